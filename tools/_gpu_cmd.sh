python -m pytest tests -m gpu -x -q > gpurun_out/r03_final5_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03_final5_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_final5_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03_final5_smoke.log
python bench.py > gpurun_out/r03_final5_bench.json 2> gpurun_out/r03_final5_bench.err
