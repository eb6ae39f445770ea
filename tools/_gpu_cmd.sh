python -m pytest tests -m gpu -x -q > gpurun_out/r03_r_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03_r_tests.log
tail -1 gpurun_out/r03_r_tests.log | grep -q "rc=0" && python tools/microbench.py > gpurun_out/r03_r_microbench.txt 2>&1 && python bench.py > gpurun_out/r03_r_bench.json 2> gpurun_out/r03_r_bench.err
