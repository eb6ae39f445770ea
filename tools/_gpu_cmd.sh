python -m pytest tests -m gpu -x -q > gpurun_out/r03_final4_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03_final4_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_final4_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r03_final4_smoke.log
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round4.log 2>&1
