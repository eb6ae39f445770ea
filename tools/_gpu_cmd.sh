python -m pytest tests -m gpu -x -q > gpurun_out/r03_final3_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03_final3_tests.log
python tools/microbench.py 2>&1 | grep "ba step\|ba build (points" > gpurun_out/r03_final3_microbench_ba.txt
