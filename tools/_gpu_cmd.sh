python -m pytest tests -x -q -m gpu -k "ba_ or bench_workload or e2e or pipeline or async_lanes or c5_end" > gpurun_out/r03_ah_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03_ah_tests.log
python tools/ab_inproc.py --reps 30 --passes 3 "join-early:SFMX_JOIN_C_EARLY=1" "join-late:" > gpurun_out/r03_ab_inproc_join.txt 2>&1
