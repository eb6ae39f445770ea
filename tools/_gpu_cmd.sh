python -m pytest tests -x -q -m gpu -k "bench_workload or e2e or pipeline or async_lanes or c5_end or cli" > gpurun_out/r03_al_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03_al_tests.log
python tools/ab_inproc.py --reps 30 --passes 3 "verify-late:SFMX_VERIFY_LATE=1" "verify-early:" > gpurun_out/r03_ab_inproc_verify.txt 2>&1
