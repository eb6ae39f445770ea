python -m pytest tests -x -q -m gpu -k "ba_ or bench_workload or e2e or pipeline_vs or async_lanes" > gpurun_out/r03_ae_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03_ae_tests.log
timeout -k 10 400 python tools/ba_wgstamps.py > gpurun_out/r03_ba_wgstamps2.txt 2>&1
python tools/ab_inproc.py --reps 30 --passes 3 "points-global:SFMX_BA_POINTS=global" "points-lds:" > gpurun_out/r03_ab_inproc_points.txt 2>&1
