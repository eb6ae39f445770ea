python -m pytest tests -x -q -m gpu -k "ba_ or bench_workload or e2e or async_lanes" > gpurun_out/r03_av_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03_av_tests.log
python tools/ab_inproc.py --reps 30 --passes 3 "pts2:" "pts1:SFMX_BA_PTS=1" "pts4:SFMX_BA_PTS=4" > gpurun_out/r03_ab_inproc_pts2.txt 2>&1
