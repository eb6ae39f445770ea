python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "corner_schedules or corner_pick or shi_" > gpurun_out/r03_be_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03_be_tests.log
python tools/ab_inproc.py --reps 30 --passes 3 "sweeps:SFMX_SHI_MODE=sweeps" "tile:SFMX_SHI_MODE=tile" > gpurun_out/r03_ab_inproc_shi2.txt 2>&1
