python tools/ab_inproc.py --reps 30 --passes 3 "sweeps:" "tile:SFMX_SHI_MODE=tile" > gpurun_out/r03_ab_inproc_shi.txt 2>&1
