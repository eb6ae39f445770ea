python tools/ab_inproc.py --reps 30 --passes 3 "tile3:" "tile2:SFMX_SHI_MODE=tile,2;" "tile3+pf3:SFMX_PREFETCH_WORKERS=3" "sweeps:SFMX_SHI_MODE=sweeps" > gpurun_out/r03_ab_inproc_shi3.txt 2>&1
