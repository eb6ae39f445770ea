python tools/ab_inproc.py --reps 30 --passes 3 "arena-malloc:SFMX_ARENA_POOL=0" "arena-pool:" > gpurun_out/r03_ab_inproc_arena.txt 2>&1
