python tools/ab_inproc.py --reps 30 --passes 3 "pad0:" "pad2560:SFMX_KLT_LDS_PAD=2560" "pad9000:SFMX_KLT_LDS_PAD=9000" > gpurun_out/r03_ab_inproc_kltlds.txt 2>&1
