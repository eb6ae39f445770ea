python tools/ab_inproc.py --reps 30 --passes 3 "replay1:" "replay2:SFMX_REPLAY_THREADS=2" "replay4:SFMX_REPLAY_THREADS=4" > gpurun_out/r03_ab_inproc_replay.txt 2>&1
