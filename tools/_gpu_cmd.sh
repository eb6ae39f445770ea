for env in "X=1" "SFMX_PREFETCH_WORKERS=4" "SFMX_PREFETCH_WORKERS=8"; do echo "== C5 $env"; env $env python tools/bench_c5.py --frames 60 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['frames_per_s'], d['ms_per_frame'], d['passes_bit_identical'], d['host_seconds'])"; done > gpurun_out/r03_m_c5.txt 2>&1
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r03_m_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03_m_tests.log
C3_FRAMES=1000 C3_PASSES=2 C5_FRAMES=60 bash tools/profile_round.sh r03 > gpurun_out/r03_m_profile.log 2>&1; echo "profile rc=$?" >> gpurun_out/r03_m_profile.log
SFMX_RANSAC_LANES=2 python tools/bench_c3.py --frames 1000 --passes 2 > gpurun_out/r03_m_c3_lanes2.json 2>/dev/null
