SFMX_TRACE_PHASES=1 timeout -k 10 300 python tools/pass_times.py > gpurun_out/r03_z_pass_times.txt 2>&1
