SFMX_RANSAC_TICKS=1 timeout -k 10 300 python tools/ransac_hyp_probe.py > gpurun_out/r03_w_hyp_probe.txt 2> gpurun_out/r03_w_hyp_ticks.txt
