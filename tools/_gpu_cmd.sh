python tools/klt_pipe_probe.py > gpurun_out/r03_n_klt_pipe.txt 2>&1
for K in 0; do for P in 0 1; do SFMX_KLT_STAMPS=1 SFMX_KLT_K=$K SFMX_KLT_PIPE=$P python tools/klt_stamps.py 2>&1 | grep -E "klt stamps mean|^T" | sed "s/^/PIPE=$P /"; done; done > gpurun_out/r03_n_stamps.txt 2>&1
