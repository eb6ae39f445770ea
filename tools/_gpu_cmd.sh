for P in 1 2 4; do SFMX_BA_PTS=$P python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ba_" > gpurun_out/r03_ao_tests_$P.log 2>&1; echo "PTS=$P rc=$?" >> gpurun_out/r03_ao_tests.log; done
SFMX_BA_REDUCE=plain python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ba_" >> gpurun_out/r03_ao_tests_plain.log 2>&1; echo "reduce plain rc=$?" >> gpurun_out/r03_ao_tests.log
for P in 1 2 4; do echo "== SFMX_BA_PTS=$P"; STAMPS_ALONE_ONLY=1 SFMX_BA_PTS=$P timeout -k 10 200 python tools/ba_wgstamps.py 2>&1 | grep -v amdgpu; done > gpurun_out/r03_ba_wgstamps5.txt
echo "== SFMX_BA_REDUCE=plain" >> gpurun_out/r03_ba_wgstamps5.txt; STAMPS_ALONE_ONLY=1 SFMX_BA_REDUCE=plain timeout -k 10 200 python tools/ba_wgstamps.py 2>&1 | grep "k_ba_reduce" >> gpurun_out/r03_ba_wgstamps5.txt
python tools/ab_inproc.py --reps 24 --passes 3 "pts4+plain:SFMX_BA_PTS=4,SFMX_BA_REDUCE=plain" "pts4:SFMX_BA_PTS=4" "pts2:SFMX_BA_PTS=2" "pts1:SFMX_BA_PTS=1" > gpurun_out/r03_ab_inproc_pts.txt 2>&1
