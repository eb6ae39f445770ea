python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "matrix_core or sum_schedules or loop_variants" > gpurun_out/r03_aa_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03_aa_tests.log
