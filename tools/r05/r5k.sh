mkdir -p gpurun_out/r5k
FDIPT_LIB=$PWD/framedipt_amd/lib/libfdipt_hip_zpol2.so timeout 600 python -m pytest tests/test_gpu_sizes.py -x -q -k "fp16_at_size" > gpurun_out/r5k/pytest_zpol2.txt 2>&1; tail -4 gpurun_out/r5k/pytest_zpol2.txt
timeout 2300 python -m pytest tests -m gpu -x -q > gpurun_out/r5k/pytest_gpu.txt 2>&1; tail -6 gpurun_out/r5k/pytest_gpu.txt
