mkdir -p gpurun_out/r5p
timeout 1500 python -m pytest tests/test_gpu_round5.py -x -q -k "bench_line or batch_of_64" > gpurun_out/r5p/pytest.txt 2>&1; tail -15 gpurun_out/r5p/pytest.txt
