mkdir -p gpurun_out/r5m
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5m/drv_1.json 2> gpurun_out/r5m/drv_1.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5m/drv_2.json 2> gpurun_out/r5m/drv_2.err
timeout 2300 python -m pytest tests -m gpu -x -q > gpurun_out/r5m/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r5m/pytest_gpu.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5m/drv_3.json 2> gpurun_out/r5m/drv_3.err
