mkdir -p gpurun_out/r5q
timeout 2300 python -m pytest tests -m gpu -x -q > gpurun_out/r5q/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r5q/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5q/smoke.txt 2>&1; tail -1 gpurun_out/r5q/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5q/drv.json 2> gpurun_out/r5q/drv.err; head -c 400 gpurun_out/r5q/drv.json
