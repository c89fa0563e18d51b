mkdir -p gpurun_out/r5f
timeout 2300 python -m pytest tests -m gpu -x -q > gpurun_out/r5f/pytest_gpu.txt 2>&1; tail -8 gpurun_out/r5f/pytest_gpu.txt
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F > gpurun_out/r5f/k20.json 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F --kernel-flags 16 > gpurun_out/r5f/k20_unfolded.json 2>&1
python3 bench.py --gpus 1 --steps 100 --warmup 5 $F > gpurun_out/r5f/k100.json 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5f/prof -- python3 /root/repo/bench.py --gpus 1 --steps 100 --warmup 5 $F > /root/repo/gpurun_out/r5f/prof.log 2>&1
