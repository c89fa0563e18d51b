mkdir -p gpurun_out/r5n
timeout 2300 python -m pytest tests -m gpu -x -q > gpurun_out/r5n/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r5n/pytest_gpu.txt
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
for i in 1 2 3; do
python3 bench.py --gpus 1 --steps 100 --warmup 5 $F > gpurun_out/r5n/k100_ticket_$i.json 2>&1
python3 bench.py --gpus 1 --steps 100 --warmup 5 $F --kernel-flags 16 > gpurun_out/r5n/k100_unfolded_$i.json 2>&1
done
python tools/soak_step_graph.py 2 300 300 8 > gpurun_out/r5n/soak.txt 2>&1; tail -2 gpurun_out/r5n/soak.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5n/prof -- python3 /root/repo/bench.py --gpus 1 --steps 100 --warmup 5 $F > /root/repo/gpurun_out/r5n/prof.log 2>&1
