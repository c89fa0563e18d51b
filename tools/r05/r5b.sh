set -x
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/r5b/pytest_r5.txt 2>&1; tail -5 gpurun_out/r5b/pytest_r5.txt
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F > gpurun_out/r5b/k20_graph8.json 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F --eager > gpurun_out/r5b/k20_eager.json 2>&1
python3 bench.py --gpus 1 $F > gpurun_out/r5b/full_graph8.json 2>&1
python3 bench.py --gpus 1 $F --graph-chunk 1 > gpurun_out/r5b/full_graph1.json 2>&1
python3 bench.py --gpus 1 $F --graph-chunk 32 > gpurun_out/r5b/full_graph32.json 2>&1
python3 bench.py --gpus 1 $F --eager > gpurun_out/r5b/full_eager.json 2>&1
# a slow host: the bench pinned to one core shared with 3 spinning processes
for i in 1 2 3; do (taskset -c 5 timeout 60 python3 -c "while True: pass" &) ; done
taskset -c 5 python3 bench.py --gpus 1 --steps 20 --warmup 5 $F > gpurun_out/r5b/k20_graph8_busy.json 2>&1
taskset -c 5 python3 bench.py --gpus 1 --steps 20 --warmup 5 $F --eager > gpurun_out/r5b/k20_eager_busy.json 2>&1
sleep 30
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5b/drv_cmd.json 2> gpurun_out/r5b/drv_cmd.err
