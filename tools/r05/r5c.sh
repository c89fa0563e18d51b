mkdir -p gpurun_out/r5c
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
which rocm-smi amd-smi
( while true; do rocm-smi --showuse --json > /dev/null 2>&1; done ) &
SMI=$!
sleep 2
for i in 1 2 3; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F > gpurun_out/r5c/smi_graph_$i.json 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F --eager > gpurun_out/r5c/smi_eager_$i.json 2>&1
done
kill $SMI
sleep 3
for i in 1 2 3; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F > gpurun_out/r5c/quiet_graph_$i.json 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F --eager > gpurun_out/r5c/quiet_eager_$i.json 2>&1
done
( while true; do cat /sys/class/drm/card*/device/gpu_busy_percent > /dev/null 2>&1; done ) &
SMI=$!
sleep 1
for i in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F > gpurun_out/r5c/sysfs_graph_$i.json 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 $F --eager > gpurun_out/r5c/sysfs_eager_$i.json 2>&1
done
kill $SMI
