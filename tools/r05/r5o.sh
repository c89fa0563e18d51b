mkdir -p gpurun_out/r5o
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
for i in 1 2; do
python3 bench.py --gpus 1 --samples-per-gpu 64 --steps 24 --warmup 3 $F > gpurun_out/r5o/b64_base_$i.json 2>&1
python3 bench.py --gpus 1 --samples-per-gpu 64 --steps 24 --warmup 3 $F --kernel-flags 128 > gpurun_out/r5o/b64_rows32_$i.json 2>&1
done
