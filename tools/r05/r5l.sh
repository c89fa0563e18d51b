mkdir -p gpurun_out/r5l
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
timeout 900 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_parity.py -x -q -k "fp32" > gpurun_out/r5l/pytest_fp32.txt 2>&1; tail -2 gpurun_out/r5l/pytest_fp32.txt
for i in 1 2; do
python3 bench.py --gpus 1 --precision fp32 --steps 8 --warmup 2 $F > gpurun_out/r5l/c4_fp32_$i.json 2>&1
python3 bench.py --gpus 1 --config c5 --steps 6 --warmup 2 $F > gpurun_out/r5l/c5_fp32_$i.json 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5l/prof_c4 -- python3 /root/repo/bench.py --gpus 1 --precision fp32 --steps 8 --warmup 2 $F > /root/repo/gpurun_out/r5l/prof_c4.log 2>&1
