mkdir -p gpurun_out/r5j
timeout 1500 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_round3.py -x -q -k "fp32 or limit or inner or batch_n1000 or reference_goldens or free_running or teacher" > gpurun_out/r5j/pytest_fp32.txt 2>&1; tail -5 gpurun_out/r5j/pytest_fp32.txt
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
python3 bench.py --gpus 1 --precision fp32 --steps 8 --warmup 2 $F > gpurun_out/r5j/c4_fp32.json 2>&1
python3 bench.py --gpus 1 --config c5 --steps 6 --warmup 2 $F > gpurun_out/r5j/c5_fp32.json 2>&1
python3 bench.py --gpus 1 --precision fp32 --steps 8 --warmup 2 $F --kernel-flags 4 > gpurun_out/r5j/c4_fp32_generic.json 2>&1
python3 bench.py --gpus 1 --config c5 --steps 6 --warmup 2 $F --kernel-flags 4 > gpurun_out/r5j/c5_fp32_generic.json 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5j/prof_c5 -- python3 /root/repo/bench.py --gpus 1 --config c5 --steps 6 --warmup 2 $F > /root/repo/gpurun_out/r5j/prof_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5j/prof_c4 -- python3 /root/repo/bench.py --gpus 1 --precision fp32 --steps 8 --warmup 2 $F > /root/repo/gpurun_out/r5j/prof_c4.log 2>&1
