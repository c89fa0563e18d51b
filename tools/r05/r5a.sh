set -x
mkdir -p gpurun_out/r5a
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5a/drv_$i.json 2> gpurun_out/r5a/drv_$i.err; done
python3 bench.py --gpus 1 --no-cpu-baseline --no-reference-precision --no-all-samples > gpurun_out/r5a/full500.json 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-precision --no-all-samples > gpurun_out/r5a/k20_only.json 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d /root/repo/gpurun_out/r5a/trace -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-precision --no-all-samples > /root/repo/gpurun_out/r5a/trace.log 2>&1
ls -la /root/repo/gpurun_out/r5a/trace/*/ | head
nproc; lscpu | head -20
