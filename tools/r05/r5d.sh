mkdir -p gpurun_out/r5d
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5d/quiet -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $F --eager > $R/gpurun_out/r5d/quiet.log 2>&1
( while true; do cat /sys/class/drm/card*/device/gpu_busy_percent > /dev/null 2>&1; done ) &
SMI=$!
sleep 1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5d/poll -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $F --eager > $R/gpurun_out/r5d/poll.log 2>&1
kill $SMI
sleep 1
# a slow poller like a harness would run: one read every 0.5 s
( while true; do cat /sys/class/drm/card*/device/gpu_busy_percent > /dev/null 2>&1; sleep 0.5; done ) &
SMI=$!
cd $R
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 $F > gpurun_out/r5d/slowpoll_graph_$i.json 2>&1; done
kill $SMI
ls /sys/class/drm/; cat /sys/class/drm/card*/device/gpu_busy_percent
