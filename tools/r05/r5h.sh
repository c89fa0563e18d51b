mkdir -p gpurun_out/r5h
timeout 1500 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_parity.py tests/test_gpu_round5.py -x -q -k "fp32 or limit or inner or batch_n1000" > gpurun_out/r5h/pytest_fp32.txt 2>&1; tail -5 gpurun_out/r5h/pytest_fp32.txt
F="--no-cpu-baseline --no-reference-precision --no-all-samples"
python3 bench.py --gpus 1 --precision fp32 --steps 8 --warmup 2 $F > gpurun_out/r5h/c4_fp32.json 2>&1
python3 bench.py --gpus 1 --config c5 --steps 6 --warmup 2 $F > gpurun_out/r5h/c5_fp32.json 2>&1
python3 bench.py --gpus 1 --precision fp32 --steps 8 --warmup 2 $F --kernel-flags 4 > gpurun_out/r5h/c4_fp32_generic.json 2>&1
for rep in 1 2; do
for v in base zpol1 zpol2 zpol3 zpol5; do
  if [ $v = base ]; then L=""; else L="FDIPT_LIB=$PWD/framedipt_amd/lib/libfdipt_hip_$v.so"; fi
  env $L python3 bench.py --gpus 1 --steps 100 --warmup 5 $F > gpurun_out/r5h/et_${v}_$rep.json 2>&1
done
done
