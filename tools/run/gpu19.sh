# round 4, call 19: non-temporal stores also for the fp32 / bf16 16-byte stores of the residual epilogue: new vs the bf16-outputs-only build
mkdir -p gpurun_out/r04_nt
timeout -k 10 400 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -k "gemm" > gpurun_out/r04_nt/tests2.log 2>&1; rc=$?; tail -3 gpurun_out/r04_nt/tests2.log; [ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  echo "== all-nt $r"; timeout -k 10 200 python tools/gemm_bf16_time.py 2048 b16 fold || exit 1
  echo "== bf16-out-nt $r"; VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_nt1.so timeout -k 10 200 python tools/gemm_bf16_time.py 2048 b16 fold || exit 1
done
F="--config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2; do
  timeout -k 10 300 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('all-nt     ', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
  VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_nt1.so timeout -k 10 300 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16-out-nt', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
done
