# round 4, call 18: non-temporal stores for the bf16 outputs of the ping-pong GEMM (QKV, fc1): new vs base library, interleaved
mkdir -p gpurun_out/r04_nt
timeout -k 10 400 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -k "gemm" > gpurun_out/r04_nt/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_nt/tests.log; [ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  echo "== new $r"; timeout -k 10 200 python tools/gemm_bf16_time.py 2048 b16 fold || exit 1
  echo "== base $r"; VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_base.so timeout -k 10 200 python tools/gemm_bf16_time.py 2048 b16 fold || exit 1
done
F="--config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2; do
  timeout -k 10 300 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
  VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_base.so timeout -k 10 300 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
done
