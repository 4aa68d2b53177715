# round 4, call 20: non-temporal stores for the fp32 QKV / fc1 outputs: new vs base (libvit_mi355x_nt1.so: same fp32 kernels without), interleaved
mkdir -p gpurun_out/r04_nt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > gpurun_out/r04_nt/tests3.log 2>&1; rc=$?; tail -3 gpurun_out/r04_nt/tests3.log; [ $rc -eq 0 ] || exit $rc
export VIT_TOOL_ARMS="qkv fold,fc1 fold"
for r in 1 2; do
  echo "== nt $r"; timeout -k 10 200 python tools/gemm_f32_fold.py 4 || exit 1
  echo "== base $r"; VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_nt1.so timeout -k 10 200 python tools/gemm_f32_fold.py 4 || exit 1
done
unset VIT_TOOL_ARMS
F="--steps 20 --warmup 3 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2 3; do
  timeout -k 10 200 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt  ', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
  VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_nt1.so timeout -k 10 200 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
done
