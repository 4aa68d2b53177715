# round 4, call 22: non-temporal policy on the residual reads (fp32 x, once per launch) of the bf16 residual epilogue vs base, interleaved
for r in 1 2 3; do
  for v in base ntr; do
    echo "== $v $r"; VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_$v.so timeout -k 10 200 python tools/gemm_bf16_time.py 2048 b16 fold | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v['ms_min'] for k,v in d['gemms'].items()})" || exit 1
  done
done
