# round 4, call 21: non-temporal policy on the LDS-DMA loads of the bf16 GEMM: activations (x) or weights (w) vs base, interleaved
for r in 1 2; do
  for v in base ntx ntw; do
    echo "== $v $r"; VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_$v.so timeout -k 10 200 python tools/gemm_bf16_time.py 2048 b16 fold | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v['ms_min'] for k,v in d['gemms'].items()})" || exit 1
  done
done
