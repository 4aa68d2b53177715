# round 4, call 10: LayerNorm tail without the cache invalidate: parity, then per-launch times (fence build, no-fence build, base build)
set -e
mkdir -p gpurun_out/r04_lntail
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "layernorm_tail" > gpurun_out/r04_lntail/tests2.log 2>&1 || { tail -30 gpurun_out/r04_lntail/tests2.log; exit 1; }
tail -2 gpurun_out/r04_lntail/tests2.log
echo "== no fence"; timeout -k 10 200 python tools/gemm_f32_lntail.py 4 | tee gpurun_out/r04_lntail/lntail_nofence.jsonl
echo "== fence"; VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_fence.so timeout -k 10 200 python tools/gemm_f32_lntail.py 4 | tee gpurun_out/r04_lntail/lntail_fence.jsonl
echo "== base"; VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_base.so timeout -k 10 200 python tools/gemm_f32_lntail.py 4 | tee gpurun_out/r04_lntail/lntail_base.jsonl
