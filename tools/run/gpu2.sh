set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "gemm" > gpurun_out/r04/gputest2.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r04/gputest2.log
if [ $rc -ne 0 ]; then exit $rc; fi
VIT_TOOL_VARIANTS=2,3 timeout -k 10 300 python3 tools/gemm_bf16_time.py 2048 b16 > gpurun_out/r04/swp_ab_b16.log 2>&1 || exit 1
VIT_TOOL_VARIANTS=2,3 timeout -k 10 300 python3 tools/gemm_bf16_time.py 2048 b16 fold > gpurun_out/r04/swp_ab_b16_fold.log 2>&1 || exit 1
VIT_TOOL_VARIANTS=2,3 timeout -k 10 300 python3 tools/gemm_bf16_time.py 1024 l16_384 fold > gpurun_out/r04/swp_ab_l16_fold.log 2>&1 || exit 1
VIT_TOOL_DATA=zeros VIT_TOOL_VARIANTS=2,3 timeout -k 10 300 python3 tools/gemm_bf16_time.py 2048 b16 > gpurun_out/r04/swp_ab_b16_zeros.log 2>&1 || exit 1
cat gpurun_out/r04/swp_ab_*.log
