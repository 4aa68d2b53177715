set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "gemm" > gpurun_out/r04/gputest4.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r04/gputest4.log
if [ $rc -ne 0 ]; then exit $rc; fi
VIT_TOOL_VARIANTS=2,3,4 timeout -k 10 300 python3 tools/gemm_bf16_time.py 2048 b16 > gpurun_out/r04/swp2_ab_b16.log 2>&1 || exit 1
VIT_TOOL_DATA=zeros VIT_TOOL_VARIANTS=2,3,4 timeout -k 10 300 python3 tools/gemm_bf16_time.py 2048 b16 > gpurun_out/r04/swp2_ab_b16_zeros.log 2>&1 || exit 1
VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_probe.so VIT_TOOL_DATA=zeros timeout -k 10 300 python3 tools/gemm_bf16_probe.py 2048 probe > gpurun_out/r04/swp2_probe_zeros.log 2>&1 || exit 1
cat gpurun_out/r04/swp2_*.log
