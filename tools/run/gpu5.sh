set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
export VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_probe.so
timeout -k 10 300 python3 tools/gemm_bf16_group.py 8,1,2,4,16 > gpurun_out/r04/group16_time.log 2>&1 || { tail -5 gpurun_out/r04/group16_time.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/traffic16_fetch -- python3 tools/gemm_bf16_group.py 8,1,2,4,16 run > gpurun_out/r04/group16_run.log 2>&1 || { tail -5 gpurun_out/r04/group16_run.log; exit 1; }
python3 tools/gemm_bf16_group.py summarize gpurun_out/traffic16_fetch > gpurun_out/r04/group16_traffic.log 2>&1
cat gpurun_out/r04/group16_time.log gpurun_out/r04/group16_traffic.log
