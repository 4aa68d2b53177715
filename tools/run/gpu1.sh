set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest1.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r04/gputest1.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r04/bench1.json 2> gpurun_out/r04/bench1.err; rc=$?
echo "bench rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python3 tools/gemm_f32_traffic.py time > gpurun_out/r04/traffic_time.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/traffic_fetch -- python3 tools/gemm_f32_traffic.py run > gpurun_out/r04/traffic_run1.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/traffic_write -- python3 tools/gemm_f32_traffic.py run > gpurun_out/r04/traffic_run2.log 2>&1 || exit 1
python3 tools/gemm_f32_traffic.py summarize gpurun_out/traffic_fetch gpurun_out/traffic_write > gpurun_out/r04/traffic.log 2>&1
tail -30 gpurun_out/r04/traffic.log
