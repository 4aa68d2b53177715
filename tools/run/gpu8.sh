set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -x -q > gpurun_out/r04/gputest8.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r04/gputest8.log
if [ $rc -ne 0 ]; then exit $rc; fi
OLD=$PWD/vision-transformer-opencl_amd/build/ab/libvit_mi355x_old.so
for i in 1 2; do
  VIT_HIP_LIBRARY=$OLD timeout -k 10 300 python3 tools/gemm_f32_traffic.py time > gpurun_out/r04/sgpr_time_old_$i.log 2>&1 || exit 1
  timeout -k 10 300 python3 tools/gemm_f32_traffic.py time > gpurun_out/r04/sgpr_time_new_$i.log 2>&1 || exit 1
  for w in old new; do echo "== $w $i"; grep -E '"(outproj|fc2)", "group_m": 1, "pieces": true|"(qkv|fc1)", "group_m": 4' gpurun_out/r04/sgpr_time_${w}_$i.log; done
done
for i in 1 2; do
for w in old new; do
if [ $w = old ]; then export VIT_HIP_LIBRARY=$OLD; else unset VIT_HIP_LIBRARY; fi
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs --steps 20 > gpurun_out/r04/bench8_${w}_$i.json 2>/dev/null || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/r04/bench8_${w}_$i.json'));print('$w',d['value'],d['roofline']['frac'],d['roofline']['whole_model_frac'],d['roofline']['stage_ms_per_step'])"
done; done
