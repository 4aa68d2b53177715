# round 4, call 16: helper-piece instantiations without the per-step accumulator waits (unpark ends with a compiler-visible vmcnt(0)):
# op tests, launch times new / base interleaved, forward A/B
mkdir -p gpurun_out/r04_unpark
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > gpurun_out/r04_unpark/tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_unpark/tests.log
[ $rc -eq 0 ] || exit $rc
export VIT_TOOL_ARMS="fc1 fold,outproj stats,fc2 stats,qkv fold"
for r in 1 2; do
  echo "== new $r"; timeout -k 10 200 python tools/gemm_f32_fold.py 4 || exit 1
  echo "== base $r"; VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_base.so timeout -k 10 200 python tools/gemm_f32_fold.py 4 || exit 1
done
unset VIT_TOOL_ARMS
F="--steps 20 --warmup 3 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2 3; do
  timeout -k 10 200 python bench.py $F > gpurun_out/r04_unpark/new_$r.json || exit 1
  VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_base.so timeout -k 10 200 python bench.py $F > gpurun_out/r04_unpark/base_$r.json || exit 1
done
python - <<'PY'
import json, glob
for k in ("new", "base"):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/r04_unpark/{k}_*.json"))]
    print(k, [round(x["value"], 1) for x in v], [round(x["ms_per_step"], 3) for x in v], v[-1]["roofline"]["stage_ms_per_step"])
PY
