# round 4, call 9: LayerNorm in the tail of the fp32 residual GEMMs -- parity, then old library vs new on one box (interleaved)
set -e
mkdir -p gpurun_out/r04_lntail
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -x -q > gpurun_out/r04_lntail/tests.log 2>&1 || { tail -30 gpurun_out/r04_lntail/tests.log; exit 1; }
tail -3 gpurun_out/r04_lntail/tests.log
F="--steps 20 --warmup 3 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2 3; do
  timeout -k 10 200 python bench.py $F > gpurun_out/r04_lntail/new_$r.json
  VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_base.so timeout -k 10 200 python bench.py $F > gpurun_out/r04_lntail/base_$r.json
done
python - <<'PY'
import json, glob
for k in ("new", "base"):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/r04_lntail/{k}_*.json"))]
    print(k, [round(x["value"], 1) for x in v], [round(x["ms_per_step"], 3) for x in v])
PY
