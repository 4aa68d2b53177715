# round 4, call 12: fp32 LayerNorm fold with the statistics in the residual epilogue -- op and forward tests, launch times, forward A/B
mkdir -p gpurun_out/r04_fold32
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -q -x > gpurun_out/r04_fold32/tests2.log 2>&1; rc=$?
tail -25 gpurun_out/r04_fold32/tests2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/gemm_f32_fold.py 4 | tee gpurun_out/r04_fold32/launch_times.jsonl || exit 1
F="--steps 20 --warmup 3 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2 3; do
  timeout -k 10 200 python bench.py $F > gpurun_out/r04_fold32/fold2_$r.json || exit 1
  timeout -k 10 200 python bench.py $F --ln-fold -1 > gpurun_out/r04_fold32/plain2_$r.json || exit 1
done
python - <<'PY'
import json, glob
for k in ("fold2", "plain2"):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/r04_fold32/{k}_*.json"))]
    print(k, [round(x["value"], 1) for x in v], [round(x["ms_per_step"], 3) for x in v], v[-1]["roofline"]["stage_ms_per_step"], v[-1].get("golden"))
PY
