# round 4, call 11: fp32 LayerNorm fold -- whole GPU suite, then folded vs unfolded engine at the metric batch (one process build, interleaved)
mkdir -p gpurun_out/r04_fold32
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04_fold32/tests.log 2>&1; rc=$?
tail -25 gpurun_out/r04_fold32/tests.log
[ $rc -eq 0 ] || exit $rc
F="--steps 20 --warmup 3 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2 3; do
  timeout -k 10 200 python bench.py $F > gpurun_out/r04_fold32/fold_$r.json || exit 1
  timeout -k 10 200 python bench.py $F --ln-fold -1 > gpurun_out/r04_fold32/plain_$r.json || exit 1
done
python - <<'PY'
import json, glob
for k in ("fold", "plain"):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/r04_fold32/{k}_*.json"))]
    print(k, [round(x["value"], 1) for x in v], [round(x["ms_per_step"], 3) for x in v], v[-1]["roofline"]["stage_ms_per_step"], v[-1].get("golden"))
PY
