# round 4, call 13: whole GPU suite on the fold build; one-image latency with and without the fold
mkdir -p gpurun_out/r04_fold32
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04_fold32/tests_all.log 2>&1; rc=$?
tail -8 gpurun_out/r04_fold32/tests_all.log
[ $rc -eq 0 ] || exit $rc
F="--batch 1 --lanes 1 --no-stage-brackets --steps 200 --warmup 20 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2; do
  timeout -k 10 200 python bench.py $F > gpurun_out/r04_fold32/b1_fold_$r.json || exit 1
  timeout -k 10 200 python bench.py $F --ln-fold -1 > gpurun_out/r04_fold32/b1_plain_$r.json || exit 1
done
python - <<'PY'
import json, glob
for k in ("b1_fold", "b1_plain"):
    v = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"gpurun_out/r04_fold32/{k}_*.json"))]
    print(k, [round(x["value"], 1) for x in v], [round(x["ms_per_step"], 4) for x in v])
PY
