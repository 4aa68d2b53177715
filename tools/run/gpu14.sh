# round 4, call 14: the fold at smaller batches (statistics kernel where the LayerNorm kernel ran): img/s with and without
mkdir -p gpurun_out/r04_fold32
for b in 8 32 64 128; do
  F="--batch $b --lanes 1 --steps 40 --warmup 5 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
  timeout -k 10 200 python bench.py $F > gpurun_out/r04_fold32/sb_fold_$b.json || exit 1
  timeout -k 10 200 python bench.py $F --ln-fold -1 > gpurun_out/r04_fold32/sb_plain_$b.json || exit 1
done
python - <<'PY'
import json
for b in (8, 32, 64, 128):
    row = []
    for k in ("fold", "plain"):
        d = json.loads(open(f"gpurun_out/r04_fold32/sb_{k}_{b}.json").read().strip().splitlines()[-1])
        st = d["roofline"]["stage_ms_per_step"]
        row.append((k, round(d["value"], 1), d["ms_per_step"], st["ln"], st["qkv"], st["outproj"], st["fc1"], st["fc2"]))
    print(b, row)
PY
