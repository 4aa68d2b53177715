# round 4, call 17: does replaying the forward as one hipGraph move the metric batch?  (direct launches vs --graph, interleaved)
F="--steps 20 --warmup 3 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs --no-stage-brackets"
for r in 1 2 3; do
  timeout -k 10 200 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('direct', d['value'], d['ms_per_step'])"
  timeout -k 10 200 python bench.py $F --graph | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph ', d['value'], d['ms_per_step'])"
done
