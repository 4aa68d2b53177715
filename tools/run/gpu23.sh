# round 4, call 23: the one-tile-per-workgroup 128x128 kernel (tile 10) against the persistent walk (auto) inside the forward, stage by stage
F="--steps 20 --warmup 3 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2 3; do
  timeout -k 10 200 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('auto  ', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
  timeout -k 10 200 python bench.py $F --gemm-tile 10 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile10', d['value'], d['ms_per_step'], d['roofline']['stage_ms_per_step'])"
done
