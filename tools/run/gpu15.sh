# round 4, call 15: row values preloaded in the one-tile kernels: op + forward tests, then the small-batch sweep again
mkdir -p gpurun_out/r04_fold32
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -q -x > gpurun_out/r04_fold32/tests3.log 2>&1; rc=$?
tail -5 gpurun_out/r04_fold32/tests3.log
[ $rc -eq 0 ] || exit $rc
bash tools/run/gpu14.sh
F="--batch 1 --lanes 1 --no-stage-brackets --steps 200 --warmup 20 --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
for r in 1 2; do
  timeout -k 10 200 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1 fold', d['value'], d['ms_per_step'])"
  timeout -k 10 200 python bench.py $F --ln-fold -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1 plain', d['value'], d['ms_per_step'])"
done
