set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for lanes in 1 2 1 2; do
  timeout -k 10 300 python3 bench.py --config 2 --lanes $lanes --steps 5 --warmup 2 --no-cpu-baseline --no-c-surface --no-clock-probe > gpurun_out/r04/lanes_b16_$lanes.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/r04/lanes_b16_$lanes.json'));print('b16 bf16 lanes',$lanes,d['value'],d['ms_per_step'])"
done
for lanes in 1 2; do
  timeout -k 10 300 python3 bench.py --config 4 --lanes $lanes --steps 3 --warmup 1 --no-cpu-baseline --no-c-surface --no-clock-probe > gpurun_out/r04/lanes_l16_$lanes.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('gpurun_out/r04/lanes_l16_$lanes.json'));print('l16 bf16 lanes',$lanes,d['value'],d['ms_per_step'])"
done
