set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite_0.log 2>&1; tail -2 $O/gpu_suite_0.log
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
B="--no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs --lanes 1 --steps 2 --warmup 1"
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $O/sq_l16 -- python3 bench.py $B --config 4 > /dev/null 2> $O/sq_l16.err
python3 tools/sq_issue_summary.py $O/sq_l16 $O/attn_issue_bf16_l16_384.csv
timeout -k 10 400 rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $O/sq_b16 -- python3 bench.py $B --config 2 > /dev/null 2> $O/sq_b16.err
python3 tools/sq_issue_summary.py $O/sq_b16 $O/attn_issue_bf16.csv
timeout -k 10 300 python3 bench.py > $O/bench_0.json 2> $O/bench_0.err; tail -c 1500 $O/bench_0.json
rm -rf $O/sq_l16 $O/sq_b16
