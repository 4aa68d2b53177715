#!/bin/bash
# One parametrised command list for a GPU-box call (replaces the 23 one-shot tools/run/gpu*.sh of rounds 1-4):
#     gpurun --timeout 1200 -- 'bash tools/gpu_session.sh <step> [<step> ...]'
# Steps run in the order given, joined by `set -e` (a failed or killed step ends the call: no GPU step runs behind a dead one).
# Everything lands under gpurun_out/$VIT_ROUND (default r05).  Steps:
#   suite               python -m pytest tests -m gpu -x -q              -> gpu_suite.log
#   pytest:<tag>:<expr> python -m pytest tests -m gpu -q -k <expr>                -> pytest_<tag>.log
#   abtest:<tag>:<lib>:<expr>  the same tests against another build of the library, all failures listed -> pytest_<tag>.log
#   bench               python bench.py (the driver's default line)       -> bench_default.json
#   bench:<tag>:<args>  python bench.py <args> (',' separates arguments)  -> bench_<tag>.json
#   sq:<tag>:<args>     SQ wait / issue counters of `bench.py <args>`     -> attn_issue_<tag>.csv (tools/sq_issue_summary.py)
#   tool:<tag>:<script>[:<args>]   python tools/<script> <args>, product library     -> <tag>.log
#   probe:<tag>:<script>[:<args>]  the same against libvit_mi355x_probe.so           -> <tag>.log
#   ab:<tag>:<lib>:<script>[:<args>]  the same against another build of the library (path relative to the repo)
#   profiles            tools/collect_profiles.sh $VIT_ROUND  (kernel stats + PMC passes -> gpurun_out/profiles/<round>/)
set -eo pipefail
export TMPDIR=/tmp
R=${VIT_ROUND:-r05}
O=gpurun_out/$R
mkdir -p "$O"
QUIET="--no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
for step in "$@"; do
    IFS=: read -r kind tag a b c <<< "$step"
    case "$kind" in
    suite)
        timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > "$O/gpu_suite.log" 2>&1 || { tail -30 "$O/gpu_suite.log"; exit 1; }
        tail -1 "$O/gpu_suite.log" ;;
    pytest)
        timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -s -k "$a" > "$O/pytest_$tag.log" 2>&1 || { tail -40 "$O/pytest_$tag.log"; exit 1; }
        tail -5 "$O/pytest_$tag.log" ;;
    abtest)
        VIT_HIP_LIBRARY=$PWD/$a timeout -k 10 900 python3 -m pytest tests -m gpu -q -k "$b" > "$O/pytest_$tag.log" 2>&1 || true
        grep -E "^(FAILED|ERROR)|passed|failed" "$O/pytest_$tag.log" | tail -30 ;;
    bench)
        if [ -z "$tag" ]; then tag=default; fi
        # shellcheck disable=SC2086
        timeout -k 10 500 python3 bench.py ${a//,/ } > "$O/bench_$tag.json" 2> "$O/bench_$tag.err"
        python3 - "$O/bench_$tag.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r = d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], "ms | dominant", r["frac"], "whole", r.get("whole_model_frac"), "| stages", r.get("stage_ms_per_step"))
for o in d.get("other_configs") or []:
    print("   ", o["workload"][:40], o["value"], o["roofline"].get("whole_model_frac"), o["roofline"].get("stage_ms_per_step"))
if d.get("c_surface"): print("    c_surface", d["c_surface"]["value"], round(d["c_surface"]["value"] / d["value"], 4))
PY
        ;;
    sq)
        # shellcheck disable=SC2086
        timeout -k 10 500 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d "$O/sq_$tag" -- \
            python3 bench.py $QUIET --lanes 1 --steps 2 --warmup 1 ${a//,/ } > /dev/null 2> "$O/sq_$tag.err"
        python3 tools/sq_issue_summary.py "$O/sq_$tag" "$O/attn_issue_$tag.csv"
        rm -rf "$O/sq_$tag" ;;
    tool|probe|ab)
        lib=""
        if [ "$kind" = probe ]; then lib=$PWD/vision-transformer-opencl_amd/libvit_mi355x_probe.so; fi
        if [ "$kind" = ab ]; then lib=$PWD/$a; a=$b; b=$c; fi
        # shellcheck disable=SC2086
        VIT_HIP_LIBRARY=$lib timeout -k 10 900 python3 "tools/$a" ${b//,/ } > "$O/$tag.log" 2>&1 || { tail -30 "$O/$tag.log"; exit 1; }
        tail -25 "$O/$tag.log" ;;
    profiles)
        bash tools/collect_profiles.sh "$R" ;;
    *) echo "unknown step $step"; exit 2 ;;
    esac
done
