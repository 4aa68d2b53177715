#!/bin/bash
# Re-creates profiles/<round>/ on a GPU box:  gpurun -- 'bash tools/collect_profiles.sh r01'
# One rocprofv3 command per file; PMC passes are separate runs with --kernel-trace only (no --stats, no
# sys/hip traces); the program comes directly after `--` (no env/bash hop).  Results land in
# gpurun_out/profiles/<round>/ -- copy them into profiles/<round>/ and commit.
set -eo pipefail
R=${1:-r05}
OUT=gpurun_out/profiles/$R
W=gpurun_out/prof_work
rm -rf "$W" "$OUT"; mkdir -p "$W" "$OUT"
export TMPDIR=/tmp
# the GPU box has no .git: `git rev-parse --short HEAD > .commit_id` before the gpurun call (the file is git-ignored and travels with the snapshot)
export VIT_COMMIT=${VIT_COMMIT:-$(cat .commit_id 2>/dev/null || true)}
export VIT_DEVICE=$(python3 -c "import importlib,sys; sys.path.insert(0,'.'); b=importlib.import_module('vision-transformer-opencl_amd.binding'); i=b.device_info(0); print((i['name'] or 'MI355X pool box'), '(' + i['arch'] + ',', i['compute_units'], 'CUs)')" 2>/dev/null)

stats() {  # name, bench args...
    local name=$1; shift
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$W/$name" -- \
        python3 bench.py --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs "$@" > "$OUT/bench_$name.json" 2> "$W/$name.err"
    cp "$(find "$W/$name" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_$name.csv"
    echo "stats $name done"
}
pmc() {  # name, counter, bench args...
    local name=$1 ctr=$2; shift 2
    # shellcheck disable=SC2086  ($ctr may hold several counter names)
    timeout -k 10 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$W/$name" -- \
        python3 bench.py --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs --lanes 1 --steps 2 --warmup 1 "$@" > /dev/null 2> "$W/$name.err"
    echo "pmc $name done"
}

stats f32_lanes1 --lanes 1
stats f32_default
pmc f32_fetch FETCH_SIZE
pmc f32_write WRITE_SIZE
python3 tools/pmc_summary.py "$W/f32_fetch" "$W/f32_write" "$OUT/hbm_traffic_pmc_f32" 256
pmc f32_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16"
python3 tools/mfma_summary.py "$W/f32_mfma" "$OUT/mfma_util_f32.csv"
if [ "$2" != "f32only" ]; then
    stats bf16_batch2048 --dtype bf16 --batch 2048 --steps 5 --warmup 2
    pmc bf16_fetch FETCH_SIZE --dtype bf16 --batch 2048
    pmc bf16_write WRITE_SIZE --dtype bf16 --batch 2048
    python3 tools/pmc_summary.py "$W/bf16_fetch" "$W/bf16_write" "$OUT/hbm_traffic_pmc_bf16" 2048
    pmc bf16_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16" --dtype bf16 --batch 2048
    python3 tools/mfma_summary.py "$W/bf16_mfma" "$OUT/mfma_util_bf16.csv"
    # BASELINE.json configs[4]: ViT-L/16-384, batch 1024, bf16: unprofiled line, kernel stats, HBM traffic and matrix-pipe passes
    L16="--model l16_384 --dtype bf16 --batch 1024"
    timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-c-surface --no-clock-probe --no-other-configs $L16 --steps 3 --warmup 1 \
        > "$OUT/bench_bf16_l16_384_batch1024.json" 2> "$W/l16.err"
    stats bf16_l16_384_batch1024 $L16 --steps 3 --warmup 1
    pmc l16_fetch FETCH_SIZE $L16
    pmc l16_write WRITE_SIZE $L16
    python3 tools/pmc_summary.py "$W/l16_fetch" "$W/l16_write" "$OUT/hbm_traffic_pmc_bf16_l16_384" 1024
    pmc l16_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16" $L16
    python3 tools/mfma_summary.py "$W/l16_mfma" "$OUT/mfma_util_bf16_l16_384.csv"
fi
# the unprofiled headline run, with the CPU baseline and the parity block
timeout -k 10 500 python3 bench.py > "$OUT/bench_f32_default_unprofiled.json" 2> "$W/unprofiled.err"
echo "collect_profiles: done -> $OUT"
