#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01/hbm_traffic_pmc_f32 [batch=256]

Units and correction (MI355X_MICROARCH.md, HBM / rocprofv3 section; checked here on LayerNorm, whose
traffic is known exactly: 151,296 KB read, 75,689 "KB" counted): both counters are in KiB; on gfx950
FETCH_SIZE counts half of the streamed read bytes, WRITE_SIZE is exact.  So

    traffic_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

Writes <out>.csv (one row per kernel name x grid size) and <out>.json ({short kernel key: bytes}),
which bench.py reads for roofline.traffic.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def read_counter(dirname, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirname}")
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                key = (row["Kernel_Name"], int(row["Grid_Size"]))
                acc[key][0] += float(row["Counter_Value"])
                acc[key][1] += 1
    return acc


def short_name(name):
    """'void (anonymous namespace)::gemm_f32_nt_kernel<128, ...>(vitgemm::GemmParams)' -> 'gemm_f32_nt_kernel<128, ...>'"""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("vitgemm::", "").replace("vitattn::", "")
    cut = name.rfind("(")
    if cut > 0 and name.endswith(")"):
        name = name[:cut]
    return name.strip()


EPI = {"0": "EPI_BIAS", "1": "EPI_BIAS_GELU", "2": "EPI_BIAS_RESIDUAL", "3": "EPI_BIAS_LN", "4": "EPI_BIAS_GELU_LN", "5": "EPI_RESIDUAL_STATS"}
EPI16 = {"0": "BF16", "1": "BF16_GELU", "2": "F32_RESIDUAL", "3": "F32_EMBED"}


def bench_key(short):
    """Kernel name in bench.py's vocabulary (its STAGE_KERNEL table) for a demangled instantiation."""
    m = re.match(r"(\w+)<(.*)>$", short)
    if not m:
        return short
    base, targs = m.group(1), [a.strip() for a in m.group(2).split(",")]
    if base == "gemm_f32_nt_persistent_kernel":
        return f"{base}<{EPI.get(targs[4], targs[4])}>"
    if base == "gemm_f32_nt_kernel":
        return f"{base}<A_PATCHES>" if targs[5] == "1" else f"{base}<{EPI.get(targs[4], targs[4])}>"
    if base in ("gemm_bf16_nt_kernel", "gemm_bf16_pp_kernel"):
        return f"{base}<{EPI16.get(targs[0], targs[0])}>"
    return base


def main():
    if len(sys.argv) not in (4, 5):
        raise SystemExit(__doc__)
    batch = int(sys.argv[4]) if len(sys.argv) == 5 else 256
    fetch = read_counter(sys.argv[1], "FETCH_SIZE")
    write = read_counter(sys.argv[2], "WRITE_SIZE")
    rows = []
    for key in sorted(set(fetch) | set(write)):
        name, grid = key
        if name.startswith("__amd_rocclr") or "at::native" in name:
            continue
        fs, fn = fetch.get(key, (0.0, 0))
        ws, wn = write.get(key, (0.0, 0))
        f_avg = fs / fn if fn else 0.0
        w_avg = ws / wn if wn else 0.0
        rows.append((short_name(name), grid, max(fn, wn), f_avg, w_avg, (2.0 * f_avg + w_avg) * 1024.0))
    rows.sort(key=lambda r: -r[5] * r[2])
    out = sys.argv[3]
    with open(out + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_size", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "traffic_MB_per_launch"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3]), round(r[4]), round(r[5] / 1e6, 1)])
    # launch-weighted mean per kernel name over its large grids (the tiny head GEMM shares a template with QKV)
    by_name = defaultdict(lambda: [0.0, 0])
    for r in rows:
        if r[5] < 50e6:
            continue
        by_name[bench_key(r[0])][0] += r[5] * r[2]
        by_name[bench_key(r[0])][1] += r[2]
    import datetime
    import os
    import subprocess
    dev = os.environ.get("VIT_DEVICE")   # tools/collect_profiles.sh: name and arch as the library's vithip_get_device_info reports them
    if not dev:
        try:   # provenance: bench.py marks these figures as replayed (roofline.replayed_from)
            txt = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=60).stdout
            names = [ln.split(":", 1)[1].strip() for ln in txt.splitlines() if "Marketing Name" in ln]
            dev = next((n for n in names if "Instinct" in n or "Radeon" in n or "MI3" in n), None)
        except Exception:
            dev = None
    with open(out + ".json", "w") as f:
        json.dump({"unit": "bytes per launch", "batch": batch, "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
                   "device": dev, "commit": os.environ.get("VIT_COMMIT"),
                   "collected": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"),
                   "kernels": {k: round(v[0] / v[1]) for k, v in by_name.items()}}, f, indent=1)
    for r in rows[:12]:
        print(f"{r[5] / 1e6:9.1f} MB/launch  x{r[2]:4d}  {r[0]}")


if __name__ == "__main__":
    main()
