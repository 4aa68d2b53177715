#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from a rocprofv3 PMC pass.
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 \\
              --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --lanes 1 --steps 2 --warmup 1 ...
    python tools/mfma_summary.py DIR out.csv
MFMA busy % = sum over SIMDs of SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8
(rocprofv3 reports the sum over the 8 XCDs) -- the formula of the derived counter MfmaUtil.  MOPS counters are in units of
512 flops (SQ_INSTS_VALU_MFMA_MOPS_*): matrix TFLOP/s = MOPS * 512 / duration."""
import csv, glob, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short_name
d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
dur = defaultdict(float)
files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
with open(files[-1], newline="") as f:
    seen = set()
    for r in csv.DictReader(f):
        k = (short_name(r["Kernel_Name"]), int(r["Grid_Size"]))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k))
            cnt[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0 or k[0].startswith("__amd") or "at::native" in k[0]:
        continue
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    util = 100.0 * busy / (gui / 8.0 * 1024.0)
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    tflops = mops * 512.0 / (dur[k] * 1e-9) / 1e12 if dur[k] else 0.0
    clock = gui / 8.0 / (dur[k] * 1e-9) / 1e9 if dur[k] else 0.0
    rows.append((dur[k], k[0], k[1], cnt[k], dur[k] / cnt[k] / 1e3, util, tflops, clock))
rows.sort(reverse=True)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "launches", "avg_us", "mfma_busy_percent", "mfma_tflops_from_mops", "clock_ghz_from_gui_active"])
    for r in rows:
        w.writerow([r[1], r[2], r[3], round(r[4], 1), round(r[5], 1), round(r[6], 1), round(r[7], 2)])
for r in rows[:10]:
    print(f"{r[5]:5.1f} % MFMA busy  {r[6]:7.1f} TF (MOPS)  {r[7]:.2f} GHz  {r[4]:9.1f} us x{r[3]:4d}  {r[1]}")
