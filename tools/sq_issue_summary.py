#!/usr/bin/env python3
"""Per-kernel wave-issue breakdown from a rocprofv3 SQ counter pass (one row per kernel, averages per launch).
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \\
              SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d DIR -- python3 bench.py ...
    python tools/sq_issue_summary.py DIR [DIR2 ...] out.csv
Several DIRs (separate passes of the same command) are merged per kernel: every counter is averaged over the launches of
the pass that collected it.  Shares are of SQ_WAVE_CYCLES (CDNA4 guide, PMC slots: WAIT_ANY = parked at s_waitcnt / barrier,
WAIT_INST_ANY = issue-stalled, ACTIVE_INST_ANY = issuing; the three are disjoint and add up to about the wave-cycles)."""
import csv, glob, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short_name

*dirs, out = sys.argv[1:]
val = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
n = defaultdict(lambda: defaultdict(int))       # kernel -> counter -> launches
dur = defaultdict(float)
ndur = defaultdict(int)
for d in dirs:
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        sys.exit(f"no counter_collection.csv under {d}")
    seen = set()
    with open(files[-1], newline="") as f:
        for r in csv.DictReader(f):
            k = (short_name(r["Kernel_Name"]), int(r["Grid_Size"]))
            if k[0].startswith("__amd") or "at::native" in k[0]:
                continue
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
            if (d, r["Dispatch_Id"]) not in seen:
                seen.add((d, r["Dispatch_Id"]))
                dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                ndur[k] += 1
counters = sorted({c for k in val for c in val[k]})
rows = []
for k in val:
    avg = {c: val[k][c] / n[k][c] for c in val[k]}
    rows.append((dur[k], k, avg))
rows.sort(key=lambda r: r[0], reverse=True)
share_of = [c for c in counters if c != "SQ_WAVE_CYCLES" and c.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_BUSY_CYCLES"))]
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "launches", "avg_us"] + counters + [f"{c}_share_of_wave_cycles" for c in share_of])
    for d_, k, avg in rows:
        wc = avg.get("SQ_WAVE_CYCLES", 0.0)
        w.writerow([k[0], k[1], ndur[k], round(d_ / ndur[k] / 1e3, 1)] + [round(avg.get(c, 0.0), 1) for c in counters]
                   + [round(avg.get(c, 0.0) / wc, 4) if wc else "" for c in share_of])
for d_, k, avg in rows[:8]:
    wc = avg.get("SQ_WAVE_CYCLES", 0.0)
    parts = "  ".join(f"{c[3:]} {avg[c] / wc:5.1%}" for c in share_of if c in avg and wc)
    print(f"{d_ / ndur[k] / 1e3:9.1f} us x{ndur[k]:4d}  {k[0][:70]}\n      {parts}")
