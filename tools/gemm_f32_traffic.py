#!/usr/bin/env python3
"""Bytes from beyond the L2s per launch of the fp32 persistent GEMM, per launch CONFIGURATION (tile-walk group, helper pieces on /
off), at the four encoder shapes of batch 256 -- and the launch time of the same configurations.  GPU box only.

    python3 tools/gemm_f32_traffic.py time                       # HIP-event times, interleaved rounds (un-profiled)
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/traffic_fetch -- python3 tools/gemm_f32_traffic.py run
    python3 tools/gemm_f32_traffic.py summarize gpurun_out/traffic_fetch [gpurun_out/traffic_write]

`run` launches every configuration REPS times in a fixed order and writes that order to gpurun_out/traffic_labels.json; all
launches carry the same kernel name, so `summarize` pairs the profiler's dispatches of that kernel (in dispatch order) with the
labels.  Units and the gfx950 correction are those of tools/pmc_summary.py: traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB.
"""
import ctypes as C
import csv
import glob
import importlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LABELS = os.path.join(ROOT, "gpurun_out", "traffic_labels.json")
REPS = 4
M = 50432
# shape name -> (N, K, epilogue)
SHAPES = {"qkv": (2304, 768, 0), "fc1": (3072, 768, 1), "outproj": (768, 768, 2), "fc2": (768, 3072, 2)}


def configs():
    out = []
    for shape in SHAPES:
        for group in (8, 1, 2, 4, 16):
            for ws in ((1, 0) if shape in ("outproj", "fc2", "fc1") and group == 8 else (1,) if shape != "qkv" else (0,)):
                out.append((shape, group, ws))
    return out


def algorithmic_bytes(shape):
    N, K, epi = SHAPES[shape]
    return 4 * (M * K + N * K + M * N * (2 if epi == 2 else 1))


def make_launcher():
    import numpy as np
    B = importlib.import_module("vision-transformer-opencl_amd.binding")
    L = B.lib()
    rng = np.random.default_rng(0)
    ws = B.gemm_workspace()
    bufs = {}
    for shape, (N, K, epi) in SHAPES.items():
        dA = B.DeviceArray.from_numpy(rng.uniform(-1, 1, (M, K)).astype(np.float32))
        dW = B.DeviceArray.from_numpy(rng.uniform(-.05, .05, (N, K)).astype(np.float32))
        db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
        dC = B.DeviceArray((M, N))
        bufs[shape] = (dA, dW, db, dC)

    def launch(shape, group, use_ws):
        N, K, epi = SHAPES[shape]
        dA, dW, db, dC = bufs[shape]
        # tile 9 = the persistent walk whatever the auto rule would pick; residual aliases C as in the engine
        args = B.CGemmArgs(dA.ptr, K, dW.ptr, K, db.ptr, dC.ptr if epi == 2 else None, N, dC.ptr, N, M, N, K, epi, 9, group,
                           ws if use_ws else None, 0)
        B.hip_check(L.vithip_gemm_f32(None, C.byref(args)), "vithip_gemm_f32")

    return B, launch


def mode_run():
    B, launch = make_launcher()
    labels = []
    for shape, group, ws in configs():
        for _ in range(REPS):
            launch(shape, group, ws)
            labels.append(f"{shape} group_m={group} pieces={'on' if ws else 'off'}")
    B.hip_check(B.lib().vithip_device_sync(), "sync")
    os.makedirs(os.path.dirname(LABELS), exist_ok=True)
    json.dump(labels, open(LABELS, "w"))
    print(f"{len(labels)} launches")


def mode_time(rounds=3):
    from tools.gemm_probe import timed
    B, launch = make_launcher()
    res = defaultdict(list)
    for _ in range(rounds):
        for cfg in configs():
            res[cfg].append(timed(lambda: launch(*cfg), reps=4, warm=1))
    for (shape, group, ws), ms in res.items():
        N, K, _ = SHAPES[shape]
        best = min(ms)
        print(json.dumps({"shape": shape, "group_m": group, "pieces": bool(ws), "ms_min": round(best, 4),
                          "ms_median": round(sorted(ms)[len(ms) // 2], 4), "tflops": round(2.0 * M * N * K / (best * 1e-3) / 1e12, 1)}), flush=True)


def read_counter(dirname, counter):
    rows = []
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and "gemm_f32_nt_persistent_kernel" in row["Kernel_Name"]:
                    rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]


def mode_summarize(fetch_dir, write_dir=None):
    labels = json.load(open(LABELS))
    fetch = read_counter(fetch_dir, "FETCH_SIZE")
    write = read_counter(write_dir, "WRITE_SIZE") if write_dir else None
    if len(fetch) != len(labels) or (write is not None and len(write) != len(labels)):
        raise SystemExit(f"{len(labels)} labels, {len(fetch)} FETCH_SIZE rows, {len(write) if write else '-'} WRITE_SIZE rows")
    acc = defaultdict(lambda: [0.0, 0.0, 0])
    for i, lab in enumerate(labels):
        acc[lab][0] += fetch[i]
        acc[lab][1] += write[i] if write else 0.0
        acc[lab][2] += 1
    for lab, (f, w, n) in acc.items():
        shape = lab.split()[0]
        rec = {"config": lab, "fetch_MB": round(2 * f / n * 1024 / 1e6, 1)}
        if write:
            rec["write_MB"] = round(w / n * 1024 / 1e6, 1)
            rec["traffic_MB"] = round((2 * f + w) / n * 1024 / 1e6, 1)
            rec["over_algorithmic"] = round((2 * f + w) / n * 1024 / algorithmic_bytes(shape), 3)
        rec["algorithmic_MB"] = round(algorithmic_bytes(shape) / 1e6, 1)
        print(json.dumps(rec))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "time"
    if mode == "run":
        mode_run()
    elif mode == "time":
        mode_time()
    elif mode == "summarize":
        mode_summarize(*sys.argv[2:4])
    else:
        raise SystemExit(__doc__)
