#!/usr/bin/env python3
"""bf16 ping-pong GEMM at the ViT-B/16 batch-2048 shapes for several L2 groupings of its tile walk (probe build:
vithip_gemm_bf16_set_group), interleaved rounds in one process: does the launch time follow the bytes fetched from beyond the L2s?
GPU box only.

    VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_probe.so python3 tools/gemm_bf16_group.py [groups, e.g. 8,1,2,4,16]
    ... under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR --` with argv[2] = run: one launch order, labels in
    gpurun_out/traffic16_labels.json (summarised by tools/gemm_f32_traffic.py-style pairing: `summarize DIR`)
"""
import ctypes as C, csv, glob, importlib, json, os, sys
from collections import defaultdict
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LABELS = os.path.join(ROOT, "gpurun_out", "traffic16_labels.json")
M = 2048 * 197
SHAPES = {"qkv": (2304, 768, 0), "fc1": (3072, 768, 1), "outproj": (768, 768, 2), "fc2": (768, 3072, 2)}


def algorithmic_bytes(shape):
    N, K, epi = SHAPES[shape]
    return 2 * (M * K + N * K) + (8 * M * N if epi == 2 else 2 * M * N)


def summarize(dirname):
    labels = json.load(open(LABELS))
    rows = []
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == "FETCH_SIZE" and "gemm_bf16_pp_kernel" in row["Kernel_Name"]:
                    rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    rows.sort()
    if len(rows) != len(labels):
        raise SystemExit(f"{len(labels)} labels, {len(rows)} rows")
    acc = defaultdict(list)
    for lab, (_, v) in zip(labels, rows):
        acc[lab].append(v)
    for lab, v in acc.items():
        print(json.dumps({"config": lab, "fetch_MB": round(2 * sum(v) / len(v) * 1024 / 1e6, 1),
                          "algorithmic_read_MB": round(2 * (M * SHAPES[lab.split()[0]][1] + SHAPES[lab.split()[0]][0] * SHAPES[lab.split()[0]][1]) / 1e6, 1)}))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "summarize":
        return summarize(sys.argv[2])
    groups = [int(g) for g in (sys.argv[1] if len(sys.argv) > 1 else "8,1,2,4,16").split(",")]
    run_mode = len(sys.argv) > 2 and sys.argv[2] == "run"
    B = importlib.import_module("vision-transformer-opencl_amd.binding")
    from tools.gemm_probe import timed
    L = B.lib()
    L.vithip_gemm_bf16.argtypes = [C.c_void_p, C.POINTER(B.CGemmBf16Args)]
    rng = np.random.default_rng(0)
    labels = []
    for shape, (N, K, epi) in SHAPES.items():
        a = rng.integers(0x3c00, 0x4000, size=(M, K), dtype=np.uint16)
        a[::2] |= 0x8000
        dA = B.DeviceArray.from_numpy(a)
        dW = B.DeviceArray.from_numpy(B.to_bf16_bits(rng.uniform(-.05, .05, (N, K)).astype(np.float32)))
        db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
        dC = B.DeviceArray((M, N), np.float32 if epi == 2 else np.uint16)
        args = B.CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, dC.ptr if epi == 2 else None, N, dC.ptr, N, M, N, K, epi, 2, None, None, None, 0, None)
        launch = lambda: B.hip_check(L.vithip_gemm_bf16(None, C.byref(args)))
        if run_mode:
            for g in groups:
                B.hip_check(L.vithip_gemm_bf16_set_group(g))
                for _ in range(3):
                    launch()
                    labels.append(f"{shape} group_m={g}")
            B.hip_check(L.vithip_device_sync())
        else:
            ms = defaultdict(list)
            for _ in range(3):
                for g in groups:
                    B.hip_check(L.vithip_gemm_bf16_set_group(g))
                    ms[g].append(timed(launch, reps=5, warm=2))
            for g, t in ms.items():
                print(json.dumps({"shape": shape, "group_m": g, "ms_min": round(min(t), 4), "ms_median": round(sorted(t)[1], 4),
                                  "tflops": round(2.0 * M * N * K / (min(t) * 1e-3) / 1e12, 1)}), flush=True)
        for d in (dA, dW, db, dC):
            d.free()
    L.vithip_gemm_bf16_set_group(0)
    if run_mode:
        os.makedirs(os.path.dirname(LABELS), exist_ok=True)
        json.dump(labels, open(LABELS, "w"))


if __name__ == "__main__":
    main()
