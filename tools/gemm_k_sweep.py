#!/usr/bin/env python3
"""Per-tile overhead of the fp32 persistent GEMM: time M = 50432 (batch 256), N = 2304 at K = 384 .. 3072 and fit
ms = a * K + b: a = the K loop (asymptotic TFLOP/s), b = what a launch pays regardless of K (epilogues, the partial last round
of tiles, launch and pipeline fill).  Round 3 measured b = 6.7 % of the K = 768 launch and then ruled three explanations out, each
with an A/B in one process on one device: hiding the epilogue behind the next tile's K loop (second accumulator set: no change, so
the two workgroups of a CU already cover each other's epilogues), 16-byte instead of 4-byte stores (transposed accumulators: no
change: the stores are not address-path-bound), a start-up skew between workgroups (no change: no chip-wide store burst).
GPU box only.   python tools/gemm_k_sweep.py [epilogue 0|1|2]"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
epi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M, N = 50432, 2304
L = B.lib()
rng = np.random.default_rng(0)
pts = []
for K in (384, 768, 1536, 3072):
    dA = B.DeviceArray.from_numpy(rng.uniform(-1, 1, (M, K)).astype(np.float32))
    dW = B.DeviceArray.from_numpy(rng.uniform(-.05, .05, (N, K)).astype(np.float32))
    db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
    dC = B.DeviceArray((M, N))
    args = B.CGemmArgs(dA.ptr, K, dW.ptr, K, db.ptr, dC.ptr if epi == 2 else None, N, dC.ptr, N, M, N, K, epi, 9, 0, None, 0)
    ms = min(timed(lambda: B.hip_check(L.vithip_gemm_f32(None, C.byref(args))), reps=5, warm=2) for _ in range(3))
    pts.append((K, ms))
    print(json.dumps({"K": K, "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}))
    for d in (dA, dW, db, dC):
        d.free()
k = np.array([p[0] for p in pts], float); t = np.array([p[1] for p in pts], float)
a, b = np.polyfit(k, t, 1)
print(json.dumps({"fit_ms": {"per_K": a, "const": b}, "overhead_share_at_K768": round(b / (a * 768 + b), 4),
                  "asymptotic_tflops": round(2.0 * M * N / a / 1e9, 1)}))
