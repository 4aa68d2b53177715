#!/usr/bin/env python3
"""Is the fp32 GEMM power-limited?  The same launch (QKV shape, product library) on operands of different bit activity:
uniform random, one repeated row, all ones, all zeros.  Same instructions, same memory traffic; only the toggling of the
operand and accumulator bits differs.  GPU box only.   python tools/gemm_f32_data_power.py"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
M, N, K = 50432, 2304, 768
L = B.lib()
rng = np.random.default_rng(0)
kinds = {
    "uniform": (rng.uniform(-1, 1, (M, K)).astype(np.float32), rng.uniform(-.05, .05, (N, K)).astype(np.float32)),
    "one_row_repeated": (np.tile(rng.uniform(-1, 1, (1, K)).astype(np.float32), (M, 1)), np.tile(rng.uniform(-.05, .05, (1, K)).astype(np.float32), (N, 1))),
    "ones": (np.ones((M, K), np.float32), np.ones((N, K), np.float32)),
    "zeros": (np.zeros((M, K), np.float32), np.zeros((N, K), np.float32)),
}
dev = {k: (B.DeviceArray.from_numpy(a), B.DeviceArray.from_numpy(w)) for k, (a, w) in kinds.items()}
db = B.DeviceArray.from_numpy(np.zeros((N,), np.float32))
dC = B.DeviceArray((M, N))
for rnd in range(3):
    row = {}
    for k, (dA, dW) in dev.items():
        args = B.CGemmArgs(dA.ptr, K, dW.ptr, K, db.ptr, None, N, dC.ptr, N, M, N, K, 0, 9, 0, None, 0)
        row[k] = round(min(timed(lambda: B.hip_check(L.vithip_gemm_f32(None, C.byref(args))), reps=5, warm=2) for _ in range(3)), 4)
    print(json.dumps({"round": rnd, "ms": row, "tflops": {k: round(2.0 * M * N * K / v / 1e9, 1) for k, v in row.items()}}), flush=True)
