#!/usr/bin/env python3
"""Time the fused attention kernel at the metric shape (256 images x 12 heads x 197 tokens). GPU box only."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
n, T, heads = 256, 197, 12
D = heads * 64
rng = np.random.default_rng(0)
dq = B.DeviceArray.from_numpy(rng.uniform(-1.5, 1.5, (n * T, 3 * D)).astype(np.float32))
do = B.DeviceArray((n * T, D))
L = B.lib()
ms = [timed(lambda: B.hip_check(L.vithip_attention_f32(None, dq.ptr, do.ptr, n, T, heads)), reps=5, warm=2) for _ in range(3)]
flop = 2.0 * n * 2 * heads * T * T * 64
print(json.dumps({"attention_ms": [round(m, 4) for m in ms], "tflops": round(flop / (min(ms) * 1e-3) / 1e12, 1)}))

import ctypes as C
dbg = B.DeviceArray((n * heads, 8), np.uint64)
L.vithip_attention_set_debug_buffer.argtypes = [C.c_void_p]
L.vithip_attention_set_debug_buffer(dbg.ptr)
for _ in range(2):
    B.hip_check(L.vithip_attention_f32(None, dq.ptr, do.ptr, n, T, heads))
d = dbg.numpy().astype(np.int64)
L.vithip_attention_set_debug_buffer(None)
med = lambda a: int(np.median(a))
print(json.dumps({"wave0_cycles_median": {"stage_kv": med(d[:, 1] - d[:, 0]), "qk": med(d[:, 2] - d[:, 1]),
                                          "softmax": med(d[:, 3] - d[:, 2]), "pv": med(d[:, 4] - d[:, 3]),
                                          "store": med(d[:, 5] - d[:, 4]), "total": med(d[:, 5] - d[:, 0])}}))
