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
# probe build: 8 cycle stamps per wave of every persistent workgroup's second (image, head) item
dbg = B.DeviceArray((256 * 8, 8), np.uint64)
L.vithip_attention_set_debug_buffer.argtypes = [C.c_void_p]
L.vithip_attention_set_debug_buffer(dbg.ptr)
for _ in range(2):
    B.hip_check(L.vithip_attention_f32(None, dq.ptr, do.ptr, n, T, heads))
d = dbg.numpy().astype(np.int64).reshape(256, 8, 8)
L.vithip_attention_set_debug_buffer(None)
for mode in (1, 2, 3, 4, 8, 15):
    L.vithip_attention_set_probe_mode(mode)
    m = min(timed(lambda: B.hip_check(L.vithip_attention_f32(None, dq.ptr, do.ptr, n, T, heads)), reps=5, warm=2) for _ in range(3))
    print(json.dumps({"probe_mode": mode, "ms": round(m, 4)}))
L.vithip_attention_set_probe_mode(0)
names = ["s_jobA", "s_job2", "wait_barrier1", "pv_jobA", "pv_job2", "wait_barrier2"]
for w in range(8):
    seg = np.median(d[:, w, 1:7] - d[:, w, 0:6], axis=0).astype(int)
    print(json.dumps({"wave": w, **dict(zip(names, seg.tolist())), "item_total": int(np.median(d[:, w, 6] - d[:, w, 0]))}))

# item cycles (s_memtime, wave 0 and wave 4) under each timing-only mode: tells a clock effect (same cycles, shorter time) from a structural one
for mode in (0, 1, 2, 8):
    L.vithip_attention_set_probe_mode(mode)
    L.vithip_attention_set_debug_buffer(dbg.ptr)
    for _ in range(2):
        B.hip_check(L.vithip_attention_f32(None, dq.ptr, do.ptr, n, T, heads))
    dd = dbg.numpy().astype(np.int64).reshape(256, 8, 8)
    L.vithip_attention_set_debug_buffer(None)
    seg = {w: np.median(dd[:, w, 1:7] - dd[:, w, 0:6], axis=0).astype(int).tolist() for w in (0, 4)}
    print(json.dumps({"probe_mode": mode, "item_cycles_wave0": int(np.median(dd[:, 0, 6] - dd[:, 0, 0])), "segments_wave0": seg[0], "segments_wave4": seg[4]}))
L.vithip_attention_set_probe_mode(0)
