#!/usr/bin/env python3
"""Time vithip_patch_embed_f32 (implicit GEMM over NCHW fp32 images) at batch 256, per tile shape (probe build: tile override).
GPU box only.   VIT_HIP_LIBRARY=.../libvit_mi355x_probe.so python tools/embed_f32_time.py [batch]"""
import importlib, json, os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = B.lib()
rng = np.random.default_rng(0)
img = B.DeviceArray.from_numpy(rng.uniform(-2, 2, (n, 3, 224, 224)).astype(np.float32))
w = B.DeviceArray.from_numpy(rng.uniform(-.05, .05, (768, 768)).astype(np.float32))
b = B.DeviceArray((768,)); cls = B.DeviceArray((768,)); pos = B.DeviceArray((197, 768)); x = B.DeviceArray((n * 197, 768))
flop = 2.0 * n * 196 * 768 * 768
for rnd in range(2):
    for tile in (0, 10, 3):   # 0 = pipelined 128x64 (the default), 10 = pipelined 128x128, 3 = classic 128x64 (the round-2 kernel)
        if hasattr(L, "vithip_gemm_set_tile"):
            L.vithip_gemm_set_tile(tile)
        elif tile:
            continue
        ms = min(timed(lambda: B.hip_check(L.vithip_patch_embed_f32(None, img.ptr, w.ptr, b.ptr, cls.ptr, pos.ptr, x.ptr, n, 224, 16, 3, 768)), reps=5, warm=2) for _ in range(3))
        print(json.dumps({"tile": tile, "embed_ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1)}))
if hasattr(L, "vithip_gemm_set_tile"):
    L.vithip_gemm_set_tile(0)
