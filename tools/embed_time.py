#!/usr/bin/env python3
"""Time vithip_patch_embed_bf16_implicit at batch 2048 (ViT-B/16). GPU box only."""
import importlib, json, os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
pkg = importlib.import_module("vision-transformer-opencl_amd")
cfg = pkg.VIT_B16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = B.lib()
L.vithip_patch_embed_bf16_implicit.argtypes = [C.c_void_p] * 7 + [C.c_int] * 5
img = B.DeviceArray((n, 3, 224, 224)); w = B.DeviceArray((768, 768), np.uint16)
b = B.DeviceArray((768,)); cls = B.DeviceArray((768,)); pos = B.DeviceArray((197, 768)); x = B.DeviceArray((n * 197, 768))
ms = min(timed(lambda: B.hip_check(L.vithip_patch_embed_bf16_implicit(None, img.ptr, w.ptr, b.ptr, cls.ptr, pos.ptr, x.ptr, n, 224, 16, 3, 768)), reps=5, warm=2) for _ in range(3))
print(json.dumps({"implicit_embed_ms": round(ms, 4)}))
