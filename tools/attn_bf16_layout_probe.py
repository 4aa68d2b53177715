#!/usr/bin/env python3
"""Is the bf16 attention kernel bound by the strided 128-B pieces it reads?  Same work and bytes, two layouts:
(n images, 12 heads): K/V rows of a head are 128 B every 4608 B;  (12n images, 1 head): 128 B every 384 B.
GPU box only."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
L = B.lib()
L.vithip_attention_bf16io.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
n, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048), 197
rng = np.random.default_rng(0)
qkv = rng.integers(0x3c00, 0x3f80, size=(n * T * 2304,), dtype=np.uint16)
qkv[::2] |= 0x8000
dq = B.DeviceArray.from_numpy(qkv)
do = B.DeviceArray((n * T * 768,), np.uint16)
for label, nn, heads in (("12 heads (row stride 4608 B)", n, 12), ("1 head, 12x images (row stride 384 B)", 12 * n, 1),
                         ("12 heads again", n, 12)):
    ms = min(timed(lambda: B.hip_check(L.vithip_attention_bf16io(None, dq.ptr, do.ptr, nn, T, heads)), reps=5, warm=2) for _ in range(3))
    gb = (n * T * 2304 * 2 + n * T * 768 * 2) / 1e9
    print(f"{label:40s}: {ms:.3f} ms  {gb / ms:.2f} TB/s")
