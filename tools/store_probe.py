#!/usr/bin/env python3
"""What does a 16-B-per-lane global store cost on a CU?  (GPU box only)  See vithip_probe_store."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
L = B.lib()
L.vithip_probe_store.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
iters = 64
for blocks in (16, 256):
    for waves in (1, 4, 8):
        for mode, stride, label in ((0, 1024, "1 KB contiguous"), (2, 4608, "8 rows x 128 B, stride 4608"), (1, 4608, "16 rows x 64 B, stride 4608"),
                                    (1, 64, "16 rows x 64 B, stride 64 (contiguous)"), (1, 3072, "16 rows x 64 B, stride 3072")):
            nw = blocks * waves
            out = B.DeviceArray((nw * iters * 16 * max(stride, 1024) // 4 + 1024,), np.float32)
            cyc = B.DeviceArray((nw * 2,), np.uint64)
            for _ in range(2):
                B.hip_check(L.vithip_probe_store(None, out.ptr, blocks, waves * 64, iters, mode, stride, cyc.ptr))
            c = cyc.numpy().reshape(nw, 2).astype(np.float64)
            print(f"{blocks:3d} CUs x {waves} waves, {label:40s}: issue {c[:,0].mean()/iters:7.1f} cyc/store/wave, complete {c[:,1].mean()/iters:7.1f}"
                  f"  -> {waves*1024/ (c[:,1].mean()/iters):6.1f} B/clk/CU")
            out.free(); cyc.free()
