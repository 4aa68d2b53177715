#!/usr/bin/env python3
"""Where the fp32 LayerNorm fold's consumer epilogue spends its 1-2 %: the stamped instantiation of the persistent walk (probe
library, tile code 129: per workgroup [start, loop start, SUM of epilogue cycles, end, ..., tiles, K-steps]) on the metric batch's QKV and
fc1 shapes, once with the plain epilogue (bias / bias + GELU on normalised rows) and once with the fold's (raw rows, gamma-folded
weights, rstd * (acc - mean * colsum) + bias').  GPU box only, probe library:
    VIT_HIP_LIBRARY=.../libvit_mi355x_probe.so python3 tools/gemm_f32_fold_stamps.py"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
L = B.lib()
M, D, H = 50432, 768, 3072
rng = np.random.default_rng(0)
f = lambda *shape, a=1.0: B.DeviceArray.from_numpy(rng.uniform(-a, a, shape).astype(np.float32))
x, rows = f(M, D), B.DeviceArray.from_numpy(np.stack([rng.uniform(0.5, 2.0, M), rng.uniform(-1, 1, M)], 1).astype(np.float32))
L.vithip_gemm_set_debug_buffer.argtypes = [C.c_void_p]
for name, N, epi in (("qkv", 3 * D, 0), ("fc1", H, 1)):
    W, b, cs, out = f(N, D, a=.03), f(N, a=.1), f(N, a=.1), B.DeviceArray((M, N))
    nwg = ((M + 127) // 128) * ((N + 127) // 128)
    for fold in (False, True, False, True):
        a = B.CGemmArgs(x.ptr, D, W.ptr, D, b.ptr, None, N, out.ptr, N, M, N, D, epi, 0, 0, None, 0,
                        rows.ptr if fold else None, cs.ptr if fold else None, None, None)
        dbg = B.DeviceArray((nwg, 8), np.uint64)
        L.vithip_gemm_set_debug_buffer(dbg.ptr)
        B.hip_check(L.vithip_gemm_set_tile(129))
        ms = timed(lambda: B.hip_check(L.vithip_gemm_f32(None, C.byref(a))), reps=3, warm=1)
        d = dbg.numpy().astype(np.int64)
        d = d[d[:, 7] > 0]
        tot, epi_c, steps, tiles = d[:, 3] - d[:, 0], d[:, 2], d[:, 7], d[:, 6]
        print(json.dumps({"gemm": name, "fold": fold, "event_ms": round(ms, 4), "wgs": int(len(d)),
                          "wg_total_cycles_median": int(np.median(tot)),
                          "loop_cycles_per_k_step_median": round(float(np.median((d[:, 3] - d[:, 1] - epi_c) / steps)), 1),
                          "epilogue_cycles_per_tile_median": round(float(np.median(epi_c / tiles))),
                          "epilogue_share_of_wg_time": round(float(np.median(epi_c / tot)), 4)}), flush=True)
        L.vithip_gemm_set_debug_buffer(None)
        L.vithip_gemm_set_tile(0)
        dbg.free()
