#!/usr/bin/env python3
"""Interleaved A/B of vithip_gemm_f32 of SEVERAL builds of the library in one process on one device: the fold's consumer GEMMs
(QKV, fc1 with GELU) at the metric batch (M = 50,432) with the centred weight (ln_colsum NULL), and the same shapes unfolded.
GPU box only.     python3 tools/gemm_f32_ab.py <rounds> <lib.so> [<lib.so> ...]       (builds: tools/build_variant.sh)
Every library's output is compared with the first one's bit for bit."""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
rounds, paths = int(sys.argv[1]), sys.argv[2:]
M, D, H = 50432, 768, 3072
rng = np.random.default_rng(0)
f = lambda *shape, a=1.0: B.DeviceArray.from_numpy(rng.uniform(-a, a, shape).astype(np.float32))
x, rows = f(M, D), B.DeviceArray.from_numpy(np.stack([rng.uniform(0.5, 1.5, M), rng.uniform(-1, 1, M)], 1).astype(np.float32))
ws = B.gemm_workspace()
fns = []
for p in paths:
    L = C.CDLL(os.path.join(ROOT, p))
    L.vithip_gemm_f32.argtypes = [C.c_void_p, C.POINTER(B.CGemmArgs)]
    fns.append(L.vithip_gemm_f32)
for name, N, epi in (("qkv", 3 * D, 0), ("fc1", H, 1)):
    W, b = f(N, D, a=.03), f(N, a=.1)
    outs = [B.DeviceArray((M, N)) for _ in paths]
    for fold in (False, True):
        ms = {p: [] for p in paths}
        for rnd in range(rounds):
            for p, fn, o in zip(paths, fns, outs):
                a = B.CGemmArgs(x.ptr, D, W.ptr, D, b.ptr, None, N, o.ptr, N, M, N, D, epi, 0, 0, ws, 0, rows.ptr if fold else None, None, None, None)
                ms[p].append(timed(lambda: B.hip_check(fn(None, C.byref(a)), "gemm"), reps=6, warm=2))
        ref = outs[0].numpy()
        for p, o in zip(paths, outs):
            print(json.dumps({"gemm": name, "fold": fold, "lib": os.path.basename(p), "us_min": round(min(ms[p]) * 1e3, 1),
                              "us_median": round(float(np.median(ms[p])) * 1e3, 1),
                              "differing_outputs_vs_first": int((o.numpy().view(np.uint32) != ref.view(np.uint32)).sum())}), flush=True)
    del W, b, outs
