#!/usr/bin/env python3
"""Interleaved A/B of vithip_gemm_bf16 of SEVERAL builds of the library in one process on one device, at the ViT shapes in their
LayerNorm-folded forms (what the engine launches).  GPU box only.
    python3 tools/gemm_bf16_ab.py <batch> <b16|l16_384> <qkv+fc1+...> <lib.so> [<lib.so> ...]   (builds: tools/build_variant.sh)
Every library's output is compared with the first one's bit for bit."""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
batch, model, shapes, paths = int(sys.argv[1]), sys.argv[2], sys.argv[3].split("+"), sys.argv[4:]
T, D = (197, 768) if model == "b16" else (577, 1024)
M = batch * T
SHAPES = {"qkv": (3 * D, D, 0), "outproj": (D, D, 2), "fc1": (4 * D, D, 1), "fc2": (D, 4 * D, 2)}
fns = []
for p in paths:
    L = C.CDLL(os.path.join(ROOT, p))
    L.vithip_gemm_bf16.argtypes = [C.c_void_p, C.POINTER(B.CGemmBf16Args)]
    fns.append(L.vithip_gemm_bf16)
for name in shapes:
    N, K, epi = SHAPES[name]
    rng = np.random.default_rng(0)
    a = rng.integers(0x3c00, 0x4000, size=(M, K), dtype=np.uint16)
    a[::2] |= 0x8000
    dA = B.DeviceArray.from_numpy(a)
    dW = B.DeviceArray.from_numpy(B.to_bf16_bits(rng.uniform(-.05, .05, (N, K)).astype(np.float32)))
    db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
    outs = [B.DeviceArray((M, N), np.float32 if epi == 2 else np.uint16) for _ in paths]
    if epi == 2:
        extra = [B.DeviceArray((M, N), np.uint16), B.DeviceArray((B.ln_strips(N), M, 2), np.float32)]
        tail = (None, None, extra[0].ptr, N, extra[1].ptr)
        res = B.DeviceArray.from_numpy(rng.uniform(-1, 1, (M, N)).astype(np.float32))
    else:
        extra = [B.DeviceArray.from_numpy(rng.uniform(0.5, 1.5, (M, 2)).astype(np.float32)),
                 B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))]
        tail = (extra[0].ptr, extra[1].ptr, None, 0, None)
        res = None
    ms = {p: [] for p in paths}
    for rnd in range(5):
        for p, f, o in zip(paths, fns, outs):
            args = B.CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, res.ptr if res else None, N, o.ptr, N, M, N, K, epi, 0, *tail)
            ms[p].append(timed(lambda: B.hip_check(f(None, C.byref(args))), reps=5, warm=2 if rnd == 0 else 1))
    ref = outs[0].numpy()
    for p, o in zip(paths, outs):
        t = ms[p]
        print(json.dumps({"shape": name, "lib": os.path.basename(p), "ms": [round(x, 4) for x in t], "median_ms": round(float(np.median(t)), 4),
                          "tflops": round(2.0 * M * N * K / (float(np.median(t)) * 1e-3) / 1e12, 1),
                          "differing_outputs_vs_first": int((o.numpy().view(np.uint8) != ref.view(np.uint8)).sum())}), flush=True)
    del dA, dW, db, outs, extra, res
