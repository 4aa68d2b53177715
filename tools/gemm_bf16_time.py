#!/usr/bin/env python3
"""Time vithip_gemm_bf16 (product library, or the one VIT_HIP_LIBRARY names) at the ViT shapes.  GPU box only.

    python tools/gemm_bf16_time.py [batch] [b16|l16_384] [fold]     fold: the LayerNorm-folded forms (consumer qkv / fc1, producer outproj / fc2)

VIT_TOOL_DATA=zeros|ones: operands without bit activity instead of random ones -- the same instructions and traffic at a lower
power draw (the bf16 GEMMs run at the clock the power limit leaves them).
"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
L = B.lib()
L.vithip_gemm_bf16.argtypes = [C.c_void_p, C.POINTER(B.CGemmBf16Args)]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
model = sys.argv[2] if len(sys.argv) > 2 else "b16"
fold = len(sys.argv) > 3 and sys.argv[3] == "fold"
T, D = (197, 768) if model == "b16" else (577, 1024)
M = batch * T
SHAPES = {"qkv": (M, 3 * D, D, 0), "outproj": (M, D, D, 2), "fc1": (M, 4 * D, D, 1), "fc2": (M, D, 4 * D, 2)}
VARIANTS = [int(v) for v in os.environ.get("VIT_TOOL_VARIANTS", "0").split(",")]   # vithip_gemm_bf16_args.variant: 0 auto, 1 two-stage, 2 ping-pong (A/B in one process)
ROUNDS = int(os.environ.get("VIT_TOOL_ROUNDS", "3"))
out = {}
for name, (M_, N, K, epi) in SHAPES.items():
    rng = np.random.default_rng(0)
    data = os.environ.get("VIT_TOOL_DATA", "random")
    if data == "random":
        a = rng.integers(0x3c00, 0x4000, size=(M_, K), dtype=np.uint16)
        a[::2] |= 0x8000
        w = B.to_bf16_bits(rng.uniform(-.05, .05, (N, K)).astype(np.float32))
    else:
        a = np.full((M_, K), 0x3f80 if data == "ones" else 0, np.uint16)
        w = np.full((N, K), 0x3f80 if data == "ones" else 0, np.uint16)
    dA = B.DeviceArray.from_numpy(a)
    dW = B.DeviceArray.from_numpy(w)
    db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
    dC = B.DeviceArray((M_, N), np.float32 if epi == 2 else np.uint16)
    extra = []
    if fold and epi == 2:
        extra = [B.DeviceArray((M_, N), np.uint16), B.DeviceArray((B.ln_strips(N), M_, 2), np.float32)]
        tail = (None, None, extra[0].ptr, N, extra[1].ptr)
    elif fold:
        extra = [B.DeviceArray.from_numpy(rng.uniform(0.5, 1.5, (M_, 2)).astype(np.float32)),
                 B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))]
        tail = (extra[0].ptr, extra[1].ptr, None, 0, None)
    else:
        tail = (None, None, None, 0, None)
    ms = {v: [] for v in VARIANTS}
    for _ in range(ROUNDS):   # interleaved rounds in one process on one device: the only comparison that means anything
        for v in VARIANTS:
            args = B.CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, dC.ptr if epi == 2 else None, N, dC.ptr, N, M_, N, K, epi, v, *tail)
            ms[v].append(timed(lambda: B.hip_check(L.vithip_gemm_bf16(None, C.byref(args))), reps=5, warm=2))
    out[name] = {("variant%d" % v if len(VARIANTS) > 1 else "auto"): {"ms_min": round(min(t), 4), "ms_median": round(sorted(t)[len(t) // 2], 4),
                                                                       "tflops": round(2.0 * M_ * N * K / (min(t) * 1e-3) / 1e12, 1)}
                 for v, t in ms.items()}
    if len(VARIANTS) == 1:
        out[name] = out[name]["auto"] | {"ms": out[name]["auto"]["ms_min"]}
    for d in [dA, dW, db, dC] + extra:
        d.free()
print(json.dumps({"batch": batch, "model": model, "fold": fold, "data": os.environ.get("VIT_TOOL_DATA", "random"), "gemms": out}))
