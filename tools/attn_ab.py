#!/usr/bin/env python3
"""Interleaved A/B of the bf16 attention entry of SEVERAL builds of the library in one process on one device (CDNA4 guide rule 24;
devices of the pool differ by a few per cent, so numbers of different gpurun calls do not compare).  GPU box only.
    python3 tools/attn_ab.py <n> <tokens> <heads> <lib.so> [<lib.so> ...]      (paths relative to the repo; builds: tools/build_variant.sh)
Every library's output is compared with the first one's (max |difference| of the bf16 values) -- timing arms must agree."""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
n, T, heads = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
paths = sys.argv[4:]
D = heads * 64
rng = np.random.default_rng(0)
vals = rng.uniform(-1.5, 1.5, (n * T, 3 * D)).astype(np.float32)
vals[:, :D] *= np.float32(B.QSCALE)
dq = B.DeviceArray.from_numpy(B.to_bf16_bits(vals))
del vals
outs = [B.DeviceArray((n * T, D), np.uint16) for _ in paths]
fns = []
for p in paths:
    L = C.CDLL(os.path.join(ROOT, p))
    f = L.vithip_attention_bf16io_qscaled
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    fns.append(f)
flop = 2.0 * n * 2 * heads * T * T * 64
res = {p: [] for p in paths}
for rnd in range(int(os.environ.get("VIT_TOOL_ROUNDS", "4"))):
    for p, f, o in zip(paths, fns, outs):
        res[p].append(timed(lambda: B.hip_check(f(None, dq.ptr, o.ptr, n, T, heads, T)), reps=5, warm=2 if rnd == 0 else 1))
ref = B.from_bf16_bits(outs[0].numpy()).astype(np.float64)
for p, o in zip(paths, outs):
    d = float(np.abs(B.from_bf16_bits(o.numpy()).astype(np.float64) - ref).max())
    ms = res[p]
    print(json.dumps({"lib": os.path.basename(p), "ms": [round(m, 4) for m in ms], "median_ms": round(float(np.median(ms)), 4),
                      "tflops": round(flop / (float(np.median(ms)) * 1e-3) / 1e12, 1), "max_diff_vs_first": d}))
