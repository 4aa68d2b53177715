#!/usr/bin/env python3
"""Time and stamp the streamed bf16 attention kernel at ViT-L/16-384 (probe build for the stamps); `2048 197 12` times the resident
kernel at ViT-B/16's shape.  GPU box only.
VIT_TOOL_DATA=zeros: an all-zero qkv (the same instructions at a lower power draw: is the kernel clock-limited?)."""
import importlib, json, os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
n, T, heads = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 577, int(sys.argv[3]) if len(sys.argv) > 3 else 16
D = heads * 64
rng = np.random.default_rng(0)
vals = rng.uniform(-1.5, 1.5, (n * T, 3 * D)).astype(np.float32)
if os.environ.get("VIT_TOOL_DATA") == "zeros":
    vals[:] = 0
dq = B.DeviceArray.from_numpy(B.to_bf16_bits(vals))
vals[:, :D] *= np.float32(B.QSCALE)          # what the engine's folded in_proj writes into the Q columns
dqs = B.DeviceArray.from_numpy(B.to_bf16_bits(vals))
del vals
do = B.DeviceArray((n * T, D), np.uint16)
L = B.lib()
L.vithip_attention_bf16io.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
flop = 2.0 * n * 2 * heads * T * T * 64
L.vithip_attention_bf16io_qscaled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
for rnd in range(2):   # interleaved A/B in one process: plain q, then the q-scaled entry (same bits: timing only)
    ms = [timed(lambda: B.hip_check(L.vithip_attention_bf16io(None, dq.ptr, do.ptr, n, T, heads)), reps=5, warm=2) for _ in range(3)]
    print(json.dumps({"attention_ms": [round(m, 4) for m in ms], "tflops": round(flop / (min(ms) * 1e-3) / 1e12, 1)}))
    ms = [timed(lambda: B.hip_check(L.vithip_attention_bf16io_qscaled(None, dqs.ptr, do.ptr, n, T, heads, T, 0)), reps=5, warm=2) for _ in range(3)]
    print(json.dumps({"qscaled_attention_ms": [round(m, 4) for m in ms], "tflops": round(flop / (min(ms) * 1e-3) / 1e12, 1)}))
if hasattr(L, "vithip_attention_set_debug_buffer"):
    dbg = B.DeviceArray((256 * 8, 16), np.uint64)
    L.vithip_attention_set_debug_buffer.argtypes = [C.c_void_p]
    L.vithip_attention_set_debug_buffer(dbg.ptr)
    for _ in range(2):
        B.hip_check(L.vithip_attention_bf16io(None, dq.ptr, do.ptr, n, T, heads))
    d = dbg.numpy().astype(np.int64).reshape(256, 8, 16)
    L.vithip_attention_set_debug_buffer(None)
    for w in (0, 3, 7):
        seg = np.median(d[:, w, 1:13] - d[:, w, 0:12], axis=0).astype(int)
        print(json.dumps({"wave": w, "segments(top-barrier, then per step: compute, barrier) -- or, for a -DST_UNIT_STAMPS=1 build, the second step of item 1: "
                          "[first score burst, second burst, max (0,0), decide+exp+sums (0,0), pack+PV (0,0), next block's burst, max (0,1), exp (0,1), PV (0,1)]": seg.tolist()}))
