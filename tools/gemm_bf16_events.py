#!/usr/bin/env python3
"""Per-K-step event log of workgroup 0 of the ping-pong bf16 GEMM (variant 4): where the cycles of a tile go.
    python tools/gemm_bf16_events.py [batch] [qkv|outproj|fc2]      (GPU box only)
"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
L = B.lib()
L.vithip_gemm_bf16.argtypes = [C.c_void_p, C.POINTER(B.CGemmBf16Args)]
L.vithip_gemm_bf16_set_debug_buffer.argtypes = [C.c_void_p]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
name = sys.argv[2] if len(sys.argv) > 2 else "qkv"
max_wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
M = batch * 197
M_, N, K, epi = {"qkv": (M, 2304, 768, 0), "outproj": (M, 768, 768, 2), "fc2": (M, 768, 3072, 2)}[name]
rng = np.random.default_rng(0)
a = rng.integers(0x3c00, 0x4000, size=(M_, K), dtype=np.uint16)
a[::2] |= 0x8000
dA = B.DeviceArray.from_numpy(a)
dW = B.DeviceArray.from_numpy(B.to_bf16_bits(rng.uniform(-.05, .05, (N, K)).astype(np.float32)))
db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
dC = B.DeviceArray((M_, N), np.float32 if epi == 2 else np.uint16)
dbg = B.DeviceArray((8 * 512,), np.uint32)
args = B.CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, dC.ptr if epi == 2 else None, N, dC.ptr, N, M_, N, K, epi)
B.hip_check(L.vithip_gemm_bf16_set_debug_buffer(dbg.ptr))
L.vithip_gemm_bf16_set_variant(4)
L.vithip_gemm_bf16_set_max_workgroups(max_wgs)
for _ in range(3):
    B.hip_check(L.vithip_gemm_bf16(None, C.byref(args)))
ev = dbg.numpy().reshape(8, 512)
L.vithip_gemm_bf16_set_variant(0)
nk = K // 64
print(f"max workgroups {max_wgs or 'all'}")
print(f"{name}: M={M_} N={N} K={K}, {nk} K steps per tile; times in cycles since the wave's first event")
for w in (0, 4):
    e = ev[w]
    e = e[e != 0]
    tags, t = e >> 28, (e & 0x0fffffff).astype(np.int64)
    t = (t - t[0]) & 0x0fffffff
    print(f"--- wave {w} (group {w >> 2}): {len(e)} events")
    ks = t[tags == 1]
    d = np.diff(ks)
    ntile = min(6, len(ks) // nk)
    for ti in range(min(ntile, 3)):
        seg = d[ti * nk:(ti + 1) * nk]
        print(f"tile {ti}: K-step durations {' '.join(str(int(x)) for x in seg)}")
    eb, ee = t[tags == 2], t[tags == 3]
    print("epilogues (begin, duration):", [(int(b), int(x - b)) for b, x in zip(eb[:6], ee[:6])])
    print("K-step 0 starts of tiles:", [int(x) for x in ks[::nk][:8]])
    print("kernel end:", int(t[tags == 4][0]) if (tags == 4).any() else None)
