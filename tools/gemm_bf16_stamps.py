#!/usr/bin/env python3
"""In-kernel timeline of the ping-pong bf16 GEMM (variant 3 = stamped build): s_memtime (100 MHz-independent
shader clock) around the load / MFMA sections of K step 3 of workgroup 0's first tile, per wave.  GPU box only.

    python tools/gemm_bf16_stamps.py [batch] [shape: qkv|outproj|fc1|fc2]
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
M = batch * 197
M_, N, K, epi = {"qkv": (M, 2304, 768, 0), "outproj": (M, 768, 768, 2), "fc1": (M, 3072, 768, 1), "fc2": (M, 768, 3072, 2)}[name]
rng = np.random.default_rng(0)
a = rng.integers(0x3c00, 0x4000, size=(M_, K), dtype=np.uint16)
a[::2] |= 0x8000
dA = B.DeviceArray.from_numpy(a)
dW = B.DeviceArray.from_numpy(B.to_bf16_bits(rng.uniform(-.05, .05, (N, K)).astype(np.float32)))
db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
dC = B.DeviceArray((M_, N), np.float32 if epi == 2 else np.uint16)
dbg = B.DeviceArray((8 * 32,), np.uint64)
args = B.CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, dC.ptr if epi == 2 else None, N, dC.ptr, N, M_, N, K, epi)
B.hip_check(L.vithip_gemm_bf16_set_debug_buffer(dbg.ptr))
L.vithip_gemm_bf16_set_variant(3)
for _ in range(3):
    B.hip_check(L.vithip_gemm_bf16(None, C.byref(args)))
L.vithip_gemm_bf16_set_variant(0)
s = dbg.numpy().reshape(8, 32).astype(np.int64) & 0xffffffff
d = lambda x, y: int((x - y) & 0xffffffff)
print(f"{name}: M={M_} N={N} K={K}  (cycles of the s_memtime clock)")
print("wave  " + "  ".join(f"L{p:d}   bar   M{p:d}   bar " for p in range(4)) + "  | kstep  tile(loop)  epilogue  kernel")
for w in range(8):
    r = s[w]
    cells = []
    for p in range(4):
        t0 = r[0] if p == 0 else r[5 * (p - 1) + 4]
        cells.append(f"{d(r[5*p+1], t0):4d} {d(r[5*p+2], r[5*p+1]):5d} {d(r[5*p+3], r[5*p+2]):5d} {d(r[5*p+4], r[5*p+3]):5d} ")
    print(f"{w:4d}  " + "  ".join(cells) + f"  | {d(r[19], r[0]):5d}  {d(r[21], r[20]):9d}  {d(r[22], r[21]):8d}  {d(r[23], r[20]):7d}"
          f"  | L0: reads {d(r[24], r[0])} dma {d(r[25], r[24])} wait {d(r[1], r[25])}  L3: reads {d(r[26], r[14])} dma {d(r[27], r[26])} wait {d(r[16], r[27])}")
