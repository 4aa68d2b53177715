#!/usr/bin/env python3
"""Does fp32 VALU work on a co-resident wave slow the fp32 MFMA pipe of the same SIMD? (GPU box only)"""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
L = B.lib()
L.vithip_probe_mfma_vs_valu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
out = B.DeviceArray((16,))
iters = 4000  # 128k MFMAs x 64 cycles = 8.2M cycles per MFMA wave
res = {}
for valu_iters in (0, 16000, 32000, 64000, 128000):
    ms = timed(lambda: B.hip_check(L.vithip_probe_mfma_vs_valu(None, out.ptr, 256, iters, valu_iters)), reps=3, warm=1)
    res[valu_iters] = {"ms": round(ms, 3), "mfma_cycles_M": iters * 32 * 64 / 1e6, "valu_instr_M": valu_iters * 64 / 1e6}
print(json.dumps(res))

L.vithip_probe_mfma_vs_gelu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
res = {}
for mode in (0, 1):
    for mf in (0, 4000):
        for gi in (0, 20000):
            if mf == 0 and gi == 0:
                continue
            ms = timed(lambda: B.hip_check(L.vithip_probe_mfma_vs_gelu(None, out.ptr, 256, mf, gi, mode)), reps=3, warm=1)
            res[f"mode{mode}_mfma{mf}_gelu{gi}"] = round(ms, 3)
print(json.dumps({"gelu_probe_ms (8 gelu per iter per lane)": res}))
