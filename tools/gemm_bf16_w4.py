#!/usr/bin/env python3
"""EXPERIMENT (probe library): the four-wave bf16 GEMM (tools/probes/vit_gemm_bf16_w4.hip, 128 x 128 per wave) against the shipped
ping-pong kernel -- first that it computes the same product, then launch times at the ViT shapes, interleaved.  GPU box only.

    VIT_HIP_LIBRARY=$PWD/vision-transformer-opencl_amd/libvit_mi355x_probe.so python3 tools/gemm_bf16_w4.py [batch] [b16|l16_384]
"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed  # noqa: E402

L = B.lib()
L.vithip_gemm_bf16.argtypes = [C.c_void_p, C.POINTER(B.CGemmBf16Args)]


def run(dA, dW, db, dC, M, N, K, epi, variant):
    B.hip_check(L.vithip_gemm_bf16_set_variant(variant), "set_variant")
    args = B.CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, None, N, dC.ptr, N, M, N, K, epi, 0, None, None, None, 0, None)
    return lambda: B.hip_check(L.vithip_gemm_bf16(None, C.byref(args)), "gemm_bf16")


def check():
    rng = np.random.default_rng(1)
    for (M, N, K, epi) in ((256 * 9 + 77, 768, 768, 0), (256 * 3, 1024, 256, 1), (300, 320, 128, 0)):
        a = B.to_bf16_bits(rng.uniform(-1, 1, (M, K)).astype(np.float32))
        w = B.to_bf16_bits(rng.uniform(-.05, .05, (N, K)).astype(np.float32))
        b = rng.uniform(-.1, .1, (N,)).astype(np.float32)
        dA, dW, db = B.DeviceArray.from_numpy(a), B.DeviceArray.from_numpy(w), B.DeviceArray.from_numpy(b)
        outs = {}
        for v in (2, 5):
            dC = B.DeviceArray.from_numpy(np.full((M, N), 0x7fc0, np.uint16))
            run(dA, dW, db, dC, M, N, K, epi, v)()
            outs[v] = B.from_bf16_bits(dC.numpy()).astype(np.float64)
        exact = B.from_bf16_bits(a).astype(np.float64) @ B.from_bf16_bits(w).astype(np.float64).T + b
        if epi == 1:
            from scipy.special import erf
            exact = 0.5 * exact * (1 + erf(exact / np.sqrt(2)))
        e2, e5 = np.abs(outs[2] - exact).max(), np.abs(outs[5] - exact).max()
        differ = int((outs[2] != outs[5]).sum())
        print(json.dumps({"check": [M, N, K, epi], "max_err_pp": e2, "max_err_w4": e5, "elements_that_differ": differ, "of": M * N}))
        assert e5 <= 2.0 ** -8 * np.abs(exact).max() + 1e-6, "w4 kernel is wrong"


def time_shapes(batch, model):
    T, D = (197, 768) if model == "b16" else (577, 1024)
    M = batch * T
    rng = np.random.default_rng(0)
    for name, (N, K, epi) in {"qkv": (3 * D, D, 0), "fc1": (4 * D, D, 1), "fc2-shaped, bf16 out": (D, 4 * D, 0)}.items():
        a = rng.integers(0x3c00, 0x4000, size=(M, K), dtype=np.uint16)
        a[::2] |= 0x8000
        dA = B.DeviceArray.from_numpy(a)
        dW = B.DeviceArray.from_numpy(B.to_bf16_bits(rng.uniform(-.05, .05, (N, K)).astype(np.float32)))
        db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
        dC = B.DeviceArray((M, N), np.uint16)
        arms = {"pp": run(dA, dW, db, dC, M, N, K, epi, 2), "w4": run(dA, dW, db, dC, M, N, K, epi, 5)}
        if epi == 0:
            B.hip_check(L.vithip_gemm_bf16_set_variant(0), "set_variant")
            for code, label in ((501, "w4 without the LDS-DMA"), (502, "w4 without MFMAs and fragment reads"),
                                (505, "w4 copies alone, 3 steps in flight"), (504, "w4 copies alone, 2 steps in flight")):
                args = B.CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, None, N, dC.ptr, N, M, N, K, code, 0, None, None, None, 0, None)
                arms[label] = (lambda a_=args: (B.hip_check(L.vithip_gemm_bf16_set_variant(0), "v"), B.hip_check(L.vithip_gemm_bf16(None, C.byref(a_)), "gemm")))
        res = {k: [] for k in arms}
        for _ in range(3):
            for k, fn in arms.items():
                if k in ("pp", "w4"):
                    B.hip_check(L.vithip_gemm_bf16_set_variant(2 if k == "pp" else 5), "set_variant")
                res[k].append(timed(fn, reps=5, warm=2))
        print(json.dumps({"shape": name, "M": M, "N": N, "K": K,
                          "arms": {k: {"ms_min": round(min(v), 4), "tflops": round(2.0 * M * N * K / (min(v) * 1e-3) / 1e12, 1)} for k, v in res.items()}}), flush=True)
        for d in (dA, dW, db, dC):
            d.free()
    B.hip_check(L.vithip_gemm_bf16_set_variant(0), "set_variant")


if __name__ == "__main__":
    check()
    time_shapes(int(sys.argv[1]) if len(sys.argv) > 1 else 2048, sys.argv[2] if len(sys.argv) > 2 else "b16")
