#!/usr/bin/env python3
"""The fp32 LayerNorm fold at the metric batch's shapes (M = 50,432), launch by launch: the consumer GEMMs (QKV, fc1) with and
without the fold epilogue (the latter with the column sums subtracted in the epilogue and with the CENTRED weight that needs none),
the residual GEMMs (out_proj, fc2) with and without the row statistics in their epilogue, and the
passes the fold replaces or adds (LayerNorm kernel, statistics kernel, finalise).  HIP-event times, arms interleaved.  GPU box only.

    python3 tools/gemm_f32_fold.py [rounds]            # VIT_TOOL_ARMS="fc1,fc1 fold" to time a subset
"""
import ctypes as C
import importlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M, D, H = 50432, 768, 3072


def main():
    import numpy as np
    from tools.gemm_probe import timed
    B = importlib.import_module("vision-transformer-opencl_amd.binding")
    L = B.lib()
    L.vithip_rowstats_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    L.vithip_rowstats_finalize_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rng = np.random.default_rng(0)
    ws = B.gemm_workspace()
    f = lambda *shape, a=1.0: B.DeviceArray.from_numpy(rng.uniform(-a, a, shape).astype(np.float32))
    x, y, hbuf, qkv = f(M, D), f(M, D), f(M, H), B.DeviceArray((M, 3 * D))
    rows, part = f(M, 2), B.DeviceArray((D // 64, M, 2))
    g, be = f(D), f(D)
    Wq, bq, cq = f(3 * D, D, a=.03), f(3 * D, a=.1), f(3 * D, a=.1)
    W1, b1, c1 = f(H, D, a=.03), f(H, a=.1), f(H, a=.1)
    Wo, bo = f(D, D, a=.03), f(D, a=.1)
    W2, b2 = f(D, H, a=.02), f(D, a=.1)

    def gemm(A, K, W, b, Cc, N, epi, res=None, ln=False, stats=False, centred=False):
        a = B.CGemmArgs(A.ptr, K, W.ptr, K, b.ptr, res.ptr if res else None, N, Cc.ptr, N, M, N, K, epi, 0, 0, ws, 0,
                        rows.ptr if ln else None, (cq if N == 3 * D else c1).ptr if ln and not centred else None,
                        rows.ptr if stats else None, part.ptr if stats else None)
        return lambda: B.hip_check(L.vithip_gemm_f32(None, C.byref(a)), "gemm")

    arms = {
        "qkv": gemm(y, D, Wq, bq, qkv, 3 * D, 0), "qkv fold": gemm(x, D, Wq, bq, qkv, 3 * D, 0, ln=True),
        "qkv fold centred": gemm(x, D, Wq, bq, qkv, 3 * D, 0, ln=True, centred=True),
        "fc1": gemm(y, D, W1, b1, hbuf, H, 1), "fc1 fold": gemm(x, D, W1, b1, hbuf, H, 1, ln=True),
        "fc1 fold centred": gemm(x, D, W1, b1, hbuf, H, 1, ln=True, centred=True),
        "outproj": gemm(y, D, Wo, bo, x, D, 2, res=x), "outproj stats": gemm(y, D, Wo, bo, x, D, 2, res=x, stats=True),
        "fc2": gemm(hbuf, H, W2, b2, x, D, 2, res=x), "fc2 stats": gemm(hbuf, H, W2, b2, x, D, 2, res=x, stats=True),
        "layernorm kernel": lambda: B.hip_check(L.vithip_layernorm_f32(None, x.ptr, D, y.ptr, D, g.ptr, be.ptr, M, D), "ln"),
        "rowstats kernel": lambda: B.hip_check(L.vithip_rowstats_f32(None, x.ptr, D, rows.ptr, M, D), "rs"),
        "finalize kernel": lambda: B.hip_check(L.vithip_rowstats_finalize_f32(None, part.ptr, M, D, rows.ptr), "fin"),
    }
    only = [a.strip() for a in os.environ.get("VIT_TOOL_ARMS", "").split(",") if a.strip()]
    if only:
        arms = {k: v for k, v in arms.items() if k in only}
    res = defaultdict(list)
    for _ in range(rounds):
        for name, fn in arms.items():
            res[name].append(timed(fn, reps=6, warm=2))
    for name, ms in res.items():
        print(json.dumps({"arm": name, "us_min": round(min(ms) * 1e3, 1), "us_median": round(sorted(ms)[len(ms) // 2] * 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
