#!/usr/bin/env python3
"""Time the fp32 GEMM kernel variants and the pure-MFMA probe with HIP events (GPU box only).

    python tools/gemm_probe.py [--shapes fc1,fc2,qkv,outproj] [--tiles 1,2,3,101,102,103,104]
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")

SHAPES = {"qkv": (50432, 2304, 768), "outproj": (50432, 768, 768), "fc1": (50432, 3072, 768),
          "fc2": (50432, 768, 3072), "head": (256, 1000, 768)}


def timed(fn, reps=5, warm=2):
    L = B.lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.vithip_event_create(C.byref(e0)); L.vithip_event_create(C.byref(e1))
    for _ in range(warm):
        fn()
    L.vithip_event_record(e0, None)
    for _ in range(reps):
        fn()
    L.vithip_event_record(e1, None)
    L.vithip_event_sync(e1)
    ms = C.c_float()
    L.vithip_event_elapsed_ms(C.byref(ms), e0, e1)
    return ms.value / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="fc1,fc2,qkv,outproj")
    ap.add_argument("--tiles", default="1,2,3,101,102,103,104")
    ap.add_argument("--epilogue", type=int, default=0)
    ap.add_argument("--groups", default="8")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--stamp-tile", type=int, default=105, help="105 classic, 125 pipelined, 126 pipelined+GELU")
    ap.add_argument("--stamps", action="store_true", help="run the clock-stamp probe (tile 105)")
    args = ap.parse_args()
    L = B.lib()
    L.vithip_probe_mfma_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    out = B.DeviceArray((16,))
    res = {}
    for blocks, threads in ((256, 256), (512, 256), (256, 512), (1024, 256)):
        iters = 4000
        ms = timed(lambda: B.hip_check(L.vithip_probe_mfma_f32(None, out.ptr, blocks, threads, iters)), reps=3, warm=1)
        flop = blocks * (threads // 64) * iters * 32 * 4096.0
        res[f"mfma_{blocks}x{threads}"] = round(flop / (ms * 1e-3) / 1e12, 1)
    print(json.dumps({"pure_mfma_tflops": res}))
    rng = np.random.default_rng(0)
    for name in args.shapes.split(","):
        M, N, K = SHAPES[name]
        dA = B.DeviceArray.from_numpy(rng.uniform(-1, 1, (M, K)).astype(np.float32))
        dW = B.DeviceArray.from_numpy(rng.uniform(-.05, .05, (N, K)).astype(np.float32))
        db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
        dC = B.DeviceArray((M, N))
        ga = B.CGemmArgs(dA.ptr, K, dW.ptr, K, db.ptr, dC.ptr, N, dC.ptr, N, M, N, K, args.epilogue)
        row = {}
        variants = [(t, g) for t in [int(x) for x in args.tiles.split(",")] for g in [int(x) for x in args.groups.split(",")]]
        samples = {v: [] for v in variants}
        for _ in range(args.rounds):  # interleaved rounds in one process (device-to-device spread is ~10 %)
            for (t, g) in variants:
                B.hip_check(L.vithip_gemm_set_tile(t))
                B.hip_check(L.vithip_gemm_set_group(g))
                samples[(t, g)].append(timed(lambda: B.hip_check(L.vithip_gemm_f32(None, C.byref(ga))), reps=3, warm=1))
        for (t, g), ms in samples.items():
            med = float(np.median(ms))
            row[f"t{t}g{g}"] = {"ms_med": round(med, 3), "ms_min": round(min(ms), 3),
                                "tflops_med": round(2.0 * M * N * K / (med * 1e-3) / 1e12, 1)}
        if args.stamps:
            nwg = ((M + 127) // 128) * ((N + 127) // 128)
            dbg = B.DeviceArray((nwg, 8), np.uint64)
            L.vithip_gemm_set_debug_buffer.argtypes = [C.c_void_p]
            L.vithip_gemm_set_debug_buffer(dbg.ptr)
            B.hip_check(L.vithip_gemm_set_tile(args.stamp_tile))
            ms105 = timed(lambda: B.hip_check(L.vithip_gemm_f32(None, C.byref(ga))))
            d = dbg.numpy().astype(np.int64)
            if args.stamp_tile == 129:  # persistent kernel: [start, loop start, sum of epilogues, end, rt0, rt1, tiles, steps]
                d = d[d[:, 7] > 0]
                tot, epi, steps, tiles = d[:, 3] - d[:, 0], d[:, 2], d[:, 7], d[:, 6]
                clk = tot / np.maximum((d[:, 5] - d[:, 4]).astype(np.float64), 1) * 100.0
                print(json.dumps({name + "_persistent_stamps": {
                    "event_ms": round(ms105, 3), "wgs": int(len(d)), "span_us": round(float((d[:, 5].max() - d[:, 4].min()) / 100.0), 1),
                    "clock_mhz_median": round(float(np.median(clk))), "wg_total_cycles_median": int(np.median(tot)),
                    "prologue_cycles_median": int(np.median(d[:, 1] - d[:, 0])),
                    "cycles_per_step_median": round(float(np.median((d[:, 3] - d[:, 1] - epi) / steps))),
                    "epilogue_cycles_per_tile_median": round(float(np.median(epi / tiles))),
                    "tiles_per_wg": [int(tiles.min()), int(tiles.max())]}}))
                L.vithip_gemm_set_debug_buffer(None)
                dbg.free()
                L.vithip_gemm_set_tile(0)
                continue
            tot, pro, loop, epi = d[:, 3] - d[:, 0], d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
            rt = (d[:, 5] - d[:, 4]).astype(np.float64)  # 100 MHz ticks
            clk = tot / np.maximum(rt, 1) * 100.0  # MHz
            span = (d[:, 5].max() - d[:, 4].min()) / 100.0  # us
            nk = K // 32
            print(json.dumps({name + "_stamps": {
                "event_ms": round(ms105, 3), "wgs": int(nwg), "kernel_span_us": round(float(span), 1),
                "clock_mhz_median": round(float(np.median(clk)), 0), "clock_mhz_p10": round(float(np.percentile(clk, 10)), 0),
                "wg_total_cycles_median": int(np.median(tot)), "prologue_cycles_median": int(np.median(pro)),
                "loop_cycles_per_iter_median": round(float(np.median(loop)) / nk, 0),
                "loop_cycles_per_iter_p10_p90": [round(float(np.percentile(loop, 10)) / nk, 0), round(float(np.percentile(loop, 90)) / nk, 0)],
                "epilogue_cycles_median": int(np.median(epi)),
                "xcc_histogram": np.bincount(d[:, 6] & 15, minlength=8).tolist()}}))
            L.vithip_gemm_set_debug_buffer(None)
            dbg.free()
        L.vithip_gemm_set_tile(0)
        L.vithip_gemm_set_group(8)
        print(json.dumps({name: row}))
        for d in (dA, dW, db, dC):
            d.free()


if __name__ == "__main__":
    main()
