#!/usr/bin/env python3
"""Race screen for the ping-pong bf16 GEMM (its LDS-DMA pipeline is ordered only by counted waits and barriers, so a
mis-placed wait shows up as a rare wrong tile, not as a failing unit test): every shape / epilogue / barrier schedule, plain
and in its LayerNorm-folded role (consumer: row pairs and column sums ride the DMA stream; producer: bf16 copy + row sums),
is run REPS times on the same random operands and every output must equal the first one bit for bit.  Product library;
GPU box only.
    python tools/gemm_bf16_race_screen.py [reps=40]"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
L = B.lib()
L.vithip_gemm_bf16.argtypes = [C.c_void_p, C.POINTER(B.CGemmBf16Args)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(1)
bad = 0
for (M, N, K) in [(256 * 37 + 19, 2052, 192), (256 * 80, 1024, 128), (256 * 150, 768, 768), (256 * 64, 3072, 768), (256 * 64, 768, 3072)]:
    a = rng.integers(0x3c00, 0x4000, size=(M, K), dtype=np.uint16)
    a[::2] |= 0x8000
    dA = B.DeviceArray.from_numpy(a)
    dW = B.DeviceArray.from_numpy(B.to_bf16_bits(rng.uniform(-.05, .05, (N, K)).astype(np.float32)))
    db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
    dR = B.DeviceArray.from_numpy(rng.uniform(-2, 2, (M, N)).astype(np.float32))
    dRows = B.DeviceArray.from_numpy(rng.uniform(0.5, 1.5, (M, 2)).astype(np.float32))
    dCs = B.DeviceArray.from_numpy(rng.uniform(-.2, .2, (N,)).astype(np.float32))
    for epi in (0, 1, 2):
        for fold in (False, True):
            if fold and epi == 2 and N % 8:
                continue  # the producer's bf16 copy needs ldx16 % 8 == 0
            if True:
                dC = B.DeviceArray((M, N), np.float32 if epi == 2 else np.uint16)
                extra = []
                tail = (None, None, None, 0, None)
                if fold and epi == 2:
                    extra = [B.DeviceArray((M, N), np.uint16), B.DeviceArray((B.ln_strips(N), M, 2), np.float32)]
                    tail = (None, None, extra[0].ptr, N, extra[1].ptr)
                elif fold:
                    tail = (dRows.ptr, dCs.ptr, None, 0, None)
                args = B.CGemmBf16Args(dA.ptr, K, dW.ptr, K, db.ptr, dR.ptr if epi == 2 else None, N, dC.ptr, N, M, N, K, epi,
                                       2, *tail)
                first, diffs = None, 0
                for r in range(reps):
                    B.hip_check(L.vithip_gemm_bf16(None, C.byref(args)))
                    out = [dC.numpy()] + [e.numpy() for e in extra]
                    if first is None:
                        first = out
                    elif not all(np.array_equal(x, y) for x, y in zip(out, first)):
                        diffs += 1
                bad += diffs
                print(f"M={M} N={N} K={K} epilogue {epi} fold={int(fold)}: {reps} runs, {diffs} differ", flush=True)
                for d in [dC] + extra:
                    d.free()
    for d in (dA, dW, db, dR, dRows, dCs):
        d.free()
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
