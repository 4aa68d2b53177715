#!/usr/bin/env python3
"""Where the fp32 persistent GEMM's time goes: the product kernel against timing-only instantiations with one part switched off
(csrc/vit_gemm_persistent.hip, template parameter DBG = n: 1 no epilogue stores, 2 no staging loads in the K loop, 3 no K-loop
barrier, 4 no fragment reads, 5 no staging ds_writes, 6 = 2 + 5; results wrong by construction).  They exist in the probe library
only (tile codes 130 + n through vithip_gemm_set_tile).  Every variant runs in its own process, the set is repeated so that drift
shows.  GPU box only.

    make -C vision-transformer-opencl_amd probes           # libvit_mi355x_probe.so
    python tools/gemm_f32_switchoff.py [rounds]

Read the numbers with DESIGN 4.1 item 11 in mind: every such build freezes the MFMAs' operand data, and the fp32 GEMM is
power-limited (tools/gemm_f32_data_power.py) -- most of what a build "saves" is clock, not the instructions it dropped.
"""
import ctypes as C, importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"qkv": (50432, 2304, 768, 0), "fc1": (50432, 3072, 768, 1), "outproj": (50432, 768, 768, 2), "fc2": (50432, 768, 3072, 2),
          "qkv_k3072": (50432, 2304, 3072, 0)}


def child():
    import numpy as np
    sys.path.insert(0, ROOT)
    B = importlib.import_module("vision-transformer-opencl_amd.binding")
    from tools.gemm_probe import timed
    L = B.lib()
    dbg = int(os.environ["PG_CHILD"])
    B.hip_check(L.vithip_gemm_set_tile(130 + dbg if dbg else 9), "vithip_gemm_set_tile")
    rng = np.random.default_rng(0)
    out = {}
    for name, (M, N, K, epi) in SHAPES.items():
        zero = os.environ.get("VIT_TOOL_DATA") == "zeros"   # quiet operands: no power effect, the structure alone
        dA = B.DeviceArray.from_numpy(np.zeros((M, K), np.float32) if zero else rng.uniform(-1, 1, (M, K)).astype(np.float32))
        dW = B.DeviceArray.from_numpy(np.zeros((N, K), np.float32) if zero else rng.uniform(-.05, .05, (N, K)).astype(np.float32))
        db = B.DeviceArray.from_numpy(rng.uniform(-.1, .1, (N,)).astype(np.float32))
        dC = B.DeviceArray((M, N))
        dR = B.DeviceArray((M, N))
        args = B.CGemmArgs(dA.ptr, K, dW.ptr, K, db.ptr, dR.ptr if epi == 2 else None, N, dC.ptr, N, M, N, K, epi, 9, 0, None, 0)
        out[name] = round(min(timed(lambda: B.hip_check(L.vithip_gemm_f32(None, C.byref(args))), reps=5, warm=2) for _ in range(3)), 4)
        for d in (dA, dW, db, dC, dR):
            d.free()
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("PG_CHILD") is not None:
        child()
        sys.exit(0)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    names = {0: "as shipped", 1: "no epilogue", 2: "no staging loads", 3: "no barrier", 4: "no fragment reads",
             5: "no ds_writes", 6: "no staging at all"}
    lib = os.path.join(ROOT, "vision-transformer-opencl_amd", "libvit_mi355x_probe.so")
    for r in range(rounds):
        for n, what in names.items():
            env = dict(os.environ, PG_CHILD=str(n), VIT_HIP_LIBRARY=lib)
            res = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
            line = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else res.stderr[-300:]
            print(json.dumps({"round": r, "build": what, "ms": json.loads(line) if line.startswith("{") else line}), flush=True)
