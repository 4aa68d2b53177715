#!/usr/bin/env python3
"""Time vithip_attention_f32 at the metric shape (256 images x 12 heads x 197 tokens) with HIP events. GPU box only.
VIT_TOOL_DATA=zeros: an all-zero qkv (the same instructions at a lower power draw)."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("vision-transformer-opencl_amd.binding")
from tools.gemm_probe import timed
n, T, heads = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 197, 12
D = heads * 64
rng = np.random.default_rng(0)
vals = rng.uniform(-1.5, 1.5, (n * T, 3 * D)).astype(np.float32)
if os.environ.get("VIT_TOOL_DATA") == "zeros":
    vals[:] = 0
dq = B.DeviceArray.from_numpy(vals)
do = B.DeviceArray((n * T, D))
L = B.lib()
ms = [timed(lambda: B.hip_check(L.vithip_attention_f32(None, dq.ptr, do.ptr, n, T, heads)), reps=10, warm=3) for _ in range(3)]
flop = 2.0 * n * 2 * heads * T * T * 64
print(json.dumps({"attention_ms": [round(m, 4) for m in ms], "tflops": round(flop / (min(ms) * 1e-3) / 1e12, 1),
                  "frac_of_157.3": round(flop / (min(ms) * 1e-3) / 1e12 / 157.3, 3)}))
