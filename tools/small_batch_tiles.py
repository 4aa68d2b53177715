#!/usr/bin/env python3
"""Device-resident ViT-B/16 fp32 forward of a SMALL batch against vit_engine_options.gemm_tile (every fp32 GEMM of the engine on one
tile code; 0 = the dispatcher's own choice per launch): does a poorly filled launch want smaller tiles than the dispatcher picks?
GPU box only.    python3 tools/small_batch_tiles.py [images ...]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vision-transformer-opencl_amd")
B = importlib.import_module("vision-transformer-opencl_amd.binding")
cfg = pkg.VIT_B16
sizes = [int(a) for a in sys.argv[1:]] or [12, 40, 64]
W = pkg.synth.make_weights(cfg, 1234)
imgs = pkg.synth.make_images(cfg, max(sizes), 99)
d_img, d_out = B.DeviceArray.from_numpy(imgs), B.DeviceArray((max(sizes), cfg.num_classes))
engines = {}
for t in (0, 7, 8, 9, 10, 11):
    e = B.Engine(cfg, max_batch=max(sizes), gemm_tile=t, profile=True)
    e.load_weights(W)
    engines[t] = e
for n in sizes:
    for t, e in engines.items():
        ts = []
        for r in range(4):
            e.sync(); t0 = time.perf_counter()
            e.forward_device(d_img.ptr, n, d_out.ptr); e.sync()
            if r: ts.append(1e3 * (time.perf_counter() - t0))
        e.reset_stage_times(); e.forward_device(d_img.ptr, n, d_out.ptr); e.sync()
        st = e.stage_times()["stages"]
        print(json.dumps({"images": n, "gemm_tile": t, "ms": round(min(ts), 3),
                          "stage_ms": {k: round(v["ms"], 3) for k, v in st.items() if k in ("qkv", "outproj", "fc1", "fc2", "attn")}}))
