#!/usr/bin/env python3
"""Device-resident forward time of ViT-B/16 fp32 against the number of images in the call (8 .. 256): where the tile walks'
last rounds are full -- the input of vit_engine_forward_host's choice of pieces.  GPU box only."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vision-transformer-opencl_amd")
B = importlib.import_module("vision-transformer-opencl_amd.binding")
cfg = pkg.VIT_B16
e = B.Engine(cfg, max_batch=256)
e.load_weights(pkg.synth.make_weights(cfg, 1234))
imgs = pkg.synth.make_images(cfg, 256, 99)
d_img, d_out = B.DeviceArray.from_numpy(imgs), B.DeviceArray((256, cfg.num_classes))
out = {}
for n in list(range(8, 257, 8)):
    ts = []
    for r in range(4):
        e.sync(); t0 = time.perf_counter()
        e.forward_device(d_img.ptr, n, d_out.ptr); e.sync()
        if r: ts.append(1e3 * (time.perf_counter() - t0))
    out[n] = min(ts)
t256 = out[256]
for n, t in out.items():
    print(json.dumps({"images": n, "ms": round(t, 3), "ms_per_image": round(t / n, 4), "vs_256_rate": round((t256 / 256) / (t / n), 4)}))
best = min(((out[a] + out[256 - a], a) for a in out if a < 256 and 256 - a in out))
print(json.dumps({"best_two_piece_split_of_256": best[1], "sum_ms": round(best[0], 3), "one_piece_ms": round(t256, 3)}))
