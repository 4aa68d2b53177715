#!/usr/bin/env python3
"""Device-resident bf16 forward of ViT-B/16: img/s against the number of images per forward_device call, 2,048 images in all
(2048 / n calls back to back on one stream).  The question: at 2,048 images per call no activation of a layer survives in the
256-MB Infinity Cache between the launch that writes it and the one that reads it (x 1.24 GB, qkv 1.86 GB, h 2.48 GB) and the GEMMs
run at 3-4.6 TB/s of HBM traffic; do smaller calls, whose layer working set fits, run faster in spite of their emptier tile walks?
GPU box only.     python3 tools/batch_time_sweep_bf16.py [b16|l16_384]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vision-transformer-opencl_amd")
B = importlib.import_module("vision-transformer-opencl_amd.binding")
model = sys.argv[1] if len(sys.argv) > 1 else "b16"
cfg, total, sizes = (pkg.VIT_B16, 2048, [32, 48, 64, 96, 128, 192, 256, 512, 1024, 2048]) if model == "b16" else (pkg.VIT_L16_384, 1024, [16, 32, 64, 128, 256, 512, 1024])
e = B.Engine(cfg, max_batch=total, dtype="bf16")
e.load_weights(pkg.synth.make_weights(cfg, 1234))
nimg = 256
imgs = pkg.synth.make_images(cfg, nimg, 99)
d_img, d_out = B.DeviceArray.from_numpy(np.concatenate([imgs] * (total // nimg))), B.DeviceArray((total, cfg.num_classes))
per = cfg.in_chans * cfg.img_size * cfg.img_size * 4
res = {}
for rnd in range(3):
    for n in sizes:
        e.sync(); t0 = time.perf_counter()
        for off in range(0, total, n):
            e.forward_device(d_img.ptr + off * per, min(n, total - off), d_out.ptr + off * cfg.num_classes * 4)   # never past the 2,048 images
        e.sync()
        if rnd: res.setdefault(n, []).append(1e3 * (time.perf_counter() - t0))
for n, ts in res.items():
    print(json.dumps({"images_per_call": n, "calls": total // n, "ms_for_all": round(min(ts), 2), "img_per_s": round(total / min(ts) * 1e3, 1)}), flush=True)
