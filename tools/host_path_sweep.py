#!/usr/bin/env python3
"""vit_engine_forward_host (what ViT_opencl() runs underneath, ViT_opencl.c:785-883) at the metric batch: ms per call for several
first-piece sizes (vit_engine_options.host_first_piece), interleaved, against the device-resident forward of the same engine.
GPU box only.    python3 tools/host_path_sweep.py [images] [first first ...]        (0 = the engine's own choice)"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("vision-transformer-opencl_amd")
B = importlib.import_module("vision-transformer-opencl_amd.binding")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
firsts = [int(a) for a in sys.argv[2:]] if len(sys.argv) > 2 else [0, 32, 48, 64, 80, 96, 128]
cfg = pkg.VIT_B16
W = pkg.synth.make_weights(cfg, 1234)
imgs = pkg.synth.make_images(cfg, n, 99)
rows = [np.ascontiguousarray(imgs[i]) for i in range(n)]                  # separately addressed images, pageable
probs = np.empty((n, cfg.num_classes), np.float32)
in_ptrs = (B.f32p * n)(*[r.ctypes.data_as(B.f32p) for r in rows])
out_ptrs = (B.f32p * n)(*[probs[i].ctypes.data_as(B.f32p) for i in range(n)])
engines = {}
for f in firsts:
    e = B.Engine(cfg, max_batch=256, host_first_piece=f)
    e.load_weights(W)
    engines[f] = e
ref = None
res = {f: [] for f in firsts}
for rnd in range(5):
    for f, e in engines.items():
        t0 = time.perf_counter()
        rc = B.lib().vit_engine_forward_host(e._h, in_ptrs, n, out_ptrs)
        dt = time.perf_counter() - t0
        assert rc == 0
        if ref is None:
            ref = probs.copy()
        assert np.array_equal(probs, ref), f                                # rows bit-identical whatever the cut
        if rnd > 0:
            res[f].append(1e3 * dt)
# device-resident time of the same batch (one engine, batch already in HBM)
e = engines[firsts[0]]
d_img, d_out = B.DeviceArray.from_numpy(imgs), B.DeviceArray((n, cfg.num_classes))
dev = []
for rnd in range(5):
    e.sync(); t0 = time.perf_counter()
    e.forward_device(d_img.ptr, n, d_out.ptr); e.sync()
    if rnd > 0:
        dev.append(1e3 * (time.perf_counter() - t0))
dms = min(dev)
print(json.dumps({"images": n, "device_resident_ms": round(dms, 3)}))
for f in firsts:
    best = min(res[f])
    print(json.dumps({"host_first_piece": f, "ms": [round(x, 2) for x in res[f]], "best_ms": round(best, 3), "share_of_device_rate": round(dms / best, 4)}))
