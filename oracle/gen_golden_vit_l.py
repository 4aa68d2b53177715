"""Generate tests/golden/vit_l16_384_e2e.npz: ViT-L/16 at 384x384 (BASELINE.json configs[4]) through the CPU oracle.

    python oracle/gen_golden_vit_l.py          (about a minute on 8 cores)

The reference has NO ViT-L: its dimensions are macros fixed at ViT-B/16 (/root/reference/ViT_seq.c:10-21), so its own
`ViT_seq()` cannot produce these numbers.  They come from oracle/vit_cpu_ref.c, the restatement whose macros became a runtime
config and which is pinned bit for bit against the compiled reference at ViT-B/16 (tests/test_oracle_vs_reference.py,
tests/golden/vit_b16_e2e.npz).  Parity status of this fixture: oracle-pinned at B/16, restated here -- the arithmetic per
operation is the reference's (same C functions), only the loop bounds differ (24 layers, D = 1024, 16 heads, H = 4096, 577 tokens).

What is recorded (inputs are NOT stored: synth.make_weights(VIT_L16_384, weight_seed) / synth.make_images(..., image_seed)
regenerate them; the C twin of that generator is checked bit for bit in tests/test_host_io.py):

    weight_seed, image_seed, n_images
    probs   [n][1000]   softmax outputs of the oracle
    logits  [n][1000]
    cls_rows [n][25][1024]  class-token row after the embedding and after each of the 24 encoders (a tap for debugging)

The sums run in sequential fp32 order whatever the thread count (OpenMP splits rows, never a dot product), so the file does not
depend on the machine it was written on; tests/test_oracle_golden.py re-runs image 0 and compares bit for bit.

Test infrastructure only -- nothing here is imported by the product.
"""
from __future__ import annotations

import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

pkg = importlib.import_module("vision-transformer-opencl_amd")
synth = pkg.synth
OUT = os.path.join(ROOT, "tests", "golden", "vit_l16_384_e2e.npz")

WEIGHT_SEED, IMAGE_SEED, N_IMAGES = 1234, 99, 2   # the seeds bench.py uses for its weights and for the b16 golden rows


def main():
    cfg = synth.VIT_L16_384
    ocfg = po.Config(cfg.img_size, cfg.patch_size, cfg.in_chans, cfg.num_classes, cfg.embed_dim, cfg.depth, cfg.num_heads,
                     cfg.hidden_dim)
    po.set_threads(min(16, os.cpu_count() or 1))
    W = synth.make_weights(cfg, WEIGHT_SEED)
    imgs = synth.make_images(cfg, N_IMAGES, IMAGE_SEED)
    probs = np.empty((N_IMAGES, cfg.num_classes), np.float32)
    logits = np.empty_like(probs)
    cls_rows = np.empty((N_IMAGES, cfg.depth + 1, cfg.embed_dim), np.float32)
    for i in range(N_IMAGES):
        t = time.time()
        probs[i], logits[i], stages = po.forward_image(ocfg, imgs[i], W, want_stages=True)
        cls_rows[i] = stages[:, 0, :]
        print(f"image {i}: {time.time() - t:.1f} s, top-1 {int(probs[i].argmax())} p = {float(probs[i].max()):.4f}")
    np.savez_compressed(OUT, weight_seed=WEIGHT_SEED, image_seed=IMAGE_SEED, n_images=N_IMAGES, probs=probs, logits=logits,
                        cls_rows=cls_rows)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
