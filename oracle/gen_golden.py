"""Generate tests/golden/*.npz with the REFERENCE ITSELF (oracle/_ref/libvitseq_ref.so).

Run in the build container only (needs /root/reference to have been compiled by
`make -C oracle ref`):

    python oracle/gen_golden.py

What is recorded (inputs are NOT stored: they are regenerated from the seeds by
vision-transformer-opencl_amd/synth.py, whose C twin is checked bit-for-bit in the tests):

  vit_b16_e2e.npz
      weight_seed, image_seed, n_images
      probs   [n][1000]  -- written by the reference's ViT_seq(ImageData*, Network*, float**)
      logits  [n][1000]  -- from the reference's own functions called stage by stage
                            (Conv2d ... Encoder x12 ... layer_norm, linear_layer), whose Softmax
                            is asserted bit-identical to `probs`
      stage_cls [n][14][768]  -- class-token row after the embedding, after each encoder, after encoder_ln
      stage_sum [n][13]       -- float64 sum of every stage tensor (embedding + 12 encoders)
  ops_b16.npz
      small slices of the reference's per-op outputs on seeded inputs (layer_norm, linear_layer,
      multihead_attn, mlp_block, Softmax, gelu) for op-level parity without the reference.

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
OUT = os.path.join(ROOT, "tests", "golden")

WEIGHT_SEED, IMAGE_SEED, N_IMAGES = 1234, 99, 2


def e2e():
    cfg = synth.VIT_B16
    ref = po.Reference()
    W = synth.make_weights(cfg, WEIGHT_SEED)
    imgs = synth.make_images(cfg, N_IMAGES, IMAGE_SEED)
    t = time.time()
    probs = ref.vit_seq(list(imgs), W)
    print(f"ViT_seq on {N_IMAGES} images: {time.time() - t:.1f} s")
    logits = np.empty((N_IMAGES, 1000), np.float32)
    stage_cls = np.empty((N_IMAGES, 14, 768), np.float32)
    stage_sum = np.empty((N_IMAGES, 13), np.float64)
    for i in range(N_IMAGES):
        x = ref.embed(imgs[i], W)
        stage_cls[i, 0] = x[0]
        stage_sum[i, 0] = x.astype(np.float64).sum()
        for l in range(12):
            x = ref.encoder(x, W[4 + 12 * l: 16 + 12 * l])
            stage_cls[i, l + 1] = x[0]
            stage_sum[i, l + 1] = x.astype(np.float64).sum()
        y = ref.layer_norm(x, W[148], W[149])
        stage_cls[i, 13] = y[0]
        lg = ref.linear(np.ascontiguousarray(y[:1]), W[150], W[151])[0]
        logits[i] = lg
        p = ref.softmax(lg)
        assert np.array_equal(p.view(np.uint32), probs[i].view(np.uint32)), "stage-wise != ViT_seq"
    print("top-1:", probs.argmax(1), probs.max(1))
    np.savez_compressed(os.path.join(OUT, "vit_b16_e2e.npz"), weight_seed=WEIGHT_SEED, image_seed=IMAGE_SEED,
                        n_images=N_IMAGES, probs=probs, logits=logits, stage_cls=stage_cls, stage_sum=stage_sum)


def ops():
    """Per-op goldens: inputs from synth.uniform(seed=4242, index=k), outputs sliced to stay small."""
    ref = po.Reference()
    S = 4242
    u = lambda k, n, a: synth.uniform(S, k, n, -a, a)  # noqa: E731
    T, D, H = 197, 768, 3072
    out = {"seed": S}
    x = u(0, T * D, 2.0).reshape(T, D)
    g, b = synth.uniform(S, 1, D, 0.5, 1.5), u(2, D, 0.5)
    out["layer_norm_rows0_8"] = ref.layer_norm(x, g, b)[:8]
    w, bias = u(3, D * D, 0.05).reshape(D, D), u(4, D, 0.1)
    out["linear_rows0_4"] = ref.linear(x, w, bias)[:4]
    in_w, in_b = u(5, 3 * D * D, 0.05).reshape(3 * D, D), u(6, 3 * D, 0.1)
    xs = u(7, T * D, 1.0).reshape(T, D)
    mha = ref.multihead_attn(xs, in_w, in_b, w, bias)
    out["mha_rows"] = mha[[0, 1, 100, 196]]
    w1, b1 = u(8, H * D, 0.05).reshape(H, D), u(9, H, 0.1)
    w2, b2 = u(10, D * H, 0.03).reshape(D, H), u(11, D, 0.1)
    out["mlp_rows"] = ref.mlp_block(xs, w1, b1, w2, b2)[[0, 1, 100, 196]]
    lg = u(12, 1000, 6.0)
    out["softmax"] = ref.softmax(lg)
    gx = u(13, 4096, 4.0)
    out["gelu"] = ref.gelu(gx)
    np.savez_compressed(os.path.join(OUT, "ops_b16.npz"), **out)


if __name__ == "__main__":
    if not po.have_reference():
        sys.exit("oracle/_ref/libvitseq_ref.so is missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    po.set_threads(1)
    ops()
    e2e()
    print("wrote", sorted(os.listdir(OUT)))
