"""Generate tests/golden/real_weights_b16.npz: the reference's REAL tensors through the REFERENCE's own functions.

Run in the build container only (needs /root/reference/Network and oracle/_ref/libvitseq_ref.so):

    python oracle/gen_golden_real.py

The reference ships 116 of its 152 weight blobs (36 GEMM weights are absent upstream), so an end-to-end
real-weight run is impossible; these are the partial real-weight checks SURVEY.md 8(c) lists as possible:

  embed   real class_token, conv_proj, pos_embedding, layer-0 ln_1 (W0-W5):
            Conv2d -> flatten_transpose -> class_token -> pos_emb -> layer_norm   (ViT_seq.c:25-121,356-362)
          on one seeded synthetic image; rows ROWS of both stages are stored.
  outproj real out_proj weight/bias + ln_2 weight/bias of the two layers whose out_proj has the largest |w|:
            r = x + linear_layer(a)  (ViT_seq.c:219-227,286-288),  layer_norm(r)  (ln_2, ViT_seq.c:291)
          on seeded inputs shaped like real activations: a ~ U(-1,1) with a few x12 columns, the residual stream x
          ~ U(-2,2) with three "massive" channels (x30), as ViT-B's residual stream has.
  head    real encoder.ln + heads.head (W148-W151, first HEAD_CLASSES classes):
            layer_norm -> linear_layer  (ViT_seq.c:429-435) on 8 seeded class rows (rows 0-7 of a 197-row seeded tensor).

Inputs are regenerated from seeds by the tests (synth.uniform); the real tensors that are needed ARE stored (after
the loader's 1e-6 rounding, Network.c:184-187, applied by THIS repo's load_weights -- asserted equal to round6(raw
file) here), since the GPU box has no /root/reference.  Data only: tensors and outputs, no reference source.
"""
from __future__ import annotations

import glob
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

pkg = importlib.import_module("vision-transformer-opencl_amd")
binding = importlib.import_module("vision-transformer-opencl_amd.binding")
synth = pkg.synth
NET = "/root/reference/Network"
OUT = os.path.join(ROOT, "tests", "golden", "real_weights_b16.npz")

SEED = 20260
ROWS = [0, 1, 2, 3, 50, 99, 100, 150, 195, 196]
HEAD_CLASSES = 256
T, D = 197, 768


def outlier_cols(x: np.ndarray, cols, scale: float) -> np.ndarray:
    x = x.copy()
    x[:, cols] *= np.float32(scale)
    return x


def activation_inputs(layer: int):
    """(a, x): attention output and residual stream, seeded per layer (the tests regenerate them)."""
    a = synth.uniform(SEED, 100 + layer, T * D, -1.0, 1.0).reshape(T, D)
    x = synth.uniform(SEED, 200 + layer, T * D, -2.0, 2.0).reshape(T, D)
    return outlier_cols(a, [7, 300, 301, 640], 12.0), outlier_cols(x, [5, 381, 759], 30.0)


def main():
    if not po.have_reference() or not os.path.isdir(NET):
        sys.exit("needs /root/reference/Network and oracle/_ref/libvitseq_ref.so (make -C oracle ref)")
    cfg = synth.VIT_B16
    W = binding.load_weight_dir(NET, cfg.n_weights)          # THIS repo's loader on the real files
    present = [i for i, w in enumerate(W) if w is not None]
    assert len(present) == 116, len(present)
    shapes = cfg.weight_shapes()
    for i in present:                                        # size per the index map, bytes = round6(raw)
        f = glob.glob(os.path.join(NET, f"Weight_{i}_*.bin"))
        assert len(f) == 1
        raw = np.fromfile(f[0], "<f4")
        assert raw.size == int(np.prod(shapes[i])) == W[i].size, (i, raw.size)
        assert np.array_equal(W[i], synth.round6(raw)), i
        assert np.array_equal(W[i], po.round_weights(raw)), i
    missing = sorted(set(range(cfg.n_weights)) - set(present))
    assert all((i - 4) % 12 in (2, 8, 10) for i in missing)  # exactly the in_proj / fc1 / fc2 weights
    ref = po.Reference()
    out = {"seed": SEED, "rows": np.array(ROWS), "head_classes": HEAD_CLASSES, "present": np.array(present)}

    # ---- embed + ln_1 of layer 0 -------------------------------------------------------------------
    image = synth.make_images(cfg, 1, SEED)[0]
    w = [W[i].reshape(shapes[i]) for i in range(6)]
    x = ref.embed(image, w)
    y = ref.layer_norm(x, w[4], w[5])
    for i in range(6):
        out[f"w{i}"] = w[i]
    out["embed_rows"], out["ln1_rows"] = x[ROWS], y[ROWS]
    out["embed_sum"], out["ln1_sum"] = x.astype(np.float64).sum(), y.astype(np.float64).sum()

    # ---- out_proj + residual + ln_2, the two layers with the largest |w| -------------------------
    amax = {l: float(np.abs(W[4 + 12 * l + 4]).max()) for l in range(12)}
    layers = sorted(sorted(amax, key=amax.get)[-2:])
    out["outproj_layers"] = np.array(layers)
    for l in layers:
        b = 4 + 12 * l
        ow, ob = W[b + 4].reshape(D, D), W[b + 5]
        g2, b2 = W[b + 6], W[b + 7]
        a, xres = activation_inputs(l)
        r = xres + ref.linear(a, ow, ob)                     # ViT_seq.c:286-288: element-wise fp32 add
        z = ref.layer_norm(r, g2, b2)
        out[f"outproj_w_{l}"], out[f"outproj_b_{l}"] = ow, ob
        out[f"ln2_w_{l}"], out[f"ln2_b_{l}"] = g2, b2
        out[f"resid_rows_{l}"], out[f"ln2_rows_{l}"] = r[ROWS], z[ROWS]
        out[f"resid_sum_{l}"] = r.astype(np.float64).sum()
        print(f"layer {l}: out_proj |w|max {amax[l]:.3f}, ln_2.weight std {g2.std():.3f}, |r|max {np.abs(r).max():.1f}")

    # ---- encoder.ln + heads.head -------------------------------------------------------------------
    # (the reference's layer_norm always normalises 197 rows, ViT_seq.c:103-121: feed it 197, keep the first 8)
    xf = outlier_cols(synth.uniform(SEED, 300, T * D, -2.0, 2.0).reshape(T, D), [5, 381, 759], 30.0)
    lw, lb = W[148], W[149]
    hw, hb = W[150].reshape(1000, D)[:HEAD_CLASSES].copy(), W[151][:HEAD_CLASSES].copy()
    zf = np.ascontiguousarray(ref.layer_norm(xf, lw, lb)[:8])
    out["ln_w"], out["ln_b"], out["head_w"], out["head_b"] = lw, lb, hw, hb
    out["final_ln"], out["head_logits"] = zf, ref.linear(zf, hw, hb)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) / 1e6, "MB")


if __name__ == "__main__":
    po.set_threads(1)
    main()
