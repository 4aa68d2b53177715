"""Real-weight vectors (tests/golden/real_weights_b16.npz, written by oracle/gen_golden_real.py from the reference's own
functions on the reference's own tensors): the CPU restatement must reproduce them bit for bit, and -- where the
reference tree is mounted -- this repo's load_weights() must reproduce the 116 real blobs exactly.

End-to-end real-weight parity stays UNPINNED: Data/input-100.bin and the 36 in_proj / fc1 / fc2 weight blobs are absent
upstream, so answer_result.txt cannot be reproduced by anyone from this tree.
"""
import glob
import os

import numpy as np
import pytest

from vit_amd import binding as B
from vit_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "real_weights_b16.npz")
NET = "/root/reference/Network"
T, D = 197, 768


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32),
                          np.ascontiguousarray(b, np.float32).view(np.uint32))


def outlier_cols(x, cols, scale):
    x = x.copy()
    x[:, cols] *= np.float32(scale)
    return x


def activation_inputs(seed, layer):
    """Same streams as oracle/gen_golden_real.py."""
    a = synth.uniform(seed, 100 + layer, T * D, -1.0, 1.0).reshape(T, D)
    x = synth.uniform(seed, 200 + layer, T * D, -2.0, 2.0).reshape(T, D)
    return outlier_cols(a, [7, 300, 301, 640], 12.0), outlier_cols(x, [5, 381, 759], 30.0)


def head_input(seed):
    return outlier_cols(synth.uniform(seed, 300, T * D, -2.0, 2.0).reshape(T, D), [5, 381, 759], 30.0)


@pytest.mark.skipif(not os.path.isdir(NET), reason="the reference tree is only mounted in the build container")
def test_loader_reproduces_the_116_real_blobs_and_the_fixture_holds_them():
    cfg = synth.VIT_B16
    W = B.load_weight_dir(NET, cfg.n_weights)
    present = [i for i, w in enumerate(W) if w is not None]
    assert len(present) == 116
    shapes = cfg.weight_shapes()
    L = B.lib()
    cc = B.CConfig.of(cfg)
    import ctypes as C
    for i in present:
        raw = np.fromfile(glob.glob(os.path.join(NET, f"Weight_{i}_*.bin"))[0], "<f4")
        assert raw.size == L.vit_config_weight_size(C.byref(cc), i) == int(np.prod(shapes[i]))
        assert same_bits(W[i], synth.round6(raw))                 # Network.c:184-187: roundf(w * 1e6f) / 1e6f
    assert all((i - 4) % 12 in (2, 8, 10) for i in set(range(152)) - set(present))   # in_proj / fc1 / fc2 weights
    g = np.load(GOLD)
    assert list(g["present"]) == present
    for i in range(6):
        assert same_bits(g[f"w{i}"].ravel(), W[i])
    for l in g["outproj_layers"]:
        b = 4 + 12 * int(l)
        assert same_bits(g[f"outproj_w_{l}"].ravel(), W[b + 4]) and same_bits(g[f"ln2_w_{l}"], W[b + 6])
    assert same_bits(g["head_w"].ravel(), W[150][:int(g["head_classes"]) * D]) and same_bits(g["ln_w"], W[148])


def test_restatement_matches_real_weight_vectors(oracle):
    from conftest import oracle_config
    g = np.load(GOLD)
    seed, rows = int(g["seed"]), list(g["rows"])
    cfg = synth.VIT_B16
    image = synth.make_images(cfg, 1, seed)[0]
    w = [g[f"w{i}"] for i in range(6)]
    x = oracle.embed(oracle_config(cfg), image, w)
    assert same_bits(x[rows], g["embed_rows"]) and x.astype(np.float64).sum() == float(g["embed_sum"])
    y = oracle.layer_norm(x, w[4], w[5])
    assert same_bits(y[rows], g["ln1_rows"]) and y.astype(np.float64).sum() == float(g["ln1_sum"])
    for l in g["outproj_layers"]:
        a, xres = activation_inputs(seed, int(l))
        r = xres + oracle.linear(a, g[f"outproj_w_{l}"], g[f"outproj_b_{l}"])
        assert same_bits(r[rows], g[f"resid_rows_{l}"]) and r.astype(np.float64).sum() == float(g[f"resid_sum_{l}"])
        assert same_bits(oracle.layer_norm(r, g[f"ln2_w_{l}"], g[f"ln2_b_{l}"])[rows], g[f"ln2_rows_{l}"])
    z = oracle.layer_norm(head_input(seed), g["ln_w"], g["ln_b"])[:8]
    assert same_bits(z, g["final_ln"])
    assert same_bits(oracle.linear(np.ascontiguousarray(z), g["head_w"], g["head_b"]), g["head_logits"])
