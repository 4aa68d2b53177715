"""The CPU restatement (oracle/vit_cpu_ref.c) against the reference's own ViT_seq.c, bit for bit.

oracle/_ref/libvitseq_ref.so is compiled from /root/reference/ViT_seq.c by `make -C oracle ref`
(build container only; the prebuilt file travels to the GPU box).  Where it is absent these
tests skip and the golden vectors (test_oracle_golden.py), which the same reference build
emitted, carry the pin.
"""
import numpy as np
import pytest

from vit_amd import synth
from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_reference(), reason="oracle/_ref/libvitseq_ref.so not built here")

T, D, H = 197, 768, 3072


def u(k, shape, a, seed=31337):
    return synth.uniform(seed, k, int(np.prod(shape)), -a, a).reshape(shape)


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture(scope="module")
def ref():
    return po.Reference()


def test_layer_norm(ref, oracle):
    x = u(0, (T, D), 3.0) + 0.7
    g, b = synth.uniform(31337, 1, D, 0.2, 1.8), u(2, (D,), 0.5)
    assert same_bits(ref.layer_norm(x, g, b), oracle.layer_norm(x, g, b))


def test_linear_layer(ref, oracle):
    x, w, b = u(3, (T, D), 1.0), u(4, (256, D), 0.05), u(5, (256,), 0.1)
    assert same_bits(ref.linear(x, w, b), oracle.linear(x, w, b))


def test_gelu_and_softmax(ref, oracle):
    x = u(6, (4096,), 6.0)
    assert same_bits(ref.gelu(x), oracle.gelu(x))
    lg = u(7, (1000,), 8.0)
    assert same_bits(ref.softmax(lg), oracle.softmax(lg))


def test_multihead_attn(ref, oracle):
    x = u(8, (T, D), 1.0)
    in_w, in_b = u(9, (3 * D, D), 0.06), u(10, (3 * D,), 0.1)
    out_w, out_b = u(11, (D, D), 0.05), u(12, (D,), 0.1)
    assert same_bits(ref.multihead_attn(x, in_w, in_b, out_w, out_b), oracle.multihead_attn(x, in_w, in_b, out_w, out_b, 12))


def test_mlp_block(ref, oracle):
    x = u(13, (T, D), 1.0)
    w1, b1, w2, b2 = u(14, (H, D), 0.05), u(15, (H,), 0.1), u(16, (D, H), 0.03), u(17, (D,), 0.1)
    assert same_bits(ref.mlp_block(x, w1, b1, w2, b2), oracle.mlp_block(x, w1, b1, w2, b2))


def test_embedding_and_encoder(ref, oracle):
    cfg = synth.VIT_B16
    W = [synth.make_weight(cfg, i, 77) for i in range(16)]
    img = synth.make_images(cfg, 1, 78)[0]
    e_ref = ref.embed(img, W)
    assert same_bits(e_ref, oracle.embed(po.Config(), img, W))
    assert same_bits(ref.encoder(e_ref, W[4:16]), oracle.encoder(e_ref, W[4:16], 12))


def test_whole_forward_one_image(ref, oracle):
    """ViT_seq(ImageData*, Network*, float**) itself (13 s on one core) == vitref_forward_image."""
    cfg = synth.VIT_B16
    W = synth.make_weights(cfg, 4321)
    img = synth.make_images(cfg, 1, 8765)
    p_ref = ref.vit_seq(list(img), W)[0]
    p_my, _, _ = oracle.forward_image(po.Config(), img[0], W)
    assert same_bits(p_ref, p_my)
    assert abs(float(p_ref.sum()) - 1.0) < 1e-5
