"""bf16 variant (BASELINE.json configs[2]): op-level checks of the bf16 MFMA GEMM.

Inputs are rounded to bf16 on the host first, so the oracle (fp32 arithmetic on the SAME values) differs
from the kernel only by fp32 accumulation order; bf16 outputs are additionally rounded once
(relative 2^-8).  Tolerances are written where they are used.
"""
import numpy as np
import pytest

from vit_amd import binding as B
from vit_amd import synth

pytestmark = pytest.mark.gpu


def u(k, shape, a, seed=999):
    return synth.uniform(seed, k, int(np.prod(shape)), -a, a).reshape(shape)


def test_f32_to_bf16_matches_round_to_nearest_even():
    x = np.concatenate([u(0, (4096,), 3.0), np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 65504.0, 1e-30, -1e30], np.float32)])
    x = x[: (x.size // 4) * 4]
    assert np.array_equal(B.f32_to_bf16_device(x), B.to_bf16_bits(x))


@pytest.fixture(params=[1, 2], ids=["two-stage", "ping-pong"])
def variant(request, monkeypatch):
    """Both bf16 GEMM kernels behind vithip_gemm_bf16 (0 = auto picks ping-pong whenever K >= 128; the two-stage kernel is the
    fallback for K < 128) -- selected per call (vithip_gemm_bf16_args.variant): the library has no process-wide tuning state."""
    v = request.param
    plain = B.gemm_bf16
    monkeypatch.setattr(B, "gemm_bf16", lambda *a, **k: plain(*a, variant=v, **k))
    yield v


def test_gemm_bf16_identity_asymmetric_exact(variant):
    K = 256
    A = B.to_bf16_bits(np.eye(K, dtype=np.float32))
    Wf = B.from_bf16_bits(B.to_bf16_bits(u(1, (256, K), 1.0)))      # values exactly representable in bf16
    got = B.gemm_bf16(A, B.to_bf16_bits(Wf), np.zeros(256, np.float32), residual=np.zeros((K, 256), np.float32),
                      epilogue=B.BF16_EPI_F32_RESIDUAL)
    assert np.array_equal(got, Wf.T.copy())


@pytest.mark.parametrize("M,N,K", [(512, 768, 768), (197, 2304, 768), (300, 100, 64), (1000, 768, 3072), (5, 12, 128)])
def test_gemm_bf16_residual_fp32_out(oracle, variant, M, N, K):
    if variant >= 2 and K < 128:
        pytest.skip("ping-pong kernel needs two K steps; auto mode uses the two-stage kernel here")
    Ab, Wb = B.to_bf16_bits(u(2, (M, K), 1.0)), B.to_bf16_bits(u(3, (N, K), 0.05))
    b, R = u(4, (N,), 0.1), u(5, (M, N), 2.0)
    ref = R + oracle.linear(B.from_bf16_bits(Ab), B.from_bf16_bits(Wb), b)
    got = B.gemm_bf16(Ab, Wb, b, residual=R, epilogue=B.BF16_EPI_F32_RESIDUAL)
    assert float(np.abs(got - ref).max()) <= 2e-5 * float(np.abs(ref).max())   # fp32 accumulation order only


@pytest.mark.parametrize("epi", [B.BF16_EPI_BF16, B.BF16_EPI_BF16_GELU])
def test_gemm_bf16_bf16_out(oracle, variant, epi):
    M, N, K = 515, 3072, 768
    Ab, Wb, b = B.to_bf16_bits(u(6, (M, K), 1.0)), B.to_bf16_bits(u(7, (N, K), 0.08)), u(8, (N,), 0.1)
    ref = oracle.linear(B.from_bf16_bits(Ab), B.from_bf16_bits(Wb), b)
    if epi == B.BF16_EPI_BF16_GELU:
        ref = oracle.gelu(ref)
    got = B.from_bf16_bits(B.gemm_bf16(Ab, Wb, b, epilogue=epi))
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 1e-5).all(), float(err.max())       # one bf16 rounding of the result


@pytest.mark.parametrize("epi", [B.BF16_EPI_BF16, B.BF16_EPI_BF16_GELU, B.BF16_EPI_F32_RESIDUAL])
@pytest.mark.parametrize("M,N,K", [(256 * 37 + 19, 2048 + 4, 192), (256 * 80, 1024, 128), (9000, 768, 448)])
def test_gemm_bf16_persistent_many_tiles(variant, epi, M, N, K):
    """More tiles than workgroups (the persistent loop crosses tile boundaries with the load pipeline running),
    odd and even K-step counts, ragged last tile row and a ragged (N % 8 == 4) last tile column.  Too large for
    the scalar oracle, so the reference is a float64 matmul of the same bf16-exact operands."""
    Ab, Wb, b = B.to_bf16_bits(u(9, (M, K), 1.0)), B.to_bf16_bits(u(10, (N, K), 0.08)), u(11, (N,), 0.1)
    ref = B.from_bf16_bits(Ab).astype(np.float64) @ B.from_bf16_bits(Wb).astype(np.float64).T + b
    if epi == B.BF16_EPI_F32_RESIDUAL:
        R = u(12, (M, N), 2.0)
        got = B.gemm_bf16(Ab, Wb, b, residual=R, epilogue=epi)
        assert float(np.abs(got - (ref + R)).max()) <= 2e-5 * float(np.abs(ref + R).max())
        return
    if epi == B.BF16_EPI_BF16_GELU:
        from scipy.special import erf
        ref = 0.5 * ref * (1.0 + erf(ref / np.sqrt(2.0)))
    got = B.from_bf16_bits(B.gemm_bf16(Ab, Wb, b, epilogue=epi))
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 1e-5).all(), float(err.max())


@pytest.mark.parametrize("implicit", [False, True], ids=["two-pass", "implicit-gemm"])
@pytest.mark.parametrize("cfg,n", [(synth.VIT_B16, 3), (synth.VIT_SMALL, 5)])
def test_patch_embed_bf16(oracle, cfg, n, implicit):
    """Patch embedding on the bf16 pipe vs the oracle's conv/flatten/class-token/pos_emb on the SAME bf16-rounded
    pixels and conv weights: only the fp32 accumulation order differs."""
    from conftest import oracle_config
    W = [synth.make_weight(cfg, i, 5) for i in range(4)]
    imgs = synth.make_images(cfg, n, 6)
    got = B.patch_embed_bf16(cfg, imgs, W[1], W[2], W[0], W[3], implicit=implicit)
    Wr = list(W)
    Wr[1] = B.from_bf16_bits(B.to_bf16_bits(W[1])).reshape(W[1].shape)
    imgs_r = B.from_bf16_bits(B.to_bf16_bits(imgs)).reshape(imgs.shape)
    ocfg = oracle_config(cfg)
    for i in range(n):
        ref = oracle.embed(ocfg, imgs_r[i], Wr)
        assert float(np.abs(got[i] - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


def test_layernorm_bf16_out(oracle):
    x = u(10, (197, 768), 3.0) + 0.5
    g, b = synth.uniform(999, 11, 768, 0.5, 1.5), u(12, (768,), 0.5)
    ref = oracle.layer_norm(x, g, b)
    got = B.from_bf16_bits(B.layernorm_bf16out(x, g, b))
    assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 1e-5).all()          # one bf16 rounding


@pytest.mark.parametrize("mfma", [1, 0, 2], ids=["bf16-mfma", "fp32-mfma", "bf16-mfma-qscaled"])
@pytest.mark.parametrize("n,T,heads", [(2, 197, 12), (1, 50, 2), (1, 224, 1), (1, 33, 1), (1, 577, 2), (2, 300, 1), (1, 225, 1), (1, 740, 2)])
def test_attention_bf16_io(oracle, n, T, heads, mfma):
    """bf16 Q/K/V in, bf16 out.  mfma=1: both products on bf16 MFMA, P rounded to bf16 (resident kernel up to
    224 tokens, streamed online-softmax kernel up to 704, chunked kernel beyond); mfma=0: fp32 MFMA on the widened values; mfma=2: as 1 with the Q columns
    holding QSCALE * q (what the engine's folded in_proj writes): the streamed kernel then starts its score accumulators at -max."""
    D = heads * 64
    vals = u(13, (n * T, 3 * D), 1.5)
    if mfma == 2:
        vals[:, :D] *= np.float32(B.QSCALE)                                    # the engine's in_proj produces QSCALE * q
    bits = B.to_bf16_bits(vals)
    qkv = B.from_bf16_bits(bits)                                               # the exact values the kernel sees
    got = B.from_bf16_bits(B.attention_bf16io(bits, n, T, heads, f32math=not mfma, q_scaled=mfma == 2)).reshape(n, T, D)
    if mfma == 2:
        qkv[:, :D] /= np.float32(B.QSCALE)                                     # what the reference is to scale by 1/8 itself
    # fp32 inside + one bf16 rounding of the output; with bf16 P add sum_j |dp_j v_j| <~ 2^-9 |v| sqrt(sum p^2)
    slack = 1e-3 if mfma else 2e-5
    for i in range(n):
        blk = qkv[i * T:(i + 1) * T]
        q, k, v = (np.ascontiguousarray(blk[:, j * D:(j + 1) * D]) for j in range(3))
        ref = oracle.attention_core(q, k, v, heads)
        assert (np.abs(got[i] - ref) <= 2.0 ** -8 * np.abs(ref) + slack).all(), float(np.abs(got[i] - ref).max())


@pytest.mark.parametrize("qs", [False, True], ids=["plain-q", "qscaled"])
@pytest.mark.parametrize("T", [577, 300])
def test_attention_bf16_streamed_reference_maximum_moves(oracle, T, qs):
    """The streamed kernel (225..704 tokens) keeps a row's running maximum as a reference that moves only when the scores outgrow
    it by more than 2^8: uniform random scores never do after the first sub-chunk, so this input makes them -- every row's score
    rises along the keys (by 4..15 in the exponent per 64-key sub-chunk, every row at its own rate; the second head's fall instead) -- and
    O, the row sum and the pending P must all be rescaled exactly once each time."""
    heads, n = 2, 1
    D = heads * 64
    rng = np.random.default_rng(5)
    j = np.arange(T, dtype=np.float32)[:, None] / T
    rate = np.linspace(0.3, 1.0, T, dtype=np.float32)[:, None]                   # per query row
    q = np.concatenate([rate * 2.0 * np.ones((T, 64), np.float32), -rate * 2.0 * np.ones((T, 64), np.float32)], axis=1)
    k = np.concatenate([j * 6.0 * np.ones((T, 64), np.float32), j * 6.0 * np.ones((T, 64), np.float32)], axis=1)
    q += rng.uniform(-0.05, 0.05, q.shape).astype(np.float32)
    k += rng.uniform(-0.05, 0.05, k.shape).astype(np.float32)
    v = rng.uniform(-1.5, 1.5, (T, D)).astype(np.float32)
    if qs:   # (the accumulators of a unit may have been started at a reference that the unit before it then moved: corrected by one add)
        q = q * np.float32(B.QSCALE)
    bits = B.to_bf16_bits(np.concatenate([q, k, v], axis=1))
    qkv = B.from_bf16_bits(bits)
    got = B.from_bf16_bits(B.attention_bf16io(bits, n, T, heads, q_scaled=qs)).reshape(T, D)
    if qs:
        qkv[:, :D] /= np.float32(B.QSCALE)
    qq, kk, vv = (np.ascontiguousarray(qkv[:, i * D:(i + 1) * D]) for i in range(3))
    ref = oracle.attention_core(qq, kk, vv, heads)
    s0 = (qq[:, :64] @ kk[:, :64].T) * 0.125 * 1.4426950408889634                 # head 0: rising, head 1: falling
    assert float((s0[:, -1] - s0[:, 63]).min()) > 24.0                            # the exponent really outgrows 2^8 several times
    # P is rounded to bf16 (relative 2^-9) before it multiplies V: |dO| <= 2^-9 sum_j p_j |v_j| / sum_j p_j <= 2^-9 max|v| = 2.9e-3 on
    # these rows, whose weight sits on the last few keys; + the rounding of the output itself
    assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 4e-3).all(), float(np.abs(got - ref).max())


@pytest.mark.parametrize("qs", [False, True], ids=["plain-q", "qscaled"])
@pytest.mark.parametrize("T,heads,n", [(577, 16, 40), (300, 3, 200), (231, 1, 700)])
def test_attention_bf16_streamed_head_switch_is_bit_identical_to_single_head_launches(T, heads, n, qs):
    """The streamed kernel's head switch (round 5): a workgroup that walks SEVERAL (image, head) items retires a query block inside the
    head's last step -- stores, state reset, the next head's Q refill -- and proves with COUNTED vmcnt waits that the refills have
    landed (the count assumes a fixed order of the wave's vector-memory operations).  A launch of one image has at most `heads` items:
    every workgroup takes the first-item path, which waits for everything.  Both must give the same bits for every image, on shapes
    where a wave owns three, two and one query blocks (19, 10, 8 blocks over 8 waves), with 2.5 / 2.3 / 2.7 items per workgroup so
    that first, middle and last heads of a walk all occur; twice, for run-to-run determinism of the walk."""
    D = heads * 64
    rng = np.random.default_rng(11)
    vals = rng.uniform(-1.5, 1.5, (n * T, 3 * D)).astype(np.float32)
    if qs:
        vals[:, :D] *= np.float32(B.QSCALE)
    bits = B.to_bf16_bits(vals)
    whole = B.attention_bf16io(bits, n, T, heads, q_scaled=qs).reshape(n, T, D)
    assert np.array_equal(B.attention_bf16io(bits, n, T, heads, q_scaled=qs).reshape(n, T, D), whole)
    for i in list(range(0, n, max(1, n // 12))) + [n - 1]:                        # a dozen images spread over the walk, and the last
        one = B.attention_bf16io(bits[i * T:(i + 1) * T], 1, T, heads, q_scaled=qs).reshape(T, D)
        assert np.array_equal(whole[i], one), i


@pytest.mark.parametrize("T,heads,n", [(197, 12, 70), (224, 3, 230), (50, 2, 700), (33, 1, 1300)])
def test_attention_bf16_resident_item_walk_is_bit_identical_to_single_item_launches(T, heads, n):
    """The resident kernel's walk (round 5): K and V of the NEXT (image, head) item arrive by LDS-DMA in a second LDS image while the
    current one multiplies, one barrier per item -- every wave's pieces landed, everybody done with the image about to be overwritten.
    A launch of one image has at most `heads` items, one per workgroup: it never changes images.  Launches with 3.3 / 2.7 / 2.7 / 2.5
    items per workgroup (256 workgroups, 512 for the short sequences) must give every image the same bits as the image alone -- with
    197 and 224 tokens (7 key tiles: the last one ragged / full), 50 and 33 (rows past the last token zero-filled by the descriptor's
    range, two workgroups per CU); twice, for run-to-run determinism."""
    D = heads * 64
    rng = np.random.default_rng(13)
    vals = rng.uniform(-1.5, 1.5, (n * T, 3 * D)).astype(np.float32)
    vals[:, :D] *= np.float32(B.QSCALE)
    bits = B.to_bf16_bits(vals)
    whole = B.attention_bf16io(bits, n, T, heads, q_scaled=True).reshape(n, T, D)
    assert np.array_equal(B.attention_bf16io(bits, n, T, heads, q_scaled=True).reshape(n, T, D), whole)
    for i in list(range(0, n, max(1, n // 12))) + [n - 1]:
        one = B.attention_bf16io(bits[i * T:(i + 1) * T], 1, T, heads, q_scaled=True).reshape(T, D)
        assert np.array_equal(whole[i], one), i


BF16_PROB_TOL = 2e-2   # bf16 activations carry 8 significant bits; the fp32 bar (1e-4) does not apply here


@pytest.fixture(params=[0, -1], ids=["ln-folded", "ln-kernels"])
def ln_fold(request):
    """vit_engine_options.ln_fold: 0 = auto (the encoder LayerNorms folded into the GEMMs either side wherever the widths allow:
    every model here but VIT_TINY), -1 = one LayerNorm kernel per LayerNorm.  Both must meet the same bar."""
    return request.param


@pytest.mark.parametrize("cfg,n", [(synth.VIT_TINY, 5), (synth.VIT_SMALL, 6)])
def test_bf16_forward_small_models(oracle, cfg, n, ln_fold):
    from conftest import oracle_config
    W = synth.make_weights(cfg, 21)
    eng = B.Engine(cfg, max_batch=4, dtype="bf16", ln_fold=ln_fold)
    eng.load_weights(W)
    imgs = synth.make_images(cfg, n, 100)
    probs = eng.forward(imgs)
    ref = oracle.forward(oracle_config(cfg), imgs, W)
    assert float(np.abs(probs - ref).max()) <= BF16_PROB_TOL
    assert (probs.argmax(1) == ref.argmax(1)).all()
    eng.close()


def test_bf16_forward_beyond_704_tokens(oracle, ln_fold):
    """730 tokens: past the streamed attention kernel's range, so the chunked kernel runs -- with the LayerNorm fold on it receives
    Q columns that already hold QSCALE * q (the in_proj fold) and must not scale them again."""
    from conftest import oracle_config
    cfg = synth.ModelConfig(img_size=432, patch_size=16, in_chans=3, num_classes=10, embed_dim=128, depth=2, num_heads=2, hidden_dim=256)
    assert cfg.tokens == 730
    W = synth.make_weights(cfg, 23)
    eng = B.Engine(cfg, max_batch=2, dtype="bf16", ln_fold=ln_fold)
    eng.load_weights(W)
    imgs = synth.make_images(cfg, 3, 101)
    probs = eng.forward(imgs)
    ref = oracle.forward(oracle_config(cfg), imgs, W)
    assert float(np.abs(probs - ref).max()) <= BF16_PROB_TOL
    assert (probs.argmax(1) == ref.argmax(1)).all()
    eng.close()


def test_bf16_forward_b16_vs_reference_golden(ln_fold):
    """ViT-B/16 with bf16 MFMA GEMMs against the fp32 reference vectors: same top-1, probabilities within
    BF16_PROB_TOL (measured 2.4e-3)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vit_b16_e2e.npz"))
    cfg = synth.VIT_B16
    eng = B.Engine(cfg, max_batch=8, dtype="bf16", lanes=2, ln_fold=ln_fold)
    eng.load_weights(synth.make_weights(cfg, int(g["weight_seed"])))
    imgs = synth.make_images(cfg, int(g["n_images"]), int(g["image_seed"]))
    probs = eng.forward(imgs)
    err = float(np.abs(probs - g["probs"]).max())
    print("bf16 max |dprob| vs fp32 reference:", err)
    assert err <= BF16_PROB_TOL
    assert (probs.argmax(1) == g["probs"].argmax(1)).all()
    eng.close()


def test_bf16_forward_b16_batch2048_position_invariance_and_golden():
    """BASELINE.json configs[2] at its full size: 2,048 images, two lanes, LayerNorm fold on -- 1,576-row-tile persistent
    ping-pong GEMMs and 12,288 (image, head) items per lane for the resident attention, shapes no 8-image test reaches.
    The batch is 16 distinct images x 128 copies in a shuffled order: copies must give bit-identical rows wherever they sit
    (any lane, any tile, any workgroup), rows sum to 1, and the two golden images inside still meet the bf16 bar against the
    reference's fp32 probabilities.  The bf16 twin of test_b16_batch256_position_invariance_and_golden."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vit_b16_e2e.npz"))
    cfg = synth.VIT_B16
    n = 2048
    base = synth.make_images(cfg, 16, int(g["image_seed"]))            # images 0, 1 are the golden ones
    idx = (np.arange(n) % 16)[np.random.default_rng(2).permutation(n)]
    eng = B.Engine(cfg, max_batch=n, dtype="bf16", lanes=2)
    eng.load_weights(synth.make_weights(cfg, int(g["weight_seed"])))
    d_in, d_out = B.DeviceArray.from_numpy(base[idx]), B.DeviceArray((n, cfg.num_classes))
    eng.forward_device(d_in.ptr, n, d_out.ptr)                          # ONE chunk of 2,048 (what bench.py --config 2 times)
    eng.sync()
    probs = d_out.numpy()
    eng.close()
    for k in range(16):
        rows = probs[idx == k]
        assert (rows == rows[0]).all(), f"image {k}: rows differ with batch position"
    assert np.allclose(probs.sum(1), 1.0, atol=1e-5)
    for k in (0, 1):
        row = probs[idx == k][0]
        assert float(np.abs(row - g["probs"][k]).max()) <= BF16_PROB_TOL
        assert int(row.argmax()) == int(g["probs"][k].argmax())


def test_bf16_forward_vit_l_geometry(oracle, ln_fold):
    """ViT-L/16-384 width and sequence length (D=1024, 16 heads, H=4096, 577 tokens), 2 layers, bf16 GEMMs
    (K = 1024 / 4096) + the chunked attention with bf16 I/O, against the fp32 oracle."""
    from conftest import oracle_config
    cfg = synth.ModelConfig(img_size=384, embed_dim=1024, depth=2, num_heads=16, hidden_dim=4096)
    W = synth.make_weights(cfg, 31)
    eng = B.Engine(cfg, max_batch=2, dtype="bf16", ln_fold=ln_fold)
    eng.load_weights(W)
    imgs = synth.make_images(cfg, 2, 32)
    probs = eng.forward(imgs)
    ref = oracle.forward(oracle_config(cfg), imgs, W)
    assert float(np.abs(probs - ref).max()) <= BF16_PROB_TOL
    assert (probs.argmax(1) == ref.argmax(1)).all()
    eng.close()


def test_ln_fold_is_invariant_to_lanes_chunks_and_pruning():
    """The folded forward keeps the engine's invariances: the same bits whatever the lane split, the chunking, the position in
    the batch (row sums are added in a fixed order, no atomics) or exact last-layer pruning (the class rows go through the same
    folded GEMMs with their own (rstd, mean*rstd) pairs).  Folded vs LayerNorm kernels: close, not equal."""
    cfg = synth.ModelConfig(img_size=64, embed_dim=256, depth=3, num_heads=4, hidden_dim=512, num_classes=100)
    W = synth.make_weights(cfg, 77)
    imgs = synth.make_images(cfg, 11, 78)
    outs = {}
    for key, kw in {"base": dict(max_batch=16), "lanes3": dict(max_batch=16, lanes=3), "chunks": dict(max_batch=4, lanes=2),
                    "pruned": dict(max_batch=16, prune_last_layer=True), "kernels": dict(max_batch=16, ln_fold=-1)}.items():
        eng = B.Engine(cfg, dtype="bf16", **kw)
        eng.load_weights(W)
        outs[key] = eng.forward(imgs)
        eng.close()
    for same in ("lanes3", "chunks", "pruned"):
        assert np.array_equal(outs[same], outs["base"]), same
    assert np.array_equal(outs["base"][::-1], B_forward_reversed(cfg, W, imgs))
    d = float(np.abs(outs["kernels"] - outs["base"]).max())
    assert 0 < d <= 5e-3, d
    assert (outs["kernels"].argmax(1) == outs["base"].argmax(1)).all()


def test_ln_fold_option_is_validated():
    """ln_fold = 1 insists on the fold and fails loudly where the ping-pong GEMM cannot carry it (embed_dim < 128); 0 (auto) falls
    back to LayerNorm kernels there, and fp32 engines ignore the option (their path keeps the reference's operation order)."""
    narrow = synth.ModelConfig(img_size=32, num_classes=10, embed_dim=64, depth=1, num_heads=1, hidden_dim=128)
    with pytest.raises(B.VitError, match="ln_fold"):
        B.Engine(narrow, max_batch=2, dtype="bf16", ln_fold=1)
    for kw in (dict(dtype="bf16", ln_fold=0), dict(dtype="f32", ln_fold=1)):
        eng = B.Engine(narrow, max_batch=2, **kw)
        eng.close()


def B_forward_reversed(cfg, W, imgs):
    eng = B.Engine(cfg, max_batch=16, dtype="bf16")
    eng.load_weights(W)
    out = eng.forward(imgs[::-1].copy())
    eng.close()
    return out


def test_facade_bf16_by_environment():
    """VIT_HIP_DTYPE=bf16 switches the drop-in ViT_opencl() surface to the bf16 variant (read at initialize time)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vit_b16_e2e.npz"))
    W = synth.make_weights(synth.VIT_B16, int(g["weight_seed"]))
    imgs = synth.make_images(synth.VIT_B16, 2, int(g["image_seed"]))
    os.environ["VIT_HIP_DTYPE"] = "bf16"
    try:
        probs = B.facade_forward(imgs, W, use_reference_names=True)
    finally:
        del os.environ["VIT_HIP_DTYPE"]
    ref = g["probs"][:2]
    assert (probs.argmax(1) == ref.argmax(1)).all()
    err = float(np.abs(probs - ref).max())
    assert 1e-5 < err <= 2e-2, err     # within the bf16 bar, and really the bf16 path (fp32 gives ~1e-6)
