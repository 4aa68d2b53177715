"""Op-level parity: every HIP kernel, called through the C-ABI, against the CPU oracle.

The oracle (oracle/vit_cpu_ref.c) is bit-identical to the reference's ViT_seq.c; the kernels sum
in a different order (MFMA k-blocking, wave reductions), so the bar is the north star's fp32
tolerance: 1e-4 absolute on softmax outputs, and a relative 2e-5 of the tensor's magnitude on
unnormalised activations (the measured oracle noise is ~6e-6).
"""
import numpy as np
import pytest

from vit_amd import binding as B
from vit_amd import synth

pytestmark = pytest.mark.gpu

REL = 2e-5


def u(k, shape, a, seed=777):
    n = int(np.prod(shape))
    return synth.uniform(seed, k, n, -a, a).reshape(shape)


def close(a, b, rel=REL):
    scale = float(np.abs(b).max()) + 1e-30
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert err <= rel * scale, f"max err {err:.3e} > {rel} * {scale:.3e}"


# ---- GEMM ---------------------------------------------------------------------------------------

def test_gemm_identity_asymmetric_exact():
    """A = I with an asymmetric W: C must be W^T exactly (catches any fragment/row/col swap)."""
    K = 256
    A = np.eye(K, dtype=np.float32)
    W = u(1, (160, K), 1.0)
    bias = np.zeros(160, np.float32)
    C = B.gemm(A, W, bias)
    assert np.array_equal(C, W.T.copy())


@pytest.mark.parametrize("M,N,K", [(197, 768, 768), (394, 2304, 768), (300, 1000, 768), (5, 10, 128),
                                   (129, 33, 64), (1, 1000, 768), (640, 256, 3072)])
def test_gemm_bias(oracle, M, N, K):
    A, W, b = u(2, (M, K), 1.0), u(3, (N, K), 0.05), u(4, (N,), 0.1)
    close(B.gemm(A, W, b), oracle.linear(A, W, b))


def test_gemm_bias_gelu(oracle):
    M, N, K = 394, 3072, 768
    A, W, b = u(5, (M, K), 1.0), u(6, (N, K), 0.08), u(7, (N,), 0.1)
    ref = oracle.linear(A, W, b)
    ref = oracle.gelu(ref)
    got = B.gemm(A, W, b, epilogue=B.EPI_BIAS_GELU)
    close(got, ref)


def test_gemm_bias_residual(oracle):
    M, N, K = 394, 768, 3072
    A, W, b, R = u(8, (M, K), 1.0), u(9, (N, K), 0.03), u(10, (N,), 0.1), u(11, (M, N), 2.0)
    ref = R + oracle.linear(A, W, b)  # ViT_seq.c:297-299: residual + mlp_out
    close(B.gemm(A, W, b, residual=R, epilogue=B.EPI_BIAS_RESIDUAL), ref)


@pytest.mark.parametrize("tile", [0, 6, 7, 8, 9, 10, 11])
def test_gemm_tile_variants(oracle, tile):
    M, N, K = 515, 200, 96
    A, W, b = u(12, (M, K), 1.0), u(13, (N, K), 0.1), u(14, (N,), 0.1)
    close(B.gemm(A, W, b, tile=tile), oracle.linear(A, W, b))
    close(B.gemm(A, W, b, tile=tile, group_m=1), oracle.linear(A, W, b))


@pytest.mark.parametrize("M,N,K", [(197, 768, 768), (515, 200, 256), (1, 1000, 768), (33, 40, 128)])
def test_gemm_latency_tile(oracle, M, N, K):
    """Tile 12 (32x32 workgroup tiles, 16x16 per wave, K step 128; csrc/vit_gemm_latency.hip) against the oracle, all three
    epilogues, ragged M and N, a single row."""
    A, W, b, R = u(30, (M, K), 1.0), u(31, (N, K), 0.1), u(32, (N,), 0.1), u(33, (M, N), 2.0)
    lin = oracle.linear(A, W, b)
    close(B.gemm(A, W, b, tile=12), lin)
    close(B.gemm(A, W, b, epilogue=B.EPI_BIAS_GELU, tile=12), oracle.gelu(lin))
    close(B.gemm(A, W, b, residual=R, epilogue=B.EPI_BIAS_RESIDUAL, tile=12), R + lin)
    with pytest.raises(B.VitError):
        B.gemm(A[:, :96].copy(), W[:, :96].copy(), b, tile=12)   # K % 128 != 0: the caller must pick another tile


@pytest.mark.parametrize("T", [197, 176, 224, 130])
def test_attention_small_batch_split_is_bit_identical(T):
    """With few (image, head) items the resident attention kernel cuts a head's query blocks over up to four workgroups (parts);
    a row's arithmetic must not depend on that: one image alone (12 items: 4 parts) equals the same image inside a batch that
    fills the chip (312 items: 1 part) bit for bit -- including the tail block, which is cut in four by keys either way."""
    heads, n = 12, 26
    qkv = u(70, (n * T, 3 * heads * 64), 1.5)
    big = B.attention(qkv, n, T, heads)
    for i in (0, 7, n - 1):
        one = B.attention(qkv[i * T:(i + 1) * T].copy(), 1, T, heads)
        assert np.array_equal(one, big[i * T:(i + 1) * T]), (T, i)
    two = B.attention(qkv[3 * T:5 * T].copy(), 2, T, heads)
    assert np.array_equal(two, big[3 * T:5 * T]), T


def test_mfma_shapes_add_their_products_like_an_fma_chain(tmp_path):
    """The fact the latency tile rests on, checked on the device in front of us: v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32
    and a chain of v_fma_f32 over the same k order produce the same bits (tools/probes/mfma_order_probe.hip, K = 3072)."""
    import json, os, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    src = os.path.join(os.path.dirname(__file__), "..", "tools", "probes", "mfma_order_probe.hip")
    exe = str(tmp_path / "mfma_order_probe")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", src, "-o", exe], check=True, timeout=300)
    out = subprocess.run([exe], check=True, timeout=60, capture_output=True, text=True).stdout
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec["of"] == 1024
    assert rec["differ_32x32x2_vs_16x16x4"] == 0 and rec["differ_32x32x2_vs_fma_chain"] == 0 and rec["differ_16x16x4_vs_fma_chain"] == 0, rec


def test_gemm_tile_shapes_are_bit_identical():
    """The engine picks tile shapes by problem size (32x32 latency tiles or 64x64 for one image, 128x64, 128x128 persistent ...)
    and documents that a row's result does not depend on how many rows it is computed with (batch position, lanes,
    prune_last_layer): every output must sum its k in the same order, one fp32 rounding per product, whatever the tile -- including
    tile 12, whose v_mfma_f32_16x16x4_f32 adds four products per instruction where the others' 32x32x2 adds two
    (tools/probes/mfma_order_probe.hip: both equal a chain of fused multiply-adds in k order)."""
    M, N, K = 333, 200, 768
    A, W, b, R = u(40, (M, K), 1.0), u(41, (N, K), 0.1), u(42, (N,), 0.1), u(43, (M, N), 2.0)
    for epi, res in ((B.EPI_BIAS, None), (B.EPI_BIAS_GELU, None), (B.EPI_BIAS_RESIDUAL, R)):
        ref = B.gemm(A, W, b, residual=res, epilogue=epi, tile=10)
        for tile in (0, 6, 7, 8, 9, 11, 12):
            assert np.array_equal(B.gemm(A, W, b, residual=res, epilogue=epi, tile=tile), ref), (epi, tile)
        assert np.array_equal(B.gemm(A[:7], W, b, residual=None if res is None else res[:7], epilogue=epi), ref[:7]), epi


def test_gemm_helper_pieces_are_bit_identical_and_reusable():
    """Persistent walk with a workspace: the tiles of the partial last round start on an idle workgroup and finish on their
    owner (csrc/vit_gemm_persistent.hip).  600 tiles on 512 workgroups: 88 owners hand 12 of 24 K-steps to a helper.  The
    accumulation chain moves, it is not split: results must equal the one-workgroup-per-tile kernel bit for bit, for every
    epilogue, and again when the same workspace serves the next launch (every flag is back to 0 when a launch ends).
    handover_test = 1 makes the helpers run their pieces last: the owners then find nothing, withdraw their request and
    compute the whole tile -- no path may store a tile whose accumulators it did not wait for."""
    import ctypes as C
    M, N, K = 128 * 100, 768, 768
    A, W, b, R = u(50, (M, K), 1.0), u(51, (N, K), 0.05), u(52, (N,), 0.1), u(53, (M, N), 2.0)
    L = B.lib()
    info = B.device_info(0)
    owners = (M // 128) * (N // 128) - 2 * info["compute_units"]
    assert owners == 88 or info["compute_units"] != 256
    ws = B.gemm_workspace()
    dA, dW, db, dR = (B.DeviceArray.from_numpy(a) for a in (A, W, b, R))
    B.gemm_workspace_stats(ws)
    for epi, res in ((B.EPI_BIAS, None), (B.EPI_BIAS_GELU, None), (B.EPI_BIAS_RESIDUAL, R)):
        ref = B.gemm(A, W, b, residual=res, epilogue=epi, tile=10)
        for late in (0, 1, 0):
            dC = B.DeviceArray.from_numpy(np.full((M, N), 7.0, np.float32))
            args = B.CGemmArgs(dA.ptr, K, dW.ptr, K, db.ptr, dR.ptr if res is not None else None, N, dC.ptr, N, M, N, K, epi, 9, 0, ws, late)
            B.hip_check(L.vithip_gemm_f32(None, C.byref(args)), "vithip_gemm_f32")
            assert np.array_equal(dC.numpy(), ref), (epi, late)
            st = B.gemm_workspace_stats(ws)
            assert st["taken"] + st["recomputed"] == owners, (st, late)      # every owner decided exactly once
            if late:
                assert st["recomputed"] > 0, st                               # (the helpers were late for most or all of them)
            else:
                assert st["taken"] >= owners * 3 // 4, st                     # an idle device: the pieces are there
            assert not B.gemm_workspace_flags(ws).any(), (epi, late)          # parked pieces consumed, withdrawn marks cleared
    # What an aborted launch may leave behind: "parked" (1) and "withdrawn" (2) marks of EARLIER launches on flags whose slots
    # hold nothing of this launch.  Flag words carry the number of the launch that wrote them, so they read as empty.
    ref = B.gemm(A, W, b, residual=R, epilogue=B.EPI_BIAS_RESIDUAL, tile=10)
    stale = np.zeros(owners, np.int32)
    stale[0::3] = (1 << 2) | 1          # launch 1 parked a piece here and nobody took it
    stale[1::3] = (2 << 2) | 2          # launch 2's owner withdrew and its helper never got to clear the mark
    stale[2::3] = 1                     # a word of the generation-less protocol of earlier builds
    for late in (0, 1):
        B.gemm_workspace_set_flags(ws, stale)
        dC = B.DeviceArray.from_numpy(np.full((M, N), 7.0, np.float32))
        args = B.CGemmArgs(dA.ptr, K, dW.ptr, K, db.ptr, dR.ptr, N, dC.ptr, N, M, N, K, B.EPI_BIAS_RESIDUAL, 9, 0, ws, late)
        B.hip_check(L.vithip_gemm_f32(None, C.byref(args)), "vithip_gemm_f32")
        assert np.array_equal(dC.numpy(), ref), late
        st = B.gemm_workspace_stats(ws)
        assert st["taken"] + st["recomputed"] == owners, (st, late)
        assert not B.gemm_workspace_flags(ws).any(), late
    L.vithip_gemm_f32_workspace_destroy(C.c_void_p(ws))


def test_gemm_rejects_bad_k():
    A, W, b = u(15, (8, 40), 1.0), u(16, (8, 40), 1.0), u(17, (8,), 1.0)
    with pytest.raises(B.VitError):
        B.gemm(A, W, b)  # K = 40 is not a multiple of 32
    for tile in (1, 2, 3, 4, 5, 13):   # the tile codes of the retired not-pipelined kernels, and one past the last
        with pytest.raises(B.VitError):
            B.gemm(np.zeros((8, 64), np.float32), np.zeros((8, 64), np.float32), b, tile=tile)


# ---- LayerNorm ------------------------------------------------------------------------------------

@pytest.mark.parametrize("rows,dim", [(197, 768), (50, 128), (7, 1024), (3, 192), (1000, 768)])
def test_layernorm(oracle, rows, dim):
    x = u(20, (rows, dim), 3.0) + 0.5
    g, b = synth.uniform(777, 21, dim, 0.5, 1.5), u(22, (dim,), 0.5)
    close(B.layernorm(x, g, b), oracle.layer_norm(x, g, b))


def test_layernorm_constant_row_uses_eps(oracle):
    """var == 0: the result is decided by the 1e-6 epsilon the OpenCL kernel forgot."""
    x = np.full((4, 768), 1.25, np.float32)
    g, b = np.ones(768, np.float32), np.zeros(768, np.float32)
    got = B.layernorm(x, g, b)
    assert np.isfinite(got).all()
    close(got, oracle.layer_norm(x, g, b), rel=1e-3)


# ---- attention ------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,T,heads", [(2, 197, 12), (3, 5, 2), (2, 50, 3), (1, 33, 1), (1, 224, 2), (1, 32, 1),
                                       (2, 129, 2)])
def test_attention(oracle, n, T, heads):
    D = heads * 64
    qkv = u(30, (n * T, 3 * D), 1.5)
    got = B.attention(qkv, n, T, heads).reshape(n, T, D)
    for i in range(n):
        blk = qkv[i * T:(i + 1) * T]
        q, k, v = (np.ascontiguousarray(blk[:, j * D:(j + 1) * D]) for j in range(3))
        ref = oracle.attention_core(q, k, v, heads)
        err = float(np.abs(got[i] - ref).max())
        assert err <= 1e-5 * max(1.0, float(np.abs(ref).max())), f"image {i}: {err}"


def test_attention_peaked_rows(oracle):
    """Large scores (|s| ~ 60): the max subtraction must keep expf in range."""
    n, T, heads = 1, 197, 2
    D = heads * 64
    qkv = u(31, (T, 3 * D), 8.0)
    got = B.attention(qkv, n, T, heads)
    q, k, v = (np.ascontiguousarray(qkv[:, j * D:(j + 1) * D]) for j in range(3))
    ref = oracle.attention_core(q, k, v, heads)
    assert np.isfinite(got).all()
    assert float(np.abs(got - ref).max()) <= 2e-4 * float(np.abs(ref).max())


@pytest.mark.parametrize("n,T,heads", [(1, 225, 1), (2, 300, 2), (1, 448, 1), (2, 577, 2), (1, 700, 1)])
def test_attention_long_sequences_chunked(oracle, n, T, heads):
    """More than 224 tokens: K/V stream through LDS in chunks with an online softmax (ViT-L/16-384: 577)."""
    D = heads * 64
    qkv = u(32, (n * T, 3 * D), 2.5)
    got = B.attention(qkv, n, T, heads).reshape(n, T, D)
    for i in range(n):
        blk = qkv[i * T:(i + 1) * T]
        q, k, v = (np.ascontiguousarray(blk[:, j * D:(j + 1) * D]) for j in range(3))
        ref = oracle.attention_core(q, k, v, heads)
        err = float(np.abs(got[i] - ref).max())
        assert err <= 2e-5 * max(1.0, float(np.abs(ref).max())), f"image {i}: {err}"


def test_attention_chunked_running_max_moves(oracle):
    """Force the rescale branch: the row maximum sits in the LAST chunk for half of the queries."""
    T, heads = 500, 1
    qkv = u(33, (T, 192), 1.0)
    qkv[450:, 64:128] *= 6.0  # large keys at the end of the sequence
    got = B.attention(qkv, 1, T, heads)
    q, k, v = (np.ascontiguousarray(qkv[:, j * 64:(j + 1) * 64]) for j in range(3))
    ref = oracle.attention_core(q, k, v, heads)
    assert float(np.abs(got - ref).max()) <= 2e-5 * float(np.abs(ref).max())


# ---- patch embedding ------------------------------------------------------------------------------

# patch sizes either side of the K step of 32: 8 (four pixel rows per K step), 16 (two), 32 (one), 64 (a pixel row spans two K steps:
# the scalar offset of a K step then carries the start inside the row) and 12 (neither divides the other: classic loop)
_PATCH_CFGS = [synth.ModelConfig(img_size=s, patch_size=p, in_chans=c, num_classes=10, embed_dim=64, depth=1, num_heads=1, hidden_dim=128)
               for s, p, c in ((32, 8, 2), (64, 32, 1), (128, 64, 1), (192, 64, 3), (48, 12, 2))]


@pytest.mark.parametrize("cfg,n", [(synth.VIT_TINY, 3), (synth.VIT_SMALL, 2), (synth.VIT_B16, 2)] + [(c, 5) for c in _PATCH_CFGS])
def test_patch_embed(oracle, cfg, n):
    from conftest import oracle_config
    W = [synth.make_weight(cfg, i, 5) for i in range(4)]
    imgs = synth.make_images(cfg, n, 6)
    got = B.patch_embed(cfg, imgs, W[1], W[2], W[0], W[3])
    ocfg = oracle_config(cfg)
    for i in range(n):
        close(got[i], oracle.embed(ocfg, imgs[i], W))


# ---- softmax + top-1 ------------------------------------------------------------------------------

def test_softmax_top1(oracle):
    logits = u(40, (37, 1000), 6.0)
    probs, label, prob = B.softmax_top1(logits)
    for r in range(logits.shape[0]):
        ref = oracle.softmax(logits[r])
        assert float(np.abs(probs[r] - ref).max()) <= 1e-6
        assert label[r] == int(ref.argmax())
        assert prob[r] == probs[r, label[r]]
    assert np.allclose(probs.sum(1), 1.0, atol=1e-5)


def test_softmax_top1_tie_takes_first():
    logits = np.zeros((2, 10), np.float32)
    logits[1, 3] = logits[1, 7] = 2.0
    _, label, _ = B.softmax_top1(logits)
    assert list(label) == [0, 3]  # Main.c:64-68 only replaces on strictly greater


def test_gelu_epilogue_matches_libm_erf_over_range(oracle):
    """The kernel's polynomial erf against the oracle's erff on a dense sweep of pre-activations
    (-8..8): one GEMM with K = 32 whose single non-zero product passes the sweep value through."""
    n = 4096
    xs = np.linspace(-8.0, 8.0, n).astype(np.float32)
    A = np.zeros((n, 32), np.float32)
    A[:, 0] = xs
    W = np.zeros((32, 32), np.float32)
    W[:, 0] = 1.0
    got = B.gemm(A, W, np.zeros(32, np.float32), epilogue=B.EPI_BIAS_GELU)[:, 0]
    ref = oracle.gelu(xs)
    assert float(np.abs(got - ref).max()) <= 1e-6
