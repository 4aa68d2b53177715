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


def _rowstats_model(x):
    """vithip_rowstats_f32's documented order in numpy fp32: per 64-column strip u = x[c + l] + x[c + 32 + l] and
    w = fma(x[c + 32 + l], x[c + 32 + l], x[c + l] * x[c + l]) for l < 32, a butterfly 16, 8, 4, 2, 1, strips ascending from 0."""
    x = np.asarray(x, np.float32)
    rows, dim = x.shape
    s = np.zeros(rows, np.float32)
    ss = np.zeros(rows, np.float32)
    for c in range(0, dim, 64):
        x0, x1 = x[:, c:c + 32], x[:, c + 32:c + 64]
        u_ = x0 + x1
        sq0 = (x0 * x0).astype(np.float32)
        w_ = (x1.astype(np.longdouble) * x1.astype(np.longdouble) + sq0.astype(np.longdouble)).astype(np.float32)  # one rounding: the fma
        for off in (16, 8, 4, 2, 1):
            idx = np.arange(32) ^ off
            u_ = u_ + u_[:, idx]
            w_ = w_ + w_[:, idx]
        s = s + u_[:, 0]
        ss = ss + w_[:, 0]
    mean = s / np.float32(dim)
    var = ss / np.float32(dim) - mean * mean
    rstd = (np.float32(1.0) / np.sqrt((var.astype(np.float64) + 1e-6).astype(np.float32))).astype(np.float32)
    return np.stack([rstd, mean], axis=1)


@pytest.mark.parametrize("rows,dim", [(1, 64), (37, 192), (333, 768), (130, 1024), (9, 1984)])
def test_rowstats_f32_follows_its_documented_order(rows, dim):
    """(rstd, mean) per row, bit for bit the documented summation order (a GEMM epilogue that holds the row can reproduce
    it), and the LayerNorm statistics of ViT_seq.c:103-121 to fp32 rounding."""
    x = (u(60, (rows, dim), 3.0) + np.float32(0.7)).astype(np.float32)
    got = B.rowstats_f32(x)
    assert np.array_equal(got, _rowstats_model(x))
    x64 = x.astype(np.float64)
    mean, var = x64.mean(1), x64.var(1)
    rstd = 1.0 / np.sqrt(var + 1e-6)
    assert np.abs(got[:, 0] - rstd).max() <= 2e-6 * rstd.max() and np.abs(got[:, 1] - mean).max() <= 2e-6 * np.abs(mean).max() + 1e-6


def test_gemm_layernorm_fold_matches_layernorm_then_gemm():
    """LN(x) . W^T + b = rstd * (x . (gamma W)^T - mean * colsum) + (b + W . beta): the folded GEMM on the raw rows against the
    LayerNorm kernel followed by the plain GEMM (the reference's order, ViT_seq.c:103-147), both epilogues; every tile shape gives
    the same bits as the others (the fold's value is two fused multiply-adds, the inner one a rank-1 matrix instruction in the
    32x32 kernels and a v_fma in the 16x16 latency kernel -- the same rounding)."""
    M, D, N = 333, 768, 328
    x = (u(61, (M, D), 2.0) + u(62, (1, D), 1.0)).astype(np.float32)   # rows with a common offset per column: the mean term matters
    gamma, beta = (1.0 + u(63, (D,), 0.5)).astype(np.float32), u(64, (D,), 0.5)
    W, b = u(65, (N, D), 0.05), u(66, (N,), 0.1)
    Wf, colsum, bias_f = B.ln_fold_weights_f32(W, b, gamma, beta)
    assert np.array_equal(Wf, (gamma[None, :] * W).astype(np.float32))
    assert np.allclose(colsum, (gamma[None, :].astype(np.float64) * W).sum(1), rtol=0, atol=2e-6)
    assert np.allclose(bias_f, b + W.astype(np.float64) @ beta.astype(np.float64), rtol=0, atol=2e-6)
    rows = B.rowstats_f32(x)
    y = B.layernorm(x, gamma, beta)
    for epi in (B.EPI_BIAS, B.EPI_BIAS_GELU):
        want = B.gemm(y, W, b, epilogue=epi, tile=10)
        ref = B.gemm(x, Wf, bias_f, epilogue=epi, tile=10, ln=(rows, colsum))
        err = float(np.abs(ref - want).max())
        assert err <= 2e-5, (epi, err)      # values of order 1: a few fp32 ulps of the un-normalised accumulation
        for tile in (0, 6, 7, 8, 9, 11, 12):
            assert np.array_equal(B.gemm(x, Wf, bias_f, epilogue=epi, tile=tile, ln=(rows, colsum)), ref), (epi, tile)
        assert np.array_equal(B.gemm(x[:7], Wf, bias_f, epilogue=epi, ln=(rows[:7], colsum)), ref[:7]), epi
    with pytest.raises(B.VitError):    # the fold is a property of the bias / bias+GELU epilogues
        B.gemm(x, Wf, bias_f, residual=np.zeros((M, N), np.float32), epilogue=B.EPI_BIAS_RESIDUAL, ln=(rows, colsum))


def test_gemm_layernorm_fold_persistent_walk_with_helper_pieces():
    """The fold in the persistent walk: the rows' pairs travel through LDS (fetched when a tile begins, published in its last
    K-step); 600 tiles on 512 workgroups with a workspace, so that tiles started by a helper are finished -- and rescaled -- by
    their owner; a ragged last row block.  Same bits as the one-tile-per-workgroup kernel."""
    M, D, N = 128 * 49 + 57, 768, 1536
    x = (u(67, (M, D), 2.0) + u(68, (1, D), 1.0)).astype(np.float32)
    gamma, beta = (1.0 + u(69, (D,), 0.5)).astype(np.float32), u(70, (D,), 0.5)
    W, b = u(71, (N, D), 0.05), u(72, (N,), 0.1)
    Wf, colsum, bias_f = B.ln_fold_weights_f32(W, b, gamma, beta)
    rows = B.rowstats_f32(x)
    for epi in (B.EPI_BIAS, B.EPI_BIAS_GELU):
        ref = B.gemm(x, Wf, bias_f, epilogue=epi, tile=10, ln=(rows, colsum))
        for late in (0, 1):
            st = {}
            got = B.gemm(x, Wf, bias_f, epilogue=epi, tile=9, workspace=True, handover_test=late, stats=st, ln=(rows, colsum))
            assert np.array_equal(got, ref), (epi, late)
            assert st["taken"] + st["recomputed"] > 0, st
        assert np.array_equal(B.gemm(x, Wf, bias_f, epilogue=epi, tile=9, ln=(rows, colsum)), ref), epi


@pytest.mark.parametrize("K", [64, 96, 128, 160, 192])
def test_gemm_layernorm_fold_persistent_walk_short_tiles(K):
    """The persistent consumer proves its copy of the rows' pairs landed by counting: "at most 2 * NS = 16 loads outstanding" at the top
    of a tile's last K-step -- true once three steps' loads are behind the copy, so tiles of fewer than four K-steps (K = 64: two,
    K = 96: three) wait for everything instead; K = 128 is the shortest tile on the counted branch, K = 160 / 192 five and six steps.
    Both branches, odd and even step counts, many tiles per workgroup (1,200 tiles on 512), both epilogues."""
    M, N = 128 * 75 + 9, 2048
    x = (u(90, (M, K), 2.0) + u(91, (1, K), 1.0)).astype(np.float32)
    gamma, beta = (1.0 + u(92, (K,), 0.5)).astype(np.float32), u(93, (K,), 0.5)
    W, b = u(94, (N, K), 0.05), u(95, (N,), 0.1)
    if K % 64 == 0:
        Wf, colsum, bias_f = B.ln_fold_weights_f32(W, b, gamma, beta)
        rows = B.rowstats_f32(x)
    else:
        # the statistics and fold kernels take widths of 64 k only (the engine's condition for the fold); the GEMM's consumer side --
        # what this test is about -- takes any K of whole 32-deep steps: odd step counts get their operands from numpy
        Wf = (gamma[None, :] * W).astype(np.float32)
        colsum = Wf.sum(axis=1, dtype=np.float32)
        bias_f = (b + W.astype(np.float64) @ beta.astype(np.float64)).astype(np.float32)
        mean = x.mean(axis=1, dtype=np.float32)
        var = (x * x).mean(axis=1, dtype=np.float32) - mean * mean
        rows = np.stack([(1.0 / np.sqrt(var.astype(np.float64) + 1e-6)).astype(np.float32), mean], axis=1)
    want = ((x.astype(np.float64) - rows[:, 1:].astype(np.float64)) * rows[:, :1].astype(np.float64) * gamma + beta) @ W.astype(np.float64).T + b
    for epi in (B.EPI_BIAS, B.EPI_BIAS_GELU):
        ref = B.gemm(x, Wf, bias_f, epilogue=epi, tile=10, ln=(rows, colsum))
        assert np.array_equal(B.gemm(x, Wf, bias_f, epilogue=epi, tile=9, ln=(rows, colsum)), ref), epi
        assert np.array_equal(B.gemm(x, Wf, bias_f, epilogue=epi, tile=9, workspace=True, ln=(rows, colsum)), ref), epi
    assert float(np.abs(B.gemm(x, Wf, bias_f, epilogue=B.EPI_BIAS, tile=9, ln=(rows, colsum)) - want).max()) <= 2e-5


@pytest.mark.parametrize("M,N,K", [(128 * 49 + 57, 768, 768), (128 * 40, 1024, 256), (128 * 171, 768, 128), (300, 768, 128), (128 * 30, 192, 128)])
def test_gemm_residual_row_statistics_in_the_epilogue(M, N, K):
    """vithip_gemm_args.stats_out: (rstd, mean) of the rows a residual GEMM stores.  The persistent walk takes the sums in
    its epilogue (a lane's two accumulator columns first, then a reduce-scatter in the butterfly order), one launch finalises them;
    every other kernel is followed by vithip_rowstats_f32.  Whoever produces them, they are the bits of vithip_rowstats_f32 on
    the stored C, and C is what the plain residual GEMM stores.  Shapes: helper pieces + a ragged row block, 1024 columns with
    short tiles, 1,026 tiles (the auto rule picks the persistent walk), a small problem, a width that is not whole 128-column tiles
    (never in the epilogue)."""
    A, W, b, R = u(80, (M, K), 1.0), u(81, (N, K), 0.05), u(82, (N,), 0.1), (u(83, (M, N), 2.0) + u(84, (1, N), 1.0)).astype(np.float32)
    ref = B.gemm(A, W, b, residual=R, epilogue=B.EPI_BIAS_RESIDUAL, tile=10)
    want = B.rowstats_f32(ref)
    for tile, ws, late in ((9, True, 0), (9, True, 1), (9, False, 0), (0, True, 0), (10, False, 0)):
        st = {}
        got = B.gemm(A, W, b, residual=R, epilogue=B.EPI_BIAS_RESIDUAL, tile=tile, workspace=ws, handover_test=late, row_stats=st)
        assert np.array_equal(got, ref), (tile, ws, late)
        assert np.array_equal(st["rows"], want), (tile, ws, late, st["in_epilogue"])
        assert st["in_epilogue"] == int(N % 128 == 0 and (tile == 9 or (tile == 0 and (M + 127) // 128 * (N // 128) >= 1024))), (tile, st)
    st = {"scratch": False}     # no scratch lent: the statistics pass runs behind the GEMM
    assert np.array_equal(B.gemm(A, W, b, residual=R, epilogue=B.EPI_BIAS_RESIDUAL, tile=9, row_stats=st), ref)
    assert st["in_epilogue"] == 0 and np.array_equal(st["rows"], want)
    with pytest.raises(B.VitError):
        B.gemm(A, W, b, epilogue=B.EPI_BIAS, row_stats={})


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


@pytest.mark.parametrize("M,N,K", [(128 * 49 + 57, 2304, 768), (300, 768, 768), (128 * 30, 192, 128), (197, 3072, 768)])
def test_gemm_layernorm_fold_with_centred_weights(M, N, K):
    """Round 5, what the engine runs: the folded weight with its column mean taken out (vithip_ln_fold_weights_f32_centered),
    so that  x . Wc^T = x . (gamma W)^T - mean * colsum  comes out of the GEMM itself and the epilogue only scales (ln_colsum NULL).
    Against LayerNorm-then-GEMM (ViT_seq.c:103-121, 134-147) within the 2e-5 bar of every fp32 op test, on rows whose mean is as large
    as their spread; every tile shape -- the kernels that skip the centring instruction and the ones that run it on a zero -- gives the
    same bits; what is left of the weights' column sums is of the order of their rounding."""
    x = (u(70, (M, K), 2.0) + u(71, (M, 1), 2.0)).astype(np.float32)
    gamma, beta = (1.0 + u(72, (K,), 0.5)).astype(np.float32), u(73, (K,), 0.5)
    W, b = u(74, (N, K), 0.05), u(75, (N,), 0.1)
    Wc, resid, bias_f = B.ln_fold_weights_f32_centered(W, b, gamma, beta)
    gw = gamma[None, :].astype(np.float64) * W.astype(np.float64)
    assert np.allclose(Wc, gw - gw.mean(axis=1, keepdims=True), rtol=0, atol=2.0 ** -24 * np.abs(gw).max() * 2)
    assert float(np.abs(resid).max()) <= K * 2.0 ** -24 * float(np.abs(gw).max())
    _, _, bias_plain = B.ln_fold_weights_f32(W, b, gamma, beta)
    assert np.array_equal(bias_f, bias_plain)
    rows = B.rowstats_f32(x) if K % 64 == 0 else None
    want = B.gemm(B.layernorm(x, gamma, beta), W, b, epilogue=B.EPI_BIAS, tile=10).astype(np.float64)
    for epi in (B.EPI_BIAS, B.EPI_BIAS_GELU):
        ref = B.gemm(x, Wc, bias_f, epilogue=epi, tile=10, ln=(rows, None))
        for tile in (0, 8, 9, 11):
            assert np.array_equal(B.gemm(x, Wc, bias_f, epilogue=epi, tile=tile, ln=(rows, None)), ref), (epi, tile)
        assert np.array_equal(B.gemm(x, Wc, bias_f, epilogue=epi, tile=9, workspace=True, ln=(rows, None)), ref), epi
        # a zero column sum through the centring path is the same thing, bit for bit
        assert np.array_equal(B.gemm(x, Wc, bias_f, epilogue=epi, tile=9, ln=(rows, np.zeros(N, np.float32))), ref), epi
    got = B.gemm(x, Wc, bias_f, epilogue=B.EPI_BIAS, ln=(rows, None)).astype(np.float64)
    assert float(np.abs(got - want).max()) <= 2e-5
