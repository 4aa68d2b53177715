"""LayerNorm folded into the bf16 GEMMs either side of it (bf16 variant; the LayerNorm is ViT_seq.c:103-121).

    LN(x) . W^T + b  =  rstd * (x . (gamma*W)^T)  -  rstd * mean * colsum(gamma*W)  +  (b + W . beta)

Producer = the residual GEMM in front (stores bf16(x) and per-row partial sums), consumer = the GEMM behind (raw bf16 rows,
folded weight, epilogue rescale).  References are float64 numpy on the same bf16-exact operands; tolerances are written where used.
"""
import numpy as np
import pytest

from vit_amd import binding as B
from vit_amd import synth

pytestmark = pytest.mark.gpu


def u(k, shape, a, seed=4242):
    return synth.uniform(seed, k, int(np.prod(shape)), -a, a).reshape(shape)


def ln_rows_ref(x):
    """(rstd, mean * rstd) per row, the reference's definition: var = E[x^2] - mean^2, 1/sqrt(var + 1e-6)."""
    x = x.astype(np.float64)
    mean = x.mean(axis=1)
    var = (x * x).mean(axis=1) - mean * mean
    rstd = 1.0 / np.sqrt(var + 1e-6)
    return np.stack([rstd, mean * rstd], axis=1)


def test_fold_weights_matches_definition():
    N, K = 300, 768
    W, b, g, be = u(0, (N, K), 0.05), u(1, (N,), 0.1), 1.0 + u(2, (K,), 0.5), u(3, (K,), 0.2)
    Wf, cs, bf = B.ln_fold_weights(W, b, g, be)
    assert np.array_equal(Wf, B.to_bf16_bits(g[None, :] * W))                       # one fp32 multiply, one bf16 rounding
    assert np.allclose(cs, B.from_bf16_bits(Wf).astype(np.float64).sum(axis=1), rtol=0, atol=2e-5)   # sums of the ROUNDED values
    assert np.allclose(bf, b + W.astype(np.float64) @ be.astype(np.float64), rtol=0, atol=2e-6)


@pytest.mark.parametrize("rows,dim", [(1000, 768), (77, 1024), (5, 64)])
def test_rowstats_bf16(rows, dim):
    x = u(4, (rows, dim), 2.0) + u(5, (rows, 1), 1.0)          # rows with different means
    x16, rs = B.rowstats_bf16(x)
    assert np.array_equal(x16, B.to_bf16_bits(x))
    assert np.allclose(rs, ln_rows_ref(x), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("M,N,K", [(256 * 3 + 37, 768, 768), (600, 1024, 256), (256 * 70 + 5, 768, 128), (130, 320, 192)])
def test_producer_stores_bf16_copy_and_row_sums(M, N, K):
    """The residual epilogue with the fold: C bit-identical to the plain residual epilogue, x16 = bf16(C), and the strips' partial
    sums add up to the row sums (fp32 summation order only).  Shapes: ragged last tile row, more tiles than workgroups, N not a
    multiple of the tile (320: the strips of the absent columns must come out as zeros)."""
    Ab, Wb, b, R = B.to_bf16_bits(u(6, (M, K), 1.0)), B.to_bf16_bits(u(7, (N, K), 0.05)), u(8, (N,), 0.1), u(9, (M, N), 2.0)
    plain = B.gemm_bf16(Ab, Wb, b, residual=R, epilogue=B.BF16_EPI_F32_RESIDUAL)
    Cf, x16, part = B.gemm_bf16(Ab, Wb, b, residual=R, epilogue=B.BF16_EPI_F32_RESIDUAL, ln_producer=True)
    assert np.array_equal(Cf, plain)
    assert np.array_equal(x16, B.to_bf16_bits(Cf))
    assert part.shape == (B.ln_strips(N), M, 2)
    c64 = Cf.astype(np.float64)
    for k in range(part.shape[0]):                                                 # every strip on its own
        cols = c64[:, 64 * k:64 * (k + 1)]
        assert np.allclose(part[k, :, 0], cols.sum(axis=1), rtol=0, atol=2e-4), k
        assert np.allclose(part[k, :, 1], (cols * cols).sum(axis=1), rtol=2e-6, atol=2e-4), k
    rs = B.rowstats_finalize(part, N)
    assert np.allclose(rs, ln_rows_ref(Cf), rtol=5e-5, atol=2e-6)


@pytest.mark.parametrize("epi", [B.BF16_EPI_BF16, B.BF16_EPI_BF16_GELU])
@pytest.mark.parametrize("M,N,K", [(515, 2304, 768), (256 * 40 + 3, 1024 + 4, 256), (197, 3072, 768)])
def test_consumer_equals_layernorm_then_gemm(epi, M, N, K):
    """rstd * (x16 . Wf^T) - mean*rstd * colsum + bias_f against float64 LayerNorm(x16) . W^T + b on the same bf16 rows; the only
    differences are the bf16 rounding of gamma*W (2^-9 relative per weight, a random walk over K) and fp32 accumulation, then one
    bf16 rounding of the result.  Rows carry a mean of the size of their spread, so the mean term matters."""
    x = u(10, (M, K), 1.5) + u(11, (M, 1), 1.0)
    W, b, g, be = u(12, (N, K), 0.05), u(13, (N,), 0.1), 1.0 + u(14, (K,), 0.5), u(15, (K,), 0.2)
    x16, rs = B.rowstats_bf16(x)
    Wf, cs, bf = B.ln_fold_weights(W, b, g, be)
    got = B.from_bf16_bits(B.gemm_bf16(x16, Wf, bf, epilogue=epi, ln_rows=rs, ln_colsum=cs))
    xb = B.from_bf16_bits(x16).astype(np.float64)
    rs64 = ln_rows_ref(x)                                       # statistics of the fp32 rows, as the engine has them
    y = (xb * rs64[:, :1] - rs64[:, 1:]) * g + be               # LayerNorm of the rounded rows with those statistics
    ref = y @ W.astype(np.float64).T + b
    if epi == B.BF16_EPI_BF16_GELU:
        from scipy.special import erf
        ref = 0.5 * ref * (1.0 + erf(ref / np.sqrt(2.0)))
    # bf16(gamma * W): independent relative errors uniform in +-2^-9 per weight -> sigma of their sum; 7 sigma over ~1e7 outputs
    sigma = 2.0 ** -9 / np.sqrt(3.0) * np.sqrt((y * y) @ ((g * W).astype(np.float64) ** 2).T)
    err = np.abs(got - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 7.0 * sigma + 1e-4).all(), float((err - 2.0 ** -8 * np.abs(ref) - 7.0 * sigma).max())


def test_fold_arguments_are_validated():
    Ab, Wb, b = B.to_bf16_bits(u(16, (256, 128), 1.0)), B.to_bf16_bits(u(17, (256, 128), 0.1)), np.zeros(256, np.float32)
    rs, cs = np.ones((256, 2), np.float32), np.zeros(256, np.float32)
    with pytest.raises(RuntimeError):      # consumer operands on the residual epilogue
        B.gemm_bf16(Ab, Wb, b, residual=np.zeros((256, 256), np.float32), epilogue=B.BF16_EPI_F32_RESIDUAL, ln_rows=rs, ln_colsum=cs)
    with pytest.raises(RuntimeError):      # only one of the two consumer operands
        B.gemm_bf16(Ab, Wb, b, epilogue=B.BF16_EPI_BF16, ln_rows=rs)
    with pytest.raises(RuntimeError):      # the two-stage kernel has no fold
        B.gemm_bf16(Ab, Wb, b, epilogue=B.BF16_EPI_BF16, ln_rows=rs, ln_colsum=cs, variant=1)


def test_fp32_fold_on_near_constant_rows_stays_inside_its_stated_bound():
    """include/vit_hip_kernels.h (ln_rows): the fp32 fold computes rstd * (x . Wf^T - mean * colsum) + b', so on a row of small
    variance the cancellation error of the bracket is scaled by rstd (at most 1 / sqrt(1e-6) = 1e3), where LayerNorm-then-GEMM
    (ViT_seq.c:103-121, then 134-147) has no such term.  Rows from "ordinary" down to exactly constant, against a float64
    LayerNorm-then-GEMM on the same (rstd, mean) pairs (the reference's operation order): the difference stays within
    rstd * 8 * 2^-24 * sum_k |x_k * Wf_k|  (the fp32 product chain's rounding, amplified) + 2e-5, and within the 2e-5 bar of every
    other fp32 op test for rows with the spread of real residual rows (rstd of order 1).
    The constant rows are 2.0, -0.5 and 0.0 everywhere -- values whose mean and mean of squares are exact in fp32.  (Rows that are
    merely ALMOST constant around a large mean are outside what the reference itself defines: its var = E[x^2] - mean^2 in fp32
    goes negative by more than the 1e-6 epsilon there and ViT_seq.c:113-116 takes the square root of a negative number -- the first
    form of this test, with such rows, got NaN from the fold and would have got it from the reference.)"""
    K, N = 768, 768
    spreads = [1.0, 1e-1, 1e-2]
    x = np.empty((128 * (len(spreads) + 1), K), np.float32)
    for i, sp in enumerate(spreads):
        x[128 * i:128 * (i + 1)] = u(40 + i, (128, 1), 2.0) + np.float32(sp) * u(50 + i, (128, K), 1.0)
    x[128 * len(spreads):] = np.repeat(np.array([2.0, -0.5, 0.0, 2.0], np.float32), 32)[:, None]
    gamma, beta = (1.0 + u(60, (K,), 0.5)).astype(np.float32), u(61, (K,), 0.5)
    W, b = u(62, (N, K), 0.05), u(63, (N,), 0.1)
    Wf, colsum, bias_f = B.ln_fold_weights_f32(W, b, gamma, beta)
    rows = B.rowstats_f32(x)                                             # (rstd, mean), the reference's formula in fp32
    assert np.isfinite(rows).all()
    assert np.allclose(rows[128 * len(spreads):, 0], 1e3, rtol=1e-6)      # constant rows: var = 0 exactly, rstd = 1 / sqrt(1e-6)
    got = B.gemm(x, Wf, bias_f, epilogue=B.EPI_BIAS, ln=(rows, colsum)).astype(np.float64)
    assert np.isfinite(got).all()
    # LayerNorm-then-GEMM in float64 with the SAME (rstd, mean) pairs: what is measured is the fold's arithmetic, not the statistics
    # (two fp32 evaluations of var = E[x^2] - mean^2 in different summation orders -- the LayerNorm kernel's and this one -- already
    # differ by several per cent in rstd on the spread-0.01 rows: that ill-conditioning is the reference's formula, fold or no fold)
    x64, r64 = x.astype(np.float64), rows.astype(np.float64)
    ref = ((x64 - r64[:, 1:]) * r64[:, :1] * gamma + beta) @ W.astype(np.float64).T + b
    amp = r64[:, :1] * (np.abs(x64) @ np.abs(Wf.astype(np.float64)).T)
    err = np.abs(got - ref)
    assert (err <= 8 * 2.0 ** -24 * amp + 2e-5).all(), float((err - 8 * 2.0 ** -24 * amp).max())
    worst = {("spread %g" % sp): float(err[128 * i:128 * (i + 1)].max()) for i, sp in enumerate(spreads)}
    worst["constant rows (rstd 1e3)"] = float(err[128 * len(spreads):].max())
    print("fp32 fold vs float64 LayerNorm-then-GEMM on the same statistics, max |difference| per kind of row:", worst)
    assert worst["spread 1"] <= 2e-5            # rows like the residual stream's (spread ~ mean, rstd of order 1): the usual bar
    assert worst["spread 0.1"] <= 2e-4          # rstd ~ 17: measured 1.1e-4
    # and the unfolded path (LayerNorm kernel, then the plain GEMM) agrees where the statistics are well-conditioned
    unfolded = B.gemm(B.layernorm(x[:128], gamma, beta), W, b, epilogue=B.EPI_BIAS).astype(np.float64)
    assert float(np.abs(got[:128] - unfolded).max()) <= 2e-5
    # the CENTRED weight the engine folds with since round 5 (column mean of gamma * W taken out of every row of it, no colsum term in
    # the epilogue): the same bound with its own weight in the amplification, and the same bars
    Wc, _, bias_c = B.ln_fold_weights_f32_centered(W, b, gamma, beta)
    got_c = B.gemm(x, Wc, bias_c, epilogue=B.EPI_BIAS, ln=(rows, None)).astype(np.float64)
    assert np.isfinite(got_c).all()
    amp_c = r64[:, :1] * (np.abs(x64) @ np.abs(Wc.astype(np.float64)).T)
    err_c = np.abs(got_c - ref)
    assert (err_c <= 8 * 2.0 ** -24 * amp_c + 2e-5).all(), float((err_c - 8 * 2.0 ** -24 * amp_c).max())
    worst_c = {("spread %g" % sp): float(err_c[128 * i:128 * (i + 1)].max()) for i, sp in enumerate(spreads)}
    worst_c["constant rows (rstd 1e3)"] = float(err_c[128 * len(spreads):].max())
    print("the same with the centred weight:", worst_c)
    assert worst_c["spread 1"] <= 2e-5 and worst_c["spread 0.1"] <= 2e-4
    assert float(np.abs(got_c[:128] - unfolded).max()) <= 2e-5
