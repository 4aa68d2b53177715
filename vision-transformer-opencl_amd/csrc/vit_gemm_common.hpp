// csrc/vit_gemm_common.hpp -- pieces shared by the fp32 GEMM kernels (vit_gemm.hip, vit_gemm_persistent.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vit_hip_kernels.h"

namespace vitgemm {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KALIGN = 32;  // K must be a multiple of this (covers both K steps below)

enum { A_DENSE = 0, A_PATCHES = 1 };
// epilogue codes beyond the public three (vit_hip_kernels.h): the consumer side of the LayerNorm fold (ln_rows / ln_colsum set)
// and the producer side: the residual epilogue that also leaves the row sums of what it stored (row_partials set; persistent walk)
enum { EPI_BIAS_LN = 3, EPI_BIAS_GELU_LN = 4, EPI_RESIDUAL_STATS = 5 };

struct GemmParams {
    const float *A;
    const float *W;
    const float *bias;
    const float *R;
    float *C;
    int lda, ldw, ldr, ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    int group_m;     // tile rows per L2 group (see tile_coords)
    // A_PATCHES only
    const float *pos;
    unsigned long long *dbg;  // DBG == 5: 8 stamps per workgroup
    int patches;     // patches per image (G*G)
    int grid;        // G = img/patch
    int patch;       // P
    int img;         // S
    int chans;       // C
    // persistent kernel, helper pieces (vit_gemm_persistent.hip): workspace lent by the caller (NULL: off), piece length
    void *sk_ws;     // device side of the workspace handle: [4 KB of flags and counters][sk_slots x 64 KB]
    int sk_slots;
    int sk_x;
    int sk_late;     // tests: helpers run their pieces last (vithip_gemm_args.handover_test)
    int sk_gen;      // launch number on this workspace (1 .. 2^28): flag words of other generations read as empty
    // LayerNorm fold, consumer side (EPI_BIAS_LN, EPI_BIAS_GELU_LN): (rstd, mean) per row of A, column sums of the folded W
    const float *ln_rows;
    const float *ln_colsum;
    // producer side (EPI_RESIDUAL_STATS): [N / 64][M][2] partial (sum, sum of squares) of the stored rows per 64-column strip;
    // stats_in_epilogue: set by the dispatcher when the kernel it chose writes them (the caller then only finalises)
    float *row_partials;
    int stats_in_epilogue;
};

// erf(x) = sign(x) * (1 - exp(t*q(t))), t = min(|x|, 4), q = degree-7 minimax fit of log(erfc(t))/t
// (fitted against scipy.special.erf; max |error| 1.2e-7 when evaluated in fp32, i.e. fp32 rounding
// noise -- libm's erff costs ~45 VALU instructions and two divergent branches per element, which made
// the fc1 epilogue 10 % of that GEMM; this form is 14 straight-line instructions).
__device__ __forceinline__ float erf_fp32(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    float q = -3.144179208902642e-05f;
    q = fmaf(q, t, 3.0881378916092217e-04f);
    q = fmaf(q, t, -1.0324339382350445e-03f);
    q = fmaf(q, t, -5.368884885683656e-04f);
    q = fmaf(q, t, 1.95839274674654e-02f);
    q = fmaf(q, t, -1.0291960090398788e-01f);
    q = fmaf(q, t, -6.36597752571106e-01f);
    q = fmaf(q, t, -1.128380298614502f);
    const float e = __builtin_amdgcn_exp2f(q * t * 1.4426950408889634f);  // v_exp_f32
    return copysignf(1.0f - e, x);
}

// GELU of the reference: 0.5f * x * (1.0f + erff(x / sqrtf(2.0f)))  (ViT_seq.c:231-233)
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erf_fp32(x * 0.70710678118654752440f));
}

// GELU of 8 values in lock-step.  One erf is a chain of ~20 DEPENDENT VALU instructions; hipcc
// schedules consecutive calls one chain after the other, and a dependent fp32 op beside a busy matrix
// pipe issues every ~13 cycles (measured: 19k cycles for the 64 values of a lane, 9 % of fc1).  Eight
// chains advanced together -- the scheduling fences keep the compiler from re-serialising them --
// issue back to back.
__device__ __forceinline__ void gelu_erf_x8(float (&y)[8]) {
    // the same arithmetic as gelu_erf() value by value (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 round like their scalar
    // forms), two values per instruction where a packed form exists: the polynomial, the scalings and the final products --
    // 21 VALU instructions per pair instead of 38.  Every one of them is paid for in fp32 matrix-pipe time (item 4 of DESIGN 4.1).
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 yy[4], u[4], t[4], q[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        yy[v] = v2{y[2 * v], y[2 * v + 1]};
        u[v] = yy[v] * v2{0.70710678118654752440f, 0.70710678118654752440f};
        t[v] = v2{fminf(fabsf(u[v][0]), 4.0f), fminf(fabsf(u[v][1]), 4.0f)};
        q[v] = __builtin_elementwise_fma(v2{-3.144179208902642e-05f, -3.144179208902642e-05f}, t[v],
                                         v2{3.0881378916092217e-04f, 3.0881378916092217e-04f});
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr float c[6] = {-1.0324339382350445e-03f, -5.368884885683656e-04f, 1.95839274674654e-02f,
                            -1.0291960090398788e-01f, -6.36597752571106e-01f, -1.128380298614502f};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int v = 0; v < 4; ++v) q[v] = __builtin_elementwise_fma(q[v], t[v], v2{c[k], c[k]});
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) q[v] = q[v] * t[v] * v2{1.4426950408889634f, 1.4426950408889634f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) q[v] = v2{__builtin_amdgcn_exp2f(q[v][0]), __builtin_amdgcn_exp2f(q[v][1])};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const v2 e = v2{1.0f, 1.0f} - q[v];
        const v2 sgn = v2{copysignf(e[0], u[v][0]), copysignf(e[1], u[v][1])};
        const v2 r = v2{0.5f, 0.5f} * yy[v] * (v2{1.0f, 1.0f} + sgn);
        y[2 * v] = r[0];
        y[2 * v + 1] = r[1];
    }
}

// GELU for a bf16 destination, two values per instruction (v_pk_fma_f32).
//   gelu(y) = max(y, 0) - h(|y|),   h = 0.5 |y| erfc(|y|/sqrt2) = t * exp2(Q(t)),  t = min(|y|, 4 sqrt2),
//   Q(t) = log2(erfc(t/sqrt2)) - 1 as a degree-6 polynomial in |y| itself (the fit on [0,4] of erfc's argument with the
//   powers of 1/sqrt2 folded into the coefficients: no scaling multiply, and |y| is a source modifier of the v_min).
// Max abs error 4.7e-6, at most 4.4 % of half a bf16 ulp of the result anywhere: invisible after the bf16
// rounding that follows, at 6.5 VALU instructions per value (6 + 1 v_pk_fma_f32, v_min, v_max, v_exp_f32 per pair and
// value) instead of 19 for the fp32-accurate form above
// (ViT_seq.c:231-233 is the fp32 definition; the bf16 path's parity bar is in tests/test_gpu_bf16.py).
__device__ __forceinline__ f32x2 gelu_bf16_x2(f32x2 y) {
    constexpr float kClamp = 5.656854249492381f;  // 4 sqrt2
    const f32x2 t = f32x2{fminf(__builtin_fabsf(y.x), kClamp), fminf(__builtin_fabsf(y.y), kClamp)};
    constexpr float c[7] = {-1.000037431716919f,   -1.1505244970321655f,   -0.4605136811733246f,  -0.05179140716791153f,
                            0.007414536084979773f, -0.0006474481779150665f, 2.524326555430889e-05f};
    f32x2 q = f32x2{c[6], c[6]};
#pragma unroll
    for (int k = 5; k >= 0; --k) q = __builtin_elementwise_fma(q, t, f32x2{c[k], c[k]});
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    return __builtin_elementwise_fma(-t, e, __builtin_elementwise_max(y, f32x2{0.f, 0.f}));
}

// The same values for NP pairs advanced TOGETHER (the scheduling fences keep hipcc from putting the chains back one after the other):
// one pair is a chain of dependent v_pk_fma_f32, and on gfx950 a packed fma that reads the previous one's result needs a wait state --
// hipcc fills it with `s_nop 0`, 5 per pair in the one-pair form (1,255 in the GELU kernel: a quarter of its vector issue slots beside
// the 9 v_pk_fma_f32, 2 v_exp_f32, 4 v_min / v_max and the conversion of a pair).  With a second chain in between there is nothing to pad.
template <int NP>
__device__ __forceinline__ void gelu_bf16_lockstep(f32x2 (&y)[NP]) {
    constexpr float kClamp = 5.656854249492381f;  // 4 sqrt2
    constexpr float c[7] = {-1.000037431716919f,   -1.1505244970321655f,   -0.4605136811733246f,  -0.05179140716791153f,
                            0.007414536084979773f, -0.0006474481779150665f, 2.524326555430889e-05f};
    f32x2 t[NP], q[NP];
#pragma unroll
    for (int v = 0; v < NP; ++v) {
        t[v] = f32x2{fminf(__builtin_fabsf(y[v].x), kClamp), fminf(__builtin_fabsf(y[v].y), kClamp)};
        q[v] = __builtin_elementwise_fma(f32x2{c[6], c[6]}, t[v], f32x2{c[5], c[5]});
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 4; k >= 0; --k) {
#pragma unroll
        for (int v = 0; v < NP; ++v) q[v] = __builtin_elementwise_fma(q[v], t[v], f32x2{c[k], c[k]});
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int v = 0; v < NP; ++v) q[v] = f32x2{__builtin_amdgcn_exp2f(q[v].x), __builtin_amdgcn_exp2f(q[v].y)};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < NP; ++v) y[v] = __builtin_elementwise_fma(-t[v], q[v], __builtin_elementwise_max(y[v], f32x2{0.f, 0.f}));
}

// Workgroup id -> tile id such that ids sharing an XCD (id % 8) get consecutive tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return base + idx;
}

// Tile id -> (tm, tn).  Tiles are walked in groups of `group_m` tile rows: inside a group the id
// runs down the rows first, then across the columns, so the ~64 workgroups resident on one XCD
// cover a compact group_m x (64/group_m) block of tiles and share both their A row-panels and their
// W column-panels through that XCD's 4 MB L2 (instead of sweeping all of W per A panel).
__device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, int group_m, int &tm, int &tn) {
    const int per_group = group_m * tiles_n;
    const int g = tile / per_group, in_g = tile - g * per_group;
    const int first = g * group_m;
    const int rows = tiles_m - first < group_m ? tiles_m - first : group_m;
    tn = in_g / rows;
    tm = first + (in_g - tn * rows);
}


// ---- epilogue ---------------------------------------------------------------------------------
// Stores one workgroup tile.  A lane holds column n of 16 rows per accumulator; each store
// instruction writes two 128-B row segments.  vmcnt counts stores as well as loads, so any load that
// is waited for between stores serialises them on the full write latency (measured: 70k cycles per
// tile, 29 % of the kernel).  Hence: the bias is passed in registers (fetched long before), interior
// tiles take a branch-free path, and the residual / pos_emb operands of one accumulator are loaded
// in batches before their stores.
// LayerNorm fold, consumer side (EPI_BIAS_LN / EPI_BIAS_GELU_LN): what the epilogue needs besides the bias
template <int TN, int TM = 1>
struct FoldOperands {
    float colsum[TN];   // column sums of the folded weight at this lane's columns (fetched with the bias, long before)
    const f32x2 *rows;  // (rstd, mean) of the tile's rows m0, m0 + 1, ...: p.ln_rows + 2 m0, or the persistent walk's LDS copy of them
    // one tile per workgroup: the lane's own values, fetched with the bias when the kernel starts -- the bias's rule: no load
    // pending in the epilogue (17 per 32-row block otherwise); `preloaded` says so
    float rstd[TM][16], mean[TM];
    bool preloaded;
};
// the lane's (rstd of its 16 rows, mean of row r) per 32-row block, for FoldOperands::rstd / mean
template <int TN, int TM>
__device__ __forceinline__ void fold_preload(FoldOperands<TN, TM> &fold, const float *ln_rows, int M, int m_tile, int r, int h) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mr = m_tile + i * 32 + r;
        fold.mean[i] = ln_rows[2 * (size_t)(mr < M ? mr : M - 1) + 1];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int m = m_tile + i * 32 + 4 * h + (v & 3) + 8 * (v >> 2);
            fold.rstd[i][v] = ln_rows[2 * (size_t)(m < M ? m : M - 1)];
        }
    }
    fold.preloaded = true;
}
// The fold's value, the same two fused multiply-adds in every kernel:
//     acc' = fma(-mean, colsum, acc)     x . (gamma W)^T - mean * colsum(gamma W) = (x - mean) . (gamma W)^T
//     y    = fma(rstd, acc', bias_f)
// The 32x32 kernels take the first one as ONE more v_mfma_f32_32x32x2_f32 per accumulator -- a rank-1 update with the k = 0
// operands (-mean of the lane's row, colsum of the lane's column) and zeros at k = 1: the instruction adds its two products to
// the accumulator one after the other, each with one rounding (tools/probes/mfma_order_probe.hip), so the result is
// fma(0, 0, fma(-mean, colsum, acc)) = the line above -- four matrix instructions per wave and tile instead of 64 vector ones.
__device__ __forceinline__ float fold_center(float acc, float mean, float colsum) { return __builtin_fmaf(-mean, colsum, acc); }
__device__ __forceinline__ float fold_scale(float centered, float rstd, float bias) { return __builtin_fmaf(rstd, centered, bias); }

#ifndef FOLD_HOIST
#define FOLD_HOIST 1  // centred-weight fold: read the rstd of all the lane's rows before the first block's values (A/B switch)
#endif
template <int BM, int BN, int WM, int WN, int EPI, int AMODE>
__device__ __forceinline__ void epilogue_store(const GemmParams &p, const f32x16 (&acc)[WM / 32][WN / 32],
                                               const float (&bias_r)[WN / 32], int m0, int n0, int wm, int wn,
                                               int r, int h, const FoldOperands<WN / 32, WM / 32> &fold = FoldOperands<WN / 32, WM / 32>{}) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr bool FOLD = EPI == EPI_BIAS_LN || EPI == EPI_BIAS_GELU_LN;
    constexpr bool GELU = EPI == VITHIP_EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_LN;
    constexpr int ST_NT = 2;  // cache policy of the bias / GELU stores: non-temporal (qkv and the hidden layer are read by the NEXT launch)
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);  // workgroup-uniform
    if (interior) {
        if constexpr (FOLD) {
            // acc' = acc - mean * colsum as one rank-1 MFMA per accumulator: A operand = -mean of row r of the block (k = 0, lanes
            // h = 0; the h = 1 lanes supply k = 1: zero), B operand = colsum of column r.  One 32-row block at a time: its two
            // matrix instructions, the 16 rstd of the lane's rows (LDS reads in the persistent walk -- fetched once per block:
            // behind the scheduling fences of the GELU batches every batch would wait for its own), then the batches.
            const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7fffffff, 0x00020000);
            // Centred weights (p.ln_colsum == NULL, vithip_ln_fold_weights_f32_centered): the accumulator IS the centred product and
            // the block goes straight to the scaling.  (With the colsum 0 the matrix instruction below would return the accumulator bit
            // for bit -- fma(-mean, 0, acc) -- so skipping it changes no result, in this kernel or against the ones that do not skip.)
            // What the skip is worth is not the four instructions (0.26 % of a K = 768 tile) but where they sit: in the epilogue of one
            // workgroup while the CU's other workgroup saturates the matrix pipe, each waits its turn and the FMAs wait for it -- stamps
            // put the fold's epilogue at 9.5k cycles per QKV tile against 5.1k for the plain one (round 5).
            const bool center = p.ln_colsum != nullptr;  // workgroup-uniform
            // Two code paths, chosen once per tile (a copy "centered = acc" on the path that does not centre cost 60 registers).
            // HOIST: the rstd of ALL the lane's rows first (LDS reads in the persistent walk) -- one exposed LDS round trip per tile
            // instead of one per 32-row block: they queue behind the fragment reads of the CU's other workgroup, and the scheduling
            // fences of the GELU batches keep the second block's reads from being issued before the first block's stores.
            auto rstd_of_block = [&](int i, float (&dst)[16]) __attribute__((always_inline)) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lr = wm * WM + i * 32 + 4 * h + 8 * g;  // tile-local row of registers 4g .. 4g + 3
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[4 * g + e] = fold.preloaded ? fold.rstd[i][4 * g + e] : fold.rows[lr + e].x;
                }
            };
            auto scale_and_store = [&](int i, const f32x16 (&src)[TN], const float (&row_rstd)[16]) __attribute__((always_inline)) {
                const int mb = m0 + wm * WM + i * 32 + 4 * h;  // row of register 0
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WN + j * 32 + r;
                    const int c_off = (mb * p.ldc + n) * 4, c_row = p.ldc * 4;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {  // two batches of 8: values first (eight erf chains in lock-step), then 8 stores
                        float y[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) y[q] = fold_scale(src[j][half * 8 + q], row_rstd[half * 8 + q], bias_r[j]);
                        if constexpr (GELU) gelu_erf_x8(y);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int v = half * 8 + q;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[q]), c_rsrc, c_off, ((v & 3) + 8 * (v >> 2)) * c_row, ST_NT);
                        }
                    }
                }
            };
            if (center) {
                float b_op[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) b_op[j] = h == 0 ? fold.colsum[j] : 0.0f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float mean = fold.preloaded ? fold.mean[i] : fold.rows[wm * WM + i * 32 + r].y;
                    const float a_op = h == 0 ? -mean : 0.0f;
                    f32x16 centered[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) centered[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op, b_op[j], acc[i][j], 0, 0, 0);
                    float row_rstd[16];
                    rstd_of_block(i, row_rstd);
                    scale_and_store(i, centered, row_rstd);
                }
            } else {
                float row_rstd[TM][16];
#pragma unroll
                for (int i = 0; i < (FOLD_HOIST ? TM : 0); ++i) rstd_of_block(i, row_rstd[i]);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (!FOLD_HOIST) rstd_of_block(i, row_rstd[i]);
                    scale_and_store(i, acc[i], row_rstd[i]);
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + r;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * WM + i * 32 + 4 * h;  // row of register 0
                if constexpr (AMODE == A_PATCHES) {
                    float add[16];
                    size_t orow[16];
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int m = mb + (v & 3) + 8 * (v >> 2);
                        const int im = m / p.patches, pp = m - im * p.patches;
                        add[v] = p.pos[(size_t)(pp + 1) * p.N + n];
                        orow[v] = (size_t)m + im + 1;
                    }
#pragma unroll
                    for (int v = 0; v < 16; ++v) p.C[orow[v] * p.ldc + n] = acc[i][j][v] + bias_r[j] + add[v];
                } else {
                    // Buffer addressing: SGPR descriptor + ONE per-lane byte offset per accumulator + the row's offset as the
                    // instruction's scalar operand.  With 64-bit pointers every one of the 16 rows cost vector address
                    // arithmetic (matrix-pipe time on gfx950) and a register pair; the persistent residual kernel spilled.
                    // The launcher keeps C and the residual below 2 GiB.
                    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7fffffff, 0x00020000);
                    const int c_off = (mb * p.ldc + n) * 4, c_row = p.ldc * 4;
                    if constexpr (EPI == VITHIP_EPI_BIAS_RESIDUAL) {
                        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.R), 0, 0x7fffffff, 0x00020000);
                        const int r_off = (mb * p.ldr + n) * 4, r_row = p.ldr * 4;
                        // two batches of 8 loads-then-stores: enough in flight, half the registers
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            float res[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int v = half * 8 + q;
                                res[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, r_off, ((v & 3) + 8 * (v >> 2)) * r_row, 0));
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int v = half * 8 + q;
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i][j][v] + bias_r[j] + res[q]), c_rsrc, c_off,
                                                                      ((v & 3) + 8 * (v >> 2)) * c_row, 0);
                            }
                        }
                    } else {
                        // two batches of 8: values first (eight erf chains in lock-step), then 8 stores
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            float y[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) y[q] = acc[i][j][half * 8 + q] + bias_r[j];
                            if constexpr (GELU) gelu_erf_x8(y);
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int v = half * 8 + q;
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[q]), c_rsrc, c_off, ((v & 3) + 8 * (v >> 2)) * c_row, ST_NT);
                            }
                        }
                    }
                }
            }
        }
    } else {
        // edge tiles (last partial M tile, N not a multiple of the tile): per-element guards
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + r;
            const bool n_ok = n < p.N;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = m0 + wm * WM + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                    if (n_ok && m < p.M) {
                        float y;
                        if constexpr (FOLD) {
                            const f32x2 row = fold.rows[m - m0];
                            y = fold_scale(fold_center(acc[i][j][v], row.y, fold.colsum[j]), row.x, bias_r[j]);
                        }
                        else y = acc[i][j][v] + bias_r[j];
                        if constexpr (AMODE == A_PATCHES) {
                            const int im = m / p.patches, pp = m - im * p.patches;
                            y += p.pos[(size_t)(pp + 1) * p.N + n];
                            p.C[((size_t)m + im + 1) * p.ldc + n] = y;
                        } else {
                            if constexpr (GELU) y = gelu_erf(y);
                            if constexpr (EPI == VITHIP_EPI_BIAS_RESIDUAL) y += p.R[(size_t)m * p.ldr + n];
                            p.C[(size_t)m * p.ldc + n] = y;
                        }
                    }
                }
            }
        }
    }
}

// ---- bf16 GEMM (vit_gemm_bf16.hip: launcher + two-stage kernel; vit_gemm_bf16_pp.hip: ping-pong kernel)
struct Bf16Params {
    const unsigned short *A;
    const unsigned short *W;
    const float *bias;
    const float *R;
    void *C;
    int lda, ldw, ldr, ldc;
    int M, N, K;
    int tiles_m, tiles_n, group_m;
    unsigned long long *dbg;  // stamped probe build only
    int patches;              // F32_EMBED epilogue: patches per image
    // LayerNorm folded into the GEMMs either side of it (ping-pong kernel only; vithip_gemm_bf16_args has the contract):
    const float *ln_rows;     // consumer (BF16 / BF16_GELU): [M][2] = (rstd, mean * rstd) of every row of A
    const float *ln_colsum;   // consumer: [N] column sums of the gamma-folded weight
    unsigned short *x16;      // producer (F32_RESIDUAL): bf16 copy of C, row stride ldx16
    int ldx16;
    float *partials;          // producer: [4 * tiles_n][M][2] partial (sum, sum of squares) per row and 64-column strip
};
int launch_gemm_bf16_pp(hipStream_t stream, const Bf16Params &p, int epilogue, int cus);
int launch_gemm_bf16_w4(hipStream_t stream, const Bf16Params &p, int epilogue, int cus);  // probe build: tools/probes/vit_gemm_bf16_w4.hip
// vit_gemm_latency.hip: 32x32 workgroup tiles on v_mfma_f32_16x16x4_f32 (bit-identical to the 32x32x2 kernels); K % 128 == 0
int launch_gemm_f32_latency(hipStream_t stream, GemmParams &p, int epilogue);
// vit_patch_embed_bf16.hip: patch embedding on the bf16 pipe as one implicit GEMM over the NCHW fp32 images
int launch_patch_embed_bf16(hipStream_t s, const float *images, const unsigned short *conv_w16, const float *conv_b,
                            const float *cls, const float *pos, float *x, int n_images, int img_size, int patch_size,
                            int in_chans, int embed_dim);

// ---- residual epilogue that also leaves the row statistics' partial sums (LayerNorm fold, producer side) ---------------------
// 16 per-lane values -> lane r of each 32-lane half holds the total of value (r >> 1) & 15 over the half's lanes, added in the
// butterfly order 16, 8, 4, 2, 1 (own + partner at every level): the order of vithip_rowstats_f32, at 15 + 1 exchanges instead
// of 16 x 5.
__device__ __forceinline__ float reduce_scatter16(const float (&a)[16], int r) {
    float b[8], c[4], d[2], e;
    {
        const bool up = (r & 16) != 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = (up ? a[k + 8] : a[k]) + __shfl_xor(up ? a[k] : a[k + 8], 16);
    }
    {
        const bool up = (r & 8) != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = (up ? b[k + 4] : b[k]) + __shfl_xor(up ? b[k] : b[k + 4], 8);
    }
    {
        const bool up = (r & 4) != 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) d[k] = (up ? c[k + 2] : c[k]) + __shfl_xor(up ? c[k] : c[k + 2], 4);
    }
    {
        const bool up = (r & 2) != 0;
        e = (up ? d[1] : d[0]) + __shfl_xor(up ? d[0] : d[1], 2);
    }
    return e + __shfl_xor(e, 1);
}

// C = acc + bias + residual as epilogue_store<EPI_BIAS_RESIDUAL> stores it, and p.row_partials[strip][m] = (sum, sum of squares)
// of the stored values of row m over the wave's 64-column strip, in the documented order of vithip_rowstats_f32: column c + l and
// c + 32 + l first (the lane's two accumulator columns), then the butterfly.  WN = 64 (one strip per wave), N % BN == 0.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epilogue_store_residual_stats(const GemmParams &p, const f32x16 (&acc)[WM / 32][WN / 32],
                                                              const float (&bias_r)[WN / 32], int m0, int n0, int wm, int wn,
                                                              int r, int h) {
    static_assert(WN == 64, "one 64-column strip per wave");
    constexpr int TM = WM / 32;
    const bool interior = m0 + BM <= p.M;  // workgroup-uniform
    const int strip = (n0 + wn * WN) >> 6;
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.R), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * WM + i * 32 + 4 * h;  // row of register 0
        float y[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * WN + j * 32 + r;
            const int c_off = (mb * p.ldc + n) * 4, c_row = p.ldc * 4, r_off = (mb * p.ldr + n) * 4, r_row = p.ldr * 4;
            if (interior) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {  // two batches of 8 loads-then-stores, as in epilogue_store
                    float res[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int v = half * 8 + q;
                        res[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, r_off, ((v & 3) + 8 * (v >> 2)) * r_row, 0));
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int v = half * 8 + q;
                        y[j][v] = acc[i][j][v] + bias_r[j] + res[q];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[j][v]), c_rsrc, c_off, ((v & 3) + 8 * (v >> 2)) * c_row, 0);
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = mb + (v & 3) + 8 * (v >> 2);
                    y[j][v] = 0.0f;
                    if (m < p.M) {
                        y[j][v] = acc[i][j][v] + bias_r[j] + p.R[(size_t)m * p.ldr + n];
                        p.C[(size_t)m * p.ldc + n] = y[j][v];
                    }
                }
            }
        }
        float u[16], w[16];
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                u[v] = y[0][v] + y[1][v];
                const float sq0 = y[0][v] * y[0][v];
                w[v] = __builtin_fmaf(y[1][v], y[1][v], sq0);
            }
        }
        const float us = reduce_scatter16(u, r), ws = reduce_scatter16(w, r);
        const int v = (r >> 1) & 15;
        const int m = mb + (v & 3) + 8 * (v >> 2);
        if (!(r & 1) && m < p.M) *reinterpret_cast<f32x2 *>(p.row_partials + ((size_t)strip * p.M + m) * 2) = f32x2{us, ws};
    }
}

}  // namespace vitgemm
