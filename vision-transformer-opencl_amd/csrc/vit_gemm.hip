// csrc/vit_gemm.hip -- fp32 "NT" GEMM on the CDNA4 matrix cores with fused epilogues.
//
// Covers every linear-shaped stage of the reference's forward (SURVEY.md 8a rows a1, a6, a8,
// a10, a11, a12): C = epilogue(A[M][K] . W[N][K]^T + bias).  The reference runs these as
// 16x16-tiled / one-work-item-per-output OpenCL kernels with a separate bias, GELU and CPU
// residual pass (kernel.cl:208-284,374-533; ViT_opencl.c:369-380,662-672,758-777); here one
// kernel per layer does the contraction on v_mfma_f32_32x32x2_f32 and applies bias, exact-erf
// GELU (ViT_seq.c:231-233) or the residual add (ViT_seq.c:286-288,297-299) in registers.
//
// Design (gfx950):
//  * Workgroup = 256 threads = 4 waves (one per SIMD); tile BM x BN, K step 32.  A wave owns a
//    WM x WN sub-tile as (WM/32)x(WN/32) 32x32 accumulators.
//  * Both operands are K-contiguous, so a lane fetches its MFMA fragments as one ds_read_b128:
//    lane l (r = l&31, h = l>>5) reads 4 consecutive k of row r at k-offset 8c+4h.  MFMA step s
//    then multiplies k = 8c+s (lanes 0-31) and k = 8c+4+s (lanes 32-63) -- the same k
//    permutation on A and W, which a dot product does not care about.  4 MFMAs (256 cycles of
//    matrix pipe) per pair of LDS reads: the kernel is MFMA-issue bound by construction.
//  * LDS rows are padded to 36 floats: the four 16-lane groups of a ds_read_b128 then touch
//    16 distinct 4-bank slots (conflict-free), and the 8-lane groups of the staging
//    ds_write_b128 are contiguous.
//  * Register-staged double buffering: global loads of tile t+1 are issued before the MFMAs
//    of tile t and written to the other LDS buffer after them; one barrier per K step.
//  * 2 workgroups per CU (72 KB LDS each at 128x128) so one group's barrier/epilogue hides
//    under the other's MFMAs.
//  * Workgroup ids are remapped so that the ids that share an XCD (id mod 8) walk one
//    contiguous run of tiles, N fastest: the A row-panel of a tile row is fetched into that
//    XCD's L2 once and reused by all its N tiles.
//  * The implicit-GEMM variant (patch embedding) gathers A straight from NCHW images and
//    fuses "+ pos_emb" and the token-row remap into the store.
#include <new>

#include "vit_gemm_common.hpp"
#ifdef VIT_PROBES
#include "vit_probes.h"
#endif

namespace vitgemm {
int launch_persistent(hipStream_t stream, GemmParams &p, int epilogue, int group_m);  // vit_gemm_persistent.hip
int launch_persistent_stamped(hipStream_t stream, GemmParams &p, int epilogue, int group_m);
int launch_persistent_switchoff(hipStream_t stream, GemmParams &p, int epilogue, int group_m, int dbg);  // probe build
int persistent_piece_steps(int M, int N, int K, int slots, int wgs);
}

namespace {

using namespace vitgemm;

// The product library carries NO mutable process-wide state: tile shape and L2 group size are per-call fields of
// vithip_gemm_args (0 = auto).  Only the probe build (make probes, -DVIT_PROBES, libvit_mi355x_probe.so) has the
// process-wide overrides the tools/ scripts flip between timings, and the timing-only kernel variants.
#ifdef VIT_PROBES
int g_gemm_tile = 0;   // 0 = no override; see vithip_gemm_set_tile()
int g_gemm_group = 0;  // 0 = no override; see vithip_gemm_set_group()
unsigned long long *g_gemm_dbg = nullptr;  // stamp buffer of the DBG == 5 probe
#endif

// DBG != 0 builds timing-only probes (tools/gemm_probe.py): 1 = no global loads and no LDS stores in
// the K loop, 2 = no global loads, 3 = no LDS stores, 4 = as 1 without the barrier.  Results are wrong
// by construction; they bound what the MFMA + LDS-read stream can reach.
// BK = K step per LDS tile (32 or 16).  LDS rows are padded to BK + 4 floats: 16-B aligned and
// conflict-free for ds_read_b128 at both sizes (row strides of 36 and 20 dwords).
template <int BM, int BN, int WM, int WN, int EPI, int AMODE, int DBG = 0, int BK = 32, bool PIPE = false>
__global__ __launch_bounds__(256, (BK == 16 ? 3 : 1)) void gemm_f32_nt_kernel(const GemmParams p) {
    constexpr int LDS_LD = BK + 4;
    constexpr int ROWS_PER_PASS = 256 / (BK / 4);  // tile rows covered by one staging load per thread
    constexpr int WGN = BN / WN;           // waves along N
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_CHUNKS = BM * (BK / 4) / 256;  // float4 per thread per A tile
    constexpr int B_CHUNKS = BN * (BK / 4) / 256;
    static_assert((BM / WM) * WGN == 4, "4 waves per workgroup");
    static_assert(A_CHUNKS >= 1 && B_CHUNKS >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];
    float *const As0 = lds;
    float *const Bs0 = lds + 2 * BM * LDS_LD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    unsigned long long st_clk0 = 0, st_rt0 = 0, st_clk1 = 0, st_clk2 = 0;
    if constexpr (DBG == 5) {
        st_clk0 = __builtin_amdgcn_s_memtime();
        st_rt0 = __builtin_amdgcn_s_memrealtime();
    }

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread global source pointers for the staging loads -----------------------
    const int ld_row = tid / (BK / 4);          // + ROWS_PER_PASS per chunk
    const int ld_kc = (tid % (BK / 4)) * 4;     // float offset inside the K step
    const float *a_src[A_CHUNKS];
    const float *b_src[B_CHUNKS];
    // Dense operands are fetched with buffer loads (SGPR descriptor + 32-bit per-thread byte offset + SGPR
    // K offset): stepping through K then costs no vector instruction (fp32 VALU time is matrix-pipe time
    // on gfx950).  The launcher keeps operands below 2 GB.
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.W), 0, 0x7fffffff, 0x00020000);
    int a_boff[A_CHUNKS], b_boff[B_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        int m = m0 + ld_row + i * ROWS_PER_PASS;
        m = m < p.M ? m : p.M - 1;
        if constexpr (AMODE == A_DENSE) {
            a_src[i] = p.A + (size_t)m * p.lda + ld_kc;
            a_boff[i] = (m * p.lda + ld_kc) * 4;
        } else {
            // row m = (image, patch); k = (ic, kh, kw): resolved per K step in load_a()
            const int im = m / p.patches, pp = m - im * p.patches;
            const int oh = pp / p.grid, ow = pp - oh * p.grid;
            a_src[i] = p.A + ((size_t)im * p.chans * p.img + (size_t)oh * p.patch) * p.img + ow * p.patch;
            // pipelined loop (launcher: patch*patch % BK == 0 and BK % patch == 0 or patch % BK == 0): a thread's four k stay in one
            // pixel row and its (kh, kw) inside the K step never change, so the gather is (per-thread byte offset) + (scalar offset
            // of the K step) like a dense operand -- buffer loads, no vector address arithmetic in the loop
            a_boff[i] = (int)(((((size_t)im * p.chans * p.img + (size_t)oh * p.patch + ld_kc / p.patch) * p.img) + ow * p.patch + ld_kc % p.patch) * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
        int n = n0 + ld_row + i * ROWS_PER_PASS;
        n = n < p.N ? n : p.N - 1;
        b_src[i] = p.W + (size_t)n * p.ldw + ld_kc;
        b_boff[i] = (n * p.ldw + ld_kc) * 4;
    }

    f32x4 a_stage[A_CHUNKS], b_stage[B_CHUNKS];

    auto load_global = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            if constexpr (AMODE == A_DENSE) {
                a_stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_boff[i], k0 * 4, 0));
            } else {
                const int k = k0 + ld_kc;
                const int pp2 = p.patch * p.patch;
                const int ic = k / pp2, rem = k - ic * pp2;
                const int kh = rem / p.patch, kw = rem - kh * p.patch;
                a_stage[i] = *reinterpret_cast<const f32x4 *>(
                    a_src[i] + ((size_t)ic * p.img + kh) * p.img + kw);
            }
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) b_stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_boff[i], k0 * 4, 0));
    };
    auto store_lds = [&](int buf) {
        float *As = As0 + buf * BM * LDS_LD;
        float *Bs = Bs0 + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i)
            *reinterpret_cast<f32x4 *>(As + (ld_row + i * ROWS_PER_PASS) * LDS_LD + ld_kc) = a_stage[i];
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i)
            *reinterpret_cast<f32x4 *>(Bs + (ld_row + i * ROWS_PER_PASS) * LDS_LD + ld_kc) = b_stage[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    // bias for this lane's columns, fetched now so that no load is pending in the epilogue
    float bias_r[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + r;
        bias_r[j] = n < p.N ? p.bias[n] : 0.0f;
    }
    FoldOperands<TN, TM> fold{};  // LayerNorm fold (consumer): the folded weight's column sums and the lane's rows' pairs beside the bias
    if constexpr (EPI == EPI_BIAS_LN || EPI == EPI_BIAS_GELU_LN) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + r;
            fold.colsum[j] = (p.ln_colsum && n < p.N) ? p.ln_colsum[n] : 0.0f;  // NULL: centred weights, nothing to subtract
        }
        fold.rows = reinterpret_cast<const f32x2 *>(p.ln_rows) + m0;
        fold_preload(fold, p.ln_rows, p.M, m0 + wm * WM, r, h);
    }

    const int nk = p.K / BK;
    const int a_frag_off = (wm * WM + r) * LDS_LD + h * 4;
    const int b_frag_off = (wn * WN + r) * LDS_LD + h * 4;
    if constexpr (PIPE) {
        // ---- software-pipelined K loop ---------------------------------------------------------
        // Per K step: the fragments of chunk c+1 are read while chunk c multiplies; the staged tile
        // t+1 is written to the other LDS buffer and the loads of tile t+2 are issued in the shadow
        // of chunk 0; the single barrier sits BEFORE the last chunk's MFMAs, and the first fragments
        // of the next tile are fetched right after it, so barrier skew and LDS latency hide under
        // 16 MFMAs instead of stalling the matrix pipe.
        constexpr int NC = BK / 8;                 // 8-deep chunks per K step
        constexpr int NS = A_CHUNKS + B_CHUNKS;    // staged float4 per thread per K step
        constexpr int NM = 4 * TM * TN;            // MFMAs per chunk
        f32x4 af[2][TM], bf[2][TN];
        auto read_frags = [&](int buf, int c, int set) {
            const float *As = As0 + buf * BM * LDS_LD + a_frag_off + c * 8;
            const float *Bs = Bs0 + buf * BN * LDS_LD + b_frag_off + c * 8;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const f32x4 *>(As + i * 32 * LDS_LD);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const f32x4 *>(Bs + j * 32 * LDS_LD);
        };
        // one staging slot: write the float4 loaded a K step ago, then reload it for two steps ahead
        auto restage_slot = [&](int q, int buf, int k0, int ka) {
            if (q < A_CHUNKS) {
                float *As = As0 + buf * BM * LDS_LD;
                *reinterpret_cast<f32x4 *>(As + (ld_row + q * ROWS_PER_PASS) * LDS_LD + ld_kc) = a_stage[q];
                a_stage[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_boff[q], ka, 0));
            } else {
                const int qb = q - A_CHUNKS;
                float *Bs = Bs0 + buf * BN * LDS_LD;
                *reinterpret_cast<f32x4 *>(Bs + (ld_row + qb * ROWS_PER_PASS) * LDS_LD + ld_kc) = b_stage[qb];
                b_stage[qb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_boff[qb], k0 * 4, 0));
            }
        };
        // byte offset of K step k0 inside a row's operand: dense k0 * 4; patches: channel k0 / P^2, pixel row (k0 % P^2) / P
        auto a_koff = [&](int k0) {
            if constexpr (AMODE == A_DENSE) return k0 * 4;
            const int pp2 = p.patch * p.patch;
            const int ic = k0 / pp2, kh = (k0 - ic * pp2) / p.patch;  // wave-uniform: scalar divisions, once per K step
            // + the K step's start inside the pixel row: 0 for patch <= 32, 32 / 64 / ... floats for patch 64, 96, ... (whose rows span
            // several K steps; the thread's own ld_kc % patch is in a_boff)
            return ((ic * p.img + kh) * p.img + (k0 - ic * pp2 - kh * p.patch)) * 4;
        };
        auto load_global_pipe = [&](int k0) {
            const int ka = a_koff(k0);
#pragma unroll
            for (int i = 0; i < A_CHUNKS; ++i) a_stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_boff[i], ka, 0));
#pragma unroll
            for (int i = 0; i < B_CHUNKS; ++i) b_stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_boff[i], k0 * 4, 0));
        };
        load_global_pipe(0);
        store_lds(0);
        load_global_pipe(nk > 1 ? BK : 0);
        __syncthreads();
        read_frags(0, 0, 0);
        if constexpr (DBG == 5) st_clk1 = __builtin_amdgcn_s_memtime();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // Branch-free body: past the end the staging just re-reads the last K step and writes a
            // buffer nobody reads again.
            const int k_ahead = (kt + 2 < nk ? kt + 2 : nk - 1) * BK;
            const int ka_ahead = a_koff(k_ahead);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c + 1 < NC) read_frags(cur, c + 1, (c + 1) & 1);
                if (c == NC - 1) {
                    __syncthreads();
                    read_frags(cur ^ 1, 0, NC & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int idx = 0; idx < NM; ++idx) {
                    const int s2 = idx / (TM * TN), i = (idx / TN) % TM, j = idx % TN;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c & 1][i][s2], bf[c & 1][j][s2], acc[i][j],
                                                                     0, 0, 0);
                    if (c == 0) {
                        // spread the NS restage slots evenly between the MFMAs of chunk 0
                        const int slot_before = (idx * NS) / NM, slot_after = ((idx + 1) * NS) / NM;
                        if (slot_after > slot_before) {
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = slot_before; q < slot_after; ++q) restage_slot(q, cur ^ 1, k_ahead, ka_ahead);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            cur ^= 1;
        }
    } else {
        load_global(0);
        store_lds(0);
        if constexpr (DBG != 0) store_lds(1);
        __syncthreads();


        if constexpr (DBG == 5) st_clk1 = __builtin_amdgcn_s_memtime();
        if constexpr (DBG >= 10) {  // placement experiment: shift the K loop by 4-byte steps
#pragma unroll
            for (int q = 0; q < DBG - 10; ++q) asm volatile("s_nop 0");
        }
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if constexpr (DBG == 0 || DBG == 3 || DBG == 5) {
                if (more) load_global((kt + 1) * BK);
            }

            const float *As = As0 + cur * BM * LDS_LD + a_frag_off;
            const float *Bs = Bs0 + cur * BN * LDS_LD + b_frag_off;
#pragma unroll
            for (int c = 0; c < BK / 8; ++c) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4 *>(As + i * 32 * LDS_LD + c * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4 *>(Bs + j * 32 * LDS_LD + c * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            }

            if constexpr (DBG == 0 || DBG == 2 || DBG == 5) {
                if (more) store_lds(cur ^ 1);
            } else if constexpr (DBG == 3) {
#pragma unroll
                for (int i = 0; i < A_CHUNKS; ++i) asm volatile("" ::"v"(a_stage[i]));
#pragma unroll
                for (int i = 0; i < B_CHUNKS; ++i) asm volatile("" ::"v"(b_stage[i]));
            }
            if constexpr (DBG != 4) __syncthreads();
            cur ^= 1;
        }

    }

    if constexpr (DBG == 5) st_clk2 = __builtin_amdgcn_s_memtime();
    epilogue_store<BM, BN, WM, WN, EPI, AMODE>(p, acc, bias_r, m0, n0, wm, wn, r, h, fold);
    if constexpr (DBG == 5) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long c3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long *d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = st_clk0; d[1] = st_clk1; d[2] = st_clk2; d[3] = c3; d[4] = st_rt0; d[5] = r3;
            d[6] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID[3:0]
            d[7] = (unsigned long long)tile;
        }
    }
}

// x[img][0][:] = cls + pos[0] (class_token + pos_emb of the reference, ViT_seq.c:72-101)
__global__ void cls_rows_kernel(const float *cls, const float *pos, float *x, int n_images, int tokens, int dim) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_images * dim) return;
    const int im = idx / dim, d = idx - im * dim;
    x[(size_t)im * tokens * dim + d] = cls[d] + pos[d];
}

template <int BM, int BN, int WM, int WN, int AMODE, int BK = 32, bool PIPE = false>
int launch_tile(hipStream_t stream, GemmParams &p, int epilogue) {
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const dim3 grid(p.tiles_m * p.tiles_n), block(256);
    if constexpr (AMODE == A_PATCHES) {
        hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS, A_PATCHES, 0, BK, PIPE>), grid, block, 0, stream, p);
    } else {
        switch (epilogue) {
            case VITHIP_EPI_BIAS:
                hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS, A_DENSE, 0, BK, PIPE>), grid, block, 0, stream, p);
                break;
            case VITHIP_EPI_BIAS_GELU:
                hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS_GELU, A_DENSE, 0, BK, PIPE>), grid, block, 0, stream, p);
                break;
            case VITHIP_EPI_BIAS_RESIDUAL:
                hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS_RESIDUAL, A_DENSE, 0, BK, PIPE>), grid, block, 0, stream, p);
                break;
            case EPI_BIAS_LN:
                hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, EPI_BIAS_LN, A_DENSE, 0, BK, PIPE>), grid, block, 0, stream, p);
                break;
            case EPI_BIAS_GELU_LN:
                hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, EPI_BIAS_GELU_LN, A_DENSE, 0, BK, PIPE>), grid, block, 0, stream, p);
                break;
            default:
                return static_cast<int>(hipErrorInvalidValue);
        }
    }
    return static_cast<int>(hipGetLastError());
}

#ifdef VIT_PROBES
template <int DBG, int EPI = VITHIP_EPI_BIAS, bool PIPE = false>
int launch_probe(hipStream_t stream, GemmParams &p) {
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = (p.N + 127) / 128;
    hipLaunchKernelGGL((gemm_f32_nt_kernel<128, 128, 64, 64, EPI, A_DENSE, DBG, 32, PIPE>),
                       dim3(p.tiles_m * p.tiles_n), dim3(256), 0, stream, p);
    return static_cast<int>(hipGetLastError());
}
#endif

template <int AMODE>
int dispatch(hipStream_t stream, GemmParams &p, int epilogue, int tile, int group_m) {
    // Tile rows per L2 group of the XCD-aware walk.  Measured per launch at batch 256 (tools/gemm_f32_traffic.py, PMC FETCH_SIZE /
    // WRITE_SIZE passes, round 4), bytes from beyond the L2s over algorithmic: N = 768 (6 tile columns: fc2 / out_proj) 3.11 / 1.58
    // with groups of 8 rows, 2.34 / 1.34 in plain N-fastest order (group 1: the 64 tiles an XCD holds at a time are ~11 whole tile
    // rows, every A panel read by one XCD instead of two); wide N (QKV 18 columns, fc1 24) 2.90 / 2.82 with 8, 2.81 / 2.71 with 4,
    // 3.6-4.6 with 1, 2 or 16.  The launch TIMES do not move with any of this (+-0.3 %, same tool): the extra reads are served by
    // the Infinity Cache, not by HBM -- the choice is bytes, not milliseconds.
    const int cols128 = (p.N + 127) / 128;
    p.group_m = group_m > 0 ? group_m : (cols128 <= 8 ? 1 : 4);
#ifdef VIT_PROBES
    if (g_gemm_tile) tile = g_gemm_tile;
    if (g_gemm_group) p.group_m = g_gemm_group;
    if constexpr (AMODE == A_DENSE) {
        switch (tile) {  // timing-only probes, never selected by the engine
            case 101: return launch_probe<1>(stream, p);
            case 102: return launch_probe<2>(stream, p);
            case 103: return launch_probe<3>(stream, p);
            case 104: return launch_probe<4>(stream, p);
            case 105: p.dbg = g_gemm_dbg; return g_gemm_dbg ? launch_probe<5>(stream, p) : static_cast<int>(hipErrorInvalidValue);
            case 129: p.dbg = g_gemm_dbg; return g_gemm_dbg ? vitgemm::launch_persistent_stamped(stream, p, epilogue, p.group_m) : static_cast<int>(hipErrorInvalidValue);
            case 125: p.dbg = g_gemm_dbg; return g_gemm_dbg ? launch_probe<5, VITHIP_EPI_BIAS, true>(stream, p) : static_cast<int>(hipErrorInvalidValue);
            case 126: p.dbg = g_gemm_dbg; return g_gemm_dbg ? launch_probe<5, VITHIP_EPI_BIAS_GELU, true>(stream, p) : static_cast<int>(hipErrorInvalidValue);
            case 131: case 132: case 133: case 134: case 135: case 136:  // persistent walk with one part switched off
                return vitgemm::launch_persistent_switchoff(stream, p, epilogue, p.group_m, tile - 130);
            default: break;
        }
    }
#endif
    if constexpr (AMODE == A_PATCHES) {
        // 0.7 % of the FLOPs (the gather indices and the token-row remap of the fused epilogue cost ~40 VGPRs on top of the dense
        // kernel): 128x64 tiles
        // pipelined loop when a K step of 32 is a whole number of pixel rows or a part of one (patch 8, 16, 32, ...) and the images
        // are addressable by 32-bit byte offsets; the classic (not software-pipelined) loop otherwise
        const bool pipe_ok = (32 % p.patch == 0 || p.patch % 32 == 0) && (p.patch * p.patch) % 32 == 0 &&
                             (size_t)(p.M / p.patches + 1) * p.chans * p.img * p.img * 4 < 0x7fffffffull;
#ifdef VIT_PROBES  // tools/embed_f32_time.py: other tile shapes of the gather
        if (tile == 1) return launch_tile<128, 128, 64, 64, A_PATCHES>(stream, p, epilogue);
        if (tile == 2) return launch_tile<256, 128, 128, 64, A_PATCHES>(stream, p, epilogue);
        if (tile == 10) return launch_tile<128, 128, 64, 64, A_PATCHES, 32, true>(stream, p, epilogue);
        if (tile == 3) return launch_tile<128, 64, 64, 32, A_PATCHES>(stream, p, epilogue);
#endif
        if (!pipe_ok) return launch_tile<128, 64, 64, 32, A_PATCHES>(stream, p, epilogue);
        // measured at batch 256 (tools/embed_f32_time.py, two interleaved rounds): classic 128x64 0.556 ms, pipelined 128x128
        // 0.504-0.556, pipelined 128x64 0.499-0.531 = 118 TFLOP/s (0.75 of the fp32 matrix peak): every 64-wide N tile re-reads its
        // pixels through the L2s, 12 x 154 MB per launch
        return launch_tile<128, 64, 64, 32, A_PATCHES, 32, true>(stream, p, epilogue);
    }
    switch (tile) {
        case 9: return vitgemm::launch_persistent(stream, p, epilogue, p.group_m);   // persistent, cross-tile pipelined
        case 6: return launch_tile<128, 128, 64, 64, AMODE, 16, true>(stream, p, epilogue);
        case 7: return launch_tile<256, 128, 128, 64, AMODE, 32, true>(stream, p, epilogue);
        case 8: return launch_tile<128, 64, 64, 32, AMODE, 32, true>(stream, p, epilogue);
        case 10: return launch_tile<128, 128, 64, 64, AMODE, 32, true>(stream, p, epilogue);  // pipelined
        case 11: return launch_tile<64, 64, 32, 32, AMODE, 32, true>(stream, p, epilogue);    // one 32x32 accumulator per wave
        case 12:  // latency tile: 16x16 per wave on 16x16x4 MFMA (vit_gemm_latency.hip)
            if (p.K % 128) return static_cast<int>(hipErrorInvalidValue);
            return vitgemm::launch_gemm_f32_latency(stream, p, epilogue);
        default: {
            // auto.  Large problems: the persistent walk wins where the epilogue is light on registers (bias, bias+GELU:
            // fc1 22.0 vs 23.0 ms per step); the residual epilogue needs 255 VGPRs there and is faster one tile per
            // workgroup (fc2 21.7 vs 22.5 ms).  Small problems (few rounds of 128x128 tiles over the 512 workgroup
            // slots) are tile-quantisation-bound: smaller tiles multiply the workgroups (rounds 1-4: 128x64 below 2,048 tiles; round 5,
            // below: 64x64, and the walk from 1,280).  The thresholds leave every GEMM of the batch-256 metric (>= 1182 tiles per lane)
            // on the large-problem kernels.
            const long tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
            // Latency regime (the reference's own use is ONE image, Main.c:45-46): when even 128x64 tiles leave CUs without
            // a workgroup, the time of a GEMM is one wave's K loop -- 2 accumulators x K/2 MFMAs of 64 cycles -- so 64x64
            // tiles (one 32x32 accumulator per wave) halve it and quadruple the workgroups: fc2 of one image 110 -> 50 us.
            // Every output still sums its k in the same order with the same instruction: results are bit-identical.
            // Fewer than ~160 tiles of 64x64 (one image: QKV 144, out_proj and fc2 48; four images: out_proj / fc2 156): the chain of
            // ONE 32x32 accumulator is the whole GEMM.  32x32 workgroup tiles on v_mfma_f32_16x16x4_f32 (vit_gemm_latency.hip)
            // quadruple the waves and shorten the chain 2.3x, bit-identically: fc2 of one image 61 -> 27 us, out_proj 20 -> 11.
            if (p.K % 128 == 0 && (long)((p.M + 63) / 64) * ((p.N + 63) / 64) < 160)
                return vitgemm::launch_gemm_f32_latency(stream, p, epilogue);
            if ((long)((p.M + 127) / 128) * ((p.N + 63) / 64) < 256)
                return launch_tile<64, 64, 32, 32, AMODE, 32, true>(stream, p, epilogue);
            // Round 5 (tools/small_batch_tiles.py: every GEMM of the ViT-B/16 forward at 12 ... 128 images on every tile code): below the
            // persistent walk's range the 64x64 tile beats the 128x64 one at every size measured -- four times the workgroups of a
            // 128x128 walk for the partial last round to spread over (fc2 at 48 images 4.50 -> 3.96 ms per 12 launches, out_proj 1.30 ->
            // 1.09; at 96 images 7.93 -> 7.81 / 2.23 -> 2.11) -- and the persistent walk pays from ~1,280 tiles on, not from 2,048 (QKV at
            // 48 images, 1,332 tiles: 3.30 -> 2.97 ms; fc1 at 40 images, 1,488 tiles: 3.70 -> 3.38).  A 40-image forward 12.6 -> ~11.9 ms,
            // 12 images 5.2 -> 4.8: the first piece of a host call (vit_engine_forward_host) and every small batch.  Same bits on every
            // tile, so the choice is time only.
            if (epilogue == VITHIP_EPI_BIAS_RESIDUAL) {
                if (tiles < 1024) return launch_tile<64, 64, 32, 32, AMODE, 32, true>(stream, p, epilogue);
                // with a workspace the persistent walk hands the first K-steps of the partial last round's tiles to idle
                // workgroups (fc2 / out_proj at batch 256: 480 -> 448 steps per workgroup); without one the residual
                // epilogue is marginally faster one tile per workgroup (fc2 21.30 vs 21.43 ms per step)
                // (also when the caller wants the row statistics of the stored rows: the persistent epilogue takes them on the way)
                if ((p.row_partials && p.N % 128 == 0) || (p.sk_ws && vitgemm::persistent_piece_steps(p.M, p.N, p.K, p.sk_slots, 0) > 0))
                    return vitgemm::launch_persistent(stream, p, epilogue, p.group_m);
                return launch_tile<128, 128, 64, 64, AMODE, 32, true>(stream, p, epilogue);
            }
            if (tiles < 1280) return launch_tile<64, 64, 32, 32, AMODE, 32, true>(stream, p, epilogue);
            return vitgemm::launch_persistent(stream, p, epilogue, p.group_m);
        }
    }
}

bool aligned16(const void *ptr) { return (reinterpret_cast<size_t>(ptr) & 15) == 0; }

}  // namespace

extern "C" {

#ifdef VIT_PROBES
// Probe build only: process-wide overrides of the per-call tile / group fields (0 = none).
int vithip_gemm_set_tile(int tile) {
    if ((tile < 0 || tile > 12) && (tile < 101 || tile > 136)) return static_cast<int>(hipErrorInvalidValue);
    g_gemm_tile = tile;
    return 0;
}

// Probe 105 (full kernel + clock stamps) writes 8 x u64 per workgroup into this device buffer.
int vithip_gemm_set_debug_buffer(void *buf) {
    g_gemm_dbg = static_cast<unsigned long long *>(buf);
    return 0;
}

int vithip_gemm_set_group(int group_m) {
    if (group_m < 0 || group_m > 1024) return static_cast<int>(hipErrorInvalidValue);
    g_gemm_group = group_m;
    return 0;
}
#endif

// ---- the scratch of vithip_gemm_args.workspace ----------------------------------------------------------------------------
// A host-side handle around UNCACHED device memory (accesses bypass the per-XCD L2s, so the workgroup that parks accumulators
// and the one that picks them up need no cache maintenance): [4 KB: one flag per owner workgroup, two counters]
// [one 64 KB slot per owner].  Owners are the first R < grid <= 2 x CUs workgroups, so 2 x CUs slots cover every launch on the
// device the workspace was made on.
struct GemmWorkspace {
    void *dev;
    int slots;
    int device;
    int gen;  // launches that used it so far: every launch tags its flag words with its number (vit_gemm_persistent.hip)
};
static size_t workspace_dev_bytes(int slots) { return 4096 + (size_t)slots * 128 * 128 * 4; }

size_t vithip_gemm_f32_workspace_bytes(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return workspace_dev_bytes(2 * cus);
}

int vithip_gemm_f32_workspace_create(void **ws) {
    if (!ws) return static_cast<int>(hipErrorInvalidValue);
    *ws = nullptr;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return static_cast<int>(e);
    GemmWorkspace *w = new (std::nothrow) GemmWorkspace{nullptr, 2 * cus, dev, 0};
    if (!w) return static_cast<int>(hipErrorOutOfMemory);
    e = hipExtMallocWithFlags(&w->dev, workspace_dev_bytes(w->slots), hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(w->dev, 0, 4096);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (w->dev) (void)hipFree(w->dev);
        delete w;
        return static_cast<int>(e);
    }
    *ws = w;
    return 0;
}
int vithip_gemm_f32_workspace_destroy(void *ws) {
    if (!ws) return 0;
    GemmWorkspace *w = static_cast<GemmWorkspace *>(ws);
    const hipError_t e = hipFree(w->dev);
    delete w;
    return static_cast<int>(e);
}
// Hand-overs since the last call: *taken = tiles an owner finished from a parked piece, *recomputed = tiles an owner ran whole
// because the piece was not there when it looked (never a wrong result, x K-steps of one workgroup lost).  Blocking (the
// stream's launches must be complete for the numbers to be final); clears the counters.
int vithip_gemm_f32_workspace_stats(void *ws, int *taken, int *recomputed) {
    if (!ws) return static_cast<int>(hipErrorInvalidValue);
    GemmWorkspace *w = static_cast<GemmWorkspace *>(ws);
    int c[2] = {0, 0};
    char *at = static_cast<char *>(w->dev) + 1016 * sizeof(int);
    hipError_t e = hipMemcpy(c, at, sizeof(c), hipMemcpyDeviceToHost);
    if (e == hipSuccess && (c[0] || c[1])) e = hipMemset(at, 0, sizeof(c));
    if (taken) *taken = c[0];
    if (recomputed) *recomputed = c[1];
    return static_cast<int>(e);
}
// The device side of the handle (tests read the flags through it).
void *vithip_gemm_f32_workspace_device_ptr(void *ws) { return ws ? static_cast<GemmWorkspace *>(ws)->dev : nullptr; }

// Would vithip_gemm_f32(a) take the row statistics in its epilogue?  (the persistent walk: tile 9, or auto with >= 1024 tiles)
int vithip_gemm_f32_stats_in_epilogue(const vithip_gemm_args *a) {
    if (!a || !a->stats_out || !a->stats_partials || a->epilogue != VITHIP_EPI_BIAS_RESIDUAL || a->M <= 0 || a->N <= 0 || a->N % 128 ||
        a->N > 2048)
        return 0;
    const long tiles = (long)((a->M + 127) / 128) * (a->N / 128);
    return (a->tile == 9 || (a->tile == 0 && tiles >= 1024)) ? 1 : 0;
}

int vithip_gemm_f32(vithip_stream_t stream, const vithip_gemm_args *a) {
    if (!a || !a->A || !a->W || !a->bias || !a->C) return static_cast<int>(hipErrorInvalidValue);
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->K % KALIGN != 0) return static_cast<int>(hipErrorInvalidValue);
    if (a->lda % 4 || a->ldw % 4 || a->lda < a->K || a->ldw < a->K || a->ldc < a->N)
        return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(a->A) || !aligned16(a->W)) return static_cast<int>(hipErrorInvalidValue);
    if ((size_t)a->M * a->lda * 4 >= 0x7fffffffull || (size_t)a->N * a->ldw * 4 >= 0x7fffffffull)
        return static_cast<int>(hipErrorInvalidValue);  // 32-bit buffer offsets: split the batch (vit_engine's lane cap does)
    if (a->epilogue == VITHIP_EPI_BIAS_RESIDUAL && (!a->residual || a->ldr < a->N))
        return static_cast<int>(hipErrorInvalidValue);
    if ((size_t)a->M * a->ldc * 4 >= 0x7fffffffull ||
        (a->epilogue == VITHIP_EPI_BIAS_RESIDUAL && (size_t)a->M * a->ldr * 4 >= 0x7fffffffull))
        return static_cast<int>(hipErrorInvalidValue);  // the epilogue addresses C and the residual through 32-bit buffer offsets too
    GemmParams p{};
    p.A = a->A; p.W = a->W; p.bias = a->bias; p.R = a->residual; p.C = a->C;
    p.lda = a->lda; p.ldw = a->ldw; p.ldr = a->ldr; p.ldc = a->ldc;
    p.M = a->M; p.N = a->N; p.K = a->K;
    if (a->workspace) {
        GemmWorkspace *w = static_cast<GemmWorkspace *>(a->workspace);
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != w->device) return static_cast<int>(hipErrorInvalidDevice);  // made on another device
        p.sk_ws = w->dev;
        p.sk_slots = w->slots;
        p.sk_late = a->handover_test == 1;
        // The generation is a HOST counter baked into the launch's parameters: it numbers the launches ENQUEUED on this workspace.  A
        // hipGraph that captured this launch replays the same number every time (vit_engine_options.use_graph), so under replay the
        // tag does not tell one replay's words from the previous replay's -- what protects a replay is what protects every launch:
        // each path of the hand-over leaves its flag word 0 (taken: the owner clears it; withdrawn: the helper clears it), so a
        // COMPLETED launch leaves no word behind.  The generation only covers what that cannot: words left by a launch that was
        // aborted mid-way (a fault, a killed process sharing nothing -- i.e. in practice never under replay either, since an aborted
        // replay takes the graph's stream down with it).
        w->gen = w->gen >= (1 << 28) ? 1 : w->gen + 1;  // (launches on one workspace are ordered by contract: one stream, one caller)
        p.sk_gen = w->gen;
    }
    if (a->handover_test < 0 || a->handover_test > 1) return static_cast<int>(hipErrorInvalidValue);
    if (a->tile < 0 || (a->tile > 0 && a->tile < 6) || a->tile > 12 || a->group_m < 0 || a->group_m > 1024) return static_cast<int>(hipErrorInvalidValue);
    int epilogue = a->epilogue;
    if (a->ln_rows || a->ln_colsum) {  // LayerNorm fold, consumer side (ln_colsum NULL: the weight is the CENTRED one, see the header)
        if (!a->ln_rows || (epilogue != VITHIP_EPI_BIAS && epilogue != VITHIP_EPI_BIAS_GELU) ||
            (reinterpret_cast<size_t>(a->ln_rows) & 7))
            return static_cast<int>(hipErrorInvalidValue);
        p.ln_rows = a->ln_rows;
        p.ln_colsum = a->ln_colsum;
        epilogue = epilogue == VITHIP_EPI_BIAS ? vitgemm::EPI_BIAS_LN : vitgemm::EPI_BIAS_GELU_LN;
    }
    if (a->stats_out) {  // ... producer side
        if (epilogue != VITHIP_EPI_BIAS_RESIDUAL || a->N % 64 || a->N > 2048 || (reinterpret_cast<size_t>(a->stats_out) & 7) ||
            (reinterpret_cast<size_t>(a->stats_partials) & 7))
            return static_cast<int>(hipErrorInvalidValue);
        if (vithip_gemm_f32_stats_in_epilogue(a)) p.row_partials = a->stats_partials;
    }
    const int rc = dispatch<A_DENSE>(static_cast<hipStream_t>(stream), p, epilogue, a->tile, a->group_m);
    if (rc != 0 || !a->stats_out) return rc;
    if (p.stats_in_epilogue) return vithip_rowstats_finalize_f32(stream, a->stats_partials, a->M, a->N, a->stats_out);
    return vithip_rowstats_f32(stream, a->C, (size_t)a->ldc, a->stats_out, a->M, a->N);
}

int vithip_patch_embed_f32(vithip_stream_t stream, const float *images, const float *conv_w,
                           const float *conv_b, const float *cls, const float *pos, float *x,
                           int n_images, int img_size, int patch_size, int in_chans, int embed_dim) {
    if (!images || !conv_w || !conv_b || !cls || !pos || !x || n_images <= 0)
        return static_cast<int>(hipErrorInvalidValue);
    if (patch_size % 4 || img_size % patch_size) return static_cast<int>(hipErrorInvalidValue);
    const int K = in_chans * patch_size * patch_size;
    if (K % KALIGN || img_size % 4) return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(images) || !aligned16(conv_w)) return static_cast<int>(hipErrorInvalidValue);
    const int G = img_size / patch_size;
    GemmParams p{};
    p.A = images; p.W = conv_w; p.bias = conv_b; p.C = x; p.pos = pos;
    p.lda = 0; p.ldw = K; p.ldc = embed_dim;
    p.M = n_images * G * G; p.N = embed_dim; p.K = K;
    p.patches = G * G; p.grid = G; p.patch = patch_size; p.img = img_size; p.chans = in_chans;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int total = n_images * embed_dim;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, s, cls, pos, x,
                       n_images, G * G + 1, embed_dim);
    int e = static_cast<int>(hipGetLastError());
    if (e) return e;
    return dispatch<A_PATCHES>(s, p, VITHIP_EPI_BIAS, 0, 0);
}

}  // extern "C"
