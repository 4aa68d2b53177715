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
#include <hip/hip_runtime.h>

#include "vit_hip_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;          // K step per LDS tile
constexpr int LDS_LD = BK + 4;  // padded LDS row, floats (144 B: 16-B aligned, conflict-free)

enum { A_DENSE = 0, A_PATCHES = 1 };

struct GemmParams {
    const float *A;
    const float *W;
    const float *bias;
    const float *R;
    float *C;
    int lda, ldw, ldr, ldc;
    int M, N, K;
    int tiles_m, tiles_n;
    // A_PATCHES only
    const float *pos;
    int patches;     // patches per image (G*G)
    int grid;        // G = img/patch
    int patch;       // P
    int img;         // S
    int chans;       // C
};

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x / 1.41421356237309504880f));
}

// Workgroup id -> tile id such that ids sharing an XCD (id % 8) get consecutive tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return base + idx;
}

template <int BM, int BN, int WM, int WN, int EPI, int AMODE>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const GemmParams p) {
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

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread global source pointers for the staging loads -----------------------
    const int ld_row = tid >> 3;          // 0..31 (+32 per chunk)
    const int ld_kc = (tid & 7) * 4;      // float offset inside the K step
    const float *a_src[A_CHUNKS];
    const float *b_src[B_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        int m = m0 + ld_row + i * 32;
        m = m < p.M ? m : p.M - 1;
        if constexpr (AMODE == A_DENSE) {
            a_src[i] = p.A + (size_t)m * p.lda + ld_kc;
        } else {
            // row m = (image, patch); k = (ic, kh, kw): resolved per K step in load_a()
            const int im = m / p.patches, pp = m - im * p.patches;
            const int oh = pp / p.grid, ow = pp - oh * p.grid;
            a_src[i] = p.A + ((size_t)im * p.chans * p.img + (size_t)oh * p.patch) * p.img + ow * p.patch;
        }
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
        int n = n0 + ld_row + i * 32;
        n = n < p.N ? n : p.N - 1;
        b_src[i] = p.W + (size_t)n * p.ldw + ld_kc;
    }

    f32x4 a_stage[A_CHUNKS], b_stage[B_CHUNKS];

    auto load_global = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            if constexpr (AMODE == A_DENSE) {
                a_stage[i] = *reinterpret_cast<const f32x4 *>(a_src[i] + k0);
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
        for (int i = 0; i < B_CHUNKS; ++i) b_stage[i] = *reinterpret_cast<const f32x4 *>(b_src[i] + k0);
    };
    auto store_lds = [&](int buf) {
        float *As = As0 + buf * BM * LDS_LD;
        float *Bs = Bs0 + buf * BN * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i)
            *reinterpret_cast<f32x4 *>(As + (ld_row + i * 32) * LDS_LD + ld_kc) = a_stage[i];
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i)
            *reinterpret_cast<f32x4 *>(Bs + (ld_row + i * 32) * LDS_LD + ld_kc) = b_stage[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    const int nk = p.K / BK;
    load_global(0);
    store_lds(0);
    __syncthreads();

    const int a_frag_off = (wm * WM + r) * LDS_LD + h * 4;
    const int b_frag_off = (wn * WN + r) * LDS_LD + h * 4;

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_global((kt + 1) * BK);

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

        if (more) store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds column n (128-B coalesced segments per half-wave) ----------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + r;
        const bool n_ok = n < p.N;
        const float bias = n_ok ? p.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * WM + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (n_ok && m < p.M) {
                    float y = acc[i][j][v] + bias;
                    if constexpr (AMODE == A_PATCHES) {
                        const int im = m / p.patches, pp = m - im * p.patches;
                        y += p.pos[(size_t)(pp + 1) * p.N + n];
                        p.C[((size_t)m + im + 1) * p.ldc + n] = y;
                    } else {
                        if constexpr (EPI == VITHIP_EPI_BIAS_GELU) y = gelu_erf(y);
                        if constexpr (EPI == VITHIP_EPI_BIAS_RESIDUAL) y += p.R[(size_t)m * p.ldr + n];
                        p.C[(size_t)m * p.ldc + n] = y;
                    }
                }
            }
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

template <int BM, int BN, int WM, int WN, int AMODE>
int launch_tile(hipStream_t stream, GemmParams &p, int epilogue) {
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    const dim3 grid(p.tiles_m * p.tiles_n), block(256);
    if constexpr (AMODE == A_PATCHES) {
        hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS, A_PATCHES>), grid, block, 0, stream, p);
    } else {
        switch (epilogue) {
            case VITHIP_EPI_BIAS:
                hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS, A_DENSE>), grid, block, 0, stream, p);
                break;
            case VITHIP_EPI_BIAS_GELU:
                hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS_GELU, A_DENSE>), grid, block, 0, stream, p);
                break;
            case VITHIP_EPI_BIAS_RESIDUAL:
                hipLaunchKernelGGL((gemm_f32_nt_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS_RESIDUAL, A_DENSE>), grid, block, 0, stream, p);
                break;
            default:
                return static_cast<int>(hipErrorInvalidValue);
        }
    }
    return static_cast<int>(hipGetLastError());
}

int g_gemm_tile = 0;  // 0 = auto; see vithip_gemm_set_tile()

template <int AMODE>
int dispatch(hipStream_t stream, GemmParams &p, int epilogue) {
    switch (g_gemm_tile) {
        case 2: return launch_tile<256, 128, 128, 64, AMODE>(stream, p, epilogue);
        case 3: return launch_tile<128, 64, 64, 32, AMODE>(stream, p, epilogue);
        default: return launch_tile<128, 128, 64, 64, AMODE>(stream, p, epilogue);
    }
}

bool aligned16(const void *ptr) { return (reinterpret_cast<size_t>(ptr) & 15) == 0; }

}  // namespace

extern "C" {

// Tuning hook (bench/tests): 0/1 = 128x128, 2 = 256x128, 3 = 128x64 workgroup tiles.
int vithip_gemm_set_tile(int tile) {
    if (tile < 0 || tile > 3) return static_cast<int>(hipErrorInvalidValue);
    g_gemm_tile = tile;
    return 0;
}

int vithip_gemm_f32(vithip_stream_t stream, const vithip_gemm_args *a) {
    if (!a || !a->A || !a->W || !a->bias || !a->C) return static_cast<int>(hipErrorInvalidValue);
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->K % BK != 0) return static_cast<int>(hipErrorInvalidValue);
    if (a->lda % 4 || a->ldw % 4 || a->lda < a->K || a->ldw < a->K || a->ldc < a->N)
        return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(a->A) || !aligned16(a->W)) return static_cast<int>(hipErrorInvalidValue);
    if (a->epilogue == VITHIP_EPI_BIAS_RESIDUAL && (!a->residual || a->ldr < a->N))
        return static_cast<int>(hipErrorInvalidValue);
    GemmParams p{};
    p.A = a->A; p.W = a->W; p.bias = a->bias; p.R = a->residual; p.C = a->C;
    p.lda = a->lda; p.ldw = a->ldw; p.ldr = a->ldr; p.ldc = a->ldc;
    p.M = a->M; p.N = a->N; p.K = a->K;
    return dispatch<A_DENSE>(static_cast<hipStream_t>(stream), p, a->epilogue);
}

int vithip_patch_embed_f32(vithip_stream_t stream, const float *images, const float *conv_w,
                           const float *conv_b, const float *cls, const float *pos, float *x,
                           int n_images, int img_size, int patch_size, int in_chans, int embed_dim) {
    if (!images || !conv_w || !conv_b || !cls || !pos || !x || n_images <= 0)
        return static_cast<int>(hipErrorInvalidValue);
    if (patch_size % 4 || img_size % patch_size) return static_cast<int>(hipErrorInvalidValue);
    const int K = in_chans * patch_size * patch_size;
    if (K % BK || img_size % 4) return static_cast<int>(hipErrorInvalidValue);
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
    return dispatch<A_PATCHES>(s, p, VITHIP_EPI_BIAS);
}

}  // extern "C"
