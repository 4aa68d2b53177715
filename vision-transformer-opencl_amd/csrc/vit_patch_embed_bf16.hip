// csrc/vit_patch_embed_bf16.hip -- patch embedding on the bf16 matrix pipe as ONE implicit GEMM over the NCHW fp32 images
// (vithip_patch_embed_bf16_implicit).  NOT the engine's default: measured slower than the two-pass path, see below.
//
// Reference: Conv2d + flatten_transpose + class_token + pos_emb (ViT_seq.c:25-101; OpenCL: kernel.cl:120-175 plus host
// loops, ViT_opencl.c:126-180,806-810).  conv_proj with stride = kernel = patch is a GEMM over the patches:
//   x[img][1 + patch][d] = bias[d] + sum_k pixel(img, patch, k) * W[d][k] + pos[1 + patch][d],   k = (c * p + kh) * p + kw
// The engine's bf16 path runs it as two passes: patchify_bf16_kernel writes every image as bf16 patch rows and the
// ping-pong GEMM reads them back through LDS-DMA (1.0 ms at batch 2048).  SURVEY.md 8f rank 4 asks for the gather inside the
// A-tile loader instead.  LDS-DMA moves bytes -- it cannot turn fp32 pixels into bf16 -- so that needs a register-staged A
// loader, which is this kernel:
//   * 128 x 128 output tile, 4 waves (64 x 64 each, 2 x 2 accumulators of v_mfma_f32_32x32x16_bf16), K step 32, three
//     workgroups per CU, three K steps of loads in flight in registers;
//   * A: a thread fetches 4 float4 of pixels per K step (its k offset inside the step is fixed; (c, kh, kw) advance
//     incrementally), converts them to bf16 and writes 8-byte halves of the 16-byte chunks of a [128][32] bf16 LDS tile;
//     W: 2 x 16 bytes of the bf16 conv weight.  64-byte rows, chunk position = chunk ^ ((row >> 2) & 3): conflict-free
//     ds_read_b128 fragment reads;
//   * epilogue: + bias + pos_emb, row m -> token row m + image + 1 (the class row is written by cls_rows_embed_kernel);
//   * tile id -> (tm, tn) with tn fastest and the XCD remap: the N tiles that re-read an M tile's pixels share an L2.
// Measured at batch 2048 (tools/embed_time.py): 2.4 ms against 1.0 ms for the two passes.  With the stages switched off one
// at a time: MFMA + LDS + barriers 0.49 ms, + W loads 0.53, + pixel loads 1.43, + epilogue 1.05, and three K steps of
// prefetch instead of one change nothing -- it is bound by bytes through the L2s, not by latency: every 128-wide N tile re-reads
// its pixels as fp32 (1.23 GB x 6 N tiles + W 3.7 GB = 11 GB) where the two passes move bf16 through 256 x 256 tiles (5.5 GB
// over both kernels).  Matching them takes 256-wide tiles with a register-staged A operand, i.e. the ping-pong kernel's
// structure minus LDS-DMA on one side; the staging pass is the better design on this chip.  Kept as a tested alternative.
#include "vit_gemm_common.hpp"

namespace vitgemm {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef unsigned short bf16_t;

constexpr int EBM = 128, EBN = 128, EBK = 32;
constexpr int ETHREADS = 256;
constexpr int ETILE = EBM * EBK;  // bf16 elements of one operand tile (8 KB)

struct EmbedParams {
    const float *images;
    const bf16_t *W;      // [N][K] bf16
    const float *bias;    // [N]
    const float *pos;     // [1 + P][N]
    float *x;             // [n][1 + P][N]
    int M, N, K;          // M = n_images * P
    int P, G, p, S, C;    // patches per image, grid, patch size, image size, channels
    int tiles_m, tiles_n;
    unsigned magic_P;     // ceil(2^32 / P): m / P == __umulhi(m, magic_P) for m < 2^32 / P
};

__global__ __launch_bounds__(ETHREADS, 3) void patch_embed_bf16_kernel(const EmbedParams q) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[4 * ETILE];  // A0 A1 B0 B1: 32 KB
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    const int m0 = tm * EBM, n0 = tn * EBN;

    // ---- A loader: float4 c4 (4 consecutive k) of patch rows ar + 32 i ------------------------------------------
    const int c4 = tid & 7, ar = tid >> 3;
    unsigned a_base[4];  // pixel offset of the patch origin (channel 0), per row (< 2^32 elements: checked by the launcher)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + ar + 32 * i;
        m = m < q.M ? m : q.M - 1;
        const int im = (int)__umulhi((unsigned)m, q.magic_P), pp = m - im * q.P;
        const int py = pp / q.G, px = pp - py * q.G;
        a_base[i] = ((unsigned)im * q.C * q.S + (unsigned)py * q.p) * q.S + (unsigned)px * q.p;
    }
    // this thread's k = k0 + 4 c4 -> (c, kh, kw), advanced by EBK per step with carries
    const int pp2 = q.p * q.p;
    int kc, kh, kw;
    {
        const int k = 4 * c4;
        kc = k / pp2;
        const int rem = k - kc * pp2;
        kh = rem / q.p;
        kw = rem - kh * q.p;
    }
    // ---- B loader: 16-byte chunk bc of weight rows br + 64 i
    const int bc = tid & 3, br = tid >> 2;
    const bf16_t *b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int n = n0 + br + 64 * i;
        n = n < q.N ? n : q.N - 1;
        b_src[i] = q.W + (size_t)n * q.K + 8 * bc;
    }

    // Staging registers: DEPTH K steps in flight.  A step's MFMAs take 256 cycles per wave, a pixel load under this access
    // pattern comes back after 3-4k: with one step of prefetch every step waited for its own loads (2.8 ms at batch 2048).
    constexpr int DEPTH = 3;
    f32x4 a_st[DEPTH][4];
    uint4 b_st[DEPTH][2];
    auto load_step = [&](int slot, int k0) __attribute__((always_inline)) {
        const unsigned koff = ((unsigned)kc * q.S + kh) * q.S + kw;
#pragma unroll
        for (int i = 0; i < 4; ++i) a_st[slot][i] = *reinterpret_cast<const f32x4 *>(q.images + (size_t)(a_base[i] + koff));
#pragma unroll
        for (int i = 0; i < 2; ++i) b_st[slot][i] = *reinterpret_cast<const uint4 *>(b_src[i] + k0);
        kw += EBK;
        while (kw >= q.p) { kw -= q.p; ++kh; }
        while (kh >= q.p) { kh -= q.p; ++kc; }
    };
    auto store_step = [&](int slot, int buf) __attribute__((always_inline)) {
        bf16_t *As = lds + buf * ETILE, *Bs = lds + (2 + buf) * ETILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ar + 32 * i;
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)a_st[slot][i][e];
            // 16-byte chunk c4 >> 1 of the 64-byte row, swizzled; this thread's half (c4 & 1)
            *reinterpret_cast<bf16x4 *>(As + row * EBK + (((c4 >> 1) ^ ((row >> 2) & 3)) * 8) + 4 * (c4 & 1)) = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = br + 64 * i;
            *reinterpret_cast<uint4 *>(Bs + row * EBK + ((bc ^ ((row >> 2) & 3)) * 8)) = b_st[slot][i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    const int nk = q.K / EBK;  // multiple of DEPTH is not required: the slot index is (step % DEPTH), unrolled below
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nk) load_step(d, d * EBK);
    store_step(0, 0);
    __syncthreads();
    const int sw = (r >> 2) & 3;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const bf16_t *As = lds + buf * ETILE + (wm * 64 + r) * EBK;
        const bf16_t *Bs = lds + (2 + buf) * ETILE + (wn * 64 + r) * EBK;
#pragma unroll
        for (int ks = 0; ks < EBK / 16; ++ks) {
            const int off = (((2 * ks + h) ^ sw) & 3) * 8;  // rows r and r + 32 share (row >> 2) & 3
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(As + i * 32 * EBK + off);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const bf16x8 *>(Bs + j * 32 * EBK + off);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    // step t: refill the slot step t used (its registers went to LDS before the last barrier) with step t + DEPTH, multiply
    // LDS buffer t & 1, then move step t + 1 from its registers to the other LDS buffer.  Unrolled by DEPTH * 2 so that slot
    // and buffer indices are constants.
    for (int t0 = 0; t0 < nk; t0 += 2 * DEPTH) {
#pragma unroll
        for (int u = 0; u < 2 * DEPTH; ++u) {
            const int t = t0 + u;
            if (t < nk) {
                if (t + DEPTH < nk) load_step(u % DEPTH, (t + DEPTH) * EBK);
                __builtin_amdgcn_sched_barrier(0);
                compute(u & 1);
                __builtin_amdgcn_sched_barrier(0);  // keeps the conversions below (and their vmcnt wait) behind the MFMAs
                if (t + 1 < nk) {
                    store_step((u + 1) % DEPTH, (u + 1) & 1);
                    __syncthreads();
                }
            }
        }
    }

    // ---- epilogue: lane = column n of 16 rows per accumulator (rows (v & 3) + 8 (v >> 2) + 4 h) ------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        const bool n_ok = n < q.N;
        const float bn = n_ok ? q.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = m0 + wm * 64 + i * 32 + 4 * h;
            // vmcnt retires in order and counts stores: a pos_emb load that is waited for between two stores serialises
            // them on the full memory latency (64 round trips per lane made the first version's epilogue a third of its
            // tile).  The operands are fetched 8 at a time, then their 8 stores go out back to back.
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float add[8];
                unsigned orow[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int v = 8 * half + e;
                    int m = mb + (v & 3) + 8 * (v >> 2);
                    m = m < q.M ? m : q.M - 1;
                    const unsigned im = __umulhi((unsigned)m, q.magic_P), pp = (unsigned)m - im * q.P;
                    add[e] = q.pos[(size_t)(pp + 1) * q.N + (n_ok ? n : 0)];
                    orow[e] = (unsigned)m + im + 1;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int v = 8 * half + e;
                    const int m = mb + (v & 3) + 8 * (v >> 2);
                    if (n_ok && m < q.M) q.x[(size_t)orow[e] * q.N + n] = acc[i][j][v] + bn + add[e];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

__global__ void cls_rows_embed_kernel(const float *cls, const float *pos, float *x, int n_images, int tokens, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_images * dim) return;
    const int im = i / dim, d = i - im * dim;
    x[(size_t)im * tokens * dim + d] = cls[d] + pos[d];  // class_token + pos_emb row 0 (ViT_seq.c:72-101)
}

}  // namespace

// images [n][C][S][S] fp32, conv_w16 [D][C*p*p] bf16 -> x [n][1 + G*G][D] fp32.  Returns a hipError_t value.
int launch_patch_embed_bf16(hipStream_t s, const float *images, const unsigned short *conv_w16, const float *conv_b,
                            const float *cls, const float *pos, float *x, int n_images, int img_size, int patch_size,
                            int in_chans, int embed_dim) {
    const int G = img_size / patch_size, P = G * G, K = in_chans * patch_size * patch_size;
    EmbedParams q{};
    q.images = images; q.W = conv_w16; q.bias = conv_b; q.pos = pos; q.x = x;
    q.M = n_images * P; q.N = embed_dim; q.K = K;
    q.P = P; q.G = G; q.p = patch_size; q.S = img_size; q.C = in_chans;
    q.tiles_m = (q.M + EBM - 1) / EBM;
    q.tiles_n = (q.N + EBN - 1) / EBN;
    q.magic_P = (unsigned)((0x100000000ull + (unsigned)P - 1) / (unsigned)P);
    const int total = n_images * embed_dim;
    hipLaunchKernelGGL(cls_rows_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, s, cls, pos, x, n_images, P + 1, embed_dim);
    hipLaunchKernelGGL(patch_embed_bf16_kernel, dim3(q.tiles_m * q.tiles_n), dim3(ETHREADS), 0, s, q);
    return static_cast<int>(hipGetLastError());
}

}  // namespace vitgemm
