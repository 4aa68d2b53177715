// csrc/vit_gemm_bf16.hip -- bf16 MFMA "NT" GEMM (fp32 accumulate) for the bf16 variant of the forward
// (BASELINE.json configs[2]: ViT-B/16, batch 2048, bf16 MFMA): the C-ABI entry points, the
// patchify kernel of the bf16 patch embedding, and the FIRST GEMM kernel (two LDS stages, one barrier per K step),
// which now serves K < 128 and as variant 1 for A/B runs; the kernel the engine uses is csrc/vit_gemm_bf16_pp.hip.
//
//   C = epilogue(A[M][K] . W[N][K]^T + bias),  A and W bf16 (K contiguous), bias fp32, accumulate fp32.
//   epilogues:  BF16       C bf16 = acc + bias                      (QKV in_proj)
//               BF16_GELU  C bf16 = gelu_erf(acc + bias)            (fc1; ViT_seq.c:231-233)
//               F32_RES    C fp32 = acc + bias + R fp32, R may be C (out_proj, fc2: the residual stream
//                                                                    stays fp32, ViT_seq.c:286-288,297-299)
//
// Design (gfx950, after the measurements of the CDNA4 guide: 256x256 tile, direct-to-LDS loads, XOR swizzle):
//  * v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16 block, 16x the fp32 MFMA rate, so operand delivery
//    decides everything.  Workgroup = 8 waves (2 per SIMD), tile 256x256, K step 64 (128-B rows in LDS);
//    a wave owns 128(m) x 64(n): 4x2 accumulators, 6 ds_read_b128 per 8 MFMAs.  HBM/L2 -> LDS traffic is
//    64 KB per 2048 matrix-pipe cycles = 32 B/clk/CU.
//  * global_load_lds_dwordx4 (LDS-DMA, 16 B per lane): no staging registers and no ds_write; each wave
//    instruction fills 8 rows x 128 B.  The LDS image is lane-linear, so the bank-conflict swizzle
//    (16-B chunk c of row r lives at chunk c ^ ((r >> 1) & 7)) is applied to the per-lane SOURCE address
//    and again on the ds_read side.  Two LDS stages; the loads of step t+1 are issued before the MFMAs
//    of step t and drained (vmcnt(0)) at the single barrier that ends the step.
//  * Operand roles are swapped (MFMA A = W rows, B = activation rows): a lane then owns ONE output row m
//    and 4 consecutive columns n per register group, so the epilogue packs 4 bf16 (8 B) or one float4 per
//    store instead of scattering 2-byte elements.
//  * Tile walk: XCD-aware remap + groups of 8 tile rows (as the fp32 kernel).
#include "vit_device.hpp"
#include "vit_gemm_common.hpp"
#ifdef VIT_PROBES
#include "vit_probes.h"
#endif

namespace {

using vitgemm::erf_fp32;
using vitgemm::tile_coords;
using vitgemm::xcd_remap;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

constexpr int TBM = 256, TBN = 256, TBK = 64;      // bf16 elements
constexpr int ROWB = TBK * 2;                      // 128 bytes per LDS row
constexpr int STAGE_BYTES = (TBM + TBN) * ROWB;    // 64 KB
constexpr int THREADS = 512;

using vitgemm::Bf16Params;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// DBG: timing-only probes (tools/gemm_bf16_probe.py): 1 = no LDS-DMA inside the K loop (MFMA + ds_read only),
// 2 = no MFMA/ds_read inside the K loop (LDS-DMA stream only).  Results are wrong by construction.
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(THREADS, 2) void gemm_bf16_nt_kernel(const Bf16Params p) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves: 128 rows x 64 columns each

    // Persistent walk: one workgroup per CU takes tiles first, first + nwg, ...; the first K step of the
    // NEXT tile is fetched (LDS-DMA) before the epilogue of the current one, so the load latency that would
    // open every tile hides under the stores.
    const int total = p.tiles_m * p.tiles_n, nwg = gridDim.x;
    int tile = xcd_remap(blockIdx.x, nwg);
    if (tile >= total) return;  // workgroup-uniform
    int m0 = 0, n0 = 0;

    // ---- staging: wave w fills rows [32w, 32w+32) of the A tile and of the W tile, 4 DMA instructions
    // each (8 rows x 128 B per instruction).  Lane l lands at (row q*8 + l/8, chunk l%8) and therefore
    // fetches source chunk (l%8) ^ ((row>>1)&7) of that row.
    const char *a_src[4];
    const char *w_src[4];
    auto set_tile = [&](int t) {
        int tm, tn;
        tile_coords(t, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
        m0 = tm * TBM;
        n0 = tn * TBN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 32 + q * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + row;
            m = m < p.M ? m : p.M - 1;
            int n = n0 + row;
            n = n < p.N ? n : p.N - 1;
            a_src[q] = reinterpret_cast<const char *>(p.A + (size_t)m * p.lda) + chunk * 16;
            w_src[q] = reinterpret_cast<const char *>(p.W + (size_t)n * p.ldw) + chunk * 16;
        }
    };
    set_tile(tile);
    auto stage = [&](int buf, int k0) {
        char *a_dst = lds + buf * STAGE_BYTES + wave * 32 * ROWB;
        char *w_dst = a_dst + TBM * ROWB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_global_load_lds((gbl_void *)(a_src[q] + (size_t)k0 * 2), (lds_void *)(a_dst + q * 8 * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(w_src[q] + (size_t)k0 * 2), (lds_void *)(w_dst + q * 8 * ROWB), 16, 0, 0);
        }
    };

    // ---- fragment addresses: row (wave tile row + r), logical 16-B chunk 2*ks + h, swizzled ------
    int a_off[4], w_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wm * 128 + i * 32 + r;
        a_off[i] = row * ROWB;  // + ((2*ks + h) ^ sw) * 16 per k16 step, sw below
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + j * 32 + r;
        w_off[j] = TBM * ROWB + row * ROWB;
    }
    const int sw = (r >> 1) & 7;  // every fragment row of this lane is r (mod 32): same swizzle key

    const int nk = p.K / TBK;
    int par = 0;  // LDS stage holding K step 0 of the current tile
    stage(0, 0);
    for (;;) {
        f32x16 acc[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.0f;
        // bias of this tile's columns, fetched now (drained by the wait below, used a whole tile later)
        f32x4 b4[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int nb = n0 + wn * 64 + j * 32 + 8 * g + 4 * h;
                nb = nb + 4 <= p.N ? nb : p.N - 4;  // clamped address; such groups are never stored
                b4[j][g] = *reinterpret_cast<const f32x4 *>(p.bias + nb);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const int cur = par ^ (kt & 1);
            if (DBG != 1 && kt + 1 < nk) stage(cur ^ 1, (kt + 1) * TBK);
            const char *base = lds + cur * STAGE_BYTES;
#pragma unroll
            for (int ks = 0; ks < (DBG == 2 ? 0 : TBK / 16); ++ks) {
                const int coff = (((2 * ks + h) ^ sw) & 7) * 16;
                bf16x8 wf[2], af[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const bf16x8 *>(base + w_off[j] + coff);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(base + a_off[i] + coff);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA of the next stage has landed
            __syncthreads();                                    // ... and everybody is done reading this one
        }
        par ^= (nk & 1) ^ 1;  // the stage NOT used by the last K step: free for the next tile's first step

        // ---- epilogue through LDS --------------------------------------------------------------------
        // In registers a lane owns one row and 4 consecutive columns per group, so direct stores would be
        // 16-B (bf16) / 32-B (fp32) fragments scattered over 32 rows per instruction -- measured: the store
        // phase cost as much as the whole K loop.  The tile is therefore transposed through the (now idle)
        // LDS in row blocks and leaves as full 512-B / 1-KB row segments, 16 B per lane; the residual is read
        // the same coalesced way.  Row pitches 520 B / 1040 B keep the block writes conflict-free.
        auto finish = [&](float t) {
            if constexpr (EPI == VITHIP_BF16_EPI_BF16_GELU) t = 0.5f * t * (1.0f + erf_fp32(t * 0.70710678118654752440f));
            return t;
        };
        if constexpr (EPI == VITHIP_BF16_EPI_F32_RESIDUAL) {
            constexpr int PITCH = TBN * 4 + 16;  // 1040 B
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {  // 64 rows per pass: waves wm == pass/2, accumulators i = 2*(pass&1), +1
                if (wm == (pass >> 1)) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int i = 2 * (pass & 1) + ii;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                f32x4 y;
#pragma unroll
                                for (int q = 0; q < 4; ++q) y[q] = acc[j][i][4 * g + q] + b4[j][g][q];
                                *reinterpret_cast<f32x4 *>(lds + (ii * 32 + r) * PITCH + (wn * 64 + j * 32 + 8 * g + 4 * h) * 4) = y;
                            }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 8; ++it) {  // 64 rows x 64 chunks of 16 B over 512 threads
                    const int c = tid + it * THREADS;
                    const int row = c >> 6, col4 = (c & 63) * 4;
                    const int m = m0 + pass * 64 + row, n = n0 + col4;
                    if (m < p.M && n + 4 <= p.N) {
                        const f32x4 y = *reinterpret_cast<const f32x4 *>(lds + row * PITCH + col4 * 4);
                        const f32x4 res = *reinterpret_cast<const f32x4 *>(p.R + (size_t)m * p.ldr + n);
                        *reinterpret_cast<f32x4 *>(static_cast<float *>(p.C) + (size_t)m * p.ldc + n) = y + res;
                    }
                }
                __syncthreads();
            }
        } else {
            constexpr int PITCH = TBN * 2 + 8;  // 520 B
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {  // 128 rows per pass: the waves with wm == pass
                if (wm == pass) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                bf16x4 y;
#pragma unroll
                                for (int q = 0; q < 4; ++q) y[q] = (__bf16)finish(acc[j][i][4 * g + q] + b4[j][g][q]);
                                *reinterpret_cast<bf16x4 *>(lds + (i * 32 + r) * PITCH + (wn * 64 + j * 32 + 8 * g + 4 * h) * 2) = y;
                            }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 8; ++it) {  // 128 rows x 32 chunks of 16 B over 512 threads
                    const int c = tid + it * THREADS;
                    const int row = c >> 5, col8 = (c & 31) * 8;
                    const int m = m0 + pass * 128 + row, n = n0 + col8;
                    if (m < p.M && n + 8 <= p.N) {
                        const uint4 y = *reinterpret_cast<const uint4 *>(lds + row * PITCH + col8 * 2);
                        *reinterpret_cast<uint4 *>(static_cast<bf16_t *>(p.C) + (size_t)m * p.ldc + n) = y;
                    } else if (m < p.M && n < p.N) {  // ragged N (N % 8 != 0): element-wise tail
                        for (int q = 0; q < 8 && n + q < p.N; ++q)
                            static_cast<bf16_t *>(p.C)[(size_t)m * p.ldc + n + q] =
                                *reinterpret_cast<const bf16_t *>(lds + row * PITCH + (col8 + q) * 2);
                    }
                }
                __syncthreads();
            }
        }

        const int next = tile + nwg;
        const bool has_next = next < total;  // workgroup-uniform
        if (has_next) {
            set_tile(next);
            stage(par, 0);
        }
        if (!has_next) break;
        tile = next;
    }
}

// fp32 -> bf16 (round to nearest even; NaN stays NaN through v_cvt_pk_bf16_f32), 4 elements per thread.
__global__ void f32_to_bf16_kernel(const float *__restrict__ src, bf16_t *__restrict__ dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(src)[i];
        bf16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (__bf16)v[q];
        reinterpret_cast<bf16x4 *>(dst)[i] = o;
    }
}

// images fp32 NCHW -> patch rows bf16 [n * G * G][C * P * P], column order (c, kh, kw) = the conv weight's.
// One thread converts 8 consecutive pixels of an image row (32 B in, 16 B out).
__global__ void patchify_bf16_kernel(const float *__restrict__ images, bf16_t *__restrict__ patches, int n_images, int S, int P,
                                     int C) {
    const int G = S / P, x8n = S / 8;
    const size_t total = (size_t)n_images * C * S * x8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x8 = (int)(i % x8n);
        size_t r = i / x8n;
        const int y = (int)(r % S);
        r /= S;
        const int c = (int)(r % C);
        const int im = (int)(r / C);
        const float *src = images + (((size_t)im * C + c) * S + y) * S + x8 * 8;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(src), b = *reinterpret_cast<const f32x4 *>(src + 4);
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[q] = (__bf16)a[q];
            o[4 + q] = (__bf16)b[q];
        }
        const int px = (x8 * 8) / P, kw = (x8 * 8) % P, py = y / P, kh = y % P;
        bf16_t *dst = patches + ((size_t)im * G * G + (size_t)py * G + px) * ((size_t)C * P * P) + ((size_t)c * P + kh) * P + kw;
        *reinterpret_cast<bf16x8 *>(dst) = o;
    }
}

__global__ void cls_rows_bf16path_kernel(const float *cls, const float *pos, float *x, int n_images, int tokens, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_images * dim) return;
    const int im = i / dim, d = i - im * dim;
    x[(size_t)im * tokens * dim + d] = cls[d] + pos[d];  // class_token + pos_emb row 0 (ViT_seq.c:72-101)
}

bool aligned16(const void *ptr) { return (reinterpret_cast<size_t>(ptr) & 15) == 0; }
// The kernel variant is a per-call field of vithip_gemm_bf16_args; the product library has no mutable process-wide state.  The probe build (-DVIT_PROBES) adds process-wide overrides and instrumented kernels.
#ifdef VIT_PROBES
int g_variant = 0;  // 0 none, 1 two-stage kernel, 2 ping-pong, 3 stamped ping-pong, 4 event-log ping-pong, 5 four-wave experiment (vit_gemm_bf16_w4.hip)
unsigned long long *g_dbg = nullptr;
int g_max_wgs = 0;    // event-log build: cap on persistent workgroups (0 = one per CU)
int g_group16 = 0;    // tile rows per L2 group of the walk (0 = the launcher's choice)
#endif

}  // namespace

extern "C" {

int vithip_f32_to_bf16(vithip_stream_t stream, const float *src, unsigned short *dst, size_t count) {
    if (!src || !dst || count % 4 || !aligned16(src) || (reinterpret_cast<size_t>(dst) & 7))
        return static_cast<int>(hipErrorInvalidValue);
    if (count == 0) return 0;
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, n4);
    return static_cast<int>(hipGetLastError());
}

#ifdef VIT_PROBES
int vithip_gemm_bf16_set_variant(int variant) {
    if (variant < 0 || variant > 5) return static_cast<int>(hipErrorInvalidValue);
    g_variant = variant;
    return 0;
}

int vithip_gemm_bf16_set_group(int group_m) {
    if (group_m < 0 || group_m > 1024) return static_cast<int>(hipErrorInvalidValue);
    g_group16 = group_m;
    return 0;
}

int vithip_gemm_bf16_set_max_workgroups(int n) {
    g_max_wgs = n < 0 ? 0 : n;
    return 0;
}

int vithip_gemm_bf16_set_debug_buffer(void *buf) {
    g_dbg = static_cast<unsigned long long *>(buf);
    return 0;
}
#endif

int vithip_patch_embed_bf16(vithip_stream_t stream, const float *images, const unsigned short *conv_w16, const float *conv_b,
                            const float *cls, const float *pos, float *x, unsigned short *patches16, int n_images,
                            int img_size, int patch_size, int in_chans, int embed_dim) {
    if (!images || !conv_w16 || !conv_b || !cls || !pos || !x || !patches16 || n_images <= 0)
        return static_cast<int>(hipErrorInvalidValue);
    if (patch_size <= 0 || patch_size % 8 || img_size % patch_size || in_chans <= 0 || embed_dim % 4)
        return static_cast<int>(hipErrorInvalidValue);
    const int G = img_size / patch_size, P = G * G, K = in_chans * patch_size * patch_size;
    if (K % TBK || K < 2 * TBK) return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(images) || !aligned16(conv_w16) || !aligned16(conv_b) || !aligned16(pos) || !aligned16(x) || !aligned16(patches16))
        return static_cast<int>(hipErrorInvalidValue);
    if ((size_t)n_images * P > 0x7fffffffu / 2) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t work = (size_t)n_images * in_chans * img_size * (img_size / 8);
    size_t blocks = (work + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(patchify_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, images, patches16, n_images, img_size,
                       patch_size, in_chans);
    const int total = n_images * embed_dim;
    hipLaunchKernelGGL(cls_rows_bf16path_kernel, dim3((total + 255) / 256), dim3(256), 0, s, cls, pos, x, n_images, P + 1, embed_dim);
    Bf16Params p{};
    p.A = patches16; p.W = conv_w16; p.bias = conv_b; p.R = pos; p.C = x;
    p.lda = K; p.ldw = K; p.ldr = embed_dim; p.ldc = embed_dim;
    p.M = n_images * P; p.N = embed_dim; p.K = K;
    p.tiles_m = (p.M + TBM - 1) / TBM;
    p.tiles_n = (p.N + TBN - 1) / TBN;
    p.group_m = 8;
    p.patches = P;
    const int g_cus = vitdev::current_cus();  // of the current device, asked per call (engines of several devices share the process)
    if (g_cus <= 0) return static_cast<int>(hipErrorInvalidDevice);
    return vitgemm::launch_gemm_bf16_pp(s, p, VITHIP_BF16_EPI_F32_EMBED, g_cus);
}

int vithip_patch_embed_bf16_implicit(vithip_stream_t stream, const float *images, const unsigned short *conv_w16, const float *conv_b,
                                     const float *cls, const float *pos, float *x, int n_images, int img_size, int patch_size,
                                     int in_chans, int embed_dim) {
    if (!images || !conv_w16 || !conv_b || !cls || !pos || !x || n_images <= 0) return static_cast<int>(hipErrorInvalidValue);
    if (patch_size <= 0 || patch_size % 4 || img_size % patch_size || img_size % 4 || in_chans <= 0 || embed_dim <= 0)
        return static_cast<int>(hipErrorInvalidValue);
    const int K = in_chans * patch_size * patch_size;
    if (K % 32 || (long long)n_images * (img_size / patch_size) * (img_size / patch_size) > (1ll << 24) ||
        (long long)n_images * in_chans * img_size * img_size >= (1ll << 32))
        return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(images) || !aligned16(conv_w16) || !aligned16(x)) return static_cast<int>(hipErrorInvalidValue);
    return vitgemm::launch_patch_embed_bf16(static_cast<hipStream_t>(stream), images, conv_w16, conv_b, cls, pos, x, n_images, img_size,
                                            patch_size, in_chans, embed_dim);
}

int vithip_gemm_bf16(vithip_stream_t stream, const vithip_gemm_bf16_args *a) {
    if (!a || !a->A || !a->W || !a->bias || !a->C) return static_cast<int>(hipErrorInvalidValue);
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->K % TBK || a->N % 4) return static_cast<int>(hipErrorInvalidValue);
    if (a->lda % 8 || a->ldw % 8 || a->lda < a->K || a->ldw < a->K || a->ldc < a->N || a->ldc % 4)
        return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(a->A) || !aligned16(a->W) || !aligned16(a->bias) || !aligned16(a->C))
        return static_cast<int>(hipErrorInvalidValue);
    if (a->epilogue == VITHIP_BF16_EPI_F32_RESIDUAL && (!a->residual || a->ldr < a->N || a->ldr % 4 || !aligned16(a->residual)))
        return static_cast<int>(hipErrorInvalidValue);
    Bf16Params p{};
    p.A = a->A; p.W = a->W; p.bias = a->bias; p.R = a->residual; p.C = a->C;
    p.lda = a->lda; p.ldw = a->ldw; p.ldr = a->ldr; p.ldc = a->ldc;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.tiles_m = (p.M + TBM - 1) / TBM;
    p.tiles_n = (p.N + TBN - 1) / TBN;
    // tile rows per L2 group of the walk (tools/gemm_bf16_group.py, batch 2048, round 4): N = 768 (three tile columns: out_proj,
    // fc2) fetches 6.56 GB per fc2 launch with groups of 8 rows and 4.95 GB with 2, at 1 % less time; wide N (QKV, fc1) is best at 8
    p.group_m = p.tiles_n <= 4 ? 2 : 8;
    if (a->variant < 0 || a->variant > 2) return static_cast<int>(hipErrorInvalidValue);
    // LayerNorm fold: consumer (ln_rows + ln_colsum on the bf16 epilogues) or producer (x16 + row_partials on the residual one)
    const bool ln_consumer = a->ln_rows || a->ln_colsum, ln_producer = a->x16 || a->row_partials;
    if (ln_consumer) {
        if (!a->ln_rows || !a->ln_colsum || ln_producer || (a->epilogue != VITHIP_BF16_EPI_BF16 && a->epilogue != VITHIP_BF16_EPI_BF16_GELU) ||
            (reinterpret_cast<size_t>(a->ln_rows) & 7) || !aligned16(a->ln_colsum))
            return static_cast<int>(hipErrorInvalidValue);
        p.ln_rows = a->ln_rows;
        p.ln_colsum = a->ln_colsum;
    }
    if (ln_producer) {
        if (!a->x16 || !a->row_partials || a->epilogue != VITHIP_BF16_EPI_F32_RESIDUAL || a->ldx16 < a->N || a->ldx16 % 8 ||
            !aligned16(a->x16) || (reinterpret_cast<size_t>(a->row_partials) & 7))
            return static_cast<int>(hipErrorInvalidValue);
        p.x16 = a->x16;
        p.ldx16 = a->ldx16;
        p.partials = a->row_partials;
    }
    int variant = a->variant;
#ifdef VIT_PROBES
    if (g_group16) p.group_m = g_group16;
    if (g_variant) variant = g_variant;
#endif
    const int g_cus = vitdev::current_cus();  // of the current device, asked per call (engines of several devices share the process)
    if (g_cus <= 0) return static_cast<int>(hipErrorInvalidDevice);
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < g_cus ? total : g_cus), block(THREADS);  // one persistent workgroup per CU
    hipStream_t s = static_cast<hipStream_t>(stream);
    // ping-pong kernel: needs two K steps per tile (its bias slot is recycled every second tile) and
    // operands addressable through 32-bit buffer offsets inside one tile (always true: 256 rows)
#ifdef VIT_PROBES
    if (a->epilogue >= 201 && a->epilogue <= 204) return vitgemm::launch_gemm_bf16_pp(s, p, a->epilogue, g_cus);  // timing probes
    if ((g_variant == 5 && a->epilogue != VITHIP_BF16_EPI_F32_RESIDUAL && !ln_consumer) || (a->epilogue >= 501 && a->epilogue <= 505))
        return vitgemm::launch_gemm_bf16_w4(s, p, a->epilogue, g_cus);  // four-wave experiment
#endif
    const bool pp_ok = p.K >= 2 * TBK && a->epilogue >= 0 && a->epilogue <= VITHIP_BF16_EPI_F32_RESIDUAL &&
                       (size_t)p.lda * 2 * 256 < (1u << 31) && (size_t)p.ldw * 2 * 256 < (1u << 31);
    if ((variant >= 2 || ln_consumer || ln_producer) && (!pp_ok || variant == 1)) return static_cast<int>(hipErrorInvalidValue);
#ifdef VIT_PROBES
    if (g_variant == 3) {
        if (!g_dbg) return static_cast<int>(hipErrorInvalidValue);
        p.dbg = g_dbg;
        return vitgemm::launch_gemm_bf16_pp(s, p, 100 + a->epilogue, g_cus);
    }
    if (g_variant == 4) {  // event-log build (bf16 and fp32-residual epilogues)
        if (!g_dbg || a->epilogue == VITHIP_BF16_EPI_BF16_GELU) return static_cast<int>(hipErrorInvalidValue);
        p.dbg = g_dbg;
        return vitgemm::launch_gemm_bf16_pp(s, p, 300 + a->epilogue, g_max_wgs ? g_max_wgs : g_cus);
    }
#endif
    if (variant != 1 && pp_ok) return vitgemm::launch_gemm_bf16_pp(s, p, a->epilogue, g_cus);
    switch (a->epilogue) {
#ifdef VIT_PROBES
        case 101: hipLaunchKernelGGL((gemm_bf16_nt_kernel<VITHIP_BF16_EPI_BF16, 1>), grid, block, 0, s, p); break;
        case 102: hipLaunchKernelGGL((gemm_bf16_nt_kernel<VITHIP_BF16_EPI_BF16, 2>), grid, block, 0, s, p); break;
#endif
        case VITHIP_BF16_EPI_BF16: hipLaunchKernelGGL(gemm_bf16_nt_kernel<VITHIP_BF16_EPI_BF16>, grid, block, 0, s, p); break;
        case VITHIP_BF16_EPI_BF16_GELU: hipLaunchKernelGGL(gemm_bf16_nt_kernel<VITHIP_BF16_EPI_BF16_GELU>, grid, block, 0, s, p); break;
        case VITHIP_BF16_EPI_F32_RESIDUAL: hipLaunchKernelGGL(gemm_bf16_nt_kernel<VITHIP_BF16_EPI_F32_RESIDUAL>, grid, block, 0, s, p); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

}  // extern "C"
