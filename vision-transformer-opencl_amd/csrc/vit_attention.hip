// csrc/vit_attention.hip -- fused scaled-dot-product attention for one (image, head) per workgroup.
//
// Reference: ViT_seq.c:156-215 (scores = q.k / sqrtf(head_dim); row softmax with max
// subtraction; head_out = P.V) -- the OpenCL path runs it as 36 launches per layer chained by
// events (ViT_opencl.c:546-573; kernel.cl:208-284,289-365) and materialises the 197x197
// scores of all 12 heads in global memory.  Here K and V of the head are staged once in LDS
// (2 x 224 x 64 fp32 = 118 KB of the 160 KB), each wave owns a block of 32 query rows and the
// scores never leave registers.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32):
//   S^T tile = K_tile(32 keys x 64) . Q_block^T  -> the accumulator holds, per lane,
//       query = lane&31 and 16 keys (key = 32*kt + (v&3) + 8*(v>>2) + 4*(lane>>5)).
//       A row of the softmax is therefore lane-local (+ one exchange with lane^32).
//   O^T tile = V_tile^T(32 d x keys) . P^T(keys x 32 queries): register v of the S^T
//       accumulator is already the B operand of this product (k = lane>>5 selects
//       key_a / key_a+4), so P goes from softmax to the second MFMA with no data movement.
// The softmax normaliser is applied to O (32 values per lane) instead of P (112 per lane).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vit_device.hpp"
#include "vit_hip_kernels.h"
#ifdef VIT_PROBES
#include "vit_probes.h"
#endif

namespace vitattn {
#ifdef VIT_PROBES
extern unsigned long long *g_res_dbg;
extern unsigned long long *g_stream_dbg;
extern int g_res_mode;
#endif
int attention_f32_resident(hipStream_t s, const float *qkv, float *out, int n_images, int tokens, int heads, int q_rows);  // vit_attention_resident.hip
int attention_bf16_stream(hipStream_t s, const unsigned short *qkv, unsigned short *out, int n_images, int tokens, int heads, bool q_scaled);  // vit_attention_stream.hip
}

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned short bf16_t;  // raw bf16 bits (bf16 variant of the forward: qkv and the output are bf16,
                                // scores, softmax and accumulation stay fp32)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

template <typename IO> __device__ __forceinline__ f32x4 load4(const IO *p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t *p) {
    const uint2 b = *reinterpret_cast<const uint2 *>(p);  // 4 bf16: value = bits << 16
    f32x4 v;
    v[0] = __uint_as_float(b.x << 16);
    v[1] = __uint_as_float(b.x & 0xffff0000u);
    v[2] = __uint_as_float(b.y << 16);
    v[3] = __uint_as_float(b.y & 0xffff0000u);
    return v;
}
template <typename IO> __device__ __forceinline__ void store4(IO *p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t *p, f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (__bf16)v[q];
    *reinterpret_cast<bf16x4 *>(p) = o;
}

constexpr int HD = 64;          // head_dim (ViT-B/16 and ViT-L/16)
constexpr int K_LD = HD + 4;    // padded K row (floats): conflict-free ds_read_b128 over 32 rows
constexpr int ATT_THREADS = 512;
#ifndef ATT_STAGGER
#define ATT_STAGGER 0   // measured at ViT-B/16 batch 2048: 0 / 8 / 14 / 20 x 64 cycles = 0.582 / 0.586 / 0.582 / 0.591 ms: nothing, off
#endif
constexpr int ATT_WAVES = ATT_THREADS / 64;
#ifndef ATT_PROBE
#define ATT_PROBE 0  // timing-only builds of the resident bf16 kernel (tools/build_variant.sh -DATT_PROBE=<bits>; results WRONG): 1 = no v_exp_f32,
#endif               // 2 = no score MFMAs, 4 = no P.V MFMAs, 8 = waves 4-7 idle (no SIMD partner), 16 = no LDS-DMA after the first item

#ifdef VIT_PROBES
unsigned long long *g_attn_dbg = nullptr;  // see vithip_attention_set_debug_buffer() (probe build only)
#else
constexpr unsigned long long *g_attn_dbg = nullptr;
#endif

template <int NKT, typename IO>  // NKT = number of 32-key tiles: tokens <= 32*NKT; IO = float or bf16 bits
__global__ __launch_bounds__(ATT_THREADS) void attention_f32_kernel(const IO *__restrict__ qkv,
                                                                    IO *__restrict__ out,
                                                                    int tokens, int heads, int q_rows,
                                                                    unsigned long long *__restrict__ dbg) {
    // dbg != nullptr (tools/attn_probe.py --stamps): cycle stamps of wave 0 at the phase boundaries
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (dbg) t0 = __builtin_amdgcn_s_memtime();
    constexpr int KEYS = NKT * 32;
    __shared__ __attribute__((aligned(16))) float lds[KEYS * K_LD + KEYS * HD];
    // V first: every V address is then lane_base + a 16-bit immediate (no per-key address VGPRs)
    float *const Vs = lds;
    float *const Ks = lds + KEYS * HD;

    const int head = blockIdx.x, img = blockIdx.y;
    const int D = heads * HD, ld = 3 * D;
    const int tid = threadIdx.x;
    const IO *base = qkv + (size_t)img * tokens * ld + head * HD;

    // ---- stage K and V of this head: 16 float4 per row, rows >= tokens are zero ----------
    // All loads of a thread are issued before the first LDS store, so the ~100 KB of a head arrive
    // at bandwidth rather than as NKT serial round trips.
    {
        const int c4 = (tid & 15) * 4;
        constexpr int ROWS_PER_PASS = ATT_THREADS / 16;  // 32
        f32x4 kreg[NKT], vreg[NKT];
#pragma unroll
        for (int it = 0; it < NKT; ++it) {
            const int row = (tid >> 4) + it * ROWS_PER_PASS;
            const int srow = row < tokens ? row : tokens - 1;  // clamped address, value discarded below
            const IO *src = base + (size_t)srow * ld + c4;
            kreg[it] = load4<IO>(src + D);
            vreg[it] = load4<IO>(src + 2 * D);
        }
#pragma unroll
        for (int it = 0; it < NKT; ++it) {
            const int row = (tid >> 4) + it * ROWS_PER_PASS;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4 *>(Ks + row * K_LD + c4) = row < tokens ? kreg[it] : zero;
            *reinterpret_cast<f32x4 *>(Vs + row * HD + c4) = row < tokens ? vreg[it] : zero;
        }
    }
    __syncthreads();
    if (dbg) t1 = __builtin_amdgcn_s_memtime();

    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nqt = (q_rows + 31) >> 5;  // q_rows <= tokens: only the first q_rows query rows are computed and stored

    // One 32-query block per wave (tokens <= 224 => at most 7 of the 8 waves have work).  No loop
    // over blocks: it would let the compiler hoist ~100 per-key addresses and predicates out of it.
    if (wave < nqt) {
        const int q0 = wave * 32;
        // Q fragment: lane holds Q[q0+r][8c + 4h + s], c = 0..7, s = 0..3 (rows past the end clamp)
        f32x4 qf[8];
        {
            int qrow = q0 + r;
            qrow = qrow < tokens ? qrow : tokens - 1;
            const IO *qsrc = base + (size_t)qrow * ld + h * 4;
#pragma unroll
            for (int c = 0; c < 8; ++c) qf[c] = load4<IO>(qsrc + c * 8);
        }

        // ---- S^T = K . Q^T -----------------------------------------------------------
        // K fragments are double-buffered in registers: the 8 ds_read_b128 of tile kt+1 are issued before
        // the 32 MFMAs of tile kt, so their LDS latency never stalls the matrix pipe.
        f32x16 st[NKT];
        f32x4 kf[2][8];
        auto read_k = [&](int kt, int set) {
            const float *kp = Ks + (kt * 32 + r) * K_LD + h * 4;
#pragma unroll
            for (int c = 0; c < 8; ++c) kf[set][c] = *reinterpret_cast<const f32x4 *>(kp + c * 8);
        };
        read_k(0, 0);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) st[kt][v] = 0.0f;
            if (kt + 1 < NKT) read_k(kt + 1, (kt + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kt & 1][c][s], qf[c][s], st[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        if (dbg) t2 = __builtin_amdgcn_s_memtime();
        // Only the last key tile can hold keys >= tokens (NKT = ceil(tokens/32)).
        // fp32 VALU work is paid for in matrix-pipe time on gfx950 (it shares the lanes of
        // v_mfma_f32_32x32x2_f32), so the softmax is kept to 4 instructions per score:
        // max, then p = exp2(s*c - m*c) as one fma + v_exp_f32 with c = log2(e)/sqrtf(64), then add.
        float mx = -INFINITY;
        const int rem = tokens - (NKT - 1) * 32;  // valid keys in the last tile, 1..32
        const int h4 = 4 * h;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                if (kt == NKT - 1) {
                    const int kloc = (v & 3) + 8 * (v >> 2);  // + 4h = key index inside the tile
                    st[kt][v] = h4 < rem - kloc ? st[kt][v] : -INFINITY;
                }
                mx = fmaxf(mx, st[kt][v]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        constexpr float kScale = 0.125f * 1.4426950408889634f;  // (1/sqrtf(64)) * log2(e)
        const float mxs = -mx * kScale;
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float e = __builtin_amdgcn_exp2f(fmaf(st[kt][v], kScale, mxs));  // exp2(-inf) = 0: masked keys
                st[kt][v] = e;
                sum += e;
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;

        if (dbg) t3 = __builtin_amdgcn_s_memtime();
        // ---- O^T = V^T . P^T -----------------------------------------------------------
        // Groups of 4 accumulator registers = 4 consecutive keys per lane half (8 MFMAs, 8 ds_read_b32);
        // the V values of group g+1 are fetched before the MFMAs of group g.  Tiles before the last are
        // full; in the last tile a group is skipped when its first key is past the end (wave-uniform).
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[dt][v] = 0.0f;
        constexpr int NG = NKT * 4;
        float va[2][8];
        const float *vbase = Vs + 4 * h * HD + r;
        auto read_v = [&](int g, int set) {
            const int key0 = (g >> 2) * 32 + 8 * (g & 3);  // key of register 4*(g&3) on lanes 0-31
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                va[set][2 * q] = vbase[(key0 + q) * HD];
                va[set][2 * q + 1] = vbase[(key0 + q) * HD + 32];
            }
        };
        read_v(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int kt = g >> 2, v0 = 4 * (g & 3);
            if (g + 1 < NG) read_v(g + 1, (g + 1) & 1);  // rows past `tokens` are zero-filled, always readable
            __builtin_amdgcn_sched_barrier(0);
            if (kt < NKT - 1 || kt * 32 + 8 * (g & 3) < tokens) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[g & 1][2 * q], st[kt][v0 + q], o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[g & 1][2 * q + 1], st[kt][v0 + q], o[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        if (dbg) t4 = __builtin_amdgcn_s_memtime();
        // ---- store: lane owns query q0+r; registers 4g..4g+3 are 4 consecutive d ---------
        if (q0 + r < q_rows) {
            IO *dst = out + ((size_t)img * tokens + q0 + r) * D + head * HD + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w;
                    w[0] = o[dt][4 * g + 0] * inv;
                    w[1] = o[dt][4 * g + 1] * inv;
                    w[2] = o[dt][4 * g + 2] * inv;
                    w[3] = o[dt][4 * g + 3] * inv;
                    store4<IO>(dst + dt * 32 + 8 * g, w);
                }
        }
    }
    if (dbg && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long *d = dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3; d[4] = t4; d[5] = __builtin_amdgcn_s_memtime();
    }
}

// ---------------------------------------------------------------------------------------------------
// Any sequence length: K/V of the head stream through LDS in chunks of 224 keys with an online softmax
// (running row maximum m, running sum l, O rescaled by exp2((m_old - m_new) c) when the maximum moves).
// Needed once K and V of one head no longer fit the 160 KB LDS: ViT-L/16-384 has 577 tokens
// (2 x 577 x 64 fp32 = 295 KB).  Same fragment scheme as the resident kernel above; grid =
// (heads, images, blocks of 8 query tiles).
constexpr int CKT = 7;           // key tiles per chunk
constexpr int CKEYS = CKT * 32;  // 224 keys per chunk

template <typename IO>
__global__ __launch_bounds__(ATT_THREADS) void attention_f32_chunked_kernel(const IO *__restrict__ qkv,
                                                                            IO *__restrict__ out, int tokens,
                                                                            int heads) {
    __shared__ __attribute__((aligned(16))) float lds[CKEYS * K_LD + CKEYS * HD];
    float *const Vs = lds;
    float *const Ks = lds + CKEYS * HD;

    const int head = blockIdx.x, img = blockIdx.y;
    const int D = heads * HD, ld = 3 * D;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const IO *base = qkv + (size_t)img * tokens * ld + head * HD;

    const int q0 = (blockIdx.z * ATT_WAVES + wave) * 32;
    const bool active = q0 < tokens;  // wave-uniform

    f32x4 qf[8];
    {
        int qrow = q0 + r;
        qrow = qrow < tokens ? qrow : tokens - 1;
        const IO *qsrc = base + (size_t)qrow * ld + h * 4;
#pragma unroll
        for (int c = 0; c < 8; ++c) qf[c] = load4<IO>(qsrc + c * 8);
    }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) o[dt][v] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;  // l_run: this lane half's share of the row sum
    constexpr float kScale = 0.125f * 1.4426950408889634f;
    const int h4 = 4 * h;

    const int nchunks = (tokens + CKEYS - 1) / CKEYS;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int key_base = ch * CKEYS;
        const int ckeys = tokens - key_base < CKEYS ? tokens - key_base : CKEYS;  // valid keys in this chunk
        if (ch > 0) __syncthreads();  // everybody is done reading the previous chunk
        {
            const int c4 = (tid & 15) * 4;
            constexpr int ROWS_PER_PASS = ATT_THREADS / 16;
            f32x4 kreg[CKT], vreg[CKT];
#pragma unroll
            for (int it = 0; it < CKT; ++it) {
                const int row = (tid >> 4) + it * ROWS_PER_PASS;
                const int srow = key_base + (row < ckeys ? row : ckeys - 1);
                const IO *src = base + (size_t)srow * ld + c4;
                kreg[it] = load4<IO>(src + D);
                vreg[it] = load4<IO>(src + 2 * D);
            }
#pragma unroll
            for (int it = 0; it < CKT; ++it) {
                const int row = (tid >> 4) + it * ROWS_PER_PASS;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(Ks + row * K_LD + c4) = row < ckeys ? kreg[it] : zero;
                *reinterpret_cast<f32x4 *>(Vs + row * HD + c4) = row < ckeys ? vreg[it] : zero;
            }
        }
        __syncthreads();
        if (!active) continue;  // wave-uniform; the barriers above are still reached by every wave

        // ---- S^T = K . Q^T for the chunk; tiles past the end are skipped (and masked below) ------
        f32x16 st[CKT];
#pragma unroll
        for (int kt = 0; kt < CKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) st[kt][v] = 0.0f;
            if (kt * 32 < ckeys) {
                const float *kp = Ks + (kt * 32 + r) * K_LD + h * 4;
                f32x4 kf[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) kf[c] = *reinterpret_cast<const f32x4 *>(kp + c * 8);
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
                        st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c][s2], qf[c][s2], st[kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- online softmax -------------------------------------------------------------------------
        float cmax = -INFINITY;
        if (ckeys < CKEYS) {  // only the last chunk can be ragged
#pragma unroll
            for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int kloc = kt * 32 + (v & 3) + 8 * (v >> 2);
                    st[kt][v] = h4 < ckeys - kloc ? st[kt][v] : -INFINITY;
                }
        }
#pragma unroll
        for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
            for (int v = 0; v < 16; ++v) cmax = fmaxf(cmax, st[kt][v]);
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float m_new = fmaxf(m_run, cmax);                                   // finite: every chunk has a key
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kScale);     // first chunk: exp2(-inf) = 0
        m_run = m_new;
        const float mxs = -m_new * kScale;
        float csum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float e = __builtin_amdgcn_exp2f(fmaf(st[kt][v], kScale, mxs));
                st[kt][v] = e;
                csum += e;
            }
        l_run = l_run * alpha + csum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[dt][v] *= alpha;

        // ---- O^T += V^T . P^T -----------------------------------------------------------------------
        const float *vbase = Vs + 4 * h * HD + r;
#pragma unroll
        for (int g = 0; g < CKT * 4; ++g) {
            const int kt = g >> 2, v0 = 4 * (g & 3);
            const int key0 = kt * 32 + 8 * (g & 3);
            if (key0 < ckeys) {  // wave-uniform; rows past the end are zero-filled and their P is 0
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vbase[(key0 + q) * HD], st[kt][v0 + q], o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vbase[(key0 + q) * HD + 32], st[kt][v0 + q], o[1], 0, 0, 0);
                }
            }
            if ((g & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (active && q0 + r < tokens) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        IO *dst = out + ((size_t)img * tokens + q0 + r) * D + head * HD + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 w;
                w[0] = o[dt][4 * g + 0] * inv;
                w[1] = o[dt][4 * g + 1] * inv;
                w[2] = o[dt][4 * g + 2] * inv;
                w[3] = o[dt][4 * g + 3] * inv;
                store4<IO>(dst + dt * 32 + 8 * g, w);
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// bf16 MFMA attention for the bf16 variant (tokens <= 224).  Same structure as attention_f32_kernel,
// but both products run on v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate), which turns the kernel
// from matrix-bound into HBM-bound (2.5 GB of qkv/out per layer at batch 2048):
//   * K of the head sits in LDS as bf16 rows of 128 B with the GEMM's XOR swizzle (conflict-free
//     ds_read_b128 of the A fragments);
//   * V sits row-major like K (128-B rows, 16-B chunk c of row k at c ^ 4*((k>>1)&1)); the P.V product needs, per
//     lane, 8 keys of ONE d as its A fragment, which gfx950's transposing LDS read delivers directly:
//     ds_read_b64_tr_b16 hands lane i of every 16-lane group column i of a 4-key x 16-d block.  (The first version
//     transposed V while staging, 32 two-byte LDS stores per thread: a quarter of the kernel.)
//   * P goes from the S^T accumulator to the B operand of P.V in registers: registers 8s..8s+7 of a
//     32x32 accumulator, packed to bf16, are exactly the k-step-s fragment whose element j of lane half h
//     is key 16s + 8(j>>2) + 4h + (j&3) -- the V^T fragment is gathered in that same order.
// Softmax, max and sums are fp32; P is rounded to bf16 once (the documented cost of the bf16 variant).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4s;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4v lds_short4v;
typedef __attribute__((address_space(3))) void lds_void_t;

// A fragment of O^T = V^T . P^T for d-tile dt and the 16 keys starting at key16 (a multiple of 16), in the key order
// of the P fragment: element j of lane half h is key key16 + 8(j>>2) + 4h + (j&3).  Vs: [keys][64] bf16, swizzled as
// above.  Per 16-lane group the transposing read takes a block of 4 keys x 16 d: lane 4q+p of the group supplies the
// address of key row q, d columns 4p..4p+3, lane i receives column i.  EXEC must be all ones (wave-uniform callers).
__device__ __forceinline__ bf16x8 v_fragment_tr(const bf16_t *Vs, int key16, int dt, int lane) {
    const int h = lane >> 5, g2 = (lane >> 4) & 1, q = (lane & 15) >> 2, pq = lane & 3;
    const int row = key16 + 4 * h + q;                     // (row >> 1) & 1 == (q >> 1) & 1: key16 + 4h is a multiple of 4
    const int chunk = (dt * 4 + 2 * g2 + (pq >> 1)) ^ (4 * ((q >> 1) & 1));
    const bf16_t *ptr = Vs + row * HD + chunk * 8 + 4 * (pq & 1);
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v *)ptr);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v *)(ptr + 8 * HD));
    typedef short short8v __attribute__((ext_vector_type(8)));
    const short8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// Persistent: a workgroup walks (image, head) items.  K and V of the NEXT item arrive by LDS-DMA in a second LDS image and its Q
// fragments in registers while the current item multiplies, so the HBM round trip that used to open every workgroup hides under
// the matrix and softmax work, and ONE barrier separates two items (round 5; rounds 2-4 staged K / V through 28 registers per
// lane and wrote them to a single LDS image between two barriers: 0.572 -> 0.561 ms per launch at batch 2048, 238 -> 213 VGPRs,
// same bits, profiles/r05/experiments/attention_resident_bf16_dma.jsonl).
template <int NKT>
__global__ __launch_bounds__(ATT_THREADS) void attention_bf16_kernel(const bf16_t *__restrict__ qkv,
                                                                     bf16_t *__restrict__ out, int tokens, int heads,
                                                                     int items, int q_rows, float kScale) {
    // kScale: (1/sqrtf(64)) * log2(e) for plain q, 1 when the Q columns already hold that multiple of q (vithip_attention_bf16io_qscaled)
    constexpr int KEYS = NKT * 32;
    constexpr int IMAGE = 2 * KEYS * HD;    // one head's K and V
    __shared__ __attribute__((aligned(1024))) bf16_t lds[2 * IMAGE];  // the current item's image and the next one's
    const bf16_t *Ks = lds;                 // [KEYS][64], chunk-swizzled for row reads
    const bf16_t *Vs = lds + KEYS * HD;     // [KEYS][64], chunk-swizzled for transposing reads

    const int D = heads * HD, ld = 3 * D;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nqt = (q_rows + 31) >> 5;  // only the first q_rows query rows are computed and stored
    const int q0 = wave * 32;
    const bool computes = wave < ((ATT_PROBE & 8) ? 4 : nqt);  // wave-uniform

    bf16x8 qf[4];
    auto item_base = [&](int item) { return qkv + (size_t)(item / heads) * tokens * ld + (item % heads) * HD; };
    // The head's K and V go straight to LDS (buffer_load ... lds: no staging registers, no ds_write, no write phase between two
    // barriers), 8 rows x 128 B per wave instruction, pieces dealt round-robin to the 8 waves.  The LDS side of an LDS-DMA is
    // linear (lane l lands at byte 16 l of the piece), so both swizzles are applied to the SOURCE chunk; rows past the last token
    // are outside the descriptor's range and read as zero.
    // (All of a wave's pieces go out in one burst when the item starts.  Dealt out in four portions through the item instead -- after
    // the scores, after the maximum, in the middle of P.V -- the launch takes 0.584 against 0.558 ms: the earlier a piece is asked
    // for the better, and an LDS-DMA between the products costs issue slots.)
    auto dma = [&](int item, int img_idx) __attribute__((always_inline)) {
        const bf16_t *base = item_base(item);
        const unsigned bytes = (unsigned)(tokens - 1) * ld * 2 + HD * 2;  // up to the end of the last token's 64 values
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(base + D), 0, bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(base + 2 * D), 0, bytes, 0x00020000);
        bf16_t *kd = lds + img_idx * IMAGE, *vd = kd + KEYS * HD;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int k = 0; k < (KEYS / 8 + ATT_WAVES - 1) / ATT_WAVES; ++k) {
            const int piece = wv + ATT_WAVES * k;  // 8 rows
            if (piece < KEYS / 8) {
                const int row = 8 * piece + (lane >> 3), cp = lane & 7;
                const int kvoff = row * ld * 2 + ((cp ^ ((row >> 1) & 7)) * 16);
                const int vvoff = row * ld * 2 + ((cp ^ (4 * ((row >> 1) & 1))) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_t *)(kd + piece * 8 * HD), 16, kvoff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void_t *)(vd + piece * 8 * HD), 16, vvoff, 0, 0, 0);
            }
        }
    };
    auto fetch_q = [&](int item) {  // this wave's Q fragments
        const bf16_t *base = item_base(item);
        int qrow = q0 + r;
        qrow = qrow < tokens ? qrow : tokens - 1;
        const bf16_t *qsrc = base + (size_t)qrow * ld + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(qsrc + ks * 16);
    };

    int item = blockIdx.x;
    if (item >= items) return;  // workgroup-uniform
    int cur = 0;  // LDS image of the current item
    dma(item, 0);
    fetch_q(item);
    for (;;) {
        bf16x8 qc[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qc[ks] = qf[ks];
        // every wave's pieces of THIS item have landed (they were issued a whole item ago, the Q fragments with them), and behind the
        // barrier everybody has finished reading the other image (the previous item), which the next item's pieces now overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Ks = lds + cur * IMAGE;
        Vs = Ks + KEYS * HD;
        __syncthreads();
        const int next = item + gridDim.x;
        if (next < items) {  // in flight during everything below
            if (!(ATT_PROBE & 16)) dma(next, cur ^ 1);
            fetch_q(next);
        }

        if (computes) {
            // The two waves of a SIMD (w and w + 4) leave the barrier together and would run their matrix phases (S, then P.V) and
            // their softmax in step -- twice the matrix time, no overlap of the two pipes.  Waves 4-7 start ATT_STAGGER x 64 cycles
            // late: their S falls into their partners' softmax, their softmax into the partners' P.V (vit_attention_stream.hip).
            if (ATT_STAGGER > 0 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_sleep(ATT_STAGGER);
            // ---- S^T = K . Q^T ------------------------------------------------------------------------
            // (Tried in round 5: the key tiles in pairs, so that the four dependent products of a tile alternate with another tile's --
            // hipcc lays most of a tile's four back to back.  0.5607 against 0.5623 ms: the SIMD's other wave already fills those gaps.)
            f32x16 st[NKT];
            const int sw = (r >> 1) & 7;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int v = 0; v < 16; ++v) st[kt][v] = 0.0f;
                const bf16_t *krow = Ks + (kt * 32 + r) * HD;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(krow + (((2 * ks + h) ^ sw) & 7) * 8);
                    if (!(ATT_PROBE & 2)) st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qc[ks], st[kt], 0, 0, 0);
                }
            }

            // ---- row softmax (fp32) ---------------------------------------------------------------------
            float mx = -INFINITY;
            const int rem = tokens - (NKT - 1) * 32;
            const int h4 = 4 * h;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    if (kt == NKT - 1) {
                        const int kloc = (v & 3) + 8 * (v >> 2);
                        st[kt][v] = h4 < rem - kloc ? st[kt][v] : -INFINITY;
                    }
                    mx = fmaxf(mx, st[kt][v]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mxs = -mx * kScale;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int v = 0; v < 16; ++v) st[kt][v] = (ATT_PROBE & 1) ? fmaf(st[kt][v], kScale, mxs) : __builtin_amdgcn_exp2f(fmaf(st[kt][v], kScale, mxs));

            // ---- O^T = V^T . P^T, and the row sums from the same pipe ---------------------------------------
            // One more MFMA per 16-key group with an all-ones A operand: every element of `lsum` becomes the sum of this lane's
            // column of P over the group's keys (both half-waves' keys: the product runs over k) -- the sum of the probabilities AS
            // ROUNDED to bf16, the weights P.V really uses.  16 x NKT v_add_f32 per lane and the half-wave exchange leave the vector
            // pipe, the busier one in this kernel.  Only element 0 is read and the elements are independent sums: the other 15
            // registers stay uninitialised (the empty asm "defines" them without an instruction).
            f32x16 lsum;
            asm volatile("" : "=v"(lsum));
            lsum[0] = 0.0f;
            const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
            f32x16 o[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int v = 0; v < 16; ++v) o[dt][v] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (kt < NKT - 1 || kt * 32 + 16 * s2 < tokens) {  // wave-uniform: skip 16-key blocks past the end
                        bf16x8 pf;
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)st[kt][8 * s2 + j];
                        if constexpr (ATT_PROBE & 4) {  // the conversions and the transposing reads stay, the products go
#pragma unroll
                            for (int dt = 0; dt < 2; ++dt) {
                                const bf16x8 vf = v_fragment_tr(Vs, kt * 32 + 16 * s2, dt, lane);
                                asm volatile("" ::"v"(vf), "v"(pf));
                            }
                            continue;
                        }
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
                            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_fragment_tr(Vs, kt * 32 + 16 * s2, dt, lane), pf, o[dt], 0, 0, 0);
                        lsum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lsum, 0, 0, 0);
                    }
                }
            }
            const float inv = 1.0f / lsum[0];

            {
                // 16-byte stores: v_permlane32_swap trades column group g of the upper half-wave for group g + 1 of the lower one,
                // so lanes 0-31 own d = 8g..8g+7 and lanes 32-63 d = 8g+8..8g+15 of their rows (half the store instructions of the
                // 8-byte row-per-lane form; the exchange needs every lane of the computing wave)
                const int img = item / heads, head = item % heads;
                bf16_t *dst = out + ((size_t)img * tokens + q0 + r) * D + head * HD + 8 * h;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        bf16x4s w0, w1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            w0[q] = (__bf16)(o[dt][4 * g + q] * inv);
                            w1[q] = (__bf16)(o[dt][4 * (g + 1) + q] * inv);
                        }
                        const uint2 a = __builtin_bit_cast(uint2, w0), c = __builtin_bit_cast(uint2, w1);
                        auto sx = __builtin_amdgcn_permlane32_swap(a.x, c.x, false, false);
                        auto sy = __builtin_amdgcn_permlane32_swap(a.y, c.y, false, false);
                        const uint4 piece = {sx[0], sy[0], sx[1], sy[1]};
                        if (q0 + r < q_rows) *reinterpret_cast<uint4 *>(dst + dt * 32 + 8 * g) = piece;
                    }
            }
        }
        if (next >= items) break;
        item = next;
        cur ^= 1;
    }
}

// bf16 MFMA attention for any sequence length: K (swizzled rows) and V (transposed) stream through LDS in
// chunks of 224 keys with the online softmax of attention_f32_chunked_kernel.  ViT-L/16-384 (577 tokens).
__global__ __launch_bounds__(ATT_THREADS) void attention_bf16_chunked_kernel(const bf16_t *__restrict__ qkv,
                                                                             bf16_t *__restrict__ out, int tokens,
                                                                             int heads, float kScale) {
    // kScale: (1/sqrtf(64)) * log2(e) for plain q, 1 when the Q columns already hold that multiple of q (as in attention_bf16_kernel)
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * CKEYS * HD];
    bf16_t *const Ks = lds;
    bf16_t *const Vs = lds + CKEYS * HD;

    const int head = blockIdx.x, img = blockIdx.y;
    const int D = heads * HD, ld = 3 * D;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bf16_t *base = qkv + (size_t)img * tokens * ld + head * HD;

    const int q0 = (blockIdx.z * ATT_WAVES + wave) * 32;
    const bool active = q0 < tokens;  // wave-uniform
    bf16x8 qf[4];
    {
        int qrow = q0 + r;
        qrow = qrow < tokens ? qrow : tokens - 1;
        const bf16_t *qsrc = base + (size_t)qrow * ld + h * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(qsrc + ks * 16);
    }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) o[dt][v] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const int h4 = 4 * h, sw = (r >> 1) & 7;

    // the K/V rows of chunk ch+1 are fetched into registers while chunk ch is multiplied (one workgroup per CU fits,
    // so nothing else hides that HBM round trip)
    const int nchunks = (tokens + CKEYS - 1) / CKEYS;
    const int c8 = tid & 7;
    constexpr int ROWS_PER_PASS = ATT_THREADS / 8;
    constexpr int PASSES = (CKEYS + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
    uint4 kreg[PASSES], vreg[PASSES];
    auto fetch = [&](int ch) {
        const int key_base = ch * CKEYS;
        const int ckeys = tokens - key_base < CKEYS ? tokens - key_base : CKEYS;
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int row = (tid >> 3) + it * ROWS_PER_PASS;
            const int srow = key_base + (row < ckeys ? row : ckeys - 1);
            const bf16_t *src = base + (size_t)srow * ld + c8 * 8;
            kreg[it] = *reinterpret_cast<const uint4 *>(src + D);
            vreg[it] = *reinterpret_cast<const uint4 *>(src + 2 * D);
        }
    };
    fetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int key_base = ch * CKEYS;
        const int ckeys = tokens - key_base < CKEYS ? tokens - key_base : CKEYS;
        if (ch > 0) __syncthreads();
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int row = (tid >> 3) + it * ROWS_PER_PASS;
            if (row < CKEYS) {
                const bool ok = row < ckeys;
                const uint4 zero = {0u, 0u, 0u, 0u};
                const uint4 kv = ok ? kreg[it] : zero, vv = ok ? vreg[it] : zero;
                *reinterpret_cast<uint4 *>(Ks + row * HD + ((c8 ^ ((row >> 1) & 7)) * 8)) = kv;
                *reinterpret_cast<uint4 *>(Vs + row * HD + ((c8 ^ (4 * ((row >> 1) & 1))) * 8)) = vv;
            }
        }
        __syncthreads();
        if (ch + 1 < nchunks) fetch(ch + 1);
        if (!active) continue;

        f32x16 st[CKT];
#pragma unroll
        for (int kt = 0; kt < CKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) st[kt][v] = 0.0f;
            if (kt * 32 < ckeys) {
                const bf16_t *krow = Ks + (kt * 32 + r) * HD;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(krow + (((2 * ks + h) ^ sw) & 7) * 8);
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kt], 0, 0, 0);
                }
            }
        }
        if (ckeys < CKEYS) {
#pragma unroll
            for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int kloc = kt * 32 + (v & 3) + 8 * (v >> 2);
                    st[kt][v] = h4 < ckeys - kloc ? st[kt][v] : -INFINITY;
                }
        }
        float cmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
            for (int v = 0; v < 16; ++v) cmax = fmaxf(cmax, st[kt][v]);
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float m_new = fmaxf(m_run, cmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kScale);
        m_run = m_new;
        const float mxs = -m_new * kScale;
        float csum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < CKT; ++kt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float e = __builtin_amdgcn_exp2f(fmaf(st[kt][v], kScale, mxs));
                st[kt][v] = e;
                csum += e;
            }
        l_run = l_run * alpha + csum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[dt][v] *= alpha;

#pragma unroll
        for (int kt = 0; kt < CKT; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (kt * 32 + 16 * s2 < ckeys) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (__bf16)st[kt][8 * s2 + j];
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_fragment_tr(Vs, kt * 32 + 16 * s2, dt, lane), pf, o[dt], 0, 0, 0);
                }
            }
        }
    }

    if (active && q0 + r < tokens) {
        const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
        bf16_t *dst = out + ((size_t)img * tokens + q0 + r) * D + head * HD + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 w;
                w[0] = o[dt][4 * g + 0] * inv;
                w[1] = o[dt][4 * g + 1] * inv;
                w[2] = o[dt][4 * g + 2] * inv;
                w[3] = o[dt][4 * g + 3] * inv;
                store4<bf16_t>(dst + dt * 32 + 8 * g, w);
            }
    }
}

template <int NKT>
int launch_bf16(hipStream_t s, const bf16_t *qkv, bf16_t *out, int n_images, int tokens, int heads, int q_rows, bool q_scaled) {
    const int items = heads * n_images;
    const int cus = vitdev::current_cus();
    if (cus <= 0) return static_cast<int>(hipErrorInvalidDevice);
    const int per_cu = NKT <= 3 ? 2 : 1;  // VGPR-limited residency (8 waves per workgroup)
    const int grid = items < cus * per_cu ? items : cus * per_cu;
    hipLaunchKernelGGL(attention_bf16_kernel<NKT>, dim3(grid), dim3(ATT_THREADS), 0, s, qkv, out, tokens, heads, items, q_rows,
                       q_scaled ? 1.0f : 0.125f * 1.4426950408889634f);
    return static_cast<int>(hipGetLastError());
}

template <int NKT, typename IO>
int launch(hipStream_t s, const IO *qkv, IO *out, int n_images, int tokens, int heads, int q_rows) {
    hipLaunchKernelGGL((attention_f32_kernel<NKT, IO>), dim3(heads, n_images), dim3(ATT_THREADS), 0, s, qkv, out,
                       tokens, heads, q_rows, g_attn_dbg);
    return static_cast<int>(hipGetLastError());
}

template <typename IO>
int attention_dispatch(hipStream_t s, const IO *qkv, IO *out, int n_images, int tokens, int heads, int q_rows) {
    if (!qkv || !out || n_images <= 0 || tokens <= 0 || heads <= 0 || q_rows <= 0 || q_rows > tokens)
        return static_cast<int>(hipErrorInvalidValue);
    if ((reinterpret_cast<size_t>(qkv) & 15) || (reinterpret_cast<size_t>(out) & 15))
        return static_cast<int>(hipErrorInvalidValue);
    const int nkt = (tokens + 31) / 32;
    if constexpr (std::is_same<IO, float>::value) {
        // fp32 I/O up to 224 tokens: the 16x16x4 kernel with LDS-DMA staging (vit_attention_resident.hip)
        if (tokens <= 224) return vitattn::attention_f32_resident(s, qkv, out, n_images, tokens, heads, q_rows);
    }
    switch (nkt) {
        case 1: return launch<1, IO>(s, qkv, out, n_images, tokens, heads, q_rows);
        case 2: return launch<2, IO>(s, qkv, out, n_images, tokens, heads, q_rows);
        case 3: return launch<3, IO>(s, qkv, out, n_images, tokens, heads, q_rows);
        case 4: return launch<4, IO>(s, qkv, out, n_images, tokens, heads, q_rows);
        case 5: return launch<5, IO>(s, qkv, out, n_images, tokens, heads, q_rows);
        case 6: return launch<6, IO>(s, qkv, out, n_images, tokens, heads, q_rows);
        case 7: return launch<7, IO>(s, qkv, out, n_images, tokens, heads, q_rows);
        default: {  // > 224 tokens: K/V stream through LDS in chunks, online softmax (computes every query row)
            const int qblocks = (nkt + ATT_WAVES - 1) / ATT_WAVES;
            hipLaunchKernelGGL(attention_f32_chunked_kernel<IO>, dim3(heads, n_images, qblocks), dim3(ATT_THREADS), 0, s,
                               qkv, out, tokens, heads);
            return static_cast<int>(hipGetLastError());
        }
    }
}

}  // namespace

#ifdef VIT_PROBES
extern "C" int vithip_attention_set_probe_mode(int mode) {
    vitattn::g_res_mode = mode;
    return 0;
}
// Probe hook: 8 x u64 cycle stamps per (image, head) workgroup; nullptr (default) disables them.
extern "C" int vithip_attention_set_debug_buffer(void *buf) {
    g_attn_dbg = static_cast<unsigned long long *>(buf);
    vitattn::g_res_dbg = g_attn_dbg;
    vitattn::g_stream_dbg = g_attn_dbg;
    return 0;
}
#endif

extern "C" int vithip_attention_f32(vithip_stream_t stream, const float *qkv, float *out,
                                    int n_images, int tokens, int heads) {
    return attention_dispatch<float>(static_cast<hipStream_t>(stream), qkv, out, n_images, tokens, heads, tokens);
}

// bf16 variant: qkv and out hold bf16 bits; K/V are widened to fp32 while staged into LDS and all
// arithmetic (fp32 MFMA, softmax) is the same as above; the output is rounded to bf16 once.
static int attention_bf16io_q(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out, int n_images, int tokens,
                              int heads, int q_rows, bool bf16_mfma = true, bool q_scaled = false);

// As vithip_attention_bf16io / _rows, for Q columns that already hold 0.125 * log2(e) * q (the factor of the scores' exponent,
// folded into the in_proj weights by the engine: q is then rounded to bf16 once, as before, and the kernels save the scaling).
// q_rows < tokens needs tokens <= 224.
extern "C" int vithip_attention_bf16io_qscaled(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out,
                                               int n_images, int tokens, int heads, int q_rows) {
    if (q_rows != tokens && tokens > 224) return static_cast<int>(hipErrorInvalidValue);
    return attention_bf16io_q(stream, qkv, out, n_images, tokens, heads, q_rows, true, true);
}

extern "C" int vithip_attention_bf16io(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out,
                                       int n_images, int tokens, int heads) {
    return attention_bf16io_q(stream, qkv, out, n_images, tokens, heads, tokens);
}

// Same I/O, fp32 arithmetic: K/V widened to fp32 in LDS, both products on the fp32 matrix pipe (the cross-check of the
// bf16-MFMA kernel: only the final rounding of the output to bf16 differs from vithip_attention_f32).
extern "C" int vithip_attention_bf16io_f32math(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out,
                                               int n_images, int tokens, int heads) {
    return attention_bf16io_q(stream, qkv, out, n_images, tokens, heads, tokens, false);
}

// Only the first q_rows query rows of every image (the class token for q_rows = 1; tokens <= 224): used by the engine's
// exact last-layer pruning -- everything after the last attention depends on row 0 alone.
extern "C" int vithip_attention_f32_rows(vithip_stream_t stream, const float *qkv, float *out, int n_images, int tokens,
                                         int heads, int q_rows) {
    if (tokens > 224) return static_cast<int>(hipErrorInvalidValue);
    return attention_dispatch<float>(static_cast<hipStream_t>(stream), qkv, out, n_images, tokens, heads, q_rows);
}

extern "C" int vithip_attention_bf16io_rows(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out,
                                            int n_images, int tokens, int heads, int q_rows) {
    if (tokens > 224) return static_cast<int>(hipErrorInvalidValue);
    return attention_bf16io_q(stream, qkv, out, n_images, tokens, heads, q_rows);
}

static int attention_bf16io_q(vithip_stream_t stream, const unsigned short *qkv, unsigned short *out, int n_images, int tokens,
                              int heads, int q_rows, bool bf16_mfma, bool q_scaled) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (q_rows <= 0 || q_rows > tokens) return static_cast<int>(hipErrorInvalidValue);
    if (!qkv || !out || n_images <= 0 || tokens <= 0 || heads <= 0 || (reinterpret_cast<size_t>(qkv) & 15) ||
        (reinterpret_cast<size_t>(out) & 15))
        return static_cast<int>(hipErrorInvalidValue);
    if (bf16_mfma) {  // bf16 matrix path while K/V of a head fit LDS; longer sequences use the chunked kernel
        switch ((tokens + 31) / 32) {
            case 1: return launch_bf16<1>(s, qkv, out, n_images, tokens, heads, q_rows, q_scaled);
            case 2: return launch_bf16<2>(s, qkv, out, n_images, tokens, heads, q_rows, q_scaled);
            case 3: return launch_bf16<3>(s, qkv, out, n_images, tokens, heads, q_rows, q_scaled);
            case 4: return launch_bf16<4>(s, qkv, out, n_images, tokens, heads, q_rows, q_scaled);
            case 5: return launch_bf16<5>(s, qkv, out, n_images, tokens, heads, q_rows, q_scaled);
            case 6: return launch_bf16<6>(s, qkv, out, n_images, tokens, heads, q_rows, q_scaled);
            case 7: return launch_bf16<7>(s, qkv, out, n_images, tokens, heads, q_rows, q_scaled);
            default: {
                // 225..704 tokens (ViT-L/16-384: 577): all query blocks of a head in one persistent workgroup, K/V streamed
                // once through an LDS-DMA ring (vit_attention_stream.hip); longer sequences: the chunked kernel below
                if (tokens <= 704) return vitattn::attention_bf16_stream(s, qkv, out, n_images, tokens, heads, q_scaled);
                const int qblocks = ((tokens + 31) / 32 + ATT_WAVES - 1) / ATT_WAVES;
                hipLaunchKernelGGL(attention_bf16_chunked_kernel, dim3(heads, n_images, qblocks), dim3(ATT_THREADS), 0, s, qkv,
                                   out, tokens, heads, q_scaled ? 1.0f : 0.125f * 1.4426950408889634f);
                return static_cast<int>(hipGetLastError());
            }
        }
    }
    if (q_scaled) return static_cast<int>(hipErrorInvalidValue);  // the fp32-arithmetic cross-check takes plain q
    return attention_dispatch<bf16_t>(s, qkv, out, n_images, tokens, heads, q_rows);
}
