// csrc/vit_attention_stream.hip -- bf16 attention for sequences that do not fit the LDS whole (225..768 tokens):
// ViT-L/16-384 has 577 tokens (BASELINE.json configs[4]).
//
// Reference semantics: ViT_seq.c:156-215 (scores / sqrtf(64), row softmax with max subtraction, P.V); bf16 operands,
// fp32 softmax and accumulation (the bf16 variant's bar is in tests/test_gpu_bf16.py).
//
// The first version (attention_bf16_chunked_kernel, vit_attention.hip) ran at 15 % of the bf16 matrix roofline at
// ViT-L/16-384, batch 1024 (3.7 ms per launch).  Where the time went:
//   * grid = (head, image, block of 256 queries): K/V of a head were staged THREE times, and the third query block holds
//     65 of 256 rows (577 = 2 x 256 + 65): 5 of its 8 waves only waited at barriers;
//   * 224-key chunks: 577 = 224 + 224 + 129, the third chunk 42 % padding; every chunk single-buffered between two
//     barriers, its rows fetched through registers (32 VGPRs).
// This version: one persistent workgroup per CU walks (image, head) items and keeps ALL query blocks of the head -- a wave
// owns up to three 32-row blocks (19 blocks over 8 waves: 3,3,3,2,2,2,2,2; per SIMD 5,5,5,4) with their online-softmax
// state (running max, sum, O^T accumulators) in registers.  K/V stream ONCE per head through a double-buffered LDS ring in
// chunks of 128 keys, sized to the sequence (19 key tiles = 4+4+4+4+3, no padded tile), filled by LDS-DMA
// (buffer_load ... lds) one chunk ahead -- the next item's first chunk behind the current item's last -- so a step is
// [issue DMA of step s+1] [S = K.Q^T, online softmax, O += V^T.P^T for my blocks] [one barrier].
// MFMA layouts (v_mfma_f32_32x32x16_bf16, transposing LDS reads for V^T, P from accumulator to B operand in registers)
// are those of attention_bf16_kernel.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vit_hip_kernels.h"

namespace vitattn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) short4v lds_short4v;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int SHD = 64;                  // head_dim
constexpr int ST_WAVES = 8, ST_THREADS = ST_WAVES * 64;
constexpr int SKT = 4;                   // key tiles (of 32) per chunk
constexpr int SKEYS = SKT * 32;          // 128 keys
constexpr int SBUF = 2 * SKEYS * SHD;    // bf16 elements of one ring slot: K[128][64] then V[128][64]
constexpr int MAXB = 3;                  // query blocks per wave: tokens <= 8 * 3 * 32 = 768
constexpr int SUB = 2;                   // key tiles whose scores are in registers at a time
constexpr float kScaleS = 0.125f * 1.4426950408889634f;

// A fragment of O^T = V^T . P^T for d-tile dt and the 16 keys from key16 (see attention_bf16_kernel): lane 4q+p of a
// 16-lane group addresses key row q, d columns 4p..4p+3 of a 4 x 16 block and receives column i.  EXEC all ones.
__device__ __forceinline__ bf16x8 v_frag_tr(const bf16_t *Vs, int key16, int dt, int lane) {
    const int h = lane >> 5, g2 = (lane >> 4) & 1, q = (lane & 15) >> 2, pq = lane & 3;
    const int row = key16 + 4 * h + q;
    const int chunk = (dt * 4 + 2 * g2 + (pq >> 1)) ^ (4 * ((q >> 1) & 1));
    const bf16_t *ptr = Vs + row * SHD + chunk * 8 + 4 * (pq & 1);
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v *)ptr);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v *)(ptr + 8 * SHD));
    const short8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

__global__ __launch_bounds__(ST_THREADS) void attention_bf16_stream_kernel(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ out,
                                                                           int tokens, int heads, int n_items) {
    __shared__ __attribute__((aligned(1024))) bf16_t lds[2 * SBUF];  // 2 x (16 KB + 16 KB)

    const int D = heads * SHD, ld = 3 * D;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int tk = tokens;  // made opaque once per step (asm below) so that hipcc keeps the tail masks out of hoisted registers

    // chunk plan: nkt key tiles in nch chunks of at most SKT tiles, sizes differing by at most one
    const int nkt = (tokens + 31) >> 5;
    const int nch = (nkt + SKT - 1) / SKT;
    const int base_t = nkt / nch, extra = nkt - base_t * nch;  // chunks [0, extra) hold base_t + 1 tiles
    auto chunk_tiles = [&](int ch) { return base_t + (ch < extra ? 1 : 0); };
    auto chunk_first = [&](int ch) { return ch * base_t + (ch < extra ? ch : extra); };  // first tile of chunk ch

    auto item_base = [&](int item) { return qkv + (size_t)(item / heads) * tokens * ld + (item % heads) * SHD; };

    // ---- LDS-DMA of one chunk: 8 rows of 128 B per wave instruction; lane = 8 row_l + pos holds source chunk pos ^ swz(row)
    //   K: stored position = chunk ^ ((row >> 1) & 7), row = 8 t + row_l  ->  (4 (t & 1) + (row_l >> 1)) & 7
    //   V: stored position = chunk ^ (4 ((row >> 1) & 1)): independent of t
    // The per-lane offsets are recomputed from the lane id at every step (a dozen VALU instructions): kept across the step
    // they would sit in registers this kernel does not have, and come back from scratch behind a vmcnt(0).
    // The DMA instruction is written as inline asm.  With the builtin, hipcc assumes that an LDS read may alias a pending
    // LDS-DMA whenever the ring slot is a run-time value and puts s_waitcnt vmcnt(0) in front of EVERY fragment read,
    // which drains the ring in the step that is supposed to hide it; specialising the step per slot doubled the live
    // state at the join (148 spilled VGPRs).  The ordering the hardware needs is explicit instead: ring_barrier() below.
    typedef int int4v __attribute__((ext_vector_type(4)));
    auto make_rsrc4 = [&](const bf16_t *base) __attribute__((always_inline)) {  // raw buffer descriptor: base, stride 0, no bounds
        const unsigned long long a = reinterpret_cast<unsigned long long>(base);
        int4v r4;
        r4[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        r4[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
        r4[2] = 0x7fffffff;
        r4[3] = 0x00020000;
        return r4;
    };
    auto dma16 = [&](int4v r4, const bf16_t *lds_dst, int voff, int soff) __attribute__((always_inline)) {
        const unsigned m0v = (unsigned)(size_t)(lds_void *)lds_dst;  // LDS byte address of the wave's 1 KB piece
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(m0v), "v"(voff), "s"(r4), "s"(soff) : "memory", "m0");
    };
    // every wave's DMA pieces have landed (vmcnt counts them) and every wave has finished reading the other slot
    auto ring_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto dma_chunk = [&](int item, int ch, int slot) __attribute__((always_inline)) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));  // opaque: nothing below is loop-invariant to the compiler
        const int row_l = lane_l >> 3, pos = lane_l & 7;
        const int lane_row = row_l * ld * 2;  // bytes
        const int kofs = lane_row + ((pos ^ ((4 * (wave & 1) + (row_l >> 1)) & 7)) << 4);  // t & 1 == wave & 1
        const int vofs = lane_row + ((pos ^ (4 * ((row_l >> 1) & 1))) << 4);
        const bf16_t *base = item_base(item);
        const int key0 = chunk_first(ch) * 32;
        const int rows = chunk_tiles(ch) * 32;  // 96 or 128 here, any multiple of 32 up to SKEYS in general
        const int4v rk = make_rsrc4(base + D), rv = make_rsrc4(base + 2 * D);
        bf16_t *Kd = lds + slot * SBUF, *Vd = Kd + SKEYS * SHD;
#pragma unroll
        for (int k = 0; k < SKEYS / 8 / ST_WAVES; ++k) {  // 16 groups of 8 rows, 2 per wave
            const int t = wave + ST_WAVES * k;
            if (8 * t < rows) {  // wave-uniform
                const int krow0 = key0 + 8 * t;
                if (krow0 + 7 < tk) {  // all eight rows exist: per-lane offsets are the three precomputed ones
                    const int so = krow0 * ld * 2;
                    dma16(rk, Kd + t * 512, kofs, so);
                    dma16(rv, Vd + t * 512, vofs, so);
                } else {  // rows past the last token: clamped (finite values; those keys are masked / weigh 0)
                    const int lrow = 8 * t + row_l;
                    int srow = krow0 + row_l;
                    srow = srow < tk ? srow : tk - 1;
                    const int so = srow * ld * 2;
                    dma16(rk, Kd + t * 512, so + ((pos ^ ((lrow >> 1) & 7)) << 4), 0);
                    dma16(rv, Vd + t * 512, so + ((pos ^ (4 * ((lrow >> 1) & 1))) << 4), 0);
                }
            }
        }
    };

    // ---- this wave's query blocks: wave, wave + 8, wave + 16 ------------------------------------------------------
    const int nblk = nkt;  // 32-row query blocks = 32-key tiles
    bf16x8 qf[MAXB][4];
    f32x16 o[MAXB][2];
    float m_run[MAXB], l_run[MAXB];
    auto load_q = [&](int item) __attribute__((always_inline)) {
        const bf16_t *base = item_base(item);
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (wave + ST_WAVES * b < nblk) {
                int qrow = (wave + ST_WAVES * b) * 32 + r;
                qrow = qrow < tokens ? qrow : tokens - 1;
                const bf16_t *qsrc = base + (size_t)qrow * ld + h * 8;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[b][ks] = *reinterpret_cast<const bf16x8 *>(qsrc + ks * 16);
            }
        }
    };

    // One step = one chunk of one item, reading ring slot `slot` while the DMA of the following step fills the other one.
    const int sw = (r >> 1) & 7, h4 = 4 * h;
    auto step = [&](int slot, int item, int ch, int next_item, int next_ch) __attribute__((always_inline)) {
        asm volatile("" : "+s"(tk));
        if (next_item >= 0) dma_chunk(next_item, next_ch, slot ^ 1);
        const bf16_t *Ks = lds + slot * SBUF, *Vs = Ks + SKEYS * SHD;
        const int key_base = chunk_first(ch) * 32;
        int valid = chunk_tiles(ch) * 32;                      // keys of this chunk ...
        valid = tk - key_base < valid ? tk - key_base : valid;  // ... that exist
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (wave + ST_WAVES * b >= nblk) continue;  // wave-uniform
            // The chunk is taken in sub-chunks of SUB key tiles (scores of 64 keys in registers at a time: three blocks of
            // O^T accumulators plus their Q fragments leave room for no more); each runs one online-softmax update.
            // A sub-chunk is always computed whole: keys past `valid` (a chunk of 3 tiles, the end of the sequence) hold
            // older, finite data in LDS and are masked to -inf, i.e. weigh exactly 0.
#pragma unroll
            for (int k0 = 0; k0 < SKT; k0 += SUB) {
                if (k0 * 32 >= valid) continue;  // wave-uniform
                f32x16 st[SUB];
#pragma unroll
                for (int u = 0; u < SUB; ++u) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) st[u][v] = 0.0f;
                    const bf16_t *krow = Ks + ((k0 + u) * 32 + r) * SHD;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(krow + (((2 * ks + h) ^ sw) & 7) * 8);
                        st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[b][ks], st[u], 0, 0, 0);
                    }
                }
                if ((k0 + SUB) * 32 > valid) {  // wave-uniform: the sub-chunk reaches past the valid keys
#pragma unroll
                    for (int u = 0; u < SUB; ++u)
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            const int kloc = (k0 + u) * 32 + (v & 3) + 8 * (v >> 2) + h4;
                            st[u][v] = kloc < valid ? st[u][v] : -INFINITY;
                        }
                }
                // ---- online softmax: running max m, running sum l, O rescaled by 2^((m_old - m_new) c) when m moved
                float cmax = -INFINITY;
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) cmax = fmaxf(cmax, st[u][v]);
                cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
                const float m_old = m_run[b];
                const float m_new = fmaxf(m_old, cmax);  // finite: the sub-chunk has at least one valid key
                m_run[b] = m_new;
                const float mxs = -m_new * kScaleS;
                float csum = 0.0f;
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(st[u][v], kScaleS, mxs));  // exp2(-inf) = 0: masked keys
                        st[u][v] = e;
                        csum += e;
                    }
                if (__any(m_new != m_old)) {  // wave-uniform: most sub-chunks leave every row's maximum where it was
                    const float alpha = __builtin_amdgcn_exp2f((m_old - m_new) * kScaleS);  // first one: exp2(-inf) = 0
                    l_run[b] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int v = 0; v < 16; ++v) o[b][dt][v] *= alpha;
                }
                l_run[b] += csum;
                // ---- O^T += V^T . P^T
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 pf;
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)st[u][8 * s2 + j];
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
                            o[b][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_frag_tr(Vs, (k0 + u) * 32 + 16 * s2, dt, lane), pf, o[b][dt], 0, 0, 0);
                    }
            }
        }
    };
    auto finish_item = [&](int item) __attribute__((always_inline)) {  // normalise and store this wave's blocks
        const int img = item / heads, head = item % heads;
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            const int row = (wave + ST_WAVES * b) * 32 + r;
            const float inv = 1.0f / (l_run[b] + __shfl_xor(l_run[b], 32));  // all lanes take part in the exchange
            if (wave + ST_WAVES * b < nblk && row < tokens) {
                bf16_t *dst = out + ((size_t)img * tokens + row) * D + head * SHD + 4 * h;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 w;
#pragma unroll
                        for (int q = 0; q < 4; ++q) w[q] = (__bf16)(o[b][dt][4 * g + q] * inv);
                        *reinterpret_cast<bf16x4 *>(dst + dt * 32 + 8 * g) = w;
                    }
            }
        }
    };

    int item = blockIdx.x;
    if (item >= n_items) return;  // workgroup-uniform
    const int stride = gridDim.x;
    // the ring starts zeroed: stale rows that a partial sub-chunk multiplies by 0 must be finite from the first step on
    for (int i = tid; i < 2 * SBUF / 8; i += ST_THREADS) reinterpret_cast<uint4 *>(lds)[i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    dma_chunk(item, 0, 0);
    int slot = 0;

    for (;;) {  // items
        load_q(item);
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            m_run[b] = -INFINITY;
            l_run[b] = 0.0f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int v = 0; v < 16; ++v) o[b][dt][v] = 0.0f;
        }
        // Q (ordinary loads) and the chunk that is about to be read (LDS-DMA) are both complete after this barrier
        ring_barrier();
        const int next_item = item + stride;
        for (int ch = 0; ch < nch; ++ch) {
            // the step after this one: the next chunk of this head, or the first chunk of the next head
            const int ni = ch + 1 < nch ? item : (next_item < n_items ? next_item : -1);
            const int nc = ch + 1 < nch ? ch + 1 : 0;
            step(slot, item, ch, ni, nc);
            if (ch == nch - 1) finish_item(item);
            slot ^= 1;
            if (ch + 1 < nch) ring_barrier();  // the next chunk has landed, everybody is done with this one
            // (after the last chunk the barrier is the one at the top of the next item, behind its Q loads)
        }
        if (next_item >= n_items) break;
        item = next_item;
    }
}

// 224 < tokens <= 768.  Returns a hipError_t value.
int attention_bf16_stream(hipStream_t s, const unsigned short *qkv, unsigned short *out, int n_images, int tokens, int heads) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return static_cast<int>(hipErrorInvalidDevice);
        cus = v;
    }
    if (tokens > ST_WAVES * MAXB * 32) return static_cast<int>(hipErrorInvalidValue);
    const int items = n_images * heads;
    const int grid = items < cus ? items : cus;
    hipLaunchKernelGGL(attention_bf16_stream_kernel, dim3(grid), dim3(ST_THREADS), 0, s, qkv, out, tokens, heads, items);
    return static_cast<int>(hipGetLastError());
}

}  // namespace vitattn
