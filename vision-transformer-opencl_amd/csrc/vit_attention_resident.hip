// csrc/vit_attention_resident.hip -- fp32 scaled-dot-product attention, K/V of one head resident in LDS (tokens <= 224).
//
// Reference: ViT_seq.c:156-215 (scores = q.k / sqrtf(head_dim); row softmax with max subtraction; head_out = P.V).
// This is the kernel behind vithip_attention_f32 for ViT-B/16 (197 tokens); the first version (attention_f32_kernel in
// vit_attention.hip, still used for the bf16-I/O cross-check variant) ran at 50 % of the fp32 matrix roofline.  What
// the stamps of that kernel showed (tools/attn_probe.py, 73k cycles per (image, head) on a CU):
//   * 197 tokens = 6 x 32 + 5: with 32x32 MFMA tiles a wave owns 32 query rows, 7 waves on 4 SIMDs make two SIMDs do
//     two full blocks (one of them for 5 valid rows) and both products run over 224 padded keys;
//   * staging the head's 100 KB of K/V (9.3k cycles) and the output stores (2k) had nothing to overlap with: the LDS
//     holds one head, so one workgroup per CU, and it waited for its own loads.
// This version:
//   * v_mfma_f32_16x16x4_f32 (exact fp32, same rate as 32x32x2): 16-row query blocks and 16-key tiles, so 197 tokens
//     pad to 208 keys and the work splits into 13 half-size blocks that are dealt to the 8 waves so that every SIMD gets
//     3.25 blocks: waves 0-3 take two blocks, waves 4-7 one, and the odd 13th block (the 5 tail rows) is cut in four by
//     KEYS between waves 4-7, whose partial softmax results (running max, sum, O) meet in LDS and are merged by
//     wave 4 in a fixed order (deterministic, no atomics);
//   * persistent workgroups (one per CU) walk the (image, head) items and stage by LDS-DMA (buffer_load ... lds: no
//     registers, zero-fill past the last token by the buffer bounds) in the OTHER phase's shadow:
//         [ S = K.Q^T + softmax of item i   ||  DMA V(i)   ]  barrier
//         [ O = V^T.P^T of item i, stores   ||  DMA K(i+1), Q(i+1) -> registers ]  barrier
//     K is dead once every wave has its scores, V once every wave has its output, so one LDS image of each suffices.
//
// MFMA mapping (lane = 16 g + n):
//   S^T tile (16 keys x 16 queries) = K_tile . Q^T: A = K[key 16kt+n][d], B = Q[q0+n][d] with d = 16c + 4g + j for MFMA
//       (c, j) -- the same permutation of d on both operands, chosen so that a lane reads its four j as one 16-byte word;
//       accumulator register i of lane (n, g) = score(query q0+n, key 16kt + 4g + i): a softmax row is lane-local plus two
//       lane exchanges (xor 16, xor 32).
//   O^T (64 d x 16 queries) += V^T . P^T, 4 keys per MFMA: register i of the S^T tile IS the B operand (k = g selects
//       key 16kt + 4g + i), A = V[that key][4n + j] for output tile j -- one 16-byte LDS read per 4 MFMAs -- and a lane
//       ends up with O[q0+n][16g .. 16g+15]: four 16-byte stores.
//   K rows are 256 B in LDS with their 16-byte chunks XOR-swizzled by (row & 15) on the SOURCE address of the DMA (the LDS
//   side of an LDS-DMA is linear), which makes the A-fragment reads conflict-free; V is read a whole row per 16 lanes.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vit_device.hpp"
#include "vit_hip_kernels.h"

namespace vitattn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64;            // head_dim
constexpr int ROWB = HD * 4;      // bytes of one K or V row in LDS
constexpr int RES_WAVES = 8;
constexpr int RES_THREADS = RES_WAVES * 64;
constexpr int PART_LD = 68;       // floats per query row of a tail partial: 64 x O, max, sum, pad
constexpr float kScale = 0.125f * 1.4426950408889634f;  // (1 / sqrtf(64)) * log2(e): p = exp2(s*kScale - max*kScale)

// NKT 16-key tiles: tokens <= 16 * NKT.  PARTS: the small-batch instantiation that spreads a head over parts_arg workgroups
// (a run-time `parts` costs the metric configuration 1 % in registers and spills: it gets its own code)
template <int NKT, bool PARTS = false>
__global__ __launch_bounds__(RES_THREADS) void attention_f32_resident_kernel(const float *__restrict__ qkv,
                                                                             float *__restrict__ out, int tokens, int heads,
                                                                             int n_items, int q_rows, int parts_arg
#ifdef VIT_PROBES
                                                                             , unsigned long long *__restrict__ dbg, int mode
#endif
) {
#ifndef VIT_PROBES
    constexpr unsigned long long *dbg = nullptr;  // instrumentation exists in the probe build only
    constexpr int mode = 0;
#endif
    // mode (probe build only, else 0): timing experiments with WRONG results: 1 = no softmax arithmetic, 2 = no LDS fragment
    // reads, 4 = waves 4-7 idle, 8 = no LDS-DMA after the first item
    // dbg != nullptr (probe build, tools/attn_probe.py): per wave 8 cycle stamps of the workgroup's SECOND item
    const int parts = PARTS ? parts_arg : 1;
    constexpr int KEYS = NKT * 16;
    // a tail block is split by KEYS between four waves: key tiles [QB0, QB1), [QB1, QB2), [QB2, QB3), [QB3, NKT)
    constexpr int QB1 = (NKT + 3) / 4, QB2 = (2 * NKT + 3) / 4, QB3 = (3 * NKT + 3) / 4;
    __shared__ __attribute__((aligned(1024))) char lds[2 * KEYS * ROWB + 4 * 16 * PART_LD * 4];
    char *const Ks = lds;
    char *const Vs = lds + KEYS * ROWB;
    float *const part = reinterpret_cast<float *>(lds + 2 * KEYS * ROWB);

    // `tk` = tokens, made opaque once per item (asm below): hipcc otherwise hoists every tokens-dependent mask, predicate
    // and tail-row offset of the unrolled job bodies out of the item loop -- ~60 VGPRs and 300 spilled SGPRs of invariants
    int tk = tokens;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int D = heads * HD, ld = 3 * D;

    // ---- which query blocks this wave owns (wave-uniform) -------------------------------------------------------
    const int nblk = (q_rows + 15) >> 4;
    const bool split = nblk > RES_WAVES && (nblk & 1);  // odd block count: the last block is cut in four by keys
    const int nfull = split ? nblk - 1 : nblk;
    // `parts` > 1 (few items: small batches): a work id is (item, part) and a workgroup takes only the part's share of the full
    // blocks -- per = ceil(nfull / parts) <= 8 of them, one per wave -- so that one head's query blocks spread over `parts` CUs
    // instead of leaving most of the chip idle; every part stages the head's whole K and V.  The last part also owns the tail
    // block, cut in four over its waves 4-7 exactly as below: a row's arithmetic does not depend on the split.
    const int per = (nfull + parts - 1) / parts;
    const bool hasB = parts == 1 && wave + RES_WAVES < nfull;  // two full blocks: only in the one-part walk
    // Four one-block waves on four different SIMDs (SIMD = wave & 3) share the tail block, on the least loaded SIMDs: with 9 or 13
    // blocks every SIMD carries the same number of full blocks (waves 4-7 take the quarters: 2.25 / 3.25 blocks per SIMD); with 11,
    // waves 0 and 1 carry two blocks, so the quarters go to SIMDs 2 and 3 (waves 2, 3, 6, 7: 2.5 against 3).  Quarter 0 merges.
    struct Own { bool hasA, has2; int tq, blkA, blkB; };
    auto own = [&](int w) __attribute__((always_inline)) {  // what this wave computes of work id w (wave-uniform)
        Own o;
        if (parts == 1) {
            o.hasA = wave < nfull && !((mode & 4) && wave >= 4);
            o.tq = !split ? -1
                 : nblk == 11 ? (wave == 2 ? 0 : wave == 3 ? 1 : wave == 6 ? 2 : wave == 7 ? 3 : -1)
                              : (wave >= 4 ? wave - 4 : -1);
            o.blkA = wave;
            o.blkB = o.tq >= 0 ? nblk - 1 : wave + RES_WAVES;
        } else {
            const int part = w % parts, b0 = part * per;
            const int nloc = nfull - b0 < per ? nfull - b0 : per;  // may be <= 0 for a trailing part
            o.hasA = wave < nloc;
            o.tq = (split && part == parts - 1 && wave >= 4) ? wave - 4 : -1;
            o.blkA = b0 + wave;
            o.blkB = nblk - 1;
        }
        o.has2 = hasB || o.tq >= 0;
        return o;
    };
    const Own own0 = own(blockIdx.x);
    const int blkA = own0.blkA, blkB = own0.blkB;  // the two-block role (parts == 1): constant over the walk

    // ---- LDS-DMA of one head's K or V: 4 rows (1 KB) per wave instruction, rows past the last token read as zero ---
    const int row_in = lane >> 4, cpos = lane & 15;
    auto dma = [&](const float *head_base, char *dst, bool swizzle) __attribute__((always_inline)) {
        // bytes addressable through the descriptor: everything up to the end of the last token's 64 floats
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(head_base), 0, (tokens - 1) * ld * 4 + ROWB, 0x00020000);
#pragma unroll
        for (int k = 0; k < (NKT * 4 + RES_WAVES - 1) / RES_WAVES; ++k) {
            const int t = wave + RES_WAVES * k;  // 4-row group
            if (t < NKT * 4) {
                const int row = 4 * t + row_in;
                const int chunk = swizzle ? (cpos ^ (row & 15)) : cpos;
                // the whole offset sits in the per-lane operand: that is the one the descriptor's bounds check sees
                const int voff = row * ld * 4 + chunk * 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(dst + t * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    auto item_base = [&](int w) -> const float * {  // w = work id = item * parts + part
        const int item = w / parts;
        const int img = item / heads, head = item - img * heads;
        return qkv + (size_t)img * tokens * ld + head * HD;
    };
    // Q rows through a buffer descriptor on the head's (scalar) base and ONE 32-bit lane offset, recomputed from the lane id at every
    // call: as 64-bit per-lane pointers the two blocks' row addresses were loop-invariant register pairs that the allocator spilled in
    // the two-block role, and the reload's vmcnt(0) sat between the two blocks' loads -- an exposed round trip per item (round 5).
    auto load_q = [&](const float *base, int blk, f32x4 (&qf)[4]) __attribute__((always_inline)) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        int qrow = blk * 16 + (lane_l & 15);
        qrow = qrow < tokens ? qrow : tokens - 1;  // rows past the end: clamped address, never stored
        const int voff = (qrow * ld + 4 * (lane_l >> 4)) * 4;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, voff + 64 * c, 0, 0));
    };

    // per-lane LDS offsets
    int koff[4];  // K fragment chunk c of this lane's row: byte offset inside the row, swizzled
#pragma unroll
    for (int c = 0; c < 4; ++c) koff[c] = ((4 * c + g) ^ n) * 16;
    const char *const kbase = Ks + n * ROWB;
    const char *const vbase = Vs + (4 * g) * ROWB + n * 16;

    // ---- the sub-steps of a job = (NB query blocks, key tiles [KT0, KT1)) ---------------------------------------------
    // A wave that owns two blocks runs them TOGETHER: every K / V fragment read from LDS feeds the MFMAs of both blocks
    // (half the LDS reads per MFMA), and the two blocks are two independent accumulation chains -- v_mfma_f32_16x16x4_f32
    // issues every 32 cycles but a DEPENDENT one only every 40 (MI355X_MICROARCH.md).  A single block takes its key tiles
    // in pairs for the same reason.  Fragments are fetched one 16-deep chunk ahead of the MFMAs that use them.
    auto s_mm = [&](auto nb_c, auto b0_c, auto kt0_c, auto kt1_c, const f32x4 (&qf)[2][4], f32x4 (&st)[2][NKT]) __attribute__((always_inline)) {
        constexpr int B0 = decltype(b0_c)::value;  // first register set used (a single block may live in set 1)
        constexpr int NB = decltype(nb_c)::value, KT0 = decltype(kt0_c)::value, KT1 = decltype(kt1_c)::value;
        constexpr int TP = NB == 1 ? 2 : 1;                    // key tiles per step
        constexpr int NS = (KT1 - KT0 + TP - 1) / TP;          // steps (the last may hold a single tile)
        f32x4 kf[2][TP] = {};
        auto read_kc = [&](int sidx, int c, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                const int kt = KT0 + sidx * TP + u;
                if (kt < KT1 && !(mode & 2)) kf[set][u] = *reinterpret_cast<const f32x4 *>(kbase + kt * (16 * ROWB) + koff[c]);
            }
        };
        read_kc(0, 0, 0);
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
#pragma unroll
            for (int u = 0; u < TP; ++u)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (KT0 + sidx * TP + u < KT1) st[B0 + b][KT0 + sidx * TP + u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int step = 4 * sidx + c;  // running chunk index: buffer = step & 1
                if (c < 3) read_kc(sidx, c + 1, (step + 1) & 1);
                else if (sidx + 1 < NS) read_kc(sidx + 1, 0, (step + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int u = 0; u < TP; ++u)
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const int kt = KT0 + sidx * TP + u;
                            if (kt < KT1)
                                st[B0 + b][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[step & 1][u][j], qf[B0 + b][c][j], st[B0 + b][kt], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // row softmax of one block over the job's keys (unnormalised: p = 2^((s - max) c); max and sum are returned).
    // fp32 VALU work is paid for in matrix-pipe time on gfx950, so it is kept to max, one fma + v_exp_f32, add per score;
    // keys >= tokens exist only in the last tile(s).
    auto softmax = [&](auto kt0_c, auto kt1_c, f32x4 (&st)[NKT], float &m_out, float &l_out) __attribute__((always_inline)) {
        constexpr int KT0 = decltype(kt0_c)::value, KT1 = decltype(kt1_c)::value;
        if (mode & 1) { m_out = 0.f; l_out = 1.f; return; }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = KT0; kt < KT1; ++kt) {
            if (16 * kt + 16 > tk) {  // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) st[kt][i] = (16 * kt + 4 * g + i) < tk ? st[kt][i] : -INFINITY;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, st[kt][i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mxs = -mx * kScale;
        float sum = 0.0f;
#pragma unroll
        for (int kt = KT0; kt < KT1; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = __builtin_amdgcn_exp2f(fmaf(st[kt][i], kScale, mxs));  // exp2(-inf) = 0: masked keys
                st[kt][i] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        m_out = mx;
        l_out = sum;
    };
    auto pv_mm = [&](auto nb_c, auto b0_c, auto kt0_c, auto kt1_c, const f32x4 (&st)[2][NKT], f32x4 (&o)[2][4]) __attribute__((always_inline)) {
        constexpr int B0 = decltype(b0_c)::value;
        constexpr int NB = decltype(nb_c)::value, KT0 = decltype(kt0_c)::value, KT1 = decltype(kt1_c)::value;
        constexpr int NG = (KT1 - KT0) * 4;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[B0 + b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // one 16-byte read feeds 4 NB MFMAs: reads run VA groups ahead so that their latency is covered
        constexpr int VA = NB == 1 ? 3 : 2;
        f32x4 vf[VA + 1] = {};
        auto read_v = [&](int gi) __attribute__((always_inline)) {  // gi = 4 (kt - KT0) + i
            if (!(mode & 2)) vf[gi % (VA + 1)] = *reinterpret_cast<const f32x4 *>(vbase + ((KT0 + (gi >> 2)) * 16 + (gi & 3)) * ROWB);
        };
#pragma unroll
        for (int gi = 0; gi < VA && gi < NG; ++gi) read_v(gi);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi + VA < NG) read_v(gi + VA);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    o[B0 + b][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[gi % (VA + 1)][j], st[B0 + b][KT0 + (gi >> 2)][gi & 3], o[B0 + b][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Output rows: buffer stores on the image's (scalar) base with one 32-bit lane offset, recomputed per call (see load_q: the
    // 64-bit row pointer was the last spilled value of the two-block role, reloaded behind a vmcnt(0) right after the next item's Q
    // loads had been issued).  Everything that varies goes into the LANE offset and the immediate: for a 16-byte buffer store with
    // a register in soffset hipcc pads no wait state in front of a write of the data registers (tools/check_inline_asm.py rule iv).
    auto store_rows = [&](int w, int blk, const f32x4 (&o)[4], float scale) __attribute__((always_inline)) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int row = blk * 16 + (lane_l & 15);
        if (row < q_rows) {
            const int item = w / parts;
            const int img = item / heads, head = item - img * heads;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)img * tokens * D, 0, tokens * D * 4, 0x00020000);
            const int voff = (row * D + head * HD + 16 * (lane_l >> 4)) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 w;
                w[0] = o[0][i] * scale; w[1] = o[1][i] * scale; w[2] = o[2][i] * scale; w[3] = o[3][i] * scale;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), ro, voff + 16 * i, 0, 0);
            }
        }
    };
    auto write_partial = [&](int which, const f32x4 (&o)[4], float m, float l) __attribute__((always_inline)) {
        float *p = part + (which * 16 + n) * PART_LD;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 w;
            w[0] = o[0][i]; w[1] = o[1][i]; w[2] = o[2][i]; w[3] = o[3][i];
            *reinterpret_cast<f32x4 *>(p + 16 * g + 4 * i) = w;
        }
        if (g == 0) { p[64] = m; p[65] = l; }
    };
    // the four quarters of a tail block, merged in a fixed order: O = sum_k O_k a_k / sum_k l_k a_k, a_k = 2^((m_k - M) c)
    auto merge_tail = [&](int w) __attribute__((always_inline)) {
        const int item = w / parts;
        const float *pk[4];
        float a[4], M = -INFINITY, den = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pk[k] = part + (k * 16 + n) * PART_LD;
            M = fmaxf(M, pk[k][64]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = __builtin_amdgcn_exp2f((pk[k][64] - M) * kScale);
            den += pk[k][65] * a[k];
        }
        const float inv = 1.0f / den;
        const int row = (nblk - 1) * 16 + n;
        if (row < q_rows) {
            const int img = item / heads, head = item - img * heads;
            float *dst = out + ((size_t)img * tokens + row) * D + head * HD + 16 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) w += *reinterpret_cast<const f32x4 *>(pk[k] + 16 * g + 4 * i) * a[k];
                *reinterpret_cast<f32x4 *>(dst + 4 * i) = w * inv;
            }
        }
    };

    using std::integral_constant;
    constexpr integral_constant<int, 0> c0{};
    constexpr integral_constant<int, QB1> q1{};
    constexpr integral_constant<int, QB2> q2{};
    constexpr integral_constant<int, QB3> q3{};
    constexpr integral_constant<int, NKT> cn{};

    // ---- persistent walk over the (image, head) items ----------------------------------------------------------------
    int item = blockIdx.x;  // work id (= item when parts == 1)
    const int step = gridDim.x;
    const int n_work = n_items * parts;
    if (item >= n_work) return;  // workgroup-uniform
    // LDS-DMA is issued by the one-block waves among 4-7 -- all four up to 13 blocks, waves 6 and 7 with 14 (waves 4, 5 then
    // carry two blocks and run the other role): an issue costs 60-185 cycles of the issuing wave (MI355X_MICROARCH.md), and
    // the two-block waves are on the phase's critical path.
    const bool dma4 = nfull <= 12;                 // workgroup-uniform
    const bool dma_wave = wave >= (dma4 ? RES_WAVES - 4 : RES_WAVES - 2);
    // A DMA wave takes the 4-row groups t = ND k + w0.  Per-lane source offsets of a group are (t & 3)-periodic apart from the
    // group's row base, which goes into the instruction's scalar offset: one VGPR for K with four DMA waves (t & 3 = w0), two
    // with two (t & 3 = w0, 2 + w0), one for V.  Rows past the last token are not zero-filled here but CLAMPED to the last
    // token (its values are finite, the keys are masked / weigh 0): no bounds check involved, and only the last groups pay
    // per-lane arithmetic.
    // The three lane offsets are recomputed from the lane id at every call (half a dozen VALU instructions): kept across the
    // item loop they were what the allocator spilled (256 VGPRs, 5-7 spilled in the metric instantiation), and a spill reload
    // is a vector load the compiler waits for with vmcnt(0) -- here in the middle of the DMA issue, i.e. the issuing wave sat out
    // the round trip of the pieces it had just issued, two or three times per phase (round 5).
    const int w0 = dma4 ? (wave & 3) : (wave & 1);
    auto dma_nd = [&](auto nd_c, const float *head_base, char *dst, bool swizzle) __attribute__((always_inline)) {
        constexpr int ND = decltype(nd_c)::value;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));  // opaque: nothing below is loop-invariant to the compiler
        const int row_in = lane_l >> 4, cpos = lane_l & 15;
        const int lane_row = row_in * ld * 4;
        const int koffA = lane_row + ((cpos ^ (w0 << 2 | row_in)) << 4);
        const int koffB = lane_row + ((cpos ^ ((2 + w0) << 2 | row_in)) << 4);
        const int voffV = lane_row + (cpos << 4);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(head_base), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int k = 0; k < (NKT * 4 + ND - 1) / ND; ++k) {
            const int t = ND * k + w0;  // 4-row group
            if (t < NKT * 4) {
                // tokens > 16 (NKT - 1): only the groups of the LAST key tile can reach past the last token
                if (ND * k + ND - 1 < 4 * (NKT - 1) || 4 * t + 3 < tk) {  // first part known at compile time; wave-uniform
                    const int voff = swizzle ? ((ND == 2 && (k & 1)) ? koffB : koffA) : voffV;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(dst + t * 1024), 16, voff, t * 16 * ld, 0, 0);
                } else {
                    const int lrow = 4 * t + row_in;
                    const int srow = lrow < tk ? lrow : tk - 1;
                    const int voff = srow * ld * 4 + ((swizzle ? (cpos ^ (lrow & 15)) : cpos) << 4);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(dst + t * 1024), 16, voff, 0, 0, 0);
                }
            }
        }
    };
    auto dma2 = [&](const float *head_base, char *dst, bool swizzle) __attribute__((always_inline)) {
        if (dma4) dma_nd(integral_constant<int, 4>{}, head_base, dst, swizzle);
        else dma_nd(integral_constant<int, 2>{}, head_base, dst, swizzle);
    };

    constexpr integral_constant<int, 1> one{};
    constexpr integral_constant<int, 2> two{};
#ifdef VIT_PROBES
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RES_STAMP(k) do { if (dbg && iter == 1) ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RES_STAMP(k) do { } while (0)
#endif

    // The item loop exists twice, once per wave role, so that each role gets its own register allocation: a two-block
    // wave is at the VGPR limit (104 score + 32 output registers) and does nothing but matrix work; the one-block waves
    // have room for the tail block, the merge and the LDS-DMA.  Both loops execute the same barriers in the same order
    // (s_barrier counts arrivals; which copy of the code a wave arrives from does not matter).
    // The SIMD issues from its oldest ready wave first: waves 0-3 would run both their blocks at full speed while their
    // SIMD partners (waves 4-7, each a single dependent-heavier stream) crawl, and then finish alone.  The partners go first.
    if (!hasB) __builtin_amdgcn_s_setprio(2);
    if (hasB) {
        // ============================ role: two full blocks, fused =====================================================
        f32x4 qf[2][4];
        {
            const float *base = item_base(item);
            load_q(base, blkA, qf[0]);
            load_q(base, blkB, qf[1]);
            dma(base + D, Ks, true);  // the first K: by everybody, nothing to hide behind yet
        }
        __syncthreads();
        int iter = 0;
        for (;;) {
            asm volatile("" : "+s"(tk));
            RES_STAMP(0);
            f32x4 st[2][NKT];
            float mA, lA, mB, lB;
            s_mm(two, c0, c0, cn, qf, st);
            softmax(c0, cn, st[0], mA, lA);
            RES_STAMP(1);
            softmax(c0, cn, st[1], mB, lB);
            RES_STAMP(2);
            __syncthreads();  // every wave has its scores (K is dead) and V has landed
            RES_STAMP(3);
            const int next = item + step;
            f32x4 o[2][4];
            pv_mm(two, c0, c0, cn, st, o);
            if (next < n_work) {  // Q of the next item (this role is at the register limit until here)
                const float *nb = item_base(next);
                load_q(nb, blkA, qf[0]);
                load_q(nb, blkB, qf[1]);
            }
            store_rows(item, blkA, o[0], 1.0f / lA);
            RES_STAMP(4);
            store_rows(item, blkB, o[1], 1.0f / lB);
            RES_STAMP(5);
            __syncthreads();  // V is dead, K/Q of the next item have landed
            RES_STAMP(6);
            ++iter;
            if (next >= n_work) break;
            item = next;
        }
    } else {
        // ============================ role: one block (or none), tail halves, LDS-DMA =================================
        f32x4 qf[2][4];
        Own cur = own0;  // this wave's share of the current work id; with parts > 1 it changes from one to the next
        {
            const float *base = item_base(item);
            if (cur.hasA) load_q(base, cur.blkA, qf[0]);
            if (cur.has2) load_q(base, cur.blkB, qf[1]);
            dma(base + D, Ks, true);
        }
        __syncthreads();
        int tail_item = -1;  // the wave with quarter 0: an item whose tail partials wait in LDS
        int iter = 0;
        for (;;) {
            asm volatile("" : "+s"(tk));
            RES_STAMP(0);
            // ---- phase 1: scores + softmax (reads K); V(item) lands meanwhile
            const float *base = item_base(item);
            if (dma_wave && (!(mode & 8) || iter == 0)) dma2(base + 2 * D, Vs, false);
            if (tail_item >= 0) {  // only the wave with quarter 0
                merge_tail(tail_item);
                tail_item = -1;
            }
            f32x4 st[2][NKT];
            float mA = 0.f, lA = 1.f, mB = 0.f, lB = 1.f;
            const bool hasA = cur.hasA;
            const int tq = cur.tq;
            if (hasA) {
                s_mm(one, c0, c0, cn, qf, st);
                softmax(c0, cn, st[0], mA, lA);
            }
            RES_STAMP(1);
            // this wave's quarter of the tail block's keys (wave-uniform)
            if (tq == 0) { s_mm(one, one, c0, q1, qf, st); softmax(c0, q1, st[1], mB, lB); }
            else if (tq == 1) { s_mm(one, one, q1, q2, qf, st); softmax(q1, q2, st[1], mB, lB); }
            else if (tq == 2) { s_mm(one, one, q2, q3, qf, st); softmax(q2, q3, st[1], mB, lB); }
            else if (tq == 3) { s_mm(one, one, q3, cn, qf, st); softmax(q3, cn, st[1], mB, lB); }
            RES_STAMP(2);
            __syncthreads();
            RES_STAMP(3);
            // ---- phase 2: O = V^T . P^T (reads V); K and Q of the next item land meanwhile
            const int next = item + step;
            Own nxt = cur;
            if (next < n_work) {  // workgroup-uniform
                const float *nb = item_base(next);
                if (dma_wave && !(mode & 8)) dma2(nb + D, Ks, true);
                nxt = own(next);
                if (nxt.hasA) load_q(nb, nxt.blkA, qf[0]);  // consumed after the barrier below
                if (nxt.has2) load_q(nb, nxt.blkB, qf[1]);
            }
            f32x4 o[2][4];
            if (hasA) {
                pv_mm(one, c0, c0, cn, st, o);
                store_rows(item, cur.blkA, o[0], 1.0f / lA);
            }
            RES_STAMP(4);
            if (tq == 0) {
                pv_mm(one, one, c0, q1, st, o);
                write_partial(0, o[1], mB, lB);
                tail_item = item;  // this wave merges the four partials after the next barrier
            } else if (tq == 1) {
                pv_mm(one, one, q1, q2, st, o);
                write_partial(1, o[1], mB, lB);
            } else if (tq == 2) {
                pv_mm(one, one, q2, q3, st, o);
                write_partial(2, o[1], mB, lB);
            } else if (tq == 3) {
                pv_mm(one, one, q3, cn, st, o);
                write_partial(3, o[1], mB, lB);
            }
            RES_STAMP(5);
            __syncthreads();  // V is dead, the partials are visible, K/Q of the next item have landed
            RES_STAMP(6);
            ++iter;
            if (next >= n_work) break;
            item = next;
            cur = nxt;
        }
        if (tail_item >= 0) merge_tail(tail_item);
    }
#undef RES_STAMP
#ifdef VIT_PROBES
    if (dbg && lane == 0) {
        unsigned long long *d = dbg + ((size_t)blockIdx.x * RES_WAVES + wave) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = ts[k];
    }
#endif
}

#ifdef VIT_PROBES
unsigned long long *g_res_dbg = nullptr;  // vithip_attention_set_debug_buffer (probe build): 8 stamps per wave
int g_res_mode = 0;                       // vithip_attention_set_probe_mode
#endif

template <int NKT>
int launch_resident(hipStream_t s, const float *qkv, float *out, int n_images, int tokens, int heads, int q_rows, int cus) {
    const int items = n_images * heads;
    // Few items (small batches): cut every head's query blocks into `parts` shares for as many workgroups, while that still
    // leaves a share at least two full blocks (one image: 12 heads x 4 parts instead of 12 workgroups on 256 CUs).
    const int nblk = (q_rows + 15) / 16, nfull = (nblk > RES_WAVES && (nblk & 1)) ? nblk - 1 : nblk;
    int parts = 1;
    while (parts < 4 && items * (parts + 1) <= cus && (nfull + parts) / (parts + 1) >= 2) ++parts;
    const int work = items * parts;
    const int grid = work < cus ? work : cus;  // one workgroup per CU: the LDS holds one head
#ifdef VIT_PROBES
    if (parts > 1)
        hipLaunchKernelGGL((attention_f32_resident_kernel<NKT, true>), dim3(grid), dim3(RES_THREADS), 0, s, qkv, out, tokens, heads, items,
                           q_rows, parts, g_res_dbg, g_res_mode);
    else
        hipLaunchKernelGGL((attention_f32_resident_kernel<NKT, false>), dim3(grid), dim3(RES_THREADS), 0, s, qkv, out, tokens, heads, items,
                           q_rows, 1, g_res_dbg, g_res_mode);
#else
    if (parts > 1)
        hipLaunchKernelGGL((attention_f32_resident_kernel<NKT, true>), dim3(grid), dim3(RES_THREADS), 0, s, qkv, out, tokens, heads, items,
                           q_rows, parts);
    else
        hipLaunchKernelGGL((attention_f32_resident_kernel<NKT, false>), dim3(grid), dim3(RES_THREADS), 0, s, qkv, out, tokens, heads, items,
                           q_rows, 1);
#endif
    return static_cast<int>(hipGetLastError());
}

// tokens <= 224, q_rows <= tokens.  Returns a hipError_t value.
int attention_f32_resident(hipStream_t s, const float *qkv, float *out, int n_images, int tokens, int heads, int q_rows) {
    const int cus = vitdev::current_cus();
    if (cus <= 0) return static_cast<int>(hipErrorInvalidDevice);
    switch ((tokens + 15) / 16) {
        case 1: return launch_resident<1>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 2: return launch_resident<2>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 3: return launch_resident<3>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 4: return launch_resident<4>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 5: return launch_resident<5>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 6: return launch_resident<6>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 7: return launch_resident<7>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 8: return launch_resident<8>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 9: return launch_resident<9>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 10: return launch_resident<10>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 11: return launch_resident<11>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 12: return launch_resident<12>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 13: return launch_resident<13>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        case 14: return launch_resident<14>(s, qkv, out, n_images, tokens, heads, q_rows, cus);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}

}  // namespace vitattn
