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
//
// Round 5: the head switch no longer stops the workgroup.  (i) A wave normalises and stores a query block, re-initialises its
// state and refills its Q slot for the NEXT head right after the block's last unit inside the head's last step -- beside the other
// waves' units instead of as one burst behind a barrier.  (ii) Nothing waits for those refills with vmcnt(0) any more: every
// vector-memory operation of the kernel is either inline-asm LDS-DMA or a bounds-checked buffer store with a scalar base, their
// order per wave is fixed, so the waits are COUNTED (vmcnt retires in order): the barrier at the top of a head waits for the
// head's first K/V chunk and the wave's first Q block only, and the first scores of the wave's second / third block wait for
// that block's Q.  (iii) The 14 spilled VGPRs are gone (store addresses are one 32-bit lane offset + a scalar base, the Q / K
// fragment addresses one lane register and an XOR per k-step): a spill reload is a vector load the compiler waits for with
// vmcnt(0) -- which drained the ring's DMA inside the step and serialised the output stores behind each other.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vit_device.hpp"
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SHD = 64;                  // head_dim
constexpr int ST_WAVES = 8, ST_THREADS = ST_WAVES * 64;
constexpr int SKT = 4;                   // key tiles (of 32) per chunk
constexpr int SKEYS = SKT * 32;          // 128 keys
constexpr int SBUF = 2 * SKEYS * SHD;    // bf16 elements of one ring slot: K[128][64] then V[128][64]
constexpr int MAXB = 3;                  // query blocks per wave: tokens <= 8 * 3 * 32 = 768
constexpr int SUB = 2;                   // key tiles whose scores are in registers at a time
constexpr float kScaleS = 0.125f * 1.4426950408889634f;
// Row sums from the matrix pipe (softmax_pv): built, exact to the bf16 rounding of P, and measured SLOWER here (2.37 against 2.22 ms per
// launch at ViT-L/16-384, batch 1024: the four extra MFMAs per unit cost more than the 32 v_add_f32 they replace -- the resident
// kernel of vit_attention.hip, whose units are longer, gains 1 % from the same change and keeps it).  Off.
#ifndef ST_MFMA_ROWSUM
#define ST_MFMA_ROWSUM 0
#endif
// The two waves of a SIMD (w and w + 4) run the same program and leave every barrier together: left alone they issue their score /
// P.V MFMAs at the same time (each burst then takes twice as long: one matrix pipe) and their softmax at the same time -- the unit
// is [512 MFMA cycles][576 VALU][...] per wave and NOTHING overlaps (stamps: 2,150 cycles per unit with the partner active, 1,216
// alone).  ST_STAGGER: waves 4-7 start every step ST_STAGGER x 64 cycles late (they issue the step's ring DMA first, then sleep),
// so that their matrix bursts fall into their partners' softmax and vice versa (CDNA4 guide, two waves per SIMD, item 9).  They
// own one query block fewer than waves 0-2, so the delay is off the step's critical path.  ST_DMA_LATE: waves 0-3 issue their share
// of the ring DMA after their first unit instead of at the step's start.
// Measured (interleaved, one device, ms per launch at ViT-L/16-384 batch 1024; round 4's kernel 2.317): no stagger 2.245, 4 x 64
// cycles 2.189, 8 x 64 2.220; a static s_setprio(1) for waves 4-7 on top: 2.33 (worse).  The model above promised far more than the
// 2.5 % the stagger gives: the partners fall back into step inside the step (profiles/r05/experiments).
#ifndef ST_STAGGER
#define ST_STAGGER 4
#endif
#ifndef ST_DMA_LATE
#define ST_DMA_LATE 1
#endif
#ifndef ST_PRIO
#define ST_PRIO 0
#endif
// ST_QREG: the Q fragments of a wave's first ST_QREG query blocks live in registers for the whole head (read from their LDS slot once,
// in the head's first step) instead of being re-read for every (block, 64-key) unit -- 4 of a score burst's 12 ds_read_b128.  Round 2
// had moved ALL of Q to LDS to make room for the second score buffer; with the step's hoisted constants gone (256 -> 211 VGPRs) all
// three blocks' fragments fit again (247 VGPRs, no scratch).  Measured, interleaved on one device: 0 / 1 / 2 / 3 blocks = 2.322 /
// 2.279 / 2.299 / 2.280 ms per launch at ViT-L/16-384 batch 1024 -- 1-2 %, the same bits; the LDS slots stay as the refill's landing place.
#ifndef ST_QREG
#define ST_QREG 3
#endif
// Tried on top of this structure and measured (profiles/r05/experiments/attention_stream_ab.jsonl): priority for the waves with the most
// query blocks instead (the same as none); the NEXT unit's score MFMAs issued between the exponentials of the current one, two per
// eight v_exp_f32, pinned with opaque asm -- same bits, 2.47 against 2.31 ms per launch: the burst in front of the softmax is faster.
#ifdef VIT_PROBES
unsigned long long *g_stream_dbg = nullptr;
#endif

// fp32 adds stay single instructions in this file (Makefile: -fno-slp-vectorize): beside MFMAs the packed forms (v_pk_add_f32 /
// v_pk_fma_f32, which hipcc makes of adjacent scalar operations under plain -O3) cost more than the two scalar instructions
// they replace (CDNA4 guide, cycle constants).  Not inline asm: hipcc inserts no wait state between a v_exp_f32 and an asm
// statement that reads its result (the transcendental-use hazard), and the sums came out wrong.
// max over the two half-waves' values (lane l and lane l ^ 32) without an LDS round trip (ds_bpermute): v_permlane32_swap
__device__ __forceinline__ float max_halves(float x) {
    // the instruction swaps the upper half of its first register with the lower half of its second: afterwards either half
    // holds {its own value, its partner's} in (a, b) or (b, a).  Inline asm: given the builtin with two copies of one value
    // hipcc (ROCm 7.2) drops the max of the two results; the s_nop covers the VALU-write -> permlane-read wait states it would
    // have inserted itself.
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

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

// QS: the Q columns of qkv hold c * q, c = 0.125 * log2(e) (the engine folds c into the in_proj weights, so q is rounded to bf16
// once either way).  The scores then come out of the matrix pipe in the exponent's units, and the accumulator of a score tile is
// INITIALISED with -m (the row's reference maximum): s - m costs no instruction, and the scale-and-subtract FMA of every score --
// a fifth of the softmax's VALU time, which is what bounds this kernel -- is gone.  The reference can move between that
// initialisation and the softmax (the units are software-pipelined): the rows it moved for are then corrected by one add.
template <bool QS>
__global__ __launch_bounds__(ST_THREADS) void attention_bf16_stream_kernel(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ out,
                                                                           int tokens, int heads, int n_items
#ifdef VIT_PROBES
                                                                           , unsigned long long *__restrict__ dbg
#endif
) {
#ifdef VIT_PROBES
    unsigned long long ts[16] = {};
    int nts = 0, it_no = 0;
#ifndef ST_UNIT_STAMPS
#define ST_UNIT_STAMPS 0   // 1 (probe build, tools/build_variant.sh -DVIT_PROBES -DST_UNIT_STAMPS=1): stamps INSIDE the units of block 0 in the second step of item 1
#endif
    [[maybe_unused]] bool ustamp_on = false;
#define ST_STAMP() do { if (!ST_UNIT_STAMPS && dbg && it_no == 1 && nts < 16) ts[nts++] = __builtin_amdgcn_s_memtime(); } while (0)
#define ST_USTAMP() do { if (ST_UNIT_STAMPS && dbg && ustamp_on && nts < 16) { __builtin_amdgcn_sched_barrier(0); ts[nts++] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define ST_STAMP() do { } while (0)
#define ST_USTAMP() do { } while (0)
#endif
    // ring: 2 x (K 16 KB + V 16 KB); behind it the Q blocks of the head (32 rows x 128 B each, swizzled like K), indexed by
    // block: a wave reads and refills only ITS blocks, so Q needs no barrier of its own
    extern __shared__ __attribute__((aligned(1024))) bf16_t lds[];
    bf16_t *const Qs = lds + 2 * SBUF;

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
    // "all but my n youngest vector-memory operations are done" for a wave-uniform run-time n (always even here: DMA pieces come in
    // pairs, stores in fours).  A count the switch does not hold waits for everything.  vmcnt retires in issue order and counts the
    // LDS-DMA pieces and the stores alike.
    auto wait_vm = [&](int n) __attribute__((always_inline)) {
        switch (n) {
#define ST_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
            ST_VMW(2) ST_VMW(4) ST_VMW(6) ST_VMW(8) ST_VMW(10) ST_VMW(12) ST_VMW(14) ST_VMW(16) ST_VMW(18) ST_VMW(20)
#undef ST_VMW
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    // LDS-DMA pieces this wave issues for chunk ch (dma_chunk below): two per group of 8 rows it owns
    auto chunk_dmas = [&](int ch) __attribute__((always_inline)) {
        const int rows = chunk_tiles(ch) * 32;
        return 2 * ((8 * wave < rows ? 1 : 0) + (8 * (wave + ST_WAVES) < rows ? 1 : 0));
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
    f32x16 o[MAXB][2];
    float m_run[MAXB], l_run[MAXB];
    [[maybe_unused]] bf16x8 qreg[ST_QREG > 0 ? ST_QREG : 1][4];
    constexpr bool MSUM = QS && ST_MFMA_ROWSUM;  // row sums from the matrix pipe (see softmax_pv)
    const int h4 = 4 * h;
    // Q block b of this wave for `item` -> LDS (4 wave instructions of 8 rows: ST_QDMA pieces); rows past the end are clamped
    constexpr int ST_QDMA = 4, ST_STORES = 4;  // vector-memory operations per block: Q refill, output stores (finish_block)
    auto dma_q_block = [&](int item, int b) __attribute__((always_inline)) {
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int row_l = lane_l >> 3, pos = lane_l & 7;
        const int4v rq = make_rsrc4(item_base(item));
        const int blk = wave + ST_WAVES * b;
#pragma unroll
        for (int t = 0; t < ST_QDMA; ++t) {
            const int lrow = 8 * t + row_l;
            int srow = blk * 32 + lrow;
            srow = srow < tk ? srow : tk - 1;
            dma16(rq, Qs + blk * (32 * SHD) + t * 512, srow * ld * 2 + ((pos ^ ((lrow >> 1) & 7)) << 4), 0);
        }
    };
    const int nb_wave = wave < nblk ? (nblk - 1 - wave) / ST_WAVES + 1 : 0;  // query blocks of this wave (wave-uniform)
    auto init_block = [&](int b) __attribute__((always_inline)) {
        m_run[b] = -INFINITY;
        l_run[b] = 0.0f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[b][dt][v] = 0.0f;
    };
    // Normalise and store block b of `item`.  Buffer stores: base (the image's rows, scalar) + one 32-bit lane offset, the 16-byte
    // piece's place as the instruction's immediate offset -- no 64-bit address per lane -- and rows past the last token are dropped
    // by the descriptor's range check (num_records = the image's bytes) instead of a predicate.  Exactly ST_STORES instructions.
    // A lane holds d = 8g + 4h + (0..3) of its row per column group g: 8-byte pieces, and a row-per-lane store of those is
    // issue-bound (16 instructions of 64 scattered 8-byte pieces per block).  v_permlane32_swap trades group g of the upper
    // half-wave for group g + 1 of the lower one: afterwards lanes 0-31 own d = 8g..8g+7 and lanes 32-63 d = 8g+8..8g+15 of their
    // rows -- one 16-byte store per pair of groups, half the instructions (T21 of the CDNA4 guide).
    auto finish_block = [&](int item, int b) __attribute__((always_inline)) {
        const int img = item / heads, head = item % heads;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)img * tokens * D, 0, tokens * D * 2, 0x00020000);
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        // The block's and head's offset goes into the LANE offset, not into the instruction's scalar offset: for a 16-byte buffer
        // store whose soffset is a register hipcc (ROCm 7.2) pads no wait state in front of a write of the data registers -- LLVM's
        // hazard recognizer exempts exactly that form -- and on gfx950 the next v_mul then overwrote the data of the lanes the store
        // reads last ((lane >> 2) & 3 == 3: query rows 12-15 and 28-31 of a block came out wrong, from run to run, on the
        // second-dispatched waves).  With soffset = 0 the two wait states are there.  tools/check_inline_asm.py rule (iv).
        const int voff = ((lane_l & 31) * D + 8 * (lane_l >> 5)) * 2 + ((wave + ST_WAVES * b) * 32 * D + head * SHD) * 2;
        float inv;
        if constexpr (MSUM) inv = 1.0f / l_run[b];                              // both half-waves hold the whole row sum
        else inv = 1.0f / (l_run[b] + __shfl_xor(l_run[b], 32));
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                bf16x4 w0, w1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w0[q] = (__bf16)(o[b][dt][4 * g + q] * inv);
                    w1[q] = (__bf16)(o[b][dt][4 * (g + 1) + q] * inv);
                }
                uint2 a = __builtin_bit_cast(uint2, w0), c = __builtin_bit_cast(uint2, w1);
                auto sx = __builtin_amdgcn_permlane32_swap(a.x, c.x, false, false);
                auto sy = __builtin_amdgcn_permlane32_swap(a.y, c.y, false, false);
                const u32x4 piece = {sx[0], sy[0], sx[1], sy[1]};  // lower half: [own g | upper's g]; upper: [lower's g+1 | own g+1]
                __builtin_amdgcn_raw_buffer_store_b128(piece, ro, voff + (dt * 32 + 8 * g) * 2, 0, 0);  // + immediate
            }
    };

    // One step = one chunk of one item, reading ring slot `slot` while the DMA of the following step fills the other one.
    //
    // Inside a step a wave walks its (block, 64-key sub-chunk) units.  The kernel is bound by the softmax's VALU work, not
    // by the matrix pipe (per unit 16 MFMAs = 512 cycles against 32 v_exp_f32 at quarter rate plus scale / sum / max /
    // bf16 packing), and a unit is a dependent chain S -> softmax -> PV: run one after the other the two never overlapped
    // (1,900 cycles per unit measured).  So the units are software-pipelined -- the score MFMAs of the NEXT unit are issued
    // before the softmax of the current one, two score buffers alternate -- and the element-wise work is written on float
    // pairs (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two values per instruction).
    //
    // first: the item's first step -- the Q blocks of this wave may still be on their way (refilled during the previous item's last
    //        step): block b's first scores wait for "all but q_younger[b] of my operations", counted.
    // last:  the item's last step -- each block is normalised, stored, re-initialised and its Q slot refilled for next_item right
    //        after its last unit.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto step = [&](int slot, int item, int ch, int next_item, int next_ch, bool first, bool last, bool ch0) __attribute__((always_inline)) {
        asm volatile("" : "+s"(tk));
        const int ncd = next_item >= 0 ? chunk_dmas(next_ch) : 0;  // operations the DMA below adds in front of everything later
        const bool dma_late = ST_DMA_LATE && wave < 4;  // wave-uniform
        if (next_item >= 0 && !dma_late) dma_chunk(next_item, next_ch, slot ^ 1);
        if (ST_STAGGER > 0 && wave >= 4) __builtin_amdgcn_s_sleep(ST_STAGGER);
        const bf16_t *Ks = lds + slot * SBUF, *Vs = Ks + SKEYS * SHD;
        const int key_base = chunk_first(ch) * 32;
        int valid = chunk_tiles(ch) * 32;                      // keys of this chunk ...
        valid = tk - key_base < valid ? tk - key_base : valid;  // ... that exist
        const bool two = SUB * 32 < valid;                      // wave-uniform: the chunk has a second sub-chunk

        f32x16 st[2][SUB];  // two score buffers
        [[maybe_unused]] float m_init[2] = {0.0f, 0.0f};  // QS: the reference the buffer's accumulators were initialised with (finite)
        // scores of sub-chunk k0 of block b -> buffer `buf`.  A sub-chunk is always computed whole: keys past `valid` (a
        // chunk of 3 tiles, the end of the sequence) hold older, finite data in LDS and are masked to -inf below.
        auto scores = [&](int buf, int b, int k0) __attribute__((always_inline)) {
            // Row r of a 32-row block (Q or K alike) at r * 128 B, its 16-byte chunk c at position c ^ ((r >> 1) & 7); k-step ks
            // wants chunk 2 ks + h, i.e. position ((h ^ sw) ^ 2 ks): ONE lane register and an XOR per k-step.  Recomputed from the
            // lane id here: kept across the unit's softmax these addresses were what the allocator spilled.
            int lane_l = lane;
            asm volatile("" : "+v"(lane_l));
            const int rr = lane_l & 31, hh = lane_l >> 5;
            const int frag_l = rr * SHD + (((hh ^ (rr >> 1)) & 7) << 3);  // bf16 elements
            if (first && k0 == 0 && b > 0) wait_vm((nb_wave - 1 - b) * (ST_QDMA + ST_STORES) + ncd);  // block b's Q has landed
            const bf16_t *qblk = Qs + (wave + ST_WAVES * b) * (32 * SHD);
            bf16x8 qf[4];
            if (b < ST_QREG) {
                if (ch0 && k0 == 0) {  // the head's first unit of this block: its Q slot -> registers, for all of the head's units
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) qreg[b < ST_QREG ? b : 0][ks] = *reinterpret_cast<const bf16x8 *>(qblk + (frag_l ^ (ks << 4)));
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[ks] = qreg[b < ST_QREG ? b : 0][ks];
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(qblk + (frag_l ^ (ks << 4)));
            }
            float init = 0.0f;
            if constexpr (QS) {
                const float mr = m_run[b];
                m_init[buf] = mr == -INFINITY ? 0.0f : mr;  // a block's first unit has no reference yet
                init = -m_init[buf];
            }
#pragma unroll
            for (int u = 0; u < SUB; ++u)
#pragma unroll
                for (int v = 0; v < 16; ++v) st[buf][u][v] = init;
            // K fragments are read one k-step AHEAD of the MFMAs that use them (two register sets): left to itself hipcc issues
            // read - wait - MFMA for every instruction, i.e. every MFMA pays an LDS round trip
            bf16x8 kf[2][SUB];
            auto read_k = [&](int ks, int set) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < SUB; ++u)
                    kf[set][u] = *reinterpret_cast<const bf16x8 *>(Ks + (k0 + u) * (32 * SHD) + (frag_l ^ (ks << 4)));
            };
            read_k(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) read_k(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < SUB; ++u)  // the SUB tiles are independent accumulation chains
                    st[buf][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks & 1][u], qf[ks], st[buf][u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // online softmax of buffer `buf` (running max m, running sum l, O rescaled by 2^((m_old - m_new) c) when m moved),
        // then O^T += V^T . P^T
        auto softmax_pv = [&](int buf, int b, int k0) __attribute__((always_inline)) {
            if ((k0 + SUB) * 32 > valid) {  // wave-uniform: the sub-chunk reaches past the valid keys
                int h4l = h4;
                asm volatile("" : "+v"(h4l));  // opaque: the key indices below are made HERE, in the rare branch, not hoisted as 64 loop invariants
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int kloc = (k0 + u) * 32 + (v & 3) + 8 * (v >> 2) + h4l;
                        st[buf][u][v] = kloc < valid ? st[buf][u][v] : -INFINITY;
                    }
            }
            f32x2 mx2 = {-INFINITY, -INFINITY};
#pragma unroll
            for (int u = 0; u < SUB; ++u)
#pragma unroll
                for (int v = 0; v < 16; v += 2) mx2 = __builtin_elementwise_max(mx2, f32x2{st[buf][u][v], st[buf][u][v + 1]});
            float cmax = fmaxf(mx2[0], mx2[1]);
            cmax = QS ? max_halves(cmax) : fmaxf(cmax, __shfl_xor(cmax, 32));
            // The running maximum is a REFERENCE, not a bound: it moves only when a row's scores outgrow it by more than
            // kDefer in the exponent (T13 of the CDNA4 guide).  P then reaches 2^kDefer instead of 1 -- the same relative
            // precision in bf16 and in the fp32 sums -- and the O-wide rescale below, which used to run in almost every
            // sub-chunk (some row of 32 nearly always finds a slightly larger score), runs a few times per head.
            constexpr float kDefer = 8.0f;
            const float m_old = m_run[b];
            if (b == 0) ST_USTAMP();  // maximum over the unit's 32 scores done
            f32x2 sum2 = {0.0f, 0.0f};
            float m_new;
            if constexpr (QS) {
                // st = s - m_init (already in the exponent's units).  Common case: the reference stays (m_new == m_old == m_init)
                // and the exponentials are taken of the accumulators as they stand.
                const float smax = cmax + m_init[buf];
                m_new = smax - m_old > kDefer ? smax : m_old;  // first unit: m_old = -inf -> smax (finite)
                m_run[b] = m_new;
                const float delta = m_init[buf] - m_new;       // 0 unless the reference moved since the initialisation
                if (__any(delta != 0.0f)) {                    // wave-uniform
                    const f32x2 d2 = {delta, delta};
#pragma unroll
                    for (int u = 0; u < SUB; ++u)
#pragma unroll
                        for (int v = 0; v < 16; v += 2) {
                            const f32x2 t = f32x2{st[buf][u][v], st[buf][u][v + 1]} + d2;
                            st[buf][u][v] = t[0];
                            st[buf][u][v + 1] = t[1];
                        }
                }
                [[maybe_unused]] float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const float e = __builtin_amdgcn_exp2f(st[buf][u][v]);  // exp2(-inf) = 0: masked keys
                        st[buf][u][v] = e;
                        if constexpr (!MSUM) s4[v & 3] += e;
                    }
                if constexpr (!MSUM) sum2 = f32x2{s4[0] + s4[1], s4[2] + s4[3]};
            } else {
                m_new = (cmax - m_old) * kScaleS > kDefer ? cmax : m_old;  // first sub-chunk: m_old = -inf -> cmax (finite)
                m_run[b] = m_new;
                const f32x2 sc2 = {kScaleS, kScaleS}, mxs2 = {-m_new * kScaleS, -m_new * kScaleS};
#pragma unroll
                for (int u = 0; u < SUB; ++u)
#pragma unroll
                    for (int v = 0; v < 16; v += 2) {
                        const f32x2 t = __builtin_elementwise_fma(f32x2{st[buf][u][v], st[buf][u][v + 1]}, sc2, mxs2);
                        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};  // exp2(-inf) = 0: masked keys
                        st[buf][u][v] = e[0];
                        st[buf][u][v + 1] = e[1];
                        sum2 += e;
                    }
            }
            if (__any(m_new != m_old)) {  // wave-uniform: most sub-chunks leave every row's maximum where it was
                const float alpha = __builtin_amdgcn_exp2f((m_old - m_new) * (QS ? 1.0f : kScaleS));  // first one: exp2(-inf) = 0
                l_run[b] *= alpha;
                const f32x2 a2 = {alpha, alpha};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int v = 0; v < 16; v += 2) {
                        const f32x2 w = f32x2{o[b][dt][v], o[b][dt][v + 1]} * a2;
                        o[b][dt][v] = w[0];
                        o[b][dt][v + 1] = w[1];
                    }
            }
            if constexpr (!MSUM) l_run[b] += sum2[0] + sum2[1];
            if (b == 0) ST_USTAMP();  // decision, exponentials, sums, rescale done
            // P.V: the V^T fragments of 16-key group g+1 are read before the MFMAs of group g.
            // MSUM: the row sums come out of the matrix pipe too -- one more MFMA per 16-key group with an all-ones A operand gives
            // every register of `lsum` the sum of this lane's column of P over the group's 16 keys (both half-waves' keys: the
            // product runs over k), i.e. the sum of the probabilities AS ROUNDED to bf16, the weights P.V really uses.  32 v_add_f32
            // per unit leave the vector pipe, which is the busier one here (30 % matrix-busy against 55 % vector-busy per SIMD).
            // Only element 0 of the sums' accumulator is ever read and its 16 elements are independent sums, so 15 of the registers stay
            // uninitialised (the empty asm "defines" them without an instruction) and element 0 starts as the running sum itself:
            // no zeroing, no add afterwards.
            bf16x8 vf[2][2];
            [[maybe_unused]] f32x16 lsum;
            if constexpr (MSUM) {
                asm volatile("" : "=v"(lsum));
                lsum[0] = l_run[b];
            }
            int lane_v = lane;
            asm volatile("" : "+v"(lane_v));  // the V^T fragment addresses are made per unit, not kept across the step
            auto read_v = [&](int g, int set) __attribute__((always_inline)) {  // g = 2 u + s2
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) vf[set][dt] = v_frag_tr(Vs, (k0 + (g >> 1)) * 32 + 16 * (g & 1), dt, lane_v);
            };
            read_v(0, 0);
#pragma unroll
            for (int g = 0; g < 2 * SUB; ++g) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (__bf16)st[buf][g >> 1][8 * (g & 1) + j];
                if (g + 1 < 2 * SUB) read_v(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[b][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[g & 1][dt], pf, o[b][dt], 0, 0, 0);
                if constexpr (MSUM) {
                    const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
                    lsum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lsum, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MSUM) l_run[b] = lsum[0];
            if (b == 0) ST_USTAMP();  // packing + P.V done
        };
        // a block's last unit of the head is behind it: out with it, and in with the next head's Q
        auto retire_block = [&](int b) __attribute__((always_inline)) {
            finish_block(item, b);
            init_block(b);
            if (next_item >= 0) dma_q_block(next_item, b);
        };

        // pipeline over this wave's blocks (wave, wave + 8, wave + 16: contiguous in b)
        ST_USTAMP();  // step start (behind the ring DMA issue / the stagger)
        if (wave < nblk) scores(0, 0, 0);
        ST_USTAMP();  // first score burst issued
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (wave + ST_WAVES * b >= nblk) continue;  // wave-uniform
            if (two) scores(1, b, SUB);                 // in the matrix pipe while the softmax below runs on the VALU
            if (b == 0) ST_USTAMP();  // second score burst issued
            softmax_pv(0, b, 0);
            if (b == 0 && dma_late && next_item >= 0) dma_chunk(next_item, next_ch, slot ^ 1);  // (in front of everything the waits count)
            if (last && !two) retire_block(b);
            if (b + 1 < MAXB && wave + ST_WAVES * (b + 1) < nblk) scores(0, b + 1, 0);  // the next block's first scores
            if (b == 0) ST_USTAMP();  // next block's score burst issued
            if (two) softmax_pv(1, b, SUB);
            if (last && two) retire_block(b);
        }
    };

    int item = blockIdx.x;
    if (item >= n_items) return;  // workgroup-uniform
    const int stride = gridDim.x;
    if (ST_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);  // the second-dispatched half loses the issue arbitration otherwise (T5, static form)
    // the ring starts zeroed: stale rows that a partial sub-chunk multiplies by 0 must be finite from the first step on
    for (int i = tid; i < 2 * SBUF / 8; i += ST_THREADS) reinterpret_cast<uint4 *>(lds)[i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    dma_chunk(item, 0, 0);
    int slot = 0;
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        init_block(b);
        if (wave + ST_WAVES * b < nblk) dma_q_block(item, b);
    }
    bool first_item = true;
    for (;;) {  // items
        // The head's first K/V chunk (every wave's pieces: hence the barrier) and THIS wave's first Q block have landed; its other
        // Q blocks may still be in flight.  Operations of this wave younger than block 0's refill: the (stores + refill) of its
        // other blocks.  The very first head has no stores in the queue: wait for everything.
        ST_STAMP();
        wait_vm(first_item ? 0 : (nb_wave - 1) * (ST_QDMA + ST_STORES));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ST_STAMP();
        const int next_item = item + stride;
        for (int ch = 0; ch < nch; ++ch) {
            // the step after this one: the next chunk of this head, or the first chunk of the next head
            const int ni = ch + 1 < nch ? item : (next_item < n_items ? next_item : -1);
            const int nc = ch + 1 < nch ? ch + 1 : 0;
#ifdef VIT_PROBES
            ustamp_on = it_no == 1 && ch == 1;
#endif
            step(slot, item, ch, ni, nc, ch == 0 && !first_item, ch == nch - 1, ch == 0);
            ST_STAMP();
            slot ^= 1;
            if (ch + 1 < nch) ring_barrier();  // the next chunk has landed, everybody is done with this one
            ST_STAMP();
            // (after the last chunk the barrier is the one at the top of the next item, behind the counted wait)
        }
        if (next_item >= n_items) break;
        item = next_item;
        first_item = false;
#ifdef VIT_PROBES
        ++it_no;
#endif
    }
#ifdef VIT_PROBES
    if (dbg && lane == 0) {
        unsigned long long *d = dbg + ((size_t)blockIdx.x * ST_WAVES + wave) * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = ts[k];
    }
#endif
#undef ST_STAMP
#undef ST_USTAMP
}

// 224 < tokens <= 704 (the head's Q blocks share the LDS with the K/V ring).  q_scaled: the Q columns hold 0.125 * log2(e) * q.
// Returns a hipError_t value.
template <bool QS>
static int launch_stream(hipStream_t s, const unsigned short *qkv, unsigned short *out, int tokens, int heads, int items, int grid,
                         size_t lds_bytes, int dev) {
    static vitdev::PerDeviceOnce attr_set;  // the attribute belongs to this device's copy of the kernel
    if (!attr_set.is_done(dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attention_bf16_stream_kernel<QS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return static_cast<int>(e);
        attr_set.set(dev);
    }
#ifdef VIT_PROBES
    hipLaunchKernelGGL(attention_bf16_stream_kernel<QS>, dim3(grid), dim3(ST_THREADS), lds_bytes, s, qkv, out, tokens, heads, items, g_stream_dbg);
#else
    hipLaunchKernelGGL(attention_bf16_stream_kernel<QS>, dim3(grid), dim3(ST_THREADS), lds_bytes, s, qkv, out, tokens, heads, items);
#endif
    return static_cast<int>(hipGetLastError());
}

int attention_bf16_stream(hipStream_t s, const unsigned short *qkv, unsigned short *out, int n_images, int tokens, int heads, bool q_scaled) {
    int dev = 0;
    const int cus = vitdev::current_cus(&dev);
    if (cus <= 0) return static_cast<int>(hipErrorInvalidDevice);
    const int nblk = (tokens + 31) / 32;
    const size_t lds_bytes = (size_t)(2 * SBUF + nblk * 32 * SHD) * sizeof(bf16_t);  // ring + the head's Q blocks
    // at least two K/V chunks per head: the counted waits of the kernel assume that a head's first and last step are different steps
    if (tokens > ST_WAVES * MAXB * 32 || nblk <= SKT || lds_bytes > 160 * 1024) return static_cast<int>(hipErrorInvalidValue);
    const int items = n_images * heads;
    const int grid = items < cus ? items : cus;
    return q_scaled ? launch_stream<true>(s, qkv, out, tokens, heads, items, grid, lds_bytes, dev)
                    : launch_stream<false>(s, qkv, out, tokens, heads, items, grid, lds_bytes, dev);
}

}  // namespace vitattn
