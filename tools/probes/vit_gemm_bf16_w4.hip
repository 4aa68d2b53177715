// tools/probes/vit_gemm_bf16_w4.hip -- EXPERIMENT (probe build only, tools/gemm_bf16_w4.py): bf16 "NT" GEMM with FOUR waves per workgroup,
// one per SIMD, each owning a 128 x 128 corner of the 256 x 256 tile.
//
// Why: the shipped kernel (vit_gemm_bf16_pp.hip, 8 waves x 128 x 64) reads 384 B of LDS per v_mfma_f32_16x16x32_bf16 -- 96 B/clk per
// CU of the 128 the LDS has -- and is 41-62 % MFMA-busy.  A 128 x 128 wave tile reads 8 fragments for 16 MFMAs of 32x32x16: 256 B per
// instruction, 64 B/clk per CU.  The price: 256 accumulator registers per lane (AGPRs: one wave per SIMD has 512 registers), nobody
// to run while a wave waits (its own fragment reads are issued one k-chunk ahead), and an epilogue nothing overlaps.
//
// Staging: LDS-DMA (buffer_load_dwordx4 ... lds, 16 B per lane, inline assembly so that the compiler does not drain it in front of
// every LDS read), a ring of four 32-KB slots (K step 32), filled three steps ahead straight through tile boundaries, counted
// vmcnt waits; one workgroup barrier per K step, placed behind the first half of the step's MFMAs; wave-private epilogue.
// (First form, two 64-KB stages one step ahead: the copy's round trip, 2-3 us, was longer than the 2,048 cycles of a step --
// QKV 638 TFLOP/s against the shipped kernel's 1,059; its loop alone, without the copies, ran K = 3072 at 1,393.)
#ifdef VIT_PROBES
#include "vit_device.hpp"
#include "vit_gemm_common.hpp"

namespace vitgemm {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef unsigned short bf16_t;

constexpr int WBM = 256, WBN = 256, WBK = 32;   // K step 32: 64-B LDS rows, four 32-KB ring slots, loads three steps ahead
constexpr int ROWB = WBK * 2;                     // 64 bytes per LDS row
constexpr int SLOT_BYTES = (WBM + WBN) * ROWB;    // 32 KB
constexpr int NSLOT = 4;
constexpr int BIAS_OFF = NSLOT * SLOT_BYTES;      // two 1-KB bias slots (tile parity)
constexpr int SCR_PITCH = 128 + 8;                // wave-private epilogue scratch: 32 rows x 64 bf16, padded
constexpr int SCR_OFF = BIAS_OFF + 2 * 1024;
constexpr int SCR_BYTES = 32 * SCR_PITCH;         // 4352 B per wave
constexpr int LDS_TOTAL = SCR_OFF + 4 * SCR_BYTES;
constexpr int WTHREADS = 256;

// One global -> LDS instruction (16 B per lane, lane-linear on the LDS side), hidden from the compiler's wait-count pass
__device__ __forceinline__ void dma16(unsigned lds_addr, int voff, __amdgpu_buffer_rsrc_t rsrc, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

// The lane id, recomputed where it is used once per tile (two instructions) instead of living in a register across the K loop:
// with 256 accumulators and 64 fragment registers the allocator spilled such values, and a spill reload is a vector load -- the
// compiler waits for it with vmcnt(0), which drains the whole staging queue it knows nothing about.
__device__ __forceinline__ int fresh_lane() {
    int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}

// One accumulator element out of its AGPR, where the program says: left to itself the register allocator copies ALL 256
// accumulators into VGPRs at the top of the epilogue (vector instructions cannot read AGPRs), and spills.
__device__ __forceinline__ float acc_read(float in_agpr) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(in_agpr));
    return t;
}

// DBG (timing only, results wrong): 1 = no LDS-DMA in the K loop, 2 = no MFMA / fragment reads, 5 = the copies alone (no MFMA, no
// epilogue), 4 = the same with TWO steps in flight instead of three (is the feed rate bytes-in-flight over latency?)
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(WTHREADS, 1) void gemm_bf16_w4_kernel(const Bf16Params p) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves: 128 rows x 128 columns each
    const unsigned lds_base = (unsigned)(size_t)lds;

    const int total = p.tiles_m * p.tiles_n, nwg = gridDim.x;
    const int first = xcd_remap(blockIdx.x, nwg);
    if (first >= total) return;  // workgroup-uniform
    const int nk = p.K / WBK;    // >= 4 (launcher)

    // ---- load cursor: the (tile, K step) the next staging instruction fetches.  Wave w fills rows [64w, 64w + 64) of the A part
    // and of the W part of a slot, 4 instructions each (16 rows x 64 B per instruction): lane l lands at (row 16q + l/4, chunk
    // l%4) and fetches source chunk (l%4) ^ ((row >> 2) & 3) -- the swizzle that keeps 8 consecutive rows of one logical chunk on
    // 8 different 16-byte bank groups for the fragment reads.
    int l_tile = first, l_kt = 0, l_issued = 0;  // l_issued: staging steps issued so far (ring slot = l_issued % NSLOT)
    int a_voff[4], w_voff[4];
    __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
    auto set_load_tile = [&](int t, int bias_slot) {
        const int lane = fresh_lane();
        int tm, tn;
        tile_coords(t, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
        const int m0 = tm * WBM, n0 = tn * WBN;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.A) + (size_t)m0 * p.lda, 0, 0x7fffffff, 0x00020000);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.W) + (size_t)n0 * p.ldw, 0, 0x7fffffff, 0x00020000);
        if (wave == 0) {  // the tile's 256 bias values -> their LDS slot: older than the step's own loads; columns past N read as zero
            const int nleft = p.N - n0 < WBN ? p.N - n0 : WBN;
            const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias) + n0, 0, nleft * 4, 0x00020000);
            dma16(lds_base + BIAS_OFF + bias_slot * 1024, lane * 16, b_rsrc, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 64 + q * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 2) & 3);
            const int m = m0 + row < p.M ? row : p.M - 1 - m0;
            const int n = n0 + row < p.N ? row : p.N - 1 - n0;
            a_voff[q] = m * p.lda * 2 + chunk * 16;
            w_voff[q] = n * p.ldw * 2 + chunk * 16;
        }
    };
    int l_tiles_done = 0;  // tiles the load cursor has started (bias slot parity)
    auto issue_step = [&]() {  // one K step of one tile into the next ring slot; past the last tile: harmless re-reads (uniform queue depth)
        if (DBG == 1 && l_issued >= 3) { ++l_issued; return; }
        const unsigned slot = lds_base + (l_issued % NSLOT) * SLOT_BYTES;
        if (l_kt == 0 && l_tile < total) {
            set_load_tile(l_tile, l_tiles_done & 1);
            ++l_tiles_done;
        }
        const int koff = l_kt * WBK * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dma16(slot + (wave * 64 + q * 16) * ROWB, a_voff[q], a_rsrc, koff);
            dma16(slot + WBM * ROWB + (wave * 64 + q * 16) * ROWB, w_voff[q], w_rsrc, koff);
        }
        ++l_issued;
        if (l_tile < total && ++l_kt == nk) {
            l_kt = 0;
            l_tile += nwg;
            if (l_tile >= total) l_kt = nk - 1;  // (stays on the last step of the last tile)
        }
    };

    // ---- fragment addresses: row (wave tile row + r), logical 16-B chunk 2 ks + h of the 64-B row, swizzled by (row >> 2) & 3 --
    int a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = (wm * 128 + i * 32 + r) * ROWB;
        w_off[i] = WBM * ROWB + (wn * 128 + i * 32 + r) * ROWB;
    }
    const int sw = (r >> 2) & 3;  // every fragment row of this lane is r (mod 32): same swizzle key
    bf16x8 wf[2][4], af[2][4];    // set 0: k-chunk 0 of a step, set 1: k-chunk 1
    auto read_frags = [&](int set, int slot, int ks) {
        const char *base = lds + slot * SLOT_BYTES + (((2 * ks + h) ^ sw) & 3) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[set][j] = *reinterpret_cast<const bf16x8 *>(base + w_off[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[set][i] = *reinterpret_cast<const bf16x8 *>(base + a_off[i]);
    };

    // ---- prologue: three steps in flight, the first one waited for ----------------------------------------------------------
    issue_step();
    issue_step();
    if (DBG != 4) issue_step();
    if (DBG == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // step 0 (and wave 0's bias copy) landed; steps 1, 2 may be in flight
    __syncthreads();
    read_frags(0, 0, 0);

    int gstep = 0;          // compute step, counted across tiles (ring slot = gstep % NSLOT)
    int since_epilogue = 2;  // compute steps since this wave's last 32 epilogue stores entered its queue (>= 2: none in the way)
    f32x16 acc[4][4];        // [j: 32 columns][i: 32 rows]
    for (int tile = first, tcount = 0; tile < total; tile += nwg, ++tcount) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.0f;
        for (int kt = 0; kt < nk; ++kt, ++gstep) {
            const int slot = gstep % NSLOT;
            // k-chunk 1 of this step, then the MFMAs of k-chunk 0: 512 cycles of matrix work queued ...
            constexpr bool NO_MATH = DBG == 2 || DBG == 4 || DBG == 5;
            if (!NO_MATH) read_frags(1, slot, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (!NO_MATH) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][j], af[0][i], acc[j][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ... in whose shadow the next step becomes visible: its loads are older than the 8 of step + 2 (and than this wave's 32
            // epilogue stores when those came in between), vmcnt retires in order; the barrier also frees the slot of step - 1
            if (DBG == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (since_epilogue < 2) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            ++since_epilogue;
            __syncthreads();
            issue_step();  // step + 3 into the slot of step - 1
            if (!NO_MATH) read_frags(0, (gstep + 1) % NSLOT, 0);  // k-chunk 0 of the next step (the next tile's first, at a boundary)
            __builtin_amdgcn_sched_barrier(0);
            if (!NO_MATH) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][j], af[1][i], acc[j][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (DBG == 4 || DBG == 5) continue;
        // ---- epilogue, wave-private: a 32-row x 64-column block at a time through 4 KB of LDS, out as 128-B row segments -------
        // In registers a lane owns row r of a 32-row block and 4 consecutive columns per group (operand roles swapped: MFMA A = W
        // rows).  No workgroup barrier: the other waves are already in the next tile's K loop.  Bias from its LDS slot (copied with
        // the tile's first K step), so the epilogue holds no vector load the compiler would drain the staging queue for.
        {
            int tm, tn;
            tile_coords(tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
            const int m_w = tm * WBM + wm * 128, n_w = tn * WBN + wn * 128;
            const float *bias_l = reinterpret_cast<const float *>(lds + BIAS_OFF + (tcount & 1) * 1024) + wn * 128;
            char *const scr = lds + SCR_OFF + wave * SCR_BYTES;
            // buffer stores: rows past M fall outside the descriptor and are dropped by the hardware, so a wave ALWAYS issues its 32
            // store instructions -- the counted waits of the K loop depend on that number.  One vector offset (the lane's row and
            // 16-B chunk inside a 8-row x 64-column piece), everything else in the scalar offset.  (N % 64 == 0: launcher.)
            const int rows_left = p.M - m_w < 128 ? (p.M - m_w > 0 ? p.M - m_w : 0) : 128;
            // (every part through readfirstlane: the clamp above is selected as a vector instruction, and a descriptor with one
            // word in a VGPR turns each of the 32 stores into a waterfall loop)
            const unsigned long long c_addr = reinterpret_cast<unsigned long long>(static_cast<bf16_t *>(p.C) + (size_t)m_w * p.ldc + n_w);
            const unsigned c_lo = __builtin_amdgcn_readfirstlane((unsigned)c_addr), c_hi = __builtin_amdgcn_readfirstlane((unsigned)(c_addr >> 32));
            const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)c_hi << 32) | c_lo), 0,
                                                                                   __builtin_amdgcn_readfirstlane(rows_left * p.ldc * 2), 0x00020000);
            const int ln = fresh_lane(), r = ln & 31, h = ln >> 5;
            const int voff = ((ln >> 3) * p.ldc + (ln & 7) * 8) * 2;
            const char *const scr_rd = scr + (ln >> 3) * SCR_PITCH + (ln & 7) * 16;
            char *const scr_wr = scr + r * SCR_PITCH + h * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {  // columns [64 jp, 64 jp + 64) of the wave's 128
                    __builtin_amdgcn_sched_barrier(0);  // (one piece at a time: hoisting all 256 accumulator reads to the top costs 200 registers)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * jp + jj;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias_l + j * 32 + 8 * g + 4 * h);
                            bf16x4 y;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float t = acc_read(acc[j][i][4 * g + q]) + b4[q];
                                if constexpr (EPI == VITHIP_BF16_EPI_BF16_GELU) t = gelu_erf(t);
                                y[q] = (__bf16)t;
                            }
                            *reinterpret_cast<bf16x4 *>(scr_wr + (jj * 32 + 8 * g) * 2) = y;
                        }
                    }
                    // (same wave: LDS operations complete in order, the reads below see the writes above)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {  // 32 rows x 8 chunks of 16 B over 64 lanes
                        const uint4 y = *reinterpret_cast<const uint4 *>(scr_rd + it * 8 * SCR_PITCH);
                        // (a 64-column piece past N: same instruction, every lane out of the descriptor's range)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), c_rsrc, n_w + jp * 64 + 64 <= p.N ? voff : 0x7ffffff0,
                                                               ((i * 32 + it * 8) * p.ldc + jp * 64) * 2, 0);
                    }
                }
            since_epilogue = 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no copy into this workgroup's LDS may outlive it
}

}  // namespace

// probe build: variant 5 of vithip_gemm_bf16 (bf16 epilogues only; N % 8 == 0)
int launch_gemm_bf16_w4(hipStream_t s, const Bf16Params &p, int epilogue, int cus) {
    if (p.N % 64 || p.K % WBK || p.K < 4 * WBK || (size_t)p.lda * 2 * 256 >= (1u << 31) || (size_t)p.ldw * 2 * 256 >= (1u << 31))
        return static_cast<int>(hipErrorInvalidValue);
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < cus ? total : cus), block(WTHREADS);
    switch (epilogue) {
        case VITHIP_BF16_EPI_BF16: hipLaunchKernelGGL(gemm_bf16_w4_kernel<VITHIP_BF16_EPI_BF16>, grid, block, 0, s, p); break;
        case VITHIP_BF16_EPI_BF16_GELU: hipLaunchKernelGGL(gemm_bf16_w4_kernel<VITHIP_BF16_EPI_BF16_GELU>, grid, block, 0, s, p); break;
        case 501: hipLaunchKernelGGL((gemm_bf16_w4_kernel<VITHIP_BF16_EPI_BF16, 1>), grid, block, 0, s, p); break;
        case 502: hipLaunchKernelGGL((gemm_bf16_w4_kernel<VITHIP_BF16_EPI_BF16, 2>), grid, block, 0, s, p); break;
        case 504: hipLaunchKernelGGL((gemm_bf16_w4_kernel<VITHIP_BF16_EPI_BF16, 4>), grid, block, 0, s, p); break;
        case 505: hipLaunchKernelGGL((gemm_bf16_w4_kernel<VITHIP_BF16_EPI_BF16, 5>), grid, block, 0, s, p); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

}  // namespace vitgemm
#endif  // VIT_PROBES
