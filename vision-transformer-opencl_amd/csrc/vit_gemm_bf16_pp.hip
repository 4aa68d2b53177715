// csrc/vit_gemm_bf16_pp.hip -- bf16 MFMA "NT" GEMM, ping-pong schedule (the tuned kernel behind
// vithip_gemm_bf16; csrc/vit_gemm_bf16.hip keeps the simple two-stage kernel as variant 1).
//
//   C = epilogue(A[M][K] . W[N][K]^T + bias)       (same contract and epilogues as vit_gemm_bf16.hip)
//
// Why a second kernel: in the two-stage kernel the two waves of a SIMD read LDS at the same time and then
// both want the matrix pipe, and only one K step of LDS-DMA is in flight.  Here (CDNA4 guide, 256x256
// "8-phase" recipe, re-derived for this operand layout):
//   * 8 waves = two GROUPS of four (group g owns tile rows [128g, 128g+128), wave (g, wc) the columns
//     [64wc, 64wc+64)); a SIMD hosts one wave of each group.  Group 1 runs ONE barrier behind group 0, so while
//     one group is in its MFMA section the other is in its load section (ds_read + LDS-DMA issue): the
//     matrix pipe of every SIMD always has a feeder.  s_setprio(1) marks the MFMA section.
//   * A K step (64) of the wave's 128x64 output is four PHASES, one 64(m) x 32(n) quadrant each, 16
//     v_mfma_f32_16x16x32_bf16 per phase; the quadrant order (m0,n0) (m0,n1) (m1,n1) (m1,n0) means phase 0
//     reads 8 X + 4 W fragments, phase 1 4 W, phase 2 8 X, phase 3 nothing.
//   * The 64 KB of a K step are four 16-KB HALF-TILES, cut so that each is read in exactly one phase:
//       kind 0  X'0 = token rows   {128g' + [0,64)}    read in phase 0
//       kind 1  W'0 = feature rows {64wc' + [0,32)}    read in phase 0
//       kind 2  W'1 = feature rows {64wc' + [32,64)}   read in phase 1
//       kind 3  X'1 = token rows   {128g' + [64,128)}  read in phase 2
//     One half-tile is issued per phase (2 LDS-DMA instructions per wave), SIX half-tiles ahead of its
//     first reader: half-tile j (= 4 * kstep + kind, counted across tiles) is issued in phase j - 6, waited
//     for with `s_waitcnt vmcnt(8)` at the end of the load section of phase j - 2 (8 = the four younger
//     half-tiles), and first read in phase j - 1 or later -- i.e. one full phase after the wait, as the
//     barrier rules of LDS-DMA require.  The 8 LDS slots (2 K-step parities x 4 kinds) are re-filled no
//     earlier than two phases after their last reader.  The stream runs straight through tile boundaries
//     (persistent workgroups), so there is no per-tile pipeline fill.
//   * Bias: the 256 floats of a tile ride the same stream (one extra 1-KB LDS-DMA by wave 0 with the
//     tile's first W'0), so the epilogue needs no vector-memory load that would make hipcc drain the
//     LDS-DMA queue.  Extra VMEM operations (that DMA, epilogue stores) only make vmcnt(8) stricter.
//   * Epilogue from registers: a lane holds token (lane & 15) and 4 consecutive features per accumulator
//     (operand roles swapped: MFMA A = W rows).  bf16 outputs are transposed 16 tokens at a time through a
//     wave-private 2-KB LDS scratch and leave as 128-B row segments; fp32 (+residual) goes out as 64-B
//     segments directly.
#include "vit_gemm_common.hpp"

namespace vitgemm {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef unsigned short bf16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

#ifndef PP_LNF_RS
#define PP_LNF_RS 0
#endif
#ifndef PP_STAGGER
#define PP_STAGGER 0  // timing experiment: start-up phases of the persistent workgroups of the short fp32-residual GEMM, in cycles (0 = off, the product)
#endif
#ifndef PP_STAGGER_ALL
#define PP_STAGGER_ALL 0  // timing experiment: the phases for every K
#endif
#ifndef PP_GELU_LOCKSTEP
#define PP_GELU_LOCKSTEP 4  // pairs of the GELU epilogue advanced together: 0 (one chain after the other), 2, 4 (fc1 -1.0 % / -0.7 %, same bits)
#endif
constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int PTHREADS = 512;
constexpr int SLOT = 128 * 128;               // one half-tile: 128 rows x 64 bf16
constexpr int RING = 8 * SLOT;                // 128 KB
constexpr int BIAS_OFF = RING;                // two 1-KB bias slots (tile parity)
constexpr int SCRATCH = RING + 2 * 1024;      // 8 x 2 KB: first epilogue transpose buffer of every wave
constexpr int LDS_BYTES = SCRATCH + 8 * 2048;

#define PP_BARRIER()                            \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// STAMP: timing-only instrumentation for tools/gemm_bf16_probe.py (workgroup 0 records s_memtime around the
// sections of K step 3 of its first tile, and around that tile's epilogue; 32 values per wave in p.dbg).
// Barriers of the phase pipeline: ONE per phase and wave -- group 0 after its MFMA sections, group 1 after its load sections.
// Between two barriers group 0 runs [load p, MFMA p] and group 1 [MFMA p-1, load p]: while one group is in its MFMA section the
// other is in its load section, but nothing forces them to switch at the same moment.  Every half-tile is waited for (counted
// vmcnt) before a barrier that precedes its first read, and refilled two phases after its last read.  (A schedule with a
// barrier after every section of every wave -- twice as many -- was 1-3.5 % slower and is gone; so is a start-up skew between
// the persistent workgroups, which never measured outside the noise.)
#define PP_BARRIER_L()          \
    do {                        \
        if (g == 1) PP_BARRIER(); \
    } while (0)
#define PP_BARRIER_M()          \
    do {                        \
        if (g == 0) PP_BARRIER(); \
    } while (0)
#define PP_MFMA(a, b, c, x, y, z) (DBG == 2 ? (c) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z))
#define PP_LDS_FRAG(ptr) (DBG == 3 ? bf16x8{} : *reinterpret_cast<const bf16x8 *>(ptr))
#define PP_STAMP(idx)                                                                                   \
    do {                                                                                                \
        if constexpr (STAMP == 1) {                                                                     \
            if (stamp_on) stamps[idx] = (unsigned)__builtin_amdgcn_s_memtime();                                   \
        }                                                                                               \
    } while (0)

// Sum over the 8 lanes that share lane >> 3 (two quad steps and a mirror inside the half row): every lane ends with the total.
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return v;
}

// DBG (timing-only builds, results wrong by construction): 1 = no LDS-DMA inside the K loop, 2 = no MFMA, 3 = no fragment reads, 4 = no epilogue.
// LNF: the LayerNorm that follows (F32_RESIDUAL) or precedes (BF16, BF16_GELU) this GEMM is folded into it:
//   producer  F32_RESIDUAL + LNF: besides C (fp32) the epilogue stores bf16(C) to p.x16 and, per row and 64-column strip
//             of a wave, the partial (sum, sum of squares) of the fp32 values to p.partials -- vithip_rowstats_finalize turns
//             them into (rstd, mean * rstd) per row;
//   consumer  BF16 / BF16_GELU + LNF: A holds the UN-normalised bf16 rows, W the gamma-folded weight, bias the beta-folded
//             bias, and the epilogue applies  v = rstd_m * acc - (mean * rstd)_m * colsum_n + bias_n  (= LN(x) . W^T + b).
//             The tile's 256 row pairs and 256 column sums ride the LDS-DMA stream like the bias (waves 1..3).
template <int EPI, int STAMP = 0, int DBG = 0, bool LNF = false>
__global__ __launch_bounds__(PTHREADS) void gemm_bf16_pp_kernel(const Bf16Params p) {
    static_assert(!(LNF && STAMP), "the event log and the LayerNorm slots share LDS");
    constexpr int LN_OFF = LDS_BYTES;  // per tile parity: 1 KB column sums + 2 KB (rstd, mean*rstd) pairs
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES + (STAMP == 2 ? 8 * 1024 : 0) + (LNF ? 2 * 3072 : 0)];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;
    // STAMP == 2: event log of workgroup 0, 256 words per wave in LDS: (tag << 28) | (s_memtime & 0x0fffffff);
    // tags: 1 K-step start, 2 epilogue begin, 3 epilogue end.  Dumped to p.dbg (8 x 512 u32) at the end.
    [[maybe_unused]] int log_n = 0;
    auto log_event = [&](unsigned tag) __attribute__((always_inline)) {
        if constexpr (STAMP == 2) {
            if (blockIdx.x == 0 && log_n < 256) {
                const unsigned t = ((unsigned)__builtin_amdgcn_s_memtime() & 0x0fffffffu) | (tag << 28);
                if (lane == 0) *reinterpret_cast<unsigned *>(lds + LDS_BYTES + wave * 1024 + log_n * 4) = t;
                ++log_n;
            }
        }
    };

    // Feature (column) order inside a wave's 64 columns.  MFMA row i = 4a + q of column group j can be fed ANY W
    // row, so the mapping is chosen for the store pattern of the epilogue:
    //   natural (fp32 out): n = 16j + 4a + q          -> a lane's float4 per j; 4 lanes x 16 B = 64 B per token row
    //   paired  (bf16 out): n = 32(j>>1) + 8a + 4(j&1) + q -> a lane owns 8 consecutive features per j-pair: one 16-B
    //                       store, 4 lanes x 16 B = 64 B per row, no transpose through LDS
    // Both keep quadrant j>>1 = features [32(j>>1), +32), i.e. the W'0 / W'1 half-tiles.  The W slots get their
    // own XOR key so that the 16 rows one fragment read touches stay conflict-free in either order.
    constexpr bool F32OUT = EPI == VITHIP_BF16_EPI_F32_RESIDUAL || EPI == VITHIP_BF16_EPI_F32_EMBED;
    constexpr bool PAIRED = !F32OUT;

    const int total = p.tiles_m * p.tiles_n, nwg = gridDim.x;
    const int first = xcd_remap(blockIdx.x, nwg);
    if (first >= total) return;  // workgroup-uniform
    const int nk = p.K / PBK;
    // Timing experiment, OFF in the product (tools/build_variant.sh -DPP_STAGGER=<cycles>): start-up phases for the short fp32-residual
    // GEMM (out_proj of ViT-B/16: K = 768, a tile is 12 K-steps and an epilogue that moves 640 KB) -- the persistent workgroups of an XCD
    // start in four phases PP_STAGGER cycles apart, so that their read-modify-write epilogues (every CU's at the same moment otherwise,
    // 164 MB asked of the HBM at once) do not coincide.  Round 5, profiles/r05/experiments/bf16_gemm_start_phases.jsonl: out_proj at
    // batch 2048 -2.3 / -4.3 / -0.6 % on three boxes with 6,000-20,000 cycles, -4 % at batch 1024, +3.8 % at batch 256 (the wait is
    // 12 us of a 138-us launch); fc2 (K = 3072) and the bf16-output GEMMs only pay the wait (+1.6 ... +4.6 % at 12,000 / 45,000),
    // ViT-L's out_proj (K = 1024) is inside the noise.  Not a rule one can ship: the sign depends on the batch and the size on the box.
    if constexpr (PP_STAGGER > 0 && EPI == VITHIP_BF16_EPI_F32_RESIDUAL) {
        const int ph = (blockIdx.x >> 3) & 3;
        if (ph && (nk <= 12 || PP_STAGGER_ALL)) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)ph * PP_STAGGER) __builtin_amdgcn_s_sleep(64);
        }
    }

    // ---- LDS-DMA source offsets: per half-tile two instructions (q) of 8 rows x 128 B.  Lane l lands at
    // (row L = 16*wave + 8q + l/8, chunk l%8) of the slot and fetches source chunk (l%8) ^ ((L>>1)&7).
    int xvoff[2][2], wvoff[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int L = wave * 16 + q * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((L >> 1) & 7);
        const int wkey = PAIRED ? (((L >> 1) & 1) | (((L >> 3) & 3) << 1)) : ((L >> 1) & 7);
        const int wchunk = (lane & 7) ^ wkey;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int xrow = (L >> 6) * 128 + half * 64 + (L & 63);
            const int wrow = (L >> 5) * 64 + half * 32 + (L & 31);
            xvoff[half][q] = xrow * p.lda * 2 + chunk * 16;
            wvoff[half][q] = wrow * p.ldw * 2 + wchunk * 16;
        }
    }

    [[maybe_unused]] bool in_loop = false;
    // ---- load cursor (workgroup-uniform): the K step whose half-tiles are being issued
    int l_tile = first, l_kt = 0, l_tpar = 0;
    bool l_valid = true;
    __amdgpu_buffer_rsrc_t l_xr, l_wr, l_br;
    __amdgpu_buffer_rsrc_t n_xr, n_wr, n_br;  // descriptors of the cursor's NEXT tile, prepared ahead of the switch
    [[maybe_unused]] int l_m0 = 0, l_n0 = 0, n_m0 = 0, n_n0 = 0;  // LNF: origin of the cursor's tile and of its next tile
    auto tile_rsrc = [&](int tile, __amdgpu_buffer_rsrc_t &xr, __amdgpu_buffer_rsrc_t &wr, __amdgpu_buffer_rsrc_t &br, int &m0o, int &n0o) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords(tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
        const int m0 = tm * PBM, n0 = tn * PBN;
        m0o = m0, n0o = n0;
        const int mrows = p.M - m0 < PBM ? p.M - m0 : PBM, nrows = p.N - n0 < PBN ? p.N - n0 : PBN;
        xr = make_rsrc(p.A + (size_t)m0 * p.lda, (unsigned)mrows * p.lda * 2);  // rows past M read as zero
        wr = make_rsrc(p.W + (size_t)n0 * p.ldw, (unsigned)nrows * p.ldw * 2);
        br = make_rsrc(p.bias + n0, (unsigned)nrows * 4);
    };
    tile_rsrc(l_tile, l_xr, l_wr, l_br, l_m0, l_n0);
    n_xr = l_xr, n_wr = l_wr, n_br = l_br, n_m0 = l_m0, n_n0 = l_n0;
    // The tile switch costs two integer divisions (tile_coords); in a load section that was +700 cycles on the critical
    // path once per tile.  prepare_next() runs between the MFMAs of phase 1 of the last K step before the switch.
    auto prepare_next = [&]() __attribute__((always_inline)) {
        if (l_kt == nk - 1 && l_tile + nwg < total) tile_rsrc(l_tile + nwg, n_xr, n_wr, n_br, n_m0, n_n0);
    };
    auto advance = [&]() __attribute__((always_inline)) {
        if (++l_kt == nk) {
            l_kt = 0;
            l_tile += nwg;
            l_tpar ^= 1;
            l_valid = l_tile < total;
            l_xr = n_xr, l_wr = n_wr, l_br = n_br, l_m0 = n_m0, l_n0 = n_n0;
        }
    };
    // issue half-tile KIND of the cursor's K step into the slot of K-step parity `parbit`
    auto issue = [&](auto kind_c, int parbit) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value;
        if (!l_valid) return;
        if (DBG == 1 && in_loop) return;
        char *dst = lds + (parbit * 4 + KIND) * SLOT + wave * 2048;
        const int koff = l_kt * (PBK * 2);
        if constexpr (KIND == 1) {
            if (l_kt == 0 && wave == 0)  // the tile's bias row: lanes 0..63 x 16 B = 256 floats
                __builtin_amdgcn_raw_ptr_buffer_load_lds(l_br, (lds_void *)(lds + BIAS_OFF + l_tpar * 1024), 16, lane * 16, 0, 0, 0);
            if constexpr (LNF && !(EPI == VITHIP_BF16_EPI_F32_RESIDUAL || EPI == VITHIP_BF16_EPI_F32_EMBED)) {
                if (l_kt == 0 && wave >= 1 && wave <= 3) {  // the tile's column sums (wave 1) and row pairs (waves 2, 3)
                    char *ln = lds + LN_OFF + l_tpar * 3072;
                    if (wave == 1) {
                        const int nrows = p.N - l_n0 < PBN ? p.N - l_n0 : PBN;
                        const __amdgpu_buffer_rsrc_t r = make_rsrc(p.ln_colsum + l_n0, (unsigned)nrows * 4);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)ln, 16, lane * 16, 0, 0, 0);
                    } else {
                        const int mrows = p.M - l_m0 < PBM ? p.M - l_m0 : PBM;  // rows past M read as zero: v = bias, never stored
                        const __amdgpu_buffer_rsrc_t r = make_rsrc(p.ln_rows + (size_t)l_m0 * 2, (unsigned)mrows * 8);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(ln + (wave - 1) * 1024), 16, (wave - 2) * 1024 + lane * 16, 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (KIND == 0 || KIND == 3) {
            constexpr int H = KIND == 3;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(l_xr, (lds_void *)dst, 16, xvoff[H][0], koff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(l_xr, (lds_void *)(dst + 1024), 16, xvoff[H][1], koff, 0, 0);
        } else {
            constexpr int H = KIND == 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(l_wr, (lds_void *)dst, 16, wvoff[H][0], koff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(l_wr, (lds_void *)(dst + 1024), 16, wvoff[H][1], koff, 0, 0);
        }
    };
    // vmcnt retires in issue order and counts stores too.  During the first K step after an epilogue the
    // half-tile being waited for is OLDER than that epilogue's memory operations, so these may stay in flight on
    // top of the four younger half-tiles -- otherwise every tile would begin by draining its predecessor's stores.
    // Only when the epilogue issued exactly its nominal operations (interior tile, no predication): a count
    // larger than what was really issued would let a needed half-tile slip.
    constexpr int EPI_OPS = F32OUT ? 55 : 16;  // stores (+ loads); 8 + 55 = vmcnt's maximum
    int epi_slack = 0;  // waits left for which the half-tile needed is older than an unpredicated epilogue's operations
    auto wait_loads = [&]() __attribute__((always_inline)) {
        if (!l_valid) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (epi_slack > 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + EPI_OPS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        if (epi_slack > 0) --epi_slack;
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

    // ---- fragment read addresses.  Row L of a slot at byte L*128; 16-B chunk c of the row at (c ^ ((L>>1)&7)).
    // Every fragment row of this lane is l15 (mod 16), so the key is (l15>>1); k step ks (32 deep) uses
    // logical chunks 4ks + l4, i.e. the two k steps differ by bit 6 of the byte offset.
    const int key = (l15 >> 1) & 7;
    const int c0 = ((l4 ^ key) & 7) * 16;
    const int xbase = (g * 64 + l15) * 128 + c0;   // + kind*SLOT + i*2048 (+ parity*64K), ^64 for ks = 1
    // W fragment of column group jj (0/1 inside the quadrant): natural row 16jj + l15, paired row 8(l15>>2) + 4jj + (l15&3)
    const int wrow0 = PAIRED ? 8 * (l15 >> 2) + (l15 & 3) : l15;
    const int wkey = PAIRED ? (((l15 >> 1) & 1) | ((l15 >> 2) << 1)) : key;  // same for both jj
    const int wbase = (wc * 32 + wrow0) * 128 + ((l4 ^ wkey) & 7) * 16;       // + kind*SLOT + jj*WJ
    constexpr int WJ = PAIRED ? 4 * 128 : 16 * 128;

    int c_tile = first, c_tpar = 0, par = 0;
    [[maybe_unused]] unsigned stamps[24];
    [[maybe_unused]] bool stamp_on = false;
    [[maybe_unused]] bool first_tile = true;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xq[4][2], wq0[2][2], wq1[2][2];

    // ---- epilogue of one finished tile (wave-private; no barrier): bias (+GELU | +residual), store, clear
    auto epilogue_body = [&](int e_tile, int e_tpar) __attribute__((always_inline)) {
        if constexpr (DBG == 4) return;  // timing probe: no epilogue at all
        int tm, tn;
        tile_coords(e_tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
        const int mw = tm * PBM + g * 128, nw = tn * PBN + wc * 64;  // this wave's 128 x 64 block
        f32x4 b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            b4[j] = *reinterpret_cast<const f32x4 *>(lds + BIAS_OFF + e_tpar * 1024 +
                                                     (wc * 64 + (PAIRED ? 32 * (j >> 1) + 8 * l4 + 4 * (j & 1) : j * 16 + 4 * l4)) * 4);
        [[maybe_unused]] f32x4 s4[4];  // LNF consumer: column sums of this lane's features, (rstd, mean*rstd) of its 8 tokens
        [[maybe_unused]] f32x2 rs_all[8];
        [[maybe_unused]] const char *ln_rows_lds = nullptr;  // + i * 128: this lane's (rstd, mean*rstd) pair of m-tile i
        if constexpr (LNF && PAIRED) {
            const char *ln = lds + LN_OFF + e_tpar * 3072;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s4[j] = *reinterpret_cast<const f32x4 *>(ln + (wc * 64 + 32 * (j >> 1) + 8 * l4 + 4 * (j & 1)) * 4);
            ln_rows_lds = ln + 1024 + (g * 128 + l15) * 8;
#if PP_LNF_RS == 1
#pragma unroll
            for (int i = 0; i < 8; ++i) rs_all[i] = *reinterpret_cast<const f32x2 *>(ln_rows_lds + i * 128);
#endif
        }
        // Full-line accesses.  In registers a lane owns row (token) l15 and two 16-B chunks of it, so a plain store
        // instruction would write 16 rows x 64 B -- measured (tools/store_probe.py) at 14-19 B/clk per CU, against
        // 35-57 B/clk for 8 rows x 128 B with lane = 8 * row + chunk (the coalescer works on adjacent lanes).  Every
        // 16-row x 128-B block therefore goes through a wave-private 2-KB LDS buffer (unpadded; chunk c of row r sits
        // at c ^ (r & 7), conflict-free both ways), two buffers per wave so that block i+1 is written while block i
        // is read back: the first in SCRATCH, the second in this wave's eighth of the X'1 slot of the K step that
        // just finished (nobody reads or fills that slot before the next phase 1, which is behind a barrier).
        // Inline asm: hipcc would put `s_waitcnt vmcnt(0)` in front of ordinary LDS loads that follow LDS stores
        // while LDS-DMA is in flight (it cannot tell these buffers from the DMA ring) and drain the prefetch queue.
        // (Every inline-asm store below ends in `s_nop 1`: the next vector instruction may overwrite the store's data registers, and the
        // wait states a store of more than 8 bytes needs in front of such a write are padded by hipcc for its own instructions only.
        // Found in round 4 on an experimental epilogue -- a v_mov in the slot behind a store cost it single dwords, differently from
        // run to run; the shipped epilogues had the same exposure and only their register allocation kept them clear of it.)
        const unsigned lds0 = (unsigned)(size_t)(lds_void *)lds;
        const unsigned buf_a = lds0 + SCRATCH + wave * 2048;
        const unsigned buf_b = lds0 + ((par ^ 1) * 4 + 3) * SLOT + wave * 2048;
        const unsigned w_off = l15 * 128 + ((l4 ^ (l15 & 7)) * 16);                      // chunk l4; chunk 4 + l4 is at ^64
        const unsigned r_off = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) * 16);  // rows 8..15 at + 1024
        auto put = [&](int b, u32x4 c0, u32x4 c1) __attribute__((always_inline)) {
            const unsigned base = (b & 1) ? buf_b : buf_a;
            asm volatile("ds_write_b128 %0, %1" ::"v"(base + w_off), "v"(c0) : "memory");
            asm volatile("ds_write_b128 %0, %1" ::"v"(base + (w_off ^ 64)), "v"(c1) : "memory");
        };
        auto get = [&](int b, u32x4 &ra, u32x4 &rb) __attribute__((always_inline)) {  // issue only; wait with PP_LGKM0
            const unsigned base = ((b & 1) ? buf_b : buf_a) + r_off;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(ra), "=&v"(rb) : "v"(base) : "memory");
        };
#define PP_LGKM0(x, y) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y)::"memory")
        const int row8 = lane >> 3, ch8 = lane & 7;  // after the transpose: row (0..7, +8 for the second value), 16-B chunk
        const bool interior = tm * PBM + PBM <= p.M && tn * PBN + PBN <= p.N;  // workgroup-uniform
        epi_slack = interior ? 4 : 0;
        if constexpr (F32OUT) {
            float *C = static_cast<float *>(p.C);
            // Blocks: (m-tile i, column pair pj) = 16 rows x 32 floats; natural mapping: column groups 2pj, 2pj+1 of a
            // lane are chunks l4 and 4 + l4.  The residual is read through the transposed (full-line) addresses.  Its
            // reads are HBM round trips that nothing hides, and hipcc, with LDS-DMA in flight, puts vmcnt(0) in front
            // of every use of an ordinary load -- between the stores, which serialised them on the write latency (29k
            // cycles per tile measured).  So they are inline asm with counted waits (vmcnt retires in issue order):
            // one block (2 loads) per batch, four batches in flight.  Counted waits need every load and store to be
            // issued, so only tiles fully inside the matrix take that path; edge tiles predicate and wait for all.
            constexpr int NB = 16, AHEAD = EPI == VITHIP_BF16_EPI_F32_EMBED ? 2 : 4;  // pos_emb is cache-resident: short look-ahead, fewer registers
            f32x4 res[AHEAD][2];
            // running pointers (block order: (i, pj) = (0,0) (0,1) (1,0) ...; inside a block rows +0 and +8)
            const float *rbase = p.R + (size_t)(mw + row8) * p.ldr + nw + ch8 * 4;
            float *cbase = C + (size_t)(mw + row8) * p.ldc + nw + ch8 * 4;
            size_t roff = 0, coff = 0;
            const size_t r8 = (size_t)8 * p.ldr, c8 = (size_t)8 * p.ldc;
            // LNF producer: bf16 copy of the new rows and per-row partial sums over this wave's 64 columns (strip 4*tn + wc)
            [[maybe_unused]] float ps_a = 0.f, pq_a = 0.f, ps_b = 0.f, pq_b = 0.f;
            [[maybe_unused]] uint2 keep_a{0, 0}, keep_b{0, 0};
            // The residual epilogue addresses memory as (wave-uniform 64-bit base in SGPRs) + (one 32-bit byte offset per lane): the
            // block (m-tile, column pair) and the +8 row step move the SCALAR base, so a lane keeps one register per stream instead of
            // pointer, running offset and row step (6): that is what lets the look-ahead stay 4 blocks deep with the fold's extra
            // stores.  F32_EMBED keeps pointers (its rows are re-mapped per image).
            constexpr bool SADDR = EPI == VITHIP_BF16_EPI_F32_RESIDUAL;
            [[maybe_unused]] const char *r_s = reinterpret_cast<const char *>(p.R + (size_t)mw * p.ldr + nw);
            [[maybe_unused]] char *c_s = reinterpret_cast<char *>(C + (size_t)mw * p.ldc + nw);
            [[maybe_unused]] const unsigned r_l = (unsigned)(row8 * p.ldr + ch8 * 4) * 4u, c_l = (unsigned)(row8 * p.ldc + ch8 * 4) * 4u;
            [[maybe_unused]] char *x_s = nullptr, *p_s = nullptr;
            [[maybe_unused]] unsigned x_l = 0, x_lw = 0, p_l = 0;
            if constexpr (LNF) {
                x_s = reinterpret_cast<char *>(p.x16 + (size_t)mw * p.ldx16 + nw);
                p_s = reinterpret_cast<char *>(p.partials + ((size_t)(tn * 4 + wc) * p.M + mw) * 2);
                x_l = (unsigned)(row8 * p.ldx16 + ch8 * 4) * 2u;
                x_lw = x_l + ((ch8 & 1) ? 56u : 0u);  // 16-byte stores: even lanes at their own place in block 0, odd lanes 4 elements back in block 1
                p_l = (unsigned)row8 * 8u;
            }
            // scalar byte offset of block blk (m-tile blk >> 1, column pair blk & 1) and of its +8 rows, for a row stride ld and an element size
            auto blk_bytes = [](int blk, int ab, int ld, int esz) __attribute__((always_inline)) {
                return ((size_t)((blk >> 1) * 16 + ab * 8) * ld + (blk & 1) * 32) * esz;
            };
            const int m_left = p.M - (mw + row8), n_left = p.N - (nw + ch8 * 4);
            auto in_range = [&](int blk, int ab) __attribute__((always_inline)) {
                return (blk >> 1) * 16 + ab * 8 < m_left && (blk & 1) * 32 < n_left;
            };
            // F32_EMBED (patch embedding): GEMM row m = image * P + patch goes to token row m + image + 1 of x (row 0 of
            // every image is the class token), and the "residual" is pos_emb row patch + 1 (ViT_seq.c:52-101).  Two
            // running (patch, image) cursors, one for the look-ahead loads and one for the stores, advanced by 8 rows at
            // a time -- one division per tile instead of 64 hoisted ones.
            struct RowCursor { int pp, im; };
            [[maybe_unused]] auto step8 = [&](RowCursor c) __attribute__((always_inline)) {
                c.pp += 8;
                while (c.pp >= p.patches) {
                    c.pp -= p.patches;
                    ++c.im;
                }
                return c;
            };
            [[maybe_unused]] RowCursor lcur{0, 0}, scur{0, 0};
            if constexpr (EPI == VITHIP_BF16_EPI_F32_EMBED) {
                const int m = mw + row8;
                lcur.im = m / p.patches;
                lcur.pp = m - lcur.im * p.patches;
                scur = lcur;
            }
            auto load_res = [&](int blk, auto interior_c) __attribute__((always_inline)) {
                constexpr bool INTERIOR = decltype(interior_c)::value;
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) {
                    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float *ptr = rbase + roff + ab * r8;
                    if constexpr (EPI == VITHIP_BF16_EPI_F32_EMBED) {
                        const RowCursor c = ab ? step8(lcur) : lcur;
                        ptr = p.R + (size_t)(c.pp + 1) * p.ldr + nw + (blk & 1) * 32 + ch8 * 4;
                    }
                    if constexpr (SADDR) {
                        const char *sb = r_s + blk_bytes(blk, ab, p.ldr, 4);
                        if constexpr (INTERIOR) {
                            asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(r) : "v"(r_l), "s"(sb) : "memory");
                        } else if (in_range(blk, ab)) {
                            asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "+v"(r) : "v"(r_l), "s"(sb) : "memory");
                        }
                    } else if constexpr (INTERIOR) {
                        // asynchronous: the value is only valid after the counted wait below.  Safe only in straight-line
                        // code (no predication, so no compiler-made copies of `r` before the data has arrived).
                        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r) : "v"(ptr) : "memory");
                    } else if (in_range(blk, ab)) {
                        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(r) : "v"(ptr) : "memory");
                    }
                    res[blk % AHEAD][ab] = r;
                }
                if constexpr (!SADDR) {
                    roff += (blk & 1) ? 2 * r8 - 32 : 32;  // next block: other column pair, or the next m-tile
                    asm volatile("" : "+v"(roff));
                }
                if constexpr (EPI == VITHIP_BF16_EPI_F32_EMBED) {
                    if (blk & 1) lcur = step8(step8(lcur));
                    asm volatile("" : "+v"(lcur.pp), "+v"(lcur.im));
                }
            };
            // (Tried: adding the residual IN the accumulator layout first -- 32 loads of 16 rows x 64 B, 8 in flight, nothing but loads
            // in the queue -- and then transposing and storing the finished values, so that no load waits behind the stores of
            // earlier blocks (vmcnt retires in issue order).  Same bits; out_proj 0.72 -> 0.76 ms, fc2 unchanged: the epilogue is not
            // bound by that ordering but by its 512 KB through the 64-B/clk address path plus the LDS transposes.  Removed.)
            auto put_blk = [&](int blk) __attribute__((always_inline)) {
                const int i = blk >> 1, pj = blk & 1;
                put(blk, __builtin_bit_cast(u32x4, acc[i][2 * pj] + b4[2 * pj]), __builtin_bit_cast(u32x4, acc[i][2 * pj + 1] + b4[2 * pj + 1]));
            };
            auto run = [&](auto interior_c) __attribute__((always_inline)) {
                constexpr bool INTERIOR = decltype(interior_c)::value;
                u32x4 va, vb;
#pragma unroll
                for (int blk = 0; blk < AHEAD; ++blk) load_res(blk, interior_c);
                put_blk(0);
                get(0, va, vb);
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) {
                    if (blk + 1 < NB) put_blk(blk + 1);
                    // residual of this block: younger operations = 2 loads per block still ahead + the 2 stores of each
                    // block already stored since those loads were issued -> a constant 2 * (AHEAD - 1) + 2 * (AHEAD - 1)
                    // in steady state; simply wait for everything on edge tiles
                    if constexpr (INTERIOR) {
                        // stores a block issues: 2 (C); LNF, odd blocks: + 2 (bf16 copy of the m-tile) + 2 (its row sums)
                        auto stores_in = [](int b) constexpr { return 2 + ((LNF && (b & 1)) ? 4 : 0); };
                        const int younger_loads = 2 * ((blk + AHEAD - 1 < NB ? blk + AHEAD - 1 : NB - 1) - blk);
                        int younger_stores = 0;
                        for (int b = blk - AHEAD + 1 > 0 ? blk - AHEAD + 1 : 0; b < blk; ++b) younger_stores += stores_in(b);
                        switch (younger_loads + younger_stores) {
#define PP_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(res[blk % AHEAD][0]), "+v"(res[blk % AHEAD][1])::"memory"); break;
                            PP_VMW(0) PP_VMW(2) PP_VMW(4) PP_VMW(6) PP_VMW(8) PP_VMW(10) PP_VMW(12) PP_VMW(14) PP_VMW(16)
                            PP_VMW(18) PP_VMW(20) PP_VMW(22) PP_VMW(24) PP_VMW(26) PP_VMW(28) PP_VMW(30)
#undef PP_VMW
                            default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(res[blk % AHEAD][0]), "+v"(res[blk % AHEAD][1])::"memory"); break;
                        }
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(res[blk % AHEAD][0]), "+v"(res[blk % AHEAD][1])::"memory");
                    }
                    PP_LGKM0(va, vb);
                    const f32x4 ya = __builtin_bit_cast(f32x4, va) + res[blk % AHEAD][0];
                    const f32x4 yb = __builtin_bit_cast(f32x4, vb) + res[blk % AHEAD][1];
                    float *ca = cbase + coff, *cb = cbase + coff + c8;
                    if constexpr (EPI == VITHIP_BF16_EPI_F32_EMBED) {
                        const RowCursor c1 = step8(scur);
                        const int mrow = mw + (blk >> 1) * 16 + row8;
                        ca = C + (size_t)(mrow + scur.im + 1) * p.ldc + nw + (blk & 1) * 32 + ch8 * 4;
                        cb = C + (size_t)(mrow + 8 + c1.im + 1) * p.ldc + nw + (blk & 1) * 32 + ch8 * 4;
                        if (blk & 1) scur = step8(c1);
                        asm volatile("" : "+v"(scur.pp), "+v"(scur.im));
                    }
                    if constexpr (SADDR) {
                        if (INTERIOR || in_range(blk, 0))
                            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(c_l), "v"(ya), "s"(c_s + blk_bytes(blk, 0, p.ldc, 4)) : "memory");
                        if (INTERIOR || in_range(blk, 1))
                            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(c_l), "v"(yb), "s"(c_s + blk_bytes(blk, 1, p.ldc, 4)) : "memory");
                    } else {
                        if (INTERIOR || in_range(blk, 0)) *reinterpret_cast<f32x4 *>(ca) = ya;
                        if (INTERIOR || in_range(blk, 1)) *reinterpret_cast<f32x4 *>(cb) = yb;
                        coff += (blk & 1) ? 2 * c8 - 32 : 32;
                        asm volatile("" : "+v"(coff));
                    }
                    if constexpr (LNF) {
                        const bool in_a = INTERIOR || in_range(blk, 0), in_b = INTERIOR || in_range(blk, 1);
                        auto pk4 = [](f32x4 v) __attribute__((always_inline)) {
                            bf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                            return __builtin_bit_cast(uint2, o);
                        };
                        if constexpr (INTERIOR) {
                            // 16-byte stores: lane pairs (ch8 even / odd) trade pieces so that the even lane owns 8 consecutive bf16 of
                            // the m-tile's first 32-column block and the odd lane 8 of its second: per row 8 lanes x 16 B = one 128-B line
                            const uint2 ca = pk4(ya), cb = pk4(yb);
                            if ((blk & 1) == 0) {
                                keep_a = ca, keep_b = cb;
                            } else {
                                const bool odd = ch8 & 1;
                                auto swap2 = [](uint2 v) __attribute__((always_inline)) {
                                    return uint2{(unsigned)__builtin_amdgcn_mov_dpp((int)v.x, 0xB1, 0xf, 0xf, true),
                                                 (unsigned)__builtin_amdgcn_mov_dpp((int)v.y, 0xB1, 0xf, 0xf, true)};
                                };
                                const uint2 ra = swap2(odd ? keep_a : ca), rb = swap2(odd ? keep_b : cb);  // even lanes send their block-1 piece
                                const u32x4 oa = odd ? u32x4{ra.x, ra.y, ca.x, ca.y} : u32x4{keep_a.x, keep_a.y, ra.x, ra.y};
                                const u32x4 ob = odd ? u32x4{rb.x, rb.y, cb.x, cb.y} : u32x4{keep_b.x, keep_b.y, rb.x, rb.y};
                                asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(x_lw), "v"(oa), "s"(x_s + blk_bytes(blk - 1, 0, p.ldx16, 2)) : "memory");
                                asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(x_lw), "v"(ob), "s"(x_s + blk_bytes(blk - 1, 1, p.ldx16, 2)) : "memory");
                            }
                        } else {
                            if (in_a) asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(x_l), "v"(pk4(ya)), "s"(x_s + blk_bytes(blk, 0, p.ldx16, 2)) : "memory");
                            if (in_b) asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(x_l), "v"(pk4(yb)), "s"(x_s + blk_bytes(blk, 1, p.ldx16, 2)) : "memory");
                        }
                        // columns outside N add nothing; rows outside M are summed but never stored
                        const f32x4 za = in_a ? ya : f32x4{0.f, 0.f, 0.f, 0.f}, zb = in_b ? yb : f32x4{0.f, 0.f, 0.f, 0.f};
                        const float sa = (za[0] + za[1]) + (za[2] + za[3]), qa = (za[0] * za[0] + za[1] * za[1]) + (za[2] * za[2] + za[3] * za[3]);
                        const float sb = (zb[0] + zb[1]) + (zb[2] + zb[3]), qb = (zb[0] * zb[0] + zb[1] * zb[1]) + (zb[2] * zb[2] + zb[3] * zb[3]);
                        if ((blk & 1) == 0) {
                            ps_a = sa, pq_a = qa, ps_b = sb, pq_b = qb;
                        } else {  // both 32-column blocks of m-tile blk >> 1 are in: one (sum, sum of squares) pair per row
                            const int i16 = (blk >> 1) * 16;
                            const f32x2 ta = f32x2{sum8_dpp(ps_a + sa), sum8_dpp(pq_a + qa)}, tb = f32x2{sum8_dpp(ps_b + sb), sum8_dpp(pq_b + qb)};
                            if (ch8 == 0) {
                                if (INTERIOR || i16 < m_left) asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(p_l), "v"(ta), "s"(p_s + (size_t)i16 * 8) : "memory");
                                if (INTERIOR || i16 + 8 < m_left) asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(p_l), "v"(tb), "s"(p_s + (size_t)(i16 + 8) * 8) : "memory");
                            }
                        }
                    }
                    if (blk + 1 < NB) get(blk + 1, va, vb);
                    if (blk + AHEAD < NB) load_res(blk + AHEAD, interior_c);
                    __builtin_amdgcn_sched_barrier(0);  // keep the blocks apart: hoisting them together costs registers
                }
            };
            if (interior)
                run(std::true_type{});
            else
                run(std::false_type{});
            // (Tried: touching the next tile's residual lines from here, one dword per 128-B line, so that the next
            // epilogue's reads hit the caches.  vmcnt retires in order, so the first counted wait after the slack window
            // also waits for those prefetches: out_proj 646 -> 554-577 TFLOP/s, fc2 1020 -> 957.  Removed.)
        } else {
            bf16_t *C = static_cast<bf16_t *>(p.C);
            // Blocks: m-tile i = 16 rows x 64 bf16; paired mapping: accumulators (i, 2k), (i, 2k+1) of a lane are
            // features 32k + 8*l4 + [0,8), i.e. chunk 4k + l4.
            auto pack = [&](int i, u32x4 &c0, u32x4 &c1) __attribute__((always_inline)) {
                [[maybe_unused]] f32x4 r4, m4;
                if constexpr (LNF) {
#if PP_LNF_RS == 1
                    const f32x2 rs = rs_all[i];
#else
                    const f32x2 rs = *reinterpret_cast<const f32x2 *>(ln_rows_lds + i * 128);
#endif
                    r4 = f32x4{rs.x, rs.x, rs.x, rs.x}, m4 = f32x4{rs.y, rs.y, rs.y, rs.y};
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    bf16x8 ob;
                    f32x2 v[4];  // the chunk's four pairs: (jj, h) = (0,0) (0,1) (1,0) (1,1)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        f32x4 t4;
                        if constexpr (LNF) {  // LN(x).W^T + b = rstd * (x.Wf^T) - mean*rstd * colsum + b'
                            t4 = __builtin_elementwise_fma(acc[i][2 * k + jj], r4, b4[2 * k + jj] - m4 * s4[2 * k + jj]);
                        } else {
                            t4 = acc[i][2 * k + jj] + b4[2 * k + jj];
                        }
                        v[2 * jj] = f32x2{t4[0], t4[1]};
                        v[2 * jj + 1] = f32x2{t4[2], t4[3]};
                    }
                    if constexpr (EPI == VITHIP_BF16_EPI_BF16_GELU) {
#if PP_GELU_LOCKSTEP == 4
                        gelu_bf16_lockstep<4>(v);
#elif PP_GELU_LOCKSTEP == 2
                        {
                            f32x2 a[2] = {v[0], v[1]}, b[2] = {v[2], v[3]};
                            gelu_bf16_lockstep<2>(a);
                            gelu_bf16_lockstep<2>(b);
                            v[0] = a[0], v[1] = a[1], v[2] = b[0], v[3] = b[1];
                        }
#else
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_bf16_x2(v[e]);
#endif
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ob[2 * e] = (__bf16)v[e].x;
                        ob[2 * e + 1] = (__bf16)v[e].y;
                    }
                    (k ? c1 : c0) = __builtin_bit_cast(u32x4, ob);
                }
            };
            auto run = [&](auto interior_c) __attribute__((always_inline)) {
                constexpr bool INTERIOR = decltype(interior_c)::value;
                bf16_t *cbase = C + (size_t)(mw + row8) * p.ldc + nw + 8 * ch8;
                size_t off = 0;  // advanced by 8 rows per store (one running offset, not sixteen precomputed addresses)
                const size_t step8 = (size_t)8 * p.ldc;
                const int m_left = p.M - (mw + row8), n = nw + 8 * ch8;  // edge tiles: rows i*16 + ab*8 < m_left are inside
                u32x4 c0, c1, va, vb;
                pack(0, c0, c1);
                put(0, c0, c1);
                get(0, va, vb);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < 7) {
                        pack(i + 1, c0, c1);
                        put(i + 1, c0, c1);
                    }
                    PP_LGKM0(va, vb);
#pragma unroll
                    for (int ab = 0; ab < 2; ++ab) {
                        const u32x4 v = ab ? vb : va;
                        bf16_t *dst = cbase + off;
                        if (INTERIOR || (i * 16 + ab * 8 < m_left && n + 8 <= p.N)) {
                            // non-temporal: qkv / h are read by the NEXT launch, a gigabyte later -- kept out of the L2's way they cost
                            // QKV 0.6 % and fc1 1.6 % less (A/B of two libraries, batch 2048); the residual epilogue's stores gain nothing
                            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst));
                        } else if (i * 16 + ab * 8 < m_left && n < p.N) {  // ragged N (N % 8 == 4): first half of the chunk
                            *reinterpret_cast<uint2 *>(dst) = uint2{v.x, v.y};
                        }
                        off += step8;
                        asm volatile("" : "+v"(off));
                    }
                    if (i < 7) get(i + 1, va, vb);
                    __builtin_amdgcn_sched_barrier(0);  // keep the m-tiles apart: hoisting them together costs registers
                }
            };
            if (interior)
                run(std::true_type{});
            else
                run(std::false_type{});
        }
#undef PP_LGKM0
    };

    auto epilogue = [&](int e_tile, int e_tpar) __attribute__((always_inline)) {
        log_event(2);
        epilogue_body(e_tile, e_tpar);
        log_event(3);
    };

    // ---- prologue: half-tiles 0..5 (K step 0 complete, kinds 0 and 1 of K step 1)
    issue(K0{}, 0);
    issue(K1{}, 0);
    issue(K2{}, 0);
    issue(K3{}, 0);
    advance();
    issue(K0{}, 1);
    issue(K1{}, 1);
    wait_loads();  // kinds 0 and 1 of K step 0 have landed (four younger half-tiles may be in flight)
    PP_BARRIER();  // everybody: half-tiles 0 and 1 are visible

    if constexpr (STAMP == 1) stamps[20] = (unsigned)__builtin_amdgcn_s_memtime();
    in_loop = true;

    // ---- the four phases of a K step.  `cur` (LDS byte offset of the K step's parity) is refreshed per K step.
    const char *xb0, *xb1, *wb0, *wb1;
    auto kstep_bases = [&]() __attribute__((always_inline)) {
        const char *cur = lds + (par << 16);
        xb0 = cur + xbase;
        xb1 = cur + (xbase ^ 64);
        wb0 = cur + wbase;
        wb1 = cur + (wbase ^ 64);
    };
    auto reads0 = [&]() __attribute__((always_inline)) {  // W'0 (4 fragments) + X'0 (8 fragments)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wq0[j][0] = PP_LDS_FRAG(wb0 + 1 * SLOT + j * WJ);
            wq0[j][1] = PP_LDS_FRAG(wb1 + 1 * SLOT + j * WJ);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xq[i][0] = PP_LDS_FRAG(xb0 + 0 * SLOT + i * 2048);
            xq[i][1] = PP_LDS_FRAG(xb1 + 0 * SLOT + i * 2048);
        }
    };
    auto mfma0 = [&]() __attribute__((always_inline)) {  // quadrant (m0, n0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = PP_MFMA(wq0[j][ks], xq[i][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto phase0 = [&]() __attribute__((always_inline)) {
        reads0();
        issue(K2{}, par ^ 1);
        wait_loads();
        PP_STAMP(1);
        PP_BARRIER_L();
        PP_STAMP(2);
        mfma0();
        PP_STAMP(3);
        PP_BARRIER_M();
        PP_STAMP(4);
    };
    auto phase1 = [&]() __attribute__((always_inline)) {  // quadrant (m0, n1)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wq1[j][0] = PP_LDS_FRAG(wb0 + 2 * SLOT + j * WJ);
            wq1[j][1] = PP_LDS_FRAG(wb1 + 2 * SLOT + j * WJ);
        }
        issue(K3{}, par ^ 1);
        wait_loads();
        PP_STAMP(6);
        PP_BARRIER_L();
        PP_STAMP(7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][2 + j] = PP_MFMA(wq1[j][ks], xq[i][ks], acc[i][2 + j], 0, 0, 0);
            if (ks == 0) prepare_next();  // scalar work under the matrix pipe
        }
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(8);
        PP_BARRIER_M();
        PP_STAMP(9);
    };
    auto phase2 = [&]() __attribute__((always_inline)) {  // quadrant (m1, n1)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xq[i][0] = PP_LDS_FRAG(xb0 + 3 * SLOT + i * 2048);
            xq[i][1] = PP_LDS_FRAG(xb1 + 3 * SLOT + i * 2048);
        }
        advance();
        issue(K0{}, par);
        wait_loads();
        PP_STAMP(11);
        PP_BARRIER_L();
        PP_STAMP(12);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][2 + j] = PP_MFMA(wq1[j][ks], xq[i][ks], acc[4 + i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(13);
        PP_BARRIER_M();
        PP_STAMP(14);
    };
    auto phase3 = [&]() __attribute__((always_inline)) {  // quadrant (m1, n0); no fragment reads
        issue(K1{}, par);
        wait_loads();
        PP_STAMP(16);
        PP_BARRIER_L();
        PP_STAMP(17);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][j] = PP_MFMA(wq0[j][ks], xq[i][ks], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(18);
        PP_BARRIER_M();
        PP_STAMP(19);
    };

    // ---- tile loop.  Epilogues: group 1 runs its own right after its last phase (before the load section of its
    // next phase 0); group 0 DEFERS its by one barrier interval, to just before the MFMAs of its next phase 0 -- the
    // same interval, so the two epilogues overlap each other instead of each stalling the other group at a barrier
    // (measured before: two back-to-back epilogues of 5-10k cycles per 37k-cycle tile).  K step 0 of every tile is
    // peeled so that this happens in straight-line code and the accumulators are (re)defined unconditionally.
    bool have_prev = false;
    int prev_tile = 0, prev_tpar = 0;
    for (;;) {
        // ================= K step 0 =================
        log_event(1);
        kstep_bases();
        if (g == 0 && have_prev) {
            issue(K2{}, par ^ 1);
            wait_loads();
            PP_BARRIER_L();
            epilogue(prev_tile, prev_tpar);
            __builtin_amdgcn_sched_barrier(0);
            reads0();  // after the epilogue: its temporaries and these fragments do not fit together
        } else {
            reads0();
            issue(K2{}, par ^ 1);
            wait_loads();
            PP_BARRIER_L();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        mfma0();
        PP_BARRIER_M();
        phase1();
        phase2();
        phase3();
        par ^= 1;
        // ================= K steps 1 .. nk-1 =================
        for (int kt = 1; kt < nk; ++kt) {
            if constexpr (STAMP == 1) stamp_on = blockIdx.x == 0 && first_tile && kt == 3;
            PP_STAMP(0);
            log_event(1);
            kstep_bases();
            phase0();
            phase1();
            phase2();
            phase3();
            par ^= 1;
        }
        // ================= tile finished =================
        if constexpr (STAMP == 1) {
            if (first_tile) stamps[21] = (unsigned)__builtin_amdgcn_s_memtime();
        }
        if (g == 1) epilogue(c_tile, c_tpar);
        if constexpr (STAMP == 1) {
            if (first_tile) stamps[22] = (unsigned)__builtin_amdgcn_s_memtime();
            first_tile = false;
        }
        prev_tile = c_tile;
        prev_tpar = c_tpar;
        have_prev = true;
        c_tpar ^= 1;
        c_tile += nwg;
        if (c_tile >= total) break;
    }
    if (g == 0) epilogue(prev_tile, prev_tpar);  // group 0's last tile
    if constexpr (STAMP == 1) {
        stamps[23] = (unsigned)__builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0 && lane == 0)
            for (int k = 0; k < 24; ++k) p.dbg[wave * 32 + k] = stamps[k];
    }
    if constexpr (STAMP == 2) {
        log_event(4);
        if (blockIdx.x == 0) {
            unsigned *out = reinterpret_cast<unsigned *>(p.dbg);
            for (int k = lane; k < 512; k += 64)
                out[wave * 512 + k] = k < log_n ? *reinterpret_cast<unsigned *>(lds + LDS_BYTES + wave * 1024 + k * 4) : 0u;
        }
    }
}

}  // namespace

int launch_gemm_bf16_pp(hipStream_t s, const Bf16Params &p, int epilogue, int cus) {
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < cus ? total : cus), block(PTHREADS);  // one persistent workgroup per CU
    if (p.ln_rows || p.x16) {  // LayerNorm folded in (consumer: ln_rows + ln_colsum; producer: x16 + partials)
        switch (epilogue) {
            case VITHIP_BF16_EPI_BF16: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16, 0, 0, true>), grid, block, 0, s, p); break;
            case VITHIP_BF16_EPI_BF16_GELU: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16_GELU, 0, 0, true>), grid, block, 0, s, p); break;
            case VITHIP_BF16_EPI_F32_RESIDUAL: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_F32_RESIDUAL, 0, 0, true>), grid, block, 0, s, p); break;
            default: return static_cast<int>(hipErrorInvalidValue);
        }
        return static_cast<int>(hipGetLastError());
    }
    switch (epilogue) {
        case VITHIP_BF16_EPI_BF16: hipLaunchKernelGGL(gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16>, grid, block, 0, s, p); break;
        case VITHIP_BF16_EPI_BF16_GELU: hipLaunchKernelGGL(gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16_GELU>, grid, block, 0, s, p); break;
        case VITHIP_BF16_EPI_F32_RESIDUAL: hipLaunchKernelGGL(gemm_bf16_pp_kernel<VITHIP_BF16_EPI_F32_RESIDUAL>, grid, block, 0, s, p); break;
        case VITHIP_BF16_EPI_F32_EMBED: hipLaunchKernelGGL(gemm_bf16_pp_kernel<VITHIP_BF16_EPI_F32_EMBED>, grid, block, 0, s, p); break;
#ifdef VIT_PROBES  // timing-only (DBG) and stamped / event-log (STAMP) instantiations: probe build only
        case 201: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16, 0, 1>), grid, block, 0, s, p); break;
        case 202: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16, 0, 2>), grid, block, 0, s, p); break;
        case 203: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16, 0, 3>), grid, block, 0, s, p); break;
        case 204: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16, 0, 4>), grid, block, 0, s, p); break;
        case 300 + VITHIP_BF16_EPI_BF16: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16, 2>), grid, block, 0, s, p); break;
        case 300 + VITHIP_BF16_EPI_F32_RESIDUAL: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_F32_RESIDUAL, 2>), grid, block, 0, s, p); break;
        case 100 + VITHIP_BF16_EPI_BF16: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16, 1>), grid, block, 0, s, p); break;
        case 100 + VITHIP_BF16_EPI_BF16_GELU: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16_GELU, 1>), grid, block, 0, s, p); break;
        case 100 + VITHIP_BF16_EPI_F32_RESIDUAL: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_F32_RESIDUAL, 1>), grid, block, 0, s, p); break;
#endif
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

}  // namespace vitgemm
