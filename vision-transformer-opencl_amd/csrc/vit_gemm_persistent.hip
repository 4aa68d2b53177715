// csrc/vit_gemm_persistent.hip -- persistent, cross-tile pipelined variant of the fp32 NT GEMM.
//
// Same math, tile (BM x BN, K step 32) and fragment scheme as vit_gemm.hip.  What changes is the
// schedule: the grid is 2 workgroups per CU and each workgroup walks a strided list of tiles, treating
// (tile, k-step) as ONE flat sequence.  The staging pipeline (global -> registers -> LDS, two steps
// ahead) therefore runs straight through tile boundaries: while the last K steps of tile i multiply,
// the first K steps of tile i+1 are already being fetched, and the only non-matrix work left at a
// boundary is the store of the finished accumulators.  Measured per 128x128 tile of fc1 in the
// one-tile-per-workgroup kernel: 12.9k cycles of prologue (load latency, LDS fill) + ~10k of epilogue
// + ~5 % of workgroup launch gaps around 187k cycles of K loop; this kernel removes the prologue and
// the launch gaps.
//
// fp32 VALU instructions and v_mfma_f32_32x32x2_f32 share the SIMD's fp32 lanes (tools/valu_probe.py:
// 1 M partner-wave v_fma add 2.4 M cycles to an MFMA-bound loop), so every VALU instruction in the
// loop or the epilogue is paid for in matrix throughput; the bookkeeping below is scalar (SALU)
// wherever it is wave-uniform.
//
// Tile quantisation and the helper pieces (round 2).  fc2 / out_proj at batch 256 are 394 x 6 = 2,364 tiles on 512 workgroups:
// 4.62 rounds paid as 5 -- 316 workgroups own five tiles, 196 four, and for the length of one tile 38 % of the chip idles
// (7.7 % of the kernel; the stage model 0.923 x 0.95 matched the measured 0.877 of the clock-limited peak).  The owners of a
// fifth tile now hand the FIRST x K-steps of it to a four-tile workgroup: the helper runs that piece before its own tiles,
// parks the 64 KB of accumulators in a workspace slot and raises the slot's flag; the owner, at the very end of its walk, loads
// them as its INITIAL accumulators and continues with K-steps x..nk.  Every output still sums its k in the sequential order with
// the same instruction -- the hand-over moves an accumulation chain between workgroups, it does not split it -- so the result is
// bit-identical to the one-workgroup tile (tests/test_gpu_ops.py::test_gemm_tile_shapes_are_bit_identical), which plain
// split-K / stream-K fix-ups are not.  With c = ceil(owners / helpers) pieces per helper and x = floor(nk / (c + 1)) every workgroup
// ends within full x nk + c x steps: fc2 448 instead of 480 (ideal 443.25), out_proj 112 / 120, fc1 444 / 456.
// Dependencies: NOBODY WAITS (round 3).  A helper runs its pieces first thing and waits for nobody.  An owner looks at its
// slot's flag ONCE, three K-steps before its load cursor enters the extra tile (about 380 steps after the helper finished in the
// resident case): piece there -> its walk ends with [x, nk) from the parked accumulators; piece not there (helper not resident
// yet: another lane's or process's kernels hold its CU, ...) -> the owner withdraws the request (flag 0 -> 2) and runs the
// whole tile [0, nk) itself, the helper finds the 2 when it parks and clears it.  Either way every output sums its k in the
// same order: no code path stores a tile whose accumulators it did not wait for, nothing spins, nothing can time out, and
// the hand-over needs no co-residency of the grid (lanes > 1, shared devices).  The two outcomes are counted in the
// workspace (vithip_gemm_f32_workspace_stats): a recomputed piece costs x K-steps of one workgroup, never a wrong bit.
#include "vit_gemm_common.hpp"

namespace vitgemm {

constexpr int PBK = 32;           // K step
constexpr int PLD = PBK + 4;      // padded LDS row (floats)
// workspace: [SK_HEADER_BYTES of ints: flag per owner | two counters] [slots of 128 x 128 x 4 bytes]
constexpr int SK_HEADER_BYTES = 4096, SK_MAX_OWNERS = 1000, SK_STAT_TAKEN = 1016, SK_STAT_RECOMPUTED = 1017;

// SK: helper pieces compiled in (launches without them use the SK = false instantiation: no segment bookkeeping in its registers)
// DBG: timing-only switch-off instantiations for tools/gemm_f32_switchoff.py, results wrong by construction; instantiated in the
// probe build only (libvit_mi355x_probe.so, tile codes 131-136), the product library holds DBG = 0 alone: 1 no epilogue stores,
// 2 no staging loads inside the K loop, 3 no K-loop barrier, 4 no fragment reads, 5 no staging ds_writes, 6 = 2 + 5.
// (What such builds measure is mostly the POWER of frozen operand data, not the removed instructions: DESIGN 4.1 item 11.)
template <int BM, int BN, int WM, int WN, int EPI, bool STAMP = false, bool SK = false, int DBG = 0>
__global__ __launch_bounds__(256, 2) void gemm_f32_nt_persistent_kernel(const GemmParams p) {
    constexpr int ROWS_PER_PASS = 256 / (PBK / 4);
    constexpr int WGN = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_CHUNKS = BM * (PBK / 4) / 256;
    constexpr int B_CHUNKS = BN * (PBK / 4) / 256;
    constexpr int NC = PBK / 8;              // 8-deep chunks per K step
    constexpr int NS = A_CHUNKS + B_CHUNKS;  // staged float4 per thread per K step
    constexpr int NM = 4 * TM * TN;          // MFMAs per chunk
    static_assert((BM / WM) * WGN == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * PLD];
    __shared__ int sk_decision;  // owner: 1 = the helper's piece is there (written by thread 0, read by all after a barrier)
    // LayerNorm fold, consumer epilogues: (rstd, mean) of the tile's BM rows = 1 KB, copied here when the tile BEGINS by one
    // global -> LDS instruction of wave 0 (16 bytes per lane), written as inline assembly so that it stays off the compiler's
    // vmcnt book-keeping: a value fetched at the tile's start and consumed at its end -- in registers or through the
    // buffer-load-to-LDS builtin alike -- makes the compiler wait for EVERYTHING in flight at the point of use (it cannot count
    // the restage loads of a run-time number of K-steps in between): a drained staging pipeline once per tile, measured +0.8 us
    // per tile (fc1 +2 %).  vmcnt retires in order, so "at most 16 outstanding" at the top of the tile's LAST K-step means the
    // copy, at least three K-steps and 24 loads old by then, has landed; the barrier inside that step publishes it.  Two buffers,
    // alternating: wave 0 issues the next tile's copy as soon as ITS epilogue is done, while the other waves may still be reading
    // the finished tile's pairs; a buffer's next writer is two tiles (many barriers) later.  (An unknown younger operation in the
    // queue only makes the compiler's own counted waits wait for one load more than they need.)
    constexpr bool FOLD = EPI == EPI_BIAS_LN || EPI == EPI_BIAS_GELU_LN;
    __shared__ __attribute__((aligned(16))) f32x2 fold_rows_lds[FOLD ? 2 * BM : 2];
    static_assert(!FOLD || BM == 128, "one 16-byte lane copy of 64 lanes = 128 rows");
    float *const As0 = lds;
    float *const Bs0 = lds + 2 * BM * PLD;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    unsigned long long st0 = 0, st_rt0 = 0, st_loop0 = 0, st_epi = 0;
    if constexpr (STAMP) {
        st0 = __builtin_amdgcn_s_memtime();
        st_rt0 = __builtin_amdgcn_s_memrealtime();
    }

    const int total = p.tiles_m * p.tiles_n;
    const int nwg = gridDim.x;
    const int first = xcd_remap(blockIdx.x, nwg);  // XCD-mates take neighbouring tiles of every round
    if (first >= total) return;                    // workgroup-uniform
    const int nk = p.K / PBK;

    // ---- this workgroup's walk: segments (tile, k0, k1, partial in, partial out), all wave-uniform ----------------------
    //   helper (first >= R, pieces on):  its pieces [0, x) of the owners' extra tiles, then `full` whole tiles
    //   owner  (first <  R):             `full` whole tiles, then its extra tile: [x, nk) from the parked partial, or [0, nk)
    const int full = total / nwg, R = total - full * nwg;
    const int x = SK ? p.sk_x : 0;               // K-steps of an extra tile that its helper runs (0: no pieces)
    const int H = nwg - R;
    const bool owner = first < R;
    const int npre = (x > 0 && !owner && R > 0) ? (R - (first - R) + H - 1) / H : 0;  // owners first-R, first-R+H, ...
    // p.sk_late (tests): a helper runs its pieces AFTER its own tiles, i.e. too late for the owners' look at the flag
    const int pstart = (SK && p.sk_late) ? full : 0, fstart = (SK && p.sk_late) ? 0 : npre;
    const int nseg = npre + full + (owner ? 1 : 0);
    bool took = false;                           // owner: the parked piece is the start of its extra tile (set by decide())
    struct Seg { int tile, k0, k1, in, out; };
    auto get_seg = [&](int i) -> Seg {
        Seg g;
        if (i >= pstart && i < pstart + npre) {
            const int o = (first - R) + (i - pstart) * H;
            g.tile = full * nwg + o; g.k0 = 0; g.k1 = x; g.in = -1; g.out = o;
        } else if (i < npre + full) {
            g.tile = first + (i - fstart) * nwg; g.k0 = 0; g.k1 = nk; g.in = -1; g.out = -1;
        } else {
            g.tile = first + full * nwg; g.k0 = took ? x : 0; g.k1 = nk; g.in = took ? first : -1; g.out = -1;
        }
        return g;
    };

    const int ld_row = tid / (PBK / 4);
    const int ld_kc = (tid % (PBK / 4)) * 4;
    // Staging loads are buffer loads: SGPR descriptor + per-thread 32-bit byte offset + SGPR K offset, so
    // stepping through K costs no vector instruction (a 64-bit v_lshl_add per load otherwise -- and fp32
    // VALU time comes out of the matrix pipe's).  Offsets fit 32 bits: the largest operand (fc2's A at
    // batch 256) is 620 MB; the launcher refuses operands >= 2 GB for this kernel.
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.W), 0, 0x7fffffff, 0x00020000);
    int a_src[A_CHUNKS];  // byte offsets
    int b_src[B_CHUNKS];
    f32x4 a_stage[A_CHUNKS], b_stage[B_CHUNKS];

    auto set_sources = [&](int tile) {
        int tm, tn;
        tile_coords(tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            int m = tm * BM + ld_row + i * ROWS_PER_PASS;
            m = m < p.M ? m : p.M - 1;
            a_src[i] = (m * p.lda + ld_kc) * 4;
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            int n = tn * BN + ld_row + i * ROWS_PER_PASS;
            n = n < p.N ? n : p.N - 1;
            b_src[i] = (n * p.ldw + ld_kc) * 4;
        }
    };
    auto load_bias = [&](int n0, float (&dst)[TN]) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + r;
            dst[j] = n < p.N ? p.bias[n] : 0.0f;
        }
    };

    // ---- the owner's one look at its slot (workgroup-uniform: thread 0 decides, a barrier and an LDS word tell the rest) -----
    // flag o: (generation << 2) | state, state 1 = piece parked, 2 = request withdrawn; p.sk_gen numbers the launches that use
    // the workspace, so ONLY a word written by THIS launch means anything: whatever an aborted earlier launch left behind (a
    // stale "parked" included) reads as empty.  Relaxed agent-scope atomics on an uncached word; what orders the DATA against the
    // flag is on the helper's side (park()).  A 1 of this generation is only ever written behind this launch's data.
    int *const sk_flags = reinterpret_cast<int *>(p.sk_ws);
    const int SK_PARKED = (p.sk_gen << 2) | 1, SK_WITHDRAWN = (p.sk_gen << 2) | 2;
    int steps = 0;                               // K-steps of the whole walk (loop trip count; decide() may shorten it)
    auto decide = [&]() {
        if (tid == 0) {
            int *const f = sk_flags + first;
            int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (v != SK_PARKED) {  // not there: withdraw (whatever stale word is in the way); the helper may park in between
                int expected = v;
                if (__hip_atomic_compare_exchange_strong(f, &expected, SK_WITHDRAWN, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    v = SK_WITHDRAWN;
                    break;
                }
                v = expected;
            }
            // taken: the slot is free again as soon as this KERNEL ends (its next writer is a later launch on this stream)
            if (v == SK_PARKED) __hip_atomic_store(f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sk_flags + (v == SK_PARKED ? SK_STAT_TAKEN : SK_STAT_RECOMPUTED), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sk_decision = v == SK_PARKED;
        }
        __syncthreads();
        took = __builtin_amdgcn_readfirstlane(sk_decision) != 0;
        if (took) steps -= x;
    };

    // ---- load cursor: the (segment, k-step) the NEXT staging load will fetch ----------------------
    int seg_l = 0, k_l, kend_l;
    {
        const Seg g = get_seg(0);
        set_sources(g.tile);
        k_l = g.k0;
        kend_l = g.k1;
    }
    auto advance_load_cursor = [&]() {
        if (++k_l == kend_l) {
            // past the last segment the old sources stay: harmless re-reads into a buffer nobody uses
            if (++seg_l < nseg) {
                // the load cursor runs three K-steps ahead of the MFMAs: entering the owner's extra tile is where its first
                // K-step -- x or 0 -- has to be known
                if (SK && owner && x > 0 && seg_l == nseg - 1) decide();
                const Seg g = get_seg(seg_l);
                set_sources(g.tile);
                k_l = g.k0;
                kend_l = g.k1;
            } else {
                k_l = kend_l - 1;
            }
        }
    };
    auto load_step = [&]() {
        const int k0 = k_l * PBK;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) a_stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_src[i], k0 * 4, 0));
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) b_stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_src[i], k0 * 4, 0));
    };
    auto store_step = [&](int buf) {
        float *As = As0 + buf * BM * PLD, *Bs = Bs0 + buf * BN * PLD;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i)
            *reinterpret_cast<f32x4 *>(As + (ld_row + i * ROWS_PER_PASS) * PLD + ld_kc) = a_stage[i];
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i)
            *reinterpret_cast<f32x4 *>(Bs + (ld_row + i * ROWS_PER_PASS) * PLD + ld_kc) = b_stage[i];
    };
    // one staging slot: write the float4 fetched a step ago, then refetch it for two steps ahead
    auto restage_slot = [&](int q, int buf, int k0) {
        if (q < A_CHUNKS) {
            float *As = As0 + buf * BM * PLD;
            if (DBG != 5 && DBG != 6) *reinterpret_cast<f32x4 *>(As + (ld_row + q * ROWS_PER_PASS) * PLD + ld_kc) = a_stage[q];
            if (DBG != 2 && DBG != 6) a_stage[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_src[q], k0 * 4, 0));
        } else {
            const int qb = q - A_CHUNKS;
            float *Bs = Bs0 + buf * BN * PLD;
            if (DBG != 5 && DBG != 6) *reinterpret_cast<f32x4 *>(Bs + (ld_row + qb * ROWS_PER_PASS) * PLD + ld_kc) = b_stage[qb];
            if (DBG != 2 && DBG != 6) b_stage[qb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_src[qb], k0 * 4, 0));
        }
    };

    const int a_frag_off = (wm * WM + r) * PLD + h * 4;
    const int b_frag_off = (wn * WN + r) * PLD + h * 4;
    f32x4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int buf, int c, int set) {
        const float *As = As0 + buf * BM * PLD + a_frag_off + c * 8;
        const float *Bs = Bs0 + buf * BN * PLD + b_frag_off + c * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const f32x4 *>(As + i * 32 * PLD);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const f32x4 *>(Bs + j * 32 * PLD);
    };

    // ---- parked accumulators: slot o = 64 KB, float4 q of thread t at ((q * 256 + t) * 16) bytes --------------------------
    // Every access to a slot is a SYSTEM-scope (sc0 sc1) buffer access to UNCACHED device memory: the store is written
    // through to memory and acknowledged (vmcnt) once it is there, the load is served from memory, whichever XCD either side
    // runs on.  A release fence at agent scope would add buffer_wbl2 -- a write-back of the XCD's whole L2, there for ORDINARY
    // stores, of which the hand-over has none (measured with such fences: out_proj +22 %, the A / W panels of the XCD's 63
    // other workgroups went out with it).  What a release needs of these stores is that they have COMPLETED before the flag
    // goes up: every thread waits for its own (s_waitcnt vmcnt(0), written out: a workgroup-scope fence only emits
    // lgkmcnt(0) on gfx950), then the workgroup barrier, then thread 0's flag.
    f32x16 acc[TM][TN];
    constexpr int SC_SYS = 0x11;  // cache policy of the raw buffer builtins on gfx94x/95x: bit 0 = sc0, bit 4 = sc1
    auto slot_rsrc = [&](int o) {
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.sk_ws) + SK_HEADER_BYTES + (size_t)o * (BM * BN * 4), 0, BM * BN * 4, 0x00020000);
    };
    auto park = [&](int o) {  // helper: accumulators -> slot o, then the flag
        const __amdgpu_buffer_rsrc_t rs = slot_rsrc(o);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q4 + e];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, tid * 16,
                                                           ((i * TN + j) * 4 + q4) * 4096, SC_SYS);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            // anything but this launch's "withdrawn" -> parked.  Found the owner's withdrawal (it computes the tile itself): clear it
            int cur = __hip_atomic_load(sk_flags + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (cur == SK_WITHDRAWN) {
                    __hip_atomic_store(sk_flags + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                int expected = cur;
                if (__hip_atomic_compare_exchange_strong(sk_flags + o, &expected, SK_PARKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                cur = expected;  // only the owner writes besides us, and only once
            }
        }
    };
    auto unpark = [&](int o) {  // owner, after decide() saw the flag: the piece is its initial accumulators
        const __amdgpu_buffer_rsrc_t rs = slot_rsrc(o);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, ((i * TN + j) * 4 + q4) * 4096, SC_SYS));
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q4 + e] = v[e];
                }
        // The piece has LANDED before the K loop goes on -- said with the builtin, which the compiler's wait-count pass reads
        // (inline assembly it does not): accumulators "possibly still in flight" at the loop header made it put vmcnt(12) / (8) /
        // (5) / (1) in front of the first four MFMAs of EVERY K-step of the helper-piece instantiations, i.e. it drained the staging
        // queue at the top of each step where the plain walk only has its eight counted vmcnt(7).  Once per owner workgroup.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), nothing else
    };

    // ---- compute cursor --------------------------------------------------------------------------
    int seg_c = 0, k_c, kend_c, m0, n0, out_c;
    [[maybe_unused]] int seg_k0 = 0;  // first K-step of the compute cursor's segment
    float bias_r[TN];
    FoldOperands<TN, TM> fold{};
    [[maybe_unused]] int fold_buf = 0;  // which half of fold_rows_lds the compute cursor's tile uses
    auto begin_segment = [&](int i) {
        const Seg g = get_seg(i);
        int tm, tn;
        tile_coords(g.tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
        m0 = tm * BM;
        n0 = tn * BN;
        k_c = g.k0;
        seg_k0 = g.k0;
        kend_c = g.k1;
        out_c = g.out;
        if (g.out < 0) load_bias(n0, bias_r);  // consumed at the segment's end
        if constexpr (FOLD) {
            if (g.out < 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WN + j * 32 + r;
                    fold.colsum[j] = (p.ln_colsum && n < p.N) ? p.ln_colsum[n] : 0.0f;  // NULL: centred weights
                }
                if (wave == 0) {  // the rows' pairs -> LDS (see fold_rows_lds); rows past M read as zero, never stored
                    const int left = p.M - m0 < BM ? p.M - m0 : BM;
                    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ln_rows) + (size_t)m0 * 2, 0, left * 8, 0x00020000);
                    const unsigned dst = (unsigned)(size_t)(fold_rows_lds + fold_buf * BM);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(lane * 16), "s"(rr) : "memory", "m0");
                }
                fold.rows = fold_rows_lds + fold_buf * BM;
                fold_buf ^= 1;
            }
        }
        if (SK && g.in >= 0) {
            unpark(g.in);
        } else {
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[ii][jj][v] = 0.0f;
        }
    };
    begin_segment(0);

    for (int i = 0; i < nseg; ++i) {  // (an owner's extra tile counts in full until decide() says otherwise)
        const Seg g = get_seg(i);
        steps += g.k1 - g.k0;
    }
    // ---- prologue (once per workgroup, not per tile): steps 0 and 1 -----------------------------
    load_step();
    advance_load_cursor();
    store_step(0);
    load_step();
    advance_load_cursor();
    __syncthreads();
    read_frags(0, 0, 0);
    if (DBG == 4) read_frags(0, 1, 1);

    if constexpr (STAMP) st_loop0 = __builtin_amdgcn_s_memtime();
    int cur = 0;
    for (int g = 0; g < steps; ++g) {
        const int k_ahead = k_l * PBK;  // offset of the step the restage loads fetch (step g + 2)
        if constexpr (FOLD) {
            if (k_c + 1 == kend_c && out_c < 0) {  // the tile's last step: the copy of its rows' pairs has landed (see fold_rows_lds)
                // vmcnt retires in order: once at most FOLD_WAIT = 2 * NS operations are outstanding, everything older than the
                // restage loads of the last two K-steps is done.  The copy was issued before this tile's first restage load, and a
                // tile of FOLD_MIN_STEPS steps has issued (FOLD_MIN_STEPS - 1) * NS >= FOLD_WAIT + NS of them by now.
                constexpr int FOLD_WAIT = 2 * NS, FOLD_MIN_STEPS = 4;
                static_assert(FOLD_WAIT <= 63, "vmcnt is a 6-bit count on gfx950");
                static_assert((FOLD_MIN_STEPS - 1) * NS >= FOLD_WAIT + NS, "the copy must be older than the loads the wait leaves in flight");
                if (kend_c - seg_k0 >= FOLD_MIN_STEPS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FOLD_WAIT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC && DBG != 4) read_frags(cur, c + 1, (c + 1) & 1);
            if (c == NC - 1) {
                // every wave has read buffer `cur` and written buffer `cur^1` (chunk 0): swap point.
                if (DBG != 3) __syncthreads();
                if (DBG != 4) read_frags(cur ^ 1, 0, NC & 1);  // first fragments of step g + 1 (maybe the next tile)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < NM; ++idx) {
                const int s2 = idx / (TM * TN), i = (idx / TN) % TM, j = idx % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c & 1][i][s2], bf[c & 1][j][s2], acc[i][j], 0, 0, 0);
                if (c == 0) {
                    // the NS restage slots spread evenly between the MFMAs of chunk 0
                    const int slot_before = (idx * NS) / NM, slot_after = ((idx + 1) * NS) / NM;
                    if (slot_after > slot_before) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = slot_before; q < slot_after; ++q) restage_slot(q, cur ^ 1, k_ahead);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (c == 0) advance_load_cursor();  // scalar; re-bases the sources once per segment
        }
        cur ^= 1;

        if (++k_c == kend_c) {  // segment finished: store the tile (or park the piece), start the next one
            unsigned long long e0 = 0;
            if constexpr (STAMP) e0 = __builtin_amdgcn_s_memtime();
            if (SK && out_c >= 0) park(out_c);
            else if (DBG != 1 || p.M < 0) {  // (M < 0: never; keeps the MFMAs alive)
                if constexpr (EPI == EPI_RESIDUAL_STATS) epilogue_store_residual_stats<BM, BN, WM, WN>(p, acc, bias_r, m0, n0, wm, wn, r, h);
                else epilogue_store<BM, BN, WM, WN, EPI, A_DENSE>(p, acc, bias_r, m0, n0, wm, wn, r, h, fold);
            }
            if constexpr (STAMP) st_epi += __builtin_amdgcn_s_memtime() - e0;
            if (++seg_c < nseg) begin_segment(seg_c);
        }
    }
    if constexpr (STAMP) {
        const unsigned long long c3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long *d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = st0; d[1] = st_loop0; d[2] = st_epi; d[3] = c3; d[4] = st_rt0; d[5] = r3;
            d[6] = (unsigned long long)nseg; d[7] = (unsigned long long)steps;
        }
    }
}

// 2 workgroups per CU of the CURRENT device (engines of several devices share the process: nothing is cached process-wide)
static int persistent_wgs() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return 2 * cus;
}

// K-steps of a last-round tile that a helper workgroup runs (0: the hand-over does not pay) for the 128 x 128 persistent walk
// on a grid of `wgs` workgroups (0: ask the current device) with `slots` parking slots
int persistent_piece_steps(int M, int N, int K, int slots, int wgs) {
    if (wgs <= 0) wgs = persistent_wgs();
    if (wgs <= 0) return 0;
    const int total = ((M + 127) / 128) * ((N + 127) / 128);
    const int nwg = total < wgs ? total : wgs;
    const int full = total / nwg, R = total - full * nwg, nk = K / PBK;
    if (full < 1 || R <= 0 || R >= nwg || R > slots || R > SK_MAX_OWNERS) return 0;
    const int c = (R + (nwg - R) - 1) / (nwg - R);
    const int x = nk / (c + 1);
    // a hand-over costs about 3 K-steps (64 KB out, 64 KB in through uncached memory, the flag): worth it when the walk gets
    // at least 8 steps shorter (measured at batch 256: fc2 -5 %, out_proj -5 % (nk = 24, 8 of 120 saved), fc1 -1.7 %;
    // QKV would save 6 of 336 with six 3-step pieces per helper: off)
    return (x >= 4 && nk - c * x >= 8) ? x : 0;
}

template <int BM, int BN, int WM, int WN>
int launch_persistent_tile(hipStream_t stream, GemmParams &p, int epilogue, int group_m) {
    const int wgs = persistent_wgs();
    if (wgs <= 0) return static_cast<int>(hipErrorInvalidDevice);
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.group_m = group_m;
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < wgs ? total : wgs), block(256);
    // helper pieces (see the head of this file): on when the caller lent a workspace, a partial last round exists and the
    // piece is long enough to be worth a 64 KB hand-over (>= 4 K-steps)
    p.sk_x = (p.sk_ws && BM == 128 && BN == 128) ? persistent_piece_steps(p.M, p.N, p.K, p.sk_slots, wgs) : 0;
    if (p.sk_x > 0) {
        switch (epilogue) {
            case VITHIP_EPI_BIAS:
                hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS, false, true>), grid, block, 0, stream, p);
                break;
            case VITHIP_EPI_BIAS_GELU:
                hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS_GELU, false, true>), grid, block, 0, stream, p);
                break;
            case VITHIP_EPI_BIAS_RESIDUAL:
                hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS_RESIDUAL, false, true>), grid, block, 0, stream, p);
                break;
            case EPI_BIAS_LN:
                hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, EPI_BIAS_LN, false, true>), grid, block, 0, stream, p);
                break;
            case EPI_BIAS_GELU_LN:
                hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, EPI_BIAS_GELU_LN, false, true>), grid, block, 0, stream, p);
                break;
            case EPI_RESIDUAL_STATS:
                hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, EPI_RESIDUAL_STATS, false, true>), grid, block, 0, stream, p);
                break;
            default:
                return static_cast<int>(hipErrorInvalidValue);
        }
        return static_cast<int>(hipGetLastError());
    }
    switch (epilogue) {
        case VITHIP_EPI_BIAS:
            hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS>), grid, block, 0, stream, p);
            break;
        case VITHIP_EPI_BIAS_GELU:
            hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS_GELU>), grid, block, 0, stream, p);
            break;
        case VITHIP_EPI_BIAS_RESIDUAL:
            hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, VITHIP_EPI_BIAS_RESIDUAL>), grid, block, 0, stream, p);
            break;
        case EPI_BIAS_LN:
            hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, EPI_BIAS_LN>), grid, block, 0, stream, p);
            break;
        case EPI_BIAS_GELU_LN:
            hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, EPI_BIAS_GELU_LN>), grid, block, 0, stream, p);
            break;
        case EPI_RESIDUAL_STATS:
            hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<BM, BN, WM, WN, EPI_RESIDUAL_STATS>), grid, block, 0, stream, p);
            break;
        default:
            return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

#ifdef VIT_PROBES
// Stamped probe build (tools/gemm_probe.py --stamp-tile 129): p.dbg receives 8 x u64 per workgroup.
int launch_persistent_stamped(hipStream_t stream, GemmParams &p, int epilogue, int group_m) {
    const int wgs = persistent_wgs();
    if (wgs <= 0) return static_cast<int>(hipErrorInvalidDevice);
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = (p.N + 127) / 128;
    p.group_m = group_m;
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < wgs ? total : wgs), block(256);
    if (epilogue == VITHIP_EPI_BIAS_GELU)
        hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, VITHIP_EPI_BIAS_GELU, true>), grid, block, 0, stream, p);
    else if (epilogue == EPI_BIAS_GELU_LN)  // the LayerNorm fold's consumer epilogues (tools/gemm_f32_fold_stamps.py)
        hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, EPI_BIAS_GELU_LN, true>), grid, block, 0, stream, p);
    else if (epilogue == EPI_BIAS_LN)
        hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, EPI_BIAS_LN, true>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, VITHIP_EPI_BIAS, true>), grid, block, 0, stream, p);
    return static_cast<int>(hipGetLastError());
}


// Switch-off builds (tools/gemm_f32_switchoff.py; tile codes 131-136 of the probe library): the persistent walk without helper
// pieces and with one part of the kernel removed.  Timing only.
template <int DBG>
static int launch_switchoff_dbg(hipStream_t stream, const GemmParams &p, int epilogue, dim3 grid) {
    switch (epilogue) {
        case VITHIP_EPI_BIAS: hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, VITHIP_EPI_BIAS, false, false, DBG>), grid, dim3(256), 0, stream, p); break;
        case VITHIP_EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, VITHIP_EPI_BIAS_GELU, false, false, DBG>), grid, dim3(256), 0, stream, p); break;
        case VITHIP_EPI_BIAS_RESIDUAL: hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, VITHIP_EPI_BIAS_RESIDUAL, false, false, DBG>), grid, dim3(256), 0, stream, p); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}
int launch_persistent_switchoff(hipStream_t stream, GemmParams &p, int epilogue, int group_m, int dbg) {
    const int wgs = persistent_wgs();
    if (wgs <= 0) return static_cast<int>(hipErrorInvalidDevice);
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = (p.N + 127) / 128;
    p.group_m = group_m;
    p.sk_x = 0;
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < wgs ? total : wgs);
    switch (dbg) {
        case 1: return launch_switchoff_dbg<1>(stream, p, epilogue, grid);
        case 2: return launch_switchoff_dbg<2>(stream, p, epilogue, grid);
        case 3: return launch_switchoff_dbg<3>(stream, p, epilogue, grid);
        case 4: return launch_switchoff_dbg<4>(stream, p, epilogue, grid);
        case 5: return launch_switchoff_dbg<5>(stream, p, epilogue, grid);
        case 6: return launch_switchoff_dbg<6>(stream, p, epilogue, grid);
        default: return static_cast<int>(hipErrorInvalidValue);
    }
}
#endif

// Entry used by vit_gemm.hip's dispatcher.  Needs at least 4 K steps per tile (K >= 128).
int launch_persistent(hipStream_t stream, GemmParams &p, int epilogue, int group_m) {
    // residual GEMM whose caller also wants the row statistics of what it stores: in the epilogue when the columns are whole tiles
    if (epilogue == VITHIP_EPI_BIAS_RESIDUAL && p.row_partials && p.N % 128 == 0) {
        epilogue = EPI_RESIDUAL_STATS;
        p.stats_in_epilogue = 1;
    }
    return launch_persistent_tile<128, 128, 64, 64>(stream, p, epilogue, group_m);
}

}  // namespace vitgemm
