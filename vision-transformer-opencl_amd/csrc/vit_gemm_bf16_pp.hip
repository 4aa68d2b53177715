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

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int PTHREADS = 512;
constexpr int SLOT = 128 * 128;               // one half-tile: 128 rows x 64 bf16
constexpr int RING = 8 * SLOT;                // 128 KB
constexpr int SCR_PITCH = 144;                // bytes per scratch row (16 rows x 64 bf16 + pad)
constexpr int SCR_WAVE = 16 * SCR_PITCH;      // 2304 B per wave
constexpr int BIAS_OFF = RING + 8 * SCR_WAVE; // two 1-KB bias slots (tile parity)
constexpr int LDS_BYTES = BIAS_OFF + 2 * 1024;

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
#define PP_STAMP(idx)                                                                                   \
    do {                                                                                                \
        if constexpr (STAMP) {                                                                          \
            if (stamp_on) stamps[idx] = (unsigned)__builtin_amdgcn_s_memtime();                                   \
        }                                                                                               \
    } while (0)

template <int EPI, bool STAMP = false>
__global__ __launch_bounds__(PTHREADS) void gemm_bf16_pp_kernel(const Bf16Params p) {
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;

    const int total = p.tiles_m * p.tiles_n, nwg = gridDim.x;
    const int first = xcd_remap(blockIdx.x, nwg);
    if (first >= total) return;  // workgroup-uniform
    const int nk = p.K / PBK;

    // ---- LDS-DMA source offsets: per half-tile two instructions (q) of 8 rows x 128 B.  Lane l lands at
    // (row L = 16*wave + 8q + l/8, chunk l%8) of the slot and fetches source chunk (l%8) ^ ((L>>1)&7).
    int xvoff[2][2], wvoff[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int L = wave * 16 + q * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((L >> 1) & 7);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int xrow = (L >> 6) * 128 + half * 64 + (L & 63);
            const int wrow = (L >> 5) * 64 + half * 32 + (L & 31);
            xvoff[half][q] = xrow * p.lda * 2 + chunk * 16;
            wvoff[half][q] = wrow * p.ldw * 2 + chunk * 16;
        }
    }

    // ---- load cursor (workgroup-uniform): the K step whose half-tiles are being issued
    int l_tile = first, l_kt = 0, l_tpar = 0;
    bool l_valid = true;
    __amdgpu_buffer_rsrc_t l_xr, l_wr, l_br;
    auto set_load_tile = [&]() {
        int tm, tn;
        tile_coords(l_tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
        const int m0 = tm * PBM, n0 = tn * PBN;
        const int mrows = p.M - m0 < PBM ? p.M - m0 : PBM, nrows = p.N - n0 < PBN ? p.N - n0 : PBN;
        l_xr = make_rsrc(p.A + (size_t)m0 * p.lda, (unsigned)mrows * p.lda * 2);  // rows past M read as zero
        l_wr = make_rsrc(p.W + (size_t)n0 * p.ldw, (unsigned)nrows * p.ldw * 2);
        l_br = make_rsrc(p.bias + n0, (unsigned)nrows * 4);
    };
    set_load_tile();
    auto advance = [&]() {
        if (++l_kt == nk) {
            l_kt = 0;
            l_tile += nwg;
            l_tpar ^= 1;
            l_valid = l_tile < total;
            if (l_valid) set_load_tile();
        }
    };
    // issue half-tile KIND of the cursor's K step into the slot of K-step parity `parbit`
    auto issue = [&](auto kind_c, int parbit) {
        constexpr int KIND = decltype(kind_c)::value;
        if (!l_valid) return;
        char *dst = lds + (parbit * 4 + KIND) * SLOT + wave * 2048;
        const int koff = l_kt * (PBK * 2);
        if constexpr (KIND == 1) {
            if (l_kt == 0 && wave == 0)  // the tile's bias row: lanes 0..63 x 16 B = 256 floats
                __builtin_amdgcn_raw_ptr_buffer_load_lds(l_br, (lds_void *)(lds + BIAS_OFF + l_tpar * 1024), 16, lane * 16, 0, 0, 0);
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
    auto wait_loads = [&]() {
        if (l_valid)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    const int wbase = (wc * 32 + l15) * 128 + c0;  // + kind*SLOT + j*2048

    // ---- compute cursor
    int c_tile = first, c_kt = 0, c_tpar = 0, par = 0;
    [[maybe_unused]] unsigned stamps[24];
    [[maybe_unused]] bool stamp_on = false;
    [[maybe_unused]] bool first_tile = true;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xq[4][2], wq0[2][2], wq1[2][2];

    // ---- prologue: half-tiles 0..5 (K step 0 complete, kinds 0 and 1 of K step 1)
    issue(K0{}, 0);
    issue(K1{}, 0);
    issue(K2{}, 0);
    issue(K3{}, 0);
    advance();
    issue(K0{}, 1);
    issue(K1{}, 1);
    wait_loads();  // kinds 0 and 1 of K step 0 have landed (four younger half-tiles may be in flight)
    PP_BARRIER();
    if (g == 1) PP_BARRIER();  // group 1 runs one barrier behind group 0 from here on

    if constexpr (STAMP) stamps[20] = (unsigned)__builtin_amdgcn_s_memtime();
    for (;;) {
        if constexpr (STAMP) stamp_on = blockIdx.x == 0 && first_tile && c_kt == 3;
        PP_STAMP(0);
        const char *cur = lds + (par << 16);
        const char *xb0 = cur + xbase, *xb1 = cur + (xbase ^ 64);
        const char *wb0 = cur + wbase, *wb1 = cur + (wbase ^ 64);

        // ================= phase 0: quadrant (m0, n0) =================
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wq0[j][0] = *reinterpret_cast<const bf16x8 *>(wb0 + 1 * SLOT + j * 2048);
            wq0[j][1] = *reinterpret_cast<const bf16x8 *>(wb1 + 1 * SLOT + j * 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xq[i][0] = *reinterpret_cast<const bf16x8 *>(xb0 + 0 * SLOT + i * 2048);
            xq[i][1] = *reinterpret_cast<const bf16x8 *>(xb1 + 0 * SLOT + i * 2048);
        }
        issue(K2{}, par ^ 1);
        wait_loads();
        PP_STAMP(1);
        PP_BARRIER();
        PP_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq0[j][ks], xq[i][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(3);
        PP_BARRIER();
        PP_STAMP(4);

        // ================= phase 1: quadrant (m0, n1) =================
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wq1[j][0] = *reinterpret_cast<const bf16x8 *>(wb0 + 2 * SLOT + j * 2048);
            wq1[j][1] = *reinterpret_cast<const bf16x8 *>(wb1 + 2 * SLOT + j * 2048);
        }
        issue(K3{}, par ^ 1);
        wait_loads();
        PP_STAMP(6);
        PP_BARRIER();
        PP_STAMP(7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq1[j][ks], xq[i][ks], acc[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(8);
        PP_BARRIER();
        PP_STAMP(9);

        // ================= phase 2: quadrant (m1, n1) =================
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xq[i][0] = *reinterpret_cast<const bf16x8 *>(xb0 + 3 * SLOT + i * 2048);
            xq[i][1] = *reinterpret_cast<const bf16x8 *>(xb1 + 3 * SLOT + i * 2048);
        }
        advance();
        issue(K0{}, par);
        wait_loads();
        PP_STAMP(11);
        PP_BARRIER();
        PP_STAMP(12);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq1[j][ks], xq[i][ks], acc[4 + i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(13);
        PP_BARRIER();
        PP_STAMP(14);

        // ================= phase 3: quadrant (m1, n0) =================
        issue(K1{}, par);
        wait_loads();
        PP_STAMP(16);
        PP_BARRIER();
        PP_STAMP(17);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq0[j][ks], xq[i][ks], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_STAMP(18);
        PP_BARRIER();
        PP_STAMP(19);

        par ^= 1;
        if (++c_kt < nk) continue;

        // ================= epilogue of tile c_tile (wave-private; no barrier) =================
        if constexpr (STAMP) {
            if (first_tile) stamps[21] = (unsigned)__builtin_amdgcn_s_memtime();
        }
        {
            int tm, tn;
            tile_coords(c_tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
            const int mw = tm * PBM + g * 128, nw = tn * PBN + wc * 64;  // this wave's 128 x 64 block
            f32x4 b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b4[j] = *reinterpret_cast<const f32x4 *>(lds + BIAS_OFF + c_tpar * 1024 + (wc * 64 + j * 16 + 4 * l4) * 4);
            if constexpr (EPI == VITHIP_BF16_EPI_F32_RESIDUAL) {
                float *C = static_cast<float *>(p.C);
                // The residual reads are HBM round trips that nothing hides (the accumulators are final only now), and
                // hipcc, with LDS-DMA in flight, puts vmcnt(0) in front of every use of an ordinary load -- between
                // the stores that serialised them on the write latency (29k cycles per tile measured).  So the loads
                // are inline asm with counted waits (vmcnt retires in issue order): four batches of 8 loads (2 m-tiles),
                // two batches in flight.
                // Counted waits need every load and store to be issued, so only tiles that lie fully inside
                // the matrix take this path; edge tiles predicate their accesses and wait for everything.
                f32x4 res[2][2][4];
                const bool interior = tm * PBM + PBM <= p.M && tn * PBN + PBN <= p.N;  // workgroup-uniform
                auto load_batch = [&](int b, auto interior_c) {  // m-tiles 2b, 2b+1 into res[b & 1]
                    constexpr bool INTERIOR = decltype(interior_c)::value;
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int m = mw + (2 * b + ii) * 16 + l15;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int n = nw + j * 16 + 4 * l4;
                            f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
                            const float *ptr = p.R + (size_t)m * p.ldr + n;
                            if (INTERIOR || (m < p.M && n < p.N))
                                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r) : "v"(ptr) : "memory");
                            res[b & 1][ii][j] = r;
                        }
                    }
                };
                auto store_batch = [&](int b, auto interior_c) {
                    constexpr bool INTERIOR = decltype(interior_c)::value;
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int i = 2 * b + ii;
                        const int m = mw + i * 16 + l15;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int n = nw + j * 16 + 4 * l4;
                            if (INTERIOR || (m < p.M && n < p.N))
                                *reinterpret_cast<f32x4 *>(C + (size_t)m * p.ldc + n) = acc[i][j] + b4[j] + res[b & 1][ii][j];
                        }
                    }
                };
#define PP_WAIT_RES(N, b)                                                                                             \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                            \
                 : "+v"(res[b][0][0]), "+v"(res[b][0][1]), "+v"(res[b][0][2]), "+v"(res[b][0][3]), "+v"(res[b][1][0]), \
                   "+v"(res[b][1][1]), "+v"(res[b][1][2]), "+v"(res[b][1][3])::"memory")
                if (interior) {
                    using T = std::true_type;
                    load_batch(0, T{});
                    load_batch(1, T{});
                    PP_WAIT_RES(8, 0);   // younger: the 8 loads of batch 1
                    store_batch(0, T{});
                    load_batch(2, T{});
                    PP_WAIT_RES(16, 1);  // younger: 8 stores + 8 loads
                    store_batch(1, T{});
                    load_batch(3, T{});
                    PP_WAIT_RES(16, 0);
                    store_batch(2, T{});
                    PP_WAIT_RES(8, 1);   // younger: the 8 stores of batch 2
                    store_batch(3, T{});
                } else {
                    using F = std::false_type;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        load_batch(b, F{});
                        if (b & 1)
                            PP_WAIT_RES(0, 1);
                        else
                            PP_WAIT_RES(0, 0);
                        store_batch(b, F{});
                    }
                }
#undef PP_WAIT_RES
            } else {
                bf16_t *C = static_cast<bf16_t *>(p.C);
                // wave-private transpose scratch, addressed with inline asm: hipcc puts `s_waitcnt vmcnt(0)` in
                // front of an ordinary LDS load that follows an LDS store while LDS-DMA is in flight (it cannot
                // tell the scratch from the DMA ring), which would drain the prefetch queue once per tile.
                const unsigned scr = (unsigned)(size_t)(lds_void *)(lds + RING + wave * SCR_WAVE);
                const unsigned scr_w = scr + l15 * SCR_PITCH + l4 * 8;                 // + j*32
                const unsigned scr_r = scr + (lane >> 3) * SCR_PITCH + (lane & 7) * 16;  // + h*8*SCR_PITCH
                // m-tile i: 4 x ds_write_b64 (lane's 4 features per column group) -> 2 x ds_read_b128 (8 features of one
                // token) -> 2 x 16-B stores; the bias/GELU/convert arithmetic of m-tile i+1 runs while the reads of
                // m-tile i are in flight (LDS executes a wave's operations in order, so write -> read needs no wait).
                auto pack = [&](int i, uint2(&ob)[4]) {
                    float y[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[j * 4 + q] = acc[i][j][q] + b4[j][q];
                    if constexpr (EPI == VITHIP_BF16_EPI_BF16_GELU) {
                        gelu_erf_x8(*reinterpret_cast<float(*)[8]>(&y[0]));
                        gelu_erf_x8(*reinterpret_cast<float(*)[8]>(&y[8]));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf16x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = (__bf16)y[j * 4 + q];
                        ob[j] = __builtin_bit_cast(uint2, o);
                    }
                };
                u32x4 v0, v1;
                auto write_read = [&](const uint2(&ob)[4]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(scr_w), "v"(ob[j]), "n"(j * 32) : "memory");
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3"
                                 : "=&v"(v0), "=&v"(v1)
                                 : "v"(scr_r), "n"(8 * SCR_PITCH)
                                 : "memory");
                };
                uint2 ob[4];
                pack(0, ob);
                write_read(ob);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < 7) pack(i + 1, ob);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1)::"memory");
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u32x4 v = h ? v1 : v0;
                        const int m = mw + i * 16 + h * 8 + (lane >> 3), n = nw + (lane & 7) * 8;
                        if (m < p.M && n + 8 <= p.N) {
                            *reinterpret_cast<u32x4 *>(C + (size_t)m * p.ldc + n) = v;
                        } else if (m < p.M && n < p.N) {  // ragged N (N % 8 == 4): first half of the chunk
                            *reinterpret_cast<uint2 *>(C + (size_t)m * p.ldc + n) = uint2{v.x, v.y};
                        }
                    }
                    if (i < 7) write_read(ob);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (STAMP) {
            if (first_tile) stamps[22] = (unsigned)__builtin_amdgcn_s_memtime();
            first_tile = false;
        }
        c_kt = 0;
        c_tpar ^= 1;
        c_tile += nwg;
        if (c_tile >= total) break;
    }
    if (g == 0) PP_BARRIER();  // pairs with group 1's last barrier
    if constexpr (STAMP) {
        stamps[23] = (unsigned)__builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0 && lane == 0)
            for (int k = 0; k < 24; ++k) p.dbg[wave * 32 + k] = stamps[k];
    }
}

}  // namespace

int launch_gemm_bf16_pp(hipStream_t s, const Bf16Params &p, int epilogue, int cus) {
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < cus ? total : cus), block(PTHREADS);  // one persistent workgroup per CU
    switch (epilogue) {
        case VITHIP_BF16_EPI_BF16: hipLaunchKernelGGL(gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16>, grid, block, 0, s, p); break;
        case VITHIP_BF16_EPI_BF16_GELU: hipLaunchKernelGGL(gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16_GELU>, grid, block, 0, s, p); break;
        case VITHIP_BF16_EPI_F32_RESIDUAL: hipLaunchKernelGGL(gemm_bf16_pp_kernel<VITHIP_BF16_EPI_F32_RESIDUAL>, grid, block, 0, s, p); break;
        case 100 + VITHIP_BF16_EPI_BF16: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16, true>), grid, block, 0, s, p); break;
        case 100 + VITHIP_BF16_EPI_BF16_GELU: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_BF16_GELU, true>), grid, block, 0, s, p); break;
        case 100 + VITHIP_BF16_EPI_F32_RESIDUAL: hipLaunchKernelGGL((gemm_bf16_pp_kernel<VITHIP_BF16_EPI_F32_RESIDUAL, true>), grid, block, 0, s, p); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

}  // namespace vitgemm
