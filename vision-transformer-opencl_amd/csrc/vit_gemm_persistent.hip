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
#include "vit_gemm_common.hpp"

namespace vitgemm {

constexpr int PBK = 32;           // K step
constexpr int PLD = PBK + 4;      // padded LDS row (floats)

template <int BM, int BN, int WM, int WN, int EPI, bool STAMP = false>
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
    const int my_tiles = (total - first + nwg - 1) / nwg;
    const int nk = p.K / PBK;

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

    // ---- load cursor: the (tile, k-step) the NEXT staging load will fetch ----------------------
    int tile_l = first, k_l = 0;
    set_sources(tile_l);
    auto advance_load_cursor = [&]() {
        if (++k_l == nk) {
            k_l = 0;
            tile_l += nwg;
            // past the last tile the old sources stay: harmless re-reads into a buffer nobody uses
            if (tile_l < total) set_sources(tile_l);
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
            *reinterpret_cast<f32x4 *>(As + (ld_row + q * ROWS_PER_PASS) * PLD + ld_kc) = a_stage[q];
            a_stage[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_src[q], k0 * 4, 0));
        } else {
            const int qb = q - A_CHUNKS;
            float *Bs = Bs0 + buf * BN * PLD;
            *reinterpret_cast<f32x4 *>(Bs + (ld_row + qb * ROWS_PER_PASS) * PLD + ld_kc) = b_stage[qb];
            b_stage[qb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_src[qb], k0 * 4, 0));
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

    // ---- compute cursor --------------------------------------------------------------------------
    int tile_c = first, k_c = 0, m0, n0;
    {
        int tm, tn;
        tile_coords(tile_c, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
        m0 = tm * BM;
        n0 = tn * BN;
    }
    float bias_r[TN];
    load_bias(n0, bias_r);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    // ---- prologue (once per workgroup, not per tile): steps 0 and 1 -----------------------------
    load_step();
    advance_load_cursor();
    store_step(0);
    load_step();
    advance_load_cursor();
    __syncthreads();
    read_frags(0, 0, 0);

    if constexpr (STAMP) st_loop0 = __builtin_amdgcn_s_memtime();
    int cur = 0;
    const int steps = my_tiles * nk;
    for (int g = 0; g < steps; ++g) {
        const int k_ahead = k_l * PBK;  // offset of the step the restage loads fetch (step g + 2)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC) read_frags(cur, c + 1, (c + 1) & 1);
            if (c == NC - 1) {
                // every wave has read buffer `cur` and written buffer `cur^1` (chunk 0): swap point.
                __syncthreads();
                read_frags(cur ^ 1, 0, NC & 1);  // first fragments of step g + 1 (maybe the next tile)
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
            if (c == 0) advance_load_cursor();  // scalar; re-bases the sources once per tile
        }
        cur ^= 1;

        if (++k_c == nk) {  // tile finished: store it, restart the accumulators, move on
            unsigned long long e0 = 0;
            if constexpr (STAMP) e0 = __builtin_amdgcn_s_memtime();
            epilogue_store<BM, BN, WM, WN, EPI, A_DENSE>(p, acc, bias_r, m0, n0, wm, wn, r, h);
            if constexpr (STAMP) st_epi += __builtin_amdgcn_s_memtime() - e0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;
            k_c = 0;
            tile_c += nwg;
            if (tile_c < total) {
                int tm, tn;
                tile_coords(tile_c, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
                m0 = tm * BM;
                n0 = tn * BN;
                load_bias(n0, bias_r);  // consumed a whole tile later
            }
        }
    }
    if constexpr (STAMP) {
        const unsigned long long c3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long *d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = st0; d[1] = st_loop0; d[2] = st_epi; d[3] = c3; d[4] = st_rt0; d[5] = r3;
            d[6] = (unsigned long long)my_tiles; d[7] = (unsigned long long)steps;
        }
    }
}

int g_persistent_wgs = 0;  // 2 per CU, queried once

template <int BM, int BN, int WM, int WN>
int launch_persistent_tile(hipStream_t stream, GemmParams &p, int epilogue, int group_m) {
    if (g_persistent_wgs == 0) {
        int dev = 0, cus = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return static_cast<int>(e);
        g_persistent_wgs = 2 * cus;
    }
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.group_m = group_m;
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < g_persistent_wgs ? total : g_persistent_wgs), block(256);
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
        default:
            return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

#ifdef VIT_PROBES
// Stamped probe build (tools/gemm_probe.py --stamp-tile 129): p.dbg receives 8 x u64 per workgroup.
int launch_persistent_stamped(hipStream_t stream, GemmParams &p, int epilogue, int group_m) {
    if (g_persistent_wgs == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return static_cast<int>(hipErrorInvalidDevice);
        g_persistent_wgs = 2 * cus;
    }
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = (p.N + 127) / 128;
    p.group_m = group_m;
    const int total = p.tiles_m * p.tiles_n;
    const dim3 grid(total < g_persistent_wgs ? total : g_persistent_wgs), block(256);
    if (epilogue == VITHIP_EPI_BIAS_GELU)
        hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, VITHIP_EPI_BIAS_GELU, true>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_nt_persistent_kernel<128, 128, 64, 64, VITHIP_EPI_BIAS, true>), grid, block, 0, stream, p);
    return static_cast<int>(hipGetLastError());
}

#endif

// Entry used by vit_gemm.hip's dispatcher.  Needs at least 4 K steps per tile (K >= 128).
int launch_persistent(hipStream_t stream, GemmParams &p, int epilogue, int group_m) {
    return launch_persistent_tile<128, 128, 64, 64>(stream, p, epilogue, group_m);
}

}  // namespace vitgemm
