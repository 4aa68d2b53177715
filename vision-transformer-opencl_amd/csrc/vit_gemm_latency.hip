// csrc/vit_gemm_latency.hip -- fp32 NT GEMM for the latency regime: 16x16 outputs per wave on v_mfma_f32_16x16x4_f32.
//
// When a GEMM has fewer 64x64 tiles than the chip has CUs (one image: fc2 = 48 tiles on 256 CUs), its time is ONE wave's
// accumulation chain: K/2 dependent v_mfma_f32_32x32x2_f32 of 64 cycles (fc2: 98k cycles = 41 us; 0.73 of the 1.84 ms of an image).
// A chain cannot be cut without changing the rounding -- but it can be made of shorter links: v_mfma_f32_16x16x4_f32 adds FOUR
// products per instruction with a dependent-issue distance of 40 cycles, i.e. 10 cycles per k instead of 32, for a quarter of
// the outputs -- and a 16x16 wave tile means four times the waves, which is exactly what an under-filled chip has room for.
//
// Bit-identical to every other fp32 GEMM kernel here: tools/probes/mfma_order_probe.hip shows that both instructions add their
// products one after the other in k order with one fp32 rounding each (both equal a chain of v_fma_f32), so only the ORDER of the
// k matters.  The 32x32x2 kernels read a lane's four k as one 16-byte LDS word at 8c + 4h (h = lane >> 5), which makes their order
// inside every chunk of eight 0 4 1 5 2 6 3 7; here lane group g = lane >> 4 supplies k = 8c + {0,4,1,5}[g] to the first MFMA of
// the chunk and that + 2 to the second (one ds_read2_b32 per operand and chunk).
// tests/test_gpu_ops.py::test_gemm_tile_shapes_are_bit_identical covers it (tile 12).
//
// Shape: workgroup = 4 waves = a 32x32 tile; K step 128 through a double-buffered LDS tile, global loads one K step ahead in
// registers; one barrier per K step = per 64 MFMAs.  No persistent walk, no XCD games: these launches have a few hundred
// workgroups and live in the L2s.
// LDS image.  A lane reads single floats (k = 8c + 4 (g & 1) + (g >> 1), and + 2), 32 lanes per LDS cycle -- all at the SAME
// element of their 16-byte word, so in any layout made of plain float4 they can reach only 8 of the 32 banks.  Rows are therefore
// 136 floats apart (34 words of 16 B: rows r, r+1, r+2, r+3 are two words apart, with the k group's own word in between) and
// the four floats of every word are ROTATED by (row >> 2) & 3 on the way in: (row & 3, g & 1) picks the word, row >> 2 the
// element -- 16 rows x 2 groups on 32 different banks.  The rotation costs four v_cndmask per staged float4, on a VALU that the
// 40-cycle dependent MFMA chain leaves idle.
#include <type_traits>

#include "vit_gemm_common.hpp"

namespace vitgemm {

namespace {

constexpr int LBM = 32, LBN = 32, LBK = 128, LLD = LBK + 8;
constexpr int LCH = (LBM * LBK / 4) / 256;  // float4 per thread and operand and K step (= 4)

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_nt_latency_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (LBM + LBN) * LLD];
    float *const As0 = lds;
    float *const Bs0 = lds + 2 * LBM * LLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
    const int m0 = tm * LBM, n0 = tn * LBN;

    // staging: thread t loads float4 number q (0..3) of row (t >> 5) + 8 q ... 32 threads cover the 128 floats of a row
    const int ld_row = tid >> 5, ld_kc = (tid & 31) * 4;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.W), 0, 0x7fffffff, 0x00020000);
    int a_off[LCH], b_off[LCH];
#pragma unroll
    for (int q = 0; q < LCH; ++q) {
        int m = m0 + ld_row + 8 * q, n = n0 + ld_row + 8 * q;
        m = m < p.M ? m : p.M - 1;  // rows past the edge: clamped copies, never stored
        n = n < p.N ? n : p.N - 1;
        a_off[q] = (m * p.lda + ld_kc) * 4;
        b_off[q] = (n * p.ldw + ld_kc) * 4;
    }
    // two staging register sets: the tile of K step t + 1 waits in one for its turn in LDS while the loads of t + 2 fill the other
    // (about one and a half K steps = 2,000 cycles between a load and its use: an L2 / Infinity Cache round trip)
    f32x4 a_st[2][LCH], b_st[2][LCH];
    auto load_global = [&](int set, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < LCH; ++q) {
            a_st[set][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_off[q], k0 * 4, 0));
            b_st[set][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_off[q], k0 * 4, 0));
        }
    };
    const bool rot1 = (ld_row >> 2) & 1;  // row = ld_row + 8 q: (row >> 2) & 3 = (rot1 + 2 q) & 3
    auto rotated = [&](f32x4 v, int q) __attribute__((always_inline)) {  // stored[(i + s) & 3] = v[i], s = (row >> 2) & 3
        const f32x4 t = (q & 1) ? f32x4{v[2], v[3], v[0], v[1]} : v;     // by 2 q: known at compile time
        return rot1 ? f32x4{t[3], t[0], t[1], t[2]} : t;
    };
    auto store_lds = [&](int set, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < LCH; ++q) {
            *reinterpret_cast<f32x4 *>(As0 + buf * LBM * LLD + (ld_row + 8 * q) * LLD + ld_kc) = rotated(a_st[set][q], q);
            *reinterpret_cast<f32x4 *>(Bs0 + buf * LBN * LLD + (ld_row + 8 * q) * LLD + ld_kc) = rotated(b_st[set][q], q);
        }
    };

    const int n = n0 + 16 * wn + n16;
    const float bias = n < p.N ? p.bias[n] : 0.0f;
    // LayerNorm fold (consumer): the column sum and the lane's four rows' (rstd, mean), fetched now like the bias -- this kernel's
    // time IS its dependent chain, a load in the epilogue would be on it
    [[maybe_unused]] float colsum = 0.0f;
    [[maybe_unused]] f32x2 rowpair[4] = {};
    if constexpr (EPI == EPI_BIAS_LN || EPI == EPI_BIAS_GELU_LN) {
        colsum = (p.ln_colsum && n < p.N) ? p.ln_colsum[n] : 0.0f;  // NULL: centred weights
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int m = m0 + 16 * wm + 4 * g + v;
            rowpair[v] = reinterpret_cast<const f32x2 *>(p.ln_rows)[m < p.M ? m : p.M - 1];
        }
    }
    // fragment elements of this lane inside a chunk of eight k: k = {0, 4, 1, 5}[g] for the chunk's first MFMA, + 2 for its
    // second; in LDS: word 2c + (g & 1) of the row, elements rotated by (row >> 2) & 3 = (n16 >> 2) & 3
    const int pos1 = ((g >> 1) + (n16 >> 2)) & 3, pos2 = (pos1 + 2) & 3;
    const int a_frag = (16 * wm + n16) * LLD + 4 * (g & 1), b_frag = (16 * wn + n16) * LLD + 4 * (g & 1);

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / LBK;
    load_global(0, 0);
    store_lds(0, 0);
    if (nk > 1) load_global(1, LBK);
    if (nk > 2) load_global(0, 2 * LBK);
    __syncthreads();
    constexpr int NCH = LBK / 8, AHEAD = 3;  // chunks of eight k per K step; fragment reads run AHEAD chunks before their MFMAs
    auto step = [&](int kt, auto set_c) __attribute__((always_inline)) {  // K step kt; `set` holds the tile of kt + 1
        constexpr int SET = decltype(set_c)::value;
        const int cur = kt & 1;
        const float *As = As0 + cur * LBM * LLD + a_frag, *Bs = Bs0 + cur * LBN * LLD + b_frag;
        float a0[AHEAD + 1], a1[AHEAD + 1], b0[AHEAD + 1], b1[AHEAD + 1];
        auto read = [&](int c) __attribute__((always_inline)) {
            a0[c % (AHEAD + 1)] = As[8 * c + pos1], a1[c % (AHEAD + 1)] = As[8 * c + pos2];
            b0[c % (AHEAD + 1)] = Bs[8 * c + pos1], b1[c % (AHEAD + 1)] = Bs[8 * c + pos2];
        };
#pragma unroll
        for (int c = 0; c < AHEAD; ++c) read(c);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + AHEAD < NCH) read(c + AHEAD);
            __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD chunks in front: left alone hipcc sinks them next to their use
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c % (AHEAD + 1)], b0[c % (AHEAD + 1)], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c % (AHEAD + 1)], b1[c % (AHEAD + 1)], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (c == 3 && kt + 1 < nk) {  // the next K step into the other LDS buffer, and its registers off to K step kt + 3
                store_lds(SET, cur ^ 1);
                if (kt + 3 < nk) load_global(SET, (kt + 3) * LBK);
            }
        }
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, std::integral_constant<int, 1>{});
        if (kt + 1 < nk) step(kt + 1, std::integral_constant<int, 0>{});
    }

    // accumulator register v of lane (n16, g): C[m0 + 16 wm + 4 g + v][n]
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int m = m0 + 16 * wm + 4 * g + v;
        if (m < p.M && n < p.N) {
            float y;
            if constexpr (EPI == EPI_BIAS_LN || EPI == EPI_BIAS_GELU_LN)  // LayerNorm fold, consumer (vit_gemm_common.hpp)
                y = fold_scale(fold_center(acc[v], rowpair[v].y, colsum), rowpair[v].x, bias);
            else y = acc[v] + bias;
            if constexpr (EPI == VITHIP_EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_LN) y = gelu_erf(y);
            if constexpr (EPI == VITHIP_EPI_BIAS_RESIDUAL) y += p.R[(size_t)m * p.ldr + n];
            p.C[(size_t)m * p.ldc + n] = y;
        }
    }
}

}  // namespace

// K % 128 == 0 (the caller falls back to the 64x64 tile otherwise); operands below 2 GiB as for every fp32 kernel.
int launch_gemm_f32_latency(hipStream_t stream, GemmParams &p, int epilogue) {
    p.tiles_m = (p.M + LBM - 1) / LBM;
    p.tiles_n = (p.N + LBN - 1) / LBN;
    const dim3 grid(p.tiles_m * p.tiles_n), block(256);
    switch (epilogue) {
        case VITHIP_EPI_BIAS: hipLaunchKernelGGL(gemm_f32_nt_latency_kernel<VITHIP_EPI_BIAS>, grid, block, 0, stream, p); break;
        case VITHIP_EPI_BIAS_GELU: hipLaunchKernelGGL(gemm_f32_nt_latency_kernel<VITHIP_EPI_BIAS_GELU>, grid, block, 0, stream, p); break;
        case VITHIP_EPI_BIAS_RESIDUAL: hipLaunchKernelGGL(gemm_f32_nt_latency_kernel<VITHIP_EPI_BIAS_RESIDUAL>, grid, block, 0, stream, p); break;
        case EPI_BIAS_LN: hipLaunchKernelGGL(gemm_f32_nt_latency_kernel<EPI_BIAS_LN>, grid, block, 0, stream, p); break;
        case EPI_BIAS_GELU_LN: hipLaunchKernelGGL(gemm_f32_nt_latency_kernel<EPI_BIAS_GELU_LN>, grid, block, 0, stream, p); break;
        default: return static_cast<int>(hipErrorInvalidValue);
    }
    return static_cast<int>(hipGetLastError());
}

}  // namespace vitgemm
