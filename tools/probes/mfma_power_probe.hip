// What does the chip sustain on fp32 MFMAs alone, by instruction shape and operand activity?  512 workgroups x 4 waves (two waves
// per SIMD, as the GEMM), 64 accumulator registers per wave, eight A and eight B operand registers cycled as a K loop would:
//   v_mfma_f32_32x32x2_f32 : 2 x 2 accumulators of 16 registers -- per 4096 flop 2 operand + 32 accumulator register accesses
//   v_mfma_f32_16x16x4_f32 : 4 x 4 accumulators of 4 registers  -- per 4096 flop 4 operand + 16 accumulator register accesses
// (both add their k-products in the same order: tools/probes/mfma_order_probe.hip) on random and on all-zero operands.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_power_probe.hip -o gpurun_out/mfma_power_probe && ./gpurun_out/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_loop(const float *src, float *dst, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(t * 16 + i) & 0xfffff];
        b[i] = src[(t * 16 + 8 + i) & 0xfffff] * 0.01f;
    }
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + 4 * i) & 7], b[(s + 4 * j) & 7], acc[i][j], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += acc[i][j][v];
        dst[t] = s;
    } else {
        f32x4 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s)   // the same 32 x 4096 flop per iteration as above: 4 x 16 instructions of 2048
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(s + 2 * i) & 7], b[(s + 2 * j) & 7], acc[i][j], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) s += acc[i][j][v];
        dst[t] = s;
    }
}

// The fp32 GEMM's inner loop without its global side: operand fragments come from LDS (random contents, 16 ds_read_b128 per 32 k
// and wave as in csrc/vit_gemm_persistent.hip), 64 accumulator registers, two waves per SIMD.  Same LDS bytes and flops either way:
//   SHAPE 32: per 8 k   2 + 2 fragment reads feed 16 v_mfma_f32_32x32x2_f32
//   SHAPE 16: per 16 k  4 + 4 fragment reads feed 64 v_mfma_f32_16x16x4_f32
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void gemm_like_loop(const float *src, float *dst, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];   // 64 KB
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 16384; i += 256) lds[i] = src[(blockIdx.x * 16384 + i) & 0xfffff];
    __syncthreads();
    const f32x4 *frag = reinterpret_cast<const f32x4 *>(lds);   // 4096 float4
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 af[2], bf[2];
                const int base = (it * 67 + c * 1031 + (t >> 6) * 257) & 4095;
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = frag[(base + i * 64 + lane) & 4095];
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = frag[(base + 2048 + j * 64 + lane) & 4095];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][k], bf[j][k] * 0.01f, acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += acc[i][j][v];
    } else {
        f32x4 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x4 af[4], bf[4];
                const int base = (it * 67 + c * 1031 + (t >> 6) * 257) & 4095;
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = frag[(base + i * 64 + lane) & 4095];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = frag[(base + 2048 + j * 64 + lane) & 4095];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][k], bf[j][k] * 0.01f, acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) s += acc[i][j][v];
    }
    dst[blockIdx.x * 256 + t] = s;
}

template <int SHAPE>
static double run_gemm_like(const float *src, float *dst, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(gemm_like_loop<SHAPE>, dim3(512), dim3(256), 0, 0, src, dst, iters / 4);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gemm_like_loop<SHAPE>, dim3(512), dim3(256), 0, 0, src, dst, iters);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("HIP error\n"); exit(2); }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 3.0 * 512 * 4 * (double)iters * 262144.0;
    return flop / (ms * 1e-3) / 1e12;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// bf16: 16x16x32 (4 x 4 accumulators of 4 registers) or 32x32x16 (2 x 2 of 16); eight A and eight B operand quads cycled
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_bf16_loop(const float *src, float *dst, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32x4 ua, ub;
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // two bf16 per word: the upper halves of two random floats (sign, exponent, 7 mantissa bits)
            const unsigned x = __builtin_bit_cast(unsigned, src[(t * 64 + i * 8 + e) & 0xfffff]), y = __builtin_bit_cast(unsigned, src[(t * 64 + i * 8 + 4 + e) & 0xfffff]);
            ua[e] = (x >> 16) | (y & 0xffff0000u);
            ub[e] = (y >> 16) | (x & 0xffff0000u);
        }
        a[i] = __builtin_bit_cast(bf16x8, ua);
        b[i] = __builtin_bit_cast(bf16x8, ub);
    }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(k + 2 * i) & 7], b[(k + 2 * j) & 7], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) s += acc[i][j][v];
    } else {
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 4 * i) & 7], b[(k + 4 * j) & 7], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += acc[i][j][v];
    }
    dst[t] = s;
}

template <int SHAPE>
static double run_bf16(const float *src, float *dst, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_bf16_loop<SHAPE>, dim3(512), dim3(256), 0, 0, src, dst, iters / 4);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma_bf16_loop<SHAPE>, dim3(512), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("HIP error\n"); exit(2); }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 3.0 * 512 * 4 * (double)iters * 64 * 16384;   // 64 x 16x16x32 (= 32 x 32x32x16) per iteration
    return flop / (ms * 1e-3) / 1e12;
}

template <int SHAPE>
static double run(const float *src, float *dst, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(512), dim3(256), 0, 0, src, dst, iters / 4);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(512), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("HIP error\n"); exit(2); }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 3.0 * 512 * 4 * (double)iters * 32 * 4096;
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t n = 1 << 20;
    std::vector<float> h(n);
    srand(11);
    for (auto &x : h) x = rand() / (float)RAND_MAX * 2.f - 1.f;
    float *rnd, *zero, *dst;
    hipMalloc(&rnd, n * 4); hipMalloc(&zero, n * 4); hipMalloc(&dst, 512 * 256 * 4);
    hipMemcpy(rnd, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(zero, 0, n * 4);
    const int iters = 20000;   // ~ 35 ms per launch
    for (int round = 0; round < 3; ++round) {
        const double r32 = run<32>(rnd, dst, iters), r16 = run<16>(rnd, dst, iters), z32 = run<32>(zero, dst, iters), z16 = run<16>(zero, dst, iters);
        printf("{\"round\": %d, \"tflops\": {\"32x32x2_random\": %.1f, \"16x16x4_random\": %.1f, \"32x32x2_zeros\": %.1f, \"16x16x4_zeros\": %.1f}}\n",
               round, r32, r16, z32, z16);
    }
    for (int round = 0; round < 3; ++round) {
        const int itg = 2500;   // x 262144 flop per wave and iteration
        const double r32 = run_gemm_like<32>(rnd, dst, itg), r16 = run_gemm_like<16>(rnd, dst, itg), z32 = run_gemm_like<32>(zero, dst, itg), z16 = run_gemm_like<16>(zero, dst, itg);
        printf("{\"round\": %d, \"fragments_from_lds_tflops\": {\"32x32x2_random\": %.1f, \"16x16x4_random\": %.1f, \"32x32x2_zeros\": %.1f, \"16x16x4_zeros\": %.1f}}\n",
               round, r32, r16, z32, z16);
    }
    for (int round = 0; round < 3; ++round) {
        const int it16 = 80000;   // ~ 35 ms per launch at 2.4 PFLOP/s
        const double r16 = run_bf16<16>(rnd, dst, it16), r32 = run_bf16<32>(rnd, dst, it16), z16 = run_bf16<16>(zero, dst, it16), z32 = run_bf16<32>(zero, dst, it16);
        printf("{\"round\": %d, \"bf16_tflops\": {\"16x16x32_random\": %.0f, \"32x32x16_random\": %.0f, \"16x16x32_zeros\": %.0f, \"32x32x16_zeros\": %.0f}}\n",
               round, r16, r32, z16, z32);
    }
    return 0;
}
