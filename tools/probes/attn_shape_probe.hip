// Would v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16 buy the bf16 attention kernels anything?  An upper bound without
// re-deriving their layouts: the instruction MIX of one unit of the streamed kernel (csrc/vit_attention_stream.hip: 32 query rows x 64
// keys -- 8 score MFMAs, 32 exponentials + ~48 other vector instructions per lane, 8 P.V MFMAs fed by the packed probabilities) in
// a register-only loop, two waves per SIMD as the kernels run, once per shape (16 + 16 instructions of 16x16x32 for the same FLOPs).
// The dependency chain is kept: scores -> scale / max / exp2 / sum / pack -> operand of the P.V MFMAs.  No LDS, no memory in the
// loop: whatever the shape can give (issue slots, clock under the power limit) shows here undiluted.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/attn_shape_probe.hip -o gpurun_out/attn_shape_probe && ./gpurun_out/attn_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}

// VALU: 0 = MFMAs only, 1 = with the softmax mix between the two products
template <int SHAPE, int VALU>
__global__ __launch_bounds__(256, 2) void attn_like_loop(const float *src, float *dst, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bf16x8 q[4], k[4], v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x4 a, b, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // two bf16 per word: upper halves of random floats
            const unsigned x = __builtin_bit_cast(unsigned, src[(t * 48 + i * 12 + e) & 0xfffff]), y = __builtin_bit_cast(unsigned, src[(t * 48 + i * 12 + 4 + e) & 0xfffff]);
            const unsigned z = __builtin_bit_cast(unsigned, src[(t * 48 + i * 12 + 8 + e) & 0xfffff]);
            a[e] = (x >> 16) | (y & 0xffff0000u);
            b[e] = (y >> 16) | (z & 0xffff0000u);
            c[e] = (z >> 16) | (x & 0xffff0000u);
        }
        q[i] = __builtin_bit_cast(bf16x8, a);
        k[i] = __builtin_bit_cast(bf16x8, b);
        v[i] = __builtin_bit_cast(bf16x8, c);
    }
    float run_max = 0.f, run_sum = 0.f, out = 0.f;
    float sc[32];  // the lane's 32 scores of a unit
    if constexpr (SHAPE == 32) {
        f32x16 o[2] = {};
        for (int it = 0; it < iters; ++it) {
            f32x16 s[2] = {};
#pragma unroll
            for (int b = 0; b < 2; ++b)  // S^T = K . Q^T: two 32-key blocks, d = 64 in four k-chunks of 16
#pragma unroll
                for (int c = 0; c < 4; ++c) s[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k[(c + b) & 3], q[c], s[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) sc[16 * b + e] = s[b][e];
            bf16x8 pf[4];
            if constexpr (VALU) {
                float m = run_max;
#pragma unroll
                for (int e = 0; e < 32; ++e) m = fmaxf(m, sc[e]);                                    // 32 v_max
#pragma unroll
                for (int e = 0; e < 32; ++e) sc[e] = __builtin_amdgcn_exp2f(sc[e] * 0.01f - m * 0.01f);  // 32 v_fma + 32 v_exp
#pragma unroll
                for (int e = 0; e < 32; ++e) run_sum += sc[e];                                       // 32 v_add
                run_max = m * 0.5f;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {                                                            // 16 v_cvt_pk
                u32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = pack2(sc[8 * g + 2 * e], sc[8 * g + 2 * e + 1]);
                pf[g] = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d)  // O^T += V^T . P^T: two 32-wide d blocks, 64 keys in four k-chunks of 16
#pragma unroll
                for (int g = 0; g < 4; ++g) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[(g + d) & 3], pf[g], o[d], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) out += o[d][e];
    } else {
        f32x4 o[8] = {};
        for (int it = 0; it < iters; ++it) {
            f32x4 s[8] = {};
#pragma unroll
            for (int b = 0; b < 8; ++b)  // 2 x 4 blocks of 16 x 16, d = 64 in two k-chunks of 32
#pragma unroll
                for (int c = 0; c < 2; ++c) s[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k[(c + b) & 3], q[(c + (b >> 2)) & 3], s[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[4 * b + e] = s[b][e];
            bf16x8 pf[4];
            if constexpr (VALU) {
                float m = run_max;
#pragma unroll
                for (int e = 0; e < 32; ++e) m = fmaxf(m, sc[e]);
#pragma unroll
                for (int e = 0; e < 32; ++e) sc[e] = __builtin_amdgcn_exp2f(sc[e] * 0.01f - m * 0.01f);
#pragma unroll
                for (int e = 0; e < 32; ++e) run_sum += sc[e];
                run_max = m * 0.5f;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = pack2(sc[8 * g + 2 * e], sc[8 * g + 2 * e + 1]);
                pf[g] = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int d = 0; d < 8; ++d)  // 2 x 4 blocks of 16 x 16, 64 keys in two k-chunks of 32
#pragma unroll
                for (int g = 0; g < 2; ++g) o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v[(g + d) & 3], pf[(2 * (d & 1) + g) & 3], o[d], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e) out += o[d][e];
    }
    dst[t] = out + run_sum + run_max;
}

template <int SHAPE, int VALU>
static double run(const float *src, float *dst, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((attn_like_loop<SHAPE, VALU>), dim3(512), dim3(256), 0, 0, src, dst, iters / 4);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((attn_like_loop<SHAPE, VALU>), dim3(512), dim3(256), 0, 0, src, dst, iters);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) {
        printf("HIP error\n");
        exit(2);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0;
}

int main() {
    const size_t n = 1 << 20;
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f;
    }
    float *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, n * 4) != hipSuccess || hipMalloc(&dst, 512 * 256 * 4) != hipSuccess) return 1;
    (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    // per iteration and wave: 16 x 32768 flop (32x32x16) = 32 x 16384 (16x16x32) per lane group = 524,288 x 2 flop... counted per wave
    const double flop = 512.0 * 4 * (double)iters * 16 * 32 * 32 * 16 * 2;
    for (int round = 0; round < 2; ++round) {
        const double a = run<32, 0>(src, dst, iters), b = run<16, 0>(src, dst, iters), c = run<32, 1>(src, dst, iters), d = run<16, 1>(src, dst, iters);
        printf("{\"round\": %d, \"mfma_only_ms\": {\"32x32x16\": %.3f, \"16x16x32\": %.3f}, \"mfma_only_tflops\": {\"32x32x16\": %.0f, \"16x16x32\": %.0f}, "
               "\"with_softmax_mix_ms\": {\"32x32x16\": %.3f, \"16x16x32\": %.3f}, \"with_softmax_mix_tflops\": {\"32x32x16\": %.0f, \"16x16x32\": %.0f}}\n",
               round, a, b, flop / (a * 1e-3) / 1e12, flop / (b * 1e-3) / 1e12, c, d, flop / (c * 1e-3) / 1e12, flop / (d * 1e-3) / 1e12);
    }
    return 0;
}
