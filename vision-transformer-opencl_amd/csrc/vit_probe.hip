// csrc/vit_probe.hip -- measurement probes (not on the forward path).
//
// vithip_probe_mfma_f32: a register-only v_mfma_f32_32x32x2_f32 loop, 4 independent accumulators
// per wave, to read the fp32 matrix rate (and so the clock) the chip sustains; the GEMM roofline
// fraction in bench.py is quoted against the spec peak, this probe says how much of the gap is
// the kernel's and how much the clock's.
#ifdef VIT_PROBES  // whole file: probe build only (make probes -> libvit_mi355x_probe.so)
#include <hip/hip_runtime.h>

#include "vit_hip_kernels.h"
#include "vit_probes.h"
#include "vit_gemm_common.hpp"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_f32_loop_kernel(float *out, int iters, float seed) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    if (s == 12345.678f) out[0] = s;  // keep the chain alive without a store on the common path
}

// Waves 0-3 of each 512-thread block run the MFMA loop, waves 4-7 a dependent-free v_fma_f32 loop of
// `valu_iters` x 64 instructions: does fp32 VALU work on the partner wave slow the fp32 matrix pipe?
__global__ __launch_bounds__(512) void mfma_vs_valu_kernel(float *out, int iters, int valu_iters, float seed) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
        float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) s += acc[i][v];
        if (s == 12345.678f) out[0] = s;
    } else {
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = seed * (i + 1) + threadIdx.x * 1e-4f;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], 0.999f, 0.001f);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
        if (s == 12345.678f) out[1] = s;
    }
}

// As above, but waves 4-7 evaluate GELU: mode 0 = one value at a time, 1 = eight in lock-step.
template <int MODE>
__global__ __launch_bounds__(512) void mfma_vs_gelu_kernel(float *out, int iters, int gelu_iters, float seed) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
        float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) s += acc[i][v];
        if (s == 12345.678f) out[0] = s;
    } else {
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = seed * (i + 1) + threadIdx.x * 1e-3f - 1.0f;
        for (int it = 0; it < gelu_iters; ++it) {
            if constexpr (MODE == 1) {
                vitgemm::gelu_erf_x8(y);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = vitgemm::gelu_erf(y[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] += 0.37f;  // keep the inputs moving
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += y[i];
        if (s == 12345.678f) out[1] = s;
    }
}
// Store-path probe: every wave issues `iters` x 16-byte-per-lane stores.  mode 0: 1 KB contiguous per instruction;
// mode 1: 16 rows x 64 B (row stride `stride` bytes), the GEMM epilogue's fragment pattern; mode 2: 8 rows x 128 B.
// Reports s_memtime cycles from the first store issue to the last ISSUE (not completion) and to completion.
__global__ __launch_bounds__(512) void store_probe_kernel(float4 *out, int iters, int mode, size_t stride, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwave = blockDim.x >> 6;
    char *base = reinterpret_cast<char *>(out) + ((size_t)blockIdx.x * nwave + wave) * (size_t)iters * 16 * stride;
    size_t off;
    if (mode == 0) off = (size_t)lane * 16;
    else if (mode == 1) off = (size_t)(lane & 15) * stride + (lane >> 4) * 16;
    else off = (size_t)(lane >> 3) * stride + (lane & 7) * 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const size_t adv = mode == 0 ? (size_t)it * 1024 : (mode == 1 ? (size_t)it * 16 * stride : (size_t)it * 8 * stride);
        *reinterpret_cast<float4 *>(base + off + adv) = v;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        cyc[((size_t)blockIdx.x * nwave + wave) * 2] = t1 - t0;
        cyc[((size_t)blockIdx.x * nwave + wave) * 2 + 1] = t2 - t0;
    }
}

}  // namespace

extern "C" int vithip_probe_mfma_vs_gelu(vithip_stream_t stream, float *out, int blocks, int iters, int gelu_iters, int mode) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == 1) hipLaunchKernelGGL(mfma_vs_gelu_kernel<1>, dim3(blocks), dim3(512), 0, s, out, iters, gelu_iters, 0.37f);
    else hipLaunchKernelGGL(mfma_vs_gelu_kernel<0>, dim3(blocks), dim3(512), 0, s, out, iters, gelu_iters, 0.37f);
    return static_cast<int>(hipGetLastError());
}

extern "C" int vithip_probe_mfma_vs_valu(vithip_stream_t stream, float *out, int blocks, int iters, int valu_iters) {
    hipLaunchKernelGGL(mfma_vs_valu_kernel, dim3(blocks), dim3(512), 0, static_cast<hipStream_t>(stream), out, iters,
                       valu_iters, 0.37f);
    return static_cast<int>(hipGetLastError());
}

// Launches blocks x threads; every wave issues iters*32 MFMAs (iters*32*4096 flop).
extern "C" int vithip_probe_mfma_f32(vithip_stream_t stream, float *out, int blocks, int threads, int iters) {
    hipLaunchKernelGGL(mfma_f32_loop_kernel, dim3(blocks), dim3(threads), 0, static_cast<hipStream_t>(stream), out,
                       iters, 0.37f);
    return static_cast<int>(hipGetLastError());
}

extern "C" int vithip_probe_store(vithip_stream_t stream, void *out, int blocks, int threads, int iters, int mode, size_t stride, void *cycles) {
    hipLaunchKernelGGL(store_probe_kernel, dim3(blocks), dim3(threads), 0, static_cast<hipStream_t>(stream),
                       static_cast<float4 *>(out), iters, mode, stride, static_cast<unsigned long long *>(cycles));
    return static_cast<int>(hipGetLastError());
}
#endif  // VIT_PROBES
