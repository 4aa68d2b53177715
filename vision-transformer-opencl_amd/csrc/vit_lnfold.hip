// csrc/vit_lnfold.hip -- the small kernels around the LayerNorm fold of the bf16 forward (the fold itself lives in the
// epilogues of csrc/vit_gemm_bf16_pp.hip; include/vit_hip_kernels.h states the algebra).
//
//   ln_fold_weights_kernel   once per weight upload: Wf = bf16(gamma * W), colsum = sum_k Wf, bias_f = bias + W . beta
//   rowstats_bf16_kernel     first LayerNorm of the stack: x16 = bf16(x), rows = (rstd, mean*rstd)   (one wave per row, HBM-bound)
//   rowstats_finalize_kernel per residual GEMM: strips x (sum, sum of squares) -> (rstd, mean*rstd), fixed summation order
//
// Statistics follow ViT_seq.c:103-121 as csrc/vit_rowops.hip does: var = E[x^2] - mean^2, 1/sqrtf((double)var + 1e-6).
#include <hip/hip_runtime.h>

#include "vit_hip_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ f32x2 finish(float s, float ss, int dim) {
    const float mean = s / (float)dim;
    const float var = ss / (float)dim - mean * mean;
    const float inv_std = 1.0f / sqrtf((float)((double)var + 1e-6));
    return f32x2{inv_std, mean * inv_std};
}

// one wave per output feature n
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const float *__restrict__ W, const float *__restrict__ bias,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta,
                                                              unsigned short *__restrict__ Wf, float *__restrict__ colsum,
                                                              float *__restrict__ bias_f, int N, int K, int scale_rows, float scale) {
    const int lane = threadIdx.x & 63;
    const int n = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (n >= N) return;  // wave-uniform
    const float sc = n < scale_rows ? scale : 1.0f;  // output features [0, scale_rows) are produced `scale` times as large
    const float *src = W + (size_t)n * K;
    unsigned short *dst = Wf + (size_t)n * K;
    float cs = 0.f, bs = 0.f;
    for (int c = lane * 4; c < K; c += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(src + c);
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + c);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = (__bf16)(sc * (g[j] * w[j]));
            cs += (float)o[j];
            bs += b[j] * w[j];
        }
        *reinterpret_cast<bf16x4 *>(dst + c) = o;
    }
    cs = wave_sum(cs);
    bs = wave_sum(bs);
    if (lane == 0) {
        colsum[n] = cs;
        bias_f[n] = sc * (bias[n] + bs);
    }
}

// (rstd, mean) for the fp32 fold: every operation rounded on its own -- contraction is switched off for the body (HIP's
// __fmul_rn & co. are plain operators that the compiler may still fuse; its __fsqrt_rn is the approximate square root)
__device__ __forceinline__ f32x2 finish_f32(float s, float ss, int dim) {
#pragma clang fp contract(off)
    const float mean = s / (float)dim;
    const float sq = mean * mean;
    const float var = ss / (float)dim - sq;
    const float inv_std = 1.0f / __builtin_sqrtf((float)((double)var + 1e-6));
    return f32x2{inv_std, mean};
}

// fp32 fold: one wave per output feature n; the products gamma * W are the fp32 values the GEMM will multiply, their sum and the
// beta term are taken in double and rounded once
__global__ __launch_bounds__(256) void ln_fold_weights_f32_kernel(const float *__restrict__ W, const float *__restrict__ bias,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                  float *__restrict__ Wf, float *__restrict__ colsum,
                                                                  float *__restrict__ bias_f, int N, int K, int center) {
    const int lane = threadIdx.x & 63;
    const int n = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (n >= N) return;  // wave-uniform
    const float *src = W + (size_t)n * K;
    float *dst = Wf + (size_t)n * K;
    double cs = 0.0, bs = 0.0;
    for (int c = lane * 4; c < K; c += 256) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(src + c);
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = g[j] * w[j];  // (nothing to fuse with: the sums below are double)
            cs += (double)o[j];
            bs += (double)b[j] * (double)w[j];
        }
        *reinterpret_cast<f32x4 *>(dst + c) = o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cs += __shfl_xor(cs, off);
        bs += __shfl_xor(bs, off);
    }
    if (center) {
        // CENTRED form (vithip_ln_fold_weights_f32_centered): Wc[k] = gamma[k] W[k] - cbar, cbar = sum_k gamma[k] W[k] / K.  Then
        //     x . Wc^T = x . (gamma W)^T - cbar * sum_k x[k] = x . (gamma W)^T - mean * colsum          (mean = sum_k x[k] / K)
        // i.e. the GEMM itself delivers the centred product and its epilogue has nothing to subtract.  One rounding per weight (the
        // difference is taken in double); what is left of the column sum after that rounding is written to `colsum` for the record
        // (~K^0.5 * 2^-24 * |W|: its product with mean * rstd is 1e-6 of the output and nobody adds it).
        const double cbar = cs / (double)K;
        double rs = 0.0;
        for (int c = lane * 4; c < K; c += 256) {
            f32x4 o = *reinterpret_cast<const f32x4 *>(dst + c);   // gamma * W as stored above (this lane wrote it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (float)((double)o[j] - cbar);
                rs += (double)o[j];
            }
            *reinterpret_cast<f32x4 *>(dst + c) = o;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rs += __shfl_xor(rs, off);
        cs = rs;
    }
    if (lane == 0) {
        colsum[n] = (float)cs;
        bias_f[n] = (float)((double)bias[n] + bs);
    }
}

// Row statistics in the canonical order of the fp32 fold (include/vit_hip_kernels.h, vithip_rowstats_f32): per 64-column strip
// lane l < 32 holds u = x[c + l] + x[c + 32 + l] and w = fmaf(x[c + 32 + l], x[c + 32 + l], x[c + l] * x[c + l]), a butterfly
// 16, 8, 4, 2, 1 over the 32 lanes sums them, strips are added in ascending order starting from 0.  One wave per row, the two
// halves of the wave on two strips at a time.  NP = strip pairs per row (0: run-time count).
template <int NP>
__global__ __launch_bounds__(256) void rowstats_f32_kernel(const float *__restrict__ x, size_t ldx, float *__restrict__ rows_out,
                                                           int rows, int dim) {
    const int lane = threadIdx.x & 63, l = lane & 31, half = lane >> 5;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    const int strips = dim >> 6;
    const int np = NP > 0 ? NP : (strips + 1) >> 1;
    for (int row = wave; row < rows; row += nwaves) {
        const float *src = x + (size_t)row * ldx;
        float s = 0.f, ss = 0.f;
        auto pair = [&](int k, float x0, float x1) {
#pragma clang fp contract(off)
            float u = x0 + x1;
            const float sq0 = x0 * x0;
            float w = __builtin_fmaf(x1, x1, sq0);
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                u += __shfl_xor(u, off);
                w += __shfl_xor(w, off);
            }
            const float u0 = __shfl(u, 0), u1 = __shfl(u, 32), w0 = __shfl(w, 0), w1 = __shfl(w, 32);
            s += u0;
            ss += w0;
            if (2 * k + 1 < strips) {  // wave-uniform
                s += u1;
                ss += w1;
            }
        };
        if constexpr (NP > 0) {
            float x0[NP], x1[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {  // every load of the row first
                const int strip = 2 * k + half;
                const bool ok = strip < strips;
                x0[k] = ok ? src[strip * 64 + l] : 0.f;
                x1[k] = ok ? src[strip * 64 + 32 + l] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) pair(k, x0[k], x1[k]);
        } else {
            for (int k = 0; k < np; ++k) {
                const int strip = 2 * k + half;
                const bool ok = strip < strips;
                pair(k, ok ? src[strip * 64 + l] : 0.f, ok ? src[strip * 64 + 32 + l] : 0.f);
            }
        }
        if (lane == 0) *reinterpret_cast<f32x2 *>(rows_out + (size_t)row * 2) = finish_f32(s, ss, dim);
    }
}

constexpr int RS_MAX_VEC = 8;  // float4 per lane: dim <= 2048
template <int NVEC>
__global__ __launch_bounds__(256) void rowstats_bf16_kernel(const float *__restrict__ x, size_t ldx, unsigned short *__restrict__ x16,
                                                            size_t ldx16, float *__restrict__ rows_out, int rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    for (int row = wave; row < rows; row += nwaves) {
        const float *src = x + (size_t)row * ldx;
        unsigned short *dst = x16 + (size_t)row * ldx16;
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < dim) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(src + c);
                s += (v[0] + v[1]) + (v[2] + v[3]);
                ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                *reinterpret_cast<bf16x4 *>(dst + c) = o;
            }
        }
        s = wave_sum(s);
        ss = wave_sum(ss);
        if (lane == 0) *reinterpret_cast<f32x2 *>(rows_out + (size_t)row * 2) = finish(s, ss, dim);
    }
}

__global__ __launch_bounds__(256) void rowstats_finalize_kernel(const float *__restrict__ partials, int strips, int rows, int dim,
                                                                float *__restrict__ rows_out) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float s = 0.f, ss = 0.f;
    for (int k = 0; k < strips; ++k) {  // strips in ascending column order, the same for every row and every launch
        const f32x2 v = *reinterpret_cast<const f32x2 *>(partials + ((size_t)k * rows + row) * 2);
        s += v.x;
        ss += v.y;
    }
    *reinterpret_cast<f32x2 *>(rows_out + (size_t)row * 2) = finish(s, ss, dim);
}

// fp32 fold: strips of a residual GEMM's epilogue (csrc/vit_gemm_common.hpp, epilogue_store_residual_stats) -> (rstd, mean),
// strips added in ascending order from 0 and finished as vithip_rowstats_f32 does: the same bits as that kernel on the stored rows
__global__ __launch_bounds__(256) void rowstats_finalize_f32_kernel(const float *__restrict__ partials, int strips, int rows, int dim,
                                                                    float *__restrict__ rows_out) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float s = 0.f, ss = 0.f;
    for (int k = 0; k < strips; ++k) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(partials + ((size_t)k * rows + row) * 2);
        s += v.x;
        ss += v.y;
    }
    *reinterpret_cast<f32x2 *>(rows_out + (size_t)row * 2) = finish_f32(s, ss, dim);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, size_t src_stride, float *__restrict__ dst,
                                                          size_t dst_stride, int rows, int width) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * width) return;
    const size_t r = i / width, c = i - r * width;
    dst[r * dst_stride + c] = src[r * src_stride + c];
}

bool aligned16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int vithip_ln_strips(int N) { return N > 0 ? 4 * ((N + 255) / 256) : 0; }

int vithip_ln_fold_weights_scaled(vithip_stream_t stream, const float *W, const float *bias, const float *gamma, const float *beta,
                                  unsigned short *Wf, float *colsum, float *bias_f, int N, int K, int scale_rows, float scale) {
    if (!W || !bias || !gamma || !beta || !Wf || !colsum || !bias_f || N <= 0 || K <= 0 || K % 4 || scale_rows < 0 || scale_rows > N)
        return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(W) || !aligned16(gamma) || !aligned16(beta) || (reinterpret_cast<size_t>(Wf) & 7))
        return static_cast<int>(hipErrorInvalidValue);
    hipLaunchKernelGGL(ln_fold_weights_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), W, bias, gamma,
                       beta, Wf, colsum, bias_f, N, K, scale_rows, scale);
    return static_cast<int>(hipGetLastError());
}

int vithip_ln_fold_weights(vithip_stream_t stream, const float *W, const float *bias, const float *gamma, const float *beta,
                           unsigned short *Wf, float *colsum, float *bias_f, int N, int K) {
    return vithip_ln_fold_weights_scaled(stream, W, bias, gamma, beta, Wf, colsum, bias_f, N, K, 0, 1.0f);
}

int vithip_ln_fold_weights_f32(vithip_stream_t stream, const float *W, const float *bias, const float *gamma, const float *beta,
                               float *Wf, float *colsum, float *bias_f, int N, int K) {
    if (!W || !bias || !gamma || !beta || !Wf || !colsum || !bias_f || N <= 0 || K <= 0 || K % 4)
        return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(W) || !aligned16(gamma) || !aligned16(beta) || !aligned16(Wf)) return static_cast<int>(hipErrorInvalidValue);
    hipLaunchKernelGGL(ln_fold_weights_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), W, bias, gamma,
                       beta, Wf, colsum, bias_f, N, K, 0);
    return static_cast<int>(hipGetLastError());
}

int vithip_ln_fold_weights_f32_centered(vithip_stream_t stream, const float *W, const float *bias, const float *gamma, const float *beta,
                                        float *Wc, float *residual_colsum, float *bias_f, int N, int K) {
    if (!W || !bias || !gamma || !beta || !Wc || !residual_colsum || !bias_f || N <= 0 || K <= 0 || K % 4)
        return static_cast<int>(hipErrorInvalidValue);
    if (!aligned16(W) || !aligned16(gamma) || !aligned16(beta) || !aligned16(Wc)) return static_cast<int>(hipErrorInvalidValue);
    hipLaunchKernelGGL(ln_fold_weights_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), W, bias, gamma,
                       beta, Wc, residual_colsum, bias_f, N, K, 1);
    return static_cast<int>(hipGetLastError());
}

int vithip_rowstats_f32(vithip_stream_t stream, const float *x, size_t ldx, float *rows_out, int rows, int dim) {
    if (!x || !rows_out || rows <= 0 || dim <= 0 || dim % 64 || dim > 2048 || ldx < (size_t)dim ||
        (reinterpret_cast<size_t>(x) & 3) || (reinterpret_cast<size_t>(rows_out) & 7))
        return static_cast<int>(hipErrorInvalidValue);
    int blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RSF_LAUNCH(NP) hipLaunchKernelGGL(rowstats_f32_kernel<NP>, dim3(blocks), dim3(256), 0, s, x, ldx, rows_out, rows, dim)
    switch ((dim / 64 + 1) / 2) {
        case 2: RSF_LAUNCH(2); break;   // 192, 256
        case 6: RSF_LAUNCH(6); break;   // 768
        case 8: RSF_LAUNCH(8); break;   // 1024
        default: RSF_LAUNCH(0); break;
    }
#undef RSF_LAUNCH
    return static_cast<int>(hipGetLastError());
}

int vithip_rowstats_bf16(vithip_stream_t stream, const float *x, size_t ldx, unsigned short *x16, size_t ldx16, float *rows_out,
                         int rows, int dim) {
    if (!x || !x16 || !rows_out || rows <= 0 || dim <= 0 || dim % 4 || dim > 64 * 4 * RS_MAX_VEC || ldx % 4 || ldx16 % 4 ||
        ldx < (size_t)dim || ldx16 < (size_t)dim || !aligned16(x) || (reinterpret_cast<size_t>(x16) & 7) ||
        (reinterpret_cast<size_t>(rows_out) & 7))
        return static_cast<int>(hipErrorInvalidValue);
    const int nvec = (dim + 255) / 256;
    int blocks = (rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RS_LAUNCH(NV) hipLaunchKernelGGL(rowstats_bf16_kernel<NV>, dim3(blocks), dim3(256), 0, s, x, ldx, x16, ldx16, rows_out, rows, dim)
    switch (nvec) {
        case 1: RS_LAUNCH(1); break;
        case 2: RS_LAUNCH(2); break;
        case 3: RS_LAUNCH(3); break;
        case 4: RS_LAUNCH(4); break;
        default: RS_LAUNCH(8); break;
    }
#undef RS_LAUNCH
    return static_cast<int>(hipGetLastError());
}

int vithip_rowstats_finalize(vithip_stream_t stream, const float *partials, int strips, int rows, int dim, float *rows_out) {
    if (!partials || !rows_out || strips <= 0 || rows <= 0 || dim <= 0 || (reinterpret_cast<size_t>(partials) & 7) ||
        (reinterpret_cast<size_t>(rows_out) & 7))
        return static_cast<int>(hipErrorInvalidValue);
    hipLaunchKernelGGL(rowstats_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), partials,
                       strips, rows, dim, rows_out);
    return static_cast<int>(hipGetLastError());
}

int vithip_rowstats_finalize_f32(vithip_stream_t stream, const float *partials, int rows, int dim, float *rows_out) {
    if (!partials || !rows_out || rows <= 0 || dim <= 0 || dim % 64 || (reinterpret_cast<size_t>(partials) & 7) ||
        (reinterpret_cast<size_t>(rows_out) & 7))
        return static_cast<int>(hipErrorInvalidValue);
    hipLaunchKernelGGL(rowstats_finalize_f32_kernel, dim3((rows + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), partials,
                       dim / 64, rows, dim, rows_out);
    return static_cast<int>(hipGetLastError());
}

int vithip_gather_rows_f32(vithip_stream_t stream, const float *src, size_t src_stride, float *dst, size_t dst_stride, int rows,
                           int width) {
    if (!src || !dst || rows <= 0 || width <= 0 || src_stride < (size_t)width || dst_stride < (size_t)width)
        return static_cast<int>(hipErrorInvalidValue);
    const size_t total = (size_t)rows * width;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                       src_stride, dst, dst_stride, rows, width);
    return static_cast<int>(hipGetLastError());
}

}  // extern "C"
