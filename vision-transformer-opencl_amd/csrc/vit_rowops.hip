// csrc/vit_rowops.hip -- row-wise kernels: LayerNorm and the final softmax + top-1.
//
// LayerNorm follows ViT_seq.c:103-121 (NOT kernel.cl:6-80, which drops the epsilon):
//   mean = sum/dim, var = sum_sq/dim - mean*mean, inv_std = 1/sqrtf((double)var + 1e-6),
//   y = (x - mean) * inv_std * gamma + beta.
// One 64-lane wave per row: the row is read once as float4 (dim <= 2048 stays in registers),
// the two sums are reduced with wave shuffles, no LDS and no barrier.  HBM-bound by design.
//
// softmax_top1 follows ViT_seq.c:304-324 and the argmax of Main.c:62-70 (first maximum wins).
#include <hip/hip_runtime.h>

#include "vit_hip_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LN_THREADS = 256;
constexpr int LN_MAX_VEC = 8;  // float4 per lane: dim <= 64*4*8 = 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// OUT = float, or unsigned short (bf16 bits) for the bf16 variant of the forward: statistics and the
// affine map are fp32 either way, only the store rounds.
template <int NVEC, typename OUT>
__global__ __launch_bounds__(LN_THREADS) void layernorm_f32_kernel(const float *__restrict__ x, size_t ldx,
                                                                   OUT *__restrict__ y, size_t ldy,
                                                                   const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta,
                                                                   int rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * LN_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * LN_THREADS) >> 6;
    for (int row = wave; row < rows; row += nwaves) {
        const float *src = x + (size_t)row * ldx;
        f32x4 v[NVEC];
        float s = 0.0f, ss = 0.0f;
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < dim) {
                v[i] = *reinterpret_cast<const f32x4 *>(src + c);
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
                ss += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            }
        }
        s = wave_sum(s);
        ss = wave_sum(ss);
        const float mean = s / (float)dim;
        const float var = ss / (float)dim - mean * mean;
        const float inv_std = 1.0f / sqrtf((float)((double)var + 1e-6));
        OUT *dst = y + (size_t)row * ldy;
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < dim) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + c);
                const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + c);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * inv_std * g[j] + b[j];
                if constexpr (sizeof(OUT) == 4) {
                    *reinterpret_cast<f32x4 *>(dst + c) = o;
                } else {
                    bf16x4 ob;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ob[j] = (__bf16)o[j];
                    *reinterpret_cast<bf16x4 *>(dst + c) = ob;
                }
            }
        }
    }
}

constexpr int SM_THREADS = 256;

// One workgroup per row.
__global__ __launch_bounds__(SM_THREADS) void softmax_top1_f32_kernel(const float *__restrict__ logits, int ld_logits,
                                                                      float *__restrict__ probs, int ld_probs,
                                                                      int *__restrict__ top1_label,
                                                                      float *__restrict__ top1_prob, int classes) {
    __shared__ float red_f[SM_THREADS / 64];
    __shared__ int red_i[SM_THREADS / 64];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *src = logits + (size_t)row * ld_logits;
    float *dst = probs + (size_t)row * ld_probs;

    float mx = -INFINITY;
    for (int c = tid; c < classes; c += SM_THREADS) mx = fmaxf(mx, src[c]);
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
#pragma unroll
    for (int w = 1; w < SM_THREADS / 64; ++w) mx = fmaxf(mx, red_f[w]);
    __syncthreads();

    float sum = 0.0f;
    for (int c = tid; c < classes; c += SM_THREADS) {
        const float e = expf(src[c] - mx);
        dst[c] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red_f[wave] = sum;
    __syncthreads();
    sum = 0.0f;
#pragma unroll
    for (int w = 0; w < SM_THREADS / 64; ++w) sum += red_f[w];
    __syncthreads();

    // normalise (each thread re-reads only its own writes) and track the first maximum probability
    float best = -1.0f;
    int arg = 0x7fffffff;
    for (int c = tid; c < classes; c += SM_THREADS) {
        const float pr = dst[c] / sum;
        dst[c] = pr;
        if (pr > best) { best = pr; arg = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oa = __shfl_xor(arg, off);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) { red_f[wave] = best; red_i[wave] = arg; }
    __syncthreads();
    if (tid == 0) {
        best = red_f[0]; arg = red_i[0];
        for (int w = 1; w < SM_THREADS / 64; ++w)
            if (red_f[w] > best || (red_f[w] == best && red_i[w] < arg)) { best = red_f[w]; arg = red_i[w]; }
        if (top1_label) top1_label[row] = arg;
        if (top1_prob) top1_prob[row] = best;
    }
}

template <int NVEC, typename OUT>
int launch_ln(hipStream_t s, const float *x, size_t ldx, OUT *y, size_t ldy, const float *gamma,
              const float *beta, int rows, int dim) {
    const int rows_per_block = LN_THREADS / 64;
    int blocks = (rows + rows_per_block - 1) / rows_per_block;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
    hipLaunchKernelGGL((layernorm_f32_kernel<NVEC, OUT>), dim3(blocks), dim3(LN_THREADS), 0, s, x, ldx, y, ldy, gamma,
                       beta, rows, dim);
    return static_cast<int>(hipGetLastError());
}

template <typename OUT>
int layernorm_dispatch(vithip_stream_t stream, const float *x, size_t ldx, OUT *y, size_t ldy, const float *gamma,
                       const float *beta, int rows, int dim) {
    if (!x || !y || !gamma || !beta || rows <= 0 || dim <= 0) return static_cast<int>(hipErrorInvalidValue);
    if (dim % 4 || dim > 64 * 4 * LN_MAX_VEC || ldx % 4 || ldy % 4) return static_cast<int>(hipErrorInvalidValue);
    if ((reinterpret_cast<size_t>(x) & 15) || (reinterpret_cast<size_t>(y) & 15) ||
        (reinterpret_cast<size_t>(gamma) & 15) || (reinterpret_cast<size_t>(beta) & 15))
        return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nvec = (dim + 255) / 256;
    switch (nvec) {
        case 1: return launch_ln<1, OUT>(s, x, ldx, y, ldy, gamma, beta, rows, dim);
        case 2: return launch_ln<2, OUT>(s, x, ldx, y, ldy, gamma, beta, rows, dim);
        case 3: return launch_ln<3, OUT>(s, x, ldx, y, ldy, gamma, beta, rows, dim);
        case 4: return launch_ln<4, OUT>(s, x, ldx, y, ldy, gamma, beta, rows, dim);
        default: return launch_ln<LN_MAX_VEC, OUT>(s, x, ldx, y, ldy, gamma, beta, rows, dim);
    }
}

}  // namespace

extern "C" {

int vithip_layernorm_f32(vithip_stream_t stream, const float *x, size_t ldx, float *y, size_t ldy,
                         const float *gamma, const float *beta, int rows, int dim) {
    return layernorm_dispatch<float>(stream, x, ldx, y, ldy, gamma, beta, rows, dim);
}

int vithip_layernorm_f32_bf16out(vithip_stream_t stream, const float *x, size_t ldx, unsigned short *y, size_t ldy,
                                 const float *gamma, const float *beta, int rows, int dim) {
    return layernorm_dispatch<unsigned short>(stream, x, ldx, y, ldy, gamma, beta, rows, dim);
}

int vithip_softmax_top1_f32(vithip_stream_t stream, const float *logits, int ld_logits, float *probs,
                            int ld_probs, int *top1_label, float *top1_prob, int rows, int classes) {
    if (!logits || !probs || rows <= 0 || classes <= 0 || ld_logits < classes || ld_probs < classes)
        return static_cast<int>(hipErrorInvalidValue);
    hipLaunchKernelGGL(softmax_top1_f32_kernel, dim3(rows), dim3(SM_THREADS), 0, static_cast<hipStream_t>(stream),
                       logits, ld_logits, probs, ld_probs, top1_label, top1_prob, classes);
    return static_cast<int>(hipGetLastError());
}

}  // extern "C"
