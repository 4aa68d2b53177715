// csrc/vit_attention.hip -- fused scaled-dot-product attention for one (image, head) per workgroup.
//
// Reference: ViT_seq.c:156-215 (scores = q.k / sqrtf(head_dim); row softmax with max
// subtraction; head_out = P.V) -- the OpenCL path runs it as 36 launches per layer chained by
// events (ViT_opencl.c:546-573; kernel.cl:208-284,289-365) and materialises the 197x197
// scores of all 12 heads in global memory.  Here K and V of the head are staged once in LDS
// (2 x 224 x 64 fp32 = 118 KB of the 160 KB), each wave owns a block of 32 query rows and the
// scores never leave registers.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32):
//   S^T tile = K_tile(32 keys x 64) . Q_block^T  -> the accumulator holds, per lane,
//       query = lane&31 and 16 keys (key = 32*kt + (v&3) + 8*(v>>2) + 4*(lane>>5)).
//       A row of the softmax is therefore lane-local (+ one exchange with lane^32).
//   O^T tile = V_tile^T(32 d x keys) . P^T(keys x 32 queries): register v of the S^T
//       accumulator is already the B operand of this product (k = lane>>5 selects
//       key_a / key_a+4), so P goes from softmax to the second MFMA with no data movement.
// The softmax normaliser is applied to O (32 values per lane) instead of P (112 per lane).
#include <hip/hip_runtime.h>

#include "vit_hip_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64;          // head_dim (ViT-B/16 and ViT-L/16)
constexpr int K_LD = HD + 4;    // padded K row (floats): conflict-free ds_read_b128 over 32 rows
constexpr int ATT_THREADS = 512;
constexpr int ATT_WAVES = ATT_THREADS / 64;

template <int NKT>  // number of 32-key tiles: tokens <= 32*NKT
__global__ __launch_bounds__(ATT_THREADS) void attention_f32_kernel(const float *__restrict__ qkv,
                                                                    float *__restrict__ out,
                                                                    int tokens, int heads) {
    constexpr int KEYS = NKT * 32;
    __shared__ __attribute__((aligned(16))) float lds[KEYS * K_LD + KEYS * HD];
    // V first: every V address is then lane_base + a 16-bit immediate (no per-key address VGPRs)
    float *const Vs = lds;
    float *const Ks = lds + KEYS * HD;

    const int head = blockIdx.x, img = blockIdx.y;
    const int D = heads * HD, ld = 3 * D;
    const int tid = threadIdx.x;
    const float *base = qkv + (size_t)img * tokens * ld + head * HD;

    // ---- stage K and V of this head: 16 float4 per row, rows >= tokens are zero ----------
    {
        const int c4 = (tid & 15) * 4;
        for (int row = tid >> 4; row < KEYS; row += ATT_THREADS / 16) {
            f32x4 k = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
            if (row < tokens) {
                const float *src = base + (size_t)row * ld + c4;
                k = *reinterpret_cast<const f32x4 *>(src + D);
                v = *reinterpret_cast<const f32x4 *>(src + 2 * D);
            }
            *reinterpret_cast<f32x4 *>(Ks + row * K_LD + c4) = k;
            *reinterpret_cast<f32x4 *>(Vs + row * HD + c4) = v;
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nqt = (tokens + 31) >> 5;

    // One 32-query block per wave (tokens <= 224 => at most 7 of the 8 waves have work).  No loop
    // over blocks: it would let the compiler hoist ~100 per-key addresses and predicates out of it.
    if (wave < nqt) {
        const int q0 = wave * 32;
        // Q fragment: lane holds Q[q0+r][8c + 4h + s], c = 0..7, s = 0..3 (rows past the end clamp)
        f32x4 qf[8];
        {
            int qrow = q0 + r;
            qrow = qrow < tokens ? qrow : tokens - 1;
            const float *qsrc = base + (size_t)qrow * ld + h * 4;
#pragma unroll
            for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const f32x4 *>(qsrc + c * 8);
        }

        // ---- S^T = K . Q^T -----------------------------------------------------------
        f32x16 st[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) st[kt][v] = 0.0f;
            const float *kp = Ks + (kt * 32 + r) * K_LD + h * 4;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4 *>(kp + c * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[c][s], st[kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the next tile's K reads from piling up in VGPRs
        }

        // ---- row softmax over keys (ViT_seq.c:174-191) --------------------------------
        // Only the last key tile can hold keys >= tokens (NKT = ceil(tokens/32)).
        float mx = -INFINITY;
        const int rem = tokens - (NKT - 1) * 32;  // valid keys in the last tile, 1..32
        const int h4 = 4 * h;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float s = st[kt][v] * 0.125f;  // / sqrtf(64), exact
                if (kt == NKT - 1) {
                    const int kloc = (v & 3) + 8 * (v >> 2);  // + 4h = key index inside the tile
                    s = h4 < rem - kloc ? s : -INFINITY;
                }
                st[kt][v] = s;
                mx = fmaxf(mx, s);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float e = expf(st[kt][v] - mx);  // exp(-inf) = 0 for masked keys
                st[kt][v] = e;
                sum += e;
                if ((v & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;

        // ---- O^T = V^T . P^T -----------------------------------------------------------
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[dt][v] = 0.0f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int key_a = kt * 32 + (v & 3) + 8 * (v >> 2);  // lanes 0-31; lanes 32-63: +4
                // tiles before the last are full (NKT = ceil(tokens/32)); the test is wave-uniform
                if (kt < NKT - 1 || key_a < tokens) {
                    const float *vp = Vs + (key_a + 4 * h) * HD + r;
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], st[kt][v], o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], st[kt][v], o[1], 0, 0, 0);
                }
                if ((v & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- store: lane owns query q0+r; registers 4g..4g+3 are 4 consecutive d ---------
        if (q0 + r < tokens) {
            float *dst = out + ((size_t)img * tokens + q0 + r) * D + head * HD + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w;
                    w[0] = o[dt][4 * g + 0] * inv;
                    w[1] = o[dt][4 * g + 1] * inv;
                    w[2] = o[dt][4 * g + 2] * inv;
                    w[3] = o[dt][4 * g + 3] * inv;
                    *reinterpret_cast<f32x4 *>(dst + dt * 32 + 8 * g) = w;
                }
        }
    }
}

template <int NKT>
int launch(hipStream_t s, const float *qkv, float *out, int n_images, int tokens, int heads) {
    hipLaunchKernelGGL(attention_f32_kernel<NKT>, dim3(heads, n_images), dim3(ATT_THREADS), 0, s, qkv, out,
                       tokens, heads);
    return static_cast<int>(hipGetLastError());
}

}  // namespace

extern "C" int vithip_attention_f32(vithip_stream_t stream, const float *qkv, float *out,
                                    int n_images, int tokens, int heads) {
    if (!qkv || !out || n_images <= 0 || tokens <= 0 || heads <= 0) return static_cast<int>(hipErrorInvalidValue);
    if ((reinterpret_cast<size_t>(qkv) & 15) || (reinterpret_cast<size_t>(out) & 15))
        return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nkt = (tokens + 31) / 32;
    switch (nkt) {
        case 1: return launch<1>(s, qkv, out, n_images, tokens, heads);
        case 2: return launch<2>(s, qkv, out, n_images, tokens, heads);
        case 3: return launch<3>(s, qkv, out, n_images, tokens, heads);
        case 4: return launch<4>(s, qkv, out, n_images, tokens, heads);
        case 5: return launch<5>(s, qkv, out, n_images, tokens, heads);
        case 6: return launch<6>(s, qkv, out, n_images, tokens, heads);
        case 7: return launch<7>(s, qkv, out, n_images, tokens, heads);
        default: return static_cast<int>(hipErrorInvalidValue);  // > 224 tokens: K/V tiling not built yet
    }
}
