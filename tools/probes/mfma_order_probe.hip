// Does v_mfma_f32_16x16x4_f32 add its four k-products in the same order, with the same roundings, as two consecutive
// v_mfma_f32_32x32x2_f32?  (If so a 16x16 latency tile is bit-identical to the 32x32 tiles of the fp32 GEMM.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_order_probe.hip -o gpurun_out/mfma_order_probe && ./gpurun_out/mfma_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *A, const float *B, int K, float *c32, float *c16, float *cseq) {
    const int l = threadIdx.x;
    // 32x32x2: whole 32x32 tile, k in pairs
    f32x16 acc = {};
    for (int k = 0; k < K; k += 2) {
        const float a = A[(l & 31) * K + k + (l >> 5)], b = B[(l & 31) * K + k + (l >> 5)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 16; ++v) c32[((v & 3) + 8 * (v >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[v];
    // 16x16x4: four quadrants, k in quads
    for (int qm = 0; qm < 2; ++qm)
        for (int qn = 0; qn < 2; ++qn) {
            f32x4 d = {};
            for (int k = 0; k < K; k += 4) {
                const float a = A[(16 * qm + (l & 15)) * K + k + (l >> 4)], b = B[(16 * qn + (l & 15)) * K + k + (l >> 4)];
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
            }
            for (int v = 0; v < 4; ++v) c16[(16 * qm + 4 * (l >> 4) + v) * 32 + 16 * qn + (l & 15)] = d[v];
        }
    // plain sequential fma chain per output (lane l computes outputs l, l + 64, ...)
    for (int o = l; o < 1024; o += 64) {
        const int m = o >> 5, n = o & 31;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[m * K + k], B[n * K + k], s);
        cseq[o] = s;
    }
}

int main() {
    const int K = 3072;
    std::vector<float> A(32 * K), B(32 * K);
    srand(7);
    for (auto &x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    for (auto &x : B) x = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float *dA, *dB, *d32, *d16, *dseq;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d32, 4096); hipMalloc(&d16, 4096); hipMalloc(&dseq, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, K, d32, d16, dseq);
    std::vector<float> c32(1024), c16(1024), cseq(1024);
    hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(c16.data(), d16, 4096, hipMemcpyDeviceToHost);
    if (hipMemcpy(cseq.data(), dseq, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 2; }
    int d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < 1024; ++i) {
        d1 += memcmp(&c32[i], &c16[i], 4) != 0;
        d2 += memcmp(&c32[i], &cseq[i], 4) != 0;
        d3 += memcmp(&c16[i], &cseq[i], 4) != 0;
    }
    printf("{\"K\": %d, \"differ_32x32x2_vs_16x16x4\": %d, \"differ_32x32x2_vs_fma_chain\": %d, \"differ_16x16x4_vs_fma_chain\": %d, \"of\": 1024}\n", K, d1, d2, d3);
    return 0;
}
