// tools/mfma_probe.hip -- v_mfma_f32_32x32x2_f32 on gfx950: operand / result layout and arithmetic, against a k-ordered fmaf chain on the host.
//   A[i][k] (32 x K), B[k][j] (K x 32):  D[i][j] = fma(A[i][K-1], B[K-1][j], ... fma(A[i][0], B[0][j], +0))
// layout assumed (and checked here): a-operand lane l = A[l % 32][2 s + l / 32], b-operand lane l = B[2 s + l / 32][l % 32],
// result register r of lane l = D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(const float* A, const float* B, float* D, int K) {
    const int l = threadIdx.x;
    f16v acc = {0};
    for (int s = 0; s < K / 2; ++s) {
        const float a = A[(l % 32) * K + 2 * s + l / 32], b = B[(2 * s + l / 32) * 32 + l % 32];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32] = acc[r];
}
int main() {
    const int K = 200;
    std::vector<float> A(32 * K), B(K * 32), D(32 * 32), W(32 * 32);
    uint32_t g = 7;
    auto rnd = [&]() { g = g * 1664525u + 1013904223u; return (float) (g >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : A) v = rnd() < 0.2f ? 0.0f : rnd();
    for (auto& v : B) v = rnd();
    for (int i = 0; i < 32; ++i) { A[i * K + 3] = 1e-41f; B[5 * 32 + i] = 3e-40f; }          // subnormal inputs / products
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float c = 0; for (int k = 0; k < K; ++k) c = fmaf(A[i * K + k], B[k * 32 + j], c); W[i * 32 + j] = c; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) if (memcmp(&D[i], &W[i], 4)) { if (bad < 5) printf("mismatch at [%d][%d]: %.9g vs %.9g\n", i / 32, i % 32, D[i], W[i]); ++bad; }
    printf("mfma_f32_32x32x2f32 vs k-ordered fmaf chain (K=%d, zeros and subnormals included): %d of 1024 differ\n", K, bad);
    return bad != 0;
}
