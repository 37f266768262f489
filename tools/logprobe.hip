// tools/logprobe.hip -- accuracy of the hardware log2 (v_log_f32) behind log_mode 1, exhaustively over
// every float y in [1, 2^14): max relative error of log2(y)*ln2/3 against the fp64 log(y)/3, per binade
// and for the first mantissa steps above 1.0 (where log(y) -> 0 and a relative bound is hardest).
//   hipcc --offload-arch=gfx950 -O2 tools/logprobe.hip -o tools/bin/logprobe && tools/bin/logprobe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void probe(uint32_t first, uint32_t count, double* max_rel, double* max_rel_corr) {
    double worst = 0, worst_c = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float y = __builtin_bit_cast(float, first + i);
        const double want = log((double) y) / 3.0;
        if (want == 0) continue;
        const float fast = __builtin_amdgcn_logf(y) * (float) (0.6931471805599453094 / 3.0);
        const double e = fabs(((double) fast - want) / want);
        if (e > worst) worst = e;
        // candidate correction: log1p polynomial below 1 + 2^-5
        const float x = y - 1.0f;
        float alt = fast;
        if (x < 0.03125f) {
            float p = fmaf(x, -0.25f, 1.0f / 3.0f);
            p = fmaf(x, p, -0.5f);
            p = fmaf(x, p, 1.0f);
            alt = (x * p) * (1.0f / 3.0f);
            // degree-5 term
        }
        const double ec = fabs(((double) alt - want) / want);
        if (ec > worst_c) worst_c = ec;
    }
    // block max via atomics on the bit pattern (positive doubles order like integers)
    atomicMax(reinterpret_cast<unsigned long long*>(max_rel), (unsigned long long) __double_as_longlong(worst));
    atomicMax(reinterpret_cast<unsigned long long*>(max_rel_corr), (unsigned long long) __double_as_longlong(worst_c));
}

static void run(const char* label, uint32_t first, uint32_t count) {
    double *d, h[2] = {0, 0};
    hipMalloc(&d, 16); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, nullptr, first, count, d, d + 1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipFree(d);
    printf("%-34s max rel err: hw log2 %.3e   with log1p poly below 1+2^-5 %.3e\n", label, h[0], h[1]);
}

int main() {
    const uint32_t one = 0x3f800000u;
    run("y = 1 + k*2^-23, k in [1,16)", one + 1, 15);
    run("k in [16,256)", one + 16, 240);
    run("k in [256,4096)", one + 256, 4096 - 256);
    run("k in [4096,65536)", one + 4096, 65536 - 4096);
    run("k in [65536, 2^18)   (y<1.03125)", one + 65536, (1u << 18) - 65536);
    run("y in [1.03125, 2)", one + (1u << 18), (1u << 23) - (1u << 18));
    for (int e = 1; e < 14; ++e) {
        char l[64]; snprintf(l, sizeof l, "y in [2^%d, 2^%d)", e, e + 1);
        run(l, one + ((uint32_t) e << 23), 1u << 23);
    }
    return 0;
}
