// tools/ldsdma_probe.hip -- does a wave's global_load_lds_dwordx4 gather (per-lane global addresses, lane-linear LDS image)
// read back what the epilogue's old-state landing expects?  hipcc --offload-arch=gfx950 -O3 -o ldsdma_probe ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct alignas(16) f4 { float x, y, z, w; };
typedef __attribute__((address_space(3))) void lds_void;
__global__ void __launch_bounds__(256) probe(const float* __restrict__ src, float* __restrict__ dst, int pieces) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* chunk = lds + (size_t) wave * pieces * 1024;
    // lane's piece p comes from a scattered place: float4 index (p * 256 + (tid * 7) % 256)
    for (int p = 0; p < pieces; ++p) {
        const f4* g = reinterpret_cast<const f4*>(src) + (size_t) p * 256 + (tid * 7) % 256;
        __builtin_amdgcn_global_load_lds(g, (lds_void*) (chunk + p * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    for (int p = 0; p < pieces; ++p) {
        const f4 v = *reinterpret_cast<const f4*>(chunk + p * 1024 + lane * 16);
        reinterpret_cast<f4*>(dst)[(size_t) p * 256 + tid] = v;
    }
}
int main() {
    const int pieces = 16, n = pieces * 256 * 4;
    std::vector<float> h(n), o(n, -1.0f);
    for (int i = 0; i < n; ++i) h[i] = (float) i;
    float *ds, *dd;
    hipMalloc(&ds, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(ds, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 4 * pieces * 1024, 0, ds, dd, pieces);
    hipMemcpy(o.data(), dd, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int p = 0; p < pieces; ++p) for (int t = 0; t < 256; ++t) for (int c = 0; c < 4; ++c) {
        const float want = h[((size_t) p * 256 + (t * 7) % 256) * 4 + c];
        if (o[((size_t) p * 256 + t) * 4 + c] != want) ++bad;
    }
    printf("ldsdma probe: %d mismatches of %d (%s)\n", bad, n, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
