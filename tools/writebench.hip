// tools/writebench.hip -- how the HBM write rate of MI355X depends on the store pattern: every workgroup trip
// writes one contiguous chunk of CH bytes (256 threads, 16 B per lane per store), chunks handed out grid-stride.
// The frame kernel's spectrum stores are 16 KiB rows written by 128 lanes; this shows what other shapes reach.
//   hipcc --offload-arch=gfx950 -O3 tools/writebench.hip -o tools/bin/writebench && tools/bin/writebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_write_chunks(u4* __restrict__ out, size_t chunks, int items_per_chunk) {
    const u4 v = {1, 2, 3, 4};
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        u4* p = out + c * (size_t) items_per_chunk;
        for (int i = threadIdx.x; i < items_per_chunk; i += THREADS) p[i] = v;
    }
}
template <typename F> static double time_ms(F launch, int iters = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms / iters);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
int main() {
    const size_t GiB = 1ull << 30, total = 2 * GiB;
    void* b; if (hipMalloc(&b, total) != hipSuccess) return 1;
    hipMemset(b, 0, total);
    printf("pure write of 2 GiB, TB/s by (chunk bytes per workgroup trip) x (grid)  [256-thread workgroups | 128-thread]\n");
    for (int chunk : {1024, 4096, 16384, 32768, 65536, 262144, 1048576}) {
        printf("chunk %7d:", chunk);
        for (int grid : {512, 1024, 2048, 4096, 16384}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_write_chunks<256>, dim3(grid), dim3(256), 0, nullptr, (u4*) b, total / chunk, chunk / 16); });
            printf("  g%-5d %.2f", grid, total / ms / 1e9);
        }
        printf("  |");
        for (int grid : {1024, 4096}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_write_chunks<128>, dim3(grid), dim3(128), 0, nullptr, (u4*) b, total / chunk, chunk / 16); });
            printf("  g%-5d %.2f", grid, total / ms / 1e9);
        }
        printf("\n");
    }
    return 0;
}
