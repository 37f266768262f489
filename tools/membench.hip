// tools/membench.hip -- what HBM bandwidth a trivial streaming kernel reaches on this box for the
// traffic mix of the frame kernel (1 byte read : 2 bytes written, 1 GiB in / 2 GiB out, the sizes of
// BASELINE configs[1]), next to a 1:1 copy, a pure read and a pure write of the same total bytes.
// The frame kernel's roofline fraction is quoted against the 8 TB/s peak; this tool gives the
// achievable ceiling for its read/write mix.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/bin/membench && tools/bin/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// s16 pairs -> f32: 16 B in, 32 B out per item
__global__ void __launch_bounds__(256) k_convert(const u4* __restrict__ in, f4* __restrict__ out, size_t items) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t) gridDim.x * blockDim.x) {
        const u4 u = in[i];
        f4 a, b;
        a.x = (float) (int16_t) (u.x & 0xffff); a.y = (float) (int16_t) (u.x >> 16);
        a.z = (float) (int16_t) (u.y & 0xffff); a.w = (float) (int16_t) (u.y >> 16);
        b.x = (float) (int16_t) (u.z & 0xffff); b.y = (float) (int16_t) (u.z >> 16);
        b.z = (float) (int16_t) (u.w & 0xffff); b.w = (float) (int16_t) (u.w >> 16);
        out[2 * i] = a; out[2 * i + 1] = b;
    }
}
// same traffic, but written the way the frame kernel writes: each workgroup owns whole 16 KiB rows
__global__ void __launch_bounds__(256) k_convert_rows(const u4* __restrict__ in, f4* __restrict__ out, size_t rows) {
    // a "frame" = 16 KiB of PCM (1024 items) -> 32 KiB of floats
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const u4* pin = in + r * 1024;
        f4* pout = out + r * 2048;
        u4 u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = pin[j * 256 + threadIdx.x];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f4 a, b;
            a.x = (float) (int16_t) (u[j].x & 0xffff); a.y = (float) (int16_t) (u[j].x >> 16);
            a.z = (float) (int16_t) (u[j].y & 0xffff); a.w = (float) (int16_t) (u[j].y >> 16);
            b.x = (float) (int16_t) (u[j].z & 0xffff); b.y = (float) (int16_t) (u[j].z >> 16);
            b.z = (float) (int16_t) (u[j].w & 0xffff); b.w = (float) (int16_t) (u[j].w >> 16);
            pout[j * 256 + threadIdx.x] = a; pout[1024 + j * 256 + threadIdx.x] = b;
        }
    }
}
__global__ void __launch_bounds__(256) k_copy(const u4* __restrict__ in, u4* __restrict__ out, size_t items) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t) gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_read(const u4* __restrict__ in, u4* __restrict__ out, size_t items) {
    u4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t) gridDim.x * blockDim.x) acc ^= in[i];
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;      // never true in practice; keeps the loads
}
__global__ void __launch_bounds__(256) k_write(u4* __restrict__ out, size_t items) {
    const u4 v = {1, 2, 3, 4};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t) gridDim.x * blockDim.x) out[i] = v;
}

template <typename F> static double time_ms(F launch, int iters = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
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
    const size_t GiB = 1ull << 30;
    void *a, *b;
    if (hipMalloc(&a, 2 * GiB) != hipSuccess || hipMalloc(&b, 2 * GiB) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, 2 * GiB); hipMemset(b, 2, 2 * GiB);
    const double total = 3.0 * GiB;
    for (int grid : {1024, 2048, 4096, 16384}) {
        double ms;
        ms = time_ms([&] { hipLaunchKernelGGL(k_convert, dim3(grid), dim3(256), 0, nullptr, (const u4*) a, (f4*) b, GiB / 16); });
        printf("grid %5d  convert 1 GiB s16 -> 2 GiB f32 (1:2)      %.3f ms  %.2f TB/s\n", grid, ms, total / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_convert_rows, dim3(grid), dim3(256), 0, nullptr, (const u4*) a, (f4*) b, GiB / 16384); });
        printf("grid %5d  same, one 16 KiB frame per workgroup trip  %.3f ms  %.2f TB/s\n", grid, ms, total / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, nullptr, (const u4*) a, (u4*) b, (3 * GiB / 2) / 16); });
        printf("grid %5d  copy 1.5 GiB -> 1.5 GiB (1:1)              %.3f ms  %.2f TB/s\n", grid, ms, total / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, nullptr, (const u4*) a, (u4*) b, (2 * GiB) / 16); });
        printf("grid %5d  read 2 GiB                                 %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * GiB / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, nullptr, (u4*) b, (2 * GiB) / 16); });
        printf("grid %5d  write 2 GiB                                %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * GiB / ms / 1e9);
    }
    return 0;
}
