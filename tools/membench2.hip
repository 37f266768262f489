// tools/membench2.hip -- which frame -> workgroup assignment lets a do-nothing kernel stream the frame kernel's
// traffic mix (1 byte of s16 PCM read : 2 bytes of f32 spectra written) fastest on MI355X?
// Round 1 measured 5.1-5.3 TB/s for "one 16 KiB frame per workgroup trip, grid-strided" (tools/membench.hip) while a
// pure write reaches 4.0-6.2 TB/s depending on the chunk a workgroup writes at a time: the pattern matters.
// Knobs swept here: frames per workgroup trip (K consecutive frames), grid-strided vs contiguous-block assignment,
// frame size (N = 4096 / 8192 / 16384), workgroup count and size.
//   hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o tools/bin/membench2 && tools/bin/membench2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void cvt(const u4 u, f4& a, f4& b) {
    a.x = (float) (int16_t) (u.x & 0xffff); a.y = (float) (int16_t) (u.x >> 16);
    a.z = (float) (int16_t) (u.y & 0xffff); a.w = (float) (int16_t) (u.y >> 16);
    b.x = (float) (int16_t) (u.z & 0xffff); b.y = (float) (int16_t) (u.z >> 16);
    b.z = (float) (int16_t) (u.w & 0xffff); b.w = (float) (int16_t) (u.w >> 16);
}

// frame = fin 16-byte items of PCM -> 2 rows of fin items of floats each.  A trip = K consecutive frames.
// mode 0: trips dealt grid-strided; mode 1: every workgroup owns a contiguous block of trips;
// mode 2: grid-strided, but the workgroup's start offset inside the frame is rotated by its index (staggered rows)
template <int TPB>
__global__ void __launch_bounds__(TPB) k_frames(const u4* __restrict__ in, f4* __restrict__ out, uint32_t frames, uint32_t fin,
                                                uint32_t K, int mode) {
    const uint32_t trips = frames / K;
    const uint32_t per = (trips + gridDim.x - 1) / gridDim.x;
    const uint32_t items = fin * K;                 // 16-byte input items of one trip
    for (uint32_t i = 0; i < per; ++i) {
        const uint32_t t = mode == 1 ? blockIdx.x * per + i : blockIdx.x + i * gridDim.x;
        if (t >= trips) break;
        const u4* pin = in + (size_t) t * items;
        f4* pout = out + (size_t) t * items * 2;
        const uint32_t rot = mode == 2 ? (blockIdx.x * 4 * TPB) % items : 0;
        for (uint32_t j0 = 0; j0 < items; j0 += 4 * TPB) {
            u4 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { uint32_t q = j0 + j * TPB + threadIdx.x + rot; q = q >= items ? q - items : q; u[j] = pin[q]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t q = j0 + j * TPB + threadIdx.x + rot; q = q >= items ? q - items : q;
                const uint32_t f = q / fin, w = q % fin;     // frame within the trip, item within the frame
                f4 a, b; cvt(u[j], a, b);
                pout[(size_t) f * 2 * fin + w] = a;
                pout[(size_t) f * 2 * fin + fin + w] = b;
            }
        }
    }
}

// R16-like mix: 1 byte read : 1 byte written (s16 PCM in, unorm16 texels out)
template <int TPB>
__global__ void __launch_bounds__(TPB) k_frames_r16(const u4* __restrict__ in, u4* __restrict__ out, uint32_t frames, uint32_t fin, uint32_t K) {
    const uint32_t trips = frames / K, items = fin * K;
    for (uint32_t t = blockIdx.x; t < trips; t += gridDim.x) {
        const u4* pin = in + (size_t) t * items;
        u4* pout = out + (size_t) t * items;
        for (uint32_t j0 = 0; j0 < items; j0 += 4 * TPB) {
            u4 u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = pin[j0 + j * TPB + threadIdx.x];
#pragma unroll
            for (int j = 0; j < 4; ++j) { u4 v = u[j]; v.x ^= 0x8000u; pout[j0 + j * TPB + threadIdx.x] = v; }
        }
    }
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
    hipEventDestroy(e0); hipEventDestroy(e1);
    return t[t.size() / 2];
}

int main() {
    const size_t GiB = 1ull << 30;
    void *a, *b;
    if (hipMalloc(&a, GiB) != hipSuccess || hipMalloc(&b, 2 * GiB) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, GiB); hipMemset(b, 2, 2 * GiB);
    const double total = 3.0 * GiB;
    printf("# 1 GiB s16 PCM -> 2 GiB f32 (the traffic of BASELINE configs[1]); TB/s by pattern.  mode: 0 grid-strided trips, 1 contiguous block of trips per workgroup, 2 grid-strided + staggered start\n");
    for (uint32_t n : {4096u, 8192u, 16384u}) {
        const uint32_t fin = n * 4 / 16;                 // 16-byte items of PCM per frame
        const uint32_t frames = (uint32_t) (GiB / (n * 4));
        for (int mode : {0, 1, 2}) {
            for (uint32_t K : {1u, 2u, 4u, 16u}) {
                printf("N=%5u mode %d K=%2u :", n, mode, K);
                for (int grid : {512, 1024, 2048, 4096}) {
                    double ms = time_ms([&] { hipLaunchKernelGGL(k_frames<256>, dim3(grid), dim3(256), 0, nullptr, (const u4*) a, (f4*) b, frames, fin, K, mode); });
                    printf("  g%-5d %.2f", grid, total / ms / 1e9);
                }
                double ms = time_ms([&] { hipLaunchKernelGGL(k_frames<512>, dim3(1024), dim3(512), 0, nullptr, (const u4*) a, (f4*) b, frames, fin, K, mode); });
                printf("  | 512thr g1024 %.2f\n", total / ms / 1e9);
                fflush(stdout);
            }
        }
    }
    printf("# 1 GiB s16 PCM -> 1 GiB u16 texels (GLV_OP_R16 traffic, 1:1)\n");
    for (uint32_t n : {4096u, 16384u}) {
        const uint32_t fin = n * 4 / 16, frames = (uint32_t) (GiB / (n * 4));
        for (uint32_t K : {1u, 4u}) {
            printf("N=%5u K=%2u :", n, K);
            for (int grid : {512, 1024, 2048, 4096}) {
                double ms = time_ms([&] { hipLaunchKernelGGL(k_frames_r16<256>, dim3(grid), dim3(256), 0, nullptr, (const u4*) a, (u4*) b, frames, fin, K); });
                printf("  g%-5d %.2f", grid, 2.0 * GiB / ms / 1e9);
            }
            printf("\n");
        }
    }
    return 0;
}
