// tools/membench3.hip -- does the width of the PCM loads matter?  The frame kernel reads its s16 PCM with 8 bytes per lane
// (one complex point of both channels) and writes 16 bytes per lane.  Same 1 GiB -> 2 GiB convert as membench2 (one 16 KiB
// frame per 256-thread workgroup trip, K = 4 frames per trip), with 4-, 8- and 16-byte loads, and pure reads of each width.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 cv(uint32_t a, uint32_t b) { f4 r; r.x = (float) (int16_t) (a & 0xffff); r.y = (float) (int16_t) (a >> 16); r.z = (float) (int16_t) (b & 0xffff); r.w = (float) (int16_t) (b >> 16); return r; }
template <int W> struct LT; template <> struct LT<16> { typedef u4 T; }; template <> struct LT<8> { typedef u2 T; }; template <> struct LT<4> { typedef uint32_t T; };
// W bytes per lane per load; 16 loads of 8 B in flight per lane like the kernel's A phase (scaled for the other widths)
template <int W>
__global__ void __launch_bounds__(256) k_conv(const char* __restrict__ in, f4* __restrict__ out, uint32_t chunks) {
    typedef typename LT<W>::T T;
    constexpr int NL = 64 / W * 2;                 // loads per lane per 32 KiB chunk of input (256 lanes)
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const T* pin = reinterpret_cast<const T*>(in + (size_t) c * 32768);
        f4* pout = out + (size_t) c * 4096;         // 64 KiB of floats
        T v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = pin[j * 256 + threadIdx.x];
        uint32_t w[NL * W / 4];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            if constexpr (W == 16) { w[4 * j] = v[j].x; w[4 * j + 1] = v[j].y; w[4 * j + 2] = v[j].z; w[4 * j + 3] = v[j].w; }
            else if constexpr (W == 8) { w[2 * j] = v[j].x; w[2 * j + 1] = v[j].y; }
            else w[j] = v[j];
        }
#pragma unroll
        for (int j = 0; j < NL * W / 8; ++j) pout[j * 256 + threadIdx.x] = cv(w[2 * j], w[2 * j + 1]);
    }
}
template <int W>
__global__ void __launch_bounds__(256) k_read(const char* __restrict__ in, uint32_t* __restrict__ out, uint32_t chunks) {
    typedef typename LT<W>::T T;
    constexpr int NL = 64 / W * 2;
    uint32_t acc = 0;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const T* pin = reinterpret_cast<const T*>(in + (size_t) c * 32768);
        T v[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = pin[j * 256 + threadIdx.x];
#pragma unroll
        for (int j = 0; j < NL; ++j) { if constexpr (W == 16) acc ^= v[j].x ^ v[j].w; else if constexpr (W == 8) acc ^= v[j].x ^ v[j].y; else acc ^= v[j]; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <typename F> static double time_ms(F launch, int iters = 20) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    for (int i = 0; i < 30; ++i) launch();
    std::vector<float> t;
    for (int rep = 0; rep < 5; ++rep) {
        (void) hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) launch();
        (void) hipEventRecord(e1, nullptr); (void) hipEventSynchronize(e1);
        float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1); t.push_back(ms / iters);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
int main() {
    const size_t GiB = 1ull << 30;
    void *a, *b;
    if (hipMalloc(&a, GiB) != hipSuccess || hipMalloc(&b, 2 * GiB) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void) hipMemset(a, 1, GiB); (void) hipMemset(b, 2, 2 * GiB);
    const uint32_t chunks = (uint32_t) (GiB / 32768);
    for (int grid : {1024, 2048}) {
        double m16 = time_ms([&] { hipLaunchKernelGGL(k_conv<16>, dim3(grid), dim3(256), 0, nullptr, (const char*) a, (f4*) b, chunks); });
        double m8 = time_ms([&] { hipLaunchKernelGGL(k_conv<8>, dim3(grid), dim3(256), 0, nullptr, (const char*) a, (f4*) b, chunks); });
        double m4 = time_ms([&] { hipLaunchKernelGGL(k_conv<4>, dim3(grid), dim3(256), 0, nullptr, (const char*) a, (f4*) b, chunks); });
        printf("grid %d convert 1 GiB -> 2 GiB: 16-byte loads %.3f ms %.2f TB/s | 8-byte loads %.3f ms %.2f TB/s | 4-byte loads %.3f ms %.2f TB/s\n", grid,
               m16, 3.0 * GiB / m16 / 1e9, m8, 3.0 * GiB / m8 / 1e9, m4, 3.0 * GiB / m4 / 1e9);
        double r16 = time_ms([&] { hipLaunchKernelGGL(k_read<16>, dim3(grid), dim3(256), 0, nullptr, (const char*) a, (uint32_t*) b, chunks); });
        double r8 = time_ms([&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, nullptr, (const char*) a, (uint32_t*) b, chunks); });
        double r4 = time_ms([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, nullptr, (const char*) a, (uint32_t*) b, chunks); });
        printf("grid %d read 1 GiB: 16-byte %.3f ms %.2f TB/s | 8-byte %.3f ms %.2f TB/s | 4-byte %.3f ms %.2f TB/s\n", grid, r16, GiB / r16 / 1e9, r8, GiB / r8 / 1e9, r4, GiB / r4 / 1e9);
    }
    return 0;
}
