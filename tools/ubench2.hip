// tools/ubench2.hip -- do two waves on one gfx950 SIMD share ONE vector-ALU issue slot whatever they execute, or do
// fp64 / conversions / transcendentals run beside packed fp32?  One 512-thread workgroup per CU = two waves per SIMD;
// waves 0-3 run instruction kind A, waves 4-7 kind B, each ITERS x 8 instructions.  If the kinds shared nothing the
// mixed time would be max(tA, tB); if they share the issue slot it is (tA + tB) / 2 of the pure pairs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/bin/ubench2 && tools/bin/ubench2
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 4096
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ float body(float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = seed * 0.5 + 1.0;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a0}, p3 = {a3, a2}, pb = {1.0001f, 0.9999f};
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == 0) {          // v_pk_mul_f32 x8
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (KIND == 1) {   // v_mul_f64 x8
            asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
        } else if (KIND == 2) {   // cvt f32->f64 x4 + cvt f64->f32 x4
            asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n"
                         "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if (KIND == 3) {   // v_log_f32 x8
            asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                         "v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (KIND == 4) {   // v_pk_fma_f32 x8
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                         "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (KIND == 5) {   // ds_read_b64 x8 (LDS beside the ALU)
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "ds_read_b64 %0, %4 offset:2048\n ds_read_b64 %1, %4 offset:2560\n ds_read_b64 %2, %4 offset:3072\n ds_read_b64 %3, %4 offset:3584\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((unsigned) (threadIdx.x & 63) * 8u));
        }
    }
    return a0 + a1 + a2 + a3 + (float) (d0 + d1 + d2 + d3) + p0.x + p1.y + p2.x + p3.y;
}

template <int KA, int KB>
__global__ void __launch_bounds__(512) k(float* out, float seed) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 512] = seed;
    __syncthreads();
    float r;
    if (threadIdx.x < 256) r = body<KA>(seed); else r = body<KB>(seed);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds[threadIdx.x];
}

template <int KA, int KB>
float run(float* d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(512), 0, 0, d_out, 1.0f);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(512), 0, 0, d_out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * 512);
    const char* names[6] = {"v_pk_mul_f32", "v_mul_f64", "cvt f32<->f64", "v_log_f32", "v_pk_fma_f32", "ds_read_b64"};
    float t[6][6] = {};
    (void) run<0, 0>(d_out);   // clocks
    t[0][0] = run<0, 0>(d_out); t[1][1] = run<1, 1>(d_out); t[2][2] = run<2, 2>(d_out); t[3][3] = run<3, 3>(d_out);
    t[4][4] = run<4, 4>(d_out); t[5][5] = run<5, 5>(d_out);
    t[0][1] = run<0, 1>(d_out); t[0][2] = run<0, 2>(d_out); t[0][3] = run<0, 3>(d_out); t[0][4] = run<0, 4>(d_out); t[0][5] = run<0, 5>(d_out);
    t[1][2] = run<1, 2>(d_out); t[1][3] = run<1, 3>(d_out);
    const double per = 1e6 / (double) (ITERS * 8) ;   // ms -> ns per wave instruction when ONE wave of the pair issues alone
    printf("two waves per SIMD, %d x 8 instructions per wave; ns per instruction pair-slot (time / instructions of one wave)\n", ITERS);
    for (int a = 0; a < 6; ++a) printf("  %-14s + %-14s : %.3f ms  (%.2f ns per instruction of each wave)\n", names[a], names[a], t[a][a], t[a][a] * per);
    const int pa[7] = {0, 0, 0, 0, 0, 1, 1}, pb[7] = {1, 2, 3, 4, 5, 2, 3};
    for (int q = 0; q < 7; ++q) {
        const int a = pa[q], b = pb[q];
        const float mx = t[a][a] > t[b][b] ? t[a][a] : t[b][b];
        printf("  %-14s + %-14s : %.3f ms   shared slot would give %.3f, independent pipes %.3f\n", names[a], names[b], t[a][b],
               0.5f * (t[a][a] + t[b][b]), 0.5f * mx > 0 ? mx * 0.5f + 0.0f : 0.0f);
    }
    return 0;
}
