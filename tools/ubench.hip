// tools/ubench.hip -- VALU issue-rate microbenchmark for gfx950 (what does one wave64
// instruction of each kind cost on a SIMD?).  Informs the kernel's instruction selection
// (packed vs plain f32, fp64 window product, fp64 log polynomial, transcendentals).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define ITERS 2048

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    float b = seed * 0.5f + 1.0f;
    double db = b;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b};
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == 0) {   // v_mul_f32 x8 independent
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 1) {   // v_pk_mul_f32 x4 (8 flops-lanes like KIND 0)
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (KIND == 2) {   // v_mul_f64 x8
            asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
        } else if (KIND == 3) {   // v_fma_f64 x8
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                         "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
        } else if (KIND == 4) {   // cvt f32->f64 x4 + cvt f64->f32 x4
            asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n"
                         "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if (KIND == 5) {   // v_log_f32 x8
            asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                         "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 6) {   // v_fma_f32 x8
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if (KIND == 7) {   // v_pk_add_f32 with neg modifier x8
            asm volatile("v_pk_add_f32 %0, %0, %4 neg_lo:[0,1]\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4 neg_lo:[0,1]\n v_pk_add_f32 %3, %3, %4\n"
                         "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4 neg_lo:[0,1]\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4 neg_lo:[0,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (KIND == 8) {   // v_mov_b32 x8
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                         "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 9) {   // v_cvt_f32_i32 + v_add_u32 mix x8
            asm volatile("v_cvt_f32_i32 %0, %1\n v_add_u32 %1, %1, %2\n v_cvt_f32_i32 %2, %3\n v_add_u32 %3, %3, %4\n"
                         "v_cvt_f32_i32 %4, %5\n v_add_u32 %5, %5, %6\n v_cvt_f32_i32 %6, %7\n v_add_u32 %7, %7, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 10) {   // v_add_f64 x8
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float) (d0 + d1 + d2 + d3) + p0.x + p1.y + p2.x + p3.y;
}

template <int KIND>
void run(const char* name, int instr_per_iter, float* d_out, int waves_per_simd) {
    const int cus = 256;
    dim3 grid(cus * waves_per_simd), block(256);   // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d_out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d_out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves each issuing ITERS*instr_per_iter instructions
    const double instr = (double) waves_per_simd * ITERS * instr_per_iter;
    const double ns_per = ms * 1e6 / instr;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %5.2f cycles @2.4GHz, %5.2f @2.0GHz)\n",
           name, waves_per_simd, ms, ns_per, ns_per * 2.4, ns_per * 2.0);
}

int main() {
    float* d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256);
    for (int w : {1, 2, 4}) {
        run<0>("v_mul_f32", 8, d_out, w);
        run<6>("v_fma_f32", 8, d_out, w);
        run<1>("v_pk_mul_f32", 8, d_out, w);
        run<7>("v_pk_add_f32 (neg_lo)", 8, d_out, w);
        run<2>("v_mul_f64", 8, d_out, w);
        run<10>("v_add_f64", 8, d_out, w);
        run<3>("v_fma_f64", 8, d_out, w);
        run<4>("v_cvt_f64_f32 / v_cvt_f32_f64", 8, d_out, w);
        run<5>("v_log_f32", 8, d_out, w);
        run<8>("v_mov_b32", 8, d_out, w);
        run<9>("v_cvt_f32_i32 / v_add_u32", 8, d_out, w);
    }
    return 0;
}
