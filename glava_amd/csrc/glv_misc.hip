// glv_misc.hip -- the small kernels and the size dispatch.
//
//   glv_post_kernel    gravity / average / wrange on spectra already in HBM (the single-op
//                      drop-ins glv_gravity, glv_average, glv_wrange; glava/render.c:720-781)
//   glv_unpack_kernel  s16 interleaved -> planar f32 (glv_unpack_s16; glava/fifo.c:94-110)
#include <hip/hip_runtime.h>

#include "glv_frame.h"
#include "glv_launch.h"

namespace glv {

// One lane owns one pair of floats (8 B) of a row and runs the same state machine as the
// fused epilogue (apply_state).
__global__ void __launch_bounds__(256) glv_post_kernel(const FrameArgs a, const uint32_t n) {
    const size_t pairs_per_row = n / 2;
    const size_t total = (size_t) a.units * pairs_per_row;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t) gridDim.x * blockDim.x) {
        const size_t row = i / pairs_per_row;
        const uint32_t off = (uint32_t) (i % pairs_per_row) * 8u;     // byte offset of the pair in its row
        cf val = ld<cf>(static_cast<const float*>(a.in) + row * n, off);
        if (a.ops & OP_WRANGE) {                                                  // render.c:777-779
            const float p = val.x + 1.0f, q = val.y + 1.0f;
            val.x = p / 2.0f; val.y = q / 2.0f;
        }
        val = apply_state(val, off, row, n, a);
        st<cf>(a.out + row * n, off, val);
    }
}

__global__ void __launch_bounds__(256) glv_unpack_kernel(const int16_t* __restrict__ pcm, size_t frames, int mono,
                                                         float* __restrict__ l, float* __restrict__ r) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += (size_t) gridDim.x * blockDim.x) {
        const uint32_t u = reinterpret_cast<const uint32_t*>(pcm)[i];
        const int a = (int16_t) (u & 0xffffu), b = (int16_t) (u >> 16);
        if (mono) { const float s = unpack_s16_mono(a, b); l[i] = s; r[i] = s; }
        else { l[i] = unpack_s16(a); r[i] = unpack_s16(b); }
    }
}

static int capped_grid(size_t items, int block) {
    size_t g = (items + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;     // 256 CUs x 8 resident 256-thread blocks, grid-stride beyond
    if (g < 1) g = 1;
    return (int) g;
}

hipError_t launch_post(const FrameArgs& a, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(glv_post_kernel, dim3(capped_grid((size_t) a.units * (n / 2), 256)), dim3(256), 0, st, a, n);
    return hipGetLastError();
}

hipError_t launch_unpack(const int16_t* pcm, size_t frames, int mono, float* l, float* r, hipStream_t st) {
    hipLaunchKernelGGL(glv_unpack_kernel, dim3(capped_grid(frames, 256)), dim3(256), 0, st, pcm, frames, mono, l, r);
    return hipGetLastError();
}

hipError_t launch_frame(int log_nn, int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    switch (log_nn) {
        case 8:  return launch_frame_8(in_mode, log_mode, a, grid, st);
        case 9:  return launch_frame_9(in_mode, log_mode, a, grid, st);
        case 10: return launch_frame_10(in_mode, log_mode, a, grid, st);
        case 11: return launch_frame_11(in_mode, log_mode, a, grid, st);
        case 12: return launch_frame_12(in_mode, log_mode, a, grid, st);
        case 13: return launch_frame_13(in_mode, log_mode, a, grid, st);
    }
    return hipErrorInvalidValue;
}

int frame_slots(int log_nn) {
    switch (log_nn) {
        case 8: return frame_slots_8(); case 9: return frame_slots_9(); case 10: return frame_slots_10();
        case 11: return frame_slots_11(); case 12: return frame_slots_12(); case 13: return frame_slots_13();
    }
    return 1;
}

}  // namespace glv
