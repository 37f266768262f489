// glv_launch.h -- host-visible launch entry points of the kernel translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "glv_frame.h"

namespace glv {

// per-size production launchers, one translation unit each (glv_inst.hip -DGLV_LOG_NN=k)
#define GLV_DECL_INST(K) \
    hipError_t launch_frame_##K(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st); \
    int frame_slots_##K();
GLV_DECL_INST(8) GLV_DECL_INST(9) GLV_DECL_INST(10) GLV_DECL_INST(11) GLV_DECL_INST(12) GLV_DECL_INST(13)
#undef GLV_DECL_INST

// glv_misc.hip
hipError_t launch_frame(int log_nn, int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st);
int frame_slots(int log_nn);      // channel rows one workgroup takes per iteration of its persistent loop
hipError_t launch_post(const FrameArgs& a, uint32_t n, hipStream_t st);
hipError_t launch_unpack(const int16_t* pcm, size_t frames, int mono, float* l, float* r, hipStream_t st);

}  // namespace glv
