// glv_launch.h -- host-visible launch entry points of the kernel translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "glv_frame.h"

namespace glv {

// what the host needs to know about one kernel configuration of one size (glv_inst.hip Tuned<K, V>)
struct FrameGeometry {
    int lanes;          // lanes cooperating on one row
    int resident;       // workgroups that fit one CU
    int rounds;         // rounds of resident workgroups a large launch is cut into
    int rows_per_trip;  // channel rows one workgroup takes per trip of its persistent loop
    int bar_batch;      // steps per batch of the fused GLV_OP_BARS loop: its work lists are padded to multiples of this
    int lds_bytes, log_e, slots, twreg, winlds;   // for diagnostics / the wisdom file's comments
    int nbuf;           // exchange regions per row (0: split exchange -- no room to park a finished row: bars are a second launch)
    int live_points;    // complex points [0, live_points) of a row are what the GLV_OP_BARS_ONLY kernel class of this configuration keeps alive (a compile-time
                        // share of the last pass's blocks, glv_frame.h LIVE_RBLOCKS): the host takes that class only when the bars sample nothing beyond
};

// per-size production launchers, one translation unit each (glv_inst.hip -DGLV_LOG_NN=k)
#define GLV_DECL_INST(K) \
    hipError_t launch_frame_##K(int in_mode, int log_mode, int variant, const FrameArgs& a, int grid, hipStream_t st); \
    int frame_variants_##K(); \
    int frame_variant_ok_##K(int in_mode, int log_mode, int variant); \
    FrameGeometry frame_geometry_##K(int variant);
GLV_DECL_INST(7) GLV_DECL_INST(8) GLV_DECL_INST(9) GLV_DECL_INST(10) GLV_DECL_INST(11) GLV_DECL_INST(12) GLV_DECL_INST(13) GLV_DECL_INST(14)
#undef GLV_DECL_INST

// glv_misc.hip
hipError_t launch_frame(int log_nn, int in_mode, int log_mode, int variant, const FrameArgs& a, int grid, hipStream_t st);
int frame_variants(int log_nn);                                        // kernel configurations built for this size (>= 1)
bool frame_variant_ok(int log_nn, int in_mode, int log_mode, int variant);   // is `variant` built for this input / log mode
FrameGeometry frame_geometry(int log_nn, int variant);
hipError_t launch_post(const FrameArgs& a, uint32_t n, hipStream_t st);
hipError_t launch_bufscale(const float* in, float* out, size_t total_out, uint32_t k, hipStream_t st);
hipError_t launch_lerp(const float* s0, const float* e0, float* out, size_t total, float mod, hipStream_t st);
hipError_t launch_smooth(float* rows, size_t nrows, uint32_t n, const int* smin, const int* smax, uint32_t asz, uint32_t reach,
                         uint32_t max_window, hipStream_t st);
// the s16 window table as float pairs (glv_core.h WinSplit): split = float [n/2][4]; d_fail_shifted = int [2], zeroed by the caller
hipError_t launch_window_split(const double* w_tab, float* split, uint32_t n, int* d_fail_shifted, hipStream_t st);
hipError_t launch_window_split_check(const double* w_tab, const float* split, uint32_t n, unsigned long long* d_mismatches, hipStream_t st);
// rt: the tables of the many-bars kernels (>= 256 bars; glv_tables.h make_bar_mtiles): tiles of 32 bars with their weights, the bars'
// weight sums, and -- when they could be cut -- the rounds of the matrix-core kernel for an LDS ring of ring_bins (160, 288, 448 or 832) bins
// mode != 0 (glv_params.sample_mode maximum / hybrid): the block-transposed weights of glv_bars_mode_kernel (glv_tables.h make_bar_mode_blocks) and the
// bins of a row its bars sample; the other tables are then unused
struct BarRowsTables {
    const BarMTile* mtiles; uint32_t ntiles; const float* wt; const float* wsum;
    const BarTile* rounds; uint32_t nrounds, ring_bins;
    uint32_t mode = 0; float hybrid_weight = 0.65f; const BarModeBlock* mblocks = nullptr; uint32_t nmblocks = 0; const float* mw = nullptr; uint32_t mode_bins = 0;
};
hipError_t launch_bars(const float* spec, float* bars_out, size_t nrows, uint32_t n, uint32_t bars, uint32_t nsteps,
                       const BarItem* items, const BarDesc* desc, const float* tap_w, hipStream_t st, bool r16 = false, const BarRowsTables* rt = nullptr);
hipError_t prepare_bars_rows(uint32_t n, const BarRowsTables* rt);      // function attributes of the kernel launch_bars would pick
// the tables of the i8 matrix-core kernel for texel rows (glv_tables.h make_bar_itiles): tiles (origin a multiple of 16 bins, steps of 32),
// the digit planes of the integer weights in operand layout, per bar the rounding constant and shift, the rounds for a ring of ring_bins bins
struct BarIRowsTables {
    const BarMTile* tiles; uint32_t ntiles; const void* wq; const BarIFin* fin;
    const BarTile* rounds; uint32_t nrounds, ring_bins;
};
hipError_t launch_bars_i8(const void* rows, bool rows_f32, void* bars_out, size_t nrows, uint32_t n, uint32_t bars, const BarIRowsTables* rt, hipStream_t st, bool r16);
hipError_t prepare_bars_i8(uint32_t n, const BarIRowsTables* rt);     // function attributes of the kernels launch_bars_i8 would pick
hipError_t launch_ring_planar(const void* ring, int is_f32, uint32_t n, uint32_t rot, int mono, size_t streams, float* out, hipStream_t st);
hipError_t launch_unpack(const int16_t* pcm, size_t frames, int mono, float* l, float* r, hipStream_t st);

}  // namespace glv
