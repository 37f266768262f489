// glv_inst.hip -- production instantiations of glv_frame_kernel for ONE transform size.
// Compiled once per size with -DGLV_LOG_NN=k (k = log2(nn) = log2(N) - 1, 7..14) so the eight
// sizes build in parallel.  The knob set per size is the measured best of tools/tune.py
// (profiles/tune_r01_final.txt, earlier sweeps in profiles/tune_r01.txt); see DESIGN.md "Kernel configuration".
#include "glv_kernel_tmpl.h"
#include "glv_launch.h"

#ifndef GLV_LOG_NN
#error "compile with -DGLV_LOG_NN=<7..14>"
#endif

namespace glv {

template <int LOG_NN> struct Tuned;
#define GLV_TUNED(K, LE, S, NB, TR, WL, OC, PF, TL, WP, WPS, RD) \
    template <> struct Tuned<K> { static constexpr int log_e = LE, slots = S, nbuf = NB, occ = OC; \
                                  static constexpr bool winlds = WL; static constexpr int twreg = TR, tiltreg = TL, prefetch = PF, wpre = WP, wpre_s = WPS, rounds = RD; };
// measured best of tools/tune.py on MI355X (profiles/tune_r01_final.txt), equal bytes per size class:
//         log2(nn) LOG_E SLOTS NBUF TWREG  WINLDS OCC PREFETCH TILTREG WPRE WPRE_S ROUNDS   (knob values: glv_kernel_tmpl.h; ROUNDS: persistent workgroups launched = ROUNDS x what fits the chip)
GLV_TUNED(7,       3,    16,   1,   true,  true,  4,  1,       true,   0,   0,   2)    // N=256    E=8:  3+3+1 (16 lanes per row; not tuned: coverage of setbufsize 256)
GLV_TUNED(8,       3,    16,   1,   true,  true,  4,  1,       true,   0,   0,   2)    // N=512    E=8:  3+3+2
GLV_TUNED(9,       3,    4,    1,   true,  true,  4,  1,       true,   0,   0,   2)    // N=1024   E=8:  3+3+3
GLV_TUNED(10,      4,    4,    1,   true,  true,  2,  1,       true,   0,   0,   2)    // N=2048   E=16: 4+4+3, four 128-lane rows (r02 sweep_13: 0.627 vs 0.668 ms for E=8)
GLV_TUNED(11,      4,    2,    1,   true,  true,  2,  1,       true,   0,   0,   2)    // N=4096   E=16: 4+4+3
GLV_TUNED(12,      4,    2,    1,   true,  true,  2,  1,       true,   0,   0,   1)    // N=8192   E=16: 4+4+4; two slots share the 64 KiB LDS window
GLV_TUNED(13,      5,    1,    1,   2,     false, 2,  1,       2,      0,   0,   2)    // N=16384  E=32: 5+5+3; pass-1 twiddles from an 8 KiB LDS table, tilt computed
                                                                                  //          (WPRE, the window prefetch ahead of the stores, measured no gain: profiles/r02)
GLV_TUNED(14,      5,    1,    1,   2,     false, 2,  1,       2,      0,   0,   1)    // N=32768  E=32: 5+5+4, one 512-lane row per CU (135 KiB exchange region), pass-1 twiddles from LDS (sweep_13)
#undef GLV_TUNED

#define GLV_CAT2(a, b) a##b
#define GLV_CAT(a, b) GLV_CAT2(a, b)

template <int IN_MODE, int LOG_MODE>
static hipError_t launch_one(const FrameArgs& a, int grid, hipStream_t st) {
    using TU = Tuned<GLV_LOG_NN>;
    return launch_variant<GLV_LOG_NN, IN_MODE, LOG_MODE, TU::slots, TU::nbuf, TU::twreg, TU::winlds, TU::occ, TU::prefetch, TU::tiltreg, TU::log_e, TU::wpre, TU::wpre_s>(a, grid, st);
}

template <int IN_MODE>
static hipError_t launch_log(int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    switch (log_mode) {
        case 0: return launch_one<IN_MODE, 0>(a, grid, st);
        case 1: return launch_one<IN_MODE, 1>(a, grid, st);
        case 2: return launch_one<IN_MODE, 2>(a, grid, st);
    }
    return hipErrorInvalidValue;
}

hipError_t GLV_CAT(launch_frame_, GLV_LOG_NN)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    switch (in_mode) {
        case IN_S16_STEREO: return launch_log<IN_S16_STEREO>(log_mode, a, grid, st);
        case IN_S16_RING:   return launch_log<IN_S16_RING>(log_mode, a, grid, st);
        case IN_F32_PLANAR: return launch_log<IN_F32_PLANAR>(log_mode, a, grid, st);
        case IN_F32_STEREO: return launch_log<IN_F32_STEREO>(log_mode, a, grid, st);
        case IN_F32_RING:   return launch_log<IN_F32_RING>(log_mode, a, grid, st);
    }
    return hipErrorInvalidValue;
}

// lanes that cooperate on one row (fused bars need whole waves per row)
int GLV_CAT(frame_lanes_, GLV_LOG_NN)() { return Frame<GLV_LOG_NN, Tuned<GLV_LOG_NN>::log_e>::T; }

// workgroups of this size's production kernel that fit one CU (LDS and the launch-bounds wave budget)
int GLV_CAT(frame_resident_, GLV_LOG_NN)() {
    using TU = Tuned<GLV_LOG_NN>;
    using FR = Frame<GLV_LOG_NN, TU::log_e>;
    constexpr size_t lds = frame_lds_bytes<GLV_LOG_NN, TU::log_e, TU::slots, TU::nbuf, TU::winlds, TU::twreg>();
    constexpr int by_lds = (int) (160 * 1024 / lds);
    constexpr int waves = FR::T * TU::slots / 64 > 0 ? FR::T * TU::slots / 64 : 1;
    constexpr int by_waves = TU::occ * 4 / waves;
    constexpr int r = by_lds < by_waves ? by_lds : by_waves;
    return r > 0 ? r : 1;
}

// rounds of resident workgroups a large launch is cut into (glv_api.cpp frame_grid).  Two rounds even out CU-to-CU
// differences; the sizes whose ONE 512-thread workgroup per CU pays a 64 KiB window staging prologue with nothing else on
// the CU to overlap it run one round (N=8192, 32768 streams: 0.697 ms with 256 workgroups, 0.723 ms with 512).
int GLV_CAT(frame_rounds_, GLV_LOG_NN)() { return Tuned<GLV_LOG_NN>::rounds; }

// channel rows one workgroup takes per trip of its persistent loop (grid sizing): a pipelined s16
// slot takes a whole frame (2 rows), a single-slot workgroup both rows of its frame in sequence; for
// other inputs the figure only makes the grid slightly smaller than strictly necessary.
int GLV_CAT(frame_slots_, GLV_LOG_NN)() {
    using TU = Tuned<GLV_LOG_NN>;
    return (TU::prefetch == 1 || TU::slots == 1) ? 2 * TU::slots : TU::slots;
}

}  // namespace glv
