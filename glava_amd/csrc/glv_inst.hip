// glv_inst.hip -- production instantiations of glv_frame_kernel for ONE transform size.
// Compiled per size with -DGLV_LOG_NN=k (k = log2(nn) = log2(N) - 1, 7..14) and per PART with -DGLV_INST_PART=p so that the
// 24 translation units build in parallel and balance over the cores (the large sizes compile slowest):
//   part 0  s16 frames and the s16 ring, configuration 0; the size's dispatcher and geometry functions
//   part 1  the f32 inputs (planar rows, interleaved stereo, the f32 ring), configuration 0
//   part 2  configuration 1 (the runner-up plan of the size), every input
// The knob set per size is the measured best of tools/tune.py
// (profiles/tune_r01_final.txt, earlier sweeps in profiles/tune_r01.txt); see DESIGN.md "Kernel configuration".
#include "glv_kernel_tmpl.h"
#include "glv_launch.h"

#ifndef GLV_LOG_NN
#error "compile with -DGLV_LOG_NN=<7..14>"
#endif
#ifndef GLV_INST_PART
#error "compile with -DGLV_INST_PART=<0..2>"
#endif

namespace glv {

// Tuned<K, V>: kernel configuration V of size K.  V = 0 is the measured best of tools/tune.py on the MI355X the sweeps ran on
// (profiles/tune_r01_final.txt and the r02 sweeps); V >= 1 are the runners-up that came within a few per cent there -- a
// different plan (points per lane, slots per workgroup, where the tables live), not just a different grid.  Which one wins
// depends on the operator chain, the log mode and the part (CU count, clocks, LDS per CU), so the choice is made at run time:
// glv_batch_autotune times every candidate on the batch's device and records (variant, workgroups) in the launch wisdom
// (glv_api.cpp) -- the role of glfft's FFTWisdom::study (glfft/glfft_wisdom.cpp:235-446: work-group shape and radix split per
// transform, timed on the target).
template <int LOG_NN, int V> struct Tuned;
template <int LOG_NN> struct NumVariants { static constexpr int value = 1; };
#define GLV_TUNED(K, V, LE, S, NB, TR, WL, OC, PF, TL, WP, WPS, RD) \
    template <> struct Tuned<K, V> { static constexpr int log_e = LE, slots = S, nbuf = NB, occ = OC; \
                                     static constexpr bool winlds = WL; static constexpr int twreg = TR, tiltreg = TL, prefetch = PF, wpre = WP, wpre_s = WPS, rounds = RD; };
#define GLV_NVARIANTS(K, NV) template <> struct NumVariants<K> { static constexpr int value = NV; };
//     log2(nn) V  LOG_E SLOTS NBUF TWREG  WINLDS OCC PREFETCH TILTREG WPRE WPRE_S ROUNDS   (knob values: glv_kernel_tmpl.h; ROUNDS: persistent workgroups launched = up to ROUNDS x what fits the chip.
//     Round 4 (tools/grid_ab.py, profiles/r04/grid_ab.txt): N <= 4096 runs 1-4 % faster in EVERY kernel class with eight rounds than with two -- the hardware hands the next
//     workgroup to whichever CU finishes first, which evens out the tail a two-round launch ends on; N = 8192 / 32768 keep one round (a 64 KiB window staging prologue per workgroup))
GLV_TUNED(7,    0, 3,    16,   1,   true,  true,  4,  1,       true,   0,   0,   8)    // N=256    E=8:  3+3+1 (16 lanes per row; not tuned: coverage of setbufsize 256)
GLV_TUNED(8,    0, 3,    16,   1,   true,  true,  4,  1,       true,   0,   0,   8)    // N=512    E=8:  3+3+2
GLV_TUNED(8,    1, 4,    16,   1,   true,  true,  2,  1,       true,   0,   0,   8)    //          E=16: 4+4, one exchange less (0.629 vs 0.598 ms in the r01 sweep; equal with log_mode 0)
GLV_TUNED(9,    0, 3,    4,    1,   true,  true,  4,  1,       true,   0,   0,   8)    // N=1024   E=8:  3+3+3
GLV_TUNED(9,    1, 3,    2,    1,   true,  true,  4,  1,       true,   0,   0,   8)    //          two rows per workgroup instead of four (smaller workgroups, more of them)
GLV_TUNED(10,   0, 4,    4,    1,   true,  true,  2,  1,       true,   0,   0,   8)    // N=2048   E=16: 4+4+3, four 128-lane rows (r02 sweep_13: 0.627 vs 0.668 ms for E=8)
GLV_TUNED(10,   1, 3,    4,    1,   true,  true,  4,  1,       true,   0,   0,   8)    //          E=8:  3+3+3+1, four 256-lane rows, 4 waves per SIMD (production until round 2)
GLV_TUNED(11,   0, 4,    2,    1,   true,  true,  2,  1,       true,   0,   0,   8)    // N=4096   E=16: 4+4+3
GLV_TUNED(11,   1, 3,    2,    1,   true,  true,  3,  1,       true,   0,   0,   8)    //          E=8:  3+3+3+2, 3 waves per SIMD (0.672 vs 0.647 ms; 0.917 vs 0.909 with log_mode 0)
GLV_TUNED(12,   0, 4,    2,    1,   true,  true,  2,  1,       true,   0,   0,   1)    // N=8192   E=16: 4+4+4; two slots share the 64 KiB LDS window
GLV_TUNED(12,   1, 4,    1,    1,   true,  false, 2,  1,       true,   0,   0,   2)    //          one row per workgroup, two workgroups per CU, window through L2 (tie with log_mode 0 in the r01 sweep)
GLV_TUNED(13,   0, 5,    1,    1,   2,     false, 2,  1,       3,      0,   0,   4)    // N=16384  E=32: 5+5+3; pass-1 twiddles from an 8 KiB LDS table, tilt computed (log_mode 1: two fused ops per value)
                                                                                       //          (WPRE, the window prefetch ahead of the stores, measured no gain: profiles/r02)
GLV_TUNED(13,   1, 5,    2,    0,   4,     false, 2,  1,       3,      0,   0,   4)    //          split exchange (the row crosses LDS one float component at a time): two rows per 512-thread workgroup and
                                                                                       //          EVERY twiddle in LDS, no L2 gather (tied the production plan in r03 sweep_xsplit; fused bars fall back to two launches)
GLV_TUNED(14,   0, 5,    1,    1,   2,     false, 2,  1,       3,      0,   0,   1)    // N=32768  E=32: 5+5+4, one 512-lane row per CU (135 KiB exchange region), pass-1 twiddles from LDS (sweep_13)
GLV_TUNED(14,   1, 5,    1,    1,   0,     false, 2,  1,       3,      0,   0,   1)    //          every pass's per-lane twiddles gathered from the L2-resident table (no LDS copy: 8 KiB more for the exchange)
GLV_NVARIANTS(8, 2) GLV_NVARIANTS(9, 2) GLV_NVARIANTS(10, 2) GLV_NVARIANTS(11, 2) GLV_NVARIANTS(12, 2) GLV_NVARIANTS(13, 2) GLV_NVARIANTS(14, 2)
#undef GLV_TUNED
#undef GLV_NVARIANTS

#define GLV_CAT2(a, b) a##b
#define GLV_CAT(a, b) GLV_CAT2(a, b)

constexpr int kNV = NumVariants<GLV_LOG_NN>::value;

// The runner-up is built for every input and the log modes a deployment runs (0 and 1); log_mode 2 (audit) has configuration 0 only.
static bool variant_built(int in_mode, int log_mode, int variant) {
    (void) in_mode;
    if (variant == 0) return true;
    return variant > 0 && variant < kNV && (log_mode == 0 || log_mode == 1);
}

template <int IN_MODE, int LOG_MODE, int V>
static hipError_t launch_one(const FrameArgs& a, int grid, hipStream_t st) {
    using TU = Tuned<GLV_LOG_NN, V>;
    return launch_variant<GLV_LOG_NN, IN_MODE, LOG_MODE, TU::slots, TU::nbuf, TU::twreg, TU::winlds, TU::occ, TU::prefetch, TU::tiltreg, TU::log_e, TU::wpre, TU::wpre_s>(a, grid, st);
}

// the three parts' entry points (each defined by the translation unit compiled with that GLV_INST_PART)
hipError_t GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part0)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st);
hipError_t GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part1)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st);
hipError_t GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part2)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st);

template <int IN_MODE, int V>
static hipError_t launch_log(int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    switch (log_mode) {
        case 0: return launch_one<IN_MODE, 0, V>(a, grid, st);
        case 1: return launch_one<IN_MODE, 1, V>(a, grid, st);
        case 2: if constexpr (V == 0) return launch_one<IN_MODE, 2, 0>(a, grid, st); else return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}

#if GLV_INST_PART == 0
hipError_t GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part0)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    switch (in_mode) {
        case IN_S16_STEREO: return launch_log<IN_S16_STEREO, 0>(log_mode, a, grid, st);
        case IN_S16_RING:   return launch_log<IN_S16_RING, 0>(log_mode, a, grid, st);
    }
    return hipErrorInvalidValue;
}
#elif GLV_INST_PART == 1
hipError_t GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part1)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    switch (in_mode) {
        case IN_F32_PLANAR: return launch_log<IN_F32_PLANAR, 0>(log_mode, a, grid, st);
        case IN_F32_STEREO: return launch_log<IN_F32_STEREO, 0>(log_mode, a, grid, st);
        case IN_F32_RING:   return launch_log<IN_F32_RING, 0>(log_mode, a, grid, st);
    }
    return hipErrorInvalidValue;
}
#else
hipError_t GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part2)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    if constexpr (kNV > 1) {
        switch (in_mode) {
            case IN_S16_STEREO: return launch_log<IN_S16_STEREO, 1>(log_mode, a, grid, st);
            case IN_S16_RING:   return launch_log<IN_S16_RING, 1>(log_mode, a, grid, st);
            case IN_F32_PLANAR: return launch_log<IN_F32_PLANAR, 1>(log_mode, a, grid, st);
            case IN_F32_STEREO: return launch_log<IN_F32_STEREO, 1>(log_mode, a, grid, st);
            case IN_F32_RING:   return launch_log<IN_F32_RING, 1>(log_mode, a, grid, st);
        }
    }
    return hipErrorInvalidValue;
}
#endif

#if GLV_INST_PART == 0
hipError_t GLV_CAT(launch_frame_, GLV_LOG_NN)(int in_mode, int log_mode, int variant, const FrameArgs& a, int grid, hipStream_t st) {
    if (variant == 1) return GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part2)(in_mode, log_mode, a, grid, st);
    if (variant != 0) return hipErrorInvalidValue;
    if (in_mode == IN_S16_STEREO || in_mode == IN_S16_RING) return GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part0)(in_mode, log_mode, a, grid, st);
    return GLV_CAT(GLV_CAT(launch_frame_, GLV_LOG_NN), _part1)(in_mode, log_mode, a, grid, st);
}

// what the host needs to know about configuration `variant` of this size (glv_launch.h FrameGeometry)
template <int V>
static FrameGeometry geometry_of() {
    using TU = Tuned<GLV_LOG_NN, V>;
    using FR = Frame<GLV_LOG_NN, TU::log_e>;
    FrameGeometry g;
    g.lanes = FR::T;                                       // lanes that cooperate on one row (fused bars need whole waves per row)
    g.bar_batch = bar_batch_of(GLV_LOG_NN);
    // workgroups that fit one CU (LDS and the launch-bounds wave budget)
    constexpr size_t lds = frame_lds_bytes<GLV_LOG_NN, TU::log_e, TU::slots, TU::nbuf, TU::winlds, TU::twreg>();
    constexpr int by_lds = (int) (160 * 1024 / lds);
    constexpr int waves = FR::T * TU::slots / 64 > 0 ? FR::T * TU::slots / 64 : 1;
    constexpr int by_waves = TU::occ * 4 / waves;
    constexpr int r = by_lds < by_waves ? by_lds : by_waves;
    g.resident = r > 0 ? r : 1;
    // rounds of resident workgroups a large launch is cut into (glv_api.cpp frame_grid).  Two rounds even out CU-to-CU
    // differences; the sizes whose ONE 512-thread workgroup per CU pays a 64 KiB window staging prologue with nothing else on
    // the CU to overlap it run one round (N=8192, 32768 streams: 0.697 ms with 256 workgroups, 0.723 ms with 512).
    g.rounds = TU::rounds;
    // channel rows one workgroup takes per trip of its persistent loop (grid sizing): a pipelined s16 slot takes a whole
    // frame (2 rows), a single-slot workgroup both rows of its frame in sequence; for other inputs the figure only makes
    // the grid slightly smaller than strictly necessary.
    g.rows_per_trip = (TU::prefetch == 1 || TU::slots == 1) ? 2 * TU::slots : TU::slots;
    g.lds_bytes = (int) lds;
    g.log_e = TU::log_e; g.slots = TU::slots; g.twreg = TU::twreg; g.winlds = TU::winlds ? 1 : 0; g.nbuf = TU::nbuf;
    g.live_points = FR::LIVE_POINTS;
    return g;
}

int GLV_CAT(frame_variants_, GLV_LOG_NN)() { return kNV; }
int GLV_CAT(frame_variant_ok_, GLV_LOG_NN)(int in_mode, int log_mode, int variant) { return variant_built(in_mode, log_mode, variant) ? 1 : 0; }
FrameGeometry GLV_CAT(frame_geometry_, GLV_LOG_NN)(int variant) {
    if constexpr (kNV > 1) { if (variant == 1) return geometry_of<1>(); }
    return geometry_of<0>();
}

#endif   // GLV_INST_PART == 0

}  // namespace glv
