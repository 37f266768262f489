// glv_inst.hip -- production instantiations of glv_frame_kernel for ONE transform size.
// Compiled once per size with -DGLV_LOG_NN=k (k = log2(nn) = log2(N) - 1, 8..13) so the six
// sizes build in parallel.  The knob set per size is the measured best of tools/tune.py
// (profiles/tune_r01.txt); see DESIGN.md "Kernel configuration".
#include "glv_kernel_tmpl.h"
#include "glv_launch.h"

#ifndef GLV_LOG_NN
#error "compile with -DGLV_LOG_NN=<8..13>"
#endif

namespace glv {

template <int LOG_NN> struct Tuned;
//                                      SLOTS NBUF TWREG WINLDS OCC
template <> struct Tuned<8>  { static constexpr int slots = 16, nbuf = 2; static constexpr bool twreg = true, winlds = true; static constexpr int occ = 2; };
template <> struct Tuned<9>  { static constexpr int slots = 8,  nbuf = 2; static constexpr bool twreg = true, winlds = true; static constexpr int occ = 2; };
template <> struct Tuned<10> { static constexpr int slots = 4,  nbuf = 2; static constexpr bool twreg = true, winlds = true; static constexpr int occ = 2; };
template <> struct Tuned<11> { static constexpr int slots = 2,  nbuf = 1; static constexpr bool twreg = true, winlds = true; static constexpr int occ = 2; };
template <> struct Tuned<12> { static constexpr int slots = 1,  nbuf = 1; static constexpr bool twreg = true, winlds = false; static constexpr int occ = 2; };
template <> struct Tuned<13> { static constexpr int slots = 1,  nbuf = 1; static constexpr bool twreg = true, winlds = false; static constexpr int occ = 2; };

#define GLV_CAT2(a, b) a##b
#define GLV_CAT(a, b) GLV_CAT2(a, b)

hipError_t GLV_CAT(launch_frame_, GLV_LOG_NN)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    using TU = Tuned<GLV_LOG_NN>;
    constexpr int K = GLV_LOG_NN;
    if (in_mode == IN_S16_STEREO) {
        if (log_mode == 0) return launch_variant<K, IN_S16_STEREO, 0, TU::slots, TU::nbuf, TU::twreg, TU::winlds, TU::occ>(a, grid, st);
        return launch_variant<K, IN_S16_STEREO, 1, TU::slots, TU::nbuf, TU::twreg, TU::winlds, TU::occ>(a, grid, st);
    }
    if (log_mode == 0) return launch_variant<K, IN_F32_PLANAR, 0, TU::slots, TU::nbuf, TU::twreg, TU::winlds, TU::occ>(a, grid, st);
    return launch_variant<K, IN_F32_PLANAR, 1, TU::slots, TU::nbuf, TU::twreg, TU::winlds, TU::occ>(a, grid, st);
}

int GLV_CAT(frame_slots_, GLV_LOG_NN)() { return Tuned<GLV_LOG_NN>::slots; }

}  // namespace glv
