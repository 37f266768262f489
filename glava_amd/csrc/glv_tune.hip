// glv_tune.hip -- knob sweep of glv_frame_kernel for ONE size (default N = 4096, -DGLV_TUNE_LOG_NN=k).
// Built into its own shared object (libglvtune.so) that only tools/tune.py loads; the
// product library does not contain these instantiations.
#include <cstring>
#include <vector>

#include "glv_kernel_tmpl.h"
#include "glv_winsplit.h"
#include "glv_tables.h"

#ifndef GLV_TUNE_LOG_NN
#define GLV_TUNE_LOG_NN 11
#endif

namespace glv {
namespace {

struct Variant {
    const char* desc;
    hipError_t (*launch)(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st);
    int slots;
};

template <int SLOTS, int NBUF, int TWREG, bool WINLDS, int OCC, int PF = 0, int TL = 0, int LE = 4, int WP = 0, int WPS = 0>
hipError_t launch_v(int in_mode, int log_mode, const FrameArgs& a, int grid, hipStream_t st) {
    constexpr int K = GLV_TUNE_LOG_NN;
    if (in_mode != IN_S16_STEREO) return hipErrorInvalidValue;
#if !defined(GLV_TUNE_NO_LOG0)
    if (log_mode == 0) return launch_variant<K, IN_S16_STEREO, 0, SLOTS, NBUF, TWREG, WINLDS, OCC, PF, TL, LE, WP, WPS>(a, grid, st);
#endif
    return launch_variant<K, IN_S16_STEREO, 1, SLOTS, NBUF, TWREG, WINLDS, OCC, PF, TL, LE, WP, WPS>(a, grid, st);
}

#define V(S, NB, TR, WL, OC) { "slots=" #S " nbuf=" #NB " twreg=" #TR " winlds=" #WL " occ=" #OC, launch_v<S, NB, TR, WL, OC>, S }
#define VX(S, NB, TR, WL, OC, PF, TL) { "slots=" #S " nbuf=" #NB " twreg=" #TR " winlds=" #WL " occ=" #OC " pf=" #PF " tiltreg=" #TL, launch_v<S, NB, TR, WL, OC, PF, TL>, S }
#define VE(S, NB, TR, WL, OC, PF, TL, LE) { "slots=" #S " nbuf=" #NB " twreg=" #TR " winlds=" #WL " occ=" #OC " pf=" #PF " tiltreg=" #TL " log_e=" #LE, launch_v<S, NB, TR, WL, OC, PF, TL, LE>, S }
#define VW(S, NB, TR, WL, OC, PF, TL, LE, WP, WPS) { "slots=" #S " nbuf=" #NB " twreg=" #TR " winlds=" #WL " occ=" #OC " pf=" #PF " tiltreg=" #TL " log_e=" #LE " wpre=" #WP " wpre_s=" #WPS, launch_v<S, NB, TR, WL, OC, PF, TL, LE, WP, WPS>, S }
#define VP(S, NB, TR, WL, OC) { "slots=" #S " nbuf=" #NB " twreg=" #TR " winlds=" #WL " occ=" #OC " prefetch", launch_v<S, NB, TR, WL, OC, true>, S }
const Variant kVariants[] = {
#ifdef GLV_TUNE_VARIANTS
    GLV_TUNE_VARIANTS
#else
    V(2, 1, false, false, 2), V(2, 1, false, false, 3), V(2, 1, false, false, 4),
    V(2, 1, false, true, 2),  V(2, 1, false, true, 4),
    V(2, 1, true, false, 2),  V(2, 1, true, false, 3),
    V(4, 1, false, false, 3), V(4, 1, false, false, 4), V(4, 1, false, true, 4),
    V(2, 2, false, false, 4), V(4, 2, false, false, 4),
#endif
};
#undef V
#undef VP
#undef VX
#undef VE
#undef VW

}  // namespace
}  // namespace glv

extern "C" {
int glv_tune_count(void) { return (int) (sizeof(glv::kVariants) / sizeof(glv::kVariants[0])); }
const char* glv_tune_describe(int i) { return glv::kVariants[i].desc; }
int glv_tune_slots(int i) { return glv::kVariants[i].slots; }
int glv_tune_log_nn(void) { return GLV_TUNE_LOG_NN; }
int glv_tune_launch(int i, int in_mode, int log_mode, const glv::FrameArgs* a, int grid, void* stream) {
    return (int) glv::kVariants[i].launch(in_mode, log_mode, *a, grid, (hipStream_t) stream);
}

// Self-contained timing of one variant: `iters` launches over `units` stereo frames of s16 PCM
// already in HBM (FFT + magnitude only), HIP events around the whole run on `stream`.
// Returns average milliseconds per launch in *ms.  Tables are created once per process.
// extra_ops: OR'ed into OP_FFT (OP_R16: d_out receives uint16 texels; OP_GRAVITY: d_grav = float [units*2][n] state)
int glv_tune_run2(int i, const void* d_pcm, float* d_out, unsigned units, int log_mode, int grid, int iters,
                  void* stream, float* ms, unsigned extra_ops, float* d_grav);
// the same with a history ring (OP_AVERAGE): d_hist = float [units*2][F][n], F = 5 windowed
int glv_tune_run3(int i, const void* d_pcm, float* d_out, unsigned units, int log_mode, int grid, int iters,
                  void* stream, float* ms, unsigned extra_ops, float* d_grav, float* d_hist);
int glv_tune_run(int i, const void* d_pcm, float* d_out, unsigned units, int log_mode, int grid, int iters,
                 void* stream, float* ms) {
    return glv_tune_run2(i, d_pcm, d_out, units, log_mode, grid, iters, stream, ms, 0u, nullptr);
}
int glv_tune_run2(int i, const void* d_pcm, float* d_out, unsigned units, int log_mode, int grid, int iters,
                  void* stream, float* ms, unsigned extra_ops, float* d_grav) {
    return glv_tune_run3(i, d_pcm, d_out, units, log_mode, grid, iters, stream, ms, extra_ops, d_grav, nullptr);
}
int glv_tune_run3(int i, const void* d_pcm, float* d_out, unsigned units, int log_mode, int grid, int iters,
                  void* stream, float* ms, unsigned extra_ops, float* d_grav, float* d_hist) {
    using namespace glv;
    static cf* d_tw = nullptr;
    static double* d_win = nullptr;
    static float* d_win_split = nullptr;
    static LogEntry* d_log = nullptr;
    static float* d_tilt = nullptr;
    static float* d_tilt_fast = nullptr;
    constexpr int NN = 1 << GLV_TUNE_LOG_NN, N = 2 * NN;
    if (!d_tw) {
        std::vector<cf> tw(NN, cf{0.0f, 0.0f});
        std::vector<double> win(N);
        make_twiddles(tw.data(), NN);
        make_window(win.data(), N);
        if (hipMalloc(&d_tw, sizeof(cf) * NN) != hipSuccess) return -1;
        if (hipMalloc(&d_win, sizeof(double) * N) != hipSuccess) return -1;
        (void) hipMemcpy(d_tw, tw.data(), sizeof(cf) * NN, hipMemcpyHostToDevice);
        (void) hipMemcpy(d_win, win.data(), sizeof(double) * N, hipMemcpyHostToDevice);
        // the s16 kernels read the window as float pairs (glv_core.h WinSplit): the same device-side search as the product's
        if (hipMalloc(&d_win_split, sizeof(float) * 2 * N) != hipSuccess) return -1;
        int* d_fs = nullptr;
        if (hipMalloc(&d_fs, 2 * sizeof(int)) != hipSuccess) return -1;
        (void) hipMemset(d_fs, 0, 2 * sizeof(int));
        (void) launch_window_split_impl(d_win, d_win_split, N, d_fs, nullptr);
        (void) hipDeviceSynchronize();
        (void) hipFree(d_fs);
        LogEntry lt[kLogTabMaxSize];
        make_log_table(lt);
        if (hipMalloc(&d_log, sizeof(lt)) != hipSuccess) return -1;
        (void) hipMemcpy(d_log, lt, sizeof(lt), hipMemcpyHostToDevice);
        std::vector<float> tl(N);
        make_tilt(tl.data(), N, 10.2f, 0.3f);
        if (hipMalloc(&d_tilt, sizeof(float) * N) != hipSuccess) return -1;
        (void) hipMemcpy(d_tilt, tl.data(), sizeof(float) * N, hipMemcpyHostToDevice);
        make_tilt(tl.data(), N, 10.2f, 0.3f, true);
        if (hipMalloc(&d_tilt_fast, sizeof(float) * N) != hipSuccess) return -1;
        (void) hipMemcpy(d_tilt_fast, tl.data(), sizeof(float) * N, hipMemcpyHostToDevice);
    }
    FrameArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = d_pcm; a.out = d_out; a.tw = d_tw; a.win = d_win; a.win_split = d_win_split; a.logtab = d_log; a.tilt = log_mode == 1 ? d_tilt_fast : d_tilt; a.units = units * 2; a.ops = OP_FFT | extra_ops; a.grav = d_grav; a.grav_w = d_grav;
    a.F = d_hist ? 5 : 1; a.hist = d_hist; a.avg_window = 1;
    make_frame_weights(a.wts, a.F, true, 0);
    a.inv_n = 1.0f / (float) N; a.fft_scale = 10.2f; a.one_minus_cutoff = 1.0f - 0.3f;
    a.g = 4.2f * (1.0f / 86.1328125f); a.F_as_float = (float) a.F;
    hipStream_t st = (hipStream_t) stream;
    if (grid <= 0) {
        const unsigned slots = (unsigned) kVariants[i].slots;          // at most one frame per slot per trip
        const unsigned wgs = (units * 2 + slots - 1) / slots;
        grid = (int) (wgs < 2048u ? wgs : 2048u);
    }
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    hipError_t e = kVariants[i].launch(IN_S16_STEREO, log_mode, a, grid, st);   // warm-up + attribute setup
    if (e != hipSuccess) return (int) e;
    (void) hipEventRecord(e0, st);
    for (int it = 0; it < iters; ++it) {
        e = kVariants[i].launch(IN_S16_STEREO, log_mode, a, grid, st);
        if (e != hipSuccess) return (int) e;
    }
    (void) hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) return -2;
    float t = 0;
    (void) hipEventElapsedTime(&t, e0, e1);
    *ms = t / (float) iters;
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    return 0;
}
}
