// tests/emu/glv_emu.cpp -- host "kernel emulator" (test infrastructure).
//
// Runs the very same per-thread phase functions the gfx950 kernel runs (glava_amd/csrc/
// glv_frame.h), but walks tid = 0..T-1 sequentially per phase where the kernel has T lanes
// and a barrier, with a plain array standing in for LDS.  It exists so that the index
// maps, the pass plan, the twiddle gathering and the epilogue state machine are checked
// against the oracle on the CPU-only container; it is NOT a fallback and the product
// library never links it.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC (tests/conftest.py does it).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../glava_amd/csrc/glv_frame.h"
#include "../../glava_amd/csrc/glv_tables.h"

using namespace glv;

template <int LOG_NN, int LOG_MODE, int LOG_E>
struct Emu {
    using FR = Frame<LOG_NN, LOG_E>;
    static constexpr int T = FR::T, P = FR::P, NN = FR::NN, N = FR::N;

    struct Thread { cf v[FR::E]; };

    template <int PASS>
    static void run_pass(std::vector<Thread>& th, std::vector<cf>& lds, const cf* table) {
        for (int tid = 0; tid < T; ++tid) {
            if constexpr (PASS > 0) FR::template exchange_read<PASS>(th[tid].v, lds.data(), tid);
        }
        // (barrier) -- all reads done before the next exchange overwrites the region
        for (int tid = 0; tid < T; ++tid) {
            cf tw[FR::template PassInfo<PASS>::NTW];
            FR::template gather_tw<PASS>(tw, table, tid);
            FR::template compute<PASS>(th[tid].v, tw);
            if constexpr (PASS < P - 1) FR::template exchange_write<PASS>(lds.data(), th[tid].v, tid);
        }
        // (barrier)
        if constexpr (PASS < P - 1) run_pass<PASS + 1>(th, lds, table);
    }

    // one channel row given register-resident inputs -> out_row
    static void finish(std::vector<Thread>& th, float* out_row, size_t row, const FrameArgs& a) {
        const bool st = (a.ops & (OP_GRAVITY | OP_AVERAGE)) != 0, raw = (a.ops & OP_RAW) != 0;
        for (int tid = 0; tid < T; ++tid) {
            if (raw && st)       FR::template epilogue<LOG_MODE, EPI_RAW_STATE>(th[tid].v, out_row, row, tid, a, a.logtab);
            else if (raw)        FR::template epilogue<LOG_MODE, EPI_RAW>(th[tid].v, out_row, row, tid, a, a.logtab);
            else if (st)         FR::template epilogue<LOG_MODE, EPI_MAG_STATE>(th[tid].v, out_row, row, tid, a, a.logtab);
            else                 FR::template epilogue<LOG_MODE, EPI_MAG>(th[tid].v, out_row, row, tid, a, a.logtab);
        }
    }

    // one channel row of an s16 frame (row = 2*frame + channel), exactly what one kernel slot does
    static void row_s16(const int16_t* frame, size_t row, const FrameArgs& a) {
        std::vector<Thread> th(T);
        std::vector<cf> lds(FR::XREGION);
        for (int tid = 0; tid < T; ++tid) {
            typename FR::Raw raw;
            if (a.rot) FR::template load_pcm<true>(raw, frame, tid, a.rot);
            else       FR::template load_pcm<false>(raw, frame, tid, 0);
            FR::unpack_window(th[tid].v, raw, a.win, tid, (uint32_t) (row & 1), a.mono != 0);
        }
        run_pass<0>(th, lds, a.tw);
        finish(th, a.out + row * N, row, a);
    }
    static void row_f32_stereo(const float* frame, size_t row, const FrameArgs& a) {
        std::vector<Thread> th(T);
        std::vector<cf> lds(FR::XREGION);
        for (int tid = 0; tid < T; ++tid)
            FR::load_f32_stereo_window(th[tid].v, frame, a.win, tid, (uint32_t) (row & 1), a.mono != 0);
        run_pass<0>(th, lds, a.tw);
        finish(th, a.out + row * N, row, a);
    }
    static void row_f32(const float* in_row, size_t row, const FrameArgs& a) {
        std::vector<Thread> th(T);
        std::vector<cf> lds(FR::XREGION);
        for (int tid = 0; tid < T; ++tid) FR::load_f32_window(th[tid].v, in_row, a.win, tid);
        run_pass<0>(th, lds, a.tw);
        finish(th, a.out + row * N, row, a);
    }
};

template <int LOG_NN, int LOG_MODE, int LOG_E>
static void run_units(int in_mode, const FrameArgs& a) {
    using EM = Emu<LOG_NN, LOG_MODE, LOG_E>;
    for (uint32_t u = 0; u < a.units; ++u) {
        if (in_mode == IN_S16_STEREO) EM::row_s16((const int16_t*) a.in + (size_t) (u >> 1) * 2 * EM::N, u, a);
        else if (in_mode == IN_F32_STEREO) EM::row_f32_stereo((const float*) a.in + (size_t) (u >> 1) * 2 * EM::N, u, a);
        else EM::row_f32((const float*) a.in + (size_t) u * EM::N, u, a);
    }
}

template <int LOG_MODE, int LOG_E>
static int dispatch(int log_nn, int in_mode, const FrameArgs& a) {
    switch (log_nn) {
        case 8:  run_units<8, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 9:  run_units<9, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 10: run_units<10, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 11: run_units<11, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 12: run_units<12, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 13: run_units<13, LOG_MODE, LOG_E>(in_mode, a); return 0;
    }
    return 1;
}

extern "C" {

// n: real samples per channel.  in: s16 [units/2][n][2] (in_mode 0) or f32 [units][n] (in_mode 1);
// units = channel rows.
// grav / hist may be NULL when the op is not requested.  Returns 0 on success.
int glvemu_process(int n, int in_mode, const void* in, float* out, float* grav, float* hist,
                   unsigned units, unsigned ops, unsigned F, unsigned head, int mono, int avg_window,
                   int avg_kind, int log_mode, float fft_scale, float fft_cutoff, float gravity_step, float ur,
                   unsigned rot, int log_e) {
    int log_nn = 0;
    while ((2 << log_nn) < n) ++log_nn;
    if ((2 << log_nn) != n) return 2;
    const int nn = n / 2;
    std::vector<cf> tw(nn);
    std::vector<double> win(n);
    make_twiddles(tw.data(), nn);
    make_window(win.data(), n);
    LogEntry lt[kLogTabSize];
    make_log_table(lt);
    std::vector<float> tl(n);
    make_tilt(tl.data(), n, fft_scale, fft_cutoff, log_mode == 1);
    FrameArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.grav = grav; a.hist = hist; a.tw = tw.data(); a.win = win.data(); a.logtab = lt; a.tilt = tl.data();
    a.units = units; a.ops = ops; a.F = F; a.head = head; a.mono = mono; a.avg_window = avg_window; a.rot = rot;
    a.inv_n = 1.0f / (float) n; a.fft_scale = fft_scale; a.one_minus_cutoff = 1.0f - fft_cutoff;
    a.g = gravity_step * (1.0f / ur); a.F_as_float = (float) F;
    if (F > 16) return 3;
    make_frame_weights(a.wts, F, avg_window != 0, avg_kind);
    if (log_e == 5) return log_mode == 0 ? dispatch<0, 5>(log_nn, in_mode, a) : log_mode == 1 ? dispatch<1, 5>(log_nn, in_mode, a) : dispatch<2, 5>(log_nn, in_mode, a);
    if (log_e == 3) return log_mode == 0 ? dispatch<0, 3>(log_nn, in_mode, a) : log_mode == 1 ? dispatch<1, 3>(log_nn, in_mode, a) : dispatch<2, 3>(log_nn, in_mode, a);
    return log_mode == 0 ? dispatch<0, 4>(log_nn, in_mode, a) : log_mode == 1 ? dispatch<1, 4>(log_nn, in_mode, a) : dispatch<2, 4>(log_nn, in_mode, a);
}

}  // extern "C"
