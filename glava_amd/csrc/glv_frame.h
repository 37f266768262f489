// glv_frame.h -- the phases one FFT "slot" (T = nn/16 cooperating threads) runs per frame.
//
// Shared between the gfx950 kernels (glv_kernels.hip) and the host emulator (tests/emu):
// every function takes the thread id explicitly and touches memory only through the
// pointers it is given, so the emulator can call the same phase for tid = 0..T-1 in turn
// where the kernel has T lanes and an s_barrier.
//
// Data flow for one channel of one frame (nn = N/2 complex points, P = ceil(log2(nn)/4) passes):
//   load_inputs      HBM -> registers, fused unpack (fifo.c:94-110) + window (render.c:792-795)
//   pass<0>          four radix-2 stages in registers (twiddles uniform: scalar loads)
//   exchange_write / barrier / exchange_read        through LDS (XOR-swizzled after pass 0)
//   pass<1> ... pass<P-1>
//   epilogue         abs/log/tilt (render.c:842-846) + gravity (720-736) + average (738-771),
//                    registers -> HBM
#pragma once

#include "glv_core.h"

namespace glv {

enum InMode { IN_S16_STEREO = 0, IN_F32_PLANAR = 1 };
enum Epi { EPI_RAW = 0, EPI_MAG = 1, EPI_MAG_STATE = 2, EPI_RAW_STATE = 3 };

// ops bits as in include/glv_spectrum.h
enum : uint32_t { OP_FFT = 1u, OP_GRAVITY = 2u, OP_AVERAGE = 4u, OP_RAW = 8u, OP_WRANGE = 16u, OP_BARS = 32u };

struct FrameArgs {
    const void* in;        // IN_S16_STEREO: int16 [units][n][2];  IN_F32_PLANAR: float [units][n]
    float* out;            // [rows][n], rows = units*2 (s16) or units (f32)
    float* grav;           // [rows][n] gravity state (only when gravity without average)
    float* hist;           // [rows][F][n] history ring (average); doubles as gravity state
    const cf* tw;          // nn-1 twiddles, layout glv::tw_offset
    const double* win;     // n window values (render.c:660 as expanded at :794)
    uint32_t units;        // stereo frames (s16) or channel rows (f32) to process
    uint32_t ops;
    uint32_t F, head;      // ring: `head` receives the current frame; ages oldest..newest are
                           // head+1, ..., head+F-1, head  (mod F)
    uint32_t mono;         // fifo.c:98-102
    uint32_t avg_window;
    uint32_t rot;          // s16 ring mode: rotation of the window start, in complex points (pairs of frames)
    float inv_n, fft_scale, one_minus_cutoff, g, F_as_float;
    double wts[16];        // window_frame weights, oldest first (render.c:661 as expanded at :766)
};

// ring slot of age f (0 = oldest .. F-1 = newest = head)
GLV_HD uint32_t ring_slot(uint32_t head, uint32_t f, uint32_t F) {
    uint32_t s = head + 1 + f;
    return s >= F ? s - F : s;   // head < F, f < F  =>  s < 2F
}

// gravity (render.c:720-736) and average (render.c:738-771) for the float pair (n0, n0+1) of
// channel row `row`; n = floats per row.  State traffic is 8 bytes per lane, lanes contiguous.
// With both operators the newest ring slot doubles as the gravity state ("applied" of
// render.c:724 is by construction the previous gravity output, i.e. the previous newest slot).
GLV_HD cf apply_state(cf val, int n0, size_t row, uint32_t n, const FrameArgs& a) {
    if (a.ops & OP_AVERAGE) {
        float* h = a.hist + row * (size_t) a.F * n + n0;
        const uint32_t F = a.F;
        cf acc = { 0.0f, 0.0f }, prev = { 0.0f, 0.0f };
        if (F == 1) prev = *reinterpret_cast<const cf*>(h + (size_t) a.head * n);
        for (uint32_t f = 0; f + 1 < F; ++f) {                               // oldest .. second newest
            prev = *reinterpret_cast<const cf*>(h + (size_t) ring_slot(a.head, f, F) * n);
            if (a.avg_window) {                                              // render.c:759, double product
                acc.x = (float) ((double) acc.x + a.wts[f] * (double) prev.x);
                acc.y = (float) ((double) acc.y + a.wts[f] * (double) prev.y);
            } else { acc.x = acc.x + prev.x; acc.y = acc.y + prev.y; }
        }
        if (a.ops & OP_GRAVITY) {
            val.x = gravity(val.x, prev.x, a.g); val.y = gravity(val.y, prev.y, a.g);
        }
        *reinterpret_cast<cf*>(h + (size_t) a.head * n) = val;
        if (a.avg_window) {
            acc.x = (float) ((double) acc.x + a.wts[F - 1] * (double) val.x);
            acc.y = (float) ((double) acc.y + a.wts[F - 1] * (double) val.y);
        } else { acc.x = acc.x + val.x; acc.y = acc.y + val.y; }
        val.x = acc.x / a.F_as_float;                                        // render.c:761
        val.y = acc.y / a.F_as_float;
    } else if (a.ops & OP_GRAVITY) {
        cf* gs = reinterpret_cast<cf*>(a.grav + row * (size_t) n + n0);
        const cf st = *gs;
        val.x = gravity(val.x, st.x, a.g); val.y = gravity(val.y, st.y, a.g);
        *gs = val;
    }
    return val;
}

template <int LOG_NN>
struct Frame {
    using PL = Plan<LOG_NN>;
    static constexpr int NN = PL::NN, N = 2 * NN, T = PL::T, P = PL::P, E = 16;

    // ---- input: v[i] <- windowed sample pair (complex point) c = i*T + tid ---------------------
    // s16: one 8-byte load holds complex point c of BOTH channels: (L[2c], R[2c], L[2c+1], R[2c+1]).
    struct Pcm { uint32_t lo[E], hi[E]; };   // lo = L[2c] | R[2c] << 16, hi = L[2c+1] | R[2c+1] << 16

    // `rot` (complex points) rotates the read position for the FIFO ring mode (fifo.c:91-92 keeps
    // the newest samples at the end of the buffer; the device ring is circular instead).
    GLV_HD static void load_pcm(Pcm& p, const int16_t* frame, int tid, uint32_t rot) {
        const u32x2* src = reinterpret_cast<const u32x2*>(frame);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const u32x2 u = src[(uint32_t) (i * T + tid + rot) & (uint32_t) (NN - 1)];
            p.lo[i] = u.x; p.hi[i] = u.y;
        }
    }
    // fifo.c:98-102 mono mix, applied once to the packed samples so that the per-channel unpack
    // below stays branch free: both channels become ((L + R) / 2) (C int division).
    GLV_HD static void mono_mix(Pcm& p) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int m0 = ((int) (int16_t) (p.lo[i] & 0xffffu) + (int) (int16_t) (p.lo[i] >> 16)) / 2;
            const int m1 = ((int) (int16_t) (p.hi[i] & 0xffffu) + (int) (int16_t) (p.hi[i] >> 16)) / 2;
            p.lo[i] = ((uint32_t) m0 & 0xffffu) | ((uint32_t) m0 << 16);
            p.hi[i] = ((uint32_t) m1 & 0xffffu) | ((uint32_t) m1 << 16);
        }
    }
    template <int CH>
    GLV_HD static void unpack_window(cf (&v)[E], const Pcm& p, const double* win, int tid) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int c = i * T + tid;
            const int s0 = CH == 0 ? (int) (int16_t) (p.lo[i] & 0xffffu) : (int) (int16_t) (p.lo[i] >> 16);
            const int s1 = CH == 0 ? (int) (int16_t) (p.hi[i] & 0xffffu) : (int) (int16_t) (p.hi[i] >> 16);
            const d2 w = reinterpret_cast<const d2*>(win)[c];
            v[i].x = apply_window(unpack_s16(s0), w.x);
            v[i].y = apply_window(unpack_s16(s1), w.y);
        }
    }
    GLV_HD static void load_f32_window(cf (&v)[E], const float* row, const double* win, int tid) {
        const cf* src = reinterpret_cast<const cf*>(row);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int c = i * T + tid;
            const cf u = src[c];
            const d2 w = reinterpret_cast<const d2*>(win)[c];
            v[i].x = apply_window(u.x, w.x);
            v[i].y = apply_window(u.y, w.y);
        }
    }

    // ---- twiddle gather for pass PASS (per thread; invariant across frames) ---------------------
    template <int PASS>
    struct PassInfo {
        static constexpr int RB = PL::rb(PASS), R = 1 << RB, NG = E >> RB;
        static constexpr int LOG_L0 = PL::log_l0(PASS), L0 = 1 << LOG_L0;
        static constexpr int NTW = NG * (R - 1);
    };

    template <int PASS>
    GLV_HD static void gather_tw(cf (&tw)[PassInfo<PASS>::NTW], const cf* table, int tid) {
        using PI = PassInfo<PASS>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi) {
            const int G = gi * T + tid;
            const int k0 = G & (PI::L0 - 1);
#pragma unroll
            for (int s = 0; s < PI::RB; ++s)
#pragma unroll
                for (int ks = 0; ks < (1 << s); ++ks)
                    tw[gi * (PI::R - 1) + (1 << s) - 1 + ks] = table[SubPass<PI::RB>::tw_index(PI::L0, k0, s, ks)];
        }
    }

    // ---- compute pass PASS in registers -----------------------------------------------------------
    template <int PASS>
    GLV_HD static void compute(cf (&v)[E], const cf (&tw)[PassInfo<PASS>::NTW]) {
        using PI = PassInfo<PASS>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi) {
            cf(&vg)[PI::R] = *reinterpret_cast<cf(*)[PI::R]>(&v[gi * PI::R]);
            const cf(&tg)[PI::R > 1 ? PI::R - 1 : 1] =
                *reinterpret_cast<const cf(*)[PI::R > 1 ? PI::R - 1 : 1]>(&tw[gi * (PI::R - 1)]);
            SubPass<PI::RB>::run(vg, tg);
        }
    }

    // element index (within the nn-point array after pass PASS) held in register slot (gi, r)
    template <int PASS>
    GLV_HD static constexpr int out_index(int tid, int gi, int r) {
        using PI = PassInfo<PASS>;
        const int G = gi * T + tid;
        const int k0 = G & (PI::L0 - 1);
        const int jt = G >> PI::LOG_L0;
        return jt * (PI::R * PI::L0) + bitrev(r, PI::RB) * PI::L0 + k0;
    }
    // element index that register slot (gi, i) of pass PASS must be fed with
    template <int PASS>
    GLV_HD static constexpr int in_index(int tid, int gi, int i) {
        using PI = PassInfo<PASS>;
        return i * (NN / PI::R) + gi * T + tid;
    }

    template <int PASS>
    GLV_HD static void exchange_write(cf* xbuf, const cf (&v)[E], int tid) {
        using PI = PassInfo<PASS>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi)
#pragma unroll
            for (int r = 0; r < PI::R; ++r)
                xbuf[lds_index(PASS, out_index<PASS>(tid, gi, r))] = v[gi * PI::R + r];
    }
    // read the inputs of pass PASS from the exchange written after pass PASS-1
    template <int PASS>
    GLV_HD static void exchange_read(cf (&v)[E], const cf* xbuf, int tid) {
        using PI = PassInfo<PASS>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi)
#pragma unroll
            for (int i = 0; i < PI::R; ++i)
                v[gi * PI::R + i] = xbuf[lds_index(PASS - 1, in_index<PASS>(tid, gi, i))];
    }

    // ---- epilogue: registers of the last pass -> HBM ------------------------------------------------
    // One complex point = floats n0 (even) and n0+1 of channel row `row`; 8-byte stores, lanes
    // contiguous.  EPI selects the operator chain at compile time (the kernel switches on a.ops
    // once per frame, outside the unrolled element loop):
    //   EPI_RAW  raw FFT output              EPI_MAG   abs/log/tilt
    //   EPI_MAG_STATE  abs/log/tilt followed by gravity and/or average (apply_state)
    template <int LOG_MODE, int EPI>
    GLV_HD static void epilogue(const cf (&v)[E], float* out_row, size_t row, int tid, const FrameArgs& a) {
        using PI = PassInfo<P - 1>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi)
#pragma unroll
            for (int r = 0; r < PI::R; ++r) {
                const int q = out_index<P - 1>(tid, gi, r);
                const int n0 = 2 * q;
                cf val = v[gi * PI::R + r];
                if constexpr (EPI == EPI_MAG || EPI == EPI_MAG_STATE) {
                    const float y0 = __builtin_fabsf(val.x) + 1.0f, y1 = __builtin_fabsf(val.y) + 1.0f;   // render.c:843-844
                    val.x = log_third<LOG_MODE>(y0) * tilt(n0, a.inv_n, a.fft_scale, a.one_minus_cutoff);  // :845
                    val.y = log_third<LOG_MODE>(y1) * tilt(n0 + 1, a.inv_n, a.fft_scale, a.one_minus_cutoff);
                }
                if constexpr (EPI == EPI_MAG_STATE || EPI == EPI_RAW_STATE) val = apply_state(val, n0, row, (uint32_t) N, a);
                *reinterpret_cast<cf*>(out_row + n0) = val;
            }
    }
};

}  // namespace glv
