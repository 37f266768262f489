// glv_core.h -- per-thread arithmetic and index maps of the spectrum kernels.
//
// Everything here is `GLV_HD` so the *same* code is compiled (a) by hipcc into the gfx950
// kernels in glv_kernels.hip and (b) by g++ into tests/emu (a host "kernel emulator" that
// walks the phases thread by thread) -- index maps and butterflies get exercised against
// the oracle on the CPU before a GPU is ever touched.
//
// What is restated from the reference (paths relative to jarcode-foss/glava):
//   glava/fifo.c:94-110      s16 -> f32 unpack
//   glava/render.c:660,794   the (phase-shifted Hamming) window, double product
//   glava/render.c:797-840   radix-2 DIT FFT, float twiddle recurrence
//   glava/render.c:842-846   abs / log / tilt
//   glava/render.c:720-771   gravity, average
// The *schedule* is new: a Stockham autosort decomposition into in-register radix-2^RB
// sub-passes (RB <= 4) whose every radix-2 butterfly performs exactly the reference's six
// individually rounded float operations with the reference's recurrence-generated twiddle,
// so results are bit-identical while the data moves through LDS only ceil(log2(nn)/4)-1
// times.  Compile with -ffp-contract=off (no FMA contraction anywhere in this file).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GLV_HD __host__ __device__ __forceinline__
#else
#define GLV_HD inline
#endif

namespace glv {

struct alignas(8) cf { float x, y; };        // one complex point == two consecutive floats of the reference's data[]
struct alignas(8) u32x2 { uint32_t x, y; };  // four interleaved s16 samples (L,R,L,R)
struct alignas(16) d2 { double x, y; };      // two consecutive window values

// ---- twiddle table layout ------------------------------------------------------------------
// One table per FFT size: stage with complex half-size L (L = 1,2,4,...,nn/2; reference
// mmax = 2L, render.c:814-839) occupies entries [L-1, 2L-1) -- nn-1 entries in total.  The
// stages' tables are NOT subsets of each other (each runs its own float recurrence).
GLV_HD constexpr int tw_offset(int L) { return L - 1; }

// The reference's six-rounding butterfly (render.c:826-832):  t = w*b;  b = a - t;  a = a + t.
GLV_HD void butterfly(cf& a, cf& b, const cf w) {
    const float p0 = w.x * b.x, p1 = w.y * b.y;
    const float p2 = w.x * b.y, p3 = w.y * b.x;
    const float tr = p0 - p1;
    const float ti = p2 + p3;
    const cf hi = { a.x - tr, a.y - ti };
    const cf lo = { a.x + tr, a.y + ti };
    a = lo; b = hi;
}

GLV_HD constexpr int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

// ---- one in-register Stockham sub-pass of radix R = 2^RB --------------------------------------
// Input:  v[i] = x[i * (nn/R) + G]   (i = top RB bits of the element index, G = the rest)
// The sub-pass covers the RB radix-2 stages with half sizes L0, 2*L0, ..., L0 << (RB-1).
// With G = jt * L0 + k0 (k0 = G mod L0), slot r of the result is the element with index
//     jt * (R*L0) + bitrev(r, RB) * L0 + k0
// of the array after those stages (Stockham: natural order in, natural order out).
// Stage s pairs slots that differ in bit (RB-1-s); its twiddle is
//     W[L0 << s][ k0 + L0 * ksub ],  ksub = the already-processed slot bits, first-processed
//     bit least significant,
// which the caller gathers into tw[(1 << s) - 1 + ksub]  (R - 1 twiddles per sub-pass).
template <int RB>
struct SubPass {
    static constexpr int R = 1 << RB;

    GLV_HD static constexpr int ksub_of(int r0, int s) {
        int k = 0;
        for (int t = 0; t < s; ++t) k |= ((r0 >> (RB - 1 - t)) & 1) << t;
        return k;
    }

    GLV_HD static void run(cf (&v)[R], const cf (&tw)[R > 1 ? R - 1 : 1]) {
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int bit = 1 << (RB - 1 - s);
#pragma unroll
            for (int r0 = 0; r0 < R; ++r0) {
                if (r0 & bit) continue;
                butterfly(v[r0], v[r0 | bit], tw[(1 << s) - 1 + ksub_of(r0, s)]);
            }
        }
    }

    // index into the size-nn twiddle table of the (s, ksub) twiddle for group constant k0
    GLV_HD static constexpr int tw_index(int L0, int k0, int s, int ksub) {
        return tw_offset(L0 << s) + k0 + L0 * ksub;
    }
};

// ---- pass plan: log2(nn) bits split into sub-passes of at most 4 bits --------------------------
// E = 16 elements per thread, T = nn/16 threads cooperate on one FFT.  Passes 0..P-2 are
// radix 16; the last takes the remaining bits (1..4) and handles 16 >> RB groups per thread.
template <int LOG_NN>
struct Plan {
    static constexpr int NN = 1 << LOG_NN;
    static constexpr int E = 16;
    static constexpr int T = NN / E;
    static constexpr int P = (LOG_NN + 3) / 4;
    GLV_HD static constexpr int rb(int pass) { return pass < P - 1 ? 4 : LOG_NN - 4 * (P - 1); }
    GLV_HD static constexpr int log_l0(int pass) { return 4 * pass; }
};

// LDS address (in complex units, within one FFT's exchange region) of element index q for the
// exchange that follows pass `pass`.  Only the first pass writes with a lane stride of 16
// elements (q = 16*tid + e); XOR-ing the low four bits with the next four makes both that
// write (16-lane ds_write_b64 groups) and the following contiguous read (32-lane ds_read_b64
// groups) bank-conflict free without padding.  Later exchanges are conflict free as is.
GLV_HD constexpr int lds_index(int pass, int q) { return pass == 0 ? (q ^ ((q >> 4) & 15)) : q; }

// ---- scalar pieces ----------------------------------------------------------------------------
// fifo.c:105-106: (float) s16 / (float) 65535, IEEE single division.
// Evaluated as one correctly-rounded-by-construction sequence: q0 = v*rcp; r = fma(-q0, 65535, v);
// q = fma(r, rcp, q0).  tests/test_host_logic.py checks all 65536 inputs against the division on
// the CPU, tests/test_gpu_parity.py does the same on the device.
GLV_HD float unpack_s16(int v) {
    const float fv = (float) v;
    const float rcp = 1.0f / 65535.0f;            // constant-folded, correctly rounded
    const float q0 = fv * rcp;
    const float r = __builtin_fmaf(-q0, 65535.0f, fv);
    return __builtin_fmaf(r, rcp, q0);
}
// fifo.c:98-102 mono mix: C int arithmetic, truncation toward zero.
GLV_HD float unpack_s16_mono(int l, int r) { return unpack_s16((l + r) / 2); }

// render.c:794: data[i] *= window(i, N)  -- float * double -> double -> float.
GLV_HD float apply_window(float x, double w) { return (float) ((double) x * w); }

// render.c:845 tilt factor; inv_n = 1/N is a power of two so n*inv_n == (float)n/(float)N exactly.
GLV_HD float tilt(int n, float inv_n, float fft_scale, float one_minus_cutoff) {
    const float a = (float) n * inv_n;
    const float b = a * fft_scale;
    const float t = b + one_minus_cutoff;
    return t > 1.0f ? t : 1.0f;
}

// render.c:730-734
GLV_HD float gravity(float b, float applied, float g) {
    return (b >= applied ? b : applied) - g;
}

// render.c:844: (float)(log((double)y) / 3) with y = |x| + 1.0f already rounded to float (y >= 1).
//   mode 0  "strict": fp64 log + fp64 divide, the reference's expression verbatim
//   mode 1  "fast":   fp32 log, <= ~3e-7 relative (the parity bar is 1e-5)
template <int LOG_MODE> GLV_HD float log_third(float y);
template <> GLV_HD float log_third<0>(float y) { return (float) (::log((double) y) / 3); }
template <> GLV_HD float log_third<1>(float y) { return ::logf(y) * (1.0f / 3.0f); }

}  // namespace glv
