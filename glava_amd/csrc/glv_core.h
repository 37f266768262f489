// glv_core.h -- per-thread arithmetic and index maps of the spectrum kernels.
//
// Everything here is `GLV_HD` so the *same* code is compiled (a) by hipcc into the gfx950
// kernels (glv_kernel_tmpl.h via glv_inst.hip, glv_misc.hip) and (b) by g++ into tests/emu (a host "kernel emulator" that
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
// sub-passes (RB <= 5) whose every radix-2 butterfly performs exactly the reference's six
// individually rounded float operations with the reference's recurrence-generated twiddle,
// so results are bit-identical while the data moves through LDS only ceil(log2(nn)/log2(E))-1
// times.  Compile with -ffp-contract=off (no FMA contraction anywhere in this file).
#pragma once

#include <math.h>
#include <stdint.h>

// Tuning knobs (A/B builds of tools/tune.py and tools/rows_bench*.sh only) are quarantined behind ONE switch: a translation unit that defines
// any of them without -DGLV_TUNE_BUILD does not compile, and glava_amd/build.py's build() -- the product -- never passes either
// (tests/test_abi.py checks the recorded command lines).  (Round 5 removed the timing-experiment and rejected-variant switches of rounds
// 1-4 from the kernels: profiles/r05/removed_experiment_scaffolding.diff has them.)
#if !defined(GLV_TUNE_BUILD)
#if defined(GLV_ROWS_RB) || defined(GLV_BAR_BATCH_BIG) || defined(GLV_STATE_PAIR_MAX)
#error "tuning macros are for tools/tune.py A/B builds: compile with -DGLV_TUNE_BUILD (glava_amd.build.build_variant does); the product never defines them"
#endif
#endif

#if defined(__HIPCC__)
#define GLV_HD __host__ __device__ __forceinline__
#else
#define GLV_HD inline
#endif

// Scheduling fence: nothing is moved across it by the backend's instruction scheduler.  Used to
// bound how many table loads the compiler clusters (each in-flight window pair costs 4 VGPRs).
#if defined(__HIP_DEVICE_COMPILE__)
#define GLV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define GLV_SCHED_FENCE() ((void) 0)
#endif

namespace glv {

struct alignas(8) cf { float x, y; };        // one complex point == two consecutive floats of the reference's data[]
struct alignas(8) u32x2 { uint32_t x, y; };  // four interleaved s16 samples (L,R,L,R)
struct alignas(16) d2 { double x, y; };      // two consecutive window values
struct alignas(16) cf2 { cf a, b; };         // two consecutive complex points (one 16-byte access)

// ---- twiddle table layout ------------------------------------------------------------------
// One table per FFT size: stage with complex half-size L (L = 1,2,4,...,nn/2; reference
// mmax = 2L, render.c:814-839) occupies entries [L, 2L) -- entry 0 is unused, nn entries in total, so that
// every stage starts on an even entry: two consecutive twiddles W[L][2j], W[L][2j+1] are one aligned 16-byte
// piece (the last pass gathers them with one load per adjacent pair of groups).  The stages' tables are NOT
// subsets of each other (each runs its own float recurrence).
GLV_HD constexpr int tw_offset(int L) { return L; }

// The reference's six-rounding butterfly (render.c:826-832):  t = w*b;  b = a - t;  a = a + t.
//   tr = wr*b.re - wi*b.im;  ti = wr*b.im + wi*b.re     (four products, two sums, each rounded)
// On gfx950 a complex point lives in an aligned VGPR pair and the whole butterfly is five packed
// IEEE f32 instructions whose op_sel / neg modifiers do the broadcast, swap and sign work, so
// no operand ever has to be moved into place:
//   p02 = (wr*b.re, wr*b.im)      v_pk_mul_f32  W.lo broadcast
//   p13 = (wi*b.im, wi*b.re)      v_pk_mul_f32  W.hi broadcast, B halves swapped
//   t   = (p0 - p1, p2 + p3)      v_pk_add_f32  neg_lo on the second source
//   hi  = a - t ; lo = a + t      v_pk_add_f32  (neg_lo+neg_hi) / plain
// The host (emulator) build performs the same ten IEEE operations one by one.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float glv_f2 __attribute__((ext_vector_type(2)));
// The three dependent steps of one butterfly, separately callable so that a group of independent
// butterflies can be issued step by step (4-8 independent packed ops between dependent ones:
// no forwarding stalls / hazard nops from back-to-back dependent v_pk instructions).
template <bool SCALAR_W>
__device__ __forceinline__ void bf_mul(glv_f2& p02, glv_f2& p13, const cf& b, const cf& w) {
    const glv_f2 B = { b.x, b.y }, W = { w.x, w.y };
    if constexpr (SCALAR_W) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p02) : "s"(W), "v"(B));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p13) : "s"(W), "v"(B));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p02) : "v"(W), "v"(B));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p13) : "v"(W), "v"(B));
    }
}
__device__ __forceinline__ void bf_t(glv_f2& t, const glv_f2& p02, const glv_f2& p13) {
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(t) : "v"(p02), "v"(p13));
}
__device__ __forceinline__ void bf_out(cf& a, cf& b, const glv_f2& t) {
    const glv_f2 A = { a.x, a.y };
    glv_f2 hi, lo;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(A), "v"(t));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(A), "v"(t));
    a.x = lo.x; a.y = lo.y; b.x = hi.x; b.y = hi.y;
}
#endif
template <bool SCALAR_W = false>
GLV_HD void butterfly(cf& a, cf& b, const cf w) {
#if defined(__HIP_DEVICE_COMPILE__)
    glv_f2 p02, p13, t;
    bf_mul<SCALAR_W>(p02, p13, b, w);
    bf_t(t, p02, p13);
    bf_out(a, b, t);
#else
    const float p0 = w.x * b.x, p1 = w.y * b.y;
    const float p2 = w.x * b.y, p3 = w.y * b.x;
    const float tr = p0 - p1;
    const float ti = p2 + p3;
    const cf hi = { a.x - tr, a.y - ti };
    const cf lo = { a.x + tr, a.y + ti };
    a = lo; b = hi;
#endif
}

// Twiddle exactly (1, +0) -- the k = 0 column of every stage (render.c:821-822 start values).
//   tr = 1*b.re - (+0)*b.im = b.re,  ti = 1*b.im + (+0)*b.re = b.im   bit for bit, PROVIDED b holds
// no -0.0 (then (-0) - (-0) would give +0) and no Inf/NaN.  That holds for every value this
// path produces: PCM samples and window factors are finite, v/65535 and x*w (w > 0.07) are never
// -0, and a +- t under round-to-nearest yields -0 only from (-0) operands (induction over the
// stages).  f32 input may contain -0.0f or non-finite samples, so the f32 kernels do not take this shortcut
// (SubPass::run UNIT_SHORTCUT = false).
GLV_HD void butterfly_unit(cf& a, cf& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const glv_f2 A = { a.x, a.y }, B = { b.x, b.y };
    glv_f2 hi, lo;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(A), "v"(B));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(A), "v"(B));
    a.x = lo.x; a.y = lo.y; b.x = hi.x; b.y = hi.y;
#else
    const cf hi = { a.x - b.x, a.y - b.y };
    const cf lo = { a.x + b.x, a.y + b.y };
    a = lo; b = hi;
#endif
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

    // FIRST: the sub-pass starts at L0 = 1, so k0 = 0 for every lane: the twiddles are wave-uniform
    // (scalar registers) and the ksub = 0 twiddle of each stage is exactly (1, +0).
    // UNIT_SHORTCUT: that twiddle's butterfly is evaluated as a +- b (butterfly_unit) -- bit-exact when b holds no -0.0, Inf
    // or NaN, which is guaranteed for s16 input only.  The f32 input paths pass false and perform the reference's full
    // multiply-add form with the table's (1, +0), so that (+0) * Inf = NaN and the sign of an exact zero come out as in the
    // reference (tests/test_concurrency_state.py::test_f32_special_values_against_the_compiled_reference).
    template <bool FIRST, bool UNIT_SHORTCUT = true>
    GLV_HD static void run(cf (&v)[R], const cf (&tw)[R > 1 ? R - 1 : 1]) {
        constexpr bool SC = FIRST && UNIT_SHORTCUT;
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int bit = 1 << (RB - 1 - s);
#if defined(__HIP_DEVICE_COMPILE__)
            // the R/2 butterflies of a stage are independent: issue them step-wise in groups of GRP
            constexpr int NB = R / 2, GRP = NB < 4 ? NB : 4;
#pragma unroll
            for (int g0 = 0; g0 < NB; g0 += GRP) {
                glv_f2 p02[GRP], p13[GRP], t[GRP];
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int r0 = lo_slot(g0 + j, bit);
                    if (!(SC && ksub_of(r0, s) == 0)) bf_mul<FIRST>(p02[j], p13[j], v[r0 | bit], tw[(1 << s) - 1 + ksub_of(r0, s)]);
                }
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int r0 = lo_slot(g0 + j, bit);
                    if (!(SC && ksub_of(r0, s) == 0)) bf_t(t[j], p02[j], p13[j]);
                }
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int r0 = lo_slot(g0 + j, bit);
                    if (SC && ksub_of(r0, s) == 0) butterfly_unit(v[r0], v[r0 | bit]);
                    else bf_out(v[r0], v[r0 | bit], t[j]);
                }
            }
#else
#pragma unroll
            for (int r0 = 0; r0 < R; ++r0) {
                if (r0 & bit) continue;
                if (SC && ksub_of(r0, s) == 0) butterfly_unit(v[r0], v[r0 | bit]);
                else butterfly<FIRST>(v[r0], v[r0 | bit], tw[(1 << s) - 1 + ksub_of(r0, s)]);
            }
#endif
        }
    }
    // the idx-th slot (in increasing order) whose `bit` is clear
    GLV_HD static constexpr int lo_slot(int idx, int bit) { return ((idx & ~(bit - 1)) << 1) | (idx & (bit - 1)); }

    // index into the size-nn twiddle table of the (s, ksub) twiddle for group constant k0
    GLV_HD static constexpr int tw_index(int L0, int k0, int s, int ksub) {
        return tw_offset(L0 << s) + k0 + L0 * ksub;
    }
};

// ---- pass plan: log2(nn) bits split into sub-passes of at most LOG_E bits ------------------------
// E = 2^LOG_E points per lane (16 or 8), T = nn/E lanes cooperate on one FFT.  Passes 0..P-2 are
// radix E; the last takes the remaining bits and handles E >> RB groups per lane.  E = 16 means
// fewer LDS exchanges (N=4096: 4+4+3), E = 8 half the registers per lane and twice the waves per
// row (3+3+3+2) -- which one wins is a per-size tuning decision (glv_inst.hip).
template <int LOG_NN, int LOG_E = 4>
struct Plan {
    static constexpr int NN = 1 << LOG_NN;
    static constexpr int E = 1 << LOG_E;
    static constexpr int T = NN / E;
    static constexpr int P = (LOG_NN + LOG_E - 1) / LOG_E;
    GLV_HD static constexpr int rb(int pass) { return pass < P - 1 ? LOG_E : LOG_NN - LOG_E * (P - 1); }
    GLV_HD static constexpr int log_l0(int pass) { return LOG_E * pass; }
};

// LDS address (in complex units, within one FFT's exchange region) of element index q for the
// exchange that follows pass `pass`.  Only the first pass writes with a lane stride of E
// elements (q = E*tid + e); padding one element per E turns that into a stride of E+1 -- conflict
// free for the 16-lane ds_write_b64 groups -- and keeps every access of a phase at
// `lane base + compile-time constant` (an XOR swizzle would need a separate address VGPR per
// element).  The following read (q = i*nn/R + tid) sees a few 2-way conflicts per 32-lane group
// (+1 LDS cycle); later exchanges are contiguous per 16 lanes and conflict free as they are.
GLV_HD constexpr int lds_index(int pass, int q, int log_e = 4) { return pass == 0 ? q + (q >> log_e) : q; }

// `uniform base + 32-bit byte offset` access (see glv_frame.h "addressing")
template <typename V> GLV_HD V ld(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const V*>(static_cast<const char*>(base) + byte_off);
}
template <typename V> GLV_HD void st(void* base, uint32_t byte_off, const V& v) {
    *reinterpret_cast<V*>(static_cast<char*>(base) + byte_off) = v;
}

// TILTREG 3 (log_mode 1 only): the folded tilt factor tilt(n) * ln2/3 from two fused multiply-adds instead of the reference's
// separately rounded chain.  tilt(n) = max(n/N * fft_scale + (1 - fft_cutoff), 1) is linear in the float index n above its
// clamp, so   tilt(n) * k  ~=  max(fma(c, S, B), k),   S = fl(fft_scale * (1/N) * k),  B = fma((float) n_base, S, fl(one_minus_cutoff * k)),
// n = n_base + c with c >= 0 a compile-time constant.  Two instructions per value where the exact evaluation costs six; at
// most four roundings of 2^-24 (S, O, B, the final fma) against the real value, as many as the reference's own chain has:
// <= 4.5e-7 relative to the reference's (thrice rounded) product tilt * k for every index (tests/test_emulator.py
// test_fused_tilt_formula_is_within_its_bound), next to the 1.8e-7 of the hardware log -- the contract of log_mode 1 is 1e-5.
// log_mode 0 and 2 keep the reference's own float operations (TILTREG 2).
constexpr float kLn2Third = (float) (0.69314718055994530942 / 3.0);
struct TiltLin { float S, O, K; };
GLV_HD TiltLin tilt_lin(float inv_n, float fft_scale, float one_minus_cutoff) {
    TiltLin t;
    t.S = (fft_scale * inv_n) * kLn2Third;          // inv_n is a power of two: the first product is exact
    t.O = one_minus_cutoff * kLn2Third;
    t.K = kLn2Third;
    return t;
}
GLV_HD float tilt_lin_base(const TiltLin& t, int n_base) { return __builtin_fmaf((float) n_base, t.S, t.O); }
GLV_HD float tilt_lin_at(const TiltLin& t, float base, int c) { return __builtin_fmaxf(__builtin_fmaf((float) c, t.S, base), t.K); }

// ---- scalar pieces ----------------------------------------------------------------------------
// fifo.c:105-106: (float) s16 / (float) 65535, IEEE single division.
// Evaluated without a divide: 1/65535 = c_hi + c_lo + (< 2^-48 relative), c_hi = 0x1.0001p-16, c_lo = 0x1.0001p-48, and
//     q = fma(v, c_hi, v * c_lo)
// is within 2^-47 (relative) of v/65535 before its one rounding, while v/65535 (|v| <= 65535) is never closer than
// 2^-40 to a float rounding boundary (65535 x is an integer, a boundary times 65535 is an odd multiple of a power of
// two) -- so q IS the correctly rounded quotient: two instructions per sample instead of the three of the
// Newton-style sequence of round 1.  tests/test_emulator.py checks every integer argument against the division
// on the CPU, tests/test_gpu_parity.py the 65536 sample values on the device, tests/test_gl_storage.py the 65536 texels.
GLV_HD float div_65535(float fv) {
    const float c_hi = 0x1.0001p-16f, c_lo = 0x1.0001p-48f;
    return __builtin_fmaf(fv, c_hi, fv * c_lo);
}
GLV_HD float unpack_s16(int v) { return div_65535((float) v); }
// fifo.c:98-102 mono mix: C int arithmetic, truncation toward zero.
GLV_HD float unpack_s16_mono(int l, int r) { return unpack_s16((l + r) / 2); }

// render.c:794: data[i] *= window(i, N)  -- float * double -> double -> float.
GLV_HD float apply_window(float x, double w) { return (float) ((double) x * w); }
// The same product for s16 samples without fp64: x takes only the 65536 values k / 65535, and for every window value w there
// is a float pair (hi, lo) -- hi = (float) w, lo = (float) (w - hi), for 14 of the 65 280 window positions of all sizes moved
// by one ulp -- with  fma(x, hi, x * lo) == (float) ((double) x * w)  for ALL of them (the pair is searched and the identity
// checked for every sample value on the device when a batch is created: glv_misc.hip glv_window_split_kernel; every (k, i) of
// every size again by tests/test_window_split.py).  Two packed instructions per complex point instead of six fp64-rate ones
// (two conversions up, two products, two conversions down): the kernels run at the package power limit, so the cheaper
// operations are worth more than their issue slots.  Layout: one WinSplit per complex point, in place of its two doubles.
struct alignas(16) WinSplit { float hi0, hi1, lo0, lo1; };    // window of samples 2c, 2c + 1
GLV_HD float apply_window_split(float x, float hi, float lo) { return __builtin_fmaf(x, hi, x * lo); }

// render.c:845 tilt factor; inv_n = 1/N is a power of two so n*inv_n == (float)n/(float)N exactly.
GLV_HD float tilt(int n, float inv_n, float fft_scale, float one_minus_cutoff) {
    const float a = (float) n * inv_n;
    const float b = a * fft_scale;
    const float t = b + one_minus_cutoff;
    return t > 1.0f ? t : 1.0f;
}

// tilt as the kernels multiply it: log_mode 1 folds ln2/3 in (log2 -> log/3) with ONE float multiply,
// the same expression on the host (glv_tables.h make_tilt) and on the device (kLn2Third: defined with tilt_lin above)
template <bool FOLD_LN2_3>
GLV_HD float tilt_factor(int n, float inv_n, float fft_scale, float one_minus_cutoff) {
    const float t = tilt(n, inv_n, fft_scale, one_minus_cutoff);
    return FOLD_LN2_3 ? t * kLn2Third : t;
}

// The GL_R16 texel a float becomes when handle_audio uploads the finished buffer
// (render.c:521-524: glTexImage1D(GL_TEXTURE_1D, 0, GL_R16, sz, 0, GL_RED, GL_FLOAT, buf)): clamp to [0, 1],
// scale by 65535, round to nearest (ties to even).  x * 65535 has at most 40 significant bits, so the double
// product is exact and rint() is the only rounding.  NaN clamps to 0 (fmax/fmin semantics).  On the device this
// is one v_cvt_pknorm_u16_f32 per two values (tests/test_gpu_parity.py checks every one of the 2^32 floats).
GLV_HD uint32_t unorm16(float x) {
    const float c = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f);
    return (uint32_t) __builtin_rint((double) c * 65535.0);
}
GLV_HD uint32_t pack_unorm16(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short glv_us2 __attribute__((ext_vector_type(2)));
    const glv_us2 p = __builtin_amdgcn_cvt_pknorm_u16(lo, hi);
    return __builtin_bit_cast(uint32_t, p);
#else
    return unorm16(lo) | (unorm16(hi) << 16);
#endif
}

// what a shader reads back from a GL_R16 texel: c / 65535 (OpenGL 4.6 eq. 2.1), correctly rounded -- div_65535
// (checked for all 65536 texel values by tests/test_gl_storage.py)
GLV_HD float unorm16_to_float(uint32_t c) { return div_65535((float) c); }
// a float written to a GL_R16 render target / texture and read back (render.c:523, :1718: every 1-D texture and FBO of
// the audio passes is GL_R16): clamped to [0, 1] and quantised to 16 bits
GLV_HD float through_r16(float x) { return unorm16_to_float(unorm16(x)); }

// render.c:730-734
GLV_HD float gravity(float b, float applied, float g) {
    return (b >= applied ? b : applied) - g;
}

// x / F for the final division of an averaging pass (average_pass.frag: `/ _AVG_FRAMES`) on GL_R16 values -- x is a weighted sum of
// at most 64 texel values c / 65535, i.e. 0 or a normal float in [2^-20, 128) -- correctly rounded without the divide expansion
// (~10 instructions: scale, reciprocal, three refinement steps, fix-up): with r = RN(1 / F), q0 = RN(x r), the remainder
// rem = x - q0 F is exact in one fused multiply-add and q = RN(q0 + rem r) is the correctly rounded quotient (Markstein).  Checked
// against the division for every float of that range and every F (tests/test_gl_storage.py through the host emulator).
GLV_HD float div_frames(float x, float F, float rcpF) {
    const float q0 = x * rcpF;
    const float rem = __builtin_fmaf(-q0, F, x);
    return __builtin_fmaf(rem, rcpF, q0);
}

// ---- GL_R16 state (glv_params.gl_storage == 1) ---------------------------------------------------------------------------
// Every value the GL passes of render.c:2188-2265 keep between frames lives in a GL_R16 texture (render.c:523, :1718), i.e. IS
// a 16-bit integer: the gravity store and the ring of averaged frames are kept as what they are -- uint16 texels, two per
// uint32 (one complex point: low half = the even float of the row) -- instead of the floats c / 65535 they read back as.
// Lossless by construction (through_r16(x) == unorm16_to_float(unorm16(x))), half the state traffic.
// the two floats a shader reads back from a packed texel pair (OpenGL 4.6 eq. 2.1: c / 65535, correctly rounded: div_65535)
GLV_HD cf texels_to_float(uint32_t p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const glv_f2 K = {(float) (p & 0xffffu), (float) (p >> 16)};        // v_cvt_f32_u32 with a word selector each
    // the two constants as SCALAR register pairs (one scalar operand per VOP3P instruction is allowed): as vector operands the backend
    // rebuilt each 64-bit pair with a v_mov_b32 in front of every use -- two more instructions per texel pair, 80 pairs per lane and row
    const glv_f2 CH = {0x1.0001p-16f, 0x1.0001p-16f}, CL = {0x1.0001p-48f, 0x1.0001p-48f};
    glv_f2 t, q;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(K), "s"(CL));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(q) : "v"(K), "s"(CH), "v"(t));
    return cf{q.x, q.y};
#else
    return cf{unorm16_to_float(p & 0xffffu), unorm16_to_float(p >> 16)};
#endif
}
// The gravity pass on texels: store' = Q(max(tex, store) - g)  (GL_MAX blend render.c:2199-2210 -- exact on texel values, the
// float comparison of c / 65535 is the integer comparison of c -- then gravity_pass.frag's subtraction and the write to the
// GL_R16 target).  For a given g the whole step is a function of ONE 16-bit integer m = max(tex, store), and for almost every g
// (whenever g * 65535 is not within rounding noise of a half-integer) that function is  m -> max(m - D, 0)  with one integer
// D: the host checks this for all 65 536 values of m whenever g changes (glv_tables.h gravity_r16_integer_step) and passes
// `sub` = D | D << 16, `exact_int` = 1; the step is then two packed 16-bit integer instructions per complex point
// (v_pk_max_u16, v_pk_sub_u16 clamp) instead of two conversions, the two-instruction quotient, the subtraction and the
// conversion back.  Where the check fails (exact_int = 0) the float operations run as written.
GLV_HD uint32_t gravity_r16(uint32_t tex, uint32_t store, float g, uint32_t sub, uint32_t exact_int) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t m;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(m) : "v"(tex), "v"(store));
    if (exact_int) {                                                    // uniform (callers with many texels test it ONCE around their loop: gravity_r16_int / _flt)
        uint32_t r;
        asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(m), "v"(sub));
        return r;
    }
    const cf f = texels_to_float(m);
    return pack_unorm16(f.x - g, f.y - g);
#else
    const uint32_t a0 = tex & 0xffffu, a1 = tex >> 16, b0 = store & 0xffffu, b1 = store >> 16;
    const uint32_t m0 = a0 > b0 ? a0 : b0, m1 = a1 > b1 ? a1 : b1;
    if (exact_int) {
        const uint32_t d = sub & 0xffffu;
        return (m0 > d ? m0 - d : 0u) | ((m1 > d ? m1 - d : 0u) << 16);
    }
    return unorm16(unorm16_to_float(m0) - g) | (unorm16(unorm16_to_float(m1) - g) << 16);
#endif
}

// The two arms of gravity_r16 on their own, for callers that hold many texel pairs: a uniform test per PAIR stays a pair of branches per
// pair in the unrolled code (the backend does not merge them across the inline assembly: 32 branches per lane and row in the frame
// kernel's GL epilogue); tested once around the loop it costs nothing.  Same instructions, same bits.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t gravity_r16_int(uint32_t tex, uint32_t store, uint32_t sub_uniform) {
    uint32_t m, r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(m) : "v"(tex), "v"(store));
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(m), "s"(sub_uniform));
    return r;
}
__device__ __forceinline__ uint32_t gravity_r16_flt(uint32_t tex, uint32_t store, float g) {
    uint32_t m;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(m) : "v"(tex), "v"(store));
    const cf f = texels_to_float(m);
    return pack_unorm16(f.x - g, f.y - g);
}
#else
inline uint32_t gravity_r16_int(uint32_t tex, uint32_t store, uint32_t sub) { return gravity_r16(tex, store, 0.0f, sub, 1u); }
inline uint32_t gravity_r16_flt(uint32_t tex, uint32_t store, float g) { return gravity_r16(tex, store, g, 0u, 0u); }
#endif

// render.c:844: (float)(log((double)y) / 3) with y = |x| + 1.0f already rounded to float (y >= 1).
//   mode 0  fp64: table-driven double-precision log (relative error ~2^-50) times 1/3, rounded to
//           float once -- equals the reference's float result except when the double value sits
//           within ~1e-15 relative of a float rounding boundary (probability ~1e-8 per value)
//   mode 1  fast: the hardware log2 (v_log_f32, 1 ulp) times (ln2/3 * tilt) in one multiply;
//           <= ~3e-7 relative, the parity bar for magnitudes is 1e-5
//   mode 2  audit: the device libm's fp64 log and a true fp64 division, the reference's
//           expression verbatim (slow; for cross-checking mode 0)
struct LogEntry;
template <int LOG_MODE> GLV_HD float log_third(float y, const LogEntry* tab, int bits);

#if defined(__HIP_DEVICE_COMPILE__)
#define GLV_LOG2F(x) __builtin_amdgcn_logf(x)
#else
#define GLV_LOG2F(x) ::log2f(x)
#endif

// log table: for j = 0 .. 2^B - 1, c_j = 1 + j / 2^B (the value of the top B mantissa bits),
// tab[j] = { 2^-23/c_j, log(c_j)/3 } rounded to double.  Generated on the host by glv::make_log_table
// (glv_tables.h); the kernel stages it into LDS (random 16-byte gathers are what LDS is good at).
// B = 9 (8 KiB) for N <= 8192 since round 3: r < 2^-9 lets the polynomial below end one term earlier (one v_fma_f64 per value less,
// log_mode 0 77.7 -> 78.8 M frames/s at N=4096) and still gives the reference's float for EVERY float in [1, 2^14)
// (tests/test_gpu_parity.py test_magnitude_stage_every_float: 0 of 117 440 512 differ); B = 10 would not leave two
// N=4096 workgroups room in a CU's LDS.
// The table in HBM always has 2^9 entries; a kernel stages the 2^B it uses (every 2^(9-B)-th entry) into LDS, B by transform
// size: two N=16384 workgroups per CU have no 4 KiB to spare (with B = 9 only one fits: 0.80 -> 1.04 ms), so N >= 16384 keeps B = 8.
constexpr int kLogTabMaxBits = 9;
constexpr int kLogTabMaxSize = 1 << kLogTabMaxBits;
GLV_HD constexpr int log_tab_bits_of(int log_nn) { return log_nn >= 13 ? 8 : 9; }
struct alignas(16) LogEntry { double inv_c, log_c3; };

#if defined(__HIP_DEVICE_COMPILE__)
// d = fma(a, b, c) with c in scalar registers: VOP3 form, so the constant is not clobbered and
// need not be re-materialised into a VGPR pair before every use (what v_fmac_f64 would force).
__device__ __forceinline__ double fma_sc(double a, double b, double c_uniform) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
}
#define GLV_FMA_SC(a, b, c) fma_sc(a, b, c)
#else
#define GLV_FMA_SC(a, b, c) __builtin_fma(a, b, c)
#endif

// y >= 1 finite.  y = 2^e * m, m in [1,2); c = m truncated to B = `bits` mantissa bits; r = (m - c)/c in
// [0, 2^-B).  log(y)/3 = e*(ln2/3) + log(c)/3 + r*P(r),
//   P(r) = (1 - r/2 + r^2/3 - r^3/4 + r^4/5 [- r^5/6 when B = 8])/3      (truncation r^5/6 < 2^-47 relative at B = 9)
// All polynomial arithmetic is ours (fused): only the final float matters, and it equals the
// reference's (float)(log(y)/3) for every float the stage can see (exhaustive test, above).
GLV_HD float log_third_table(float y, const LogEntry* tab, int bits) {      // tab: 2^bits entries (bits is a compile-time constant at every call site)
    const uint32_t u = __builtin_bit_cast(uint32_t, y);
    const int e = (int) (u >> 23) - 127;
    const LogEntry t = ld<LogEntry>(tab, (u >> (19 - bits)) & (uint32_t) (((1 << bits) - 1) << 4));   // j * 16 bytes
    // r = (m - c) / c with m - c = k * 2^-23, k = the low 23 - B mantissa bits: the table carries 2^-23 / c_j, so r is one
    // exact integer conversion and one product (scaling by a power of two commutes with the rounding: the same bits
    // as (double) (m - c) * (1 / c_j), two instructions fewer)
    const double r = (double) (u & ((1u << (23 - bits)) - 1u)) * t.inv_c;
    double p = 1.0 / 15.0;
    if (bits < 9) p = GLV_FMA_SC(-1.0 / 18.0, r, 1.0 / 15.0);
    p = GLV_FMA_SC(p, r, -1.0 / 12.0);
    p = GLV_FMA_SC(p, r, 1.0 / 9.0);
    p = GLV_FMA_SC(p, r, -1.0 / 6.0);
    p = GLV_FMA_SC(p, r, 1.0 / 3.0);
    const double hi = __builtin_fma((double) e, 0.6931471805599453094 / 3.0, t.log_c3);
    return (float) __builtin_fma(r, p, hi);
}
template <> GLV_HD float log_third<0>(float y, const LogEntry* tab, int bits) { return log_third_table(y, tab, bits); }
// y = +Inf or NaN (f32 input rows holding non-finite samples): the reference's (float)(log(y) / 3) is y itself (Inf) or NaN.
// The hardware log of mode 1 and libm's of mode 2 do that on their own; the table-driven log of mode 0 decodes the exponent
// field and needs the select.  s16 input cannot produce such values (|FFT output| <= n / 2), its kernels skip it.
template <int LOG_MODE, bool NONFINITE>
GLV_HD float log_third_nf(float y, const LogEntry* tab, int bits = kLogTabMaxBits) {
    const float r = log_third<LOG_MODE>(y, tab, bits);
    if constexpr (LOG_MODE == 0 && NONFINITE) return __builtin_bit_cast(uint32_t, y) >= 0x7f800000u ? y : r;
    else return r;
}
// mode 1 returns log2(y); the ln2/3 factor is folded into the tilt table the kernel multiplies with
// (glv_tables.h make_tilt with fold_ln2_3), one multiply less per value
template <> GLV_HD float log_third<1>(float y, const LogEntry*, int) { return GLV_LOG2F(y); }
template <> GLV_HD float log_third<2>(float y, const LogEntry*, int) { return (float) (::log((double) y) / 3); }

}  // namespace glv
