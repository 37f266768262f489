// glv_frame.h -- the phases one FFT "slot" (T = nn/E cooperating lanes, E = 8, 16 or 32 points per lane) runs
// per channel row, and the arithmetic of the operators that follow the transform.
//
// Shared between the gfx950 kernels (glv_kernel_tmpl.h, glv_misc.hip) and the host emulator (tests/emu):
// every function takes the lane id explicitly and touches memory only through the pointers it is given,
// so the emulator can call the same phase for tid = 0..T-1 in turn where the kernel has T lanes and a
// slot-scoped synchronisation.
//
// Data flow for one channel row (nn = N/2 complex points, P = ceil(log2(nn)/log2(E)) passes):
//   load_pcm / load_f32*_raw     HBM -> registers (one frame / row ahead of its use)
//   unpack_window / window_*     fused unpack (fifo.c:94-110) + window (render.c:792-795)
//   compute<0>                   log2(E) radix-2 stages in registers (twiddles uniform: SGPRs)
//   exchange_write / sync / exchange_read     through LDS (padded by one point per E after pass 0)
//   compute<1> ... compute<P-1>
//   epilogue                     abs/log/tilt (render.c:842-846) + gravity (720-736) + average (738-771),
//                                registers -> HBM (or -> LDS when the bars are computed in the kernel)
//   bar_item_*                   smooth_audio() bin averaging (smooth.glsl:13-40) in 64-tap chunks
#pragma once

#include "glv_core.h"

namespace glv {

enum InMode { IN_S16_STEREO = 0, IN_F32_PLANAR = 1, IN_S16_RING = 2, IN_F32_STEREO = 3, IN_F32_RING = 4 };
enum Epi { EPI_RAW = 0, EPI_MAG = 1, EPI_MAG_STATE = 2, EPI_RAW_STATE = 3 };

// ops bits as in include/glv_spectrum.h
enum : uint32_t { OP_FFT = 1u, OP_GRAVITY = 2u, OP_AVERAGE = 4u, OP_RAW = 8u, OP_WRANGE = 16u, OP_BARS = 32u, OP_SMOOTH = 64u, OP_MAGNITUDE = 128u, OP_R16 = 256u, OP_PRIVATE_STATE = 512u };

// one output bar of GLV_OP_BARS: taps are consecutive bins [first_bin, first_bin + count) with weights
// tap_w[tap_offset ...]; weight_sum = float sum of the weights in tap order (smooth.glsl:31-36)
struct BarDesc { uint32_t first_bin, count, tap_offset; float weight_sum; };
struct BarModeBlock { uint32_t w_off, maxcount; };      // bars [64 b, 64 b + 64) of glv_bars_mode_kernel: where their [tap][lane] weights start, taps of the longest (rounded up to kBarModeUnroll)
constexpr uint32_t kBarModeUnroll = 4;
// One work item of GLV_OP_BARS: a chunk of bar_chunk_of(n) = 16 / 32 / 64 consecutive taps of one bar, taken by one group of
// bar_lanes_of(n) = 2 / 4 / 8 lanes.  Everything a step needs, ready to use (the loop is instruction-bound): w_byte = byte offset
// of the chunk's weights in tap_w (zero-padded by make_bar_taps; an all-zero block for padding items, which
// therefore add exactly 0), tex_byte = byte offset of the chunk's first tap in the row, res = the bar this chunk
// completes, or `bars` (a dump slot) when it does not, keep = 0.0f when the chunk opens a bar (the running total
// restarts), 1.0f otherwise.
struct alignas(16) BarItem { uint32_t w_byte, tex_byte, res; float keep; };
// Many bars (>= 256; glv_tables.h make_bar_mtiles): one tile of 32 consecutive bars -- first bin, pairs of bins, where its 64
// weights per pair start -- and one round of glv_bars_rows_kernel: tiles [k0, k1) whose bins [origin, end) sit in the LDS ring.
struct alignas(16) BarTile { uint32_t k0, k1, origin, end; };
struct alignas(16) BarMTile { uint32_t k0, origin, steps, w_off; };
// texel rows (glv_tables.h make_bar_itiles): per bar the epilogue's rounding constant and shift, texel = (floor(T / 2^16) + c) >> s, s = P - 16 in [1, 15];
// a bar whose weights sum to 0 has every digit 0, c = 0 and s = kBarIFinNone (the same expression then gives texel 0; the float form NaN)
struct alignas(8) BarIFin { uint32_t c, s; };
constexpr uint32_t kBarIFinNone = 16;
constexpr uint32_t kBarILookAhead = 4;      // steps of zero weights behind the last tile's: the kernel's weight requests run ahead of the step it computes

struct FrameArgs {
    const void* in;        // s16: int16 [units/2][n][2] (a unit is one channel row of a frame);  f32 planar: float [units][n];
                           // f32 stereo (PulseAudio layout, pulse_input.c:155-178): float [units/2][n][2]
    float* out;            // [units][n]
    const float* grav;     // [rows][n] gravity state read by gravity without average: the previous gravity output
    float* grav_w;         // [rows][n] where that chain stores its new state.  transform_gravity's output IS its new state
                           // (render.c:733-734), so the host points this at the caller's output buffer (and leaves `out` NULL)
                           // unless a private copy was asked for; == grav for the in-place form
    float* hist;           // [rows][F][n] history ring (average); doubles as gravity state
    const cf* tw;          // nn entries (entry 0 unused), layout glv::tw_offset
    const double* win;     // n window values (render.c:660 as expanded at :794)
    const void* win_split; // the same as float pairs for s16 samples (glv_core.h WinSplit [n/2]); which one a kernel reads: win_split_of
    const LogEntry* logtab; // kLogTabMaxSize entries, log_mode 0 (glv_core.h)
    const float* tilt;     // n tilt factors max(n/N*fft_scale + (1 - fft_cutoff), 1) (render.c:845), host-generated
    uint32_t units;        // channel rows to process (2 per stereo frame)
    uint32_t ops;
    uint32_t F, head;      // ring: `head` receives the current frame; ages oldest..newest are
                           // head+1, ..., head+F-1, head  (mod F)
    uint32_t mono;         // fifo.c:98-102
    uint32_t avg_window;
    uint32_t gl_storage;   // the GL passes' GL_R16 storage is modelled (apply_state / gl16_state_block).  1: the state arrays
                           // (`grav`, `grav_w`, `hist`) hold uint16 texels -- uint16 [rows][n] / [rows][F][n], the pointers are
                           // reinterpreted; 2: the same values kept as the floats c / 65535 (the pass-by-pass form: glv_post_kernel only)
    uint32_t grav_sub;     // gl_storage 1: the gravity step as a packed 16-bit integer subtraction D | D << 16, valid when
    uint32_t grav_int;     //   grav_int != 0 (glv_core.h gravity_r16, checked on the host for every texel value)
    uint32_t bars_r16;     // fused bars of the GL_R16 chain: bars_out is uint16 [units][bars] (the smooth pass's render target)
    uint32_t log_mode;     // glv_post_kernel's OP_MAGNITUDE (the frame kernels take it as a template parameter)
    uint32_t rot;          // ring modes (RING kernels): index of the ring's oldest stereo frame = where the window starts
    float inv_n, fft_scale, one_minus_cutoff, g, F_as_float;
    float F_rcp;           // RN(1 / F): the GL_R16 chain's final division (glv_core.h div_frames)
    uint32_t out_limit;    // gl_storage 1, rows handed to a later kernel only (the bars of a second launch): bytes of a float row that are
                           // written at all -- the pre-smoothing pass samples bins below 0.31 n, the rest of the row is not stored (0: no limit);
                           // texel rows (OP_R16: what the i8 matrix-core pass reads) stop at the same bin, i.e. at half as many bytes
    uint32_t live_points;  // gl_storage 1, GLV_OP_BARS_ONLY batches (0: every point): complex points [0, live_points) of a row are all the chain's consumer (the
                           // bars of a second launch) ever samples -- the LIVE kernel class keeps the gravity store and the ring, and computes magnitude,
                           // upload, gravity and average, for the last pass's blocks of L0 points that hold them, and touches nothing beyond
    double wts[64];        // window_frame weights, oldest first (render.c:661 as expanded at :766); GLV_MAX_AVG_FRAMES
    float wts32[64];       // the same rounded to float: the GL passes' arithmetic is the shader's, 32-bit (weighted_texels)
    // fused GLV_OP_BARS (stateful kernels, lanes-per-row a multiple of 64): the finished row goes to the
    // slot's LDS region instead of HBM and only the bars leave the chip
    const BarDesc* bar_desc;
    const BarItem* bar_items;   // [bar_nsteps + batch][groups] work lists (glv_tables.h make_bar_items), groups = T/8
    const float* bar_w;
    float* bars_out;            // [units][bars], nullptr = not fused
    uint32_t bars;
    uint32_t bar_nsteps;        // multiple of bar_batch_of(log2 nn)
};

// ---- GLV_OP_BARS arithmetic (smooth.glsl:25-40; tex clamped to [0,1] like the GL_R16 texture the
// shader samples, render.c:523) ---------------------------------------------------------------------
// A bar's taps are cut into chunks of 64; a chunk is summed by a group of 8 lanes: lane l takes the eight CONSECUTIVE taps
// 8l .. 8l+7 (two 16-byte loads of their weights) and runs two fused-multiply-add chains over them, one over its even taps
// and one over its odd taps, each from +0 in tap order (e = fma(x0, w0, 0), fma(x2, w2, e), ...: a v_pk_fma_f32 does one
// link of both chains); the lane's sum is e + o; the group adds its 8 lane sums with a fixed DPP pattern (neighbours,
// pairs of pairs, the two quads); the bar's running total takes the chunk totals in chunk order.  This order IS the
// contract of GLV_OP_BARS (smooth.glsl's loop adds tap by tap; what a GL driver's compiler makes of that loop -- fma or
// not, unrolled or not -- is not defined): the oracle's glvo_bars_chunked follows it and the GPU tests demand its bits;
// against the tap-by-tap order of the shader text it differs by summation rounding only (<= 2e-4 relative,
// tests/test_glsl_twins.py).  Which group of which wave takes a chunk does not enter the arithmetic, so the fused epilogue
// (row in LDS, T/8 groups per row) and glv_bars_kernel (row in HBM, 32 groups per row) give the same bits.
// A bar's taps are counted by BIN: its range is contiguous, a bin that round(s) skips (s += 1.0 rounding up across a binade
// boundary) is a tap of weight +0.
// FROM 256 BARS UP (the pre-smoothing pass, bars == n) the contract is simpler: ONE fused-multiply-add chain over the bar's taps in
// bin order from +0, acc = fma(w, x, acc), then / weight_sum -- what the matrix cores compute for 32 bars x 64 rows at a time
// (glv_tables.h make_bar_mtiles, glv_misc.hip glv_bars_rows_kernel / glv_bars_seq_kernel; never fused into the frame kernel).
// Why this shape: the loop is VALU-issue bound (the chip runs at its power limit, time follows the instruction count).
// Round 2's version (16 lanes x 4 taps, mul + add, flags unpacked from a bit field) spent 41 instructions per 4 taps;
// this one spends ~32 per 8: per-step bookkeeping is amortised over twice the taps, one DPP level is gone, multiplies
// and adds are fused and packed, and the end-of-bar logic is a multiply by `keep` and an unconditional store.
// (One WAVE per bar, the very first version, spent ~100 instructions per bar on mostly idle lanes.)
constexpr int kBarTaps = 8;            // consecutive taps per lane
// Lanes per group, i.e. the chunk a bar's taps are cut into (8 taps per lane): a bar is at least one chunk, and its width grows
// with the transform size (80 bars: ~15 taps at n = 1024, ~29 at 2048, ~57 at 4096, ~230 at 16384), so small transforms take
// small chunks -- 16 taps (2 lanes) up to n = 1024, 32 (4 lanes) at n = 2048, 64 (8 lanes) above: with 64-tap chunks a row of
// n = 1024 was ten steps of eight groups at 23 % useful taps (fft -> gravity -> 80 bars 0.83 ms; with 16-tap chunks four steps
// of 32 groups: 0.73 ms).  The chunk size is part of the documented summation order (glvo_bars_chunked restates it).
GLV_HD constexpr int bar_lanes_of(uint32_t n) { return n <= 1024u ? 2 : (n == 2048u ? 4 : 8); }
GLV_HD constexpr uint32_t bar_chunk_of(uint32_t n) { return (uint32_t) (bar_lanes_of(n) * kBarTaps); }
// Which window table an s16 kernel multiplies by: the float pairs (two packed instructions per complex point instead of six
// fp64-rate ones) measured +5 % (N=4096 -> GL_R16), +6 % (N=8192), +4..9 % (N=16384 gravity chains and bars), +2..15 % (N=32768)
// and 0 % on the N=4096 f32 pass; the one loser is the stateless f32 pass of N=16384 (-2.3 %, 20.8 vs 20.3 M frames/s on the
// same box, profiles/r03/ab_split.txt), which keeps the fp64 product.  Same bits either way.
GLV_HD constexpr bool win_split_of(int log_nn, int stateful) { return !(log_nn == 13 && stateful == 0); }
constexpr int kBarBatch = 2;           // work-list steps whose loads are issued together (glv_bars_kernel; the fused loop: bar_batch_of)
// The fused loop's batch per transform size: a batch is one exposed L2 round trip (the weights; ~0.5 us per row that nothing in
// the workgroup covers), so the large sizes, whose rows are 10 steps of 32 groups, take six steps per trip (N=16384
// fft -> gravity -> 80 bars 0.748 -> 0.672 ms) -- where the epilogue's registers are free again; the sizes with 2-6 steps per row
// lose more to padded steps and register pressure than the saved trip is worth (N=8192: 0.652 / 0.648 / 0.68 / 0.73 ms with
// 2 / 3 / 4 / 6; N=4096: 0.569 / 0.566 / 0.662 with 2 / 4 / 6; N=1024: 0.83 / 0.87 / 1.2 / 1.8: profiles/r03/ab_bars_phase.txt).
#if !defined(GLV_BAR_BATCH_BIG)
#define GLV_BAR_BATCH_BIG 6
#endif
GLV_HD constexpr int bar_batch_of(int log_nn) { return log_nn >= 13 ? GLV_BAR_BATCH_BIG : kBarBatch; }
struct BarTaps { float t[kBarTaps], w[kBarTaps]; };
// sub = lane index within the group (0..7).  A chunk reaches up to 63 floats past its bar's last tap (zero weights there:
// whatever is read is clamped to [0, 1], NaN -> 0, by bar_item_lane_sum before it meets its zero weight, so it adds
// exactly 0) -- never past the row: smooth_audio()'s taps end at 0.288 n (scale_audio(1) = -log(0.1) / 8), so
// first_bin + 64 * chunks <= n for every n >= 128 (checked when the tables are made, glv_tables.h bar_chunks_in_row).
// VEC (row in HBM): the lane's eight taps are two 16-byte loads at a 4-byte-aligned address (chunks start at arbitrary
// bins; gfx950 global loads need dword alignment only) -- eight dword loads per lane were eight times the L1 requests
// for the same lines.  !VEC (row in LDS): dword reads, paired by the compiler (ds_read2_b32).
struct alignas(16) BarW4 { float w[4]; };
struct __attribute__((packed, aligned(4))) BarT4 { float t[4]; };
template <bool VEC = true>
GLV_HD BarTaps bar_item_load(const float* tex_row, const float* tap_w, const BarItem& it, int sub) {
    BarTaps s;
    const uint32_t lane_byte = 4u * (uint32_t) kBarTaps * (uint32_t) sub;
#pragma unroll
    for (int h = 0; h < kBarTaps / 4; ++h) {
        const BarW4 w4 = ld<BarW4>(tap_w, it.w_byte + lane_byte + 16u * (uint32_t) h);   // chunks start on 64-float boundaries
#pragma unroll
        for (int i = 0; i < 4; ++i) s.w[4 * h + i] = w4.w[i];
    }
    const uint32_t base = it.tex_byte + lane_byte;
    if constexpr (VEC) {
#pragma unroll
        for (int h = 0; h < kBarTaps / 4; ++h) {
            const BarT4 t4 = *reinterpret_cast<const BarT4*>(reinterpret_cast<const char*>(tex_row) + base + 16u * (uint32_t) h);
#pragma unroll
            for (int i = 0; i < 4; ++i) s.t[4 * h + i] = t4.t[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < kBarTaps; ++i) s.t[i] = ld<float>(tex_row, base + 4u * (uint32_t) i);
    }
    return s;
}
GLV_HD float bar_item_lane_sum(const BarTaps& s) {
#if defined(__HIP_DEVICE_COMPILE__)
    // packed: clamp two taps (x * 1 with the clamp modifier: [0, 1], NaN -> 0), then one link of both chains
    const glv_f2 ones = {1.0f, 1.0f};
    glv_f2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < kBarTaps; i += 2) {
        glv_f2 t = {s.t[i], s.t[i + 1]};
        const glv_f2 w = {s.w[i], s.w[i + 1]};
        asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(t) : "v"(t), "s"(ones));          // (uniform constant: a scalar pair, not a v_mov_b64 per step)
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(acc) : "v"(t), "v"(w), "v"(acc));
    }
    return acc.x + acc.y;
#else
    float e = 0.0f, o = 0.0f;
    auto clamp01 = [](float x) { return x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f; };   // NaN -> 0
    for (int i = 0; i < kBarTaps; i += 2) {
        e = __builtin_fmaf(clamp01(s.t[i]), s.w[i], e);
        o = __builtin_fmaf(clamp01(s.t[i + 1]), s.w[i + 1], o);
    }
    return e + o;
#endif
}
#if defined(__HIPCC__)
// sum over each group of GL = 2 / 4 / 8 lanes (aligned inside a DPP row), result in every lane of the group: VALU-speed cross-lane
// adds (a ds_bpermute shuffle is an LDS round trip each)
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int GL> __device__ __forceinline__ float group_sum(float v) {
    static_assert(GL == 2 || GL == 4 || GL == 8, "lanes per group");
    v = v + dpp_move<0xB1>(v);                             // quad_perm [1,0,3,2]: neighbours
    if constexpr (GL >= 4) v = v + dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]: pairs of pairs
    if constexpr (GL >= 8) v = v + dpp_move<0x141>(v);     // row_half_mirror: the other quad of each 8
    return v;
}
#endif

// ---- addressing -------------------------------------------------------------------------------
// Every HBM/LDS access of a phase is `uniform base + 32-bit byte offset`, the offsets of the 16
// accesses differing only by compile-time constants.  That is the shape the gfx950 backend turns
// into  global_load ... v_off, s[base:base+1] offset:imm  /  ds_read ... v_off offset:imm  -- one
// offset VGPR per phase instead of sixteen 64-bit addresses kept alive across the frame loop.
// (ld<V>(base, byte_off) / st<V>(base, byte_off, v) live in glv_core.h)

// ring slot of age f (0 = oldest .. F-1 = newest = head)
GLV_HD uint32_t ring_slot(uint32_t head, uint32_t f, uint32_t F) {
    uint32_t s = head + 1 + f;
    return s >= F ? s - F : s;   // head < F, f < F  =>  s < 2F
}

// gravity (render.c:720-736) and average (render.c:738-771) for the float pair at byte offset
// `off` of channel row `row`; n = floats per row.  State traffic is 8 bytes per lane, lanes
// contiguous.  With both operators the newest ring slot doubles as the gravity state ("applied"
// of render.c:724 is by construction the previous gravity output, i.e. the previous newest slot).
//
// a.gl_storage (glv_params.gl_storage): the GL twin's storage model, render.c:2188-2265 -- every intermediate lives in a
// GL_R16 texture, so it is clamped to [0, 1] and quantised to 16 bits (through_r16) where a pass writes it: the uploaded
// buffer (render.c:521-524), the gravity store after GL_MAX and the in-place subtraction (:2199-2228), the ring copy
// (exact) and the average (:2230-2265, only when avg_frames > 1).  Consequences the float state machine does not have:
// gravity's fixed point under silence is 0 (not -g), nothing exceeds 1.
// ---- GL_R16 state, kept as texels (gl_storage == 1) ------------------------------------------------------------------------
// The pass structure of apply_state's gl_storage branch below on uint16 state: `tex` = the uploaded point as two packed texels
// (pack_unorm16 of the transform's output), `off` = the point's byte offset in an f32 row (its texel pair sits at off / 2).
// Returns the texel pair the chain ends with (gravity store, or the average when F > 1).
// One term of average_pass.frag's sum,  r += window(I, ...) * texelFetch(tI).r  -- in the shader's own arithmetic: GLSL `float`
// is 32-bit, the weight is a constant the GLSL compiler folded to a float, product and sum are each rounded to float (no
// contraction: what a non-fusing implementation such as Mesa's llvmpipe on x86-64 executes).  Rounds 2-3 had borrowed the CPU
// operator's  float * double -> double -> float  (render.c:759) for this pass too; against the reference's own llvmpipe texels
// the two are indistinguishable (profiles/r04/gl_average_models.txt: the same texels differ, the exact half-texel ties), the
// 32-bit form costs two packed instructions per complex point and frame instead of ten.
GLV_HD void weighted_add(cf& acc, cf f, float w, bool windowed) {
#if defined(__HIP_DEVICE_COMPILE__)
    const glv_f2 T = {f.x, f.y}, A = {acc.x, acc.y};
    glv_f2 s;
    if (windowed) {
        const glv_f2 W = {w, w};                               // uniform: a scalar register pair (as a vector operand it was rebuilt with a v_mov_b64 per use)
        glv_f2 p;
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(T), "s"(W));
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(A), "v"(p));
    } else asm("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(A), "v"(T));
    acc.x = s.x; acc.y = s.y;
#else
    if (windowed) { const float p0 = w * f.x, p1 = w * f.y; acc.x = acc.x + p0; acc.y = acc.y + p1; }
    else { acc.x = acc.x + f.x; acc.y = acc.y + f.y; }
#endif
}
GLV_HD void weighted_texels(cf& acc, uint32_t p, float w, bool windowed) { weighted_add(acc, texels_to_float(p), w, windowed); }
GLV_HD uint32_t apply_state_r16(uint32_t tex, uint32_t off, size_t row, uint32_t n, const FrameArgs& a) {
    const uint32_t F = a.F, o = off / 2u;
    const bool ring = (a.ops & OP_AVERAGE) != 0;
    uint16_t* h = ring ? reinterpret_cast<uint16_t*>(a.hist) + row * (size_t) F * n : nullptr;         // uniform
    if (a.ops & OP_GRAVITY) {
        const uint16_t* gs = ring ? h + (size_t) (F == 1 ? a.head : ring_slot(a.head, F - 2, F)) * n
                                  : reinterpret_cast<const uint16_t*>(a.grav) + row * (size_t) n;
        tex = gravity_r16(tex, ld<uint32_t>(gs, o), a.g, a.grav_sub, a.grav_int);
        if (!ring) st<uint32_t>(reinterpret_cast<uint16_t*>(a.grav_w) + row * (size_t) n, o, tex);
    }
    if (ring) {
        cf acc = { 0.0f, 0.0f };
        for (uint32_t f = 0; f + 1 < F; ++f)                               // oldest .. second newest
            weighted_texels(acc, ld<uint32_t>(h + (size_t) ring_slot(a.head, f, F) * n, o), a.wts32[f], a.avg_window != 0);
        st<uint32_t>(h + (size_t) a.head * n, o, tex);
        if (F > 1) {                                                       // render.c:2230: no averaging pass for one frame
            weighted_texels(acc, tex, a.wts32[F - 1], a.avg_window != 0);
            tex = pack_unorm16(div_frames(acc.x, a.F_as_float, a.F_rcp), div_frames(acc.y, a.F_as_float, a.F_rcp));
        }
    }
    return tex;
}

GLV_HD cf apply_state(cf val, uint32_t off, size_t row, uint32_t n, const FrameArgs& a) {
    if (a.gl_storage == 1) return texels_to_float(apply_state_r16(pack_unorm16(val.x, val.y), off, row, n, a));
    if (a.gl_storage) {
        val.x = through_r16(val.x); val.y = through_r16(val.y);              // the upload
        const uint32_t F = a.F;
        const bool ring = (a.ops & OP_AVERAGE) != 0;
        float* h = ring ? a.hist + row * (size_t) F * n : nullptr;            // uniform
        if (a.ops & OP_GRAVITY) {
            const float* gs = ring ? h + (size_t) (F == 1 ? a.head : ring_slot(a.head, F - 2, F)) * n : a.grav + row * (size_t) n;
            const cf st0 = ld<cf>(gs, off);                                   // the store texture == the previous newest ring slot
            val.x = through_r16(gravity(val.x, st0.x, a.g)); val.y = through_r16(gravity(val.y, st0.y, a.g));
            if (!ring) st<cf>(a.grav_w + row * (size_t) n, off, val);
        }
        if (ring) {
            cf acc = { 0.0f, 0.0f };
            for (uint32_t f = 0; f + 1 < F; ++f)                              // oldest .. second newest; the shader's 32-bit arithmetic
                weighted_add(acc, ld<cf>(h + (size_t) ring_slot(a.head, f, F) * n, off), a.wts32[f], a.avg_window != 0);
            st<cf>(h + (size_t) a.head * n, off, val);
            if (F > 1) {                                                      // render.c:2230: no averaging pass for one frame
                weighted_add(acc, val, a.wts32[F - 1], a.avg_window != 0);
                val.x = through_r16(div_frames(acc.x, a.F_as_float, a.F_rcp)); val.y = through_r16(div_frames(acc.y, a.F_as_float, a.F_rcp));
            }
        }
        return val;
    }
    if (a.ops & OP_AVERAGE) {
        float* h = a.hist + row * (size_t) a.F * n;                          // uniform
        const uint32_t F = a.F;
        cf acc = { 0.0f, 0.0f }, prev = { 0.0f, 0.0f };
        if (F == 1) prev = ld<cf>(h + (size_t) a.head * n, off);
        for (uint32_t f = 0; f + 1 < F; ++f) {                               // oldest .. second newest
            prev = ld<cf>(h + (size_t) ring_slot(a.head, f, F) * n, off);
            if (a.avg_window) {                                              // render.c:759, double product
                acc.x = (float) ((double) acc.x + a.wts[f] * (double) prev.x);
                acc.y = (float) ((double) acc.y + a.wts[f] * (double) prev.y);
            } else { acc.x = acc.x + prev.x; acc.y = acc.y + prev.y; }
        }
        if (a.ops & OP_GRAVITY) {
            val.x = gravity(val.x, prev.x, a.g); val.y = gravity(val.y, prev.y, a.g);
        }
        st<cf>(h + (size_t) a.head * n, off, val);
        if (a.avg_window) {
            acc.x = (float) ((double) acc.x + a.wts[F - 1] * (double) val.x);
            acc.y = (float) ((double) acc.y + a.wts[F - 1] * (double) val.y);
        } else { acc.x = acc.x + val.x; acc.y = acc.y + val.y; }
        val.x = acc.x / a.F_as_float;                                        // render.c:761
        val.y = acc.y / a.F_as_float;
    } else if (a.ops & OP_GRAVITY) {
        const cf st0 = ld<cf>(a.grav + row * (size_t) n, off);               // uniform base
        val.x = gravity(val.x, st0.x, a.g); val.y = gravity(val.y, st0.y, a.g);
        st<cf>(a.grav_w + row * (size_t) n, off, val);
    }
    return val;
}

// The same state machine for NV values of one lane at once, loads first: for every history slot the
// NV loads are issued back to back before any of them is consumed, so a lane keeps NV (not 1) HBM
// requests in flight.  (The one-value form above is load -> wait -> use per slot and per value: fine
// for glv_post_kernel's one-pair-per-lane grid, latency-bound inside the frame kernel's epilogue.)
// GLV_STATE_PAIR_MAX: largest log2(nn) whose stateful epilogue requests two history frames per trip (PAIR below);
// N=8192 spills 8-12 VGPRs for it and still gains (profiles/r02/ab_pair.txt), N=16384 would spill 50-60
#ifndef GLV_STATE_PAIR_MAX
#define GLV_STATE_PAIR_MAX 12
#endif
template <int NV, bool PAIR = false>
GLV_HD void apply_state_block(cf (&val)[NV], const uint32_t (&off)[NV], size_t row, uint32_t n, const FrameArgs& a) {
    if (a.ops & OP_AVERAGE) {
        float* h = a.hist + row * (size_t) a.F * n;                          // uniform
        const uint32_t F = a.F;
        cf acc[NV], prev[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) { acc[e].x = 0.0f; acc[e].y = 0.0f; prev[e].x = 0.0f; prev[e].y = 0.0f; }
        if (F == 1) {
#pragma unroll
            for (int e = 0; e < NV; ++e) prev[e] = ld<cf>(h + (size_t) a.head * n, off[e]);
        }
        // oldest .. second newest, two history frames per trip (PAIR): their 2 NV loads are in flight
        // together -- the epilogue is a chain of dependent HBM round trips (F - 1 per block), halving their number is
        // worth more than the NV extra registers; the accumulation order (render.c:757-760) is unchanged
        uint32_t f = 0;
        if constexpr (PAIR) for (; f + 2 < F; f += 2) {
            const float* hs0 = h + (size_t) ring_slot(a.head, f, F) * n;         // uniform
            const float* hs1 = h + (size_t) ring_slot(a.head, f + 1, F) * n;     // uniform
            cf p0[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) p0[e] = ld<cf>(hs0, off[e]);
#pragma unroll
            for (int e = 0; e < NV; ++e) prev[e] = ld<cf>(hs1, off[e]);
            const double w0 = a.wts[f], w1 = a.wts[f + 1];
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (a.avg_window) {                                          // render.c:759, double product
                    acc[e].x = (float) ((double) acc[e].x + w0 * (double) p0[e].x);
                    acc[e].y = (float) ((double) acc[e].y + w0 * (double) p0[e].y);
                    acc[e].x = (float) ((double) acc[e].x + w1 * (double) prev[e].x);
                    acc[e].y = (float) ((double) acc[e].y + w1 * (double) prev[e].y);
                } else {
                    acc[e].x = acc[e].x + p0[e].x; acc[e].y = acc[e].y + p0[e].y;
                    acc[e].x = acc[e].x + prev[e].x; acc[e].y = acc[e].y + prev[e].y;
                }
            }
        }
        for (; f + 1 < F; ++f) {
            const float* hs = h + (size_t) ring_slot(a.head, f, F) * n;      // uniform
#pragma unroll
            for (int e = 0; e < NV; ++e) prev[e] = ld<cf>(hs, off[e]);
            const double w = a.wts[f];
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (a.avg_window) {                                          // render.c:759, double product
                    acc[e].x = (float) ((double) acc[e].x + w * (double) prev[e].x);
                    acc[e].y = (float) ((double) acc[e].y + w * (double) prev[e].y);
                } else { acc[e].x = acc[e].x + prev[e].x; acc[e].y = acc[e].y + prev[e].y; }
            }
        }
        const double wl = a.wts[F - 1];
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            if (a.ops & OP_GRAVITY) { val[e].x = gravity(val[e].x, prev[e].x, a.g); val[e].y = gravity(val[e].y, prev[e].y, a.g); }
            st<cf>(h + (size_t) a.head * n, off[e], val[e]);
            if (a.avg_window) {
                acc[e].x = (float) ((double) acc[e].x + wl * (double) val[e].x);
                acc[e].y = (float) ((double) acc[e].y + wl * (double) val[e].y);
            } else { acc[e].x = acc[e].x + val[e].x; acc[e].y = acc[e].y + val[e].y; }
            val[e].x = acc[e].x / a.F_as_float;                              // render.c:761
            val[e].y = acc[e].y / a.F_as_float;
        }
    } else if (a.ops & OP_GRAVITY) {
        const float* gs = a.grav + row * (size_t) n;                         // uniform
        float* gw = a.grav_w + row * (size_t) n;
        cf st0[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) st0[e] = ld<cf>(gs, off[e]);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            val[e].x = gravity(val[e].x, st0[e].x, a.g); val[e].y = gravity(val[e].y, st0[e].y, a.g);
            st<cf>(gw, off[e], val[e]);
        }
    }
}

// gl_storage == 1 inside the frame kernel: apply_state_r16 for NV complex points of one lane at once, loads first (the shape of
// apply_state_block; same arithmetic, same order).  PAIRED: points j and j + 1 (j even) are adjacent in the row -- their four
// texels are one 8-byte access.  TWO: two history frames per trip.
// KPRE > 0 (the live class, glv_kernel_tmpl.h): the oldest `npre` <= KPRE ring slots of the lane's points were requested BEFORE the row's transform
// (gl16_state_prefetch, same addresses) and are in `pre` -- the HBM latency of the state ran under the FFT passes instead of in front of the
// average; the sum takes them in the same order, oldest first, so not a bit changes.  Only chains that average over F >= 2 frames prefetch.
constexpr int kLivePre = 4;
// TAIL / `limit` (the live classes): the last TAIL entries are the lane's points of the last live block of the row, which the bars sample only in part --
// a lane whose point lies at or beyond byte offset `limit` (of a float row) neither loads nor stores it (its arithmetic runs on whatever the register holds;
// nothing reads the result): the block's dead tail is not moved -- 1216 of the 1536 bins kept at n = 4096.
template <int NV, bool PAIRED, int TAIL = 0>
GLV_HD uint32_t gl16_state_prefetch(uint32_t (&pre)[kLivePre][NV], const uint32_t (&off)[NV], size_t row, uint32_t n, const FrameArgs& a, uint32_t limit = 0xffffffffu) {
    const uint32_t F = a.F;
    if (!(a.ops & OP_AVERAGE) || F < 2u) return 0u;
    const uint32_t npre = F - 1u < (uint32_t) kLivePre ? F - 1u : (uint32_t) kLivePre;
    const uint16_t* h = reinterpret_cast<const uint16_t*>(a.hist) + row * (size_t) F * n;
    uint32_t lb = off[0] / 2u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lb));
#endif
#pragma unroll
    for (int k = 0; k < kLivePre; ++k) {
        if ((uint32_t) k >= npre) break;
        const uint16_t* base = h + (size_t) ring_slot(a.head, (uint32_t) k, F) * n;
        if constexpr (PAIRED) {
#pragma unroll
            for (int e = 0; e < NV; e += 2) {
                if (e >= NV - TAIL && off[e] >= limit) { pre[k][e] = 0u; pre[k][e + 1] = 0u; continue; }
                const u32x2 t = ld<u32x2>(base, lb + (off[e] - off[0]) / 2u); pre[k][e] = t.x; pre[k][e + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (e >= NV - TAIL && off[e] >= limit) { pre[k][e] = 0u; continue; }
                pre[k][e] = ld<uint32_t>(base, lb + (off[e] - off[0]) / 2u);
            }
        }
    }
    return npre;
}
template <int NV, bool PAIRED, bool TWO, int KPRE = 0, int TAIL = 0>
GLV_HD void gl16_state_block(uint32_t (&tex)[NV], const uint32_t (&off)[NV], size_t row, uint32_t n, const FrameArgs& a,
                             const uint32_t (*pre)[NV] = nullptr, uint32_t npre = 0, uint32_t limit = 0xffffffffu) {
    // Every access is `uniform base + lane offset + compile-time constant` (off[e] - off[0] is a constant: the points of a lane are a fixed
    // pattern).  The lane offset is re-defined opaquely per call: inside the history loop the backend otherwise hoists its zero-extension out of
    // the loop and then forms a 64-bit address per access and trip with v_lshl_add_u64 (two registers each) instead of the
    // `global_load v_off, s[base:base+1] offset:imm` form -- one 32-bit offset register for all of a slot's accesses.
    auto lane_base = [&]() -> uint32_t {
        uint32_t lb = off[0] / 2u;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(lb));
#endif
        return lb;
    };
    auto load = [&](uint32_t (&dst)[NV], const uint16_t* base) {
        const uint32_t lb = lane_base();
        if constexpr (PAIRED) {
#pragma unroll
            for (int e = 0; e < NV; e += 2) {
                if (e >= NV - TAIL && off[e] >= limit) { dst[e] = 0u; dst[e + 1] = 0u; continue; }
                const u32x2 t = ld<u32x2>(base, lb + (off[e] - off[0]) / 2u); dst[e] = t.x; dst[e + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (e >= NV - TAIL && off[e] >= limit) { dst[e] = 0u; continue; }
                dst[e] = ld<uint32_t>(base, lb + (off[e] - off[0]) / 2u);
            }
        }
    };
    auto store = [&](uint16_t* base, const uint32_t (&src)[NV]) {
        const uint32_t lb = lane_base();
        if constexpr (PAIRED) {
#pragma unroll
            for (int e = 0; e < NV; e += 2) {
                if (e >= NV - TAIL && off[e] >= limit) continue;
                st<u32x2>(base, lb + (off[e] - off[0]) / 2u, u32x2{src[e], src[e + 1]});
            }
        } else {
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                if (e >= NV - TAIL && off[e] >= limit) continue;
                st<uint32_t>(base, lb + (off[e] - off[0]) / 2u, src[e]);
            }
        }
    };
    const uint32_t F = a.F;
    // Uniform choices -- window or plain sum, the gravity step as an integer subtraction or in floats -- are made ONCE around the loops over the
    // lane's points, not inside weighted_texels / gravity_r16 for every point: in the unrolled code each inner test was a pair of scalar
    // branches of its own (the backend does not merge identical tests across the inline assembly between them), 12 pairs per point at
    // F = 5, ~190 per lane and row (profiles/r05/uniform_branches.txt).  A plain sum is the windowed form with weight 1 (1 * t == t, bit for bit).
    const bool windowed = a.avg_window != 0;
    auto gravity_all = [&](const uint32_t (&st)[NV]) {
        if (a.grav_int) {
#pragma unroll
            for (int e = 0; e < NV; ++e) tex[e] = gravity_r16_int(tex[e], st[e], a.grav_sub);
        } else {
#pragma unroll
            for (int e = 0; e < NV; ++e) tex[e] = gravity_r16_flt(tex[e], st[e], a.g);
        }
    };
    if (a.ops & OP_AVERAGE) {
        uint16_t* h = reinterpret_cast<uint16_t*>(a.hist) + row * (size_t) F * n;                      // uniform
        cf acc[NV];
        uint32_t prev[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) { acc[e].x = 0.0f; acc[e].y = 0.0f; prev[e] = 0u; }
        if (F == 1) load(prev, h + (size_t) a.head * n);                     // the gravity store of a one-frame ring is the slot itself
        uint32_t f = 0;
        if constexpr (KPRE > 0) {                                            // the prefetched slots, oldest first (uniform tests: npre, F)
#pragma unroll
            for (int k = 0; k < KPRE; ++k) {
                if ((uint32_t) k >= npre) break;
                const float w = windowed ? a.wts32[k] : 1.0f;
#pragma unroll
                for (int e = 0; e < NV; ++e) weighted_texels(acc[e], pre[k][e], w, true);
                if ((uint32_t) k + 2u == F) {                                // slot F - 2: the previous newest == the gravity store
#pragma unroll
                    for (int e = 0; e < NV; ++e) prev[e] = pre[k][e];
                }
            }
            f = npre;
        }
        if constexpr (TWO && KPRE == 0) for (; f + 2 < F; f += 2) {          // oldest .. second newest, two frames per trip
            uint32_t p0[NV];
            load(p0, h + (size_t) ring_slot(a.head, f, F) * n);
            load(prev, h + (size_t) ring_slot(a.head, f + 1, F) * n);
            const float w0 = windowed ? a.wts32[f] : 1.0f, w1 = windowed ? a.wts32[f + 1] : 1.0f;
#pragma unroll
            for (int e = 0; e < NV; ++e) { weighted_texels(acc[e], p0[e], w0, true); weighted_texels(acc[e], prev[e], w1, true); }
        }
        for (; f + 1 < F; ++f) {
            load(prev, h + (size_t) ring_slot(a.head, f, F) * n);
            const float w = windowed ? a.wts32[f] : 1.0f;
#pragma unroll
            for (int e = 0; e < NV; ++e) weighted_texels(acc[e], prev[e], w, true);
        }
        // prev == the previous newest slot == the gravity store (the same texels: render.c:2232-2243 copies it into the ring)
        if (a.ops & OP_GRAVITY) gravity_all(prev);
        store(h + (size_t) a.head * n, tex);
        if (F > 1) {                                                         // render.c:2230: no averaging pass for one frame
            const float wl = windowed ? a.wts32[F - 1] : 1.0f;
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                weighted_texels(acc[e], tex[e], wl, true);
                tex[e] = pack_unorm16(div_frames(acc[e].x, a.F_as_float, a.F_rcp), div_frames(acc[e].y, a.F_as_float, a.F_rcp));
            }
        }
    } else if (a.ops & OP_GRAVITY) {
        uint32_t st0[NV];
        load(st0, reinterpret_cast<const uint16_t*>(a.grav) + row * (size_t) n);
        gravity_all(st0);
        store(reinterpret_cast<uint16_t*>(a.grav_w) + row * (size_t) n, tex);
    }
}

template <int LOG_NN, int LOG_E = 4>
struct Frame {
    using PL = Plan<LOG_NN, LOG_E>;
    static constexpr int NN = PL::NN, N = 2 * NN, T = PL::T, P = PL::P, E = PL::E;
    // complex points of one LDS exchange region: the pass-0 exchange is padded by one point per E
    static constexpr int XREGION = NN + NN / E;

    // ---- input: v[i] <- windowed sample pair (complex point) c = i*T + tid ---------------------
    // s16: one 8-byte load holds complex point c of BOTH channels: (L[2c], R[2c], L[2c+1], R[2c+1]).
    // A slot transforms ONE channel row per iteration; the two channels of a frame are handled by
    // neighbouring slots of the same workgroup at the same time, so the second reader of a frame's
    // PCM hits the CU's L1 / the XCD's L2 and HBM still sees every PCM byte once -- while no lane has
    // to park the other channel's samples in registers across a whole transform.
    struct Raw { uint32_t x[E], y[E]; };     // x = L[2c] | R[2c] << 16,  y = L[2c+1] | R[2c+1] << 16

    // RING: rotate the read position for the FIFO ring mode -- fifo.c:91-92 keeps the newest samples at the end of
    // the buffer by memmove; the device ring is circular instead and `rot` is the index of its OLDEST frame.  fifo.c
    // accepts any sample_sz (fifo.c:38,81,91), so rot may be odd: complex point c is then the frame pair
    // (rot + 2c, rot + 2c + 1) mod n, which is neither 8-byte aligned nor contiguous across the wrap -- two dword loads;
    // with rot even (every sample_sz that is a multiple of 8 bytes) the pair is one aligned 8-byte load.  Uniform branch.
    template <bool RING>
    GLV_HD static void load_pcm(Raw& p, const void* frame, int tid, uint32_t rot) {
        if constexpr (RING) {
            if (rot & 1u) {
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const uint32_t f0 = (uint32_t) (2 * (i * T + tid)) + rot;
                    p.x[i] = ld<uint32_t>(frame, (f0 & (uint32_t) (N - 1)) * 4u);
                    p.y[i] = ld<uint32_t>(frame, ((f0 + 1u) & (uint32_t) (N - 1)) * 4u);
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < E; ++i) {
            uint32_t off;
            if constexpr (RING) off = ((uint32_t) (i * T + tid + (rot >> 1)) & (uint32_t) (NN - 1)) * 8u;
            else off = (uint32_t) tid * 8u + (uint32_t) (i * T) * 8u;
            const u32x2 u = ld<u32x2>(frame, off);
            p.x[i] = u.x; p.y[i] = u.y;
        }
    }
    // one sample of channel `ch` (0 = left/low half, 1 = right/high half) -- fifo.c:105-106;
    // mono: fifo.c:98-102, ((L + R) / 2) with C int division, for both channels.
    GLV_HD static float sample(uint32_t packed, uint32_t ch_shift, bool mono) {
        if (mono) return unpack_s16(((int) (int16_t) (packed & 0xffffu) + (int) (int16_t) (packed >> 16)) / 2);
        return unpack_s16((int) (int16_t) ((packed >> ch_shift) & 0xffffu));
    }
    // WCHUNK window pairs are fetched per scheduling fence: two chunks in flight hide the L2/LDS
    // latency of the table while keeping the transient footprint at 2*WCHUNK*4 VGPRs.
    static constexpr int WCHUNK = E < 4 ? E : 4;
    // WPRE: the window values of the lane's first WPRE points were fetched ahead (window_prefetch).  Kernels whose
    // window table is not in LDS (N=16384) issue these loads BEFORE the epilogue's spectrum stores: vmcnt retires
    // in order, so a table load issued after the stores cannot be consumed until every store has drained, while
    // one issued before them only waits for itself (and has the whole magnitude stage to arrive).
    template <int WPRE> struct WinPre { d2 w[WPRE > 0 ? WPRE : 1]; };
    template <int WPRE>
    GLV_HD static void window_prefetch(WinPre<WPRE>& wp, const void* win, int tid) {
#pragma unroll
        for (int j = 0; j < WPRE; ++j) wp.w[j] = ld<d2>(win, (uint32_t) tid * 16u + (uint32_t) (j * T) * 16u);
    }
    // one complex point: samples 2c, 2c + 1 of the channel, unpacked (fifo.c:105-106) and windowed (render.c:794).
    // SPLIT: `wv` holds the point's WinSplit (glv_core.h) instead of two doubles; on the device the pair goes through four
    // packed instructions (quotient by 65535: glv_core.h div_65535; window: apply_window_split).
    template <bool MONO, bool SPLIT>
    GLV_HD static void unpack_point(cf& out, uint32_t px, uint32_t py, uint32_t ch_shift, const d2& wv) {
        if constexpr (SPLIT) {
            const WinSplit q = __builtin_bit_cast(WinSplit, wv);
#if defined(__HIP_DEVICE_COMPILE__)
            glv_f2 K;
            if constexpr (MONO) {
                K.x = (float) (((int) (int16_t) (px & 0xffffu) + (int) (int16_t) (px >> 16)) / 2);
                K.y = (float) (((int) (int16_t) (py & 0xffffu) + (int) (int16_t) (py >> 16)) / 2);
            } else {
                K.x = (float) (int) (int16_t) ((px >> ch_shift) & 0xffffu);
                K.y = (float) (int) (int16_t) ((py >> ch_shift) & 0xffffu);
            }
            const glv_f2 CH = {0x1.0001p-16f, 0x1.0001p-16f}, CL = {0x1.0001p-48f, 0x1.0001p-48f};
            const glv_f2 HI = {q.hi0, q.hi1}, LO = {q.lo0, q.lo1};
            glv_f2 t, S, X;
            asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(K), "v"(CL));
            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(S) : "v"(K), "v"(CH), "v"(t));
            asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(S), "v"(LO));
            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(X) : "v"(S), "v"(HI), "v"(t));
            out.x = X.x; out.y = X.y;
#else
            out.x = apply_window_split(sample(px, ch_shift, MONO), q.hi0, q.lo0);
            out.y = apply_window_split(sample(py, ch_shift, MONO), q.hi1, q.lo1);
#endif
        } else {
            out.x = apply_window(sample(px, ch_shift, MONO), wv.x);   // render.c:794
            out.y = apply_window(sample(py, ch_shift, MONO), wv.y);
        }
    }
    template <bool MONO, int WPRE = 0, bool SPLIT = false>
    GLV_HD static void unpack_window_impl(cf (&v)[E], const Raw& p, const void* win, int tid, uint32_t ch_shift, const d2* wpre = nullptr) {
        static_assert(WPRE % WCHUNK == 0 && WPRE <= E, "WPRE: whole chunks");
        d2 w[2][WCHUNK];
        if constexpr (WPRE < E) {
#pragma unroll
            for (int j = 0; j < WCHUNK; ++j) w[0][j] = ld<d2>(win, (uint32_t) tid * 16u + (uint32_t) ((WPRE + j) * T) * 16u);
        }
        if constexpr (WPRE > 0) {
            GLV_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < WPRE; ++i) unpack_point<MONO, SPLIT>(v[i], p.x[i], p.y[i], ch_shift, wpre[i]);
        }
#pragma unroll
        for (int c0 = WPRE; c0 < E; c0 += WCHUNK) {
            const int cur = ((c0 - WPRE) / WCHUNK) & 1;
            if (c0 + WCHUNK < E) {
#pragma unroll
                for (int j = 0; j < WCHUNK; ++j)
                    w[cur ^ 1][j] = ld<d2>(win, (uint32_t) tid * 16u + (uint32_t) ((c0 + WCHUNK + j) * T) * 16u);
            }
            GLV_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < WCHUNK; ++j) {
                const int i = c0 + j;
                unpack_point<MONO, SPLIT>(v[i], p.x[i], p.y[i], ch_shift, w[cur][j]);
            }
            GLV_SCHED_FENCE();
        }
    }
    // three code versions (mono, left, right): with a compile-time shift the channel select becomes the
    // operand selector of the conversion (v_cvt_f32_i32_sdwa WORD_0 / WORD_1) instead of a shift per sample
    template <int WPRE = 0, bool SPLIT = false>
    GLV_HD static void unpack_window(cf (&v)[E], const Raw& p, const void* win, int tid, uint32_t ch, bool mono, const d2* wpre = nullptr) {
        if (mono) unpack_window_impl<true, WPRE, SPLIT>(v, p, win, tid, 0, wpre);
        else if (ch) unpack_window_impl<false, WPRE, SPLIT>(v, p, win, tid, 16u, wpre);
        else unpack_window_impl<false, WPRE, SPLIT>(v, p, win, tid, 0u, wpre);
    }
    GLV_HD static void load_f32_window(cf (&v)[E], const void* row, const void* win, int tid) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const cf u = ld<cf>(row, (uint32_t) tid * 8u + (uint32_t) (i * T) * 8u);
            const d2 w = ld<d2>(win, (uint32_t) tid * 16u + (uint32_t) (i * T) * 16u);
            v[i].x = apply_window(u.x, w.x);
            v[i].y = apply_window(u.y, w.y);
        }
    }

    // ---- pipelined f32 inputs: samples are fetched into registers one row / frame ahead (kernel
    // PREFETCH 1) and windowed from there
    struct RawF { cf p[E]; };                                   // planar: complex point i*T + tid of the row
    GLV_HD static void load_f32_raw(RawF& r, const void* row, int tid) {
#pragma unroll
        for (int i = 0; i < E; ++i) r.p[i] = ld<cf>(row, (uint32_t) tid * 8u + (uint32_t) (i * T) * 8u);
    }
    // window table reads in chunks of WCHUNK like unpack_window_impl
    template <typename GET>
    GLV_HD static void window_points(cf (&v)[E], const void* win, int tid, GET get) {
        d2 w[2][WCHUNK];
#pragma unroll
        for (int j = 0; j < WCHUNK; ++j) w[0][j] = ld<d2>(win, (uint32_t) tid * 16u + (uint32_t) (j * T) * 16u);
#pragma unroll
        for (int c0 = 0; c0 < E; c0 += WCHUNK) {
            const int cur = (c0 / WCHUNK) & 1;
            if (c0 + WCHUNK < E) {
#pragma unroll
                for (int j = 0; j < WCHUNK; ++j)
                    w[cur ^ 1][j] = ld<d2>(win, (uint32_t) tid * 16u + (uint32_t) ((c0 + WCHUNK + j) * T) * 16u);
            }
            GLV_SCHED_FENCE();
#pragma unroll
            for (int j = 0; j < WCHUNK; ++j) {
                const int i = c0 + j;
                const cf u = get(i);
                v[i].x = apply_window(u.x, w[cur][j].x);
                v[i].y = apply_window(u.y, w[cur][j].y);
            }
            GLV_SCHED_FENCE();
        }
    }
    GLV_HD static void window_f32_raw(cf (&v)[E], const RawF& r, const void* win, int tid) {
        window_points(v, win, tid, [&](int i) { return r.p[i]; });
    }

    // interleaved stereo f32 (pulse_input.c:159-176): one 16-byte load = complex point c of both
    // channels (L[2c], R[2c], L[2c+1], R[2c+1]); mono = (L + R) / 2 in float (pulse_input.c:167)
    struct alignas(16) f4 { float a, b, c, d; };
    // RING: the frame is a circular buffer whose oldest complex point sits at `rot` (glv_batch_ring_update_f32)
    template <bool RING = false>
    GLV_HD static void load_f32_stereo_window(cf (&v)[E], const void* frame, const void* win, int tid, uint32_t ch, bool mono,
                                              uint32_t rot = 0) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            f4 u;
            if constexpr (RING) {      // frames (rot + 2c, rot + 2c + 1) mod n, 8 bytes each (rot = oldest frame of the ring, any parity)
                const uint32_t f0 = (uint32_t) (2 * (i * T + tid)) + rot;
                const cf lo = ld<cf>(frame, (f0 & (uint32_t) (N - 1)) * 8u), hi = ld<cf>(frame, ((f0 + 1u) & (uint32_t) (N - 1)) * 8u);
                u.a = lo.x; u.b = lo.y; u.c = hi.x; u.d = hi.y;
            } else u = ld<f4>(frame, (uint32_t) tid * 16u + (uint32_t) (i * T) * 16u);
            const d2 w = ld<d2>(win, (uint32_t) tid * 16u + (uint32_t) (i * T) * 16u);
            float s0, s1;
            if (mono) { s0 = (u.a + u.b) / 2; s1 = (u.c + u.d) / 2; }
            else { s0 = ch ? u.b : u.a; s1 = ch ? u.d : u.c; }
            v[i].x = apply_window(s0, w.x);
            v[i].y = apply_window(s1, w.y);
        }
    }

    // interleaved stereo f32, one channel of complex point i*T + tid: floats 4c + ch and 4c + 2 + ch of the
    // frame.  Two dword loads per point keep the footprint at E registers pairs (a 16-byte load would park
    // the other channel in registers across a whole transform: 64 VGPRs more than the kernel has); the
    // other halves of the same cache lines are read by the channel's sibling row one transform later.
    template <bool RING = false>
    GLV_HD static void load_f32s_raw(RawF& r, const void* frame, int tid, uint32_t ch, uint32_t rot = 0) {
        const char* base = static_cast<const char*>(frame) + ch * 4u;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            if constexpr (RING) {      // rot = oldest frame of the ring (any parity, pulse_input.c accepts any sample_sz)
                const uint32_t f0 = (uint32_t) (2 * (i * T + tid)) + rot;
                r.p[i].x = ld<float>(base, (f0 & (uint32_t) (N - 1)) * 8u);
                r.p[i].y = ld<float>(base, ((f0 + 1u) & (uint32_t) (N - 1)) * 8u);
            } else {
                const uint32_t poff = (uint32_t) tid * 16u + (uint32_t) (i * T) * 16u;
                r.p[i].x = ld<float>(base, poff);
                r.p[i].y = ld<float>(base, poff + 8u);
            }
        }
    }

    // ---- twiddle gather for pass PASS (per thread; invariant across frames) ---------------------
    template <int PASS>
    struct PassInfo {
        static constexpr int RB = PL::rb(PASS), R = 1 << RB, NG = E >> RB;
        static constexpr int LOG_L0 = PL::log_l0(PASS), L0 = 1 << LOG_L0;
        static constexpr int NTW = NG * (R - 1);
    };

    // Which of the nn/R radix-R groups of pass PASS does lane `tid` take as its gi-th group?
    // Intermediate passes: G = gi*T + tid (lanes contiguous: conflict-free LDS, one base address).
    // Last pass: the lane's groups come in ADJACENT pairs (G = 2*tid + {0, 1} [+ 2T per further pair]), so its outputs come
    // in runs of two consecutive complex points: 16-byte ds_read_b128 / global_store_dwordx4 instead of 8-byte ones
    // (the spectrum store is issue-bound, not bandwidth-bound: half the instructions, half the time), and consecutive lanes
    // hold consecutive 16-byte pieces (whole cache lines per wave instruction).
    // (With ONE group per lane in the last pass -- N=1024 at E=8, N=8192 at E=16 -- the stores are 8-byte ones: trading registers between
    // lanes 16 apart for 16-byte stores measured no gain, profiles/r02/sweep_3.txt; the variant is in profiles/r05/removed_experiment_scaffolding.diff.)
    template <int PASS>
    GLV_HD static constexpr int group_of(int tid, int gi) {
        // last pass: the lane's groups come in adjacent PAIRS (one 16-byte access per pair), pair p of lane t at 2*(p*T + t):
        // the 64 lanes of a store (or state load) instruction then cover 1 KiB CONTIGUOUS bytes.  With all NG groups of a lane
        // adjacent (round 2) a lane's 16-byte pieces were 8*NG bytes apart: at NG = 4 (N=16384) every wave store wrote
        // half of each 128-byte line it touched and a second instruction came back for the other half.
        if (PASS == P - 1 && PassInfo<PASS>::NG >= 2) return (gi >> 1) * (2 * T) + 2 * tid + (gi & 1);
        return PASS == P - 1 ? tid * PassInfo<PASS>::NG + gi : gi * T + tid;
    }

    // BIAS: `table` points at entry BIAS of the twiddle table (the LDS copy of the middle passes' range)
    template <int PASS, int BIAS = 0>
    GLV_HD static void gather_tw(cf (&tw)[PassInfo<PASS>::NTW], const cf* table, int tid) {
        using PI = PassInfo<PASS>;
        // last pass with paired groups (group_of): groups gi, gi + 1 are adjacent, so are their twiddles -- W[L][k0 + ...] and
        // W[L][k0 + 1 + ...], k0 even, every stage's table starting on an even entry (tw_offset) -- one 16-byte load per pair
        constexpr bool PAIRED = PASS == P - 1 && PI::NG >= 2 && (BIAS % 2) == 0 && group_of<PASS>(0, 1) == group_of<PASS>(0, 0) + 1;
#pragma unroll
        for (int gi = 0; gi < PI::NG; gi += (PAIRED ? 2 : 1)) {
            const int G = group_of<PASS>(tid, gi);
            const int k0 = G & (PI::L0 - 1);
#pragma unroll
            for (int s = 0; s < PI::RB; ++s)
#pragma unroll
                for (int ks = 0; ks < (1 << s); ++ks) {
                    if constexpr (PAIRED) {
                        const cf2 two = ld<cf2>(table, (uint32_t) (SubPass<PI::RB>::tw_index(PI::L0, k0, s, ks) - BIAS) * 8u);
                        tw[gi * (PI::R - 1) + (1 << s) - 1 + ks] = two.a;
                        tw[(gi + 1) * (PI::R - 1) + (1 << s) - 1 + ks] = two.b;
                    } else tw[gi * (PI::R - 1) + (1 << s) - 1 + ks] = table[SubPass<PI::RB>::tw_index(PI::L0, k0, s, ks) - BIAS];
                }
        }
    }

    // ---- compute pass PASS in registers -----------------------------------------------------------
    template <int PASS, bool UNIT_SHORTCUT = true>
    GLV_HD static void compute(cf (&v)[E], const cf (&tw)[PassInfo<PASS>::NTW]) {
        using PI = PassInfo<PASS>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi) {
            cf(&vg)[PI::R] = *reinterpret_cast<cf(*)[PI::R]>(&v[gi * PI::R]);
            const cf(&tg)[PI::R > 1 ? PI::R - 1 : 1] =
                *reinterpret_cast<const cf(*)[PI::R > 1 ? PI::R - 1 : 1]>(&tw[gi * (PI::R - 1)]);
            SubPass<PI::RB>::template run<PASS == 0, UNIT_SHORTCUT>(vg, tg);
        }
    }

    // element index (within the nn-point array after pass PASS) held in register slot (gi, r)
    template <int PASS>
    GLV_HD static constexpr int out_index(int tid, int gi, int r) {
        using PI = PassInfo<PASS>;
        const int G = group_of<PASS>(tid, gi);
        if (PASS == P - 1) return bitrev(r, PI::RB) * PI::L0 + G;        // R*L0 == nn: jt == 0, k0 == G
        const int k0 = G & (PI::L0 - 1);
        const int jt = G >> PI::LOG_L0;
        return jt * (PI::R * PI::L0) + bitrev(r, PI::RB) * PI::L0 + k0;
    }
    // element index that register slot (gi, i) of pass PASS must be fed with
    template <int PASS>
    GLV_HD static constexpr int in_index(int tid, int gi, int i) {
        using PI = PassInfo<PASS>;
        return i * (NN / PI::R) + group_of<PASS>(tid, gi);
    }

    // LDS byte offset of element q in the exchange after pass PASS (see glv::lds_index)
    template <int PASS>
    GLV_HD static void exchange_write(void* xbuf, const cf (&v)[E], int tid) {
        using PI = PassInfo<PASS>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi)
#pragma unroll
            for (int r = 0; r < PI::R; ++r)
                st<cf>(xbuf, (uint32_t) lds_index(PASS, out_index<PASS>(tid, gi, r), LOG_E) * 8u, v[gi * PI::R + r]);
    }
    // read the inputs of pass PASS from the exchange written after pass PASS-1; in the last pass a
    // lane's groups are adjacent (group_of), so two points come with one 16-byte read
    template <int PASS>
    GLV_HD static void exchange_read(cf (&v)[E], const void* xbuf, int tid) {
        using PI = PassInfo<PASS>;
        if constexpr (PASS == P - 1 && PI::NG >= 2 && PASS - 1 != 0) {
#pragma unroll
            for (int i = 0; i < PI::R; ++i)
#pragma unroll
                for (int gi = 0; gi < PI::NG; gi += 2) {
                    const cf2 two = ld<cf2>(xbuf, (uint32_t) in_index<PASS>(tid, gi, i) * 8u);
                    v[gi * PI::R + i] = two.a;
                    v[(gi + 1) * PI::R + i] = two.b;
                }
        } else if constexpr (PASS - 1 == 0 && (NN / PI::R) % E == 0) {
            // reading the padded pass-0 layout: pad(q) = q + q/E is additive when one term is a multiple
            // of E, and i * nn/R is -- so the address is pad(group) + a compile-time constant (written as
            // pad(i * nn/R + group) the backend re-derives the padding with three integer ops per point)
#pragma unroll
            for (int gi = 0; gi < PI::NG; ++gi) {
                const uint32_t base = (uint32_t) lds_index(0, group_of<PASS>(tid, gi), LOG_E) * 8u;
#pragma unroll
                for (int i = 0; i < PI::R; ++i)
                    v[gi * PI::R + i] = ld<cf>(xbuf, base + (uint32_t) lds_index(0, i * (NN / PI::R), LOG_E) * 8u);
            }
        } else {
#pragma unroll
            for (int gi = 0; gi < PI::NG; ++gi)
#pragma unroll
                for (int i = 0; i < PI::R; ++i)
                    v[gi * PI::R + i] = ld<cf>(xbuf, (uint32_t) lds_index(PASS - 1, in_index<PASS>(tid, gi, i), LOG_E) * 8u);
        }
    }

    // (The last exchange of N=1024 as wavefront shuffles -- v_permlane32_swap / v_permlane16_swap / DPP row_ror, 40 vector instructions in
    // place of 16 LDS accesses -- gave the same bits and no speed-up: profiles/r04/shuffle_ab.txt; code in profiles/r05/removed_experiment_scaffolding.diff.
    // Lane-crossing instructions stay where they pay: the DPP reductions of the bars, glv_frame.h group_sum.)

    // ---- split exchange (kernel knob NBUF = 0): the row crosses LDS one float component at a time -- all real parts
    // (write, barrier, read), then all imaginary parts -- so the exchange region is XREGION floats instead of XREGION complex
    // points: half the LDS per row in flight, at the price of two more barriers per exchange and 4-byte LDS accesses.  The
    // index maps are those of exchange_write / exchange_read; no register pressure is added (a lane holds E old imaginary and
    // E new real parts in the middle).  H = 0: .x, H = 1: .y
    template <int PASS, int H>
    GLV_HD static void exchange_write_half(void* xbuf, const cf (&v)[E], int tid) {
        using PI = PassInfo<PASS>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi)
#pragma unroll
            for (int r = 0; r < PI::R; ++r)
                st<float>(xbuf, (uint32_t) lds_index(PASS, out_index<PASS>(tid, gi, r), LOG_E) * 4u, H == 0 ? v[gi * PI::R + r].x : v[gi * PI::R + r].y);
    }
    template <int PASS, int H>
    GLV_HD static void exchange_read_half(cf (&v)[E], const void* xbuf, int tid) {
        using PI = PassInfo<PASS>;
        auto put = [&](int slot, float f) { if (H == 0) v[slot].x = f; else v[slot].y = f; };
        if constexpr (PASS == P - 1 && PI::NG >= 2 && PASS - 1 != 0) {
            struct alignas(8) f2 { float a, b; };
#pragma unroll
            for (int i = 0; i < PI::R; ++i)
#pragma unroll
                for (int gi = 0; gi < PI::NG; gi += 2) {                      // adjacent groups: one 8-byte read
                    const f2 two = ld<f2>(xbuf, (uint32_t) in_index<PASS>(tid, gi, i) * 4u);
                    put(gi * PI::R + i, two.a);
                    put((gi + 1) * PI::R + i, two.b);
                }
        } else if constexpr (PASS - 1 == 0 && (NN / PI::R) % E == 0) {
#pragma unroll
            for (int gi = 0; gi < PI::NG; ++gi) {
                const uint32_t base = (uint32_t) lds_index(0, group_of<PASS>(tid, gi), LOG_E) * 4u;
#pragma unroll
                for (int i = 0; i < PI::R; ++i)
                    put(gi * PI::R + i, ld<float>(xbuf, base + (uint32_t) lds_index(0, i * (NN / PI::R), LOG_E) * 4u));
            }
        } else {
#pragma unroll
            for (int gi = 0; gi < PI::NG; ++gi)
#pragma unroll
                for (int i = 0; i < PI::R; ++i)
                    put(gi * PI::R + i, ld<float>(xbuf, (uint32_t) lds_index(PASS - 1, in_index<PASS>(tid, gi, i), LOG_E) * 4u));
        }
    }

    // ---- epilogue: registers of the last pass -> HBM ------------------------------------------------
    // One complex point = floats n0 (even) and n0+1 of channel row `row`; 8-byte stores, lanes
    // contiguous.  EPI selects the operator chain at compile time (the kernel switches on a.ops
    // once per frame, outside the unrolled element loop):
    //   EPI_RAW  raw FFT output              EPI_MAG   abs/log/tilt
    //   EPI_MAG_STATE  abs/log/tilt followed by gravity and/or average (apply_state)
    // per-lane tilt factors of the E output points (loop invariant: a slot always owns the same bins)
    GLV_HD static void gather_tilt(cf (&tl)[E], const float* tilt, int tid) {
        using PI = PassInfo<P - 1>;
#pragma unroll
        for (int gi = 0; gi < PI::NG; ++gi)
#pragma unroll
            for (int r = 0; r < PI::R; ++r)
                tl[gi * PI::R + r] = ld<cf>(tilt, (uint32_t) out_index<P - 1>(tid, gi, r) * 8u);
    }

    // ---- output pairs ---------------------------------------------------------------------------------------------
    // The epilogue leaves the row in 16-byte pieces where the last pass gives a lane two or more groups: the same register slot of
    // neighbouring groups is two memory-adjacent complex points.
    static constexpr bool PAIRS = (E >> PL::rb(P - 1)) >= 2;
    // byte offset (f32 rows) of the lane's pair p = 0 .. E/2-1
    GLV_HD static uint32_t pair_offset(int tid, int p) {
        using PI = PassInfo<P - 1>;
        {
            const int idx = 2 * p;
            return (uint32_t) out_index<P - 1>(tid, idx % PI::NG, idx / PI::NG) * 8u;
        }
    }

    // TILTREG 0: tilt factors read from the table per row; 1: from `tl_reg` (registers, gathered once per
    // kernel); 2: evaluated in registers with the reference's float operations (no memory at all); 3: with log_mode 1 the
    // folded factor from two fused multiply-adds (glv_core.h tilt_lin; tl_reg[0].x carries the lane's base term), else as 2
    // R16: the output row is uint16 [n] (GL_R16 texels, glv_core.h unorm16) instead of float [n]; state stays f32
    // NONFINITE: the row may hold Inf / NaN (f32 input): log_mode 0 then needs log_third_nf's select
    // LIVE (GLV_OP_BARS_ONLY with the bars fused, kernel class 8; stateful only): magnitude, state and the row in LDS for the lane's LIVE_SLOTS live
    // register slots only, enumerated block by block of the row (see epilogue_gl16 below) -- what the fused bars never sample is neither computed nor kept
    template <int LOG_MODE, int EPI, int TILTREG = 0, bool R16 = false, bool NONFINITE = false, bool LIVE = false>
    GLV_HD static void epilogue(const cf (&v)[E], float* out_row, size_t row, int tid, const FrameArgs& a,
                                const LogEntry* logtab, const cf* tl_reg = nullptr) {
        using PI = PassInfo<P - 1>;
        constexpr bool STATE = EPI == EPI_MAG_STATE || EPI == EPI_RAW_STATE;
        static_assert(!LIVE || STATE, "the live class is a stateful one");
        constexpr int NS = LIVE ? (E >> PI::RB) * ((3 * PI::R + 7) / 8) : E;       // == LIVE_SLOTS (declared below)
        // register slot (gi, r) of the lane's idx-th point: r outer, gi inner; LIVE: r in the order of the row's blocks
        auto slot_r = [](int idx) constexpr -> int { return LIVE ? bitrev(idx / PI::NG, PI::RB) : idx / PI::NG; };
        // TILTREG 3: the lane's base term, re-defined opaquely per row -- the factors derived from it are loop invariant and
        // LLVM would otherwise hoist all 2E of them out of the row loop (and spill them)
        float tilt_base = 0.0f;
        if constexpr (TILTREG == 3 && LOG_MODE == 1) {
            tilt_base = tl_reg[0].x;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(tilt_base));
#endif
        }
        // magnitude of register slot (gi, r): abs/log/tilt, or the raw value
        auto value = [&](int gi, int r) -> cf {
            cf val = v[gi * PI::R + r];
            if constexpr (EPI == EPI_MAG || EPI == EPI_MAG_STATE) {
                const int q = out_index<P - 1>(tid, gi, r);     // = lane base + compile-time constant
                const float y0 = __builtin_fabsf(val.x) + 1.0f, y1 = __builtin_fabsf(val.y) + 1.0f;   // render.c:843-844
                cf tl;                                                                                      // :845 factors
                if constexpr (TILTREG == 1) tl = tl_reg[gi * PI::R + r];
                else if constexpr (TILTREG == 3 && LOG_MODE == 1) {
                    // n = 2 q = 2 * (lane part) + 2 * (compile-time part): the lane part went into tl_reg[0].x once per kernel
                    const int c = 2 * out_index<P - 1>(0, gi, r);
                    const TiltLin tlin = tilt_lin(a.inv_n, a.fft_scale, a.one_minus_cutoff);
                    tl.x = tilt_lin_at(tlin, tilt_base, c);
                    tl.y = tilt_lin_at(tlin, tilt_base, c + 1);
                } else if constexpr (TILTREG == 2 || TILTREG == 3) {
                    tl.x = tilt_factor<LOG_MODE == 1>(2 * q, a.inv_n, a.fft_scale, a.one_minus_cutoff);
                    tl.y = tilt_factor<LOG_MODE == 1>(2 * q + 1, a.inv_n, a.fft_scale, a.one_minus_cutoff);
                } else tl = ld<cf>(a.tilt, (uint32_t) q * 8u);
                val.x = log_third_nf<LOG_MODE, NONFINITE>(y0, logtab, log_tab_bits_of(LOG_NN)) * tl.x;
                val.y = log_third_nf<LOG_MODE, NONFINITE>(y1, logtab, log_tab_bits_of(LOG_NN)) * tl.y;
            }
            return val;
        };
        // one finished pair / point -> the output row (f32 or GL_R16 texels)
        auto store_pair = [&](uint32_t off, const cf2& two) {
            if constexpr (R16) {
                const u32x2 q = { pack_unorm16(two.a.x, two.a.y), pack_unorm16(two.b.x, two.b.y) };
                st<u32x2>(out_row, off / 2u, q);
            } else st<cf2>(out_row, off, two);
        };
        auto store_point = [&](uint32_t off, const cf& one) {
            if constexpr (R16) st<uint32_t>(out_row, off / 2u, pack_unorm16(one.x, one.y));
            else st<cf>(out_row, off, one);
        };
        if constexpr (!STATE) {
            // stateless: value -> store, pair by pair (nothing but the pair in flight)
            {
#pragma unroll
                for (int r = 0; r < PI::R; ++r) {
                    if constexpr (PI::NG >= 2) {
#pragma unroll
                        for (int gi = 0; gi < PI::NG; gi += 2) {             // adjacent groups: 16-byte stores
                            cf2 two;
                            two.a = value(gi, r);
                            two.b = value(gi + 1, r);
                            store_pair((uint32_t) out_index<P - 1>(tid, gi, r) * 8u, two);
                        }
                    } else {
                        store_point((uint32_t) out_index<P - 1>(tid, 0, r) * 8u, value(0, r));
                    }
                }
            }
        } else {
            // gravity without average: ONE state value per point, so all of the row's state can be requested
            // up front (2E registers) and its latency hides behind the log/tilt arithmetic -- in blocks, as below,
            // every block's loads were a separate exposed round trip
            if (!(a.ops & OP_AVERAGE) && (a.ops & OP_GRAVITY) && E <= 32) {
                const float* gs = a.grav + row * (size_t) N;
                float* gw = a.grav_w + row * (size_t) N;
                
                cf st0[NS];
#pragma unroll
                for (int idx = 0; idx < NS; ++idx)
                    st0[idx] = ld<cf>(gs, (uint32_t) out_index<P - 1>(tid, idx % PI::NG, slot_r(idx)) * 8u);
#pragma unroll
                for (int idx = 0; idx < NS; idx += (PI::NG >= 2 ? 2 : 1)) {
                    const int r = slot_r(idx), gi = idx % PI::NG;
                    const uint32_t off = (uint32_t) out_index<P - 1>(tid, gi, r) * 8u;
                    cf va = value(gi, r);
                    va.x = gravity(va.x, st0[idx].x, a.g); va.y = gravity(va.y, st0[idx].y, a.g);
                    if constexpr (PI::NG >= 2) {
                        cf vb = value(gi + 1, r);
                        vb.x = gravity(vb.x, st0[idx + 1].x, a.g); vb.y = gravity(vb.y, st0[idx + 1].y, a.g);
                        cf2 two; two.a = va; two.b = vb;
                        st<cf2>(gw, off, two);
                        if (out_row != nullptr) store_pair(off, two);
                    } else {
                        st<cf>(gw, off, va);
                        if (out_row != nullptr) store_point(off, va);
                    }
                }
                return;
            }
            // stateful: the lane's E points in blocks of BLK, loads of a block issued together
            // (apply_state_block); BLK bounds the registers the history loads need
            constexpr int BLK = LIVE ? NS : (E / 2 > 0 ? E / 2 : 1);
#pragma unroll
            for (int h0 = 0; h0 < NS; h0 += BLK) {
                cf val[BLK];
                uint32_t off[BLK];
                {
                    // enumeration order: r outer, gi inner => adjacent groups sit next to each other in val[]
#pragma unroll
                    for (int j = 0; j < BLK; ++j) {
                        const int idx = h0 + j, r = slot_r(idx), gi = idx % PI::NG;
                        val[j] = value(gi, r);
                        off[j] = (uint32_t) out_index<P - 1>(tid, gi, r) * 8u;
                    }
                }
                apply_state_block<BLK, (LOG_NN <= GLV_STATE_PAIR_MAX)>(val, off, row, (uint32_t) N, a);
                if (out_row == nullptr) continue;                     // uniform: output aliased to the gravity state
#pragma unroll
                for (int j = 0; j < BLK; ++j) {
                    if constexpr (PAIRS) {
                        if (j % 2 == 0) { cf2 two; two.a = val[j]; two.b = val[j + 1]; store_pair(off[j], two); }
                    } else {
                        store_point(off[j], val[j]);
                    }
                }
            }
        }
    }

    // ---- epilogue of the GL_R16 chain (glv_params.gl_storage == 1): render.c:2188-2265 inside the transform's launch ------------
    // magnitude -> upload quantisation (render.c:521-524) -> GL_MAX + gravity pass -> ring copy -> average pass on uint16 state
    // (gl16_state_block), then the finished texels leave as
    //   TO_LDS        floats c / 65535 into the slot's LDS region (fused bars: smooth_audio() samples them from there),
    //   a.ops & R16   GL_R16 texels, out_row = uint16 [n] (what the reference's `av` texture holds): 4 N bytes per frame,
    //   otherwise     the floats c / 65535, out_row = float [n].
    // Every stored value is a 16-bit integer by construction, so state and output move 2 bytes per value: with F = 5 a stereo
    // frame costs 4 N (PCM) + 16 N (four ring slots) + 4 N (newest slot) + 4 N (texels) = 28 N bytes where the pass-by-pass form
    // (f32 intermediates, three launches) moved ~80 N.
    static constexpr int GL16_BLK = E <= 16 ? E : E / 2;       // points per block: the whole lane where the registers allow it
    // LIVE (GLV_OP_BARS_ONLY batches, kernel class 7): the last pass leaves point bitrev(r) * L0 + G in register slot (gi, r), so the row's
    // blocks of L0 points are the same register slots in every lane.  The class keeps alive the first LIVE_RBLOCKS of the last pass's R
    // blocks -- three eighths of the row (one half where the last pass is radix 2 or 4): smooth_audio() at the shipped parameters samples
    // bins below 0.30 n -- as a COMPILE-TIME count: the lane's live points are one state block of LIVE_SLOTS entries with no test inside
    // (a run-time count cost a scalar branch per point and 208 bytes of scratch: profiles/r06/live_chain.txt).  The host takes this
    // class only when the bins the bars sample fit (glv_launch.h FrameGeometry::live_points), else the full chain of class 5.
    static constexpr int LIVE_RBLOCKS = (3 * PassInfo<P - 1>::R + 7) / 8;
    static constexpr int LIVE_SLOTS = LIVE_RBLOCKS * PassInfo<P - 1>::NG;
    static constexpr int LIVE_POINTS = LIVE_RBLOCKS * PassInfo<P - 1>::L0;
    static_assert(LIVE_SLOTS <= GL16_BLK && LIVE_SLOTS <= (E / 2 > 0 ? E / 2 : 1), "the live points of a lane are one state block");
    // the live class's state prefetch (gl16_state_prefetch): where its 4 x LIVE_SLOTS registers fit -- E <= 16
    static constexpr bool LIVE_PREFETCH = LOG_E <= 4;
    struct LivePre { uint32_t t[kLivePre][LIVE_SLOTS]; uint32_t n; };
    // byte offset (of a float row) behind the last point the bars sample, when it lies inside the LAST live block (else: no limit) -- uniform
    GLV_HD static uint32_t live_limit(const FrameArgs& a) {
        return a.live_points > (uint32_t) ((LIVE_RBLOCKS - 1) * PassInfo<P - 1>::L0) && a.live_points < (uint32_t) LIVE_POINTS ? a.live_points * 8u : 0xffffffffu;
    }
    GLV_HD static void live_offsets(uint32_t (&off)[LIVE_SLOTS], int tid) {      // byte offsets (of a float row) of the lane's live points, block by block of the row
        using PI = PassInfo<P - 1>;
#pragma unroll
        for (int j = 0; j < LIVE_SLOTS; ++j) off[j] = (uint32_t) out_index<P - 1>(tid, j % PI::NG, bitrev(j / PI::NG, PI::RB)) * 8u;
    }
    GLV_HD static void live_prefetch(LivePre& lp, size_t row, int tid, const FrameArgs& a) {
        uint32_t off[LIVE_SLOTS];
        live_offsets(off, tid);
        lp.n = gl16_state_prefetch<LIVE_SLOTS, (PassInfo<P - 1>::NG >= 2), PassInfo<P - 1>::NG>(lp.t, off, row, (uint32_t) N, a, live_limit(a));
    }
    template <int LOG_MODE, int TILTREG, bool NONFINITE, bool TO_LDS, bool LIVE = false>
    GLV_HD static void epilogue_gl16(const cf (&v)[E], float* out_row, size_t row, int tid, const FrameArgs& a,
                                     const LogEntry* logtab, const cf* tl_reg = nullptr, const LivePre* lp = nullptr) {
        using PI = PassInfo<P - 1>;
        constexpr bool PAIRED = PI::NG >= 2;
        constexpr int SLOTS = LIVE ? LIVE_SLOTS : E, BLK = LIVE ? LIVE_SLOTS : GL16_BLK;
        float tilt_base = 0.0f;
        if constexpr (TILTREG == 3 && LOG_MODE == 1) {
            tilt_base = tl_reg[0].x;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(tilt_base));
#endif
        }
        // the uploaded texel pair of register slot (gi, r): abs/log/tilt (render.c:842-846), clamp + quantise (:521-524)
        auto texels = [&](int gi, int r) -> uint32_t {
            const cf val = v[gi * PI::R + r];
            const int q = out_index<P - 1>(tid, gi, r);
            const float y0 = __builtin_fabsf(val.x) + 1.0f, y1 = __builtin_fabsf(val.y) + 1.0f;
            cf tl;
            if constexpr (TILTREG == 1) tl = tl_reg[gi * PI::R + r];
            else if constexpr (TILTREG == 3 && LOG_MODE == 1) {
                const int c = 2 * out_index<P - 1>(0, gi, r);
                const TiltLin tlin = tilt_lin(a.inv_n, a.fft_scale, a.one_minus_cutoff);
                tl.x = tilt_lin_at(tlin, tilt_base, c);
                tl.y = tilt_lin_at(tlin, tilt_base, c + 1);
            } else if constexpr (TILTREG == 2 || TILTREG == 3) {
                tl.x = tilt_factor<LOG_MODE == 1>(2 * q, a.inv_n, a.fft_scale, a.one_minus_cutoff);
                tl.y = tilt_factor<LOG_MODE == 1>(2 * q + 1, a.inv_n, a.fft_scale, a.one_minus_cutoff);
            } else tl = ld<cf>(a.tilt, (uint32_t) q * 8u);
            return pack_unorm16(log_third_nf<LOG_MODE, NONFINITE>(y0, logtab, log_tab_bits_of(LOG_NN)) * tl.x,
                                log_third_nf<LOG_MODE, NONFINITE>(y1, logtab, log_tab_bits_of(LOG_NN)) * tl.y);
        };
        const bool r16_out = (a.ops & OP_R16) != 0;                                // uniform
#pragma unroll
        for (int h0 = 0; h0 < SLOTS; h0 += BLK) {
            uint32_t tex[BLK], off[BLK];
            // enumeration order: r outer, gi inner => adjacent groups sit next to each other (LIVE: r in the order of the row's blocks)
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                const int idx = h0 + j, r = LIVE ? bitrev(idx / PI::NG, PI::RB) : idx / PI::NG, gi = idx % PI::NG;
                tex[j] = texels(gi, r);
                off[j] = (uint32_t) out_index<P - 1>(tid, gi, r) * 8u;
            }
            if constexpr (LIVE && LIVE_PREFETCH) {
                if (lp != nullptr) gl16_state_block<BLK, PAIRED, (LOG_NN <= GLV_STATE_PAIR_MAX), kLivePre, PI::NG>(tex, off, row, (uint32_t) N, a, lp->t, lp->n, live_limit(a));
                else gl16_state_block<BLK, PAIRED, (LOG_NN <= GLV_STATE_PAIR_MAX), 0, PI::NG>(tex, off, row, (uint32_t) N, a, nullptr, 0u, live_limit(a));
            } else if constexpr (LIVE) gl16_state_block<BLK, PAIRED, (LOG_NN <= GLV_STATE_PAIR_MAX), 0, PI::NG>(tex, off, row, (uint32_t) N, a, nullptr, 0u, live_limit(a));
            else gl16_state_block<BLK, PAIRED, (LOG_NN <= GLV_STATE_PAIR_MAX)>(tex, off, row, (uint32_t) N, a);
            if (TO_LDS || !r16_out) {
#pragma unroll
                for (int j = 0; j < BLK; j += (PAIRED ? 2 : 1)) {
                    // (out_limit: a store instruction covers consecutive points across its lanes, so whole instructions fall away)
                    if (!TO_LDS && a.out_limit != 0u && off[j] >= a.out_limit) continue;
                    if constexpr (PAIRED) { cf2 two; two.a = texels_to_float(tex[j]); two.b = texels_to_float(tex[j + 1]); st<cf2>(out_row, off[j], two); }
                    else st<cf>(out_row, off[j], texels_to_float(tex[j]));
                }
            } else {
#pragma unroll
                for (int j = 0; j < BLK; j += (PAIRED ? 2 : 1)) {
                    if (a.out_limit != 0u && off[j] >= a.out_limit) continue;      // (out_limit counts bytes of a FLOAT row: off[j] is that offset)
                    if constexpr (PAIRED) st<u32x2>(out_row, off[j] / 2u, u32x2{tex[j], tex[j + 1]});
                    else st<uint32_t>(out_row, off[j] / 2u, tex[j]);
                }
            }
        }
    }
};

}  // namespace glv
