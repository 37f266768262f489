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
#include <thread>
#include <algorithm>

#include "../../glava_amd/csrc/glv_frame.h"
#include "../../glava_amd/csrc/glv_tables.h"

using namespace glv;

template <int LOG_NN, int LOG_MODE, int LOG_E>
struct Emu {
    using FR = Frame<LOG_NN, LOG_E>;
    static constexpr int T = FR::T, P = FR::P, NN = FR::NN, N = FR::N;

    struct Thread { cf v[FR::E]; };

    // SC: the unit-twiddle shortcut of pass 0 (s16 input only, like Body<..., UNIT_SHORTCUT> in the kernel)
    template <int PASS, bool SC>
    static void run_pass(std::vector<Thread>& th, std::vector<cf>& lds, const cf* table) {
        for (int tid = 0; tid < T; ++tid) {
            if constexpr (PASS > 0) FR::template exchange_read<PASS>(th[tid].v, lds.data(), tid);
        }
        // (barrier) -- all reads done before the next exchange overwrites the region
        for (int tid = 0; tid < T; ++tid) {
            cf tw[FR::template PassInfo<PASS>::NTW];
            FR::template gather_tw<PASS>(tw, table, tid);
            FR::template compute<PASS, SC>(th[tid].v, tw);
            if constexpr (PASS < P - 1) FR::template exchange_write<PASS>(lds.data(), th[tid].v, tid);
        }
        // (barrier)
        if constexpr (PASS < P - 1) run_pass<PASS + 1, SC>(th, lds, table);
    }

    // one channel row given register-resident inputs -> out_row
    // NF: the row may hold non-finite values (f32 input), as in the kernel's epilogue instantiations
    template <bool NF>
    static void finish(std::vector<Thread>& th, float* out_row, size_t row, const FrameArgs& a) {
        const bool st = (a.ops & (OP_GRAVITY | OP_AVERAGE)) != 0, raw = (a.ops & OP_RAW) != 0;
        if (a.gl_storage == 1 && st && !raw) {       // the fused GL_R16 chain (kernel class 5): texels or their floats, as a.ops & OP_R16 says
            float* o = (a.ops & OP_R16) ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(a.out) + row * N) : out_row;
            for (int tid = 0; tid < T; ++tid) FR::template epilogue_gl16<LOG_MODE, 0, NF, false>(th[tid].v, o, row, tid, a, a.logtab);
            return;
        }
        for (int tid = 0; tid < T; ++tid) {
            if (raw && st)       FR::template epilogue<LOG_MODE, EPI_RAW_STATE, 0, false, NF>(th[tid].v, out_row, row, tid, a, a.logtab);
            else if (raw)        FR::template epilogue<LOG_MODE, EPI_RAW, 0, false, NF>(th[tid].v, out_row, row, tid, a, a.logtab);
            else if (st)         FR::template epilogue<LOG_MODE, EPI_MAG_STATE, 0, false, NF>(th[tid].v, out_row, row, tid, a, a.logtab);
            else                 FR::template epilogue<LOG_MODE, EPI_MAG, 0, false, NF>(th[tid].v, out_row, row, tid, a, a.logtab);
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
        run_pass<0, true>(th, lds, a.tw);
        finish<false>(th, a.out + row * N, row, a);
    }
    static void row_f32_stereo(const float* frame, size_t row, const FrameArgs& a) {
        std::vector<Thread> th(T);
        std::vector<cf> lds(FR::XREGION);
        for (int tid = 0; tid < T; ++tid)
            FR::load_f32_stereo_window(th[tid].v, frame, a.win, tid, (uint32_t) (row & 1), a.mono != 0);
        run_pass<0, false>(th, lds, a.tw);
        finish<true>(th, a.out + row * N, row, a);
    }
    static void row_f32(const float* in_row, size_t row, const FrameArgs& a) {
        std::vector<Thread> th(T);
        std::vector<cf> lds(FR::XREGION);
        for (int tid = 0; tid < T; ++tid) FR::load_f32_window(th[tid].v, in_row, a.win, tid);
        run_pass<0, false>(th, lds, a.tw);
        finish<true>(th, a.out + row * N, row, a);
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
        case 7:  run_units<7, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 8:  run_units<8, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 9:  run_units<9, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 10: run_units<10, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 11: run_units<11, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 12: run_units<12, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 13: run_units<13, LOG_MODE, LOG_E>(in_mode, a); return 0;
        case 14: run_units<14, LOG_MODE, LOG_E>(in_mode, a); return 0;
    }
    return 1;
}

extern "C" {

// n: real samples per channel.  in: s16 [units/2][n][2] (in_mode 0) or f32 [units][n] (in_mode 1);
// units = channel rows.
// grav / hist may be NULL when the op is not requested.  Returns 0 on success.
// gl_storage 1: grav / hist are uint16 arrays (texels), out is uint16 rows when ops has OP_R16; force_float_gravity: evaluate the
// gravity step with the float expression even where the integer form is proven (both must give the same texels)
int glvemu_process_gl(int n, int in_mode, const void* in, float* out, float* grav, float* hist,
                      unsigned units, unsigned ops, unsigned F, unsigned head, int mono, int avg_window,
                      int avg_kind, int log_mode, float fft_scale, float fft_cutoff, float gravity_step, float ur,
                      unsigned rot, int log_e, int gl_storage, int force_float_gravity);
int glvemu_process(int n, int in_mode, const void* in, float* out, float* grav, float* hist,
                   unsigned units, unsigned ops, unsigned F, unsigned head, int mono, int avg_window,
                   int avg_kind, int log_mode, float fft_scale, float fft_cutoff, float gravity_step, float ur,
                   unsigned rot, int log_e) {
    return glvemu_process_gl(n, in_mode, in, out, grav, hist, units, ops, F, head, mono, avg_window, avg_kind, log_mode, fft_scale, fft_cutoff,
                             gravity_step, ur, rot, log_e, 0, 0);
}
int glvemu_process_gl(int n, int in_mode, const void* in, float* out, float* grav, float* hist,
                      unsigned units, unsigned ops, unsigned F, unsigned head, int mono, int avg_window,
                      int avg_kind, int log_mode, float fft_scale, float fft_cutoff, float gravity_step, float ur,
                      unsigned rot, int log_e, int gl_storage, int force_float_gravity) {
    int log_nn = 0;
    while ((2 << log_nn) < n) ++log_nn;
    if ((2 << log_nn) != n) return 2;
    const int nn = n / 2;
    std::vector<cf> tw(nn, cf{0.0f, 0.0f});
    std::vector<double> win(n);
    make_twiddles(tw.data(), nn);
    make_window(win.data(), n);
    LogEntry lt[kLogTabMaxSize];
    make_log_table(lt, log_tab_bits_of(log_nn));         // the emulated kernel reads the table it would have staged into LDS
    std::vector<float> tl(n);
    make_tilt(tl.data(), n, fft_scale, fft_cutoff, log_mode == 1);
    FrameArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.grav = grav; a.grav_w = grav; a.hist = hist; a.tw = tw.data(); a.win = win.data(); a.logtab = lt; a.tilt = tl.data();
    a.units = units; a.ops = ops; a.F = F; a.head = head; a.mono = mono; a.avg_window = avg_window; a.rot = rot;
    a.inv_n = 1.0f / (float) n; a.fft_scale = fft_scale; a.one_minus_cutoff = 1.0f - fft_cutoff;
    a.g = gravity_step * (1.0f / ur); a.F_as_float = (float) F; a.F_rcp = 1.0f / (float) F;
    a.gl_storage = (uint32_t) gl_storage;
    a.grav_int = gravity_r16_integer_step(a.g, &a.grav_sub) && !force_float_gravity ? 1u : 0u;
    if (F > 64) return 3;
    make_frame_weights(a.wts, F, avg_window != 0, avg_kind);
    for (unsigned f = 0; f < F; ++f) a.wts32[f] = (float) a.wts[f];
    if (log_e == 5) return log_mode == 0 ? dispatch<0, 5>(log_nn, in_mode, a) : log_mode == 1 ? dispatch<1, 5>(log_nn, in_mode, a) : dispatch<2, 5>(log_nn, in_mode, a);
    if (log_e == 3) return log_mode == 0 ? dispatch<0, 3>(log_nn, in_mode, a) : log_mode == 1 ? dispatch<1, 3>(log_nn, in_mode, a) : dispatch<2, 3>(log_nn, in_mode, a);
    return log_mode == 0 ? dispatch<0, 4>(log_nn, in_mode, a) : log_mode == 1 ? dispatch<1, 4>(log_nn, in_mode, a) : dispatch<2, 4>(log_nn, in_mode, a);
}


// glv_post_kernel's loop on the host: gravity / average (optionally with the GL_R16 storage model, glv_params.gl_storage:
// 2 = float state, 1 = grav / hist point at uint16 texel arrays) on planar rows through the same apply_state the kernel calls.
// Returns 0 on success.
int glvemu_post_state(const float* in, float* out, float* grav, float* hist, int n, unsigned rows, unsigned ops, unsigned F,
                      unsigned head, int avg_window, int avg_kind, int gl_storage, float gravity_step, float ur) {
    if (F == 0 || F > 64) return 3;
    FrameArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.grav = grav; a.grav_w = grav; a.hist = hist; a.units = rows; a.ops = ops; a.F = F; a.head = head;
    a.avg_window = avg_window; a.gl_storage = gl_storage; a.g = gravity_step * (1.0f / ur); a.F_as_float = (float) F; a.F_rcp = 1.0f / (float) F;
    a.grav_int = gravity_r16_integer_step(a.g, &a.grav_sub) ? 1u : 0u;       // gl_storage 1: grav / hist are uint16 arrays
    make_frame_weights(a.wts, F, avg_window != 0, avg_kind);
    for (unsigned f = 0; f < F; ++f) a.wts32[f] = (float) a.wts[f];
    for (unsigned r = 0; r < rows; ++r)
        for (int q = 0; q < n / 2; ++q) {
            const uint32_t off = (uint32_t) q * 8u;
            cf val = ld<cf>(in + (size_t) r * n, off);
            val = apply_state(val, off, r, (uint32_t) n, a);
            st<cf>(out + (size_t) r * n, off, val);
        }
    return 0;
}

}  // extern "C"

// ---- GLV_OP_BARS on the host: the same tables, work lists and chunk arithmetic as the kernels; the
// DPP sum over a group of GL = 2 / 4 / 8 lanes (glv_frame.h group_sum) is restated as the data movement its steps perform.
namespace {
float group_sum_host(const float* lane, int gl) {
    float v[8], t[8];
    for (int i = 0; i < gl; ++i) v[i] = lane[i];
    auto step = [&](auto src) { for (int i = 0; i < gl; ++i) t[i] = v[i] + v[src(i)]; for (int i = 0; i < gl; ++i) v[i] = t[i]; };
    step([](int i) { return i ^ 1; });                                // quad_perm [1,0,3,2]
    if (gl >= 4) step([](int i) { return i ^ 2; });                   // quad_perm [2,3,0,1]
    if (gl >= 8) step([](int i) { return 7 - i; });                   // row_half_mirror
    return v[0];
}
}  // namespace

extern "C" {
// glv_core.h div_65535 (the s16 unpack and the GL_R16 texel readback) for every integer in [lo, hi]
void glvemu_div_65535(int lo, int hi, float* out) {
    for (int v = lo; v <= hi; ++v) out[v - lo] = glv::div_65535((float) v);
}
}

extern "C" {
// bars of `nrows` rows of n floats through work lists for `groups` groups of bar_lanes_of(n) lanes.  steps_out (may be NULL)
// receives the step count; returns 0 on success.
int glvemu_bars_rows(const float* tex, int n, int bars, float smooth_factor, float phase, int bins, float* out);
int glvemu_bars(const float* spec, size_t nrows, int n, int bars, float smooth_factor, int groups, float* out, unsigned* steps_out, float phase) {
    if ((uint32_t) bars >= glv::kBarSeqMin) {              // like launch_bars: from 256 bars up the chain kernels and their tile tables
        for (size_t r = 0; r < nrows; ++r)
            if (glvemu_bars_rows(spec + r * (size_t) n, n, bars, smooth_factor, phase, 288, out + r * (size_t) bars)) return -1;
        if (steps_out) *steps_out = 0;
        return 0;
    }
    using namespace glv;
    std::vector<BarDesc> desc;
    std::vector<float> w;
    make_bar_taps(desc, w, (uint32_t) n, (uint32_t) bars, smooth_factor, phase);
    if (!bar_chunks_in_row(desc, (uint32_t) n)) return 1;
    const uint32_t zero_off = (uint32_t) w.size();
    const uint32_t chunk = bar_chunk_of((uint32_t) n);
    const int gl = bar_lanes_of((uint32_t) n);
    w.resize(w.size() + chunk, 0.0f);
    std::vector<BarItem> items;
    const uint32_t nsteps = make_bar_items(items, desc, (uint32_t) groups, zero_off, chunk);
    if (steps_out) *steps_out = nsteps;
    for (size_t row = 0; row < nrows; ++row) {
        const float* tex = spec + row * (size_t) n;
        for (int g = 0; g < groups; ++g) {
            float total = 0.0f;
            for (uint32_t s = 0; s < nsteps; ++s) {
                const BarItem it = items[(size_t) s * groups + g];
                float lane[8];
                for (int sub = 0; sub < gl; ++sub) lane[sub] = bar_item_lane_sum(bar_item_load(tex, w.data(), it, sub));
                total = __builtin_fmaf(total, it.keep, group_sum_host(lane, gl));
                if (it.res != (uint32_t) bars) out[row * bars + it.res] = total / desc[it.res].weight_sum;
            }
        }
    }
    return 0;
}

// work-list invariants for the tests: returns 0 when every chunk of every bar appears exactly once, a bar's
// chunks sit in ONE group's list in increasing order, the last chunk (only) carries the end flag, padding
// items point at zero weights, and the step count is a multiple of kBarBatch with one batch of padding rows.
int glvemu_bar_items_check(int n, int bars, float smooth_factor, int groups) {
    using namespace glv;
    std::vector<BarDesc> desc;
    std::vector<float> w;
    make_bar_taps(desc, w, (uint32_t) n, (uint32_t) bars, smooth_factor);
    const uint32_t kBarChunk = bar_chunk_of((uint32_t) n);
    for (const BarDesc& d : desc) { if (d.count == 0 || d.first_bin + d.count > (uint32_t) n || d.tap_offset % kBarChunk) return 1; }
    const uint32_t zero_off = (uint32_t) w.size();
    w.resize(w.size() + kBarChunk, 0.0f);
    std::vector<BarItem> items;
    const uint32_t nsteps = make_bar_items(items, desc, (uint32_t) groups, zero_off, kBarChunk);
    if (nsteps % kBarBatch || items.size() != (size_t) (nsteps + kBarBatch) * groups) return 2;
    std::vector<int> owner(desc.size(), -1);
    std::vector<uint32_t> next_chunk(desc.size(), 0);
    for (int g = 0; g < groups; ++g) {
        bool open = false;                         // a bar of this group has chunks still to come
        for (uint32_t s = 0; s < nsteps + kBarBatch; ++s) {
            const BarItem it = items[(size_t) s * groups + g];
            if (it.w_byte == zero_off * 4u) { if (it.tex_byte != 0 || it.res != (uint32_t) bars || it.keep != 0.0f || open) return 3; continue; }
            if (s >= nsteps) return 4;
            // which bar: the one whose weights this chunk points into
            uint32_t k = 0;
            while (k < desc.size() && !(it.w_byte / 4u >= desc[k].tap_offset && it.w_byte / 4u < desc[k].tap_offset + desc[k].count)) ++k;
            if (k >= desc.size()) return 5;
            if (owner[k] == -1) owner[k] = g; else if (owner[k] != g) return 6;
            const uint32_t i0 = next_chunk[k];
            if (it.w_byte != (desc[k].tap_offset + i0) * 4u || it.tex_byte != (desc[k].first_bin + i0) * 4u) return 7;
            const bool last = i0 + kBarChunk >= desc[k].count;
            if (it.res != (last ? k : (uint32_t) bars) || it.keep != (i0 == 0 ? 0.0f : 1.0f) || open != (i0 != 0)) return 8;
            open = !last;
            for (uint32_t j = 0; j < kBarChunk; ++j) if (i0 + j >= desc[k].count && w[it.w_byte / 4u + j] != 0.0f) return 9;
            next_chunk[k] = i0 + kBarChunk;
        }
    }
    for (size_t k = 0; k < desc.size(); ++k) if (next_chunk[k] < desc[k].count) return 10;
    return 0;
}
}  // extern "C"

extern "C" {
// glv_tables.h gravity_r16_integer_step: 1 when the gravity pass on texels is m -> max(m - D, 0) for every texel value (D -> *d)
int glvemu_gravity_step(float gravity_step, float ur, unsigned* d) {
    uint32_t sub = 0;
    const bool ok = glv::gravity_r16_integer_step(gravity_step * (1.0f / ur), &sub);
    if (d) *d = sub & 0xffffu;
    return ok ? 1 : 0;
}
}

extern "C" {
// glv_core.h div_frames against the division for every float with bits in [lo_bits, hi_bits] (step `stride` in bit space):
// returns the number of mismatches (first one -> *bad_bits)
unsigned long long glvemu_div_frames_check(unsigned F, unsigned lo_bits, unsigned hi_bits, unsigned stride, unsigned* bad_bits) {
    const float Ff = (float) F, r = 1.0f / Ff;
    unsigned long long bad = 0;
    for (unsigned long long u = lo_bits; u <= hi_bits; u += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t) u);
        const float want = x / Ff, got = glv::div_frames(x, Ff, r);
        if (__builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, got)) { if (!bad && bad_bits) *bad_bits = (unsigned) u; ++bad; }
    }
    return bad;
}
}

extern "C" {
// glv_tables.h make_bar_mtiles (the host tables of the many-bars kernels): 0 when the tiles cover every bar once in order, 32 at a
// time, from a first bin that is a multiple of 4 over a whole number of step groups; the weights in MFMA operand layout hold exactly
// the bars' weights at the bars' bins and +0 elsewhere; the rounds cover every tile once in order, at most tiles_per_round at a time,
// fit the ring (their own bins and what the next round adds), and are monotone.  -1: no rounds for this ring (the matrix-core
// kernel is then not used); -2: no tables at all (fewer than 256 bars).
int glvemu_bar_tiles_check(int n, int bars, float smooth_factor, float phase, int bins, int tiles_per_round, unsigned* nrounds_out, unsigned* max_count_out) {
    using namespace glv;
    std::vector<BarDesc> desc;
    std::vector<float> w;
    make_bar_taps(desc, w, (uint32_t) n, (uint32_t) bars, smooth_factor, phase);
    uint32_t mc = 0;
    for (const BarDesc& d : desc) mc = d.count > mc ? d.count : mc;
    if (max_count_out) *max_count_out = mc;
    std::vector<BarMTile> mt;
    std::vector<BarTile> rounds;
    std::vector<float> wt, wsum;
    if (!make_bar_mtiles(mt, wt, wsum, rounds, desc, w, (uint32_t) n, (uint32_t) bins, (uint32_t) tiles_per_round)) return -2;
    if (mt.size() != ((size_t) bars + 31) / 32 || wsum.size() != mt.size() * 64) return 7;
    for (size_t T = 0; T < mt.size(); ++T) {
        const BarMTile& t = mt[T];
        if (t.k0 != 32 * T || (t.origin & 3u) || t.steps == 0 || t.steps % kBarStepPad) return 1;
        for (uint32_t j = 0; j < 32; ++j) {
            const uint32_t k = t.k0 + j;
            if (k < (uint32_t) bars && (desc[k].first_bin < t.origin || desc[k].first_bin + desc[k].count > t.origin + 2 * t.steps)) return 3;
            for (uint32_t i = 0; i < 2 * t.steps; ++i) {
                const uint32_t bin = t.origin + i;
                const float want = k < (uint32_t) bars && bin >= desc[k].first_bin && bin < desc[k].first_bin + desc[k].count ? w[desc[k].tap_offset + bin - desc[k].first_bin] : 0.0f;
                if (__builtin_bit_cast(uint32_t, wt[t.w_off + i * 32 + j]) != __builtin_bit_cast(uint32_t, want)) return 6;
            }
            const float ws = k < (uint32_t) bars ? desc[k].weight_sum : 1.0f;
            if (wsum[2 * k] != ws || wsum[2 * k + 1] != (bar_rcp_division_ok(ws) ? 1.0f / ws : 0.0f)) return 5;
        }
    }
    if (rounds.empty()) return -1;
    if (nrounds_out) *nrounds_out = (unsigned) rounds.size();
    uint32_t next = 0;
    for (size_t i = 0; i < rounds.size(); ++i) {
        const BarTile& r = rounds[i];
        if (r.k0 != next || r.k1 <= r.k0 || r.k1 - r.k0 > (uint32_t) tiles_per_round) return 2;
        if ((r.origin & 3u) || (r.end & 3u) || r.end <= r.origin || r.end - r.origin > (uint32_t) bins || r.end > (uint32_t) n) return 4;
        for (uint32_t T = r.k0; T < r.k1; ++T)
            for (uint32_t k = mt[T].k0; k < mt[T].k0 + 32 && k < (uint32_t) bars; ++k)
                if (desc[k].first_bin < r.origin || desc[k].first_bin + desc[k].count > r.end) return 8;
        if (i && (r.end < rounds[i - 1].end || r.origin < rounds[i - 1].origin || r.end - rounds[i - 1].origin > (uint32_t) bins)) return 9;
        next = r.k1;
    }
    if (next != mt.size()) return 10;
    return 0;
}

// The many-bars kernels' arithmetic on the host, off the same tables: per tile and bar ONE fmaf chain over the tile's whole (padded)
// bin range, the other bars' bins and the padding weighing +0 -- what the MFMA computes.  Texels past the row (padding) read as the
// row's last one.  out: bars floats.  Returns 0, or -2 when there are no tables.
int glvemu_bars_rows(const float* tex, int n, int bars, float smooth_factor, float phase, int bins, float* out) {
    using namespace glv;
    std::vector<BarDesc> desc;
    std::vector<float> w;
    make_bar_taps(desc, w, (uint32_t) n, (uint32_t) bars, smooth_factor, phase);
    std::vector<BarMTile> mt;
    std::vector<BarTile> rounds;
    std::vector<float> wt, wsum;
    if (!make_bar_mtiles(mt, wt, wsum, rounds, desc, w, (uint32_t) n, (uint32_t) bins, 4u)) return -2;
    auto clamp01 = [](float v) { return v > 0.0f ? (v < 1.0f ? v : 1.0f) : 0.0f; };      // NaN -> 0
    for (const BarMTile& t : mt)
        for (uint32_t j = 0; j < 32 && t.k0 + j < (uint32_t) bars; ++j) {
            float acc = 0.0f;
            for (uint32_t i = 0; i < 2 * t.steps; ++i) {
                const uint32_t bin = t.origin + i;
                acc = fmaf(clamp01(tex[bin < (uint32_t) n ? bin : (uint32_t) n - 1]), wt[t.w_off + i * 32 + j], acc);
            }
            out[t.k0 + j] = acc / wsum[2 * (t.k0 + j)];
        }
    return 0;
}

// The i8 matrix-core form of the many-bars pass over TEXEL rows (glv_tables.h make_bar_itiles, glv_misc.hip glv_bars_rows_i8_kernel) on the
// host, off the same tables, the way the kernel consumes them: the ring of `bins` bins per row as two planes of signed bytes (c ^ 0x8080),
// filled round by round (a slot is OVERWRITTEN when the ring wraps, exactly as on the device: a table whose rounds let a wave read a bin
// that is gone shows up here), per tile / step / digit the 64 x 16 bytes of the b-operand against the 16 bytes per lane of the a-operand,
// four int32 accumulators, and the kernel's epilogue  texel = (uint32) ((a3 << 8) + a2 + ((a1 + (a0 >> 8)) >> 8) + c) >> s.
// out16 / outf: bars values (either may be NULL).  Returns 0, -2 without tables, -3 without rounds for this ring.
int glvemu_bars_int(const uint16_t* tex, int n, int bars, float smooth_factor, float phase, int bins, uint16_t* out16, float* outf) {
    using namespace glv;
    std::vector<BarDesc> desc;
    std::vector<float> w;
    make_bar_taps(desc, w, (uint32_t) n, (uint32_t) bars, smooth_factor, phase);
    std::vector<BarMTile> it;
    std::vector<BarTile> rounds;
    std::vector<int8_t> wq;
    std::vector<BarIFin> fin;
    if (!make_bar_itiles(it, wq, fin, rounds, desc, w, (uint32_t) n, (uint32_t) bins, 4u)) return -2;
    if (rounds.empty()) return -3;
    const uint32_t S = (uint32_t) bins;
    std::vector<int8_t> ph(S, 99), pl(S, -77);                                   // garbage where nothing was parked: must meet weight 0 only
    auto park = [&](uint32_t b0, uint32_t b1) {
        for (uint32_t b = b0; b < b1; ++b) {
            const uint16_t c = tex[b < (uint32_t) n ? b : (uint32_t) n - 1];
            ph[b % S] = (int8_t) ((c >> 8) ^ 0x80); pl[b % S] = (int8_t) ((c & 255) ^ 0x80);
        }
    };
    uint32_t filled_to = rounds[0].end;
    park(rounds[0].origin, rounds[0].end);
    std::vector<long long> before;
    for (size_t t = 0; t < rounds.size(); ++t) {
        const BarTile& R = rounds[t];
        const uint32_t next_end = t + 1 < rounds.size() ? rounds[t + 1].end : filled_to;
        // the round's tiles are evaluated TWICE -- before and after the next round's bins are parked: on the device the other waves park
        // them while this one still reads, so a table is only right if nothing a tile reads with a non-zero weight changes
        for (int pass = 0; pass < 2; ++pass) {
            size_t at = 0;
            for (uint32_t Ti = R.k0; Ti < R.k1; ++Ti) {
                const BarMTile& T = it[Ti];
                if (T.origin % 16u) return 1;
                for (uint32_t j = 0; j < 32 && T.k0 + j < (uint32_t) bars; ++j) {
                    int32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    for (uint32_t s = 0; s < T.steps; ++s)
                        for (uint32_t half = 0; half < 2; ++half)
                            for (uint32_t q = 0; q < 16; ++q) {
                                const uint32_t slot = (T.origin + 32 * s + 16 * half + q) % S;
                                const int32_t h = ph[slot], l = pl[slot];
                                const int8_t* wb = wq.data() + (size_t) T.w_off * 16u + (((size_t) s * 3u) * 64u + half * 32u + j) * 16u + q;
                                const int32_t w0 = wb[0], w1 = wb[(size_t) 64 * 16], w2 = wb[(size_t) 2 * 64 * 16];
                                a3 += h * w2; a2 += h * w1 + l * w2; a1 += h * w0 + l * w1; a0 += l * w0;
                            }
                    const BarIFin f = fin[T.k0 + j];
                    const int P = (int) f.s + 16;
                    const long long tot = ((long long) a3 << 24) + ((long long) a2 << 16) + ((long long) a1 << 8) + a0 + ((long long) 32896 << P);
                    if (pass == 0) before.push_back(tot); else if (before[at++] != tot) return 2;
                    if (out16) out16[T.k0 + j] = (uint16_t) (((uint32_t) ((a3 << 8) + a2 + ((a1 + (a0 >> 8)) >> 8)) + f.c) >> f.s);      // (weights that sum to 0: 0 >> 16)
                    if (outf) outf[T.k0 + j] = f.s == kBarIFinNone ? __builtin_nanf("") : (float) (ldexp((double) tot, -P) / 65535.0);
                }
            }
            if (pass == 0 && next_end > filled_to) { park(filled_to, next_end); filled_to = next_end; }
        }
        before.clear();
    }
    return 0;
}

// The rows kernel's three-instruction division by a bar's weight sum (glv_tables.h bar_rcp_division_ok; glv_misc.hip): for every bar
// of the table, q0 = a * r, rem = fma(-q0, b, a), q = fma(rem, r, q0) against a / b for EVERY significand of a in two binades around
// b -- every b_stride-th distinct weight sum -- (scaling a by a power of two scales everything exactly while nothing leaves the normal range, which the kernel's 2^-90 guard and
// totals <= 2^8 ensure), and at the guard's edge.  Returns the number of mismatches; *checked = quotients compared.
unsigned long long glvemu_div_rcp_check(int n, int bars, float smooth_factor, float phase, int nthreads, int b_stride, unsigned long long* checked) {
    using namespace glv;
    std::vector<BarDesc> desc;
    std::vector<float> w;
    make_bar_taps(desc, w, (uint32_t) n, (uint32_t) bars, smooth_factor, phase);
    std::vector<float> bs;
    for (const BarDesc& d : desc) bs.push_back(d.weight_sum);
    std::sort(bs.begin(), bs.end());
    bs.erase(std::unique(bs.begin(), bs.end()), bs.end());
    std::vector<unsigned long long> bad((size_t) nthreads, 0ull), cnt((size_t) nthreads, 0ull);
    std::vector<std::thread> th;
    for (int ti = 0; ti < nthreads; ++ti)
        th.emplace_back([&, ti]() {
            for (size_t i = (size_t) ti * (size_t) b_stride; i < bs.size(); i += (size_t) nthreads * (size_t) b_stride) {
                const float b = bs[i];
                if (!bar_rcp_division_ok(b)) continue;
                const float r = 1.0f / b;
                int eb; (void) frexpf(b, &eb);
                for (int e : {eb - 1, eb, -89 + eb, 7}) {                            // a in [2^(e-1), 2^e) ... incl. the smallest and largest totals the fast path sees
                    const uint32_t base = __builtin_bit_cast(uint32_t, ldexpf(0.5f, e));
                    for (uint32_t m = 0; m < (1u << 23); ++m) {
                        const float a = __builtin_bit_cast(float, base + m);
                        const float q0 = a * r, rm = fmaf(-q0, b, a), q = fmaf(rm, r, q0);
                        if (__builtin_bit_cast(uint32_t, q) != __builtin_bit_cast(uint32_t, a / b)) ++bad[(size_t) ti];
                    }
                    cnt[(size_t) ti] += 1ull << 23;
                }
            }
        });
    for (auto& t : th) t.join();
    unsigned long long nb = 0, nc = 0;
    for (int ti = 0; ti < nthreads; ++ti) { nb += bad[(size_t) ti]; nc += cnt[(size_t) ti]; }
    if (checked) *checked = nc;
    return nb;
}

}
