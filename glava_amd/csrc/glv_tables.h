// glv_tables.h -- host-side generation of the constant tables the kernels consume.
//
// These are the only places libm's cos()/sin() are evaluated; they run on the host with the
// same glibc the reference would use, so the tables carry the reference's exact bits:
//   window    glava/render.c:660 (macro) as expanded at the call site :794
//   twiddles  glava/render.c:817-836 (float recurrence seeded by double sin)
//   weights   glava/render.c:661 (macro) as expanded at :766
// Host compile flags must not contract or reassociate (-ffp-contract=off, no -ffast-math).
#pragma once

#include <math.h>
#include <stddef.h>

#include <algorithm>
#include <vector>

#include "glv_core.h"
#include "glv_frame.h"

namespace glv {

static const double kTwoPi = 6.28318530718;   // render.c:63 (TWOPI), deliberately not 2*M_PI

// window(i, s->sz - 1) with the macro's unparenthesised `sz`:
//   0.53836 - 0.46164 * cos(TWOPI * i / N - 1)
inline void make_window(double* w, size_t n) {
    for (size_t i = 0; i < n; ++i)
        w[i] = 0.53836 - (0.46164 * cos(kTwoPi * (double) i / (double) n - 1));
}

// Average weights by AGE f (0 = oldest frame ... F-1 = newest), as apply_state consumes them.
//   kind 0, CPU twin (render.c:751-766): window_frame(f, avg_frames - 1) with the macro's unparenthesised
//           `sz`: 0.6 - 0.4*cos(TWOPI*f/F - 1); f = 0 is the oldest frame.
//   kind 1, GL twin (average_pass.frag:19-45, common.glsl:13): window(I, _AVG_FRAMES - 1) -- same macro quirk,
//           Hamming coefficients -- with I = 0 the MOST RECENT frame (render.c:2247-2256), i.e. I = F-1-f;
//           no window at all when F == 2 (average_pass.frag:27-29).
inline void make_frame_weights(double* w, size_t F, bool use_window, int kind) {
    for (size_t f = 0; f < F; ++f) {
        if (!use_window || (kind == 1 && F == 2)) w[f] = 1.0;
        else if (kind == 0) w[f] = 0.6 - (0.4 * cos(kTwoPi * (double) f / (double) F - 1));
        else w[f] = 0.53836 - (0.46164 * cos(kTwoPi * (double) (F - 1 - f) / (double) F - 1));
    }
}

// The gravity pass on GL_R16 texels (glv_core.h gravity_r16): is  m -> unorm16(unorm16_to_float(m) - g)  the integer step
// m -> max(m - D, 0) for EVERY texel value m?  Checked exhaustively (65 536 evaluations of the float expression the pass is
// defined by); *sub receives D | D << 16.  False for steps that raise values (g < 0) or where g * 65535 sits so close to a
// half-integer that float rounding decides texel by texel -- the kernels then evaluate the float expression as written.
inline bool gravity_r16_integer_step(float g, uint32_t* sub) {
    const uint32_t top = unorm16(unorm16_to_float(65535u) - g);
    const uint32_t d = 65535u - top;                    // the only candidate: the step taken from the largest texel
    *sub = d | (d << 16);
    if (top > 65535u) return false;
    for (uint32_t m = 0; m < 65536u; ++m)
        if (unorm16(unorm16_to_float(m) - g) != (m > d ? m - d : 0u)) return false;
    return true;
}

// All radix-2 stages of an nn-point transform: stage with half size L at [L-1, 2L-1).
inline void make_twiddles(cf* table, size_t nn) {
    for (size_t L = 1; L < nn; L <<= 1) {
        const size_t mmax = 2 * L;
        const float theta = (float) (-(2 * M_PI / (double) mmax));
        const float wtemp = (float) sin(0.5 * (double) theta);
        const float wpr = (float) (-2.0 * (double) wtemp * (double) wtemp);
        const float wpi = (float) sin((double) theta);
        float wr = 1.0f, wi = 0.0f;
        cf* t = table + tw_offset((int) L);
        for (size_t k = 0; k < L; ++k) {
            t[k].x = wr; t[k].y = wi;
            const float wt = wr;
            const float a = wr * wpr, b = wi * wpi;
            const float ab = a - b;
            wr = wr + ab;
            const float c = wi * wpr, d = wt * wpi;
            const float cd = c + d;
            wi = wi + cd;
        }
    }
}

// render.c:845 per-bin tilt factors, evaluated with the reference's float operations.
// fold_ln2_3: log_mode 1 only -- the table carries tilt * (ln2/3) so the kernel multiplies log2(y) once.
inline void make_tilt(float* t, size_t n, float fft_scale, float fft_cutoff, bool fold_ln2_3 = false) {
    const float inv_n = 1.0f / (float) n;              // n is a power of two: exact
    const float omc = 1.0F - fft_cutoff;
    for (size_t i = 0; i < n; ++i) {
        t[i] = fold_ln2_3 ? tilt_factor<true>((int) i, inv_n, fft_scale, omc) : tilt_factor<false>((int) i, inv_n, fft_scale, omc);
    }
}

// transform_smooth window bounds (render.c:699-707): they depend only on the output index t.
//   asz = ceil(sz / smooth_ratio); smin = floor(e^max(log t - d, 0)); smax = min(ceil(e^(log t + d)), sz - 1)
inline size_t make_smooth_bounds(int* smin, int* smax, size_t sz, float smooth_distance, float smooth_ratio) {
    const double kE = 2.7182818284590452353;           // render.c:692
    const size_t asz = (size_t) ceil(sz / smooth_ratio);
    for (size_t t = 0; t < asz; ++t) {
        const float db = (float) log((double) (int) t);
        float lo = db - smooth_distance;
        if (!(lo > 0)) lo = 0;
        smin[t] = (int) floor(powf((float) kE, lo));
        const int hi = (int) ceil(powf((float) kE, db + smooth_distance));
        smax[t] = hi < (int) sz - 1 ? hi : (int) sz - 1;
    }
    return asz;
}

// smooth_audio() taps of every bar (shaders/glava/util/smooth.glsl:13-59 in float, as the GLSL would):
//   idx = (k + phase) / bars (phase 0: the modules' bar positions; 0.5: the texel centres of smooth_pass.frag);
//   smin/smax = scale_audio(clamp(idx -/+ factor)) * n,  scale_audio(u) = -log(1 - SAMPLE_RANGE u) / SAMPLE_SCALE   (smooth.glsl:13-15)
//   m = (smax - smin)/2, rm = smin + m;  for s = smin; s <= smax; s += 1:  w = ROUND_FORMULA(clamp((m - |rm - s|)/m))
//   sample bin int(round(s)).  Consecutive s round to consecutive bins, so a bar is a contiguous bin range.
// BarShape: the GLSL `#define`s of smooth_parameters.glsl:17-42 (glv_params round_formula / sample_scale / sample_range); `inclusive`:
// SAMPLE_MODE average walks s <= smax (smooth.glsl:34), maximum and hybrid s < smax (:43, :54).
struct BarShape { uint32_t round_formula = 0; float scale = 8.0f, range = 0.9f; bool inclusive = true; };
inline float bar_round_formula(uint32_t formula, float x) {                       // common.glsl:17-22
    if (formula == 1u) return sqrtf(1.0f - ((x - 1.0f) * (x - 1.0f)));            // circular
    if (formula == 2u) return x;                                                  // linear
    return (0.5f * sinf((3.14159265359f * x) - (3.14159265359f / 2.0f))) + 0.5f;  // sinusoidal
}
inline void make_bar_taps(std::vector<BarDesc>& desc, std::vector<float>& tap_w, uint32_t n, uint32_t bars, float smooth_factor, float phase = 0.0f, const BarShape& shape = BarShape{}) {
    const uint32_t chunk = bar_chunk_of(n);
    auto scale = [&](float u) { return -logf((-shape.range * u) + 1.0f) / shape.scale; };
    auto clamp01 = [](float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); };
    desc.resize(bars);
    tap_w.clear();
    for (uint32_t k = 0; k < bars; ++k) {
        const float idx = phase == 0.0f ? (float) k / (float) bars : ((float) k + phase) / (float) bars;   // gl_FragCoord.x / w
        const float smin = scale(clamp01(idx - smooth_factor)) * (float) n;
        const float smax = scale(clamp01(idx + smooth_factor)) * (float) n;
        const float m = (smax - smin) / 2.0f, rm = smin + m;
        BarDesc d{};
        d.tap_offset = (uint32_t) tap_w.size();
        float weight = 0.0f;
        bool first = true;
        uint32_t prev_bin = 0;
        for (float sx = smin; shape.inclusive ? sx <= smax : sx < smax; sx += 1.0f) {
            const float w = bar_round_formula(shape.round_formula, clamp01((m - fabsf(rm - sx)) / m));
            const uint32_t bin = (uint32_t) (int) roundf(sx);
            if (first) { d.first_bin = bin; first = false; }
            else if (bin != prev_bin + 1) {          // (never for step 1.0; keep the range contiguous regardless)
                for (uint32_t g = prev_bin + 1; g < bin; ++g) tap_w.push_back(0.0f);
            }
            prev_bin = bin;
            weight += w;
            tap_w.push_back(w);
        }
        d.count = (uint32_t) tap_w.size() - d.tap_offset;
        d.weight_sum = weight;
        // zero-pad to whole chunks: the kernels load a chunk's weights unconditionally
        while ((tap_w.size() - d.tap_offset) % chunk) tap_w.push_back(0.0f);
        desc[k] = d;
    }
}

// SAMPLE_MODE maximum / hybrid (glv_misc.hip glv_bars_mode_kernel): blocks of 64 bars, lane l = bar 64 b + l; the block's weights as [tap j][lane], +0 behind a
// bar's own taps up to the block's longest bar (and for bars past the last one)
inline void make_bar_mode_blocks(std::vector<BarModeBlock>& blocks, std::vector<float>& mw, const std::vector<BarDesc>& desc, const std::vector<float>& tap_w) {
    blocks.clear(); mw.clear();
    const uint32_t bars = (uint32_t) desc.size();
    for (uint32_t k0 = 0; k0 < bars; k0 += 64u) {
        BarModeBlock b{(uint32_t) mw.size(), 0u};
        for (uint32_t k = k0; k < k0 + 64u && k < bars; ++k) b.maxcount = desc[k].count > b.maxcount ? desc[k].count : b.maxcount;
        b.maxcount = (b.maxcount + kBarModeUnroll - 1u) / kBarModeUnroll * kBarModeUnroll;      // the kernel walks whole groups of taps
        mw.resize(mw.size() + (size_t) b.maxcount * 64u, 0.0f);
        for (uint32_t k = k0; k < k0 + 64u && k < bars; ++k)
            for (uint32_t j = 0; j < desc[k].count; ++j) mw[b.w_off + (size_t) j * 64u + (k - k0)] = tap_w[desc[k].tap_offset + j];
        blocks.push_back(b);
    }
}

// every chunk of every bar lies inside the row (the kernels read whole chunks: glv_frame.h bar_item_load)
inline bool bar_chunks_in_row(const std::vector<BarDesc>& desc, uint32_t n) {
    const uint32_t chunk = bar_chunk_of(n);
    for (const BarDesc& d : desc)
        if ((uint64_t) d.first_bin + (d.count + chunk - 1) / chunk * chunk > n) return false;
    return true;
}

// GLV_OP_BARS work lists for `groups` groups (of chunk / 8 lanes) per row: every bar's chunks go, in order, to one
// group (longest bars first, each to the least loaded group); step s of group g is items[s * groups + g].
// Lists are padded with all-zero-weight items (`zero_off`: `chunk` zeros in tap_w; they restart the running
// total and store it -- an exact 0 -- to the dump slot `bars`) to a multiple of `batch` steps, plus one more
// batch that the kernels' look-ahead reads.  Returns the step count.
inline uint32_t make_bar_items(std::vector<BarItem>& items, const std::vector<BarDesc>& desc, uint32_t groups, uint32_t zero_off, uint32_t chunk, uint32_t batch = kBarBatch) {
    std::vector<std::vector<BarItem>> list(groups);
    std::vector<uint32_t> order(desc.size());
    const uint32_t bars = (uint32_t) desc.size();
    for (uint32_t k = 0; k < bars; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return desc[a].count > desc[b].count; });
    for (uint32_t k : order) {
        uint32_t g = 0;
        for (uint32_t c = 1; c < groups; ++c) if (list[c].size() < list[g].size()) g = c;
        const BarDesc& d = desc[k];
        for (uint32_t i0 = 0; i0 < d.count; i0 += chunk)
            list[g].push_back(BarItem{(d.tap_offset + i0) * 4u, (d.first_bin + i0) * 4u, i0 + chunk >= d.count ? k : bars, i0 == 0 ? 0.0f : 1.0f});
    }
    uint32_t nsteps = 0;
    for (auto& l : list) nsteps = l.size() > nsteps ? (uint32_t) l.size() : nsteps;
    nsteps = (nsteps + batch - 1) / batch * batch;
    items.assign((size_t) (nsteps + batch) * groups, BarItem{zero_off * 4u, 0u, bars, 0.0f});
    for (uint32_t g = 0; g < groups; ++g)
        for (uint32_t s2 = 0; s2 < list[g].size(); ++s2) items[(size_t) s2 * groups + g] = list[g][s2];
    return nsteps;
}

// ---- many bars (bars >= kBarSeqMin: the pre-smoothing pass, bars == n) ------------------------------------------------------------
// From 256 bars up a bar is ONE fused-multiply-add chain over its taps in bin order, from +0:
//     acc = fma(w[b], x[b], acc),  b = first_bin .. first_bin + count - 1;   bar = acc / weight_sum
// (glv_frame.h "GLV_OP_BARS arithmetic").  A tap of weight +0 leaves the chain untouched (x is clamped to [0, 1]: 0 * x = +0,
// acc + 0 = acc), so a tile of 32 bars can run all its chains over ONE common bin range -- 32 bars x 64 rows x 2 bins are one
// v_mfma_f32_32x32x2_f32 pair (bit for bit a k-ordered fmaf chain on gfx950: tools/mfma_probe.hip), which is what
// glv_bars_rows_kernel does; glv_bars_seq_kernel walks the same tables with one lane per bar (few rows, or bars too long for the
// LDS ring).
//   mtiles[T]: bars [32 T, 32 T + 32): first bin `origin` (a multiple of 4), `steps` pairs of bins (a multiple of kBarStepPad: the
//              kernels run whole groups of steps; the padding weighs +0), where its weights start in wt
//   wt:        per tile and step 64 floats -- [bin of the pair 0..1][bar of the tile 0..31] -- exactly the b-operand of the MFMA,
//              one coalesced load per step; +0 outside a bar's own taps / for bars past the last one
//   wsum:      per bar {weight_sum, 1 / weight_sum}, padded with {1, 1} to whole tiles.  The reciprocal is 0 where the kernels must
//              divide the long way (bar_rcp_division_ok)
//   rounds:    consecutive tiles (at most tiles_per_round, one per wave) whose bins [origin, end) fit an LDS ring of `bins` bins
//              together with what the NEXT round adds to it: end(t) - origin(t) <= bins and end(t + 1) - origin(t) <= bins; empty
//              when some tile does not fit or the first bins are not monotone (glv_bars_rows_kernel is then not used)
//
// bar_rcp_division_ok: a / b as q0 = a * r, rem = fma(-q0, b, a), q = fma(rem, r, q0) with r = RN(1 / b) is the correctly rounded
// quotient (Markstein 1990) unless b's significand is all ones; a is a sum of [0, 1] texels times the weights, b their sum:
// no overflow, and the kernels take the long way themselves for 0 < a < 2^-90 (where rem would not be exact).  tests/test_emulator.py
// checks every significand of a against the weight sums of the shipped sizes.
constexpr uint32_t kBarSeqMin = 256, kBarTileBars = 32, kBarStepPad = 8, kBarLookAhead = 32;      // (steps; the rows kernel's weight look-ahead is at most kBarLookAhead)
inline bool bar_rcp_division_ok(float b) {
    const uint32_t u = __builtin_bit_cast(uint32_t, b);
    return b >= 0x1p-60f && b <= 0x1p60f && (u & 0x7fffffu) != 0x7fffffu;
}
// rounds of consecutive tiles for an LDS ring of `bins` bins (shared by make_bar_mtiles and make_bar_itiles): see make_bar_mtiles
template <class OriginOf, class EndOf>
inline bool cut_bar_rounds(std::vector<BarTile>& rounds, uint32_t nt, OriginOf origin_of, EndOf end_of, uint32_t n, uint32_t bins, uint32_t tiles_per_round) {
    rounds.clear();
    uint32_t T = 0, prev_origin = 0;
    bool fits = true;
    while (fits && T < nt) {
        BarTile r{T, T, origin_of(T), rounds.empty() ? 0u : rounds.back().end};       // k0, k1: TILE indices here; ends are kept monotone
        uint32_t H = T;
        while (H < nt && H - T < tiles_per_round) {
            const uint32_t e = end_of(H);
            if (e > n) { fits = false; break; }
            const uint32_t end = e > r.end ? e : r.end;
            if (end - r.origin > bins || (!rounds.empty() && end - prev_origin > bins)) break;
            // the tile after this one must be able to open the next round: its bins are added while this round still reads from r.origin
            if (H > T && H + 1 < nt && end_of(H + 1) - r.origin > bins) break;
            r.end = end;
            ++H;
        }
        if (H == T) fits = false;                                                // a single tile does not fit
        r.k1 = H;
        rounds.push_back(r);
        prev_origin = r.origin;
        T = H;
    }
    if (!fits) rounds.clear();
    return fits;
}
inline bool make_bar_mtiles(std::vector<BarMTile>& mtiles, std::vector<float>& wt, std::vector<float>& wsum, std::vector<BarTile>& rounds,
                            const std::vector<BarDesc>& desc, const std::vector<float>& tap_w, uint32_t n, uint32_t bins, uint32_t tiles_per_round) {
    mtiles.clear(); wt.clear(); wsum.clear(); rounds.clear();
    const uint32_t bars = (uint32_t) desc.size();
    if (bars < kBarSeqMin || tiles_per_round == 0) return false;
    const uint32_t nt = (bars + kBarTileBars - 1) / kBarTileBars;
    bool monotone = true;
    auto tile_end = [&](uint32_t H) {                                           // last bin + 1 of the tile's own taps, rounded up to a slot
        uint32_t e = 0;
        const uint32_t k1 = (H + 1) * kBarTileBars < bars ? (H + 1) * kBarTileBars : bars;
        for (uint32_t k = H * kBarTileBars; k < k1; ++k) e = desc[k].first_bin + desc[k].count > e ? desc[k].first_bin + desc[k].count : e;
        return (e + 3u) & ~3u;
    };
    for (uint32_t T = 0; T < nt; ++T) {
        const uint32_t k0 = T * kBarTileBars, k1 = k0 + kBarTileBars < bars ? k0 + kBarTileBars : bars;
        uint32_t lo = 0xffffffffu;
        for (uint32_t k = k0; k < k1; ++k) lo = desc[k].first_bin < lo ? desc[k].first_bin : lo;
        BarMTile t{k0, lo & ~3u, 0u, 0u};
        t.steps = (tile_end(T) - t.origin + 1u) / 2u;
        t.steps = (t.steps + kBarStepPad - 1u) / kBarStepPad * kBarStepPad;
        if (T && t.origin < mtiles[T - 1].origin) monotone = false;
        for (uint32_t j = 0; j < kBarTileBars; ++j) {
            const float ws = k0 + j < k1 ? desc[k0 + j].weight_sum : 1.0f;
            wsum.push_back(ws);
            wsum.push_back(bar_rcp_division_ok(ws) ? 1.0f / ws : 0.0f);
        }
        mtiles.push_back(t);
    }
    // rounds for the LDS ring (the padded steps read, with weight +0, whatever the ring holds there: only a tile's own bins must fit)
    if (monotone && bins % 4u == 0) cut_bar_rounds(rounds, nt, [&](uint32_t T) { return mtiles[T].origin; }, tile_end, n, bins, tiles_per_round);
    // the weights, laid out as the rows kernel consumes them: wave w of a workgroup takes tile k0 + w of every round, so its tiles
    // follow one another in memory -- ONE stream per wave, which the kernel reads a fixed number of steps ahead straight across tile
    // boundaries (without rounds: tile order)
    std::vector<uint32_t> order;
    if (!rounds.empty()) {
        for (uint32_t wv = 0; wv < tiles_per_round; ++wv)
            for (const BarTile& r : rounds)
                if (r.k0 + wv < r.k1) order.push_back(r.k0 + wv);
    } else {
        for (uint32_t i = 0; i < nt; ++i) order.push_back(i);
    }
    for (uint32_t Ti : order) {
        BarMTile& t = mtiles[Ti];
        const uint32_t k0 = t.k0, k1 = k0 + kBarTileBars < bars ? k0 + kBarTileBars : bars;
        t.w_off = (uint32_t) wt.size();
        for (uint32_t i = 0; i < 2u * t.steps; ++i)
            for (uint32_t j = 0; j < kBarTileBars; ++j) {
                const uint32_t k = k0 + j, bin = t.origin + i;
                wt.push_back(k < k1 && bin >= desc[k].first_bin && bin < desc[k].first_bin + desc[k].count ? tap_w[desc[k].tap_offset + bin - desc[k].first_bin] : 0.0f);
            }
    }
    wt.insert(wt.end(), 64u * kBarLookAhead, 0.0f);                             // what the look-ahead reads past the last tile
    return true;
}

// ---- many bars over TEXEL rows (the GL chains, gl_storage != 0: what the reference's pre-smoothing pass samples is a GL_R16 texture,
// render.c:2277-2303) -- exact integer arithmetic on the i8 matrix cores (glv_misc.hip glv_bars_rows_i8_kernel) ----------------------
// The texels c_j are 16-bit integers, so the weighted mean  sum w_j c_j / sum w_j  can be computed EXACTLY once the weights are integers:
//     ws = sum_j (double) w_j (tap order);   P = max(17, 21 + ceil(log2 ws))   (so that every W_j <= 2^22)
//     W_j = llrint(ldexp((double) w_j, P) / ws);   the first largest W_j takes the residue 2^P - sum W_j   =>  sum_j W_j == 2^P exactly
//     texel = floor(sum_j W_j c_j / 2^P + 1/2)          (round to nearest, a tie -- an exact half -- goes up; OpenGL 4.6 2.3.5 leaves it open)
//     float form (bars not as texels): (float) ((double) sum_j W_j c_j * 2^-P / 65535.0)
// w_j are the shader's float weights (make_bar_taps).  The integer weights are within 2^-P (relative to their sum: <= 2^-22 of the largest)
// of the float ones, so the result lies within 65535 / 2^21 = 0.03 texel steps of the mean with the float weights in exact arithmetic
// (typically 0.002) -- closer to it than any float summation order is (tests/test_gl_reference.py holds it to the reference's own
// llvmpipe texels tie-aware, like every other form).  A bar whose weights sum to 0 (0 / 0 in the shader) is texel 0 / float NaN.
// On the device c - 32896 = 256 h + l and W = 65536 w2 + 256 w1 + w0 in balanced signed bytes: six v_mfma_i32_32x32x32_i8 per
// 32 rows x 32 bars x 32 bins into four accumulators (2^24: h w2; 2^16: h w1 + l w2; 2^8: h w0 + l w1; 2^0: l w0), all exact.
//   itiles[T]: bars [32 T, 32 T + 32): origin (a multiple of 16 bins), steps of 32 bins, w_off = first 16-byte vector of its weights in wq
//   wq:        per tile and step [digit 0..2][lane 0..63][16 bytes]: lane l, byte j = digit of W for bar k0 + l % 32 at bin
//              origin + 32 step + 16 (l / 32) + j (0 outside the bar's own taps) -- the b-operand of the MFMA, one coalesced load per digit
//   fin[k]:    {c, s}: texel = (uint32) (floor(T / 2^16) + c) >> s with s = P - 16 in [1, 15] and c = 32896 * 2^s + 2^(s-1); weights that sum to 0: {0, kBarIFinNone}, every digit 0
//   rounds:    as make_bar_mtiles (tile ends rounded up to 8 bins: the ring is filled 8 texels = 16 bytes at a time)
// bins per step; steps of zeros behind the last tile: the kernel requests a wave's weights PF = 3 steps ahead of the step it runs, so the
// stream that lies last in memory is read up to three steps past its last tile (glv_misc.hip glv_bars_rows_i8_kernel)
constexpr uint32_t kBarIStepBins = 32;                    // (kBarILookAhead: glv_frame.h, shared with the kernel)
// the integer weights of one bar (W: count values); returns P, or -1 when the float weights sum to 0 / NaN, or -2 when P would exceed 31
inline int bar_int_weights(const float* w, uint32_t count, std::vector<int32_t>& W) {
    W.assign(count, 0);
    double ws = 0.0;
    for (uint32_t j = 0; j < count; ++j) ws += (double) w[j];
    if (!(ws > 0.0) || !(ws < 1e30)) return -1;
    int ex = 0;
    const double f = frexp(ws, &ex);                       // ws = f 2^ex, f in [0.5, 1)
    int P = 21 + (f == 0.5 ? ex - 1 : ex);                 // 21 + ceil(log2 ws)
    if (P < 17) P = 17;
    if (P > 31) return -2;
    int64_t sum = 0;
    uint32_t jmax = 0;
    for (uint32_t j = 0; j < count; ++j) {
        W[j] = (int32_t) llrint(ldexp((double) w[j], P) / ws);
        sum += W[j];
        if (W[j] > W[jmax]) jmax = j;
    }
    W[jmax] += (int32_t) (((int64_t) 1 << P) - sum);
    return P;
}
inline void bar_int_digits(int32_t W, int8_t d[3]) {       // W = 65536 d2 + 256 d1 + d0, every digit in [-128, 127]
    const int32_t d0 = ((W + 128) & 255) - 128, W1 = (W - d0) >> 8, d1 = ((W1 + 128) & 255) - 128, d2 = (W1 - d1) >> 8;
    d[0] = (int8_t) d0; d[1] = (int8_t) d1; d[2] = (int8_t) d2;
}
inline bool make_bar_itiles(std::vector<BarMTile>& itiles, std::vector<int8_t>& wq, std::vector<BarIFin>& fin, std::vector<BarTile>& rounds,
                            const std::vector<BarDesc>& desc, const std::vector<float>& tap_w, uint32_t n, uint32_t bins, uint32_t tiles_per_round) {
    itiles.clear(); wq.clear(); fin.clear(); rounds.clear();
    const uint32_t bars = (uint32_t) desc.size();
    if (bars < kBarSeqMin || tiles_per_round == 0 || bins % 32u) return false;
    const uint32_t nt = (bars + kBarTileBars - 1) / kBarTileBars;
    // the integer weights of every bar
    std::vector<std::vector<int32_t>> W(bars);
    fin.assign((size_t) nt * kBarTileBars, BarIFin{0u, kBarIFinNone});
    for (uint32_t k = 0; k < bars; ++k) {
        const int P = bar_int_weights(tap_w.data() + desc[k].tap_offset, desc[k].count, W[k]);
        if (P == -2) return false;
        if (P < 0) continue;                                                    // weights sum to 0: {0, kBarIFinNone}, every digit 0
        // bar_int_digits splits into three SIGNED bytes: |W| must stay below 2^23 - 2^16 (true while every shader weight is <= 1 -- smooth.glsl's window
        // functions are -- and its share of the sum <= 1; a weight function that breaks it takes the f32 form, announced by glv_batch_bars_arithmetic)
        for (const int32_t x : W[k]) if (x > 127 * 65536 || x < -127 * 65536) return false;   // d2 in [-127, 127] with room for the carries of d0, d1
        const uint32_t s = (uint32_t) P - 16u;
        fin[k] = BarIFin{(32896u << s) + (1u << (s - 1u)), s};
    }
    bool monotone = true;
    auto tile_end = [&](uint32_t H) {                                           // last bin + 1 of the tile's own taps, rounded up to a fill unit
        uint32_t e = 0;
        const uint32_t k1 = (H + 1) * kBarTileBars < bars ? (H + 1) * kBarTileBars : bars;
        for (uint32_t k = H * kBarTileBars; k < k1; ++k) e = desc[k].first_bin + desc[k].count > e ? desc[k].first_bin + desc[k].count : e;
        return (e + 7u) & ~7u;
    };
    for (uint32_t T = 0; T < nt; ++T) {
        const uint32_t k0 = T * kBarTileBars, k1 = k0 + kBarTileBars < bars ? k0 + kBarTileBars : bars;
        uint32_t lo = 0xffffffffu;
        for (uint32_t k = k0; k < k1; ++k) lo = desc[k].first_bin < lo ? desc[k].first_bin : lo;
        BarMTile t{k0, lo & ~15u, 0u, 0u};
        if (lo == 0xffffffffu || tile_end(T) <= t.origin) return false;         // a tile without taps: the kernel's step countdown starts at steps >= 1
        t.steps = (tile_end(T) - t.origin + kBarIStepBins - 1u) / kBarIStepBins;
        if (T && t.origin < itiles[T - 1].origin) monotone = false;
        itiles.push_back(t);
    }
    if (monotone) cut_bar_rounds(rounds, nt, [&](uint32_t T) { return itiles[T].origin; }, tile_end, n, bins, tiles_per_round);
    std::vector<uint32_t> order;                                                // one stream per wave, as make_bar_mtiles
    if (!rounds.empty()) {
        for (uint32_t wv = 0; wv < tiles_per_round; ++wv)
            for (const BarTile& r : rounds)
                if (r.k0 + wv < r.k1) order.push_back(r.k0 + wv);
    } else {
        for (uint32_t i = 0; i < nt; ++i) order.push_back(i);
    }
    for (uint32_t Ti : order) {
        BarMTile& t = itiles[Ti];
        const uint32_t k0 = t.k0, k1 = k0 + kBarTileBars < bars ? k0 + kBarTileBars : bars;
        t.w_off = (uint32_t) (wq.size() / 16u);
        wq.resize(wq.size() + (size_t) t.steps * 3u * 64u * 16u, 0);
        int8_t* base = wq.data() + (size_t) t.w_off * 16u;
        for (uint32_t k = k0; k < k1; ++k)
            for (uint32_t i = 0; i < desc[k].count; ++i) {
                const uint32_t rel = desc[k].first_bin + i - t.origin, step = rel / kBarIStepBins, half = (rel % kBarIStepBins) / 16u, j = rel % 16u;
                int8_t d[3];
                bar_int_digits(W[k][i], d);
                for (uint32_t q = 0; q < 3; ++q) base[(((size_t) step * 3u + q) * 64u + half * 32u + (k - k0)) * 16u + j] = d[q];
            }
    }
    wq.resize(wq.size() + (size_t) kBarILookAhead * 3u * 64u * 16u, 0);         // what the look-ahead reads past the last tile
    return true;
}

// log_mode 0 table (glv_core.h log_third_table): c_j = 1 + j / 2^bits, { 2^-23 / c_j, log(c_j)/3 }.
inline void make_log_table(LogEntry* t, int bits = kLogTabMaxBits) {
    for (int j = 0; j < (1 << bits); ++j) {
        const double c = 1.0 + (double) j / (double) (1 << bits);
        t[j].inv_c = ldexp(1.0 / c, -23);      // exact scaling: r = k * inv_c for the integer k = (m - c) * 2^23
        t[j].log_c3 = log(c) / 3.0;
    }
}

}  // namespace glv
