// glv_misc.hip -- the small kernels and the size dispatch.
//
//   glv_post_kernel    gravity / average / wrange on spectra already in HBM (the single-op
//                      drop-ins glv_gravity, glv_average, glv_wrange; glava/render.c:720-781) and
//                      the magnitude stage alone (glv_magnitude; render.c:842-846)
//   glv_unpack_kernel  s16 interleaved -> planar f32 (glv_unpack_s16; glava/fifo.c:94-110)
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "glv_frame.h"
#include "glv_launch.h"
#include "glv_winsplit.h"

namespace glv {

// One lane owns one pair of floats (8 B) of a row and runs the same state machine as the
// fused epilogue (apply_state).
__global__ void __launch_bounds__(256) glv_post_kernel(const FrameArgs a, const uint32_t n) {
    const size_t pairs_per_row = n / 2;
    const size_t total = (size_t) a.units * pairs_per_row;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t) gridDim.x * blockDim.x) {
        const size_t row = i / pairs_per_row;
        const uint32_t off = (uint32_t) (i % pairs_per_row) * 8u;     // byte offset of the pair in its row
        cf val = ld<cf>(static_cast<const float*>(a.in) + row * n, off);
        if (a.ops & OP_MAGNITUDE) {                                               // render.c:842-846
            const float y0 = __builtin_fabsf(val.x) + 1.0f, y1 = __builtin_fabsf(val.y) + 1.0f;
            const cf tl = ld<cf>(a.tilt, off);
            if (a.log_mode == 0)      { val.x = log_third_nf<0, true>(y0, a.logtab) * tl.x; val.y = log_third_nf<0, true>(y1, a.logtab) * tl.y; }
            else if (a.log_mode == 1) { val.x = log_third<1>(y0, a.logtab, kLogTabMaxBits) * tl.x; val.y = log_third<1>(y1, a.logtab, kLogTabMaxBits) * tl.y; }
            else                      { val.x = log_third<2>(y0, a.logtab, kLogTabMaxBits) * tl.x; val.y = log_third<2>(y1, a.logtab, kLogTabMaxBits) * tl.y; }
        }
        if (a.ops & OP_WRANGE) {                                                  // render.c:777-779
            const float p = val.x + 1.0f, q = val.y + 1.0f;
            val.x = p / 2.0f; val.y = q / 2.0f;
        }
        val = apply_state(val, off, row, n, a);
        if (a.out) {
            if (a.ops & OP_R16) st<uint32_t>(reinterpret_cast<uint16_t*>(a.out) + row * n, off / 2u, pack_unorm16(val.x, val.y));   // render.c:521-524
            else st<cf>(a.out + row * n, off, val);
        }
    }
}

__global__ void __launch_bounds__(256) glv_unpack_kernel(const int16_t* __restrict__ pcm, size_t frames, int mono,
                                                         float* __restrict__ l, float* __restrict__ r) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += (size_t) gridDim.x * blockDim.x) {
        const uint32_t u = reinterpret_cast<const uint32_t*>(pcm)[i];
        const int a = (int16_t) (u & 0xffffu), b = (int16_t) (u >> 16);
        if (mono) { const float s = unpack_s16_mono(a, b); l[i] = s; r[i] = s; }
        else { l[i] = unpack_s16(a); r[i] = unpack_s16(b); }
    }
}

// The device rings as the reference's backends publish them in audio_out_l / audio_out_r (glava/fifo.h:9-20): planar f32,
// oldest sample first (fifo.c:91-92 / pulse_input.c:155-156 keep that order by memmove; the device rings are circular and
// `rot` is the index of their oldest frame).  s16 ring: the unpack of fifo.c:94-110; f32 ring: the deinterleave of
// pulse_input.c:159-176; mono: the respective (L + R) / 2 into both outputs.
__global__ void __launch_bounds__(256) glv_ring_planar_kernel(const void* __restrict__ ring, int is_f32, uint32_t n, uint32_t rot, int mono,
                                                              size_t streams, float* __restrict__ out) {
    const size_t total = streams * n;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t) gridDim.x * blockDim.x) {
        const size_t s = i / n;
        const uint32_t t = (uint32_t) (i % n);
        const size_t src = s * n + ((t + rot) & (n - 1));
        float l, r;
        if (is_f32) {
            const cf u = static_cast<const cf*>(ring)[src];
            if (mono) { l = (u.x + u.y) / 2; r = l; } else { l = u.x; r = u.y; }              // pulse_input.c:167
        } else {
            const uint32_t u = static_cast<const uint32_t*>(ring)[src];
            const int a = (int16_t) (u & 0xffffu), b = (int16_t) (u >> 16);
            if (mono) { l = unpack_s16_mono(a, b); r = l; } else { l = unpack_s16(a); r = unpack_s16(b); }
        }
        out[(s * 2) * n + t] = l;
        out[(s * 2 + 1) * n + t] = r;
    }
}

// ---- rd_update prelude (glava/render.c:1765-1809) ------------------------------------------------
// bufscale: mean of k consecutive samples, float accumulation in index order, one float division
__global__ void __launch_bounds__(256) glv_bufscale_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           size_t total_out, uint32_t k) {
    const float fk = (float) k;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total_out; i += (size_t) gridDim.x * blockDim.x) {
        float accum = 0.0f;
        for (uint32_t a = 0; a < k; ++a) accum = accum + in[i * k + a];     // rows are n_out*k long: i*k stays inside the row
        out[i] = accum / fk;
    }
}
// keyframe interpolation: s + (e - s) * mod, mod = min(uratio * kcounter, 1) computed on the host
__global__ void __launch_bounds__(256) glv_lerp_kernel(const float* __restrict__ s0, const float* __restrict__ e0,
                                                       float* __restrict__ out, size_t total, float mod) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t) gridDim.x * blockDim.x) {
        const float d = e0[i] - s0[i];
        const float p = d * mod;
        out[i] = s0[i] + p;
    }
}

// ---- CPU-path transform_smooth (glava/render.c:694-718) --------------------------------------------
// In place and sequentially dependent inside a row (output t reads inputs that earlier outputs already
// replaced), so one lane walks one row; rows are independent.  smin/smax depend only on t and come from
// the host (powf/log/floor/ceil of the reference's libm, glv_tables.h).
//
// The walk only ever touches the first `reach` floats of a row (reach = max smax + 1, about 1.01 n / smooth_ratio) and
// writes the first `asz`.  A workgroup (one wave) therefore stages that prefix of R rows in LDS with coalesced loads --
// row j at float offset j * stride, stride odd, so the 64 lanes' accesses to the same index of their own rows fall in
// different banks --, lanes 0..R-1 walk their rows there (every tap an LDS read instead of a strided global one: round 2
// measured 29 ms for 131 072 rows of N=4096 with the walk in global memory, every lane on its own cache line), and the
// first asz floats of each row go back coalesced.  R = as many rows as fit the workgroup's LDS (host: launch_smooth).
__global__ void __launch_bounds__(64) glv_smooth_kernel(float* __restrict__ rows, size_t nrows, uint32_t n,
                                                        const int* __restrict__ smin, const int* __restrict__ smax, uint32_t asz,
                                                        uint32_t reach, uint32_t rows_per_wg) {
    extern __shared__ float smooth_lds[];
    const uint32_t stride = reach | 1u;
    const uint32_t lane = threadIdx.x;
    const size_t row0 = (size_t) blockIdx.x * rows_per_wg;
    if (row0 >= nrows) return;
    const uint32_t R = (uint32_t) (nrows - row0 < rows_per_wg ? nrows - row0 : rows_per_wg);
    // staging: eight loads of a lane in flight before the first is parked in LDS (one at a time, every 256-byte piece
    // of a row was its own exposed HBM round trip)
    for (uint32_t j = 0; j < R; ++j) {
        const float* src = rows + (row0 + j) * n;
        float* dst = smooth_lds + (size_t) j * stride;
        for (uint32_t i0 = lane; i0 < reach; i0 += 64 * 8) {
            float tmp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) tmp[k] = i0 + 64u * k < reach ? src[i0 + 64u * k] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (i0 + 64u * k < reach) dst[i0 + 64u * k] = tmp[k];
        }
    }
    __syncthreads();
    if (lane < R) {
        float* b = smooth_lds + (size_t) lane * stride;
        // the bounds of eight steps are fetched together (uniform: scalar loads) -- one memory round trip per eight
        // outputs instead of two per output, which is what a step of the first LDS version waited for
        for (uint32_t t0 = 0; t0 < asz; t0 += 8) {
            int lo[8], hi[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = t0 + k < asz ? t0 + k : asz - 1;
                lo[k] = smin[t]; hi[k] = smax[t];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (t0 + k >= asz) break;
                float avg = 0.0f;
                int count = 0;
                const int q1 = hi[k];
                // eight taps per trip, read together; taps past smax read as 0, which the reference's `if (b[s])` skips anyway
                for (int q0 = lo[k]; q0 <= q1; q0 += 8) {
                    float x[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = q0 + i <= q1 ? b[q0 + i] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (x[i] != 0.0f) { avg = avg + x[i]; ++count; }   // `if (b[s])`: NaN counts, +-0 does not
                }
                b[t0 + k] = avg / (float) count;                    // 0/0 = NaN at t = 0, as in the reference
            }
        }
    }
    __syncthreads();
    for (uint32_t j = 0; j < R; ++j) {
        float* dst = rows + (row0 + j) * n;
        const float* src = smooth_lds + (size_t) j * stride;
        for (uint32_t i = lane; i < asz; i += 64) dst[i] = src[i];
    }
}

// The same walk for many rows at once: a wave takes 64 rows, one per lane, and keeps only a sliding window of each in LDS
// -- a ring of W floats per lane (position q lives in slot q mod W; lane stride W + 1, odd: the lanes' accesses to the
// same slot of their own rows hit different banks).  Inputs enter in chunks of kSmoothChunk positions (coalesced: half a
// wave reads one row's chunk), finished outputs leave in chunks of the same size, the walk in between reads and writes
// LDS only.  W >= (largest window) + 2 chunks: a slot is reused for position p + W only after p has left every later
// window (smin is monotone) and, being a finished output, has been written back.  33 KiB of LDS per wave at W = 128
// (the defaults at N=4096: largest window 23 taps) instead of one 4 KiB row prefix per lane: 4 waves = 256 rows in
// flight per CU against 30, which is what a latency-bound dependent walk needs.
constexpr uint32_t kSmoothChunk = 32;
template <uint32_t W>
__global__ void __launch_bounds__(64) glv_smooth_ring_kernel(float* __restrict__ rows, size_t nrows, uint32_t n,
                                                             const int* __restrict__ smin, const int* __restrict__ smax, uint32_t asz,
                                                             uint32_t reach) {
    extern __shared__ float smooth_lds[];
    constexpr uint32_t LS = W + 1;                         // lane stride (floats)
    constexpr uint32_t CH = kSmoothChunk;
    const uint32_t lane = threadIdx.x;
    const size_t row0 = (size_t) blockIdx.x * 64;
    if (row0 >= nrows) return;
    const uint32_t R = (uint32_t) (nrows - row0 < 64 ? nrows - row0 : 64);
    const uint32_t sub = lane & (CH - 1), half = lane / CH;           // two rows per load / store instruction
    uint32_t loaded = 0, written = 0;
    // positions [c0, c1) (c1 - c0 <= CH) of every row: HBM -> ring, eight row pairs in flight
    auto load_chunk = [&](uint32_t c0, uint32_t c1) {
        const uint32_t pos = c0 + sub;
        for (uint32_t j0 = 0; j0 < R; j0 += 16) {
            float tmp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t j = j0 + 2u * k + half;
                tmp[k] = (j < R && pos < c1) ? rows[(row0 + j) * n + pos] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t j = j0 + 2u * k + half;
                if (j < R && pos < c1) smooth_lds[j * LS + (pos & (W - 1))] = tmp[k];
            }
        }
    };
    // finished outputs [c0, c1) of every row: ring -> HBM
    auto store_chunk = [&](uint32_t c0, uint32_t c1) {
        const uint32_t pos = c0 + sub;
        for (uint32_t j0 = 0; j0 < R; j0 += 2) {
            const uint32_t j = j0 + half;
            if (j < R && pos < c1) rows[(row0 + j) * n + pos] = smooth_lds[j * LS + (pos & (W - 1))];
        }
    };
    auto wave_sync = [&]() {                               // the workgroup is one wave: LDS is in order, the compiler must be too
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    float* b = smooth_lds + (size_t) lane * LS;
    for (uint32_t t0 = 0; t0 < asz; t0 += 8) {
        int lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t t = t0 + k < asz ? t0 + k : asz - 1;
            lo[k] = smin[t]; hi[k] = smax[t];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t t = t0 + k;
            if (t >= asz) break;
            // outputs that are complete chunks leave first (they also free the slots the next inputs take)
            while (t - written >= CH) { wave_sync(); store_chunk(written, written + CH); written += CH; }
            // inputs up to this step's last tap
            // ... and up to t itself: the output takes position t's slot, which no later chunk may then overwrite
            uint32_t need = hi[k] + 1 > (int) t + 1 ? (uint32_t) (hi[k] + 1) : t + 1;
            need = need < reach ? need : reach;
            while (loaded < need) {
                const uint32_t c1 = loaded + CH < reach ? loaded + CH : reach;
                wave_sync();
                load_chunk(loaded, c1);
                loaded = c1;
            }
            wave_sync();
            if (lane < R) {
                float avg = 0.0f;
                int count = 0;
                const int q1 = hi[k];
                for (int q0 = lo[k]; q0 <= q1; q0 += 8) {
                    float x[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = q0 + i <= q1 ? b[(uint32_t) (q0 + i) & (W - 1)] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (x[i] != 0.0f) { avg = avg + x[i]; ++count; }   // `if (b[s])`: NaN counts, +-0 does not
                }
                b[t & (W - 1)] = avg / (float) count;                   // 0/0 = NaN at t = 0, as in the reference
            }
        }
    }
    wave_sync();
    while (written < asz) { const uint32_t c1 = written + CH < asz ? written + CH : asz; store_chunk(written, c1); written = c1; }
}

// ---- smooth_audio() bar sampling (shaders/glava/util/smooth.glsl:13-40, radial/1.frag:58-70) --------
// The tap positions and weights of a bar depend only on (bar, n, smooth_factor) -- not on the data -- so
// they are generated once per batch on the host (glv_tables.h make_bar_taps: SAMPLE_MODE average,
// ROUND_FORMULA sinusoidal, SAMPLE_SCALE 8, SAMPLE_RANGE 0.9) together with the work lists
// (make_bar_items).  One 256-thread workgroup = 256 / GL groups of GL = bar_lanes_of(n) lanes per row; arithmetic: glv_frame.h.
// r16: bars_out is uint16 [nrows][bars], the GL_R16 texel of every value (what the reference's smooth pass renders into,
// render.c:2277-2303 with bind_1d_fbo's GL_R16 texture) instead of float
template <int GL>
__global__ void __launch_bounds__(256) glv_bars_kernel(const float* __restrict__ spec, float* __restrict__ bars_out,
                                                       size_t nrows, uint32_t n, uint32_t bars, uint32_t nsteps,
                                                       const BarItem* __restrict__ items, const BarDesc* __restrict__ desc,
                                                       const float* __restrict__ tap_w, int r16) {
    constexpr uint32_t G = 256 / GL;
    const int sub = threadIdx.x & (GL - 1);
    const uint32_t g = threadIdx.x / GL;
    for (size_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const float* tex = spec + row * n;
        float total = 0.0f;
        for (uint32_t s0 = 0; s0 < nsteps; s0 += kBarBatch) {
            BarItem it[kBarBatch];
            BarTaps tp[kBarBatch];
#pragma unroll
            for (int b = 0; b < kBarBatch; ++b) it[b] = items[(size_t) (s0 + b) * G + g];
#pragma unroll
            for (int b = 0; b < kBarBatch; ++b) tp[b] = bar_item_load(tex, tap_w, it[b], sub);
#pragma unroll
            for (int b = 0; b < kBarBatch; ++b) {
                total = __builtin_fmaf(total, it[b].keep, group_sum<GL>(bar_item_lane_sum(tp[b])));
                const uint32_t k = it[b].res;
                if (k != bars && sub == 0) {
                    const float v = total / desc[k].weight_sum;
                    if (r16) reinterpret_cast<uint16_t*>(bars_out)[row * bars + k] = (uint16_t) unorm16(v);
                    else bars_out[row * bars + k] = v;
                }
            }
        }
    }
}

// The same for short work lists (nsteps == NS: 80 bars of a row up to N=4096 are 2-4 steps of the 32 groups): the group's
// items, the lane's weights and the weight sums do not depend on the row, so they are fetched ONCE per workgroup and stay in
// registers; a row then costs one round trip (its taps: 2 NS 16-byte loads per lane) instead of a chain of three (item ->
// weights / taps per batch), and RI rows are in flight per workgroup trip.  N=1024 x 262144 rows: 2.1 -> 0.41 ms.
template <int NS, int RI, int GL>
__global__ void __launch_bounds__(256) glv_bars_short_kernel(const float* __restrict__ spec, float* __restrict__ bars_out,
                                                             size_t nrows, uint32_t n, uint32_t bars,
                                                             const BarItem* __restrict__ items, const BarDesc* __restrict__ desc,
                                                             const float* __restrict__ tap_w, int r16) {
    constexpr uint32_t G = 256 / GL;
    const int sub = threadIdx.x & (GL - 1);
    const uint32_t g = threadIdx.x / GL;
    const uint32_t lane_byte = 4u * (uint32_t) kBarTaps * (uint32_t) sub;
    BarItem it[NS];
    BarTaps tw[NS];              // .w: the lane's weights of step s (.t unused)
    float wsum[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) it[s] = items[(size_t) s * G + g];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int h = 0; h < kBarTaps / 4; ++h) {
            const BarW4 w4 = ld<BarW4>(tap_w, it[s].w_byte + lane_byte + 16u * (uint32_t) h);
#pragma unroll
            for (int i = 0; i < 4; ++i) tw[s].w[4 * h + i] = w4.w[i];
        }
        wsum[s] = it[s].res != bars ? desc[it[s].res].weight_sum : 1.0f;
    }
    for (size_t row0 = (size_t) blockIdx.x * RI; row0 < nrows; row0 += (size_t) gridDim.x * RI) {
        BarT4 t[RI][NS][kBarTaps / 4];
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const size_t row = row0 + i < nrows ? row0 + i : nrows - 1;
            const char* tex = reinterpret_cast<const char*>(spec + row * n);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int h = 0; h < kBarTaps / 4; ++h)
                    t[i][s][h] = *reinterpret_cast<const BarT4*>(tex + it[s].tex_byte + lane_byte + 16u * (uint32_t) h);
        }
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const size_t row = row0 + i;
            float total = 0.0f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                BarTaps tp = tw[s];
#pragma unroll
                for (int q = 0; q < kBarTaps; ++q) tp.t[q] = t[i][s][q / 4].t[q % 4];
                total = __builtin_fmaf(total, it[s].keep, group_sum<GL>(bar_item_lane_sum(tp)));
                const uint32_t k = it[s].res;
                if (k != bars && sub == 0 && row < nrows) {
                    const float v = total / wsum[s];
                    if (r16) reinterpret_cast<uint16_t*>(bars_out)[row * bars + k] = (uint16_t) unorm16(v);
                    else bars_out[row * bars + k] = v;
                }
            }
        }
    }
}

// The same bars for MANY bars and MANY rows (the pre-smoothing pass of render.c:2277-2303: bars == n, one output per texel, ~58
// taps each at n = 4096: 237 K multiply-adds per row): ONE LANE PER ROW, EIGHT BARS PER WAVE AT A TIME.
//   * The taps and weights of a bar do not depend on the row, so with 64 rows side by side in a wave every weight is wave-uniform: a
//     SCALAR register, streamed through the scalar cache (s_load_dwordx16) -- no vector register, no vector load, no broadcast.
//   * The eight bars of a group start at one bin (glv_tables.h make_bar_taps, "group rule"), so they walk the same texels in the same
//     octets: the lane reads an octet of its row ONCE from LDS (two ds_read_b128) and uses it for eight bars.
//   * A tap pair is one v_pk_fma_f32 for 64 rows: {even chain, odd chain} += {x[2i], x[2i+1]} * {w[2i], w[2i+1]} with the weight pair as
//     an SGPR-pair operand -- 4 packed instructions per octet and bar, + 1 add for the octet sum, + the tree.
//     (Round 4's first version took one bar per wave, one v_fmac_f32_dpp and half a ds_read2st64_b32 per tap and row: LDS-bandwidth
//     bound at 1.33 ms for 32 K rows of N = 4096 -- profiles/r04/rows_parts.txt: 1.18 ms of it with no HBM traffic at all.)
// A workgroup is eight waves on the SAME 64 rows.  The rows' texels live in LDS as a RING of S bins, [bin / 4 mod S / 4][row][4]
// (conflict-free b128 accesses both ways: the fill's global loads deliver exactly one such slot).  The bars' first bins grow slowly
// (4096 bars cover 1250 bins), so a ROUND -- eight groups, wave w takes group w (host table, glv_tables.h make_bar_groups) -- needs
// only a few bins the previous round did not have: they are requested before the round's arithmetic and written behind it, into
// slots no wave of the round reads (the table guarantees end(t + 1) - origin(t) <= S).  Every texel is read from HBM / L2 once per
// range of rounds.  The round's 64 results per row are parked in LDS and leave as contiguous row segments (measured: storing a
// group's eight results per row straight from the registers -- 64 lines per store instruction, rows a power of two apart -- costs
// 0.79 instead of 0.61 ms at N = 4096).  Two workgroups per CU = four waves per SIMD cover the scalar and LDS latencies (an s_load
// result needs lgkmcnt(0): the prefetch distance is half a step, the rest is the other waves).  What bounds the kernel is that
// scalar stream: 256 B per octet step and wave, 0.56 GB per launch at N = 4096 x 32 K rows, every half step an L2 round trip of
// ~1300 cycles (profiles/r04/rows_kernel.txt: the vector ALU is busy 40 % of the time, waves wait on lgkmcnt; the loads do not
// hit in the scalar cache -- made to read one address they queue at one L2 channel and the kernel is twice as slow).
// The summation order is the documented one (glv_frame.h "GLV_OP_BARS arithmetic": chunks of 16 / 32 / 64 taps, per chunk 2 / 4 / 8
// octet sums, each the sum of two fused-multiply-add chains, combined pairwise, chunk totals in order), walked octet by octet with a
// three-deep stack of partial sums: the same bits as glv_bars_kernel and the fused epilogue.  The final division by the bar's weight
// sum is three instructions with the host's reciprocal (glv_tables.h bar_rcp_division_ok: the correctly rounded quotient).
constexpr int kRowsWaves = 8, kRowsTileBars = 64, kRowsStagePitch = 65;     // N = 4096: 2 x (60 KiB ring + 16.3 KiB stage) fit a CU's 160 KiB
template <int S, int GL>
__global__ void __launch_bounds__(64 * kRowsWaves) glv_bars_rows_kernel(const float* __restrict__ spec, void* __restrict__ bars_out, size_t nrows, uint32_t n,
                                                                        uint32_t bars, const BarTile* __restrict__ tiles, uint32_t ntiles, uint32_t tiles_per_wg,
                                                                        const BarGroupDesc* __restrict__ groups, const float* __restrict__ wg,
                                                                        const float* __restrict__ wsum, int r16) {
#if defined(__HIP_DEVICE_COMPILE__)                     /* packed-f32 inline assembly: the host pass sees an empty stub */
    extern __shared__ float rows_lds[];                 // [S / 4][64] x 4 texels: the ring | [64][65] finished outputs of the round
    constexpr uint32_t NS = S / 4;                      // slots
    static_assert(S % 8 == 0 && kRowsTileBars == 8 * kRowsWaves, "a wave per group of eight bars; a step's two slots never straddle the ring's end");
    BarW4* win = reinterpret_cast<BarW4*>(rows_lds);
    float* stage = rows_lds + (size_t) S * 64;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const size_t row0 = (size_t) blockIdx.x * 64;
    if (row0 >= nrows) return;
    const uint32_t R = (uint32_t) (nrows - row0 < 64 ? nrows - row0 : 64);
    const float* src = spec + (row0 + (lane < R ? lane : R - 1)) * (size_t) n;
    const uint32_t t_begin = blockIdx.y * tiles_per_wg, t_end = t_begin + tiles_per_wg < ntiles ? t_begin + tiles_per_wg : ntiles;
    const glv_f2 ones = {1.0f, 1.0f};
    BarW4* win_lane = win + lane;
    struct HalfW { glv_f2 w[16]; };                     // a half step: bars 4 h .. 4 h + 3 of the group x 4 tap pairs, 32 SGPRs
    auto load_half = [&](const float* wp) {
        HalfW h;
        const glv_f2* p = reinterpret_cast<const glv_f2*>(wp);
#pragma unroll
        for (int q = 0; q < 16; ++q) h.w[q] = p[q];
        return h;
    };
    // 4 bins of this lane's row: one slot of the ring; clamped here, once per texel, instead of once per tap: [0, 1] like the GL_R16
    // texel the shader samples, NaN -> 0 (v_pk_mul_f32 x, 1.0 clamp -- the operation bar_item_lane_sum applies to every tap: same bits)
    auto fetch = [&](uint32_t bin) {
#if defined(GLV_EXP_ROWS_NOFILL)        /* timing experiment (wrong results): no row loads */
        return BarW4{{(float) bin, 0.5f, 0.25f, (float) lane}};
#else
        return ld<BarW4>(src, bin * 4u);
#endif
    };
    auto park = [&](const BarW4& v, uint32_t bin) {
        glv_f2 lo = {v.w[0], v.w[1]}, hi = {v.w[2], v.w[3]};
        asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(lo) : "v"(lo), "v"(ones));
        asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(hi) : "v"(hi), "v"(ones));
        win_lane[(size_t) ((bin / 4u) % NS) * 64] = BarW4{{lo.x, lo.y, hi.x, hi.y}};
    };
    if (t_begin >= t_end) return;
    // the first round's whole window: four loads of a wave are in flight before the first is parked
    uint32_t filled_to;
    {
        const BarTile T = tiles[t_begin];
        const uint32_t ncol = (T.end - T.origin) / 4u;
        for (uint32_t cb = wave * 4u; cb < ncol; cb += kRowsWaves * 4) {
            BarW4 v4[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) v4[q] = fetch(T.origin + 4u * (cb + q < ncol ? cb + q : cb));
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                if (cb + q < ncol) park(v4[q], T.origin + 4u * (cb + q));
        }
        filled_to = T.end;
    }
    __syncthreads();
    for (uint32_t t = t_begin; t < t_end; ++t) {
        const BarTile T = tiles[t];                                             // uniform: scalar loads
        const bool valid = T.k0 + 8u * wave < T.k1;
        const BarGroupDesc g = groups[valid ? T.k0 / 8u + wave : T.k0 / 8u];
        // what the next round adds to the ring: requested now, parked behind this round's arithmetic (up to two slots per wave in
        // registers, which is what a round adds at most in practice; the rest after them)
        const uint32_t next_end = t + 1 < t_end ? tiles[t + 1].end : filled_to;
        const uint32_t nnew = next_end > filled_to ? (next_end - filled_to) / 4u : 0u;
        BarW4 pre[2];
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) pre[q] = fetch(filled_to + 4u * (wave + kRowsWaves * q < nnew ? wave + kRowsWaves * q : 0u));
        // the next round's new slots, none of which this round's groups read.  Parked BEFORE this round's stores are issued: vmcnt
        // retires in order, so a wait for these loads behind the stores would wait for the stores (which take microseconds)
        auto park_new = [&]() {
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q)
                if (wave + kRowsWaves * q < nnew) park(pre[q], filled_to + 4u * (wave + kRowsWaves * q));
            for (uint32_t c = wave + 2u * kRowsWaves; c < nnew; c += kRowsWaves) park(fetch(filled_to + 4u * c), filled_to + 4u * c);
        };
#if defined(GLV_EXP_ROWS_NOCOMPUTE)
        if (valid) park_new();
#else
        if (valid) {
            const uint32_t steps = (uint32_t) __builtin_amdgcn_readfirstlane((int) g.steps);
#if defined(GLV_EXP_ROWS_SAMEW)
            const float* wp = wg;
#else
            const float* wp = wg + (uint32_t) __builtin_amdgcn_readfirstlane((int) g.w_off);
#endif
            uint32_t slot = (uint32_t) __builtin_amdgcn_readfirstlane((int) g.slot0);       // even; a step reads slots slot, slot + 1, then moves on two (wrapping)
            // per pair of bars {2 i, 2 i + 1}: the running total and the stack of partial sums of the chunk under way
            glv_f2 tot[4], p0[4], p1[4], p2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) tot[i] = p0[i] = p1[i] = p2[i] = glv_f2{0.0f, 0.0f};
            HalfW wcur = load_half(wp);
            const float* wnext = wp + 32;                                       // the weight stream, one half step ahead: a running scalar pointer
            BarW4 xa = win_lane[(size_t) slot * 64], xc = win_lane[(size_t) slot * 64 + 64];
            slot = slot + 2u == NS ? 0u : slot + 2u;
            auto pk_add = [](glv_f2 a2, glv_f2 b2) { glv_f2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a2), "v"(b2)); return r; };
            // one octet of texels x eight bars; L = the octet's place in its chunk (compile time: the tree of partial sums --
            // group_sum's order: neighbours, pairs of pairs, the two quads -- built as the octets arrive)
            auto step = [&](auto LC) {
                constexpr int L = decltype(LC)::value;
                const BarW4 na = win_lane[(size_t) slot * 64], nc = win_lane[(size_t) slot * 64 + 64];       // the next step's texels (one step past the group's end is read and dropped)
                slot = slot + 2u == NS ? 0u : slot + 2u;
                const glv_f2 x01 = {xa.w[0], xa.w[1]}, x23 = {xa.w[2], xa.w[3]}, x45 = {xc.w[0], xc.w[1]}, x67 = {xc.w[2], xc.w[3]};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const HalfW wn = load_half(wnext);                            // one half step ahead (64 floats of slack follow the table)
#if defined(GLV_EXP_ROWS_SAMEW)         /* timing experiment (wrong results): every half step reads the same 128 bytes -- scalar-cache hits */
                    wnext = wp + (((wnext - wp) + 32) & 32);
#else
                    wnext += 32;
#endif
                    // {even chain, odd chain} of the half step's four bars, interleaved (a dependent packed op two slots later costs a wait
                    // state); first link: fma(x, w, +0) == x * w (both >= +0)
                    glv_f2 ac[4], o[2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm("v_pk_mul_f32 %0, %1, %2" : "=v"(ac[q]) : "v"(x01), "s"(wcur.w[4 * q + 0]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ac[q]) : "v"(x23), "s"(wcur.w[4 * q + 1]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ac[q]) : "v"(x45), "s"(wcur.w[4 * q + 2]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ac[q]) : "v"(x67), "s"(wcur.w[4 * q + 3]));
                    asm("v_add_f32 %0, %1, %2" : "=v"(o[0].x) : "v"(ac[0].x), "v"(ac[0].y));
                    asm("v_add_f32 %0, %1, %2" : "=v"(o[0].y) : "v"(ac[1].x), "v"(ac[1].y));
                    asm("v_add_f32 %0, %1, %2" : "=v"(o[1].x) : "v"(ac[2].x), "v"(ac[2].y));
                    asm("v_add_f32 %0, %1, %2" : "=v"(o[1].y) : "v"(ac[3].x), "v"(ac[3].y));
#pragma unroll
                    for (int bp = 0; bp < 2; ++bp) {
                        const int i = 2 * h + bp;
                        if constexpr (L == 0) p0[i] = o[bp];
                        else if constexpr (L == 1) p0[i] = pk_add(p0[i], o[bp]);
                        else if constexpr (L == 2 || L == 4) p1[i] = o[bp];
                        else if constexpr (L == 5) p1[i] = pk_add(p1[i], o[bp]);
                        else if constexpr (L == 3) p0[i] = pk_add(p0[i], pk_add(p1[i], o[bp]));
                        else if constexpr (L == 6) p2[i] = o[bp];
                        else p0[i] = pk_add(p0[i], pk_add(p1[i], pk_add(p2[i], o[bp])));
                    }
                    wcur = wn;
                }
                xa = na; xc = nc;
                GLV_SCHED_FENCE();
            };
            // whole chunks: after the last octet everything is folded into p0
            const uint32_t full = steps / (uint32_t) GL, rem = steps % (uint32_t) GL;
            for (uint32_t c = 0; c < full; ++c) {
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                if constexpr (GL >= 4) { step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{}); }
                if constexpr (GL >= 8) {
                    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
                    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) tot[i] = pk_add(tot[i], p0[i]);
            }
            // the last, partial chunk: the octets the bars do not have are +0 and x + 0 == x for x >= +0, so what is parked is added
            // in the tree's order and nothing else: rem = 1, 2, 4: p0;  3, 5, 6: p0 + p1;  7: p0 + (p1 + p2)
            if (rem) {
                for (uint32_t r = 0; r < rem; ++r) {
                    switch (r) {
                        case 0: step(std::integral_constant<int, 0>{}); break;
                        case 1: step(std::integral_constant<int, 1>{}); break;
                        case 2: step(std::integral_constant<int, GL >= 4 ? 2 : 0>{}); break;
                        case 3: step(std::integral_constant<int, GL >= 4 ? 3 : 0>{}); break;
                        case 4: step(std::integral_constant<int, GL >= 8 ? 4 : 0>{}); break;
                        case 5: step(std::integral_constant<int, GL >= 8 ? 5 : 0>{}); break;
                        default: step(std::integral_constant<int, GL >= 8 ? 6 : 0>{}); break;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    glv_f2 sum = p0[i];
                    if (rem == 7) sum = pk_add(sum, pk_add(p1[i], p2[i]));
                    else if (rem == 3 || rem == 5 || rem == 6) sum = pk_add(sum, p1[i]);
                    tot[i] = pk_add(tot[i], sum);
                }
            }
            park_new();
            // total / weight sum: with the host's reciprocal where that is the correctly rounded quotient (glv_tables.h
            // bar_rcp_division_ok; not for a total so small that the remainder below would be inexact), else the long way
            const uint32_t kg = T.k0 + 8u * wave;
            struct WS { glv_f2 v[8]; };                                         // {weight sum, its reciprocal or 0} of the eight bars: scalar loads
            const WS ws = *reinterpret_cast<const WS*>(wsum + 2u * kg);
            float q[8];
            bool fast = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) fast = fast && ws.v[j].y != 0.0f;
            float tmax = 0.0f;                                                  // is some total in (0, 2^-90)?  (totals are >= +0)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a0 = tot[i].x == 0.0f ? 1.0f : tot[i].x, a1 = tot[i].y == 0.0f ? 1.0f : tot[i].y;
                const float mn = a0 < a1 ? a0 : a1;
                tmax = i == 0 ? mn : (mn < tmax ? mn : tmax);
            }
            fast = fast && !__builtin_amdgcn_readfirstlane((int) (__ballot(tmax < 0x1p-90f) != 0ull));
            if (fast) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = j & 1 ? tot[j / 2].y : tot[j / 2].x, bsum = ws.v[j].x, r = ws.v[j].y;
                    const float q0 = a * r;
                    const float rm = __builtin_fmaf(-q0, bsum, a);
                    q[j] = __builtin_fmaf(rm, r, q0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) q[j] = (j & 1 ? tot[j / 2].y : tot[j / 2].x) / ws.v[j].x;
            }
            // parked for the round's flush: scattered 16 / 32-byte stores straight from here (one line per lane and instruction) cost more
            // than the arithmetic -- every later wait for a load then waits for their acknowledgements (vmcnt retires in order)
#pragma unroll
            for (int j = 0; j < 8; ++j) stage[(size_t) (8u * wave + (uint32_t) j) * kRowsStagePitch + lane] = q[j];
        }
#endif
        if (!valid) park_new();
        filled_to = next_end > filled_to ? next_end : filled_to;
        __syncthreads();
        // the round's m bars of R rows: every row's m values are one contiguous segment of the output; wave w takes rows w, w + 8, ...
        {
            const uint32_t m = T.k1 - T.k0;
            const float* sp = stage + (size_t) lane * kRowsStagePitch;
            const size_t at0 = (row0 + wave) * (size_t) bars + T.k0 + lane;
#pragma unroll
            for (uint32_t i = 0; i < 64u / kRowsWaves; ++i) {
                const uint32_t jr = wave + kRowsWaves * i;
#if defined(GLV_EXP_ROWS_NOFLUSH)       /* timing experiment (wrong results): one store in 64 */
                if (jr < R && lane < m && lane == 0) {
#else
                if (jr < R && lane < m) {
#endif
                    const float v = sp[jr];
                    const size_t at = at0 + (size_t) (kRowsWaves * i) * bars;
                    if (r16) reinterpret_cast<uint16_t*>(bars_out)[at] = (uint16_t) pack_unorm16(v, 0.0f);
                    else reinterpret_cast<float*>(bars_out)[at] = v;
                }
            }
        }
        __syncthreads();                                                        // the stage is free for the next round
    }
#endif
}

// the s16 window as float pairs: glv_winsplit.h (shared with the knob-sweep harness glv_tune.hip)
hipError_t launch_window_split(const double* w_tab, float* split, uint32_t n, int* d_fail_shifted, hipStream_t st) {
    return launch_window_split_impl(w_tab, split, n, d_fail_shifted, st);
}
hipError_t launch_window_split_check(const double* w_tab, const float* split, uint32_t n, unsigned long long* d_mismatches, hipStream_t st) {
    return launch_window_split_check_impl(w_tab, split, n, d_mismatches, st);
}

static int capped_grid(size_t items, int block) {
    size_t g = (items + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;     // 256 CUs x 8 resident 256-thread blocks, grid-stride beyond
    if (g < 1) g = 1;
    return (int) g;
}

hipError_t launch_post(const FrameArgs& a, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(glv_post_kernel, dim3(capped_grid((size_t) a.units * (n / 2), 256)), dim3(256), 0, st, a, n);
    return hipGetLastError();
}

hipError_t launch_unpack(const int16_t* pcm, size_t frames, int mono, float* l, float* r, hipStream_t st) {
    hipLaunchKernelGGL(glv_unpack_kernel, dim3(capped_grid(frames, 256)), dim3(256), 0, st, pcm, frames, mono, l, r);
    return hipGetLastError();
}

hipError_t launch_ring_planar(const void* ring, int is_f32, uint32_t n, uint32_t rot, int mono, size_t streams, float* out, hipStream_t st) {
    hipLaunchKernelGGL(glv_ring_planar_kernel, dim3(capped_grid(streams * n, 256)), dim3(256), 0, st, ring, is_f32, n, rot, mono, streams, out);
    return hipGetLastError();
}

hipError_t launch_bufscale(const float* in, float* out, size_t total_out, uint32_t k, hipStream_t st) {
    hipLaunchKernelGGL(glv_bufscale_kernel, dim3(capped_grid(total_out, 256)), dim3(256), 0, st, in, out, total_out, k);
    return hipGetLastError();
}
hipError_t launch_lerp(const float* s0, const float* e0, float* out, size_t total, float mod, hipStream_t st) {
    hipLaunchKernelGGL(glv_lerp_kernel, dim3(capped_grid(total, 256)), dim3(256), 0, st, s0, e0, out, total, mod);
    return hipGetLastError();
}
hipError_t launch_smooth(float* rows, size_t nrows, uint32_t n, const int* smin, const int* smax, uint32_t asz, uint32_t reach,
                         uint32_t max_window, hipStream_t st) {
    // ring kernel: 64 rows per wave, W >= largest window + 2 chunks (glv_smooth_ring_kernel); few rows or huge windows:
    // the row-prefix kernel
    const uint32_t need_w = max_window + 2 * kSmoothChunk;
    if (nrows >= 64 && need_w <= 512) {
        const unsigned wgs = (unsigned) ((nrows + 63) / 64);
#define GLV_RING(WW)                                                                                                           \
        do {                                                                                                                   \
            const size_t lds = sizeof(float) * 64 * (WW + 1);                                                                  \
            if (lds > 64 * 1024) {                                                                                             \
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(glv_smooth_ring_kernel<WW>),                  \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);                     \
                if (e != hipSuccess) return e;                                                                                 \
            }                                                                                                                  \
            hipLaunchKernelGGL(glv_smooth_ring_kernel<WW>, dim3(wgs), dim3(64), lds, st, rows, nrows, n, smin, smax, asz, reach); \
            return hipGetLastError();                                                                                          \
        } while (0)
        if (need_w <= 128) GLV_RING(128);
        if (need_w <= 256) GLV_RING(256);
        GLV_RING(512);
#undef GLV_RING
    }
    // rows per workgroup: what fits 64 KiB of LDS (two workgroups per CU), at most one row per lane, at least one
    // (reach <= n <= 32768 floats = 128 KiB: a row always fits the 160 KiB of a CU once the limit is raised)
    const size_t row_bytes = sizeof(float) * (size_t) (reach | 1u);
    size_t budget = 64 * 1024;
    if (row_bytes > budget) {
        budget = row_bytes;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(glv_smooth_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) budget);
        if (e != hipSuccess) return e;
    }
    size_t rpw = budget / row_bytes;
    if (rpw > 64) rpw = 64;
    const size_t wgs = (nrows + rpw - 1) / rpw;
    hipLaunchKernelGGL(glv_smooth_kernel, dim3((unsigned) wgs), dim3(64), rpw * row_bytes, st, rows, nrows, n, smin, smax, asz, reach, (uint32_t) rpw);
    return hipGetLastError();
}
template <int GL>
static void launch_bars_gl(const float* spec, float* bars_out, size_t nrows, uint32_t n, uint32_t bars, uint32_t nsteps,
                           const BarItem* items, const BarDesc* desc, const float* tap_w, hipStream_t st, int r) {
    auto grid = [&](size_t trips) { const size_t cap = 256 * 8; return dim3((unsigned) (trips < cap ? (trips ? trips : 1) : cap)); };   // grid-stride beyond
    if (nsteps == 2) hipLaunchKernelGGL((glv_bars_short_kernel<2, 2, GL>), grid((nrows + 1) / 2), dim3(256), 0, st, spec, bars_out, nrows, n, bars, items, desc, tap_w, r);
    else if (nsteps == 4) hipLaunchKernelGGL((glv_bars_short_kernel<4, 2, GL>), grid((nrows + 1) / 2), dim3(256), 0, st, spec, bars_out, nrows, n, bars, items, desc, tap_w, r);
    else hipLaunchKernelGGL((glv_bars_kernel<GL>), grid(nrows), dim3(256), 0, st, spec, bars_out, nrows, n, bars, nsteps, items, desc, tap_w, r);
}
template <int S, int GL>
static hipError_t launch_bars_rows(const float* spec, float* bars_out, size_t nrows, uint32_t n, uint32_t bars, const BarTile* tiles, uint32_t ntiles,
                                   const BarGroupDesc* groups, const float* wg, const float* wsum, hipStream_t st, int r) {
    const size_t lds = sizeof(float) * ((size_t) 64 * S + (size_t) kRowsTileBars * kRowsStagePitch);
    static std::atomic<bool> done[64] = {};
    if (lds > 64 * 1024) {
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(glv_bars_rows_kernel<S, GL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
        }
    }
    if (nrows == 0) return hipSuccess;                                          // prepare_bars_rows: the attribute only
    // 64 rows per workgroup in x, ranges of rounds in y: two resident workgroups per CU, twice over (a range start refills the whole
    // ring; N = 4096: 32 K rows 0.60 / 0.61 / 0.62 ms with 1 / 2 / 4 ranges, 8 K rows 0.48 / 0.19 / 0.18 ms with 1 / 4 / 16)
    const uint32_t xb = (uint32_t) ((nrows + 63) / 64);
    uint32_t yb = xb >= 1024 ? 1 : (1024 + xb - 1) / xb;
#if defined(GLV_TUNE_BUILD)
    if (const char* o = std::getenv("GLV_ROWS_YB")) yb = (uint32_t) atoi(o);       // tools/rows_bench: the split of the rounds over blockIdx.y
#endif
    if (yb > ntiles) yb = ntiles;
    const uint32_t tpw = (ntiles + yb - 1) / yb;
    yb = (ntiles + tpw - 1) / tpw;
    hipLaunchKernelGGL((glv_bars_rows_kernel<S, GL>), dim3(xb, yb), dim3(64 * kRowsWaves), lds, st, spec, static_cast<void*>(bars_out), nrows, n, bars, tiles,
                       ntiles, tpw, groups, wg, wsum, r);
    return hipGetLastError();
}

// the > 64 KiB dynamic-LDS opt-in of the rows kernel that launch_bars would pick for (n, rt), set ahead of the first launch (a
// process call is then a plain launch)
hipError_t prepare_bars_rows(uint32_t n, const BarRowsTables* rt) {
    if (rt == nullptr || rt->tiles == nullptr || rt->ntiles == 0) return hipSuccess;
#define GLV_ROWS(SS, GG) launch_bars_rows<SS, GG>(nullptr, nullptr, 0, n, 0, rt->tiles, rt->ntiles, rt->groups, rt->wg, rt->wsum, nullptr, 0)
    switch (bar_lanes_of(n)) {
        case 2: if (rt->tile_bins == 160) return GLV_ROWS(160, 2); break;
        case 4: if (rt->tile_bins == 160) return GLV_ROWS(160, 4); break;
        default:
            if (rt->tile_bins == 160) return GLV_ROWS(160, 8);
            if (rt->tile_bins == 240) return GLV_ROWS(240, 8);
            break;
    }
#undef GLV_ROWS
    return hipSuccess;
}

hipError_t launch_bars(const float* spec, float* bars_out, size_t nrows, uint32_t n, uint32_t bars, uint32_t nsteps,
                       const BarItem* items, const BarDesc* desc, const float* tap_w, hipStream_t st, bool r16, const BarRowsTables* rt) {
    const int r = r16 ? 1 : 0;
    // many bars of many rows (the pre-smoothing pass): one lane per row, eight bars per wave, weights as scalars (glv_bars_rows_kernel),
    // when the host could cut the bars into tiles that fit an LDS window of rt->tile_bins bins (glv_tables.h make_bar_groups)
    if (rt != nullptr && rt->tiles != nullptr && rt->ntiles != 0 && bars >= 256 && nrows >= 256) {
#define GLV_ROWS(SS, GG) launch_bars_rows<SS, GG>(spec, bars_out, nrows, n, bars, rt->tiles, rt->ntiles, rt->groups, rt->wg, rt->wsum, st, r)
        switch (bar_lanes_of(n)) {
            case 2: if (rt->tile_bins == 160) return GLV_ROWS(160, 2); break;
            case 4: if (rt->tile_bins == 160) return GLV_ROWS(160, 4); break;
            default:
                if (rt->tile_bins == 160) return GLV_ROWS(160, 8);
                if (rt->tile_bins == 240) return GLV_ROWS(240, 8);
                break;
        }
#undef GLV_ROWS
    }
    switch (bar_lanes_of(n)) {                                   // the work lists were made for 256 / bar_lanes_of(n) groups
        case 2: launch_bars_gl<2>(spec, bars_out, nrows, n, bars, nsteps, items, desc, tap_w, st, r); break;
        case 4: launch_bars_gl<4>(spec, bars_out, nrows, n, bars, nsteps, items, desc, tap_w, st, r); break;
        default: launch_bars_gl<8>(spec, bars_out, nrows, n, bars, nsteps, items, desc, tap_w, st, r); break;
    }
    return hipGetLastError();
}

#define GLV_BY_SIZE(log_nn, CALL, DEFAULT)                                                                            \
    switch (log_nn) {                                                                                                   \
        case 7: return CALL(7); case 8: return CALL(8); case 9: return CALL(9); case 10: return CALL(10);               \
        case 11: return CALL(11); case 12: return CALL(12); case 13: return CALL(13); case 14: return CALL(14);         \
    }                                                                                                                   \
    return DEFAULT

hipError_t launch_frame(int log_nn, int in_mode, int log_mode, int variant, const FrameArgs& a, int grid, hipStream_t st) {
#define GLV_CALL(K) launch_frame_##K(in_mode, log_mode, variant, a, grid, st)
    GLV_BY_SIZE(log_nn, GLV_CALL, hipErrorInvalidValue);
#undef GLV_CALL
}
int frame_variants(int log_nn) {
#define GLV_CALL(K) frame_variants_##K()
    GLV_BY_SIZE(log_nn, GLV_CALL, 1);
#undef GLV_CALL
}
bool frame_variant_ok(int log_nn, int in_mode, int log_mode, int variant) {
#define GLV_CALL(K) frame_variant_ok_##K(in_mode, log_mode, variant) != 0
    GLV_BY_SIZE(log_nn, GLV_CALL, false);
#undef GLV_CALL
}
FrameGeometry frame_geometry(int log_nn, int variant) {
#define GLV_CALL(K) frame_geometry_##K(variant)
    GLV_BY_SIZE(log_nn, GLV_CALL, FrameGeometry{});
#undef GLV_CALL
}
#undef GLV_BY_SIZE

}  // namespace glv
