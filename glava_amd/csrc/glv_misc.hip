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
// ROUND_FORMULA / SAMPLE_SCALE / SAMPLE_RANGE from glv_params: sinusoidal, 8, 0.9 as shipped) together with the work lists
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

// MANY bars (>= 256: the pre-smoothing pass of render.c:2277-2303, bars == n -- one output per texel, ~58 taps each at n = 4096,
// 237 K multiply-adds per row).  From 256 bars up a bar is one fused-multiply-add chain over its taps in bin order (glv_frame.h
// "GLV_OP_BARS arithmetic", glv_tables.h make_bar_mtiles), and a tap of weight +0 leaves a chain untouched -- so the 32 chains of a
// tile of consecutive bars can all run over the tile's common bin range, which makes the pass a banded matrix product:
//     out[row][bar] = sum over bins of x[row][bin] * w[bin][bar]          (then / weight_sum[bar])
//
// glv_bars_rows_kernel (whenever the host could cut rounds for an LDS ring): the matrix cores.  One v_mfma_f32_32x32x2_f32 is 32 rows x 32 bars x 2 bins, and on gfx950 it
// IS the k-ordered fmaf chain (tools/mfma_probe.hip: bit for bit, subnormals and zeros included) -- the documented arithmetic, at the
// f32 matrix rate.  A workgroup is four waves on the SAME 64 rows.  The rows' texels live in LDS as a RING of S bins, [bin mod S][row]
// (the a-operand of a step -- lane l: row l % 32 of the half, bin 2 s + l / 32 -- is one conflict-free ds_read_b32), clamped to
// [0, 1] (NaN -> 0) once when they are parked; the weights of a step are one coalesced 256-byte load in b-operand layout, requested
// eight steps ahead.  The bars' first bins grow slowly (4096 bars cover 1250 bins), so a ROUND -- four tiles, wave w takes tile w
// (host table) -- needs only a few bins the previous round did not have: requested before the round's arithmetic, written behind it
// into slots no wave of the round reads (the table guarantees end(t + 1) - origin(t) <= S); one barrier per round.  With rows as the
// result's rows a result register holds, across lanes 0..31 / 32..63, 32 consecutive bars of ONE row: every store instruction is two
// whole 128-byte (texels: 64-byte) row segments, straight from the accumulators, and a lane's 32 results all belong to one bar --
// one weight sum, one reciprocal (glv_tables.h bar_rcp_division_ok: three instructions give the correctly rounded quotient).
// Round 4's earlier versions, for the record (profiles/r04/rows_kernel.txt; N = 4096 x 32 K rows): one lane per row and one bar
// per wave with v_fmac_f32_dpp, LDS-bandwidth-bound, 1.33 ms; eight bars per wave sharing the texels with the weights as SGPR pairs
// for v_pk_fma_f32 (chunked summation, groups of bars padded to one first bin), bound by the L2 round trips of its scalar weight
// stream, 0.60 ms.
constexpr int kRowsWaves = 4;
constexpr size_t kRowsMin = 1;           // rows from which the matrix-core kernel is used: any (tools/sm_small.py)
#if !defined(GLV_ROWS_RB)               /* rows per workgroup: 64 (two MFMAs per step on the same weights) or 32; tools/rows_bench A/B builds override */
#define GLV_ROWS_RB 64
#endif
typedef float glv_f16v __attribute__((ext_vector_type(16)));
template <int S, int RB>
__global__ void __launch_bounds__(64 * kRowsWaves) glv_bars_rows_kernel(const float* __restrict__ spec, void* __restrict__ bars_out, size_t nrows, uint32_t n,
                                                                        uint32_t bars, const BarTile* __restrict__ rounds, uint32_t nrounds, uint32_t rounds_per_wg,
                                                                        const BarMTile* __restrict__ mtiles, const float* __restrict__ wt,
                                                                        const float* __restrict__ wsum, int r16) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ float rows_lds[];                 // [S][RB]: the ring
    static_assert(S % 8 == 0 && (RB == 64 || RB == 32), "a parked slot of four bins never straddles the ring's end; one or two MFMAs per step");
    constexpr uint32_t CPI = 64 / RB;                   // columns of four bins one fetch instruction covers (the lanes beyond RB rows take the next column)
    constexpr int NB = 1, PF = 8 * NB;                  // steps of weights in flight: NB banks of 8 registers (tiles are whole banks: glv_tables.h kBarStepPad)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const size_t row0 = (size_t) blockIdx.x * RB;
    if (row0 >= nrows) return;
    const uint32_t R = (uint32_t) (nrows - row0 < RB ? nrows - row0 : RB);
    const uint32_t frow = lane % (uint32_t) RB, fcol = lane / (uint32_t) RB;      // fill: this lane's row and which column of the instruction
    const float* src = spec + (row0 + (frow < R ? frow : R - 1)) * (size_t) n;
    const uint32_t t_begin = blockIdx.y * rounds_per_wg, t_end = t_begin + rounds_per_wg < nrounds ? t_begin + rounds_per_wg : nrounds;
    if (t_begin >= t_end) return;
    const glv_f2 ones = {1.0f, 1.0f};
    // 4 bins of this lane's row (column `col` of four bins from bin0); clamped here, once per texel: [0, 1] like the GL_R16 texel the shader samples, NaN -> 0 (v_pk_mul_f32
    // x, 1.0 clamp -- the operation bar_item_lane_sum applies to every tap)
    auto fetch = [&](uint32_t bin) {
        return ld<BarW4>(src, (bin + 4u <= n ? bin : n - 4u) * 4u);      // (a dummy request at the row's very end stays inside the row: ADVICE r4)
    };
    auto park = [&](const BarW4& v, uint32_t bin) {
        glv_f2 lo = {v.w[0], v.w[1]}, hi = {v.w[2], v.w[3]};
        asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(lo) : "v"(lo), "v"(ones));
        asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(hi) : "v"(hi), "v"(ones));
        float* at = rows_lds + (size_t) (bin % (uint32_t) S) * RB + frow;
        at[0] = lo.x; at[RB] = lo.y; at[2 * RB] = hi.x; at[3 * RB] = hi.y;
    };
    // instruction i of a run of `ncol` columns starting at bin0: this lane's column, clamped for the load, and whether it exists
    auto col_of = [&](uint32_t i) { return i * CPI + fcol; };
    // the ring starts as zeros: the padded steps of a tile read slots nothing was parked in yet, with weight +0 -- 0 * x must be +0
    for (uint32_t i = threadIdx.x; i < (uint32_t) S * RB; i += 64 * kRowsWaves) rows_lds[i] = 0.0f;
    __syncthreads();
    // the first round's whole window: four loads of a wave are in flight before the first is parked
    uint32_t filled_to;
    {
        const BarTile T = rounds[t_begin];
        const uint32_t ncol = (T.end - T.origin) / 4u, ninst = (ncol + CPI - 1u) / CPI;
        for (uint32_t ib = wave * 4u; ib < ninst; ib += kRowsWaves * 4) {
            BarW4 v4[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) v4[q] = fetch(T.origin + 4u * (col_of(ib + q) < ncol ? col_of(ib + q) : 0u));
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                if (col_of(ib + q) < ncol) park(v4[q], T.origin + 4u * col_of(ib + q));
        }
        filled_to = T.end;
    }
    __syncthreads();
    // lane part of an a-operand address: row (lane % 32) of a half, the odd bin of the pair for lanes 32..63
    const float* xlane = rows_lds + (lane >> 5) * (uint32_t) RB + (lane & 31u);
    // The weights: wave w takes tile k0 + w of every round and the host laid those tiles out one behind the other (glv_tables.h
    // make_bar_mtiles), so the wave reads ONE stream, PF steps ahead, straight across tile boundaries -- a register is reloaded as soon
    // as its step has used it, and the loads a tile's first steps need were issued before the previous tile's stores (vmcnt retires in
    // order: a load behind the stores waits for them).  bank: which of the NB banks of 8 registers the next 8 steps use.
    float w[PF];
    const float* wp = nullptr;                          // stream position of the NEXT load (lane-offset)
    uint32_t bank = 0;
    for (uint32_t t = t_begin; t < t_end; ++t) {
        const BarTile T = rounds[t];                                            // uniform: scalar loads
        const bool valid = T.k0 + wave < T.k1;
        const BarMTile M = mtiles[valid ? T.k0 + wave : T.k0];
        // what the next round adds to the ring: requested now, parked behind this round's arithmetic (up to two slots per wave in
        // registers -- what a round adds at most in practice --, the rest after them)
        const uint32_t next_end = t + 1 < t_end ? rounds[t + 1].end : filled_to;
        const uint32_t nnew = next_end > filled_to ? (next_end - filled_to) / 4u : 0u;
        BarW4 pre[2];
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) pre[q] = fetch(filled_to + 4u * (col_of(wave + kRowsWaves * q) < nnew ? col_of(wave + kRowsWaves * q) : 0u));
    auto park_new = [&]() {
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q)
                if (col_of(wave + kRowsWaves * q) < nnew) park(pre[q], filled_to + 4u * col_of(wave + kRowsWaves * q));
            for (uint32_t i = wave + 2u * kRowsWaves; i * CPI < nnew; i += kRowsWaves)
                if (col_of(i) < nnew) park(fetch(filled_to + 4u * col_of(i)), filled_to + 4u * col_of(i));
        };
        if (valid) {
            const uint32_t steps = (uint32_t) __builtin_amdgcn_readfirstlane((int) M.steps);        // a multiple of 8 (glv_tables.h kBarStepPad)
            uint32_t sb = (uint32_t) __builtin_amdgcn_readfirstlane((int) (M.origin % (uint32_t) S));   // ring slot of the step's even bin
            glv_f16v acc0 = {0}, acc1 = {0};                                    // rows 0..31 / 32..63 of the block (RB == 64) x the tile's 32 bars
            if (wp == nullptr) {                                                // the wave's first tile: fill the pipeline
                wp = wt + (uint32_t) __builtin_amdgcn_readfirstlane((int) M.w_off) + lane;
#pragma unroll
                for (int i = 0; i < PF; ++i) w[i] = wp[(size_t) i * 64];
                wp += (size_t) PF * 64;
                bank = 0;
            }
            // the texels two steps ahead; a step is two MFMAs (rows 0..31 and 32..63 of the block) on the same weights
            float xa0 = xlane[(size_t) sb * RB], xb0 = RB == 64 ? xlane[(size_t) sb * RB + 32] : 0.0f;
            sb = sb + 2u == (uint32_t) S ? 0u : sb + 2u;
            float xa1 = xlane[(size_t) sb * RB], xb1 = RB == 64 ? xlane[(size_t) sb * RB + 32] : 0.0f;
            sb = sb + 2u == (uint32_t) S ? 0u : sb + 2u;
            auto eight = [&](auto BC) {                                         // eight steps on bank B
                constexpr int B = decltype(BC)::value;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float na = xlane[(size_t) sb * RB], nb = RB == 64 ? xlane[(size_t) sb * RB + 32] : 0.0f;      // (up to two steps past the tile's end are read and dropped)
                    sb = sb + 2u == (uint32_t) S ? 0u : sb + 2u;
                    const float wcur = w[8 * B + u];
                    w[8 * B + u] = wp[(size_t) u * 64];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa0, wcur, acc0, 0, 0, 0);
                    if constexpr (RB == 64) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xb0, wcur, acc1, 0, 0, 0);
                    xa0 = xa1; xb0 = xb1; xa1 = na; xb1 = nb;
                }
                wp += (size_t) 8 * 64;
            };
            if constexpr (NB == 1) {
                // (a taken branch between MFMAs is dear -- four-step blocks: 0.41 instead of 0.33 ms --, so two blocks per trip where there are two)
                uint32_t s8 = 0;
                for (; s8 + 16u <= steps; s8 += 16u) { eight(std::integral_constant<int, 0>{}); eight(std::integral_constant<int, 0>{}); }
                if (s8 < steps) eight(std::integral_constant<int, 0>{});
            } else {
                for (uint32_t s8 = 0; s8 < steps; s8 += 8u) {
                    if (bank == 0) eight(std::integral_constant<int, 0>{});
                    else if (bank == 1) eight(std::integral_constant<int, (NB > 1 ? 1 : 0)>{});
                    else if (bank == 2) eight(std::integral_constant<int, (NB > 2 ? 2 : 0)>{});
                    else eight(std::integral_constant<int, (NB > 3 ? 3 : 0)>{});
                    bank = bank + 1u == (uint32_t) NB ? 0u : bank + 1u;
                }
            }
            park_new();
            // a lane's 32 results are one bar (k0 + lane % 32) of the rows 8 (r / 4) + 4 (lane / 32) + r % 4 (+ 32 for acc1)
            const uint32_t kb = M.k0 + (lane & 31u);
            const glv_f2 ws = *reinterpret_cast<const glv_f2*>(wsum + 2u * (size_t) kb);       // {weight sum, its reciprocal or 0} (padded to whole tiles)
            float tmin = 1.0f;                                                  // is some total in (0, 2^-90)?  (totals are >= +0)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a0 = acc0[r] == 0.0f ? 1.0f : acc0[r], a1 = RB == 32 || acc1[r] == 0.0f ? 1.0f : acc1[r];
                const float mn = a0 < a1 ? a0 : a1;
                tmin = mn < tmin ? mn : tmin;
            }
            // total / weight sum: with the host's reciprocal where that is the correctly rounded quotient (glv_tables.h bar_rcp_division_ok;
            // not for a total so small that the remainder would be inexact), else the long way -- the whole wave one way or the other
            const bool fast = __ballot(ws.y == 0.0f || tmin < 0x1p-90f) == 0ull;
            const size_t at0 = (row0 + 4u * (lane >> 5)) * (size_t) bars + kb;
            auto put = [&](int r, float v) {
                const uint32_t jr = 32u * (uint32_t) (r / 16) + 8u * (uint32_t) ((r & 15) / 4) + 4u * (lane >> 5) + (uint32_t) (r & 3);
                if (jr < R && kb < bars) {
                    const size_t at = at0 + (size_t) (32u * (uint32_t) (r / 16) + 8u * (uint32_t) ((r & 15) / 4) + (uint32_t) (r & 3)) * bars;
                    if (r16) reinterpret_cast<uint16_t*>(bars_out)[at] = (uint16_t) pack_unorm16(v, 0.0f);
                    else reinterpret_cast<float*>(bars_out)[at] = v;
                }
            };
            if (fast) {
#pragma unroll
                for (int r = 0; r < RB / 2; ++r) {
                    const float a = r < 16 ? acc0[r & 15] : acc1[r & 15];
                    const float q0 = a * ws.y;
                    const float rm = __builtin_fmaf(-q0, ws.x, a);
                    put(r, __builtin_fmaf(rm, ws.y, q0));
                }
            } else {
#pragma unroll
                for (int r = 0; r < RB / 2; ++r) put(r, (r < 16 ? acc0[r & 15] : acc1[r & 15]) / ws.x);
            }
        } else {
            park_new();
        }
        filled_to = next_end > filled_to ? next_end : filled_to;
        __syncthreads();
    }
#endif
}

// The same arithmetic with one lane per bar, off the same tables: tiles no LDS ring takes (n = 32768; a few hundred bars spread over a
// long row).  A wave = two tiles of one row; the row is read through L1 (32 lanes share a texel).  (Even two rows -- a single GLava
// instance -- are quicker on the matrix cores where rounds exist: N = 4096 14.9 us against 27.1 us, tools/sm_small.py.)
__global__ void __launch_bounds__(256) glv_bars_seq_kernel(const float* __restrict__ spec, void* __restrict__ bars_out, size_t nrows, uint32_t n, uint32_t bars,
                                                          const BarMTile* __restrict__ mtiles, uint32_t ntiles, const float* __restrict__ wt,
                                                          const float* __restrict__ wsum, int r16) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t units = nrows * (size_t) ((ntiles + 1u) / 2u);
    for (size_t u = (size_t) blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); u < units; u += (size_t) gridDim.x * (blockDim.x / 64)) {
        const size_t row = u / ((ntiles + 1u) / 2u);
        const uint32_t T = 2u * (uint32_t) (u % ((ntiles + 1u) / 2u)) + (lane >> 5);
        if (T >= ntiles) continue;
        const BarMTile M = mtiles[T];
        const float* x = spec + row * (size_t) n;
        const float* wp = wt + M.w_off + (lane & 31u);
        float acc = 0.0f;
        for (uint32_t i = 0; i < 2u * M.steps; ++i) {
            const uint32_t bin = M.origin + i;
            float xv = x[bin < n ? bin : n - 1u];                               // (padded steps: weight +0)
#if defined(__HIP_DEVICE_COMPILE__)
            asm("v_mul_f32 %0, 1.0, %1 clamp" : "=v"(xv) : "v"(xv));            // [0, 1], NaN -> 0: what the rows kernel parks
#else
            xv = xv > 0.0f ? (xv < 1.0f ? xv : 1.0f) : 0.0f;
#endif
            acc = __builtin_fmaf(xv, wp[(size_t) i * 32], acc);
        }
        const uint32_t kb = M.k0 + (lane & 31u);
        if (kb < bars) {
            const float v = acc / wsum[2u * (size_t) kb];
            if (r16) reinterpret_cast<uint16_t*>(bars_out)[row * bars + kb] = (uint16_t) unorm16(v);
            else reinterpret_cast<float*>(bars_out)[row * bars + kb] = v;
        }
    }
}

// SAMPLE_MODE maximum / hybrid (shaders/glava/util/smooth.glsl:41-59; glv_params.sample_mode has the contract, the tests' CPU checker restates it
// as glvo_bars_mode_at): a maximum is not a matrix product and the hybrid's average is the shader's own chain of float additions, so a bar is ONE lane
// walking its taps in bin order -- v = x * w (rounded), vmax = max(vmax, v), avg = avg + v -- for RR rows at a time (RR = 8 / 4 / 1 independent chains per lane).
// A workgroup parks the bins the bars sample of its RR rows in LDS, clamped to [0, 1] (NaN -> 0) like the texels the shader fetches (LDS = false: rows
// too long for that are read through L1); wave w then takes the 64-bar blocks w, w + 4, ...: lane l of a block is bar 64 blk + l, its weights come
// block-transposed (glv_tables.h make_bar_mode_blocks: [tap][lane], one coalesced 256-byte load per tap) and padded with +0 up to the block's longest
// bar -- x * +0 = +0 changes neither the maximum (vmax >= +0) nor the sum.  MODE 1: bar = vmax;  MODE 2: bar = vmax * (1 - H) + (avg / weight) * H,
// every operation rounded on its own (no fused multiply-add: the shader's expression as written).
template <int MODE, int RR, bool LDS>
__global__ void __launch_bounds__(256) glv_bars_mode_kernel(const float* __restrict__ spec, void* __restrict__ bars_out, size_t nrows, uint32_t n, uint32_t bars,
                                                           const BarDesc* __restrict__ desc, const BarModeBlock* __restrict__ blocks, uint32_t nblocks,
                                                           const float* __restrict__ mw, uint32_t bins, float hyb, float one_minus_hyb, int r16) {
    extern __shared__ __attribute__((aligned(16))) float glv_mode_rows[];        // [bins][RR]: a tap's RR rows are one 16-byte read per four rows
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    auto clamp01 = [](float x) { return x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f; };   // NaN -> 0
    for (size_t row0 = (size_t) blockIdx.x * RR; row0 < nrows; row0 += (size_t) gridDim.x * RR) {
        if constexpr (LDS) {
            __syncthreads();                                                      // the previous rows have been read
            for (uint32_t i = threadIdx.x; i < (uint32_t) RR * bins; i += 256u) {
                const uint32_t r = i / bins, bin = i - r * bins;                  // (coalesced row reads; the transposing writes are the small side)
                const size_t row = row0 + r < nrows ? row0 + r : nrows - 1;
                glv_mode_rows[bin * (uint32_t) RR + r] = clamp01(spec[row * (size_t) n + bin]);
            }
            __syncthreads();
        }
        for (uint32_t blk = wave; blk < nblocks; blk += 4u) {
            const BarModeBlock B = blocks[blk];
            const uint32_t k = blk * 64u + lane;
            const BarDesc d = desc[k < bars ? k : bars - 1u];
            const float* wp = mw + B.w_off + lane;
            float vmax[RR], avg[RR];
#pragma unroll
            for (int r = 0; r < RR; ++r) { vmax[r] = 0.0f; avg[r] = 0.0f; }
            // the taps in groups of JU: the group's weights are requested together (one L2 round trip per group instead of per tap; the host pads a
            // block's weights with +0 to a multiple of JU taps)
            constexpr uint32_t JU = kBarModeUnroll;
            for (uint32_t j0 = 0; j0 < B.maxcount; j0 += JU) {
                float w[JU];
#pragma unroll
                for (uint32_t u = 0; u < JU; ++u) w[u] = wp[(size_t) (j0 + u) * 64u];
#pragma unroll
                for (uint32_t u = 0; u < JU; ++u) {
                    uint32_t bin = d.first_bin + j0 + u;
                    bin = bin < bins ? bin : bins - 1u;                           // (past the bar's own taps: weight +0)
                    float x[RR];
                    if constexpr (LDS && RR % 4 == 0) {
#pragma unroll
                        for (int q = 0; q < RR / 4; ++q) {
                            const float4 t = *reinterpret_cast<const float4*>(&glv_mode_rows[bin * (uint32_t) RR + 4u * (uint32_t) q]);
                            x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
                        }
                    } else if constexpr (LDS) {
#pragma unroll
                        for (int r = 0; r < RR; ++r) x[r] = glv_mode_rows[bin * (uint32_t) RR + (uint32_t) r];
                    } else {
#pragma unroll
                        for (int r = 0; r < RR; ++r) x[r] = clamp01(spec[(row0 + r < nrows ? row0 + r : nrows - 1) * (size_t) n + bin]);
                    }
#pragma unroll
                    for (int r = 0; r < RR; ++r) {
                        const float v = __fmul_rn(x[r], w[u]);
                        vmax[r] = vmax[r] < v ? v : vmax[r];                      // smooth.glsl:48-49 / :56-57
                        if constexpr (MODE == 2) avg[r] = __fadd_rn(avg[r], v);
                    }
                }
            }
            if (k < bars) {
#pragma unroll
                for (int r = 0; r < RR; ++r) {
                    if (row0 + r >= nrows) break;
                    float v = vmax[r];
                    if constexpr (MODE == 2) v = __fadd_rn(__fmul_rn(vmax[r], one_minus_hyb), __fmul_rn(avg[r] / d.weight_sum, hyb));   // smooth.glsl:51
                    if (r16) reinterpret_cast<uint16_t*>(bars_out)[(row0 + r) * bars + k] = (uint16_t) unorm16(v);
                    else reinterpret_cast<float*>(bars_out)[(row0 + r) * bars + k] = v;
                }
            }
        }
    }
}
template <int MODE>
static hipError_t launch_bars_mode(const float* spec, void* bars_out, size_t nrows, uint32_t n, uint32_t bars, const BarDesc* desc, const BarRowsTables& rt, hipStream_t st, int r16) {
    if (rt.mblocks == nullptr || rt.mw == nullptr || rt.nmblocks == 0 || rt.mode_bins == 0 || rt.mode_bins > n) return hipErrorInvalidValue;
    const uint32_t bins = rt.mode_bins;
    const float h = rt.hybrid_weight, omh = 1.0f - rt.hybrid_weight;
    auto grid_of = [&](int rr) { const size_t g = (nrows + (size_t) rr - 1) / (size_t) rr; return dim3((unsigned) (g < 256u * 8u ? (g ? g : 1) : 256u * 8u)); };
#define GLV_MODE_LAUNCH(RR, LDSF, BYTES) hipLaunchKernelGGL((glv_bars_mode_kernel<MODE, RR, LDSF>), grid_of(RR), dim3(256), BYTES, st, spec, bars_out, nrows, n, bars, desc, rt.mblocks, rt.nmblocks, rt.mw, bins, h, omh, r16)
    // rows per workgroup by what their sampled bins take of the 64 KiB a launch may ask for without an attribute: 8 (n <= 4096 as shipped), 4, 1; else through L1
    if ((size_t) bins * 32u <= 64u * 1024u && nrows >= 8) GLV_MODE_LAUNCH(8, true, (size_t) bins * 32u);
    else if ((size_t) bins * 16u <= 64u * 1024u) GLV_MODE_LAUNCH(4, true, (size_t) bins * 16u);
    else if ((size_t) bins * 4u <= 64u * 1024u) GLV_MODE_LAUNCH(1, true, (size_t) bins * 4u);
    else GLV_MODE_LAUNCH(4, false, 0);
#undef GLV_MODE_LAUNCH
    return hipGetLastError();
}

// MANY bars over TEXEL rows (the library's GL chains, gl_storage != 0: the pre-smoothing pass of render.c:2277-2303 samples a GL_R16
// texture) -- EXACT integer arithmetic on the i8 matrix cores (glv_tables.h make_bar_itiles has the contract; the tests' CPU checker restates
// it as glvo_bars_int_at).  The texels are 16-bit integers; with the bar's weights as integers W that sum to 2^P the weighted mean
// is sum W c / 2^P exactly.  c - 32896 = 256 h + l and W = 65536 w2 + 256 w1 + w0 in balanced signed bytes: one step is 32 rows x 32 bars
// x 32 bins as six v_mfma_i32_32x32x32_i8 (h w2 | h w1 + l w2 | h w0 + l w1 | l w0: four int32 accumulator tiles, nothing rounds) -- 28 x
// the multiply-add rate of the f32 form, which this kernel replaces wherever the rows are texels.  Same shape as glv_bars_rows_kernel
// otherwise: four waves on the same RB rows, wave w takes tile w of every round, the rows' texels in an LDS ring of S bins -- here as two
// planes of signed bytes [row][bin] (row pitch S + 16 bytes: the a-operand, 16 consecutive bins of one row per lane, is a conflict-free
// ds_read_b128), split and biased ONCE when they are parked, 8 texels = one 16-byte load at a time; the weights one coalesced 1 KiB line per
// digit and step, one step ahead, one stream per wave straight across tile boundaries.  A slot nothing was parked in yet, or that the next
// round is overwriting, only ever meets weight 0, and 0 x anything is 0 here -- no clearing, no clamping, no NaN.  The epilogue is five
// integer instructions per texel: floor(T / 2^16) = (a3 << 8) + a2 + ((a1 + (a0 >> 8)) >> 8), texel = (that + c) >> s with the host's
// c = 32896 2^s + 2^(s - 1), s = P - 16 (round to nearest, an exact half up).
typedef int glv_i4v __attribute__((ext_vector_type(4)));
typedef int glv_i2v __attribute__((ext_vector_type(2)));
typedef int glv_i16v __attribute__((ext_vector_type(16)));
template <int S, int RB, bool F32IN, bool R16>
__global__ void __launch_bounds__(64 * kRowsWaves, 2) glv_bars_rows_i8_kernel(const void* __restrict__ rows_in, void* __restrict__ bars_out, size_t nrows, uint32_t n,
                                                                              uint32_t bars, const BarTile* __restrict__ rounds, uint32_t nrounds, uint32_t rounds_per_wg,
                                                                              const BarMTile* __restrict__ tiles, const glv_i4v* __restrict__ wq,
                                                                              const BarIFin* __restrict__ fin) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(S % 32 == 0 && (RB == 64 || RB == 32), "16-bin chunks never straddle the ring's end and (S + 16) / 16 is odd; one or two row groups");
    extern __shared__ __attribute__((aligned(16))) char i8_lds[];       // [2 planes][RB rows][S + 16 bytes]
    constexpr uint32_t PITCH = S + 16, S16 = S / 16, G = RB / 32, CPI = 64 * kRowsWaves / RB;       // CPI: columns of 8 bins one sweep of the workgroup fetches
    char* plane_h = i8_lds;
    char* plane_l = i8_lds + (size_t) RB * PITCH;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const size_t row0 = (size_t) blockIdx.x * RB;
    if (row0 >= nrows) return;
    const uint32_t R = (uint32_t) (nrows - row0 < RB ? nrows - row0 : RB);
    const uint32_t t_begin = blockIdx.y * rounds_per_wg, t_end = t_begin + rounds_per_wg < nrounds ? t_begin + rounds_per_wg : nrounds;
    if (t_begin >= t_end) return;
    const uint32_t frow = threadIdx.x % (uint32_t) RB, fcol = threadIdx.x / (uint32_t) RB;
    const size_t srow = row0 + (frow < R ? frow : R - 1);                 // (a partial row block repeats its last row; its stores are masked)
    const char* src = static_cast<const char*>(rows_in) + srow * (size_t) n * (F32IN ? 4u : 2u);
    struct Tex8 { uint32_t d[4]; };                                     // 8 texels, two per dword
    auto fetch = [&](uint32_t bin) -> Tex8 {
        Tex8 v;
        bin = bin + 8u <= n ? bin : n - 8u;                             // (a dummy request -- nothing new to park -- at the row's very end stays inside the row)
        if constexpr (F32IN) {                                          // rows of floats c / 65535 (the pass-by-pass chain): back to the texels, exactly
            const BarW4 a = ld<BarW4>(src, bin * 4u), b = ld<BarW4>(src, bin * 4u + 16u);
            v.d[0] = pack_unorm16(a.w[0], a.w[1]); v.d[1] = pack_unorm16(a.w[2], a.w[3]);
            v.d[2] = pack_unorm16(b.w[0], b.w[1]); v.d[3] = pack_unorm16(b.w[2], b.w[3]);
        } else {
            const glv_i4v a = *reinterpret_cast<const glv_i4v*>(src + bin * 2u);
            v.d[0] = (uint32_t) a.x; v.d[1] = (uint32_t) a.y; v.d[2] = (uint32_t) a.z; v.d[3] = (uint32_t) a.w;
        }
        return v;
    };
    auto park = [&](const Tex8& v, uint32_t bin) {                       // low bytes / high bytes of the 8 texels, biased to signed (c ^ 0x8080)
        const uint32_t l0 = __builtin_amdgcn_perm(v.d[1], v.d[0], 0x06040200u) ^ 0x80808080u, l1 = __builtin_amdgcn_perm(v.d[3], v.d[2], 0x06040200u) ^ 0x80808080u;
        const uint32_t h0 = __builtin_amdgcn_perm(v.d[1], v.d[0], 0x07050301u) ^ 0x80808080u, h1 = __builtin_amdgcn_perm(v.d[3], v.d[2], 0x07050301u) ^ 0x80808080u;
        const uint32_t at = frow * PITCH + bin % (uint32_t) S;
        *reinterpret_cast<uint2*>(plane_l + at) = make_uint2(l0, l1);
        *reinterpret_cast<uint2*>(plane_h + at) = make_uint2(h0, h1);
    };
    // the first round's whole window
    uint32_t filled_to;
    {
        const BarTile T = rounds[t_begin];
        const uint32_t ncol = (T.end - T.origin) / 8u;
        for (uint32_t c0 = fcol; c0 < ncol; c0 += 4u * CPI) {
            Tex8 v4[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) v4[q] = fetch(T.origin + 8u * (c0 + q * CPI < ncol ? c0 + q * CPI : 0u));
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                if (c0 + q * CPI < ncol) park(v4[q], T.origin + 8u * (c0 + q * CPI));
        }
        filled_to = T.end;
    }
    __syncthreads();
    // a-operand: lane l reads 16 consecutive bins (one half of the step's 32) of row l % 32 of a group
    const uint32_t arow = (lane & 31u) * PITCH, ahalf = lane >> 5;
    // The weights: wave w takes tile k0 + w of every round and the host laid those tiles out one behind the other, so the wave reads ONE
    // stream of steps straight across tile boundaries, PF steps ahead: a step's three digit fragments sit in bank (step mod PF), and the bank
    // is reloaded as soon as its step has used it (the stream ends in PF steps of zeros).  Register banks mean an unrolled loop, and a tile may
    // end after any step: the loop below runs over the wave's STREAM, and what a tile's last step is followed by -- parking the next round's
    // bins, the epilogue, the round's barrier, the next tile's set-up -- hangs off each of the PF steps as a side block.
    constexpr int PF = 3;
    static_assert((uint32_t) PF < kBarILookAhead, "the stream that lies last in memory is read PF steps past its last tile: the host pads kBarILookAhead steps");
    const glv_i4v* wp = nullptr;                                        // stream position of the NEXT load (lane-offset)
    glv_i4v wb[PF][3];
    glv_i16v acc[G][4];
    uint32_t t = t_begin, left = 0, ck = 0, next_end = filled_to, nnew = 0;
    BarMTile M = tiles[0];                                              // (overwritten before use)
    Tex8 pre[2];
    auto park_new = [&]() {
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q)
            if (fcol + q * CPI < nnew) park(pre[q], filled_to + 8u * (fcol + q * CPI));
        for (uint32_t c = fcol + 2u * CPI; c < nnew; c += CPI) park(fetch(filled_to + 8u * c), filled_to + 8u * c);
        filled_to = next_end > filled_to ? next_end : filled_to;
    };
    // opens round t: sets up the wave's tile, requests what the NEXT round adds to the ring (parked behind this round's arithmetic) and the
    // tile's epilogue constants.  false: the wave has no tile in this round
    // the epilogue constants {c, s} of the wave's tile: requested a ROUND ahead (the next tile's descriptor is known by then), behind the previous
    // tile's last step -- the full wait in front of that tile's stores covers them, so their first use never waits
    BarIFin fcur = BarIFin{0u, 0u}, fnext = BarIFin{0u, 0u};
    // (the descriptors of round t + 1 -- uniform scalar loads, the tile's dependent on the round's -- are requested while round t runs)
    BarTile Tn = rounds[t_begin];
    BarMTile Mn = tiles[Tn.k0 + wave < Tn.k1 ? Tn.k0 + wave : Tn.k0];
    auto open_round = [&]() -> bool {
        const BarTile T = Tn;
        const bool valid = T.k0 + wave < T.k1;
        if (valid) M = Mn;
        Tn = rounds[t + 1 < t_end ? t + 1 : t];
        Mn = tiles[Tn.k0 + wave < Tn.k1 ? Tn.k0 + wave : Tn.k0];
        next_end = t + 1 < t_end ? Tn.end : filled_to;
        nnew = next_end > filled_to ? (next_end - filled_to) / 8u : 0u;
        if (valid) {
            left = (uint32_t) __builtin_amdgcn_readfirstlane((int) M.steps);
            ck = ((uint32_t) __builtin_amdgcn_readfirstlane((int) (M.origin >> 4)) + ahalf) % S16;      // this lane's 16-bin chunk of the step, in the ring
#pragma unroll
            for (uint32_t g = 0; g < G; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[g][q] = glv_i16v{0};
        }
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) pre[q] = fetch(filled_to + 8u * (fcol + q * CPI < nnew ? fcol + q * CPI : 0u));
        fcur = fnext;
        return valid;
    };
    // rounds without a tile for this wave: park, join the barrier, go on.  false: no round is left
    auto next_tile = [&]() -> bool {
        while (t < t_end) {
            if (open_round()) return true;
            fnext = fin[Mn.k0 + (lane & 31u)];
            // (awaited HERE: a compiler-visible load left in flight across the loop's back edge makes the backend guard every later reuse of its
            // register with a vmcnt(0) -- in front of every step's LDS reads, a drain of the result stores per step)
            asm volatile("" : "+v"(fnext.c), "+v"(fnext.s));
            park_new();
            __syncthreads();
            ++t;
        }
        return false;
    };
    // behind a tile's last step
    // Loads and stores retire on ONE counter, so behind a tile's result stores a wait for a weight fragment is in effect a wait for the stores as
    // well (vmcnt(6) in step() stays CORRECT -- loads retire in order among themselves -- it just lasts until the stores have left too).  Hence
    // every load is awaited BEFORE the stores are issued: the fragments of the next PF steps are then in their registers and those steps (most
    // tiles have no more) wait for nothing while the stores drain behind the round's barrier; `fresh` counts the steps that still need no wait.
    // That full wait includes the weight requests of the tile's last steps, a moment old.  Round 6: ALL of the epilogue's arithmetic (which needs no
    // load: its constants were requested a round ahead) sits between the last step and the wait, so that the L2 round trip of those requests runs
    // under ~1.5 K cycles of vector work instead of in front of them; parking the next round's texels follows the wait.
    uint32_t fresh = 0;
    auto close_tile = [&]() {
        fnext = fin[Mn.k0 + (lane & 31u)];                                      // (padded to whole tiles; Mn: the wave's tile of the next round, or a valid one)
        // a lane's 16 results of a group are one bar (k0 + lane % 32) of the rows 32 g + 8 (r / 4) + 4 (lane / 32) + r % 4
        const uint32_t kb = M.k0 + (lane & 31u);
        const BarIFin f = fcur;
        using OutT = std::conditional_t<R16, uint16_t, float>;
        auto result = [&](uint32_t g, int r) -> OutT {
            const int a0 = acc[g][0][r], a1 = acc[g][1][r], a2 = acc[g][2][r], a3 = acc[g][3][r];
            if constexpr (R16) {
                // (a bar whose weights sum to 0 has c = 0, s = 16 and all-zero digits: 0 >> 16 -- no branch)
                const uint32_t t16 = (uint32_t) ((a3 << 8) + a2 + ((a1 + (a0 >> 8)) >> 8));
                return (uint16_t) ((t16 + f.c) >> f.s);
            } else {
                const int P = (int) f.s + 16;
                const long long tot = ((long long) a3 << 24) + ((long long) a2 << 16) + ((long long) a1 << 8) + a0 + ((long long) 32896 << P);
                return f.s == kBarIFinNone ? __builtin_nanf("") : (float) (__builtin_ldexp((double) tot, -P) / 65535.0);
            }
        };
        auto out_of = [](uint32_t x) -> OutT { if constexpr (R16) return (uint16_t) x; else return __builtin_bit_cast(float, x); };
        // all results first (they take the place of the accumulators they come from) ...
        uint32_t res[G][16];                                                    // (a float's bits, or the texel)
#pragma unroll
        for (uint32_t g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (R16) res[g][r] = result(g, r);
                else res[g][r] = __builtin_bit_cast(uint32_t, result(g, r));
            }
        // ... pinned in front of the wait (sixteen at a time: an asm statement takes thirty operands) ...
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            uint32_t (&x)[16] = res[g];
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]),
                         "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
        }
        // ... then every load this wave has in flight -- the weight stream's next PF steps ...
        __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0)
        asm volatile("" : "+v"(pre[0].d[0]), "+v"(pre[0].d[1]), "+v"(pre[0].d[2]), "+v"(pre[0].d[3]), "+v"(pre[1].d[0]), "+v"(pre[1].d[1]), "+v"(pre[1].d[2]), "+v"(pre[1].d[3]) : : "memory");
        fresh = PF;
        park_new();                                                             // (the ring's new texels: requested when the round opened)
        // ... then the stores, `global_store v_off, v_data, s[base:base+1]`: a uniform row base in scalar registers, walked from row to row by
        // scalar additions, + one 32-bit lane offset -- no vector address arithmetic per store.  (What it takes: the row OFFSET goes through an asm
        // statement so that the backend neither folds the lane offset into a 64-bit vector address it then walks with a v_lshl_add_u64 per store nor
        // keeps 64 hoisted row offsets in scalar registers -- the offset, not the pointer: a pointer that went through an asm statement loses its
        // address space and the stores become FLAT ones -- and the lane offset is re-defined opaquely in every basic block that stores: its
        // zero-extension must sit next to the store for the addressing mode to be matched.)
        // register r of a group is row 8 (r / 4) + r % 4 (+ 4 for the upper lanes: in the lane offset) of its 32
        uint32_t loff = (4u * (lane >> 5) * bars + kb) * (uint32_t) sizeof(OutT);
        const size_t one_row = (size_t) bars * sizeof(OutT);
        size_t ro = row0 * one_row;                                             // uniform: offset of the row the next store goes to
        // a partial last block: rows of the group this lane may store
        uint32_t rlim = R > 4u * (lane >> 5) ? R - 4u * (lane >> 5) : 0u;
        asm volatile("" : "+v"(rlim));
        if (kb < bars) {
            if (R == (uint32_t) RB) {                                           // whole block (uniform): no row checks, one basic block
                asm volatile("" : "+v"(loff));
#pragma unroll
                for (uint32_t g = 0; g < G; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            asm volatile("" : "+s"(ro));
                            st<OutT>(static_cast<char*>(bars_out) + ro, loff, out_of(res[g][4 * q + i]));
                            ro += i == 3 ? 5 * one_row : one_row;
                        }
            } else {
#pragma unroll
                for (uint32_t g = 0; g < G; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            asm volatile("" : "+s"(ro));
                            if (32u * g + 8u * (uint32_t) q + (uint32_t) i < rlim) {
                                uint32_t lo = loff;
                                asm volatile("" : "+v"(lo));
                                st<OutT>(static_cast<char*>(bars_out) + ro, lo, out_of(res[g][4 * q + i]));
                            }
                            ro += i == 3 ? 5 * one_row : one_row;
                        }
            }
        }
        __syncthreads();
        ++t;
    };
    // The weight stream is requested and awaited by hand: behind a bank's three requests at least the other two banks' six have been issued
    // by the time the bank is used, and loads retire in order: vmcnt(6).  (Left to the compiler, every control-flow merge behind a tile's end
    // made the next wait a wait for everything, the fragments requested a moment ago included: an L2 round trip per tile.)  The compiler does
    // not know these loads are in flight: its own waits (for the ring's texels, the epilogue's constants) can only come out stricter than
    // necessary, never too lax.
    auto wload = [&](glv_i4v& d0, glv_i4v& d1, glv_i4v& d2) {
        asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:1024\n\tglobal_load_dwordx4 %2, %3, off offset:2048"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2) : "v"(wp) : "memory");
        wp += 3 * 64;
    };
    auto step = [&](auto BC) {                                                  // one step of 32 bins on bank B
        constexpr int B = decltype(BC)::value;
        glv_i4v ah[G], al[G];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            ah[g] = *reinterpret_cast<const glv_i4v*>(plane_h + g * 32u * PITCH + arow + ck * 16u);
            al[g] = *reinterpret_cast<const glv_i4v*>(plane_l + g * 32u * PITCH + arow + ck * 16u);
        }
        ck = ck + 2u >= S16 ? ck + 2u - S16 : ck + 2u;
        if (fresh != 0) --fresh;
        else asm volatile("s_waitcnt vmcnt(6)" : "+v"(wb[B][0]), "+v"(wb[B][1]), "+v"(wb[B][2]) : : "memory");
        const glv_i4v w0 = wb[B][0], w1 = wb[B][1], w2 = wb[B][2];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            acc[g][3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah[g], w2, acc[g][3], 0, 0, 0);
            acc[g][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah[g], w1, acc[g][2], 0, 0, 0);
            acc[g][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah[g], w0, acc[g][1], 0, 0, 0);
            acc[g][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al[g], w0, acc[g][0], 0, 0, 0);
        }
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
            acc[g][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al[g], w2, acc[g][2], 0, 0, 0);
            acc[g][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al[g], w1, acc[g][1], 0, 0, 0);
        }
        // the bank's next step, PF steps on: requested once its MFMAs have been issued (they read the registers when they issue)
        asm volatile("" : : "v"(acc[0][1]) : "memory");
        wload(wb[B][0], wb[B][1], wb[B][2]);
    };
    fnext = fin[Mn.k0 + (lane & 31u)];                                          // the first tile's epilogue constants
    asm volatile("" : "+v"(fnext.c), "+v"(fnext.s));
    if (!next_tile()) return;
    // the wave's first tile: fill the pipeline
    wp = wq + (uint32_t) __builtin_amdgcn_readfirstlane((int) M.w_off) + lane;
#pragma unroll
    for (int b = 0; b < PF; ++b) wload(wb[b][0], wb[b][1], wb[b][2]);
    for (;;) {
        step(std::integral_constant<int, 0>{});
        if (--left == 0) { close_tile(); if (!next_tile()) break; }
        step(std::integral_constant<int, 1>{});
        if (--left == 0) { close_tile(); if (!next_tile()) break; }
        step(std::integral_constant<int, 2>{});
        if (--left == 0) { close_tile(); if (!next_tile()) break; }
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
template <int S, int RB>
static hipError_t launch_bars_rows(const float* spec, float* bars_out, size_t nrows, uint32_t n, uint32_t bars, const BarRowsTables& rt, hipStream_t st, int r) {
    const size_t lds = sizeof(float) * (size_t) RB * S;
    static std::atomic<bool> done[64] = {};
    if (lds > 64 * 1024) {
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(glv_bars_rows_kernel<S, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
        }
    }
    if (nrows == 0) return hipSuccess;                                          // prepare_bars_rows: the attribute only
    // RB rows per workgroup in x, ranges of rounds in y: 512 workgroups = two per CU, once (a range start refills the whole ring; N = 4096,
    // ms with 1 / 2 / 4 / 8 / 16 ranges: 32 K rows 0.321 / 0.328 / 0.337 / 0.351 / 0.370, 8 K rows 0.213 / 0.137 / 0.093 / 0.099 / 0.102,
    // 2 K rows 0.206 / 0.131 / 0.072 / 0.048 / 0.045)
    const uint32_t xb = (uint32_t) ((nrows + RB - 1) / RB);
    uint32_t yb = xb >= 512 ? 1 : (512 + xb - 1) / xb;
#if defined(GLV_TUNE_BUILD)
    if (const char* o = std::getenv("GLV_ROWS_YB")) yb = (uint32_t) atoi(o);       // tools/rows_bench: the split of the rounds over blockIdx.y
#endif
    if (yb > rt.nrounds) yb = rt.nrounds;
    const uint32_t rpw = (rt.nrounds + yb - 1) / yb;
    yb = (rt.nrounds + rpw - 1) / rpw;
    hipLaunchKernelGGL((glv_bars_rows_kernel<S, RB>), dim3(xb, yb), dim3(64 * kRowsWaves), lds, st, spec, static_cast<void*>(bars_out), nrows, n, bars, rt.rounds,
                       rt.nrounds, rpw, rt.mtiles, rt.wt, rt.wsum, r);
    return hipGetLastError();
}

// the > 64 KiB dynamic-LDS opt-in of the rows kernel that launch_bars would pick for rt, set ahead of the first launch (a process call
// is then a plain launch)
hipError_t prepare_bars_rows(uint32_t n, const BarRowsTables* rt) {
    if (rt == nullptr || rt->rounds == nullptr || rt->nrounds == 0) return hipSuccess;
    if (rt->ring_bins == 160) return launch_bars_rows<160, GLV_ROWS_RB>(nullptr, nullptr, 0, n, 0, *rt, nullptr, 0);
    if (rt->ring_bins == 288) return launch_bars_rows<288, GLV_ROWS_RB>(nullptr, nullptr, 0, n, 0, *rt, nullptr, 0);
    if (rt->ring_bins == 448) return launch_bars_rows<448, 32>(nullptr, nullptr, 0, n, 0, *rt, nullptr, 0);
    if (rt->ring_bins == 832) return launch_bars_rows<832, 32>(nullptr, nullptr, 0, n, 0, *rt, nullptr, 0);
    return hipSuccess;
}

// the i8 kernel for ring_bins in {160, 288, 448, 832} (64 rows per workgroup) or 1600 (32 rows: the bars of n = 32768); nrows == 0: the
// dynamic-LDS attribute only
template <int S, int RB, bool F32IN, bool R16>
static hipError_t launch_bars_i8_one(const void* rows, void* bars_out, size_t nrows, uint32_t n, uint32_t bars, const BarIRowsTables& rt, hipStream_t st) {
    const size_t lds = (size_t) 2 * RB * (S + 16);
    static std::atomic<bool> done[64] = {};
    if (lds > 64 * 1024) {
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(glv_bars_rows_i8_kernel<S, RB, F32IN, R16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
        }
    }
    if (nrows == 0) return hipSuccess;
    // RB rows per workgroup in x, ranges of rounds in y (512 workgroups, as glv_bars_rows_kernel)
    const uint32_t xb = (uint32_t) ((nrows + RB - 1) / RB);
    uint32_t yb = xb >= 512 ? 1 : (512 + xb - 1) / xb;
#if defined(GLV_TUNE_BUILD)
    if (const char* o = std::getenv("GLV_ROWS_YB")) yb = (uint32_t) atoi(o);
#endif
    if (yb > rt.nrounds) yb = rt.nrounds;
    const uint32_t rpw = (rt.nrounds + yb - 1) / yb;
    yb = (rt.nrounds + rpw - 1) / rpw;
    hipLaunchKernelGGL((glv_bars_rows_i8_kernel<S, RB, F32IN, R16>), dim3(xb, yb), dim3(64 * kRowsWaves), lds, st, rows, bars_out, nrows, n, bars, rt.rounds, rt.nrounds, rpw,
                       rt.tiles, reinterpret_cast<const glv_i4v*>(rt.wq), rt.fin);
    return hipGetLastError();
}
template <bool F32IN, bool R16>
static hipError_t launch_bars_i8_in(const void* rows, void* bars_out, size_t nrows, uint32_t n, uint32_t bars, const BarIRowsTables& rt, hipStream_t st) {
    switch (rt.ring_bins) {
        case 160: return launch_bars_i8_one<160, 64, F32IN, R16>(rows, bars_out, nrows, n, bars, rt, st);
        case 288: return launch_bars_i8_one<288, 64, F32IN, R16>(rows, bars_out, nrows, n, bars, rt, st);
        case 448: return launch_bars_i8_one<448, 64, F32IN, R16>(rows, bars_out, nrows, n, bars, rt, st);
        case 832: return launch_bars_i8_one<832, 64, F32IN, R16>(rows, bars_out, nrows, n, bars, rt, st);
        case 1600: return launch_bars_i8_one<1600, 32, F32IN, R16>(rows, bars_out, nrows, n, bars, rt, st);
    }
    return hipErrorInvalidValue;
}
// rows: uint16 [nrows][n] texels (rows_f32 false) or float [nrows][n] holding texel values c / 65535 (true)
hipError_t launch_bars_i8(const void* rows, bool rows_f32, void* bars_out, size_t nrows, uint32_t n, uint32_t bars, const BarIRowsTables* rt, hipStream_t st, bool r16) {
    if (rt == nullptr || rt->tiles == nullptr || rt->rounds == nullptr || rt->nrounds == 0) return hipErrorInvalidValue;
    if (rows_f32) return r16 ? launch_bars_i8_in<true, true>(rows, bars_out, nrows, n, bars, *rt, st) : launch_bars_i8_in<true, false>(rows, bars_out, nrows, n, bars, *rt, st);
    return r16 ? launch_bars_i8_in<false, true>(rows, bars_out, nrows, n, bars, *rt, st) : launch_bars_i8_in<false, false>(rows, bars_out, nrows, n, bars, *rt, st);
}
hipError_t prepare_bars_i8(uint32_t n, const BarIRowsTables* rt) {
    if (rt == nullptr || rt->rounds == nullptr || rt->nrounds == 0) return hipSuccess;
    hipError_t e = launch_bars_i8_in<false, true>(nullptr, nullptr, 0, n, 0, *rt, nullptr);
    if (e == hipSuccess) e = launch_bars_i8_in<false, false>(nullptr, nullptr, 0, n, 0, *rt, nullptr);
    if (e == hipSuccess) e = launch_bars_i8_in<true, true>(nullptr, nullptr, 0, n, 0, *rt, nullptr);
    if (e == hipSuccess) e = launch_bars_i8_in<true, false>(nullptr, nullptr, 0, n, 0, *rt, nullptr);
    return e;
}

hipError_t launch_bars(const float* spec, float* bars_out, size_t nrows, uint32_t n, uint32_t bars, uint32_t nsteps,
                       const BarItem* items, const BarDesc* desc, const float* tap_w, hipStream_t st, bool r16, const BarRowsTables* rt) {
    const int r = r16 ? 1 : 0;
    // SAMPLE_MODE maximum / hybrid: one lane per bar and row, any number of bars
    if (rt != nullptr && rt->mode == 1u) return launch_bars_mode<1>(spec, bars_out, nrows, n, bars, desc, *rt, st, r);
    if (rt != nullptr && rt->mode == 2u) return launch_bars_mode<2>(spec, bars_out, nrows, n, bars, desc, *rt, st, r);
    // many bars: one fma chain per bar (glv_tables.h make_bar_mtiles) -- on the matrix cores when the host could cut the tiles into
    // rounds for the LDS ring, one lane per bar otherwise
    if (bars >= 256) {
        if (rt == nullptr || rt->mtiles == nullptr || rt->ntiles == 0) return hipErrorInvalidValue;
        if (rt->rounds != nullptr && rt->nrounds != 0 && nrows >= kRowsMin) {
            // (the long bars of n = 8192 / 16384 need a longer ring: 32 rows per workgroup there, one MFMA per step)
            if (rt->ring_bins == 160) return launch_bars_rows<160, GLV_ROWS_RB>(spec, bars_out, nrows, n, bars, *rt, st, r);
            if (rt->ring_bins == 288) return launch_bars_rows<288, GLV_ROWS_RB>(spec, bars_out, nrows, n, bars, *rt, st, r);
            if (rt->ring_bins == 448) return launch_bars_rows<448, 32>(spec, bars_out, nrows, n, bars, *rt, st, r);
            if (rt->ring_bins == 832) return launch_bars_rows<832, 32>(spec, bars_out, nrows, n, bars, *rt, st, r);
        }
        const size_t units = nrows * (size_t) ((rt->ntiles + 1u) / 2u);
        const size_t wgs = (units + 3) / 4;
        hipLaunchKernelGGL(glv_bars_seq_kernel, dim3((unsigned) (wgs < 256 * 16 ? (wgs ? wgs : 1) : 256 * 16)), dim3(256), 0, st, spec, static_cast<void*>(bars_out),
                           nrows, n, bars, rt->mtiles, rt->ntiles, rt->wt, rt->wsum, r);
        return hipGetLastError();
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
