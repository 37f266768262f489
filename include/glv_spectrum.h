/*
 * glv_spectrum.h -- C ABI of the MI355X-native GLava audio-spectrum path.
 *
 * Drop-in boundary for GLava's  fifo/pulse_input -> window -> FFT -> magnitude ->
 * gravity/average  pipeline.  Plain C: pointers, sizes, int status
 * returns.  No exceptions, no exit(), no torch/HIP types (a HIP stream is passed as
 * `void*`).  Every entry point names the reference interface it replaces; paths are
 * relative to the jarcode-foss/glava tree.
 *
 * The library is libglvspectrum.so (glava_amd/csrc), hand-written HIP for gfx950.  There
 * is NO CPU fallback: without a usable HIP device every compute entry point returns
 * GLV_ERR_NO_DEVICE and glv_last_error() says why.
 */
#ifndef GLV_SPECTRUM_H
#define GLV_SPECTRUM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLV_ABI_VERSION 7      /* 5 (round 5): + glv_gl_texture; GLV_OP_BARS over texel rows (gl_storage != 0, 256 bars or more) is the exact integer mean
                                  6 (round 6): + GLV_OP_BARS_ONLY, glv_batch_live_bins, glv_batch_bars_arithmetic, glv_batch_tune_placement; GLV_OP_R16 in a creation mask is a hint
                                  7 (round 6): glv_params grew smooth_audio()'s shape -- round_formula, sample_mode, sample_hybrid_weight, sample_scale, sample_range
                                               (appended; all-zero == the shipped shape, so a caller that zero-fills the tail keeps ABI 6's results); GLV_BARS_F32_SEQ */

/* status codes (0 = ok).  The reference has no error channel: it prints and calls
 * glava_abort() (glava/glava.h:17, glava/render.c passim); the in-tree shim maps any
 * non-zero status to that. */
enum {
    GLV_OK = 0,
    GLV_ERR_INVALID = 1,     /* bad argument (n not a power of two in [256,32768], NULL, ...) */
    GLV_ERR_NO_DEVICE = 2,   /* no HIP device / kernels for gfx950 not loadable */
    GLV_ERR_HIP = 3,         /* a HIP runtime call failed; see glv_last_error() */
    GLV_ERR_NOMEM = 4,
    GLV_ERR_STATE = 5        /* operator used on a batch/state that lacks the required buffers */
};

/* Operator selection for the batched calls (bit set).  The order of application is
 * fixed and is the reference's: fft -> gravity -> average (glava/render.c:2140-2156). */
enum {
    GLV_OP_FFT      = 1u << 0,  /* window + FFT + abs/log/tilt   == transform_fft     render.c:783-847 */
    GLV_OP_GRAVITY  = 1u << 1,  /*                               == transform_gravity render.c:720-736 */
    GLV_OP_AVERAGE  = 1u << 2,  /*                               == transform_average render.c:738-771 */
    GLV_OP_RAW      = 1u << 3,  /* testing aid: with GLV_OP_FFT, skip abs/log/tilt and emit the raw
                                   interleaved (Re,Im) FFT output, which must be bit-identical to the
                                   reference's data[] at render.c:840 */
    GLV_OP_WRANGE   = 1u << 4,  /* (b+1)/2                       == transform_wrange  render.c:773-781;
                                   exclusive with GLV_OP_FFT (the wave module requests window,wrange) */
    GLV_OP_BARS     = 1u << 5,  /* smooth_audio() bin averaging + bar lookup (shaders/glava/util/
                                   smooth.glsl:13-64, radial/1.frag:58-70): emit `bars` values per
                                   channel instead of n bins; d_out is float [streams][2][bars].
                                   Inputs are clamped to [0, 1] (NaN -> 0) like the GL_R16 texels the shader
                                   samples; the taps' products are summed in a documented, fixed order (below 256
                                   bars: chunks of 16 / 32 / 64 taps by n, fused multiply-add chains; from 256 bars up -- the
                                   pre-smoothing pass, bars = n -- ONE fused multiply-add chain per bar in bin order, which the
                                   matrix cores compute: glava_amd/csrc/glv_frame.h "GLV_OP_BARS arithmetic";
                                   oracle/glv_oracle.c glvo_bars_chunked restates both) -- within 2e-4 relative of the
                                   shader's tap-by-tap loop, identical bits on every device path.  A bar whose weights
                                   sum to 0 is 0 / 0 as in the shader.
                                   Inside the GL chains (gl_storage != 0, gravity / average in the chain: the rows the bars sample are
                                   GL_R16 TEXELS, as in the reference's pre-smoothing pass, render.c:2277-2303) 256 bars or more are computed
                                   EXACTLY (ABI 5): per bar the shader's float weights w_j become integers W_j = llrint(w_j 2^P / sum w) with the
                                   first largest taking the residue so that sum W_j == 2^P (P = max(17, 21 + ceil(log2 sum w))), and
                                   texel = floor(sum W_j c_j / 2^P + 1/2) on the 16-bit texels c_j; without GLV_OP_R16 the float
                                   (float) ((double) sum W_j c_j 2^-P / 65535).  Integer arithmetic on the i8 matrix cores: no summation order,
                                   within 0.03 texel steps of the mean with the float weights in exact arithmetic
                                   (glava_amd/csrc/glv_tables.h make_bar_itiles; oracle/glv_oracle.c glvo_bars_int_at restates it) */
    GLV_OP_SMOOTH   = 1u << 6,  /* CPU-path log-window mean  == transform_smooth   render.c:694-718;
                                   applied last, in place on each row (after fft/gravity/average) */
    GLV_OP_MAGNITUDE = 1u << 7, /* the magnitude stage alone: b = (float)(log(|b| + 1.0f) / 3) * tilt(i),
                                   the tail of transform_fft (render.c:842-846) on planar f32 rows that
                                   already hold FFT output; exclusive with GLV_OP_FFT (which includes it);
                                   runs before gravity/average when combined with them */
    GLV_OP_R16      = 1u << 8,  /* output as GL_R16 texels: d_out is uint16 [streams][2][n] with
                                   texel = round_to_nearest_even(clamp(x, 0, 1) * 65535) -- what the only consumer of
                                   the spectra makes of them (handle_audio uploads every finished buffer with
                                   glTexImage1D(GL_TEXTURE_1D, 0, GL_R16, sz, 0, GL_RED, GL_FLOAT, buf), render.c:521-524).
                                   Applied last, to the output only: gravity / average state stays f32.  Halves the
                                   write side of the pass (8n instead of 12n bytes per s16 frame).  Alone (no other
                                   op) it quantises planar f32 rows.  With GLV_OP_BARS the bars are what is quantised (d_out:
                                   uint16 [streams][2][bars]; the spectra feeding them stay f32): with gl_storage = 1, bars = n and
                                   bar_phase = 0.5 that is the texture every stock module samples -- upload, gravity, average and
                                   pre-smoothing pass of render.c:2188-2303 in one call.  Excludes GLV_OP_RAW, GLV_OP_SMOOTH.
                                   In glv_batch_create's ops_mask (ABI 6) the bit is a HINT and nothing else: a FLOAT chain (gl_storage 0)
                                   with GLV_OP_BARS whose bars the transform kernel can compute itself gets the internal spectra rows --
                                   which only its bars AS TEXELS need (they leave through a second launch) -- only when the mask carries
                                   GLV_OP_R16 too; without it such a call is refused (GLV_ERR_STATE), with it nothing else changes. */
    GLV_OP_PRIVATE_STATE = 1u << 9, /* (ABI 3 flag, accepted and ignored since ABI 4: a batch-owned gravity state is the default again) */
    GLV_OP_RING_S16 = 1u << 10, /* glv_batch_create's ops_mask only: allocate (and zero, == the calloc'd rings of
                                   glava.c:487-494) the s16 device ring of glv_batch_ring_update_s16 / _append_s16; ring calls on
                                   a batch created without it are refused (GLV_ERR_STATE: nothing is allocated after creation) */
    GLV_OP_RING_F32 = 1u << 11, /* the same for the interleaved f32 ring of glv_batch_ring_update_f32 */
    GLV_OP_BARS_ONLY = 1u << 13, /* glv_batch_create's ops_mask only (ABI 6), with GLV_OP_BARS: the caller promises that every stateful call on this
                                   batch asks for GLV_OP_BARS -- the bars are all that is ever looked at (GLava's shipped pipeline: the modules sample
                                   the pre-smoothed texture and nothing else, smooth.glsl:62; BASELINE configs[2]: the radial module's bars).
                                   smooth_audio() samples bins below scale_audio(1) * n plus half a window (0.288 n with the shipped
                                   SAMPLE_RANGE 0.9 / SAMPLE_SCALE 8; glv_params.sample_range / sample_scale), so what the reference's passes compute beyond that is dead: the chain then keeps its gravity
                                   store and ring, and computes magnitude / upload / gravity / average, only for a compile-time share of the row
                                   that covers the bins the bars sample (3/8 of it; glv_batch_live_bins tells whether the batch runs that way) --
                                   15 n instead of 28 n bytes per frame at n = 4096, F = 5 on GL_R16 state.  Live kernel classes exist for the
                                   GL_R16 chains (gl_storage 1: bars in a second launch or fused) and for float chains whose bars the transform
                                   kernel computes itself (fewer than 256 bars, every kernel configuration of the size fusable); any other
                                   batch (and log_mode 2) runs the full chain: the bars are bit-identical either way (no smooth_factor makes a
                                   bar sample further: smooth_audio() clamps its positions to [0, 1]).  A stateful call without GLV_OP_BARS and glv_batch_gravity_state are refused
                                   (GLV_ERR_STATE); so is a glv_batch_set_params that would take a batch which has run its live class to the
                                   full chain (the state beyond the live bins was never kept) -- until glv_batch_reset */
    GLV_OP_OUTPUT_IS_STATE = 1u << 12 /* opt-in, with a chain that ENDS in gravity (GLV_OP_GRAVITY without AVERAGE / SMOOTH / RAW, f32
                                   rows out, no gl_storage): transform_gravity stores every value twice, to its `applied` array and to
                                   the buffer (render.c:733-734) -- with this flag ONE array is kept: the call writes the spectra
                                   once, into d_out, and the NEXT update of the batch reads its `applied` values from there.  The
                                   caller promises to leave that buffer intact until the next update has been issued (the same
                                   buffer every time, or several in rotation; stream order is all the synchronisation needed) and
                                   not to alias it with the input.  The chain then moves SURVEY 8d row B's 20 n bytes per frame
                                   instead of 28 n.  Without the flag the state lives in a buffer the batch owns (the default: a
                                   caller may post-process, reuse or free d_out at will) */
};
/* Bits glv_batch_create's ops_mask understands: GLV_OP_GRAVITY / GLV_OP_AVERAGE (state arrays), GLV_OP_BARS (the internal
 * spectra rows, where the announced chain cannot compute its bars inside the transform's launch), GLV_OP_SMOOTH, GLV_OP_RING_S16 /
 * GLV_OP_RING_F32 (device rings).  The tables are cheap and always made at creation (tilt, bar taps and work lists, smooth
 * bounds: an operator the mask did not announce only goes without them when its parameters are unusable); buffers of spectrum
 * size are made for announced operators only.  The process calls themselves never allocate, never copy synchronously, change no
 * function attribute, and can be captured into a hipGraph from the first one on; an operator whose buffers the batch was not
 * created with is refused with GLV_ERR_STATE. */

/* Mirrors the fields of the private `struct gl_data` that the path reads
 * (glava/render.c:166-207) and of `struct audio_data` (glava/fifo.h:9-20). */
typedef struct glv_params {
    uint32_t n;             /* audio_buf_sz / bsz: real samples per channel; power of two in [256, 32768]
                               (#request setbufsize, render.c:1176).  Complex FFT length is n/2 (render.c:786) */
    uint32_t channels;      /* 2 = stereo, 1 = mirror/mono mix (fifo.c:98-102, setmirror render.c:1053) */
    float fft_scale;        /* render.c:845, default 10.2 (render.c:930) */
    float fft_cutoff;       /* render.c:845, default 0.3  (render.c:931) */
    float gravity_step;     /* render.c:728, default 4.2  (render.c:911) */
    float ur;               /* updates per second used by gravity (render.c:728); the reference measures it
                               (render.c:2387); ideal value rate/(sample_sz/4).  Like the reference, any value is
                               taken: 0 (an interval without updates) makes the step infinite and the output -inf */
    uint32_t avg_frames;    /* F, render.c:743; 1..GLV_MAX_AVG_FRAMES */
    uint32_t avg_window;    /* bool, render.c:745/765 */
    uint32_t avg_window_kind; /* 0: CPU twin 0.6/0.4, oldest-first (render.c:661,751-766)
                                 1: GL twin 0.53836/0.46164, newest-first (common.glsl:13, average_pass.frag:19-45,
                                    render.c:2247-2256) */
    uint32_t log_mode;      /* render.c:844 evaluates log() in fp64 and divides by 3 in fp64, then rounds to float:
                               1: (default) hardware log2 * (ln2/3 * tilt) in fp32: <= 1.8e-7 relative error
                                  against the reference's float on EVERY float input of the stage (exhaustive
                                  test, tests/test_gpu_parity.py::test_magnitude_stage_every_float; bar 1e-5).  At n >= 16384
                                  the factor ln2/3 * tilt itself comes from one fused multiply-add per value (<= 4.5e-7 from
                                  the reference's thrice-rounded product): <= 6.3e-7 in total
                               0: bit-faithful: fp64 table-driven log (rel. error ~2^-50) and fp64 /3 -- the
                                  reference's float result on every input of the same exhaustive test
                               2: audit: device libm fp64 log + true fp64 division (slow) */
    /* GLV_OP_BARS parameters (shaders/glava/smooth_parameters.glsl) */
    uint32_t bars;          /* bars per channel (radial.glsl:9 NBARS 160 => 80) */
    float smooth_factor;    /* SMOOTH_FACTOR, smooth_parameters.glsl:72, default 0.025 */
    /* GLV_OP_SMOOTH parameters (render.c:917-918, #request setsmooth / setsmoothratio render.c:1201-1206) */
    float smooth_distance;  /* default 0.01 */
    float smooth_ratio;     /* default 4 */
    uint32_t gl_storage;    /* 0 (default): gravity / average keep float state, like the CPU operators (render.c:720-771).
                               1: the storage of the GL passes is modelled (render.c:2188-2265 with setaccelfft, GLava's shipped
                               default rc.glsl:211: every intermediate is a GL_R16 texture, render.c:523, :1718): the uploaded
                               buffer, the gravity store after each step and the average are clamped to [0, 1] and quantised to 16
                               bits; no averaging pass when avg_frames == 1 (render.c:2230).  Usually combined with
                               avg_window_kind = 1.  The state IS kept as 16-bit texels (uint16 store and ring) and an FFT chain
                               runs as ONE launch: upload quantisation, GL_MAX + gravity pass, ring copy and average pass are the
                               transform kernel's epilogue (with GLV_OP_BARS, where a row's bars fit, also the bars); the output is
                               GL_R16 texels (GLV_OP_R16) or their floats c / 65535.  28 n bytes per frame at avg_frames = 5.
                               2: the same values, pass by pass as the reference runs them: the transform writes f32 spectra, a
                               second kernel applies gravity / average on f32 state (every value a float c / 65535).  Bit-identical
                               to 1 on every input; kept as the checker of 1 and for the chains 1 does not fuse.  The storage
                               class (1 vs 0 / 2) is fixed when a batch or state is created. */
    float bar_phase;        /* GLV_OP_BARS evaluates smooth_audio() at idx = (k + bar_phase) / bars, k = 0 .. bars-1.
                               0 (default): the bar positions of the modules (radial/1.frag:58-70: pos = k / (NBARS / 2)).
                               0.5 with bars == n: the texel centres of the reference's pre-smoothing pass
                               (util/smooth_pass.frag: smooth_audio(tex, sz, gl_FragCoord.x / w), render.c:2277-2303) -- the
                               texture every stock module samples when setsmoothpass is on (the default) */
    /* smooth_audio()'s SHAPE (ABI 7): the GLSL `#define`s of shaders/glava/smooth_parameters.glsl:17-42, which a user overrides in
     * ~/.config/glava/smooth_parameters.glsl or a module's config (the reference re-defines them textually, glsl_ext.c:143-157).  0 in every
     * field is the shipped shape.  GLV_OP_BARS evaluates exactly this smooth_audio(); integration/glava_hip_shim.c reads the defines out of
     * the processed shader text the host compiles. */
    uint32_t round_formula; /* ROUND_FORMULA (smooth_parameters.glsl:17, the macros of util/common.glsl:17-22):
                               GLV_ROUND_SINUSOIDAL 0 (shipped), GLV_ROUND_CIRCULAR 1, GLV_ROUND_LINEAR 2 */
    uint32_t sample_mode;   /* SAMPLE_MODE (smooth_parameters.glsl:28; smooth.glsl:9-11, :32-59):
                               GLV_SAMPLE_AVERAGE 0 (shipped): the weighted mean over s = smin; s <= smax -- every arithmetic GLV_OP_BARS documents above;
                               GLV_SAMPLE_MAXIMUM 1: max over s = smin; s < smax of fl(x_j * w_j), from +0 (order-free: bit-exact by construction);
                               GLV_SAMPLE_HYBRID  2: fl(fl(vmax * fl(1 - H)) + fl(fl(avg / weight) * H)) with v_j = fl(x_j * w_j), vmax their maximum,
                                  avg = ONE chain of float additions of the v_j in bin order from +0 (the shader's loop as written, no fused
                                  multiply-add), weight the tap-order float sum of the w_j; s < smax.
                               Modes 1 and 2 run one lane per bar and row (glv_bars_mode_kernel), never inside the transform's launch and never on the
                               matrix cores (a maximum is not a matrix product); texel rows (gl_storage != 0) enter as x = c / 65535 in float (what
                               texelFetch returns) and leave, with GLV_OP_R16, as round_to_nearest_even(clamp(bar, 0, 1) * 65535) */
    float sample_hybrid_weight; /* SAMPLE_HYBRID_WEIGHT, smooth_parameters.glsl:31; 0 = the shipped 0.65; (0, 1] */
    float sample_scale;     /* SAMPLE_SCALE, smooth_parameters.glsl:35; 0 = the shipped 8 */
    float sample_range;     /* SAMPLE_RANGE, smooth_parameters.glsl:40; 0 = the shipped 0.9.  scale_audio(1) = -log(1 - range) / scale must not
                               exceed 1 (the shader would fetch texels beyond the texture: undefined in GL, GLV_ERR_INVALID here) */
} glv_params;
enum { GLV_ROUND_SINUSOIDAL = 0, GLV_ROUND_CIRCULAR = 1, GLV_ROUND_LINEAR = 2 };
enum { GLV_SAMPLE_AVERAGE = 0, GLV_SAMPLE_MAXIMUM = 1, GLV_SAMPLE_HYBRID = 2 };

#define GLV_MAX_AVG_FRAMES 64

/* Fill `p` with the shipped defaults (shaders/glava/rc.glsl:181-211,
 * shaders/glava/smooth_parameters.glsl:46-67): n=4096, stereo, 10.2/0.3, 4.2,
 * ur = 22050/256, F=5 windowed. */
void glv_params_default(glv_params* p);

/* Version / diagnostics. */
int         glv_abi_version(void);
const char* glv_last_error(void);          /* thread-local, never NULL */
int         glv_device_count(void);        /* 0 when no HIP device is usable */

/* ------------------------------------------------------------------------------------
 * 1. Single-stream drop-ins (host pointers, in place).  Same shape as the reference's
 *    operator seam  void apply(struct gl_data*, void** udata, void* data)
 *    (glava/render.c:106-118, table at :849-856).  `glv_state` replaces the `*udata`
 *    slot (lazily calloc'd by the reference, render.c:662-666, and free()d by
 *    rd_destroy, render.c:2463-2469); here it is created/destroyed explicitly.
 * ------------------------------------------------------------------------------------ */
typedef struct glv_state glv_state;   /* per (channel, stream) device-side gravity + history state */

int glv_state_create(const glv_params* p, int device, glv_state** out);
int glv_state_reset(glv_state* s);            /* zero gravity/history (== fresh calloc) */
int glv_state_destroy(glv_state* s);

/* == transform_fft(gl, _, &(struct gl_sampler_data){buf, p->n})   glava/render.c:783-847 */
int glv_fft(const glv_params* p, glv_state* s, float* buf);
/* == transform_gravity                                             glava/render.c:720-736 */
int glv_gravity(const glv_params* p, glv_state* s, float* buf);
/* == transform_average                                             glava/render.c:738-771 */
int glv_average(const glv_params* p, glv_state* s, float* buf);
/* == transform_wrange                                              glava/render.c:773-781 */
int glv_wrange(const glv_params* p, glv_state* s, float* buf);
/* == transform_smooth                                              glava/render.c:694-718 */
int glv_smooth(const glv_params* p, glv_state* s, float* buf);
/* == the magnitude tail of transform_fft alone (abs, +1, log/3, tilt) glava/render.c:842-846 */
int glv_magnitude(const glv_params* p, glv_state* s, float* buf);
/* fft -> gravity -> average in one launch (what handle_audio does per bind, render.c:2140-2156) */
int glv_fft_gravity_average(const glv_params* p, glv_state* s, float* buf);
/* the GL_R16 texels glTexImage1D(..., GL_R16, ..., GL_FLOAT, buf) stores for buf[0..n) (render.c:521-524):
 * texels[i] = round_to_nearest_even(clamp(buf[i], 0, 1) * 65535); buf is not modified */
int glv_texels_r16(const glv_params* p, glv_state* s, const float* buf, uint16_t* texels);
/* == the accel_fft branch of handle_audio, from the per-frame transform_fft to the texture the module samples (glava/render.c:2176-2303,
 * GLava's shipped configuration: rc.glsl:211 setaccelfft true, smooth_parameters.glsl:78 setsmoothpass true): transform_fft of the n samples
 * in buf (not modified), the GL_R16 upload (:521-524), GL_MAX store + gravity pass (:2199-2228), ring copy + average pass (:2230-2265; none
 * when avg_frames == 1) and -- smooth_pass != 0 -- the pre-smoothing pass (:2277-2303, util/smooth_pass.frag) on 16-bit state: one launch
 * (+ one for the pre-smoothing pass).  texels: the n GL_R16 texels the last pass's render target holds -- upload them with
 * glTexImage1D(GL_TEXTURE_1D, 0, GL_R16, n, 0, GL_RED, GL_UNSIGNED_SHORT, texels).  The state must have been created with gl_storage = 1
 * (normally avg_window_kind = 1; for the pre-smoothing pass bars = n and bar_phase = 0.5); it holds the gravity store and the ring like
 * the reference's gr_store / gr.out textures do.  integration/render_hip.patch binds this into handle_audio. */
int glv_gl_texture(const glv_params* p, glv_state* s, const float* buf, int smooth_pass, uint16_t* texels);

/* == the unpack loop of the FIFO backend, glava/fifo.c:94-110 (and :67-79 when pcm == NULL):
 * `frames` interleaved stereo s16 frames -> planar f32.  Runs on the device (the bit-exactness
 * of int16 -> f32 /65535f is part of the parity contract). */
int glv_unpack_s16(int device, const int16_t* pcm, size_t frames, int channels, float* l, float* r);

/* rd_update prelude on device buffers (glava/render.c:1765-1809); stream-ordered, no state:
 *   bufscale  d_out[r][t] = mean(d_in[r][t*k .. t*k+k-1]), float accumulation (render.c:1768-1781)
 *   lerp      d_out = d_start + (d_end - d_start) * min(uratio * kcounter, 1)   (render.c:1794-1809) */
int glv_prelude_bufscale(int device, const float* d_in, float* d_out, size_t rows, uint32_t n_out, uint32_t k, void* hip_stream);
int glv_prelude_lerp(int device, const float* d_start, const float* d_end, float* d_out, size_t count,
                     float uratio, int kcounter, void* hip_stream);

/* ------------------------------------------------------------------------------------
 * 2. Batched extension (device pointers, stream-ordered).  No reference counterpart:
 *    B independent stereo streams, one update ("frame") of every stream per call.
 *    Layouts (row-major):
 *      d_pcm   int16 [streams][n][2]     interleaved LRLR, exactly the bytes a FIFO delivers
 *      d_f32   float [streams][2][n]     planar, what glava.c:528-537 snapshots into lb/rb
 *      d_spec  float [streams][2][n]     out; index 2k = |Re Z_k| path, 2k+1 = |Im Z_k| path
 *      d_bars  float [streams][2][bars]  out when GLV_OP_BARS
 *    State (gravity [streams][2][n], history ring [streams][2][F][n]) is owned by the
 *    batch and allocated according to `ops_mask` given at creation.
 * ------------------------------------------------------------------------------------ */
/* device memory for C hosts that do not want the HIP headers (the audio backend of INTEGRATION.md section 2):
 * thin wrappers over hipMalloc / hipFree / hipMemcpyAsync on `device`; copies are ordered on hip_stream
 * (NULL = the default stream) and glv_device_sync waits for it. */
int glv_device_malloc(int device, size_t bytes, void** d_ptr);
int glv_device_free(int device, void* d_ptr);
int glv_device_upload(int device, void* d_dst, const void* h_src, size_t bytes, void* hip_stream);
int glv_device_download(int device, void* h_dst, const void* d_src, size_t bytes, void* hip_stream);
int glv_device_sync(int device, void* hip_stream);

typedef struct glv_batch glv_batch;

int glv_batch_create(const glv_params* p, uint32_t streams, unsigned ops_mask, int device, glv_batch** out);
/* Change the scalar knobs of a batch (fft_scale, fft_cutoff, gravity_step, ur, avg_window*, log_mode, channels, bars,
 * smooth_*, bar_phase, gl_storage 0 <-> 2): regenerates and uploads whatever tables depend on them -- synchronously; this is the
 * only place besides creation that allocates or copies.  n, avg_frames and the state's storage class (gl_storage == 1 or not)
 * are fixed at creation (GLV_ERR_STATE).  State (gravity, history, rings) is kept. */
int glv_batch_set_params(glv_batch* b, const glv_params* p);
int glv_batch_reset(glv_batch* b);
int glv_batch_destroy(glv_batch* b);

/* one update of every stream from s16 PCM already resident in HBM.  Stream-ordered: nothing is allocated, nothing is copied
 * synchronously; the call can be captured into a hipGraph.
 *
 * Chains that end in gravity (GLV_OP_GRAVITY without AVERAGE / SMOOTH / RAW): the state lives in a buffer the batch owns and
 * d_out receives a copy (28 n bytes per frame); d_out may be NULL -- the spectra are then left in the state buffer only, read
 * them through glv_batch_gravity_state (20 n).  With GLV_OP_OUTPUT_IS_STATE d_out itself becomes the state (20 n; see the flag
 * for what the caller promises).  (With GLV_OP_BARS a chain ending in gravity keeps the spectra in the state internally.) */
int glv_batch_process_s16(glv_batch* b, const int16_t* d_pcm, float* d_out, unsigned ops, void* hip_stream);
/* device pointer to the gravity state float [streams][2][n] == the latest output of a chain ending in gravity: the
 * batch-owned buffer, or the caller's d_out of the latest call when that doubles as the state (GLV_OP_OUTPUT_IS_STATE).
 * GLV_ERR_STATE if the batch was created without GLV_OP_GRAVITY, after fused gravity + average calls (the state is
 * then the newest slot of the history ring, float [rows][F][n] -- not an array of this shape), and with gl_storage == 1 (the
 * state is uint16 texels). */
int glv_batch_gravity_state(glv_batch* b, const float** d_state);
/* same from planar f32 (the lb/rb snapshot) */
int glv_batch_process_f32(glv_batch* b, const float* d_f32, float* d_out, unsigned ops, void* hip_stream);

/* same from interleaved stereo f32 frames, float [streams][n][2] -- the layout the PulseAudio backend
 * receives (glava/pulse_input.c:155-178); channels == 1 mixes (L + R) / 2 in float (pulse_input.c:167) */
int glv_batch_process_f32_stereo(glv_batch* b, const float* d_pcm, float* d_out, unsigned ops, void* hip_stream);

/* FIFO ring mode (glava/fifo.c:91-112): the batch keeps an n-frame s16 ring per stream in
 * HBM; each call appends `new_frames` (= sample_sz/4, fifo.c:38,91) stereo frames per
 * stream (d_new int16 [streams][new_frames][2]; NULL => poll-timeout zero fill,
 * fifo.c:67-79), then transforms the whole window. */
int glv_batch_ring_update_s16(glv_batch* b, const int16_t* d_new, uint32_t new_frames, float* d_out,
                              unsigned ops, void* hip_stream);
/* the same ring kept in interleaved f32 frames -- the PulseAudio backend's update (glava/pulse_input.c:
 * 155-178: shift both rings left by sample_sz/4, append sample_sz/4 stereo f32 frames; channels == 1 mixes
 * (L + R) / 2 in float).  d_new: float [streams][new_frames][2]; no zero-fill path (pa_simple_read blocks). */
int glv_batch_ring_update_f32(glv_batch* b, const float* d_new, uint32_t new_frames, float* d_out, unsigned ops,
                              void* hip_stream);

/* The rings on their own (the audio-backend seam, glava/fifo.h:9-20): append an update without transforming, and read the
 * rings the way the reference's backends publish them in audio_out_l / audio_out_r -- planar f32, oldest sample first
 * (fifo.c:91-110: memmove + unpack; pulse_input.c:155-176: memmove + deinterleave; channels == 1: the mono mix in both).
 * d_planar: float [streams][2][n].  f32_ring: 0 = the s16 ring, 1 = the interleaved f32 ring.  A backend that keeps
 * struct audio_data's contract downloads this; one that publishes spectra calls glv_batch_ring_update_* instead. */
int glv_batch_ring_append_s16(glv_batch* b, const int16_t* d_new /* NULL = zero fill, fifo.c:67-79 */, uint32_t new_frames, void* hip_stream);
int glv_batch_ring_append_f32(glv_batch* b, const float* d_new, uint32_t new_frames, void* hip_stream);
int glv_batch_ring_planar(glv_batch* b, int f32_ring, float* d_planar, void* hip_stream);

/* smooth_audio() bar sampling of spectra already in HBM (d_spec float [streams][2][n]) into
 * d_bars float [streams][2][bars]; what GLV_OP_BARS runs after the transform. */
int glv_batch_bars(glv_batch* b, const float* d_spec, float* d_bars, void* hip_stream);

/* Kernel-time accounting for the roofline report: HIP events recorded on the caller's
 * stream around every launch between begin/end; returns accumulated milliseconds and the
 * number of launches of the dominant (FFT) kernel. */
int glv_batch_timing_begin(glv_batch* b);
int glv_batch_timing_end(glv_batch* b, double* kernel_ms, uint64_t* launches);

/* Algorithmic HBM bytes one process call moves for the given ops: what the CHAIN must move -- input, state
 * read and written, output.  Rows that one launch of a multi-launch chain hands to the next (the uint16 `av` rows in front of the
 * pre-smoothing pass, bars that are not computed inside the transform's launch) are traffic of the organisation, not of the problem, and
 * are not counted -- except gl_storage 2, whose pass-by-pass f32 round trip (+16 n) is its definition.  A chain that ends in gravity is
 * counted with its output copy (28 n) unless GLV_OP_OUTPUT_IS_STATE is in `ops` (20 n; a NULL d_out moves those 20 n too). */
uint64_t glv_batch_algorithmic_bytes(const glv_batch* b, unsigned ops, int input_is_s16);

/* GLV_OP_BARS_ONLY batches: the bins of a row the chain keeps alive, [0, glv_batch_live_bins) (what the bars sample, in whole store
 * instructions); 0 for every other batch (all n bins live).  The kernels round it up to their last pass's block. */
uint32_t glv_batch_live_bins(const glv_batch* b);

/* Launch-geometry override for tuning (workgroups of the persistent frame kernel; 0 = automatic). */
int glv_batch_set_grid(glv_batch* b, int grid);
int glv_batch_last_grid(const glv_batch* b);      /* workgroups the last frame-kernel launch of this batch used */
int glv_batch_last_launches(const glv_batch* b);  /* kernels the last process / ring-update call of this batch launched (1 for every
                                                     FFT chain that runs fused; the ring copies of an update are not kernels) */
/* Kernel configurations of the batch's size: the library carries, per transform size, the configuration that won the
 * build-time sweeps (variant 0) and the runners-up that came close (a different radix split / points per lane, rows per
 * workgroup, placement of the window and twiddle tables); for s16 frame / ring input with log_mode 0 or 1 -- other
 * inputs have variant 0 only.  glv_batch_set_variant forces one (-1 = automatic: wisdom, else 0). */
int glv_batch_variants(const glv_batch* b);
int glv_batch_set_variant(glv_batch* b, int variant);
int glv_batch_last_variant(const glv_batch* b);   /* configuration the last frame-kernel launch of this batch used */
int glv_batch_describe_variant(const glv_batch* b, int variant, char* buf, size_t len);

/* Launch wisdom -- the role of glfft's FFTWisdom (glfft/glfft_wisdom.cpp:235-446: candidate work-group shapes and radix
 * splits of a transform are timed on the target and the winner remembered per transform description) for this path.
 * glv_batch_autotune times EVERY kernel configuration built for the batch's size on several workgroup counts each, on the
 * batch's device with the caller's buffers (real updates: a stateful batch is reset afterwards), and records the winning
 * (variant, workgroups) process-wide; every later launch with the same description -- device name + compute-unit count, n,
 * input kind, kernel class of `ops`, log_mode, avg_frames of an averaging chain, log2 of the stream count -- uses it.
 * Save / load carry the table across processes (a text file, one entry per line; entries of other devices are kept but
 * never match); the file named by the environment variable GLV_WISDOM is loaded when the first batch is created. */
int glv_batch_autotune(glv_batch* b, const int16_t* d_pcm, float* d_out, unsigned ops, void* hip_stream, int* best_grid, float* best_ms);
/* Placement wisdom (ABI 6).  A stateful chain (GLV_OP_GRAVITY / GLV_OP_AVERAGE) runs at one of two or three speeds, 7 - 14 % apart, decided by where its
 * state arrays lie in physical memory relative to the caller's OUTPUT buffer (profiles/r06/modes.txt: K batches of one process keep K speeds; one batch
 * changes speed with the output buffer; no allocator choice, TLB or clock effect).  This call times the batch's current placement with the caller's real
 * buffers and then up to `candidates` - 1 fresh allocations of the state arrays (all alive together while it runs: up to `candidates` x the state's size,
 * bounded by the free device memory), keeps the fastest, frees the rest and RESETS the state -- call it once, before the first real update, with the
 * buffers the updates will use.  *first_ms / *best_ms (may be NULL): ms per update of the placement the batch was created with / now has. */
int glv_batch_tune_placement(glv_batch* b, const int16_t* d_pcm, void* d_out, unsigned ops, int candidates, void* hip_stream, float* first_ms, float* best_ms);
int glv_wisdom_save(const char* path);
int glv_wisdom_load(const char* path);
int glv_wisdom_clear(void);
int glv_wisdom_count(void);

/* Self test of the s16 window product.  The reference multiplies a float sample by a double window value and rounds the
 * double product to float (render.c:794).  For s16 input the kernels compute the same bits without fp64: the window value
 * is a float pair (hi, lo) with fma(x, hi, x * lo) == (float) ((double) x * w) for every one of the 65536 sample values x =
 * k / 65535 -- found and checked on the device when the batch is created.  This call re-checks EVERY (sample value, window
 * position) pair of the batch's size against the fp64 product on the device: *mismatches (must be 0) and the number of
 * positions whose low part had to be moved off (float) (w - hi) (*shifted, may be NULL). */
int glv_batch_window_selftest(glv_batch* b, unsigned long long* mismatches, int* shifted);

/* Name of the kernel the last process call launched (for matching rocprofv3 rows). */
const char* glv_batch_kernel_name(const glv_batch* b);

/* Which arithmetic GLV_OP_BARS over TEXEL rows (gl_storage != 0) runs on this batch with its current parameters (ABI 6):
 *   GLV_BARS_NONE          the batch has no bar tables (bars == 0 / GLV_OP_BARS not announced)
 *   GLV_BARS_F32_CHAIN     fewer than 256 bars: smooth_audio()'s float chain per bar (the modules' bars; fused into the transform where it fits)
 *   GLV_BARS_I8_EXACT      256 bars or more, the documented form: exact integer weighted means on the i8 matrix cores (oracle glvo_bars_int_at)
 *   GLV_BARS_F32_MATRIX    256 bars or more where the integer tables could NOT be made -- a bar wider than the largest LDS ring (1600 bins: very
 *                          large smooth_factor, or n = 32768 with smooth_factor slightly above 0.025), 2^P scaling beyond 31 bits, a weight
 *                          >= 2^23 / a tile without steps, or GLV_NO_BARS_I8 in the environment at creation: one f32 fma chain per bar on
 *                          the f32 matrix cores (bit-equal to smooth_audio()'s float order, NOT to glvo_bars_int_at)
 *   GLV_BARS_F32_SEQ       (ABI 7) sample_mode maximum / hybrid: one lane per bar and row, the shader's loop in float as glv_params.sample_mode documents
 *                          (oracle glvo_bars_mode_at), float and texel rows alike
 * Float rows (gl_storage 0) always take the float forms.  Callers and parity tests that depend on the exact form check this. */
enum { GLV_BARS_NONE = 0, GLV_BARS_F32_CHAIN = 1, GLV_BARS_F32_MATRIX = 2, GLV_BARS_I8_EXACT = 3, GLV_BARS_F32_SEQ = 4 };
int glv_batch_bars_arithmetic(const glv_batch* b);

/* ------------------------------------------------------------------------------------
 * 3. Several GPUs of one node (BASELINE configs[3]).  No reference counterpart.
 *    Streams are independent: device g of G owns the contiguous shard [g*B/G, (g+1)*B/G) --
 *    its PCM, state and spectra live only there -- and there is no data-path collective.
 *    glv_multi_run_s16 drives every shard from its own host thread on its own HIP stream
 *    (the same timed-region contract as bench.py: warm-up, barrier + synchronize, `steps`
 *    updates, barrier + synchronize) and then performs the ONLY communication of the path:
 *    one ncclAllGather of a 32-byte stats record per rank and one ncclAllReduce(max) of the
 *    elapsed seconds, over RCCL (xGMI inside a node).  librccl is resolved with dlopen by
 *    glv_multi_create; hosts that never call glv_multi_* do not need it, and a host without it still runs
 *    (the 32-byte records are then collected on the host: glv_multi_uses_rccl).
 * ------------------------------------------------------------------------------------ */
typedef struct glv_multi glv_multi;
typedef struct glv_multi_stats {   /* what every rank contributes to the all-gather (32 bytes) */
    uint64_t frames;               /* stereo frames the rank processed in the timed region */
    double   seconds;              /* wall clock of the rank's timed region */
    uint64_t bytes;                /* algorithmic HBM bytes of those frames (glv_batch_algorithmic_bytes) */
    double   kernel_ms;            /* HIP-event time of the rank's launches */
} glv_multi_stats;

/* contiguous balanced partition: the first (total % world) ranks take one extra stream */
void glv_multi_shard_range(uint64_t total_streams, int rank, int world, uint64_t* first, uint64_t* count);
/* devices: `ndev` distinct device ordinals (NULL = 0 .. ndev-1) */
int glv_multi_create(const glv_params* p, uint64_t total_streams, unsigned ops_mask, const int* devices, int ndev, glv_multi** out);
int glv_multi_destroy(glv_multi* m);
int glv_multi_devices(const glv_multi* m);
/* 1: the stats record travels over RCCL; 0: librccl is not installed (or GLV_MULTI_RCCL=0) and the table is assembled on the
 * host -- the data path has no collective either way.  Only in the second mode may `devices` name a device more than once. */
int glv_multi_uses_rccl(const glv_multi* m);
/* shard idx: its device, first global stream, stream count and the glv_batch that owns its state (any may be NULL) */
int glv_multi_shard(const glv_multi* m, int idx, int* device, uint64_t* first_stream, uint32_t* streams, glv_batch** batch);
/* d_pcm[idx] / d_out[idx]: that shard's buffers on its own device (int16 [streams][n][2] / float [streams][2][n], or the
 * layout `ops` implies).  stats: [ndev] records as gathered (identical on every rank), max_seconds: the all-reduced
 * maximum of the ranks' elapsed seconds; either may be NULL. */
int glv_multi_run_s16(glv_multi* m, const int16_t* const* d_pcm, float* const* d_out, unsigned ops, int warmup, int steps,
                      glv_multi_stats* stats, double* max_seconds);

#ifdef __cplusplus
}
#endif
#endif /* GLV_SPECTRUM_H */
