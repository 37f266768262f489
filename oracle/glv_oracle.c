/*
 * oracle/glv_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, one thread) of the GLava audio-spectrum hot path.
 * It exists so the HIP path can be checked where the reference itself cannot be
 * linked (the GPU box has no /root/reference) and so intermediate values the
 * reference never exposes (the FFT before abs/log) can be compared bit-for-bit.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (glava_amd/csrc) never includes, links or calls anything here.
 *
 * Pinning: the reference's tests hold no golden vector for this path (SURVEY.md 4 /
 * 8c), so this file is pinned against the *compiled reference itself*
 * (oracle/_ref/libglvref.so, built from /root/reference by oracle/Makefile):
 * tests/test_oracle.py demands bit equality on every function below that the
 * reference can execute without a GL context (the rd_update prelude is pinned through the reference's own
 * rd_update run over a null GL, tests/test_handle_audio.py; the GLSL twins at the end of this file are
 * restatements checked against an independent evaluation of the shader text, tests/test_glsl_twins.py), and the
 * vectors in tests/golden/ were produced by the compiled reference
 * (tests/golden/make_golden.py).
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).  The structure is deliberately different from the reference
 * (Stockham autosort instead of bit-reversal + in-place Danielson-Lanczos, explicit
 * twiddle tables, a ring instead of memmove) -- the *arithmetic per output bit* is
 * what is being restated.  Build with -O2 -ffp-contract=off, no -march=native, no
 * -ffast-math (meson.build:5 builds the reference at optimization=2 on baseline x86-64,
 * which has no FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GLVO_TWOPI 6.28318530718 /* glava/render.c:63 -- a truncated 2*pi, on purpose */

/* ---- a2: s16 interleaved -> planar f32 (glava/fifo.c:94-110) -------------------------
 * stereo: l = L/65535f, r = R/65535f.  mono (channels == 1): ((L + R) / 2) with C int
 * arithmetic (truncation toward zero), then /65535f, written to both channels. */
void glvo_unpack_s16(const int16_t* pcm, size_t frames, int channels, float* l, float* r) {
    for (size_t n = 0; n < frames; ++n) {
        int a = pcm[2 * n], b = pcm[2 * n + 1];
        if (channels == 1) {
            float s = (float) ((a + b) / 2) / (float) 65535;
            l[n] = s; r[n] = s;
        } else {
            l[n] = (float) a / (float) 65535;
            r[n] = (float) b / (float) 65535;
        }
    }
}

/* a3: f32 interleaved -> planar (glava/pulse_input.c:155-178); mono = (L+R)/2 in float */
void glvo_unpack_f32(const float* pcm, size_t frames, int channels, float* l, float* r) {
    for (size_t n = 0; n < frames; ++n) {
        if (channels == 1) {
            float s = (pcm[2 * n] + pcm[2 * n + 1]) / 2;
            l[n] = s; r[n] = s;
        } else { l[n] = pcm[2 * n]; r[n] = pcm[2 * n + 1]; }
    }
}

/* a2: ring update (glava/fifo.c:91-92 shift, :94-110 append, :67-79 zero fill).
 * `pcm == NULL` restates the poll-timeout branch. ring_l/ring_r hold fsz floats. */
void glvo_ring_update_s16(float* ring_l, float* ring_r, size_t fsz, const int16_t* pcm,
                          size_t new_frames, int channels) {
    memmove(ring_l, ring_l + new_frames, (fsz - new_frames) * sizeof(float));
    memmove(ring_r, ring_r + new_frames, (fsz - new_frames) * sizeof(float));
    if (pcm) glvo_unpack_s16(pcm, new_frames, channels, ring_l + fsz - new_frames, ring_r + fsz - new_frames);
    else {
        memset(ring_l + fsz - new_frames, 0, new_frames * sizeof(float));
        memset(ring_r + fsz - new_frames, 0, new_frames * sizeof(float));
    }
}

/* ---- a7(i): the window (glava/render.c:660 macro, call site :794) ---------------------
 * window(i, s->sz - 1) expands, because the macro does not parenthesise `sz`, to
 *   0.53836 - 0.46164*cos(TWOPI*(double)i/(double)N - 1)
 * i.e. Hamming coefficients, period N, shifted by one radian. */
double glvo_window(size_t i, size_t n) {
    return 0.53836 - (0.46164 * cos(GLVO_TWOPI * (double) i / (double) n - 1));
}
void glvo_window_table(double* w, size_t n) {
    for (size_t i = 0; i < n; ++i) w[i] = glvo_window(i, n);
}
void glvo_apply_window(float* data, size_t n) {
    for (size_t i = 0; i < n; ++i) data[i] = (float) ((double) data[i] * glvo_window(i, n));
}

/* ---- a7(iii): per-stage twiddles by the reference's float recurrence -------------------
 * (glava/render.c:817-836).  Stage with complex half-size L (reference mmax = 2L).
 * tw: L complex floats (re, im interleaved). */
void glvo_twiddles(float* tw, size_t L) {
    size_t mmax = 2 * L;
    float theta = -(2 * M_PI / mmax);
    float wtemp = sin(0.5 * theta);
    float wpr   = -2.0 * wtemp * wtemp;
    float wpi   = sin(theta);
    float wr = 1.0, wi = 0.0;
    for (size_t k = 0; k < L; ++k) {
        tw[2 * k] = wr; tw[2 * k + 1] = wi;
        float t = wr;
        float a = wr * wpr, b = wi * wpi; wr = wr + (a - b);
        float c = wi * wpr, d = t * wpi;  wi = wi + (c + d);
    }
}

/* a7(ii)+(iii): Stockham radix-2 autosort restatement of bit-reversal + in-place DIT
 * (glava/render.c:797-840).  Natural order in, natural order out; each butterfly does
 * exactly the reference's six roundings (render.c:826-832).  In place on data[0..2nn). */
void glvo_fft_core(float* data, size_t nn) {
    float* x = malloc(sizeof(float) * 2 * nn);
    float* y = malloc(sizeof(float) * 2 * nn);
    float* tw = malloc(sizeof(float) * 2 * (nn > 1 ? nn / 2 : 1));
    memcpy(x, data, sizeof(float) * 2 * nn);
    for (size_t L = 1; L < nn; L <<= 1) {
        glvo_twiddles(tw, L);
        size_t m = nn / (2 * L);               /* x viewed as [2][m][L], y as [m][2][L] */
        for (size_t j = 0; j < m; ++j)
            for (size_t k = 0; k < L; ++k) {
                float wr = tw[2 * k], wi = tw[2 * k + 1];
                const float* a = x + 2 * (j * L + k);
                const float* b = x + 2 * ((m + j) * L + k);
                float p0 = wr * b[0], p1 = wi * b[1]; float tr = p0 - p1;
                float p2 = wr * b[1], p3 = wi * b[0]; float ti = p2 + p3;
                float* lo = y + 2 * ((2 * j) * L + k);
                float* hi = y + 2 * ((2 * j + 1) * L + k);
                hi[0] = a[0] - tr; hi[1] = a[1] - ti;
                lo[0] = a[0] + tr; lo[1] = a[1] + ti;
            }
        float* t = x; x = y; y = t;
    }
    memcpy(data, x, sizeof(float) * 2 * nn);
    free(x); free(y); free(tw);
}

/* a7(iv): per-float abs, log, tilt (glava/render.c:842-846). */
float glvo_tilt(size_t n, size_t sz, float fft_scale, float fft_cutoff) {
    float t = (((float) n / (float) sz) * fft_scale) + (1.0F - fft_cutoff);
    return t > 1.0F ? t : 1.0F;
}
void glvo_magnitude(float* data, size_t sz, float fft_scale, float fft_cutoff) {
    for (size_t n = 0; n < sz; ++n) {
        float x = data[n];
        if (x < 0.0F) x = -x;
        x = (float) (log((double) (x + 1)) / 3);
        x *= glvo_tilt(n, sz, fft_scale, fft_cutoff);
        data[n] = x;
    }
}

/* a7: the whole transform_fft (glava/render.c:783-847).  If raw_out != NULL the FFT
 * output before abs/log/tilt is copied there (not observable in the reference). */
void glvo_transform_fft(float* data, size_t sz, float fft_scale, float fft_cutoff, float* raw_out) {
    glvo_apply_window(data, sz);
    glvo_fft_core(data, sz / 2);
    if (raw_out) memcpy(raw_out, data, sz * sizeof(float));
    glvo_magnitude(data, sz, fft_scale, fft_cutoff);
}

/* ---- a8: gravity (glava/render.c:720-736).  state: sz floats, zero-initialised. */
void glvo_gravity(float* b, float* applied, size_t sz, float gravity_step, float ur) {
    float g = gravity_step * (1.0F / ur);
    for (size_t t = 0; t < sz; ++t) {
        if (b[t] >= applied[t]) applied[t] = b[t] - g;
        else applied[t] -= g;
        b[t] = applied[t];
    }
}

/* ---- a9: F-frame (optionally windowed) average (glava/render.c:738-771).
 * The reference shifts a [F][sz] history with memmove; here it is a ring:
 * `hist` is [F][sz], `*head` is the slot that receives the current frame, and age
 * order oldest..newest is head+1, head+2, ..., head (mod F).  window_frame(f, F-1)
 * expands (macro at render.c:661, call at :766) to 0.6 - 0.4*cos(TWOPI*f/F - 1). */
double glvo_frame_weight(size_t f, size_t F, int use_window) {
    if (!use_window) return 1.0;
    return 0.6 - (0.4 * cos(GLVO_TWOPI * (double) f / (double) F - 1));
}
void glvo_average(float* b, float* hist, size_t* head, size_t sz, size_t F, int use_window) {
    memcpy(hist + (*head) * sz, b, sz * sizeof(float));
    for (size_t t = 0; t < sz; ++t) {
        float v = 0.0F;
        for (size_t f = 0; f < F; ++f) {
            size_t slot = (*head + 1 + f) % F;
            if (use_window) v = (float) ((double) v + glvo_frame_weight(f, F, 1) * (double) hist[slot * sz + t]);
            else            v = v + (float) 1 * hist[slot * sz + t];
        }
        b[t] = v / (float) F;
    }
    *head = (*head + 1) % F;
}

/* ---- a10: wrange (glava/render.c:773-781) */
void glvo_wrange(float* b, size_t sz) {
    for (size_t t = 0; t < sz; ++t) { float x = b[t] + 1.0F; b[t] = x / 2.0F; }
}

/* ---- whole stereo frame from PCM, the unit the batched GPU path processes --------------
 * pcm: [n][2] s16; out: [2][n]; grav: [2][n] state or NULL; hist: [2][F][n] or NULL.
 * Order of operators as handle_audio applies them (glava/render.c:2140-2156):
 * fft -> gravity -> average, per channel. */
void glvo_frame_s16(const int16_t* pcm, size_t n, int channels, float fft_scale, float fft_cutoff,
                    float* out, float* raw_out,
                    float* grav, float gravity_step, float ur,
                    float* hist, size_t* heads, size_t F, int use_window) {
    glvo_unpack_s16(pcm, n, channels, out, out + n);
    for (int c = 0; c < 2; ++c) {
        glvo_transform_fft(out + c * n, n, fft_scale, fft_cutoff, raw_out ? raw_out + c * n : NULL);
        if (grav) glvo_gravity(out + c * n, grav + c * n, n, gravity_step, ur);
        if (hist) glvo_average(out + c * n, hist + c * F * n, heads + c, n, F, use_window);
    }
}

/* CPU-baseline helper for bench.py ("port" kind): process `frames` stereo frames of the
 * same PCM layout the GPU sees; returns a checksum so the work cannot be elided. */
double glvo_bench_frames(const int16_t* pcm, size_t frames, size_t n, float fft_scale, float fft_cutoff) {
    float* out = malloc(sizeof(float) * 2 * n);
    double acc = 0;
    for (size_t f = 0; f < frames; ++f) {
        glvo_frame_s16(pcm + f * 2 * n, n, 2, fft_scale, fft_cutoff, out, NULL, NULL, 0, 1, NULL, NULL, 0, 0);
        acc += out[1] + out[n + 1];
    }
    free(out);
    return acc;
}

/* glvo_bench_frames on `threads` native pthreads for `seconds` of wall clock (bench.py cpu_baseline, kind "port":
 * used only where oracle/_ref is absent).  frames_done[t] = frames thread t finished; returns elapsed seconds. */
#include <pthread.h>
#include <time.h>
typedef struct { const int16_t* pcm; size_t frames, n; float scale, cutoff; volatile double deadline; double t_end; unsigned long long done; double sink; } glvo_mt_arg;
static double glvo_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void* glvo_mt_worker(void* v) {
    glvo_mt_arg* a = v;
    do { a->sink += glvo_bench_frames(a->pcm, a->frames, a->n, a->scale, a->cutoff); a->done += a->frames; } while (glvo_now() < a->deadline);
    a->t_end = glvo_now();
    return NULL;
}
double glvo_bench_mt(const int16_t* pcm, size_t frames, size_t n, float fft_scale, float fft_cutoff, int threads, double seconds,
                     unsigned long long* frames_done) {
    if (threads < 1 || threads > 4096) return -1.0;
    pthread_t* th = calloc(threads, sizeof(*th));
    glvo_mt_arg* arg = calloc(threads, sizeof(*arg));
    if (!th || !arg) { free(th); free(arg); return -3.0; }
    const double t0 = glvo_now();
    for (int t = 0; t < threads; ++t) {
        arg[t] = (glvo_mt_arg){ .pcm = pcm, .frames = frames, .n = n, .scale = fft_scale, .cutoff = fft_cutoff, .deadline = t0 + seconds };
        if (pthread_create(&th[t], NULL, glvo_mt_worker, &arg[t]) != 0) {
            /* the threads already running read arg[]: end their run (a deadline in the past) and join them before freeing */
            for (int u = 0; u < t; ++u) arg[u].deadline = 0.0;
            for (int u = 0; u < t; ++u) pthread_join(th[u], NULL);
            free(th); free(arg);
            return -2.0;
        }
    }
    double t_end = t0;
    for (int t = 0; t < threads; ++t) {
        pthread_join(th[t], NULL);
        frames_done[t] = arg[t].done;
        if (arg[t].t_end > t_end) t_end = arg[t].t_end;
    }
    free(th); free(arg);
    return t_end - t0;
}

/* ---- a11: the texture upload at the end of handle_audio (glava/render.c:521-524):
 *   glTexImage1D(GL_TEXTURE_1D, 0, GL_R16, sz, 0, GL_RED, GL_FLOAT, buf)
 * stores every float as a 16-bit unsigned normalized texel.  OpenGL 4.6 core, section 2.3.5.1 (eq. 2.3): clamp
 * to [0, 1], multiply by 2^16 - 1, convert to an integer "where rounding to nearest is preferred".  The
 * restatement rounds the EXACT product once, ties to even (x * 65535 fits a double exactly); NaN clamps to 0.
 * Which neighbour a driver picks at an exact tie / after a float-rounded product is implementation defined
 * (Mesa multiplies in float first), so this is pinned to the specification, not to a particular driver. */
uint16_t glvo_unorm16(float x) {
    float c = x > 0.0F ? x : 0.0F;          /* NaN compares false: 0 */
    if (c > 1.0F) c = 1.0F;
    return (uint16_t) rint((double) c * 65535.0);
}
void glvo_texels_r16(const float* buf, size_t sz, uint16_t* texels) {
    for (size_t i = 0; i < sz; ++i) texels[i] = glvo_unorm16(buf[i]);
}
/* the value a shader reads back from such a texel (GL 4.6 eq. 2.1): c / 65535 in float */
float glvo_unorm16_to_float(uint16_t c) { return (float) c / 65535.0F; }

/* ---- a5: rd_update prelude (glava/render.c:1765-1809).  These run inside rd_update; pinned by running the
 * reference's real rd_update over a GL that does nothing (integration/nullgl_harness.c -> oracle/_ref/
 * libglvnullgl_ref.so) and comparing what it hands to glTexImage1D with these restatements, bit for bit
 * (tests/test_handle_audio.py::test_prelude_*). ------------------------------------------------------ */
/* bufscale box decimation (render.c:1768-1781): mean of `k` consecutive samples, float accumulate
 * in index order, then one float division. */
void glvo_bufscale(const float* in, float* out, size_t n_out, size_t k) {
    for (size_t t = 0; t < n_out; ++t) {
        float accum = 0.0F;
        for (size_t a = 0; a < k; ++a) accum += in[t * k + a];
        accum /= (float) k;
        out[t] = accum;
    }
}
/* keyframe interpolation (render.c:1794-1809): s + (e - s) * min(uratio * kcounter, 1) */
void glvo_lerp(const float* start, const float* end, float* out, size_t n, float uratio, int kcounter) {
    float mod = uratio * kcounter;
    if (mod > 1.0F) mod = 1.0F;
    for (size_t t = 0; t < n; ++t) out[t] = start[t] + ((end[t] - start[t]) * mod);
}

/* ---- CPU transform_smooth (glava/render.c:694-718): in place, sequentially dependent (later
 * outputs read earlier, already replaced inputs).  Pinned against the compiled reference
 * (glvref_smooth).  The window bounds depend only on t: smin/smax tables (what the device consumes,
 * generated on the host with the same libm) are exposed separately. */
#define GLVO_E 2.7182818284590452353 /* render.c:692 */
void glvo_smooth_bounds(int* smin, int* smax, size_t sz, float smooth_distance, float smooth_ratio, size_t* asz_out) {
    size_t asz = (size_t) ceil(sz / smooth_ratio);
    for (size_t t = 0; t < asz; ++t) {
        float db = log((int) t);
        float lo = db - smooth_distance; if (!(lo > 0)) lo = 0;      /* max(db - d, 0) with the macro's `_a > _b ? _a : _b` */
        smin[t] = (int) floor(powf(GLVO_E, lo));
        int hi = (int) ceil(powf(GLVO_E, db + smooth_distance));
        smax[t] = hi < (int) sz - 1 ? hi : (int) sz - 1;
    }
    *asz_out = asz;
}
void glvo_smooth(float* b, size_t sz, float smooth_distance, float smooth_ratio) {
    size_t asz = (size_t) ceil(sz / smooth_ratio);
    for (int t = 0; t < (int) asz; ++t) {
        float db = log(t), avg = 0;
        float lo = db - smooth_distance; if (!(lo > 0)) lo = 0;
        int smin = (int) floor(powf(GLVO_E, lo));
        int hi = (int) ceil(powf(GLVO_E, db + smooth_distance));
        int smax = hi < (int) sz - 1 ? hi : (int) sz - 1;
        int count = 0;
        for (int s = smin; s <= smax; ++s)
            if (b[s]) { avg += b[s]; count++; }
        avg /= count;
        b[t] = avg;
    }
}

/* ---- a12: the GL "accel" twins (semantics only; GLSL float, cannot be executed here: no GL
 * context, so this part of the oracle is UNPINNED and tolerances in the tests are loose). ---------- */
/* average_pass.frag:19-45 with window() of common.glsl:13 as expanded at average_pass.frag:41:
 *   r = sum_I (0.53836 - 0.46164*cos(TWOPI*I/F - 1)) * t_I / F,  t_0 = most recent (render.c:2247-2256);
 * no window when F == 2 (average_pass.frag:27-29). Weight for age f (0 = oldest): I = F-1-f. */
double glvo_gl_frame_weight(size_t f, size_t F, int use_window) {
    if (!use_window || F == 2) return 1.0;
    size_t I = F - 1 - f;
    return 0.53836 - (0.46164 * cos(GLVO_TWOPI * (double) I / (double) F - 1));
}
void glvo_average_gl(float* b, float* hist, size_t* head, size_t sz, size_t F, int use_window) {
    memcpy(hist + (*head) * sz, b, sz * sizeof(float));
    for (size_t t = 0; t < sz; ++t) {
        float v = 0.0F;
        for (size_t f = 0; f < F; ++f) {
            size_t slot = (*head + 1 + f) % F;
            v = (float) ((double) v + glvo_gl_frame_weight(f, F, use_window) * (double) hist[slot * sz + t]);
        }
        b[t] = v / (float) F;
    }
    *head = (*head + 1) % F;
}
/* The GL passes with their storage (render.c:2188-2265 when setaccelfft moved gravity / average to the GPU): every
 * intermediate is a GL_R16 texture (render.c:523 upload, :1718 FBO textures), i.e. clamped to [0, 1] and quantised to 16 bits
 * wherever a pass writes:  tex = Q(row) (upload);  store = Q(max(store, tex) - g)  (GL_MAX blend :2199-2210, exact on texel
 * values, then the in-place gravity pass :2219-2228, gravity_pass.frag);  ring[head] = store (pass.frag copy :2232-2243);
 * row = Q(sum_I window(I) * t_I / F) (average_pass.frag, t_0 = newest) -- only when F > 1 (:2230), else row = store.
 * store is the previous newest ring slot (the same texels).  Q() as glvo_unorm16 / glvo_unorm16_to_float above.
 * The average is evaluated in the shader's own arithmetic: GLSL `float` is 32 bits -- `float r = 0; r += window(I) * tI; ...
 * r / _AVG_FRAMES` -- the weight a compile-time constant folded to float, every product, sum and the quotient rounded to float,
 * nothing contracted (what Mesa's llvmpipe executes on x86-64).  (Rounds 2-3 restated this pass with the CPU operator's
 * float * double products, render.c:759; against the reference's llvmpipe texels the two forms are indistinguishable --
 * profiles/r04/gl_average_models.txt -- and which neighbour an exact half-texel tie takes is the driver's choice either way:
 * tests/test_gl_reference.py compares tie-aware.)
 * Restatement (no GL here); the shader arithmetic itself is checked against tests/glsl_eval.py. */
static float glvo_q16(float x) { return glvo_unorm16_to_float(glvo_unorm16(x)); }
void glvo_gl_chain_r16(float* row, float* store, float* hist, size_t* head, size_t sz, size_t F, int use_window, int do_average,
                       float gravity_step, float ur) {
    const float g = gravity_step * (1.0F / ur);
    for (size_t t = 0; t < sz; ++t) {
        float tex = glvo_q16(row[t]);
        float st = tex >= store[t] ? tex : store[t];
        st = glvo_q16(st - g);
        store[t] = st;
        row[t] = st;
    }
    if (!do_average) return;
    memcpy(hist + (*head) * sz, row, sz * sizeof(float));
    if (F > 1)
        for (size_t t = 0; t < sz; ++t) {
            float v = 0.0F;
            for (size_t f = 0; f < F; ++f) {
                size_t slot = (*head + 1 + f) % F;
                float w = (float) glvo_gl_frame_weight(f, F, use_window);
                float p = (use_window && F != 2) ? w * hist[slot * sz + t] : hist[slot * sz + t];
                v = v + p;
            }
            row[t] = glvo_q16(v / (float) F);
        }
    *head = (*head + 1) % F;
}

/* smooth_audio() (shaders/glava/util/smooth.glsl:13-40, SAMPLE_MODE average) sampled at the bar positions the radial
 * module uses (radial/1.frag:58-70: pos = k / bars, k = 0..bars-1).  tex[] is clamped to [0,1] as the
 * GL_R16 texture would (render.c:523).
 * The SHAPE -- ROUND_FORMULA, SAMPLE_SCALE, SAMPLE_RANGE, the `#define`s of smooth_parameters.glsl:17-42 that a user's config overrides
 * (glsl_ext.c:143-157 re-defines them textually) -- is a setting of this checker: glvo_set_smooth_shape() selects it for every glvo_bars_*
 * function until changed (formula 0 sinusoidal / 1 circular / 2 linear: the macros of util/common.glsl:17-22; a scale or range of 0 selects the
 * shipped 8 / 0.9).  Test infrastructure: one thread, set - use - restore. */
static int glvo_shape_formula = 0;
static float glvo_shape_scale = 8.0F, glvo_shape_range = 0.9F;
void glvo_set_smooth_shape(int formula, float scale, float range) {
    glvo_shape_formula = formula;
    glvo_shape_scale = scale != 0 ? scale : 8.0F;
    glvo_shape_range = range != 0 ? range : 0.9F;
}
static float glvo_scale_audio(float idx) { return -logf((-glvo_shape_range * idx) + 1) / glvo_shape_scale; }   /* smooth.glsl:13-15 */
static float glvo_clamp01(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
static float glvo_sinusoidal(float x) {                                              /* ROUND_FORMULA(x), common.glsl:17-22 */
    if (glvo_shape_formula == 1) return sqrtf(1 - ((x - 1) * (x - 1)));              /* circular */
    if (glvo_shape_formula == 2) return x;                                           /* linear */
    return (0.5F * sinf((3.14159265359F * x) - (3.14159265359F / 2))) + 0.5F;        /* sinusoidal */
}
static double glvo_round_formula_exact(double x) {                                   /* the same in float64 (the *_exact functions) */
    if (glvo_shape_formula == 1) return sqrt(1 - (x - 1) * (x - 1));
    if (glvo_shape_formula == 2) return x;
    return 0.5 * sin(3.14159265358979323846 * x - 3.14159265358979323846 / 2) + 0.5;
}
/* SAMPLE_MODE for the float64 evaluations below (glvo_bars_at_exact, glvo_bars_one_exact, glvo_bars_range_exact -- the brackets the texels of a
 * GLSL implementation are held to): 0 average (s <= smax), 1 maximum, 2 hybrid with weight H (s < smax) -- smooth.glsl:32-59.  The float functions
 * take the mode as an argument (glvo_bars_mode_at) or are the averaging forms by definition. */
static int glvo_shape_mode = 0;
static double glvo_shape_hybrid = 0.65;
void glvo_set_smooth_mode(int mode, float hybrid_weight) { glvo_shape_mode = mode; glvo_shape_hybrid = hybrid_weight != 0 ? (double) hybrid_weight : 0.65; }
/* smooth_audio()'s loop from the float bounds smin / smax -- the shader's own walk in float (s += 1.0F, tap round(s)), weights, products, sums and the
 * result in float64.  half_even: round() at an exact .5 to even (Mesa) instead of away from zero (C). */
static double glvo_smooth_loop_exact(const float* tex, size_t sz, float smin, float smax, int half_even, int* cnt_out) {
    float m = (smax - smin) / 2.0F, rm = smin + m;
    double avg = 0, weight = 0, vmax = 0;
    int cnt = 0;
    for (float s = smin; glvo_shape_mode == 0 ? s <= smax : s < smax; s += 1.0F) {
        double x = ((double) m - fabs((double) rm - (double) s)) / (double) m;
        x = x < 0 ? 0 : (x > 1 ? 1 : x);
        double w = glvo_round_formula_exact(x);
        long b = (long) (int) (half_even ? rintf(s) : roundf(s));
        double tv = tex[b < (long) sz ? b : (long) sz - 1];
        tv = tv > 0 ? (tv < 1 ? tv : 1) : 0;
        double v = tv * w;
        avg += v; weight += w; ++cnt;
        if (vmax < v) vmax = v;
    }
    *cnt_out = cnt;
    if (glvo_shape_mode == 1) return vmax;
    double mean = weight > 0 ? avg / weight : 0.0;
    return glvo_shape_mode == 2 ? vmax * (1.0 - glvo_shape_hybrid) + mean * glvo_shape_hybrid : mean;
}
/* phase: smooth_audio() is evaluated at idx = (k + phase) / bars -- 0: the modules' bar positions (radial/1.frag:58-70);
 * 0.5 with bars == sz: gl_FragCoord.x / w of util/smooth_pass.frag, the reference's pre-smoothing pass (render.c:2277-2303) */
void glvo_bars_at(const float* tex, size_t sz, float* bars_out, size_t bars, float smooth_factor, float phase);
void glvo_bars(const float* tex, size_t sz, float* bars_out, size_t bars, float smooth_factor) { glvo_bars_at(tex, sz, bars_out, bars, smooth_factor, 0.0F); }
void glvo_bars_at(const float* tex, size_t sz, float* bars_out, size_t bars, float smooth_factor, float phase) {
    for (size_t k = 0; k < bars; ++k) {
        float idx = phase == 0.0F ? (float) k / (float) bars : ((float) k + phase) / (float) bars;
        float smin = glvo_scale_audio(glvo_clamp01(idx - smooth_factor)) * sz;
        float smax = glvo_scale_audio(glvo_clamp01(idx + smooth_factor)) * sz;
        float m = (smax - smin) / 2.0F, rm = smin + m;
        float avg = 0, weight = 0;
        for (float s = smin; s <= smax; s += 1.0F) {
            float w = glvo_sinusoidal(glvo_clamp01((m - fabsf(rm - s)) / m));
            weight += w;
            avg += glvo_clamp01(tex[(int) roundf(s)]) * w;
        }
        bars_out[k] = avg / weight;
    }
}

/* SAMPLE_MODE maximum (mode 1, smooth.glsl:52-58) and hybrid (mode 2, smooth.glsl:41-51) -- the shader's loops as written, in float, every operation
 * rounded on its own (this file is compiled with -ffp-contract=off): s = smin; s < smax; v = tex * w; vmax from 0 by `if (vmax < v) vmax = v`;
 * hybrid: avg += v, weight += w, result (vmax * (1 - H)) + ((avg / weight) * H).  The library's glv_bars_mode_kernel must produce these bits
 * (glv_params.sample_mode in include/glv_spectrum.h states the same contract).  tex[] clamped to [0, 1], NaN -> 0 (a GL_R16 texel). */
void glvo_bars_mode_at(const float* tex, size_t sz, float* bars_out, size_t bars, float smooth_factor, float phase, int mode, float hybrid_weight) {
    for (size_t k = 0; k < bars; ++k) {
        float idx = phase == 0.0F ? (float) k / (float) bars : ((float) k + phase) / (float) bars;
        float smin = glvo_scale_audio(glvo_clamp01(idx - smooth_factor)) * sz;
        float smax = glvo_scale_audio(glvo_clamp01(idx + smooth_factor)) * sz;
        float m = (smax - smin) / 2.0F, rm = smin + m;
        float vmax = 0, avg = 0, weight = 0;
        for (float s = smin; s < smax; s += 1.0F) {
            float w = glvo_sinusoidal(glvo_clamp01((m - fabsf(rm - s)) / m));
            float tv = tex[(int) roundf(s)];
            tv = tv > 0 ? (tv < 1 ? tv : 1) : 0;
            float v = tv * w;
            weight += w;
            avg += v;
            if (vmax < v) vmax = v;
        }
        float one_minus = 1 - hybrid_weight;
        bars_out[k] = mode == 1 ? vmax : (vmax * one_minus) + ((avg / weight) * hybrid_weight);
    }
}

/* GLV_OP_BARS as the library defines it: the taps and weights of glvo_bars (smooth.glsl:13-40), summed in the library's
 * documented order instead of tap by tap -- a bar's taps in chunks of C = 16 (sz <= 1024), 32 (sz = 2048) or 64 (sz >= 4096);
 * within a chunk C / 8 partial sums of eight consecutive taps each, every one the sum e + o of two fused-multiply-add chains
 * from +0 (e over the even taps: fma(x6, w6, fma(x4, w4, fma(x2, w2, fma(x0, w0, 0)))), o over the odd ones; zero weights past
 * the bar's end), combined pairwise (neighbours, pairs of pairs, the two quads); chunk totals added in chunk order; one
 * division by the tap-order sum of the weights.  glava_amd/csrc/glv_frame.h "GLV_OP_BARS arithmetic" is what this restates;
 * the GPU tests demand these bits, tests/test_glsl_twins.py bounds the distance to glvo_bars (summation rounding only).
 * With 256 bars or more (the pre-smoothing pass: bars == sz) the library's order is simpler: ONE fused-multiply-add chain over the
 * bar's taps in bin order from +0, acc = fmaf(w, x, acc) -- what a tile of v_mfma_f32_32x32x2_f32 computes for 32 bars x 64 rows at
 * a time (a k-ordered fmaf chain, bit for bit; taps of weight +0 -- the other bars' bins of the tile -- leave it untouched). */
void glvo_bars_chunked_at(const float* tex, size_t sz, float* bars_out, size_t bars, float smooth_factor, float phase);
void glvo_bars_chunked(const float* tex, size_t sz, float* bars_out, size_t bars, float smooth_factor) { glvo_bars_chunked_at(tex, sz, bars_out, bars, smooth_factor, 0.0F); }
void glvo_bars_chunked_at(const float* tex, size_t sz, float* bars_out, size_t bars, float smooth_factor, float phase) {
    float* x = malloc(sizeof(float) * (sz + 128));
    float* w = malloc(sizeof(float) * (sz + 128));
    const size_t C = sz <= 1024 ? 16 : (sz == 2048 ? 32 : 64);
    for (size_t k = 0; k < bars; ++k) {
        float idx = phase == 0.0F ? (float) k / (float) bars : ((float) k + phase) / (float) bars;
        float smin = glvo_scale_audio(glvo_clamp01(idx - smooth_factor)) * sz;
        float smax = glvo_scale_audio(glvo_clamp01(idx + smooth_factor)) * sz;
        float m = (smax - smin) / 2.0F, rm = smin + m;
        float weight = 0;
        size_t cnt = 0;
        long prev = -1;
        for (float s = smin; s <= smax; s += 1.0F) {
            float wt = glvo_sinusoidal(glvo_clamp01((m - fabsf(rm - s)) / m));
            weight += wt;
            /* the library counts a bar's taps by BIN: where s += 1.0F rounds up across a binade boundary (x.49997 + 1 -> (x+1).5) round(s)
             * skips a bin -- the shader has no tap there, the library a tap of weight +0 (glv_tables.h make_bar_taps), which moves the
             * taps behind it one place further in their chunk (n = 4096, bar 2848: bin 512; only the chunked order notices) */
            long bin = (long) (int) roundf(s);
            for (long q = prev + 1; prev >= 0 && q < bin; ++q) { w[cnt] = 0; x[cnt] = 0; ++cnt; }
            prev = bin;
            w[cnt] = wt;
            float tv = tex[(int) roundf(s)];
            x[cnt] = tv > 0 ? (tv < 1 ? tv : 1) : 0;          /* [0, 1] like a GL_R16 texel; NaN -> 0 (the library's clamp: v_pk_mul_f32 ... clamp) */
            ++cnt;
        }
        for (size_t p = cnt; p < ((cnt + C - 1) / C) * C; ++p) { w[p] = 0; x[p] = 0; }
        float total = 0;
        if (bars >= 256) {                                    /* one chain in bin order */
            for (size_t p = 0; p < cnt; ++p) total = fmaf(w[p], x[p], total);
            bars_out[k] = total / weight;
            continue;
        }
        for (size_t c0 = 0; c0 < cnt; c0 += C) {
            float lane[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            for (size_t l = 0; l < C / 8; ++l) {
                float e = 0, o = 0;
                for (int i = 0; i < 8; i += 2) {
                    e = fmaf(x[c0 + 8 * l + i], w[c0 + 8 * l + i], e);
                    o = fmaf(x[c0 + 8 * l + i + 1], w[c0 + 8 * l + i + 1], o);
                }
                lane[l] = e + o;
            }
            float sum = lane[0] + lane[1];
            if (C >= 32) sum = sum + (lane[2] + lane[3]);
            if (C >= 64) sum = sum + ((lane[4] + lane[5]) + (lane[6] + lane[7]));
            total = total + sum;
        }
        bars_out[k] = total / weight;
    }
    free(x); free(w);
}

/* GLV_OP_BARS over TEXEL rows with 256 bars or more -- the pre-smoothing pass inside the library's GL chains (gl_storage != 0:
 * render.c:2277-2303 samples a GL_R16 texture, so every input is a 16-bit integer c).  There the library computes the weighted mean
 * of smooth.glsl:25-40 in EXACT integer arithmetic (glava_amd/csrc/glv_tables.h "many bars over TEXEL rows" is what this restates):
 *   w_j   the shader's float weights, taps counted by bin as above (a skipped bin is a tap of weight 0)
 *   ws    = sum_j (double) w_j in tap order;  P = max(17, 21 + ceil(log2 ws))
 *   W_j   = llrint(ldexp((double) w_j, P) / ws); the first largest W_j takes the residue, so that sum_j W_j == 2^P
 *   texel = floor(sum_j W_j c_j / 2^P + 1/2);   float = (float) ((double) sum_j W_j c_j * 2^-P / 65535.0)
 * A bar whose float weights sum to 0 (0 / 0 in the shader) is texel 0 / float NaN.  Either output may be NULL.
 * Returns 0, or -1 when some bar would need P > 31 (the library then keeps its float chain). */
int glvo_bars_int_at(const uint16_t* tex, size_t sz, uint16_t* texels_out, float* floats_out, size_t bars, float smooth_factor, float phase) {
    float* w = malloc(sizeof(float) * (sz + 128));
    long* bin_of = malloc(sizeof(long) * (sz + 128));
    long long* W = malloc(sizeof(long long) * (sz + 128));
    int rc = 0;
    for (size_t k = 0; k < bars && rc == 0; ++k) {
        float idx = phase == 0.0F ? (float) k / (float) bars : ((float) k + phase) / (float) bars;
        float smin = glvo_scale_audio(glvo_clamp01(idx - smooth_factor)) * sz;
        float smax = glvo_scale_audio(glvo_clamp01(idx + smooth_factor)) * sz;
        float m = (smax - smin) / 2.0F, rm = smin + m;
        size_t cnt = 0;
        long prev = -1;
        for (float s = smin; s <= smax; s += 1.0F) {
            float wt = glvo_sinusoidal(glvo_clamp01((m - fabsf(rm - s)) / m));
            long bin = (long) (int) roundf(s);
            for (long q = prev + 1; prev >= 0 && q < bin; ++q) { w[cnt] = 0; bin_of[cnt] = q; ++cnt; }
            prev = bin;
            w[cnt] = wt; bin_of[cnt] = bin; ++cnt;
        }
        double ws = 0;
        for (size_t j = 0; j < cnt; ++j) ws += (double) w[j];
        if (!(ws > 0) || !(ws < 1e30)) {
            if (texels_out) texels_out[k] = 0;
            if (floats_out) floats_out[k] = NAN;
            continue;
        }
        int ex = 0;
        double f = frexp(ws, &ex);
        int P = 21 + (f == 0.5 ? ex - 1 : ex);
        if (P < 17) P = 17;
        if (P > 31) { rc = -1; break; }
        long long sum = 0;
        size_t jmax = 0;
        for (size_t j = 0; j < cnt; ++j) {
            W[j] = llrint(ldexp((double) w[j], P) / ws);
            sum += W[j];
            if (W[j] > W[jmax]) jmax = j;
        }
        W[jmax] += (1LL << P) - sum;
        long long total = 0;
        for (size_t j = 0; j < cnt; ++j) total += W[j] * (long long) tex[bin_of[j]];
        if (texels_out) texels_out[k] = (uint16_t) ((total + (1LL << (P - 1))) >> P);
        if (floats_out) floats_out[k] = (float) (ldexp((double) total, -P) / 65535.0);
    }
    free(w); free(bin_of); free(W);
    return rc;
}

/* smooth_audio() once more, for the tie-aware comparisons of tests/test_gl_reference.py: the same taps -- selected by the shader's
 * float bounds smin / smax exactly as in glvo_bars_at -- but weights, products, sums and the quotient in float64, so that
 * exact[k] is (to ~1e-15) the real number a float implementation approximates; ntaps[k] = taps of bar k (the float error of
 * an implementation's sum grows with it); fragile[k] = 1 when the tap SET itself hangs on the last bits of the math library:
 * the count floor(smax - smin) + 1 or the rounding of every tap position round(smin + j) would change if smin / smax moved by
 * `ulps` units in the last place (log() of two libraries may differ by that) -- such a bar gains or loses a whole tap between
 * implementations and is compared with no tolerance claim at all. */
void glvo_bars_at_exact(const float* tex, size_t sz, double* exact, int* ntaps, int* fragile, size_t bars, float smooth_factor, float phase, int ulps) {
    for (size_t k = 0; k < bars; ++k) {
        float idx = phase == 0.0F ? (float) k / (float) bars : ((float) k + phase) / (float) bars;
        float smin = glvo_scale_audio(glvo_clamp01(idx - smooth_factor)) * sz;
        float smax = glvo_scale_audio(glvo_clamp01(idx + smooth_factor)) * sz;
        int cnt = 0;
        exact[k] = glvo_smooth_loop_exact(tex, sz, smin, smax, 0, &cnt);
        ntaps[k] = cnt;
        /* scale_audio() = log, a product and a quotient in float: `ulps` units of the larger bound, four times over for the chain */
        double big = fabs((double) smax) > 1 ? fabs((double) smax) : 1;
        double eps = 4.0 * ulps * (nextafterf((float) big, INFINITY) - (float) big);
        double d = (double) smax - (double) smin, fm = (double) smin - floor((double) smin);
        fragile[k] = fabs(d - rint(d)) <= 2 * eps || fabs(fm - 0.5) <= eps;
    }
}

/* How much of a float implementation's error the WEIGHT FUNCTION amplifies, per bar (for the brackets of tests/test_gl_reference.py): the argument
 * x = (m - |rm - s|) / m is a difference of float positions, good to dx = 8 ulp(max(smax, 1)) / m, and ROUND_FORMULA circular = sqrt(1 - (x - 1)^2) has
 * an infinite slope at x = 0 -- the outermost taps of a bar get weights that hang on the last bits of x whoever evaluates them (sinusoidal is flat
 * there, linear has slope 1).  rel[k] = sum_j |w(x_j + dx) - w(x_j - dx)| / sum_j w_j bounds the relative move of a weighted mean, abs[k] = max_j of the
 * same differences the absolute move of a maximum of x_j w_j. */
void glvo_bars_weight_slack(size_t sz, double* rel, double* abs_out, size_t bars, float smooth_factor, float phase) {
    for (size_t k = 0; k < bars; ++k) {
        float idx = phase == 0.0F ? (float) k / (float) bars : ((float) k + phase) / (float) bars;
        float smin = glvo_scale_audio(glvo_clamp01(idx - smooth_factor)) * sz;
        float smax = glvo_scale_audio(glvo_clamp01(idx + smooth_factor)) * sz;
        float m = (smax - smin) / 2.0F, rm = smin + m;
        double big = fabs((double) smax) > 1 ? fabs((double) smax) : 1;
        double dx = m > 0 ? 8.0 * (nextafterf((float) big, INFINITY) - (float) big) / (double) m : 0.0;
        double acc = 0, wsum = 0, worst = 0;
        for (float s = smin; glvo_shape_mode == 0 ? s <= smax : s < smax; s += 1.0F) {
            double x = ((double) m - fabs((double) rm - (double) s)) / (double) m;
            double xc = x < 0 ? 0 : (x > 1 ? 1 : x), xp = x + dx, xm = x - dx;
            xp = xp < 0 ? 0 : (xp > 1 ? 1 : xp); xm = xm < 0 ? 0 : (xm > 1 ? 1 : xm);
            double d = fabs(glvo_round_formula_exact(xp) - glvo_round_formula_exact(xm));
            acc += d; wsum += glvo_round_formula_exact(xc);
            if (d > worst) worst = d;
        }
        rel[k] = wsum > 0 ? acc / wsum : 0.0;
        abs_out[k] = worst;
    }
}

/* One bar of smooth_audio() in float64 like glvo_bars_at_exact, with the loop's float bounds MOVED by dmin / dmax units in the last place
 * (nextafterf steps): the tap set another implementation of scale_audio() -- its log() a few ulps off -- would walk.  The loop itself
 * (s = smin; s <= smax; s += 1.0F, tap round(s)) is the shader's, in float: which taps exist and which bins they hit follows from the
 * moved bounds exactly as it would in that implementation.  half_even: GLSL leaves the direction round() takes at an exact .5 to the
 * implementation (GLSL 4.60 8.3); C's roundf goes away from zero, Mesa's llvmpipe to even -- and a tap position CAN be an exact .5
 * (n = 4096, bar 2848: 473.49997 + 39 rounds to 512.5 in float).  tests/test_gl_reference.py uses it to make the comparison TAP-SET-AWARE:
 * where glvo_bars_at_exact flags a bar as fragile, a texel is accepted iff it is right for ONE of the admissible tap sets. */
void glvo_bars_one_exact(const float* tex, size_t sz, size_t k, size_t bars, float smooth_factor, float phase, int dmin, int dmax, int half_even, double* exact, int* ntaps) {
    float idx = phase == 0.0F ? (float) k / (float) bars : ((float) k + phase) / (float) bars;
    float smin = glvo_scale_audio(glvo_clamp01(idx - smooth_factor)) * sz;
    float smax = glvo_scale_audio(glvo_clamp01(idx + smooth_factor)) * sz;
    for (int q = 0; q < abs(dmin); ++q) smin = nextafterf(smin, dmin > 0 ? INFINITY : -INFINITY);
    for (int q = 0; q < abs(dmax); ++q) smax = nextafterf(smax, dmax > 0 ? INFINITY : -INFINITY);
    if (smin < 0) smin = 0;
    *exact = glvo_smooth_loop_exact(tex, sz, smin, smax, half_even, ntaps);
}

/* What smooth_audio() may return for every bar when scale_audio()'s log() is only as accurate as GLSL implementations are held to:
 * the loop's bounds are n * (-log(1 - 0.9 u) / 8), and log() carries an ABSOLUTE error of up to log_abs inside [0.5, 2] (2^-21 is what
 * the SPIR-V / Vulkan precision table demands there; OpenGL promises no more, and Mesa's llvmpipe -- a polynomial log2 -- uses a good
 * part of it just below 1, where -log(1 - 0.9 u) is small: tests/golden `n1024_F5w_sf010`, bar 13).  So either bound may sit anywhere
 * within d = log_abs * sz / 8 (+ 4 ulps) of the value a correctly rounded log gives.  Over that box the mean is piecewise smooth and, for
 * so small a box, monotone along each axis within a piece; the pieces end where the tap SET changes: frac(smin) crossing .5 (every tap's
 * round() moves) and smax - smin crossing an integer (a tap appears).  The range is therefore taken over a 9 x 9 grid of the box, the
 * points on either side of every such crossing inside it, and both directions of round() at an exact .5.
 * vmin is evaluated on tex_lo, vmax on tex_hi (the weights are >= 0: the mean is monotone in every tap); pass the same row twice for one
 * row's range.  ntaps[k] = the largest tap count met (the float error of an implementation's sums grows with it). */
static double glvo_bars_mean_moved(const float* tex, size_t sz, float smin, float smax, int half_even, int* cnt_out) {
    return glvo_smooth_loop_exact(tex, sz, smin, smax, half_even, cnt_out);
}
void glvo_bars_range_exact(const float* tex_lo, const float* tex_hi, size_t sz, double* vmin, double* vmax, int* ntaps, size_t bars,
                           float smooth_factor, float phase, double log_abs) {
    for (size_t k = 0; k < bars; ++k) {
        float idx = phase == 0.0F ? (float) k / (float) bars : ((float) k + phase) / (float) bars;
        float smin0 = glvo_scale_audio(glvo_clamp01(idx - smooth_factor)) * sz;
        float smax0 = glvo_scale_audio(glvo_clamp01(idx + smooth_factor)) * sz;
        double big = fabs((double) smax0) > 1 ? fabs((double) smax0) : 1;
        double d = log_abs * (double) sz / (double) glvo_shape_scale + 4.0 * (nextafterf((float) big, INFINITY) - (float) big);
        float amin[9 + 3 * 3]; int na = 0;
        for (int g = -4; g <= 4; ++g) amin[na++] = (float) ((double) smin0 + d * g / 4.0);
        for (int q = -1; q <= 1; ++q) {                                  /* frac(smin) == .5 inside the box */
            double c = floor((double) smin0) + 0.5 + q;
            if (fabs(c - (double) smin0) <= d) { amin[na++] = nextafterf((float) c, -INFINITY); amin[na++] = (float) c; amin[na++] = nextafterf((float) c, INFINITY); }
        }
        double lo = INFINITY, hi = -INFINITY; int most = 0;
        for (int a = 0; a < na; ++a) {
            float smin = amin[a] < 0 ? 0 : amin[a];
            float bmax[9 + 5 * 5]; int nb = 0;
            for (int g = -4; g <= 4; ++g) bmax[nb++] = (float) ((double) smax0 + d * g / 4.0);
            double q0 = rint((double) smax0 - (double) smin);
            for (int q = -2; q <= 2; ++q) {                              /* smax - smin == an integer inside the box */
                float c = smin + (float) (q0 + q);
                if (fabs((double) c - (double) smax0) <= d && nb + 5 <= (int) (sizeof(bmax) / sizeof(bmax[0]))) {
                    float dn = nextafterf(c, -INFINITY), up = nextafterf(c, INFINITY);
                    bmax[nb++] = nextafterf(dn, -INFINITY); bmax[nb++] = dn; bmax[nb++] = c; bmax[nb++] = up; bmax[nb++] = nextafterf(up, INFINITY);
                }
            }
            for (int b = 0; b < nb; ++b)
                for (int he = 0; he < 2; ++he) {
                    int cnt;
                    double v = glvo_bars_mean_moved(tex_lo, sz, smin, bmax[b], he, &cnt);
                    if (v < lo) lo = v;
                    if (cnt > most) most = cnt;
                    if (tex_hi != tex_lo) v = glvo_bars_mean_moved(tex_hi, sz, smin, bmax[b], he, &cnt);
                    if (v > hi) hi = v;
                }
        }
        vmin[k] = lo; vmax[k] = hi; ntaps[k] = most;
    }
}

/* The library's s16 window product (glava_amd/csrc/glv_core.h apply_window_split): for every window position of size n the
 * float pair hi = (float) w, lo = (float) (w - hi), lo moved by +1, -1, +2, ... ulps until
 *     fmaf(x, hi, x * lo) == (float) ((double) x * w)        (render.c:794: float * double -> double -> float)
 * holds for every s16 sample value x = k / 65535 (fifo.c:105-106).  This is the CPU restatement of that search (the library
 * runs it on the device when a batch is created); returns the number of (k, position) pairs that still differ after it
 * (must be 0), *shifted = positions whose lo had to move, *max_shift = the largest move in ulps. */
long glvo_window_split_mismatches(size_t n, int* shifted, int* max_shift) {
    static float x[32769];
    for (int k = 0; k <= 32768; ++k) x[k] = (float) k / (float) 65535;
    long left = 0;
    int nsh = 0, maxd = 0;
    for (size_t i = 0; i < n; ++i) {
        double w = 0.53836 - (0.46164 * cos(GLVO_TWOPI * (double) i / (double) n - 1));      /* window(i, n) as render.c:660 expands at :794 */
        float hi = (float) w, lo0 = (float) (w - (double) hi);
        int ok = 0;
        for (int t = 0; t < 33 && !ok; ++t) {
            int d = (t + 1) / 2 * ((t & 1) ? 1 : -1);
            float lo = lo0;
            for (int q = 0; q < abs(d); ++q) lo = nextafterf(lo, d > 0 ? INFINITY : -INFINITY);
            int bad = 0;
            for (int k = 1; k <= 32768 && !bad; ++k) bad = fmaf(x[k], hi, x[k] * lo) != (float) ((double) x[k] * w);
            if (!bad) { ok = 1; if (d) { ++nsh; if (abs(d) > maxd) maxd = abs(d); } }
        }
        if (!ok) for (int k = 1; k <= 32768; ++k) left += fmaf(x[k], hi, x[k] * lo0) != (float) ((double) x[k] * w);
    }
    if (shifted) *shifted = nsh;
    if (max_shift) *max_shift = maxd;
    return left;
}
