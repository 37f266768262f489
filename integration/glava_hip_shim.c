/*
 * integration/glava_hip_shim.c -- the operator shim of INTEGRATION.md section 1, as real code.
 *
 * This file is meant to be #included into glava/render.c after transform_fft (render.c:847): it needs
 * the private `struct gl_data` (render.c:166-207) and `struct gl_sampler_data` (render.c:115-118).  It
 * defines drop-in operators with the reference's signature
 *     void apply(struct gl_data*, void** udata, void* data)          (render.c:106-111)
 * that forward to libglvspectrum.so through the C ABI of include/glv_spectrum.h:
 *     transform_fft_hip      == transform_fft      render.c:783-847
 *     transform_gravity_hip  == transform_gravity  render.c:720-736
 *     transform_average_hip  == transform_average  render.c:738-771
 *     transform_wrange_hip   == transform_wrange   render.c:773-781
 *     transform_fga_hip      == the fft,gravity,average triple of handle_audio (render.c:2149-2153), one launch
 *     transform_gl_hip       == the accel_fft branch behind the transform list (render.c:2176-2303): FFT, upload, GL_MAX store,
 *                               gravity / average / pre-smoothing passes -- GLava's shipped configuration -- as one call
 * plus glv_hip_release(slot), to be called from rd_destroy before it free()s the slot (render.c:2463-2469).
 *
 * integration/shim_harness.c compiles it against the unmodified reference sources (oracle/Makefile ->
 * oracle/_ref/libglvshim.so) and tests/test_gpu_parity.py::test_reference_host_through_shim drives the
 * reference's own call sequence through it on the GPU.
 */
#include <glv_spectrum.h>

/* one handle per (bind, transform) slot, stored where the reference keeps its own state
   (gl->t_data[c], render.c:2140-2156).  rd_destroy free()s every slot, so the slot holds a small
   malloc'd box; the device state behind it is released by glv_hip_release first. */
struct glv_box { unsigned long long magic; glv_state* st; };
#define GLV_BOX_MAGIC 0x676c765f626f7821ULL   /* "glv_box!": tells a box from a slot the stock operators filled (float state) */

/* Run-time knobs (environment, read once):
 *   GLAVA_HIP_DEVICE=<ordinal>     the HIP device the states live on (default 0)
 *   GLAVA_HIP_LOG_MODE=0|1|2       glv_params.log_mode: 0 (the host drop-in's default since round 6) the bit-faithful fp64 table log -- the
 *                                  reference's floats, bit for bit: one GLava instance is bound by launch latency, not by the log, so
 *                                  the patched host gives up nothing for being exact; 1 the hardware log2 (<= 1.8e-7 relative on every
 *                                  input of the stage; the batched API's default, where the log is 25 % of a launch); 2 the audit form
 *   GLAVA_HIP_GL=0                 keep the accel path's GL passes on the GL (only the per-frame FFT runs on the MI355X)
 *                                  (the smoothing SHAPE needs no knob: glv_hip_scan_shape below reads it from the shader the host compiles)
 *   GLAVA_HIP_SMOOTH_FACTOR=<f>    OVERRIDES the pre-smoothing pass's _SMOOTH_FACTOR.  Without it the factor is the host's own:
 *                                  gl_data.smooth_factor (render.c:184, `#request setsmoothfactor` render.c:1198-1200), through the
 *                                  same "%.6f" text the reference prepends to every shader as `#define _SMOOTH_FACTOR` (render.c:317-326)
 *                                  and the GLSL compiler reads back as a float literal */
static unsigned glv_hip_log_mode = 0;      /* glv_params.log_mode for new boxes (GLAVA_HIP_LOG_MODE; tests flip it) */
static int glv_hip_env_done = 0, glv_hip_dev = 0, glv_hip_gl = 1;
static float glv_hip_smooth_factor = -1.0f;   /* < 0: none given, the host's gl_data.smooth_factor is used */
static void glv_hip_env(void) {
    if (glv_hip_env_done) return;
    glv_hip_env_done = 1;
    const char* e;
    if ((e = getenv("GLAVA_HIP_DEVICE")) && e[0]) glv_hip_dev = atoi(e);
    if ((e = getenv("GLAVA_HIP_LOG_MODE")) && e[0] >= '0' && e[0] <= '2' && !e[1]) glv_hip_log_mode = (unsigned) (e[0] - '0');
    if ((e = getenv("GLAVA_HIP_GL")) && e[0] == '0') glv_hip_gl = 0;
    if ((e = getenv("GLAVA_HIP_SMOOTH_FACTOR")) && e[0]) glv_hip_smooth_factor = (float) atof(e);
}
static bool glv_hip_gl_on(void) { glv_hip_env(); return glv_hip_gl != 0; }

/* smooth_audio()'s SHAPE (glv_params ABI 7).  The host has no field for it: ROUND_FORMULA, SAMPLE_MODE, SAMPLE_HYBRID_WEIGHT, SAMPLE_SCALE and
   SAMPLE_RANGE are GLSL `#define`s of shaders/glava/smooth_parameters.glsl:17-42, which a user's ~/.config/glava/smooth_parameters.glsl or a
   module's config re-defines; glsl_ext.c:143-157 turns every `#define X` into `#ifdef X / #undef X / #endif / #define X ...`, so in the
   processed text the LAST definition is the one the GLSL compiler keeps.  shaderload() (render_hip.patch) hands every processed source
   here; the pre-smoothing pass (util/smooth_pass.frag, render.c:1634-1636) is the one shader whose smooth_audio() the MI355X replaces.
   A value that is not one of the stock names / a plain numeric literal (an expression, another macro, a definition inside a conditional
   the scan cannot evaluate -- `#if` lines between the last definition and the end of the file are not interpreted) marks the shape
   unreadable: glv_hip_gl_supported() then keeps the GL passes on the GL, whatever they render there. */
static struct { bool readable; unsigned formula, mode; float hybrid, scale, range; } glv_hip_shape = { true, 0u, 0u, 0.0f, 0.0f, 0.0f };
static const char* glv_hip_last_define(const char* text, size_t len, const char* name, size_t* vlen) {
    const size_t nl = strlen(name);
    const char* found = NULL;
    size_t i = 0;
    while (i < len) {                                                    /* line by line */
        size_t j = i;
        while (j < len && (text[j] == ' ' || text[j] == '\t')) ++j;
        if (j < len && text[j] == '#') {
            ++j;
            while (j < len && (text[j] == ' ' || text[j] == '\t')) ++j;
            if (len - j > 6 && !strncmp(text + j, "define", 6) && (text[j + 6] == ' ' || text[j + 6] == '\t')) {
                j += 6;
                while (j < len && (text[j] == ' ' || text[j] == '\t')) ++j;
                if (len - j > nl && !strncmp(text + j, name, nl) && (text[j + nl] == ' ' || text[j + nl] == '\t')) {
                    j += nl;
                    while (j < len && (text[j] == ' ' || text[j] == '\t')) ++j;
                    size_t e = j;
                    while (e < len && text[e] != '\n' && text[e] != '\r' && !(text[e] == '/' && e + 1 < len && (text[e + 1] == '/' || text[e + 1] == '*'))) ++e;
                    while (e > j && (text[e - 1] == ' ' || text[e - 1] == '\t')) --e;
                    found = text + j; *vlen = e - j;
                }
            }
        }
        while (i < len && text[i] != '\n') ++i;
        ++i;
    }
    return found;
}
static bool glv_hip_literal(const char* v, size_t vlen, float* out) {     /* a GLSL numeric literal: 8, 0.9, .65, 1e-1, 0.9f, (0.9) */
    char buf[48];
    while (vlen >= 2 && v[0] == '(' && v[vlen - 1] == ')') { ++v; vlen -= 2; }
    if (vlen == 0 || vlen >= sizeof(buf)) return false;
    memcpy(buf, v, vlen); buf[vlen] = 0;
    if (buf[vlen - 1] == 'f' || buf[vlen - 1] == 'F') buf[vlen - 1] = 0;
    char* end = NULL;
    *out = strtof(buf, &end);
    return end != buf && *end == 0 && (buf[0] == '.' || (buf[0] >= '0' && buf[0] <= '9'));
}
void glv_hip_scan_shape(const char* rpath, const char* text, size_t len) {
    static const char* formulas[] = { "sinusoidal", "circular", "linear" }, * modes[] = { "average", "maximum", "hybrid" };
    const size_t pl = rpath ? strlen(rpath) : 0;
    if (pl < 16 || strcmp(rpath + pl - 16, "smooth_pass.frag") || !text) return;
    glv_hip_shape.readable = true; glv_hip_shape.formula = 0u; glv_hip_shape.mode = 0u;
    glv_hip_shape.hybrid = glv_hip_shape.scale = glv_hip_shape.range = 0.0f;
    size_t vl = 0;
    const char* v;
    if ((v = glv_hip_last_define(text, len, "ROUND_FORMULA", &vl))) {
        unsigned k = 0;
        while (k < 3 && !(strlen(formulas[k]) == vl && !strncmp(v, formulas[k], vl))) ++k;
        if (k < 3) glv_hip_shape.formula = k; else glv_hip_shape.readable = false;
    }
    if ((v = glv_hip_last_define(text, len, "SAMPLE_MODE", &vl))) {
        unsigned k = 0;
        while (k < 3 && !(strlen(modes[k]) == vl && !strncmp(v, modes[k], vl))) ++k;
        if (k < 3) glv_hip_shape.mode = k; else glv_hip_shape.readable = false;
    }
    if ((v = glv_hip_last_define(text, len, "SAMPLE_HYBRID_WEIGHT", &vl)) && !glv_hip_literal(v, vl, &glv_hip_shape.hybrid)) glv_hip_shape.readable = false;
    if ((v = glv_hip_last_define(text, len, "SAMPLE_SCALE", &vl)) && !glv_hip_literal(v, vl, &glv_hip_shape.scale)) glv_hip_shape.readable = false;
    if ((v = glv_hip_last_define(text, len, "SAMPLE_RANGE", &vl)) && !glv_hip_literal(v, vl, &glv_hip_shape.range)) glv_hip_shape.readable = false;
    /* what the library refuses (glv_api.cpp ensure_bar_tables: positions past the end of the texture) stays on the GL as well */
    const float sc = glv_hip_shape.scale != 0.0f ? glv_hip_shape.scale : 8.0f, rg = glv_hip_shape.range != 0.0f ? glv_hip_shape.range : 0.9f;
    const float hw = glv_hip_shape.hybrid != 0.0f ? glv_hip_shape.hybrid : 0.65f;
    if (!(sc > 0.0f && sc <= 1e6f) || !(rg > 0.0f && rg < 1.0f) || !(-logf(1.0f - rg) / sc <= 1.0f) || !(hw > 0.0f && hw <= 1.0f)) glv_hip_shape.readable = false;
    if (!glv_hip_shape.readable) fprintf(stderr, "glv: the smoothing shape of '%s' is not one the MI355X path reads (SAMPLE_* / ROUND_FORMULA): its GL passes stay on the GL\n", rpath);
}

static void glv_hip_fill(const struct gl_data* d, size_t sz, glv_params* p) {
    glv_params_default(p);
    p->n            = (uint32_t) sz;          /* bsz, #request setbufsize (render.c:1176) */
    p->fft_scale    = d->fft_scale;           /* render.c:845 */
    p->fft_cutoff   = d->fft_cutoff;
    p->gravity_step = d->gravity_step;        /* render.c:728 */
    p->ur           = d->ur;                  /* measured updates/s, render.c:2387 */
    p->avg_frames   = (uint32_t) d->avg_frames;
    p->avg_window   = d->avg_window;
    glv_hip_env();
    p->log_mode     = glv_hip_log_mode;
}
/* the accel path's GL passes (render.c:2188-2303) with their GL_R16 storage: the shaders' Hamming window, newest frame first
   (average_pass.frag:19-45), the pre-smoothing pass at the texel centres (util/smooth_pass.frag) */
static void glv_hip_fill_gl(const struct gl_data* d, size_t sz, glv_params* p) {
    glv_hip_fill(d, sz, p);
    p->gl_storage = 1; p->avg_window_kind = 1;
    p->bars = (uint32_t) sz; p->bar_phase = 0.5f;
    p->round_formula = glv_hip_shape.formula; p->sample_mode = glv_hip_shape.mode;      /* glv_hip_scan_shape: the pre-smoothing shader's own defines */
    p->sample_hybrid_weight = glv_hip_shape.hybrid; p->sample_scale = glv_hip_shape.scale; p->sample_range = glv_hip_shape.range;
    if (glv_hip_smooth_factor >= 0.0f) p->smooth_factor = glv_hip_smooth_factor;
    else {
        /* what smooth.glsl:25-27 computes with: the float literal of the header line shaderload() formats (render.c:317-326) */
        char lit[64];
        snprintf(lit, sizeof(lit), "%.6f", (double) d->smooth_factor);
        p->smooth_factor = strtof(lit, NULL);
    }
}

/* Parameters outside what the library takes -- a window that is not a power of two in [256, 32768] (setbufsize is
   unchecked, and bufscale can produce any size), more than GLV_MAX_AVG_FRAMES averaging frames -- are left to the stock
   CPU operators: a host that runs on the reference must not be terminated by the shim (render_hip.patch tests this
   before it routes a call here). */
static bool glv_hip_supported(const struct gl_data* d, size_t sz) {
    return sz >= 256 && sz <= 32768 && (sz & (sz - 1)) == 0 && d->avg_frames >= 1 && d->avg_frames <= GLV_MAX_AVG_FRAMES;
}

/* ... and for the GL passes on the MI355X: a pre-smoothing factor the library builds tap tables for (a factor outside (0, 1] -- every
   bar without taps, or NaN weights in the shader -- stays on the GL, whatever it renders there) */
static bool glv_hip_gl_supported(const struct gl_data* d) { return d->smooth_factor > 0.0f && d->smooth_factor <= 1.0f && glv_hip_shape.readable; }

static glv_state* glv_hip_slot(struct gl_data* d, void** udata, size_t sz) {
    struct glv_box* b = *udata;
    if (!b) {                                  /* lazily, like ALLOC_ONCE (render.c:662-666) */
        glv_params p; glv_hip_fill(d, sz, &p);
        b = calloc(1, sizeof(*b));
        if (!b) { fprintf(stderr, "glv: out of memory\n"); glava_abort(); }
        b->magic = GLV_BOX_MAGIC;
        if (glv_state_create(&p, glv_hip_dev, &b->st) != GLV_OK) {
            fprintf(stderr, "glv: %s\n", glv_last_error());
            glava_abort();                     /* the reference's error convention (glava.h:17) */
        }
        *udata = b;
    }
    return b->st;
}

#define GLV_HIP_OPERATOR(name, call)                                           \
    void name(struct gl_data* d, void** udata, void* data) {                  \
        struct gl_sampler_data* s = (struct gl_sampler_data*) data;           \
        glv_params p; glv_hip_fill(d, s->sz, &p);                             \
        if (call(&p, glv_hip_slot(d, udata, s->sz), s->buf) != GLV_OK) {      \
            fprintf(stderr, "glv: %s\n", glv_last_error()); glava_abort();   \
        }                                                                     \
    }
GLV_HIP_OPERATOR(transform_fft_hip,     glv_fft)
GLV_HIP_OPERATOR(transform_gravity_hip, glv_gravity)
GLV_HIP_OPERATOR(transform_average_hip, glv_average)
GLV_HIP_OPERATOR(transform_wrange_hip,  glv_wrange)
GLV_HIP_OPERATOR(transform_fga_hip,     glv_fft_gravity_average)

/* The whole accel path of handle_audio behind the transform list (render.c:2176-2303: per-frame transform_fft, GL_R16 upload, GL_MAX
   store + gravity pass, ring + average pass, pre-smoothing pass) as one call: buf holds the bind's n samples (left as they are), *texels
   (malloc'd here, free()d by rd_destroy) receives the n GL_R16 texels of the texture the module samples.  The state slot holds the gravity
   store and the ring, as the reference's gr_store / gr.out textures do. */
void transform_gl_hip(struct gl_data* d, void** udata, unsigned short** texels, float* buf, size_t sz, bool smooth_pass) {
    struct glv_box* b = *udata;
    glv_params p; glv_hip_fill_gl(d, sz, &p);
    if (!b) {
        b = calloc(1, sizeof(*b));
        if (!b) { fprintf(stderr, "glv: out of memory\n"); glava_abort(); }
        b->magic = GLV_BOX_MAGIC;
        if (glv_state_create(&p, glv_hip_dev, &b->st) != GLV_OK) { fprintf(stderr, "glv: %s\n", glv_last_error()); glava_abort(); }
        *udata = b;
    }
    if (!*texels && !(*texels = calloc(sz, sizeof(unsigned short)))) { fprintf(stderr, "glv: out of memory\n"); glava_abort(); }
    if (glv_gl_texture(&p, b->st, buf, smooth_pass ? 1 : 0, *texels) != GLV_OK) { fprintf(stderr, "glv: %s\n", glv_last_error()); glava_abort(); }
}

/* rd_destroy hook: release the device state of a *_hip slot; the box itself is free()d by rd_destroy.  Slots that the
   stock operators filled (parameters the library does not take fall back to them) hold the reference's own float state --
   at least sz >= 2 floats -- and are left alone. */
void glv_hip_release(void* slot) {
    struct glv_box* b = slot;
    if (b && b->magic == GLV_BOX_MAGIC && b->st) { glv_state_destroy(b->st); b->st = NULL; }
}
