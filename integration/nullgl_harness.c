/*
 * integration/nullgl_harness.c -- TEST INFRASTRUCTURE: runs the reference's real rd_update() -- the rd_update
 * prelude (bufscale decimation, keyframe interpolation: render.c:1765-1809), handle_audio with all of its branches
 * (render.c:2113-2186: modified / !modified, accel_fft on and off, the optimize_fft truncation), the counters and
 * keyframe pushes (render.c:2347-2387) and rd_destroy's slot release (render.c:2463-2469) -- without a GL context.
 *
 * How: glava/render.c is unity-included from where it lies (GLV_RENDER_C: the untouched /root/reference/glava/render.c,
 * or the copy oracle/Makefile produces by applying integration/render_hip.patch to it; nothing is copied into this
 * repository), the renderer state (struct gl_data, one stage, two audio binds with the transformation list the bars
 * module requests) is built by hand instead of by rd_new (which needs shaders and a window), and every GL entry point
 * is resolved by glad to a no-op -- except glTexImage1D, which records what handle_audio hands to the GL_R16 texture
 * (render.c:521-524), the value the rest of GLava consumes.  Two shared objects come out of this one file:
 *     oracle/_ref/libglvnullgl_ref.so   the reference as it is            (CPU only)
 *     oracle/_ref/libglvnullgl_hip.so   the reference + render_hip.patch  (links libglvspectrum.so)
 * tests/test_handle_audio.py feeds both the same frame sequences and compares the uploads.
 */
#ifndef GLV_RENDER_C
#define GLV_RENDER_C "/root/reference/glava/render.c"
#endif
#include GLV_RENDER_C

/* symbols rd_new()/rd_update() reference from the X11 window code */
struct gl_wcb wcb_glx;
void xwin_assign_icon_bmp(struct gl_wcb* a, void* b, const char* c) { (void) a; (void) b; (void) c; }
unsigned int xwin_copyglbg(struct glava_renderer* r, unsigned int t) { (void) r; (void) t; return 0; }
bool xwin_should_render(struct gl_wcb* a, void* b) { (void) a; (void) b; return true; }
void xwin_wait_for_wm(void) {}

/* ---- a GL that does nothing ------------------------------------------------------------------------------------ */
typedef struct { int unit; size_t width; float* data; unsigned short* texels; } ng_upload;      /* data: a GL_FLOAT upload; texels: a GL_UNSIGNED_SHORT one */
static ng_upload ng_log[64];
static size_t ng_n = 0;
static int ng_unit = 0;
static GLuint ng_ids = 100;

static void* ng_noop(void) { return NULL; }
static const GLubyte* ng_GetString(GLenum name) { (void) name; return (const GLubyte*) "4.6.0 nullgl"; }
static GLenum ng_CheckFramebufferStatus(GLenum t) { (void) t; return GL_FRAMEBUFFER_COMPLETE; }
static void ng_GetIntegerv(GLenum pname, GLint* data) { *data = pname == GL_NUM_EXTENSIONS ? 1 : 0; }      /* glad's extension scan */
static const GLubyte* ng_GetStringi(GLenum name, GLuint i) { (void) name; (void) i; return (const GLubyte*) "GL_NV_texture_barrier"; }
static void ng_ActiveTexture(GLenum t) { ng_unit = (int) (t - GL_TEXTURE0); }
static void ng_GenObjects(GLsizei n, GLuint* ids) { for (GLsizei i = 0; i < n; ++i) ids[i] = ++ng_ids; }
static void ng_TexImage1D(GLenum target, GLint level, GLint ifmt, GLsizei w, GLint border, GLenum fmt, GLenum type, const void* data) {
    (void) target; (void) level; (void) border; (void) fmt;
    if (!data || ifmt != GL_R16 || (type != GL_FLOAT && type != GL_UNSIGNED_SHORT) || ng_n >= sizeof(ng_log) / sizeof(ng_log[0])) return;
    ng_upload* u = &ng_log[ng_n++];
    u->unit = ng_unit; u->width = (size_t) w; u->data = NULL; u->texels = NULL;
    if (type == GL_FLOAT) {
        u->data = malloc(sizeof(float) * (size_t) w);
        memcpy(u->data, data, sizeof(float) * (size_t) w);
    } else {                                    /* the patched accel path: the finished GL_R16 texels of the MI355X chain */
        u->texels = malloc(sizeof(unsigned short) * (size_t) w);
        memcpy(u->texels, data, sizeof(unsigned short) * (size_t) w);
    }
}
static void* ng_loader(const char* name) {
    if (!strcmp(name, "glGetString")) return (void*) ng_GetString;
    if (!strcmp(name, "glCheckFramebufferStatus")) return (void*) ng_CheckFramebufferStatus;
    if (!strcmp(name, "glGetIntegerv")) return (void*) ng_GetIntegerv;
    if (!strcmp(name, "glGetStringi")) return (void*) ng_GetStringi;
    if (!strcmp(name, "glActiveTexture")) return (void*) ng_ActiveTexture;
    if (!strcmp(name, "glTexImage1D")) return (void*) ng_TexImage1D;
    if (!strcmp(name, "glGenTextures") || !strcmp(name, "glGenFramebuffers")) return (void*) ng_GenObjects;
    return (void*) ng_noop;
}

/* ---- a window backend that does nothing ------------------------------------------------------------------------- */
static bool nw_false(void* p) { (void) p; return false; }
static bool nw_true(void* p) { (void) p; return true; }
static bool nw_offscreen(void) { return false; }
static void nw_void(void* p) { (void) p; }
static void nw_terminate(void) {}
static void nw_get2(void* p, int* a, int* b) { (void) p; *a = 64; *b = 64; }
static void nw_geometry(void* p, int x, int y, int w, int h) { (void) p; (void) x; (void) y; (void) w; (void) h; }
static double nw_get_time(void* p) { (void) p; return 1e-4; }       /* 10 000 frames until the ur/fr counters roll over */
static void nw_set_time(void* p, double t) { (void) p; (void) t; }
static struct gl_wcb null_wcb = {
    .name = "nullgl", .offscreen = nw_offscreen, .should_close = nw_false, .should_render = nw_true, .bg_changed = nw_false,
    .swap_buffers = nw_void, .raise = nw_void, .destroy = nw_void, .terminate = nw_terminate, .get_pos = nw_get2,
    .get_fbsize = nw_get2, .set_geometry = nw_geometry, .get_time = nw_get_time, .set_time = nw_set_time,
};
static struct rd_bind no_binds[1] = { { .name = NULL } };

/* ---- the hand-built renderer ------------------------------------------------------------------------------------ */
typedef struct {
    unsigned n;                 /* audio_buf_sz (setbufsize) */
    unsigned bufscale;          /* setbufscale, 1 = off (render.c:1765) */
    int interpolate;            /* setinterpolate (render.c:1792) */
    int accel_fft;              /* setaccelfft (render.c:2131) */
    unsigned avg_frames; int avg_window;
    float fft_scale, fft_cutoff, gravity_step, ur, fr;
    unsigned hip_log_mode;      /* patched build only: glv_params.log_mode of the *_hip operators */
    int smooth_pass;            /* setsmoothpass (render.c:2277) */
    int hip_gl;                 /* patched build only: the accel path's GL passes on the MI355X too (GLAVA_HIP_GL) */
    float smooth_factor;        /* setsmoothfactor (render.c:184, 1198-1200); 0 = rd_new's default 0.025 (render.c:916) */
} nullgl_cfg;

typedef struct { struct glava_renderer* r; size_t isz; } nullgl;

static struct gl_bind make_bind(const char* name, int src_type) {
    /* what `#request uniform "audio_l" audio_l` + `#request transform audio_l "window" / "fft" / "gravity" / "avg"`
       of the bars module (shaders/glava/bars/1.frag:12-24) leave behind: render.c:1218-1310 */
    struct gl_bind b = { .name = strdup(name), .type = BIND_SAMPLER1D, .src_type = src_type, .transformations = malloc(1),
                         .t_sz = 0, .gr = { .out = NULL }, .optimize_fft = false };
    static const char* req[4] = { "window", "fft", "gravity", "avg" };
    for (int q = 0; q < 4; ++q)
        for (size_t t = 0; t < sizeof(transform_functions) / sizeof(struct gl_transform); ++t)
            if (!strcmp(transform_functions[t].name, req[q])) {
                ++b.t_sz;
                b.transformations = realloc(b.transformations, b.t_sz * sizeof(void (*)(void*)));
                b.transformations[b.t_sz - 1] = transform_functions[t].apply;
            }
    return b;
}

void* nullgl_create(const nullgl_cfg* c) {
    static bool loaded = false;
    if (!loaded) {
        if (!gladLoadGLLoader((GLADloadproc) ng_loader)) return NULL;
        glad_glTextureBarrierNV = (PFNGLTEXTUREBARRIERNVPROC) ng_noop;     /* extension entry point handle_audio calls (render.c:2219) */
        loaded = true;
    }
#ifdef GLV_NULLGL_HIP
    glv_hip_env();
    glv_hip_log_mode = c->hip_log_mode;
    glv_hip_gl = c->hip_gl;
#endif
    nullgl* h = calloc(1, sizeof(*h));
    struct glava_renderer* r = calloc(1, sizeof(*r));
    struct gl_data* gl = calloc(1, sizeof(*gl));
    r->gl = gl; r->alive = true;
    gl->wcb = &null_wcb; gl->w = NULL; gl->binds = no_binds; gl->stdin_type = STDIN_TYPE_NONE;
    gl->rate = 0; gl->timecycle = 60.0F;
    gl->bufscale = c->bufscale ? c->bufscale : 1; gl->interpolate = c->interpolate != 0; gl->accel_fft = c->accel_fft != 0;
    gl->avg_frames = c->avg_frames; gl->avg_window = c->avg_window != 0;
    gl->fft_scale = c->fft_scale; gl->fft_cutoff = c->fft_cutoff; gl->gravity_step = c->gravity_step;
    gl->ur = c->ur; gl->fr = c->fr; gl->smooth_pass = c->smooth_pass != 0;
    gl->smooth_factor = c->smooth_factor != 0.0f ? c->smooth_factor : 0.025;
    gl->audio_tex_l = 11; gl->audio_tex_r = 12;
    gl->av_utex = calloc(c->avg_frames ? c->avg_frames : 1, sizeof(GLuint));   /* uniform locations of the averaging pass (render.c:1655-1660) */
    gl->stages_sz = 1;
    gl->stages = calloc(1, sizeof(struct gl_sfbo));
    gl->stages[0].shader = 1; gl->stages[0].name = strdup("bars");
    gl->stages[0].binds_sz = 2;
    gl->stages[0].binds = calloc(2, sizeof(struct gl_bind));
    gl->stages[0].binds[0] = make_bind("audio_l", SRC_AUDIO_L);
    gl->stages[0].binds[1] = make_bind("audio_r", SRC_AUDIO_R);
    /* t_count as the "transform" request handler counts it: one per entry, two more per "fft" (render.c:1254-1260) */
    gl->t_count = 2 * (4 + 2);
    gl->t_data = malloc(sizeof(void*) * gl->t_count);
    for (size_t t = 0; t < gl->t_count; ++t) gl->t_data[t] = NULL;
    if (gl->interpolate) {                                                  /* render.c:1679-1690 */
        size_t isz = c->n / gl->bufscale;
        float* ibuf = calloc(isz * 6, sizeof(float));
        for (int q = 0; q < 6; ++q) gl->interpolate_buf[q] = &ibuf[isz * q];
        h->isz = isz;
    }
    h->r = r;
    return h;
}

/* one rd_update(); up_l / up_r receive what was uploaded to the left / right audio texture (texture units 1 / 2),
 * *up_n the number of floats of each.  lb / rb are transformed in place exactly as glava.c's buffers are. */
int nullgl_update(void* hv, float* lb, float* rb, size_t bsz, int modified, float* up_l, float* up_r, size_t* up_n) {
    nullgl* h = hv;
    ng_n = 0;
    if (!rd_update(h->r, lb, rb, bsz, modified != 0)) return -1;
    int got = 0;
    *up_n = 0;
    for (size_t i = 0; i < ng_n; ++i) {
        float* dst = ng_log[i].unit == 1 ? up_l : ng_log[i].unit == 2 ? up_r : NULL;
        if (dst && ng_log[i].data) { memcpy(dst, ng_log[i].data, sizeof(float) * ng_log[i].width); *up_n = ng_log[i].width; ++got; }
        free(ng_log[i].data); free(ng_log[i].texels);
    }
    ng_n = 0;
    return got;
}
/* the same for the patched accel path with the GL passes on the MI355X: tex_l / tex_r receive the GL_R16 texels handle_audio uploaded
 * (GL_UNSIGNED_SHORT) to the left / right audio texture -- the texture the module samples; returns how many such uploads the update made
 * (0 on a frame without new audio: the textures keep the last result), *float_uploads the GL_FLOAT uploads it made besides (none) */
int nullgl_update_texels(void* hv, float* lb, float* rb, size_t bsz, int modified, unsigned short* tex_l, unsigned short* tex_r, int* float_uploads) {
    nullgl* h = hv;
    ng_n = 0;
    if (!rd_update(h->r, lb, rb, bsz, modified != 0)) return -1;
    int got = 0;
    *float_uploads = 0;
    for (size_t i = 0; i < ng_n; ++i) {
        unsigned short* dst = ng_log[i].unit == 1 ? tex_l : ng_log[i].unit == 2 ? tex_r : NULL;
        if (dst && ng_log[i].texels) { memcpy(dst, ng_log[i].texels, sizeof(unsigned short) * ng_log[i].width); ++got; }
        if (ng_log[i].data) ++*float_uploads;
        free(ng_log[i].data); free(ng_log[i].texels);
    }
    ng_n = 0;
    return got;
}
#ifdef GLV_NULLGL_HIP
/* the flag an audio backend that publishes spectra raises (integration/hipfifo.c); defined here so that the patched
   handle_audio of this library sees it without the backend being linked in */
volatile int glv_audio_publishes_spectra = 0;
void nullgl_spectra_in(int on) { glv_audio_publishes_spectra = on ? 1 : 0; }
#endif
float nullgl_ur(void* hv) { return ((nullgl*) hv)->r->gl->ur; }
int nullgl_interpolate_glsl(void* hv) { return ((nullgl*) hv)->r->gl->interpolate_glsl; }
size_t nullgl_bind_t_sz(void* hv, int b) { return ((nullgl*) hv)->r->gl->stages[0].binds[b].t_sz; }

void nullgl_destroy(void* hv) {
    nullgl* h = hv;
    rd_destroy(h->r);          /* releases the slots (and, patched, the device state behind them), frees r and gl */
    free(h);
}
