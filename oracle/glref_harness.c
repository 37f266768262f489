/*
 * oracle/glref_harness.c -- TEST INFRASTRUCTURE: the reference's GL audio path, executed.
 *
 * Runs the reference's REAL rd_new() and rd_update() -- shader tree loaded from /root/reference/shaders/glava through the
 * reference's own GLSL preprocessor (glsl_ext.c: #include, #request, #expand), programs compiled and linked, the accel_fft
 * passes of render.c:2188-2303 (GL_MAX gravity store, gravity_pass.frag, ring copy, average_pass.frag, smooth_pass.frag)
 * drawn into their GL_R16 1-D textures -- over a real OpenGL 4.5 core context: Mesa's software rasteriser (llvmpipe),
 * reached WITHOUT an X server or EGL by loading swrast_dri.so and driving its DRI screen / context / drawable interface
 * directly (GL/internal/dri_interface.h).  glava/render.c is unity-included from where it lies; nothing is copied.
 *
 * What it hands back after every update, per audio channel, as the exact 16-bit texels of the reference's textures:
 *     up   the uploaded transform_fft output            (glTexImage1D GL_R16, render.c:521-524)
 *     gr   the gravity store after the pass             (render.c:2199-2228)
 *     av   the average of the gravity ring              (render.c:2230-2265; == gr when setavgframes 1)
 *     sm   the pre-smoothed texture the modules sample  (smooth_pass.frag: smooth_audio() at every texel centre, render.c:2277-2303)
 * tests/golden/make_gl_golden.py records them as tests/golden/gl_vectors.npz; tests/test_gl_reference.py compares the
 * oracle's restatements and (on the GPU box) the HIP path's gl_storage chain with them.
 *
 * Built by oracle/Makefile into oracle/_ref/libglvglref.so when /root/reference and Mesa's swrast_dri.so are present.
 */
#include <dlfcn.h>
#include <GL/internal/dri_interface.h>

/* GLV_RENDER_C: the untouched reference (default), or the copy oracle/Makefile produces by applying integration/render_hip.patch to it
 * (-> oracle/_ref/libglvglref_hip.so, linked against the product library: the reference host WITH the MI355X binding, over the same
 * software GL -- tests/test_gl_reference.py compares the texture its modules sample with the unpatched run's) */
#ifndef GLV_RENDER_C
#define GLV_RENDER_C "/root/reference/glava/render.c"
#endif
#include GLV_RENDER_C

/* ---- symbols rd_new()/rd_update() reference from the X11 window code ------------------------------------------------ */
void xwin_assign_icon_bmp(struct gl_wcb* a, void* b, const char* c) { (void) a; (void) b; (void) c; }
unsigned int xwin_copyglbg(struct glava_renderer* r, unsigned int t) { (void) r; (void) t; return 0; }
bool xwin_should_render(struct gl_wcb* a, void* b) { (void) a; (void) b; return true; }
void xwin_wait_for_wm(void) {}

/* ---- an off-screen window backend on Mesa's swrast driver ------------------------------------------------------------ */
#define GLREF_W 64
#define GLREF_H 64
static void sw_get_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* p) { (void) d; (void) p; *x = 0; *y = 0; *w = GLREF_W; *h = GLREF_H; }
static void sw_put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* p) { (void) d; (void) op; (void) x; (void) y; (void) w; (void) h; (void) data; (void) p; }
static void sw_get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* p) { (void) d; (void) x; (void) y; (void) p; memset(data, 0, (size_t) w * h * 4); }
static void sw_put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* p) { (void) d; (void) op; (void) x; (void) y; (void) w; (void) h; (void) stride; (void) data; (void) p; }
static void sw_get_image2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* p) { (void) d; (void) x; (void) y; (void) w; (void) p; memset(data, 0, (size_t) stride * h); }
static const __DRIswrastLoaderExtension sw_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = sw_get_info, .putImage = sw_put_image, .getImage = sw_get_image, .putImage2 = sw_put_image2, .getImage2 = sw_get_image2,
};
static const __DRIextension* sw_loader_exts[] = { &sw_loader.base, NULL };
static void* (*sw_get_proc)(const char*) = NULL;
static const __DRIcoreExtension* sw_core = NULL;
static const __DRIswrastExtension* sw_drv = NULL;
static __DRIscreen* sw_screen = NULL;
static const __DRIconfig** sw_configs = NULL;

static bool sw_offscreen(void) { return true; }
static void sw_init(void) {
    if (sw_screen) return;
    const char* path = getenv("GLREF_SWRAST_DRI");
    void* drv = dlopen(path ? path : "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) { fprintf(stderr, "glref: %s\n", dlerror()); abort(); }
    const __DRIextension** (*get_exts)(void) = dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) { fprintf(stderr, "glref: swrast_dri.so has no __driDriverGetExtensions_swrast\n"); abort(); }
    const __DRIextension** exts = get_exts();
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) sw_core = (const __DRIcoreExtension*) exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw_drv = (const __DRIswrastExtension*) exts[i];
    }
    if (!sw_core || !sw_drv || sw_drv->base.version < 4) { fprintf(stderr, "glref: DRI core / swrast v4 extension missing\n"); abort(); }
    sw_screen = sw_drv->createNewScreen2(0, sw_loader_exts, exts, &sw_configs, NULL);
    if (!sw_screen) { fprintf(stderr, "glref: createNewScreen2 failed\n"); abort(); }
    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    sw_get_proc = glapi ? (void* (*)(const char*)) dlsym(glapi, "_glapi_get_proc_address") : NULL;
    if (!sw_get_proc) { fprintf(stderr, "glref: libglapi.so.0 / _glapi_get_proc_address missing\n"); abort(); }
}
static void* sw_create_and_bind(const char* name, const char* class, const char* type, const char** states, size_t states_sz,
                                int w, int h, int x, int y, int major, int minor, bool clickthrough, bool off) {
    (void) name; (void) class; (void) type; (void) states; (void) states_sz; (void) w; (void) h; (void) x; (void) y; (void) clickthrough; (void) off;
    unsigned err = 0;
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, (uint32_t) major, __DRI_CTX_ATTRIB_MINOR_VERSION, (uint32_t) minor };
    __DRIcontext* ctx = sw_drv->createContextAttribs(sw_screen, __DRI_API_OPENGL_CORE, sw_configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) { fprintf(stderr, "glref: createContextAttribs(%d.%d core) failed, error %u\n", major, minor, err); return NULL; }
    __DRIdrawable* dr = sw_drv->createNewDrawable(sw_screen, sw_configs[0], NULL);
    if (!dr || !sw_core->bindContext(ctx, dr, dr)) { fprintf(stderr, "glref: drawable / bindContext failed\n"); return NULL; }
    if (!gladLoadGLLoader((GLADloadproc) sw_get_proc)) { fprintf(stderr, "glref: glad could not load GL\n"); return NULL; }
    return ctx;
}
static bool sw_false(void* p) { (void) p; return false; }
static bool sw_true(void* p) { (void) p; return true; }
static void sw_void(void* p) { (void) p; }
static void sw_terminate(void) {}
static void sw_get_pos(void* p, int* a, int* b) { (void) p; *a = 0; *b = 0; }
static void sw_get_fbsize(void* p, int* a, int* b) { (void) p; *a = GLREF_W; *b = GLREF_H; }
static void sw_geometry(void* p, int x, int y, int w, int h) { (void) p; (void) x; (void) y; (void) w; (void) h; }
static void sw_set_int(int v) { (void) v; }
static void sw_set_bool(bool v) { (void) v; }
static double sw_get_time(void* p) { (void) p; return 1e-4; }       /* 10 000 frames until the ur / fr counters roll over: ur stays what the harness set */
static void sw_set_time(void* p, double t) { (void) p; (void) t; }
static void sw_set_visible(void* p, bool v) { (void) p; (void) v; }
static const char* sw_environment(void) { return NULL; }
struct gl_wcb wcb_glx = {
    .name = "glx", .offscreen = sw_offscreen, .init = sw_init, .create_and_bind = sw_create_and_bind,
    .should_close = sw_false, .should_render = sw_true, .bg_changed = sw_false, .swap_buffers = sw_void, .raise = sw_void,
    .destroy = sw_void, .terminate = sw_terminate, .get_pos = sw_get_pos, .get_fbsize = sw_get_fbsize, .set_geometry = sw_geometry,
    .set_swap = sw_set_int, .set_floating = sw_set_bool, .set_decorated = sw_set_bool, .set_focused = sw_set_bool,
    .set_maximized = sw_set_bool, .set_transparent = sw_set_bool, .get_time = sw_get_time, .set_time = sw_set_time,
    .set_visible = sw_set_visible, .get_environment = sw_environment,
};

/* ---- the harness ---------------------------------------------------------------------------------------------------- */
static struct rd_bind glref_no_binds[1] = { { .name = NULL } };

/* config_dir: the user's configuration directory (rc.glsl, <module>.glsl, smooth_parameters.glsl and links to the module and
 * util directories, as GLava's --copy-config lays it out); defaults_dir: the installed shaders/glava tree (the `@` includes);
 * they may be the same directory.  requests: NULL-terminated `#request` bodies applied after rc.glsl (e.g. "setbufsize 1024").
 * Note that smooth_parameters.glsl is included by the module's shaders and its #request lines (setavgframes, setavgwindow,
 * setgravitystep, ...) run AFTER these requests: to vary them, vary the config directory's copy.  ur: the updates-per-second
 * figure gravity divides by (the renderer measures it, render.c:2387; the harness pins it).  Returns the renderer, or NULL. */
void* glref_create(const char* config_dir, const char* defaults_dir, const char** requests, float ur) {
    const char* paths[3] = { config_dir, defaults_dir, NULL };
    struct glava_renderer* r = rd_new(paths, "rc.glsl", requests, "glx", glref_no_binds, STDIN_TYPE_NONE, false, false, true);
    if (!r) return NULL;
    r->gl->ur = ur;
    return r;
}

const char* glref_gl_version(void) { return (const char*) glGetString(GL_VERSION); }
const char* glref_gl_renderer(void) { return (const char*) glGetString(GL_RENDERER); }
unsigned glref_bufsize(void* h) { return (unsigned) ((struct glava_renderer*) h)->bufsize_request; }
unsigned glref_avg_frames(void* h) { return (unsigned) ((struct glava_renderer*) h)->gl->avg_frames; }
int glref_accel_fft(void* h) { return ((struct glava_renderer*) h)->gl->accel_fft; }
float glref_ur(void* h) { return ((struct glava_renderer*) h)->gl->ur; }
float glref_smooth_factor(void* h) { return ((struct glava_renderer*) h)->gl->smooth_factor; }

static void glref_read(GLuint tex, size_t n, uint16_t* out) {
    if (!out) return;
    if (!tex) { memset(out, 0, n * sizeof(uint16_t)); return; }
    glBindTexture(GL_TEXTURE_1D, tex);
    glPixelStorei(GL_PACK_ALIGNMENT, 1);
    glGetTexImage(GL_TEXTURE_1D, 0, GL_RED, GL_UNSIGNED_SHORT, out);
}

/* one rd_update(); lb / rb: the time-domain snapshots of glava.c:528-537 (transformed in place by the reference, as there).
 * texels: uint16 [2 channels][4 textures up, gr, av, sm][n] -- what the reference's GL path holds afterwards. */
int glref_update(void* h, float* lb, float* rb, size_t bsz, int modified, uint16_t* texels) {
    struct glava_renderer* r = h;
    struct gl_data* gl = r->gl;
    if (!rd_update(r, lb, rb, bsz, modified != 0)) return -1;
    glFinish();
    int found = 0;
    for (size_t s = 0; s < gl->stages_sz; ++s)
        for (size_t b = 0; b < gl->stages[s].binds_sz; ++b) {
            struct gl_bind* bind = &gl->stages[s].binds[b];
            int ch = bind->src_type == SRC_AUDIO_L ? 0 : bind->src_type == SRC_AUDIO_R ? 1 : -1;
            if (ch < 0 || !bind->optimize_fft) continue;
            uint16_t* base = texels + (size_t) ch * 4 * bsz;
#ifdef GLV_GLREF_HIP
            /* patched build, GL passes on the MI355X: the bind's own texture (render_hip.patch) is what the module samples */
            glref_read(bind->hip_tex ? bind->hip_tex : ch == 0 ? gl->audio_tex_l : gl->audio_tex_r, bsz, base);
#else
            glref_read(ch == 0 ? gl->audio_tex_l : gl->audio_tex_r, bsz, base);
#endif
            glref_read(bind->gr_store.tex, bsz, base + bsz);
            glref_read(gl->avg_frames > 1 ? bind->av.tex : bind->gr_store.tex, bsz, base + 2 * bsz);
            glref_read(bind->sm.tex, bsz, base + 3 * bsz);
            ++found;
        }
    return found;
}

void glref_destroy(void* h) { rd_destroy((struct glava_renderer*) h); }
#ifdef GLV_GLREF_HIP
/* patched build: the accel path's GL passes on the MI355X (GLAVA_HIP_GL) or on the GL, and the log mode of the operators */
void glref_hip(int gl_passes_on_hip, unsigned log_mode) { glv_hip_env(); glv_hip_gl = gl_passes_on_hip; glv_hip_log_mode = log_mode; }
volatile int glv_audio_publishes_spectra = 0;
#endif
