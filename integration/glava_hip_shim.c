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

static unsigned glv_hip_log_mode = 1;      /* glv_params.log_mode for new boxes (tests flip it) */

static void glv_hip_fill(const struct gl_data* d, size_t sz, glv_params* p) {
    glv_params_default(p);
    p->n            = (uint32_t) sz;          /* bsz, #request setbufsize (render.c:1176) */
    p->fft_scale    = d->fft_scale;           /* render.c:845 */
    p->fft_cutoff   = d->fft_cutoff;
    p->gravity_step = d->gravity_step;        /* render.c:728 */
    p->ur           = d->ur;                  /* measured updates/s, render.c:2387 */
    p->avg_frames   = (uint32_t) d->avg_frames;
    p->avg_window   = d->avg_window;
    p->log_mode     = glv_hip_log_mode;
}

/* Parameters outside what the library takes -- a window that is not a power of two in [256, 32768] (setbufsize is
   unchecked, and bufscale can produce any size), more than GLV_MAX_AVG_FRAMES averaging frames -- are left to the stock
   CPU operators: a host that runs on the reference must not be terminated by the shim (render_hip.patch tests this
   before it routes a call here). */
static bool glv_hip_supported(const struct gl_data* d, size_t sz) {
    return sz >= 256 && sz <= 32768 && (sz & (sz - 1)) == 0 && d->avg_frames >= 1 && d->avg_frames <= GLV_MAX_AVG_FRAMES;
}

static glv_state* glv_hip_slot(struct gl_data* d, void** udata, size_t sz) {
    struct glv_box* b = *udata;
    if (!b) {                                  /* lazily, like ALLOC_ONCE (render.c:662-666) */
        glv_params p; glv_hip_fill(d, sz, &p);
        b = calloc(1, sizeof(*b));
        if (!b) { fprintf(stderr, "glv: out of memory\n"); glava_abort(); }
        b->magic = GLV_BOX_MAGIC;
        if (glv_state_create(&p, /*device*/ 0, &b->st) != GLV_OK) {
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

/* rd_destroy hook: release the device state of a *_hip slot; the box itself is free()d by rd_destroy.  Slots that the
   stock operators filled (parameters the library does not take fall back to them) hold the reference's own float state --
   at least sz >= 2 floats -- and are left alone. */
void glv_hip_release(void* slot) {
    struct glv_box* b = slot;
    if (b && b->magic == GLV_BOX_MAGIC && b->st) { glv_state_destroy(b->st); b->st = NULL; }
}
