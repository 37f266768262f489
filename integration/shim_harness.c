/*
 * integration/shim_harness.c -- TEST INFRASTRUCTURE: compiles integration/glava_hip_shim.c into the
 * UNMODIFIED reference translation unit (glava/render.c is unity-included from /root/reference where it
 * lies; nothing is copied) and drives handle_audio's CPU-path call sequence (render.c:2149-2153) once with
 * the reference's operators and once with the *_hip operators, state slots and all.
 * Built by oracle/Makefile into oracle/_ref/libglvshim.so (git-ignored; travels to the GPU box prebuilt).
 */
#include "/root/reference/glava/render.c"

struct gl_wcb wcb_glx;
void xwin_assign_icon_bmp(struct gl_wcb* a, void* b, const char* c) { (void) a; (void) b; (void) c; }
unsigned int xwin_copyglbg(struct glava_renderer* r, unsigned int t) { (void) r; (void) t; return 0; }
bool xwin_should_render(struct gl_wcb* a, void* b) { (void) a; (void) b; return true; }
void xwin_wait_for_wm(void) {}

#include "glava_hip_shim.c"

typedef struct {
    float fft_scale, fft_cutoff, gravity_step, ur;
    unsigned long avg_frames;
    int avg_window;
} glvshim_params;

/* frames: float [nframes][2][n], transformed in place the way rd_update's handle_audio does for a
 * `#request transform ... "fft"` bind on the CPU path:  apply(fft); gravity; average  with slots c, c+1, c+2
 * per channel.  mode 0: the reference's operators.  mode 1: the three *_hip operators.  mode 2: the fused
 * transform_fga_hip (one launch per channel).  Slots persist across frames and are released at the end the
 * way rd_destroy would (glv_hip_release, then free). */
int glvshim_run(const glvshim_params* sp, int mode, unsigned log_mode, float* frames, size_t n, int nframes) {
    struct gl_data gl;
    memset(&gl, 0, sizeof(gl));
    gl.fft_scale = sp->fft_scale; gl.fft_cutoff = sp->fft_cutoff; gl.gravity_step = sp->gravity_step;
    gl.ur = sp->ur; gl.avg_frames = sp->avg_frames; gl.avg_window = sp->avg_window != 0;
    glv_hip_log_mode = log_mode;
    void* t_data[6] = { NULL, NULL, NULL, NULL, NULL, NULL };
    for (int f = 0; f < nframes; ++f)
        for (int ch = 0; ch < 2; ++ch) {
            struct gl_sampler_data d = { .buf = frames + ((size_t) f * 2 + ch) * n, .sz = n };
            size_t c = (size_t) ch * 3;
            if (mode == 0) {
                transform_fft(&gl, &t_data[c], &d);
                transform_gravity(&gl, &t_data[c + 1], &d);
                transform_average(&gl, &t_data[c + 2], &d);
            } else if (mode == 1) {
                transform_fft_hip(&gl, &t_data[c], &d);
                transform_gravity_hip(&gl, &t_data[c + 1], &d);
                transform_average_hip(&gl, &t_data[c + 2], &d);
            } else {
                transform_fga_hip(&gl, &t_data[c], &d);
            }
        }
    for (int t = 0; t < 6; ++t) {                       /* rd_destroy: render.c:2463-2469 */
        if (mode != 0) glv_hip_release(t_data[t]);
        free(t_data[t]);
    }
    return 0;
}
