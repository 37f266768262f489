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

/* what glv_hip_scan_shape made of the last pre-smoothing shader text it was given (tests/test_smooth_shape.py); returns `readable` */
int shim_shape(unsigned* formula, unsigned* mode, float* hybrid, float* scale, float* range) {
    *formula = glv_hip_shape.formula; *mode = glv_hip_shape.mode;
    *hybrid = glv_hip_shape.hybrid != 0.0f ? glv_hip_shape.hybrid : 0.65f;
    *scale = glv_hip_shape.scale != 0.0f ? glv_hip_shape.scale : 8.0f;
    *range = glv_hip_shape.range != 0.0f ? glv_hip_shape.range : 0.9f;
    return glv_hip_shape.readable ? 1 : 0;
}

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

/* ---- audio backend seam (glava/fifo.h): run a registered backend against a named pipe -------------------------
 * Like glava.c:469-520: look the backend up by name in audio_impls[], start its `entry` on a thread, then feed
 * `chunks` updates of `ssz` bytes and snapshot audio_out_l/r (fsz floats each) after every update the backend
 * publishes.  zero_fill[e] = 1 when update e was a poll-timeout update of zeros (timing dependent; the test
 * replays whatever happened).  Works for the reference's "fifo" backend (snapshots are the sample rings) and for
 * "hipfifo" (the same rings by default; spectra after glvshim_hipfifo_publish_spectra(1)).  Returns the number of events,
 * or a negative error. */
#include <sys/stat.h>
#include "fifo.h"
extern volatile unsigned long glv_hipfifo_zero_fills;
extern volatile int glv_hipfifo_spectra;
/* what the "hipfifo" backend publishes from its next start on: 0 the sample rings (its default), 1 finished spectra */
void glvshim_hipfifo_publish_spectra(int on) { glv_hipfifo_spectra = on ? 1 : 0; }

/* pcm: the bytes to feed, `chunks` updates of `ssz` bytes (s16 backends: sample_sz bytes of interleaved s16) or of 2 * ssz
 * bytes (name "hippulse": sample_sz / 4 interleaved stereo f32 frames, what pa_simple_read delivers per update) */
long glvshim_backend_run(const char* name, const char* fifo_path, const int16_t* pcm, size_t chunks, size_t ssz, size_t fsz,
                         int channels, float* snapshots /* [max_events][2][fsz] */, unsigned char* zero_fill, size_t max_events) {
    struct audio_impl* impl = NULL;
    for (size_t t = 0; t < audio_impls_idx; ++t)
        if (!strcmp(audio_impls[t]->name, name)) impl = audio_impls[t];
    if (!impl) return -1;
    const bool hip = !strcmp(name, "hipfifo");
    const bool f32 = !strcmp(name, "hippulse");
    const size_t wsz = f32 ? 2 * ssz : ssz;                /* bytes per update on the wire */
    unlink(fifo_path);
    if (mkfifo(fifo_path, 0600) != 0) return -2;
    float* bl = calloc(fsz, sizeof(float));
    float* br = calloc(fsz, sizeof(float));
    struct audio_data audio = {
        .audio_out_r = br, .audio_out_l = bl, .modified = false, .audio_buf_sz = fsz, .sample_sz = ssz,
        .format = -1, .rate = 22050, .source = strdup(fifo_path), .channels = channels, .terminate = 0,
        .mutex = PTHREAD_MUTEX_INITIALIZER
    };
    pthread_t thr;
    pthread_create(&thr, NULL, impl->entry, &audio);
    int wfd = open(fifo_path, O_WRONLY);                  /* blocks until the backend opened its end */
    if (wfd < 0) return -3;
    size_t ev = 0;
    unsigned long zf_seen = hip ? glv_hipfifo_zero_fills : 0;
    long rc = 0;
    for (size_t sent = 0; sent < chunks && ev < max_events; ++sent) {
        if (write(wfd, (const char*) pcm + sent * wsz, wsz) != (ssize_t) wsz) { rc = -4; break; }
        bool landed = false;
        while (!landed && ev < max_events) {
            pthread_mutex_lock(&audio.mutex);
            if (audio.modified) {
                audio.modified = false;
                bool zf;
                if (hip) { zf = glv_hipfifo_zero_fills != zf_seen; zf_seen = glv_hipfifo_zero_fills; }
                else if (f32) zf = false;                   /* a blocking read: no poll-timeout updates */
                else {
                    bool tail_zero = true, input_zero = true;
                    for (size_t q = fsz - ssz / 4; q < fsz; ++q) if (bl[q] != 0.0f || br[q] != 0.0f) { tail_zero = false; break; }
                    for (size_t q = 0; q < ssz / 2; ++q) if (pcm[sent * (ssz / 2) + q] != 0) { input_zero = false; break; }
                    zf = tail_zero && !input_zero;
                }
                memcpy(snapshots + (ev * 2 + 0) * fsz, bl, fsz * sizeof(float));
                memcpy(snapshots + (ev * 2 + 1) * fsz, br, fsz * sizeof(float));
                zero_fill[ev++] = zf;
                if (!zf) landed = true;
            }
            pthread_mutex_unlock(&audio.mutex);
            if (!landed) usleep(200);
        }
    }
    audio.terminate = 1;                                   /* noticed after the backend's next event (a timeout) */
    if (f32) { close(wfd); wfd = -1; }                     /* a blocking reader ends when the producer goes away */
    pthread_join(thr, NULL);
    if (wfd >= 0) close(wfd);
    unlink(fifo_path);
    free(bl); free(br); free(audio.source);
    return rc < 0 ? rc : (long) ev;
}
