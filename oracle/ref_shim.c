/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Thin C wrapper around the *unmodified* reference sources where they lie under
 * /root/reference (nothing is copied into this repository).  `glava/render.c` is
 * unity-included so that the private `struct gl_data` / `struct gl_sampler_data`
 * (render.c:106-118, 166-207) and the non-static operators
 *     transform_fft      render.c:783-847
 *     transform_gravity  render.c:720-736
 *     transform_average  render.c:738-771
 *     transform_wrange   render.c:773-781
 *     transform_smooth   render.c:694-718
 * are visible.  `glava/fifo.c` is compiled as its own TU (see Makefile) and its
 * `entry` (fifo.c:29-127) is reached through the `audio_impls[]` registry exactly as
 * glava.c:469-479 does.
 *
 * Output: oracle/_ref/libglvref.so (git-ignored, travels to the GPU box prebuilt).
 * The build is skipped when /root/reference is absent.
 */
#include "/root/reference/glava/render.c"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

/* Symbols that rd_new()/rd_update() reference but the audio transforms never call. */
struct gl_wcb wcb_glx;
void xwin_assign_icon_bmp(struct gl_wcb* a, void* b, const char* c) { (void) a; (void) b; (void) c; }
unsigned int xwin_copyglbg(struct glava_renderer* r, unsigned int t) { (void) r; (void) t; return 0; }
bool xwin_should_render(struct gl_wcb* a, void* b) { (void) a; (void) b; return true; }
void xwin_wait_for_wm(void) {}

typedef struct {
    float fft_scale, fft_cutoff, gravity_step, ur, smooth_distance, smooth_ratio;
    unsigned long avg_frames;
    int avg_window;
} glvref_params;

static void fill(struct gl_data* gl, const glvref_params* p) {
    memset(gl, 0, sizeof(*gl));
    gl->fft_scale       = p->fft_scale;
    gl->fft_cutoff      = p->fft_cutoff;
    gl->gravity_step    = p->gravity_step;
    gl->ur              = p->ur;
    gl->avg_frames      = p->avg_frames;
    gl->avg_window      = p->avg_window != 0;
    gl->smooth_distance = p->smooth_distance;
    gl->smooth_ratio    = p->smooth_ratio;
}

/* One persistent state slot per (channel, transform), calloc'd by the callee
 * (ALLOC_ONCE, render.c:662-666); pass the address of a NULL-initialised pointer. */
void glvref_fft(const glvref_params* p, float* buf, size_t n) {
    struct gl_data gl; fill(&gl, p);
    struct gl_sampler_data d = { .buf = buf, .sz = n };
    void* slot = NULL;
    transform_fft(&gl, &slot, &d);
}
void glvref_gravity(const glvref_params* p, void** slot, float* buf, size_t n) {
    struct gl_data gl; fill(&gl, p);
    struct gl_sampler_data d = { .buf = buf, .sz = n };
    transform_gravity(&gl, slot, &d);
}
void glvref_average(const glvref_params* p, void** slot, float* buf, size_t n) {
    struct gl_data gl; fill(&gl, p);
    struct gl_sampler_data d = { .buf = buf, .sz = n };
    transform_average(&gl, slot, &d);
}
void glvref_wrange(const glvref_params* p, float* buf, size_t n) {
    struct gl_data gl; fill(&gl, p);
    struct gl_sampler_data d = { .buf = buf, .sz = n };
    void* slot = NULL;
    transform_wrange(&gl, &slot, &d);
}
void glvref_smooth(const glvref_params* p, float* buf, size_t n) {
    struct gl_data gl; fill(&gl, p);
    struct gl_sampler_data d = { .buf = buf, .sz = n };
    void* slot = NULL;
    transform_smooth(&gl, &slot, &d);
}
void glvref_slot_free(void** slot) { free(*slot); *slot = NULL; }

/* ---- FIFO backend driven through the plugin registry (fifo.h:22-44) ------------------ */
#include "/root/reference/glava/fifo.h"

/* Feeds `chunks` updates of `ssz` bytes each through a real named pipe into the
 * reference's fifo `entry` thread and snapshots both rings after every *data* update.
 * rings_out: [chunks][2][fsz] floats (L then R).  Poll-timeout zero-fills
 * (fifo.c:67-79) that slip in while we are not writing are reported in *zero_fills
 * and also applied by the caller's model.  Returns 0 on success. */
int glvref_fifo_run(const char* fifo_path, const int16_t* pcm, size_t chunks, size_t ssz,
                    size_t fsz, int channels, float* rings_out, unsigned char* was_zero_fill,
                    size_t max_events, size_t* n_events) {
    struct audio_impl* impl = NULL;
    for (size_t t = 0; t < audio_impls_idx; ++t)
        if (!strcmp(audio_impls[t]->name, "fifo")) impl = audio_impls[t];
    if (!impl) return -1;

    unlink(fifo_path);
    if (mkfifo(fifo_path, 0600) != 0) return -2;

    float* bl = calloc(fsz, sizeof(float));
    float* br = calloc(fsz, sizeof(float));
    struct audio_data audio = {
        .audio_out_r = br, .audio_out_l = bl, .modified = false,
        .audio_buf_sz = fsz, .sample_sz = ssz, .format = -1, .rate = 22050,
        .source = strdup(fifo_path), .channels = channels, .terminate = 0,
        .mutex = PTHREAD_MUTEX_INITIALIZER
    };
    pthread_t thr;
    pthread_create(&thr, NULL, impl->entry, &audio);
    int wfd = open(fifo_path, O_WRONLY);   /* blocks until the reader opened it */
    if (wfd < 0) return -3;

    size_t ev = 0, sent = 0;
    int rc = 0;
    while (sent < chunks && ev < max_events) {
        if (write(wfd, (const char*) pcm + sent * ssz, ssz) != (ssize_t) ssz) { rc = -4; break; }
        /* wait for this chunk to land; record any zero-fill that happened first */
        bool landed = false;
        while (!landed && ev < max_events) {
            pthread_mutex_lock(&audio.mutex);
            if (audio.modified) {
                audio.modified = false;
                /* a data update ends in our samples; a zero fill ends in ssz/4 zeros.
                   Distinguish by asking whether the pipe is drained: the reader sets
                   `modified` under the same mutex right after consuming the bytes. */
                bool all_zero = true;
                for (size_t q = fsz - ssz / 4; q < fsz; ++q)
                    if (bl[q] != 0.0f || br[q] != 0.0f) { all_zero = false; break; }
                bool input_zero = true;
                for (size_t q = 0; q < ssz / 2; ++q)
                    if (pcm[sent * (ssz / 2) + q] != 0) { input_zero = false; break; }
                bool zf = all_zero && !input_zero;
                memcpy(rings_out + (ev * 2 + 0) * fsz, bl, fsz * sizeof(float));
                memcpy(rings_out + (ev * 2 + 1) * fsz, br, fsz * sizeof(float));
                was_zero_fill[ev] = zf;
                ++ev;
                if (!zf) landed = true;
            }
            pthread_mutex_unlock(&audio.mutex);
            if (!landed) usleep(200);
        }
        ++sent;
    }
    /* the thread notices `terminate` after its next event; with nothing written that
       event is the poll timeout (fifo.c:63-79,119-122), a few ms away */
    audio.terminate = 1;
    pthread_join(thr, NULL);
    close(wfd);
    unlink(fifo_path);
    free(audio.source); free(bl); free(br);
    *n_events = ev;
    return rc;
}

/* ---- CPU baseline ("reference" kind in bench.py) -----------------------------------------
 * `frames` stereo frames of interleaved s16 PCM ([frames][n][2]) through the reference's own
 * transform_fft (both channels), optionally followed by transform_gravity + transform_average
 * with per-channel slots that persist across the frames of the call (one stream's history).
 * The unpack loop is the 2-line body of fifo.c:105-106 (the surrounding `entry` is a static
 * thread function bound to a file descriptor, exercised separately by glvref_fifo_run).
 * Returns a checksum so the work cannot be optimised away. */
double glvref_bench_frames(const glvref_params* p, const int16_t* pcm, size_t frames, size_t n, int with_state) {
    struct gl_data gl; fill(&gl, p);
    float* l = malloc(sizeof(float) * n);
    float* r = malloc(sizeof(float) * n);
    void* gs[2] = { NULL, NULL };
    void* as[2] = { NULL, NULL };
    double acc = 0;
    for (size_t f = 0; f < frames; ++f) {
        const int16_t* buf = pcm + f * 2 * n;
        for (size_t q = 0, i = 0; q < 2 * n; q += 2, ++i) {
            l[i] = buf[q] / (float) 65535;
            r[i] = buf[q + 1] / (float) 65535;
        }
        struct gl_sampler_data dl = { .buf = l, .sz = n }, dr = { .buf = r, .sz = n };
        transform_fft(&gl, NULL, &dl);
        transform_fft(&gl, NULL, &dr);
        if (with_state) {
            transform_gravity(&gl, &gs[0], &dl); transform_average(&gl, &as[0], &dl);
            transform_gravity(&gl, &gs[1], &dr); transform_average(&gl, &as[1], &dr);
        }
        acc += l[1] + r[1];
    }
    free(l); free(r); free(gs[0]); free(gs[1]); free(as[0]); free(as[1]);
    return acc;
}

/* The same loop on `threads` native pthreads until `seconds` of wall clock have passed (CLOCK_MONOTONIC), every
 * thread on its own scratch buffers over the shared read-only PCM: the embarrassingly parallel bound of SURVEY.md 8d
 * (the reference itself is single-threaded per stream).  frames_done[t] receives what thread t finished; returns the
 * elapsed seconds of the slowest thread (<0 on error). */
#include <time.h>
typedef struct {
    const glvref_params* p; const int16_t* pcm; size_t frames, n; int with_state;
    volatile double deadline; double t_end; unsigned long long done; double sink;
} glvref_mt_arg;
static double glvref_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void* glvref_mt_worker(void* v) {
    glvref_mt_arg* a = v;
    do {
        a->sink += glvref_bench_frames(a->p, a->pcm, a->frames, a->n, a->with_state);
        a->done += a->frames;
    } while (glvref_now() < a->deadline);
    a->t_end = glvref_now();
    return NULL;
}
double glvref_bench_mt(const glvref_params* p, const int16_t* pcm, size_t frames, size_t n, int with_state,
                       int threads, double seconds, unsigned long long* frames_done) {
    if (threads < 1 || threads > 4096) return -1.0;
    pthread_t* th = calloc(threads, sizeof(*th));
    glvref_mt_arg* arg = calloc(threads, sizeof(*arg));
    if (!th || !arg) { free(th); free(arg); return -3.0; }
    const double t0 = glvref_now();
    for (int t = 0; t < threads; ++t) {
        arg[t] = (glvref_mt_arg){ .p = p, .pcm = pcm, .frames = frames, .n = n, .with_state = with_state, .deadline = t0 + seconds };
        if (pthread_create(&th[t], NULL, glvref_mt_worker, &arg[t]) != 0) {
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
