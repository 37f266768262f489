/*
 * integration/hipfifo.c -- an `audio_impl` backend for GLava's plugin seam (glava/fifo.h:22-44), as real
 * code: INTEGRATION.md section 2.
 *
 * Same source as the reference's "fifo" backend -- `sample_sz` bytes of interleaved s16 per update from a
 * named pipe, poll timeout => an update of zeros (fifo.c:63-79) -- but the N-sample rings live on the MI355X.
 *
 * What it publishes in audio_out_l / audio_out_r under the mutex:
 *
 *   default        the time-domain sample rings, exactly what struct audio_data defines (fifo.h:9-20) and what
 *                  fifo.c leaves there after its memmove + unpack (fifo.c:91-110): the device ring is appended to
 *                  (glv_batch_ring_append_s16) and read back in publishing order (glv_batch_ring_planar: planar f32,
 *                  oldest sample first, the s16 -> f32 unpack done on the device).  Every stock module renders from
 *                  it as from the stock backend; with integration/render_hip.patch the transforms of rd_update then
 *                  run on the MI355X through the operator shim.
 *
 *   GLAVA_HIPFIFO_SPECTRA=1 (opt-in; or glv_hipfifo_spectra = 1 before the thread starts)
 *                  the finished SPECTRA of the device-resident rings (window, FFT, magnitude of render.c:783-847 in one
 *                  launch: glv_batch_ring_update_s16), one transform per update instead of upload + transform per
 *                  bind.  This changes the meaning of audio_out_l / audio_out_r, so the backend raises
 *                  glv_audio_publishes_spectra, which the patched handle_audio reads to skip the "fft" transform of
 *                  every bind (gravity / average still run; integration/render_hip.patch).  Not for modules that bind
 *                  the raw waveform, nor for setbufscale / setinterpolate (the rd_update prelude would then decimate /
 *                  interpolate spectra).
 *
 * Everything device-side goes through the C ABI of include/glv_spectrum.h; no HIP headers here.
 *
 * Meant to be compiled next to glava/fifo.c (it self-registers with AUDIO_ATTACH like every backend and is
 * selected with --audio=hipfifo, glava.c:469-479).  integration/shim_harness.c builds it against the
 * unmodified reference headers; tests/test_gpu_parity.py::test_hipfifo_backend_through_the_registry runs it.
 */
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <glv_spectrum.h>

#include "fifo.h"

/* observable by tests: how many updates were poll-timeout zero fills (timing dependent), and the poll timeout
 * the backend is currently using (adapted to the producer's cadence exactly like fifo.c:82-87) */
volatile unsigned long glv_hipfifo_zero_fills = 0;
volatile int glv_hipfifo_timeout_ms = 50;
/* knobs the host would take from its config (rc.glsl): magnitude parameters of the transform (spectra mode) */
float glv_hipfifo_fft_scale = 10.2f, glv_hipfifo_fft_cutoff = 0.3f;
/* what is published: 0 (default) the sample rings struct audio_data defines, 1 finished spectra; -1 = take it from the
 * environment (GLAVA_HIPFIFO_SPECTRA) when the thread starts */
volatile int glv_hipfifo_spectra __attribute__((weak)) = -1;     /* weak: integration/hippulse.c carries the same definition */
/* read by the patched handle_audio (integration/render_hip.patch): non-zero while a backend publishes spectra */
volatile int glv_audio_publishes_spectra __attribute__((weak)) = 0;

static void glv_hipfifo_die(const char* what) {
    fprintf(stderr, "hipfifo backend: %s: %s\n", what, glv_last_error());
    exit(EXIT_FAILURE);                       /* the reference backends' error convention (fifo.c:47-50) */
}

static void init(struct audio_data* audio) {
    if (!audio->source) audio->source = strdup("/tmp/mpd.fifo");      /* fifo.c:23-27 */
}

static void* entry(void* data) {
    struct audio_data* audio = data;
    const size_t n = audio->audio_buf_sz, ssz = audio->sample_sz;
    const uint32_t frames = (uint32_t) (ssz / 4);                      /* stereo frames per update */
    int spectra = glv_hipfifo_spectra;
    if (spectra < 0) { const char* e = getenv("GLAVA_HIPFIFO_SPECTRA"); spectra = e && e[0] == '1'; }

    glv_params p;
    glv_params_default(&p);
    p.n = (uint32_t) n;
    p.channels = (uint32_t) audio->channels;                           /* 1 = mirror/mono mix, fifo.c:98-102 */
    p.fft_scale = glv_hipfifo_fft_scale;
    p.fft_cutoff = glv_hipfifo_fft_cutoff;
    glv_batch* batch = NULL;
    void *d_new = NULL, *d_pub = NULL;
    if (glv_batch_create(&p, 1, GLV_OP_FFT | GLV_OP_RING_S16, 0, &batch) != GLV_OK) glv_hipfifo_die("glv_batch_create");
    if (glv_device_malloc(0, ssz, &d_new) != GLV_OK || glv_device_malloc(0, 2 * n * sizeof(float), &d_pub) != GLV_OK)
        glv_hipfifo_die("glv_device_malloc");
    glv_audio_publishes_spectra = spectra;

    int fd = open(audio->source, O_RDONLY);
    if (fd == -1) {
        fprintf(stderr, "hipfifo backend: cannot open \"%s\": %s\n", audio->source, strerror(errno));
        exit(EXIT_FAILURE);
    }
    struct pollfd pfd = { .fd = fd, .events = POLLIN };
    int16_t* buf = malloc(ssz);
    float* pub = malloc(2 * n * sizeof(float));
    if (!buf || !pub) { fprintf(stderr, "hipfifo backend: out of memory\n"); exit(EXIT_FAILURE); }
    int timeout_ms = 50;                                               /* initial value of fifo.c:39 */
    struct timespec tv_last = { 0, 0 }, tv;
    int measured = 0;

    for (;;) {
        int rc, ready = poll(&pfd, 1, timeout_ms);
        const void* d_update = NULL;                                   /* NULL: nothing arrived, an update of zeros (fifo.c:67-79) */
        if (ready < 0) { fprintf(stderr, "hipfifo backend: poll: %s\n", strerror(errno)); exit(EXIT_FAILURE); }
        rc = GLV_OK;
        if (ready > 0) {
            size_t have = 0;                                           /* a full update, like read(fd, buf, ssz) */
            while (have < ssz) {
                ssize_t r = read(fd, (char*) buf + have, ssz - have);
                if (r <= 0) break;
                have += (size_t) r;
            }
            if (have < ssz) memset((char*) buf + have, 0, ssz - have);
            /* fifo.c:82-87: "Set the timeout slightly higher than the delay between samples to prevent empty
               writes" -- the delay between this read and the previous one, in whole milliseconds, plus one */
            clock_gettime(CLOCK_REALTIME, measured ? &tv : &tv_last);
            if (measured) {
                timeout_ms = (int) (((tv.tv_sec - tv_last.tv_sec) * 1000) + ((tv.tv_nsec - tv_last.tv_nsec) / 1000000)) + 1;
                tv_last = tv;
                glv_hipfifo_timeout_ms = timeout_ms;
            } else measured = 1;
            rc = glv_device_upload(0, d_new, buf, ssz, NULL);
            d_update = d_new;
        }
        if (rc == GLV_OK) {
            if (spectra) rc = glv_batch_ring_update_s16(batch, d_update, frames, d_pub, GLV_OP_FFT, NULL);
            else {
                rc = glv_batch_ring_append_s16(batch, d_update, frames, NULL);
                if (rc == GLV_OK) rc = glv_batch_ring_planar(batch, 0, d_pub, NULL);
            }
        }
        if (rc == GLV_OK) rc = glv_device_download(0, pub, d_pub, 2 * n * sizeof(float), NULL);
        if (rc == GLV_OK) rc = glv_device_sync(0, NULL);
        if (rc != GLV_OK) glv_hipfifo_die("update");

        pthread_mutex_lock(&audio->mutex);
        memcpy((void*) audio->audio_out_l, pub, n * sizeof(float));
        memcpy((void*) audio->audio_out_r, pub + n, n * sizeof(float));
        if (ready == 0) ++glv_hipfifo_zero_fills;
        audio->modified = true;
        pthread_mutex_unlock(&audio->mutex);

        if (audio->terminate == 1) break;                              /* fifo.c:119-122 */
    }
    glv_audio_publishes_spectra = 0;
    close(fd);
    free(buf); free(pub);
    glv_device_free(0, d_new); glv_device_free(0, d_pub);
    glv_batch_destroy(batch);
    return NULL;
}

AUDIO_ATTACH(hipfifo);
