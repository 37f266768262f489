/*
 * integration/hippulse.c -- the PulseAudio twin of integration/hipfifo.c: an `audio_impl` backend (glava/fifo.h:22-44)
 * with the ring / deinterleave half of the reference's "pulseaudio" backend (glava/pulse_input.c:108-190) on the MI355X.
 *
 * Per update the reference reads sample_sz / 4 interleaved stereo f32 frames (pa_simple_read of 2 * sample_sz bytes,
 * pulse_input.c:112,147), shifts both rings left by that many samples (:155-156) and appends the deinterleaved frames --
 * (L + R) / 2 into both when channels == 1 (:159-176).  Here the ring lives on the device as interleaved f32 frames
 * (glv_batch_ring_append_f32) and what is published in audio_out_l / audio_out_r is read back in publishing order
 * (glv_batch_ring_planar): exactly what struct audio_data defines (fifo.h:9-20), so every stock module and the patched
 * handle_audio (integration/render_hip.patch) consume it unchanged.  GLAVA_HIPFIFO_SPECTRA=1 switches to finished spectra
 * like hipfifo does (glv_batch_ring_update_f32; raises glv_audio_publishes_spectra).
 *
 * Capture itself is out of scope and libpulse is not part of this build environment, so the
 * capture call is the one line that differs between the two builds of this file:
 *     -DGLV_HAVE_PULSE   pa_simple_new / pa_simple_read exactly as pulse_input.c:113-150 (link -lpulse-simple);
 *     otherwise          the same bytes -- interleaved stereo f32, 2 * sample_sz per update -- read from the named pipe
 *                        audio->source (e.g. `parec --format=float32le --channels=2 > pipe`), which is also how
 *                        tests/test_gpu_parity.py::test_hippulse_backend_through_the_registry feeds it.
 */
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <glv_spectrum.h>

#include "fifo.h"

#ifdef GLV_HAVE_PULSE
#include <pulse/simple.h>
#include <pulse/error.h>
#endif

/* shared with integration/hipfifo.c: what the hip backends publish (-1: GLAVA_HIPFIFO_SPECTRA decides) and the flag the patched
 * handle_audio reads.  Weak definitions in BOTH backends, so that either links alone and both link together */
volatile int glv_hipfifo_spectra __attribute__((weak)) = -1;
volatile int glv_audio_publishes_spectra __attribute__((weak)) = 0;

static void glv_hippulse_die(const char* what) {
    fprintf(stderr, "hippulse backend: %s: %s\n", what, glv_last_error());
    exit(EXIT_FAILURE);
}

static void init(struct audio_data* audio) {
#ifndef GLV_HAVE_PULSE
    if (!audio->source) audio->source = strdup("/tmp/glava_pulse.f32");
#else
    (void) audio;                                       /* NULL = the server's default source (the reference looks the sink's monitor up, pulse_input.c:16-97) */
#endif
}

static void* entry(void* data) {
    struct audio_data* audio = data;
    const size_t n = audio->audio_buf_sz, ssz = audio->sample_sz;
    const uint32_t frames = (uint32_t) (ssz / 4);                      /* stereo frames per update (pulse_input.c:155-160) */
    const size_t bytes = (size_t) frames * 2 * sizeof(float);          /* == sizeof(float buf[ssz / 2]), pulse_input.c:112 */
    int spectra = glv_hipfifo_spectra;
    if (spectra < 0) { const char* e = getenv("GLAVA_HIPFIFO_SPECTRA"); spectra = e && e[0] == '1'; }

    glv_params p;
    glv_params_default(&p);
    p.n = (uint32_t) n;
    p.channels = (uint32_t) audio->channels;                           /* 1 = (L + R) / 2 into both rings, pulse_input.c:166-170 */
    glv_batch* batch = NULL;
    void *d_new = NULL, *d_pub = NULL;
    if (glv_batch_create(&p, 1, GLV_OP_FFT | GLV_OP_RING_F32, 0, &batch) != GLV_OK) glv_hippulse_die("glv_batch_create");
    if (glv_device_malloc(0, bytes, &d_new) != GLV_OK || glv_device_malloc(0, 2 * n * sizeof(float), &d_pub) != GLV_OK)
        glv_hippulse_die("glv_device_malloc");
    glv_audio_publishes_spectra = spectra;
    float* buf = malloc(bytes);
    float* pub = malloc(2 * n * sizeof(float));
    if (!buf || !pub) { fprintf(stderr, "hippulse backend: out of memory\n"); exit(EXIT_FAILURE); }

#ifdef GLV_HAVE_PULSE
    const pa_sample_spec ss = { .format = PA_SAMPLE_FLOAT32LE, .rate = audio->rate, .channels = 2 };
    const pa_buffer_attr pb = { .maxlength = (uint32_t) -1, .fragsize = (uint32_t) ssz };
    int error;
    pa_simple* s = pa_simple_new(NULL, "glava", PA_STREAM_RECORD, audio->source, "audio for glava", &ss, NULL, &pb, &error);
    if (!s) { fprintf(stderr, "hippulse backend: could not open pulseaudio source %s: %s\n", audio->source, pa_strerror(error)); exit(EXIT_FAILURE); }
#else
    int fd = open(audio->source, O_RDONLY);
    if (fd == -1) { fprintf(stderr, "hippulse backend: cannot open \"%s\": %s\n", audio->source, strerror(errno)); exit(EXIT_FAILURE); }
#endif

    for (;;) {
#ifdef GLV_HAVE_PULSE
        if (pa_simple_read(s, buf, bytes, &error) < 0) { fprintf(stderr, "hippulse backend: pa_simple_read() failed: %s\n", pa_strerror(error)); exit(EXIT_FAILURE); }
#else
        size_t have = 0;                                               /* a blocking read of one full update, like pa_simple_read */
        while (have < bytes) {
            ssize_t r = read(fd, (char*) buf + have, bytes - have);
            if (r <= 0) break;
            have += (size_t) r;
        }
        if (have < bytes) break;                                       /* the producer closed the pipe */
#endif
        int rc = glv_device_upload(0, d_new, buf, bytes, NULL);
        if (rc == GLV_OK) {
            if (spectra) rc = glv_batch_ring_update_f32(batch, d_new, frames, d_pub, GLV_OP_FFT, NULL);
            else {
                rc = glv_batch_ring_append_f32(batch, d_new, frames, NULL);
                if (rc == GLV_OK) rc = glv_batch_ring_planar(batch, 1, d_pub, NULL);
            }
        }
        if (rc == GLV_OK) rc = glv_device_download(0, pub, d_pub, 2 * n * sizeof(float), NULL);
        if (rc == GLV_OK) rc = glv_device_sync(0, NULL);
        if (rc != GLV_OK) glv_hippulse_die("update");

        pthread_mutex_lock(&audio->mutex);
        memcpy((void*) audio->audio_out_l, pub, n * sizeof(float));
        memcpy((void*) audio->audio_out_r, pub + n, n * sizeof(float));
        audio->modified = true;
        pthread_mutex_unlock(&audio->mutex);

        if (audio->terminate == 1) break;                              /* pulse_input.c:183-186 */
    }
    glv_audio_publishes_spectra = 0;
#ifdef GLV_HAVE_PULSE
    pa_simple_free(s);
#else
    close(fd);
#endif
    free(buf); free(pub);
    glv_device_free(0, d_new); glv_device_free(0, d_pub);
    glv_batch_destroy(batch);
    return NULL;
}

AUDIO_ATTACH(hippulse);
