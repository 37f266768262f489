"""ctypes bindings for the two CPU checkers under oracle/ (test infrastructure only).

* ``Oracle``  -> oracle/liboracle.so   (C restatement, oracle/glv_oracle.c)
* ``Ref``     -> oracle/_ref/libglvref.so (the reference's own C compiled from
  /root/reference by oracle/Makefile; absent => ``Ref.available()`` is False)

Nothing in glava_amd/ imports this module.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build_oracles() -> None:
    """(Re)build liboracle.so and, when /root/reference exists, _ref/libglvref.so."""
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def lcg_pcm(seed: int, count: int) -> np.ndarray:
    """SURVEY.md 8c generator: s = s*1664525 + 1013904223 (mod 2^32); v = int16(s >> 16)."""
    out = np.empty(count, dtype=np.int16)
    s = seed & 0xFFFFFFFF
    # vectorised closed form is not worth it; chunked python loop is fine for test sizes
    a, c = 1664525, 1013904223
    buf = np.empty(count, dtype=np.uint32)
    for i in range(count):
        s = (s * a + c) & 0xFFFFFFFF
        buf[i] = s
    out[:] = (buf >> 16).astype(np.uint16).view(np.int16)
    return out


def lcg_pcm_fast(seed: int, count: int) -> np.ndarray:
    """Same sequence as lcg_pcm, vectorised by jumping the LCG (a^k, c_k) per block."""
    a, c, m = 1664525, 1013904223, 1 << 32
    block = 1 << 12
    # first block sequentially
    n0 = min(block, count)
    s = seed & 0xFFFFFFFF
    first = np.empty(n0, dtype=np.uint64)
    for i in range(n0):
        s = (s * a + c) % m
        first[i] = s
    if count <= block:
        return (first.astype(np.uint32) >> 16).astype(np.uint16).view(np.int16)
    # jump-ahead constants for stride `block`
    A, Cc = 1, 0
    for _ in range(block):
        A, Cc = (A * a) % m, (Cc * a + c) % m
    nblocks = (count + block - 1) // block
    out = np.empty((nblocks, block), dtype=np.uint64)
    out[0] = first
    for b in range(1, nblocks):
        out[b] = (out[b - 1] * A + Cc) % m
    flat = out.reshape(-1)[:count].astype(np.uint32)
    return (flat >> 16).astype(np.uint16).view(np.int16)


class Oracle:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            path = os.path.join(ORACLE_DIR, "liboracle.so")
            if not os.path.exists(path):
                build_oracles()
            L = C.CDLL(path)
            L.glvo_unpack_s16.argtypes = [_i16p, C.c_size_t, C.c_int, _f32p, _f32p]
            L.glvo_unpack_f32.argtypes = [_f32p, C.c_size_t, C.c_int, _f32p, _f32p]
            L.glvo_ring_update_s16.argtypes = [_f32p, _f32p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
            L.glvo_window_table.argtypes = [_f64p, C.c_size_t]
            L.glvo_apply_window.argtypes = [_f32p, C.c_size_t]
            L.glvo_twiddles.argtypes = [_f32p, C.c_size_t]
            L.glvo_fft_core.argtypes = [_f32p, C.c_size_t]
            L.glvo_magnitude.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float]
            L.glvo_transform_fft.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float, C.c_void_p]
            L.glvo_gravity.argtypes = [_f32p, _f32p, C.c_size_t, C.c_float, C.c_float]
            L.glvo_average.argtypes = [_f32p, _f32p, C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_int]
            L.glvo_frame_weight.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
            L.glvo_frame_weight.restype = C.c_double
            L.glvo_wrange.argtypes = [_f32p, C.c_size_t]
            L.glvo_frame_s16.argtypes = [_i16p, C.c_size_t, C.c_int, C.c_float, C.c_float, _f32p, C.c_void_p,
                                         C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            L.glvo_bench_frames.argtypes = [_i16p, C.c_size_t, C.c_size_t, C.c_float, C.c_float]
            L.glvo_bench_frames.restype = C.c_double
            L.glvo_bench_mt.argtypes = [_i16p, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_double, C.POINTER(C.c_ulonglong)]
            L.glvo_bench_mt.restype = C.c_double
            L.glvo_bufscale.argtypes = [_f32p, _f32p, C.c_size_t, C.c_size_t]
            L.glvo_lerp.argtypes = [_f32p, _f32p, _f32p, C.c_size_t, C.c_float, C.c_int]
            L.glvo_smooth.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float]
            L.glvo_average_gl.argtypes = [_f32p, _f32p, C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_int]
            L.glvo_bars.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float]
            L.glvo_bars_chunked.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float]
            L.glvo_bars_at.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, C.c_float]
            L.glvo_bars_chunked_at.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, C.c_float]
            L.glvo_bars_mode_at.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_float]
            L.glvo_set_smooth_shape.argtypes = [C.c_int, C.c_float, C.c_float]
            L.glvo_set_smooth_mode.argtypes = [C.c_int, C.c_float]
            _f64s = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
            L.glvo_bars_weight_slack.argtypes = [C.c_size_t, _f64s, _f64s, C.c_size_t, C.c_float, C.c_float]
            _u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
            L.glvo_bars_int_at.argtypes = [_u16p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float]
            L.glvo_bars_int_at.restype = C.c_int
            L.glvo_bars_one_exact.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_double), C.POINTER(C.c_int)]
            L.glvo_texels_r16.argtypes = [_f32p, C.c_size_t, np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")]
            L.glvo_bars_at_exact.argtypes = [_f32p, C.c_size_t, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"),
                                             np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"), np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"),
                                             C.c_size_t, C.c_float, C.c_float, C.c_int]
            _f64q = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
            L.glvo_bars_range_exact.argtypes = [_f32p, _f32p, C.c_size_t, _f64q, _f64q, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"),
                                                C.c_size_t, C.c_float, C.c_float, C.c_double]
            L.glvo_gl_chain_r16.argtypes = [_f32p, _f32p, _f32p, C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_float]
            L.glvo_unorm16.argtypes = [C.c_float]; L.glvo_unorm16.restype = C.c_uint16
            L.glvo_unorm16_to_float.argtypes = [C.c_uint16]; L.glvo_unorm16_to_float.restype = C.c_float
            cls._lib = L
        return cls._lib

    # --- convenience wrappers -----------------------------------------------------------
    @classmethod
    def unpack_s16(cls, pcm: np.ndarray, channels: int = 2):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(-1)
        frames = pcm.size // 2
        l = np.empty(frames, np.float32); r = np.empty(frames, np.float32)
        cls.lib().glvo_unpack_s16(pcm, frames, channels, l, r)
        return l, r

    @classmethod
    def texels_r16(cls, buf: np.ndarray) -> np.ndarray:
        """GL_R16 texels of a float buffer (the upload of render.c:521-524)."""
        b = np.ascontiguousarray(buf, dtype=np.float32).reshape(-1)
        out = np.empty(b.size, np.uint16)
        cls.lib().glvo_texels_r16(b, b.size, out)
        return out.reshape(np.shape(buf))

    @classmethod
    def bars_int(cls, texels: np.ndarray, bars: int, smooth_factor=0.025, phase=0.5):
        """glvo_bars_int_at: the exact integer weighted mean of the many-bars pass over one row of GL_R16 texels -> (texels, floats)"""
        texels = np.ascontiguousarray(texels, dtype=np.uint16)
        t = np.zeros(bars, np.uint16); f = np.zeros(bars, np.float32)
        rc = cls.lib().glvo_bars_int_at(texels, texels.size, t.ctypes.data, f.ctypes.data, bars, smooth_factor, phase)
        assert rc == 0, rc
        return t, f

    @classmethod
    def bars_mode(cls, row: np.ndarray, bars: int, mode: int, hybrid_weight=0.65, smooth_factor=0.025, phase=0.0) -> np.ndarray:
        """glvo_bars_mode_at: SAMPLE_MODE maximum (1) / hybrid (2) of one float row, the shader's loop in float"""
        row = np.ascontiguousarray(row, dtype=np.float32)
        out = np.zeros(bars, np.float32)
        cls.lib().glvo_bars_mode_at(row, row.size, out, bars, smooth_factor, phase, mode, hybrid_weight)
        return out

    @classmethod
    @contextlib.contextmanager
    def smooth_shape(cls, formula: int = 0, scale: float = 0.0, rng: float = 0.0, mode: int = 0, hybrid_weight: float = 0.0):
        """glvo_set_smooth_shape for the duration of a `with` block: ROUND_FORMULA (0 sinusoidal, 1 circular, 2 linear), SAMPLE_SCALE, SAMPLE_RANGE
        as every glvo_bars_* function sees them (0 = the shipped 8 / 0.9); mode / hybrid_weight: SAMPLE_MODE for the float64 evaluations
        (glvo_bars_at_exact, _one_exact, _range_exact: glvo_set_smooth_mode)"""
        cls.lib().glvo_set_smooth_shape(formula, scale, rng)
        cls.lib().glvo_set_smooth_mode(mode, hybrid_weight)
        try:
            yield
        finally:
            cls.lib().glvo_set_smooth_shape(0, 0.0, 0.0)
            cls.lib().glvo_set_smooth_mode(0, 0.0)

    @classmethod
    def window_table(cls, n: int) -> np.ndarray:
        w = np.empty(n, np.float64)
        cls.lib().glvo_window_table(w, n)
        return w

    @classmethod
    def twiddles(cls, L: int) -> np.ndarray:
        tw = np.empty(2 * L, np.float32)
        cls.lib().glvo_twiddles(tw, L)
        return tw

    @classmethod
    def transform_fft(cls, data: np.ndarray, fft_scale=10.2, fft_cutoff=0.3, want_raw=False):
        d = np.array(data, dtype=np.float32, copy=True)
        raw = np.empty_like(d) if want_raw else None
        cls.lib().glvo_transform_fft(d, d.size, fft_scale, fft_cutoff,
                                     raw.ctypes.data_as(C.c_void_p) if want_raw else None)
        return (d, raw) if want_raw else d

    @classmethod
    def gravity(cls, b: np.ndarray, state: np.ndarray, gravity_step=4.2, ur=86.1328125):
        cls.lib().glvo_gravity(b, state, b.size, gravity_step, ur)

    @classmethod
    def average(cls, b: np.ndarray, hist: np.ndarray, head: "C.c_size_t", F: int, use_window=True):
        cls.lib().glvo_average(b, hist, C.byref(head), b.size, F, int(use_window))


def log1_rel(n: int) -> float:
    """bound of the default log mode's relative error on EVERY input of the magnitude stage (hardware v_log_f32; exhaustive:
    tests/test_gpu_parity.py::test_magnitude_stage_every_float measures <= 1.8e-7, <= 6.3e-7 from n = 16384 up where the tilt factor is folded)"""
    return 2e-7 if n < 16384 else 7e-7


def chain_close(got, want, peak=None, n=4096, rel=1e-5):
    """north_star's tolerance for the default log mode: 1e-5 RELATIVE on every value.  Behind gravity's subtraction (max(b, applied) - g can
    land anywhere near zero) a relative bound on the difference means nothing; what is bounded there is its ABSOLUTE error: gravity's max()
    is 1-Lipschitz and its subtraction exact up to one rounding, the average's weights are <= 1, so the error of a bin is at most the log's
    relative error times the LARGEST magnitude that bin has held so far -- `peak` (per bin), log1_rel(n) * peak.  (Until round 4 a flat
    atol = 2e-6 stood here)  peak None: a chain without gravity -- purely relative."""
    tol = rel * np.abs(want).astype(np.float64)
    if peak is not None:
        tol = tol + log1_rel(n) * np.asarray(peak, np.float64)
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    fin = np.isfinite(want)
    return bool((d[fin] <= tol[fin]).all() and (np.asarray(got)[~fin] == np.asarray(want)[~fin]).all())


class StreamOracle:
    """Stateful per-stream oracle: PCM frame in -> spectrum out, with gravity/average state.

    Mirrors the order handle_audio applies the operators (render.c:2140-2156).  With gravity it also keeps `peak`, the largest
    magnitude every bin has held (chain_close's absolute term); close(got, want) applies the tolerance."""

    def __init__(self, n, channels=2, fft_scale=10.2, fft_cutoff=0.3, gravity_step=4.2,
                 ur=86.1328125, avg_frames=5, avg_window=True, gravity=True, average=True):
        self.n, self.channels = n, channels
        self.fft_scale, self.fft_cutoff = fft_scale, fft_cutoff
        self.gravity_step, self.ur = gravity_step, ur
        self.F, self.avg_window = avg_frames, avg_window
        self.grav = np.zeros((2, n), np.float32) if gravity else None
        self.hist = np.zeros((2, avg_frames, n), np.float32) if average else None
        self.heads = (C.c_size_t * 2)(0, 0)
        self.peak = np.zeros((2, n), np.float32) if gravity else None

    def close(self, got, want, rel=1e-5) -> bool:
        return chain_close(got, want, self.peak, self.n, rel)

    def frame(self, pcm: np.ndarray, want_raw=False):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(-1)
        assert pcm.size == 2 * self.n
        if self.peak is not None:                            # the magnitudes that enter the chain (fft only, no state)
            mag = np.empty((2, self.n), np.float32)
            Oracle.lib().glvo_frame_s16(pcm, self.n, self.channels, self.fft_scale, self.fft_cutoff, mag, None, None, self.gravity_step, self.ur,
                                        None, None, self.F, int(self.avg_window))
            self.peak = np.maximum(self.peak, np.where(np.isfinite(mag), np.abs(mag), 0).astype(np.float32))
        out = np.empty((2, self.n), np.float32)
        raw = np.empty((2, self.n), np.float32) if want_raw else None
        Oracle.lib().glvo_frame_s16(
            pcm, self.n, self.channels, self.fft_scale, self.fft_cutoff, out,
            raw.ctypes.data_as(C.c_void_p) if want_raw else None,
            self.grav.ctypes.data_as(C.c_void_p) if self.grav is not None else None,
            self.gravity_step, self.ur,
            self.hist.ctypes.data_as(C.c_void_p) if self.hist is not None else None,
            C.cast(self.heads, C.c_void_p), self.F, int(self.avg_window))
        return (out, raw) if want_raw else out


class RefParams(C.Structure):
    _fields_ = [("fft_scale", C.c_float), ("fft_cutoff", C.c_float), ("gravity_step", C.c_float),
                ("ur", C.c_float), ("smooth_distance", C.c_float), ("smooth_ratio", C.c_float),
                ("avg_frames", C.c_ulong), ("avg_window", C.c_int)]


class Ref:
    """The compiled reference (transform_* of glava/render.c, fifo entry of glava/fifo.c)."""
    _lib = None
    PATH = os.path.join(ORACLE_DIR, "_ref", "libglvref.so")

    @classmethod
    def available(cls) -> bool:
        if not os.path.exists(cls.PATH) and os.path.exists("/root/reference/glava/render.c"):
            build_oracles()
        return os.path.exists(cls.PATH)

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(cls.PATH)
            P = C.POINTER(RefParams)
            L.glvref_fft.argtypes = [P, _f32p, C.c_size_t]
            L.glvref_gravity.argtypes = [P, C.POINTER(C.c_void_p), _f32p, C.c_size_t]
            L.glvref_average.argtypes = [P, C.POINTER(C.c_void_p), _f32p, C.c_size_t]
            L.glvref_wrange.argtypes = [P, _f32p, C.c_size_t]
            L.glvref_smooth.argtypes = [P, _f32p, C.c_size_t]
            L.glvref_slot_free.argtypes = [C.POINTER(C.c_void_p)]
            L.glvref_fifo_run.argtypes = [C.c_char_p, _i16p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int,
                                          _f32p, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
            L.glvref_fifo_run.restype = C.c_int
            L.glvref_bench_frames.argtypes = [P, _i16p, C.c_size_t, C.c_size_t, C.c_int]
            L.glvref_bench_frames.restype = C.c_double
            L.glvref_bench_mt.argtypes = [P, _i16p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_ulonglong)]
            L.glvref_bench_mt.restype = C.c_double
            cls._lib = L
        return cls._lib

    @staticmethod
    def params(fft_scale=10.2, fft_cutoff=0.3, gravity_step=4.2, ur=86.1328125, avg_frames=5,
               avg_window=True, smooth_distance=0.01, smooth_ratio=4.0) -> RefParams:
        return RefParams(fft_scale, fft_cutoff, gravity_step, ur, smooth_distance, smooth_ratio,
                         avg_frames, int(avg_window))

    @classmethod
    def fft(cls, data: np.ndarray, p: RefParams | None = None) -> np.ndarray:
        p = p or cls.params()
        d = np.array(data, dtype=np.float32, copy=True)
        cls.lib().glvref_fft(C.byref(p), d, d.size)
        return d


class RefStream:
    """Reference fft -> gravity -> average chain for one stereo stream (per-channel slots)."""

    def __init__(self, p: RefParams, gravity=True, average=True):
        self.p, self.use_g, self.use_a = p, gravity, average
        self.gslot = [C.c_void_p(None), C.c_void_p(None)]
        self.aslot = [C.c_void_p(None), C.c_void_p(None)]

    def frame_from_float(self, l: np.ndarray, r: np.ndarray) -> np.ndarray:
        out = np.stack([np.array(l, np.float32), np.array(r, np.float32)])
        L = Ref.lib()
        for c in range(2):
            buf = np.ascontiguousarray(out[c])
            L.glvref_fft(C.byref(self.p), buf, buf.size)
            if self.use_g:
                L.glvref_gravity(C.byref(self.p), C.byref(self.gslot[c]), buf, buf.size)
            if self.use_a:
                L.glvref_average(C.byref(self.p), C.byref(self.aslot[c]), buf, buf.size)
            out[c] = buf
        return out

    def close(self):
        for s in self.gslot + self.aslot:
            Ref.lib().glvref_slot_free(C.byref(s))


def tones_pcm(stream: int, n: int, frame: int = 0) -> np.ndarray:
    """SURVEY.md 8d "tones" generator: interleaved stereo s16 frame `frame` (n samples per channel) of
    stream `stream`: 8000 sin(2 pi f1 t/22050) + 4000 sin(2 pi f2 t/22050 + c), f1 = 110 * 2^((b mod 48)/12),
    f2 = 3 f1, channel phase c in {0, 1}, plus LCG noise >> 6."""
    t = np.arange(frame * n, (frame + 1) * n, dtype=np.float64)
    f1 = 110.0 * 2.0 ** ((stream % 48) / 12.0)
    noise = (lcg_pcm_fast(12345 + stream + 7919 * frame, 2 * n).astype(np.int32) >> 6).reshape(n, 2)
    out = np.empty((n, 2), np.int32)
    for c in range(2):
        x = 8000.0 * np.sin(2 * np.pi * f1 * t / 22050.0) + 4000.0 * np.sin(2 * np.pi * 3 * f1 * t / 22050.0 + c)
        out[:, c] = np.rint(x).astype(np.int32) + noise[:, c]
    return np.clip(out, -32768, 32767).astype(np.int16).reshape(-1)
