"""ctypes binding of the C ABI (include/glv_spectrum.h) -- plumbing, not the product.

The product is libglvspectrum.so (hand-written HIP for gfx950, glava_amd/csrc).  This
module only loads it, mirrors `glv_params`, and passes raw device pointers (from torch
tensors or any other allocator) across.  There is no fallback of any kind: if the shared
object is missing or the device is unusable, calls raise `GlvError`.

Names follow the reference's operator table (glava/render.c:849-856): fft, gravity,
avg(average), wrange.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

HERE = os.path.dirname(os.path.abspath(__file__))
# GLV_SPECTRUM_LIB: an A/B build of the SAME library (glava_amd.build --variant, tools/ab_bench.sh); never a fallback
LIB_PATH = os.environ.get("GLV_SPECTRUM_LIB") or os.path.join(HERE, "csrc", "libglvspectrum.so")

OP_FFT, OP_GRAVITY, OP_AVERAGE, OP_RAW, OP_WRANGE, OP_BARS, OP_SMOOTH, OP_MAGNITUDE, OP_R16 = 1, 2, 4, 8, 16, 32, 64, 128, 256
OP_PRIVATE_STATE, OP_RING_S16, OP_RING_F32, OP_OUTPUT_IS_STATE, OP_BARS_ONLY = 512, 1024, 2048, 4096, 8192
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_STATE = 0, 1, 2, 3, 4, 5
BARS_NONE, BARS_F32_CHAIN, BARS_F32_MATRIX, BARS_I8_EXACT, BARS_F32_SEQ = 0, 1, 2, 3, 4
ROUND_SINUSOIDAL, ROUND_CIRCULAR, ROUND_LINEAR = 0, 1, 2          # glv_params.round_formula
SAMPLE_AVERAGE, SAMPLE_MAXIMUM, SAMPLE_HYBRID = 0, 1, 2            # glv_params.sample_mode


class GlvError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"glv error {code}: {msg}")
        self.code = code


class CParams(C.Structure):
    _fields_ = [("n", C.c_uint32), ("channels", C.c_uint32), ("fft_scale", C.c_float), ("fft_cutoff", C.c_float),
                ("gravity_step", C.c_float), ("ur", C.c_float), ("avg_frames", C.c_uint32),
                ("avg_window", C.c_uint32), ("avg_window_kind", C.c_uint32), ("log_mode", C.c_uint32),
                ("bars", C.c_uint32), ("smooth_factor", C.c_float),
                ("smooth_distance", C.c_float), ("smooth_ratio", C.c_float), ("gl_storage", C.c_uint32),
                ("bar_phase", C.c_float),
                # ABI 7: smooth_audio()'s shape (smooth_parameters.glsl:17-42); 0 = the shipped value in every field
                ("round_formula", C.c_uint32), ("sample_mode", C.c_uint32), ("sample_hybrid_weight", C.c_float),
                ("sample_scale", C.c_float), ("sample_range", C.c_float)]


class MultiStats(C.Structure):
    """glv_multi_stats: the 32-byte record every rank contributes to the RCCL all-gather"""
    _fields_ = [("frames", C.c_uint64), ("seconds", C.c_double), ("bytes", C.c_uint64), ("kernel_ms", C.c_double)]


_lib = None


def lib() -> C.CDLL:
    """Load libglvspectrum.so (built in-tree by glava_amd.build); fail loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GlvError(ERR_NO_DEVICE, f"{LIB_PATH} not built -- run `python -m glava_amd.build` "
                                          "(there is no Python/CPU fallback)")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64 with the same soname as
        # /opt/rocm's.  Whichever is loaded first wins for everybody, and torch.cuda only comes up on
        # its own copy -- so make sure torch (when installed) is imported before our library pulls in
        # the system one.  This is process plumbing, not a dependency: the C ABI itself is torch-free.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        P = C.POINTER(CParams)
        vp = C.c_void_p
        L.glv_params_default.argtypes = [P]; L.glv_params_default.restype = None
        L.glv_abi_version.restype = C.c_int
        L.glv_last_error.restype = C.c_char_p
        L.glv_device_count.restype = C.c_int
        L.glv_state_create.argtypes = [P, C.c_int, C.POINTER(vp)]
        L.glv_state_reset.argtypes = [vp]
        L.glv_state_destroy.argtypes = [vp]
        for name in ("glv_fft", "glv_gravity", "glv_average", "glv_wrange", "glv_smooth", "glv_magnitude", "glv_fft_gravity_average"):
            getattr(L, name).argtypes = [P, vp, vp]
        L.glv_texels_r16.argtypes = [P, vp, vp, vp]
        L.glv_gl_texture.argtypes = [P, vp, vp, C.c_int, vp]
        L.glv_unpack_s16.argtypes = [C.c_int, vp, C.c_size_t, C.c_int, vp, vp]
        L.glv_batch_create.argtypes = [P, C.c_uint32, C.c_uint, C.c_int, C.POINTER(vp)]
        L.glv_batch_reset.argtypes = [vp]
        L.glv_batch_set_params.argtypes = [vp, P]
        L.glv_batch_destroy.argtypes = [vp]
        L.glv_batch_process_s16.argtypes = [vp, vp, vp, C.c_uint, vp]
        L.glv_batch_process_f32.argtypes = [vp, vp, vp, C.c_uint, vp]
        L.glv_batch_process_f32_stereo.argtypes = [vp, vp, vp, C.c_uint, vp]
        L.glv_batch_ring_update_s16.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint, vp]
        L.glv_batch_ring_update_f32.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint, vp]
        L.glv_batch_bars.argtypes = [vp, vp, vp, vp]
        L.glv_batch_ring_append_s16.argtypes = [vp, vp, C.c_uint32, vp]
        L.glv_batch_ring_append_f32.argtypes = [vp, vp, C.c_uint32, vp]
        L.glv_batch_ring_planar.argtypes = [vp, C.c_int, vp, vp]
        L.glv_batch_gravity_state.argtypes = [vp, C.POINTER(C.c_void_p)]
        L.glv_prelude_bufscale.argtypes = [C.c_int, vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, vp]
        L.glv_prelude_lerp.argtypes = [C.c_int, vp, vp, vp, C.c_size_t, C.c_float, C.c_int, vp]
        L.glv_batch_timing_begin.argtypes = [vp]
        L.glv_batch_timing_end.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.glv_batch_live_bins.argtypes = [vp]; L.glv_batch_live_bins.restype = C.c_uint32
        L.glv_batch_tune_placement.argtypes = [vp, vp, vp, C.c_uint, C.c_int, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.glv_batch_bars_arithmetic.argtypes = [vp]; L.glv_batch_bars_arithmetic.restype = C.c_int
        L.glv_batch_algorithmic_bytes.argtypes = [vp, C.c_uint, C.c_int]
        L.glv_batch_algorithmic_bytes.restype = C.c_uint64
        L.glv_batch_kernel_name.argtypes = [vp]; L.glv_batch_kernel_name.restype = C.c_char_p
        L.glv_batch_set_grid.argtypes = [vp, C.c_int]
        L.glv_batch_last_grid.argtypes = [vp]
        L.glv_batch_last_launches.argtypes = [vp]
        L.glv_batch_variants.argtypes = [vp]
        L.glv_batch_set_variant.argtypes = [vp, C.c_int]
        L.glv_batch_last_variant.argtypes = [vp]
        L.glv_batch_describe_variant.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
        L.glv_batch_window_selftest.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
        L.glv_batch_autotune.argtypes = [vp, vp, vp, C.c_uint, vp, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.glv_wisdom_save.argtypes = [C.c_char_p]; L.glv_wisdom_load.argtypes = [C.c_char_p]
        L.glv_multi_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.glv_multi_shard_range.restype = None
        L.glv_multi_create.argtypes = [P, C.c_uint64, C.c_uint, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.glv_multi_destroy.argtypes = [vp]
        L.glv_multi_devices.argtypes = [vp]
        L.glv_multi_uses_rccl.argtypes = [vp]
        L.glv_multi_shard.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(vp)]
        L.glv_multi_run_s16.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.c_uint, C.c_int, C.c_int, C.POINTER(MultiStats), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc != OK:
        raise GlvError(rc, lib().glv_last_error().decode())


@dataclass
class Params:
    """Mirror of glv_params; defaults are the shipped GLava configuration."""
    n: int = 4096
    channels: int = 2
    fft_scale: float = 10.2
    fft_cutoff: float = 0.3
    gravity_step: float = 4.2
    ur: float = 22050.0 / 256.0
    avg_frames: int = 5
    avg_window: bool = True
    avg_window_kind: int = 0
    log_mode: int = 1
    bars: int = 80
    smooth_factor: float = 0.025
    smooth_distance: float = 0.01
    smooth_ratio: float = 4.0
    gl_storage: int = 0
    bar_phase: float = 0.0
    round_formula: int = 0          # ROUND_SINUSOIDAL / ROUND_CIRCULAR / ROUND_LINEAR
    sample_mode: int = 0            # SAMPLE_AVERAGE / SAMPLE_MAXIMUM / SAMPLE_HYBRID
    sample_hybrid_weight: float = 0.0
    sample_scale: float = 0.0
    sample_range: float = 0.0

    def c(self) -> CParams:
        return CParams(self.n, self.channels, self.fft_scale, self.fft_cutoff, self.gravity_step, self.ur,
                       self.avg_frames, int(self.avg_window), self.avg_window_kind, self.log_mode,
                       self.bars, self.smooth_factor, self.smooth_distance, self.smooth_ratio, self.gl_storage, self.bar_phase,
                       self.round_formula, self.sample_mode, self.sample_hybrid_weight, self.sample_scale, self.sample_range)


def _ptr(x) -> C.c_void_p:
    """Raw pointer of a torch tensor / numpy array / int / None."""
    if x is None:
        return C.c_void_p(None)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if hasattr(x, "ctypes"):
        return C.c_void_p(x.ctypes.data)
    raise TypeError(type(x))


class Batch:
    """B independent stereo streams on one GPU (glv_batch)."""

    def __init__(self, params: Params, streams: int, ops_mask: int = OP_FFT, device: int = 0):
        self.params, self.streams, self.ops_mask, self.device = params, streams, ops_mask, device
        self._h = C.c_void_p(None)
        cp = params.c()
        _check(lib().glv_batch_create(C.byref(cp), streams, ops_mask, device, C.byref(self._h)))

    def set_params(self, params: Params) -> None:
        """glv_batch_set_params: change scalar knobs (synchronous: tables are regenerated); n, avg_frames and the state's storage
        class are fixed at creation"""
        cp = params.c()
        _check(lib().glv_batch_set_params(self._h, C.byref(cp)))
        self.params = params

    def process_s16(self, d_pcm, d_out, ops: int = OP_FFT, stream: int | None = None) -> None:
        """one update of every stream; stream-ordered (never allocates, never copies synchronously).  Chains ending in gravity
        keep their state in a batch-owned buffer; with OP_OUTPUT_IS_STATE d_out itself becomes the state the next update reads
        (leave it intact until then; see include/glv_spectrum.h)."""
        _check(lib().glv_batch_process_s16(self._h, _ptr(d_pcm), _ptr(d_out), ops, _ptr(stream)))

    def process_f32(self, d_in, d_out, ops: int = OP_FFT, stream: int | None = None) -> None:
        _check(lib().glv_batch_process_f32(self._h, _ptr(d_in), _ptr(d_out), ops, _ptr(stream)))

    def process_f32_stereo(self, d_in, d_out, ops: int = OP_FFT, stream: int | None = None) -> None:
        _check(lib().glv_batch_process_f32_stereo(self._h, _ptr(d_in), _ptr(d_out), ops, _ptr(stream)))

    def ring_update_s16(self, d_new, new_frames: int, d_out, ops: int = OP_FFT, stream: int | None = None) -> None:
        _check(lib().glv_batch_ring_update_s16(self._h, _ptr(d_new), new_frames, _ptr(d_out), ops, _ptr(stream)))

    def gravity_state(self) -> int:
        """device address of the gravity state float [streams][2][n] (the spectra when d_out is None)"""
        p = C.c_void_p()
        _check(lib().glv_batch_gravity_state(self._h, C.byref(p)))
        return int(p.value)

    def ring_update_f32(self, d_new, new_frames: int, d_out, ops: int = OP_FFT, stream: int | None = None) -> None:
        _check(lib().glv_batch_ring_update_f32(self._h, _ptr(d_new), new_frames, _ptr(d_out), ops, _ptr(stream)))

    def ring_append_s16(self, d_new, new_frames: int, stream: int | None = None) -> None:
        _check(lib().glv_batch_ring_append_s16(self._h, _ptr(d_new), new_frames, _ptr(stream)))

    def ring_append_f32(self, d_new, new_frames: int, stream: int | None = None) -> None:
        _check(lib().glv_batch_ring_append_f32(self._h, _ptr(d_new), new_frames, _ptr(stream)))

    def ring_planar(self, d_planar, f32_ring: bool = False, stream: int | None = None) -> None:
        """the ring as the reference's backends publish it: float [streams][2][n], oldest sample first"""
        _check(lib().glv_batch_ring_planar(self._h, int(f32_ring), _ptr(d_planar), _ptr(stream)))

    def bars(self, d_spec, d_bars, stream: int | None = None) -> None:
        _check(lib().glv_batch_bars(self._h, _ptr(d_spec), _ptr(d_bars), _ptr(stream)))

    def reset(self) -> None:
        _check(lib().glv_batch_reset(self._h))

    def timing_begin(self) -> None:
        _check(lib().glv_batch_timing_begin(self._h))

    def timing_end(self) -> tuple[float, int]:
        ms, n = C.c_double(0), C.c_uint64(0)
        _check(lib().glv_batch_timing_end(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def live_bins(self) -> int:
        return int(lib().glv_batch_live_bins(self._h))

    def bars_arithmetic(self) -> int:
        """BARS_NONE / BARS_F32_CHAIN / BARS_F32_MATRIX / BARS_I8_EXACT (include/glv_spectrum.h glv_batch_bars_arithmetic)"""
        return int(lib().glv_batch_bars_arithmetic(self._h))

    def algorithmic_bytes(self, ops: int, input_is_s16: bool = True) -> int:
        return int(lib().glv_batch_algorithmic_bytes(self._h, ops, int(input_is_s16)))

    def kernel_name(self) -> str:
        return lib().glv_batch_kernel_name(self._h).decode()

    def set_grid(self, grid: int) -> None:
        _check(lib().glv_batch_set_grid(self._h, grid))

    def last_grid(self) -> int:
        return int(lib().glv_batch_last_grid(self._h))

    def last_launches(self) -> int:
        """kernels the last process / ring-update call launched"""
        return int(lib().glv_batch_last_launches(self._h))

    def variants(self) -> int:
        """kernel configurations built for this size (glv_inst.hip Tuned<K, V>)"""
        return int(lib().glv_batch_variants(self._h))

    def set_variant(self, variant: int) -> None:
        _check(lib().glv_batch_set_variant(self._h, variant))

    def last_variant(self) -> int:
        return int(lib().glv_batch_last_variant(self._h))

    def window_selftest(self) -> tuple[int, int]:
        """(mismatches, shifted): every (s16 sample value, window position) pair of this size, float-pair product against the
        reference's fp64 product, on the device; positions whose low part was moved to make the identity hold"""
        m, sh = C.c_ulonglong(0), C.c_int(0)
        _check(lib().glv_batch_window_selftest(self._h, C.byref(m), C.byref(sh)))
        return int(m.value), int(sh.value)

    def describe_variant(self, variant: int) -> str:
        buf = C.create_string_buffer(256)
        _check(lib().glv_batch_describe_variant(self._h, variant, buf, len(buf)))
        return buf.value.decode()

    def autotune(self, d_pcm, d_out, ops: int = OP_FFT, stream: int | None = None) -> tuple[int, float]:
        """time the candidate workgroup counts on the device, record the winner in the process-wide wisdom"""
        g, ms = C.c_int(0), C.c_float(0)
        _check(lib().glv_batch_autotune(self._h, _ptr(d_pcm), _ptr(d_out), ops, _ptr(stream), C.byref(g), C.byref(ms)))
        return g.value, ms.value

    def tune_placement(self, d_pcm, d_out, ops: int, candidates: int = 6, stream: int | None = None) -> tuple[float, float]:
        """placement wisdom (include/glv_spectrum.h glv_batch_tune_placement): keep the fastest of `candidates` placements of the state arrays
        for THESE buffers; resets the state.  Returns (ms per update of the placement it had, of the one it has now)"""
        f, b = C.c_float(0), C.c_float(0)
        _check(lib().glv_batch_tune_placement(self._h, _ptr(d_pcm), _ptr(d_out), ops, candidates, _ptr(stream), C.byref(f), C.byref(b)))
        return f.value, b.value

    def close(self) -> None:
        if self._h:
            lib().glv_batch_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multi_shard_range(total_streams: int, rank: int, world: int) -> tuple[int, int]:
    """(first, count) of rank's contiguous shard -- the C twin of glava_amd.sharding.shard_range"""
    lo, cnt = C.c_uint64(0), C.c_uint64(0)
    lib().glv_multi_shard_range(total_streams, rank, world, C.byref(lo), C.byref(cnt))
    return int(lo.value), int(cnt.value)


class Multi:
    """One host, several GPUs (glv_multi): contiguous shards, one host thread per device, RCCL only for the stats."""

    def __init__(self, params: Params, total_streams: int, ops_mask: int = OP_FFT, devices: list[int] | None = None, ndev: int | None = None):
        self._h = C.c_void_p(None)
        cp = params.c()
        n = len(devices) if devices is not None else (1 if ndev is None else int(ndev))
        arr = (C.c_int * max(n, 1))(*devices) if devices is not None else None
        _check(lib().glv_multi_create(C.byref(cp), total_streams, ops_mask, arr, n, C.byref(self._h)))
        self.ndev = n

    def uses_rccl(self) -> bool:
        return bool(lib().glv_multi_uses_rccl(self._h))

    def shard(self, idx: int) -> tuple[int, int, int]:
        dev, lo, cnt = C.c_int(0), C.c_uint64(0), C.c_uint32(0)
        _check(lib().glv_multi_shard(self._h, idx, C.byref(dev), C.byref(lo), C.byref(cnt), None))
        return dev.value, int(lo.value), int(cnt.value)

    def run_s16(self, d_pcm: list, d_out: list, ops: int, warmup: int, steps: int) -> tuple[list[dict], float]:
        pin = (C.c_void_p * self.ndev)(*[_ptr(x) for x in d_pcm])
        pout = (C.c_void_p * self.ndev)(*[_ptr(x) for x in d_out])
        st = (MultiStats * self.ndev)()
        mx = C.c_double(0)
        _check(lib().glv_multi_run_s16(self._h, pin, pout, ops, warmup, steps, st, C.byref(mx)))
        return [{"frames": s.frames, "seconds": s.seconds, "bytes": s.bytes, "kernel_ms": s.kernel_ms} for s in st], mx.value

    def close(self) -> None:
        if self._h:
            lib().glv_multi_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class State:
    """Per-(stream, channel) state behind the single-buffer drop-ins (glv_state): the
    replacement for the `*udata` slot of the reference's operator seam (render.c:106-118)."""

    def __init__(self, params: Params, device: int = 0):
        self.params = params
        self._h = C.c_void_p(None)
        cp = params.c()
        _check(lib().glv_state_create(C.byref(cp), device, C.byref(self._h)))

    def _call(self, name: str, buf) -> None:
        cp = self.params.c()
        _check(getattr(lib(), name)(C.byref(cp), self._h, _ptr(buf)))

    def fft(self, buf) -> None: self._call("glv_fft", buf)                 # transform_fft
    def gravity(self, buf) -> None: self._call("glv_gravity", buf)         # transform_gravity
    def average(self, buf) -> None: self._call("glv_average", buf)         # transform_average
    def wrange(self, buf) -> None: self._call("glv_wrange", buf)           # transform_wrange
    def smooth(self, buf) -> None: self._call("glv_smooth", buf)           # transform_smooth
    def magnitude(self, buf) -> None: self._call("glv_magnitude", buf)     # tail of transform_fft
    def fft_gravity_average(self, buf) -> None: self._call("glv_fft_gravity_average", buf)

    def texels_r16(self, buf, texels) -> None:
        """the GL_R16 texels handle_audio's upload stores for buf (render.c:521-524); host buffers"""
        cp = self.params.c()
        _check(lib().glv_texels_r16(C.byref(cp), self._h, _ptr(buf), _ptr(texels)))

    def gl_texture(self, buf, texels, smooth_pass: bool = True) -> None:
        """handle_audio's accel_fft branch from the per-frame transform_fft to the texture the module samples (render.c:2176-2303);
        host buffers: n float samples in (not modified), n GL_R16 texels out; the state must have gl_storage = 1"""
        cp = self.params.c()
        _check(lib().glv_gl_texture(C.byref(cp), self._h, _ptr(buf), 1 if smooth_pass else 0, _ptr(texels)))

    def reset(self) -> None:
        _check(lib().glv_state_reset(self._h))

    def close(self) -> None:
        if self._h:
            lib().glv_state_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unpack_s16(pcm, frames: int, channels, l, r, device: int = 0) -> None:
    """fifo.c:94-110 on the device; host numpy buffers in/out."""
    _check(lib().glv_unpack_s16(device, _ptr(pcm), frames, channels, _ptr(l), _ptr(r)))


def prelude_bufscale(d_in, d_out, rows: int, n_out: int, k: int, device: int = 0, stream: int | None = None) -> None:
    """render.c:1768-1781 box decimation on device buffers."""
    _check(lib().glv_prelude_bufscale(device, _ptr(d_in), _ptr(d_out), rows, n_out, k, _ptr(stream)))


def prelude_lerp(d_start, d_end, d_out, count: int, uratio: float, kcounter: int, device: int = 0,
                 stream: int | None = None) -> None:
    """render.c:1794-1809 keyframe interpolation on device buffers."""
    _check(lib().glv_prelude_lerp(device, _ptr(d_start), _ptr(d_end), _ptr(d_out), count, uratio, kcounter, _ptr(stream)))


def wisdom_save(path: str) -> None: _check(lib().glv_wisdom_save(path.encode()))
def wisdom_load(path: str) -> None: _check(lib().glv_wisdom_load(path.encode()))
def wisdom_clear() -> None: lib().glv_wisdom_clear()
def wisdom_count() -> int: return int(lib().glv_wisdom_count())


def device_count() -> int:
    return int(lib().glv_device_count())
