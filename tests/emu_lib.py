"""Thin wrapper around the host emulator entry point (tests/emu/glv_emu.cpp)."""
import ctypes as C

import numpy as np


def emu_process(L, n, data, units, ops, grav=None, hist=None, F=5, head=0, mono=0, avg_window=1,
                avg_kind=0, log_mode=0, in_mode=0, fft_scale=10.2, fft_cutoff=0.3, gravity_step=4.2,
                ur=86.1328125, rot=0, log_e=4):
    rows = units * (2 if in_mode in (0, 3) else 1)      # `units` = stereo frames (s16 / f32 interleaved) or rows (f32 planar)
    out = np.zeros((rows, n), np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    rc = L.glvemu_process(n, in_mode, vp(data), vp(out), vp(grav), vp(hist), rows, ops, F, head, mono,
                          avg_window, avg_kind, log_mode, C.c_float(fft_scale), C.c_float(fft_cutoff),
                          C.c_float(gravity_step), C.c_float(ur), rot, log_e)
    assert rc == 0, rc
    return out


def emu_bars(L, spec, n, bars, smooth_factor=0.025, groups=16, phase=0.0):
    spec = np.ascontiguousarray(spec, dtype=np.float32).reshape(-1, n)
    out = np.zeros((spec.shape[0], bars), np.float32)
    steps = C.c_uint(0)
    rc = L.glvemu_bars(spec.ctypes.data_as(C.c_void_p), C.c_size_t(spec.shape[0]), n, bars, C.c_float(smooth_factor), groups,
                       out.ctypes.data_as(C.c_void_p), C.byref(steps), C.c_float(phase))
    assert rc == 0, rc
    return out, steps.value
