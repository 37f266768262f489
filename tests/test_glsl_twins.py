"""The GL twins of the path against an INDEPENDENT evaluation of the reference's shader text.

tests/glsl_eval.py interprets shaders/glava/util/{smooth.glsl, average_pass.frag, gravity_pass.frag, common.glsl} -- with
GLava's preprocessing (#include, #expand, ## pasting; the unparenthesised window() macro comes out exactly as the shader
compiler would see it) and float32 arithmetic; tests/golden/glsl_vectors.npz holds its outputs (generator committed,
tests/golden/make_glsl_golden.py).  Checked against them:
  * CPU: the golden vectors regenerate bit for bit where /root/reference exists; the oracle's C restatements (glvo_bars,
    glvo_average_gl, the gravity step) agree with the shader evaluation to float rounding -- so the checker of the GPU
    tests is no longer only its author's second reading of the shader;
  * CPU: the host side of GLV_OP_BARS (tap tables + work lists, through the host emulator) against the same vectors;
  * GPU: glv_batch_bars and the avg_window_kind = 1 average on the device against the same vectors.
GLSL itself still cannot run here (no GL context); what a driver's float pipeline adds (fused multiply-adds, the order
of a compiler-unrolled sum) stays outside: tolerances are a few float ulps, stated below.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_glsl_golden import AVG, BARS, tex_row  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "glsl_vectors.npz"))
ULPS = 8 * 2.0 ** -24          # relative: a handful of float ulps (summation order of ~100-tap bars)


def test_golden_vectors_regenerate_from_the_shader_text():
    if not os.path.exists("/root/reference/shaders/glava/util/smooth.glsl"):
        pytest.skip("/root/reference absent (GPU box): the committed vectors are used as they are")
    import glsl_eval as G
    n, bars, factor, seed = BARS[1]
    assert (G.smooth_audio_bars(tex_row(n, seed), bars, factor).view(np.uint32) == GOLD[f"bars_n{n}_b{bars}_f{factor}_s{seed}"].view(np.uint32)).all()
    F, win = AVG[2]
    frames = [tex_row(256, 100 + F * 10 + i) for i in range(F)]
    assert (G.average_pass(frames, bool(win)).view(np.uint32) == GOLD[f"avg_F{F}_w{win}"].view(np.uint32)).all()
    # the preprocessor output shows the macro bug the CPU twin shares: window(I, _AVG_FRAMES - 1) -> cos(TWOPI * I / F - 1)
    src, _ = G.preprocess(os.path.join(G.SHADER_ROOT, "util", "average_pass.frag"), {"_AVG_FRAMES": 5, "_AVG_WINDOW": 1})
    assert "cos(6.28318530718 * 3 / 5 - 1)" in src and "0.53836" in src


@pytest.mark.parametrize("n,bars,factor,seed", BARS)
def test_oracle_bars_equal_the_shader_evaluation(n, bars, factor, seed):
    want = GOLD[f"bars_n{n}_b{bars}_f{factor}_s{seed}"]
    got = np.empty(bars, np.float32)
    Oracle.lib().glvo_bars(tex_row(n, seed), n, got, bars, factor)
    assert np.abs(got - want).max() <= ULPS * np.abs(want).max(), np.abs(got - want).max()


@pytest.mark.parametrize("F,win", AVG)
def test_oracle_gl_average_equals_the_shader_evaluation(F, win):
    want = GOLD[f"avg_F{F}_w{win}"]
    frames = [tex_row(256, 100 + F * 10 + i) for i in range(F)]          # index 0 = newest
    hist = np.zeros((F, 256), np.float32)
    head = C.c_size_t(0)
    out = None
    for f in reversed(frames):                                          # feed oldest first; the last call averages all F
        out = f.copy()
        Oracle.lib().glvo_average_gl(out, hist, C.byref(head), 256, F, win)
    assert np.abs(out - want).max() <= ULPS * np.abs(want).max()


def test_oracle_gravity_step_equals_the_shader_evaluation():
    want = GOLD["gravity_diff0.0487"]
    store = tex_row(256, 7)
    g = np.float32(4.2) * (np.float32(1.0) / np.float32(86.1328125))      # render.c:2225: gravity_step * (1.0F / ur)
    assert (store - g).astype(np.float32).view(np.uint32).tolist() == want.view(np.uint32).tolist()


@pytest.mark.parametrize("n,bars,factor,seed", [b for b in BARS if b[1] <= 80])
def test_host_bar_tables_equal_the_shader_evaluation(emu, n, bars, factor, seed):
    """tap tables and work lists of GLV_OP_BARS (glv_tables.h) through the host emulator, 16 groups as glv_bars_kernel"""
    want = GOLD[f"bars_n{n}_b{bars}_f{factor}_s{seed}"]
    tex = tex_row(n, seed)
    got = np.empty(bars, np.float32)
    emu.glvemu_bars.argtypes = [np.ctypeslib.ndpointer(np.float32), C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_int,
                                np.ctypeslib.ndpointer(np.float32), C.c_void_p, C.c_float]
    assert emu.glvemu_bars(tex, 1, n, bars, factor, 16, got, None, 0.0) == 0
    assert np.abs(got - want).max() <= 4 * ULPS * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n,bars,factor,seed", [b for b in BARS if b[1] <= 80])
def test_device_bars_equal_the_shader_evaluation(glvlib, n, bars, factor, seed):
    import torch
    G = glvlib
    want = GOLD[f"bars_n{n}_b{bars}_f{factor}_s{seed}"]
    tex = tex_row(n, seed)
    b = G.Batch(G.Params(n=n, bars=bars, smooth_factor=factor), 1, G.OP_FFT)
    d_spec = torch.from_numpy(np.stack([tex, tex])).cuda()
    d_bars = torch.empty((2, bars), dtype=torch.float32, device="cuda")
    b.bars(d_spec, d_bars)
    got = d_bars.cpu().numpy()
    b.close()
    for r in range(2):
        assert np.abs(got[r] - want).max() <= 4 * ULPS * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("F,win", AVG)
def test_device_gl_average_equals_the_shader_evaluation(glvlib, F, win):
    """avg_window_kind = 1 (Hamming coefficients, newest frame first, no window at F == 2) on planar rows"""
    import torch
    G = glvlib
    want = GOLD[f"avg_F{F}_w{win}"]
    frames = [tex_row(256, 100 + F * 10 + i) for i in range(F)]
    p = G.Params(n=256, avg_frames=F, avg_window=bool(win), avg_window_kind=1)
    b = G.Batch(p, 1, G.OP_AVERAGE)
    d_out = torch.empty((2, 256), dtype=torch.float32, device="cuda")
    for f in reversed(frames):
        b.process_f32(torch.from_numpy(np.stack([f, f])).cuda(), d_out, G.OP_AVERAGE)
    got = d_out.cpu().numpy()
    b.close()
    for r in range(2):
        assert np.abs(got[r] - want).max() <= ULPS * np.abs(want).max()
