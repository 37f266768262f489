"""SURVEY.md 8a row a12, pinned to an EXECUTION of the reference's GL path.

tests/golden/gl_vectors.npz holds the exact 16-bit texels the reference's own rd_new / rd_update leave in their GL_R16 textures
(upload, gravity store, ring average, pre-smoothing pass; render.c:521-524, 2188-2303 with the shipped shaders
util/gravity_pass.frag, average_pass.frag, pass.frag, smooth_pass.frag + smooth.glsl) when they run over a real OpenGL 4.5 core
context -- Mesa llvmpipe, driven through swrast_dri.so's DRI interface by oracle/glref_harness.c (no X server, no EGL).
Generator: tests/golden/make_gl_golden.py (committed; needs /root/reference).

  CPU   the oracle's restatements (glvo_gl_chain_r16, glvo_bars_at) against those texels; where the harness can be built, a
        live re-run must reproduce the committed file bit for bit;
  GPU   the HIP path's gl_storage chain (avg_window_kind 1) and GLV_OP_BARS at the pre-smoothing pass's texel centres
        (bar_phase 0.5, bars == n) against the same texels.

Tolerances are what separates two conforming GL implementations, nothing more: a float -> UNORM16 conversion may take either
neighbour at a tie (OpenGL 4.6 section 2.3.5; exact ties are common in the unwindowed average: a sum of F integers over F),
and log() / sin() of the shader compiler's math library may differ from glibc's in the last ulp, which can move ONE tap in or
out of ONE bar's window.  Hence: every texel within 1 step, mismatches counted, outliers (the tap-count case) bounded."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle_lib import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "gl_vectors.npz"))
UR = float(GOLD["ur"])
CASES = [("n1024_F5w", 1024, 5, True), ("n1024_F6u", 1024, 6, False), ("n1024_F1", 1024, 1, True), ("n1024_F3w", 1024, 3, True),
         ("n1024_F2w", 1024, 2, True), ("n2048_F5w_loud", 2048, 5, True), ("n4096_F5w", 4096, 5, True)]
UP, GR, AV, SM = 0, 1, 2, 3


def texel_float(t):
    return (t.astype(np.float32) / np.float32(65535)).copy()        # what a shader reads back: c / 65535 (OpenGL 4.6 eq. 2.1), correctly rounded


def diff_stats(a, b):
    d = np.abs(a.astype(np.int64) - b.astype(np.int64))
    return int(d.max()), float((d != 0).mean()), int((d > 1).sum())


@pytest.mark.parametrize("name,n,F,win", CASES)
def test_oracle_gl_passes_against_the_reference_gl_execution(name, n, F, win):
    """gravity store: EXACT; average: within one texel step (ties); upload: within one step; pre-smoothing pass: within one step
    except where a bar gains / loses a tap to the math library (at most 2 texels per row)."""
    pcm, tex = GOLD[name + "_pcm"], GOLD[name + "_tex"]
    store = np.zeros((2, n), np.float32); hist = np.zeros((2, F, n), np.float32)
    heads = [C.c_size_t(0), C.c_size_t(0)]
    worst = {"up": 0.0, "av": 0.0, "sm": 0.0}
    for f in range(pcm.shape[0]):
        for ch in range(2):
            x = pcm[f, :, ch].astype(np.float32) / np.float32(65535)                       # fifo.c:105-106
            mx, frac, out = diff_stats(Oracle.texels_r16(Oracle.transform_fft(x)), tex[f, ch, UP])
            assert mx <= 1 and frac < 5e-3, ("upload", f, ch, mx, frac)
            worst["up"] = max(worst["up"], frac)
            # the passes, fed with the reference's own upload so that each comparison isolates one pass
            row = texel_float(tex[f, ch, UP])
            Oracle.lib().glvo_gl_chain_r16(row, store[ch], hist[ch], C.byref(heads[ch]), n, F, int(win), 1, 4.2, UR)
            assert (Oracle.texels_r16(store[ch]) == tex[f, ch, GR]).all(), ("gravity store", f, ch)
            mx, frac, out = diff_stats(Oracle.texels_r16(row), tex[f, ch, AV])
            assert mx <= 1 and frac < (0.55 if not win or F == 2 else 5e-3), ("average", f, ch, mx, frac)   # unweighted sums: exact half-texel ties
            worst["av"] = max(worst["av"], frac)
            # keep the model's state on the reference's texels (a one-step difference must not propagate into later frames)
            store[ch] = texel_float(tex[f, ch, GR])
            hist[ch][(heads[ch].value + F - 1) % F] = store[ch]
            sm = np.empty(n, np.float32)
            Oracle.lib().glvo_bars_at(texel_float(tex[f, ch, AV]), n, sm, n, 0.025, 0.5)
            mx, frac, out = diff_stats(Oracle.texels_r16(sm), tex[f, ch, SM])
            assert out <= 2 and frac < 2e-2, ("smooth pass", f, ch, mx, frac, out)
            worst["sm"] = max(worst["sm"], frac)
    print(name, worst)


def test_gl_golden_file_is_reproducible(tmp_path):
    """where /root/reference and Mesa's swrast driver exist (the build container), the reference's GL path is run again for one
    case and must give the committed texels bit for bit"""
    so = os.path.join(ROOT, "oracle", "_ref", "libglvglref.so")
    if not os.path.exists("/root/reference/shaders/glava/rc.glsl"):
        pytest.skip("needs the reference tree (build container only)")
    if not os.path.exists(so):
        from oracle_lib import build_oracles
        build_oracles()
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libglvglref.so could not be built (no Mesa swrast_dri.so / DRI headers)")
    import subprocess, sys
    # in a child process: rd_new prints deprecation warnings and glava_abort()s on errors
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import make_gl_golden as M, tempfile; "
            "pcm = M.frames_of('n1024_F3w', 1024, 6, 4); tex, v, r = M.run_case(1024, 3, True, pcm, tempfile.mkdtemp()); "
            "np.save(%r, tex)") % (os.path.join(ROOT, "tests", "golden"), str(tmp_path / "t.npy"))
    subprocess.run([sys.executable, "-c", code], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert (np.load(str(tmp_path / "t.npy")) == GOLD["n1024_F3w_tex"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,F,win", CASES)
def test_device_gl_storage_chain_against_the_reference_gl_execution(glvlib, name, n, F, win):
    """The HIP path with gl_storage = 1 (GL_R16 storage of every pass) and the GL twin's window (avg_window_kind 1), from the
    same s16 PCM the reference's renderer was fed: its GL_R16 texels against the reference's own `av` texture, frame by frame;
    then GLV_OP_BARS at the pre-smoothing pass's positions (bar_phase 0.5, bars = n) on the reference's `av` texels against
    its `sm` texture."""
    import torch
    G = glvlib
    pcm, tex = GOLD[name + "_pcm"], GOLD[name + "_tex"]
    mask = G.OP_GRAVITY | (G.OP_AVERAGE if F > 1 else 0)
    ops = G.OP_FFT | mask
    p = G.Params(n=n, avg_frames=F, avg_window=win, avg_window_kind=1, gl_storage=1, log_mode=0, ur=UR, bars=n, bar_phase=0.5)
    b = G.Batch(p, 1, mask | G.OP_BARS)
    bb = G.Batch(p, 1, G.OP_FFT | G.OP_BARS)
    full = G.Batch(p, 1, mask | G.OP_BARS)              # the whole default pipeline in one call: ... | GLV_OP_BARS | GLV_OP_R16
    d_sm = torch.zeros((2, n), dtype=torch.int16, device="cuda")
    d_q = torch.zeros((2, n), dtype=torch.int16, device="cuda")
    d_bars = torch.empty((2, n), dtype=torch.float32, device="cuda")
    for f in range(pcm.shape[0]):
        d_pcm = torch.from_numpy(np.ascontiguousarray(pcm[f])).cuda()
        b.process_s16(d_pcm, d_q, ops | G.OP_R16)
        got = d_q.cpu().numpy().view(np.uint16)
        # PCM in, the texture the stock modules sample out (upload -> gravity -> average -> pre-smoothing pass, every one GL_R16)
        full.process_s16(d_pcm, d_sm, ops | G.OP_BARS | G.OP_R16)
        got_sm = d_sm.cpu().numpy().view(np.uint16)
        for ch in range(2):
            mx, frac, out = diff_stats(got_sm[ch], tex[f, ch, SM])
            assert out <= 2 and frac < (0.6 if not win or F == 2 else 5e-2), ("end to end", f, ch, mx, frac, out)
        for ch in range(2):
            mx, frac, out = diff_stats(got[ch], tex[f, ch, AV])
            # one-step differences of the upload (0.1 % of texels) travel through max / average: a little more slack than the
            # per-pass comparison of the CPU test, still within one texel step
            assert mx <= 1 and frac < (0.55 if not win or F == 2 else 2e-2), ("chain", f, ch, mx, frac)
        av = np.stack([texel_float(tex[f, ch, AV]) for ch in range(2)])
        bb.bars(torch.from_numpy(av).cuda(), d_bars)
        sm = Oracle.texels_r16(d_bars.cpu().numpy())
        for ch in range(2):
            mx, frac, out = diff_stats(sm[ch], tex[f, ch, SM])
            assert out <= 2 and frac < 2e-2, ("smooth pass", f, ch, mx, frac, out)
    b.close(); bb.close(); full.close()
