"""glv_params.gl_storage: the GL passes with their GL_R16 storage (render.c:2188-2265).

The model (oracle/glv_oracle.c glvo_gl_chain_r16): every intermediate of the accel path is a 16-bit unorm texture -- the
uploaded buffer, the gravity store after GL_MAX and the in-place subtraction, the ring copies, the average -- so values are
clamped to [0, 1] and quantised where a pass writes them.  Checked here:
  * CPU: the host twin of glv_post_kernel (tests/emu, same apply_state as the device) against the oracle model, bit for
    bit, over many frames; the texel read-back division for all 65536 texel values; what the model changes -- gravity's
    fixed point under silence is 0 (the float state machine's is -g), nothing leaves [0, 1], F == 1 has no averaging pass;
  * CPU: one averaging pass of the model against the independent evaluation of average_pass.frag (tests/glsl_eval.py
    golden vectors) to one texel step;
  * GPU: the device chain (frame kernel, then the GL-storage pass) against the oracle model.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle_lib import Oracle, StreamOracle, lcg_pcm_fast

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OP_FFT, OP_GRAVITY, OP_AVERAGE, OP_R16 = 1, 2, 4, 256
fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def emu_post(emu, x, grav, hist, ops, F, head, win, kind, gl, gstep=4.2, ur=86.1328125):
    rows, n = x.shape
    out = np.empty_like(x)
    emu.glvemu_post_state.argtypes = [fp, fp, C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int,
                                      C.c_int, C.c_int, C.c_float, C.c_float]
    rc = emu.glvemu_post_state(np.ascontiguousarray(x), out, grav.ctypes.data_as(C.c_void_p) if grav is not None else None,
                               hist.ctypes.data_as(C.c_void_p) if hist is not None else None, n, rows, ops, F, head, int(win), kind, gl, gstep, ur)
    assert rc == 0
    return out


def spectrum_like(rng, rows, n):
    return (rng.random((rows, n), dtype=np.float32) ** 2 * np.float32(1.4) - np.float32(0.1)).astype(np.float32)   # some < 0, some > 1


def test_gravity_step_on_texels_is_an_integer_subtraction(emu):
    """glv_core.h gravity_r16: Q(c / 65535 - g) == max(c - D, 0) for EVERY texel c -- what lets the kernels run the gravity pass as
    two packed 16-bit integer instructions.  The host decides per g by trying all 65 536 texels (glv_tables.h); restated here in
    numpy for the shipped step and a spread of others, incl. steps where the identity must be REFUSED (g * 65535 at a half-integer:
    float rounding then decides texel by texel) and steps that raise values."""
    c = np.arange(65536, dtype=np.float32)
    tex = (c / np.float32(65535)).astype(np.float32)

    def q(x):
        return np.rint(np.clip(x.astype(np.float64), 0.0, 1.0) * 65535.0).astype(np.int64)

    d = C.c_uint(0)
    n_int = 0
    for gstep, ur in [(4.2, 86.1328125), (4.2, 60.0), (4.2, 144.0), (1.0, 86.1328125), (0.0, 86.1328125), (4.2, 0.0), (100.0, 1.0),
                      (3195.5 / 65535 * 86.1328125, 86.1328125), (-1.0, 86.1328125)] + [(4.2, 30.0 + 1.37 * k) for k in range(40)]:
        with np.errstate(divide="ignore"):
            g = np.float32(gstep) * (np.float32(1.0) / np.float32(ur))
        ok = emu.glvemu_gravity_step(C.c_float(gstep), C.c_float(ur), C.byref(d))
        want = q((tex - g).astype(np.float32))
        if ok:
            n_int += 1
            assert (want == np.maximum(np.arange(65536) - d.value, 0)).all(), (gstep, ur, d.value)
        else:
            assert not (want == np.maximum(np.arange(65536) - d.value, 0)).all(), (gstep, ur)
    assert emu.glvemu_gravity_step(C.c_float(4.2), C.c_float(86.1328125), C.byref(d)) == 1 and d.value == 3196     # the shipped step: g * 65535 = 3195.6
    assert emu.glvemu_gravity_step(C.c_float(-1.0), C.c_float(86.1328125), C.byref(d)) == 0                        # raises values: float path
    assert n_int >= 40


@pytest.mark.parametrize("log_e", [4, 3, 5])
@pytest.mark.parametrize("n,F,win", [(512, 5, True), (1024, 6, False), (2048, 2, True), (4096, 5, True), (1024, 1, True), (16384, 3, True)])
def test_emulated_fused_gl_chain_equals_the_oracle_model(emu, n, F, win, log_e):
    """The fused GL_R16 epilogue (glv_frame.h epilogue_gl16 / gl16_state_block: upload quantisation, GL_MAX + gravity, ring,
    average on uint16 state inside the transform's kernel) walked lane by lane on the host, from s16 PCM, against the oracle's
    transform_fft + glvo_gl_chain_r16: texel output and float output, integer and float form of the gravity step -- bit for bit."""
    if log_e == 5 and n < 2048:
        pytest.skip("E = 32 needs at least 64 complex points per lane row")
    units = 2
    rows = units * 2
    vp = lambda a: a.ctypes.data_as(C.c_void_p)      # noqa: E731
    st = {k: (np.zeros((rows, n), np.uint16), np.zeros((rows, F, n), np.uint16)) for k in ("tex", "flt", "fg")}
    ostore = np.zeros((rows, n), np.float32); ohist = np.zeros((rows, F, n), np.float32)
    oheads = [C.c_size_t(0) for _ in range(rows)]
    head = 0
    ops = OP_FFT | OP_GRAVITY | OP_AVERAGE
    for fr in range(F + 2):
        pcm = (lcg_pcm_fast(900 + fr + n, units * 2 * n) // (1 if fr % 2 else 8)).astype(np.int16)
        outs = {}
        for key, o, force in (("tex", ops | OP_R16, 0), ("flt", ops, 0), ("fg", ops | OP_R16, 1)):
            out = np.zeros((rows, n), np.uint16 if o & OP_R16 else np.float32)
            rc = emu.glvemu_process_gl(n, 0, vp(pcm), vp(out), vp(st[key][0]), vp(st[key][1]), rows, o, F, head, 0, int(win), 1, 0,
                                       C.c_float(10.2), C.c_float(0.3), C.c_float(4.2), C.c_float(86.1328125), 0, log_e, 1, force)
            assert rc == 0
            outs[key] = out
        head = (head + 1) % F
        for u in range(units):
            spec = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n])
            for c in range(2):
                r = 2 * u + c
                want = np.ascontiguousarray(spec[c])
                Oracle.lib().glvo_gl_chain_r16(want, ostore[r], ohist[r], C.byref(oheads[r]), n, F, int(win), 1, 4.2, 86.1328125)
                assert (outs["tex"][r] == Oracle.texels_r16(want)).all(), (fr, r)
                assert (bits(outs["flt"][r]) == bits(want)).all(), (fr, r)
                assert (outs["fg"][r] == outs["tex"][r]).all(), (fr, r)
        for key in st:
            assert (st[key][1] == Oracle.texels_r16(ohist)).all(), (fr, key)


def test_final_division_of_the_average_is_correctly_rounded(emu):
    """glv_core.h div_frames (three fused operations instead of the divide expansion) against the IEEE quotient for EVERY float an
    averaging pass on GL_R16 values can hand it -- [2^-24, 128], all 226 M of them for the frame counts in use, every 61st for
    all of F = 1 .. 64 -- and for zero"""
    emu.glvemu_div_frames_check.restype = C.c_ulonglong
    lo, hi = int(np.float32(2.0 ** -24).view(np.uint32)), int(np.float32(128.0).view(np.uint32))
    bad = C.c_uint(0)
    for F in (2, 3, 5, 6):
        assert emu.glvemu_div_frames_check(F, lo, hi, 1, C.byref(bad)) == 0, (F, hex(bad.value))
    for F in range(1, 65):
        assert emu.glvemu_div_frames_check(F, lo, hi, 61, C.byref(bad)) == 0, (F, hex(bad.value))
        assert emu.glvemu_div_frames_check(F, 0, 0, 1, C.byref(bad)) == 0                      # zero (denormals cannot occur: the
                                                                                                # smallest nonzero sum is ~1e-6)


def test_texel_readback_division_every_texel(emu):
    """through_r16 on k / 65535 + a hair for every k: the device-side division sequence (glv_core.h unorm16_to_float) is the
    correctly rounded c / 65535.0f of the oracle"""
    k = np.arange(65536, dtype=np.float64)
    x = (k / 65535.0).astype(np.float32).reshape(64, 1024)
    got = emu_post(emu, x, np.zeros_like(x), None, OP_GRAVITY, 1, 0, True, 1, 2, gstep=0.0)    # g = 0: store = Q(max(0, Q(x)))
    got16 = emu_post(emu, x, np.zeros(x.shape, np.uint16), None, OP_GRAVITY, 1, 0, True, 1, 1, gstep=0.0)   # the same on texel state
    assert (bits(got16) == bits(got)).all()
    want = (np.arange(65536, dtype=np.uint16).astype(np.float32) / np.float32(65535)).reshape(64, 1024)
    tex = Oracle.texels_r16(x)
    assert (tex.reshape(-1) == np.arange(65536)).all()
    assert (bits(got) == bits(want)).all()


@pytest.mark.parametrize("gl", [1, 2])
@pytest.mark.parametrize("F,win", [(5, True), (6, False), (2, True), (1, True)])
def test_host_twin_equals_the_oracle_model(emu, F, win, gl):
    """gl 2: the state as floats c / 65535 (the pass-by-pass form); gl 1: the same state as uint16 texels (what the fused
    kernel keeps) -- identical outputs, and the texel arrays ARE the float arrays' texels"""
    rng = np.random.default_rng(F)
    rows, n = 3, 512
    sdt = np.uint16 if gl == 1 else np.float32
    grav = np.zeros((rows, n), sdt)
    hist = np.zeros((rows, F, n), sdt)
    ostore = np.zeros((rows, n), np.float32); ohist = np.zeros((rows, F, n), np.float32)
    oheads = [C.c_size_t(0) for _ in range(rows)]
    head = 0
    for fr in range(2 * F + 3):
        x = spectrum_like(rng, rows, n) if fr != 3 else np.zeros((rows, n), np.float32)
        got = emu_post(emu, x, grav, hist, OP_GRAVITY | OP_AVERAGE, F, head, win, 1, gl)
        head = (head + 1) % F
        for r in range(rows):
            want = x[r].copy()
            Oracle.lib().glvo_gl_chain_r16(want, ostore[r], ohist[r], C.byref(oheads[r]), n, F, int(win), 1, 4.2, 86.1328125)
            assert (bits(got[r]) == bits(want)).all(), (fr, r)
            if gl == 1:
                assert (hist[r] == Oracle.texels_r16(ohist[r])).all()
        assert got.min() >= 0.0 and got.max() <= 1.0
        assert np.allclose(got * 65535, np.round(got * 65535), atol=1e-2)      # every output is a texel value


def test_gravity_fixed_point_is_zero_not_minus_g(emu):
    n = 256
    loud = np.full((1, n), 0.8, np.float32)
    quiet = np.zeros((1, n), np.float32)
    for gl, fixed in ((1, 0.0), (2, 0.0), (0, -(np.float32(4.2) * (np.float32(1.0) / np.float32(86.1328125))))):
        grav = np.zeros((1, n), np.uint16 if gl == 1 else np.float32)
        out = emu_post(emu, loud, grav, None, OP_GRAVITY, 1, 0, True, 1, gl)
        for _ in range(40):
            out = emu_post(emu, quiet, grav, None, OP_GRAVITY, 1, 0, True, 1, gl)
        assert np.allclose(out, fixed, atol=1e-7), (gl, out[0, :3])


def test_one_averaging_pass_agrees_with_the_shader_evaluation(emu):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "glsl_vectors.npz"))
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_glsl_golden import tex_row
    F = 5
    frames = [tex_row(256, 100 + F * 10 + i) for i in range(F)]                 # index 0 = newest; values in [0, 1]
    frames_q = [Oracle.texels_r16(f).astype(np.float32) / np.float32(65535) for f in frames]
    hist = np.zeros((1, F, 256), np.uint16)
    head = 0
    out = None
    for f in reversed(frames_q):
        out = emu_post(emu, f.reshape(1, 256), None, hist, OP_AVERAGE, F, head, True, 1, 1)
        head = (head + 1) % F
    # the shader evaluation ran on the unquantised rows: compare on the texel grid, one step of slack
    want = gold["avg_F5_w1"]
    assert np.abs(np.round(out[0] * 65535) - np.round(np.clip(want, 0, 1) * 65535)).max() <= 1.0 + 65535 * 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,F", [(1024, 5), (4096, 5), (16384, 3)])
def test_device_gl_chain_equals_the_oracle_model(glvlib, n, F):
    """s16 PCM -> frame kernel -> GL-storage pass (gravity, ring, Hamming newest-first average) on the device; f32 and R16
    outputs; the checker is the oracle's transform_fft followed by glvo_gl_chain_r16.  With the bit-faithful log the chain is
    bit-exact; the GL_R16 output is the texel of the f32 output."""
    import torch
    G = glvlib
    streams = 5
    p = G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, log_mode=0)
    b = G.Batch(p, streams, G.OP_GRAVITY | G.OP_AVERAGE)
    bq = G.Batch(p, streams, G.OP_GRAVITY | G.OP_AVERAGE)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    d_q = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
    store = np.zeros((streams * 2, n), np.float32); hist = np.zeros((streams * 2, F, n), np.float32)
    heads = [C.c_size_t(0) for _ in range(streams * 2)]
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    for fr in range(F + 3):
        pcm = (lcg_pcm_fast(3100 + fr + n, streams * 2 * n) // 16).astype(np.int16)
        d_pcm = torch.from_numpy(pcm).cuda()
        b.process_s16(d_pcm, d_out, ops)
        bq.process_s16(d_pcm, d_q, ops | G.OP_R16)
        got = d_out.cpu().numpy(); gq = d_q.cpu().numpy().view(np.uint16)
        for u in range(streams):
            spec = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n])
            for c in range(2):
                want = np.ascontiguousarray(spec[c])
                Oracle.lib().glvo_gl_chain_r16(want, store[2 * u + c], hist[2 * u + c], C.byref(heads[2 * u + c]), n, F, 1, 1, 4.2, 86.1328125)
                assert (bits(got[2 * u + c]) == bits(want)).all(), (fr, u, c)
                assert (gq[2 * u + c] == Oracle.texels_r16(want)).all()
    b.close(); bq.close()
