"""smooth_audio()'s SHAPE (glv_params ABI 7): ROUND_FORMULA, SAMPLE_MODE, SAMPLE_HYBRID_WEIGHT, SAMPLE_SCALE, SAMPLE_RANGE -- the GLSL `#define`s of
shaders/glava/smooth_parameters.glsl:17-42 that a user's or a module's configuration re-defines (VERDICT r5 missing 7).

  * CPU: the oracle under a shape (glvo_set_smooth_shape, glvo_bars_mode_at) against an INDEPENDENT evaluation of the reference's shader text with a
    user's smooth_parameters.glsl in front of it (tests/glsl_eval.py; tests/golden/glsl_vectors.npz `shape_*`, generator committed), to a few
    float ulps (summation order, and the last place of log / sin);
  * CPU: the host shim reads the shape out of the processed shader text the way the patched host hands it over (integration/glava_hip_shim.c
    glv_hip_scan_shape, compiled into the reference-host harness);
  * GPU: GLV_OP_BARS under every shape, on float rows and inside the GL chains, against the oracle -- the averaging shapes through the kernels that
    already exist (chunked chains, matrix cores, the exact integer mean: only the tap tables change), maximum / hybrid through glv_bars_mode_kernel,
    bit for bit.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_glsl_golden import SHAPES, tex_row  # noqa: E402
from oracle_lib import Oracle, lcg_pcm_fast  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "glsl_vectors.npz"))
ULPS = 8 * 2.0 ** -24


def _oracle_bars(row, bars, factor, phase, shape, chunked=False):
    """the oracle's smooth_audio() of one float row under `shape` = (round_formula, sample_mode, hybrid_weight, scale, range)"""
    formula, mode, hw, scale, rng = shape
    with Oracle.smooth_shape(formula, scale, rng):
        if mode:
            return Oracle.bars_mode(row, bars, mode, hw or 0.65, factor, phase)
        out = np.empty(bars, np.float32)
        (Oracle.lib().glvo_bars_chunked_at if chunked else Oracle.lib().glvo_bars_at)(np.ascontiguousarray(row, np.float32), row.size, out, bars, factor, phase)
        return out


@pytest.mark.parametrize("case", SHAPES, ids=[s[0] for s in SHAPES])
def test_oracle_shapes_equal_the_shader_evaluation(case):
    key, user, shape, n, bars, factor, phase, seed = case
    want = GOLD[f"shape_{key}"]
    if os.path.exists("/root/reference/shaders/glava/util/smooth.glsl") and bars <= 100:      # (the vectors regenerate from the shader text where it exists)
        import glsl_eval as G
        again = G.smooth_audio_bars(tex_row(n, seed), bars, factor, user_parameters=user, phase=phase)
        assert (again.view(np.uint32) == want.view(np.uint32)).all()
    got = _oracle_bars(tex_row(n, seed), bars, factor, phase, shape)
    # (the evaluator rounds log / sin / sqrt correctly, glibc's float functions are within an ulp: the bounds and weights may differ in the last place)
    assert np.abs(got - want).max() <= ULPS * np.abs(want).max(), np.abs(got - want).max()
    assert np.abs(want).max() > 0.05                                                           # (a vector of zeros would prove nothing)


def test_the_shipped_shape_is_the_zero_shape():
    """0 in every glv_params shape field == the shipped defines: the oracle under (0, 8, 0.9) and under zeros gives the default's bits"""
    row = tex_row(2048, 3)
    a = _oracle_bars(row, 80, 0.025, 0.0, (0, 0, 0.0, 0.0, 0.0))
    b = _oracle_bars(row, 80, 0.025, 0.0, (0, 0, 0.0, 8.0, 0.9))
    c = np.empty(80, np.float32)
    Oracle.lib().glvo_bars(row, 2048, c, 80, 0.025)
    assert (a.view(np.uint32) == b.view(np.uint32)).all() and (a.view(np.uint32) == c.view(np.uint32)).all()


def _shim():
    so = os.path.join(ROOT, "oracle", "_ref", "libglvshim.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libglvshim.so not built (needs /root/reference at build time)")
    try:
        L = C.CDLL(so)
    except OSError as e:                                  # links libglvspectrum -> the HIP runtime: loadable wherever the product is
        pytest.skip(f"libglvshim.so does not load here: {e}")
    L.glv_hip_scan_shape.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.shim_shape.argtypes = [C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.shim_shape.restype = C.c_int
    return L


def _scan(L, text, path=b"util/smooth_pass.frag"):
    L.glv_hip_scan_shape(path, text, len(text))
    f, m = C.c_uint(9), C.c_uint(9)
    h, s, r = C.c_float(-1), C.c_float(-1), C.c_float(-1)
    ok = L.shim_shape(C.byref(f), C.byref(m), C.byref(h), C.byref(s), C.byref(r))
    return ok, f.value, m.value, round(h.value, 6), round(s.value, 6), round(r.value, 6)


def test_shim_reads_the_shape_from_the_processed_shader_text():
    """what shaderload() hands over after glsl_ext.c: every `#define X v` preceded by `#ifdef X / #undef X / #endif`, the user's file after the stock one"""
    L = _shim()
    def defs(**kw):
        return "".join(f"#ifdef {k}\n#undef {k}\n#endif\n#define {k} {v}\n" for k, v in kw.items())
    stock = defs(ROUND_FORMULA="sinusoidal", SAMPLE_MODE="average", SAMPLE_HYBRID_WEIGHT="0.65", SAMPLE_SCALE="8", SAMPLE_RANGE="0.9")
    tail = "\n#define average 0\n#define maximum 1\n#define hybrid 2\nfloat f(float x) { return x * (SAMPLE_RANGE) / (SAMPLE_SCALE); }\nvoid main() { }\n"      # (uses, not definitions)
    assert _scan(L, (stock + tail).encode()) == (1, 0, 0, 0.65, 8.0, 0.9)
    user = defs(ROUND_FORMULA="circular", SAMPLE_MODE="hybrid", SAMPLE_HYBRID_WEIGHT=".4 /* mine */", SAMPLE_SCALE="6.0f", SAMPLE_RANGE="(0.8) // narrower")
    assert _scan(L, (stock + user + tail).encode()) == (1, 1, 2, 0.4, 6.0, 0.8)               # the LAST definition is the one the compiler keeps
    assert _scan(L, (user + stock + tail).encode()) == (1, 0, 0, 0.65, 8.0, 0.9)
    assert _scan(L, (stock + defs(SAMPLE_MODE="maximum", ROUND_FORMULA="linear") + tail).encode()) == (1, 2, 1, 0.65, 8.0, 0.9)
    # another shader's text changes nothing
    assert L.glv_hip_scan_shape(b"bars/1.frag", (stock + defs(SAMPLE_MODE="hybrid")).encode(), 10) is not None
    assert _scan(L, stock.encode(), path=b"bars/1.frag")[0:3] == (1, 2, 1)
    # what the scan cannot read keeps the GL passes on the GL: an expression, an unknown formula, a range the library refuses
    assert _scan(L, (stock + defs(SAMPLE_SCALE="(4 + 4)") + tail).encode())[0] == 0
    assert _scan(L, (stock + defs(ROUND_FORMULA="mycurve") + tail).encode())[0] == 0
    assert _scan(L, (stock + defs(SAMPLE_RANGE="1.0") + tail).encode())[0] == 0
    assert _scan(L, (stock + defs(SAMPLE_SCALE="1", SAMPLE_RANGE="0.9") + tail).encode())[0] == 0     # -log(0.1) / 1 = 2.3 rows
    assert _scan(L, (stock + tail).encode())[0] == 1                                                 # ... and a readable text restores it


# ----------------------------------------------------------------------------------------------------------------------------------- GPU
FLOAT_CASES = [  # n, bars, factor, phase, rows, shape
    (1024, 80, 0.025, 0.0, 5, (0, 1, 0.0, 0.0, 0.0)), (4096, 80, 0.025, 0.0, 9, (0, 2, 0.0, 0.0, 0.0)), (2048, 64, 0.05, 0.0, 4, (2, 2, 0.4, 0.0, 0.0)),
    (4096, 80, 0.025, 0.0, 6, (1, 0, 0.0, 6.0, 0.8)), (1024, 100, 0.01, 0.0, 3, (2, 0, 0.0, 4.0, 0.95)), (16384, 80, 0.025, 0.0, 3, (1, 0, 0.0, 0.0, 0.0)),
    (512, 512, 0.025, 0.5, 70, (1, 1, 0.0, 0.0, 0.0)), (4096, 4096, 0.025, 0.5, 131, (0, 2, 0.0, 0.0, 0.0)), (4096, 4096, 0.025, 0.5, 67, (2, 0, 0.0, 6.0, 0.85)),
    (16384, 16384, 0.01, 0.5, 5, (0, 1, 0.0, 0.0, 0.0)), (32768, 300, 0.025, 0.0, 3, (0, 2, 0.9, 5.0, 0.99)), (1024, 1024, 0.05, 0.5, 2, (1, 2, 1.0, 0.0, 0.0)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("n,bars,factor,phase,rows,shape", FLOAT_CASES)
def test_device_bars_of_float_rows_under_every_shape(glvlib, n, bars, factor, phase, rows, shape):
    """glv_batch_bars: the averaging shapes in the library's documented orders (glvo_bars_chunked_at under the shape), maximum / hybrid the shader's loop
    (glvo_bars_mode_at) -- bit for bit, as floats and as GL_R16 texels; inputs outside [0, 1] and NaN are clamped like texels"""
    import torch
    G = glvlib
    formula, mode, hw, scale, rng = shape
    p = G.Params(n=n, bars=bars, smooth_factor=factor, bar_phase=phase, round_formula=formula, sample_mode=mode, sample_hybrid_weight=hw,
                 sample_scale=scale, sample_range=rng)
    b = G.Batch(p, (rows + 1) // 2, G.OP_FFT | G.OP_BARS)
    assert b.bars_arithmetic() == (G.BARS_F32_SEQ if mode else (G.BARS_F32_CHAIN if bars < 256 else G.BARS_F32_MATRIX))
    nrows = 2 * ((rows + 1) // 2)
    spec = np.stack([tex_row(n, 900 + r) for r in range(nrows)])
    spec[0, :7] = [-0.5, 1.5, np.nan, np.inf, -np.inf, 0.0, 1.0]
    d_spec = torch.from_numpy(spec).cuda()
    d_bars = torch.full((nrows, bars), -1.0, dtype=torch.float32, device="cuda")
    b.bars(d_spec, d_bars)
    got = d_bars.cpu().numpy()
    b.close()
    clean = np.nan_to_num(np.clip(spec, 0, 1), nan=0.0)                                        # what the oracle's own clamp does, made explicit for the chunked form
    for r in sorted(set([0, 1, nrows - 1, nrows // 2])):
        want = _oracle_bars(clean[r], bars, factor, phase, shape, chunked=True)
        same = (got[r].view(np.uint32) == want.view(np.uint32)) | (np.isnan(got[r]) & np.isnan(want))
        assert same.all(), (r, int((~same).sum()), np.flatnonzero(~same)[:5], got[r][~same][:3], want[~same][:3])


GL_CASES = [  # n, F, factor, streams, shape
    (1024, 3, 0.025, 37, (0, 1, 0.0, 0.0, 0.0)), (4096, 5, 0.025, 70, (0, 2, 0.0, 0.0, 0.0)), (4096, 5, 0.025, 33, (1, 0, 0.0, 6.0, 0.8)),
    (2048, 2, 0.05, 9, (2, 2, 0.3, 0.0, 0.0)), (4096, 5, 0.01, 5, (2, 0, 0.0, 0.0, 0.0)), (8192, 4, 0.025, 3, (1, 1, 0.0, 7.0, 0.9)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("bars_only", [False, True])
@pytest.mark.parametrize("n,F,factor,streams,shape", GL_CASES)
def test_gl_chain_presmoothing_pass_under_every_shape(glvlib, n, F, factor, streams, shape, bars_only):
    """GLava's shipped pipeline with a user's smoothing shape: the `sm` texels of the chain (gl_storage 1, bars = n, bar_phase 0.5) against the oracle's
    smooth_audio() of the SAME chain's `av` texels -- the exact integer mean under the shape for the averaging modes (glvo_bars_int_at), the shader's float
    loop on c / 65535 for maximum / hybrid (glvo_bars_mode_at), then the GL_R16 store.  With and without GLV_OP_BARS_ONLY."""
    import torch
    G = glvlib
    formula, mode, hw, scale, rng = shape
    kw = dict(n=n, avg_frames=F, avg_window=True, avg_window_kind=1, gl_storage=1, smooth_factor=factor, round_formula=formula, sample_mode=mode,
              sample_hybrid_weight=hw, sample_scale=scale, sample_range=rng)
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    av = G.Batch(G.Params(bars=80, **kw), streams, G.OP_GRAVITY | G.OP_AVERAGE)
    sm = G.Batch(G.Params(bars=n, bar_phase=0.5, **kw), streams, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | (G.OP_BARS_ONLY if bars_only else 0))
    assert sm.bars_arithmetic() == (G.BARS_F32_SEQ if mode else G.BARS_I8_EXACT)
    o_av = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
    o_sm = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
    o_smf = torch.zeros((streams * 2, n), dtype=torch.float32, device="cuda")
    for fr in range(F + 2):
        pcm = (lcg_pcm_fast(7700 + fr + n, streams * 2 * n) // (1, 16, 4)[fr % 3]).astype(np.int16)
        d_pcm = torch.from_numpy(pcm).cuda()
        av.process_s16(d_pcm, o_av, ops | G.OP_R16)
        sm.process_s16(d_pcm, o_sm, ops | G.OP_BARS | G.OP_R16)
        assert sm.last_launches() == 2
    torch.cuda.synchronize()
    a = o_av.cpu().numpy().view(np.uint16)
    s = o_sm.cpu().numpy().view(np.uint16)
    for r in sorted(set([0, 1, streams, 2 * streams - 1])):
        if mode:
            want = Oracle.texels_r16(_oracle_bars(a[r].astype(np.float32) / np.float32(65535), n, factor, 0.5, shape))
        else:
            with Oracle.smooth_shape(formula, scale, rng):
                want, _ = Oracle.bars_int(a[r], n, factor, 0.5)
        assert (s[r] == want).all(), (r, int((s[r] != want).sum()), np.flatnonzero(s[r] != want)[:5])
    assert int(s.max()) > 1000
    # the same bars as floats (maximum / hybrid: the float itself; averaging: the integer mean's float form is covered by tests/test_gl_fused.py)
    if mode and not bars_only:
        sm.reset(); av.reset()
        pcm = lcg_pcm_fast(31, streams * 2 * n).astype(np.int16)
        d_pcm = torch.from_numpy(pcm).cuda()
        av.process_s16(d_pcm, o_av, ops | G.OP_R16)
        sm.process_s16(d_pcm, o_smf, ops | G.OP_BARS)
        a = o_av.cpu().numpy().view(np.uint16)
        want = _oracle_bars(a[1].astype(np.float32) / np.float32(65535), n, factor, 0.5, shape)
        assert (o_smf[1].cpu().numpy().view(np.uint32) == want.view(np.uint32)).all()
    av.close(); sm.close()


@pytest.mark.gpu
def test_fused_bars_and_live_classes_follow_the_shape(glvlib):
    """80 bars of the modules under an averaging shape stay inside the transform's launch (tables only); maximum / hybrid leave it (two launches) and give
    the oracle's bits on the chain's own spectra; a shape that samples past the live share switches a GLV_OP_BARS_ONLY batch to the full chain"""
    import torch
    G = glvlib
    n, streams, F = 4096, 6, 3
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    pcm = [torch.from_numpy(lcg_pcm_fast(50 + i, streams * 2 * n).astype(np.int16)).cuda() for i in range(F + 1)]
    for shape, launches in [((1, 0, 0.0, 6.0, 0.8), 1), ((0, 1, 0.0, 0.0, 0.0), 2), ((2, 2, 0.5, 0.0, 0.0), 2)]:
        formula, mode, hw, scale, rng = shape
        kw = dict(n=n, avg_frames=F, bars=80, round_formula=formula, sample_mode=mode, sample_hybrid_weight=hw, sample_scale=scale, sample_range=rng)
        full = G.Batch(G.Params(**kw), streams, G.OP_GRAVITY | G.OP_AVERAGE)
        bars = G.Batch(G.Params(**kw), streams, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS)
        o_full = torch.zeros((streams * 2, n), dtype=torch.float32, device="cuda")
        o_bars = torch.zeros((streams * 2, 80), dtype=torch.float32, device="cuda")
        for d in pcm:
            full.process_s16(d, o_full, ops)
            bars.process_s16(d, o_bars, ops | G.OP_BARS)
            assert bars.last_launches() == launches, (shape, bars.last_launches())
        torch.cuda.synchronize()
        spec, got = o_full.cpu().numpy(), o_bars.cpu().numpy()
        for r in (0, 5, 2 * streams - 1):
            want = _oracle_bars(np.nan_to_num(np.clip(spec[r], 0, 1), nan=0.0), 80, 0.025, 0.0, shape, chunked=True)
            assert (got[r].view(np.uint32) == want.view(np.uint32)).all(), (shape, r)
        full.close(); bars.close()
    # live bins: the shipped shape samples 0.288 n; SAMPLE_RANGE 0.99 / SAMPLE_SCALE 5 samples 0.92 n -- more than any live class keeps
    kw = dict(n=n, avg_frames=5, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)
    m = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_BARS_ONLY
    a = G.Batch(G.Params(**kw), 4, m)
    b = G.Batch(G.Params(sample_scale=5.0, sample_range=0.99, **kw), 4, m)
    c = G.Batch(G.Params(sample_scale=16.0, **kw), 4, m)
    assert a.live_bins() == 1216 and b.live_bins() == 0 and 0 < c.live_bins() < 1216
    for x in (a, b, c): x.close()
    # shapes the shader itself could not run are refused when the tables are made
    for bad in (dict(sample_range=1.0), dict(sample_scale=1.0), dict(sample_scale=-8.0), dict(sample_hybrid_weight=1.5, sample_mode=2), dict(round_formula=3), dict(sample_mode=3)):
        with pytest.raises(G.GlvError) as e:
            G.Batch(G.Params(n=1024, bars=80, **bad), 1, G.OP_FFT | G.OP_BARS)
        assert e.value.code == G.ERR_INVALID, bad


@pytest.mark.gpu
def test_shapes_over_random_parameters(glvlib):
    """60 seeded draws of (n, bars, smooth_factor, bar_phase, rows, formula, mode, hybrid weight, scale, range): glv_batch_bars on float rows against the oracle
    under the same shape, bit for bit -- every arithmetic the library has for bars (chunked chains below 256 bars, one chain per bar on the matrix cores or one
    lane per bar above, the mode kernel with 8 / 4 / 1 rows per lane and through L1), ragged last blocks, bars that are not a multiple of 64, taps that reach
    the row's last bins.  A shape the library refuses (positions past the row) must be refused with GLV_ERR_INVALID, not computed."""
    import torch
    G = glvlib
    rng = np.random.default_rng(20260930)
    done = refused = 0
    for trial in range(60):
        n = int(rng.choice([256, 512, 1024, 2048, 4096, 8192]))
        bars = int(rng.choice([1, 7, 64, 65, 80, 200, 255, 256, 300, n // 2, n]))
        bars = min(bars, n)
        factor = float(rng.choice([0.005, 0.01, 0.025, 0.05, 0.1]))
        phase = float(rng.choice([0.0, 0.5, 0.25]))
        rows = int(rng.choice([2, 4, 6, 10, 34, 70]))
        formula, mode = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        hw = float(rng.choice([0.0, 0.2, 0.65, 1.0]))
        scale = float(rng.choice([0.0, 3.0, 5.0, 8.0, 12.0]))
        rg = float(rng.choice([0.0, 0.5, 0.8, 0.9, 0.95, 0.99]))
        shape = (formula, mode, hw, scale, rg)
        p = G.Params(n=n, bars=bars, smooth_factor=factor, bar_phase=phase, round_formula=formula, sample_mode=mode, sample_hybrid_weight=hw, sample_scale=scale, sample_range=rg)
        sc, r_ = (scale or 8.0), (rg or 0.9)
        if -np.log(np.float32(1.0) - np.float32(r_)) / np.float32(sc) > 1.0:
            with pytest.raises(G.GlvError) as e:
                G.Batch(p, rows // 2, G.OP_FFT | G.OP_BARS)
            assert e.value.code == G.ERR_INVALID
            refused += 1
            continue
        try:
            b = G.Batch(p, rows // 2, G.OP_FFT | G.OP_BARS)
        except G.GlvError as e:                                  # a tap chunk past the row's end (large factors at the top of the range): refused, never computed
            assert e.code == G.ERR_INVALID and "leave the row" in str(e), (trial, shape, str(e))
            refused += 1
            continue
        spec = np.stack([tex_row(n, 5000 + 97 * trial + r) for r in range(rows)])
        d_bars = torch.full((rows, bars), -1.0, dtype=torch.float32, device="cuda")
        b.bars(torch.from_numpy(spec).cuda(), d_bars)
        got = d_bars.cpu().numpy()
        b.close()
        for r in (0, rows - 1):
            want = _oracle_bars(spec[r], bars, factor, phase, shape, chunked=True)
            same = (got[r].view(np.uint32) == want.view(np.uint32)) | (np.isnan(got[r]) & np.isnan(want))
            assert same.all(), (trial, n, bars, factor, phase, rows, shape, r, int((~same).sum()), np.flatnonzero(~same)[:4], got[r][~same][:3], want[~same][:3])
        done += 1
    assert done >= 40 and refused >= 1, (done, refused)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(0, 1, 0.0, 0.0, 0.0), (1, 2, 0.5, 6.0, 0.85)])
@pytest.mark.parametrize("bars_only", [False, True])
def test_gl_chain_with_the_modules_bars_under_maximum_and_hybrid(glvlib, shape, bars_only):
    """the GL_R16 chain + the 80 bars of the bars / radial modules (fewer than 256 bars: fused into the transform for the averaging modes) under SAMPLE_MODE maximum /
    hybrid: two launches, the shader's loop on the chain's own `av` texels as floats c / 65535, as GL_R16 texels and as floats"""
    import torch
    G = glvlib
    n, F, streams, bars = 4096, 5, 11, 80
    formula, mode, hw, scale, rng = shape
    kw = dict(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=bars, round_formula=formula, sample_mode=mode, sample_hybrid_weight=hw, sample_scale=scale, sample_range=rng)
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    av = G.Batch(G.Params(**kw), streams, G.OP_GRAVITY | G.OP_AVERAGE)
    bt = G.Batch(G.Params(**kw), streams, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | (G.OP_BARS_ONLY if bars_only else 0))
    bf = G.Batch(G.Params(**kw), streams, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS)
    o_av = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
    o_t = torch.zeros((streams * 2, bars), dtype=torch.int16, device="cuda")
    o_f = torch.zeros((streams * 2, bars), dtype=torch.float32, device="cuda")
    for fr in range(F + 1):
        d_pcm = torch.from_numpy((lcg_pcm_fast(9100 + fr, streams * 2 * n) // (2, 32)[fr % 2]).astype(np.int16)).cuda()
        av.process_s16(d_pcm, o_av, ops | G.OP_R16)
        bt.process_s16(d_pcm, o_t, ops | G.OP_BARS | G.OP_R16)
        bf.process_s16(d_pcm, o_f, ops | G.OP_BARS)
        assert bt.last_launches() == 2 and bf.last_launches() == 2
    a = o_av.cpu().numpy().view(np.uint16); t = o_t.cpu().numpy().view(np.uint16); f = o_f.cpu().numpy()
    for r in (0, 7, 2 * streams - 1):
        want = _oracle_bars(a[r].astype(np.float32) / np.float32(65535), bars, 0.025, 0.0, shape)
        assert (f[r].view(np.uint32) == want.view(np.uint32)).all(), (r, shape)
        assert (t[r] == Oracle.texels_r16(want)).all(), (r, shape)
    assert int(t.max()) > 1000
    for b in (av, bt, bf): b.close()
