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

Comparisons are TIE-AWARE: every pass's output texel is the rounding of a real number that can be computed
exactly from the pass's integer inputs -- the upload x * 65535, the average sum_I w_I c_I / F of the ring's integer texels, the
weighted mean of smooth_audio()'s taps.  A float implementation may land on either neighbour only where that real number lies
within its own arithmetic error of a half-integer (a genuine tie: which way it goes is the implementation's business, OpenGL 4.6
section 2.3.5); everywhere else the texel is determined, and EQUALITY is demanded -- of the reference's llvmpipe texels, of the
oracle's, of the MI355X's.  The one thing no arithmetic decides is smooth_audio()'s tap SET where a bound computed through log()
sits within a few ulps of admitting one more tap (glvo_bars_at_exact flags those bars, ~0.5 % at n = 4096): they are excluded
from the claim, and nothing else is.  No mismatch fractions, no outlier budgets."""
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
# `#request setsmoothfactor` cases (round 6): gl_data.smooth_factor reaches the pass as the header's `#define _SMOOTH_FACTOR %.6f` (render.c:317-326)
FACTOR_CASES = [("n1024_F5w_sf010", 1024, 5, True), ("n1024_F5w_sf050", 1024, 5, True), ("n4096_F5w_sf050", 4096, 5, True), ("n2048_F3w_sf010", 2048, 3, True)]
# a user's smooth_parameters.glsl with the smoothing SHAPE re-defined (round 6, VERDICT r5 missing 7): name -> (the `#define`s of the generator's
# configuration directory, glv_params' (round_formula, sample_mode, sample_hybrid_weight, sample_scale, sample_range))
SHAPE_CASES = [("n1024_F5w_maximum", 1024, 5, True), ("n1024_F5w_hybrid", 1024, 5, True), ("n4096_F5w_circular_s6_r80", 4096, 5, True), ("n2048_F3w_linear_hybrid40", 2048, 3, True)]
SHAPES = {"n1024_F5w_maximum": ({"SAMPLE_MODE": "maximum"}, (0, 1, 0.0, 0.0, 0.0)), "n1024_F5w_hybrid": ({"SAMPLE_MODE": "hybrid"}, (0, 2, 0.0, 0.0, 0.0)),
          "n4096_F5w_circular_s6_r80": ({"ROUND_FORMULA": "circular", "SAMPLE_SCALE": "6", "SAMPLE_RANGE": "0.8"}, (1, 0, 0.0, 6.0, 0.8)),
          "n2048_F3w_linear_hybrid40": ({"ROUND_FORMULA": "linear", "SAMPLE_MODE": "hybrid", "SAMPLE_HYBRID_WEIGHT": "0.4"}, (2, 2, 0.4, 0.0, 0.0))}
UP, GR, AV, SM = 0, 1, 2, 3


def shape_of(name):
    return SHAPES[name][1] if name in SHAPES else (0, 0, 0.0, 0.0, 0.0)


import contextlib


@contextlib.contextmanager
def shape_ctx(name):
    """every oracle evaluation of smooth_audio() inside the block runs under the case's shape (the float functions and the float64 brackets alike);
    for a ROUND_FORMULA other than the shipped one the brackets carry what the weight function makes of a float argument (WEIGHT_SLACK)"""
    formula, mode, hw, scale, rng = shape_of(name)
    with Oracle.smooth_shape(formula, scale, rng, mode, hw):
        if formula != 0:
            n = next(c[1] for c in SHAPE_CASES if c[0] == name)
            rel = np.empty(n, np.float64); ab = np.empty(n, np.float64)
            Oracle.lib().glvo_bars_weight_slack(n, rel, ab, n, factor_of(name), 0.5)
            WEIGHT_SLACK["rel"], WEIGHT_SLACK["abs"] = rel, ab * (1.0 if mode == 1 else (1.0 - (hw or 0.65)) if mode == 2 else 0.0)   # the maximum's share
        try:
            yield
        finally:
            WEIGHT_SLACK["rel"] = WEIGHT_SLACK["abs"] = None


def shape_params(name):
    formula, mode, hw, scale, rng = shape_of(name)
    return dict(round_formula=formula, sample_mode=mode, sample_hybrid_weight=hw, sample_scale=scale, sample_range=rng)


def oracle_smooth_texels(av_texels, n, factor, name):
    """the oracle's float form of the pass on one row of `av` texels under the case's shape -> texels"""
    formula, mode, hw, scale, rng = shape_of(name)
    if mode:
        return Oracle.texels_r16(Oracle.bars_mode(texel_float(av_texels), n, mode, hw or 0.65, factor, 0.5))
    sm = np.empty(n, np.float32)
    Oracle.lib().glvo_bars_at(texel_float(av_texels), n, sm, n, factor, 0.5)
    return Oracle.texels_r16(sm)


def factor_of(name):
    """the float the shader computes with: the golden's recorded request through the "%.6f" header text, read back as a float literal"""
    key = name + "_factor"
    return float(np.float32(float("%.6f" % float(GOLD[key])))) if key in GOLD.files else 0.025


def texel_float(t):
    return (t.astype(np.float32) / np.float32(65535)).copy()        # what a shader reads back: c / 65535 (OpenGL 4.6 eq. 2.1), correctly rounded


U = 2.0 ** -24          # unit roundoff of the float arithmetic the passes run in


def gl_weights(F, win):
    """average_pass.frag's weights by AGE (0 = oldest): window(I, ...) of common.glsl:13 with I = F - 1 - age; none for two frames"""
    return np.array([1.0 if (not win or F == 2) else 0.53836 - (0.46164 * np.cos(6.28318530718 * (F - 1 - a) / F - 1)) for a in range(F)])


def assert_tie_aware(got, exact, delta, what, skip=None):
    """got: integer texels; exact: the real value in texel units; delta: the arithmetic error an implementation may have there.
    Outside |frac(exact) - 0.5| <= delta the texel must be rint(exact); inside, one of the two neighbours."""
    got = got.astype(np.int64)
    exact = np.clip(exact, 0.0, 65535.0)
    lo = np.floor(exact)
    near = np.abs(exact - lo - 0.5) <= delta
    keep = np.ones(got.shape, bool) if skip is None else ~skip
    far_bad = (got != np.rint(exact)) & ~near & keep
    near_bad = (got != lo) & (got != lo + 1) & near & keep
    assert not far_bad.any() and not near_bad.any(), (what, int(far_bad.sum()), int(near_bad.sum()), np.flatnonzero(far_bad | near_bad)[:5])
    return int((near & keep).sum()), int(((got != np.rint(exact)) & keep).sum())


def exact_upload(spec):
    return np.clip(spec.astype(np.float64), 0.0, 1.0) * 65535.0


def exact_average(ring, F, win):
    """ring: F integer texel rows, oldest first"""
    w = gl_weights(F, win)
    return sum(w[a] * ring[a].astype(np.float64) for a in range(F)) / F


def exact_smooth(av_texels, n, factor=0.025):
    exact = np.empty(n, np.float64); nt = np.empty(n, np.int32); frag = np.empty(n, np.int32)
    Oracle.lib().glvo_bars_at_exact(texel_float(av_texels), n, exact, nt, frag, n, factor, 0.5, 4)
    return exact * 65535.0, nt, frag.astype(bool)


def d_upload(ex): return U * np.maximum(ex, 1.0) + 1e-9                      # one rounding (Mesa multiplies in float before converting)
def d_average(ex, F): return (F + 8) * U * np.maximum(ex, 1.0)               # F products, F sums, the quotient, the folded weights
# worst case of an nt-term float sum (numerator and weight sum) + the weights themselves: they are sin() values, GLSL promises no
# accuracy for sin() and Mesa's llvmpipe evaluates it with a polynomial good to ~2^-20, which a weighted mean inherits
# ... under a user's ROUND_FORMULA the weight function itself can amplify the float error of its argument (circular: an infinite slope at the outermost
# taps): WEIGHT_SLACK holds glvo_bars_weight_slack's per-bar bounds for the case being checked (None for the shipped sinusoidal: flat there, and
# covered by the 2^-18 above) -- `k`: the bar a scalar evaluation belongs to
WEIGHT_SLACK = {"rel": None, "abs": None}
def d_smooth(ex, nt, k=None):
    d = ((nt + 16) * U + 2.0 ** -18) * np.maximum(ex, 1.0)
    if WEIGHT_SLACK["rel"] is not None:
        rel, ab = (WEIGHT_SLACK["rel"], WEIGHT_SLACK["abs"]) if k is None else (WEIGHT_SLACK["rel"][k], WEIGHT_SLACK["abs"][k])
        d = d + rel * np.maximum(ex, 1.0) + ab * 65535.0
    return d


# the tap sets a fragile bar may walk: the loop's bounds up to 4 ulps away (another log()), round() at an exact .5 away from zero or to even
CANDIDATES = [(a, b, he) for he in (0, 1) for a in (0, -1, 1, -2, 2, -3, 3, -4, 4) for b in (0, -1, 1, -2, 2, -3, 3, -4, 4)]


def texel_ok(got, ex, delta):
    """one texel against the exact value of its pass (texel units): the rounded value, or either neighbour inside the tie zone"""
    ex = min(max(float(ex), 0.0), 65535.0)
    lo = np.floor(ex)
    return int(got) in (int(lo), int(lo) + 1) if abs(ex - lo - 0.5) <= delta else int(got) == int(np.rint(ex))


# What GLSL implementations are held to for log() inside [0.5, 2]: an absolute error of 2^-21 (the SPIR-V / Vulkan precision table; OpenGL
# promises no more).  smooth_audio()'s loop bounds are n * (-log(1 - 0.9 u) / 8): with such a log either bound may sit n * 2^-21 / 8 away
# from the correctly rounded one, and at the lowest bars of a SMALL smooth factor (three taps, unequal texels) that moves the weighted
# mean by more than the float arithmetic does -- Mesa's llvmpipe (a polynomial log2, least accurate just below 1) shows it: golden
# n1024_F5w_sf010, frame 5, right channel, bar 13.  glvo_bars_range_exact gives the range of the exact mean over that box of bounds.
LOG_ABS = 2.0 ** -21


def smooth_range(av_lo, av_hi, n, factor, log_abs=LOG_ABS):
    """lowest / highest admissible texel of every bar for an implementation whose log() is good to log_abs (and whose sums carry d_smooth)"""
    vmin = np.empty(n, np.float64); vmax = np.empty(n, np.float64); nt = np.empty(n, np.int32)
    Oracle.lib().glvo_bars_range_exact(texel_float(av_lo), texel_float(av_hi), n, vmin, vmax, nt, n, factor, 0.5, log_abs)
    vmin = np.clip(vmin * 65535.0, 0.0, 65535.0); vmax = np.clip(vmax * 65535.0, 0.0, 65535.0)
    lo = np.where(np.abs(vmin - np.floor(vmin) - 0.5) <= d_smooth(vmin, nt), np.floor(vmin), np.rint(vmin))
    hi = np.where(np.abs(vmax - np.floor(vmax) - 0.5) <= d_smooth(vmax, nt), np.floor(vmax) + 1, np.rint(vmax))
    return np.clip(lo, 0, 65535).astype(np.int64), np.clip(hi, 0, 65535).astype(np.int64)


def smooth_admissible(got, av_texels, n, what, factor=0.025, log_abs=0.0):
    """The pre-smoothing pass, TAP-SET-AWARE: every bar is held to the rounded exact value (tie-aware); a bar
    whose tap SET hangs on the last bits of scale_audio()'s log() (glvo_bars_at_exact flags it) is held to the exact value of ONE of the tap
    sets an implementation with bounds up to 4 ulps away, and with either direction of round() at an exact .5, would walk
    (glvo_bars_one_exact: the shader's own float loop from the moved bounds) -- no bar is excluded.  That is the standard for an
    implementation with a correctly rounded log() (the oracle, the library: log_abs = 0).  For a GLSL implementation (the reference's
    texels off llvmpipe) log_abs = LOG_ABS: a texel that misses that standard must lie inside the range the log's admissible error opens
    (smooth_range).  Returns (texels inside a tie zone, texels off the nearest rounding, fragile bars, texels only the log's error admits)."""
    ex, nt, frag = exact_smooth(av_texels, n, factor)
    g = got.astype(np.int64)
    exc = np.clip(ex, 0.0, 65535.0)
    lo = np.floor(exc); delta = d_smooth(ex, nt)
    near = np.abs(exc - lo - 0.5) <= delta
    bad = np.where(near, (g != lo) & (g != lo + 1), g != np.rint(exc))
    texf = texel_float(av_texels)
    e, c = C.c_double(0), C.c_int(0)
    for k in np.flatnonzero(frag):
        ok = False
        for dmin, dmax, he in CANDIDATES:
            Oracle.lib().glvo_bars_one_exact(texf, n, int(k), n, factor, 0.5, dmin, dmax, he, C.byref(e), C.byref(c))
            if texel_ok(got[k], e.value * 65535.0, float(d_smooth(e.value * 65535.0, c.value, int(k)))):
                ok = True
                break
        bad[k] = not ok
    wide = 0
    if bad.any() and log_abs > 0:
        rlo, rhi = smooth_range(av_texels, av_texels, n, factor, log_abs)
        still = bad & ((g < rlo) | (g > rhi))
        wide = int(bad.sum()) - int(still.sum())
        bad = still
    assert not bad.any(), (what, "no admissible tap set / bound gives these texels", np.flatnonzero(bad)[:5], g[bad][:5], ex[bad][:5])
    keep = ~frag
    return int((near & keep).sum()), int(((g != np.rint(exc)) & keep).sum()), int(frag.sum()), wide


def smooth_bounds(av_lo, av_hi, n, factor=0.025, log_abs=0.0):
    """lowest / highest admissible `sm` texel of every bar when the `av` texels may lie anywhere in [av_lo, av_hi] (the weights are >= 0:
    the mean is monotone in every tap), every tie may go either way and a fragile bar may walk any of its admissible tap sets; log_abs > 0:
    and the loop's bounds may carry a GLSL log()'s admissible error (smooth_range) -- the bracket for the reference's llvmpipe texels"""
    exl, ntl, frag = exact_smooth(av_lo, n, factor)
    exh, nth, frag_h = exact_smooth(av_hi, n, factor)
    frag = frag | frag_h
    lo = np.where(np.abs(exl - np.floor(exl) - 0.5) <= d_smooth(exl, ntl), np.floor(exl), np.rint(exl))
    hi = np.where(np.abs(exh - np.floor(exh) - 0.5) <= d_smooth(exh, nth), np.floor(exh) + 1, np.rint(exh))
    e, c = C.c_double(0), C.c_int(0)
    fl, fh = texel_float(av_lo), texel_float(av_hi)
    for k in np.flatnonzero(frag):
        for dmin, dmax, he in CANDIDATES:
            Oracle.lib().glvo_bars_one_exact(fl, n, int(k), n, factor, 0.5, dmin, dmax, he, C.byref(e), C.byref(c))
            v = e.value * 65535.0; dl = float(d_smooth(v, c.value, int(k)))
            lo[k] = min(lo[k], np.floor(v) if abs(v - np.floor(v) - 0.5) <= dl else np.rint(v))
            Oracle.lib().glvo_bars_one_exact(fh, n, int(k), n, factor, 0.5, dmin, dmax, he, C.byref(e), C.byref(c))
            v = e.value * 65535.0; dl = float(d_smooth(v, c.value, int(k)))
            hi[k] = max(hi[k], np.floor(v) + 1 if abs(v - np.floor(v) - 0.5) <= dl else np.rint(v))
    lo = np.clip(lo, 0, 65535).astype(np.int64); hi = np.clip(hi, 0, 65535).astype(np.int64)
    if log_abs > 0:
        rlo, rhi = smooth_range(av_lo, av_hi, n, factor, log_abs)
        lo = np.minimum(lo, rlo); hi = np.maximum(hi, rhi)
    return lo, hi


class ChainBounds:
    """END TO END without a +-1: the admissible range of every texel of the chain upload -> GL_MAX store + gravity ->
    ring -> average when every genuine tie may go either way.  All of these passes are monotone in their inputs (max, a subtraction of a
    constant, an average with positive weights), so two runs of the exact model -- every ambiguous texel rounded down in one, up in the
    other, each with its own gravity store and ring across the frames -- bracket whatever an implementation can produce: where no tie is in
    play the two coincide and EQUALITY is demanded."""
    def __init__(self, n, F, win):
        self.n, self.F, self.win = n, F, win
        self.store = [np.zeros(n, np.float32), np.zeros(n, np.float32)]
        self.ring = [[np.zeros(n, np.int64) for _ in range(F)] for _ in range(2)]

    def frame(self, spec):
        ex = exact_upload(spec)
        tie = np.abs(ex - np.floor(ex) - 0.5) <= d_upload(ex)
        out = []
        for side in (0, 1):
            up = np.where(tie, np.floor(ex) + side, np.rint(ex)).astype(np.uint16)
            row = texel_float(up)
            Oracle.lib().glvo_gl_chain_r16(row, self.store[side], row, C.byref(C.c_size_t(0)), self.n, self.F, int(self.win), 0, 4.2, UR)
            gr = Oracle.texels_r16(row).astype(np.int64)
            self.ring[side] = self.ring[side][1:] + [gr]
            if self.F > 1:
                exa = exact_average(self.ring[side], self.F, self.win)
                t = np.abs(exa - np.floor(exa) - 0.5) <= d_average(exa, self.F)
                out.append(np.clip(np.where(t, np.floor(exa) + side, np.rint(exa)), 0, 65535).astype(np.int64))
            else:
                out.append(gr)
        return out[0], out[1]


@pytest.mark.parametrize("name,n,F,win", CASES + FACTOR_CASES + SHAPE_CASES)
def test_gl_passes_tie_aware_reference_and_oracle(name, n, F, win):
    with shape_ctx(name):
        _gl_passes_tie_aware_reference_and_oracle(name, n, F, win)


def _gl_passes_tie_aware_reference_and_oracle(name, n, F, win):
    """Pass by pass, each fed with the reference's own input texels: the REFERENCE's llvmpipe texels and the ORACLE's restatement
    against the exact value of the pass -- gravity store exact; upload, average and pre-smoothing pass equal to the rounded exact
    value except at genuine ties (either neighbour); the smooth pass tap-set-aware (smooth_admissible: no bar excluded), also for the
    library's integer form of it; and END TO END from the PCM the reference's av / sm texels lie inside the admissible range of the exact
    model (ChainBounds / smooth_bounds: equality wherever no tie is in play)."""
    pcm, tex = GOLD[name + "_pcm"], GOLD[name + "_tex"]
    factor = factor_of(name)
    glsl_log = name + "_factor" in GOLD.files or name in SHAPES
    averaging = shape_of(name)[1] == 0
    store = np.zeros((2, n), np.float32); hist = np.zeros((2, F, n), np.float32)
    heads = [C.c_size_t(0), C.c_size_t(0)]
    ring = [[np.zeros(n, np.int64) for _ in range(F)] for _ in range(2)]
    bounds = [ChainBounds(n, F, win), ChainBounds(n, F, win)]
    seen = {"up": [0, 0], "av": [0, 0], "sm": [0, 0], "fragile": 0}
    for f in range(pcm.shape[0]):
        for ch in range(2):
            x = pcm[f, :, ch].astype(np.float32) / np.float32(65535)                       # fifo.c:105-106
            spec = Oracle.transform_fft(x)
            ex = exact_upload(spec)
            for who, got in (("reference", tex[f, ch, UP]), ("oracle", Oracle.texels_r16(spec))):
                nn, nd = assert_tie_aware(got, ex, d_upload(ex), ("upload", who, f, ch))
            seen["up"][0] += nn; seen["up"][1] += nd
            # the passes, fed with the reference's own upload so that each comparison isolates one pass
            row = texel_float(tex[f, ch, UP])
            Oracle.lib().glvo_gl_chain_r16(row, store[ch], hist[ch], C.byref(heads[ch]), n, F, int(win), 1, 4.2, UR)
            assert (Oracle.texels_r16(store[ch]) == tex[f, ch, GR]).all(), ("gravity store", f, ch)
            ring[ch] = ring[ch][1:] + [tex[f, ch, GR].astype(np.int64)]
            if F > 1:
                ex = exact_average(ring[ch], F, win)
                for who, got in (("reference", tex[f, ch, AV]), ("oracle", Oracle.texels_r16(row))):
                    nn, nd = assert_tie_aware(got, ex, d_average(ex, F), ("average", who, f, ch))
                seen["av"][0] += nn; seen["av"][1] += nd
            else:
                assert (tex[f, ch, AV] == tex[f, ch, GR]).all() and (Oracle.texels_r16(row) == tex[f, ch, GR]).all()     # render.c:2230
            # (the model's state IS the reference's: the gravity store matched exactly)
            forms = [("oracle", oracle_smooth_texels(tex[f, ch, AV], n, factor, name)), ("reference", tex[f, ch, SM])]
            if averaging: forms.insert(1, ("integer form", Oracle.bars_int(tex[f, ch, AV], n, factor, 0.5)[0]))   # the library's exact integer form of the pass (round 5)
            for who, got in forms:
                # the reference's texels come off a GLSL log(): at the shipped factor they meet the correctly-rounded-log standard all the
                # same (log_abs stays 0 there: a regression would show); at other factors the log's admissible error is part of the claim
                nn, nd, nfrag, nw = smooth_admissible(got, tex[f, ch, AV], n, ("smooth pass", who, f, ch), factor, LOG_ABS if (who == "reference" and glsl_log) else 0.0)
            seen["sm"][0] += nn; seen["sm"][1] += nd; seen["fragile"] = nfrag; seen["log"] = seen.get("log", 0) + nw
            # end to end from the PCM: the reference's own av / sm texels lie inside the admissible range of the exact model
            lo, hi = bounds[ch].frame(spec)
            assert ((lo <= tex[f, ch, AV]) & (tex[f, ch, AV] <= hi)).all(), ("chain bounds", f, ch)
            slo, shi = smooth_bounds(lo.astype(np.uint16), hi.astype(np.uint16), n, factor, LOG_ABS if glsl_log else 0.0)
            assert ((slo <= tex[f, ch, SM]) & (tex[f, ch, SM] <= shi)).all(), ("chain bounds, smooth pass", f, ch)
            seen["open"] = seen.get("open", 0) + int((hi > lo).sum()) + int((shi > slo).sum())
    assert seen["fragile"] <= 0.01 * n                                      # (bars with more than one admissible tap set: a handful)
    assert seen["log"] <= 2                                                 # (texels only a GLSL log()'s admissible error explains: one in all the goldens)
    print(name, "near-tie texels / texels the reference rounds the other way:", seen)


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
@pytest.mark.parametrize("name,n,F,win", CASES + FACTOR_CASES + SHAPE_CASES)
def test_device_gl_passes_tie_aware(glvlib, name, n, F, win):
    with shape_ctx(name):
        _device_gl_passes_tie_aware(glvlib, name, n, F, win)


def _device_gl_passes_tie_aware(glvlib, name, n, F, win):
    """The HIP path (gl_storage 1: the fused GL_R16 chain; avg_window_kind 1; GLV_OP_BARS at the pre-smoothing pass's texel centres)
    held to the same standard.  Pass by pass, fed with the reference's own texels: upload (GLV_OP_R16 of the transform), gravity +
    average (the operators on planar rows), smooth pass (glv_batch_bars) -- each equal to the rounded exact value except at genuine
    ties, the smooth pass tap-set-aware.  End to end from the PCM the reference's renderer was fed, in ONE call (upload -> gravity ->
    average -> pre-smoothing pass, all GL_R16): every texel inside the admissible range of the exact model (ChainBounds /
    smooth_bounds: the range is a single value wherever no upload / average tie and no fragile tap set is in play -- equality there),
    and the pre-smoothing pass exact on the device's own `av` texels (the integer weighted mean)."""
    import torch
    G = glvlib
    pcm, tex = GOLD[name + "_pcm"], GOLD[name + "_tex"]
    factor = factor_of(name)
    mask = G.OP_GRAVITY | (G.OP_AVERAGE if F > 1 else 0)
    ops = G.OP_FFT | mask
    p = G.Params(n=n, avg_frames=F, avg_window=win, avg_window_kind=1, gl_storage=1, log_mode=0, ur=UR, bars=n, bar_phase=0.5, smooth_factor=factor, **shape_params(name))
    averaging = shape_of(name)[1] == 0
    up = G.Batch(p, 1, G.OP_FFT)            # the transform with the texel upload
    passes = G.Batch(p, 1, mask)            # gravity / average passes on the reference's upload texels (planar rows in)
    bb = G.Batch(p, 1, G.OP_FFT | G.OP_BARS)
    full = G.Batch(p, 1, mask | G.OP_BARS)  # the whole default pipeline in one call: ... | GLV_OP_BARS | GLV_OP_R16
    chain = G.Batch(p, 1, mask)
    d_sm = torch.zeros((2, n), dtype=torch.int16, device="cuda")
    d_q = torch.zeros((2, n), dtype=torch.int16, device="cuda")
    d_bars = torch.empty((2, n), dtype=torch.float32, device="cuda")
    ring = [[np.zeros(n, np.int64) for _ in range(F)] for _ in range(2)]
    bounds = [ChainBounds(n, F, win), ChainBounds(n, F, win)]
    for f in range(pcm.shape[0]):
        d_pcm = torch.from_numpy(np.ascontiguousarray(pcm[f])).cuda()
        # upload
        up.process_s16(d_pcm, d_q, G.OP_FFT | G.OP_R16)
        got = d_q.cpu().numpy().view(np.uint16)
        for ch in range(2):
            x = pcm[f, :, ch].astype(np.float32) / np.float32(65535)
            ex = exact_upload(Oracle.transform_fft(x))
            assert_tie_aware(got[ch], ex, d_upload(ex), ("upload", f, ch))
        # gravity + average passes on the reference's upload
        rows = np.stack([texel_float(tex[f, ch, UP]) for ch in range(2)])
        passes.process_f32(torch.from_numpy(rows).cuda(), d_q, mask | G.OP_R16)
        got = d_q.cpu().numpy().view(np.uint16)
        for ch in range(2):
            ring[ch] = ring[ch][1:] + [tex[f, ch, GR].astype(np.int64)]
            if F > 1:
                ex = exact_average(ring[ch], F, win)
                assert_tie_aware(got[ch], ex, d_average(ex, F), ("average", f, ch))
            else:
                assert (got[ch] == tex[f, ch, GR]).all()
        # smooth pass on the reference's `av`
        av = np.stack([texel_float(tex[f, ch, AV]) for ch in range(2)])
        bb.bars(torch.from_numpy(av).cuda(), d_bars)
        sm = Oracle.texels_r16(d_bars.cpu().numpy())
        for ch in range(2):
            smooth_admissible(sm[ch], tex[f, ch, AV], n, ("smooth pass", f, ch), factor)
        # end to end from PCM, one call: inside the admissible range of the exact model (equality wherever no tie is in play)
        chain.process_s16(d_pcm, d_q, ops | G.OP_R16)
        full.process_s16(d_pcm, d_sm, ops | G.OP_BARS | G.OP_R16)
        got_av = d_q.cpu().numpy().view(np.uint16); got_sm = d_sm.cpu().numpy().view(np.uint16)
        for ch in range(2):
            x = pcm[f, :, ch].astype(np.float32) / np.float32(65535)
            lo, hi = bounds[ch].frame(Oracle.transform_fft(x))
            assert ((lo <= got_av[ch]) & (got_av[ch] <= hi)).all(), ("chain", f, ch, int(((got_av[ch] < lo) | (got_av[ch] > hi)).sum()))
            if F > 1:
                slo, shi = smooth_bounds(lo.astype(np.uint16), hi.astype(np.uint16), n, factor)
                bad = (got_sm[ch] < slo) | (got_sm[ch] > shi)
                assert not bad.any(), ("end to end", f, ch, int(bad.sum()), np.flatnonzero(bad)[:4])
                # ... and the pass itself is exact on the device's own `av`: the integer weighted mean (maximum / hybrid: the shader's float loop)
                want = Oracle.bars_int(got_av[ch], n, factor, 0.5)[0] if averaging else oracle_smooth_texels(got_av[ch], n, factor, name)
                assert (got_sm[ch] == want).all(), ("the pass on the device's own av", f, ch)
    for b in (up, passes, bb, full, chain): b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,F,win", [("n1024_F5w", 1024, 5, True), ("n1024_F3w", 1024, 3, True), ("n4096_F5w", 4096, 5, True)] + FACTOR_CASES + SHAPE_CASES)
def test_patched_reference_over_llvmpipe_samples_the_unpatched_texture(glvlib, tmp_path, name, n, F, win):
    with shape_ctx(name):
        _patched_reference_over_llvmpipe(glvlib, tmp_path, name, n, F, win)


def _patched_reference_over_llvmpipe(glvlib, tmp_path, name, n, F, win):
    """VERDICT r4 item 2 / r5 item 1, over a real GL: the reference's rd_new / rd_update WITH integration/render_hip.patch
    (oracle/_ref/libglvglref_hip.so: the same harness, the patched render.c, the product library) run over Mesa llvmpipe with the shipped
    shaders, fed the frames the committed goldens were recorded with by the UNPATCHED reference -- the `#request setsmoothfactor` cases
    included: the factor travels as the reference carries it (smooth_parameters.glsl of the configuration directory -> gl_data.smooth_factor
    -> the shim), no environment variable.  The bind's own texture afterwards (read back with glGetTexImage) is held to the standard of the
    library itself (test_device_gl_passes_tie_aware): every texel inside the admissible range of the exact model of the chain (ChainBounds /
    smooth_bounds with the case's factor: a single value -- EQUALITY with the unpatched run's `sm` texel -- wherever no upload / average tie
    and no fragile tap set is in play), no +-1, no excluded bars; the unpatched run's texels lie in the same range
    (test_gl_passes_tie_aware_reference_and_oracle).  The GL passes' own textures (gravity store, ring, average, smooth target) are never
    created: the work happened in the one call on the MI355X.  With GLAVA_HIP_GL off the same build runs the passes on the GL from the HIP
    FFT's upload (bit-faithful log): all four textures EQUAL the unpatched run's.  Needs the GPU, Mesa's swrast driver and the shader files
    (oracle/_ref/shaders on the GPU box)."""
    so = os.path.join(ROOT, "oracle", "_ref", "libglvglref_hip.so")
    shaders = "/root/reference/shaders/glava" if os.path.exists("/root/reference/shaders/glava/rc.glsl") else os.path.join(ROOT, "oracle", "_ref", "shaders")
    if not os.path.exists(so) or not os.path.exists(os.path.join(shaders, "rc.glsl")):
        pytest.skip("oracle/_ref/libglvglref_hip.so / the shader files are not there (built where /root/reference and Mesa are)")
    if not os.path.exists(os.environ.get("GLREF_SWRAST_DRI", "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")):
        pytest.skip("no Mesa swrast_dri.so on this box")
    import subprocess, sys
    pcm, tex = GOLD[name + "_pcm"], GOLD[name + "_tex"]
    request = float(GOLD[name + "_factor"]) if name + "_factor" in GOLD.files else None
    factor = factor_of(name)
    np.save(str(tmp_path / "pcm.npy"), pcm)
    for on_hip in (1, 0):
        # in a child process: rd_new prints deprecation warnings and glava_abort()s on errors
        code = ("import sys, numpy as np, tempfile; sys.path.insert(0, %r); import make_gl_golden as M; pcm = np.load(%r); "
                "out, v, r = M.run_case(%d, %d, %r, pcm, tempfile.mkdtemp(), so=%r, hip=(%d, 0), factor=%r, defines=%r); np.save(%r, out)"
                % (os.path.join(ROOT, "tests", "golden"), str(tmp_path / "pcm.npy"), n, F, bool(win), so, on_hip, request, SHAPES[name][0] if name in SHAPES else None,
                   str(tmp_path / "t.npy")))
        env = {k: v for k, v in os.environ.items() if k != "GLAVA_HIP_SMOOTH_FACTOR"}
        subprocess.run([sys.executable, "-c", code], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(env, GLV_SHADERS=shaders))
        got = np.load(str(tmp_path / "t.npy"))
        if not on_hip:
            assert (got == tex).all(), (name, "GL passes on the GL from the HIP upload", int((got != tex).sum()))
            continue
        bounds = [ChainBounds(n, F, win), ChainBounds(n, F, win)]
        equal = total = 0
        for f in range(pcm.shape[0]):
            for ch in range(2):
                x = pcm[f, :, ch].astype(np.float32) / np.float32(65535)
                lo, hi = bounds[ch].frame(Oracle.transform_fft(x))
                slo, shi = smooth_bounds(lo.astype(np.uint16), hi.astype(np.uint16), n, factor)
                final = got[f, ch, UP].astype(np.int64)                         # on the MI355X the bind's own texture holds the result
                bad = (final < slo) | (final > shi)
                assert not bad.any(), (name, f, ch, int(bad.sum()), np.flatnonzero(bad)[:4])
                if request is not None or name in SHAPES:                       # (the unpatched run's texture lies in the range a GLSL log() opens around it)
                    slo, shi = smooth_bounds(lo.astype(np.uint16), hi.astype(np.uint16), n, factor, LOG_ABS)
                assert ((slo <= tex[f, ch, SM]) & (tex[f, ch, SM] <= shi)).all()
                equal += int((final == tex[f, ch, SM]).sum()); total += n
                assert not got[f, ch, GR].any() and not got[f, ch, SM].any()   # the GL passes' textures were never made
        print(name, "texels equal to the unpatched llvmpipe run's: %d of %d" % (equal, total))
