"""The GL-default pipeline in ONE launch.

GLava ships `setaccelfft true` (shaders/glava/rc.glsl:211): gravity and average run as GL passes on GL_R16 textures
(glava/render.c:2188-2265, textures :523, :1718), optionally followed by the pre-smoothing pass (:2277-2303).  The library
has two forms of that chain:

  gl_storage = 2   pass by pass, as the reference runs it: the transform writes f32 spectra, a second kernel applies upload
                   quantisation / GL_MAX + gravity / ring / average on f32 state (every value a float c / 65535), a third the
                   bars.  Pinned against the oracle's glvo_gl_chain_r16 and against the reference's own llvmpipe texels
                   (tests/test_gl_storage.py, tests/test_gl_reference.py).  It is the CHECKER here.
  gl_storage = 1   the same arithmetic as the transform kernel's epilogue on 16-bit state (uint16 store + ring): one launch,
                   28 n bytes per frame at F = 5 instead of ~80 n.

This file demands that 1 gives 2's bits -- every output form (texels, floats, bars as floats and as texels, bars == n), every
input kind, both log modes, every kernel configuration of a size, gravity-only / average-only / F == 1 chains -- and that it
really is one launch."""
import ctypes as C

import numpy as np
import pytest

from oracle_lib import Oracle, StreamOracle, lcg_pcm_fast

pytestmark = pytest.mark.gpu


def _eq(a, b):
    import torch
    ia = a.view(torch.int32) if a.dtype == torch.float32 else a
    ib = b.view(torch.int32) if b.dtype == torch.float32 else b
    return bool(torch.equal(ia, ib))


@pytest.mark.parametrize("n,F,win,log_mode", [(256, 5, True, 1), (512, 5, True, 0), (1024, 5, True, 1), (1024, 6, False, 0), (2048, 2, True, 1),
                                              (2048, 1, True, 0), (4096, 5, True, 1), (4096, 5, True, 0), (4096, 3, False, 1), (8192, 5, True, 1),
                                              (8192, 4, True, 0), (16384, 3, True, 1), (16384, 5, True, 0), (32768, 2, True, 1)])
def test_fused_gl_chain_gives_the_pass_by_pass_bits(glvlib, n, F, win, log_mode):
    """s16 frames in; `av` texels, their floats, 80 bars as floats and as texels; one launch each for the fused form"""
    import torch
    G = glvlib
    streams = 7 if n <= 8192 else 3
    mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    kw = dict(n=n, avg_frames=F, avg_window=win, avg_window_kind=1, log_mode=log_mode, bars=80)
    forms = [("tex", ops | G.OP_R16, torch.int16, n), ("flt", ops, torch.float32, n), ("bars", ops | G.OP_BARS, torch.float32, 80),
             ("bars16", ops | G.OP_BARS | G.OP_R16, torch.int16, 80)]
    fused = {k: G.Batch(G.Params(gl_storage=1, **kw), streams, mask) for k, *_ in forms}
    split = {k: G.Batch(G.Params(gl_storage=2, **kw), streams, mask) for k, *_ in forms}
    out_f = {k: torch.zeros((streams * 2, w), dtype=dt, device="cuda") for k, _, dt, w in forms}
    out_s = {k: torch.zeros((streams * 2, w), dtype=dt, device="cuda") for k, _, dt, w in forms}
    for fr in range(F + 3):
        div = (1, 8, 64)[fr % 3]                                       # loud frames saturate texels, quiet ones let gravity show
        pcm = (lcg_pcm_fast(4200 + fr + n, streams * 2 * n) // div).astype(np.int16)
        if fr == 2: pcm[:] = 0                                          # silence: the store decays
        d_pcm = torch.from_numpy(pcm).cuda()
        # bars are computed inside the launch where whole waves own a row (n >= 1024): below that they are a second launch on the
        # finished rows' floats
        for k, o, _, _ in forms:
            fused[k].process_s16(d_pcm, out_f[k], o)
            assert fused[k].last_launches() == (2 if (o & G.OP_BARS) and n < 1024 else 1), (k, fused[k].last_launches())
            split[k].process_s16(d_pcm, out_s[k], o)
            assert split[k].last_launches() >= 2
        torch.cuda.synchronize()
        for k, *_ in forms:
            assert _eq(out_f[k], out_s[k]), (fr, k)
        # the floats are the texels' read-back values c / 65535
        tex_f = torch.from_numpy(out_f["tex"].cpu().numpy().view(np.uint16).astype(np.float32) / np.float32(65535)).cuda()
        assert _eq(out_f["flt"], tex_f), fr
    # byte accounting: 16-bit state
    b = fused["tex"]
    assert b.algorithmic_bytes(ops | G.OP_R16) == streams * (4 * n + 4 * n * (F - 1) + 4 * n + 4 * n)
    assert split["tex"].algorithmic_bytes(ops | G.OP_R16) == streams * (4 * n + 8 * n * (F - 1) + 8 * n + 4 * n + 16 * n)
    for d in (fused, split):
        for x in d.values(): x.close()


@pytest.mark.parametrize("n,F", [(1024, 5), (4096, 5), (16384, 3)])
def test_fused_gl_chain_equals_the_oracle_model(glvlib, n, F):
    """independently of the pass-by-pass form: transform_fft (oracle) + glvo_gl_chain_r16, bit for bit with the bit-faithful log"""
    import torch
    G = glvlib
    streams = 5
    p = G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, log_mode=0)
    b = G.Batch(p, streams, G.OP_GRAVITY | G.OP_AVERAGE)
    d_q = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
    store = np.zeros((streams * 2, n), np.float32); hist = np.zeros((streams * 2, F, n), np.float32)
    heads = [C.c_size_t(0) for _ in range(streams * 2)]
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16
    for fr in range(F + 3):
        pcm = (lcg_pcm_fast(3100 + fr + n, streams * 2 * n) // 16).astype(np.int16)
        b.process_s16(torch.from_numpy(pcm).cuda(), d_q, ops)
        gq = d_q.cpu().numpy().view(np.uint16)
        for u in range(streams):
            spec = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n])
            for c in range(2):
                want = np.ascontiguousarray(spec[c])
                Oracle.lib().glvo_gl_chain_r16(want, store[2 * u + c], hist[2 * u + c], C.byref(heads[2 * u + c]), n, F, 1, 1, 4.2, 86.1328125)
                assert (gq[2 * u + c] == Oracle.texels_r16(want)).all(), (fr, u, c)
    b.close()


@pytest.mark.parametrize("n", [1024, 4096])
def test_fused_gl_chain_every_input_kind_and_sub_chain(glvlib, n):
    """planar f32, interleaved f32, the two device rings; gravity alone, average alone; mono -- fused == pass by pass"""
    import torch
    G = glvlib
    streams, F = 5, 4
    rng = np.random.default_rng(n)
    x_pl = (rng.standard_normal((streams * 2, n)) * 0.2).astype(np.float32)
    x_st = (rng.standard_normal((streams, n, 2)) * 0.2).astype(np.float32)
    pcm = lcg_pcm_fast(77 + n, streams * 2 * n)
    d_pl, d_st, d_pcm = torch.from_numpy(x_pl).cuda(), torch.from_numpy(x_st).cuda(), torch.from_numpy(pcm).cuda()
    for sub in (G.OP_GRAVITY | G.OP_AVERAGE, G.OP_GRAVITY, G.OP_AVERAGE):
        for ch in (2, 1):
            mk = lambda gl: G.Batch(G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=gl, channels=ch), streams,     # noqa: E731
                                    sub | G.OP_RING_S16 | G.OP_RING_F32)
            for kind in ("s16", "planar", "stereo", "ring16", "ring32"):
                if ch == 1 and kind == "planar": continue
                bf, bs = mk(1), mk(2)
                of = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda"); os_ = torch.zeros_like(of)
                for u in range(F + 1):
                    for b, o in ((bf, of), (bs, os_)):
                        ops = G.OP_FFT | sub | G.OP_R16
                        if kind == "s16": b.process_s16(d_pcm, o, ops)
                        elif kind == "planar": b.process_f32(d_pl, o, ops)
                        elif kind == "stereo": b.process_f32_stereo(d_st, o, ops)
                        elif kind == "ring16": b.ring_update_s16(d_pcm[: streams * 2 * 300].reshape(streams, 300, 2).contiguous(), 300, o, ops)
                        else: b.ring_update_f32(d_st[:, :300].contiguous(), 300, o, ops)
                    assert bf.last_launches() == 1
                    assert _eq(of, os_), (sub, ch, kind, u)
                bf.close(); bs.close()


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192])
def test_fused_gl_chain_every_kernel_configuration(glvlib, n):
    import torch
    G = glvlib
    streams, F = 9, 5
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    outs = []
    nv = None
    for v in range(2):
        b = G.Batch(G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1), streams, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS)
        nv = b.variants(); b.set_variant(v)
        o = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
        ob = torch.zeros((streams * 2, 80), dtype=torch.float32, device="cuda")
        got = []
        for u in range(F + 1):
            d_pcm = torch.from_numpy((lcg_pcm_fast(910 + u + n, streams * 2 * n) // 4).astype(np.int16)).cuda()
            b.process_s16(d_pcm, o, ops | G.OP_R16); assert b.last_variant() == v
            got.append(o.clone())
        b.reset()
        for u in range(F + 1):
            d_pcm = torch.from_numpy((lcg_pcm_fast(910 + u + n, streams * 2 * n) // 4).astype(np.int16)).cuda()
            b.process_s16(d_pcm, ob, ops | G.OP_BARS)
            got.append(ob.clone())
        outs.append(got); b.close()
    assert nv == 2
    for x, y in zip(outs[0], outs[1]):
        assert _eq(x, y)


def test_gl_default_pipeline_end_to_end_one_call(glvlib):
    """PCM in, the texture every stock module samples out: upload -> gravity -> average -> pre-smoothing pass (bars == n,
    bar_phase 0.5).  The smooth pass's 4096 bars do not fit the slack behind a row in LDS, so this chain is two launches
    (the fused GL kernel writing the `av` floats, then the bars); its texels equal the pass-by-pass form's."""
    import torch
    G = glvlib
    n, F, streams = 4096, 5, 3
    kw = dict(n=n, avg_frames=F, avg_window_kind=1, bars=n, bar_phase=0.5)
    mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
    bf, bs = G.Batch(G.Params(gl_storage=1, **kw), streams, mask), G.Batch(G.Params(gl_storage=2, **kw), streams, mask)
    of = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda"); os_ = torch.zeros_like(of)
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16
    for u in range(F + 1):
        d_pcm = torch.from_numpy((lcg_pcm_fast(31 + u, streams * 2 * n) // 16).astype(np.int16)).cuda()
        bf.process_s16(d_pcm, of, ops); bs.process_s16(d_pcm, os_, ops)
        assert bf.last_launches() == 2 and bs.last_launches() == 3
        assert _eq(of, os_), u
    bf.close(); bs.close()


def test_gl_state_storage_class_is_fixed_at_creation(glvlib):
    import torch
    G = glvlib
    n, streams = 1024, 2
    p1 = G.Params(n=n, gl_storage=1, avg_window_kind=1)
    b = G.Batch(p1, streams, G.OP_GRAVITY | G.OP_AVERAGE)
    with pytest.raises(G.GlvError) as ei:
        b.set_params(G.Params(n=n, gl_storage=2, avg_window_kind=1))
    assert ei.value.code == G.ERR_STATE
    with pytest.raises(G.GlvError) as ei:
        b.gravity_state()                                           # uint16 texels, not floats
    assert ei.value.code == G.ERR_STATE
    b.set_params(G.Params(n=n, gl_storage=1, avg_window_kind=1, gravity_step=1.0))      # a knob: fine
    b.close()
    # the single-stream drop-ins keep their class too
    s = G.State(G.Params(n=n, gl_storage=0))
    buf = np.zeros(n, np.float32)
    s.params = G.Params(n=n, gl_storage=1)
    with pytest.raises(G.GlvError):
        s.gravity(buf)
    s.close()


@pytest.mark.parametrize("n,streams", [(1024, 150), (2048, 129), (4096, 165), (4096, 128)])
def test_many_bars_of_many_rows_kernel_gives_the_documented_bits(glvlib, n, streams):
    """The pre-smoothing pass at scale (bars == n, bar_phase 0.5): glv_bars_rows_kernel -- a banded matrix product on the
    matrix cores, 32 bars x 64 rows x 2 bins per v_mfma_f32_32x32x2_f32, the rows' texels in an LDS ring -- computes the documented
    order (from 256 bars up: one fma chain per bar in bin order), so its bars equal the oracle's glvo_bars_chunked_at bit for bit on
    EVERY row, whether a batch has hundreds of rows or sixteen (partial row blocks); texel output likewise (the one-lane-per-bar
    kernel: test_many_bars_kernels_agree_over_random_parameters).  Rows
    include values outside [0, 1], NaN and Inf (clamped like a GL_R16 texel)."""
    import torch
    G = glvlib
    rows = streams * 2
    rng = np.random.default_rng(n + streams)
    spec = (rng.random((rows, n), dtype=np.float32) ** 2 * np.float32(1.3) - np.float32(0.05)).astype(np.float32)
    spec[3, 100:140] = np.nan; spec[5, 7] = np.inf; spec[6, 300:310] = -np.inf
    p = G.Params(n=n, bars=n, bar_phase=0.5)
    big = G.Batch(p, streams, G.OP_FFT | G.OP_BARS)
    small = G.Batch(p, 8, G.OP_FFT | G.OP_BARS)
    d_spec = torch.from_numpy(spec).cuda()
    d_big = torch.empty((rows, n), dtype=torch.float32, device="cuda")
    big.bars(d_spec, d_big)
    got = d_big.cpu().numpy()
    d_small = torch.empty((16, n), dtype=torch.float32, device="cuda")
    for r0 in (0, rows - 16):
        small.bars(d_spec[r0:r0 + 16].contiguous(), d_small)
        assert _eq(d_small, d_big[r0:r0 + 16].contiguous()), r0
    for r in range(rows):                   # every row (round 4: n = 4096, bar 2848 has a skipped bin -- one ulp apart on one row in seven)
        want = np.empty(n, np.float32)
        Oracle.lib().glvo_bars_chunked_at(np.ascontiguousarray(spec[r]), n, want, n, 0.025, 0.5)
        assert (got[r].view(np.uint32) == want.view(np.uint32)).all(), r
    # the whole GL-default pipeline with the pre-smoothing pass, texels out: many streams (rows kernel) == few streams (bars kernel)
    kw = dict(n=n, avg_frames=3, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)
    mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
    ops = G.OP_FFT | mask | G.OP_R16
    fb, fs = G.Batch(G.Params(**kw), streams, mask), G.Batch(G.Params(**kw), 8, mask)
    ob = torch.zeros((rows, n), dtype=torch.int16, device="cuda"); os_ = torch.zeros((16, n), dtype=torch.int16, device="cuda")
    for u in range(4):
        pcm = (lcg_pcm_fast(555 + u + n, streams * 2 * n) // 8).astype(np.int16)
        d_pcm = torch.from_numpy(pcm).cuda()
        fb.process_s16(d_pcm, ob, ops); fs.process_s16(d_pcm[: 8 * 2 * n], os_, ops)
        assert fb.last_launches() == 2
        assert _eq(ob[:16].contiguous(), os_), u
    for b in (big, small, fb, fs): b.close()


@pytest.mark.parametrize("n,bars,phase,rows", [(2048, 1001, 0.0, 300), (1024, 259, 0.5, 257), (4096, 4096, 0.5, 70 * 64 + 3), (512, 512, 0.5, 1024),
                                                (8192, 8192, 0.5, 258), (16384, 16384, 0.5, 258), (16384, 256, 0.0, 300), (4096, 4096, 0.5, 6), (4096, 4096, 0.5, 70), (2048, 2048, 0.5, 64)])
def test_many_rows_kernel_with_ragged_tables(glvlib, n, bars, phase, rows):
    """The many-bars kernels away from the round numbers: a bar count that is not a multiple of 32 (a last tile of a few bars, a last
    round of fewer than four tiles), bars further apart than in the pre-smoothing pass, a row count that leaves a last workgroup of
    3 rows, more row blocks than resident workgroups, the long rings of n = 8192 / 16384 with 32 rows per workgroup (the matrix-core
    kernel); tiles too wide for any ring (256 bars at n = 16384) and few rows (the one-lane-per-bar kernel) -- every bar of every row against the oracle's chain, floats."""
    import torch
    G = glvlib
    streams = (rows + 1) // 2
    rng = np.random.default_rng(n * 7 + bars)
    spec = (rng.random((streams * 2, n), dtype=np.float32) ** 3 * np.float32(1.2) - np.float32(0.02)).astype(np.float32)
    spec[1, : n // 3] = 0.0
    p = G.Params(n=n, bars=bars, bar_phase=phase)
    b = G.Batch(p, streams, G.OP_FFT | G.OP_BARS)
    d_spec = torch.from_numpy(spec).cuda()
    d_out = torch.full((streams * 2, bars), -1.0, dtype=torch.float32, device="cuda")
    b.bars(d_spec, d_out)
    got = d_out.cpu().numpy()
    want = np.empty((streams * 2, bars), np.float32)
    for r in range(streams * 2):
        Oracle.lib().glvo_bars_chunked_at(np.ascontiguousarray(spec[r]), n, want[r], bars, 0.025, phase)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), (int((~same).sum()), np.argwhere(~same)[:4].tolist())
    b.close()


def test_many_bars_kernels_agree_over_random_parameters(glvlib, monkeypatch):
    """The matrix-core kernel against the one-lane-per-bar kernel (GLV_NO_BARS_ROWS at table creation) over a seeded sweep of sizes,
    bar counts, smoothing widths and phases -- wide bars (smooth_factor up to 0.12: hundreds of taps), bar counts that leave ragged
    tiles and rounds, rings of every size the kernel is built for, parameter sets no ring takes (both batches then run the same
    kernel) -- bit for bit, floats and texels; and one row of each against the oracle's chain."""
    import torch
    G = glvlib
    rng = np.random.default_rng(2024)
    seen_rings = 0
    for trial in range(24):
        n = int(rng.choice([256, 512, 1024, 2048, 4096, 8192]))
        bars = int(rng.choice([256, 300, 333, 512, 1000, n])) if n >= 512 else 256
        bars = min(bars, n)
        factor = float(rng.choice([0.005, 0.025, 0.05, 0.12]))
        phase = float(rng.choice([0.0, 0.5, 0.25]))
        rows = 256 + 2 * int(rng.integers(0, 40))
        streams = rows // 2
        spec = (rng.random((rows, n), dtype=np.float32) ** 2 * np.float32(1.25) - np.float32(0.04)).astype(np.float32)
        spec[2, ::9] = np.nan
        p = G.Params(n=n, bars=bars, bar_phase=phase, smooth_factor=factor)
        d_spec = torch.from_numpy(spec).cuda()
        try:
            fast = G.Batch(p, streams, G.OP_FFT | G.OP_BARS)
        except G.GlvError:
            continue                                                        # (a tap chunk would leave the row: refused at creation)
        monkeypatch.setenv("GLV_NO_BARS_ROWS", "1")
        slow = G.Batch(p, streams, G.OP_FFT | G.OP_BARS)
        monkeypatch.delenv("GLV_NO_BARS_ROWS")
        a = torch.full((rows, bars), -1.0, dtype=torch.float32, device="cuda"); b2 = torch.full_like(a, -2.0)
        fast.bars(d_spec, a); slow.bars(d_spec, b2)
        ga, gb = a.cpu().numpy(), b2.cpu().numpy()
        same = (ga.view(np.uint32) == gb.view(np.uint32)) | (np.isnan(ga) & np.isnan(gb))
        assert same.all(), (trial, n, bars, factor, phase, int((~same).sum()))
        want = np.empty(bars, np.float32)
        Oracle.lib().glvo_bars_chunked_at(np.ascontiguousarray(spec[rows - 1]), n, want, bars, factor, phase)
        ok = (ga[rows - 1].view(np.uint32) == want.view(np.uint32)) | (np.isnan(ga[rows - 1]) & np.isnan(want))
        assert ok.all(), (trial, n, bars, factor, phase)
        seen_rings += 1
        fast.close(); slow.close()
    assert seen_rings >= 12


@pytest.mark.parametrize("n,streams,F,rows_checked,bars,phase", [(1024, 70, 3, 140, 0, 0.5), (2048, 33, 2, 66, 0, 0.5), (4096, 70, 5, 140, 0, 0.5), (4096, 3, 5, 6, 0, 0.5),
                                                                 (8192, 34, 2, 12, 0, 0.5), (16384, 33, 2, 6, 0, 0.5), (32768, 5, 2, 3, 0, 0.5),
                                                                 (2048, 35, 2, 70, 1001, 0.0), (1024, 41, 2, 82, 259, 0.5), (4096, 33, 2, 66, 296, 0.25)])
def test_presmoothing_pass_of_the_gl_chains_is_the_exact_integer_mean(glvlib, n, streams, F, rows_checked, bars, phase):
    """Round 5: inside the library's GL chains the rows the pre-smoothing pass samples are GL_R16 texels (render.c:2277-2303 samples a
    texture), so the many-bars pass runs in EXACT integer arithmetic on the i8 matrix cores (glv_bars_rows_i8_kernel; contract in
    glv_tables.h make_bar_itiles): sm = floor(sum W c / 2^P + 1/2) with integer weights that sum to 2^P.  The fused chain (texel rows
    handed over as uint16), the pass-by-pass chain (the same values as floats c / 65535) and the oracle's independent restatement
    (glvo_bars_int_at on the chain's own `av` texels) must agree EXACTLY -- texels and float bits -- on every checked row: whole
    64-row blocks and a partial last one, every ring the kernel is built for (n = 32768: 32 rows per workgroup, a 1600-bin ring --
    VERDICT r4 missing 2), two launches; bar counts that are not a multiple of 8 (or of 32: a ragged last tile, a last round of fewer
    than four) take the kernel's unstaged store path."""
    import torch
    G = glvlib
    rows = streams * 2
    bars = bars or n
    kw = dict(n=n, avg_frames=F, avg_window_kind=1, bars=bars, bar_phase=phase)
    mask = G.OP_GRAVITY | G.OP_AVERAGE
    ops = G.OP_FFT | mask
    av = G.Batch(G.Params(gl_storage=1, **kw), streams, mask)
    sm16 = G.Batch(G.Params(gl_storage=1, **kw), streams, mask | G.OP_BARS)
    smf = G.Batch(G.Params(gl_storage=1, **kw), streams, mask | G.OP_BARS)
    sp16 = G.Batch(G.Params(gl_storage=2, **kw), streams, mask | G.OP_BARS)
    o_av = torch.zeros((rows, n), dtype=torch.int16, device="cuda")
    o_16 = torch.full((rows, bars), -1, dtype=torch.int16, device="cuda"); o_s16 = torch.zeros_like(o_16)
    o_f = torch.zeros((rows, bars), dtype=torch.float32, device="cuda")
    pick = np.unique(np.linspace(0, rows - 1, rows_checked).astype(int))
    for u in range(F + 1):
        pcm = (lcg_pcm_fast(9000 + u + n, streams * 2 * n) // (1, 16, 3)[u % 3]).astype(np.int16)
        d_pcm = torch.from_numpy(pcm).cuda()
        av.process_s16(d_pcm, o_av, ops | G.OP_R16)
        sm16.process_s16(d_pcm, o_16, ops | G.OP_BARS | G.OP_R16)
        assert sm16.last_launches() == 2
        smf.process_s16(d_pcm, o_f, ops | G.OP_BARS)
        sp16.process_s16(d_pcm, o_s16, ops | G.OP_BARS | G.OP_R16)
        assert _eq(o_16, o_s16), u
        if u < F - 1 and u != 1: continue                     # the oracle on the loud first frames' successor and on the full ring
        t_av = o_av.cpu().numpy().view(np.uint16); t_16 = o_16.cpu().numpy().view(np.uint16); t_f = o_f.cpu().numpy()
        for r in pick:
            w16, wf = Oracle.bars_int(t_av[r], bars, 0.025, phase)
            assert (t_16[r] == w16).all(), (u, r, int((t_16[r] != w16).sum()), np.flatnonzero(t_16[r] != w16)[:5])
            assert (t_f[r].view(np.uint32) == wf.view(np.uint32)).all(), (u, r)
    for b in (av, sm16, smf, sp16): b.close()


def test_float_chain_with_bars_as_texels_after_creation_time_allocation(glvlib):
    """gl_storage 0, GRAVITY | AVERAGE | BARS in the creation mask, 80 bars the transform kernel could compute itself --
    and then FFT | GRAVITY | AVERAGE | BARS | R16: the float chain's bars as texels leave through glv_bars_kernel, which needs the internal
    spectra rows; they must have been allocated at creation (process calls never allocate).  The texels are the quantised float bars.
    Since ABI 6 (512 MiB of rows nothing read at 16 K streams) a float chain whose every kernel configuration computes the bars itself
    gets those rows only when the creation mask carries GLV_OP_R16 as well -- the hint that texel bars will be asked for; without it the call is
    refused (GLV_ERR_STATE), never allocated mid-stream."""
    import torch
    G = glvlib
    n, streams, bars = 4096, 5, 80
    mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
    ops = G.OP_FFT | mask
    p = G.Params(n=n, bars=bars)
    bt, bf = G.Batch(p, streams, mask | G.OP_R16), G.Batch(p, streams, mask)
    ot = torch.zeros((streams * 2, bars), dtype=torch.int16, device="cuda")
    of = torch.zeros((streams * 2, bars), dtype=torch.float32, device="cuda")
    for u in range(7):
        d_pcm = torch.from_numpy((lcg_pcm_fast(77 + u, streams * 2 * n) // 4).astype(np.int16)).cuda()
        bt.process_s16(d_pcm, ot, ops | G.OP_R16)
        bf.process_s16(d_pcm, of, ops)
        assert bf.last_launches() == 1 and bt.last_launches() == 2
        assert (ot.cpu().numpy().view(np.uint16) == Oracle.texels_r16(of.cpu().numpy())).all(), u
    with pytest.raises(G.GlvError) as ei:                # no hint, no rows: refused
        bf.process_s16(d_pcm, ot, ops | G.OP_R16)
    assert ei.value.code == G.ERR_STATE
    bt.close(); bf.close()



def test_single_stream_gl_texture_is_the_batched_chain(glvlib):
    """glv_gl_texture (the drop-in integration/render_hip.patch binds into handle_audio's accel path: host samples in, the module's GL_R16
    texels out) against the batched GL-default chain on planar rows: the same texels, with and without the pre-smoothing pass, a silent
    update in between, the samples left untouched."""
    import torch
    G = glvlib
    n, F = 4096, 5
    kw = dict(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)
    mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
    for smooth in (True, False):
        st = [G.State(G.Params(**kw)) for _ in range(2)]
        b = G.Batch(G.Params(**kw), 1, mask)
        ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16 | (G.OP_BARS if smooth else 0)
        out = torch.zeros((2, n), dtype=torch.int16, device="cuda")
        for u in range(F + 2):
            x = (lcg_pcm_fast(606 + u, 2 * n).reshape(n, 2).T.astype(np.float32) / np.float32(65535 * (1, 8)[u % 2])).copy()
            if u == 3: x[:] = 0
            b.process_f32(torch.from_numpy(x).cuda(), out, ops)
            want = out.cpu().numpy().view(np.uint16)
            for ch in range(2):
                row = x[ch].copy(); tex = np.zeros(n, np.uint16)
                st[ch].gl_texture(row, tex, smooth)
                assert (row == x[ch]).all() and (tex == want[ch]).all(), (smooth, u, ch)
        for s in st: s.close()
        b.close()


def test_presmoothing_pass_of_the_gl_chains_over_random_parameters(glvlib):
    """The integer pass over a seeded sweep of sizes, bar counts (ragged last tiles and rounds), smoothing widths (wide bars: hundreds of
    taps, the long rings) and phases, F = 1 .. 3: the fused chain and the pass-by-pass chain agree bit for bit, and two rows of every
    trial equal the oracle's integer mean of the chain's own `av` texels -- or, where no ring takes the tiles (the library then keeps
    the float chain's kernels for the pass), the documented fma chain of the texels' floats."""
    import torch
    G = glvlib
    rng = np.random.default_rng(555)
    seen_int = 0
    for trial in range(16):
        n = int(rng.choice([512, 1024, 2048, 4096, 8192]))
        bars = min(n, int(rng.choice([256, 300, 333, 512, 1000, n])))
        factor = float(rng.choice([0.005, 0.025, 0.05, 0.12]))
        phase = float(rng.choice([0.0, 0.5, 0.25]))
        F = int(rng.choice([1, 2, 3]))
        streams = 65 + int(rng.integers(0, 40))
        kw = dict(n=n, avg_frames=F, avg_window_kind=1, bars=bars, bar_phase=phase, smooth_factor=factor)
        mask = G.OP_GRAVITY | (G.OP_AVERAGE if F > 1 else 0)
        ops = G.OP_FFT | mask
        try:
            sm1 = G.Batch(G.Params(gl_storage=1, **kw), streams, mask | G.OP_BARS)
        except G.GlvError:
            continue                                    # (a parameter set without bar tables at all)
        sm2 = G.Batch(G.Params(gl_storage=2, **kw), streams, mask | G.OP_BARS)
        av = G.Batch(G.Params(gl_storage=1, **kw), streams, mask)
        o1 = torch.full((streams * 2, bars), -1, dtype=torch.int16, device="cuda"); o2 = torch.zeros_like(o1)
        oa = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
        for u in range(F + 1):
            d_pcm = torch.from_numpy((lcg_pcm_fast(31 * trial + u, streams * 2 * n) // (4, 24)[u % 2]).astype(np.int16)).cuda()
            sm1.process_s16(d_pcm, o1, ops | G.OP_BARS | G.OP_R16); sm2.process_s16(d_pcm, o2, ops | G.OP_BARS | G.OP_R16)
            av.process_s16(d_pcm, oa, ops | G.OP_R16)
            assert _eq(o1, o2), (trial, n, bars, factor, phase, F, u)
        t_av = oa.cpu().numpy().view(np.uint16); t_sm = o1.cpu().numpy().view(np.uint16)
        arith = sm1.bars_arithmetic()                   # the library SAYS which arithmetic ran (ABI 6): no either-or in the comparison
        assert arith == sm2.bars_arithmetic() and arith in (G.BARS_I8_EXACT, G.BARS_F32_MATRIX), arith
        for r in (0, streams * 2 - 1):
            if arith == G.BARS_I8_EXACT:
                w16 = np.zeros(bars, np.uint16)
                rc = Oracle.lib().glvo_bars_int_at(np.ascontiguousarray(t_av[r]), n, w16.ctypes.data, None, bars, factor, phase)
                assert rc == 0 and (t_sm[r] == w16).all(), (trial, n, bars, factor, phase, F, r, rc)
                seen_int += 1
                continue
            want = np.empty(bars, np.float32)
            Oracle.lib().glvo_bars_chunked_at((t_av[r].astype(np.float32) / np.float32(65535)).copy(), n, want, bars, factor, phase)
            assert (t_sm[r] == Oracle.texels_r16(want)).all(), (trial, n, bars, factor, phase, F, r)
        for b in (sm1, sm2, av): b.close()
    assert seen_int >= 16                               # most trials run the integer pass


@pytest.mark.gpu
@pytest.mark.parametrize("n,F,factor,in_kind", [(1024, 5, 0.025, "s16"), (2048, 3, 0.05, "s16"), (4096, 5, 0.025, "s16"), (4096, 5, 0.025, "f32"), (4096, 2, 0.01, "s16"),
                                                (8192, 5, 0.025, "s16"), (16384, 3, 0.025, "s16"), (32768, 2, 0.025, "s16")])
def test_bars_only_chain_is_the_full_chain_on_the_bins_the_bars_sample(glvlib, n, F, factor, in_kind):
    """GLV_OP_BARS_ONLY (round 6): GLava's modules sample the pre-smoothed texture and nothing else (smooth.glsl:62), and smooth_audio() reaches
    bins below scale_audio(1) n = 0.288 n plus half a window -- what the reference's GL passes compute beyond is never looked at.  A batch
    created with the flag keeps the gravity store and the ring, and computes magnitude / upload / gravity / average, for those bins only
    (kernel class 7: a compile-time share of the last pass's blocks -- 3/8 of the row -- taken when the bars sample nothing beyond, else the full chain).  Its `sm` texels must equal, bit for bit and update after update
    (loud, quiet, silent frames: the gravity store and every ring slot are exercised), those of a batch without the flag, in both kernel
    configurations of the size; a stateful call without GLV_OP_BARS is refused; the algorithmic bytes shrink with the live bins."""
    import torch
    G = glvlib
    streams = 9 if n <= 8192 else 3
    kw = dict(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5, smooth_factor=factor)
    mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16
    for variant in (0, 1):
        full = G.Batch(G.Params(**kw), streams, mask)
        live = G.Batch(G.Params(**kw), streams, mask | G.OP_BARS_ONLY)
        try:
            full.set_variant(variant); live.set_variant(variant)
        except G.GlvError:
            full.close(); live.close()
            continue
        L = live.live_bins()
        assert full.live_bins() == 0 and live.bars_arithmetic() == G.BARS_I8_EXACT
        if L == 0:                                      # the bars reach beyond what the live kernel classes keep (3/8 of the row): the full chain serves
            assert factor > 0.03 or n <= 2048, (n, factor)
            assert live.algorithmic_bytes(ops, in_kind == "s16") == full.algorithmic_bytes(ops, in_kind == "s16")
        else:
            assert 0.28 * n < L <= 0.5 * n and L % 64 == 0, (L, n)
            assert live.algorithmic_bytes(ops, in_kind == "s16") < full.algorithmic_bytes(ops, in_kind == "s16")
        o_f = torch.full((streams * 2, n), -1, dtype=torch.int16, device="cuda"); o_l = torch.full_like(o_f, -2)
        for u in range(F + 3):
            pcm = (lcg_pcm_fast(8100 + u + n, streams * 2 * n) // (1, 8, 64)[u % 3]).astype(np.int16)
            if u == 2: pcm[:] = 0
            if in_kind == "s16":
                d_in = torch.from_numpy(pcm).cuda()
                full.process_s16(d_in, o_f, ops); live.process_s16(d_in, o_l, ops)
            else:
                x = (pcm.reshape(streams, n, 2).transpose(0, 2, 1).astype(np.float32) / np.float32(65535)).copy()
                d_in = torch.from_numpy(x).cuda()
                full.process_f32(d_in, o_f, ops); live.process_f32(d_in, o_l, ops)
            assert live.last_launches() == 2 and full.last_launches() == 2
            assert _eq(o_f, o_l), (n, F, variant, u, int((o_f != o_l).sum()))
        with pytest.raises(G.GlvError) as ei:
            live.process_s16(torch.zeros((streams, n, 2), dtype=torch.int16, device="cuda"), o_l, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16)
        assert ei.value.code == G.ERR_STATE
        full.close(); live.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,gl,avg,bars", [(4096, 0, False, 80), (4096, 0, True, 80), (4096, 1, True, 80), (16384, 0, False, 80), (16384, 1, True, 80),
                                           (8192, 0, True, 100), (2048, 1, True, 64), (1024, 0, True, 80)])
def test_bars_only_with_the_bars_fused_is_the_full_chain(glvlib, n, gl, avg, bars):
    """GLV_OP_BARS_ONLY where the transform kernel computes the bars itself (BASELINE configs[2]: N = 16384, gravity + the radial module's 80 bars; the bars
    module's chain; the GL_R16 chain + 80 bars) -- kernel classes 8 (float state) and 9 (GL_R16 state): magnitude, state and the row in LDS for the live blocks
    only.  The bars must equal those of a batch without the flag bit for bit over loud, quiet and silent updates in every kernel configuration of the size; one
    launch; fewer algorithmic bytes when the live class is taken (glv_batch_live_bins != 0), the same when it is not; glv_batch_gravity_state is refused."""
    import torch
    G = glvlib
    streams, F = (7 if n <= 8192 else 3), 3
    kw = dict(n=n, avg_frames=F if avg else 1, avg_window_kind=1 if gl else 0, gl_storage=gl, bars=bars)
    mask = G.OP_GRAVITY | (G.OP_AVERAGE if avg else 0) | G.OP_BARS
    ops = G.OP_FFT | mask | (G.OP_R16 if gl else 0)
    dt = torch.int16 if gl else torch.float32
    took_live = 0
    for variant in (0, 1):
        full, live = G.Batch(G.Params(**kw), streams, mask), G.Batch(G.Params(**kw), streams, mask | G.OP_BARS_ONLY)
        try:
            full.set_variant(variant); live.set_variant(variant)
        except G.GlvError:
            full.close(); live.close()
            continue
        L = live.live_bins()
        assert full.live_bins() == 0
        if L: assert 0.28 * n < L <= 0.5 * n and live.algorithmic_bytes(ops) < full.algorithmic_bytes(ops), (L, n)
        else: assert live.algorithmic_bytes(ops) == full.algorithmic_bytes(ops)
        took_live += bool(L)
        o_f = torch.full((streams * 2, bars), -1, dtype=dt, device="cuda"); o_l = torch.full_like(o_f, -2)
        for u in range(F + 3):
            pcm = (lcg_pcm_fast(9100 + u + n, streams * 2 * n) // (1, 8, 64)[u % 3]).astype(np.int16)
            if u == 2: pcm[:] = 0
            d_in = torch.from_numpy(pcm).cuda()
            full.process_s16(d_in, o_f, ops); live.process_s16(d_in, o_l, ops)
            assert live.last_launches() == full.last_launches()
            assert torch.equal(o_f.view(torch.int16), o_l.view(torch.int16)), (n, gl, avg, variant, u, L)
        if not gl and not avg:
            with pytest.raises(G.GlvError) as ei: live.gravity_state()
            assert ei.value.code == G.ERR_STATE
        full.close(); live.close()
    if n >= 2048: assert took_live >= 1, "no kernel configuration took the live class"


@pytest.mark.gpu
def test_bars_only_batch_refuses_parameters_that_need_state_it_did_not_keep(glvlib):
    """A GLV_OP_BARS_ONLY batch that has run its live class has no state beyond the live bins.  A glv_batch_set_params that takes the live class away (log_mode 2:
    the GL passes one by one, no live class) would make the full chain read state nobody maintained: refused (GLV_ERR_STATE), the batch left as it was (the next
    update still equals the unflagged batch's); after glv_batch_reset the change is accepted and the batch follows an unflagged one bit for bit.  smooth_factor can
    change freely mid-stream: smooth_audio() clamps its sample positions to [0, 1], so no factor makes a bar sample beyond scale_audio(1) n = 0.288 n."""
    import torch
    G = glvlib
    n, streams, F = 4096, 5, 3
    kw = dict(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)
    mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
    ops = G.OP_FFT | mask | G.OP_R16
    live, full = G.Batch(G.Params(smooth_factor=0.025, **kw), streams, mask | G.OP_BARS_ONLY), G.Batch(G.Params(smooth_factor=0.025, **kw), streams, mask)
    ol = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda"); of = torch.zeros_like(ol)
    def step(u):
        d = torch.from_numpy((lcg_pcm_fast(4242 + u, streams * 2 * n) // (1, 16)[u % 2]).astype(np.int16)).cuda()
        live.process_s16(d, ol, ops); full.process_s16(d, of, ops)
        assert _eq(ol, of), u
    for u in range(F + 1): step(u)
    assert live.live_bins() != 0
    for b in (live, full): b.set_params(G.Params(smooth_factor=0.3, **kw))           # any factor: the taps end at 0.288 n
    assert live.live_bins() != 0 and live.live_bins() <= 0.30 * n
    for u in range(F + 1, 2 * F + 2): step(u)
    with pytest.raises(G.GlvError) as ei:
        live.set_params(G.Params(smooth_factor=0.3, log_mode=2, **kw))               # the audit form: passes one by one, no live class
    assert ei.value.code == G.ERR_STATE and "live" in str(ei.value)
    assert live.live_bins() != 0
    step(100)                                                                        # ... and the batch is what it was
    live.reset(); full.reset()
    for b in (live, full): b.set_params(G.Params(smooth_factor=0.3, log_mode=2, **kw))
    assert live.live_bins() == 0
    for u in range(200, 200 + F + 1): step(u)
    live.close(); full.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["shipped_pipeline_64k", "configs2_8k"])
def test_bars_only_at_baseline_sizes_is_the_full_chain(glvlib, case):
    """BASELINE.json's full sizes through a size-independent property: at 65 536 streams of N = 4096 (configs[1]'s batch, GLava's shipped pipeline -> `sm`
    texels) and at 8 192 streams of N = 16384 (configs[2]: gravity + 80 radial bars) the GLV_OP_BARS_ONLY batch -- live kernel classes 7 / 8, the persistent
    workgroups' frame walk at full occupancy, the row block tails of the last workgroups -- gives the full chain's output bit for bit, update after update;
    a subset of streams is also checked against the oracle elsewhere (tests/test_gpu_parity.py), so equality here carries that to every stream.  Every row is
    compared on the device; a 64-bit checksum of checksums of the two outputs is compared as well (one number per update for the log)."""
    import torch
    G = glvlib
    if case == "shipped_pipeline_64k":
        n, streams, F, updates = 4096, 65536, 5, 6
        kw = dict(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)
        mask, ops, dt, width = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, torch.int16, n
    else:
        n, streams, F, updates = 16384, 8192, 1, 4
        kw = dict(n=n, bars=80)
        mask, ops, dt, width = G.OP_GRAVITY | G.OP_BARS, G.OP_FFT | G.OP_GRAVITY | G.OP_BARS, torch.float32, 80
    full = G.Batch(G.Params(**kw), streams, mask)
    live = G.Batch(G.Params(**kw), streams, mask | G.OP_BARS_ONLY)
    assert live.live_bins() > 0 and full.live_bins() == 0
    o_f = torch.zeros((streams * 2, width), dtype=dt, device="cuda"); o_l = torch.ones_like(o_f)
    gen = torch.Generator(device="cuda"); gen.manual_seed(4242)
    weights = torch.arange(1, streams * 2 + 1, dtype=torch.int64, device="cuda")
    for u in range(updates):
        pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda", generator=gen) // (1, 16, 4, 64)[u % 4]
        if u == 2: pcm.zero_()
        full.process_s16(pcm, o_f, ops); live.process_s16(pcm, o_l, ops)
        a = o_f.view(torch.int32) if dt == torch.float32 else o_f
        b = o_l.view(torch.int32) if dt == torch.float32 else o_l
        assert torch.equal(a, b), (case, u, int((a != b).sum()))
        rows_f = (a.to(torch.int64) & 0xffffffff).sum(dim=1); rows_l = (b.to(torch.int64) & 0xffffffff).sum(dim=1)      # a checksum per row, then one of those
        assert int((rows_f * weights).sum()) == int((rows_l * weights).sum())
        assert int(rows_f.max()) > 0 or u == 2
    full.close(); live.close()


def test_fused_gl_chain_over_random_parameters_equals_the_oracle_model(glvlib):
    """20 seeded draws of (n, F, window, gravity_step, ur, fft_scale, fft_cutoff, chain with or without the average pass) for the GL_R16 chain in one launch:
    texels equal to the oracle's transform_fft + glvo_gl_chain_r16 model, update after update -- gravity steps for which the pass is NOT the integer step
    (glv_tables.h gravity_r16_integer_step: float evaluation per texel) and steps larger than any texel among them"""
    import torch
    G = glvlib
    rng = np.random.default_rng(4711)
    for trial in range(20):
        n = int(rng.choice([512, 1024, 2048, 4096, 8192]))
        F = int(rng.choice([1, 2, 3, 5, 7]))
        win = bool(rng.integers(0, 2))
        gs = float(np.float32(rng.choice([0.0, 0.37, 4.2, 11.0, 90.0])))
        ur = float(np.float32(rng.uniform(30.0, 200.0)))
        sc, cut = float(np.float32(rng.uniform(2.0, 20.0))), float(np.float32(rng.uniform(0.0, 0.8)))
        avg = bool(rng.integers(0, 4)) and True
        streams = 3
        mask = G.OP_GRAVITY | (G.OP_AVERAGE if avg else 0)
        b = G.Batch(G.Params(n=n, avg_frames=F, avg_window=win, avg_window_kind=1, gl_storage=1, log_mode=0, gravity_step=gs, ur=ur, fft_scale=sc, fft_cutoff=cut), streams, mask)
        d_q = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
        store = np.zeros((streams * 2, n), np.float32); hist = np.zeros((streams * 2, F, n), np.float32)
        heads = [C.c_size_t(0) for _ in range(streams * 2)]
        for fr in range(F + 2):
            pcm = (lcg_pcm_fast(6200 + 13 * trial + fr, streams * 2 * n) // (8, 128)[fr % 2]).astype(np.int16)
            b.process_s16(torch.from_numpy(pcm).cuda(), d_q, G.OP_FFT | mask | G.OP_R16)
            assert b.last_launches() == 1
            gq = d_q.cpu().numpy().view(np.uint16)
            for u in range(streams):
                spec = StreamOracle(n, gravity=False, average=False, fft_scale=sc, fft_cutoff=cut).frame(pcm[u * 2 * n:(u + 1) * 2 * n])
                for c in range(2):
                    want = np.ascontiguousarray(spec[c])
                    Oracle.lib().glvo_gl_chain_r16(want, store[2 * u + c], hist[2 * u + c], C.byref(heads[2 * u + c]), n, F, int(win), int(avg), gs, ur)
                    bad = gq[2 * u + c] != Oracle.texels_r16(want)
                    assert not bad.any(), (trial, n, F, win, gs, ur, sc, cut, avg, fr, u, c, int(bad.sum()), np.flatnonzero(bad)[:4])
        b.close()
