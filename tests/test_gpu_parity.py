"""GPU (MI355X): parity of the HIP path against the oracle, called through the C ABI only.

Bars (BASELINE.json north_star):
  * bit-exact for the s16 -> f32 unpack/index path and for everything up to and including the
    FFT butterflies (GLV_OP_RAW exposes the pre-abs/log values), and for gravity/average applied
    to those raw values;
  * <= 1e-5 relative for the magnitudes after log/tilt (log_mode 0 is expected to be bit-identical
    except for last-ulp differences between the device's and glibc's fp64 log).
The oracle (oracle/liboracle.so) is the checker; it is pinned to the compiled reference by
tests/test_oracle.py.  /root/reference is never touched here.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle_lib import Oracle, Ref, RefStream, StreamOracle, chain_close, lcg_pcm_fast

pytestmark = pytest.mark.gpu

REL = 1e-5   # north_star tolerance for magnitudes


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rel_err(got, want):
    return np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want.astype(np.float64)), 1e-30)


@pytest.fixture(scope="module")
def G(glvlib):
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    assert glvlib.device_count() >= 1
    return glvlib


def run_batch(G, params, pcm_np, streams, ops, ops_mask=None, in_f32=False):
    import torch
    n = params.n
    b = G.Batch(params, streams, ops if ops_mask is None else ops_mask)
    d_in = torch.from_numpy(pcm_np).cuda()
    d_out = torch.full((streams * 2, n), float("nan"), dtype=torch.float32, device="cuda")
    (b.process_f32 if in_f32 else b.process_s16)(d_in, d_out, ops)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    b.close()
    return out


# ---- unpack --------------------------------------------------------------------------------------
def test_unpack_all_65536_inputs_bit_exact(G):
    v = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16)
    pcm = np.ascontiguousarray(np.stack([v, v[::-1]], axis=1)).reshape(-1)
    l = np.empty(65536, np.float32); r = np.empty(65536, np.float32)
    G.unpack_s16(pcm, 65536, 2, l, r)
    assert (bits(l) == bits(v.astype(np.float32) / np.float32(65535))).all()
    assert (bits(r) == bits(v[::-1].astype(np.float32) / np.float32(65535))).all()
    # mono mix: C integer (a+b)/2 truncating toward zero (fifo.c:99)
    G.unpack_s16(pcm, 65536, 1, l, r)
    wl, wr = Oracle.unpack_s16(pcm, channels=1)
    assert (bits(l) == bits(wl)).all() and (bits(r) == bits(wr)).all()
    # NULL pcm == poll-timeout zero fill (fifo.c:67-79)
    G.unpack_s16(None, 16, 2, l[:16], r[:16])
    assert not l[:16].any() and not r[:16].any()


# ---- FFT core: bit exact, every size -----------------------------------------------------------------
@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
def test_fft_raw_bit_exact_and_magnitude_within_tol(G, n):
    streams = 37                      # ragged: not a multiple of any slots-per-workgroup
    pcm = lcg_pcm_fast(2000 + n, streams * 2 * n)
    raw = run_batch(G, G.Params(n=n), pcm, streams, G.OP_FFT | G.OP_RAW)
    mag = run_batch(G, G.Params(n=n, log_mode=0), pcm, streams, G.OP_FFT)
    fast = run_batch(G, G.Params(n=n, log_mode=1), pcm, streams, G.OP_FFT)
    nbad = 0
    for u in range(streams):
        want, wraw = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n], want_raw=True)
        assert (bits(wraw) == bits(raw[2 * u:2 * u + 2])).all(), f"raw FFT differs, stream {u}"
        got = mag[2 * u:2 * u + 2]
        assert rel_err(got, want).max() <= REL
        nbad += int((bits(got) != bits(want)).sum())
        # strict log mode: at most one float ulp away anywhere
        assert np.abs(bits(got).astype(np.int64) - bits(want).astype(np.int64)).max() <= 1
        assert rel_err(fast[2 * u:2 * u + 2], want).max() <= REL
    # strict mode is bit-identical except for rare last-ulp fp64-log differences
    assert nbad <= 1e-4 * streams * 2 * n, nbad


@pytest.mark.parametrize("log_mode", [0, 1])
def test_golden_vectors_through_gpu(G, golden, log_mode):
    """Committed outputs of the compiled reference (tests/golden) vs the f32-planar GPU path, in the
    strict log mode (<= 1 float ulp) and the default hardware-log mode (<= 1e-5 relative)."""
    for key, seed, n in (("fft_n512_seed12857", 12857, 512), ("fft_n1024_seed13369", 13369, 1024),
                         ("fft_n4096_seed16441", 16441, 4096), ("fft_n16384_seed28729", 28729, 16384),
                         ("fft_kat_survey", 12345, 4096)):
        x = lcg_pcm_fast(seed, n).astype(np.float32) / np.float32(65535)
        st = G.State(G.Params(n=n, log_mode=log_mode))
        buf = x.copy()
        st.fft(buf)
        st.close()
        assert rel_err(buf, golden[key]).max() <= REL, key
        if log_mode == 0:
            assert np.abs(bits(buf).astype(np.int64) - bits(golden[key]).astype(np.int64)).max() <= 1, key
    # parameters and edge inputs
    st = G.State(G.Params(n=1024, fft_scale=3.0, fft_cutoff=0.7, log_mode=log_mode))
    buf = (lcg_pcm_fast(7, 1024).astype(np.float32) / np.float32(65535)); st.fft(buf)
    assert rel_err(buf, golden["fft_n1024_scale3_cut0p7"]).max() <= REL
    st.close()
    st = G.State(G.Params(n=1024, log_mode=log_mode))
    buf = np.zeros(1024, np.float32); st.fft(buf)
    assert (bits(buf) == bits(golden["fft_n1024_zeros"])).all()
    buf = np.full(1024, 32767 / 65535, np.float32); st.fft(buf)
    assert rel_err(buf, golden["fft_n1024_dc"]).max() <= REL
    st.close()


# ---- stateful operators ----------------------------------------------------------------------------
def oracle_raw_chain(n, F, win, gravity, average, frames_pcm, streams, gravity_step=4.2, ur=86.1328125):
    """fft(raw) -> gravity -> average on the oracle, all float ops => bit-comparable with EPI_RAW_STATE."""
    outs = []
    grav = np.zeros((streams * 2, n), np.float32)
    hist = np.zeros((streams * 2, F, n), np.float32)
    heads = [C.c_size_t(0) for _ in range(streams * 2)]
    for pcm in frames_pcm:
        out = np.empty((streams * 2, n), np.float32)
        for u in range(streams):
            _, raw = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n], want_raw=True)
            for c in range(2):
                row = np.ascontiguousarray(raw[c])
                if gravity: Oracle.gravity(row, grav[2 * u + c], gravity_step, ur)
                if average: Oracle.average(row, hist[2 * u + c], heads[2 * u + c], F, win)
                out[2 * u + c] = row
        outs.append(out)
    return outs


@pytest.mark.parametrize("F,win,ops", [(5, True, 2 | 4), (6, False, 2 | 4), (1, True, 2 | 4), (3, True, 4), (5, True, 2)])
def test_gravity_average_bit_exact_on_raw(G, F, win, ops):
    import torch
    n, streams, nframes = 1024, 5, 2 * F + 3
    frames = [lcg_pcm_fast(3000 + 17 * F + fr, streams * 2 * n) for fr in range(nframes)]
    want = oracle_raw_chain(n, F, win, bool(ops & 2), bool(ops & 4), frames, streams)
    b = G.Batch(G.Params(n=n, avg_frames=F, avg_window=win), streams, G.OP_FFT | ops)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    for fr in range(nframes):
        b.process_s16(torch.from_numpy(frames[fr]).cuda(), d_out, G.OP_FFT | G.OP_RAW | ops)
        got = d_out.cpu().numpy()
        assert (bits(got) == bits(want[fr])).all(), f"frame {fr}"
    b.close()


@pytest.mark.parametrize("n,F", [(512, 5), (4096, 5), (16384, 6)])
def test_full_chain_magnitudes(G, n, F):
    """fft -> gravity -> average with the log in between: tolerance 1e-5 relative (+ tiny absolute
    floor because gravity subtracts and can land arbitrarily close to zero)."""
    import torch
    streams, nframes = 3, F + 3
    sos = [StreamOracle(n, avg_frames=F) for _ in range(streams)]
    b = G.Batch(G.Params(n=n, avg_frames=F), streams, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    for fr in range(nframes):
        pcm = lcg_pcm_fast(4000 + n + fr, streams * 2 * n)
        b.process_s16(torch.from_numpy(pcm).cuda(), d_out, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE)
        got = d_out.cpu().numpy()
        for u in range(streams):
            want = sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n])
            assert sos[u].close(got[2 * u:2 * u + 2], want), (fr, u)
    b.close()


def test_golden_chain_through_single_stream_dropins(G, golden):
    """glv_fft / glv_gravity / glv_average used like the reference's operator table
    (render.c:2140-2156), checked against the compiled reference's committed outputs."""
    n = 1024
    for tag, F, win in (("F5w", 5, True), ("F6u", 6, False), ("F1w", 1, True)):
        want = golden[f"chain_n1024_{tag}"]
        p = G.Params(n=n, avg_frames=F, avg_window=win, ur=86.1328125)
        sl, sr = G.State(p), G.State(p)          # one state per channel, like the t_data[] slots
        fl, fr_ = G.State(p), G.State(p)         # fused variant
        peak = np.zeros((2, n), np.float32)                 # the largest magnitude every bin has held: chain_close's absolute term
        for fr in range(9):
            pcm = lcg_pcm_fast(500 + fr, 2 * n).reshape(n, 2)
            for c, (st, fs) in enumerate(((sl, fl), (sr, fr_))):
                x = pcm[:, c].astype(np.float32) / np.float32(65535)
                peak[c] = np.maximum(peak[c], Oracle.transform_fft(x.copy()))
                buf = x.copy(); st.fft(buf); st.gravity(buf); st.average(buf)
                assert chain_close(buf, want[fr, c], peak[c], n), (tag, fr, c)
                buf2 = x.copy(); fs.fft_gravity_average(buf2)
                assert chain_close(buf2, want[fr, c], peak[c], n), (tag, fr, c, "fused")
        for s in (sl, sr, fl, fr_): s.close()


def test_wrange_and_state_reset(G, golden):
    p = G.Params(n=512)
    st = G.State(p)
    b = np.zeros(512, np.float32)
    b[:256] = lcg_pcm_fast(31, 256).astype(np.float32) / np.float32(65535)
    st.wrange(b)
    assert (bits(b[:256]) == bits(golden["wrange_seed31"])).all()
    # gravity state persists, reset clears it
    x = np.full(512, 0.5, np.float32); st.gravity(x)
    y = np.zeros(512, np.float32); st.gravity(y)
    g = np.float32(4.2) * (np.float32(1.0) / np.float32(p.ur))
    assert (x == np.float32(0.5) - g).all() and (y == (np.float32(0.5) - g) - g).all()
    st.reset()
    z = np.zeros(512, np.float32); st.gravity(z)
    assert (z == np.float32(0.0) - g).all()
    st.close()


# ---- layouts / modes -----------------------------------------------------------------------------------
def test_f32_planar_batch_and_mono(G):
    n, streams = 2048, 9
    x = (np.random.default_rng(3).standard_normal((streams * 2, n)) * 0.3).astype(np.float32)
    raw = run_batch(G, G.Params(n=n), x, streams, G.OP_FFT | G.OP_RAW, in_f32=True)
    for r in range(streams * 2):
        _, wraw = Oracle.transform_fft(x[r], want_raw=True)
        assert (bits(raw[r]) == bits(wraw)).all()
    pcm = lcg_pcm_fast(88, streams * 2 * n)
    got = run_batch(G, G.Params(n=n, channels=1), pcm, streams, G.OP_FFT | G.OP_RAW)
    for u in range(streams):
        _, wraw = StreamOracle(n, channels=1, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n], want_raw=True)
        assert (bits(got[2 * u:2 * u + 2]) == bits(wraw)).all()


def test_pulse_f32_interleaved_input(G):
    """pulse_input.c:155-178 layout: float [streams][n][2]; stereo and (L+R)/2 mono; raw FFT bit-exact."""
    import torch
    n, streams = 2048, 7
    x = (np.random.default_rng(11).standard_normal((streams, n, 2)) * 0.3).astype(np.float32)
    for ch in (2, 1):
        b = G.Batch(G.Params(n=n, channels=ch), streams, G.OP_FFT)
        d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
        b.process_f32_stereo(torch.from_numpy(x).cuda(), d_out, G.OP_FFT | G.OP_RAW)
        got = d_out.cpu().numpy()
        b.close()
        for u in range(streams):
            pl = np.empty(n, np.float32); pr = np.empty(n, np.float32)
            Oracle.lib().glvo_unpack_f32(np.ascontiguousarray(x[u].reshape(-1)), n, ch, pl, pr)
            for c, row in enumerate((pl, pr)):
                _, want = Oracle.transform_fft(row, want_raw=True)
                assert (bits(got[2 * u + c]) == bits(want)).all(), (ch, u, c)


def test_prelude_bufscale_and_lerp(G):
    """rd_update prelude (render.c:1765-1809) on device buffers, bit-exact vs the restatement."""
    import torch
    rows, n_out = 6, 1024
    for k in (2, 4, 3):
        x = np.random.default_rng(k).standard_normal((rows, n_out * k)).astype(np.float32)
        d_out = torch.empty((rows, n_out), dtype=torch.float32, device="cuda")
        G.prelude_bufscale(torch.from_numpy(x).cuda(), d_out, rows, n_out, k)
        got = d_out.cpu().numpy()
        for r in range(rows):
            want = np.empty(n_out, np.float32)
            Oracle.lib().glvo_bufscale(np.ascontiguousarray(x[r]), want, n_out, k)
            assert (bits(got[r]) == bits(want)).all()
    s0 = np.random.default_rng(1).standard_normal(5000).astype(np.float32)
    e0 = np.random.default_rng(2).standard_normal(5000).astype(np.float32)
    for uratio, kc in ((0.3, 1), (0.3, 2), (0.6, 3)):      # the last one saturates at 1
        d_out = torch.empty(5000, dtype=torch.float32, device="cuda")
        G.prelude_lerp(torch.from_numpy(s0).cuda(), torch.from_numpy(e0).cuda(), d_out, 5000, uratio, kc)
        want = np.empty(5000, np.float32)
        Oracle.lib().glvo_lerp(s0, e0, want, 5000, uratio, kc)
        assert (bits(d_out.cpu().numpy()) == bits(want)).all()


def test_cpu_smooth_operator(G):
    """transform_smooth (render.c:694-718): bit-exact incl. the NaN the reference produces at t = 0."""
    for n, dist, ratio in ((512, 0.01, 4.0), (4096, 0.01, 4.0), (1024, 0.05, 2.0)):
        x = np.abs(np.random.default_rng(n).standard_normal(n)).astype(np.float32)
        x[::7] = 0
        want = x.copy(); Oracle.lib().glvo_smooth(want, n, dist, ratio)
        st = G.State(G.Params(n=n, smooth_distance=dist, smooth_ratio=ratio))
        got = x.copy(); st.smooth(got)
        st.close()
        assert ((bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))).all()


@pytest.mark.parametrize("n,dist,ratio,streams", [(512, 0.01, 4.0, 41), (4096, 0.01, 4.0, 37), (4096, 0.05, 1.0, 5), (32768, 0.01, 1.0, 2),
                                                  (1024, 0.2, 2.0, 70), (512, 0.01, 3.0, 32), (2048, 0.03, 1.5, 33), (256, 0.01, 4.0, 64),
                                                  (16384, 0.01, 4.0, 40), (4096, 0.6, 4.0, 48), (8192, 0.002, 1.0, 36)])
def test_cpu_smooth_operator_batched(G, n, dist, ratio, streams):
    """The batched transform_smooth -- 64 rows per wave with a sliding window of each in LDS (ring sizes 128 / 256 / 512),
    or, for fewer than 64 rows and for windows no ring holds, row prefixes in LDS: row counts that are and are not
    multiples of 64, output counts that are no multiple of a chunk, windows up to the whole row (ratio 1; 128 KiB of LDS at
    n = 32768), narrow and very wide windows; every row bit-equal to the oracle (NaN where the reference produces NaN),
    floats behind the written prefix untouched."""
    import torch
    rng = np.random.default_rng(n + streams)
    x = np.abs(rng.standard_normal((streams * 2, n))).astype(np.float32)
    x[:, ::7] = 0
    x[3, 5:40] = 0
    want = x.copy()
    for r_ in range(streams * 2):
        row = np.ascontiguousarray(want[r_]); Oracle.lib().glvo_smooth(row, n, dist, ratio); want[r_] = row
    b = G.Batch(G.Params(n=n, smooth_distance=dist, smooth_ratio=ratio), streams, 0)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.full_like(d_in, float("nan"))
    b.process_f32(d_in, d_out, G.OP_SMOOTH)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    b.close()
    assert ((bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))).all()


def test_gl_twin_average_and_bars(G):
    """a12 semantics (GL accel passes): Hamming-weighted newest-first average (average_pass.frag) and
    smooth_audio() bar sampling (smooth.glsl, radial/1.frag).  GLSL cannot run here, so both sides are
    restatements.  The bars follow the library's documented summation order (oracle glvo_bars_chunked): bit-equal on
    spectra given in HBM, within the magnitudes' own 1e-5 when the spectra come from the transform (hardware log)."""
    import torch
    n, streams, F, bars = 16384, 3, 5, 80
    p = G.Params(n=n, avg_frames=F, avg_window_kind=1, bars=bars)
    b = G.Batch(p, streams, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS)
    d_bars = torch.empty((streams * 2, bars), dtype=torch.float32, device="cuda")
    d_spec = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    grav = np.zeros((streams * 2, n), np.float32)
    hist = np.zeros((streams * 2, F, n), np.float32)
    heads = [C.c_size_t(0) for _ in range(streams * 2)]
    for fr in range(F + 2):
        pcm = (lcg_pcm_fast(7000 + fr, streams * 2 * n) // 64).astype(np.int16)      # keep magnitudes inside [0,1]-ish
        d_pcm = torch.from_numpy(pcm).cuda()
        b.process_s16(d_pcm, d_bars, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS)
        got_bars = d_bars.cpu().numpy()
        for u in range(streams):
            out = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n])
            for c in range(2):
                row = np.ascontiguousarray(out[c])
                Oracle.gravity(row, grav[2 * u + c])
                Oracle.lib().glvo_average_gl(row, hist[2 * u + c], C.byref(heads[2 * u + c]), n, F, 1)
                want = np.empty(bars, np.float32)
                Oracle.lib().glvo_bars_chunked(row, n, want, bars, 0.025)
                assert np.allclose(got_bars[2 * u + c], want, rtol=2e-5, atol=2e-6), (fr, u, c)   # magnitudes carry the 1e-5 log tolerance
    # bars of spectra already in HBM (glv_batch_bars)
    spec = np.abs(np.random.default_rng(5).standard_normal((streams * 2, n))).astype(np.float32) * 0.4
    b.bars(torch.from_numpy(spec).cuda(), d_bars)
    got = d_bars.cpu().numpy()
    for r in range(streams * 2):
        want = np.empty(bars, np.float32)
        Oracle.lib().glvo_bars_chunked(np.ascontiguousarray(spec[r]), n, want, bars, 0.025)
        assert (bits(got[r]) == bits(want)).all(), r          # the documented summation order, bit for bit
    b.close()


@pytest.mark.parametrize("ssz", [1024, 1000, 3000, 1004, 4, 4096])
def test_fifo_ring_mode(G, ssz):
    """glv_batch_ring_update_s16 == fifo.c:91-112 ring shift/append (+ :67-79 zero fill) followed by
    the transform of the whole window, for any sample_sz fifo.c accepts (fifo.c:38,81,91): ssz/4 stereo frames per
    update -- dividing n (1024), even but not dividing (1000, 3000), odd (1004 -> 251 frames: the window then starts
    at an odd frame of the device ring), a single frame, the whole window."""
    import torch
    n, streams, nf = 1024, 4, ssz // 4
    b = G.Batch(G.Params(n=n), streams, G.OP_FFT | G.OP_RING_S16)
    rl = np.zeros((streams, n), np.float32); rr = np.zeros((streams, n), np.float32)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    for step in range(7 if nf > 1 else 5):
        zero_fill = step == 3
        new = lcg_pcm_fast(600 + step, streams * nf * 2).reshape(streams, nf * 2)
        b.ring_update_s16(None if zero_fill else torch.from_numpy(new).cuda(), nf, d_out, G.OP_FFT | G.OP_RAW)
        got = d_out.cpu().numpy()
        for u in range(streams):
            chunk = np.ascontiguousarray(new[u])
            Oracle.lib().glvo_ring_update_s16(rl[u], rr[u], n, None if zero_fill else chunk.ctypes.data_as(C.c_void_p), nf, 2)
            _, wl = Oracle.transform_fft(rl[u], want_raw=True)
            _, wr = Oracle.transform_fft(rr[u], want_raw=True)
            assert (bits(got[2 * u]) == bits(wl)).all() and (bits(got[2 * u + 1]) == bits(wr)).all(), (step, u)
    b.close()


def test_error_behaviour(G):
    import torch
    b = G.Batch(G.Params(n=512), 2, G.OP_FFT | G.OP_RING_S16)
    d = torch.zeros(2 * 2 * 512, dtype=torch.int16, device="cuda")
    o = torch.zeros((4, 512), dtype=torch.float32, device="cuda")
    with pytest.raises(G.GlvError) as ei:
        b.process_s16(d, o, G.OP_FFT | G.OP_GRAVITY)      # state not allocated
    assert ei.value.code == G.ERR_STATE
    with pytest.raises(G.GlvError) as ei:
        b.process_s16(None, o, G.OP_FFT)
    assert ei.value.code == G.ERR_INVALID
    for bad in (0, 513):                                  # fifo.c accepts any sample_sz up to the window
        with pytest.raises(G.GlvError) as ei:
            b.ring_update_s16(d, bad, o, G.OP_FFT)
        assert ei.value.code == G.ERR_INVALID
    # a rejected update must not advance the ring: bad ops, then a good update equals a fresh batch's
    with pytest.raises(G.GlvError):
        b.ring_update_s16(d, 128, o, G.OP_FFT | G.OP_GRAVITY)
    with pytest.raises(G.GlvError):
        b.ring_update_s16(d, 128, None, G.OP_FFT)
    pcm = lcg_pcm_fast(31, 2 * 128 * 2)
    dn = torch.from_numpy(pcm).cuda()
    b.ring_update_s16(dn, 128, o, G.OP_FFT | G.OP_RAW)
    b2 = G.Batch(G.Params(n=512), 2, G.OP_FFT | G.OP_RING_S16)
    o2 = torch.zeros_like(o)
    b2.ring_update_s16(dn, 128, o2, G.OP_FFT | G.OP_RAW)
    torch.cuda.synchronize()
    assert torch.equal(o.view(torch.int32), o2.view(torch.int32))
    b.close(); b2.close()


# ---- BASELINE.json full size: 64K streams x N=4096 --------------------------------------------------
@pytest.mark.parametrize("log_mode", [1, 0])
def test_full_size_64k_streams_properties(G, log_mode):
    """configs[1]: 65536 stereo streams, N=4096, window+FFT+magnitude, in the default (hardware log) and
    the strict log mode.  Size-independent checks:
    (a) a random subset of >= 1024 streams against the oracle (magnitudes <= 1e-5 rel);
    (b) every stream that was given identical PCM produces identical output bits (indexing across
        the whole batch, all workgroups/slots);
    (c) no element left unwritten."""
    import torch
    n, streams = 4096, 65536
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    d_pcm = torch.randint(-32768, 32768, (streams, n * 2), dtype=torch.int16, device="cuda", generator=g)
    # plant duplicates of stream 7 at scattered positions
    dup = [7, 8, 1023, 4097, 32768, 65535, 40001]
    for s in dup[1:]:
        d_pcm[s] = d_pcm[7]
    d_out = torch.full((streams * 2, n), float("nan"), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n, log_mode=log_mode), streams, G.OP_FFT)
    b.process_s16(d_pcm, d_out, G.OP_FFT)
    torch.cuda.synchronize()
    assert not torch.isnan(d_out).any().item()
    ref7 = d_out[14:16]
    for s in dup[1:]:
        assert torch.equal(d_out[2 * s:2 * s + 2].view(torch.int32), ref7.view(torch.int32)), s
    rng = np.random.default_rng(99)
    subset = np.unique(np.concatenate([rng.integers(0, streams, 1040), [0, streams - 1]]))
    assert subset.size >= 1024
    idx = torch.from_numpy(subset).cuda()
    pcm_sub = d_pcm[idx].cpu().numpy()
    rows = torch.stack([2 * idx, 2 * idx + 1], dim=1).reshape(-1)
    out_sub = d_out[rows].cpu().numpy().reshape(subset.size, 2, n)
    worst, nbad = 0.0, 0
    for i in range(subset.size):
        want = StreamOracle(n, gravity=False, average=False).frame(pcm_sub[i])
        worst = max(worst, rel_err(out_sub[i], want).max())
        nbad += int((bits(out_sub[i]) != bits(want)).sum())
    assert worst <= REL, worst
    if log_mode == 0:                    # strict: bit-identical but for rare fp64 last-ulp ties
        assert nbad <= 1e-4 * subset.size * 2 * n, nbad
    b.close()


@pytest.mark.parametrize("n,F,win", [(256, 5, True), (1024, 5, True), (2048, 6, False), (4096, 5, True), (8192, 3, True),
                                     (16384, 5, True), (32768, 2, False)])
def test_against_the_compiled_reference_itself(G, n, F, win):
    """No restatement in between: the batched HIP path against oracle/_ref/libglvref.so -- the reference's own
    render.c compiled by oracle/Makefile, which travels to the GPU box as a built file -- on the same s16 PCM, seven
    updates of fft -> gravity -> average for 6 streams.  log_mode 0: identical bits, no tolerance;
    log_mode 1: north_star's 1e-5 (+ the floor gravity's subtraction needs)."""
    import torch
    if not os.path.exists(Ref.PATH):
        pytest.skip("oracle/_ref/libglvref.so was not built (needs /root/reference at build time)")
    streams, frames = 6, 7
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    for log_mode in (0, 1):
        rp = Ref.params(avg_frames=F, avg_window=win)
        refs = [RefStream(rp) for _ in range(streams)]
        peaks = [np.zeros((2, n), np.float32) for _ in range(streams)]      # per bin: the largest magnitude that has entered the chain (chain_close)
        b = G.Batch(G.Params(n=n, avg_frames=F, avg_window=int(win), log_mode=log_mode), streams, ops)
        d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
        for fr in range(frames):
            pcm = np.stack([lcg_pcm_fast(977 * n + 31 * s_ + fr, 2 * n) for s_ in range(streams)])
            if fr == 3: pcm[1] = 0                                   # a silent update (zero-fill, fifo.c:67-79)
            if fr == 4: pcm[2] = np.where(np.arange(2 * n) % 2 == 0, 32767, -32768)   # full scale
            b.process_s16(torch.from_numpy(pcm).cuda(), d_out, ops)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy().reshape(streams, 2, n)
            for s_ in range(streams):
                f = pcm[s_].astype(np.float32) / np.float32(65535)    # fifo.c:105-106 (pinned exhaustively elsewhere)
                want = refs[s_].frame_from_float(f[0::2], f[1::2])
                if log_mode == 0:
                    assert np.array_equal(bits(got[s_]), bits(want)), (n, fr, s_, int((bits(got[s_]) != bits(want)).sum()))
                else:
                    peaks[s_] = np.maximum(peaks[s_], np.stack([Oracle.transform_fft(f[0::2].copy()), Oracle.transform_fft(f[1::2].copy())]))
                    assert chain_close(got[s_], want, peaks[s_], n), (n, fr, s_)
        b.close()
        for r_ in refs: r_.close()


def test_buffers_beyond_4_GiB(G):
    """Addressing past 2^32 bytes: 196 608 streams at N=4096 (PCM 3 GiB, spectra 6 GiB, gravity state 6 GiB) and
    24 576 streams at N=32768 (spectra 6 GiB).  Rows behind the 4 GiB mark against the oracle, duplicates planted
    on both sides of it, nothing left unwritten; then a gravity update on the same batch size (state rows behind
    the mark are read and written)."""
    import torch
    for n, streams, ops_list in [(4096, 196608, ["fft", "grav"]), (32768, 24576, ["fft"])]:
        g = torch.Generator(device="cuda"); g.manual_seed(4242 + n)
        d_pcm = torch.randint(-32768, 32768, (streams, n * 2), dtype=torch.int16, device="cuda", generator=g)
        dup = [5, streams // 2 + 3, streams - 1, streams - 4097]
        for s_ in dup[1:]:
            d_pcm[s_] = d_pcm[5]
        subset = np.unique(np.concatenate([np.random.default_rng(3).integers(streams * 2 // 3, streams, 24), [0, streams - 1, streams - 2]]))
        idx = torch.from_numpy(subset).cuda()
        rows = torch.stack([2 * idx, 2 * idx + 1], dim=1).reshape(-1)
        pcm_sub = d_pcm[idx].cpu().numpy()
        for kind in ops_list:
            ops = G.OP_FFT if kind == "fft" else G.OP_FFT | G.OP_GRAVITY
            d_out = torch.full((streams * 2, n), float("nan"), dtype=torch.float32, device="cuda")
            assert d_out.numel() * 4 > (1 << 32)
            b = G.Batch(G.Params(n=n), streams, ops)
            sos = [StreamOracle(n, gravity=kind == "grav", average=False) for _ in subset]
            for upd in range(2 if kind == "grav" else 1):
                b.process_s16(d_pcm, d_out, ops)
                torch.cuda.synchronize()
                got = d_out[rows].cpu().numpy().reshape(subset.size, 2, n)
                for i in range(subset.size):
                    want = sos[i].frame(pcm_sub[i])
                    assert sos[i].close(got[i], want), (n, kind, upd, int(subset[i]))
            assert not torch.isnan(d_out).any().item()
            for s_ in dup[1:]:
                assert torch.equal(d_out[2 * s_:2 * s_ + 2].view(torch.int32), d_out[10:12].view(torch.int32)), (n, kind, s_)
            b.close()
            del d_out
        del d_pcm
        torch.cuda.empty_cache()


def test_full_size_stateful_chain_subset(G):
    """configs[1] size with the chain GLava's modules request (fft -> gravity -> average, F=5): three
    updates of 65536 streams, a random subset of streams against the stateful oracle every update."""
    import torch
    n, streams, F = 4096, 65536, 5
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    b = G.Batch(G.Params(n=n, avg_frames=F), streams, ops)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(7)
    subset = np.unique(np.concatenate([rng.integers(0, streams, 96), [0, 1, streams - 1]]))
    idx = torch.from_numpy(subset).cuda()
    rows = torch.stack([2 * idx, 2 * idx + 1], dim=1).reshape(-1)
    sos = {int(s): StreamOracle(n, avg_frames=F) for s in subset}
    g = torch.Generator(device="cuda")
    for fr in range(3):
        g.manual_seed(100 + fr)
        d_pcm = torch.randint(-32768, 32768, (streams, n * 2), dtype=torch.int16, device="cuda", generator=g)
        b.process_s16(d_pcm, d_out, ops)
        torch.cuda.synchronize()
        pcm_sub = d_pcm[idx].cpu().numpy()
        got = d_out[rows].cpu().numpy().reshape(subset.size, 2, n)
        for i, s in enumerate(subset):
            want = sos[int(s)].frame(pcm_sub[i])
            assert sos[int(s)].close(got[i], want), (fr, int(s))
    assert not torch.isnan(d_out).any().item()
    b.close()


def test_single_stream_dropin_latency(G):
    """The host-pointer drop-ins (the transform_* seam) must keep up with GLava's update rate with a wide margin:
    one stereo update = two glv_fft_gravity_average calls; 22050 Hz / 256 frames per update = 11.6 ms between
    updates.  Both stagings (mapped pinned host block, the default; device buffer + two copies, GLV_STAGING=copy)
    must give the same bits."""
    import time
    p = G.Params(n=4096)
    buf = [(lcg_pcm_fast(1 + c, 4096).astype(np.float32) / np.float32(65535)) for c in range(2)]
    res, per_update = {}, {}
    for mode in ("copy", "mapped"):
        if mode == "copy": os.environ["GLV_STAGING"] = "copy"
        else: os.environ.pop("GLV_STAGING", None)
        st = [G.State(p), G.State(p)]
        os.environ.pop("GLV_STAGING", None)
        outs = []
        for _ in range(5):
            outs = [buf[c].copy() for c in range(2)]
            for c in range(2): st[c].fft_gravity_average(outs[c])
        res[mode] = [bits(o) for o in outs]
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            for c in range(2): st[c].fft_gravity_average(buf[c].copy())
        per_update[mode] = (time.perf_counter() - t0) / reps
        for s_ in st: s_.close()
    print("single-stream stereo update through the host-pointer drop-ins: "
          f"{per_update['mapped'] * 1e6:.0f} us (mapped staging), {per_update['copy'] * 1e6:.0f} us (copy staging)")
    for c in range(2):
        assert np.array_equal(res["copy"][c], res["mapped"][c])
    assert per_update["mapped"] < 11.6e-3 / 4


@pytest.mark.parametrize("log_mode,bar", [(0, 0.0), (1, 1e-6)])
def test_magnitude_stage_every_float(G, log_mode, bar):
    """The magnitude stage alone (GLV_OP_MAGNITUDE, render.c:842-846) over EVERY float y = |b| + 1 in
    [1, 2^14) -- all the stage can see for n <= 16384 (|b| <= n/2 * 0.5 * max window).  log_mode 0 must be
    within one float ulp of (float)(log(y)/3) everywhere (and almost always equal); log_mode 1 (hardware
    log2) within 1e-6 relative, ten times inside the 1e-5 bar, including right above y = 1 where
    log(y) -> 0 makes a relative bound hardest."""
    import torch
    n, streams = 16384, 256                      # 2^23 values per call = one binade
    b = G.Batch(G.Params(n=n, fft_scale=0.0, fft_cutoff=0.0, log_mode=log_mode), streams, G.OP_FFT)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    k = np.arange(1 << 23, dtype=np.uint32)
    worst, nbad, total = 0.0, 0, 0
    for e in range(14):
        y = (np.uint32(0x3f800000 + (e << 23)) + k).view(np.float32)
        x = y - np.float32(1)                    # exact; x + 1.0f == y on the device
        assert (x + np.float32(1) == y).all()
        x[1::2] *= np.float32(-1)                # abs
        b.process_f32(torch.from_numpy(x.reshape(streams * 2, n)).cuda(), d_out, G.OP_MAGNITUDE)
        got = d_out.cpu().numpy().reshape(-1)
        want = (np.log(y.astype(np.float64)) / 3).astype(np.float32)      # tilt == 1 with fft_scale = cutoff = 0
        ulps = np.abs(bits(got).astype(np.int64) - bits(want).astype(np.int64))
        if log_mode == 0:
            assert ulps.max() <= 1, (e, int(ulps.max()))
            nbad += int((ulps != 0).sum())
        nz = want != 0
        worst = max(worst, float(np.abs((got[nz].astype(np.float64) - want[nz]) / want[nz]).max()))
        assert (got[~nz] == 0).all()
        total += y.size
    print(f"log_mode {log_mode}: max relative error over {total} floats {worst:.3e}; values differing from the reference float: {nbad}")
    assert worst <= (bar if log_mode else 1.3e-7)
    if log_mode == 0: assert nbad <= 1e-4 * total
    b.close()


def test_gravity_state_as_output(G):
    """A chain ending in gravity may leave its spectra in the state buffer only (d_out = NULL,
    render.c:733-734 stores the same value to both): identical bits to the explicit output, also for the
    bars computed from it."""
    import torch
    n, streams, bars = 2048, 11, 80
    ops = G.OP_FFT | G.OP_GRAVITY
    a = G.Batch(G.Params(n=n, bars=bars), streams, ops | G.OP_BARS)
    b = G.Batch(G.Params(n=n, bars=bars), streams, ops | G.OP_BARS)
    c = G.Batch(G.Params(n=n, bars=bars), streams, ops | G.OP_BARS)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    d_bars = torch.empty((streams * 2, bars), dtype=torch.float32, device="cuda")
    d_bars2 = torch.empty_like(d_bars)
    import ctypes
    for fr in range(4):
        d_pcm = torch.from_numpy(lcg_pcm_fast(900 + fr, streams * 2 * n)).cuda()
        a.process_s16(d_pcm, d_out, ops)
        b.process_s16(d_pcm, None, ops)
        torch.cuda.synchronize()
        # read the state buffer through torch: wrap the raw pointer
        st = torch.empty_like(d_out)
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(ctypes.c_void_p(st.data_ptr()), ctypes.c_void_p(b.gravity_state()), ctypes.c_size_t(st.numel() * 4), 3) == 0
        assert torch.equal(st.view(torch.int32), d_out.view(torch.int32)), fr
        c.process_s16(d_pcm, d_bars, ops | G.OP_BARS)            # spectra stay in c's state buffer
        a.bars(d_out, d_bars2)
        torch.cuda.synchronize()
        assert torch.equal(d_bars.view(torch.int32), d_bars2.view(torch.int32)), fr
    with pytest.raises(G.GlvError):
        a.process_s16(d_pcm, None, G.OP_FFT)                     # no state to hold the output
    for x in (a, b, c): x.close()


@pytest.mark.parametrize("n,F,bars", [(16384, 0, 80), (4096, 5, 80), (512, 5, 80), (1024, 5, 80), (1024, 0, 127), (1024, 0, 128),
                                      (2048, 0, 200), (8192, 5, 300), (256, 0, 30), (32768, 0, 80)])
def test_fused_bars_equal_unfused(G, n, F, bars):
    """GLV_OP_BARS inside the frame kernel (row in LDS, sizes whose rows are owned by whole waves) must
    give the bits of glv_batch_bars on the same spectra; N=512 / 256 exercise the unfused fallback, N=1024 with 80 and 127 bars
    the fused kernel with more bars than lanes per row (64), 128 bars the fallback behind it (bars + 1 > 2 * lanes);
    the bar counts also walk glv_bars_kernel's variants (work lists of 2, 4 and more steps)."""
    import torch
    streams = 5
    ops = G.OP_FFT | G.OP_GRAVITY | (G.OP_AVERAGE if F else 0)
    p = G.Params(n=n, bars=bars, avg_frames=max(F, 1), avg_window_kind=1)
    a, b = G.Batch(p, streams, ops | G.OP_BARS), G.Batch(p, streams, ops | G.OP_BARS)
    d_spec = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    d_b1 = torch.empty((streams * 2, bars), dtype=torch.float32, device="cuda")
    d_b2 = torch.full_like(d_b1, float("nan"))
    for fr in range(max(F, 1) + 2):
        d_pcm = torch.from_numpy((lcg_pcm_fast(77 + fr, streams * 2 * n) // 32).astype(np.int16)).cuda()
        a.process_s16(d_pcm, d_spec, ops)
        a.bars(d_spec, d_b1)
        b.process_s16(d_pcm, d_b2, ops | G.OP_BARS)
        torch.cuda.synchronize()
        assert torch.equal(d_b1.view(torch.int32), d_b2.view(torch.int32)), fr
    assert float(torch.nan_to_num(d_b1).abs().max()) > 0     # (n=256: bar 0 is 0 / 0 -- one tap of weight 0 -- in the shader too)
    a.close(); b.close()


# ---- more inputs / modes ------------------------------------------------------------------------------
from oracle_lib import tones_pcm  # noqa: E402


@pytest.mark.parametrize("n", [4096, 16384])
def test_tones_chain_parity(G, n):
    """SURVEY 8d "tones" PCM (harmonic, strongly peaked spectra -- the opposite of the noise worst case) through
    fft -> gravity -> average, consecutive frames of continuous signals, every stream against the oracle."""
    import torch
    streams, F, nframes = 6, 5, 7
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    b = G.Batch(G.Params(n=n, avg_frames=F), streams, ops)
    raw_b = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    sos = [StreamOracle(n, avg_frames=F) for _ in range(streams)]
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    for fr in range(nframes):
        pcm = np.concatenate([tones_pcm(40 * s + 3, n, fr) for s in range(streams)])
        d_pcm = torch.from_numpy(pcm).cuda()
        raw_b.process_s16(d_pcm, d_out, G.OP_FFT | G.OP_RAW)
        raw = d_out.cpu().numpy()
        b.process_s16(d_pcm, d_out, ops)
        got = d_out.cpu().numpy()
        for u in range(streams):
            _, wraw = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n], want_raw=True)
            assert (bits(raw[2 * u:2 * u + 2]) == bits(wraw)).all(), (fr, u)
            want = sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n])
            assert sos[u].close(got[2 * u:2 * u + 2], want), (fr, u)
    b.close(); raw_b.close()


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192, 16384])
def test_single_stream_mono_and_extreme_inputs(G, n):
    """streams = 1 (one workgroup, idle slots), mono mix (fifo.c:98-102), and edge PCM: silence (exact zeros out),
    full-scale alternating +-32768/32767 (largest magnitudes the stage can see), DC."""
    for ch in (2, 1):
        for kind in ("noise", "zeros", "alt", "dc"):
            if kind == "noise": pcm = lcg_pcm_fast(5 + n, 2 * n)
            elif kind == "zeros": pcm = np.zeros(2 * n, np.int16)
            elif kind == "alt": pcm = np.tile(np.array([-32768, 32767, 32767, -32768], np.int16), n // 2)
            else: pcm = np.full(2 * n, -32768, np.int16)
            raw = run_batch(G, G.Params(n=n, channels=ch), pcm, 1, G.OP_FFT | G.OP_RAW)
            mag = run_batch(G, G.Params(n=n, channels=ch, log_mode=0), pcm, 1, G.OP_FFT)
            fast = run_batch(G, G.Params(n=n, channels=ch), pcm, 1, G.OP_FFT)
            want, wraw = StreamOracle(n, channels=ch, gravity=False, average=False).frame(pcm, want_raw=True)
            assert (bits(raw) == bits(wraw)).all(), (ch, kind)
            assert np.abs(bits(mag).astype(np.int64) - bits(want).astype(np.int64)).max() <= 1, (ch, kind)
            assert rel_err(fast, want).max() <= REL, (ch, kind)
            assert np.isfinite(fast).all()
            if kind == "zeros": assert (bits(fast) == 0).all() and (bits(mag) == 0).all()


def test_ring_mode_full_window_equals_frame_mode(G):
    """A ring update that replaces the whole window (new_frames == n) must equal process_s16 on the same PCM,
    through the stateful chain with fused bars as well."""
    import torch
    n, streams, bars = 4096, 5, 80
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    p = G.Params(n=n, bars=bars)
    a, b = G.Batch(p, streams, ops | G.OP_BARS | G.OP_RING_S16), G.Batch(p, streams, ops | G.OP_BARS | G.OP_RING_S16)
    oa = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda"); ob = torch.empty_like(oa)
    ba = torch.empty((streams * 2, bars), dtype=torch.float32, device="cuda"); bb = torch.empty_like(ba)
    for fr in range(4):
        d_pcm = torch.from_numpy(lcg_pcm_fast(321 + fr, streams * 2 * n) // 16).cuda()
        if fr % 2 == 0:
            a.process_s16(d_pcm, oa, ops); b.ring_update_s16(d_pcm, n, ob, ops)
            torch.cuda.synchronize()
            assert torch.equal(oa.view(torch.int32), ob.view(torch.int32)), fr
        else:
            a.process_s16(d_pcm, ba, ops | G.OP_BARS); b.ring_update_s16(d_pcm, n, bb, ops | G.OP_BARS)
            torch.cuda.synchronize()
            assert torch.equal(ba.view(torch.int32), bb.view(torch.int32)), fr
    a.close(); b.close()


def test_config2_full_size_gravity_bars_subset(G):
    """configs[2]: N=16384, 8192 streams, fft + gravity + radial bin averaging (80 bars/channel), fused in the frame
    kernel.  Two updates; a random subset of streams against the oracle chain (spectra via d_out=NULL state, bars
    via the restatement of smooth.glsl); every bar written."""
    import torch, ctypes
    n, streams, bars = 16384, 8192, 80
    ops = G.OP_FFT | G.OP_GRAVITY
    b = G.Batch(G.Params(n=n, bars=bars), streams, ops | G.OP_BARS)
    d_bars = torch.full((streams * 2, bars), float("nan"), dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(3)
    subset = np.unique(np.concatenate([rng.integers(0, streams, 24), [0, streams - 1]]))
    idx = torch.from_numpy(subset).cuda()
    grav = {int(s): np.zeros((2, n), np.float32) for s in subset}
    peak = {int(s): np.zeros((2, n), np.float32) for s in subset}       # largest magnitude per bin so far (chain_close)
    hip = ctypes.CDLL("libamdhip64.so")
    g = torch.Generator(device="cuda")
    for fr in range(2):
        g.manual_seed(500 + fr)
        d_pcm = (torch.randint(-32768, 32768, (streams, n * 2), dtype=torch.int16, device="cuda", generator=g) // 64).to(torch.int16)
        b.process_s16(d_pcm, d_bars, ops | G.OP_BARS)
        torch.cuda.synchronize()
        assert not torch.isnan(d_bars).any().item()
        state = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
        assert hip.hipMemcpy(ctypes.c_void_p(state.data_ptr()), ctypes.c_void_p(b.gravity_state()), ctypes.c_size_t(state.numel() * 4), 3) == 0
        pcm_sub = d_pcm[idx].cpu().numpy()
        rows = torch.stack([2 * idx, 2 * idx + 1], dim=1).reshape(-1)
        got_bars = d_bars[rows].cpu().numpy().reshape(subset.size, 2, bars)
        got_spec = state[rows].cpu().numpy().reshape(subset.size, 2, n)
        for i, s in enumerate(subset):
            out = StreamOracle(n, gravity=False, average=False).frame(pcm_sub[i])
            for c in range(2):
                row = np.ascontiguousarray(out[c])
                peak[int(s)][c] = np.maximum(peak[int(s)][c], row)
                Oracle.gravity(row, grav[int(s)][c])
                assert chain_close(got_spec[i, c], row, peak[int(s)][c], n), (fr, int(s), c)
                # bars of the DEVICE's own state row in the documented order: bit for bit; of the oracle's row: the 1e-5 of the log
                want = np.empty(bars, np.float32)
                Oracle.lib().glvo_bars_chunked(np.ascontiguousarray(got_spec[i, c]), n, want, bars, 0.025)
                assert (bits(got_bars[i, c]) == bits(want)).all(), (fr, int(s), c)
                Oracle.lib().glvo_bars_chunked(row, n, want, bars, 0.025)
                # (a bar is a weighted mean of its taps: its error is at most the largest tap error, log1_rel * the largest peak)
                assert chain_close(got_bars[i, c], want, np.full(bars, peak[int(s)][c].max(), np.float32), n, rel=2e-5), (fr, int(s), c)
    b.close()


def test_bars_bits_equal_host_emulation(G, emu):
    """glv_bars_kernel (and through test_fused_bars_equal_unfused the fused epilogue) produce exactly the bits
    of the host emulation of the chunk arithmetic -- i.e. the DPP reduction adds in the documented order."""
    import torch
    from emu_lib import emu_bars
    for n in (512, 4096, 16384):
        bars, streams = 80, 3
        spec = np.abs(np.random.default_rng(n + 1).standard_normal((streams * 2, n))).astype(np.float32) * 0.5
        b = G.Batch(G.Params(n=n, bars=bars), streams, G.OP_FFT)
        d_bars = torch.empty((streams * 2, bars), dtype=torch.float32, device="cuda")
        b.bars(torch.from_numpy(spec).cuda(), d_bars)
        got = d_bars.cpu().numpy()
        want, _ = emu_bars(emu, spec, n, bars, groups=16)
        assert (bits(got) == bits(want)).all(), n
        for r in range(streams * 2):                            # ... and of the oracle's restatement of that order
            w2 = np.empty(bars, np.float32)
            Oracle.lib().glvo_bars_chunked(np.ascontiguousarray(spec[r]), n, w2, bars, 0.025)
            assert (bits(got[r]) == bits(w2)).all(), (n, r)
        b.close()


@pytest.mark.parametrize("n,bars,factor,phase", [(1024, 80, 0.025, 0.0), (1024, 1024, 0.025, 0.5), (2048, 33, 0.1, 0.0), (4096, 256, 0.01, 0.0),
                                                 (8192, 1, 0.025, 0.0), (16384, 160, 0.05, 0.0), (32768, 80, 0.025, 0.0), (256, 7, 0.3, 0.25)])
def test_bars_parameter_sweep_against_the_oracle_order(G, n, bars, factor, phase):
    """bar counts from 1 to n, narrow and wide windows, the pre-smoothing pass's texel centres (phase 0.5, bars == n), rows
    with values outside [0, 1], negative, infinite and NaN (clamped to [0, 1], NaN -> 0, as the documented contract says):
    glv_batch_bars == glvo_bars_chunked_at, bit for bit"""
    import torch
    streams = 2
    rng = np.random.default_rng(n + bars)
    spec = np.abs(rng.standard_normal((streams * 2, n))).astype(np.float32) * 0.6
    spec[1, ::5] = -spec[1, ::5]
    spec[2, ::11] = np.inf
    spec[2, 3::13] = np.nan
    spec[3, 7::17] = -np.inf
    spec[3, ::3] *= 1e-30
    b = G.Batch(G.Params(n=n, bars=bars, smooth_factor=factor, bar_phase=phase), streams, G.OP_FFT)
    d_bars = torch.full((streams * 2, bars), float("nan"), dtype=torch.float32, device="cuda")
    b.bars(torch.from_numpy(spec).cuda(), d_bars)
    got = d_bars.cpu().numpy()
    for r in range(streams * 2):
        want = np.empty(bars, np.float32)
        Oracle.lib().glvo_bars_chunked_at(np.ascontiguousarray(spec[r]), n, want, bars, factor, phase)
        assert (bits(got[r]) == bits(want)).all(), r
    assert np.isfinite(got).all()
    b.close()


def test_reference_host_through_shim(G):
    """The drop-in at the operator seam, on the reference's own types: oracle/_ref/libglvshim.so is the UNMODIFIED
    glava/render.c with integration/glava_hip_shim.c compiled into it.  handle_audio's
    CPU-path sequence fft -> gravity -> average with persistent t_data[] slots (render.c:2149-2153) runs once
    on the reference's operators and once on the *_hip operators (three launches, and the fused one)."""
    import ctypes
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libglvshim.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libglvshim.so not built (needs /root/reference at build time)")
    S = ctypes.CDLL(path)

    class P(ctypes.Structure):
        _fields_ = [("fft_scale", ctypes.c_float), ("fft_cutoff", ctypes.c_float), ("gravity_step", ctypes.c_float),
                    ("ur", ctypes.c_float), ("avg_frames", ctypes.c_ulong), ("avg_window", ctypes.c_int)]
    S.glvshim_run.argtypes = [ctypes.POINTER(P), ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    for n, F, win in ((4096, 5, 1), (1024, 6, 0)):
        p = P(10.2, 0.3, 4.2, 86.1328125, F, win)
        nframes = F + 3
        pcm = lcg_pcm_fast(4242 + n, nframes * 2 * n).reshape(nframes, n, 2)
        x = (pcm.astype(np.float32) / np.float32(65535)).transpose(0, 2, 1).copy()      # [frame][ch][n], fifo.c:105-106
        ref = x.copy()
        assert S.glvshim_run(ctypes.byref(p), 0, 0, ref.ctypes.data_as(ctypes.c_void_p), n, nframes) == 0
        for mode in (1, 2):
            for log_mode in (0, 1):
                got = x.copy()
                assert S.glvshim_run(ctypes.byref(p), mode, log_mode, got.ctypes.data_as(ctypes.c_void_p), n, nframes) == 0
                if log_mode == 0:
                    assert (bits(got) == bits(ref)).all(), (n, mode, log_mode)        # the bit-faithful log: the reference's own floats
                else:
                    peak = np.maximum.accumulate(np.stack([[Oracle.transform_fft(x[f, c].copy()) for c in range(2)] for f in range(nframes)]), axis=0)
                    assert chain_close(got, ref, peak, n), (n, mode, log_mode)
                assert np.isfinite(got).all()


@pytest.mark.parametrize("ch,nf", [(2, 256), (1, 256), (2, 375), (1, 251), (2, 250)])
def test_pulse_ring_mode(G, ch, nf):
    """glv_batch_ring_update_f32 == pulse_input.c:155-178: both rings shift left by new_frames, the new interleaved
    f32 frames are appended (channels == 1: (L + R) / 2 in float), then the whole window is transformed; any update
    size (odd, not dividing n) as pulse_input.c accepts any sample_sz."""
    import torch
    n, streams = 2048, 4
    b = G.Batch(G.Params(n=n, channels=ch), streams, G.OP_FFT | G.OP_RING_F32)
    rl = np.zeros((streams, n), np.float32); rr = np.zeros((streams, n), np.float32)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(17)
    for step in range(n // nf + 3):                      # wraps the device ring more than once
        new = (rng.standard_normal((streams, nf, 2)) * 0.3).astype(np.float32)
        b.ring_update_f32(torch.from_numpy(new).cuda(), nf, d_out, G.OP_FFT | G.OP_RAW)
        got = d_out.cpu().numpy()
        for u in range(streams):
            pl = np.empty(nf, np.float32); pr = np.empty(nf, np.float32)
            Oracle.lib().glvo_unpack_f32(np.ascontiguousarray(new[u].reshape(-1)), nf, ch, pl, pr)
            rl[u] = np.concatenate([rl[u][nf:], pl]); rr[u] = np.concatenate([rr[u][nf:], pr])
            _, wl = Oracle.transform_fft(rl[u], want_raw=True)
            _, wr = Oracle.transform_fft(rr[u], want_raw=True)
            assert (bits(got[2 * u]) == bits(wl)).all() and (bits(got[2 * u + 1]) == bits(wr)).all(), (step, u)
    with pytest.raises(G.GlvError):
        b.ring_update_f32(None, nf, d_out, G.OP_FFT)
    b.close()


def test_randomized_parameter_sweep(G):
    """Random operator parameters (fft_scale, fft_cutoff, gravity_step, ur, F in 1..16, window on/off, mono) at
    random sizes and ragged stream counts, several consecutive frames each, against the stateful oracle:
    raw FFT bit-exact, bit-faithful log mode within one float ulp before the state machines, chain <= 1e-5."""
    import torch
    rng = np.random.default_rng(20260923)
    for trial in range(24):
        n = int(rng.choice([512, 1024, 2048, 4096, 8192, 16384]))
        streams = int(rng.integers(1, 7))
        F = int(rng.integers(1, 17))
        win = bool(rng.integers(0, 2))
        ch = 1 if trial % 6 == 5 else 2
        kw = dict(fft_scale=float(np.float32(rng.uniform(0.0, 20.0))), fft_cutoff=float(np.float32(rng.uniform(0.0, 1.0))),
                  gravity_step=float(np.float32(rng.uniform(0.1, 9.0))), ur=float(np.float32(rng.uniform(20.0, 200.0))))
        ops = G.OP_FFT | (G.OP_GRAVITY if trial % 3 != 2 else 0) | (G.OP_AVERAGE if trial % 4 != 3 else 0)
        p = G.Params(n=n, channels=ch, avg_frames=F, avg_window=win, **kw)
        b = G.Batch(p, streams, ops)
        b0 = G.Batch(G.Params(n=n, channels=ch, log_mode=0, **kw), streams, G.OP_FFT)
        sos = [StreamOracle(n, channels=ch, avg_frames=F, avg_window=win, gravity=bool(ops & G.OP_GRAVITY),
                            average=bool(ops & G.OP_AVERAGE), **kw) for _ in range(streams)]
        d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
        d_strict = torch.empty_like(d_out)        # its own buffer: d_out is the gravity state of `b` between updates (output == state)
        for fr in range(min(F, 4) + 2):
            pcm = lcg_pcm_fast(int(rng.integers(1, 1 << 30)), streams * 2 * n)
            d_pcm = torch.from_numpy(pcm).cuda()
            b0.process_s16(d_pcm, d_strict, G.OP_FFT)
            strict = d_strict.cpu().numpy()
            b.process_s16(d_pcm, d_out, ops)
            got = d_out.cpu().numpy()
            for u in range(streams):
                seg = pcm[u * 2 * n:(u + 1) * 2 * n]
                mag = StreamOracle(n, channels=ch, gravity=False, average=False, **kw).frame(seg)
                assert np.abs(bits(strict[2 * u:2 * u + 2]).astype(np.int64) - bits(mag).astype(np.int64)).max() <= 1, (trial, fr, u)
                want = sos[u].frame(seg)
                assert sos[u].close(got[2 * u:2 * u + 2], want), (trial, n, F, win, ch, hex(ops), fr, u)
        b.close(); b0.close()


def _shim():
    import ctypes
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libglvshim.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libglvshim.so not built (needs /root/reference at build time)")
    S = ctypes.CDLL(path)
    S.glvshim_backend_run.restype = ctypes.c_long
    S.glvshim_backend_run.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                      ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    S.glvshim_hipfifo_publish_spectra.argtypes = [ctypes.c_int]
    return S


def run_backend(S, name, fifo, pcm, chunks, ssz, n, channels, max_events=64):
    """drive a registered audio backend (glava/fifo.h:22-44) against a named pipe; returns (snapshots [ev][2][n], zero_fill [ev])"""
    import ctypes
    snaps = np.zeros((max_events, 2, n), np.float32)
    zf = np.zeros(max_events, np.uint8)
    ev = S.glvshim_backend_run(name, fifo, pcm.ctypes.data_as(ctypes.c_void_p), chunks, ssz, n, channels,
                               snaps.ctypes.data_as(ctypes.c_void_p), zf.ctypes.data_as(ctypes.c_void_p), max_events)
    assert ev >= chunks, ev
    return snaps[:ev], zf[:ev]


@pytest.mark.parametrize("channels", [2, 1])
def test_hipfifo_backend_through_the_registry(G, golden, channels, tmp_path):
    """The audio-backend seam (glava/fifo.h:22-44) on real code: integration/hipfifo.c self-registers next to the
    reference's own "fifo" backend in audio_impls[] (both compiled into oracle/_ref/libglvshim.so against the
    unmodified reference headers); it is looked up by name and run on a thread against a named pipe exactly like
    glava.c:469-520 does.
      default mode   it publishes what struct audio_data defines (fifo.h:9-20): the time-domain sample rings of the
                     device-resident ring -- bit-equal to a replay of fifo.c's shift / append / zero fill on the oracle, to
                     the rings the REFERENCE's own fifo thread produced for the same PCM (tests/golden fifo_ch*_rings) and to
                     what the reference's backend publishes through the same driver in this very process;
      spectra mode   (opt-in) the finished spectra of those rings: transform_fft of the replayed rings."""
    S = _shim()
    fifo = str(tmp_path / "glv_hipfifo_test.fifo").encode()
    # -- default: the rings, on the golden run's parameters
    pcm = np.ascontiguousarray(golden[f"fifo_ch{channels}_pcm"])
    n, ssz = golden[f"fifo_ch{channels}_rings"].shape[2], 1024
    chunks, nf = pcm.size * 2 // ssz, ssz // 4
    S.glvshim_hipfifo_publish_spectra(0)
    snaps, zf = run_backend(S, b"hipfifo", fifo, pcm, chunks, ssz, n, channels)
    rl = np.zeros(n, np.float32); rr = np.zeros(n, np.float32)
    sent = 0
    landed = []
    for e in range(len(zf)):
        if zf[e]:
            Oracle.lib().glvo_ring_update_s16(rl, rr, n, None, nf, channels)
        else:
            chunk = np.ascontiguousarray(pcm[sent * (ssz // 2):(sent + 1) * (ssz // 2)]); sent += 1
            Oracle.lib().glvo_ring_update_s16(rl, rr, n, chunk.ctypes.data_as(C.c_void_p), nf, channels)
            landed.append(e)
        assert (bits(snaps[e, 0]) == bits(rl)).all() and (bits(snaps[e, 1]) == bits(rr)).all(), e
    assert sent == chunks
    gold_zf = golden[f"fifo_ch{channels}_zero_fill"]
    if not zf.any() and not gold_zf.any():                 # no poll timeout on either side: event for event the reference's rings
        assert (bits(snaps) == bits(golden[f"fifo_ch{channels}_rings"])).all()
    ref_snaps, ref_zf = run_backend(S, b"fifo", fifo, pcm, chunks, ssz, n, channels)
    if not zf.any() and not ref_zf.any():
        assert (bits(snaps) == bits(ref_snaps)).all()
    # -- opt-in: spectra
    n, ssz, chunks = 1024, 1024, 9
    nf = ssz // 4
    pcm = lcg_pcm_fast(777 + channels, chunks * ssz // 2)
    S.glvshim_hipfifo_publish_spectra(1)
    try:
        snaps, zf = run_backend(S, b"hipfifo", fifo, pcm, chunks, ssz, n, channels)
    finally:
        S.glvshim_hipfifo_publish_spectra(0)
    rl = np.zeros(n, np.float32); rr = np.zeros(n, np.float32)
    sent = 0
    for e in range(len(zf)):
        if zf[e]:
            Oracle.lib().glvo_ring_update_s16(rl, rr, n, None, nf, channels)
        else:
            chunk = np.ascontiguousarray(pcm[sent * (ssz // 2):(sent + 1) * (ssz // 2)]); sent += 1
            Oracle.lib().glvo_ring_update_s16(rl, rr, n, chunk.ctypes.data_as(C.c_void_p), nf, channels)
        wl = Oracle.transform_fft(rl.copy()); wr = Oracle.transform_fft(rr.copy())
        assert np.allclose(snaps[e, 0], wl, rtol=REL, atol=1e-7) and np.allclose(snaps[e, 1], wr, rtol=REL, atol=1e-7), e
    assert sent == chunks


@pytest.mark.parametrize("channels", [2, 1])
def test_hippulse_backend_through_the_registry(G, channels, tmp_path):
    """The PulseAudio twin of the seam test above: integration/hippulse.c (AUDIO_ATTACH(hippulse), found by name in the same
    registry) with its capture call replaced by a named pipe that delivers what pa_simple_read delivers -- sample_sz / 4
    interleaved stereo f32 frames per update.  What it publishes is the ring / deinterleave result of pulse_input.c:155-176
    (shift by sample_sz / 4, append, (L + R) / 2 for channels == 1), bit for bit; in spectra mode transform_fft of those rings."""
    S = _shim()
    fifo = str(tmp_path / "glv_hippulse_test.fifo").encode()
    n, ssz, chunks = 2048, 1024, 7
    nf = ssz // 4
    x = (np.random.default_rng(60 + channels).standard_normal((chunks, nf, 2)) * 0.3).astype(np.float32)
    x[2, 5, 0] = -0.0; x[3, 7, 1] = np.float32(1e-41)                       # the ring carries every float unchanged
    feed = np.ascontiguousarray(x).view(np.int16).reshape(-1)                # the driver takes bytes
    for spectra in (0, 1):
        S.glvshim_hipfifo_publish_spectra(spectra)
        try:
            snaps, zf = run_backend(S, b"hippulse", fifo, feed, chunks, ssz, n, channels)
        finally:
            S.glvshim_hipfifo_publish_spectra(0)
        assert len(zf) == chunks and not zf.any()
        rl = np.zeros(n, np.float32); rr = np.zeros(n, np.float32)
        for e in range(chunks):
            pl = np.empty(nf, np.float32); pr = np.empty(nf, np.float32)
            Oracle.lib().glvo_unpack_f32(np.ascontiguousarray(x[e].reshape(-1)), nf, channels, pl, pr)
            rl = np.concatenate([rl[nf:], pl]); rr = np.concatenate([rr[nf:], pr])
            if spectra:
                wl = Oracle.transform_fft(rl.copy()); wr = Oracle.transform_fft(rr.copy())
                assert np.allclose(snaps[e, 0], wl, rtol=REL, atol=1e-7) and np.allclose(snaps[e, 1], wr, rtol=REL, atol=1e-7), e
            else:
                assert (bits(snaps[e, 0]) == bits(rl)).all() and (bits(snaps[e, 1]) == bits(rr)).all(), e


def test_ring_append_and_planar_snapshot(G):
    """glv_batch_ring_append_* + glv_batch_ring_planar: the device rings read back in publishing order equal the oracle's
    fifo.c / pulse_input.c replay bit for bit -- any update size, mono mix, zero fill, many streams; appending and then
    transforming the ring elsewhere is the same as glv_batch_ring_update_*."""
    import torch
    n, streams = 1024, 23
    for ch in (2, 1):
        nob = G.Batch(G.Params(n=n, channels=ch), streams, G.OP_FFT)
        for call in (lambda: nob.ring_planar(torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")),
                     lambda: nob.ring_append_s16(None, 4), lambda: nob.ring_planar(torch.empty((streams, 2, n), device="cuda"), f32_ring=True)):
            with pytest.raises(G.GlvError) as ei:
                call()
            assert ei.value.code == G.ERR_STATE                           # rings exist only when the creation mask announced them
        nob.close()
        b = G.Batch(G.Params(n=n, channels=ch), streams, G.OP_FFT | G.OP_RING_S16 | G.OP_RING_F32)
        rl = np.zeros((streams, n), np.float32); rr = np.zeros((streams, n), np.float32)
        fl = np.zeros((streams, n), np.float32); fr_ = np.zeros((streams, n), np.float32)
        d_pl = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
        rng = np.random.default_rng(ch)
        for u, nf in enumerate([256, 250, 3, 1024, 999, 256]):
            zero = u == 2
            new = lcg_pcm_fast(60 + u, streams * nf * 2).reshape(streams, nf * 2)
            b.ring_append_s16(None if zero else torch.from_numpy(new).cuda(), nf)
            b.ring_planar(d_pl)
            got = d_pl.cpu().numpy()
            for s_ in range(streams):
                chunk = np.ascontiguousarray(new[s_])
                Oracle.lib().glvo_ring_update_s16(rl[s_], rr[s_], n, None if zero else chunk.ctypes.data_as(C.c_void_p), nf, ch)
            assert (bits(got[:, 0]) == bits(rl)).all() and (bits(got[:, 1]) == bits(rr)).all(), (ch, u)
            newf = (rng.standard_normal((streams, nf, 2)) * 0.3).astype(np.float32)
            b.ring_append_f32(torch.from_numpy(newf).cuda(), nf)
            b.ring_planar(d_pl, f32_ring=True)
            got = d_pl.cpu().numpy()
            for s_ in range(streams):
                pl = np.empty(nf, np.float32); pr = np.empty(nf, np.float32)
                Oracle.lib().glvo_unpack_f32(np.ascontiguousarray(newf[s_].reshape(-1)), nf, ch, pl, pr)
                fl[s_] = np.concatenate([fl[s_][nf:], pl]); fr_[s_] = np.concatenate([fr_[s_][nf:], pr])
            assert (bits(got[:, 0]) == bits(fl)).all() and (bits(got[:, 1]) == bits(fr_)).all(), (ch, u)
        b.close()


def test_f32_inputs_through_stateful_and_fused_paths(G):
    """The pipelined f32 input paths (planar rows, interleaved stereo, the PulseAudio ring) through the stateful
    kernel and the fused bars: same bits as the s16-free reference chain built from the stateless pass + the
    standalone operators (glv_post_kernel, glv_bars_kernel)."""
    import torch
    n, streams, bars, F = 4096, 6, 80, 3
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    p = G.Params(n=n, bars=bars, avg_frames=F)
    rng = np.random.default_rng(5150)
    chain = {k: G.Batch(p, streams, ops | G.OP_BARS | G.OP_RING_F32) for k in ("planar", "stereo", "ring", "planar_bars", "stereo_bars")}
    ref_fft = G.Batch(p, streams, ops | G.OP_BARS | G.OP_RING_F32)                    # stateless pass, then the operators one by one
    d_spec = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    d_ref = torch.empty_like(d_spec)
    d_bars = torch.empty((streams * 2, bars), dtype=torch.float32, device="cuda")
    d_bars_ref = torch.empty_like(d_bars)
    for fr in range(F + 2):
        x = (rng.standard_normal((streams, n, 2)) * 0.01).astype(np.float32)             # interleaved frames
        d_st = torch.from_numpy(x).cuda()
        d_pl = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()        # planar [streams][2][n]
        # reference: fft+magnitude (stateless kernel), then gravity+average on rows in HBM, then bars from HBM
        ref_fft.process_f32(d_pl, d_ref, G.OP_FFT)
        ref_fft.process_f32(d_ref, d_ref, G.OP_GRAVITY | G.OP_AVERAGE)
        ref_fft.bars(d_ref, d_bars_ref)
        chain["planar"].process_f32(d_pl, d_spec, ops)
        torch.cuda.synchronize()
        assert torch.equal(d_spec.view(torch.int32), d_ref.view(torch.int32)), ("planar", fr)
        chain["stereo"].process_f32_stereo(d_st, d_spec, ops)
        torch.cuda.synchronize()
        assert torch.equal(d_spec.view(torch.int32), d_ref.view(torch.int32)), ("stereo", fr)
        chain["ring"].ring_update_f32(d_st, n, d_spec, ops)                                # whole-window update
        torch.cuda.synchronize()
        assert torch.equal(d_spec.view(torch.int32), d_ref.view(torch.int32)), ("ring", fr)
        chain["planar_bars"].process_f32(d_pl, d_bars, ops | G.OP_BARS)
        torch.cuda.synchronize()
        assert torch.equal(d_bars.view(torch.int32), d_bars_ref.view(torch.int32)), ("planar bars", fr)
        chain["stereo_bars"].process_f32_stereo(d_st, d_bars, ops | G.OP_BARS)
        torch.cuda.synchronize()
        assert torch.equal(d_bars.view(torch.int32), d_bars_ref.view(torch.int32)), ("stereo bars", fr)
    for b in list(chain.values()) + [ref_fft]: b.close()


# ---- GLV_OP_R16: the GL_R16 texel output (render.c:521-524) -----------------------------------------------
def test_r16_every_float_bit_exact(G):
    """The quantiser on EVERY one of the 2^32 float bit patterns, through the product path (GLV_OP_R16 alone on
    planar f32 rows), against round-to-nearest-even of the exact clamp(x, 0, 1) * 65535 (oracle: glvo_unorm16)."""
    import torch
    n, streams = 16384, 4096                      # 2^27 floats per call, 32 calls
    per = streams * 2 * n
    b = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    d_out = torch.empty(per, dtype=torch.int16, device="cuda")
    base = torch.arange(per, dtype=torch.int64, device="cuda")
    bad = 0
    for c in range((1 << 32) // per):
        v = base + c * per                                        # bit patterns 0 .. 2^32-1, as signed 32-bit
        d_bits = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)
        x = d_bits.view(torch.float32)
        b.process_f32(x, d_out, G.OP_R16)
        want = torch.round(torch.nan_to_num(torch.clamp(x, 0.0, 1.0), nan=0.0).double() * 65535.0).to(torch.int32)
        got = d_out.view(torch.int16).to(torch.int32) & 0xffff
        bad += int((got != want).sum().item())
    b.close()
    assert bad == 0, f"{bad} of 2^32 floats quantise differently"
    # the same definition on the CPU side of the test infrastructure (oracle) for a sample incl. ties and edges
    xs = np.array([0.0, -0.0, 1.0, 1.5, -1.0, np.nan, np.inf, -np.inf, 0.5, 0.5 / 65535, 1.5 / 65535, 2.5 / 65535,
                   np.float32(1) - np.float32(2 ** -24), 1e-30, 7.62951e-06], np.float32)
    want = Oracle.texels_r16(xs)
    st = G.State(G.Params(n=512))
    buf = np.zeros(512, np.float32); buf[:xs.size] = xs
    tex = np.empty(512, np.uint16)
    st.texels_r16(buf, tex)
    st.close()
    assert (tex[:xs.size] == want).all(), (tex[:xs.size], want)


@pytest.mark.parametrize("n", [512, 1024, 4096, 8192, 16384])
def test_r16_fused_output_equals_quantised_f32_output(G, n):
    """fft -> R16 and fft -> gravity -> average -> R16: the texels are exactly the quantised f32 output of the same
    chain (bit for bit), the f32 state is untouched by the quantisation, and they agree with the oracle's texels of the
    reference-exact spectrum up to one step where the 1e-5 magnitude tolerance straddles a rounding boundary."""
    import torch
    streams, F = 19, 5
    for log_mode in (0, 1):
        p = G.Params(n=n, log_mode=log_mode, avg_frames=F)
        for ops, mask in ((G.OP_FFT, G.OP_FFT), (G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE, G.OP_GRAVITY | G.OP_AVERAGE),
                          (G.OP_FFT | G.OP_GRAVITY, G.OP_GRAVITY)):
            bf, bq = G.Batch(p, streams, mask), G.Batch(p, streams, mask)
            d_f = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
            d_q = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
            sos = [StreamOracle(n, avg_frames=F, gravity=bool(ops & G.OP_GRAVITY), average=bool(ops & G.OP_AVERAGE)) for _ in range(streams)]
            for fr in range(3):
                pcm = lcg_pcm_fast(4242 + 17 * fr + n, streams * 2 * n)
                d_pcm = torch.from_numpy(pcm).cuda()
                bf.process_s16(d_pcm, d_f, ops)
                bq.process_s16(d_pcm, d_q, ops | G.OP_R16)
                torch.cuda.synchronize()
                f = d_f.cpu().numpy()
                q = d_q.cpu().numpy().view(np.uint16)
                assert (q == Oracle.texels_r16(f)).all(), (n, log_mode, ops, fr)
                for u in range(streams):
                    want = Oracle.texels_r16(sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n]))
                    d = np.abs(q[2 * u:2 * u + 2].astype(np.int32) - want.astype(np.int32))
                    assert d.max() <= 1 and (d != 0).mean() < 2e-2, (n, log_mode, ops, fr, u, d.max(), (d != 0).mean())
            bf.close(); bq.close()


def test_r16_post_kernel_and_errors(G):
    """R16 behind the operators on planar rows (glv_post_kernel): wrange / gravity drop-in chains; invalid combinations."""
    import torch
    n, streams = 2048, 5
    rng = np.random.default_rng(5)
    x = (rng.random((streams * 2, n), dtype=np.float32) * 3 - 1).astype(np.float32)
    d_x = torch.from_numpy(x).cuda()
    d_q = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
    b = G.Batch(G.Params(n=n), streams, G.OP_GRAVITY)
    b.process_f32(d_x, d_q, G.OP_WRANGE | G.OP_R16)
    torch.cuda.synchronize()
    assert (d_q.cpu().numpy().view(np.uint16) == Oracle.texels_r16((x + np.float32(1)) / np.float32(2))).all()
    grav = np.zeros_like(x)
    want = x.copy()
    for r in range(streams * 2): Oracle.gravity(want[r], grav[r])
    b.process_f32(d_x, d_q, G.OP_GRAVITY | G.OP_R16)
    torch.cuda.synchronize()
    assert (d_q.cpu().numpy().view(np.uint16) == Oracle.texels_r16(want)).all()
    for bad in (G.OP_FFT | G.OP_RAW | G.OP_R16, G.OP_FFT | G.OP_SMOOTH | G.OP_R16):
        with pytest.raises(G.GlvError):
            b.process_f32(d_x, d_q, bad)
    with pytest.raises(G.GlvError):
        b.process_f32(d_x, None, G.OP_GRAVITY | G.OP_R16)
    b.close()


def test_parameters_the_reference_tolerates(G):
    """the reference takes any ur (render.c:2387 sets it to 0 after an interval without updates: the gravity step
    becomes infinite and the output -inf) and any avg_frames; neither may be an error here."""
    import torch
    n, streams = 1024, 3
    pcm = lcg_pcm_fast(8, streams * 2 * n)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n, ur=0.0), streams, G.OP_GRAVITY)
    b.process_s16(d_pcm, d_out, G.OP_FFT | G.OP_GRAVITY)
    got = d_out.cpu().numpy()
    so = StreamOracle(n, ur=0.0, average=False)
    with np.errstate(all="ignore"):
        want = so.frame(pcm[:2 * n])
    assert np.isneginf(want).all() and (bits(got[:2]) == bits(want)).all()
    b.close()
    F = 23                                            # > the 16 of round 1; GLV_MAX_AVG_FRAMES is 64
    b = G.Batch(G.Params(n=n, avg_frames=F, log_mode=0), streams, G.OP_GRAVITY | G.OP_AVERAGE)
    sos = [StreamOracle(n, avg_frames=F) for _ in range(streams)]
    for fr in range(F + 3):
        pcm = lcg_pcm_fast(900 + fr, streams * 2 * n)
        b.process_s16(torch.from_numpy(pcm).cuda(), d_out, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE)
        got = d_out.cpu().numpy()
        for u in range(streams):
            want = sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n])
            assert (bits(got[2 * u:2 * u + 2]) == bits(want)).all(), (fr, u)          # log_mode 0: the reference's bits
    # one `applied` buffer per chain: mixing fused and unfused gravity on one batch is refused, a reset allows it again
    with pytest.raises(G.GlvError) as ei:
        b.process_s16(d_pcm, d_out, G.OP_FFT | G.OP_GRAVITY)
    assert ei.value.code == G.ERR_STATE
    b.reset()
    b.process_s16(d_pcm, d_out, G.OP_FFT | G.OP_GRAVITY)
    b.close()


@pytest.mark.parametrize("n,F,win", [(512, 5, True), (4096, 5, True), (8192, 6, False), (1024, 1, True)])
def test_log_mode_0_chain_bit_exact_end_to_end(G, n, F, win):
    """with the bit-faithful log (log_mode 0) the WHOLE chain fft -> gravity -> average equals the oracle bit for
    bit over many frames -- no tolerance of any kind -- except on values where the fp64 log of the device table and
    glibc's differ in the last float ulp of the magnitude (none observed on these inputs; the exhaustive magnitude test
    bounds them)."""
    import torch
    streams = 11
    b = G.Batch(G.Params(n=n, avg_frames=F, avg_window=win, log_mode=0), streams, G.OP_GRAVITY | G.OP_AVERAGE)
    sos = [StreamOracle(n, avg_frames=F, avg_window=win) for _ in range(streams)]
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    for fr in range(2 * F + 3):
        pcm = lcg_pcm_fast(5150 + 7 * fr + n, streams * 2 * n)
        b.process_s16(torch.from_numpy(pcm).cuda(), d_out, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE)
        got = d_out.cpu().numpy()
        for u in range(streams):
            want = sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n])
            assert (bits(got[2 * u:2 * u + 2]) == bits(want)).all(), (fr, u, int((bits(got[2 * u:2 * u + 2]) != bits(want)).sum()))
    b.close()


def test_chain_over_random_parameters_bit_exact(G):
    """24 seeded draws of every scalar knob the chain reads -- n, channels, fft_scale, fft_cutoff, gravity_step, ur, avg_frames, avg_window, the operator subset --
    with the bit-faithful log: fft (-> gravity) (-> average) on s16 frames against the oracle (itself bit-equal to the compiled reference at every size,
    tests/test_oracle.py), bit for bit over 2 F + 2 updates of loud, quiet and silent frames.  The parametrised tests above fix the shipped values; this walks
    around them (tilt slopes that cross zero, gravity steps larger than the signal, one averaging frame, mono mix)."""
    import torch
    rng = np.random.default_rng(6060)
    for trial in range(24):
        n = int(rng.choice([256, 512, 1024, 2048, 4096, 8192, 16384]))
        channels = int(rng.choice([2, 2, 1]))
        F = int(rng.choice([1, 2, 3, 5, 8]))
        win = bool(rng.integers(0, 2))
        kw = dict(fft_scale=float(np.float32(rng.uniform(0.5, 30.0))), fft_cutoff=float(np.float32(rng.uniform(0.0, 0.95))),
                  gravity_step=float(np.float32(rng.choice([0.0, 0.3, 4.2, 40.0]))), ur=float(np.float32(rng.uniform(20.0, 240.0))))
        chain = int(rng.integers(0, 3))                          # 0: fft   1: fft + gravity   2: fft + gravity + average
        mask = (G.OP_GRAVITY if chain >= 1 else 0) | (G.OP_AVERAGE if chain == 2 else 0)
        ops = G.OP_FFT | mask
        streams = 5 if n <= 4096 else 2
        b = G.Batch(G.Params(n=n, channels=channels, avg_frames=F, avg_window=win, log_mode=0, **kw), streams, mask if mask else G.OP_FFT)
        sos = [StreamOracle(n, channels=channels, avg_frames=F, avg_window=win, gravity=chain >= 1, average=chain == 2, **kw) for _ in range(streams)]
        d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
        for fr in range(2 * F + 2):
            pcm = (lcg_pcm_fast(777 + 31 * trial + fr, streams * 2 * n) // (1, 64, 8)[fr % 3]).astype(np.int16)
            if fr == 1: pcm[:] = 0
            b.process_s16(torch.from_numpy(pcm).cuda(), d_out, ops)
            got = d_out.cpu().numpy()
            for u in range(streams):
                want = sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n])
                diff = bits(got[2 * u:2 * u + 2]) != bits(want)
                assert not diff.any(), (trial, n, channels, F, win, kw, chain, fr, u, int(diff.sum()), np.argwhere(diff)[:3].tolist())
        b.close()
