"""GPU (MI355X): what round 3 added to the parity suite.

  * BASELINE configs[4] as it is benchmarked: the five size classes N in {512 ... 8192} on five HIP streams AT ONCE -- every
    class's raw FFT bit-exact against the oracle, every class's magnitudes and stateful chains bit-identical to the same batch
    run alone (shared process-wide wisdom, per-instantiation launch attributes, table uploads: nothing may leak between them);
  * f32 inputs holding -0.0, denormals, +-Inf and NaN against the COMPILED reference (oracle/_ref/libglvref.so);
  * output == state for chains ending in gravity (20 N bytes per frame), GLV_OP_PRIVATE_STATE;
  * gl_storage chains with GLV_OP_SMOOTH / GLV_OP_RAW (both were dropped by the split path);
  * run-time selection of the kernel configuration (glv_inst.hip Tuned<K, V>): every variant gives the bits of variant 0.
"""
import ctypes as C

import numpy as np
import pytest

from oracle_lib import Oracle, Ref, StreamOracle, lcg_pcm_fast

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def G(glvlib):
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    return glvlib


# ---- BASELINE configs[4]: mixed sizes on concurrent HIP streams -----------------------------------------------------------
CLASSES = (512, 1024, 2048, 4096, 8192)


@pytest.mark.parametrize("log_mode", [1, 0])
def test_config4_mixed_sizes_on_concurrent_streams(G, log_mode):
    """tools/configs_bench.py / bench.py `configs[4]`: one batch per size class, equal bytes per class, each on its own HIP
    stream, all five in flight together, several rounds back to back.  Checked per class:
      raw FFT (GLV_OP_RAW) of a subset of streams        == oracle, bit for bit
      magnitudes of EVERY stream                         == the same batch's output when it ran alone, bit for bit
      fft -> gravity -> average over 4 updates           == the same chain run alone, bit for bit (state advanced correctly)
    and the first concurrent round is also the first launch of each class in this configuration (attribute set-up, table
    uploads and wisdom look-ups happen while the other classes are already running)."""
    import torch
    total = 1 << 22                                   # real samples per channel and class: 8192 streams at N=512 ... 512 at N=8192
    F, updates = 5, 4
    cls = []
    for n in CLASSES:
        s = total // n
        p = G.Params(n=n, log_mode=log_mode, avg_frames=F)
        pcm = [torch.from_numpy(lcg_pcm_fast(7000 + n + u, s * 2 * n)).cuda() for u in range(updates)]
        cls.append(dict(n=n, s=s, p=p, pcm=pcm, st=torch.cuda.Stream()))
    # alone, one class after the other on the default stream: the reference bits for "concurrent == alone"
    for c in cls:
        n, s = c["n"], c["s"]
        b = G.Batch(c["p"], s, G.OP_FFT)
        c["alone_mag"] = torch.empty((s * 2, n), dtype=torch.float32, device="cuda")
        b.process_s16(c["pcm"][0], c["alone_mag"], G.OP_FFT)
        b.close()
        bc = G.Batch(c["p"], s, G.OP_GRAVITY | G.OP_AVERAGE)
        c["alone_chain"] = []
        for u in range(updates):
            o = torch.empty((s * 2, n), dtype=torch.float32, device="cuda")
            bc.process_s16(c["pcm"][u], o, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE)
            c["alone_chain"].append(o)
        bc.close()
    torch.cuda.synchronize()
    # concurrent: fresh batches (first launches race with each other), every class on its own stream
    for c in cls:
        n, s = c["n"], c["s"]
        c["b_raw"], c["b_mag"] = G.Batch(c["p"], s, G.OP_FFT), G.Batch(c["p"], s, G.OP_FFT)
        c["b_chain"] = G.Batch(c["p"], s, G.OP_GRAVITY | G.OP_AVERAGE)
        c["raw"] = torch.empty((s * 2, n), dtype=torch.float32, device="cuda")
        c["mag"] = torch.empty((s * 2, n), dtype=torch.float32, device="cuda")
        c["chain"] = [torch.empty((s * 2, n), dtype=torch.float32, device="cuda") for _ in range(updates)]
    torch.cuda.synchronize()
    for u in range(updates):                           # round u: all five classes issued before any is waited for
        for c in cls:
            st = c["st"].cuda_stream
            if u == 0:
                c["b_raw"].process_s16(c["pcm"][0], c["raw"], G.OP_FFT | G.OP_RAW, st)
                c["b_mag"].process_s16(c["pcm"][0], c["mag"], G.OP_FFT, st)
            c["b_chain"].process_s16(c["pcm"][u], c["chain"][u], G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE, st)
    torch.cuda.synchronize()
    rng = np.random.default_rng(4)
    for c in cls:
        n, s = c["n"], c["s"]
        assert torch.equal(c["mag"].view(torch.int32), c["alone_mag"].view(torch.int32)), (n, "magnitudes differ from the solo run")
        for u in range(updates):
            assert torch.equal(c["chain"][u].view(torch.int32), c["alone_chain"][u].view(torch.int32)), (n, u, "chain differs from the solo run")
        subset = np.unique(np.concatenate([[0, s - 1], rng.integers(0, s, 14)]))
        raw = c["raw"].cpu().numpy()
        pcm0 = c["pcm"][0].cpu().numpy()
        for sidx in subset:
            _, want = StreamOracle(n, gravity=False, average=False).frame(pcm0[sidx * 2 * n:(sidx + 1) * 2 * n], want_raw=True)
            assert (bits(raw[2 * sidx:2 * sidx + 2]) == bits(want)).all(), (n, int(sidx))
        # and the chain of a few streams against the oracle itself (1e-5; bit-exact with the bit-faithful log)
        for sidx in subset[:4]:
            so = StreamOracle(n, avg_frames=F)
            for u in range(updates):
                want = so.frame(c["pcm"][u].cpu().numpy()[sidx * 2 * n:(sidx + 1) * 2 * n])
                got = c["chain"][u][2 * sidx:2 * sidx + 2].cpu().numpy()
                if log_mode == 0:
                    assert (bits(got) == bits(want)).all(), (n, int(sidx), u)
                else:
                    assert so.close(got, want), (n, int(sidx), u)
        for k in ("b_raw", "b_mag", "b_chain"):
            c[k].close()


# ---- f32 special values against the compiled reference ---------------------------------------------------------------
def _special_rows(n, rng):
    """planar f32 rows: ordinary audio with special values planted"""
    rows = []
    base = lambda: (rng.standard_normal(n) * 0.25).astype(np.float32)          # noqa: E731
    r = base(); r[::7] = -0.0; rows.append(("negative zeros", r))
    r = np.full(n, -0.0, np.float32); rows.append(("all -0.0", r))
    r = np.zeros(n, np.float32); rows.append(("all +0.0", r))
    r = base(); r[5::11] = np.float32(1e-41); r[6::13] = np.float32(-3e-45); rows.append(("denormals", r))
    r = (rng.standard_normal(n) * 1e-39).astype(np.float32); rows.append(("all denormal", r))
    r = base(); r[n // 3] = np.inf; rows.append(("one +Inf", r))
    r = base(); r[2] = -np.inf; r[3] = np.inf; rows.append(("-Inf and +Inf in one complex point", r))
    r = base(); r[n - 1] = np.nan; rows.append(("one NaN", r))
    r = base(); r[10] = np.nan; r[11] = np.inf; r[500] = -0.0; rows.append(("NaN, Inf and -0", r))
    r = base(); r[::2] = np.float32(3.0e38); r[1::2] = np.float32(-3.0e38); rows.append(("overflowing sums", r))
    return rows


def _same_value(got, want, exact):
    """NaN where the reference has NaN (payload and sign of a NaN are not part of the contract), otherwise equal bits
    (exact) or <= 1e-5 relative; infinities must match in sign"""
    gn, wn = np.isnan(got), np.isnan(want)
    if not (gn == wn).all():
        return False
    ok = ~wn
    if exact:
        return bool((bits(got)[ok] == bits(want)[ok]).all())
    inf = np.isinf(want) & ok
    if not (got[inf] == want[inf]).all():
        return False
    fin = ok & ~inf
    return bool(np.allclose(got[fin], want[fin], rtol=1e-5, atol=0.0))      # magnitudes only (no gravity in front): purely relative


@pytest.mark.parametrize("n", [1024, 4096, 16384])
def test_f32_special_values_against_the_compiled_reference(G, ref, n):
    """VERDICT r2 item 3a.  Planar (glava.c:528-537 snapshot) and interleaved (pulse_input.c:155-178) f32 input with -0.0,
    denormals, +-Inf and NaN, against the reference's own transform_fft (oracle/_ref/libglvref.so, not the restatement):
      magnitudes   log_mode 0: the reference's bits wherever it is not NaN, NaN exactly where it is NaN;
                   log_mode 1: <= 1e-5, same NaN / Inf pattern
      raw FFT      the oracle's bits (NaN pattern equal), with ONE documented exception: the sign of an exact zero.  The
                   kernels evaluate the unit twiddle (1, +0) of an f32 row with the reference's full multiply-add form (so
                   that 0 * Inf = NaN appears where the reference has it), and every other operation is the reference's, so
                   even that exception never shows on these rows -- the test asserts |got| == |want| bitwise and counts the
                   sign-of-zero differences, which must be none.
    (s16 input cannot hold any of these values: samples are finite and v/65535 * w is never -0.)"""
    import torch
    rng = np.random.default_rng(n)
    rows = _special_rows(n, rng)
    names = [k for k, _ in rows]
    x = np.stack([r for _, r in rows])
    if x.shape[0] % 2: x = np.concatenate([x, x[:1]]); names.append(names[0])
    streams = x.shape[0] // 2
    with np.errstate(all="ignore"):
        want_mag = np.stack([Ref.fft(x[r]) for r in range(x.shape[0])])
        want_raw = np.stack([Oracle.transform_fft(x[r], want_raw=True)[1] for r in range(x.shape[0])])
    # the restatement's raw output feeds the same magnitudes as the compiled reference's (pins the raw reference used below)
    with np.errstate(all="ignore"):
        for r in range(x.shape[0]):
            assert _same_value(Oracle.transform_fft(x[r]), want_mag[r], True), names[r]
    d_x = torch.from_numpy(x).cuda()
    # interleaved twin of the same rows: stream u = (row 2u, row 2u+1) as (L, R)
    xi = np.ascontiguousarray(np.stack([x[0::2], x[1::2]], axis=2))                    # [streams][n][2]
    d_xi = torch.from_numpy(xi).cuda()
    for log_mode in (0, 1):
        b = G.Batch(G.Params(n=n, log_mode=log_mode), streams, G.OP_FFT)
        d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
        for layout, call, src in (("planar", b.process_f32, d_x), ("interleaved", b.process_f32_stereo, d_xi)):
            call(src, d_out, G.OP_FFT)
            got = d_out.cpu().numpy()
            for r in range(x.shape[0]):
                assert _same_value(got[r], want_mag[r], log_mode == 0), (layout, log_mode, names[r])
            call(src, d_out, G.OP_FFT | G.OP_RAW)
            raw = d_out.cpu().numpy()
            zero_sign_diffs = 0
            for r in range(x.shape[0]):
                gn, wn = np.isnan(raw[r]), np.isnan(want_raw[r])
                assert (gn == wn).all(), (layout, names[r], "NaN pattern of the raw FFT")
                ok = ~wn
                assert ((bits(raw[r]) & 0x7fffffff)[ok] == (bits(want_raw[r]) & 0x7fffffff)[ok]).all(), (layout, names[r])
                diff = (bits(raw[r]) != bits(want_raw[r])) & ok
                assert (want_raw[r][diff] == 0).all(), (layout, names[r], "a sign differs on a non-zero value")
                zero_sign_diffs += int(diff.sum())
            assert zero_sign_diffs == 0, (layout, zero_sign_diffs)
        b.close()


# ---- output == state -------------------------------------------------------------------------------------------------------
def _hip_copy_to_host(ptr, count):
    out = np.empty(count, np.float32)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(count * 4), 2) == 0     # device -> host
    return out


@pytest.mark.parametrize("n", [1024, 4096, 16384])
def test_gravity_output_is_the_state(G, n):
    """With GLV_OP_OUTPUT_IS_STATE (opt-in since ABI 4) a chain ending in gravity writes its spectra ONCE (SURVEY 8d
    row B): the caller's buffer is the `applied` array of the next update (render.c:733-734 store the same value twice).  Same
    buffer every call, alternating buffers, the default batch-owned state and the d_out = NULL form all give the oracle's bits on
    the raw values and each other's bits on magnitudes; clobbering the output between updates is harmless by default and matters
    only to the caller who opted in."""
    import torch
    streams, updates = 7, 5
    ops = G.OP_FFT | G.OP_GRAVITY
    p = G.Params(n=n)
    same, alt, priv, none_, rawb = (G.Batch(p, streams, G.OP_GRAVITY) for _ in range(5))
    o_same = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    o_alt = [torch.empty_like(o_same) for _ in range(3)]
    o_priv, o_raw = torch.empty_like(o_same), torch.empty_like(o_same)
    sos = [StreamOracle(n, average=False) for _ in range(streams)]
    grav_raw = np.zeros((streams * 2, n), np.float32)
    OIS = G.OP_OUTPUT_IS_STATE
    assert same.algorithmic_bytes(ops | OIS) == 20 * n * streams and same.algorithmic_bytes(ops) == 28 * n * streams
    for u in range(updates):
        pcm = lcg_pcm_fast(5100 + 31 * u + n, streams * 2 * n)
        d_pcm = torch.from_numpy(pcm).cuda()
        same.process_s16(d_pcm, o_same, ops | OIS)
        alt.process_s16(d_pcm, o_alt[u % 3], ops | OIS)
        priv.process_s16(d_pcm, o_priv, ops | (G.OP_PRIVATE_STATE if u % 2 else 0))     # the ABI 3 flag is accepted and means the default
        none_.process_s16(d_pcm, None, ops)
        rawb.process_s16(d_pcm, o_raw, ops | G.OP_RAW)              # RAW chains keep a private state (the output is not the state's meaning)
        torch.cuda.synchronize()
        assert same.gravity_state() == o_same.data_ptr() and alt.gravity_state() == o_alt[u % 3].data_ptr()
        assert priv.gravity_state() not in (o_priv.data_ptr(),) and none_.gravity_state() != 0
        ref_bits = o_same.view(torch.int32)
        assert torch.equal(o_alt[u % 3].view(torch.int32), ref_bits), u
        assert torch.equal(o_priv.view(torch.int32), ref_bits), u
        st = _hip_copy_to_host(none_.gravity_state(), streams * 2 * n).reshape(streams * 2, n)
        assert (bits(st) == bits(o_same.cpu().numpy())).all(), u
        st = _hip_copy_to_host(priv.gravity_state(), streams * 2 * n).reshape(streams * 2, n)
        assert (bits(st) == bits(o_same.cpu().numpy())).all(), u
        got = o_same.cpu().numpy(); raw = o_raw.cpu().numpy()
        for s in range(streams):
            want, wraw = sos[s].frame(pcm[s * 2 * n:(s + 1) * 2 * n], want_raw=True)
            assert sos[s].close(got[2 * s:2 * s + 2], want), (u, s)
            for c in range(2):                                      # gravity on the raw values: bit-exact state machine
                row = np.ascontiguousarray(wraw[c]); Oracle.gravity(row, grav_raw[2 * s + c])
                assert (bits(raw[2 * s + c]) == bits(row)).all(), (u, s, c)
        o_priv.fill_(float("nan"))                                  # the private copy does not care
    # opted in, the caller's buffer IS the state: overwrite it and the next update starts from what it holds
    o_same.zero_()
    pcm = lcg_pcm_fast(99, streams * 2 * n); d_pcm = torch.from_numpy(pcm).cuda()
    same.process_s16(d_pcm, o_same, ops | OIS)
    fresh = G.Batch(p, streams, G.OP_GRAVITY)
    o_fresh = torch.empty_like(o_same)
    fresh.process_s16(d_pcm, o_fresh, ops)                          # a fresh batch starts from zeros too
    assert torch.equal(o_same.view(torch.int32), o_fresh.view(torch.int32))
    priv.process_s16(d_pcm, o_priv, ops)                            # ... whereas this one continues its own history
    with pytest.raises(G.GlvError):                                 # the flag is refused where the output is not the state's meaning
        rawb.process_s16(d_pcm, o_raw, ops | G.OP_RAW | OIS)
    with pytest.raises(G.GlvError):
        same.process_s16(d_pcm, None, ops | OIS)
    assert not torch.equal(o_priv.view(torch.int32), o_fresh.view(torch.int32))
    for b in (same, alt, priv, none_, rawb, fresh): b.close()


def test_gravity_state_after_fused_average_is_refused(G):
    """glv_batch_gravity_state handed out the (stale, zero) gravity-only buffer after fused gravity+average calls."""
    import torch
    n, streams = 1024, 3
    b = G.Batch(G.Params(n=n), streams, G.OP_GRAVITY | G.OP_AVERAGE)
    d_pcm = torch.from_numpy(lcg_pcm_fast(1, streams * 2 * n)).cuda()
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    assert b.gravity_state() != 0                                   # nothing applied yet: the zeroed buffer
    b.process_s16(d_pcm, d_out, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE)
    with pytest.raises(G.GlvError) as ei:
        b.gravity_state()
    assert ei.value.code == G.ERR_STATE
    b.reset()
    b.process_s16(d_pcm, d_out, G.OP_FFT | G.OP_GRAVITY)
    assert b.gravity_state() not in (0, d_out.data_ptr())           # the batch-owned state (the default since ABI 4)
    b.process_s16(d_pcm, d_out, G.OP_FFT | G.OP_GRAVITY | G.OP_OUTPUT_IS_STATE)
    assert b.gravity_state() == d_out.data_ptr()
    b.close()


def test_rings_allocated_at_creation(G):
    """GLV_OP_RING_S16 / GLV_OP_RING_F32 in the creation mask: the device rings exist before the first update, which then is
    purely stream-ordered; results equal a batch that allocates lazily."""
    import torch
    n, streams, nf = 2048, 5, 256
    a = G.Batch(G.Params(n=n), streams, G.OP_FFT | G.OP_RING_S16 | G.OP_RING_F32)
    b = G.Batch(G.Params(n=n), streams, G.OP_FFT | G.OP_RING_S16 | G.OP_RING_F32)
    oa = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda"); ob = torch.empty_like(oa)
    for u in range(3):
        new = torch.from_numpy(lcg_pcm_fast(40 + u, streams * nf * 2)).cuda()
        a.ring_update_s16(new, nf, oa, G.OP_FFT); b.ring_update_s16(new, nf, ob, G.OP_FFT)
        assert torch.equal(oa.view(torch.int32), ob.view(torch.int32)), u
        newf = torch.from_numpy((np.random.default_rng(u).standard_normal(streams * nf * 2) * 0.2).astype(np.float32)).cuda()
        a.ring_update_f32(newf, nf, oa, G.OP_FFT); b.ring_update_f32(newf, nf, ob, G.OP_FFT)
        assert torch.equal(oa.view(torch.int32), ob.view(torch.int32)), u
    a.close(); b.close()


# ---- gl_storage with SMOOTH / RAW ------------------------------------------------------------------------------
@pytest.mark.parametrize("n,F", [(1024, 5), (4096, 3)])
def test_gl_storage_chain_with_smooth_and_raw(G, n, F):
    """With gl_storage = 1 an FFT chain with gravity / average runs pass by pass (render.c:2188-2265).  That path used to
    ignore GLV_OP_RAW (magnitudes were computed anyway) and GLV_OP_SMOOTH (rows came back unsmoothed, no error).  Now:
      FFT|GRAVITY|AVERAGE|SMOOTH  == the same chain without SMOOTH, then transform_smooth (oracle) on every row, bit for bit
      FFT|RAW|GRAVITY|AVERAGE     == the GL-storage passes (oracle glvo_gl_chain_r16) applied to the oracle's RAW rows"""
    import torch
    streams = 4
    p = G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, log_mode=0)
    plain, smooth, rawb = (G.Batch(p, streams, G.OP_GRAVITY | G.OP_AVERAGE) for _ in range(3))
    o_plain = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    o_smooth, o_raw = torch.empty_like(o_plain), torch.empty_like(o_plain)
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    store = np.zeros((streams * 2, n), np.float32); hist = np.zeros((streams * 2, F, n), np.float32)
    heads = [C.c_size_t(0) for _ in range(streams * 2)]
    for u in range(F + 2):
        pcm = (lcg_pcm_fast(6100 + u + n, streams * 2 * n) // 8).astype(np.int16)
        d_pcm = torch.from_numpy(pcm).cuda()
        plain.process_s16(d_pcm, o_plain, ops)
        smooth.process_s16(d_pcm, o_smooth, ops | G.OP_SMOOTH)
        rawb.process_s16(d_pcm, o_raw, ops | G.OP_RAW)
        want = o_plain.cpu().numpy().copy()
        with np.errstate(all="ignore"):
            for r in range(streams * 2):
                row = np.ascontiguousarray(want[r]); Oracle.lib().glvo_smooth(row, n, p.smooth_distance, p.smooth_ratio); want[r] = row
        got = o_smooth.cpu().numpy()
        gn, wn = np.isnan(got), np.isnan(want)                      # transform_smooth's 0/0 at t = 0
        assert (gn == wn).all() and (bits(got)[~wn] == bits(want)[~wn]).all(), u
        assert not torch.equal(o_smooth.view(torch.int32), o_plain.view(torch.int32))
        raw = o_raw.cpu().numpy()
        for s in range(streams):
            _, wraw = StreamOracle(n, gravity=False, average=False).frame(pcm[s * 2 * n:(s + 1) * 2 * n], want_raw=True)
            for c in range(2):
                w = np.ascontiguousarray(wraw[c])
                Oracle.lib().glvo_gl_chain_r16(w, store[2 * s + c], hist[2 * s + c], C.byref(heads[2 * s + c]), n, F, 1, 1, 4.2, 86.1328125)
                assert (bits(raw[2 * s + c]) == bits(w)).all(), (u, s, c)
    for b in (plain, smooth, rawb): b.close()


# ---- kernel configurations (f4) -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192, 16384, 32768])
def test_every_kernel_variant_gives_the_same_bits(G, n):
    """glv_inst.hip builds more than one kernel configuration for these sizes (a different radix split / rows per workgroup /
    table placement; since round 4 also for N = 16384 -- the split exchange with every twiddle in LDS -- and N = 32768, and for
    the f32 inputs).  Whatever the wisdom picks, the spectra are the same: raw FFT, magnitudes (both log modes), the stateful
    chain, GL_R16 texels, fused bars, the ring mode and the f32 inputs of every variant equal variant 0 bit for bit."""
    import torch
    streams, F, nf = (37 if n <= 8192 else 9), 5, 256
    pcm = [torch.from_numpy(lcg_pcm_fast(8200 + n + u, streams * 2 * n)).cuda() for u in range(3)]
    new = torch.from_numpy(lcg_pcm_fast(8300 + n, streams * nf * 2)).cuda()
    results = []
    nv = None
    for log_mode in (1, 0):
        per_variant = []
        probe = G.Batch(G.Params(n=n, log_mode=log_mode), streams, G.OP_FFT | G.OP_BARS | G.OP_RING_S16)
        nv = probe.variants(); probe.close()
        assert nv >= 2, "this size is expected to carry a runner-up configuration"
        for v in range(nv):
            p = G.Params(n=n, log_mode=log_mode, avg_frames=F)
            out = {}
            b = G.Batch(p, streams, G.OP_FFT | G.OP_RING_S16 | G.OP_BARS); b.set_variant(v)
            assert "variant %d" % v in b.describe_variant(v)
            o = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
            b.process_s16(pcm[0], o, G.OP_FFT | G.OP_RAW); out["raw"] = o.clone(); assert b.last_variant() == v
            b.process_s16(pcm[0], o, G.OP_FFT); out["mag"] = o.clone()
            q = torch.zeros((streams * 2, n), dtype=torch.int16, device="cuda")
            b.process_s16(pcm[0], q, G.OP_FFT | G.OP_R16); out["r16"] = q.clone()
            b.ring_update_s16(new, nf, o, G.OP_FFT); out["ring"] = o.clone(); assert b.last_variant() == v
            b.close()
            bc = G.Batch(p, streams, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_RING_S16); bc.set_variant(v)
            bb = G.Batch(p, streams, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_RING_S16); bb.set_variant(v)
            d_bars = torch.empty((streams * 2, p.bars), dtype=torch.float32, device="cuda")
            for u in range(3):
                bc.process_s16(pcm[u], o, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE); out["chain%d" % u] = o.clone()
                bb.process_s16(pcm[u], d_bars, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS); out["bars%d" % u] = d_bars.clone()
            bc.close(); bb.close()
            # the f32 inputs carry the runner-up too (round 4); the audit log mode has configuration 0 only: a forced variant falls
            # back there instead of failing
            bf = G.Batch(p, streams, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE); bf.set_variant(v)
            x = torch.from_numpy((np.random.default_rng(n).standard_normal((streams * 2, n)) * 0.3).astype(np.float32)).cuda()
            xs = torch.from_numpy((np.random.default_rng(n + 1).standard_normal((streams, n, 2)) * 0.3).astype(np.float32)).cuda()
            bf.process_f32(x, o, G.OP_FFT); out["f32"] = o.clone(); assert bf.last_variant() == v
            bf.process_f32_stereo(xs, o, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE); out["f32s_chain"] = o.clone(); assert bf.last_variant() == v
            bf.close()
            if log_mode == 1:
                ba = G.Batch(G.Params(n=n, log_mode=2), streams, G.OP_FFT); ba.set_variant(v)
                ba.process_s16(pcm[0], o, G.OP_FFT); assert ba.last_variant() == 0
                ba.close()
            per_variant.append(out)
        for v in range(1, nv):
            for k, t in per_variant[0].items():
                assert torch.equal(per_variant[v][k].view(torch.int32) if t.dtype == torch.float32 else per_variant[v][k], t.view(torch.int32) if t.dtype == torch.float32 else t), (n, log_mode, v, k)
        results.append(per_variant[0])
    # and variant 0's raw FFT is the oracle's
    raw = results[0]["raw"].cpu().numpy(); pcm0 = pcm[0].cpu().numpy()
    for s in (0, streams - 1):
        _, want = StreamOracle(n, gravity=False, average=False).frame(pcm0[s * 2 * n:(s + 1) * 2 * n], want_raw=True)
        assert (bits(raw[2 * s:2 * s + 2]) == bits(want)).all()
    with pytest.raises(G.GlvError):
        b = G.Batch(G.Params(n=n), streams, G.OP_FFT | G.OP_BARS | G.OP_RING_S16)
        try: b.set_variant(nv)
        finally: b.close()


def test_create_destroy_returns_every_byte(G):
    """glv_batch_create / glv_batch_destroy over every family of tables and buffers the library can allocate -- float and GL_R16 state, rings, scratch rows, bar tables
    of every arithmetic (chunked, f32 and i8 matrix-core tiles, the SAMPLE_MODE maximum / hybrid blocks), smooth bounds, placement tuning -- leaves the device's free
    memory where it was (hipMemGetInfo through torch; 40 cycles each: a table leaked once per batch would show as 40 x its size)"""
    import torch
    torch.cuda.synchronize()
    n, streams = 4096, 64
    gl = dict(avg_window_kind=1, gl_storage=1)
    cases = [
        (G.Params(n=n), G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_SMOOTH | G.OP_RING_S16 | G.OP_RING_F32),
        (G.Params(n=n, bars=n, bar_phase=0.5, **gl), G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_BARS_ONLY),
        (G.Params(n=n, bars=n, bar_phase=0.5), G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16),
        (G.Params(n=n, bars=n, bar_phase=0.5, sample_mode=G.SAMPLE_MAXIMUM, **gl), G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS),
        (G.Params(n=n, bars=80, sample_mode=G.SAMPLE_HYBRID, round_formula=G.ROUND_LINEAR), G.OP_GRAVITY | G.OP_BARS),
        (G.Params(n=16384, bars=80, gl_storage=2), G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS),
    ]
    def cycle(count):
        for p, mask in cases:
            for _ in range(count):
                b = G.Batch(p, streams, mask)
                q = G.Params(**{**p.__dict__, "smooth_factor": 0.05})     # set_params rebuilds the bar tables in place
                b.set_params(q)
                b.close()
    cycle(2)                                                        # (the runtime's own pools settle)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(40)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), (free0, free1, free0 - free1)     # the smallest table here is > 200 KiB: 40 leaked copies would be > 8 MiB
