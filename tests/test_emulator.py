"""CPU: the kernel's per-thread phase code (glava_amd/csrc/glv_frame.h) walked by the host
emulator (tests/emu/glv_emu.cpp) must reproduce the oracle bit for bit -- this is where the
Stockham index maps, the pass plan for every size, the LDS swizzle, the twiddle gather and the
gravity/average state machine are debugged without a GPU."""
import os

import numpy as np
import pytest

from emu_lib import emu_process
from oracle_lib import StreamOracle, lcg_pcm_fast

OP_FFT, OP_GRAVITY, OP_AVERAGE, OP_RAW = 1, 2, 4, 8


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("log_e", [4, 3, 5])
@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
def test_fft_raw_and_magnitude_bit_exact(emu, oracle, n, log_e):
    """every size, both lane footprints (E = 16 and E = 8 points per lane: different pass plans)"""
    units = 2
    pcm = lcg_pcm_fast(1000 + n, units * 2 * n)
    raw = emu_process(emu, n, pcm, units, OP_FFT | OP_RAW, log_e=log_e)
    mag = emu_process(emu, n, pcm, units, OP_FFT, log_e=log_e)
    for u in range(units):
        o, r = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n], want_raw=True)
        assert (bits(r) == bits(raw[2 * u:2 * u + 2])).all()
        assert (bits(o) == bits(mag[2 * u:2 * u + 2])).all()


def test_f32_planar_input(emu, oracle):
    n = 2048
    x = (np.random.default_rng(5).standard_normal((3, n)) * 0.4).astype(np.float32)
    out = emu_process(emu, n, x, 3, OP_FFT, in_mode=1)
    for r in range(3):
        assert (bits(out[r]) == bits(oracle.transform_fft(x[r]))).all()


@pytest.mark.parametrize("F,win", [(5, True), (6, False), (1, True), (2, True)])
def test_gravity_average_state_machine(emu, oracle, F, win):
    n, units = 1024, 3
    hist = np.zeros((units * 2, F, n), np.float32)
    sos = [StreamOracle(n, avg_frames=F, avg_window=win) for _ in range(units)]
    head = 0
    for fr in range(2 * F + 2):
        pcm = lcg_pcm_fast(77 + fr, units * 2 * n)
        out = emu_process(emu, n, pcm, units, OP_FFT | OP_GRAVITY | OP_AVERAGE, hist=hist, F=F, head=head, avg_window=int(win))
        head = (head + 1) % F
        for u in range(units):
            assert (bits(sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n])) == bits(out[2 * u:2 * u + 2])).all(), (fr, u)


def test_gravity_only_and_average_only(emu, oracle):
    n, units = 512, 2
    grav = np.zeros((units * 2, n), np.float32)
    sos = [StreamOracle(n, average=False) for _ in range(units)]
    for fr in range(4):
        pcm = lcg_pcm_fast(177 + fr, units * 2 * n)
        out = emu_process(emu, n, pcm, units, OP_FFT | OP_GRAVITY, grav=grav)
        for u in range(units):
            assert (bits(sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n])) == bits(out[2 * u:2 * u + 2])).all()
    F = 4
    hist = np.zeros((units * 2, F, n), np.float32)
    sos = [StreamOracle(n, gravity=False, avg_frames=F) for _ in range(units)]
    for fr in range(6):
        pcm = lcg_pcm_fast(277 + fr, units * 2 * n)
        out = emu_process(emu, n, pcm, units, OP_FFT | OP_AVERAGE, hist=hist, F=F, head=fr % F)
        for u in range(units):
            assert (bits(sos[u].frame(pcm[u * 2 * n:(u + 1) * 2 * n])) == bits(out[2 * u:2 * u + 2])).all()


def test_mono_mix(emu, oracle):
    n = 1024
    pcm = lcg_pcm_fast(99, 2 * n)
    out = emu_process(emu, n, pcm, 1, OP_FFT, mono=1)
    want = StreamOracle(n, channels=1, gravity=False, average=False).frame(pcm)
    assert (bits(out) == bits(want)).all()
    assert (bits(out[0]) == bits(out[1])).all()


def test_fast_log_mode_within_tolerance(emu, oracle):
    """log_mode 1 (fp32 log) must stay within the 1e-5 relative bar of BASELINE.json north_star."""
    n = 4096
    pcm = lcg_pcm_fast(4, 2 * n)
    fast = emu_process(emu, n, pcm, 1, OP_FFT, log_mode=1)
    want = StreamOracle(n, gravity=False, average=False).frame(pcm)
    rel = np.abs(fast - want) / np.maximum(np.abs(want), 1e-30)
    assert rel.max() <= 1e-5, rel.max()


def test_f32_interleaved_pulse_layout(emu, oracle):
    """pulse_input.c:155-178: interleaved f32 frames, stereo and mono ((L+R)/2 in float)."""
    n = 1024
    x = (np.random.default_rng(9).standard_normal((2, n, 2)) * 0.3).astype(np.float32)    # [frames][n][2]
    out = emu_process(emu, n, x, 2, OP_FFT | OP_RAW, in_mode=3)
    for fr in range(2):
        l, r = oracle.lib().glvo_unpack_f32, None
        pl = np.empty(n, np.float32); pr = np.empty(n, np.float32)
        oracle.lib().glvo_unpack_f32(np.ascontiguousarray(x[fr].reshape(-1)), n, 2, pl, pr)
        for c, ch in enumerate((pl, pr)):
            _, want = oracle.transform_fft(ch, want_raw=True)
            assert (bits(out[2 * fr + c]) == bits(want)).all()
    mono = emu_process(emu, n, x, 2, OP_FFT | OP_RAW, in_mode=3, mono=1)
    for fr in range(2):
        pl = np.empty(n, np.float32); pr = np.empty(n, np.float32)
        oracle.lib().glvo_unpack_f32(np.ascontiguousarray(x[fr].reshape(-1)), n, 1, pl, pr)
        _, want = oracle.transform_fft(pl, want_raw=True)
        assert (bits(mono[2 * fr]) == bits(want)).all() and (bits(mono[2 * fr + 1]) == bits(want)).all()


def test_raw_state_chain_bit_exact(emu, oracle):
    """GLV_OP_RAW with gravity/average: the state machine on raw FFT values, all-float => bit-exact."""
    import ctypes as C
    from oracle_lib import Oracle
    n, F, units = 1024, 5, 2
    hist = np.zeros((units * 2, F, n), np.float32)
    ograv = np.zeros((units * 2, n), np.float32)
    ohist = np.zeros((units * 2, F, n), np.float32)
    heads = [C.c_size_t(0) for _ in range(units * 2)]
    for fr in range(7):
        pcm = lcg_pcm_fast(3000 + fr, units * 2 * n)
        out = emu_process(emu, n, pcm, units, OP_FFT | OP_RAW | OP_GRAVITY | OP_AVERAGE, hist=hist, F=F, head=fr % F)
        for u in range(units):
            _, raw = StreamOracle(n, gravity=False, average=False).frame(pcm[u * 2 * n:(u + 1) * 2 * n], want_raw=True)
            for c in range(2):
                row = np.ascontiguousarray(raw[c])
                Oracle.gravity(row, ograv[2 * u + c])
                Oracle.average(row, ohist[2 * u + c], heads[2 * u + c], F, True)
                assert (bits(row) == bits(out[2 * u + c])).all(), (fr, u, c)


def test_ring_rotation(emu, oracle):
    """FIFO ring mode: reading a circular PCM ring with a rotation == the reference's memmove ring."""
    n = 1024
    pcm = lcg_pcm_fast(55, 2 * n).reshape(n, 2)
    for rot_frames in (0, 256, 512, 768, 2, 1, 251, 1023):                          # odd: fifo.c accepts any sample_sz
        ring = np.ascontiguousarray(np.roll(pcm, rot_frames, axis=0)).reshape(-1)   # logical frame i at (i+rot)%n
        out = emu_process(emu, n, ring, 1, OP_FFT | OP_RAW, rot=rot_frames)
        _, want = StreamOracle(n, gravity=False, average=False).frame(pcm.reshape(-1), want_raw=True)
        assert (bits(out) == bits(want)).all(), rot_frames


def test_lds_padding_is_injective_and_conflict_free():
    """Model of the gfx950 LDS banking for the first exchange (MI355X_MICROARCH.md, LDS):
    ds_write_b64 is serviced in 16-lane groups over 32 4-byte banks, ds_read_b64 in 32-lane groups
    over 64 banks.  The padded index q + (q >> 4) (glv_core.h lds_index) must be injective, fit the
    region (nn + nn/16), make the pass-0 write pattern q = 16*tid + e conflict free, and leave the
    pass-1 read pattern q = i*(nn/R) + tid with at most one 2-way conflict per 32-lane group."""
    for log_nn in range(8, 14):
        nn = 1 << log_nn
        T = nn // 16
        pad = lambda q: q + (q >> 4)  # noqa: E731
        idx = [pad(q) for q in range(nn)]
        assert len(set(idx)) == nn and max(idx) < nn + nn // 16
        for e in range(16):                      # write: lanes = consecutive tids
            for g0 in range(0, min(T, 64), 16):
                banks = [(2 * pad(16 * t + e)) % 32 for t in range(g0, g0 + 16)]
                assert len(set(banks)) == 16, (log_nn, e, g0)
        rb = min(4, log_nn - 4)                  # radix bits of pass 1
        R = 1 << rb
        if log_nn - 4 <= 4 and (16 >> rb) >= 2:
            continue                             # pass 1 is the last pass with adjacent groups: different pattern
        for i in range(R):
            for g0 in range(0, min(T, 64), 32):
                lanes = list(range(g0, min(g0 + 32, T)))
                banks = [(2 * pad(i * (nn // R) + t)) % 64 for t in lanes]
                assert len(lanes) - len(set(banks)) <= 1, (log_nn, i, g0)


# ---- GLV_OP_BARS host logic: tap tables, work lists, chunk arithmetic -------------------------------------
import ctypes as C  # noqa: E402
from emu_lib import emu_bars  # noqa: E402


@pytest.mark.parametrize("n,bars", [(512, 80), (4096, 80), (16384, 80), (4096, 31), (16384, 256)])
def test_bar_work_lists(emu, n, bars):
    """every chunk of every bar exactly once, a bar's chunks in one group in order, zero-weight padding --
    for every group count the kernels use (T/8 of each size, 32 for glv_bars_kernel)"""
    for groups in (1, 4, 8, 16, 32):
        assert emu.glvemu_bar_items_check(n, bars, C.c_float(0.025), groups) == 0, groups
    assert emu.glvemu_bar_items_check(n, bars, C.c_float(0.1), 16) == 0


@pytest.mark.parametrize("n", [512, 4096, 16384])
def test_bars_emulator_vs_restatement(emu, oracle, n):
    """the kernels' chunked 8-lane arithmetic == the oracle's restatement of that documented order (glvo_bars_chunked), bit
    for bit; both differ from the tap-by-tap order of the shader text (glvo_bars, pinned to the GLSL evaluation by
    tests/test_glsl_twins.py) by summation rounding only; the result does not depend on how many groups share the work"""
    bars = 80
    spec = np.abs(np.random.default_rng(n).standard_normal((3, n))).astype(np.float32) * 0.4
    spec[1, ::7] = 1.7          # exercise the [0,1] clamp
    got16, steps = emu_bars(emu, spec, n, bars, groups=16)
    assert steps % 2 == 0 and steps > 0          # kBarBatch
    for r in range(3):
        want = np.empty(bars, np.float32)
        oracle.lib().glvo_bars_chunked(np.ascontiguousarray(spec[r]), n, want, bars, 0.025)
        assert (bits(got16[r]) == bits(want)).all(), r
        seq = np.empty(bars, np.float32)
        oracle.lib().glvo_bars(np.ascontiguousarray(spec[r]), n, seq, bars, 0.025)
        assert np.allclose(want, seq, rtol=2e-4, atol=2e-6)      # chunk order vs tap-by-tap order: rounding of the sums only
    for groups in (1, 4, 8, 32):
        got, _ = emu_bars(emu, spec, n, bars, groups=groups)
        assert (bits(got) == bits(got16)).all(), groups


@pytest.mark.parametrize("n,bars,factor,phase", [(1024, 80, 0.025, 0.0), (1024, 1024, 0.025, 0.5), (2048, 33, 0.1, 0.0), (4096, 256, 0.01, 0.0),
                                                 (8192, 1, 0.025, 0.0), (16384, 160, 0.05, 0.0), (256, 7, 0.3, 0.25)])
def test_bars_parameter_sweep_emulator_vs_restatement(emu, oracle, n, bars, factor, phase):
    """the same sweep the device runs (tests/test_gpu_parity.py test_bars_parameter_sweep_against_the_oracle_order): bar counts
    from 1 to n, narrow and wide windows, the pre-smoothing pass's positions, values outside [0, 1] / infinite / NaN"""
    rng = np.random.default_rng(n + bars)
    spec = np.abs(rng.standard_normal((4, n))).astype(np.float32) * 0.6
    spec[1, ::5] = -spec[1, ::5]
    spec[2, ::11] = np.inf
    spec[2, 3::13] = np.nan
    spec[3, 7::17] = -np.inf
    spec[3, ::3] *= 1e-30
    got, _ = emu_bars(emu, spec, n, bars, smooth_factor=factor, groups=32, phase=phase)
    for r in range(4):
        want = np.empty(bars, np.float32)
        oracle.lib().glvo_bars_chunked_at(np.ascontiguousarray(spec[r]), n, want, bars, factor, phase)
        assert (bits(got[r]) == bits(want)).all(), r
    assert np.isfinite(got).all()


def test_division_by_65535_is_correctly_rounded_for_every_integer_argument(emu):
    """glv_core.h div_65535 -- fma(v, c_hi, v * c_lo) -- against the IEEE single division of fifo.c:105-106 for every
    integer |v| <= 65535: the 65536 s16 sample values (unpack) and the 65536 GL_R16 texel values (readback)."""
    import ctypes as C
    lo, hi = -65535, 65535
    out = np.empty(hi - lo + 1, dtype=np.float32)
    emu.glvemu_div_65535.argtypes = [C.c_int, C.c_int, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")]
    emu.glvemu_div_65535.restype = None
    emu.glvemu_div_65535(lo, hi, out)
    want = np.arange(lo, hi + 1, dtype=np.float32) / np.float32(65535)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def _special_rows(n, rng):
    base = lambda: (rng.standard_normal(n) * 0.25).astype(np.float32)          # noqa: E731
    rows = []
    r = base(); r[::7] = -0.0; rows.append(r)
    rows.append(np.full(n, -0.0, np.float32)); rows.append(np.zeros(n, np.float32))
    r = base(); r[5::11] = np.float32(1e-41); r[6::13] = np.float32(-3e-45); rows.append(r)
    rows.append((rng.standard_normal(n) * 1e-39).astype(np.float32))
    r = base(); r[n // 3] = np.inf; rows.append(r)
    r = base(); r[2] = -np.inf; r[3] = np.inf; rows.append(r)
    r = base(); r[n - 1] = np.nan; rows.append(r)
    r = base(); r[10] = np.nan; r[11] = np.inf; r[500] = -0.0; rows.append(r)
    r = base(); r[::2] = np.float32(3.0e38); r[1::2] = np.float32(-3.0e38); rows.append(r)
    return np.stack(rows)


@pytest.mark.parametrize("n,log_e", [(1024, 3), (4096, 4), (16384, 5)])
def test_f32_special_values_like_the_compiled_reference(emu, oracle, ref, n, log_e):
    """f32 rows with -0.0, denormals, +-Inf, NaN through the kernel's own arithmetic (host emulator: no unit-twiddle shortcut
    for f32 input, non-finite values through the table-driven log) against the reference's transform_fft compiled from
    /root/reference: NaN exactly where the reference has NaN, identical bits elsewhere (log_mode 0); the raw FFT equals
    the restatement's bit for bit including the sign of every zero.  The GPU twin of this test is
    tests/test_concurrency_state.py::test_f32_special_values_against_the_compiled_reference."""
    x = _special_rows(n, np.random.default_rng(n))
    with np.errstate(all="ignore"):
        want = np.stack([ref.fft(x[r]) for r in range(x.shape[0])])
        want_raw = np.stack([oracle.transform_fft(x[r], want_raw=True)[1] for r in range(x.shape[0])])
        got = emu_process(emu, n, x, x.shape[0], OP_FFT, in_mode=1, log_mode=0, log_e=log_e)
        raw = emu_process(emu, n, x, x.shape[0], OP_FFT | OP_RAW, in_mode=1, log_mode=0, log_e=log_e)
    for r in range(x.shape[0]):
        gn, wn = np.isnan(got[r]), np.isnan(want[r])
        assert (gn == wn).all(), r
        assert (bits(got[r])[~wn] == bits(want[r])[~wn]).all(), r
        rn = np.isnan(want_raw[r])
        assert (np.isnan(raw[r]) == rn).all(), r
        assert (bits(raw[r])[~rn] == bits(want_raw[r])[~rn]).all(), r
    assert np.isinf(want).any() and np.isnan(want).any()          # the rows do exercise both


@pytest.mark.parametrize("n", [16384, 32768])
def test_fused_tilt_formula_is_within_its_bound(n):
    """glv_core.h tilt_lin (TILTREG 3, log_mode 1 at N >= 16384): the folded tilt factor tilt(i) * ln2/3 from a per-lane base
    term and one fused multiply-add + max per value, against the exact product of the reference's tilt (render.c:845, float
    operations) with ln2/3, for EVERY index and a few parameter sets: <= 4.5e-7 relative to that (itself three times rounded) product -- each side is
    within ~2e-7 of the real value; the contract of log_mode 1 is 1e-5."""
    f = np.float32
    k = f(0.69314718055994530942 / 3.0)

    def fma(a, b, c):
        return f(np.float64(a) * np.float64(b) + np.float64(c))          # the f32 x f32 product is exact in f64
    for scale, cutoff in ((10.2, 0.3), (0.0, 0.0), (20.0, 1.0), (3.7, 0.95)):
        inv_n = f(1.0) / f(n)
        omc = f(1.0) - f(cutoff)
        S = f(f(f(scale) * inv_n) * k); O = f(omc * k)
        idx = np.arange(n)
        worst = 0.0
        for base in (0, 4, 252, 1020, 2044):                   # the lane part 4 * tid (tid < 512) and the rest c >= 0, as in the kernel
            c = np.maximum(idx - base, 0)
            B = fma(f(base), S, O)
            got = np.maximum(f(np.float64(c.astype(f)) * np.float64(S) + np.float64(B)), k).astype(f)
            got[idx < base] = np.nan
            t = (idx.astype(f) * inv_n) * f(scale)
            t = (t.astype(f) + omc).astype(f)
            exact = np.maximum(t, f(1.0)).astype(np.float64) * np.float64(k)
            worst = max(worst, float(np.nanmax(np.abs(got.astype(np.float64) - exact) / exact)))
        assert worst <= 4.5e-7, (n, scale, cutoff, worst)


@pytest.mark.parametrize("n,bins", [(256, 160), (1024, 160), (2048, 160), (4096, 160), (4096, 288), (8192, 288)])
def test_tile_tables_of_the_many_bars_kernels(emu, n, bins):
    """From 256 bars up (the pre-smoothing pass: bars == n) a bar is one fma chain over its taps, and the kernels run off host
    tables: tiles of 32 consecutive bars with one common bin range and the weights in MFMA operand layout (+0 outside a bar's own
    taps), and -- for the matrix-core kernel -- rounds of four tiles whose bins, and what the next round adds, fit an LDS ring of
    `bins` bins.  Invariants of all of them, and which sizes get rounds at all: n <= 2048 with 160 bins, n = 4096 only with 288
    (its longest bar has 191 taps, four tiles spread them over 40 bins more), n >= 8192 not (the one-lane-per-bar kernel then)."""
    import ctypes as C
    nr, mc = C.c_uint(0), C.c_uint(0)
    rc = emu.glvemu_bar_tiles_check(n, n, C.c_float(0.025), C.c_float(0.5), bins, 4, C.byref(nr), C.byref(mc))
    if n >= 8192 or (n == 4096 and bins == 160):
        assert rc == -1, (rc, mc.value)
    else:
        assert rc == 0 and nr.value >= n // 128, (rc, nr.value, mc.value)
    # fewer bars than a power of two, a last tile that is not full, wide gaps between the bars
    assert emu.glvemu_bar_tiles_check(2048, 1001, C.c_float(0.025), C.c_float(0.0), 160, 4, C.byref(nr), C.byref(mc)) == -1     # two tiles in a row: 168 bins
    assert emu.glvemu_bar_tiles_check(2048, 1001, C.c_float(0.025), C.c_float(0.0), 288, 4, C.byref(nr), C.byref(mc)) == 0
    assert emu.glvemu_bar_tiles_check(1024, 259, C.c_float(0.025), C.c_float(0.5), 288, 4, C.byref(nr), C.byref(mc)) == 0
    assert emu.glvemu_bar_tiles_check(16384, 256, C.c_float(0.025), C.c_float(0.0), 288, 4, C.byref(nr), C.byref(mc)) == -1      # tiles, no rounds
    # below 256 bars the chunked order applies (the modules' 80 bars): no tables
    assert emu.glvemu_bar_tiles_check(4096, 80, C.c_float(0.025), C.c_float(0.0), 288, 4, C.byref(nr), C.byref(mc)) == -2


@pytest.mark.parametrize("n,bars,bins,phase", [(256, 256, 160, 0.5), (512, 512, 160, 0.5), (1024, 1024, 160, 0.5), (2048, 2048, 160, 0.5), (4096, 4096, 288, 0.5),
                                               (8192, 8192, 288, 0.5), (2048, 1001, 160, 0.0), (1024, 259, 160, 0.5), (16384, 256, 288, 0.0)])
def test_many_bars_arithmetic_is_the_documented_one(emu, oracle, n, bars, bins, phase):
    """The many-bars kernels run every chain of a tile over the tile's common, padded bin range (glvemu_bars_rows restates that off the
    same host tables); the oracle walks every bar's own taps (glvo_bars_chunked_at, bars >= 256: one fmaf chain in bin order).  Bit
    for bit the same -- taps of weight +0 leave a chain untouched -- for noise, for texels outside [0, 1], NaN and Inf (clamped), for
    an all-zero and an all-one row."""
    import ctypes as C
    rng = np.random.default_rng(n + bars)
    rows = [rng.random(n, dtype=np.float32), (rng.standard_normal(n) * 2).astype(np.float32), np.zeros(n, np.float32), np.ones(n, np.float32)]
    rows += [(rng.random(n, dtype=np.float32) ** 2 * np.float32(1.3) - np.float32(0.05)).astype(np.float32) for _ in range(8)]
    rows[1][::7] = np.nan
    rows[1][3::11] = np.inf
    fp = C.POINTER(C.c_float)
    emu.glvemu_bars_rows.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, fp]
    emu.glvemu_bars_rows.restype = C.c_int
    for tex in rows:
        got = np.full(bars, -1, np.float32)
        assert emu.glvemu_bars_rows(tex.ctypes.data_as(fp), n, bars, 0.025, phase, bins, got.ctypes.data_as(fp)) == 0
        want = np.zeros(bars, np.float32)
        oracle.lib().glvo_bars_chunked_at(np.ascontiguousarray(tex), n, want, bars, 0.025, phase)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (n, bars, int((~same).sum()))


@pytest.mark.parametrize("n,bars,bins,factor,phase", [(512, 512, 160, 0.025, 0.5), (1024, 1024, 160, 0.025, 0.5), (2048, 2048, 160, 0.025, 0.5), (4096, 4096, 288, 0.025, 0.5),
                                                      (8192, 8192, 448, 0.025, 0.5), (16384, 16384, 832, 0.025, 0.5), (32768, 32768, 1600, 0.025, 0.5),
                                                      (2048, 1001, 288, 0.025, 0.0), (1024, 259, 288, 0.025, 0.5), (4096, 4096, 832, 0.12, 0.25), (4096, 300, 832, 0.005, 0.0)])
def test_integer_tables_of_the_texel_rows_pass(emu, oracle, n, bars, bins, factor, phase):
    """Many bars over TEXEL rows (the library's GL chains): exact integer arithmetic on the i8 matrix cores off host tables -- per bar
    integer weights that sum to 2^P, as balanced signed byte digits in MFMA operand layout, the texels as two planes of signed bytes in an
    LDS ring (glv_tables.h make_bar_itiles).  glvemu_bars_int walks those tables the way the kernel does -- ring slots overwritten as the
    ring wraps, every tile evaluated before AND after the next round's bins are parked, the kernel's five-instruction epilogue -- and must
    give the oracle's independent restatement (glvo_bars_int_at: the bars' own taps, 64-bit integers): the same texels, the same float
    bits.  Saturated, zero and random rows; and the result lies within 0.03 texel steps of the weighted mean with the shader's weights
    in float64 (glvo_bars_at_exact) -- closer than any float summation order."""
    import ctypes as C
    rng = np.random.default_rng(n * 3 + bars)
    u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS"); f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    emu.glvemu_bars_int.argtypes = [u16p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, u16p, f32p]
    emu.glvemu_bars_int.restype = C.c_int
    rows = [(rng.random(n) ** 2 * 65535.99).astype(np.uint16), np.full(n, 65535, np.uint16), np.zeros(n, np.uint16), rng.integers(0, 65536, n).astype(np.uint16),
            (rng.random(n) < 0.5).astype(np.uint16) * np.uint16(65535)]
    for tex in rows:
        g16 = np.full(bars, 7, np.uint16); gf = np.full(bars, -1, np.float32)
        assert emu.glvemu_bars_int(tex, n, bars, factor, phase, bins, g16, gf) == 0
        w16, wf = oracle.bars_int(tex, bars, factor, phase)
        assert (g16 == w16).all(), (n, bars, int((g16 != w16).sum()))
        assert ((gf.view(np.uint32) == wf.view(np.uint32)) | (np.isnan(gf) & np.isnan(wf))).all()
        ex = np.empty(bars, np.float64); nt = np.empty(bars, np.int32); frag = np.empty(bars, np.int32)
        oracle.lib().glvo_bars_at_exact((tex.astype(np.float32) / np.float32(65535)).copy(), n, ex, nt, frag, bars, factor, phase, 4)
        ok = ~np.isnan(wf)
        assert np.abs(wf[ok].astype(np.float64) * 65535 - ex[ok] * 65535).max() <= 0.03 + 65535 * 2.0 ** -23
    if (n, bins) == (4096, 288):
        # a ring too small for the tiles gives no rounds (the library then falls back to the float chain's kernels), never wrong numbers
        assert emu.glvemu_bars_int(rows[0], n, bars, factor, phase, 160, g16, gf) == -3


@pytest.mark.parametrize("n,b_stride", [(1024, 16), (4096, 64)])
def test_rows_kernel_division_by_reciprocal_is_the_quotient(emu, n, b_stride):
    """The many-bars kernels divide a bar's total by its weight sum in three instructions -- q0 = a * r, rem = fma(-q0, b, a),
    q = fma(rem, r, q0) with the host's r = RN(1 / b) (Markstein) -- where the table says that is exact.  Checked for the weight
    sums of the pre-smoothing pass against a / b over EVERY significand of a in four binades (the two around b, the smallest and the
    largest totals the fast path sees): 2^25 quotients per bar, no mismatch.  By default every 16th / 64th distinct weight sum (the CPU
    suite stays short); GLV_FULL_CHECKS=1 takes them all (round 4: 1.3e11 quotients for n = 4096, none differs)."""
    import ctypes as C
    emu.glvemu_div_rcp_check.restype = C.c_ulonglong
    emu.glvemu_div_rcp_check.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    if os.environ.get("GLV_FULL_CHECKS"):
        b_stride = 1
    checked = C.c_ulonglong(0)
    bad = emu.glvemu_div_rcp_check(n, n, 0.025, 0.5, min(16, os.cpu_count() or 2), b_stride, C.byref(checked))
    assert bad == 0 and checked.value >= (n // (2 * b_stride) - 16) * (1 << 25), (bad, checked.value)


def test_tile_tables_over_random_parameters(emu, oracle):
    """A seeded sweep of sizes, bar counts (>= 256), smoothing widths, phases and ring sizes: the tile / round tables always pass their
    invariants (or report that no rounds exist: -1), and the chains walked off them equal the oracle's per-bar chains bit for bit."""
    import ctypes as C
    rng = np.random.default_rng(77)
    fp = C.POINTER(C.c_float)
    emu.glvemu_bars_rows.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, fp]
    emu.glvemu_bars_rows.restype = C.c_int
    with_rounds = 0
    for trial in range(40):
        n = int(rng.choice([256, 512, 1024, 2048, 4096]))
        bars = int(rng.integers(256, n + 1))
        factor = float(rng.choice([0.004, 0.0125, 0.025, 0.06, 0.11]))
        phase = float(rng.choice([0.0, 0.5, 0.3]))
        bins = int(rng.choice([160, 288, 448, 832]))
        nr, mc = C.c_uint(0), C.c_uint(0)
        rc = emu.glvemu_bar_tiles_check(n, bars, C.c_float(factor), C.c_float(phase), bins, 4, C.byref(nr), C.byref(mc))
        assert rc in (0, -1), (trial, n, bars, factor, phase, bins, rc)
        with_rounds += rc == 0
        if trial % 4 == 0:
            tex = (rng.random(n, dtype=np.float32) * np.float32(1.1) - np.float32(0.03)).astype(np.float32)
            got = np.full(bars, -1, np.float32)
            assert emu.glvemu_bars_rows(tex.ctypes.data_as(fp), n, bars, factor, phase, bins, got.ctypes.data_as(fp)) == 0
            want = np.zeros(bars, np.float32)
            oracle.lib().glvo_bars_chunked_at(np.ascontiguousarray(tex), n, want, bars, factor, phase)
            same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
            assert same.all(), (trial, n, bars, factor, phase)
    assert with_rounds >= 10
