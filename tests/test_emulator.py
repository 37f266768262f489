"""CPU: the kernel's per-thread phase code (glava_amd/csrc/glv_frame.h) walked by the host
emulator (tests/emu/glv_emu.cpp) must reproduce the oracle bit for bit -- this is where the
Stockham index maps, the pass plan for every size, the LDS swizzle, the twiddle gather and the
gravity/average state machine are debugged without a GPU."""
import numpy as np
import pytest

from emu_lib import emu_process
from oracle_lib import StreamOracle, lcg_pcm_fast

OP_FFT, OP_GRAVITY, OP_AVERAGE, OP_RAW = 1, 2, 4, 8


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096, 8192, 16384])
def test_fft_raw_and_magnitude_bit_exact(emu, oracle, n):
    units = 2
    pcm = lcg_pcm_fast(1000 + n, units * 2 * n)
    raw = emu_process(emu, n, pcm, units, OP_FFT | OP_RAW)
    mag = emu_process(emu, n, pcm, units, OP_FFT)
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


def test_lds_swizzle_is_a_permutation_and_conflict_free():
    """Model of the gfx950 LDS banking for the first exchange (MI355X_MICROARCH.md, LDS):
    ds_write_b64 is serviced in 16-lane groups over 32 4-byte banks, ds_read_b64 in 32-lane
    groups over 64 banks.  q ^ ((q>>4)&15) must be a bijection and conflict free for the
    pass-0 write pattern q = 16*tid + e and the pass-1 read pattern q = i*(nn/16) + tid."""
    for log_nn in range(8, 14):
        nn = 1 << log_nn
        T = nn // 16
        sw = lambda q: q ^ ((q >> 4) & 15)  # noqa: E731
        assert sorted(sw(q) for q in range(nn)) == list(range(nn))
        for e in range(16):                      # write: lanes = consecutive tids
            for g0 in range(0, min(T, 64), 16):
                banks = [(2 * sw(16 * t + e)) % 32 for t in range(g0, g0 + 16)]
                assert len(set(banks)) == 16, (log_nn, e, g0)
        rb = min(4, log_nn - 4)                  # radix bits of pass 1
        R = 1 << rb
        for i in range(R):                       # read: lanes = consecutive tids (one group per lane set)
            for g0 in range(0, min(T, 64), 32):
                lanes = range(g0, min(g0 + 32, T))
                banks = [(2 * sw(i * (nn // R) + t)) % 64 for t in lanes]
                assert len(set(banks)) == len(list(lanes)), (log_nn, i, g0)
