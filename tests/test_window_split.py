"""The s16 window product without fp64 (glava_amd/csrc/glv_core.h apply_window_split).

render.c:794 multiplies a float sample by a double window value and rounds the double product to float.  For s16 input the
library computes fma(x, hi, x * lo) with a float pair per window position instead -- bit-identical for every one of the 65536
sample values, which is searched for and checked on the device when a batch is created.  Here:
  CPU   the oracle's restatement of the search (glvo_window_split_mismatches): the pair exists for every position of the sizes
        the CPU finishes in seconds, at most a handful of positions need their low part moved, by one ulp;
  GPU   glv_batch_window_selftest: every (sample value, position) pair of EVERY size against the fp64 product, on the device;
        and, end to end, the transform of frames that hold every sample value (raw FFT bit-exact against the compiled reference's
        restatement is asserted throughout tests/test_gpu_parity.py -- with the split window in the s16 kernels since round 3)."""
import ctypes as C

import numpy as np
import pytest

from oracle_lib import Oracle


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192, 16384])
def test_oracle_split_search_finds_an_exact_pair_everywhere(n):
    L = Oracle.lib()
    L.glvo_window_split_mismatches.restype = C.c_long
    L.glvo_window_split_mismatches.argtypes = [C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    sh, mx = C.c_int(0), C.c_int(0)
    assert L.glvo_window_split_mismatches(n, C.byref(sh), C.byref(mx)) == 0
    assert sh.value <= 4 and mx.value <= 1, (sh.value, mx.value)      # 0 0 0 0 1 2 4 (and 7 at n = 32768) with glibc


@pytest.mark.gpu
@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
def test_device_split_window_equals_the_fp64_product_for_every_sample_and_position(glvlib, n):
    G = glvlib
    b = G.Batch(G.Params(n=n), 1, G.OP_FFT)
    mism, shifted = b.window_selftest()
    b.close()
    assert mism == 0
    assert shifted <= 8, shifted          # 0 / 0 / 0 / 0 / 1 / 2 / 4 / 7 with glibc's cos()


@pytest.mark.gpu
@pytest.mark.parametrize("n,mono", [(4096, False), (1024, True), (16384, False)])
def test_every_sample_value_through_the_transform(glvlib, n, mono):
    """frames whose samples walk through all 65536 s16 values at shifting positions (every value meets n / 65536-th of the window
    positions per frame; 16 frames with different offsets), stereo and mono: the raw FFT must be the reference's, bit for bit"""
    import torch
    G = glvlib
    streams = 16
    k = np.arange(n, dtype=np.int64)
    pcm = np.empty((streams, n, 2), np.int16)
    for s in range(streams):
        pcm[s, :, 0] = ((k * 16 + s * 4099 + 7) % 65536 - 32768).astype(np.int16)
        pcm[s, :, 1] = ((k * 48 + s * 8191 + 12345) % 65536 - 32768).astype(np.int16)
    b = G.Batch(G.Params(n=n, channels=1 if mono else 2), streams, G.OP_FFT)
    d_out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    b.process_s16(torch.from_numpy(pcm).cuda(), d_out, G.OP_FFT | G.OP_RAW)
    got = d_out.cpu().numpy().reshape(streams, 2, n)
    b.close()
    for s in range(streams):
        for ch in range(2):
            if mono:
                x = np.trunc((pcm[s, :, 0].astype(np.int32) + pcm[s, :, 1].astype(np.int32)) / 2).astype(np.int32)      # fifo.c:99: C int division (truncation)
                row = x.astype(np.float32) / np.float32(65535)
            else:
                row = pcm[s, :, ch].astype(np.float32) / np.float32(65535)
            _, want = Oracle.transform_fft(row, want_raw=True)
            assert (got[s, ch].view(np.uint32) == want.view(np.uint32)).all(), (s, ch)
