"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libglvref.so).

Run in the build container (needs /root/reference to build oracle/_ref):

    python tests/golden/make_golden.py

The reference's own tests pin no spectral value, so these vectors --
outputs of the unmodified glava/render.c transform_* and glava/fifo.c entry compiled
with the recipe in oracle/Makefile -- are the committed ground truth the CPU test-suite
holds the oracle restatement to, also on machines where the reference cannot be built.
Inputs are derived from the LCG of SURVEY.md 8c so the files stay small.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle_lib import Ref, RefStream, lcg_pcm_fast  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main() -> None:
    assert Ref.available(), "oracle/_ref/libglvref.so missing and /root/reference absent"
    vecs = {}
    # 1. transform_fft on s16/65535 input, several sizes (seed = 12345 + n); full outputs
    for n in (512, 1024, 4096, 16384):
        v = lcg_pcm_fast(12345 + n, n)
        x = v.astype(np.float32) / np.float32(65535)
        vecs[f"fft_n{n}_seed{12345 + n}"] = Ref.fft(x)
    # the exact SURVEY.md 8c KAT (seed 12345, N=4096)
    v = lcg_pcm_fast(12345, 4096)
    vecs["fft_kat_survey"] = Ref.fft(v.astype(np.float32) / np.float32(65535))
    # non-default fft_scale / fft_cutoff
    vecs["fft_n1024_scale3_cut0p7"] = Ref.fft(lcg_pcm_fast(7, 1024).astype(np.float32) / np.float32(65535),
                                              Ref.params(fft_scale=3.0, fft_cutoff=0.7))
    # silence and full-scale DC
    vecs["fft_n1024_zeros"] = Ref.fft(np.zeros(1024, np.float32))
    vecs["fft_n1024_dc"] = Ref.fft(np.full(1024, 32767 / 65535, np.float32))

    # 2. fft -> gravity -> average chain, 9 frames, stereo, F=5 windowed and F=6 unwindowed, F=1
    for tag, F, win in (("F5w", 5, True), ("F6u", 6, False), ("F1w", 1, True)):
        n = 1024
        rs = RefStream(Ref.params(avg_frames=F, avg_window=win))
        outs = []
        for fr in range(9):
            pcm = lcg_pcm_fast(500 + fr, 2 * n).reshape(n, 2)
            l = pcm[:, 0].astype(np.float32) / np.float32(65535)
            r = pcm[:, 1].astype(np.float32) / np.float32(65535)
            outs.append(rs.frame_from_float(l, r))
        rs.close()
        vecs[f"chain_n1024_{tag}"] = np.stack(outs)
    # gravity only (no average), 5 frames
    rs = RefStream(Ref.params(), average=False)
    outs = []
    for fr in range(5):
        pcm = lcg_pcm_fast(900 + fr, 2 * 512).reshape(512, 2)
        outs.append(rs.frame_from_float(pcm[:, 0].astype(np.float32) / np.float32(65535),
                                        pcm[:, 1].astype(np.float32) / np.float32(65535)))
    rs.close()
    vecs["chain_n512_gravity_only"] = np.stack(outs)

    # 3. wrange
    p = Ref.params()
    b = (lcg_pcm_fast(31, 256).astype(np.float32) / np.float32(65535))
    Ref.lib().glvref_wrange(C.byref(p), b, b.size)
    vecs["wrange_seed31"] = b

    # 4. FIFO backend through a real named pipe: rings after each data update (stereo + mono)
    for ch in (2, 1):
        ssz, fsz, chunks = 1024, 4096, 6
        pcm = lcg_pcm_fast(4242 + ch, chunks * ssz // 2)
        pcm[0] = -32768; pcm[1] = 32767; pcm[2] = 1; pcm[3] = -1     # SURVEY 8c unpack KATs
        rings = np.zeros((64, 2, fsz), np.float32)
        zf = np.zeros(64, np.uint8)
        nev = C.c_size_t(0)
        rc = Ref.lib().glvref_fifo_run(f"/tmp/glv_golden_{os.getpid()}.fifo".encode(), pcm, chunks, ssz, fsz, ch,
                                       rings, zf, 64, C.byref(nev))
        assert rc == 0, rc
        nev = nev.value
        vecs[f"fifo_ch{ch}_pcm"] = pcm
        vecs[f"fifo_ch{ch}_rings"] = rings[:nev]
        vecs[f"fifo_ch{ch}_zero_fill"] = zf[:nev]

    np.savez_compressed(os.path.join(OUT, "reference_vectors.npz"), **vecs)
    total = sum(v.nbytes for v in vecs.values())
    print(f"wrote {len(vecs)} arrays, {total / 1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
