"""Launch wisdom (the role of glfft's FFTWisdom): file format and table on the CPU,
measured autotuning, persistence and result-invariance on the GPU."""
import os

import numpy as np
import pytest


def test_wisdom_file_roundtrip(glvlib, tmp_path):
    G = glvlib
    G.wisdom_clear()
    assert G.wisdom_count() == 0
    p = tmp_path / "w.txt"
    p.write_text("# comment\n8192 0 0 1 15 256 0.697000\n16384 0 1 1 14 512 1.270000\nnot an entry\n4096 0 0 1 16 0 0.5\n")
    G.wisdom_load(str(p))
    assert G.wisdom_count() == 2                      # the malformed line and the grid-0 line are ignored
    q = tmp_path / "out.txt"
    G.wisdom_save(str(q))
    lines = [l for l in q.read_text().splitlines() if not l.startswith("#")]
    assert sorted(l.split()[:6] for l in lines) == [["16384", "0", "1", "1", "14", "512"], ["8192", "0", "0", "1", "15", "256"]]
    with pytest.raises(G.GlvError):
        G.wisdom_load(str(tmp_path / "missing.txt"))
    G.wisdom_clear()


@pytest.mark.gpu
def test_autotune_measures_records_and_is_used(glvlib, tmp_path):
    import torch
    G = glvlib
    G.wisdom_clear()
    n, streams = 8192, 4096
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    ref = torch.empty_like(out)
    b = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    b.process_s16(pcm, ref, G.OP_FFT)
    default_grid = b.last_grid()
    grid, ms = b.autotune(pcm, out, G.OP_FFT)
    assert grid > 0 and ms > 0 and G.wisdom_count() == 1
    b.process_s16(pcm, out, G.OP_FFT)
    torch.cuda.synchronize()
    assert b.last_grid() == grid                                   # the launch consults the wisdom
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32))   # the spectra do not depend on the workgroup count
    b.close()
    # another batch of the same description picks the entry up; a different stream count does not
    b2 = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    b2.process_s16(pcm, out, G.OP_FFT)
    assert b2.last_grid() == grid
    b2.close()
    f = tmp_path / "wisdom.txt"
    G.wisdom_save(str(f)); G.wisdom_clear()
    b3 = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    b3.process_s16(pcm, out, G.OP_FFT)
    assert b3.last_grid() == default_grid
    G.wisdom_load(str(f))
    b3.process_s16(pcm, out, G.OP_FFT)
    assert b3.last_grid() == grid
    b3.close()
    # a stateful chain: the probes are real updates, so the state is reset afterwards
    bs = G.Batch(G.Params(n=n), streams, G.OP_GRAVITY)
    g2, _ = bs.autotune(pcm, out, G.OP_FFT | G.OP_GRAVITY)
    bs.process_s16(pcm, out, G.OP_FFT | G.OP_GRAVITY)
    fresh = G.Batch(G.Params(n=n), streams, G.OP_GRAVITY)
    fresh.process_s16(pcm, ref, G.OP_FFT | G.OP_GRAVITY)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    bs.close(); fresh.close()
    G.wisdom_clear()
