"""Launch wisdom (the role of glfft's FFTWisdom): file format and table on the CPU,
measured autotuning, persistence and result-invariance on the GPU."""
import os

import numpy as np
import pytest


def test_wisdom_file_roundtrip(glvlib, tmp_path):
    """v2 format: device compute_units n input_kind ops_class log_mode log2(streams) avg_frames variant workgroups ms"""
    G = glvlib
    G.wisdom_clear()
    assert G.wisdom_count() == 0
    p = tmp_path / "w.txt"
    p.write_text("# comment\ngfx950:sramecc+:xnack- 256 8192 0 0 1 15 0 1 256 0.697000\ngfx950:sramecc+:xnack- 256 16384 0 1 1 14 5 0 512 1.270000\nnot an entry\n"
                 "gfx950 256 4096 0 0 1 16 0 0 0 0.5\n8192 0 0 1 15 256 0.697000\ngfx942 304 8192 0 0 1 15 0 0 304 0.9\n")
    G.wisdom_load(str(p))
    assert G.wisdom_count() == 3                      # the malformed line, the grid-0 line and the round-2 (v1) line are ignored ...
    msg = G.lib().glv_last_error().decode()           # ... but not silently
    assert "3 entries loaded" in msg and "3 line(s) skipped" in msg and "v1" in msg, msg
    old = tmp_path / "v1.txt"
    old.write_text("# glv launch wisdom v1\n8192 0 0 1 15 256 0.697000\n4096 0 0 1 16 512 0.650000\n")
    with pytest.raises(G.GlvError) as ei:             # a file that yields nothing is an error
        G.wisdom_load(str(old))
    assert ei.value.code == G.ERR_INVALID and "v1" in str(ei.value)
    q = tmp_path / "out.txt"
    G.wisdom_save(str(q))
    lines = [l.split() for l in q.read_text().splitlines() if not l.startswith("#")]
    assert sorted(l[:10] for l in lines) == [["gfx942", "304", "8192", "0", "0", "1", "15", "0", "0", "304"],
                                             ["gfx950:sramecc+:xnack-", "256", "16384", "0", "1", "1", "14", "5", "0", "512"],
                                             ["gfx950:sramecc+:xnack-", "256", "8192", "0", "0", "1", "15", "0", "1", "256"]]
    G.wisdom_load(str(q))                             # loading what was saved changes nothing
    assert G.wisdom_count() == 3
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


@pytest.mark.gpu
@pytest.mark.parametrize("n,streams", [(4096, 8192), (8192, 4096), (1024, 32768), (16384, 2048), (32768, 1024)])
def test_wisdom_selects_the_kernel_variant(glvlib, tmp_path, n, streams):
    """f4: the wisdom chooses WHICH kernel configuration runs (radix split / rows per workgroup / table placement), not only the
    grid.  autotune times every configuration built for the size and records the winner; the entry is keyed on the device, so a
    file written for another part is not applied; a saved file reproduces the choice in a fresh table -- also a NON-default one
    (the file is edited to name the runner-up, which the next launch then uses, with identical spectra)."""
    import torch
    G = glvlib
    G.wisdom_clear()
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    ref = torch.empty_like(out)
    b = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    nv = b.variants()
    assert nv >= 2
    b.process_s16(pcm, ref, G.OP_FFT)
    assert b.last_variant() == 0                                   # nothing tuned: the build-time choice
    grid, ms = b.autotune(pcm, out, G.OP_FFT)
    b.process_s16(pcm, out, G.OP_FFT)
    chosen = b.last_variant()
    assert 0 <= chosen < nv and b.last_grid() == grid
    f = tmp_path / "wisdom.txt"
    G.wisdom_save(str(f))
    entry = [l.split() for l in f.read_text().splitlines() if not l.startswith("#")]
    assert len(entry) == 1 and int(entry[0][8]) == chosen and int(entry[0][9]) == grid and int(entry[0][2]) == n
    dev, cus = entry[0][0], int(entry[0][1])
    assert cus > 0 and dev != "unknown"
    # the same measurements under another device's name (or CU count) are not this device's wisdom
    other = 1 - chosen if nv == 2 else (chosen + 1) % nv
    G.wisdom_clear()
    (tmp_path / "foreign.txt").write_text(" ".join(["some_other_gpu", str(cus)] + entry[0][2:8] + [str(other), "7", "1.0"]) + "\n"
                                          + " ".join([dev, str(cus + 8)] + entry[0][2:8] + [str(other), "7", "1.0"]) + "\n")
    G.wisdom_load(str(tmp_path / "foreign.txt"))
    assert G.wisdom_count() == 2
    b.process_s16(pcm, out, G.OP_FFT)
    assert b.last_variant() == 0 and b.last_grid() != 7
    # the runner-up named in this device's file is what runs next -- in this batch (cache invalidated) and in a fresh one
    G.wisdom_clear()
    (tmp_path / "edited.txt").write_text(" ".join(entry[0][:8] + [str(other), entry[0][9], entry[0][10]]) + "\n")
    G.wisdom_load(str(tmp_path / "edited.txt"))
    b.process_s16(pcm, out, G.OP_FFT)
    assert b.last_variant() == other
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    b2 = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    b2.process_s16(pcm, out, G.OP_FFT)
    assert b2.last_variant() == other and torch.equal(out.view(torch.int32), ref.view(torch.int32))
    # an averaging chain has its own entries (avg_frames is part of the key): the stateless entry does not apply to it
    bc = G.Batch(G.Params(n=n, avg_frames=5), streams, G.OP_GRAVITY | G.OP_AVERAGE)
    bc.process_s16(pcm, out, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE)
    assert bc.last_variant() == 0
    # an explicit choice beats the wisdom
    b2.set_variant(chosen if chosen != other else 0)
    b2.process_s16(pcm, out, G.OP_FFT)
    assert b2.last_variant() == (chosen if chosen != other else 0)
    for x in (b, b2, bc): x.close()
    G.wisdom_clear()


@pytest.mark.gpu
@pytest.mark.parametrize("gl", [0, 1])
def test_placement_tuning_keeps_the_results_and_never_the_slower_placement(glvlib, gl):
    """Placement wisdom (round 6, profiles/r06/modes.txt): a stateful chain's speed depends on where its state arrays lie relative to the caller's output
    buffer; glv_batch_tune_placement times fresh allocations with the caller's buffers and keeps the fastest.  Checked: the placement it ends with is not
    slower than the one it started with, the state is reset, the updates that follow equal an untuned batch's bit for bit (state on other frames changes
    no value), arguments are validated, and a stateless chain is refused."""
    import torch
    G = glvlib
    n, streams, F = 4096, 8192, 5
    kw = dict(n=n, avg_frames=F, avg_window_kind=1, gl_storage=gl)
    mask = G.OP_GRAVITY | G.OP_AVERAGE
    ops = G.OP_FFT | mask | (G.OP_R16 if gl else 0)
    dt = torch.int16 if gl else torch.float32
    pcm = [torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda") // d for d in (1, 16, 3)]
    out, ref = torch.empty((streams, 2, n), dtype=dt, device="cuda"), torch.empty((streams, 2, n), dtype=dt, device="cuda")
    tuned, plain = G.Batch(G.Params(**kw), streams, mask), G.Batch(G.Params(**kw), streams, mask)
    tuned.process_s16(pcm[0], out, ops)                                 # state that the tuning must wipe
    first, best = tuned.tune_placement(pcm[0], out, ops, candidates=4)
    assert first > 0 and 0 < best <= first * 1.001, (first, best)
    for u in range(F + 2):
        tuned.process_s16(pcm[u % 3], out, ops); plain.process_s16(pcm[u % 3], ref, ops)
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), u
    with pytest.raises(G.GlvError) as ei:
        tuned.tune_placement(pcm[0], out, G.OP_FFT, 3)
    assert ei.value.code == G.ERR_INVALID
    one, same = tuned.tune_placement(pcm[0], out, ops, candidates=1)    # one candidate: a measurement, nothing is moved
    assert one == same
    tuned.close(); plain.close()
