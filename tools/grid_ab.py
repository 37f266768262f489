#!/usr/bin/env python3
"""Persistent-workgroup count A/B at one size: every kernel class at several grids, alternating.  python tools/grid_ab.py [n] [grids...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
grids = [int(x) for x in sys.argv[2:]] or [1024, 2048]
streams = 65536 * 4096 // n
sync = torch.cuda.synchronize
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
CH = G.OP_GRAVITY | G.OP_AVERAGE
cases = [("fft f32", dict(), G.OP_FFT, G.OP_FFT), ("fft R16", dict(), G.OP_FFT, G.OP_FFT | G.OP_R16), ("fft log0", dict(log_mode=0), G.OP_FFT, G.OP_FFT),
         ("chain F=5", dict(), CH, G.OP_FFT | CH), ("chain+80 bars", dict(), CH | G.OP_BARS, G.OP_FFT | CH | G.OP_BARS),
         ("GL chain R16", dict(gl_storage=1, avg_window_kind=1), CH, G.OP_FFT | CH | G.OP_R16), ("fft+gravity", dict(), G.OP_GRAVITY, G.OP_FFT | G.OP_GRAVITY)]
for name, kw, mask, ops in cases:
    b = G.Batch(G.Params(n=n, **kw), streams, mask)
    res = {g: [] for g in grids}
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        b.process_s16(pcm, out, ops); sync()
    for rep in range(3):
        for g in grids:
            b.set_grid(g)
            for _ in range(3): b.process_s16(pcm, out, ops)
            sync()
            b.timing_begin()
            for _ in range(20): b.process_s16(pcm, out, ops)
            sync()
            ms, nl = b.timing_end()
            res[g].append(ms / nl)
    print(f"N={n} {name:14s} " + "  ".join(f"grid {g}: " + "/".join(f"{x:.3f}" for x in res[g]) for g in grids), flush=True)
    b.close()
