#!/usr/bin/env python3
"""glv_bars_kernel alone (row in HBM): 80 bars from resident spectra, per size, equal bytes of spectra."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
sync = torch.cuda.synchronize
bars = int(sys.argv[1]) if len(sys.argv) > 1 else 80
for n in (1024, 4096, 16384):
    rows = 2 * 8192 * 16384 // n
    spec = torch.rand((rows, n), dtype=torch.float32, device="cuda")
    out = torch.empty((rows, bars), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n, bars=bars), rows // 2, G.OP_FFT)
    dt = timed(lambda: b.bars(spec, out), sync)
    touched = 0.288 * n * 4 * rows
    print(f"N={n:5d} x {rows:7d} rows -> {bars} bars: {dt*1e3:.3f} ms   {rows/dt/1e6:8.2f} M rows/s   ~{touched/dt/1e12:.2f} TB/s of the row part the taps touch")
    b.close()
