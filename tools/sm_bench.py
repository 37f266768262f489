#!/usr/bin/env python3
"""The pre-smoothing pass alone (GLV_OP_BARS with bars == n at the texel centres, render.c:2277-2303) on rows already in HBM:
glv_bars_rows_kernel (>= 256 rows) against glv_bars_kernel (forced with fewer rows per call).   python tools/sm_bench.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
sync = torch.cuda.synchronize
for n in [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096]:
    streams = 16384 * 4096 // n
    spec = torch.rand((streams * 2, n), dtype=torch.float32, device="cuda")
    out = torch.empty((streams * 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n, bars=n, bar_phase=0.5), streams, G.OP_FFT | G.OP_BARS)
    dt = timed(lambda: b.bars(spec, out), sync, steps=10)
    tag = "glv_bars_seq_kernel (GLV_NO_BARS_ROWS)" if os.environ.get("GLV_NO_BARS_ROWS") else "glv_bars_rows_kernel"
    print(f"{tag}: N={n} bars=n, {streams * 2} rows: {dt * 1e3:.3f} ms  -> {streams / dt / 1e6:.2f} M stereo frames/s  ({dt * 1e9 / (streams * 2):.1f} ns per row)")
    b.close()
