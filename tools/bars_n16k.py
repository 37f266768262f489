#!/usr/bin/env python3
"""N=16384 / 32768 fused bars (fft -> gravity [-> average] -> 80 / 160 bars): the A/B benchmark of the fused loop's batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
sync = torch.cuda.synchronize
for n in (16384, 32768):
    streams = 8192 * 16384 // n
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    for bars in (80, 160):
        dbars = torch.empty((streams, 2, bars), dtype=torch.float32, device="cuda")
        for name, ops, byt in (("fft+gravity+bars        ", G.OP_FFT | G.OP_GRAVITY | G.OP_BARS, 20 * n + 8 * bars),
                               ("fft+gravity+average+bars", G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, 44 * n + 8 * bars)):
            b = G.Batch(G.Params(n=n, bars=bars), streams, ops & ~G.OP_BARS)
            dt = timed(lambda: b.process_s16(pcm, dbars, ops), sync)
            print(f"N={n:5d} x {streams:5d} {bars:3d} bars {name}: {dt*1e3:.3f} ms  {streams/dt*byt/8e12*100:5.1f} % of 8 TB/s")
            b.close()
