#!/usr/bin/env python3
"""Where does the fused-bars time go?  N=16384, fft+gravity: no bars / 1 bar / 8 / 80 / 160 bars."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
sys.path.insert(0, os.path.join(ROOT, "tools"))
from configs_bench import timed
n, streams = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 8192
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda", generator=gen)
sync = torch.cuda.synchronize
ops = G.OP_FFT | G.OP_GRAVITY
for bars in (0, 1, 8, 80, 160):
    b = G.Batch(G.Params(n=n, bars=max(bars, 1)), streams, ops)
    out = torch.empty((streams, 2, max(bars, 1)), dtype=torch.float32, device="cuda")
    if bars == 0:
        dt = timed(lambda: b.process_s16(pcm, None, ops), sync)
    else:
        dt = timed(lambda: b.process_s16(pcm, out, ops | G.OP_BARS), sync)
    print(f"N={n} bars={bars:3d}: {dt * 1e3:.3f} ms  {streams / dt / 1e6:.2f} M frames/s")
    b.close()
