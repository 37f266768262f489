#!/usr/bin/env python3
"""fft -> gravity -> average(F=5, windowed), spectra out, per size (equal PCM bytes): the chain BASELINE's metric names."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
sync = torch.cuda.synchronize
ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
for n in (2048, 4096, 8192):
    streams = 32768 * 4096 // n
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    spec = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n), streams, ops)
    dt = timed(lambda: b.process_s16(pcm, spec, ops), sync)
    print(f"N={n:5d} x {streams:6d} fft+gravity+average: {dt*1e3:.3f} ms  {streams/dt/1e6:7.2f} M frames/s  {streams/dt*52*n/8e12*100:5.1f} % of 8 TB/s (52N B/frame)")
    b.close()
