#!/usr/bin/env python3
"""The stateful chain fft -> gravity -> average(F) with spectra out, per size and F (equal PCM bytes): 4N + 8N(F+1) B/frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
sync = torch.cuda.synchronize
ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
for n, F in ((1024, 5), (4096, 5), (4096, 6), (4096, 2), (8192, 5), (16384, 5)):
    streams = 32768 * 4096 // n
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n, avg_frames=F), streams, ops)
    dt = timed(lambda: b.process_s16(pcm, out, ops), sync)
    byt = 4 * n + 8 * n * (F + 1)
    print(f"N={n:5d} F={F} x {streams:6d}: {dt*1e3:.3f} ms  {streams/dt/1e6:7.2f} M frames/s  {streams/dt*byt/8e12*100:5.1f} % of 8 TB/s ({byt//n}N B/frame)")
    b.close(); del pcm, out
