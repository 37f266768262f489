#!/usr/bin/env python3
"""The chain GLava's bars/radial modules really run -- fft -> gravity -> average(F=5, windowed) -> 80 bars per
channel -- at the shipped N=4096, 65536 streams: spectra out, bars out (fused), bars out (two kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
n, streams, bars = 4096, 65536, 80
sync = torch.cuda.synchronize
ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
b = G.Batch(G.Params(n=n, bars=bars), streams, ops | G.OP_BARS)
spec = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
dbars = torch.empty((streams, 2, bars), dtype=torch.float32, device="cuda")
dt = timed(lambda: b.process_s16(pcm, spec, ops), sync)
print(f"fft+gravity+average, spectra out       : {dt*1e3:.3f} ms  {streams/dt/1e6:6.2f} M frames/s  {streams/dt*52*n/8e12*100:5.1f} % of 8 TB/s (52N B/frame)")
dt = timed(lambda: b.process_s16(pcm, dbars, ops | G.OP_BARS), sync)
tag = "two kernels" if os.environ.get("GLV_UNFUSED_BARS") else "fused"
print(f"fft+gravity+average+bars ({tag:11s}): {dt*1e3:.3f} ms  {streams/dt/1e6:6.2f} M frames/s  {streams/dt*(44*n+640)/8e12*100:5.1f} % of 8 TB/s (44N+640 B/frame: no spectra written)")
b.close()
