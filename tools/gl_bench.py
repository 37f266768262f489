#!/usr/bin/env python3
"""The GL-default pipeline (gl_storage 1: upload quantisation, GL_MAX + gravity, ring, average on uint16 state in ONE launch) at
N=4096 x 65536 streams, F=5: `av` texels out, 80 bars as texels out; with GLV_SPECTRUM_LIB an A/B build of the library.
    python tools/gl_bench.py [n streams]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 65536 * 4096 // n
F, bars = 5, 80
sync = torch.cuda.synchronize
ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
for lm in (1, 0):
    b = G.Batch(G.Params(n=n, bars=bars, avg_frames=F, avg_window_kind=1, gl_storage=1, log_mode=lm), streams, G.OP_GRAVITY | G.OP_AVERAGE)
    q = torch.empty((streams, 2, n), dtype=torch.int16, device="cuda")
    qb = torch.empty((streams, 2, bars), dtype=torch.int16, device="cuda")
    dt = timed(lambda: b.process_s16(pcm, q, ops | G.OP_R16), sync, steps=30)
    print(f"N={n} log_mode {lm} gl_default av texels : {dt*1e3:.3f} ms  {streams/dt/1e6:6.2f} M frames/s  {streams/dt*28*n/8e12*100:5.1f} % of 8 TB/s (28N B/frame)")
    b.reset()
    dt = timed(lambda: b.process_s16(pcm, qb, ops | G.OP_BARS | G.OP_R16), sync, steps=30)
    print(f"N={n} log_mode {lm} gl_default 80 bars   : {dt*1e3:.3f} ms  {streams/dt/1e6:6.2f} M frames/s  {streams/dt*(24*n+4*bars)/8e12*100:5.1f} % of 8 TB/s (24N+320 B/frame)")
    b.close()
