#!/usr/bin/env python3
"""One GLava instance (one stereo stream, N = 4096): the default pipeline -- s16 PCM -> upload -> GL_MAX + gravity -> ring -> average ->
pre-smoothing pass -> `sm` texels -- per update, eager calls against one replayed hipGraph of F updates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from glava_amd import spectrum as G
n, F = 4096, 5
p = G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)
mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
ops = G.OP_FFT | mask | G.OP_R16
b = G.Batch(p, 1, mask)
pcm = torch.randint(-8000, 8000, (1, n, 2), dtype=torch.int16, device="cuda")
out = torch.zeros((2, n), dtype=torch.int16, device="cuda")
for _ in range(50): b.process_s16(pcm, out, ops)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): b.process_s16(pcm, out, ops)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 2000
t0 = time.perf_counter()
for _ in range(500):
    b.process_s16(pcm, out, ops); torch.cuda.synchronize()
sync = (time.perf_counter() - t0) / 500
print(f"one stream, N={n}, default pipeline with the pre-smoothing pass ({b.last_launches()} launches): {eager * 1e6:.1f} us per update back to back, "
      f"{sync * 1e6:.1f} us per update with a synchronize after each (launch + completion latency)")
b.close()
