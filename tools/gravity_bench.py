#!/usr/bin/env python3
"""SURVEY 8d row B (fft + gravity, 20N B/frame when the spectra stay in the state buffer) per size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
sync = torch.cuda.synchronize
ops = G.OP_FFT | G.OP_GRAVITY
for n in (1024, 4096, 8192, 16384):
    streams = 32768 * 4096 // n
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n), streams, ops)
    dt0 = timed(lambda: b.process_s16(pcm, None, ops), sync)
    dt1 = timed(lambda: b.process_s16(pcm, out, ops), sync)
    print(f"N={n:5d} x {streams:6d}: state only {dt0*1e3:.3f} ms {streams/dt0*20*n/8e12*100:5.1f} % (20N)   with spectra out {dt1*1e3:.3f} ms {streams/dt1*28*n/8e12*100:5.1f} % (28N moved)")
    b.close(); del pcm, out
if len(sys.argv) > 1:
    n, streams = 4096, 32768
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    for g in (256, 512, 1024, 2048, 4096):
        b = G.Batch(G.Params(n=n), streams, ops); b.set_grid(g)
        dt0 = timed(lambda: b.process_s16(pcm, None, ops), sync)
        print(f"N=4096 grid {g}: state only {dt0*1e3:.3f} ms")
        b.close()
    for nn in (512, 2048):
        s2 = 32768 * 4096 // nn
        pcm2 = torch.randint(-32768, 32768, (s2, nn, 2), dtype=torch.int16, device="cuda")
        b = G.Batch(G.Params(n=nn), s2, ops)
        dt0 = timed(lambda: b.process_s16(pcm2, None, ops), sync)
        print(f"N={nn}: state only {dt0*1e3:.3f} ms {s2/dt0*20*nn/8e12*100:5.1f} %")
        b.close()
