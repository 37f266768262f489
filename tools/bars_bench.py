#!/usr/bin/env python3
"""GLV_OP_BARS fused into the frame kernel (what GLava's bars / radial modules request), per size:
fft -> gravity -> average(F=5) -> 80 bars and fft -> gravity -> 80 bars (BASELINE configs[2]), equal PCM bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
sync = torch.cuda.synchronize
bars = 80
for n in (1024, 4096, 8192, 16384):
    streams = 32768 * 4096 // n
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    dbars = torch.empty((streams, 2, bars), dtype=torch.float32, device="cuda")
    for name, ops, byt in (("fft+gravity+average+bars", G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, 44 * n + 640),
                           ("fft+gravity+bars        ", G.OP_FFT | G.OP_GRAVITY | G.OP_BARS, 20 * n + 640)):
        b = G.Batch(G.Params(n=n, bars=bars), streams, ops & ~G.OP_BARS)
        dt = timed(lambda: b.process_s16(pcm, dbars, ops), sync)
        print(f"N={n:5d} x {streams:6d} {name}: {dt*1e3:.3f} ms  {streams/dt/1e6:7.2f} M frames/s  {streams/dt*byt/8e12*100:5.1f} % of 8 TB/s")
        b.close()
    del pcm, dbars
