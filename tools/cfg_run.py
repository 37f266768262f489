#!/usr/bin/env python3
"""A few launches of one configuration, for rocprofv3:  cfg_run.py configs2 | chain | n1024bars"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
which = sys.argv[1] if len(sys.argv) > 1 else "configs2"
if which == "configs2":
    n, streams, ops, bars = 16384, 8192, G.OP_FFT | G.OP_GRAVITY | G.OP_BARS, 80
elif which == "chain":
    n, streams, ops, bars = 4096, 32768, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE, 0
else:
    n, streams, ops, bars = 1024, 131072, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, 80
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
out = torch.empty((streams, 2, bars if bars else n), dtype=torch.float32, device="cuda")
b = G.Batch(G.Params(n=n, bars=max(bars, 1)), streams, ops & ~G.OP_BARS)
for _ in range(6):
    b.process_s16(pcm, out, ops)
torch.cuda.synchronize()
print(which, "n", n, "streams", streams, "algorithmic bytes per launch", b.algorithmic_bytes(ops) if hasattr(b, "algorithmic_bytes") else "-")
b.close()
