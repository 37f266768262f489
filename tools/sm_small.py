#!/usr/bin/env python3
"""The pre-smoothing pass on FEW rows: the matrix-core kernel against the one-lane-per-bar kernel (GLV_NO_BARS_ROWS at table creation).
python tools/sm_small.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for rows in (2, 16, 64, 128, 256, 1024):
    res = {}
    for tag in ("rows", "seq"):
        if tag == "seq": os.environ["GLV_NO_BARS_ROWS"] = "1"
        else: os.environ.pop("GLV_NO_BARS_ROWS", None)
        b = G.Batch(G.Params(n=n, bars=n, bar_phase=0.5), max(rows // 2, 1), G.OP_FFT | G.OP_BARS)
        spec = torch.rand((rows, n), dtype=torch.float32, device="cuda"); out = torch.empty_like(spec)
        for _ in range(20): b.bars(spec, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): b.bars(spec, out)
        e1.record(); torch.cuda.synchronize()
        res[tag] = e0.elapsed_time(e1) / 200
        b.close()
    print(f"N={n} {rows:5d} rows: library's choice {res['rows'] * 1e3:8.1f} us   one lane per bar {res['seq'] * 1e3:8.1f} us", flush=True)
