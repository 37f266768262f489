#!/usr/bin/env python3
"""glv_smooth_kernel (the CPU-path transform_smooth, render.c:694-718, batched): time per launch and effective traffic.
Only the first n/ratio outputs of a row are written and only ~1.01 n/ratio of its floats are read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
sync = torch.cuda.synchronize
for n, streams in ((1024, 65536), (4096, 16384), (4096, 65536), (16384, 4096)):
    x = torch.rand((streams * 2, n), dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    b = G.Batch(G.Params(n=n), streams, 0)
    t_copy = timed(lambda: b.process_f32(x, y, G.OP_WRANGE), sync)       # one grid-stride pass over the rows (reference point)
    t_sm = timed(lambda: b.process_f32(x, y, G.OP_SMOOTH), sync)
    touched = streams * 2 * (n / 4) * 4 * 2.01                           # read ~1.01 n/4 floats, write n/4 floats per row
    print(f"N={n:5d} x {streams:6d} streams: smooth {t_sm*1e3:8.3f} ms (wrange pass over the same rows {t_copy*1e3:.3f} ms)  "
          f"{streams/t_sm/1e6:7.2f} M frames/s, {touched/t_sm/1e9:7.1f} GB/s of the floats it touches")
    b.close(); del x, y
