#!/usr/bin/env python3
"""Where the transform + GL passes kernel of the SHIPPED pipeline stands next to the same kernel in the `av`-texel chain:
    python tools/k1_probe.py [streams ...]          (default 16384 65536)
For every batch size: wall ms per call (back to back, after spin-up) of
    gl_default   FFT|GRAVITY|AVERAGE|R16          one launch, whole texel rows out (28 N B / frame)
    gl_sm        ... |BARS, bars = n              two launches, K1 writes 0.31 n texels of every row, the i8 pass the `sm` rows
over the frame kernel's workgroup counts and both kernel configurations -- is the default plan (glv_api.cpp default_grid) the
right one for the chain whose second kernel leaves 268 MB of dirty lines behind?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G

n = 4096
sizes = [int(x) for x in sys.argv[1:]] or [16384, 65536]


def timed(call, seconds=0.25, reps=3):
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        for _ in range(4): call()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        k, t0 = 0, time.perf_counter()
        while True:
            for _ in range(16): call()
            k += 16
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > seconds: break
        best = min(best, (time.perf_counter() - t0) / k)
    return best * 1e3


for streams in sizes:
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    out = torch.empty((streams, 2, n), dtype=torch.int16, device="cuda")
    kw = dict(avg_window_kind=1, gl_storage=1)
    ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16
    for name, extra, pk, mask in (("gl_default", 0, dict(), G.OP_GRAVITY | G.OP_AVERAGE),
                                  ("gl_sm", G.OP_BARS, dict(bars=n, bar_phase=0.5), G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS)):
        b = G.Batch(G.Params(n=n, **kw, **pk), streams, mask)
        call = lambda: b.process_s16(pcm, out, ops | extra)
        call(); torch.cuda.synchronize()
        dgrid, dvar = b.last_grid(), b.last_variant()
        base = timed(call)
        bytes28 = 28 * n * streams
        print(f"{name:10s} streams {streams:6d}  default plan: variant {dvar} grid {dgrid:5d}  {base:.4f} ms  {bytes28 / base / 8e9:.4f} of 28N  launches {b.last_launches()}", flush=True)
        rows = []
        for v in range(b.variants()):
            b.set_variant(v)
            for g in (256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096):
                b.set_grid(g)
                try:
                    ms = timed(call, 0.12, 2)
                except Exception as e:       # a plan the size does not take
                    print("   variant", v, "grid", g, "->", e); continue
                rows.append((ms, v, g))
        rows.sort()
        print("   best:", "  ".join(f"v{v} g{g} {ms:.4f}" for ms, v, g in rows[:5]), " worst:", "  ".join(f"v{v} g{g} {ms:.4f}" for ms, v, g in rows[-2:]), flush=True)
        b.close()
