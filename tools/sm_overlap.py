#!/usr/bin/env python3
"""the GL-default pipeline with the pre-smoothing pass as TWO batches of half the streams on two HIP streams, so that
transform(k+1) and the many-bars pass (k) can be in flight together -- against one batch of all the streams on one stream.
    sm_overlap.py [streams] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n, F = 4096, 5
ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16
mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
p = G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)


def timed(fn):
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls * 1e3


pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
out = torch.empty((streams, 2, n), dtype=torch.int16, device="cuda")
one = G.Batch(p, streams, mask)
ms1 = timed(lambda: one.process_s16(pcm, out, ops))
one.close()
for parts in (2, 4):
    sub = streams // parts
    bs = [G.Batch(p, sub, mask) for _ in range(parts)]
    sts = [torch.cuda.Stream() for _ in range(parts)]

    def split():
        for i, (b, st) in enumerate(zip(bs, sts)):
            b.process_s16(pcm[i * sub:(i + 1) * sub], out[i * sub:(i + 1) * sub], ops, st.cuda_stream)
    ms = timed(split)
    print(f"sm chain N={n} x {streams} streams: one batch, one stream {ms1:.4f} ms = {streams / ms1 / 1e3:.2f} M frames/s; {parts} batches on {parts} streams {ms:.4f} ms = {streams / ms / 1e3:.2f} M frames/s")
    for b in bs: b.close()

# Round 6: can the pass of one batch RESIDE beside the transform of the other?  Both kernels fill the chip when launched alone (two workgroups per CU each:
# 222 / 235 registers), so two streams serialise them in practice.  With the transform held to ONE persistent workgroup per CU (glv_batch_set_grid) a pass
# workgroup fits beside it -- what a fused kernel with the pass as a phase would amount to.  `live`: the same with GLV_OP_BARS_ONLY.
for live in (0, 1):
    m = mask | (G.OP_BARS_ONLY if live else 0)
    one = G.Batch(p, streams, m)
    base = timed(lambda: one.process_s16(pcm, out, ops))
    one.close()
    sub = streams // 2
    for grid in (0, 256, 384, 512):
        bs = [G.Batch(p, sub, m) for _ in range(2)]
        for b in bs: b.set_grid(grid)
        sts = [torch.cuda.Stream() for _ in range(2)]

        def split2():
            for i, (b, st) in enumerate(zip(bs, sts)):
                b.process_s16(pcm[i * sub:(i + 1) * sub], out[i * sub:(i + 1) * sub], ops, st.cuda_stream)
        ms = timed(split2)
        print(f"sm chain{' (BARS_ONLY)' if live else ''} N={n} x {streams} streams: one batch {base:.4f} ms; two half batches on two streams, transform on {grid or 'its default'} workgroups "
              f"(last grid {bs[0].last_grid()}): {ms:.4f} ms = {streams / ms / 1e3:.2f} M frames/s ({(ms / base - 1) * 100:+.1f} %)")
        for b in bs: b.close()

# The same with the semantics ONE library call would have to keep: everything forked from and joined back into the caller's
# stream inside each call (chunk i's chain on internal stream i % 2), so that nothing of call k is in flight when call k + 1 starts.
main = torch.cuda.current_stream()
for parts, grid, live in [(2, 0, 0), (4, 0, 0), (8, 0, 0), (2, 256, 0), (4, 256, 0), (8, 256, 0), (4, 256, 1), (8, 256, 1), (4, 0, 1)]:
    sub = streams // parts
    bs = [G.Batch(p, sub, mask | (G.OP_BARS_ONLY if live else 0)) for _ in range(parts)]
    for b in bs: b.set_grid(grid)
    sts = [torch.cuda.Stream() for _ in range(2)]
    ev_in = torch.cuda.Event(); ev_out = [torch.cuda.Event() for _ in range(2)]

    def joined():
        ev_in.record(main)
        for st in sts: st.wait_event(ev_in)
        for i, b in enumerate(bs):
            b.process_s16(pcm[i * sub:(i + 1) * sub], out[i * sub:(i + 1) * sub], ops, sts[i % 2].cuda_stream)
        for st, ev in zip(sts, ev_out):
            ev.record(st); main.wait_event(ev)
    ms = timed(joined)
    print(f"sm chain{' (BARS_ONLY)' if live else ''} N={n} x {streams} streams: forked and joined inside every call, {parts} chunks on 2 streams, transform on {grid or 'its default'} workgroups: {ms:.4f} ms = {streams / ms / 1e3:.2f} M frames/s (one batch {ms1:.4f} / BARS_ONLY above)")
    for b in bs: b.close()
