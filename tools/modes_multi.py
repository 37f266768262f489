#!/usr/bin/env python3
"""profiles/r06/modes.txt: is the speed of a stateful chain a property of WHERE ITS STATE LIES?  K batches of the same chain created one after the other and
kept alive together (K different sets of physical frames), one PCM / output buffer for all: wall ms per call of each, twice round.    python tools/modes_multi.py [chain|gl] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
which = sys.argv[1] if len(sys.argv) > 1 else "chain"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n, streams = 4096, 65536
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
if which == "chain":
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda"); ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    mk = lambda: G.Batch(G.Params(n=n), streams, G.OP_GRAVITY | G.OP_AVERAGE)
else:
    out = torch.empty((streams, 2, n), dtype=torch.int16, device="cuda"); ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16
    mk = lambda: G.Batch(G.Params(n=n, avg_window_kind=1, gl_storage=1), streams, G.OP_GRAVITY | G.OP_AVERAGE)
bs = [mk() for _ in range(K)]
def ms(b, calls=40):
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(4): b.process_s16(pcm, out, ops)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls): b.process_s16(pcm, out, ops)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / calls * 1e3
for rnd in range(2):
    print(f"{which} round {rnd}: " + "  ".join(f"{ms(b):.4f}" for b in bs), flush=True)
# the same batch against OTHER input / output buffers
pcm2 = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda"); out2 = torch.empty_like(out)
pcm_, out_ = pcm, out
res = []
for (p_, o_) in ((pcm, out), (pcm2, out), (pcm, out2), (pcm2, out2)):
    pcm, out = p_, o_
    res.append(ms(bs[0]))
print(f"{which} batch 0 with (pcm, out) / (pcm2, out) / (pcm, out2) / (pcm2, out2): " + "  ".join(f"{x:.4f}" for x in res))
