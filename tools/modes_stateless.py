#!/usr/bin/env python3
"""profiles/r06/modes.txt, appendix: does the STATELESS pass (PCM in, spectra out: two streams in lockstep) have placement speeds too?
K output buffers and K PCM buffers of one process, the headline launch timed for each pairing on the diagonal and first row/column."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
n, streams, K = 4096, 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 6
b = G.Batch(G.Params(n=n), streams, G.OP_FFT)
pcms = [torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda") for _ in range(K)]
outs = [torch.empty((streams, 2, n), dtype=torch.float32, device="cuda") for _ in range(K)]
def ms(p, o, calls=60):
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(8): b.process_s16(p, o, G.OP_FFT)
        torch.cuda.synchronize()
    b.timing_begin()
    for _ in range(calls): b.process_s16(p, o, G.OP_FFT)
    torch.cuda.synchronize()
    k, nl = b.timing_end()
    return k / nl
print("pcm0 x out_k :", "  ".join(f"{ms(pcms[0], o):.4f}" for o in outs))
print("pcm_k x out0 :", "  ".join(f"{ms(p, outs[0]):.4f}" for p in pcms))
print("pcm_k x out_k:", "  ".join(f"{ms(p, o):.4f}" for p, o in zip(pcms, outs)))
