#!/usr/bin/env python3
"""Throughput of the other input layouts of the batched API at N=4096 (fft + magnitude):
planar f32 (the lb/rb snapshot, glava.c:528-537), interleaved stereo f32 (PulseAudio, pulse_input.c:155-178),
s16 FIFO ring updates of 256 frames (fifo.c:81-112).  Algorithmic bytes per frame: f32 16N, ring 8N + 1 KiB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from glava_amd import spectrum as G
from configs_bench import timed
n, streams = 4096, 32768
sync = torch.cuda.synchronize
out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
b = G.Batch(G.Params(n=n), streams, G.OP_FFT | G.OP_RING_S16)
x = torch.rand((streams, 2, n), dtype=torch.float32, device="cuda") - 0.5
dt = timed(lambda: b.process_f32(x, out, G.OP_FFT), sync)
print(f"planar f32      : {dt*1e3:.3f} ms  {streams/dt/1e6:7.2f} M frames/s  {streams/dt*16*n/8e12*100:5.1f} % of 8 TB/s (16N B/frame)")
xs = torch.rand((streams, n, 2), dtype=torch.float32, device="cuda") - 0.5
dt = timed(lambda: b.process_f32_stereo(xs, out, G.OP_FFT), sync)
print(f"interleaved f32 : {dt*1e3:.3f} ms  {streams/dt/1e6:7.2f} M frames/s  {streams/dt*16*n/8e12*100:5.1f} % of 8 TB/s (16N B/frame)")
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
dt = timed(lambda: b.process_s16(pcm, out, G.OP_FFT), sync)
print(f"s16 frames      : {dt*1e3:.3f} ms  {streams/dt/1e6:7.2f} M frames/s  {streams/dt*12*n/8e12*100:5.1f} % of 8 TB/s (12N B/frame)")
new = torch.randint(-32768, 32768, (streams, 256 * 2), dtype=torch.int16, device="cuda")
dt = timed(lambda: b.ring_update_s16(new, 256, out, G.OP_FFT), sync)
print(f"s16 ring update : {dt*1e3:.3f} ms  {streams/dt/1e6:7.2f} M frames/s  {streams/dt*(12*n+2048)/8e12*100:5.1f} % of 8 TB/s (12N + 2 KiB B/frame: append 1 KiB, re-read the ring)")
b.close()
