#!/usr/bin/env python3
"""One GLava instance (one stereo stream, N = 4096): the default pipeline -- s16 PCM -> upload -> GL_MAX + gravity -> ring -> average ->
pre-smoothing pass -> `sm` texels -- per update, eager calls against one replayed hipGraph of F updates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from glava_amd import spectrum as G
n, F = 4096, 5
p = G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)
mask = G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS
ops = G.OP_FFT | mask | G.OP_R16
b = G.Batch(p, 1, mask)
pcm = torch.randint(-8000, 8000, (1, n, 2), dtype=torch.int16, device="cuda")
out = torch.zeros((2, n), dtype=torch.int16, device="cuda")
for _ in range(50): b.process_s16(pcm, out, ops)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): b.process_s16(pcm, out, ops)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 2000
t0 = time.perf_counter()
for _ in range(500):
    b.process_s16(pcm, out, ops); torch.cuda.synchronize()
sync = (time.perf_counter() - t0) / 500
print(f"one stream, N={n}, default pipeline with the pre-smoothing pass ({b.last_launches()} launches): {eager * 1e6:.1f} us per update back to back, "
      f"{sync * 1e6:.1f} us per update with a synchronize after each (launch + completion latency)")
b.close()
# ... and what the patched GLava host pays per rendered frame: integration/render_hip.patch makes ONE glv_gl_texture call per bind (two binds:
# audio_l, audio_r) -- host samples in, the module's GL_R16 texels out, synchronous (pinned mapped staging, one stream synchronise per call)
sts = [G.State(p), G.State(p)]
x = (np.random.default_rng(1).integers(-8000, 8000, (2, n)).astype(np.float32) / np.float32(65535))
tex = np.zeros((2, n), np.uint16)
for _ in range(50):
    for c in range(2): sts[c].gl_texture(x[c], tex[c], True)
t0 = time.perf_counter()
for _ in range(1000):
    for c in range(2): sts[c].gl_texture(x[c], tex[c], True)
host = (time.perf_counter() - t0) / 1000
print(f"the same through the host drop-in (glv_gl_texture, two binds per update, host buffers in and out): {host * 1e6:.1f} us per update "
      f"(GLava has 11.6 ms per update at 86 updates/s; its CPU + GL passes are replaced by this)")
for s_ in sts: s_.close()
