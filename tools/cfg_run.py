#!/usr/bin/env python3
"""One configuration, warmed up and launched back to back, for rocprofv3 (tools/profile_cmd.sh):
    cfg_run.py configs2 | configs2_live | chain | n1024bars | gl_default | gl_bars | gl_bars_live | gl_sm | gl_sm64 | gl_sm64_live | gl_sm64[_live]_maximum | gl_sm64_hybrid | gl_sm64[_live]_circular | ring  [calls]
0.3 s of spin-up launches first (the first dozens of launches after idle run ~20 % slower), then `calls` launches (default 150):
the kernel-trace average then describes the warm kernel (the r03 summaries averaged 6 cold calls)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
which = sys.argv[1] if len(sys.argv) > 1 else "configs2"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 150
_shake = torch.empty(int(os.environ.get("CFG_SHAKE_MB", "0")) << 20, dtype=torch.uint8, device="cuda") if os.environ.get("CFG_SHAKE_MB") else None   # (tools/modes.py: moves every later allocation's physical frames)
kw, mask, dt, width = {}, 0, torch.float32, None
if which == "configs2":
    n, streams, ops, bars = 16384, 8192, G.OP_FFT | G.OP_GRAVITY | G.OP_BARS, 80
elif which == "configs2_live":     # ... with GLV_OP_BARS_ONLY (kernel class 8)
    n, streams, ops, bars = 16384, 8192, G.OP_FFT | G.OP_GRAVITY | G.OP_BARS, 80
    mask = G.OP_BARS | G.OP_BARS_ONLY
elif which == "chain":
    n, streams, ops, bars = 4096, 65536, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE, 0
elif which == "n1024bars":
    n, streams, ops, bars = 1024, 131072, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, 80
elif which == "gl_default":
    n, streams, ops, bars = 4096, 65536, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16, 0
    kw, dt = dict(avg_window_kind=1, gl_storage=1), torch.int16
elif which == "gl_bars":
    n, streams, ops, bars = 4096, 65536, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, 80
    kw, dt = dict(avg_window_kind=1, gl_storage=1), torch.int16
elif which == "gl_bars_live":      # ... with GLV_OP_BARS_ONLY (kernel class 9)
    n, streams, ops, bars = 4096, 65536, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, 80
    kw, dt, mask = dict(avg_window_kind=1, gl_storage=1), torch.int16, G.OP_BARS | G.OP_BARS_ONLY
elif which == "gl_sm":
    n, streams, ops, bars = 4096, 16384, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, 4096
    kw, dt, mask = dict(avg_window_kind=1, gl_storage=1, bar_phase=0.5), torch.int16, G.OP_BARS
elif which == "gl_sm64":      # the shipped chain at the batch of bench.py's gl_default.sm_out entry
    n, streams, ops, bars = 4096, 65536, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, 4096
    kw, dt, mask = dict(avg_window_kind=1, gl_storage=1, bar_phase=0.5), torch.int16, G.OP_BARS
elif which == "gl_sm64_live":  # ... with the state kept only where the pass samples (GLV_OP_BARS_ONLY)
    n, streams, ops, bars = 4096, 65536, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, 4096
    kw, dt, mask = dict(avg_window_kind=1, gl_storage=1, bar_phase=0.5), torch.int16, G.OP_BARS | G.OP_BARS_ONLY
elif which in ("gl_sm64_maximum", "gl_sm64_hybrid", "gl_sm64_circular", "gl_sm64_live_maximum", "gl_sm64_live_circular"):
    # the shipped chain under a user's smoothing shape (ABI 7): SAMPLE_MODE maximum / hybrid -> glv_bars_mode_kernel; ROUND_FORMULA circular -> other tap tables, same kernels
    n, streams, ops, bars = 4096, 65536, G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, 4096
    shape = dict(sample_mode=G.SAMPLE_MAXIMUM) if "maximum" in which else dict(sample_mode=G.SAMPLE_HYBRID) if "hybrid" in which else dict(round_formula=G.ROUND_CIRCULAR)
    kw, dt, mask = dict(avg_window_kind=1, gl_storage=1, bar_phase=0.5, **shape), torch.int16, G.OP_BARS | (G.OP_BARS_ONLY if "_live" in which else 0)
elif which == "ring":
    n, streams, ops, bars = 4096, 65536, G.OP_FFT, 0
    mask = G.OP_RING_S16
else:
    raise SystemExit("unknown configuration " + which)
pcm = torch.randint(-32768, 32768, (streams, 256 if which == "ring" else n, 2), dtype=torch.int16, device="cuda")
out = torch.empty((streams, 2, bars if bars else n), dtype=dt, device="cuda")
b = G.Batch(G.Params(n=n, bars=max(bars, 1), **kw), streams, (ops & (G.OP_GRAVITY | G.OP_AVERAGE)) | mask)
call = (lambda: b.ring_update_s16(pcm, 256, out, ops)) if which == "ring" else (lambda: b.process_s16(pcm, out, ops))
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(4): call()
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls): call()
torch.cuda.synchronize()
dt_ = (time.perf_counter() - t0) / calls
print(which, "n", n, "streams", streams, "algorithmic bytes per launch", b.algorithmic_bytes(ops), f"wall ms per call {dt_ * 1e3:.4f} over {calls} calls after spin-up",
      f"=> {b.algorithmic_bytes(ops) / dt_ / 8e12:.4f} of 8 TB/s", "launches per call", b.last_launches())
b.close()
