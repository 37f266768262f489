#!/usr/bin/env python3
"""All single-GPU/CPU configurations of BASELINE.json `configs` in one table.

    python tools/configs_bench.py [--out profiles/r01/configs.txt]

  [0] single 22050 Hz stereo stream, N=1024, bars-module chain (window,fft,gravity,avg), reference CPU path
      -> the compiled reference (oracle/_ref) on ONE host core, no GPU
  [1] 1 MI355X, 64K stereo streams, N=4096, window+FFT+magnitude              (== bench.py)
  [2] 1 MI355X, N=16384, FFT + gravity + radial-module bin averaging (80 bars/channel), 8192 streams
  [3] 8 GPUs -- the driver's job (bench.py --gpus 8); not run here
  [4] mixed N in {512,1024,2048,4096,8192}, equal bytes per size class, one HIP stream per class, all
      classes in flight together; per-class frames/s
Every GPU number is kernel-inclusive wall time between synchronisations after a clock spin-up; PCM is
resident in HBM.  Roofline fractions use the algorithmic bytes of SURVEY.md 8d and 8 TB/s.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cpu_config0(seconds=4.0):
    import numpy as np
    from oracle_lib import Ref
    if not Ref.available():
        return "config[0]: oracle/_ref not available"
    n = 1024
    p = Ref.params(avg_frames=5, avg_window=True)
    pcm = np.random.default_rng(1).integers(-32768, 32768, 256 * 2 * n, dtype=np.int16)
    Ref.lib().glvref_bench_frames(C.byref(p), pcm, 256, n, 1)
    t0 = time.perf_counter(); frames = 0
    while time.perf_counter() - t0 < seconds:
        Ref.lib().glvref_bench_frames(C.byref(p), pcm, 256, n, 1)
        frames += 256
    dt = time.perf_counter() - t0
    return (f"config[0] reference CPU path, N=1024, fft+gravity+average(F=5,windowed), 1 host core: "
            f"{frames / dt:,.0f} stereo frames/s  (real-time need: 86 updates/s)")


def timed(fn, sync, steps=20, spin=0.3):
    t_end = time.perf_counter() + spin
    while time.perf_counter() < t_end:
        fn(); sync()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    from glava_amd import build as B, spectrum as G
    B.build()
    lines = [cpu_config0()]
    sync = torch.cuda.synchronize
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)

    # [1]
    n, streams = 4096, 65536
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda", generator=gen)
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    for lm in (0, 1):
        b = G.Batch(G.Params(n=n, log_mode=lm), streams, G.OP_FFT)
        dt = timed(lambda: b.process_s16(pcm, out, G.OP_FFT), sync)
        fps = streams / dt
        lines.append(f"config[1] N=4096 x {streams} streams, window+FFT+magnitude, log_mode {lm}: {fps / 1e6:7.2f} M frames/s, "
                     f"{fps * 12 * n / 8e12 * 100:5.1f} % of 8 TB/s")
        b.close()
    b = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    out16 = torch.empty((streams, 2, n), dtype=torch.int16, device="cuda")
    dt = timed(lambda: b.process_s16(pcm, out16, G.OP_FFT | G.OP_R16), sync)
    lines.append(f"config[1'] same pass, output as GL_R16 texels (GLV_OP_R16, what handle_audio uploads): {streams / dt / 1e6:7.2f} M frames/s, "
                 f"{streams / dt * 8 * n / 8e12 * 100:5.1f} % of 8 TB/s (8N B/frame)")
    b.close()
    del pcm, out, out16
    # the larger windows at equal bytes (the sizes VERDICT r1 singles out)
    for n2, s2 in ((8192, 32768), (16384, 16384), (32768, 8192)):
        pcm2 = torch.randint(-32768, 32768, (s2, n2, 2), dtype=torch.int16, device="cuda", generator=gen)
        out2 = torch.empty((s2, 2, n2), dtype=torch.float32, device="cuda")
        b = G.Batch(G.Params(n=n2), s2, G.OP_FFT)
        dt = timed(lambda: b.process_s16(pcm2, out2, G.OP_FFT), sync)
        lines.append(f"          N={n2:5d} x {s2:5d} streams, window+FFT+magnitude: {s2 / dt / 1e6:7.2f} M frames/s, {s2 / dt * 12 * n2 / 8e12 * 100:5.1f} % of 8 TB/s ({dt * 1e3:.3f} ms)")
        b.close()
        del pcm2, out2

    # [2]
    n, streams, bars = 16384, 8192, 80
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda", generator=gen)
    dbars = torch.empty((streams, 2, bars), dtype=torch.float32, device="cuda")
    spec = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n, bars=bars), streams, G.OP_FFT | G.OP_GRAVITY)
    ops = G.OP_FFT | G.OP_GRAVITY
    dt = timed(lambda: b.process_s16(pcm, spec, ops), sync)
    lines.append(f"config[2a] N=16384 x {streams} streams, FFT+gravity (full spectra out): {streams / dt / 1e6:6.2f} M frames/s, "
                 f"{streams / dt * 20 * n / 8e12 * 100:5.1f} % of 8 TB/s (20N B/frame algorithmic, 28N moved: state and output are both written)")
    dt = timed(lambda: b.process_s16(pcm, None, ops), sync)
    lines.append(f"config[2a'] same, spectra left in the gravity state (d_out = NULL, output == state): {streams / dt / 1e6:6.2f} M frames/s, "
                 f"{streams / dt * 20 * n / 8e12 * 100:5.1f} % of 8 TB/s (20N B/frame, the traffic actually moved)")
    dt = timed(lambda: b.process_s16(pcm, dbars, ops | G.OP_BARS), sync)
    lines.append(f"config[2b] N=16384 x {streams} streams, FFT+gravity+radial bin averaging -> {bars} bars/channel: "
                 f"{streams / dt / 1e6:6.2f} M frames/s, {streams / dt * (20 * n + 640) / 8e12 * 100:5.1f} % of 8 TB/s "
                 f"(20N+640 B/frame, the traffic actually moved: bars are computed in the frame kernel from the row in LDS)")
    b.close()
    del pcm, dbars, spec

    # [4] mixed sizes, equal bytes per class, concurrent streams
    classes = []
    for n in (512, 1024, 2048, 4096, 8192):
        s = 16384 * 4096 // n
        pcm = torch.randint(-32768, 32768, (s, n, 2), dtype=torch.int16, device="cuda", generator=gen)
        out = torch.empty((s, 2, n), dtype=torch.float32, device="cuda")
        classes.append((n, s, pcm, out, G.Batch(G.Params(n=n), s, G.OP_FFT), torch.cuda.Stream()))

    def launch_all():
        for n, s, pcm, out, b, st in classes:
            b.process_s16(pcm, out, G.OP_FFT, st.cuda_stream)
    dt = timed(launch_all, sync)
    tot_bytes = sum(s * 12 * n for n, s, *_ in classes)
    lines.append(f"config[4] mixed N, 5 classes x 256 MiB PCM each on 5 HIP streams: {dt * 1e3:.3f} ms per round of all classes, "
                 f"{tot_bytes / dt / 8e12 * 100:5.1f} % of 8 TB/s aggregate")
    for n, s, pcm, out, b, st in classes:
        dt1 = timed(lambda: b.process_s16(pcm, out, G.OP_FFT, st.cuda_stream), sync, steps=10, spin=0.1)
        lines.append(f"          class N={n:5d} x {s:6d} streams alone: {s / dt1 / 1e6:8.2f} M frames/s, {s / dt1 * 12 * n / 8e12 * 100:5.1f} % of 8 TB/s")
        b.close()

    text = "\n".join(lines)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("# tools/configs_bench.py on one MI355X (see the module docstring for the configurations)\n" + text + "\n")


if __name__ == "__main__":
    main()
