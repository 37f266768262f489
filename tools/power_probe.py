#!/usr/bin/env python3
"""Package power and shader clock while one pass runs back to back (rocm-smi samples beside a launch loop).
    python tools/power_probe.py [--seconds 6]   ->  one line per pass: kernel ms, median power (W), median sclk (MHz)"""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sample():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([0-9.]+)", out)
    c = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(c.group(1)) if c else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=0, help="default: 65536 * 4096 / n (equal bytes)")
    ap.add_argument("--only", default="", help="comma list of pass names to run: f32,r16,strict (default all)")
    a = ap.parse_args()
    import statistics
    import torch
    from glava_amd import build as B, spectrum as G
    B.build()
    n = a.n
    streams = a.streams or 65536 * 4096 // n
    only = set(a.only.split(",")) if a.only else None
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    print(f"N={n} x {streams} streams, library {os.environ.get('GLV_SPECTRUM_LIB', 'product')}; idle:", sample())
    for key, name, ops, lm in (("f32", "window+FFT+magnitude -> f32", G.OP_FFT, 1), ("r16", "... -> GL_R16 texels", G.OP_FFT | G.OP_R16, 1), ("strict", "bit-faithful log -> f32", G.OP_FFT, 0)):
        if only and key not in only: continue
        b = G.Batch(G.Params(n=n, log_mode=lm), streams, G.OP_FFT)
        stop = False
        samples = []

        def watch():
            while not stop:
                samples.append(sample())
                time.sleep(0.25)
        t_end = time.perf_counter() + 1.0
        while time.perf_counter() < t_end:
            for _ in range(16): b.process_s16(pcm, out, ops)
            torch.cuda.synchronize()
        th = threading.Thread(target=watch); th.start()
        b.timing_begin()
        t_end = time.perf_counter() + a.seconds
        while time.perf_counter() < t_end:
            for _ in range(64): b.process_s16(pcm, out, ops)
            torch.cuda.synchronize()
        ms, nl = b.timing_end()
        stop = True; th.join()
        pw = [s[0] for s in samples if s[0] is not None]; ck = [s[1] for s in samples if s[1] is not None]
        print(f"{name:32s} {ms / nl:.3f} ms/launch over {nl} launches; power median {statistics.median(pw) if pw else None} W "
              f"(max {max(pw) if pw else None}); sclk median {statistics.median(ck) if ck else None} MHz (min {min(ck) if ck else None}); {len(samples)} samples")
        b.close()


if __name__ == "__main__":
    main()
