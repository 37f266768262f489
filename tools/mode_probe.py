#!/usr/bin/env python3
"""The stateful chains' two speeds (profiles/r05/run_to_run.txt) against what the part reports while they run: rocm-smi power, sclk, mclk, fclk sampled beside
SECONDS (default 10) of back-to-back launches, ms per call for every second of the run.    python tools/mode_probe.py [chain|gl] [seconds]
Round 6: run for minutes, the speed changes WITHIN one process (profiles/r06/modes.txt) -- the "modes" are states of the device in time, not of a process."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
n, streams = 4096, 65536
SECS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
chain = len(sys.argv) > 1 and sys.argv[1] == "chain"
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
if chain:
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda"); ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    b = G.Batch(G.Params(n=n), streams, G.OP_GRAVITY | G.OP_AVERAGE)
else:
    out = torch.empty((streams, 2, n), dtype=torch.int16, device="cuda"); ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16
    b = G.Batch(G.Params(n=n, avg_window_kind=1, gl_storage=1), streams, G.OP_GRAVITY | G.OP_AVERAGE)
def sample():
    o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
    g = lambda pat: (re.search(pat, o) or [None, "?"])[1]
    return "P %s W  sclk %s  mclk %s  fclk %s  socclk %s  Tjunc %s  Tmem %s" % (g(r"Package Power \(W\):\s*([0-9.]+)"), g(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)"), g(r"mclk clock level:\s*\d+:?\s*\((\d+)Mhz\)"),
            g(r"fclk clock level:\s*\d+:?\s*\((\d+)Mhz\)"), g(r"socclk clock level:\s*\S+:?\s*\((\d+)Mhz\)"), g(r"\(Sensor junction\) \(C\):\s*([0-9.]+)"), g(r"\(Sensor (?:memory|HBM 0)\) \(C\):\s*([0-9.]+)"))
print("idle:", sample(), flush=True)
stop = False
def watch():
    while not stop:
        print("   smi:", sample(), flush=True); time.sleep(5.0 if SECS > 30 else 1.0)
th = threading.Thread(target=watch); th.start()
for sec in range(SECS):
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 1.0:
        for _ in range(32): b.process_s16(pcm, out, ops)
        torch.cuda.synchronize(); k += 32
    print(f"second {sec}: {(time.perf_counter() - t0) / k * 1e3:.4f} ms per call", flush=True)
stop = True; th.join()
