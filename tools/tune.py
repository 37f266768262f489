#!/usr/bin/env python3
"""Knob sweep of glv_frame_kernel (libglvtune.so) on a real MI355X.

    python tools/tune.py [--streams 65536] [--iters 5] [--out gpurun_out/tune.txt]

For every variant compiled into glava_amd/csrc/glv_tune.hip: average kernel time over
`iters` launches (HIP events), frames/s, fraction of the 8 TB/s HBM roofline for the
algorithmic 12*N bytes per stereo frame, and a bitwise comparison of its output with the
production library's output for the same PCM.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--grids", default="0", help="comma list of grid sizes (0 = auto)")
    ap.add_argument("--log-modes", default="0,1")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    from glava_amd import build as B, spectrum as G
    B.build(tune=True)
    T = C.CDLL(os.path.join(ROOT, "glava_amd", "csrc", "libglvtune.so"))
    T.glv_tune_describe.restype = C.c_char_p
    T.glv_tune_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p,
                               C.POINTER(C.c_float)]
    n = 2 << T.glv_tune_log_nn()
    streams = a.streams
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    d_pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda", generator=g)
    d_ref = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    d_out = torch.empty_like(d_ref)
    lines = []
    for lm in [int(x) for x in a.log_modes.split(",")]:
        b = G.Batch(G.Params(n=n, log_mode=lm), streams, G.OP_FFT)
        b.process_s16(d_pcm, d_ref, G.OP_FFT)
        torch.cuda.synchronize()
        b.close()
        for grid in [int(x) for x in a.grids.split(",")]:
            for i in range(T.glv_tune_count()):
                ms = C.c_float(0)
                d_out.fill_(float("nan"))
                rc = T.glv_tune_run(i, d_pcm.data_ptr(), d_out.data_ptr(), streams, lm, grid, a.iters, None, C.byref(ms))
                torch.cuda.synchronize()
                if rc != 0:
                    lines.append(f"log={lm} grid={grid} {T.glv_tune_describe(i).decode():48s} FAILED rc={rc}")
                    continue
                same = bool(torch.equal(d_out.view(torch.int32), d_ref.view(torch.int32)))
                fps = streams / (ms.value * 1e-3)
                frac = fps * 12 * n / 8e12
                lines.append(f"log={lm} grid={grid:5d} {T.glv_tune_describe(i).decode():48s} {ms.value:9.3f} ms  "
                             f"{fps / 1e6:8.2f} Mframes/s  {100 * frac:5.1f}% of 8TB/s  bits_equal_prod={same}")
                print(lines[-1], flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
