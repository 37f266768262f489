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
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--grids", default="0", help="comma list of grid sizes (0 = auto)")
    ap.add_argument("--log-modes", default="0,1")
    ap.add_argument("--out", default="")
    ap.add_argument("--lib", default="", help="alternative libglvtune_*.so (A/B experiments)")
    ap.add_argument("--spinup-s", type=float, default=0.5)
    ap.add_argument("--extra-ops", type=int, default=0, help="ops OR'ed into GLV_OP_FFT (256 = GLV_OP_R16, 2 = GLV_OP_GRAVITY: timing only)")
    ap.add_argument("--bytes-per-frame-n", type=float, default=0.0, help="algorithmic bytes per frame in units of N (default 12; 8 with R16, 20 with gravity state-only)")
    a = ap.parse_args()
    import torch
    from glava_amd import build as B, spectrum as G
    B.build(tune=not a.lib)
    T = C.CDLL(a.lib if a.lib else os.path.join(ROOT, "glava_amd", "csrc", "libglvtune.so"))
    T.glv_tune_describe.restype = C.c_char_p
    T.glv_tune_run3.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.POINTER(C.c_float), C.c_uint, C.c_void_p, C.c_void_p]
    n = 2 << T.glv_tune_log_nn()
    streams = a.streams
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    d_pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda", generator=g)
    d_ref = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    d_out = torch.empty_like(d_ref)
    gravity = bool(a.extra_ops & (G.OP_GRAVITY | G.OP_AVERAGE))
    average = bool(a.extra_ops & G.OP_AVERAGE)
    d_hist = torch.zeros((streams, 2, 5, n), dtype=torch.float32, device="cuda") if average else None
    d_grav = torch.zeros((streams, 2, n), dtype=torch.float32, device="cuda") if gravity else None
    bpf = a.bytes_per_frame_n if a.bytes_per_frame_n else (8.0 if a.extra_ops & G.OP_R16 else 52.0 if average else 20.0 if gravity else 12.0)
    lines = []
    import statistics
    grids = [int(x) for x in a.grids.split(",")]
    for lm in [int(x) for x in a.log_modes.split(",")]:
        d_ref.zero_()
        if not gravity:       # stateful chains are checked by the parity suite; here they are timed only
            b = G.Batch(G.Params(n=n, log_mode=lm), streams, G.OP_FFT)
            b.process_s16(d_pcm, d_ref, G.OP_FFT | a.extra_ops)
            torch.cuda.synchronize()
            b.close()
        cases = [(grid, i) for grid in grids for i in range(T.glv_tune_count())]
        # clock spin-up: the first dozens of launches after idle run ~20 % slower on MI355X (bench.py does the same)
        import time
        t_end = time.perf_counter() + a.spinup_s
        ms0 = C.c_float(0)
        while time.perf_counter() < t_end:
            T.glv_tune_run3(0, d_pcm.data_ptr(), d_out.data_ptr() if (average or not gravity) else None, streams, lm, grids[0], 8, None, C.byref(ms0),
                            a.extra_ops, d_grav.data_ptr() if gravity else None, d_hist.data_ptr() if average else None)
        times = {c: [] for c in cases}
        same = {}
        # round-robin over the variants, `reps` times, so clock/thermal drift hits all of them alike
        for rep in range(a.reps):
            for (grid, i) in cases:
                ms = C.c_float(0)
                if rep == 0:
                    d_out.zero_()
                rc = T.glv_tune_run3(i, d_pcm.data_ptr(), d_out.data_ptr() if (average or not gravity) else None, streams, lm, grid, a.iters, None, C.byref(ms),
                                     a.extra_ops, d_grav.data_ptr() if gravity else None, d_hist.data_ptr() if average else None)
                torch.cuda.synchronize()
                if rc != 0:
                    times[(grid, i)].append(float("inf"))
                    continue
                times[(grid, i)].append(ms.value)
                if rep == 0 and not gravity:
                    same[(grid, i)] = bool(torch.equal(d_out.view(torch.int32), d_ref.view(torch.int32)))
        for (grid, i) in cases:
            t = times[(grid, i)]
            med, best = statistics.median(t), min(t)
            fps = streams / (med * 1e-3)
            frac = fps * bpf * n / 8e12
            lines.append(f"log={lm} grid={grid:5d} {T.glv_tune_describe(i).decode():44s} median {med:7.3f} ms (min {best:7.3f})  "
                         f"{fps / 1e6:7.2f} Mframes/s  {100 * frac:5.1f}% of 8TB/s  bits_equal_prod={same.get((grid, i))}")
            print(lines[-1], flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
