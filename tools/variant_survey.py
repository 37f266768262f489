#!/usr/bin/env python3
"""Which kernel configuration (glv_inst.hip Tuned<K, V>) wins where: every variant of every size that has more than one,
for the stateless pass, the GL_R16 output, the bit-faithful log and the fft -> gravity -> average chain, at equal bytes per
size; then what glv_batch_autotune records.  Run on the GPU box:  python tools/variant_survey.py [--out file]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--wisdom", default="")
    a = ap.parse_args()
    import torch
    from glava_amd import build as B, spectrum as G
    B.build()
    lines = []
    cases = [("fft", G.OP_FFT, G.OP_FFT, 1, 12, {}), ("fft->R16", G.OP_FFT | G.OP_R16, G.OP_FFT, 1, 8, {}), ("fft log_mode 0", G.OP_FFT, G.OP_FFT, 0, 12, {}),
             ("fft->gravity->average", G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE, G.OP_GRAVITY | G.OP_AVERAGE, 1, 52, {}),
             ("GL_R16 chain (gl_storage 1)", G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16, G.OP_GRAVITY | G.OP_AVERAGE, 1, 28, dict(gl_storage=1, avg_window_kind=1))]
    for n in (512, 1024, 2048, 4096, 8192, 16384, 32768):
        streams = 65536 * 4096 // n
        pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
        out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
        for name, ops, mask, lm, bpn, kw in cases:
            if mask != G.OP_FFT and n > 4096:
                s2 = streams // 4                              # the history ring of the chain: keep it at a few GiB
            else:
                s2 = streams
            b = G.Batch(G.Params(n=n, log_mode=lm, **kw), s2, mask)
            res = []
            for v in range(b.variants()):
                b.set_variant(v)
                t_end = time.perf_counter() + 0.2
                while time.perf_counter() < t_end:
                    for _ in range(4): b.process_s16(pcm, out, ops)
                    torch.cuda.synchronize()
                b.timing_begin()
                for _ in range(10): b.process_s16(pcm, out, ops)
                torch.cuda.synchronize()
                ms, nl = b.timing_end()
                k = ms / nl
                res.append((v, k, s2 * bpn * n / (k * 1e-3) / 8e12))
            b.set_variant(-1)
            best = min(res, key=lambda r: r[1])
            lines.append(f"N={n:5d} x {s2:6d}  {name:24s} " + "  ".join(f"v{v}: {k:7.3f} ms {100 * f:5.1f}%" for v, k, f in res) + f"   -> v{best[0]}")
            print(lines[-1], flush=True)
            if a.wisdom:
                g, ms = b.autotune(pcm, out, ops)
                b.process_s16(pcm, out, ops)
                lines.append(f"        autotune: variant {b.last_variant()} ({b.describe_variant(b.last_variant())}), {g} workgroups, {ms:.3f} ms")
                print(lines[-1], flush=True)
            b.close()
        del pcm, out
    if a.wisdom:
        G.wisdom_save(a.wisdom)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("# tools/variant_survey.py on one MI355X: kernel time and fraction of 8 TB/s per kernel configuration\n" + "\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
