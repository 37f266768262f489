#!/usr/bin/env python3
"""Golden vectors for the GL twins, produced by evaluating the reference's SHADER TEXT with
tests/glsl_eval.py (an interpreter written for this purpose; it shares nothing with the oracle or the kernels):

    shaders/glava/util/smooth.glsl        smooth_audio() at the radial / bars sampling positions k / bars
    shaders/glava/util/average_pass.frag  the windowed frame average (t0 = newest), F = 2, 3, 5, 6, window on / off
    shaders/glava/util/gravity_pass.frag  store - diff

Needs /root/reference (run in the build container); writes tests/golden/glsl_vectors.npz, which travels to the GPU box.
Inputs are regenerated from the seeds below by the tests.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import glsl_eval as G  # noqa: E402

BARS = [(512, 80, 0.025, 11), (1024, 16, 0.025, 12), (4096, 80, 0.025, 13), (4096, 31, 0.05, 14), (16384, 80, 0.025, 15), (8192, 200, 0.01, 16)]
AVG = [(2, 1), (3, 1), (5, 1), (5, 0), (6, 1)]


def tex_row(n, seed):
    """a spectrum-like row in [0, 1]: what the GL_R16 texture holds"""
    rng = np.random.default_rng(seed)
    x = rng.random(n, dtype=np.float32) ** 3 * np.float32(1.3) - np.float32(0.1)
    return np.clip(x, 0, 1).astype(np.float32)


def main():
    out = {}
    for n, bars, factor, seed in BARS:
        out[f"bars_n{n}_b{bars}_f{factor}_s{seed}"] = G.smooth_audio_bars(tex_row(n, seed), bars, factor)
    for F, win in AVG:
        frames = [tex_row(256, 100 + F * 10 + i) for i in range(F)]          # index 0 = newest (t0)
        out[f"avg_F{F}_w{win}"] = G.average_pass(frames, bool(win))
    out["gravity_diff0.0487"] = G.gravity_pass(tex_row(256, 7), np.float32(4.2) * (np.float32(1.0) / np.float32(86.1328125)))   # render.c:2225 in float
    np.savez_compressed(os.path.join(HERE, "glsl_vectors.npz"), **out)
    print("wrote", len(out), "vectors")


if __name__ == "__main__":
    main()
