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
# smooth_audio() under a user's smooth_parameters.glsl (the `#define`s of smooth_parameters.glsl:17-42 re-defined, as ~/.config/glava/ would):
# (key, the user's file, glv_params: round_formula, sample_mode, sample_hybrid_weight, sample_scale, sample_range), n, bars, factor, phase, seed
SHAPES = [("maximum", "#define SAMPLE_MODE maximum\n", (0, 1, 0.0, 0.0, 0.0), 1024, 80, 0.025, 0.0, 21),
          ("hybrid", "#define SAMPLE_MODE hybrid\n", (0, 2, 0.0, 0.0, 0.0), 4096, 80, 0.025, 0.0, 22),
          ("hybrid_w40_linear", "#define SAMPLE_MODE hybrid\n#define SAMPLE_HYBRID_WEIGHT 0.4\n#define ROUND_FORMULA linear\n", (2, 2, 0.4, 0.0, 0.0), 2048, 64, 0.05, 0.0, 23),
          ("circular_s6_r80", "#define ROUND_FORMULA circular\n#define SAMPLE_SCALE 6\n#define SAMPLE_RANGE 0.8\n", (1, 0, 0.0, 6.0, 0.8), 4096, 80, 0.025, 0.0, 24),
          ("linear_s4_r95", "#define ROUND_FORMULA linear\n#define SAMPLE_SCALE 4\n#define SAMPLE_RANGE 0.95\n", (2, 0, 0.0, 4.0, 0.95), 1024, 100, 0.01, 0.0, 25),
          ("maximum_circular_pass", "#define SAMPLE_MODE maximum\n#define ROUND_FORMULA circular\n", (1, 1, 0.0, 0.0, 0.0), 512, 512, 0.025, 0.5, 26),
          ("hybrid_pass", "#define SAMPLE_MODE hybrid\n", (0, 2, 0.0, 0.0, 0.0), 512, 512, 0.025, 0.5, 27)]


def tex_row(n, seed):
    """a spectrum-like row in [0, 1]: what the GL_R16 texture holds"""
    rng = np.random.default_rng(seed)
    x = rng.random(n, dtype=np.float32) ** 3 * np.float32(1.3) - np.float32(0.1)
    return np.clip(x, 0, 1).astype(np.float32)


def main():
    out = {}
    for n, bars, factor, seed in BARS:
        out[f"bars_n{n}_b{bars}_f{factor}_s{seed}"] = G.smooth_audio_bars(tex_row(n, seed), bars, factor)
    for key, user, _, n, bars, factor, phase, seed in SHAPES:
        out[f"shape_{key}"] = G.smooth_audio_bars(tex_row(n, seed), bars, factor, user_parameters=user, phase=phase)
    for F, win in AVG:
        frames = [tex_row(256, 100 + F * 10 + i) for i in range(F)]          # index 0 = newest (t0)
        out[f"avg_F{F}_w{win}"] = G.average_pass(frames, bool(win))
    out["gravity_diff0.0487"] = G.gravity_pass(tex_row(256, 7), np.float32(4.2) * (np.float32(1.0) / np.float32(86.1328125)))   # render.c:2225 in float
    np.savez_compressed(os.path.join(HERE, "glsl_vectors.npz"), **out)
    print("wrote", len(out), "vectors")


if __name__ == "__main__":
    main()
