#!/usr/bin/env python3
"""Round-2 experiment batch: builds the A/B tune libraries here (CPU container, `--build`), runs them on the
GPU box (`--run`, inside ONE gpurun call so box-to-box spread cancels) and writes gpurun_out/r02/*.txt.

    python tools/r02_sweep.py --build
    gpurun -- 'python tools/r02_sweep.py --run'
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NOL0 = ["-DGLV_TUNE_NO_LOG0"]
# name, log_nn, extra flags, variant list (glv_tune.hip macros)
LIBS = [
    ("r2s_n12_p0", 12, NOL0, "VW(2,1,true,true,2,1,1,4,0,0)"),
    ("r2s_n12_p3", 12, NOL0 + ["-DGLV_EXP_STOREPRIO=3"], "VW(2,1,true,true,2,1,1,4,0,0)"),
    ("r2s_n13_p0", 13, NOL0, "VW(1,1,2,false,2,1,2,5,0,0)"),
    ("r2s_n13_p3", 13, NOL0 + ["-DGLV_EXP_STOREPRIO=3"], "VW(1,1,2,false,2,1,2,5,0,0)"),
]
RUNS = [
    ("r2s_n12_p0", 32768, 0, "N=8192 default priority"),
    ("r2s_n12_p3", 32768, 0, "N=8192 epilogue at s_setprio 3"),
    ("r2s_n12_p0", 32768, 0, "N=8192 default priority (again)"),
    ("r2s_n12_p3", 32768, 0, "N=8192 epilogue at s_setprio 3 (again)"),
    ("r2s_n13_p0", 16384, 0, "N=16384 default priority"),
    ("r2s_n13_p3", 16384, 0, "N=16384 epilogue at s_setprio 3"),
    ("r2s_n13_p0", 16384, 0, "N=16384 default priority (again)"),
    ("r2s_n13_p3", 16384, 0, "N=16384 epilogue at s_setprio 3 (again)"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    from glava_amd import build as B
    if a.build:
        from concurrent.futures import ThreadPoolExecutor
        libs = [l for l in LIBS if not a.only or l[0] in a.only.split(",")]
        with ThreadPoolExecutor(max_workers=6) as ex:
            for lib in ex.map(lambda l: B.build_tune_variant(l[0], [f"-DGLV_TUNE_LOG_NN={l[1]}"] + l[2], l[3]), libs):
                print("built", lib, flush=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", os.path.join(ROOT, "tools", "membench2.hip"),
                        "-o", os.path.join(ROOT, "tools", "bin", "membench2")], check=True)
    if a.run:
        out = os.path.join(ROOT, "gpurun_out", "r02")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "membench2.txt"), "w") as f:
            subprocess.run([os.path.join(ROOT, "tools", "bin", "membench2")], stdout=f, stderr=subprocess.STDOUT)
        with open(os.path.join(out, "sweep.txt"), "w") as f:
            for run in RUNS:
                lib, streams, extra, label = run[:4]
                env = dict(os.environ); env.update(run[4] if len(run) > 4 else {})
                if a.only and lib not in a.only.split(","):
                    continue
                f.write(f"== {label}  [{lib}, streams={streams}, extra_ops={extra}]\n"); f.flush()
                subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tune.py"), "--streams", str(streams), "--log-modes", "0,1" if "_l0" in lib else "1",
                                "--lib", os.path.join(ROOT, "glava_amd", "csrc", f"libglvtune_{lib}.so"), "--extra-ops", str(extra)] + (["--grids", "256"] if "n12" in lib else []),
                               stdout=f, stderr=subprocess.STDOUT, env=env)
                f.flush()


if __name__ == "__main__":
    main()
