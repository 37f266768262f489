#!/usr/bin/env python3
"""Round-3 experiment batches: builds the A/B tune libraries here (CPU container, `--build`), runs them on the GPU box
(`--run`, inside ONE gpurun call so box-to-box spread cancels) and writes gpurun_out/r03/sweep_<tag>.txt.

    python tools/r03_sweep.py --build --batch xsplit
    gpurun -- 'python tools/r03_sweep.py --run --batch xsplit'
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NOL0 = ["-DGLV_TUNE_NO_LOG0"]
# VW(slots, nbuf, twreg, winlds, occ, prefetch, tiltreg, log_e, wpre, wpre_s); nbuf 0 = split exchange
P13 = "VW(1,1,2,false,2,1,2,5,0,0)"          # N=16384 production
X13 = "VW(2,0,4,false,2,1,2,5,0,0)"          # N=16384: two rows per 512-thread workgroup, split exchange, every twiddle in LDS
BATCHES = {
    "xsplit": dict(
        libs=[("r3a_n13_prod", 13, NOL0, P13),
              ("r3a_n13_prod_nowin", 13, NOL0 + ["-DGLV_EXP_NOWINLOAD"], P13),
              ("r3a_n13_prod_neither", 13, NOL0 + ["-DGLV_EXP_NOWINLOAD", "-DGLV_EXP_NOTWLOAD"], P13),
              ("r3a_n13_x", 13, NOL0, X13),
              ("r3a_n13_x_nowin", 13, NOL0 + ["-DGLV_EXP_NOWINLOAD"], X13)],
        runs=[("r3a_n13_prod", 16384, 0, "N=16384 production", "512"),
              ("r3a_n13_prod_nowin", 16384, 0, "N=16384 production, no window loads (wrong results)", "512"),
              ("r3a_n13_prod_neither", 16384, 0, "N=16384 production, no window and no L2 twiddle loads (wrong results)", "512"),
              ("r3a_n13_x", 16384, 0, "N=16384 split exchange, 2 rows per workgroup, all twiddles in LDS", "256,512"),
              ("r3a_n13_x_nowin", 16384, 0, "N=16384 split exchange ..., no window loads (wrong results)", "256,512"),
              ("r3a_n13_prod", 16384, 0, "N=16384 production (again)", "512")]),
    "groups": dict(
        libs=[("r3b_n13_new", 13, NOL0, P13), ("r3b_n13_old", 13, NOL0 + ["-DGLV_EXP_OLDGROUPS"], P13)],
        runs=[("r3b_n13_old", 16384, 0, "N=16384 round-2 last-pass mapping (4 adjacent groups per lane: 16-byte pieces 32 bytes apart)", "512"),
              ("r3b_n13_new", 16384, 0, "N=16384 paired mapping (a wave instruction covers 1 KiB contiguous)", "512"),
              ("r3b_n13_old", 16384, 0, "N=16384 old (again)", "512"),
              ("r3b_n13_new", 16384, 0, "N=16384 new (again)", "512"),
              ("r3b_n13_old", 8192, 2, "N=16384 x 8192 fft+gravity (state only), old", "512"),
              ("r3b_n13_new", 8192, 2, "N=16384 x 8192 fft+gravity (state only), new", "512")]),
    "tilt": dict(
        libs=[("r3c_n13_t2", 13, NOL0, "VW(1,1,2,false,2,1,2,5,0,0)"), ("r3c_n13_t3", 13, NOL0, "VW(1,1,2,false,2,1,3,5,0,0)"),
              ("r3c_n14_t2", 14, NOL0, "VW(1,1,2,false,2,1,2,5,0,0)"), ("r3c_n14_t3", 14, NOL0, "VW(1,1,2,false,2,1,3,5,0,0)")],
        runs=[("r3c_n13_t2", 16384, 0, "N=16384 tilt evaluated with the reference's operations (TILTREG 2)", "512"),
              ("r3c_n13_t3", 16384, 0, "N=16384 tilt from one fused multiply-add + max per value (TILTREG 3)", "512"),
              ("r3c_n13_t2", 16384, 0, "N=16384 TILTREG 2 (again)", "512"),
              ("r3c_n13_t3", 16384, 0, "N=16384 TILTREG 3 (again)", "512"),
              ("r3c_n14_t2", 8192, 0, "N=32768 TILTREG 2", "256"),
              ("r3c_n14_t3", 8192, 0, "N=32768 TILTREG 3", "256"),
              ("r3c_n13_t2", 8192, 2, "N=16384 x 8192 fft+gravity TILTREG 2", "512"),
              ("r3c_n13_t3", 8192, 2, "N=16384 x 8192 fft+gravity TILTREG 3", "512")]),
    "n8192e32": dict(
        libs=[("r3d_n12_prod", 12, NOL0, "VW(2,1,true,true,2,1,1,4,0,0)"), ("r3d_n12_e32s4", 12, NOL0, "VW(4,1,2,false,2,1,3,5,0,0)"),
              ("r3d_n12_e32s4_t2", 12, NOL0, "VW(4,1,2,false,2,1,2,5,0,0)")],
        runs=[("r3d_n12_prod", 32768, 0, "N=8192 production (E=16, 2 rows x 4 waves, window + twiddles resident)", "256"),
              ("r3d_n12_e32s4", 32768, 0, "N=8192 E=32: 4 rows x 2 waves per workgroup, window and last-pass twiddles through L2, fused tilt", "256,512"),
              ("r3d_n12_e32s4_t2", 32768, 0, "N=8192 E=32 4 rows, exact tilt", "256"),
              ("r3d_n12_prod", 32768, 0, "N=8192 production (again)", "256")]),
    "occ3": dict(
        libs=[("r3e_n11_prod", 11, NOL0, "VW(2,1,true,true,2,1,1,4,0,0)"),
              ("r3e_n11_s6", 11, NOL0, "VW(6,1,4,true,3,1,3,4,0,0)"),
              ("r3e_n11_s6t1", 11, NOL0, "VW(6,1,4,true,3,1,1,4,0,0)")],
        runs=[("r3e_n11_prod", 65536, 0, "N=4096 production (2 rows x 2 waves per workgroup, 2 workgroups per CU, twiddles + tilt resident)", "0"),
              ("r3e_n11_s6", 65536, 0, "N=4096 six rows per 768-thread workgroup, 3 waves per SIMD: twiddles from a 16 KiB LDS table, fused tilt", "256,512"),
              ("r3e_n11_s6t1", 65536, 0, "N=4096 six rows, twiddles from LDS, tilt resident", "256,512"),
              ("r3e_n11_prod", 65536, 256, "N=4096 production, GL_R16 output", "0"),
              ("r3e_n11_s6", 65536, 256, "N=4096 six rows, GL_R16 output", "256,512"),
              ("r3e_n11_s6t1", 65536, 256, "N=4096 six rows tilt resident, GL_R16 output", "256,512"),
              ("r3e_n11_prod", 65536, 0, "N=4096 production (again)", "0")]),
    "notw": dict(
        libs=[("r3f_n13_prod", 13, NOL0, "VW(1,1,2,false,2,1,3,5,0,0)"), ("r3f_n13_notw", 13, NOL0 + ["-DGLV_EXP_NOTWLOAD"], "VW(1,1,2,false,2,1,3,5,0,0)"),
              ("r3f_n13_neither", 13, NOL0 + ["-DGLV_EXP_NOTWLOAD", "-DGLV_EXP_NOWINLOAD"], "VW(1,1,2,false,2,1,3,5,0,0)")],
        runs=[("r3f_n13_prod", 16384, 0, "N=16384 production (folded tilt)", "512"),
              ("r3f_n13_notw", 16384, 0, "N=16384 without the last pass's L2 twiddle gather (wrong results): the prize of resident twiddles", "512"),
              ("r3f_n13_neither", 16384, 0, "N=16384 without twiddle gather and window loads (wrong results)", "512"),
              ("r3f_n13_prod", 16384, 0, "N=16384 production (again)", "512"),
              ("r3f_n13_notw", 16384, 0, "N=16384 no twiddle gather (again)", "512")]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--batch", required=True)
    a = ap.parse_args()
    from glava_amd import build as B
    batch = BATCHES[a.batch]
    if a.build:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=6) as ex:
            for lib in ex.map(lambda l: B.build_tune_variant(l[0], [f"-DGLV_TUNE_LOG_NN={l[1]}"] + l[2], l[3]), batch["libs"]):
                print("built", lib, flush=True)
    if a.run:
        out = os.path.join(ROOT, "gpurun_out", "r03")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, f"sweep_{a.batch}.txt"), "w") as f:
            for lib, streams, extra, label, grids in batch["runs"]:
                f.write(f"== {label}  [{lib}, streams={streams}, extra_ops={extra}]\n"); f.flush()
                subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tune.py"), "--streams", str(streams), "--log-modes", "1",
                                "--lib", os.path.join(ROOT, "tools", "bin", f"libglvtune_{lib}.so"), "--extra-ops", str(extra), "--grids", grids],
                               stdout=f, stderr=subprocess.STDOUT)
                f.flush()
        print(open(os.path.join(out, f"sweep_{a.batch}.txt")).read())


if __name__ == "__main__":
    main()
