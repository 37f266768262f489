#!/usr/bin/env python3
"""Register / scratch / instruction-mix summary of the kernels in a gfx950 assembly file (hipcc -save-temps *.s).

    python tools/isa_stats.py file.s [--loop]     # --loop: histogram of the longest backward-branch loop body of each kernel
"""
import collections
import re
import subprocess
import sys


def demangle(name):
    for tool in ("c++filt", "/opt/rocm/llvm/bin/llvm-cxxfilt"):
        try:
            return subprocess.run([tool, name], capture_output=True, text=True).stdout.strip() or name
        except FileNotFoundError:
            continue
    return name


def classify(op):
    if op.startswith("global_load"): return "gload"
    if op.startswith("global_store"): return "gstore"
    if op.startswith("scratch"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    s = open(sys.argv[1]).read()
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        b = m.group(2)
        g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", b) or [None, "?"])[1]    # noqa: E731
        meta[m.group(1)] = dict(vgpr=g("next_free_vgpr"), accum=g("accum_offset"), scratch=g("private_segment_fixed_size"))
    for name, info in meta.items():
        i0 = s.find("\n" + name + ":")
        if i0 < 0:
            continue
        body = s[i0:s.find("s_endpgm", i0)]
        lines = [l.strip() for l in body.split("\n")]
        ins = [(i, l.split()[0]) for i, l in enumerate(lines) if l and not l.startswith((".", ";", "//")) and not l.endswith(":")]
        hist = collections.Counter(classify(op) for _, op in ins)
        detail = collections.Counter(op for _, op in ins if classify(op) in ("gload", "gstore", "scratch"))
        dm = demangle(name)
        dm = re.sub(r"glv::glv_frame_kernel", "frame", dm)
        print(f"{dm[:110]}\n    vgpr {info['vgpr']} (accum_offset {info['accum']}) scratch {info['scratch']} B   static: {dict(hist)}\n    {dict(detail)}")


if __name__ == "__main__":
    main()
