#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: kernel stats + per-dispatch PMC averages per KERNEL (keyed on the full
template argument list: the stateless, stateful and R16 instantiations of glv_frame_kernel differ only in their last
arguments and must not be blended).  Prints plain text (committed under profiles/).

    prof_summary.py <dir> [--traffic-json OUT --n N --streams S --ops OPS --kernel SUBSTRING]

--traffic-json writes the HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section) of the one
kernel whose name contains SUBSTRING, in the format bench.py reads (profiles/hbm_traffic.json)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(out, traffic=None):
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats:", os.path.relpath(f, out))
        for row in csv.DictReader(open(f)):
            name = row.get("Name", "")[:90]
            print(f"  {name:90s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} total_ns={row.get('TotalDurationNs')} pct={row.get('Percentage')}")
    bj = os.path.join(out, "bench_stats.json")
    if os.path.exists(bj):
        print("== bench line under --kernel-trace:", open(bj).read().strip()[:1500])
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "glv_" not in k:
                continue
            agg[k.split("(")[0]][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    for k, cs in agg.items():
        print("== PMC (mean per dispatch):", k)
        for c, v in sorted(cs.items()):
            print(f"  {c:28s} {sum(v) / len(v):18.1f}   (n={len(v)})")
        g = lambda n: (sum(cs[n]) / len(cs[n])) if n in cs else None  # noqa: E731
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if g(n) is not None:
                    print(f"  {n}/SQ_WAVE_CYCLES = {g(n) / wc:.3f}")
        if g("FETCH_SIZE") is not None:
            print(f"  FETCH_SIZE KB={g('FETCH_SIZE'):.0f} (x2 gfx950 correction for wide streams => {2 * g('FETCH_SIZE') * 1024 / 1e9:.3f} GB)")
        if g("WRITE_SIZE") is not None:
            print(f"  WRITE_SIZE KB={g('WRITE_SIZE'):.0f} => {g('WRITE_SIZE') * 1024 / 1e9:.3f} GB")
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            tot = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
            print(f"  HBM traffic per launch (2*FETCH_SIZE + WRITE_SIZE) = {tot:.0f} bytes")
            if traffic and traffic["kernel"] in k:
                import json
                rec = [{"n": traffic["n"], "streams": traffic["streams"], "ops": traffic["ops"], "bytes_per_launch": int(round(tot)),
                        "kernel": k, "fetch_size_kb": g("FETCH_SIZE"), "write_size_kb": g("WRITE_SIZE"),
                        "source": "rocprofv3 --pmc FETCH_SIZE (x2 gfx950 wide-stream correction, MI355X_MICROARCH.md HBM section) and --pmc WRITE_SIZE, "
                                  "separate passes (tools/profile.sh), mean over the dispatches of exactly this kernel under `bench.py --steps 2 --warmup 1`"}]
                json.dump(rec, open(traffic["out"], "w"), indent=1)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--traffic-json", default="")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--ops", default="fft")
    ap.add_argument("--kernel", default="", help="substring of the demangled kernel name; ~ stands for a space (shell-friendly)")
    a = ap.parse_args()
    main(a.dir, {"out": a.traffic_json, "n": a.n, "streams": a.streams, "ops": a.ops, "kernel": a.kernel.replace("~", " ")} if a.traffic_json else None)
