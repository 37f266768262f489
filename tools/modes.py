#!/usr/bin/env python3
"""what the two (three) speeds of the stateful chains are.  Runs ON the GPU box:

    python tools/modes.py counters [cfg] [procs]   # per process: rocprofv3 --pmc <set> around tools/cfg_run.py <cfg>; the frame kernel's mean duration
                                                   # (dispatch timestamps of the same csv) next to the set's counters -- which counter moves with the speed?
    python tools/modes.py alloc [cfg] [procs]      # GLV_STATE_ALLOC = malloc | vmm:0 | vmm:2 | vmm:1024 | fine | uncached, `procs` processes each: wall ms per call

Output: plain text for profiles/r06/modes.txt."""
import csv, glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = [                       # (every rocprofv3 run is bounded by `timeout`: the first version of this script hung in its first --pmc run for 28 minutes)
    "TCC_HIT TCC_MISS TCC_TAG_STALL TCC_EA0_RDREQ",
    "TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL",
    "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL",
    "TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS",
    "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY",
]


def wall_ms(cfg, env=None, calls=60):
    e = dict(os.environ); e.update(env or {})
    o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cfg_run.py"), cfg, str(calls)], capture_output=True, text=True, env=e).stdout
    m = re.search(r"wall ms per call ([0-9.]+)", o)
    return float(m.group(1)) if m else float("nan")


def counters(cfg, procs):
    sets = [os.environ["GLV_MODES_SET"]] if os.environ.get("GLV_MODES_SET") else SETS
    for si, cs in enumerate(sets):
        for p in range(procs):
            d = tempfile.mkdtemp(prefix="modes_", dir="/tmp")
            cmd = ["timeout", "90", "rocprofv3", "--pmc", *cs.split(), "--output-format", "csv", "-d", d, "-o", "m", "--", sys.executable, os.path.join(ROOT, "tools", "cfg_run.py"), cfg, "40"]
            shake = str((p * 797) % 3000)                           # a different amount of memory taken first: the state lands on other physical frames
            r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", CFG_SHAKE_MB=shake))
            wm = re.search(r"wall ms per call ([0-9.]+)", r.stdout)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                rows += [x for x in csv.DictReader(open(f)) if "glv_frame_kernel" in x["Kernel_Name"]]
            by = {}
            for x in rows:
                by.setdefault(x["Dispatch_Id"], {"dur": (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6})[x["Counter_Name"]] = by.get(x["Dispatch_Id"], {}).get(x["Counter_Name"], 0.0) + float(x["Counter_Value"])
            ds = sorted(by, key=int)[-25:]                         # the warm dispatches
            if not ds:
                print(f"set {si} proc {p}: no dispatches ({r.stderr[-200:]})"); continue
            mean = lambda k: sum(by[i].get(k, 0.0) for i in ds) / len(ds)      # noqa: E731
            print(f"set {si} proc {p}: kernel {mean('dur'):.4f} ms (under pmc; wall {wm.group(1) if wm else '?'})  " + "  ".join(f"{c} {mean(c):.4g}" for c in cs.split()), flush=True)
            subprocess.run(["rm", "-rf", d])


def alloc(cfg, procs):
    pols = os.environ.get("GLV_MODES_POLICIES", "malloc vmm:0 vmm:2 vmm:64 vmm:1024 fine uncached").split()
    res = {p: [] for p in pols}
    for _ in range(procs):                                      # alternating: a drift of the box would hit every policy alike
        for pol in pols: res[pol].append(wall_ms(cfg, {"GLV_STATE_ALLOC": pol}))
    for pol in pols: print(f"GLV_STATE_ALLOC={pol:9s} {cfg}: " + "  ".join(f"{x:.4f}" for x in res[pol]), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "alloc"
    cfg = sys.argv[2] if len(sys.argv) > 2 else "chain"
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    (counters if what == "counters" else alloc)(cfg, procs)
