"""Build the gfx950 shared objects in-tree (hipcc cross-compiles without a GPU).

    python -m glava_amd.build            # libglvspectrum.so (product)
    python -m glava_amd.build --tune     # + libglvtune.so (knob sweep used by tools/tune.py)

Outputs live next to the sources (glava_amd/csrc/*.so, git-ignored, shipped to the GPU
box by gpurun).  Objects are rebuilt only when a source/header is newer.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
TUNE_BIN = os.path.join(os.path.dirname(HERE), "tools", "bin")
ARCH = "gfx950"
# -ffp-contract=off: the bit-exactness contract forbids fusing the reference's separate
# multiplies and adds (SURVEY.md 7 "hard parts"); explicit __builtin_fmaf calls are unaffected.
HIPFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
            "-fno-fast-math", "-Wall", "-Wno-unused-function"]
SIZES = (7, 8, 9, 10, 11, 12, 13, 14)
PARTS = (0, 1, 2)        # glv_inst.hip is compiled per (size, part): s16 inputs / f32 inputs / the runner-up configuration


def _inst_jobs(obj_dir: str, sizes, extra: list[str]):
    """largest sizes first: they compile slowest, the pool should not end on them"""
    return [("glv_inst.hip", os.path.join(obj_dir, f"glv_inst_{k}_{p}.o"), [f"-DGLV_LOG_NN={k}", f"-DGLV_INST_PART={p}", *extra])
            for k in sorted(sizes, reverse=True) for p in PARTS]
HEADERS = ["glv_core.h", "glv_frame.h", "glv_kernel_tmpl.h", "glv_launch.h", "glv_tables.h", "glv_winsplit.h",
           os.path.join("..", "..", "include", "glv_spectrum.h")]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build step failed: " + os.path.basename(cmd[-1]))


def _compile(src: str, obj: str, extra: list[str]) -> str:
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _newer(obj, deps):
        cmd = [_hipcc(), *HIPFLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        _run(cmd)
        open(obj + ".cmd", "w").write(" ".join(cmd) + "\n")     # what this object was compiled with (tests/test_abi.py reads it)
    return obj


def _refuse_experiment_flags(flags: list[str]) -> None:
    """the product is compiled with HIPFLAGS + the size only: experiment / tuning macros (glv_core.h) must not reach it"""
    env = " ".join(os.environ.get(k, "") for k in ("HIPCC_COMPILE_FLAGS_APPEND", "HIPFLAGS", "CXXFLAGS", "CPPFLAGS"))
    bad = [f for f in flags if f.startswith("-D") and not f.startswith(("-DGLV_LOG_NN=", "-DGLV_INST_PART="))]
    if bad or "-DGLV_" in env:
        raise RuntimeError(f"product build refuses extra macro definitions: {bad or env!r} (use build_variant for A/B libraries)")


def build_tune_variant(name: str, extra_flags: list[str], variants: str | None = None) -> str:
    """Experimental knob-sweep library tools/bin/libglvtune_<name>.so compiled with extra
    flags (and optionally a custom variant list) -- used by tools/tune.py --lib for A/B tests.
    (tools/bin is git-ignored but travels to the GPU box; experiment libraries no longer live next to the product.)"""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(TUNE_BIN, exist_ok=True)
    obj = os.path.join(OBJ, f"glv_tune_{name}.o")
    flags = ["-DGLV_TUNE_BUILD", *[f for f in extra_flags if f != "-DGLV_TUNE_BUILD"]]
    if variants:
        flags.append("-DGLV_TUNE_VARIANTS=" + variants)
    _run([_hipcc(), *HIPFLAGS, *flags, "-c", os.path.join(CSRC, "glv_tune.hip"), "-o", obj])
    lib = os.path.join(TUNE_BIN, f"libglvtune_{name}.so")
    _run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib, obj])
    os.remove(obj)
    return lib


def build_variant(name: str, extra_flags: list[str], sizes=SIZES, kernels_only: bool = False, misc: bool = False) -> str:
    """A/B copy of the product library, glava_amd/csrc/libglvspectrum_<name>.so, compiled with extra flags
    (experiment macros); loaded instead of the product with GLV_SPECTRUM_LIB=<path> (tools/ab_bench.sh).
    kernels_only: the flags touch nothing but the frame kernels of `sizes` -- only those objects are compiled, everything
    else is the product's (which must be built and current): minutes instead of the whole library per experiment."""
    obj_dir = os.path.join(OBJ, name)
    os.makedirs(obj_dir, exist_ok=True)
    extra_flags = ["-DGLV_TUNE_BUILD", *[f for f in extra_flags if f != "-DGLV_TUNE_BUILD"]]      # the experiment macros' switch (glv_core.h)
    jobs = _inst_jobs(obj_dir, sizes, extra_flags)
    reused = []
    if kernels_only:
        reused = [os.path.join(OBJ, f"glv_inst_{k}_{p}.o") for k in SIZES if k not in sizes for p in PARTS]
        reused += [os.path.join(OBJ, o) for o in ("glv_api.o", "glv_multi.o")]
        if misc:        # the flags (also) touch the kernels of glv_misc.hip: sizes=() compiles nothing else
            jobs.append(("glv_misc.hip", os.path.join(obj_dir, "glv_misc.o"), list(extra_flags)))
        else:
            reused.append(os.path.join(OBJ, "glv_misc.o"))
    else:
        jobs.append(("glv_misc.hip", os.path.join(obj_dir, "glv_misc.o"), list(extra_flags)))
        jobs.append(("glv_api.cpp", os.path.join(obj_dir, "glv_api.o"), ["-x", "hip", *extra_flags]))
        jobs.append(("glv_multi.cpp", os.path.join(obj_dir, "glv_multi.o"), ["-x", "hip", *extra_flags]))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(lambda j: _compile(*j), jobs))
    lib = os.path.join(CSRC, f"libglvspectrum_{name}.so")
    _run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib, *objs, *reused])
    return lib


def build(tune: bool = False, verbose: bool = False) -> str:
    lib = os.path.join(CSRC, "libglvspectrum.so")
    # the built library travels to the GPU box, the objects need not: nothing to do when the library's stamp (newest source
    # mtime, taken when its build STARTED -- an edit during a build must not look built) still matches the sources
    srcs = [os.path.join(CSRC, f) for f in ("glv_inst.hip", "glv_misc.hip", "glv_api.cpp", "glv_multi.cpp")] + [os.path.join(CSRC, h) for h in HEADERS]
    stamp_path = lib + ".stamp"
    stamp = repr(max(os.path.getmtime(f) for f in srcs))
    if not tune and os.path.exists(lib) and os.path.exists(stamp_path) and open(stamp_path).read() == stamp:
        if verbose:
            print("up to date", lib)
        return lib
    os.makedirs(OBJ, exist_ok=True)
    jobs = _inst_jobs(OBJ, SIZES, [])
    jobs.append(("glv_misc.hip", os.path.join(OBJ, "glv_misc.o"), []))
    jobs.append(("glv_api.cpp", os.path.join(OBJ, "glv_api.o"), ["-x", "hip"]))
    jobs.append(("glv_multi.cpp", os.path.join(OBJ, "glv_multi.o"), ["-x", "hip"]))
    for _, _, extra in jobs:
        _refuse_experiment_flags(extra)
    if tune:
        jobs.append(("glv_tune.hip", os.path.join(OBJ, "glv_tune.o"), ["-DGLV_TUNE_BUILD"]))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(lambda j: _compile(*j), jobs))
    prod = [o for o in objs if not o.endswith("glv_tune.o")]
    if _newer(lib, prod) or not os.path.exists(stamp_path) or open(stamp_path).read() != stamp:
        _run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib, *prod])
    # objects compiled from sources that changed meanwhile carry mtimes newer than those sources and would be kept: the
    # stamp says which source state this library was built FROM, the next call compares it with the state it finds
    if repr(max(os.path.getmtime(f) for f in srcs)) == stamp:
        open(stamp_path, "w").write(stamp)
    else:
        for o in prod:
            os.remove(o)
        if os.path.exists(stamp_path): os.remove(stamp_path)
    if tune:
        tlib = os.path.join(CSRC, "libglvtune.so")
        tobj = os.path.join(OBJ, "glv_tune.o")
        if _newer(tlib, [tobj]):
            _run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tlib, tobj])
    if verbose:
        print("built", lib)
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--variant", nargs="+", metavar=("NAME", "FLAG"), help="build libglvspectrum_NAME.so with extra flags; write them without the leading dash (DGLV_X=1)")
    a = ap.parse_args()
    if a.variant:
        print("built", build_variant(a.variant[0], ["-" + f.lstrip("-") for f in a.variant[1:]]))
    else:
        build(tune=a.tune, verbose=True)
