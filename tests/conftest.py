"""Test configuration.

Markers
    gpu   needs a real MI355X (run by the driver with `-m gpu` on the GPU box); everything
          else runs on the CPU-only build container (`-m "not gpu"`).

The CPU suite covers: the oracle against the compiled reference and the committed golden
vectors, the host emulator of the kernel phases against the oracle, host-side tables,
the C-ABI surface (symbols, error behaviour without a device), and the multi-process
(gloo) sharding logic.  The GPU suite is the parity suite proper and calls through the
C ABI (libglvspectrum.so) only.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle, build_oracles
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        build_oracles()
    Oracle.lib()
    return Oracle


@pytest.fixture(scope="session")
def ref():
    from oracle_lib import Ref
    if not Ref.available():
        pytest.skip("compiled reference (oracle/_ref/libglvref.so) not available")
    Ref.lib()
    return Ref


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def emu():
    """Host emulator of the kernel phases (tests/emu/glv_emu.cpp), built with g++."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "emu", "glv_emu.cpp")
    so = os.path.join(ROOT, "tests", "emu", "libglvemu.so")
    deps = [src] + [os.path.join(ROOT, "glava_amd", "csrc", h) for h in ("glv_core.h", "glv_frame.h", "glv_tables.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", "-o", so, src], check=True)
    return C.CDLL(so)


@pytest.fixture(scope="session")
def glvlib():
    """The product library; built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from glava_amd import build as B
    B.build()
    # torch ships its own HIP runtime: it must be the first one the process initialises -- a harness library that pulls in the
    # system's libamdhip64 first (oracle/_ref/libglvnullgl_hip.so) leaves torch with "No HIP GPUs are available"
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from glava_amd import spectrum
    return spectrum
