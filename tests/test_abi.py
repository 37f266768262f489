"""CPU: the C-ABI shared library builds for gfx950, loads, exports every symbol declared in
include/glv_spectrum.h, validates arguments, and -- with no GPU in this container -- refuses
to compute instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "glv_spectrum.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(glv_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_surface():
    names = declared_functions()
    for must in ("glv_fft", "glv_gravity", "glv_average", "glv_unpack_s16", "glv_batch_create",
                 "glv_batch_process_s16", "glv_batch_destroy", "glv_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(glvlib):
    L = C.CDLL(glvlib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, missing


def test_header_compiles_as_plain_c(tmp_path):
    """The boundary must be consumable from C (the reference host is C)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "glv_spectrum.h"\nint main(void){ glv_params p; (void)p; return GLV_OK; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)


def test_defaults_match_shipped_config(glvlib):
    cp = glvlib.CParams()
    glvlib.lib().glv_params_default(C.byref(cp))
    assert cp.n == 4096 and cp.channels == 2 and cp.avg_frames == 5 and cp.avg_window == 1
    assert cp.fft_scale == np.float32(10.2) and cp.fft_cutoff == np.float32(0.3)
    assert cp.gravity_step == np.float32(4.2) and cp.ur == np.float32(22050 / 256)
    assert glvlib.lib().glv_abi_version() == 4


def test_argument_validation(glvlib):
    P = glvlib.Params
    for bad in (P(n=1000), P(n=128), P(n=65536), P(channels=3), P(avg_frames=0), P(avg_frames=65), P(ur=float("nan"))):
        with pytest.raises(glvlib.GlvError) as ei:
            glvlib.Batch(bad, 4)
        assert ei.value.code == glvlib.ERR_INVALID
    with pytest.raises(glvlib.GlvError):
        glvlib.Batch(P(), 0)


def test_no_silent_cpu_fallback(glvlib):
    """Without a device the product must fail loudly (GLV_ERR_NO_DEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the fallback check is for device-less hosts")
    assert glvlib.device_count() == 0
    with pytest.raises(glvlib.GlvError) as ei:
        glvlib.Batch(glvlib.Params(), 8)
    assert ei.value.code == glvlib.ERR_NO_DEVICE
    with pytest.raises(glvlib.GlvError) as ei:
        glvlib.State(glvlib.Params())
    assert ei.value.code == glvlib.ERR_NO_DEVICE
    l = np.zeros(4, np.float32); r = np.zeros(4, np.float32)
    with pytest.raises(glvlib.GlvError) as ei:
        glvlib.unpack_s16(np.zeros(8, np.int16), 4, 2, l, r)
    assert ei.value.code == glvlib.ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """Nothing under glava_amd/ may import, include or link oracle/ or the emulator."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "glava_amd")):
        if os.path.basename(d) in ("build", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            txt = open(os.path.join(d, f), errors="ignore").read()
            if re.search(r"liboracle|glv_oracle|libglvref|oracle_lib|glv_emu|#include\s+\"[^\"]*oracle", txt):
                bad.append(os.path.join(d, f))
    assert not bad, bad
    import subprocess
    out = subprocess.run(["ldd", os.path.join(ROOT, "glava_amd", "csrc", "libglvspectrum.so")],
                         capture_output=True, text=True).stdout
    assert "oracle" not in out and "glvref" not in out and "glvemu" not in out


def test_integration_shim_library_links_the_abi():
    """oracle/_ref/libglvshim.so (the reference's render.c / fifo.c with integration/*.c compiled in) is built
    whenever /root/reference is present: it must export the harness entry points and the hipfifo backend's
    constructor, and import (not define) the C ABI it forwards to."""
    import subprocess
    path = os.path.join(ROOT, "oracle", "_ref", "libglvshim.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libglvshim.so not built (needs /root/reference at build time)")
    out = subprocess.run(["nm", "-D", path], capture_output=True, text=True, check=True).stdout
    defined = {l.split()[-1] for l in out.splitlines() if " T " in l or " D " in l or " B " in l}
    undefined = {l.split()[-1] for l in out.splitlines() if " U " in l}
    for sym in ("glvshim_run", "glvshim_backend_run", "_hipfifo_construct", "transform_fft_hip", "transform_fga_hip", "glv_hip_release"):
        assert sym in defined, sym
    for sym in ("glv_fft", "glv_gravity", "glv_average", "glv_fft_gravity_average", "glv_state_create",
                "glv_batch_ring_update_s16", "glv_device_upload", "glv_device_download"):
        assert sym in undefined, sym
