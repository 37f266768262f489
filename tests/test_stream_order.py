"""Stream-ordered for real.

Everything a process call needs -- tilt table, bar tables and work lists, smooth bounds, the internal spectra rows, device
rings, the state arrays -- is made by glv_batch_create (or glv_batch_set_params).  The process calls and ring updates launch
kernels and asynchronous device-to-device copies, nothing else:

  CPU   the bodies of the functions on that path in glava_amd/csrc/glv_api.cpp contain no allocating / synchronising HIP call;
  GPU   the FIRST glv_batch_process_s16 after creation is captured into a hipGraph (hipStreamBeginCapture, global mode: any
        allocation or synchronous copy would invalidate the capture) and replayed; results equal eager execution bit for bit,
        for BASELINE configs[1] (stateless pass), the F = 5 chain, fused bars, the GL_R16 chain and a ring update.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle_lib import lcg_pcm_fast

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORBIDDEN = ["hipMalloc", "hipFree", "hipMemcpy(", "hipMemset(", "hipMemcpyAsync(", "hipStreamSynchronize", "hipDeviceSynchronize",
             "hipHostMalloc", "hipEventSynchronize", "hipMemcpyToSymbol", "hipMemcpyFromSymbol"]


def _function_body(src, signature_re):
    m = re.search(signature_re, src)
    assert m, signature_re
    i = src.index("{", m.end() - 1)
    depth, j = 0, i
    while True:
        if src[j] == "{": depth += 1
        elif src[j] == "}":
            depth -= 1
            if depth == 0: break
        j += 1
    return src[i:j + 1]


def _strip_comments(s):
    s = re.sub(r"//[^\n]*", "", s)
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def test_process_path_has_no_allocating_or_synchronising_call():
    src = open(os.path.join(ROOT, "glava_amd", "csrc", "glv_api.cpp")).read()
    path = [r"\nint process\(glv_batch\* b, const void\* d_in,", r"\nint check_ops\(", r"\nvoid launch_plan\(", r"\nstatic int ring_append\(",
            r"\nstatic int ring_push\(", r"\nint glv_batch_process_s16\(", r"\nint glv_batch_process_f32\(", r"\nint glv_batch_process_f32_stereo\(",
            r"\nint glv_batch_ring_update_s16\(", r"\nint glv_batch_ring_update_f32\(", r"\nint glv_batch_ring_append_s16\(",
            r"\nint glv_batch_ring_append_f32\(", r"\nint glv_batch_ring_planar\(", r"\nint glv_batch_bars\(", r"\nint timed_launch_end\("]
    for sig in path:
        body = _strip_comments(_function_body(src, sig))
        # the one copy the path may issue: the asynchronous device-to-device row copy of operator-only chains
        body = body.replace("hipMemcpyAsync(d_out, d_in, sizeof(float) * (size_t) units * b->p.n, hipMemcpyDeviceToDevice, st)", "")
        for f in FORBIDDEN:
            assert f not in body, (sig, f)
    # and the helpers that DO allocate are reachable from creation / set_params only
    for helper in ("ensure_bar_tables", "ensure_smooth_tables", "batch_prepare", "set_tilt", "batch_alloc"):
        for sig in path:
            assert helper + "(" not in _strip_comments(_function_body(src, sig)), (sig, helper)


class _Hip:
    def __init__(self):
        self.L = C.CDLL("libamdhip64.so")

    def check(self, rc, what):
        assert rc == 0, (what, rc)

    def capture(self, stream_ptr, fn):
        """run fn() under stream capture on stream_ptr, return an instantiated graph"""
        self.check(self.L.hipStreamBeginCapture(C.c_void_p(stream_ptr), 0), "hipStreamBeginCapture")     # hipStreamCaptureModeGlobal
        try:
            fn()
        finally:
            graph = C.c_void_p()
            rc = self.L.hipStreamEndCapture(C.c_void_p(stream_ptr), C.byref(graph))
        self.check(rc, "hipStreamEndCapture")
        exe = C.c_void_p()
        self.check(self.L.hipGraphInstantiate(C.byref(exe), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")
        return graph, exe

    def launch(self, exe, stream_ptr):
        self.check(self.L.hipGraphLaunch(exe, C.c_void_p(stream_ptr)), "hipGraphLaunch")

    def destroy(self, graph, exe):
        self.L.hipGraphExecDestroy(exe); self.L.hipGraphDestroy(graph)


def _bits_equal(a, b):
    import torch
    return bool(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["configs1", "chain_F5", "fused_bars", "gl_default", "gl_default_sm", "gravity_out_is_state", "unfused_bars", "gl_default_sm_hybrid", "bars_maximum_live"])
def test_first_process_call_can_be_captured_and_replayed(glvlib, case):
    """capture the FIRST call(s) after glv_batch_create into a hipGraph, replay, compare with an eagerly driven twin batch"""
    import torch
    G = glvlib
    hip = _Hip()
    n, streams, F = 4096, 64, 5
    spec = {
        "configs1": dict(p=G.Params(n=n), mask=G.OP_FFT, ops=G.OP_FFT, dt=torch.float32, w=n, per_graph=1),
        "chain_F5": dict(p=G.Params(n=n, avg_frames=F), mask=G.OP_GRAVITY | G.OP_AVERAGE, ops=G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE,
                         dt=torch.float32, w=n, per_graph=F),
        "fused_bars": dict(p=G.Params(n=n, avg_frames=F), mask=G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS,
                           ops=G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, dt=torch.float32, w=80, per_graph=F),
        "gl_default": dict(p=G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1), mask=G.OP_GRAVITY | G.OP_AVERAGE,
                           ops=G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16, dt=torch.int16, w=n, per_graph=F),
        # + the pre-smoothing pass (two launches; the matrix-core kernel with its > 64 KiB LDS opt-in set at creation)
        "gl_default_sm": dict(p=G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5), mask=G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS,
                              ops=G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, dt=torch.int16, w=n, per_graph=F, streams=128),
        # ... under a user's SAMPLE_MODE (ABI 7): glv_bars_mode_kernel behind the transform, its rows in dynamic LDS below the 64 KiB a launch may ask for
        "gl_default_sm_hybrid": dict(p=G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5, sample_mode=G.SAMPLE_HYBRID, round_formula=G.ROUND_CIRCULAR),
                                     mask=G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, ops=G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, dt=torch.int16, w=n, per_graph=F),
        "bars_maximum_live": dict(p=G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5, sample_mode=G.SAMPLE_MAXIMUM),
                                  mask=G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_BARS_ONLY, ops=G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_R16, dt=torch.int16, w=n,
                                  per_graph=F),
        "gravity_out_is_state": dict(p=G.Params(n=n), mask=G.OP_GRAVITY, ops=G.OP_FFT | G.OP_GRAVITY | G.OP_OUTPUT_IS_STATE,
                                     dt=torch.float32, w=n, per_graph=1),
        "unfused_bars": dict(p=G.Params(n=512, avg_frames=F), mask=G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS,
                             ops=G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, dt=torch.float32, w=80, per_graph=F),
    }[case]
    p = spec["p"]
    nn = p.n
    streams = spec.get("streams", streams)
    pcm = [torch.from_numpy((lcg_pcm_fast(7300 + u, streams * 2 * nn) // 4).astype(np.int16)).cuda() for u in range(spec["per_graph"])]
    graph_b, eager_b = G.Batch(p, streams, spec["mask"]), G.Batch(p, streams, spec["mask"])
    o_graph = [torch.zeros((streams * 2, spec["w"]), dtype=spec["dt"], device="cuda") for _ in range(spec["per_graph"])]
    o_eager = [torch.zeros_like(o_graph[0]) for _ in range(spec["per_graph"])]
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    # per_graph consecutive updates in one graph: a stateful chain's ring head advances on the host, so one graph holds one
    # full turn of the ring (F updates) and replaying it continues the sequence exactly like F more eager calls
    out_is_state = bool(spec["ops"] & G.OP_OUTPUT_IS_STATE)

    def body():
        for u in range(spec["per_graph"]):
            graph_b.process_s16(pcm[u], o_graph[u], spec["ops"], stream=st.cuda_stream)
    graph, exe = hip.capture(st.cuda_stream, body)
    for rep in range(3):
        hip.launch(exe, st.cuda_stream)
        st.synchronize()
        for u in range(spec["per_graph"]):
            eager_b.process_s16(pcm[u], o_eager[u], spec["ops"])
        torch.cuda.synchronize()
        for u in range(spec["per_graph"]):
            assert _bits_equal(o_graph[u], o_eager[u]), (case, rep, u)
        assert not out_is_state or graph_b.gravity_state() == o_graph[0].data_ptr()
    hip.destroy(graph, exe)
    graph_b.close(); eager_b.close()


@pytest.mark.gpu
def test_ring_update_can_be_captured(glvlib):
    import torch
    G = glvlib
    hip = _Hip()
    n, streams, nf = 1024, 32, 256
    bg, be = (G.Batch(G.Params(n=n), streams, G.OP_FFT | G.OP_RING_S16) for _ in range(2))
    new = [torch.from_numpy(lcg_pcm_fast(40 + u, streams * nf * 2)).cuda() for u in range(n // nf)]
    og = [torch.zeros((streams * 2, n), dtype=torch.float32, device="cuda") for _ in new]
    oe = [torch.zeros_like(og[0]) for _ in new]
    st = torch.cuda.Stream()
    torch.cuda.synchronize()

    def body():                                                     # one full turn of the ring: the write position is host state
        for u, x in enumerate(new):
            bg.ring_update_s16(x, nf, og[u], G.OP_FFT, stream=st.cuda_stream)
    graph, exe = hip.capture(st.cuda_stream, body)
    for rep in range(2):
        hip.launch(exe, st.cuda_stream); st.synchronize()
        for u, x in enumerate(new):
            be.ring_update_s16(x, nf, oe[u], G.OP_FFT)
        torch.cuda.synchronize()
        for u in range(len(new)):
            assert _bits_equal(og[u], oe[u]), (rep, u)
    hip.destroy(graph, exe)
    bg.close(); be.close()


@pytest.mark.gpu
def test_operators_without_tables_are_refused_not_allocated(glvlib):
    """a chain whose buffers the creation mask did not announce fails with GLV_ERR_STATE instead of allocating mid-stream"""
    import torch
    G = glvlib
    n, streams = 1024, 4
    d_pcm = torch.from_numpy(lcg_pcm_fast(5, streams * 2 * n)).cuda()
    o = torch.zeros((streams * 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n), streams, G.OP_FFT)                    # no rings, no internal spectra rows
    with pytest.raises(G.GlvError) as ei:
        b.ring_update_s16(d_pcm[: streams * 2 * 128], 128, o, G.OP_FFT)
    assert ei.value.code == G.ERR_STATE
    with pytest.raises(G.GlvError) as ei:
        b.process_s16(d_pcm, torch.zeros((streams * 2, 80), device="cuda"), G.OP_FFT | G.OP_BARS)     # stateless bars need the scratch rows
    assert ei.value.code == G.ERR_STATE
    b.process_s16(d_pcm, o, G.OP_FFT)                                 # the batch still works
    # knobs change through glv_batch_set_params, not behind the library's back
    b2 = G.Batch(G.Params(n=n), streams, G.OP_FFT | G.OP_BARS)
    ob = torch.zeros((streams * 2, 80), device="cuda")
    b2.process_s16(d_pcm, ob, G.OP_FFT | G.OP_BARS)
    b2.set_params(G.Params(n=n, bars=40, fft_scale=7.0))
    ob2 = torch.zeros((streams * 2, 40), device="cuda")
    b2.process_s16(d_pcm, ob2, G.OP_FFT | G.OP_BARS)
    ref = G.Batch(G.Params(n=n, bars=40, fft_scale=7.0), streams, G.OP_FFT | G.OP_BARS)
    oref = torch.zeros_like(ob2)
    ref.process_s16(d_pcm, oref, G.OP_FFT | G.OP_BARS)
    assert _bits_equal(ob2, oref)
    with pytest.raises(G.GlvError) as ei:
        b2.set_params(G.Params(n=2 * n))
    assert ei.value.code == G.ERR_STATE
    for x in (b, b2, ref): x.close()
