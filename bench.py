#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configuration.

metric    stereo 4096-pt (setbufsize 4096) unpack+window+FFT+magnitude frames/s -- `value`, the pass configs[1] names;
          the same line carries the FFT+smooth chain the metric's name alludes to (`smooth_chain`: fft -> gravity ->
          average, what GLava's modules request), the bit-faithful log mode (`strict_log`) and the GL_R16 texel output
          (`r16_texels`), each with its own algorithmic bytes and roofline fraction
workload  configs[1]: 1 MI355X, 64K batched stereo streams, N=4096; per GPU for --gpus N
          (configs[3]: 512K streams sharded 64K-per-GPU over 8 GPUs; weak scaling, no data-path
          collective -- streams are independent; RCCL only gathers the per-rank stats record)
step      one pass of the fused HIP kernel over every stream of the rank (65536 frames)

Inputs are resident in HBM before the timed region.  One JSON line on rank 0.

    python bench.py                       # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)


def usable_cores() -> tuple[int, str]:
    """Cores this process may actually run on: the scheduler affinity mask, capped by the cgroup CPU quota
    (cpu.max of cgroup v2 / cfs quota of v1).  os.cpu_count() ignores both."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    note = f"affinity {aff}"
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max": quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0: quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    if quota is not None:
        note += f", cgroup quota {quota:.1f} cpus"
        aff = max(1, min(aff, int(quota + 0.999)))
    return aff, note


def cpu_baseline(n: int, seconds: float = 10.0) -> dict | None:
    """Time the reference's own transform_fft (oracle/_ref, kind "reference") -- or, if that library is absent,
    the C restatement (kind "port") -- on a bounded sample of the same workload (stereo frames of uniform s16 noise,
    N real samples per channel): first on ONE native thread, then on one native pthread per usable core
    (oracle/ref_shim.c glvref_bench_mt; no Python threads, no oversubscription)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import numpy as np
        from oracle_lib import Oracle, Ref
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    cores, note = usable_cores()
    frames_per_call = 64
    pcm = np.random.default_rng(12345).integers(-32768, 32768, frames_per_call * 2 * n, dtype=np.int16)
    use_ref = Ref.available()
    if use_ref:
        p = Ref.params()
        run = lambda thr, sec, done: Ref.lib().glvref_bench_mt(C.byref(p), pcm, frames_per_call, n, 0, thr, sec, done)  # noqa: E731
    else:
        run = lambda thr, sec, done: Oracle.lib().glvo_bench_mt(pcm, frames_per_call, n, 10.2, 0.3, thr, sec, done)   # noqa: E731

    def timed(threads, sec):
        done = (C.c_ulonglong * threads)()
        dt = run(threads, sec, done)
        if dt <= 0: raise RuntimeError(f"cpu baseline failed ({dt})")
        return sum(done), dt, min(done), max(done)

    one_frames, one_dt, _, _ = timed(1, min(3.0, seconds))
    tot, dt, lo, hi = timed(cores, seconds)
    one = one_frames / one_dt
    # BASELINE configs[0]: the reference's own CPU path as GLava runs it -- one stream, N=1024, the bars module's chain
    # (window, fft, gravity, avg: transform_fft -> transform_gravity -> transform_average, F = 5 windowed) on ONE core
    c0 = None
    if use_ref:
        try:
            n0 = 1024
            p0 = Ref.params(avg_frames=5, avg_window=True)
            pcm0 = np.random.default_rng(1).integers(-32768, 32768, 256 * 2 * n0, dtype=np.int16)
            done0 = (C.c_ulonglong * 1)()
            dt0 = Ref.lib().glvref_bench_mt(C.byref(p0), pcm0, 256, n0, 1, 1, min(2.0, seconds), done0)
            c0 = {"value": done0[0] / dt0, "unit": "frames/s", "cores": 1, "kind": "reference",
                  "sample": f"BASELINE configs[0]: {done0[0]} stereo frames N={n0} through the reference's transform_fft -> transform_gravity -> "
                            f"transform_average (F=5, windowed) + fifo.c unpack on 1 native thread in {dt0:.1f} s; GLava needs 86 updates/s"}
        except Exception as e:  # pragma: no cover
            c0 = {"error": str(e)}
    return {"value": tot / dt, "configs[0]": c0, "unit": "frames/s", "cores": cores, "kind": "reference" if use_ref else "port",
            "one_core": one, "per_core": tot / dt / cores, "parallel_efficiency": (tot / dt) / (one * cores),
            "host_logical_cpus": os.cpu_count(), "cores_note": note,
            "sample": f"{tot} stereo frames N={n} (uniform s16 noise, one {frames_per_call}-frame buffer looped) on {cores} native "
                      f"pthreads in {dt:.1f} s (per thread {lo}..{hi} frames) after {one_frames} frames on 1 thread in {one_dt:.1f} s; "
                      f"reference transform_fft x2 ch + fifo.c unpack, gcc -O2"}


def measured_traffic(n: int, streams: int, ops: str):
    """HBM bytes per launch from the PMC passes of tools/profile.sh (FETCH_SIZE x2 gfx950 correction
    + WRITE_SIZE, separate rocprofv3 --pmc runs), committed as profiles/hbm_traffic.json; None when
    no measurement for this exact workload is on file."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        for r in rec:
            if r["n"] == n and r["streams"] == streams and r["ops"] == ops:
                return r["bytes_per_launch"]
    except Exception:
        pass
    return None


def live_traffic(a):
    """2*FETCH_SIZE + WRITE_SIZE of the headline kernel collected IN THIS RUN: two SEPARATE `rocprofv3 --pmc` passes (one counter each, no tracing
    domain beside them -- MI355X_MICROARCH.md's HBM recipe) over this same script reduced to its headline launches (--steps 2 --warmup 1, no configs), as child processes,
    outside every timed region; mean over the dispatches of the headline kernel.  Returns (bytes per launch or None, detail dict).  Never fails the bench line: any
    problem (no rocprofv3, a profiler already wrapped around this process, a timeout) returns None with the reason and the committed measurement is reported instead."""
    import csv, glob, shutil, subprocess, tempfile
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, {"skipped": "this process already runs under a rocprofiler tool"}
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"skipped": "rocprofv3 not found"}
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt", "--no-configs", "--sustained-s", "0",
             "--no-live-traffic", "--streams", str(a.streams), "--n", str(a.n), "--ops", a.ops, "--log-mode", str(a.log_mode)]
    tmp = tempfile.mkdtemp(prefix="glv_pmc_", dir="/tmp")
    vals, counts = {}, {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            try:
                r = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "t", "--"] + child, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90)
            except subprocess.TimeoutExpired:
                return None, {"skipped": f"the {ctr} pass timed out"}
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == ctr and row.get("Kernel_Name", "").startswith("void glv::glv_frame_kernel<"):
                        rows.append(float(row.get("Counter_Value", 0)))
            if not rows:
                return None, {"skipped": f"the {ctr} pass gave no rows for the headline kernel (rocprofv3 rc {r.returncode})"}
            vals[ctr], counts[ctr] = sum(rows) / len(rows), len(rows)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return int(round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)), {"fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"], "dispatches": counts}


def streaming_ceiling(n: int, ops: str):
    """Second, honest denominator: what a do-nothing streaming kernel reaches on MI355X for
    this pass's read/write mix (tools/membench.hip, committed as profiles/stream_ceiling.json).  Only on file
    for the FFT+magnitude pass of s16 input (1 byte read : 2 written)."""
    if ops != "fft":
        return None
    try:
        return float(json.load(open(os.path.join(ROOT, "profiles", "stream_ceiling.json")))["gbs"])
    except Exception:
        return None


def extra_configs(G, torch, device, a, peak_gbs):
    """The other single-GPU configurations of BASELINE.json, timed like the headline (spin-up, warm-up, K launches bracketed by
    synchronize; kernel time from HIP events on the launch stream):
      configs[2]  N=16384 x 8192 streams, fft + gravity smoothing + 80-bar radial bin averaging (20 N + 640 B / frame, SURVEY 8d
                  row D) and the same chain with the full spectra as output (20 N, row B: the output is the state)
      configs[4]  N in {512 ... 8192}, equal bytes per class, one batch per class on its own HIP stream: every class alone, then
                  all five in flight together (aggregate)
      n8192 / n16384   the stateless pass (12 N) at equal bytes to the headline"""
    import time
    steps, warm = a.configs_steps, 2

    def run(batch, call):
        t_end = time.perf_counter() + a.spinup_s                      # the headline's spin-up (0.15 s was thinner)
        while time.perf_counter() < t_end:
            for _ in range(4): call()
            torch.cuda.synchronize()
        for _ in range(warm): call()
        torch.cuda.synchronize()
        if batch is not None: batch.timing_begin()
        t0 = time.perf_counter()
        for _ in range(steps): call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kms = None
        if batch is not None:
            ms, nl = batch.timing_end()
            kms = ms / max(nl, 1)
        return dt, kms

    def place(batch, d_in, d_out, ops):
        """placement wisdom (glv_batch_tune_placement): the fastest of six placements of the state arrays for THESE buffers; untimed, like autotune"""
        if not a.tune_placement: return None
        try:
            f, b = batch.tune_placement(d_in, d_out, ops, 6, st0)
            return {"ms_first_placement": f, "ms_best_placement": b}
        except Exception as ex:                                   # never fails the bench line
            return {"error": str(ex)}

    def entry(note, frames, bytes_per_launch, dt, kms):
        k = kms if kms else dt * 1e3
        return {"note": note, "value": frames / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "avg_kernel_ms": k,
                "algorithmic_bytes_per_launch": bytes_per_launch, "roofline_frac": bytes_per_launch / (k * 1e-3) / 1e9 / peak_gbs}

    out = {}
    gen = torch.Generator(device="cuda"); gen.manual_seed(777)
    st0 = torch.cuda.current_stream().cuda_stream
    # --- the large stateless sizes
    for n, s in ((8192, 32768), (16384, 16384)):
        pcm = torch.randint(-32768, 32768, (s, n, 2), dtype=torch.int16, device="cuda", generator=gen)
        o = torch.empty((s, 2, n), dtype=torch.float32, device="cuda")
        b = G.Batch(G.Params(n=n, log_mode=a.log_mode), s, G.OP_FFT, device=device)
        dt, kms = run(b, lambda: b.process_s16(pcm, o, G.OP_FFT, st0))
        out[f"n{n}_stateless"] = entry(f"N={n} x {s} streams, window+FFT+magnitude, 12 N B/frame", s, b.algorithmic_bytes(G.OP_FFT), dt, kms)
        b.close(); del pcm, o
    # --- configs[2]
    n, s, bars = 16384, 8192, 80
    pcm = torch.randint(-32768, 32768, (s, n, 2), dtype=torch.int16, device="cuda", generator=gen)
    o = torch.empty((s, 2, n), dtype=torch.float32, device="cuda")
    ob = torch.empty((s, 2, bars), dtype=torch.float32, device="cuda")
    gops = G.OP_FFT | G.OP_GRAVITY
    b = G.Batch(G.Params(n=n, log_mode=a.log_mode, bars=bars), s, G.OP_GRAVITY, device=device)
    pl = place(b, pcm, ob, gops | G.OP_BARS)
    dt, kms = run(b, lambda: b.process_s16(pcm, ob, gops | G.OP_BARS, st0))
    c2 = entry(f"BASELINE configs[2]: N={n} x {s} streams, fft + gravity + radial bin averaging to {bars} bars/channel (fused: bars from the row in LDS), "
               f"20 N + 640 B/frame (SURVEY 8d row D)", s, (20 * n + 8 * bars) * s, dt, kms)
    c2["placement"] = pl
    b.reset()
    gois = gops | G.OP_OUTPUT_IS_STATE
    dt, kms = run(b, lambda: b.process_s16(pcm, o, gois, st0))
    c2["spectra_out"] = entry("same size, fft + gravity with the full spectra as output == state (GLV_OP_OUTPUT_IS_STATE), 20 N B/frame (SURVEY 8d row B)", s, b.algorithmic_bytes(gois), dt, kms)
    b.reset()
    dt, kms = run(b, lambda: b.process_s16(pcm, o, gops, st0))
    c2["spectra_out_private_state"] = entry("same chain with the batch-owned state (the default): spectra written twice, 28 N B/frame", s, b.algorithmic_bytes(gops), dt, kms)
    # ... and with GLV_OP_BARS_ONLY (round 6): the 80 bars are all this configuration outputs, and smooth_audio() samples bins below 0.30 n -- the live kernel
    # class (8) computes magnitude and keeps the gravity state for 3/8 of the row only; the same bars bit for bit (tests/test_gl_fused.py).  roofline_frac is
    # against ITS OWN algorithmic bytes (4 N + 16 L + 640, L the live bins); frac_of_full_chain prices the same frames at configs[2]'s 20 N + 640 B.
    b6 = G.Batch(G.Params(n=n, log_mode=a.log_mode, bars=bars), s, G.OP_GRAVITY | G.OP_BARS | G.OP_BARS_ONLY, device=device)
    pl = place(b6, pcm, ob, gops | G.OP_BARS)
    dt, kms = run(b6, lambda: b6.process_s16(pcm, ob, gops | G.OP_BARS, st0))
    c2["bars_only"] = entry(f"BASELINE configs[2] with GLV_OP_BARS_ONLY: live bins {b6.live_bins()} of {n}; same bars bit for bit", s, b6.algorithmic_bytes(gops | G.OP_BARS), dt, kms)
    c2["bars_only"]["live_bins"] = b6.live_bins()
    c2["bars_only"]["placement"] = pl
    c2["bars_only"]["frac_of_full_chain"] = (20 * n + 8 * bars) * s / (c2["bars_only"]["avg_kernel_ms"] * 1e-3) / 1e9 / peak_gbs
    b6.close()
    out["configs[2]"] = c2
    b.close(); del pcm, o, ob
    torch.cuda.empty_cache()
    # --- the pipeline GLava ships (rc.glsl:211 setaccelfft, render.c:2188-2303): GL_R16 upload, GL_MAX + gravity pass, ring, average
    # pass -- one launch on 16-bit state (gl_storage 1): N=4096 x 64K streams, F=5
    n, s, F, bars = a.n, a.streams, 5, 80
    pcm = torch.randint(-32768, 32768, (s, n, 2), dtype=torch.int16, device="cuda", generator=gen)
    q = torch.empty((s, 2, n), dtype=torch.int16, device="cuda")
    qb = torch.empty((s, 2, bars), dtype=torch.int16, device="cuda")
    glops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
    pg = G.Params(n=n, log_mode=a.log_mode, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=bars)
    b = G.Batch(pg, s, G.OP_GRAVITY | G.OP_AVERAGE, device=device)
    pl = place(b, pcm, q, glops | G.OP_R16)
    dt, kms = run(b, lambda: b.process_s16(pcm, q, glops | G.OP_R16, st0))
    gl = entry(f"GLava's shipped pipeline (setaccelfft): N={n} x {s} streams, s16 PCM -> upload quantisation -> GL_MAX + gravity pass -> ring -> "
               f"average pass (F={F}, Hamming newest-first) -> `av` GL_R16 texels, ONE launch on uint16 state: 4N + 16N + 4N + 4N = 28 N B/frame",
               s, b.algorithmic_bytes(glops | G.OP_R16), dt, kms)
    gl["launches_per_step"] = b.last_launches()
    gl["placement"] = pl
    b.reset()
    pl = place(b, pcm, qb, glops | G.OP_BARS | G.OP_R16)
    dt, kms = run(b, lambda: b.process_s16(pcm, qb, glops | G.OP_BARS | G.OP_R16, st0))
    gl["bars_out"] = entry(f"same chain with the {bars} bars of the bars / radial modules computed in the kernel (from the finished row in LDS) and "
                           f"stored as GL_R16 texels: 24 N + {4 * bars} B/frame", s, b.algorithmic_bytes(glops | G.OP_BARS | G.OP_R16), dt, kms)
    gl["bars_out"]["launches_per_step"] = b.last_launches()
    gl["bars_out"]["placement"] = pl
    b.close()
    b7 = G.Batch(pg, s, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_BARS_ONLY, device=device)       # the same with GLV_OP_BARS_ONLY (kernel class 9)
    pl = place(b7, pcm, qb, glops | G.OP_BARS | G.OP_R16)
    dt, kms = run(b7, lambda: b7.process_s16(pcm, qb, glops | G.OP_BARS | G.OP_R16, st0))
    gl["bars_out_live"] = entry(f"the GL chain + {bars} bars with GLV_OP_BARS_ONLY: live bins {b7.live_bins()} of {n}, one launch, same bars bit for bit; roofline_frac on its own bytes",
                                s, b7.algorithmic_bytes(glops | G.OP_BARS | G.OP_R16), dt, kms)
    gl["bars_out_live"]["launches_per_step"] = b7.last_launches()
    gl["bars_out_live"]["live_bins"] = b7.live_bins()
    gl["bars_out_live"]["placement"] = pl
    gl["bars_out_live"]["frac_of_24N"] = gl["bars_out"]["algorithmic_bytes_per_launch"] / (gl["bars_out_live"]["avg_kernel_ms"] * 1e-3) / 1e9 / peak_gbs
    b7.close()
    # ... and with the pre-smoothing pass of render.c:2277-2303 behind it (bars == n at the texel centres: the texture every stock
    # module samples under setsmoothpass, GLava's shipped configuration): the fused GL kernel hands the `av` rows over as uint16
    # texels and the pass runs in exact integer arithmetic on the i8 matrix cores (round 5) -- two launches.  roofline_frac is
    # against the CHAIN's algorithmic bytes, 28 N per frame: PCM in, four ring slots read, one written, `sm` texels out (the
    # intermediate `av` rows are not counted -- traffic the organisation adds, not the problem needs)
    smops = glops | G.OP_BARS | G.OP_R16
    psm = G.Params(n=n, log_mode=a.log_mode, avg_frames=F, avg_window_kind=1, gl_storage=1, bars=n, bar_phase=0.5)
    sm_entries = {}
    for tag, s3 in (("full", s), ("quarter", max(s // 4, 1))):          # the whole batch (as every other entry), and round 4's quarter batch
        qs = torch.empty((s3, 2, n), dtype=torch.int16, device="cuda")
        b3 = G.Batch(psm, s3, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS, device=device)
        pl = place(b3, pcm, qs, smops)
        dt, kms = run(b3, lambda: b3.process_s16(pcm, qs, smops, st0))
        assert b3.algorithmic_bytes(smops) == 28 * n * s3
        sm_entries[tag] = entry(f"GLava's SHIPPED pipeline end to end: the chain above + the pre-smoothing pass (bars = n = {n}, bar_phase 0.5) -> `sm` GL_R16 texels, {s3} streams, "
                                f"two launches (glv_frame_kernel -> uint16 `av` rows -> glv_bars_rows_i8_kernel: exact integer weighted means on v_mfma_i32_32x32x32_i8); "
                                f"bytes: the chain's 28 N per frame", s3, b3.algorithmic_bytes(smops), dt, kms)
        sm_entries[tag]["launches_per_step"] = b3.last_launches()
        sm_entries[tag]["placement"] = pl
        b3.close(); del qs
    gl["sm_out"] = sm_entries["full"]
    gl["sm_out"]["quarter_batch"] = sm_entries["quarter"]
    # ... and as GLava actually consumes it (round 6, GLV_OP_BARS_ONLY): the modules sample the `sm` texture and nothing else (smooth.glsl:62),
    # and smooth_audio() reaches bins below 0.288 n + half a window -- the gravity store, ring and average of the bins beyond are dead values
    # the reference's GL passes compute for nobody.  The chain keeps and computes only the live bins; the `sm` texels are bit-identical
    # (tests/test_gl_fused.py).  roofline_frac is against THIS chain's own algorithmic bytes (PCM 4 N + live state + `sm` 4 N); frac_of_28N
    # prices the same frames at the full chain's bytes for comparison with sm_out -- it is NOT a roofline claim.
    qs = torch.empty((s, 2, n), dtype=torch.int16, device="cuda")
    b5 = G.Batch(psm, s, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | G.OP_BARS_ONLY, device=device)
    pl = place(b5, pcm, qs, smops)
    dt, kms = run(b5, lambda: b5.process_s16(pcm, qs, smops, st0))
    gl["sm_out_live"] = entry(f"the shipped pipeline with the state kept only where the pre-smoothing pass samples (GLV_OP_BARS_ONLY: live bins {b5.live_bins()} of {n}, "
                              f"kept in whole last-pass blocks), {s} streams, two launches, same `sm` texels bit for bit; bytes: 4 N + (4 (F - 1) + 4) L + 4 N per frame",
                              s, b5.algorithmic_bytes(smops), dt, kms)
    gl["sm_out_live"]["launches_per_step"] = b5.last_launches()
    gl["sm_out_live"]["placement"] = pl
    gl["sm_out_live"]["live_bins"] = b5.live_bins()
    gl["sm_out_live"]["frac_of_28N"] = 28 * n * s / (gl["sm_out_live"]["avg_kernel_ms"] * 1e-3) / 1e9 / peak_gbs
    b5.close(); del qs
    # ... and what a FUSED form (the pass as a phase of the transform's persistent workgroups) could at most buy, measured instead of
    # estimated: two batches of half the streams on two HIP streams, free-running, each transform held to ONE persistent workgroup per CU
    # (glv_batch_set_grid) so that a pass workgroup of the other half RESIDES beside it (222 + 235 registers, two waves per SIMD) -- the co-residence a fused
    # kernel would arrange by hand.  Wall clock over the same K updates of all the streams; caller-side pipelining, not the default entry.
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    for key, extra in (("sm_out", 0), ("sm_out_live", G.OP_BARS_ONLY)):
        try:
            half = s // 2
            qs = torch.empty((s, 2, n), dtype=torch.int16, device="cuda")
            hb = [G.Batch(psm, half, G.OP_GRAVITY | G.OP_AVERAGE | G.OP_BARS | extra, device=device) for _ in range(2)]
            for x in hb: x.set_grid(cus)
            trials = []
            for t in range(4):                                    # the runtime maps HIP streams onto a few hardware queues: two streams that share one never overlap
                hs = [torch.cuda.Stream(device=device) for _ in range(2)]     # (profiles/r06/overlap_grid.txt: 1 pair in 10, 1.52 instead of 1.20 ms, the same batches) -- four
                def two_halves():                                 # fresh pairs of streams, every value reported, the best one is `value`
                    for i in range(2): hb[i].process_s16(pcm[i * half:(i + 1) * half], qs[i * half:(i + 1) * half], smops, hs[i].cuda_stream)
                dt, _ = run(None, two_halves)
                trials.append(dt * 1e3)
            for x in hb: x.close()
            dt = min(trials) * 1e-3
            gl[key]["two_half_batches"] = {"note": f"the same {s} streams as two batches of {half} on two HIP streams, free-running, each transform on {cus} persistent workgroups (one per CU): the pass "
                                                   f"of one half resides beside the transform of the other -- the measured bound of what fusing the two launches could buy; wall clock; "
                                                   f"best of four fresh pairs of streams (two streams the runtime maps onto one hardware queue never overlap)",
                                           "value": s / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "ms_per_step_trials": trials, "vs_one_batch": gl[key]["ms_per_step"] / (dt * 1e3)}
            del qs
        except Exception as ex:                                   # never fails the bench line
            gl[key]["two_half_batches"] = {"error": str(ex)}
    s3 = max(s // 4, 1)
    # the pre-smoothing kernel by itself on rows already in HBM: its roofline is the f32 matrix rate (157.3 TFLOP/s dense at nominal
    # clock, MI355X_MICROARCH.md), counted on the USEFUL multiply-adds (smooth_audio()'s own taps; the tiles' padding is not counted)
    try:
        import numpy as np
        f32 = np.float32
        k = (np.arange(n, dtype=f32) + f32(0.5)) / f32(n)
        sc = lambda u: (-np.log((f32(-0.9) * u + f32(1.0)).astype(f32)).astype(f32) / f32(8.0)).astype(f32)
        smin = (sc(np.clip(k - f32(0.025), f32(0), f32(1)).astype(f32)) * f32(n)).astype(f32)
        smax = (sc(np.clip(k + f32(0.025), f32(0), f32(1)).astype(f32)) * f32(n)).astype(f32)
        taps = int(np.sum(np.floor((smax - smin).astype(np.float64)) + 1))
        rows_in = torch.rand((s3 * 2, n), dtype=torch.float32, device="cuda")
        b4 = G.Batch(G.Params(n=n, bars=n, bar_phase=0.5), s3, G.OP_FFT | G.OP_BARS, device=device)
        sync = torch.cuda.synchronize
        for _ in range(10): b4.bars(rows_in, qs4 := torch.empty((s3 * 2, n), dtype=torch.float32, device="cuda"))
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): b4.bars(rows_in, qs4)
        e1.record(); sync()
        ms = e0.elapsed_time(e1) / 20
        tf = 2.0 * taps * s3 * 2 / (ms * 1e-3) * 1e-12
        gl["sm_out"]["pass_alone_f32_rows"] = {"note": f"the FLOAT-row form of the pass (gl_storage 0; glv_batch_bars: one fma chain per bar on v_mfma_f32_32x32x2_f32), glv_bars_rows_kernel alone, "
                                                       f"{s3 * 2} float rows in HBM -> floats: {taps} useful multiply-adds per row.  (The GL chain above runs the integer form on the i8 matrix cores.)",
                                      "ms": ms, "roofline": {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3}}
        b4.close(); del rows_in, qs4
    except Exception as ex:                                   # never fails the bench line
        gl["sm_out"]["pass_alone_f32_rows"] = {"note": f"not measured: {ex}"}
    # the pass-by-pass form of the same chain (the checker: f32 intermediates, three launches), for the record
    s2 = s // 4
    b2 = G.Batch(G.Params(n=n, log_mode=a.log_mode, avg_frames=F, avg_window_kind=1, gl_storage=2), s2, G.OP_GRAVITY | G.OP_AVERAGE, device=device)
    dt, kms = run(b2, lambda: b2.process_s16(pcm, q, glops | G.OP_R16, st0))
    gl["pass_by_pass"] = entry(f"the same values pass by pass (gl_storage 2, the checker: transform -> f32 spectra -> gravity / average kernel on f32 state -> "
                               f"texels), {s2} streams; roofline_frac is against ITS OWN 64 N B/frame, frac_of_28N against the fused form's bytes", s2,
                               b2.algorithmic_bytes(glops | G.OP_R16), dt, kms)
    gl["pass_by_pass"]["launches_per_step"] = b2.last_launches()
    gl["pass_by_pass"]["frac_of_28N"] = 28 * n * s2 / (gl["pass_by_pass"]["avg_kernel_ms"] * 1e-3) / 1e9 / peak_gbs
    b2.close()
    out["gl_default"] = gl
    del q, qb
    # --- SURVEY 8e's only failure mode, probed without an 8-GPU node: the C multi-GPU driver with EIGHT shards on eight
    # host threads, all on this one device (host-gathered stats: GLV_MULTI_RCCL=0), against ONE shard of all the streams.  The device does the
    # same work either way; if launching from eight host threads serialised on the host (a lock in the launch path), the eight-shard run
    # would be slower by more than the smaller kernels explain.
    try:
        sh, msteps = 8, 20
        pm = G.Params(n=n, log_mode=a.log_mode)
        saved = os.environ.get("GLV_MULTI_RCCL")
        os.environ["GLV_MULTI_RCCL"] = "0"
        try:
            res = {}
            for nsh in (1, sh):
                m = G.Multi(pm, s, G.OP_FFT, devices=[device] * nsh)
                ins, outs = [], []
                for i in range(nsh):
                    _, lo_, cnt = m.shard(i)
                    ins.append(pcm[lo_:lo_ + cnt]); outs.append(torch.empty((cnt, 2, n), dtype=torch.float32, device="cuda"))
                m.run_s16(ins, outs, G.OP_FFT, warmup=20, steps=msteps)               # spin-up
                st_, mx = m.run_s16(ins, outs, G.OP_FFT, warmup=3, steps=msteps)
                res[nsh] = {"max_seconds": mx, "ms_per_step": mx / msteps * 1e3, "frames_per_s": sum(r["frames"] for r in st_) / mx}
                m.close(); del ins, outs
        finally:
            if saved is None: os.environ.pop("GLV_MULTI_RCCL", None)
            else: os.environ["GLV_MULTI_RCCL"] = saved
        out["multi_host_threads"] = {"note": f"glv_multi_run_s16, N={n} x {s} streams on ONE device: 1 shard / 1 host thread against {sh} shards / {sh} host threads "
                                             f"(each with its own batch and HIP stream; thread barrier + synchronize around {msteps} timed updates; stats gathered on the host)",
                                     "one_shard": res[1], f"{sh}_shards": res[sh], "ratio": res[sh]["max_seconds"] / res[1]["max_seconds"]}
    except Exception as ex:                                   # a probe, never a reason to lose the line
        out["multi_host_threads"] = {"note": f"not measured: {ex}"}
    # --- f1: the FIFO ring mode (fifo.c:91-112 on the device): append 256 new frames per stream, transform the whole window
    nf = 256
    new = torch.randint(-32768, 32768, (s, nf, 2), dtype=torch.int16, device="cuda", generator=gen)
    o = torch.empty((s, 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n, log_mode=a.log_mode), s, G.OP_FFT | G.OP_RING_S16, device=device)
    dt, kms = run(b, lambda: b.ring_update_s16(new, nf, o, G.OP_FFT, st0))
    out["ring_update"] = entry(f"SURVEY 8f1: glv_batch_ring_update_s16, N={n} x {s} streams, {nf} new frames appended per stream and update (a strided "
                               f"device-to-device copy), then window+FFT+magnitude of the whole ring read with a rotation; bytes: the transform's 12 N "
                               f"(the append moves 8 * {nf} B more); avg_kernel_ms covers copy + kernel", s, b.algorithmic_bytes(G.OP_FFT), dt, kms)
    b.close(); del pcm, new, o
    torch.cuda.empty_cache()
    # --- configs[4]
    classes = []
    for n in (512, 1024, 2048, 4096, 8192):
        s = 16384 * 4096 // n
        pcm = torch.randint(-32768, 32768, (s, n, 2), dtype=torch.int16, device="cuda", generator=gen)
        o = torch.empty((s, 2, n), dtype=torch.float32, device="cuda")
        classes.append((n, s, pcm, o, G.Batch(G.Params(n=n, log_mode=a.log_mode), s, G.OP_FFT, device=device), torch.cuda.Stream()))
    per = []
    for n, s, pcm, o, b, st in classes:
        dt, kms = run(b, lambda: b.process_s16(pcm, o, G.OP_FFT, st.cuda_stream))
        per.append(dict(entry(f"class N={n} x {s} streams alone", s, b.algorithmic_bytes(G.OP_FFT), dt, kms), n=n, streams=s))

    def all_classes():
        for n, s, pcm, o, b, st in classes:
            b.process_s16(pcm, o, G.OP_FFT, st.cuda_stream)
    dt, _ = run(None, all_classes)
    tot_bytes = sum(b.algorithmic_bytes(G.OP_FFT) for *_, b, _ in classes)
    tot_frames = sum(s for _, s, *_ in classes)
    agg = entry("BASELINE configs[4]: N in {512,1024,2048,4096,8192}, 256 MiB of PCM per class, one launch per class on five HIP streams in flight together; "
                "ms_per_step = wall clock of one round of all classes (rounds back to back)", tot_frames, tot_bytes, dt, None)
    agg["classes"] = per
    out["configs[4]"] = agg
    for *_, b, _ in classes: b.close()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=65536, help="stereo streams per GPU (configs[1]: 64K)")
    ap.add_argument("--n", type=int, default=4096, help="real samples per channel (setbufsize)")
    ap.add_argument("--ops", default="fft", choices=["fft", "fft+gravity", "fft+gravity+average"])
    ap.add_argument("--log-mode", type=int, default=1,
                    help="1 hardware log2 (library default, <= 1.8e-7 relative on every float), 0 bit-faithful fp64 table log")
    ap.add_argument("--grid", type=int, default=0, help="workgroups of the persistent kernel (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary fast-log measurement")
    ap.add_argument("--no-live-traffic", dest="live_traffic", action="store_false",
                    help="do not collect roofline.traffic with rocprofv3 --pmc child passes (the committed profiles/hbm_traffic.json is reported instead)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` key (BASELINE configs[2], configs[4], N=8192 / 16384)")
    ap.add_argument("--configs-steps", type=int, default=20, help="timed steps per entry of the `configs` key (same spin-up as the headline)")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="seconds of back-to-back headline launches behind the `sustained` key (0 = skip)")
    ap.add_argument("--check-dump", default="", help="test hook: every rank saves the PCM of its FIRST stream and the raw FFT the library "
                                                     "computes for it to <path>.rank<r>.npz (global stream index inside); tests/ compare with the oracle")
    ap.add_argument("--no-tune-placement", dest="tune_placement", action="store_false",
                    help="stateful entries: skip glv_batch_tune_placement (the fastest of six placements of the state arrays for the entry's buffers, "
                         "found outside the timed region; profiles/r06/modes.txt) and time whatever placement the allocation happened to give")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--spinup-s", type=float, default=0.3, help="untimed clock spin-up before the W warm-up steps")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    # --gpus N started as a PLAIN process (no launcher around it: RANK unset) must still drive N GPUs, or refuse -- never measure one GPU and
    # print n_gpus 1 (VERDICT r5 missing 5).  It re-executes itself under torch.distributed.run with one rank per GPU; fewer than N visible
    # devices is an error.  (GLV_BENCH_DEVICE_COUNT overrides the visible-device count and GLV_BENCH_SPAWN_DRYRUN=1 prints the command instead of
    # running it: test hooks, tests/test_bench_cli.py.)
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "RANK" not in os.environ:
        ndev = int(os.environ.get("GLV_BENCH_DEVICE_COUNT", "-1"))
        if ndev < 0: ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} HIP device(s) visible on this node -- refusing to measure fewer GPUs than asked for")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        if os.environ.get("GLV_BENCH_SPAWN_DRYRUN") == "1":
            print(json.dumps({"spawn": cmd, "devices": ndev}), flush=True)
            return
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush(); sys.stderr.flush()
        os.execve(sys.executable, cmd, env)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("GLV_BENCH_BACKEND", "nccl")      # "nccl" is RCCL over xGMI on ROCm
    if world > ndev and backend == "nccl":
        raise SystemExit(f"bench.py --gpus {a.gpus}: rank {rank} sees {ndev} HIP device(s) for {world} ranks -- one rank per GPU, refusing to share devices")
    device = local_rank % ndev          # one rank per GPU in production; the modulo only matters for the
    torch.cuda.set_device(device)       # single-GPU rehearsal of the N>1 code path (GLV_BENCH_BACKEND=gloo)
    # launched by torch.distributed.run: take the distributed code path (process group, barriers, all-reduce,
    # all-gather) even with a single rank, so that the RCCL calls of the N > 1 path are exercised on any box
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ and os.environ.get("GLV_BENCH_DIST", "1") != "0")
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    coll_dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")

    from glava_amd import build as B
    if rank == 0:
        B.build()
    if dist_on:
        dist.barrier()
    from glava_amd import spectrum as G
    from glava_amd.sharding import shard_range, gather_stats

    ops = G.OP_FFT
    if "gravity" in a.ops: ops |= G.OP_GRAVITY
    if "average" in a.ops: ops |= G.OP_AVERAGE
    n, streams = a.n, a.streams
    total_streams = streams * world                       # weak scaling: 64K streams per GPU
    lo, hi = shard_range(total_streams, rank, world)      # contiguous shard, SURVEY.md 8e
    assert hi - lo == streams

    params = G.Params(n=n, log_mode=a.log_mode)
    batch = G.Batch(params, streams, ops, device=device)
    if a.grid: batch.set_grid(a.grid)
    alt_batch = None
    alt_mode = 0 if a.log_mode == 1 else 1
    if a.log_mode in (0, 1) and not a.no_alt:  # secondary measurement: the other log mode of the same pass
        alt_batch = G.Batch(G.Params(n=n, log_mode=alt_mode), streams, ops, device=device)
        if a.grid: alt_batch.set_grid(a.grid)
    gen = torch.Generator(device="cuda"); gen.manual_seed(12345 + lo)     # the PCM of a shard is a function of its first global stream
    d_pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda", generator=gen)
    d_out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def timed(b):
        """Spin-up, W warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides.

        The spin-up (same launches, untimed, `--spinup-s` seconds of wall clock) brings the GPU to its
        sustained clock/power state: the first few dozen launches after idle run ~20 % slower on MI355X
        (ms/step 1.27 at K=3 vs 1.04 at K=50 without it), which would make `value` depend on K."""
        t_end = time.perf_counter() + a.spinup_s
        while time.perf_counter() < t_end:
            for _ in range(8):
                b.process_s16(d_pcm, d_out, ops, stream)
            torch.cuda.synchronize()
        for _ in range(a.warmup):
            b.process_s16(d_pcm, d_out, ops, stream)
        torch.cuda.synchronize()
        if dist_on: dist.barrier()
        torch.cuda.synchronize()
        b.timing_begin()                                   # HIP events on the launch stream, per launch
        t0 = time.perf_counter()
        for _ in range(a.steps):
            b.process_s16(d_pcm, d_out, ops, stream)
        torch.cuda.synchronize()
        if dist_on: dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kms, nl = b.timing_end()
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kms, nl

    elapsed, kernel_ms, launches = timed(batch)
    # the same launch back to back for --sustained-s seconds (N=1 only): the headline is a 13 ms burst after 0.3 s of spin-up, this is
    # what the part holds at its power limit (say which one a claim is made on)
    sustained = None
    if world == 1 and a.sustained_s > 0:
        batch.timing_begin()
        t0 = time.perf_counter(); nl = 0
        while time.perf_counter() - t0 < a.sustained_s:
            for _ in range(128):
                batch.process_s16(d_pcm, d_out, ops, stream)
            nl += 128
            torch.cuda.synchronize()
        s_dt = time.perf_counter() - t0
        s_kms, s_nl = batch.timing_end()
        s_k = s_kms / max(s_nl, 1) * 1e-3
        sustained = {"note": f"{s_nl} back-to-back launches of the headline pass over {s_dt:.2f} s (a synchronize every 128), HIP-event kernel time",
                     "seconds": s_dt, "launches": s_nl, "value": streams * nl / s_dt, "unit": "frames/s", "avg_kernel_ms": s_k * 1e3,
                     "roofline_frac": (batch.algorithmic_bytes(ops) / s_k / 1e9) / HBM_PEAK_GBS if s_k > 0 else 0.0}
    if a.check_dump:
        import numpy as np
        cb_ = G.Batch(G.Params(n=n), 1, G.OP_FFT, device=device)
        raw1 = torch.empty((2, n), dtype=torch.float32, device="cuda")
        cb_.process_s16(d_pcm[:1].contiguous(), raw1, G.OP_FFT | G.OP_RAW, stream)
        torch.cuda.synchronize()
        np.savez(f"{a.check_dump}.rank{rank}.npz", pcm=d_pcm[0].cpu().numpy(), raw=raw1.cpu().numpy(), first_spectrum=d_out[0].cpu().numpy(),
                 global_stream=lo, rank=rank, world=world, streams=streams)
        cb_.close()
    alt = timed(alt_batch) if alt_batch is not None else None
    # tertiary measurement (N=1 only): the chain GLava's spectrum modules actually request --
    # window,fft,gravity,avg (bars/1.frag:12-24) -- i.e. the same pass + gravity + F=5 windowed average
    chain = None
    if world == 1 and ops == G.OP_FFT and not a.no_alt:
        if alt_batch is not None: alt_batch.close(); alt_batch = None
        cops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
        cb = G.Batch(params, streams, cops, device=device)
        c_place = None
        if a.tune_placement:
            try:
                pf, pb = cb.tune_placement(d_pcm, d_out, cops, 6, stream)
                c_place = {"ms_first_placement": pf, "ms_best_placement": pb}
            except Exception as ex:
                c_place = {"error": str(ex)}
        ops_saved, ops = ops, cops
        c_el, c_kms, c_nl = timed(cb)
        ops = ops_saved
        c_k = (c_kms / max(c_nl, 1)) * 1e-3
        c_bytes = cb.algorithmic_bytes(cops)
        chain = {"note": f"fft -> gravity -> average(F=5, windowed), log_mode {a.log_mode}; algorithmic bytes 52*N per frame (SURVEY 8d row C)",
                 "algorithmic_bytes_per_launch": c_bytes, "value": streams * a.steps / c_el, "unit": "frames/s", "ms_per_step": c_el / a.steps * 1e3,
                 "avg_kernel_ms": c_k * 1e3, "roofline_frac": (c_bytes / c_k / 1e9) / HBM_PEAK_GBS if c_k > 0 else 0.0, "placement": c_place}
        cb.close()

    # the same pass with the output as GL_R16 texels (what handle_audio uploads, render.c:521-524): 8N bytes per frame
    r16 = None
    if world == 1 and ops == G.OP_FFT and not a.no_alt:
        rb = G.Batch(params, streams, G.OP_FFT, device=device)
        ops_saved, ops = ops, G.OP_FFT | G.OP_R16
        r_el, r_kms, r_nl = timed(rb)
        ops = ops_saved
        r_k = (r_kms / max(r_nl, 1)) * 1e-3
        r_bytes = rb.algorithmic_bytes(G.OP_FFT | G.OP_R16)
        r16 = {"note": f"window+FFT+magnitude with the output as GL_R16 texels (GLV_OP_R16), log_mode {a.log_mode}; algorithmic bytes 8*N per frame",
               "value": streams * a.steps / r_el, "unit": "frames/s", "ms_per_step": r_el / a.steps * 1e3,
               "avg_kernel_ms": r_k * 1e3, "roofline_frac": (r_bytes / r_k / 1e9) / HBM_PEAK_GBS if r_k > 0 else 0.0}
        rb.close()

    # BASELINE configs[2], configs[4] and the two large stateless sizes, each with its own kernel time and roofline fraction
    # (SURVEY 8d bytes per frame) -- the figures below the headline's that VERDICT r2 wants in the driver's record
    configs = None
    if world == 1 and ops == G.OP_FFT and not a.no_alt and not a.no_configs:
        configs = extra_configs(G, torch, device, a, HBM_PEAK_GBS)

    frames_rank = streams * a.steps
    stats = gather_stats({"frames": frames_rank, "seconds": elapsed, "kernel_ms": kernel_ms,
                          "bytes": batch.algorithmic_bytes(ops) * launches}, world, force=dist_on)
    # which device every rank actually drove (a scaling run on one node must show N distinct devices)
    prop = torch.cuda.get_device_properties(device)
    ident = {"rank": rank, "local_rank": local_rank, "device": device, "uuid": str(getattr(prop, "uuid", "")), "name": prop.name,
             "pci_bus_id": getattr(prop, "pci_bus_id", None), "global_stream_first": lo}
    idents = [ident]
    if dist_on:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    for srec, irec in zip(stats, idents):
        srec.update({k: irec[k] for k in ("device", "uuid", "name", "pci_bus_id", "global_stream_first")})

    if rank == 0:
        total_frames = sum(s["frames"] for s in stats)
        value = total_frames / elapsed
        traffic_now, traffic_detail = (None, {"skipped": "--no-live-traffic, or more than one rank"})
        if a.live_traffic and world == 1:
            torch.cuda.synchronize()
            traffic_now, traffic_detail = live_traffic(a)
        alg_bytes = batch.algorithmic_bytes(ops)           # per launch (all streams of the rank)
        avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
        achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        line = {
            "metric": f"stereo {n}-pt FFT+smooth frames/s; % HBM roofline -- `value`: the window+FFT+magnitude pass of BASELINE configs[1]; the FFT+smooth "
                      f"chains (gravity / average on f32 state, GLava's shipped GL_R16 chain, + its pre-smoothing pass) are in roofline.chains",
            "value": value, "unit": "frames/s", "log_mode": a.log_mode,
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist_on else 1, "distinct_devices": len({(r.get("uuid"), r.get("pci_bus_id"), r.get("device")) for r in stats}),
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{world} MI355X, {streams} batched stereo streams per GPU, N={n} "
                                   f"{'Hann(-like) window+FFT+magnitude' if ops == G.OP_FFT else a.ops}",
                       "streams_per_gpu": streams, "n": n, "ops": a.ops, "log_mode": a.log_mode, "spinup_s": a.spinup_s,
                       "input": "int16 [streams][n][2] resident in HBM", "kernel": batch.kernel_name(),
                       "collectives": (f"{backend} ({'RCCL over xGMI' if backend == 'nccl' else 'CPU rehearsal'}): barrier, all_reduce(MAX) of elapsed, "
                                       f"all_gather of one 32-byte stats record per rank; {world} rank(s)") if dist_on else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_now if traffic_now is not None else measured_traffic(n, streams, a.ops),
                         "traffic_source": ("collected in THIS run: 2*FETCH_SIZE + WRITE_SIZE (KiB, the guide's gfx950 wide-stream correction) of this kernel from two separate "
                                            "rocprofv3 --pmc passes over this script's headline launches (child processes, outside the timed region), mean per dispatch")
                                           if traffic_now is not None else
                                           ("profiles/hbm_traffic.json: 2*FETCH_SIZE + WRITE_SIZE of this kernel from separate rocprofv3 --pmc passes "
                                            "(tools/profile.sh) of the same command on an MI355X -- a committed measurement, NOT collected in this run"),
                         "traffic_live": traffic_detail, "traffic_committed": measured_traffic(n, streams, a.ops),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_ms": avg_kernel_s * 1e3,
                         "kernel": batch.kernel_name()},
        }
        ceil = streaming_ceiling(n, a.ops)
        if ceil:
            line["roofline"]["streaming_ceiling"] = {"value": ceil, "unit": "GB/s", "frac": achieved / ceil,
                                                     "note": "best do-nothing kernel for the same 1:2 read/write mix, tools/membench.hip"}
        if alt is not None:
            a_el, a_kms, a_nl = alt
            a_k = (a_kms / max(a_nl, 1)) * 1e-3
            key, note = (("strict_log", "same pass with log_mode 0 (fp64 table log, bit-faithful to the reference's float on every input)")
                         if alt_mode == 0 else
                         ("fast_log", "same pass with log_mode 1 (hardware log2, <= 1.8e-7 relative on every float; the parity bar is 1e-5)"))
            line[key] = {"note": note,
                                "value": streams * world * a.steps / a_el, "unit": "frames/s",
                                "ms_per_step": a_el / a.steps * 1e3, "avg_kernel_ms": a_k * 1e3,
                                "roofline_frac": (alg_bytes / a_k / 1e9) / HBM_PEAK_GBS if a_k > 0 else 0.0}
        if sustained is not None:
            line["sustained"] = sustained
        line["stats"] = stats                              # one record per rank, as gathered (frames, seconds, kernel_ms, bytes)
        if chain is not None:
            line["smooth_chain"] = chain
        if r16 is not None:
            line["r16_texels"] = r16
        if configs is not None:
            line["configs"] = configs
        # BASELINE.json's metric is "FFT+smooth": every smoothing chain measured in this run, one record each, INSIDE `roofline` (the key the
        # driver's parser keeps).  frac = bytes_per_frame x frames_per_launch / avg_kernel_ms / 8 TB/s on the chain's own algorithmic bytes.
        def chain_rec(e, frames, what):
            if not e or "avg_kernel_ms" not in e: return None
            by = e.get("algorithmic_bytes_per_launch", alg_bytes)
            rec = {"what": what, "frames_per_s": e["value"], "ms_per_step": e["ms_per_step"], "avg_kernel_ms": e["avg_kernel_ms"],
                   "bytes_per_frame": by / frames, "frac": e["roofline_frac"], "launches_per_step": e.get("launches_per_step", 1)}
            if "frac_of_28N" in e: rec["frac_of_28N"] = e["frac_of_28N"]
            if e.get("placement"): rec["placement"] = e["placement"]
            return rec
        chains = {"smooth_chain": chain_rec(chain, streams, "fft -> gravity -> average (F=5) on f32 state, 52 N B/frame (the CPU path's transform list: bars/1.frag:12-24)"),
                  "strict_log": chain_rec(line.get("strict_log"), streams * world, "the headline pass with the bit-faithful fp64 log (log_mode 0), 12 N B/frame")}
        if configs is not None and "gl_default" in configs:
            g = configs["gl_default"]
            chains["gl_default"] = chain_rec(g, streams, "GLava's shipped accel chain (render.c:2188-2265): upload, GL_MAX + gravity, ring, average on GL_R16 state -> `av` texels, one launch, 28 N B/frame")
            chains["gl_default_bars"] = chain_rec(g.get("bars_out"), streams, "... + the 80 bars of the bars / radial modules in the same launch")
            chains["gl_default_bars_live"] = chain_rec(g.get("bars_out_live"), streams, "... the same with GLV_OP_BARS_ONLY (live bins only); frac on ITS OWN bytes")
            chains["sm_out"] = chain_rec(g.get("sm_out"), streams, "... + the pre-smoothing pass (render.c:2277-2303) -> `sm` texels: what GLava ships end to end, two launches, 28 N B/frame")
            chains["sm_out_live"] = chain_rec(g.get("sm_out_live"), streams, "the same `sm` texels with the state kept only for the bins the pass samples (GLV_OP_BARS_ONLY); frac on ITS OWN bytes, frac_of_28N for comparison")
        line["roofline"]["chains"] = {k: v for k, v in chains.items() if v}
        try:
            mism, shifted = batch.window_selftest()
            line["window_selftest"] = {"mismatches": mism, "shifted_positions": shifted,
                                       "note": f"every (s16 sample value, window position) pair of N={n}: the kernels' float-pair window product "
                                               "against the reference's float x double -> double -> float product, on the device, outside the timed region"}
        except Exception as ex:                    # a diagnostic, never a reason to lose the line
            line["window_selftest"] = {"error": str(ex)}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, a.cpu_seconds)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    batch.close()
    if alt_batch is not None: alt_batch.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
