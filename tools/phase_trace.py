#!/usr/bin/env python3
"""Poor man's phase trace: cycles wave 0 of workgroup 0 spends in each phase of the row loop (libglvtune built
with -DGLV_EXP_PHASETIME, one variant).  python tools/phase_trace.py <lib> <streams> [log_mode]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from glava_amd import spectrum as G
G.lib()
T = C.CDLL(sys.argv[1]); streams = int(sys.argv[2]); lm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
T.glv_tune_describe.restype = C.c_char_p
T.glv_tune_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
n = 2 << T.glv_tune_log_nn()
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
ms = C.c_float(0)
buf = (C.c_ulonglong * 32)()
T.glv_tune_run(0, pcm.data_ptr(), out.data_ptr(), streams, lm, 0, 3, None, C.byref(ms)); torch.cuda.synchronize()
T.glv_tune_phase_read(buf)
T.glv_tune_run(0, pcm.data_ptr(), out.data_ptr(), streams, lm, 0, 5, None, C.byref(ms)); torch.cuda.synchronize()
T.glv_tune_phase_read(buf)
rows = max(buf[17], 1)
P = -(-T.glv_tune_log_nn() // int(sys.argv[4])) if len(sys.argv) > 4 else None
if P is None:                       # passes: slots 1..P hold compute, 1+P.. hold the exchanges (P-1 of them)
    nz = [i for i in range(1, 14) if buf[i]]
    P = (len(nz) + 1) // 2
names = {0: "A issue PCM loads", 14: "W vmcnt(0)", 15: "D epilogue", 16: "C unpack+window", 18: "loop overhead"}
for q in range(P): names[1 + q] = f"compute pass {q}"
for q in range(P - 1): names[1 + P + q] = f"exchange after pass {q} (write, sync, gather, sync, read)"
tot = sum(buf[i] for i in names)
print(f"{T.glv_tune_describe(0).decode()}  N={n} streams={streams} log={lm}: {ms.value:.3f} ms/launch, {rows} rows by the traced wave, {tot/rows:.0f} cycles/row (s_memtime, 100 MHz ref => x clk ratio)")
for i in sorted(names):
    print(f"  {names[i]:58s} {buf[i]/rows:9.0f} cycles/row  {100*buf[i]/max(tot,1):5.1f} %")
