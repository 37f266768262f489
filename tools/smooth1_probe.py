import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from glava_amd import spectrum as G
for n in (1024, 4096, 16384):
    st = G.State(G.Params(n=n))
    x = np.abs(np.random.default_rng(1).standard_normal(n)).astype(np.float32)
    for _ in range(3): st.smooth(x.copy())
    t0 = time.perf_counter()
    for _ in range(20): st.smooth(x.copy())
    print(n, "single-row glv_smooth: %.0f us per call" % ((time.perf_counter() - t0) / 20 * 1e6))
    st.close()
