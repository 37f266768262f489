#!/bin/bash
# round 4: the persistent-workgroup count after the grid rule change (default_grid, Tuned<>::rounds)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --sustained-s 0 2>&1 | tail -1 | cut -c1-400
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --sustained-s 0 --grid 1024 2>&1 | tail -1 | cut -c1-400
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --sustained-s 0 2>&1 | tail -1 | cut -c1-400
python tools/grid_ab.py 8192 256 512 1024
python tools/grid_ab.py 16384 512 1024 2048
python tools/grid_ab.py 512 2048 4096 8192 16384
python - <<'PY'
import sys, time, torch
sys.path.insert(0, "."); 
from glava_amd import spectrum as G
# smaller batches: the wgs/4 rule against one round / rounds x round
for n, streams in ((4096, 16384), (4096, 8192), (4096, 4096), (1024, 32768), (2048, 16384)):
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n), streams, G.OP_FFT)
    b.process_s16(pcm, out, G.OP_FFT); torch.cuda.synchronize()
    dflt = b.last_grid()
    res = {}
    for rep in range(3):
        for g in sorted({dflt, 512, 1024, 2048, 4096}):
            b.set_grid(g)
            for _ in range(5): b.process_s16(pcm, out, G.OP_FFT)
            torch.cuda.synchronize(); b.timing_begin()
            for _ in range(50): b.process_s16(pcm, out, G.OP_FFT)
            torch.cuda.synchronize(); ms, nl = b.timing_end()
            res.setdefault(g, []).append(ms / nl)
    print(f"N={n} streams={streams} default={dflt} " + "  ".join(f"{g}: " + "/".join(f"{x:.4f}" for x in v) for g, v in res.items()), flush=True)
    b.close()
PY
} > gpurun_out/grid2.txt 2>&1
tail -40 gpurun_out/grid2.txt
