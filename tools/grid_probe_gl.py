import sys, time, torch
sys.path.insert(0, '/root/repo')
from glava_amd import spectrum as G
n, F, streams = 4096, 5, 16384
p = G.Params(n=n, avg_frames=F, avg_window_kind=1, gl_storage=1)
ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE | G.OP_R16
pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
out = torch.empty((streams, 2, n), dtype=torch.int16, device="cuda")
b = G.Batch(p, streams, G.OP_GRAVITY | G.OP_AVERAGE)
for grid in (0, 1024, 512, 384, 256, 128):
    b.set_grid(grid)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(4): b.process_s16(pcm, out, ops)
        torch.cuda.synchronize()
    b.timing_begin()
    for _ in range(60): b.process_s16(pcm, out, ops)
    torch.cuda.synchronize()
    ms, nl = b.timing_end()
    print(f"grid {grid:5d} (last {b.last_grid()}): {ms / nl:.4f} ms per launch")
b.close()
