import sys, time, torch
sys.path.insert(0, ".")
from glava_amd import spectrum as G
n, streams = 4096, 65536
ops = G.OP_FFT | G.OP_GRAVITY | G.OP_AVERAGE
def measure(tag, pad_mb, use_stream):
    pads = [torch.empty(int(pad_mb * 1024 * 1024), dtype=torch.uint8, device="cuda")] if pad_mb else []
    pcm = torch.randint(-32768, 32768, (streams, n, 2), dtype=torch.int16, device="cuda")
    out = torch.empty((streams, 2, n), dtype=torch.float32, device="cuda")
    b = G.Batch(G.Params(n=n), streams, ops)
    st = torch.cuda.current_stream().cuda_stream if use_stream else None
    call = (lambda: b.process_s16(pcm, out, ops, st)) if use_stream else (lambda: b.process_s16(pcm, out, ops))
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(8): call()
        torch.cuda.synchronize()
    b.timing_begin()
    for _ in range(40): call()
    torch.cuda.synchronize()
    ms, nl = b.timing_end()
    print(f"{tag}: pad {pad_mb} MiB stream {use_stream}: {ms / nl:.4f} ms  pcm@{pcm.data_ptr():#x} out@{out.data_ptr():#x}", flush=True)
    b.close(); del pcm, out, pads
    torch.cuda.empty_cache()
for rep in range(2):
    measure("a", 0, False); measure("b", 0, True); measure("c", 1.37, False); measure("d", 33.1, True); measure("e", 257.3, False)
