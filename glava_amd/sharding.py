"""Multi-GPU layout of the batched path: streams are independent, so GPU g of G
owns the contiguous shard [g*B/G, (g+1)*B/G) -- PCM, state and spectra of those streams live
only on that GPU -- and there is NO data-path collective.  The only communication is one
all-gather of a fixed stats record per rank (RCCL over xGMI when the backend is "nccl";
gloo in the CPU tests), latency-bound at 32 bytes per rank.
"""
from __future__ import annotations

STAT_FIELDS = ("frames", "seconds", "kernel_ms", "bytes")


def shard_range(total_streams: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced partition; the first (total % world) ranks get one extra stream."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(total_streams, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_stats(record: dict, world: int, force: bool = False) -> list[dict]:
    """All-gather one {frames, seconds, kernel_ms, bytes} record per rank; returns the list on
    every rank (world == 1: no process group needed unless `force` asks for the collective anyway)."""
    if world == 1 and not force:
        return [dict(record)]
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([float(record[k]) for k in STAT_FIELDS], dtype=torch.float64, device=dev)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    return [{k: float(t[i].item()) for i, k in enumerate(STAT_FIELDS)} for t in allr]
