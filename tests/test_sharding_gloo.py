"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- contiguous stream shards, no data-path
collective, one all-gather of the stats record, max-over-ranks timing."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from glava_amd.sharding import shard_range
    for total in (1, 7, 64, 65536, 524288, 1000003):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(524288, 3, 8) == (3 * 65536, 4 * 65536)     # configs[3]
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from glava_amd.sharding import gather_stats, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(1001, rank, world)
    rec = {"frames": (hi - lo) * 3, "seconds": 0.5 + rank, "kernel_ms": 10.0 * (rank + 1), "bytes": (hi - lo) * 49152}
    stats = gather_stats(rec, world)
    t = torch.tensor([rec["seconds"]], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, stats, float(t.item())))
    dist.destroy_process_group()


def test_stats_gather_world2_gloo():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, stats, tmax in res:
        assert tmax == 1.5                                   # max over ranks
        assert [s["frames"] for s in stats] == [501 * 3, 500 * 3]
        assert sum(s["frames"] for s in stats) == 1001 * 3   # whole-job aggregate
        assert stats[1]["kernel_ms"] == 20.0 and stats[0]["bytes"] == 501 * 49152
