"""CPU, world_size 2, gloo: the N>1 bookkeeping of bench.py (stream sharding, barrier, max-over-ranks)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from x264vfw_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard.stream_ids(rank, world, 4)
    dist.barrier()
    mx = shard.max_over_ranks(1.0 + rank, dist)          # rank 1 is the slow one
    q.put((rank, ids, mx, shard.aggregate_fps(4, 10, world, mx)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3] and res[1][1] == [4, 5, 6, 7]            # disjoint, complete
    assert all(abs(r[2] - 2.0) < 1e-12 for r in res)                           # max over ranks
    assert all(abs(r[3] - 4 * 10 * 2 / 2.0) < 1e-9 for r in res)              # whole-job frames/s


def test_split_groups():
    assert shard.split_groups(16, 2) == [8, 8]
    assert shard.split_groups(5, 4) == [2, 1, 1, 1]
    assert shard.split_groups(3, 8) == [1, 1, 1]
    assert sum(shard.split_groups(257, 4)) == 257
    assert shard.stream_seed(0x264, 7) == 0x264 + 7
