"""CPU suite: the N>1 rank logic (shard -> local work -> all-gather of metrics -> max time) on world_size-2 gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speinet_amd.dist import gather_metrics, max_over_ranks, shard_clips_by_length, shard_units


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    units = shard_units(7, rank, world)
    # "deblur" each unit: a deterministic stand-in metric per frame
    local = torch.tensor([sum(30.0 + u for u in units), float(len(units)), float(sum(units))], dtype=torch.float64)
    allm = gather_metrics(local, dist)
    t = max_over_ranks(1.0 + rank, torch.device("cpu"), dist)
    dist.barrier()
    q.put((rank, units, allm.tolist(), t))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    for _, _, allm, t in res:
        assert allm == res[0][2]                       # every rank holds the same gathered table
        assert sum(r[1] for r in allm) == 7            # all frames accounted for exactly once
        assert abs(sum(r[0] for r in allm) - sum(30.0 + u for u in range(7))) < 1e-9
        assert t == 2.0                                # max over ranks


def test_single_process_degenerates():
    m = gather_metrics(torch.tensor([1.0, 2.0]))
    assert m.shape == (1, 2)
    assert max_over_ranks(3.5, torch.device("cpu")) == 3.5


def test_clip_balancing():
    shards = shard_clips_by_length([150, 100, 100, 50, 40, 10], 2)
    assert sorted(sum(shards, [])) == list(range(6))
    loads = [sum([150, 100, 100, 50, 40, 10][i] for i in s) for s in shards]
    assert abs(loads[0] - loads[1]) <= 50
