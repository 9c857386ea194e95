"""The N > 1 path of the training step on the CPU (gloo, world_size 2): bucketed gradient averaging and the BatchNorm-buffer
broadcast of speinet_amd.trainer reproduce the single-process full-batch step (what nn.DataParallel computes in the reference,
model/__init__.py:19-20).  The model here is a small torch module — the collectives are model-agnostic; the HIP model itself
is exercised by tests/test_gpu_train.py."""
import os
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Net(torch.nn.Sequential):
    """conv - BN - relu - conv, plus a parameter no forward ever uses (as SearchTransfer.search1/2 in the reference model)."""

    def __init__(self):
        super().__init__(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(), torch.nn.Conv2d(8, 3, 3, padding=1))
        self.unused = torch.nn.Parameter(torch.ones(5))


def _net():
    torch.manual_seed(3)
    return _Net()


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    from speinet_amd.trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _net()
        g = torch.Generator().manual_seed(11)
        x, y = torch.randn(4, 3, 12, 12, generator=g), torch.randn(4, 3, 12, 12, generator=g)
        share = slice(rank * 2, rank * 2 + 2)
        tr = Trainer(net, torch.nn.L1Loss(), lr=1e-3, weight_decay=0.1)
        # tiny buckets: several all-reduces, parameters split across them
        import speinet_amd.trainer as T
        orig = T.allreduce_gradients
        T.allreduce_gradients = lambda params, group=None: orig(params, group, bucket_bytes=256)
        tr.step(x[share], y[share])
        assert net.unused.grad is None, "a parameter without a gradient on any rank must keep grad None (Adam skips it)"
        from speinet_amd.trainer import seed_rank
        assert seed_rank(100) == 100 + rank
        ret[rank] = {k: v.clone() for k, v in net.state_dict().items()}
    finally:
        dist.destroy_process_group()


def test_two_rank_step_matches_full_batch_step():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    for k in a:
        assert torch.equal(a[k], b[k]), f"ranks disagree on {k} after the step"
    # single process, whole batch, but BatchNorm statistics per half (what the replicas of nn.DataParallel see) and the running
    # buffers of the first half only
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(4, 3, 12, 12, generator=g), torch.randn(4, 3, 12, 12, generator=g)
    halves = [_net(), _net()]                     # the two replicas: same weights, each sees its half of the batch
    for r, rep in enumerate(halves):
        torch.nn.L1Loss()(rep(x[r * 2:r * 2 + 2]), y[r * 2:r * 2 + 2]).backward()
    net = halves[0]                               # replica 0's buffers are the ones that persist
    with torch.no_grad():
        for p0, p1 in zip(net.parameters(), halves[1].parameters()):
            if p0.grad is not None:
                p0.grad = 0.5 * (p0.grad + p1.grad)
    torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=0.1).step()
    ref = net.state_dict()
    assert torch.equal(a["unused"], torch.ones(5)), "weight decay must not touch a parameter that never had a gradient"
    for k in a:
        if "num_batches" in k:
            continue
        assert torch.allclose(a[k], ref[k], atol=1e-6, rtol=1e-5), k
