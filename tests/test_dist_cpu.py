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


# ---- the one-command launcher (bench.py --gpus N, python -m speinet_amd.inference --n_GPUs N) -------------------------------------
_RANK_SCRIPT = """
import os, sys, json
import torch, torch.distributed as dist
dist.init_process_group("gloo")
from speinet_amd.dist import gather_metrics, max_over_ranks
r, w = dist.get_rank(), dist.get_world_size()
assert (r, w) == (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
allm = gather_metrics(torch.tensor([float(r), 1.0], dtype=torch.float64), dist)
t = max_over_ranks(1.0 + r, torch.device("cpu"), dist)
if r == 0:
    print(json.dumps({"n_gpus": w, "ranks": allm[:, 0].tolist(), "t": t, "argv": sys.argv[1:], "master": os.environ["MASTER_ADDR"]}))
dist.destroy_process_group()
sys.exit(int(sys.argv[1]) if r == w - 1 else 0)
"""


def test_rank_command_shape():
    from speinet_amd.dist import rank_command
    c = rank_command("bench.py", ["--gpus", "4", "--steps", "3"], 4, 29999, python="python3")
    assert c[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in c and "--nproc-per-node=4" in c
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[c.index("--master-port") + 1] == "29999"
    assert c[-5:] == ["bench.py", "--gpus", "4", "--steps", "3"]
    m = rank_command("speinet_amd.inference", ["--n_GPUs", "2"], 2, 1, python="python3", module=True)
    assert m[-4:] == ["-m", "speinet_amd.inference", "--n_GPUs", "2"]


def test_launch_ranks_two_gloo_ranks(tmp_path, capfd):
    """The parent starts two ranks as children, rank 0's line comes through its stdout, the worst exit code comes back."""
    import json
    import sys
    from speinet_amd.dist import launch_ranks
    script = tmp_path / "rank_script.py"
    script.write_text(_RANK_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), RANK="7", WORLD_SIZE="9")   # stale rank env must not leak
    rc = launch_ranks(str(script), ["0", "--flag"], 2, need_gpus=False, env=env)
    out = capfd.readouterr().out
    assert rc == 0, out
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [0.0, 1.0] and line["t"] == 2.0
    assert line["argv"] == ["0", "--flag"] and line["master"] == "127.0.0.1"
    assert launch_ranks(str(script), ["3"], 2, need_gpus=False, env=env) != 0        # a failing rank fails the launch


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` on a node with fewer than 2 GPUs is an error (exit code 2, nothing printed on stdout) — never a
    1-GPU line; the parent decides that without initialising the GPU."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("node has 2+ GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for cmd in (["bench.py", "--gpus", "2", "--steps", "1"], ["bench.py", "--train", "--gpus", "2"],
                ["-m", "speinet_amd.inference", "--n_GPUs", "2", "--data_path", "x", "--result_path", "y", "--model_path", "synthetic"]):
        p = subprocess.run([sys.executable, *cmd], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 2, (cmd, p.returncode, p.stderr[-500:])
        assert p.stdout.strip() == "" and "ranks requested" in p.stderr
