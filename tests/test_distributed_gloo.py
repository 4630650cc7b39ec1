"""World-size-2 coverage of the multi-GPU path on CPU (gloo): envs shard by global index with no
data-path collective, and the one exchange -- the all-gather of per-env episode returns -- reassembles
exactly what a single process computes.  The CPU oracle stands in for the HIP kernel (same seeds, same
global-index keyed RNG), so this exercises crowdnav.rollout's sharding + gather code."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

N_TOTAL, STEPS = 8, 40


def _actions():
    rng = np.random.default_rng(123)
    return np.stack([rng.uniform(0, 0.22, (STEPS, N_TOTAL)), rng.uniform(-2, 2, (STEPS, N_TOTAL))], 2).astype(np.float32)


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crowdnav.rollout import gather_returns, shard_range
    from oracle import oracle
    base, n = shard_range(N_TOTAL, rank, world)
    o = oracle.Oracle(n_envs=n, env_index_base=base, max_steps=15, seed=77)
    o.reset()
    acts = _actions()
    done_count = 0
    for t in range(STEPS):
        _, _, d, _ = o.step(acts[t, base:base + n].astype(np.float64), auto_reset=True)
        done_count += int(d.sum())
    local = torch.from_numpy(o.returns().astype(np.float32))
    allr = gather_returns(local)
    total = torch.tensor([done_count]); dist.all_reduce(total)
    if rank == 0:
        q.put((allr.numpy(), int(total.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_return_gather():
    from oracle import oracle
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    allr, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over all envs
    o = oracle.Oracle(n_envs=N_TOTAL, env_index_base=0, max_steps=15, seed=77)
    o.reset()
    acts = _actions()
    dc = 0
    for t in range(STEPS):
        _, _, d, _ = o.step(acts[t].astype(np.float64), auto_reset=True)
        dc += int(d.sum())
    assert total == dc and dc >= N_TOTAL
    assert np.array_equal(allr, o.returns().astype(np.float32))


def test_shard_range():
    from crowdnav.rollout import shard_range
    assert [shard_range(16384, r, 8) for r in (0, 7)] == [(0, 2048), (14336, 2048)]
    with pytest.raises(AssertionError):
        shard_range(10, 0, 4)
