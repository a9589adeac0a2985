"""world_size-2 gloo test of the N>1 path: roots are sharded with (global_batch, root_offset), no
collective on the data path, outputs gathered in root order == the un-sharded run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from muax_amd import gather_roots, shard_roots
    from oracle import pyoracle as po
    Bg, obs_dim, E, A, S = 37, 4, 8, 2, 12
    w = po.random_mlp_weights(5, obs_dim, E, A, 21, bias_scale=0.1)
    rng = np.random.default_rng(9)
    obs = rng.uniform(-1, 1, (Bg, obs_dim)).astype(np.float32)
    noise = rng.dirichlet([0.3] * A, Bg).astype(np.float32)
    off, cnt = shard_roots(Bg, world, rank)
    # the CPU oracle stands in for the device kernels here: what is under test is the sharding contract
    out = po.act_mlp(po.Mlp(w, obs_dim, E, A, 21), po.SearchCfg(S, tiebreak=1, global_batch=Bg, root_offset=off),
                     obs[off:off + cnt], [4, 2], noise[off:off + cnt], 0.25)
    action = gather_roots(torch.from_numpy(out["action"]), Bg)
    weights = gather_roots(torch.from_numpy(out["action_weights"]), Bg)
    visits = gather_roots(torch.from_numpy(out["tree"].children_visits), Bg)
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing
    if rank == 0:
        full = po.act_mlp(po.Mlp(w, obs_dim, E, A, 21), po.SearchCfg(S, tiebreak=1), obs, [4, 2], noise, 0.25)
        ok = (np.array_equal(full["action"], action.numpy()) and np.array_equal(full["action_weights"], weights.numpy())
              and np.array_equal(full["tree"].children_visits, visits.numpy()) and float(t) == world)
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_reproduces_full_batch():
    from oracle import pyoracle as po
    po.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
