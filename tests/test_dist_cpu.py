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
    import datetime
    # (a bounded rendezvous: a port that something else holds -- or a peer that died -- is an exception here within
    # seconds, not the default half-hour wait)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=30))
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


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_two_ranks(port, join_s=120):
    """-> (exit codes, result or None).  Never hangs: once a rank has FAILED the others get five more seconds (a peer of a
    rank that could not open the store would otherwise sit in the rendezvous), and whatever is alive after `join_s` is
    killed; a killed rank is reported as None."""
    import queue
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    deadline = time.time() + join_s
    while time.time() < deadline and any(p.is_alive() for p in procs):
        if any(p.exitcode not in (None, 0) for p in procs):
            deadline = min(deadline, time.time() + 5)
        time.sleep(0.2)
    codes = []
    for p in procs:
        if p.is_alive():
            p.kill()
            p.join(10)
            codes.append(None)
        else:
            codes.append(p.exitcode)
    try:
        res = q.get(timeout=5) if codes == [0, 0] else None
    except queue.Empty:
        res = None
    return codes, res


def test_two_rank_sharding_reproduces_full_batch():
    from oracle import pyoracle as po
    po.build()
    codes, res = _run_two_ranks(_free_port())  # (a port of the OS's choosing: 29500 + pid could collide with a neighbour)
    assert codes == [0, 0], f"rank exit codes {codes} (None = hung and killed)"
    assert res is True


def test_taken_port_fails_loudly_instead_of_hanging():
    """with the rendezvous port held by somebody else the run must END with an error within seconds
    (rank 0 cannot bind its store: EADDRINUSE; its peer, which would wait on the foreign socket, is reaped by the
    harness), not sit in the rendezvous."""
    import socket
    import time
    from oracle import pyoracle as po
    po.build()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        sk.listen(1)  # held for the duration of the run; it never speaks the store's protocol
        t0 = time.time()
        codes, res = _run_two_ranks(sk.getsockname()[1], join_s=90)
    took = time.time() - t0
    assert res is None and codes[0] not in (0, None) and codes[1] != 0, f"exit codes {codes} after {took:.0f} s"
    assert took < 60, f"the failure took {took:.0f} s to surface"
