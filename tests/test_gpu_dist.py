"""The N > 1 code path with the PRODUCT kernels, on the one GPU a test box has: two ranks (two processes) share
device 0 and talk over gloo.  No scaling number comes out of this -- it proves that the launcher contract of
bench.py, the (global_batch, root_offset) plumbing of act() and the one-all-reduce gradient mean of update() run
with the real HIP kernels behind them (tests/test_dist_cpu.py covers the same contracts with the oracle standing in)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


def _model(mx, seed=3):
    g = torch.Generator().manual_seed(seed)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                          mx.nn.Dynamic(8, 2, 21, generator=g))
    m = mx.MuZero(net, optimizer=mx.optimizers.create_optimizer("sgd", 0.1), device="cuda:0")
    m.init(0, np.zeros((1, 4)))
    return m


def _batch(mx, B=64, L=5, seed=0):
    rng = np.random.default_rng(seed)
    return mx.Transition(obs=rng.uniform(-1, 1, (B, L, 4)).astype(F32), a=rng.integers(0, 2, (B, L)),
                         r=rng.uniform(0, 1, (B, L)).astype(F32), Rn=rng.uniform(0, 20, (B, L)).astype(F32),
                         pi=rng.dirichlet([1, 1], (B, L)).astype(F32).reshape(B, L, 1, 2))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import muax_amd as mx
    ok = {}
    # ---- act(): each rank searches its contiguous shard of one 96-root batch with the fused kernel
    Bg = 96
    obs = np.random.default_rng(5).uniform(-1, 1, (Bg, 4)).astype(F32)
    m = _model(mx)
    off, cnt = mx.shard_roots(Bg, world, rank)
    a, pi, v = m.act(11, obs[off:off + cnt], with_pi=True, with_value=True, obs_from_batch=True, num_simulations=25,
                     device_outputs=True, global_batch=Bg, root_offset=off)
    # gloo moves host tensors: gather on the CPU (the act path itself has no collective)
    ga = mx.gather_roots(a.cpu(), Bg)
    gpi = mx.gather_roots(pi.cpu(), Bg)
    gv = mx.gather_roots(v.cpu(), Bg)
    if rank == 0:
        fa, fpi, fv = m.act(11, obs, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=25)
        ok["act"] = bool(np.array_equal(fa, ga.numpy()) and np.array_equal(fpi, gpi.numpy()) and np.array_equal(fv, gv.numpy()))
    # ---- update(): fused loss+grad kernel on the local half, ONE flat all-reduce, identical weights afterwards
    b = _batch(mx)
    half = mx.Transition(**{k: (v_[rank * 32:(rank + 1) * 32] if isinstance(v_, np.ndarray) else v_)
                            for k, v_ in b.__dict__.items()})
    loss = m.update(half, backend="hip")["loss"]
    flat = torch.cat([p.detach().reshape(-1) for mod in m.network for p in mod.parameters()]).cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    if rank == 0:
        ok["same_weights"] = bool(torch.equal(both[0], both[1]))
        full = _model(mx)
        full_loss = None
        # single-process step on the whole batch (update() finds no process group partner: world of one is
        # emulated by calling the kernel wrapper directly and applying the same SGD step)
        from muax_amd import loss as mz_loss
        fg = mz_loss.FusedLossGrad(full)
        full_loss, g = fg(b)
        want = torch.cat([p.detach().reshape(-1) for p in fg.params]) - 0.1 * g
        got = torch.cat([p.detach().reshape(-1) for p in m._fused_train.params])
        ok["dp_step"] = bool(torch.allclose(got, want, rtol=2e-5, atol=2e-6))
        ok["losses"] = (float(loss), float(full_loss))
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_shard_act_and_average_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok = q.get(timeout=5)
    assert ok["act"] and ok["same_weights"] and ok["dp_step"], ok


def test_bench_launcher_contract_with_two_ranks_on_one_gpu():
    """python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 (as the driver launches it), both ranks
    on device 0 (MUAX_BENCH_SINGLE_DEVICE) over gloo: one JSON line from rank 0, whole-job value over 2 x 4096 roots."""
    env = dict(os.environ, MUAX_BENCH_SINGLE_DEVICE="1", MUAX_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29400 + os.getpid() % 500
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-config45"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["scaling"] == "weak" and line["unit"] == "env-steps/s"
    assert line["value"] > 1e6 and abs(line["value"] - 2 * 4096 * 10 / (line["ms_per_step"] * 10e-3)) < 0.01 * line["value"]
    assert "cpu_baseline" not in line and line["roofline"]["frac"] > 0


def test_bench_plain_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form of the driver's N = 1 command): bench.py starts
    the two ranks itself; on this 1-GPU box both sit on device 0 (MUAX_BENCH_SINGLE_DEVICE) over gloo.  One JSON line,
    n_gpus 2, the ranks' devices and the backend named in config -- and the two multi-GPU configurations of BASELINE
    as sub-objects measured by BOTH ranks: config 4 (each rank a 128-root shard of the 1024-root batch) and config 5
    (the gradient all-reduce inside the timed update(), reported separately as ms_allreduce).  (The recurrent
    kernel's pair mode wants a GPU to itself: two processes sharing the device keep one workgroup per root.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MUAX_BENCH_SINGLE_DEVICE"] = "1"
    env["MZS_TOWER_PAIR"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3",
                          "--cfg4-sims", "16", "--cfg4-acts", "1", "--cfg5-iters", "5"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["scaling"] == "weak"
    assert line["config"]["launcher"] == "self" and line["config"]["backend"] == "gloo"
    assert len(line["config"]["ranks"]) == 2 and all("cuda:0" in r for r in line["config"]["ranks"])
    assert abs(line["value"] - 2 * 4096 * 10 / (line["ms_per_step"] * 10e-3)) < 0.01 * line["value"]
    assert line["value_pipelined"] > 0  # (two ranks share one GPU here: no ordering between the two rates)
    assert "api" not in line and "cpu_baseline" not in line
    c4 = line["config4_atari"]
    assert "error" not in c4, c4
    assert c4["n_gpus"] == 2 and c4["global_batch"] == 1024 and c4["roots_per_gpu"] == 128 and len(c4["ranks"]) == 2
    assert c4["backend"] == "gloo" and c4["value"] > 0 and c4["roofline"]["bound"] == "mfma"
    c5 = line["config5_gumbel_train"]
    assert "error" not in c5, c5
    u = c5["update"]
    assert c5["n_gpus"] == 2 and u["allreduce_in_timed_update"] is True and u["weights_identical_on_all_ranks_after"] is True
    assert u["ms_per_update"] > 0 and u["ms_per_update_without_allreduce"] > 0 and "ms_allreduce" in u
    assert u["ms_allreduce_alone"] > 0 and u["allreduce_bytes"] == 4 * 1564  # the default trio's 18 arrays, one message


def _check_eight_rank_line(line, launcher):
    assert line["n_gpus"] == 8 and line["config"]["launcher"] == launcher and line["config"]["backend"] == "gloo"
    ranks = line["config"]["ranks"]
    assert len(ranks) == 8 and [r.split(":")[0] for r in ranks] == [f"rank {i}" for i in range(8)]
    # the device ordinal every rank would take on a real node, as text: LOCAL_RANK i -> cuda:i
    for i, r in enumerate(ranks):
        assert f"LOCAL_RANK={i} " in r and f"cuda:{i} on a real node" in r and r.startswith(f"rank {i}: cuda:0 "), r
    c4, c5 = line["config4_atari"], line["config5_gumbel_train"]
    assert "error" not in c4 and "error" not in c5, (c4, c5)
    assert c4["n_gpus"] == 8 and c4["global_batch"] == 1024 and c4["roots_per_gpu"] == 128 and len(c4["ranks"]) == 8
    assert c4["pair_mode_wanted"] is True and c4["pair_mode_survived"] in (True, False)
    assert c5["n_gpus"] == 8 and c5["update"]["weights_identical_on_all_ranks_after"] is True
    assert c5["update"]["allreduce_in_timed_update"] is True and c5["update"]["ms_allreduce_alone"] > 0


def test_bench_eight_ranks_under_torch_distributed_run_on_one_gpu():
    """The DRIVER's form of the 8-GPU job -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 8 ...` -- dry-run on this box's one device (round 5 ran only the
    self-launch form at world 8): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher, every rank reports
    the ordinal LOCAL_RANK gives it on a real node, all three legs report n_gpus 8."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MZS_TOWER_PAIR")}
    env["MUAX_BENCH_SINGLE_DEVICE"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "8", "--steps", "5", "--warmup", "2", "--cfg4-sims", "10", "--cfg4-acts", "1",
                          "--cfg5-iters", "3", "--no-cpu-baseline"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    _check_eight_rank_line(line, "torch.distributed.run")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_8ranks_1gpu_torchrun.json"), "w") as f:
        f.write(lines[0] + "\n")


def test_bench_eight_rank_dry_run_on_one_gpu():
    """`python bench.py --gpus 8` as the driver's 8-GPU run will start it, dry-run on this box's one device
    (MUAX_BENCH_SINGLE_DEVICE: eight processes on device 0 over gloo) -- eight library loads, eight handles, eight
    pair-mode scratch allocations competing for the same CUs (pair mode is left ON: with eight processes on one GPU
    a rendezvous may be lost, and then the automatic fall-back to one workgroup per root is what gets exercised;
    `pair_mode_survived` says which), the rendezvous on a free port, and all three legs reporting n_gpus 8.  The
    driver's first world-8 run must not be the first world-8 execution ever."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MZS_TOWER_PAIR")}
    env["MUAX_BENCH_SINGLE_DEVICE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "10", "--warmup", "3",
                          "--cfg4-sims", "20", "--cfg4-acts", "1", "--cfg5-iters", "5"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    _check_eight_rank_line(line, "self")
    assert abs(line["value"] - 8 * 4096 * 10 / (line["ms_per_step"] * 10e-3)) < 0.01 * line["value"]
    assert line["value_unsettled"] > 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_8ranks_1gpu.json"), "w") as f:
        f.write(lines[0] + "\n")


def test_bench_refuses_more_gpus_than_the_box_has():
    """Without the dry-run switch a plain `--gpus 2` on a 1-GPU box must fail loudly, not print n_gpus: 1."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two devices")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MUAX_BENCH_SINGLE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "device(s) visible" in out.stderr and "{" not in out.stdout


def test_bench_line_schema_at_one_gpu():
    """The driver's N = 1 command: the line carries roofline, cpu_baseline, api, config 3/4/5 sub-objects, and
    roofline.frac is algorithmic bytes / kernel time / 8 TB/s recomputed from its own fields."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "api", "value_pipelined",
                "config3_lunarlander", "config4_atari", "config5_gumbel_train"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["vs_baseline"] is None and line["dtype"] == "f32"
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 8e12) < 2e-3 * r["frac"] + 1e-4
    assert abs(r["achieved"] - r["frac"] * 8000.0) < 0.5
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # the import probe for the reference's arithmetic source is recorded either way
    tp = c["third_party_probe"]
    assert set(("jax", "mctx", "threefry_partitionable")) <= set(tp)
    assert tp["jax"].startswith("absent") or "mctx_cpu" in tp or "error" in tp or tp["mctx"].startswith("absent")
    assert line["value_unsettled"] > 0  # BASELINE.md's protocol without the clock-settling launches, beside `value`
    c4 = line["config4_atari"]
    # (True on an undisturbed box; a lost rendezvous -- another process on the GPU -- legitimately reports False and the
    # leg's numbers are then one workgroup per root's)
    assert c4["pair_mode_wanted"] is True and c4["pair_mode_survived"] in (True, False)
    assert "mfma_busy" in c4["roofline"] and "traffic" in c4["roofline"]
    assert line["api"]["numpy"]["value"] > 0 and line["api"]["device"]["value"] > 0
    for r4 in (line["config4_atari"]["roofline"], line["config4_atari"]["recurrent_pass"]):
        assert r4["bound"] == "mfma" and r4["peak"] == 157.3
        assert abs(r4["frac"] - r4["algorithmic_flops_per_launch"] / (r4["kernel_ms"] * 1e-3) / 157.3e12) < 2e-3
    assert "mz_resnet_search_kernel" in line["config4_atari"]["roofline"]["kernel"]
    c5 = line["config5_gumbel_train"]
    assert c5["act"]["ms_per_act"] > 0 and c5["update"]["ms_per_update"] > 0 and c5["update"]["ms_allreduce"] == 0.0
    assert line["config4_atari"]["n_gpus"] == 1 and c5["n_gpus"] == 1
    # the headline is the synced act; the pipelined rate can only be higher, the kernel alone higher still
    assert line["value_pipelined"] >= 0.98 * line["value"] and r["traffic_source"].startswith("profiles/pmc_traffic.json")
    assert r["kernel_ms"] <= line["ms_per_step"]


def _rccl_worker(q, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import muax_amd as mx
    g = [torch.arange(6, dtype=torch.float32, device="cuda").reshape(2, 3), torch.full((5,), 2.0, device="cuda")]
    want = [t.clone() for t in g]
    mx.sharding.allreduce_mean_flat(g, even_if_alone=True)  # cat -> ONE RCCL all-reduce -> / world -> unpack
    t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing
    dist.barrier()
    got = mx.gather_roots(torch.arange(4, device="cuda"), 4)
    q.put(bool(all(torch.equal(a, b) for a, b in zip(g, want)) and float(t) == 1.5 and got.tolist() == [0, 1, 2, 3]))
    dist.destroy_process_group()


def test_rccl_backend_initialises_and_reduces_on_this_gpu():
    """The 'nccl' (= RCCL) branch that a multi-GPU run takes -- process-group init bound to the rank's device, the
    flat gradient all-reduce, the MAX all-reduce and barrier of bench.py -- executed over RCCL with the one rank a
    1-GPU box can hold.  (Cross-device traffic needs a second GPU; the driver's scaling run covers it.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q, 29700 + os.getpid() % 200))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_bench_rccl_branch_with_one_rank():
    """bench.py with the process group forced on for one rank: init_process_group('nccl', device_id=...), barrier,
    all_reduce(MAX) on a device tensor and all_gather_object all run through RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(MUAX_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", "--cfg4-sims", "16", "--cfg4-acts", "1", "--cfg5-iters", "5"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["config"]["backend"].startswith("rccl") and line["value"] > 1e6
    # the config-5 leg's collective ran through RCCL (one rank: the flat all-reduce timed by itself)
    u = line["config5_gumbel_train"]["update"]
    assert line["config5_gumbel_train"]["backend"].startswith("rccl") and u["ms_allreduce_alone"] > 0
    assert line["config4_atari"]["backend"].startswith("rccl") and "error" not in line["config4_atari"]
