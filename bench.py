#!/usr/bin/env python
"""bench.py -- batched MuZero act() env-steps/s at num_simulations=50 (BASELINE.json metric).

A "step" is one batched act(): root inference + 50 simulations of select -> recurrent_fn -> expand
-> backup + visit-count sampling, for every root of the batch -- ONE launch of the fused gfx950
kernel.  Workload at N=1: BASELINE configs[1] (CartPole-v1 shapes: obs 4, MLP embed 8, A=2, support
10, 4096 parallel roots, S=50) with the reference's full semantics on (Dirichlet root noise, mctx's
threefry tie-break noise, Gumbel sampling from the key).  Inputs are synthetic and already resident in
HBM when the timed region starts.  For N>1 every rank owns 4096 more roots of one global batch (weak
scaling, no collective on the data path; RCCL is used only for the barrier and the max-over-ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (roots per GPU, obs_dim, E, A, support, S)
    "cartpole": (4096, 4, 8, 2, 10, 50),       # BASELINE.json configs[1] -- the metric's config
    "lunarlander": (8192, 8, 32, 4, 10, 50),   # configs[2]
}


def haiku_style_weights(seed, obs_dim, E, A, F, H=16):
    """w ~ TruncNormal(sigma = 1/sqrt(fan_in)), b = 0 (hk.Linear default), torch.Generator(seed)."""
    g = torch.Generator().manual_seed(seed)

    def tn(i, o):
        w = torch.empty(i, o)
        torch.nn.init.trunc_normal_(w, 0.0, 1.0, -2.0, 2.0, generator=g)
        return w / np.sqrt(i)

    z = torch.zeros
    return {"repr_w": tn(obs_dim, E), "repr_b": z(E),
            "pv_w1": tn(E, H), "pv_b1": z(H), "pv_w2": tn(H, F), "pv_b2": z(F),
            "pp_w1": tn(E, H), "pp_b1": z(H), "pp_w2": tn(H, A), "pp_b2": z(A),
            "dr_w1": tn(E + A, H), "dr_b1": z(H), "dr_w2": tn(H, F), "dr_b2": z(F),
            "dn_w1": tn(E + A, H), "dn_b1": z(H), "dn_w2": tn(H, E), "dn_b2": z(E)}


def algorithmic_bytes(depth_sum_total, roots, S, A, E, obs_dim):
    """per root D*[4(5A+3)+48] + S*(8E+4A+36) + (4*obs+4E+8A+20), D from the kernel."""
    return depth_sum_total * (4 * (5 * A + 3) + 48) + roots * (S * (8 * E + 4 * A + 36)
                                                               + (4 * obs_dim + 4 * E + 8 * A + 20))


def usable_cores():
    """Host threads this process may really use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} hw threads visible"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                note += f", cgroup cpu.max quota {q}"
                n = q
    except Exception:
        pass
    return n, note


def cpu_baseline(weights, obs, noise, A, E, F, S, support, budget_s=12.0):
    """The CPU oracle (a port: our restatement of the mctx semantics, NOT the reference itself --
    jax/mctx are not installable here) timed on this box's host cores on the same workload."""
    from oracle import pyoracle as po
    po.build()
    w = {k: v.numpy() for k, v in weights.items()}
    mlp = po.Mlp(w, obs.shape[1], E, A, F, support_size=support)
    cores, cores_note = usable_cores()
    B = obs.shape[0]
    out = {}
    for label, nthreads in (("1", 1), ("all", cores)):
        cfg = po.SearchCfg(S, tiebreak=1)
        tree = po.Tree(B, S + 1, A, E)
        for _ in range(2 if nthreads > 1 else 0):  # let the OpenMP team spread over the cores
            po.act_mlp(mlp, cfg, obs, [0, 99], noise, 0.25, None, 1.0, None, nthreads=nthreads, tree=tree)
        reps, t_total = 0, 0.0
        while t_total < budget_s / 2 and reps < 200:
            t0 = time.perf_counter()
            po.act_mlp(mlp, cfg, obs, [0, reps], noise, 0.25, None, 1.0, None, nthreads=nthreads, tree=tree)
            t_total += time.perf_counter() - t0
            reps += 1
        out[label] = (B * reps / t_total, reps, t_total)
    return {"value": round(out["all"][0], 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "third_party_probe": third_party_probe(w, obs, noise, A, E, F, S, support),
            "single_thread_value": round(out["1"][0], 1),
            "sample": f"{B} roots x S={S}, {out['all'][1]} acts on {cores} threads ({out['all'][2]:.1f}s) "
                      f"and {out['1'][1]} acts on 1 thread ({out['1'][2]:.1f}s); gcc -O2 C oracle, "
                      f"root-major OpenMP; {cores_note}"}


def third_party_probe(weights, obs, noise, A, E, F, S, support, budget_s=20.0):
    """SURVEY.md section 7 step 0 / BASELINE.md section 3's third row: is the reference's arithmetic source (jax + mctx)
    importable on THIS host?  Recorded either way.  When both import, mctx.muzero_policy is jitted for the CPU backend
    around this build's OWN jnp restatement of the default trio (tools/mctx_cpu_glue.py: no reference file travels) and
    timed on the metric's workload -- a third-party baseline, never the thing measured -- and the same call's outputs
    for seeds {0, 1, 2} at 8 roots are dumped under gpurun_out/mctx_capture/ in tests/golden/mctx_fixture.py's format."""
    probe = {}
    for name in ("jax", "mctx", "haiku"):
        try:
            mod = __import__(name)
            probe[name] = getattr(mod, "__version__", "present")
        except Exception as e:  # noqa: BLE001 -- ImportError, or a broken install: both are "absent" here
            probe[name] = f"absent ({type(e).__name__})"
    if not (probe["jax"].startswith("absent") or probe["mctx"].startswith("absent")):
        try:
            import jax
            probe["threefry_partitionable"] = bool(jax.config.jax_threefry_partitionable)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import mctx_cpu_glue
            probe.update(mctx_cpu_glue.time_and_capture(weights, obs, noise, A, E, F, S, support, budget_s,
                                                        os.path.join(ROOT, "gpurun_out", "mctx_capture")))
        except Exception as e:  # noqa: BLE001
            probe["error"] = f"{type(e).__name__}: {e}"[:300]
    else:
        probe["threefry_partitionable"] = None
    return probe


def pmc_traffic(workload, field="hbm_bytes_per_launch"):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get(field)
    except Exception:
        return None


def grow_signal_pool():
    """muax_amd.utils.warm_runtime: the HIP runtime's completion-signal pool grown before anything is timed (a growth
    step inside a timed region is a 50-70 ms host stall -- the whole measurement of a 200-step run; tools/diag_stall.py)."""
    from muax_amd.utils import warm_runtime
    warm_runtime()


def fused_run(workload, B, rank, world, dev, steps, warmup, tiebreak, dist=None, backend="nccl", settle_ms=30.0,
              unsettled_first=False):
    """`steps` timed acts of the fused act() kernel on this rank's B roots of the global batch (inputs resident in
    HBM; every act followed by the host synchronisation; barrier + max over ranks for N > 1), then the same number of
    launches back to back with one synchronisation at the end (`pipelined`).

    Before the `warmup` untimed steps the GPU is brought to its working clocks: the MI355X ramps them over ~10 ms of
    continuous work (the same 20 timed steps measure 36.9 M env-steps/s after 5 warm-up launches = 0.5 ms, 37.6 M after
    20, 39.2 M after 80), and a short run would otherwise report the ramp instead of the rate an RL loop sees.  The
    settle phase is `settle_ms` of untimed launches of the same kernel (--settle-ms 0 turns it off; it is named in
    the JSON line's config)."""
    from muax_amd import MuZeroSearch, SearchConfig
    grow_signal_pool()
    _, obs_dim, E, A, support, S = WORKLOADS[workload]
    F = 2 * support + 1
    weights = haiku_style_weights(0, obs_dim, E, A, F)
    g = torch.Generator().manual_seed(1000 + rank)
    obs = torch.rand(B, obs_dim, generator=g) * 2 - 1
    noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)) if A > 1 else torch.ones(B, 1)
    search = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=tiebreak, global_batch=B * world, root_offset=B * rank), dev)
    search.set_mlp_weights(weights, obs_dim, support, 0.99)
    d_obs, d_noise = obs.to(dev), noise.to(dev)

    def step(i):
        search.act_mlp(d_obs, (0, i), dirichlet_noise=d_noise, dirichlet_fraction=0.25, temperature=1.0)

    # HIP events bracket a SAMPLE of the launches (every 10th): an event pair costs ~6 us of stream time, 5 % of the
    # kernel, so bracketing every launch would slow the very loop that is being timed.  They are created AND recorded
    # once up front: the first record() of an event makes the runtime allocate its completion signal, and a pool that
    # has to grow stalls the host for tens of milliseconds -- once, but inside a 24 ms timed region that is the whole
    # measurement (seen on this pool: one 45 ms step among 200 steps of 0.118 ms).
    ev_every = 10 if steps >= 20 else 1
    sampled = [i for i in range(steps) if i % ev_every == ev_every // 2]
    ev_sets = []
    for _ in range(3):
        evs = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in sampled}
        for a_, b_ in evs.values():
            a_.record()
            b_.record()
        ev_sets.append(evs)
    torch.cuda.synchronize()
    for i in range(warmup):  # the untimed warm-up steps are what the timed steps are: an act and the host synchronisation
        step(i)
        torch.cuda.synchronize()

    def reduce_max(x):
        if not dist:
            return x
        t = torch.tensor([x], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(sync_each, evs):
        """Exactly `steps` acts between two (barrier + synchronize) brackets; the job's time is the MAX over ranks of
        the per-rank times."""
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            if i in evs:
                evs[i][0].record()
            step(warmup + i)
            if i in evs:
                evs[i][1].record()
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0  # THIS rank's time for exactly `steps` steps
        if dist:
            dist.barrier()  # closing bracket: no rank leaves the timed region's neighbourhood before all are done
            torch.cuda.synchronize()
        return reduce_max(dt), float(np.mean([evs[i][0].elapsed_time(evs[i][1]) for i in sampled]))

    # THE TIMED REGION ("B / wall_time(act), host sync at the end included"): every act is followed by
    # the host synchronisation the reference's np.asarray / .item() imply -- the next act of an RL loop needs this
    # act's actions.  The same launches enqueued back to back with one synchronisation at the end are reported beside
    # it as `value_pipelined` (what a caller that keeps several acts in flight gets; rounds 1-3 reported that as `value`).
    # BASELINE.md section 3's own protocol first ("5 warm-up + 20 timed", nothing before the warm-up): the same steps
    # on a GPU that has not yet been brought to its working clocks -> `elapsed_unsettled`; then the settling launches,
    # the warm-up steps again, and the run `value` comes from.  (One handle, every event recorded up front: a second
    # handle / a second set of fresh events in front of the timed region brought the 45 ms signal-pool stall back.)
    elapsed_unsettled = None
    if unsettled_first and settle_ms > 0:
        elapsed_unsettled, _ = timed(True, ev_sets[2])
    if settle_ms > 0:
        t_s = time.perf_counter()
        step(0)
        torch.cuda.synchronize()
        per = max(time.perf_counter() - t_s, 2e-5)
        for i in range(min(5000, int(settle_ms * 1e-3 / per) + 1)):
            step(i)
        for i in range(warmup):
            step(i)
            torch.cuda.synchronize()
    elapsed, kernel_ms = timed(True, ev_sets[0])
    for i in range(max(10, warmup)):  # (the secondary figure gets its own untimed warm-up: a different submission pattern)
        step(i)
    # (the median of three runs: with 200 launches in flight the runtime sometimes stalls the host once for ~17 ms --
    # its queue / signal bookkeeping, box dependent -- which would double this secondary figure; `value` is one run)
    runs = sorted(timed(False, ev_sets[1]) for _ in range(3))
    pipelined, kernel_ms_pipelined = runs[1]
    depth_total = int(search.depth_sum.sum().item())  # last act's D (the same every act up to the key)
    actions = search.action.cpu()
    assert int(actions.min()) >= 0 and int(actions.max()) < A
    search.close()
    return {"elapsed": elapsed, "elapsed_unsettled": elapsed_unsettled, "kernel_ms": kernel_ms, "depth_total": depth_total,
            "weights": weights, "obs": obs,
            "noise": noise, "pipelined": pipelined, "kernel_ms_pipelined": kernel_ms_pipelined}


def roofline(workload, B, kernel_ms, depth_total, kernel_ms_timed_region=None):
    """`kernel_ms`: HIP events around sampled launches of the back-to-back loop (the stream is never empty there, so
    the interval is the kernel's duration -- the figure rocprofv3's kernel trace reports); `kernel_ms_in_timed_region`:
    the same events around launches of the synced timed region, where the interval also holds the host's enqueue
    latency on an idle stream."""
    _, obs_dim, E, A, support, S = WORKLOADS[workload]
    abytes = algorithmic_bytes(depth_total, B, S, A, E, obs_dim)
    achieved = abytes / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": pmc_traffic(workload),
            "traffic_source": "profiles/pmc_traffic.json: separate rocprofv3 --pmc passes of this command, committed; "
                              "NOT re-measured in this run",
            "kernel": "mz_act_fused_kernel", "kernel_ms": round(kernel_ms, 4),
            "kernel_ms_in_timed_region": None if kernel_ms_timed_region is None else round(kernel_ms_timed_region, 4),
            "algorithmic_bytes_per_launch": int(abytes), "mean_selection_depth": round(depth_total / (B * S), 3)}


def api_numbers(workload, B, weights, obs, dev, acts=100):
    """What a caller of the reference's own entry point gets: MuZero.act(rng_key, obs, obs_from_batch=True,
    num_simulations=50) with default arguments -- Dirichlet root noise drawn from the key on the device, tie-break
    noise, sampling -- (a) NumPy in / NumPy out: upload, draw, search, ONE download and the host synchronisation of
    every act (the next act of an RL loop needs this act's actions), (b) device tensors in / out
    (device_outputs=True): no synchronisation inside act(), acts pipeline on the stream."""
    import muax_amd as mx
    _, obs_dim, E, A, support, S = WORKLOADS[workload]
    net = mx.nn.MZNetwork(mx.nn.Representation(E), mx.nn.Prediction(A, 2 * support + 1), mx.nn.Dynamic(E, A, 2 * support + 1))
    m = mx.MuZero(net, support_size=support, device=dev)
    m.init(0, np.zeros((1, obs_dim), np.float32))
    with torch.no_grad():
        for k, p in mx.nn.mlp_trio_weights(m.network).items():
            p.copy_(weights[k].to(dev))
    m.weights_changed()
    obs_np, obs_dev = obs.numpy(), obs.to(dev)
    out = {}
    for label, kw, x in (("numpy", {}, obs_np), ("device", {"device_outputs": True}, obs_dev)):
        for i in range(10):
            m.act(i, x, obs_from_batch=True, num_simulations=S, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(acts):
            a = m.act(1000 + i, x, obs_from_batch=True, num_simulations=S, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / acts
        out[label] = {"value": round(B / dt, 1), "ms_per_act": round(dt * 1e3, 4)}
    out["unit"] = "env-steps/s"
    out["call"] = f"MuZero.act(key, obs[{B},{obs_dim}], obs_from_batch=True, num_simulations={S}), defaults otherwise; {acts} acts"
    return out


MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak of the MI355X (/opt/skills/guides/MI355X_MICROARCH.md)


def recurrent_flops_per_root(A=18, F=21):
    """Useful flops of ONE recurrent_fn call of the reference's ResNet nets on one root (muax/nn.py:118-148,313-378:
    6x6 map, 64 channels): 1x1 stem on [s, action plane] + 8 blocks x 3 convolutions 3x3 (projection, conv_0, conv_1)
    + reward / value / policy heads.  Multiply-add = 2 flops; LayerNorm / ELU / decode are not counted."""
    px = 36
    conv3 = 2 * px * 9 * 64 * 64
    stem = 2 * px * 65 * 64
    r_head = 2 * px * (65 * 64 + 64 * 64) + 2 * (px * 64 * 64 + 64 * F)
    v_head = 2 * px * (64 * 16 + 16 * 16) + 2 * (px * 16 * 16 + 16 * F)
    p_head = 2 * px * (64 * 16) + 2 * (px * 16 * 16 + 16 * A)
    return stem + 24 * conv3 + r_head + v_head + p_head


class Ranks:
    """What a sub-benchmark needs to know about the job: this rank, the world, and the two collectives bench.py uses
    outside the data path (barrier, max over ranks).  world == 1 without a process group: both are no-ops."""

    def __init__(self, rank=0, world=1, dist=None, backend="nccl", dev=None, placement=None):
        self.rank, self.world, self.dist, self.backend, self.dev = rank, world, dist, backend, dev
        self.placement = placement or []

    def bracket(self):
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
            torch.cuda.synchronize()

    def max(self, x):
        if not self.dist:
            return x
        t = torch.tensor([x], device=self.dev if self.backend == "nccl" else "cpu", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, n, sync_each=False):
        """n calls of fn between two (synchronize + barrier) brackets -> seconds per call, MAX over ranks."""
        self.bracket()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        self.bracket()
        return self.max(dt)

    def describe(self):
        return {"n_gpus": self.world, "ranks": self.placement,
                "backend": (("rccl (torch 'nccl')" if self.backend == "nccl" else self.backend) if self.dist
                            else "none (one rank)")}


def config4_atari(rk, roots=128, S=200, acts=3, tower_launches=200, global_roots=1024):
    """BASELINE configs[3]: Atari-shaped 84x84x4 frames, the reference's ResNet nets (muax/nn.py:313-395; random init),
    A = 18, num_simulations = 200, "1024 roots sharded over 8 MI355X": every rank searches ITS 128-root shard of the
    1024-root batch (global_batch = 1024, root_offset = 128 * rank: per-root PRNG streams are those of the whole batch;
    no collective on the path), through MuZero.act() with the search loop captured in one hipGraph; time = MAX over
    ranks, value = roots of all ranks / that.  The dominant kernel is the recurrent_fn launch (fp32 MFMA); rank 0
    times it with HIP events over `tower_launches` back-to-back launches (replayed from a hipGraph, as inside act())
    on the stream act() uses, at the shapes the search calls it with; `roofline` prices it against the dense fp32
    matrix peak."""
    import muax_amd as mx
    dev = rk.dev
    A, F, support = 18, 21, 10
    g = torch.Generator().manual_seed(0)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(A, F, generator=g),
            mx.nn.ResNetDynamic(A, F, generator=g))
    global_roots = max(global_roots, roots * rk.world)
    # rank 0 draws its frames as rounds 1-3 did (from the weights' generator, after the weights: the trees these
    # random-weight nets grow on them are chain-like, mean selection depth 44 -- the harder case); other ranks their own
    go = g if rk.rank == 0 else torch.Generator().manual_seed(100 + rk.rank)
    obs = torch.randint(0, 256, (roots, 84, 84, 4), generator=go).float().to(dev)
    m = mx.MuZero(*mods, capture_graph=True, device=dev)
    m.init(0, np.zeros((1, 84, 84, 4), np.float32))
    kw = dict(obs_from_batch=True, num_simulations=S, device_outputs=True, global_batch=global_roots,
              root_offset=roots * rk.rank)
    for i in range(2):
        m.act(i, obs, **kw)
    dt = rk.timed(lambda i: m.act(10 + i, obs, **kw), acts)
    dt_sync = rk.timed(lambda i: m.act(20 + i, obs, **kw), acts, sync_each=True)
    handle = list(m._policy._handles.values())[0]
    depth = float(handle.depth_sum.float().mean()) / S
    dy, pred = mods[2], mods[1]
    # the dominant kernel of an act: the ONE launch that runs all simulations (mz_resnet_search_kernel), timed with HIP
    # events on the stream act() uses, around the launch itself
    search_ms = None
    if dy.hip_search_ok(pred, (6, 6, 64), support):
        evs = []
        for i in range(acts):
            handle.time_native_loop = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            m.act(30 + i, obs, **kw)
            evs.append(handle.time_native_loop)
        torch.cuda.synchronize()
        handle.time_native_loop = None
        search_ms = float(np.mean([a_.elapsed_time(b_) for a_, b_ in evs]))
    s = torch.rand(roots, 6, 6, 64, generator=g).to(dev)
    a = torch.randint(0, A, (roots,), generator=g).to(dev)
    for _ in range(10):
        dy.hip_recurrent(pred, s, a, support)
    torch.cuda.synchronize()
    # the launches are replayed from a hipGraph (as inside act()): eager Python calls of hip_recurrent cost the host
    # about as much as the kernel costs the device, and the events would time the host
    per_graph = 50
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        dy.hip_recurrent(pred, s, a, support)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(per_graph):
            dy.hip_recurrent(pred, s, a, support)
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rk.bracket()
    reps = max(1, tower_launches // per_graph)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    kernel_ms = e0.elapsed_time(e1) / (reps * per_graph)
    rk.bracket()
    flops = recurrent_flops_per_root(A, F) * roots
    tf = flops / (kernel_ms * 1e-3) / 1e12
    pair = bool(dy.use_pair_tower) and bool(getattr(dy, "_pair_scratch", None))
    # pair mode (two workgroups per root meeting in one XCD's L2) rests on where the hardware places blocks; MuZero
    # checks the status words after every search and drops to one workgroup per root for good when a rendezvous was
    # lost -- `pair_mode_survived` says which launch shape the numbers of this leg were measured with
    pair_wanted = roots <= 128 and os.environ.get("MZS_TOWER_PAIR", "1") != "0"
    shape = "2 workgroups per root" if pair else "1 workgroup per root"
    one_pass = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "kernel": f"mz_resnet_tower{'_pair' if pair else ''}_kernel ({shape}): ONE recurrent_fn pass over the shard",
                "kernel_ms": round(kernel_ms, 4), "algorithmic_flops_per_launch": int(flops), "measured_on": "rank 0"}
    if search_ms is not None:
        tfs = flops * S / (search_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(tfs, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tfs / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                "kernel": f"mz_resnet_search_kernel ({shape}): all {S} simulations of the shard in one launch -- per simulation one "
                          f"recurrent_fn pass + mctx's expand / backward / next simulate on the root's own workgroup(s)",
                "kernel_ms": round(search_ms, 3), "launches_per_act": 1, "algorithmic_flops_per_launch": int(flops * S),
                "flops_counted": "the recurrent_fn passes only (the tree step has none to speak of)", "measured_on": "rank 0"}
        # committed counter passes of this kernel (profiles/pmc_traffic.json "atari"): HBM bytes per launch and how busy
        # the matrix pipes were, each labelled with the simulation count its pass ran at; NOT re-measured in this run
        roof["traffic"] = pmc_traffic("atari")
        roof["traffic_source"] = pmc_traffic("atari", "source")
        roof["mfma_busy"] = pmc_traffic("atari", "mfma_busy")
    else:
        roof = dict(one_pass, launches_per_act=S)
    out = {"value": round(roots * rk.world / dt_sync, 1), "unit": "env-steps/s", "ms_per_act": round(dt_sync * 1e3, 3),
           "value_pipelined": round(roots * rk.world / dt, 1), "ms_per_act_pipelined": round(dt * 1e3, 3), "acts": acts,
           "workload": f"atari: {roots} roots per GPU = rows [{roots}*rank, {roots}*(rank+1)) of a {global_roots}-root batch, "
                       f"obs 84x84x4, ResNet nets (embedding 6x6x64), A={A}, support {support}, num_simulations={S}, "
                       f"MuZero policy, simulation loop in one launch; no collective on the path",
           "roots_per_gpu": roots, "global_batch": global_roots,
           "pair_mode_wanted": pair_wanted, "pair_mode_survived": pair if pair_wanted else None,
           "mean_selection_depth": round(depth, 2), "dtype": "f32",
           "search_share_of_act": None if search_ms is None else round(search_ms / (dt * 1e3), 3),
           "roofline": roof, "recurrent_pass": one_pass}
    out.update(rk.describe())
    return out


def config5_gumbel_train(rk, B=4096, L=10, S=50, iters=50):
    """BASELINE configs[4]: Gumbel MuZero act() on 4096 roots per GPU (num_simulations = 50) and ONE k_steps = 10
    unrolled training step (loss, 18 gradients, Adam) on 4096 trajectories per GPU -- default MLP trio, both single
    fused launches.  With N > 1 ranks update() averages the flat gradient vector with ONE all-reduce (RCCL over xGMI;
    muax_amd/sharding.py, precedent muax/frameworks/acme/jax/muzero/learning.py:151) INSIDE the timed update();
    `ms_allreduce` = update() with the mean minus update() without it (dp_mean=False), both MAX over ranks, and
    `ms_allreduce_alone` = the collective on the same vector timed by itself with a synchronisation per call."""
    import muax_amd as mx
    from muax_amd.sharding import allreduce_mean_flat
    dev = rk.dev
    g = torch.Generator().manual_seed(0)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                          mx.nn.Dynamic(8, 2, 21, generator=g))
    m = mx.MuZero(net, policy="gumbel", device=dev)
    m.init(0, np.zeros((1, 4)))
    gr = torch.Generator().manual_seed(10 + rk.rank)  # every rank its own trajectories (and the same initial weights)
    obs = (torch.rand(B, 4, generator=gr) * 2 - 1).to(dev)
    rng = np.random.default_rng(rk.rank)
    batch = mx.Transition(obs=torch.rand(B, L, 4, generator=gr).to(dev), a=torch.randint(0, 2, (B, L), generator=gr).to(dev),
                          r=torch.rand(B, L, generator=gr).to(dev), Rn=(torch.rand(B, L, generator=gr) * 20).to(dev),
                          pi=torch.as_tensor(rng.dirichlet([1, 1], (B, L)).astype(np.float32)).to(dev))
    akw = dict(obs_from_batch=True, num_simulations=S, device_outputs=True, global_batch=B * rk.world, root_offset=B * rk.rank)

    def timeit(fn, n, warm=5, sync_each=False):
        for i in range(warm):
            fn(i)
        return rk.timed(fn, n, sync_each)

    t_act = timeit(lambda i: m.act(1, obs, **akw), iters)
    t_act_sync = timeit(lambda i: m.act(1, obs, **akw), iters, sync_each=True)
    t_upd = timeit(lambda i: m.update(batch), iters)  # every rank its own batch, ONE shared gradient mean per step
    flat = m._fused_train.grads if m._fused_train is not None else None
    same = None
    if rk.dist:  # data-parallel training keeps the replicas bit-identical: checked after the timed steps
        w = torch.cat([p.detach().reshape(-1) for mod in m.network for p in mod.parameters()])
        w = w if rk.backend == "nccl" else w.cpu()
        lo, hi = w.clone(), w.clone()
        rk.dist.all_reduce(lo, op=rk.dist.ReduceOp.MIN)
        rk.dist.all_reduce(hi, op=rk.dist.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
    t_ar = None
    if rk.dist and flat is not None:
        t_ar = timeit(lambda i: allreduce_mean_flat([flat], even_if_alone=True), iters, sync_each=True)
    t_upd_local = timeit(lambda i: m.update(batch, dp_mean=False), iters)  # (last: the replicas drift apart here)
    out = {"act": {"value": round(B * rk.world / t_act_sync, 1), "unit": "env-steps/s", "ms_per_act": round(t_act_sync * 1e3, 4),
                   "value_pipelined": round(B * rk.world / t_act, 1), "ms_per_act_pipelined": round(t_act * 1e3, 4)},
           "update": {"value": round(B * L * rk.world / t_upd, 1), "unit": "transitions/s",
                      "ms_per_update": round(t_upd * 1e3, 4), "ms_per_update_without_allreduce": round(t_upd_local * 1e3, 4),
                      "ms_allreduce": round((t_upd - t_upd_local) * 1e3, 4) if rk.dist else 0.0,
                      "ms_allreduce_alone": None if t_ar is None else round(t_ar * 1e3, 4),
                      "allreduce_bytes": None if flat is None else int(flat.numel() * 4),
                      "allreduce_in_timed_update": bool(rk.dist),
                      "weights_identical_on_all_ranks_after": same},
           "ms_per_iteration": round((t_act_sync + t_upd) * 1e3, 4), "iters": iters, "dtype": "f32",
           "workload": f"gumbel + train: {B} roots per GPU, Gumbel MuZero act() num_simulations={S} (max_num_considered_actions 16, "
                       f"gumbel_scale 1) + update() on {B} trajectories per GPU x k_steps={L}, default MLP trio, Adam; "
                       f"data-parallel gradient mean = one flat all-reduce per update"}
    out.update(rk.describe())
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, LOCAL_RANK ->
    device ordinal, rendezvous on a free 127.0.0.1 port), let rank 0 print the line, and fail loudly when fewer than
    N devices are visible.  MUAX_BENCH_SINGLE_DEVICE=1 (dry run on a 1-GPU box) puts every rank on device 0 over gloo."""
    import socket
    import subprocess
    n = args.gpus
    single = bool(os.environ.get("MUAX_BENCH_SINGLE_DEVICE"))
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (1 if single else n):
        raise SystemExit(f"bench.py --gpus {n}: only {have} ROCm device(s) visible "
                         f"(set MUAX_BENCH_SINGLE_DEVICE=1 to dry-run {n} ranks on device 0)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   MUAX_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cartpole", choices=sorted(WORKLOADS))
    ap.add_argument("--roots", type=int, default=0, help="roots per GPU (default: the workload's)")
    ap.add_argument("--no-tiebreak", action="store_true", help="drop mctx's threefry tie-break noise (NOT the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the api / config-3 sub-objects of the JSON line")
    ap.add_argument("--settle-ms", type=float, default=30.0,
                    help="untimed launches before the warm-up steps until the GPU clocks have ramped (see fused_run)")
    ap.add_argument("--no-unsettled", action="store_true", help="skip the un-settled run that precedes the settled one")
    ap.add_argument("--no-config45", action="store_true", help="skip the config4_atari / config5_gumbel_train sub-objects")
    ap.add_argument("--cfg4-sims", type=int, default=200, help="num_simulations of the config-4 leg (200 = BASELINE's; dry runs lower it)")
    ap.add_argument("--cfg4-acts", type=int, default=3)
    ap.add_argument("--cfg5-iters", type=int, default=50)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N "
                         f"(plain `python bench.py --gpus N` starts them itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    # dry run of the N>1 code path on a 1-GPU box: every rank on device 0 (RCCL wants one device per rank -> gloo)
    single = bool(os.environ.get("MUAX_BENCH_SINGLE_DEVICE"))
    if single:
        local_rank = 0
    backend = os.environ.get("MUAX_BENCH_BACKEND", "gloo" if single else "nccl")  # "nccl" is RCCL on ROCm
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} ROCm device(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    # MUAX_BENCH_FORCE_DIST=1: initialise the process group for one rank too, so that the RCCL branch (init, barrier,
    # max-over-ranks all-reduce on a device tensor, all_gather_object) executes on a 1-GPU box
    if world > 1 or os.environ.get("MUAX_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        import datetime
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        # a rank that dies inside a sub-benchmark must not hang the others for the default half hour
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5), **kw)

    B, obs_dim, E, A, support, S = WORKLOADS[args.workload]
    if args.roots:
        B = args.roots
    F = 2 * support + 1
    dev = torch.device("cuda", local_rank)
    run = fused_run(args.workload, B, rank, world, dev, args.steps, args.warmup, not args.no_tiebreak, dist, backend,
                    args.settle_ms, unsettled_first=not args.no_unsettled)
    unsettled = None if run["elapsed_unsettled"] is None else B * world * args.steps / run["elapsed_unsettled"]
    elapsed, kernel_ms, depth_total, weights, obs, noise = (run[k] for k in ("elapsed", "kernel_ms", "depth_total",
                                                                               "weights", "obs", "noise"))
    # every rank states the device it runs on AND the LOCAL_RANK it was launched with (the ordinal it takes on a real node:
    # in the single-device dry run they differ, on an 8-GPU node a placement bug would show here as text)
    env_local = int(os.environ.get("LOCAL_RANK", "0"))
    placement = [f"rank {rank}: cuda:{local_rank} ({torch.cuda.get_device_properties(local_rank).gcnArchName.split(':')[0]}); "
                 f"LOCAL_RANK={env_local}" + (f" (dry run on one device: cuda:{env_local} on a real node)" if single else "")]
    if dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, placement[0])
        placement = gathered
    rk = Ranks(rank, world, dist, backend, dev, placement)

    line = {}
    if rank == 0:
        line = {
            "metric": "batched act() env-steps/sec at num_simulations=50",
            # roots / wall time of an act WITH the host synchronisation at its end (rounds 1-3 reported
            # the pipelined rate here and this one as `value_synced`)
            "value": round(B * world * args.steps / elapsed, 1),
            "unit": "env-steps/s",
            # the same launches enqueued back to back, one synchronisation after the last
            # the same steps timed FIRST, right after the warm-up steps and before the clock-settling launches
            # (BASELINE.md section 3's protocol as written)
            "value_unsettled": None if unsettled is None else round(unsettled, 1),
            "value_pipelined": round(B * world * args.steps / run["pipelined"], 1),
            "ms_per_step_pipelined": round(run["pipelined"] / args.steps * 1e3, 4),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {B} roots/GPU, obs {obs_dim}, MLP embed {E}, A={A}, "
                                   f"support {support}, num_simulations={S}, dirichlet 0.25/0.3, "
                                   f"tiebreak={'threefry' if not args.no_tiebreak else 'off'}, temperature 1",
                       "roots_per_gpu": B, "num_simulations": S, "parallelism": f"roots sharded x{world}, no collective",
                       "timed_region": "steps x (one fused act() launch + torch.cuda.synchronize())",
                       "clock_settle_ms_before_warmup": args.settle_ms,
                       "backend": rk.describe()["backend"],
                       "launcher": os.environ.get("MUAX_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "none"),
                       "ranks": placement},
            "roofline": roofline(args.workload, B, run["kernel_ms_pipelined"], depth_total, kernel_ms),
        }

    def guarded(name, fn):
        # a secondary measurement must never cost the headline line: its failure is reported in its own slot.  With
        # N > 1 every rank runs fn (the legs time MAX over ranks through barriers); a rank that fails still joins the
        # others' remaining collectives only by luck, so failures are all-or-nothing: the flag is agreed first
        try:
            res = fn()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            res = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0:
            line[name] = res

    extras = not args.no_extras and args.workload == "cartpole" and not args.roots
    if rank == 0 and world == 1 and not args.no_extras:
        # the reference's own entry point on the same workload
        guarded("api", lambda: api_numbers(args.workload, B, weights, obs, dev))
    if extras and world == 1:
        def config3():  # BASELINE configs[2] (the other fused instance)
            B3 = WORKLOADS["lunarlander"][0]
            steps3 = max(20, args.steps // 4)
            r3 = fused_run("lunarlander", B3, 0, 1, dev, steps3, max(5, args.warmup // 2),
                           not args.no_tiebreak, settle_ms=args.settle_ms)
            return {"value": round(B3 * steps3 / r3["elapsed"], 1), "unit": "env-steps/s", "steps": steps3,
                    "value_pipelined": round(B3 * steps3 / r3["pipelined"], 1),
                    "ms_per_step": round(r3["elapsed"] / steps3 * 1e3, 4),
                    "workload": f"lunarlander: {B3} roots, obs 8, MLP embed 32, A=4, support 10, num_simulations=50",
                    "roofline": roofline("lunarlander", B3, r3["kernel_ms_pipelined"], r3["depth_total"], r3["kernel_ms"])}
        guarded("config3_lunarlander", config3)
    if extras and not args.no_config45:
        # BASELINE configs[3] and [4] are multi-GPU configurations: every rank runs its shard / its replica
        guarded("config4_atari", lambda: config4_atari(rk, S=args.cfg4_sims, acts=args.cfg4_acts))
        guarded("config5_gumbel_train", lambda: config5_gumbel_train(rk, iters=args.cfg5_iters))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        guarded("cpu_baseline", lambda: cpu_baseline(weights, obs.numpy(), noise.numpy(), A, E, F, S, support))
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
