"""Pair mode of the fused recurrent kernel against one workgroup per root over many batch sizes, inputs and
repeated launches (also back to back on one scratch, and inside a captured graph).
    python tools/stress_tower_pair.py [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import muax_amd as mx  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(0)
mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
        mx.nn.ResNetDynamic(18, 21, generator=g))
m = mx.MuZero(*mods)
m.init(0, np.zeros((1, 84, 84, 4), np.float32))
d, pred = mods[2], mods[1]
rng = np.random.default_rng(1)
worst, launches = 0.0, 0
for it in range(rounds):
    B = int(rng.integers(1, 129))
    s = torch.rand(B, 6, 6, 64, generator=g).cuda() * float(rng.choice([0.1, 1.0, 5.0]))
    a = torch.randint(0, 18, (B,), generator=g).cuda()
    d.use_pair_tower = False
    ref = d.hip_recurrent(pred, s, a, 10)
    d.use_pair_tower = True
    reps = int(rng.integers(1, 6))
    for _ in range(reps):
        out = d.hip_recurrent(pred, s, a, 10)
    launches += reps
    again = d.hip_recurrent(pred, s, a, 10)
    for k, (x, y, z) in enumerate(zip(ref, out, again)):
        assert torch.equal(y, z), (it, B, k, "pair mode is not deterministic")
        e = float((x - y).abs().max()) / max(1.0, float(x.abs().max()))
        worst = max(worst, e)
        if e >= 1e-3:
            # is it the kernel or the problem?  the torch modules against one workgroup per root on the same input
            d.use_hip_tower = False
            (r0, _, lg0, v0), ns0 = m._recurrent_inference(None, None, a, s)
            d.use_hip_tower = True
            t = (r0, v0, lg0, ns0)[k]
            et = float((t - x).abs().max()) / max(1.0, float(x.abs().max()))
            print(f"round {it} B={B} output {k}: pair vs single {e:.2e}; torch modules vs single {et:.2e}", flush=True)
            assert e < 20 * max(et, 1e-4), (it, B, k, e, et)
    assert d.pair_status() == 0, (it, B)
# inside a captured graph: the launch epoch lives in device memory, so replays keep their message numbers
B = 100
s = torch.rand(B, 6, 6, 64, generator=g).cuda()
a = torch.randint(0, 18, (B,), generator=g).cuda()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        d.hip_recurrent(pred, s, a, 10)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        y1 = d.hip_recurrent(pred, s, a, 10)
        y2 = d.hip_recurrent(pred, y1[3], a, 10)
    for _ in range(50):
        graph.replay()
    torch.cuda.synchronize()
    got = [t.clone() for t in y2]
    e1 = d.hip_recurrent(pred, s, a, 10)   # the same two launches eagerly, in pair mode: the same bits
    e2 = d.hip_recurrent(pred, e1[3], a, 10)
    torch.cuda.synchronize()
for x, y in zip(e2, got):
    assert torch.equal(x, y), "graph replay differs from the eager pair-mode launches"
assert d.pair_status() == 0
print(f"{rounds} batch sizes, {launches} pair-mode launches + 100 inside a replayed graph: status 0 everywhere, "
      f"worst relative difference to one workgroup per root {worst:.2e} "
      f"(ill-conditioned inputs -- a channel with a tiny range under min_max_normalize2d -- are judged against the torch modules' own distance)")
