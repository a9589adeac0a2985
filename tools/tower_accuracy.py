"""Accuracy of the fp32-MFMA recurrent kernel of the ResNet nets (mzs_resnet_tower with heads) against an fp64 CPU
evaluation of the same torch modules, next to MIOpen's fp32 evaluation of them.  Both launch shapes.
    python tools/tower_accuracy.py [seeds]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import muax_amd as mx  # noqa: E402

A, SUPPORT = 18, 10
names = ("reward", "value", "prior_logits", "next_state")
rows = {n: [] for n in names}
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    g = torch.Generator().manual_seed(100 + seed)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(A, 21, generator=g),
            mx.nn.ResNetDynamic(A, 21, generator=g))
    m = mx.MuZero(*mods)
    m.init(0, np.zeros((1, 84, 84, 4), np.float32))
    with torch.no_grad():
        for mod in mods[1:]:
            for p in mod.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g).to(p.device))
    d, pred = mods[2], mods[1]
    B = 32
    s = torch.rand(B, 6, 6, 64, generator=g)
    a = torch.randint(0, A, (B,), generator=g)
    d64, p64 = copy.deepcopy(d).cpu().double(), copy.deepcopy(pred).cpu().double()
    d64.use_hip_tower = False
    with torch.no_grad():
        r_l, ns64 = d64(s.double(), a)
        v_l, lg64 = p64(ns64)
        ref = (mx.utils.support_to_scalar(torch.softmax(r_l, -1), SUPPORT).flatten(),
               mx.utils.support_to_scalar(torch.softmax(v_l, -1), SUPPORT).flatten(), lg64, ns64)
    sc, ac = s.cuda(), a.cuda()
    d.use_hip_tower = False
    (r0, _, lg0, v0), ns0 = m._recurrent_inference(None, None, ac, sc)
    d.use_hip_tower = True
    lib = (r0, v0, lg0, ns0)
    d.use_pair_tower = False
    hip = d.hip_recurrent(pred, sc, ac, SUPPORT)
    d.use_pair_tower = True
    hip2 = d.hip_recurrent(pred, sc, ac, SUPPORT)
    assert all(torch.equal(x, y) for x, y in zip(hip, hip2)), "launch shapes differ"
    for n, h, l, x in zip(names, hip, lib, ref):
        rows[n].append((float((h.double().cpu() - x).abs().max()), float((l.double().cpu() - x).abs().max()),
                        float((h.double().cpu() - x).abs().mean()), float((l.double().cpu() - x).abs().mean()),
                        float(x.abs().max())))
print("error against fp64 over the seeds (32 roots each): max abs [HIP kernel | MIOpen fp32], mean abs [HIP | MIOpen], max |x|")
for n in names:
    r = np.array(rows[n])
    print(f"  {n:13s} max {r[:, 0].max():.2e} | {r[:, 1].max():.2e}   mean {r[:, 2].mean():.2e} | {r[:, 3].mean():.2e}   max|x| {r[:, 4].max():.2f}"
          f"   per seed HIP/MIOpen max-error ratio: " + " ".join(f"{a / b:.2f}" for a, b in zip(r[:, 0], r[:, 1])))
