"""Per-act medians (ms) of the listed 3 .. 8-action shapes at 50 simulations, 64 .. 8192 roots (product library, or
MUAX_AMD_LIB=tools/bin/libmzsearch_r04inst.so: round 4's instance list)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import haiku_style_weights
from muax_amd import MuZeroSearch, SearchConfig
from muax_amd.utils import warm_runtime
warm_runtime()
for (A, E, od, sup) in ((3, 8, 4, 10), (4, 8, 4, 10), (4, 16, 8, 10), (4, 10, 8, 10), (6, 8, 6, 10), (6, 8, 6, 20), (8, 8, 6, 10)):
    row = []
    for B in (256, 1024, 4096, 8192):
        s = MuZeroSearch(B, SearchConfig(A, 50, E, tiebreak=True))
        s.set_mlp_weights(haiku_style_weights(0, od, E, A, 2 * sup + 1), od, sup, 0.99)
        obs = (torch.rand(B, od) * 2 - 1).cuda()
        noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).cuda()
        for i in range(4): s.act_mlp(obs, (0, i), dirichlet_noise=noise)
        torch.cuda.synchronize()
        ts = []
        for i in range(20):
            t0 = time.perf_counter(); s.act_mlp(obs, (0, 10 + i), dirichlet_noise=noise); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row.append(sorted(ts)[10] * 1e3); s.close()
    print(f"A={A} E={E} support={sup} S=50: " + "  ".join(f"B={B}: {t:6.3f}" for B, t in zip((256, 1024, 4096, 8192), row)), flush=True)
