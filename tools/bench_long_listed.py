"""Per-act medians (ms) of the long-search shapes that have LISTED instances, at 64 .. 4096 roots.  Run once with the product
library and once with MUAX_AMD_LIB=tools/bin/libmzsearch_r04inst.so (a variant library built from round 4's instance list:
plain records) to compare the LONG instances with the lines they replaced (profiles/r05_generic_route.txt)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bench import haiku_style_weights
from muax_amd import MuZeroSearch, SearchConfig
from muax_amd.utils import warm_runtime
warm_runtime()
for (A, E, od, S) in ((2, 8, 4, 63), (2, 8, 4, 100), (2, 8, 4, 127), (4, 32, 8, 100)):
    row = []
    for B in (64, 256, 1024, 2048, 4096):
        s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True))
        s.set_mlp_weights(haiku_style_weights(0, od, E, A, 21), od, 10, 0.99)
        obs = (torch.rand(B, od) * 2 - 1).cuda()
        noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).cuda()
        for i in range(4): s.act_mlp(obs, (0, i), dirichlet_noise=noise)
        torch.cuda.synchronize()
        ts = []
        for i in range(20):
            t0 = time.perf_counter(); s.act_mlp(obs, (0, 10 + i), dirichlet_noise=noise); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row.append(sorted(ts)[10] * 1e3); s.close()
    print(f"A={A} E={E} S={S}: " + "  ".join(f"B={B}: {t:6.3f}" for B, t in zip((64, 256, 1024, 2048, 4096), row)), flush=True)
