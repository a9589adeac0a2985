"""Path statistics of the metric's workload (4096 CartPole-shaped roots x 50 simulations) from the ORACLE's trees -- what
the fused kernel's backup (lane e <-> path entry e, 16 entries per pass, four roots per wavefront) meets: how much of a
path is the previous simulation's, how often a wavefront needs a second pass.  CPU only (profiles/r06_backup_ab.txt).

    python tools/backup_prefix_stats.py
"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from oracle import pyoracle as po
B, obs_dim, E, A, support, S = bench.WORKLOADS['cartpole']
F=2*support+1
w = {k: v.numpy() for k,v in bench.haiku_style_weights(0, obs_dim, E, A, F).items()}
g = torch.Generator().manual_seed(1000)
obs = (torch.rand(B, obs_dim, generator=g)*2-1).numpy()
noise = torch.distributions.Dirichlet(torch.full((A,),0.3)).sample((B,)).numpy()
ref = po.act_mlp(po.Mlp(w, obs_dim, E, A, F), po.SearchCfg(S, tiebreak=1), obs, [0, 0], noise, 0.25)
par = ref["tree"].parents
N=S+1
depth = np.zeros((B,N),int)
for n in range(1,N): depth[:,n] = depth[np.arange(B), par[:,n]] + 1
# simpler: ancestors table
anc = np.full((B,N,N+1),-1,int)   # anc[b,n,l] = ancestor of n at level l
for n in range(N):
    if n==0:
        anc[:,0,0]=0; continue
    p = par[:,n]
    anc[:,n,:] = anc[np.arange(B),p,:]
    anc[np.arange(B),n,depth[:,n]] = n
shared=[]; dep=[]
for s in range(1,S):
    a = anc[:,s+1,:]; b = anc[:,s,:]
    same = (a==b) & (a>=0)
    # common prefix length
    cp = np.cumprod(same,axis=1).sum(1)
    shared.append(cp); dep.append(depth[:,s+1]+1)
shared=np.array(shared); dep=np.array(dep)
print("entries per path (depth+1) mean %.2f ; entries identical to the previous simulation's (common prefix) mean %.2f = %.0f%%" % (dep.mean(), shared.mean(), 100*shared.sum()/dep.sum()))
print("paths whose EVERY entry but the new leaf is unchanged: %.1f%%" % (100*(shared==dep-1).mean()))
# per wave: lanes that diverge
div = dep-shared   # entries that must be loaded per row
w4 = div.reshape(S-1,B//4,4)
print("diverging entries per row-sim: mean %.2f; per wave the row with most: mean %.2f; P(a wave-sim has a row with >=1 diverging entry)=%.3f" % (div.mean(), w4.max(2).mean(), (w4.max(2)>=1).mean()))
wm = (dep-1).reshape(S-1,B//4,4).max(2)
print("P(second pass: deepest row of the wave >= 16 levels) = %.3f ; mean passes %.3f" % ((wm>=16).mean(), 1+(wm>=16).mean()))
rowdeep=(dep-1>=16)
print("P(a single row >= 16 levels) = %.3f ; all four rows of a wave: %.4f" % (rowdeep.mean(), rowdeep.reshape(S-1,B//4,4).all(2).mean()))
