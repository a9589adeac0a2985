import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
B, obs_dim, E, A, support, S = bench.WORKLOADS["cartpole"]
w = bench.haiku_style_weights(0, obs_dim, E, A, 2 * support + 1)
obs = torch.rand(B, obs_dim) * 2 - 1
for _ in range(3):
    print(json.dumps(bench.api_numbers("cartpole", B, w, obs, torch.device("cuda", 0), acts=300)))
