"""Assemble profiles/<tag>_* from what tools/run_round_profiles.sh <tag> (+ the tower / atari / EZ / plugin-net runs) left
under gpurun_out/<tag>/.  Headers are kept from the existing profiles/<tag>_* files when they exist (the text above the
first data line), so that re-collecting after a re-run only swaps the measurements.
    python tools/collect_round_profiles.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
R, P = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")


def rd(name):
    with open(os.path.join(R, name)) as f:
        return f.read()


def header(path, default=""):
    """Leading '#' lines (and blank lines between them) of an existing profile file."""
    if not os.path.exists(path):
        return default
    out = []
    for ln in open(path):
        if (ln.startswith("#") and not ln.startswith("##")) or (not ln.strip() and out):
            out.append(ln)
        else:
            break
    return "".join(out)


def write(name, body, default_header=""):
    path = os.path.join(P, f"{tag}_{name}")
    text = header(path, default_header) + body
    with open(path, "w") as f:
        f.write(text)
    print("wrote", os.path.relpath(path, ROOT))


d = json.load(open(os.path.join(R, "bench.json")))
l = json.load(open(os.path.join(R, "bench_lunarlander.json")))
with open(os.path.join(P, f"{tag}_bench.json"), "w") as f:
    f.write(json.dumps(d) + "\n" + json.dumps(l) + "\n")
write("rocprofv3_cartpole4096.txt", "\n## CartPole 4096 roots (the metric)\n" + rd("prof_cartpole.txt")
      + "\n## LunarLander 8192 roots (config 3): the packed compact record, two workgroups per CU\n" + rd("prof_lunarlander.txt"))
write("rocprofv3_atari128.txt", rd("atari_trace.txt") + "\n## MFMA counters of the recurrent kernel (per-dispatch averages)\n"
      + rd("tower_pmc.txt"))
write("batch_scaling.txt", rd("batch_scaling.txt"))
write("phase_cycles.txt", rd("phase_cycles.txt"))
sections = [("## config 4's shard end to end (tools/bench_atari.py 128 200): 40.3 ms at the end of round 2\n", "atari_bench.txt"),
            ("\n## the recurrent kernel alone (tools/bench_tower.py; eager launches timed with events: includes the launch gaps)\n", "tower_bench.txt"),
            ("\n## root inference of the convolutional nets, 128 roots of 84x84x4 frames (tools/bench_root_inference.py)\n", "root_inference.txt"),
            ("\n## EfficientZero-style nets through MuZero.act(), search loop as one hipGraph (tools/bench_ez.py [roots] [S] [channels])\n", "ez_bench.txt"),
            ("\n## kernel trace of the EZ search, C = 32 (tools/rocprof_ez.sh 32)\n", "ez_trace.txt"),
            ("\n## config 5's shape on one GPU (tools/bench_cfg5.py)\n", "cfg5.txt"),
            ("\n## the representation nets' C -> C 3x3 convolutions and residual blocks on 128 images (tools/bench_repr_conv.py;\n"
             "## Python call to call, i.e. >= ~15 us per launch)\n", "repr_conv.txt")]
write("plugin_nets.txt", "".join(h + rd(n) for h, n in sections if os.path.exists(os.path.join(R, n))),
      f"# {tag} -- wall-clock numbers of the plugin-net paths (1 x MI355X; gpurun_out/{tag})\n\n")
if os.path.exists(os.path.join(R, "api_profile.txt")):
    write("api_profile.txt", rd("api_profile.txt"))
if os.path.exists(os.path.join(R, "search_phases.txt")):  # round 4: the one-launch ResNet search (tools/profile_search.py)
    write("search_phases.txt", rd("search_phases.txt"),
          f"# {tag} -- config 4's shard (128 roots x 200 simulations): act() through the three routes, and in-kernel phase timers\n"
          f"# (s_memtime, -DMZ_PROFILE build) of the one-launch search, per simulation (tools/profile_search.py run)\n\n")
if os.path.exists(os.path.join(R, "generic.txt")):  # round 4: default-trio shapes without a listed instance
    write("generic_route.txt", rd("generic.txt"),
          f"# {tag} -- act() of the default MLP trio: tuned fused instance vs the generic one-launch search (mz_mlp_generic.cuh),\n"
          f"# and shapes only the generic route serves (tools/bench_generic.py)\n\n")
if os.path.exists(os.path.join(R, "bench_2ranks_1gpu.json")):
    with open(os.path.join(P, f"{tag}_bench_2ranks_1gpu.json"), "w") as f:
        f.write(open(os.path.join(R, "bench_2ranks_1gpu.json")).read())
    print("wrote", f"profiles/{tag}_bench_2ranks_1gpu.json")


def pmc(text, counter):
    for ln in text.splitlines():
        if counter in ln and "mz_act_fused" in ln:
            return float(ln.split(counter)[1].split()[0])
    return None


pj = json.load(open(os.path.join(P, "pmc_traffic.json")))
for wl, fn in (("cartpole", "prof_cartpole.txt"), ("lunarlander", "prof_lunarlander.txt")):
    fs, ws = pmc(rd(fn), "FETCH_SIZE"), pmc(rd(fn), "WRITE_SIZE")
    if fs and ws:
        pj[wl].update({"hbm_bytes_per_launch": int((2 * fs + ws) * 1024), "fetch_size_kb_raw": fs, "write_size_kb_raw": ws})
json.dump(pj, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print("pmc_traffic.json refreshed")
