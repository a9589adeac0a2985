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
write("rocprofv3_atari128.txt", rd("atari_trace.txt") + "\n## counter passes of the search kernel (per-dispatch averages; the simulation "
      "count of the passes is in their first line)\n" + rd("tower_pmc.txt"))
write("batch_scaling.txt", rd("batch_scaling.txt"))
write("phase_cycles.txt", rd("phase_cycles.txt"))
sections = [("## config 4's shard end to end (tools/bench_atari.py 128 200, then 1024 roots on the one GPU)\n", "atari_bench.txt"),
            ("\n## the recurrent kernel alone (tools/bench_tower.py; eager launches timed with events: includes the launch gaps)\n", "tower_bench.txt"),
            ("\n## root inference of the convolutional nets, 128 roots of 84x84x4 frames (tools/bench_root_inference.py)\n", "root_inference.txt"),
            ("\n## EfficientZero-style nets through MuZero.act(), search loop as one hipGraph (tools/bench_ez.py [roots] [S] [channels])\n", "ez_bench.txt"),
            ("\n## kernel trace of the EZ search, C = 32 (tools/rocprof_ez.sh 32)\n", "ez_trace.txt"),
            ("\n## kernel trace of the EZ nets' ROOT inference, last inference of the run: gap before / duration in us / kernel "
             "(tools/rocprof_ez_root.sh)\n", "ez_root_trace.txt"),
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
    with open(os.path.join(P, f"{tag}_search_phases.txt"), "a") as f:
        if os.path.exists(os.path.join(R, "search_phases_heads.txt")):
            f.write("\n## the same with the heads' own timers (-DMZ_PROF_HEADS build: slots 3..9 time the pieces of \"heads after the tower\")\n"
                    + "".join(ln for ln in rd("search_phases_heads.txt").splitlines(True) if not ln.startswith("act()")))
        if os.path.exists(os.path.join(R, "ab_ldstree.txt")):
            f.write("\n## the tree statistics in LDS against the HBM tree, same kernels otherwise (MZS_SEARCH_LDS_TREE=1 / 0), same box,\n"
                    "## three alternations (tools/bench_atari.py 128 200)\n" + rd("ab_ldstree.txt"))
if os.path.exists(os.path.join(R, "generic.txt")):  # round 4: default-trio shapes without a listed instance
    write("generic_route.txt", rd("generic.txt"),
          f"# {tag} -- act() of the default MLP trio: tuned fused instance vs the generic one-launch search (mz_mlp_generic.cuh),\n"
          f"# and shapes only the generic route serves (tools/bench_generic.py)\n\n")
if os.path.exists(os.path.join(R, "bench_long.txt")):
    with open(os.path.join(P, f"{tag}_generic_route.txt"), "a") as f:
        f.write("\n## long searches / wide action sets through mzs_act_mlp at several batch sizes: median ms per act (tools/bench_long.py);\n"
                "## plan = (support slots, tree size, wavefronts per workgroup, LONG record)\n" + rd("bench_long.txt"))
if os.path.exists(os.path.join(R, "sync_gap.txt")):
    with open(os.path.join(P, f"{tag}_generic_route.txt"), "a") as f:
        f.write("\n## a synchronised act() of the metric's workload taken apart (tools/sync_gap.py, three runs)\n" + rd("sync_gap.txt"))
for multi in ("bench_2ranks_1gpu.json", "bench_8ranks_1gpu.json", "bench_8ranks_1gpu_torchrun.json"):
    if os.path.exists(os.path.join(R, multi)) or os.path.exists(os.path.join(ROOT, "gpurun_out", multi)):
        src = os.path.join(R, multi) if os.path.exists(os.path.join(R, multi)) else os.path.join(ROOT, "gpurun_out", multi)
        # only the JSON line(s): the ranks' "[Gloo] Rank 0 is connected ..." chatter on stdout is not part of the record
        keep = [ln for ln in open(src).read().splitlines() if ln.startswith("{")]
        with open(os.path.join(P, f"{tag}_{multi}"), "w") as f:
            f.write("\n".join(keep) + "\n")
        print("wrote", f"profiles/{tag}_{multi}")


for extra, name, hdr in (("stress_round5.txt", "stress_parity.txt",
                          f"# {tag} -- randomised parity campaigns of the round's kernels against the C oracle (tools/stress_round5.py)\n\n"),
                         ("stall.txt", "host_stall.txt",
                          f"# {tag} -- where the HIP runtime stalls the host: per-act wall times of 4000 synchronised fused acts, without / with\n"
                          f"# muax_amd.utils.warm_runtime's 2048 pre-recorded events (tools/diag_stall.py N events-per-step pre-recorded)\n\n"),
                         ("ab_split.log", "spec_order_ab.txt", "")):
    if os.path.exists(os.path.join(R, extra)):
        write(name, rd(extra), hdr)


def pmc(text, counter):
    for ln in text.splitlines():
        if counter in ln and "mz_act_fused" in ln:
            return float(ln.split(counter)[1].split()[0])
    return None


pj = json.load(open(os.path.join(P, "pmc_traffic.json")))
for wl, fn in (("cartpole", "prof_cartpole.txt"), ("lunarlander", "prof_lunarlander.txt")):
    fs, ws = pmc(rd(fn), "FETCH_SIZE"), pmc(rd(fn), "WRITE_SIZE")
    if fs and ws:
        pj[wl].update({"hbm_bytes_per_launch": int((2 * fs + ws) * 1024), "fetch_size_kb_raw": fs, "write_size_kb_raw": ws,
                       "source": f"profiles/{tag}_rocprofv3_cartpole4096.txt (tools/rocprof_passes.sh: separate --pmc passes of "
                                 f"`python bench.py --workload {wl}`)"})
json.dump(pj, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print("pmc_traffic.json refreshed")


def pmc_search(text, counter):
    for ln in text.splitlines():
        if counter in ln and "mz_resnet_search_kernel" in ln:
            return float(ln.split(counter)[1].split()[0])
    return None


if os.path.exists(os.path.join(R, "tower_pmc.txt")):
    t = rd("tower_pmc.txt")
    sims = int(t.split("counters of a ")[1].split("-simulation")[0]) if "counters of a " in t else None
    fs, ws = pmc_search(t, "FETCH_SIZE"), pmc_search(t, "WRITE_SIZE")
    busy, gui = pmc_search(t, "SQ_VALU_MFMA_BUSY_CYCLES"), pmc_search(t, "GRBM_GUI_ACTIVE")
    hit, req = pmc_search(t, "TCC_HIT_sum"), pmc_search(t, "TCC_REQ_sum")
    if fs and ws and busy and gui:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the chip has 1024 SIMDs
        pj["atari"] = {
            "hbm_bytes_per_launch": int((2 * fs + ws) * 1024), "fetch_size_kb_raw": fs, "write_size_kb_raw": ws,
            "correction": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024", "simulations_per_launch": sims,
            "kernel": "mz::mz_resnet_search_kernel<false, true, true, 2> (MuZero policy, pair mode, tree statistics in LDS, two action slots; 128 roots)",
            "mfma_busy": {"value": round(busy / (gui / 8 * 1024), 4), "simulations_per_launch": sims,
                          "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)"},
            "l2_hit_rate": None if not (hit and req) else round(hit / req, 4),
            "source": f"profiles/{tag}_rocprofv3_atari128.txt (tools/rocprof_tower_pmc.sh {sims}: separate --pmc passes)",
            "note": "memory-side requests of the 8 L2s (Infinity-Cache hits are counted): every XCD re-streams the 5.7 MB of "
                    "convolution weights once per simulation (they do not fit its 4 MB L2; the workgroups of an XCD walk them "
                    "roughly in phase, hence the L2 hit rate) + the embedding rows and the pair messages"}
        json.dump(pj, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
        print("pmc_traffic.json: atari entry refreshed")

# ---- "Headline numbers" table of profiles/README.md, generated from <tag>_bench.json (the hand-written
# table went stale); everything between the two markers is replaced
line = d
c3, c4, c5 = line.get("config3_lunarlander", {}), line.get("config4_atari", {}), line.get("config5_gumbel_train", {})
rows = [
    ("CartPole 4096 roots x 50 sims (BASELINE config 2, the metric)",
     f"**{line['value'] / 1e6:.1f} M** env-steps/s synced ({line['ms_per_step']:.4f} ms per act); un-settled {((line.get('value_unsettled') or 0) / 1e6):.1f} M; "
     f"{line['value_pipelined'] / 1e6:.1f} M back to back; `MuZero.act` NumPy in/out {line.get('api', {}).get('numpy', {}).get('value', 0) / 1e6:.1f} M",
     f"`{line['roofline']['kernel']}` {line['roofline']['kernel_ms'] * 1e3:.1f} us (HIP events)",
     f"{line['roofline']['achieved']:.0f} GB/s algorithmic = {line['roofline']['frac']:.3f} of 8 TB/s; {(line['roofline'].get('traffic') or 0) / 1e6:.2f} MB of real HBM traffic"),
]
if "roofline" in c3:
    rows.append(("LunarLander 8192 roots x 50 sims (config 3)", f"{c3['value'] / 1e6:.1f} M synced / {c3['value_pipelined'] / 1e6:.1f} M back to back",
                 f"{c3['roofline']['kernel_ms'] * 1e3:.1f} us", f"{c3['roofline']['frac']:.3f}"))
if "roofline" in c4:
    r4 = c4["roofline"]
    rows.append(("Atari-shaped search, 128 roots x 200 sims, ResNet nets (config 4, one GPU's shard)",
                 f"**{c4['ms_per_act']:.2f} ms** per act (pair mode survived: {c4.get('pair_mode_survived')})",
                 f"ONE `mz_resnet_search_kernel` launch, {r4['kernel_ms']:.2f} ms; stand-alone pass {c4['recurrent_pass']['kernel_ms'] * 1e3:.1f} us",
                 f"fp32 MFMA {r4['achieved']:.1f} TFLOP/s = **{r4['frac']:.3f}**; one pass {c4['recurrent_pass']['frac']:.3f}; matrix pipes busy "
                 f"{(r4.get('mfma_busy') or {}).get('value')}; {(r4.get('traffic') or 0) / 1e9:.1f} GB memory-side per launch"))
if "act" in c5:
    rows.append(("Gumbel act + training step (config 5, one GPU)", f"act {c5['act']['ms_per_act']:.3f} ms synced, `update()` {c5['update']['ms_per_update']:.3f} ms",
                 "one fused launch each", "-"))
table = f"<!-- headline:{tag}:begin (generated by tools/collect_round_profiles.py from {tag}_bench.json) -->\n" \
        f"Headline numbers, {tag} (generated from `{tag}_bench.json`):\n\n| | value | kernel | roofline |\n|---|---|---|---|\n" + \
        "".join(f"| {a} | {b} | {c} | {e} |\n" for a, b, c, e in rows) + f"<!-- headline:{tag}:end -->\n"
readme = os.path.join(P, "README.md")
txt = open(readme).read()
b, e = f"<!-- headline:{tag}:begin", f"<!-- headline:{tag}:end -->\n"
if b in txt:
    txt = txt[:txt.index(b)] + table + txt[txt.index(e) + len(e):]
else:
    marker = "Headline numbers, round 4"
    txt = txt.replace(marker, table + "\n" + marker, 1) if marker in txt else txt + "\n" + table
open(readme, "w").write(txt)
print("profiles/README.md: headline table of", tag, "regenerated")
