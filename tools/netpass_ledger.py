"""Instruction ledger of the fused kernel's network pass from the ISA of tools/ubench_netpass_e32.hip (-save-temps):
the code between the `; ---- pass begin` / `; ---- pass end` markers, by class, in program order segments.
    python tools/netpass_ledger.py <file.s>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
b = next(i for i, l in enumerate(lines) if "---- pass begin" in l)
e = next(i for i, l in enumerate(lines) if "---- pass end" in l)
if e > b:
    span = lines[b + 1:e]
else:
    # rotated loop: begin .. back edge, then loop head .. end
    lab = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    back = next(i for i in range(b, len(lines)) for m in [re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])]
                if m and lab.get(m.group(1), 1 << 30) < e)
    head = lab[re.search(r"(\.LBB\d+_\d+)", lines[back]).group(1)]
    span = lines[b + 1:back + 1] + lines[head:e]
body = []
labels = 0
for l in span:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not t.endswith(":"):
        continue
    if t.endswith(":"):
        labels += 1
        continue
    body.append(t)


def cls(t):
    op = t.split()[0]
    if op.startswith("v_fmac_f32_dpp") or (op.startswith("v_fmac_f32") and "row_newbcast" in t):
        return "fma chain links (v_fmac_f32_dpp)"
    if op.startswith(("v_pk_fma", "v_fma_f32", "v_fmac", "v_fmaak", "v_fmamk")):
        return "other fma (packed / scalar)"
    if "dpp" in op or "row_" in t or "quad_perm" in t or op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")):
        return "cross-lane (DPP butterflies, broadcasts, readlane)"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_div", "v_ldexp", "v_frexp", "v_rndne", "v_cvt", "v_floor", "v_trunc")):
        return "transcendental / division / conversion"
    if op.startswith("v_pk_"):
        return "packed add / mul / max"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "compare / select"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "moves"
    if op.startswith("v_"):
        return "other VALU (add, mul, max, min, integer)"
    if op.startswith("ds_"):
        return "LDS reads / writes"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "memory"
    if op == "s_nop":
        return "s_nop (DPP / MFMA hazards)"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op.startswith("s_"):
        return "SALU / branches"
    return "other"


c = collections.Counter(cls(t) for t in body)
valu = sum(v for k, v in c.items() if not k.startswith(("LDS", "memory", "s_nop", "s_waitcnt", "SALU", "other")))
print(f"network pass, straight-line body: {len(body)} instructions ({labels} labels inside: rare-path branches), {valu} of them VALU")
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"  {v:5d}  {k}")
nops = sum(int(t.split()[1]) + 1 for t in body if t.startswith("s_nop"))
print(f"  (s_nop wait states in total: {nops})")
# per segment: split at the ds_read_b128 groups and v_exp to show layer boundaries roughly
seg = collections.Counter()
ops = collections.Counter(t.split()[0] for t in body)
print("by opcode:")
for k, v in ops.most_common(40):
    print(f"  {v:5d}  {k}")
