#!/usr/bin/env python3
"""Loops (backward branches) of a kernel from scratch/isa_phases.py output: body range, phase, instruction mix."""
import collections, json, re, sys
sys.path.insert(0, "scratch")
from isa_table import load_phases, phase_of
d = json.load(open(sys.argv[1]))
ph = load_phases("alphagpu_amd/csrc/agz_tree_reg.hpp")
ins = d["ins"]
addr_idx = {x["addr"]: i for i, x in enumerate(ins)}
def ph_of(x):
    body = None
    for fn, f, ln in x["frames"]:
        if "rollout_reg_body" in fn and f == "agz_tree_reg.hpp":
            body = ln
    if body is None:
        return "network" if any("mlp_wave_body" in fn for fn, _, _ in x["frames"]) else "loop"
    return phase_of(body, ph)
loops = []
for i, x in enumerate(ins):
    if x["mn"].startswith("s_cbranch") or x["mn"] == "s_branch":
        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", x["txt"])
        if not m:
            continue
        base = ins[0]["addr"]
        tgt = base + int(m.group(1), 16)
        if tgt <= x["addr"] and tgt in addr_idx:
            j = addr_idx[tgt]
            body = ins[j:i + 1]
            c = collections.Counter(y["cls"] for y in body)
            p = collections.Counter(ph_of(y) for y in body)
            lines = sorted({ln for y in body for fn, f, ln in y["frames"] if "rollout_reg_body" in fn})
            loops.append((tgt - base, x["addr"] - base, dict(c), p.most_common(3), (lines[0], lines[-1]) if lines else None))
for l in sorted(loops):
    print(f"{l[0]:6x}-{l[1]:6x} {l[2]} {l[3]} lines {l[4]}")
