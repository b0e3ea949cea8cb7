#!/usr/bin/env python3
"""Per-phase static instruction table from scratch/isa_phases.py output.  usage: isa_table.py /tmp/isa.json [phasefile]"""
import collections
import json
import sys

PH = [  # (first line, last line, phase) of agz_tree_reg.hpp::rollout_reg_body — keep in sync with the source (PHASE markers)
]


def load_phases(path):
    """lines of the form  `// PHASE name`  in agz_tree_reg.hpp open a phase that lasts until the next marker"""
    ph = []
    for i, ln in enumerate(open(path), 1):
        if "// PHASE " in ln:
            ph.append((i, ln.split("// PHASE ")[1].strip()))
    return ph


def phase_of(line, ph):
    name = "prologue"
    for l0, n in ph:
        if line >= l0:
            name = n
        else:
            break
    return name


BODY, SRC = "rollout_eager_body", "agz_tree_eager.hpp"


def main():
    global BODY, SRC
    d = json.load(open(sys.argv[1]))
    if len(sys.argv) > 2 and sys.argv[2] == "reg":
        BODY, SRC = "rollout_reg_body", "agz_tree_reg.hpp"
    ph = load_phases("alphagpu_amd/csrc/" + SRC)
    # optional filter: only the instructions of the copy of the body whose template arguments end with this text (a kernel with age classes
    # holds two copies of the rollout loop: "... 8, 8>" rows by legal rank, "... 0, 8>" rows by action); the network body is split by its caller
    variant = sys.argv[3] if len(sys.argv) > 3 else None
    tab = collections.defaultdict(collections.Counter)
    for x in d["ins"]:
        body = None
        if variant is not None:
            names = [fn for fn, f, ln in x["frames"] if ("rollout_eager_body<" in fn or "persist_search<" in fn)]
            if not any(variant in fn.split("::")[0] + ">" or fn.rstrip().endswith(variant) or (variant in fn) for fn in names):
                continue
        for fn, f, ln in x["frames"]:
            if BODY in fn and f == SRC:
                body = ln
        if body is None:
            if any("mlp_wave_body" in fn for fn, _, _ in x["frames"]):
                p = "network body"
            else:
                p = "kernel loop"
        else:
            p = phase_of(body, ph)
        tab[p][x["cls"]] += 1
    cols = ["valu", "salu", "lds", "vmem", "mfma", "other"]
    print(f"{'phase':34s}" + "".join(f"{c:>7s}" for c in cols))
    tot = collections.Counter()
    order = [n for _, n in ph] + ["prologue", "network body", "kernel loop"]
    seen = []
    for p in order:
        if p in tab and p not in seen:
            seen.append(p)
            print(f"{p:34s}" + "".join(f"{tab[p][c]:7d}" for c in cols))
            tot.update(tab[p])
    print(f"{'total':34s}" + "".join(f"{tot[c]:7d}" for c in cols))
    json.dump({p: dict(tab[p]) for p in tab}, open(sys.argv[1].replace(".json", "_table.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
