#!/usr/bin/env python3
"""Loop nest of a kernel listing (hipcc -S -gline-tables-only): every backward branch closes a loop; prints the loops with their
static instruction counts by class (own = excluding nested loops) and the source lines they cover.
usage: isa_loops2.py file.s kernel-substring"""
import collections, re, sys
path, ksub = sys.argv[1], sys.argv[2]
files = {}; ins = []; labels = {}; cur = None; inker = False
for ln in open(path):
    s = ln.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]; continue
    if re.match(r'^_Z\w+:', ln): inker = ksub in ln; continue
    if not inker: continue
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m: labels[m.group(1)] = len(ins); continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m: cur = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    m = re.match(r'^([sv]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)\b(.*)', s)
    if m: ins.append((m.group(1), m.group(2), cur))
def cls(op):
    if 'mfma' in op: return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    return 'vmem'
loops = []
for i, (op, rest, loc) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        t = rest.strip().split()[-1]
        if t in labels and labels[t] <= i: loops.append((labels[t], i))
loops = sorted(set(loops), key=lambda x: (x[0], -x[1]))
# merge loops with same head (keep the widest)
byhead = {}
for a, b in loops: byhead[a] = max(byhead.get(a, b), b)
loops = sorted(byhead.items(), key=lambda x: (x[0], -x[1]))
def count(a, b, skip):
    c = collections.Counter(); lines = collections.Counter()
    for i in range(a, b + 1):
        if any(x <= i <= y for x, y in skip): continue
        c[cls(ins[i][0])] += 1
        if ins[i][2]: lines[ins[i][2]] += 1
    return c, lines
print("instructions", len(ins))
for a, b in loops:
    depth = sum(1 for x, y in loops if x <= a and b <= y and (x, y) != (a, b))
    inner = [(x, y) for x, y in loops if a <= x and y <= b and (x, y) != (a, b)]
    c, lines = count(a, b, inner)
    tot, _ = count(a, b, [])
    fl = collections.defaultdict(list)
    for (f, l), n in lines.items(): fl[f].append(l)
    span = "; ".join(f"{f}:{min(v)}-{max(v)}" for f, v in sorted(fl.items(), key=lambda kv: -len(kv[1]))[:3])
    print(f"{'  ' * depth}loop [{a},{b}] own {dict(c)} total-valu {tot['valu']}  {span}")
