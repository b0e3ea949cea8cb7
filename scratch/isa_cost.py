#!/usr/bin/env python3
"""Issue-cost profile of a kernel listing by loop (hipcc -S -gline-tables-only), with the per-class issue costs measured on MI355X
(scratch/ubench/*.hip: plain f32/int VALU 2 cycles per wave64, compares / selects / DPP / converts / lane reads / 3-operand
integer ops / carries 4, packed f32 5, transcendental 8).
usage: isa_cost.py file.s kernel-substring [min-instructions]"""
import collections, re, sys
path, ksub = sys.argv[1], sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 60
def klass(op):
    if not op.startswith('v_'): return None, 0
    if re.match(r'v_(rcp|exp|log|sqrt|rsq|sin|cos)', op): return 'trans', 8
    if re.match(r'v_pk_', op): return 'pk', 5
    if '_dpp' in op or '_sdwa' in op: return 'dpp', 4
    if re.match(r'v_cmp', op): return 'cmp', 4
    if re.match(r'v_cndmask', op): return 'cndmask', 4
    if re.match(r'v_cvt', op): return 'cvt', 4
    if re.match(r'v_(readlane|readfirstlane|writelane)', op): return 'lane', 4
    if re.match(r'v_(add_co|addc_co|sub_co|subb_co|subrev_co)', op): return 'carry', 4
    if re.match(r'v_(mul_lo|mul_hi|mad_u64|mad_i64|mul_u32_u24|mul_i32_i24|mad_u32_u24|mad_i32_i24)', op): return 'imul', 4
    if re.match(r'v_div_', op): return 'div', 4
    if re.match(r'v_(lshl_add|and_or|bfe|add3|lshl_or|max3|min3|med3|alignbit|perm|bfi|add_lshl|or3|xad|mbcnt|bcnt|ffbh|ffbl|bfrev)', op): return 'vop3int', 4
    if 'mfma' in op: return 'mfma', 8
    return 'simple', 2
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
loops = {}
for i, (op, rest, loc) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        t = rest.strip().split()[-1]
        if t in labels and labels[t] <= i: loops[labels[t]] = max(loops.get(labels[t], i), i)
loops = sorted(loops.items(), key=lambda x: (x[0], -x[1]))
print("instructions", len(ins))
def prof(a, b, skip):
    n = collections.Counter(); c = collections.Counter()
    for i in range(a, b + 1):
        if any(x <= i <= y for x, y in skip): continue
        k, w = klass(ins[i][0])
        if k: n[k] += 1; c[k] += w
    return n, c
for a, b in [(0, len(ins) - 1)] + loops:
    if b - a < minins: continue
    depth = sum(1 for x, y in loops if x <= a and b <= y and (x, y) != (a, b))
    inner = [(x, y) for x, y in loops if a <= x and y <= b and (x, y) != (a, b)]
    n, c = prof(a, b, inner)
    tot = sum(c.values()) or 1
    lines = [ins[i][2] for i in range(a, b + 1) if ins[i][2] and not any(x <= i <= y for x, y in inner)]
    fl = collections.defaultdict(list)
    for f, l in lines: fl[f].append(l)
    span = "; ".join(f"{f}:{min(v)}-{max(v)}" for f, v in sorted(fl.items(), key=lambda kv: -len(kv[1]))[:2])
    print(f"{'  ' * depth}loop [{a},{b}] own valu {sum(n.values())} cost {tot} cycles  {span}")
    print(f"{'  ' * depth}   " + "  ".join(f"{k} {n[k]}/{100 * c[k] // tot}%" for k in sorted(c, key=lambda x: -c[x])))
