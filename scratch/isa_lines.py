#!/usr/bin/env python3
"""Static instruction counts per source line from a `hipcc -S -gline-tables-only` listing.
usage: isa_lines.py file.s [kernel-substring] [-f file-substring] [-m] ; -m: per-mnemonic histogram of the selected lines"""
import collections
import re
import sys

def main():
    path = sys.argv[1]
    ksub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else None
    fsub = None; mn = False; lo = hi = None
    a = sys.argv[2:]
    for i, x in enumerate(a):
        if x == '-f': fsub = a[i + 1]
        if x == '-m': mn = True
        if x == '-r': lo, hi = int(a[i + 1]), int(a[i + 2])
    files = {}
    cur = None
    per = collections.defaultdict(collections.Counter)
    hist = collections.Counter()
    inker = ksub is None
    for ln in open(path):
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
            continue
        if ksub is not None:
            if re.match(r'^_Z\w+:', ln):
                inker = ksub in ln
            if not inker: continue
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
            continue
        m = re.match(r'^([sv]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)\b', s)
        if not m or cur is None: continue
        op = m.group(1)
        cls = 'valu' if op.startswith('v_') and 'mfma' not in op else ('mfma' if 'mfma' in op else ('salu' if op.startswith('s_') else ('lds' if op.startswith('ds_') else 'vmem')))
        per[cur][cls] += 1
        if (fsub is None or fsub in cur[0]) and (lo is None or lo <= cur[1] <= hi):
            hist[op] += 1
    tot = collections.Counter()
    for (f, l), c in sorted(per.items()):
        if fsub and fsub not in f: continue
        if lo is not None and not (lo <= l <= hi): continue
        tot.update(c)
        if not mn: print(f"{f}:{l:5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("total", dict(tot))
    if mn:
        for op, n in hist.most_common(60): print(f"{n:6d} {op}")

main()
