#!/usr/bin/env python3
"""Analyse the structured MFMA tests (scratch/mfma_gen.py -> mfma_probe2): print observed outputs in units of ulp(1)=2^-23."""
import json, sys
import numpy as np
d = json.load(open("scratch/mfma_tests.json"))
rows, cvals = d["rows"], d["cvals"]
f = open(sys.argv[1], "rb")
n = int(np.frombuffer(f.read(4), np.int32)[0])
f.read(n * 512 * 2 * 2)
C = np.frombuffer(f.read(n * 256 * 4), np.float32).reshape(n * 16, 16)
D = np.frombuffer(f.read(n * 256 * 4), np.float32).reshape(n * 16, 16)
def show(filter_fn, ci, title, maxrows=80):
    print("==", title, " c =", cvals[ci])
    k = 0
    for i, r in enumerate(rows):
        if r.get("pad") or not filter_fn(r):
            continue
        exact = sum(float(v) for v in r["vals"]) + float(np.float32(cvals[ci]))
        got = float(D[i, ci])
        big = r.get("big", 1.0)
        print(f"  j={r['j']:2d} m={r['m']} sign={r['sign']:+.0f} bigpos={r['big_pos']} smallpos={r['small_pos'][:3]}.. exact-base={(exact - (big if r['big_pos'] is not None else 0) - cvals[ci]) * 2**23:+.6f}ulp got-base={(got - (big if r['big_pos'] is not None else 0) - float(np.float32(cvals[ci]))) * 2**23:+.3f}ulp")
        k += 1
        if k >= maxrows: break
sel = lambda bp, sp0, sign, m: (lambda r: r.get("big") is None and r["big_pos"] == bp and r["small_pos"][0] == sp0 and r["sign"] == sign and r["m"] == m)
show(sel(0, 1, 1.0, 7), 0, "big k=0, 7 smalls same block, +, c=0")
show(sel(0, 1, 1.0, 1), 0, "big k=0, 1 small same block, +, c=0")
show(sel(0, 8, 1.0, 7), 0, "big k=0, 7 smalls in block 1, +, c=0")
show(sel(None, 0, 1.0, 7), 1, "big in C (1.0), 7 smalls block 0, +")
show(sel(0, 1, -1.0, 7), 0, "big k=0, 7 smalls same block, -, c=0")
show(sel(8, 0, 1.0, 7), 0, "big k=8 (block 1), 7 smalls in block 0, +, c=0")
