#!/usr/bin/env python3
"""Structured inputs for scratch/mfma_probe2: B = 1 (so products = A entries), row m of a tile = one set of 32 products,
column n = one accumulator value.  Writes scratch/mfma_tests.bin and scratch/mfma_tests.json (the test descriptions)."""
import json
import numpy as np

def bf16(x):
    u = np.array([x], np.float32).view(np.uint32)[0]
    assert (u & 0xFFFF) == 0, x
    return np.uint16(u >> 16)

rows = []          # each: list of 32 floats
def row(big_pos, big, small_pos, small):
    r = [0.0] * 32
    if big_pos is not None:
        r[big_pos] = big
    for p in small_pos:
        r[p] = small
    return r

for big_pos, smalls in ((None, (0, 1, 2, 3, 4, 5, 6)), (0, (1, 2, 3, 4, 5, 6, 7)), (7, (0, 1, 2, 3, 4, 5, 6)), (0, (8, 9, 10, 11, 12, 13, 14)),
                        (8, (0, 1, 2, 3, 4, 5, 6)), (31, (0, 1, 2, 3, 4, 5, 6)), (0, (24, 25, 26, 27, 28, 29, 30)), (0, (1, 9, 17, 25, 2, 10, 18))):
    for sign in (1.0, -1.0):
        for j in range(20, 34):
            for m in (1, 2, 3, 7):
                rows.append(dict(big_pos=big_pos, small_pos=list(smalls[:m]), small=sign * 2.0 ** -j, j=j, m=m, sign=sign,
                                 vals=row(big_pos, 1.0, smalls[:m], sign * 2.0 ** -j)))
# big = 1.5 / 1.9921875 (all 8 significand bits) variants for sticky / guard behaviour
for big in (1.5, 1.9921875, 1.0078125):
    for sign in (1.0, -1.0):
        for j in range(20, 32):
            for m in (1, 3, 7):
                rows.append(dict(big_pos=0, small_pos=list(range(1, 1 + m)), small=sign * 2.0 ** -j, j=j, m=m, sign=sign, big=big,
                                 vals=row(0, big, range(1, 1 + m), sign * 2.0 ** -j)))
cvals = [0.0, 1.0, -1.0, 2.0 ** -24, 1.0 + 2.0 ** -23, 3.0, -3.0, 2.0 ** -30, 0.5, 1024.0, -2.0 ** -24, 1.0 - 2.0 ** -24, 2.0, 4.0, 1e-3, -0.75]
while len(rows) % 16:
    rows.append(dict(vals=[0.0] * 32, pad=True))
n = len(rows) // 16
A = np.zeros((n, 16, 32), np.uint16)
B = np.full((n, 16, 32), bf16(1.0), np.uint16)
C = np.zeros((n, 16, 16), np.float32)
for i, r in enumerate(rows):
    for k, v in enumerate(r["vals"]):
        A[i // 16, i % 16, k] = bf16(v)
    C[i // 16, i % 16, :] = np.array(cvals, np.float32)
with open("scratch/mfma_tests.bin", "wb") as f:
    f.write(np.int32(n).tobytes()); f.write(A.tobytes()); f.write(B.tobytes()); f.write(C.tobytes())
json.dump(dict(rows=rows, cvals=cvals), open("scratch/mfma_tests.json", "w"))
print(n, "tiles,", len(rows), "product sets x", len(cvals), "accumulators")
