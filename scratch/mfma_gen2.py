#!/usr/bin/env python3
"""Inputs for scratch/mfma_probe2 (second campaign): (1) the value-head dot product of scratch/repro_policy_diff.py step by step
(/tmp/valdot.npz: b, Wv), (2) random SINGLE-BLOCK tiles (only k = 0..7 nonzero) whose products spread over up to 2^16 and whose
accumulator lies between 2^-6 and 2^9 times the largest product.  Writes scratch/mfma_tests2.bin (+ .json for set 1)."""
import sys, json, ctypes as C
import numpy as np
sys.path.insert(0, "tests"); import oracle_lib as O
L = O.lib(); L.agzo_mfma_dot.restype = C.c_float; L.agzo_mfma_dot.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
# (the vectors of set 1 were written by an analysis step of scratch/repro_policy_diff.py's capture; their tiles and the GPU's answers
#  are part of tests/golden/mfma_kat.npz, so without the file only set 2 is generated)
import os
have1 = os.path.exists("/tmp/valdot.npz")
if have1:
    z = np.load("/tmp/valdot.npz"); b, Wv = z["b"], z["Wv"]
tests = []   # (a[32], bcol[32], c)
acc = np.float32(0)
for s in range(4 if have1 else 0):                                        # the four MFMA steps of the 128-long dot, C = the model's running value
    tests.append((b[32*s:32*s+32].copy(), Wv[32*s:32*s+32].copy(), float(acc), f"step {s}"))
    for j in range(4):                                    # ... and every block alone (the other 24 k zeroed), C = the model's value before it
        a = np.zeros(32, np.uint16); a[8*j:8*j+8] = b[32*s+8*j:32*s+8*j+8]
        tests.append((a, Wv[32*s:32*s+32].copy(), float(acc), f"step {s} block {j}"))
        aa = np.ascontiguousarray(b[32*s+8*j:32*s+8*j+8]); ww = np.ascontiguousarray(Wv[32*s+8*j:32*s+8*j+8])
        acc = np.float32(L.agzo_mfma_dot(aa.ctypes.data, ww.ctypes.data, 8, float(acc)))
n1 = (len(tests) + 15) // 16
rng = np.random.default_rng(7)
n2 = 3000
n = n1 + n2
A = np.zeros((n, 16, 32), np.uint16); B = np.zeros((n, 16, 32), np.uint16); Cc = np.zeros((n, 16, 16), np.float32)
for i, (a, w, c, _) in enumerate(tests):
    A[i // 16, i % 16] = a; B[i // 16, i % 16] = w; Cc[i // 16, i % 16, i % 16] = c
def rbf(exp, size):                                       # random bf16 with the given unbiased exponents
    sign = rng.integers(0, 2, size).astype(np.uint16) << 15
    man = rng.integers(0, 128, size).astype(np.uint16)
    return sign | ((np.asarray(exp) + 127).astype(np.uint16) << 7) | man
for t in range(n1, n):
    spread = int(rng.integers(0, 17))
    for m in range(16):
        base = int(rng.integers(-20, 10))
        ex = base - rng.integers(0, spread + 1, 8); ex[rng.integers(8)] = base
        a = rbf(ex, 8)
        a[rng.random(8) < 0.1] = 0
        A[t, m, :8] = a
    for nn in range(16):
        B[t, nn, :8] = rbf(rng.integers(-1, 2, 8), 8)
    for m in range(16):
        ea = ((A[t, m, :8] >> 7) & 0xff).astype(int) - 127
        top = int(ea.max()) if (A[t, m, :8] != 0).any() else 0
        for nn in range(16):
            if rng.random() < 0.08: Cc[t, m, nn] = 0.0
            else:
                e = top + int(rng.integers(-7, 11))
                Cc[t, m, nn] = np.float32((1 if rng.random() < 0.5 else -1) * (1.0 + rng.integers(0, 1 << 23) / float(1 << 23)) * 2.0 ** e)
with open("scratch/mfma_tests2.bin", "wb") as f:
    f.write(np.int32(n).tobytes()); f.write(A.tobytes()); f.write(B.tobytes()); f.write(Cc.tobytes())
json.dump(dict(n1=n1, names=[t[3] for t in tests]), open("scratch/mfma_tests2.json", "w"))
print(n, "tiles;", len(tests), "chain tests in", n1, "tiles")
