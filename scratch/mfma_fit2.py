#!/usr/bin/env python3
"""Second-stage fit: per block of 8 k, the 8 exact products and the accumulator are aligned to the largest exponent, truncated to W
bits below it, summed exactly and rounded once (RNE) to fp32."""
import sys
import numpy as np
from fractions import Fraction

f = open(sys.argv[1], "rb")
n = int(np.frombuffer(f.read(4), np.int32)[0])
A = np.frombuffer(f.read(n * 512 * 2), np.uint16).reshape(n, 16, 32)
B = np.frombuffer(f.read(n * 512 * 2), np.uint16).reshape(n, 16, 32)
C = np.frombuffer(f.read(n * 256 * 4), np.float32).reshape(n, 16, 16)
D = np.frombuffer(f.read(n * 256 * 4), np.float32).reshape(n, 16, 16)
bf = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
Af, Bf = bf(A).astype(np.float64), bf(B).astype(np.float64)


def rne32(x):          # x: python Fraction -> fp32 (round to nearest even); fp64 has enough bits here when |terms| aligned -> use exact path
    return np.float32(float(x))      # float(Fraction) is correctly rounded to fp64; fp64->fp32 double rounding is rare but possible


def block_sum(terms, W, mode):
    nz = [t for t in terms if t != 0]
    if not nz:
        return 0.0
    E = max(int(np.floor(np.log2(abs(t)))) for t in nz)
    q = 2.0 ** (E - W)
    s = 0
    for t in nz:
        k = t / q
        k = np.trunc(k) if mode == "trunc" else np.floor(k)
        s += int(k)
    return s * q


def model(t, m, nn, W, mode, cmode):
    acc = float(C[t, m, nn])
    for k0 in range(0, 32, 8):
        prods = [Af[t, m, k] * Bf[t, nn, k] for k in range(k0, k0 + 8)]
        if cmode == "in":
            acc = float(np.float32(block_sum([acc] + prods, W, mode)))
        else:
            acc = float(np.float32(acc + block_sum(prods, W, mode)))
    return np.float32(acc)


rng = np.random.default_rng(0)
for mode_id in (0, 1, 2, 6, 7):
    tiles = [t for t in range(0, 128) if t % 8 == mode_id]
    pts = [(t, int(rng.integers(16)), int(rng.integers(16))) for t in tiles for _ in range(24)]
    print("mode", mode_id)
    for cmode in ("in", "out"):
        for tm in ("trunc", "floor"):
            for W in (23, 24, 25, 26, 27, 28, 30, 32, 40, 52):
                eq = sum(model(t, m, nn, W, tm, cmode).view(np.uint32) == D[t, m, nn].view(np.uint32) for t, m, nn in pts)
                print(f"   c {cmode:3s} {tm:5s} W={W:2d}: {eq}/{len(pts)}")
