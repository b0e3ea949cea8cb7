#!/usr/bin/env python3
"""Fit an arithmetic model of v_mfma_f32_16x16x32_bf16 to the tiles dumped by scratch/mfma_probe (gpurun_out/.../mfma_probe.bin)."""
import sys
import numpy as np

f = open(sys.argv[1], "rb")
n = int(np.frombuffer(f.read(4), np.int32)[0])
A = np.frombuffer(f.read(n * 512 * 2), np.uint16).reshape(n, 16, 32)
B = np.frombuffer(f.read(n * 512 * 2), np.uint16).reshape(n, 16, 32)
C = np.frombuffer(f.read(n * 256 * 4), np.float32).reshape(n, 16, 16)
D = np.frombuffer(f.read(n * 256 * 4), np.float32).reshape(n, 16, 16)
bf = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
Af, Bf = bf(A).astype(np.float64), bf(B).astype(np.float64)


def model_seq_fp32(t, order):
    acc = C[t].astype(np.float32).copy()
    for k in order:
        p = (Af[t][:, None, k] * Bf[t][None, :, k])
        acc = (acc.astype(np.float64) + p).astype(np.float32)
    return acc


def model_exact_then_round(t):
    s = C[t].astype(np.float64) + np.einsum("mk,nk->mn", Af[t], Bf[t])
    return s.astype(np.float32)


def model_blocks(t, blk):
    acc = C[t].astype(np.float32).copy()
    for k0 in range(0, 32, blk):
        p = np.einsum("mk,nk->mn", Af[t][:, k0:k0 + blk], Bf[t][:, k0:k0 + blk])
        acc = (acc.astype(np.float64) + p).astype(np.float32)
    return acc


def score(name, fn, tiles):
    eq = tot = 0
    for t in tiles:
        m = fn(t)
        eq += int((m.view(np.uint32) == D[t].view(np.uint32)).sum()); tot += 256
    print(f"  {name:40s} exact {eq}/{tot} = {eq / tot:.4f}")


for mode in range(8):
    tiles = [t for t in range(0, 256) if t % 8 == mode]
    print("mode", mode)
    score("sequential fp32, k ascending", lambda t: model_seq_fp32(t, range(32)), tiles)
    score("one rounding (fp64 sum)", model_exact_then_round, tiles)
    for blk in (2, 4, 8, 16):
        score(f"blocks of {blk} exact, fp32 between", lambda t, b=blk: model_blocks(t, b), tiles)
