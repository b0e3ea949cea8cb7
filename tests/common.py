"""Shared helpers for the parity tests (test infrastructure; uses the oracle)."""
import ctypes as C

import numpy as np

import oracle_lib as O

# (name, oracle kind, n, nvict): the BASELINE.json config families at oracle-sized shapes
GAMES = {
    "tictactoe": ("gobang", 3, 3),
    "gobang9": ("gobang", 9, 5),
    "gobang11": ("gobang", 11, 5),
    "gobang13": ("gobang", 13, 5),
    "connect4": ("connect4", 0, 0),
    "hex5": ("hex", 5, 0),
    "hex9": ("hex", 9, 0),
    "hex11": ("hex", 11, 0),
    "hex12": ("hex", 12, 0),
    "reversi8": ("reversi8", 0, 0),
    "reversi6": ("reversi6", 0, 0),
}


def diverse_roots(g, L, seed, max_prefix=12):
    """L non-terminal positions reached by random legal prefixes of 0..max_prefix plies."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < L:
        p = O.pos_init(g)
        k = int(rng.integers(0, max_prefix + 1)) if len(out) % 4 else 0
        ok = True
        for _ in range(k):
            legal = [a for a in range(g.A) if O.can_play(g, p, a)]
            q = O.play(g, p, legal[int(rng.integers(len(legal)))])
            if O.is_over(g, q)[0]:
                ok = False
                break
            p = q
        if ok:
            out.append(p)
    return out


def pos_bytes(positions):
    return np.frombuffer(b"".join(bytes(p) for p in positions), np.uint8).copy()


def pos_from_bytes(buf):
    buf = np.ascontiguousarray(buf, np.uint8).reshape(-1, 80)
    out = []
    for row in buf:
        p = O.Pos()
        C.memmove(C.byref(p), row.ctypes.data, 80)
        out.append(p)
    return out


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
