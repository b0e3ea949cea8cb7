"""Comparison helpers of the GPU parity tests (test infrastructure; uses the oracle).

bf16 mode is compared BIT FOR BIT against the oracle's model of the bf16 MFMA forward, and A MISMATCH FAILS.

That model was fitted to instruction outputs captured on the hardware (oracle/agz_oracle.c agzo_mfma_dot), so a future seed may hit an
un-modelled rounding case with no product bug behind it.  For that case only — and only when the developer asks for it with
AGZ_ALLOW_MFMA_MODEL_MISS=1 — `assert_bf16_search_matches` runs a diagnosis instead of failing at once: the games that differ are
searched again through the stepwise API with the GPU's own priors and values teacher-forced into the oracle (the trees must then be
bit-identical and the GPU logits within the bf16 bound of the oracle's fp32 forward, DenseNet.jl:294-304), the stepwise run must have
met a network output the model misses, AND the fused one-launch result that mismatched must equal that stepwise GPU run bit for bit
(same MFMA instruction, operand and k order in both forms) — which ties the fused kernel itself, not just the stepwise kernels, to the
teacher-forced oracle run.  The offending network outputs are printed so that the known-answer fixture (tests/golden/mfma_kat.npz) can
be extended.  Without the switch, and in every other case with it, the test fails.
"""
import os
import warnings

import numpy as np

import common
import oracle_lib as O

BF16_LOGIT_REL_FP32 = 2.0 ** -6
BF16_VALUE_TOL_FP32 = 2.0 ** -6
KEYS = ("leaf", "node_count", "visits", "q", "policy")


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype == np.float32:
        return bool((common.bits(a) == common.bits(b)).all())
    return bool(np.array_equal(a, b))


def engine_result(e, rows=None):
    r = dict(leaf=e.leaf(), node_count=e.node_count(), visits=e.root_visits(), q=e.root_q(), policy=e.policy())
    return r if rows is None else {k: v[rows] for k, v in r.items()}


def oracle_result(t):
    return dict(leaf=t.leaf(), node_count=t.newindex(), visits=t.root_visits(), q=t.root_q(), policy=t.policy())


def differing_games(a, b):
    bad = np.zeros(len(a["leaf"]), bool)
    for k in KEYS:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        d = (common.bits(x) != common.bits(y)) if x.dtype == np.float32 else (x != y)
        bad |= d.reshape(len(bad), -1).any(axis=1)
    return np.nonzero(bad)[0]


def teacher_forced_check(make_engine, og, onet, roots, ids, V, cpuct, training, seed, step, what=""):
    """Stepwise search on the GPU (agz_rollout_*), the GPU's softmaxed priors and values handed to the oracle rollout by rollout:
    leaves, planes and the final trees must be bit-identical; logits within the bf16 bound of the fp32 forward.  Returns (the list of
    (rollout, row, column, gpu, model) network outputs that differ from the bf16 MFMA model, the stepwise GPU run's result dict)."""
    L = len(roots)
    t = O.OracleTree(og, L, V)
    t.set_roots(roots, ids)
    t.reset()
    misses = []
    with make_engine(L, V) as e:
        e.set_roots(common.pos_bytes(roots), game_ids=ids)
        e.search_begin(cpuct, training, step)
        for k in range(V):
            e.rollout_select(k, last=(k == V - 1))
            t.select(seed, step, k, cpuct)
            assert same_bits(e.leaf(), t.leaf()), f"{what}: leaf @rollout {k} (teacher-forced)"
            planes = t.encode_leaves()
            assert same_bits(e.leaf_batch(), planes), f"{what}: leaf planes @rollout {k} (teacher-forced)"
            e.rollout_eval()
            pr, v = e.get_eval()
            lg, vv = e.get_logits()
            olg, ov = onet.logits(planes)
            scale = np.maximum(1.0, np.abs(olg).max(axis=1, keepdims=True))
            assert float((np.abs(lg - olg) / scale).max()) <= BF16_LOGIT_REL_FP32, f"{what}: logits outside the bf16 bound @rollout {k}"
            assert float(np.abs(vv - ov).max()) <= BF16_VALUE_TOL_FP32, f"{what}: value outside the bf16 bound @rollout {k}"
            blg, bv = onet.logits_bf16(planes)
            for r, c in zip(*np.nonzero(common.bits(lg) != common.bits(blg))):
                misses.append((k, int(r), int(c), float(lg[r, c]), float(blg[r, c])))
            for r in np.nonzero(common.bits(vv) != common.bits(bv))[0]:
                misses.append((k, int(r), -1, float(vv[r]), float(bv[r])))
            t.expand(pr, training, seed, step, k)
            t.backup(v, seed, step, k)
            e.rollout_expand_backup()
        e.search_end()
        got, ref = engine_result(e), oracle_result(t)
        for key in ("leaf", "node_count", "visits", "q"):
            assert same_bits(got[key], ref[key]), f"{what}: {key} differs with teacher-forced priors"
    return misses, got


def assert_bf16_search_matches(got, ref, fallback, what=""):
    """got / ref: engine_result / oracle_result dicts of the same games.  fallback(rows) -> misses runs teacher_forced_check on the
    games `rows` of the comparison (None: a mismatch fails at once)."""
    bad = differing_games(got, ref)
    if len(bad) == 0:
        assert np.abs(got["q"] - ref["q"]).max() <= 1e-4
        return
    detail = ", ".join(f"{k}: {int((np.asarray(got[k]) != np.asarray(ref[k])).sum())}" for k in KEYS)
    allow = os.environ.get("AGZ_ALLOW_MFMA_MODEL_MISS") == "1" and fallback is not None
    assert allow, (f"{what}: {len(bad)} games differ from the oracle ({detail}); rerun with AGZ_ALLOW_MFMA_MODEL_MISS=1 to find out "
                   f"whether an un-modelled MFMA rounding case is behind it")
    rows = bad[:8]
    misses, stepwise = fallback(rows)
    assert misses, (f"{what}: {len(bad)} games differ from the oracle ({detail}) although every network output of the stepwise "
                    f"re-run equals the bf16 MFMA model")
    for k in KEYS:                          # the fused form that mismatched == the stepwise GPU form that the oracle followed
        assert same_bits(np.asarray(got[k])[rows], stepwise[k]), f"{what}: {k} of the one-launch search differs from the stepwise GPU run"
    msg = (f"{what}: {len(bad)} games differ from the oracle's bf16 MFMA MODEL; with the GPU's priors teacher-forced the trees are "
           f"bit-identical and the logits are within the bf16 bound of the fp32 forward -> un-modelled MFMA rounding case, not a "
           f"product bug.  Network outputs to add to tests/golden/mfma_kat.npz (rollout, row, column, gpu, model): {misses[:8]}")
    print("\n" + msg)
    warnings.warn(msg)


def oracle_selfplay_slices(og, onet, n, V, cpuct, tau_plies, seed, bases):
    """the oracle's lock-step games of the ids base .. base + n - 1 for every base, the slices computed SIDE BY SIDE (the C oracle releases the
    GIL and every call owns its tree; a full-size test checks up to nine slices, each a few seconds of one to sixteen host threads)
    -> {base: O.selfplay result}"""
    from concurrent.futures import ThreadPoolExecutor
    bases = list(bases)
    with ThreadPoolExecutor(max_workers=max(1, len(bases))) as ex:
        refs = list(ex.map(lambda b: O.selfplay(og, onet, n, V, cpuct, tau_plies, seed, b), bases))
    return dict(zip(bases, refs))
