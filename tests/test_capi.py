"""C-ABI surface without a GPU: the library loads, exports every symbol include/agz.h declares, answers the
GPU-free queries, and refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import alphagpu_amd as ag
from alphagpu_amd import lib as aglib
from alphagpu_amd import mcts_gpu as M
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read()
    names = sorted(set(re.findall(r"\b(agz_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 30
    L = aglib.load_library()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


@pytest.mark.parametrize("game,n,k,A,VS,ML,img", [
    ("gobang", 3, 3, 9, 9, 9, 104), ("gobang", 9, 5, 81, 81, 81, 104), ("connect4", 0, 0, 7, 42, 42, 104),
    ("hex", 9, 0, 81, 100, 81, 104), ("reversi8", 0, 0, 65, 64, 70, 152), ("reversi6", 0, 0, 37, 36, 50, 152)])
def test_game_constants_match_reference_and_oracle(game, n, k, A, VS, ML, img):
    g = ag.GameSpec(game, n, k)
    assert (g.A, g.VS, g.FS, g.ML, g.pos_image_bytes) == (A, VS, VS, ML, img)
    og = O.make_game(game, n, k)
    assert (og.A, og.VS, og.FS, og.ML) == (g.A, g.VS, g.FS, g.ML)


def test_bad_game_parameters_rejected():
    with pytest.raises(ValueError):
        ag.GameSpec("gobang", 14, 5)
    with pytest.raises(ValueError):
        ag.GameSpec("hex", 13, 0)


def test_weight_init_matches_oracle_definition():
    g = ag.GameSpec("connect4")
    net = ag.SNetwork2.random(g, 64, 3, seed=123)
    on = O.OracleNet(O.make_game("connect4"), 64, 3, seed=123)
    for a, b in ((net.W0, on.W0), (net.Wres, on.Wres), (net.Wp, on.Wp), (net.Wv, on.Wv), (net.bp, on.bp)):
        assert np.array_equal(a, b)
    lim = np.sqrt(6.0 / (84 + 64))
    assert np.abs(net.W0).max() <= lim and np.abs(net.W0).max() > 0.9 * lim


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aglib.AgzError) as e:
        M.Engine(ag.GameSpec("gobang", 3, 3), 4, 4)
    assert "no HIP device" in str(e.value)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "alphagpu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                # comments may SAY "oracle"; nothing may include, import, link or load it
                for needle in ("agz_oracle", "oracle_lib", "libagz_oracle", "oracle/", "agzo_", "import oracle"):
                    assert needle not in txt, (f, needle)


def test_pool_sample_ring_semantics():
    g = ag.GameSpec("gobang", 3, 3)
    buf = ag.PoolSample(g, 5)
    st = np.arange(2 * 18, dtype=np.int8).reshape(2, 18) % 2
    pol = np.ones((2, 9), np.float32) / 9
    idx = [buf.push_buffer(st, pol, 1 if i % 2 == 0 else -1, i % 2) for i in range(7)]
    assert idx == [1, 2, 3, 4, 5, 1, 2] and buf.full and buf.currentIndex == 3 and buf.length_buffer() == 5
    buf.update_buffer([1, 2], -1, np.ones(9, np.int8))
    assert buf.value[0] == 1.0 and buf.value[1] == 0.0 and (buf.fstate[0] == -1).all() and (buf.fstate[1] == 1).all()


@pytest.mark.parametrize("game,n,k", [("gobang", 9, 5), ("reversi8", 0, 0), ("connect4", 0, 0), ("hex", 9, 0)])
def test_packed_record_layout_is_the_one_agz_h_documents(game, n, k):
    """include/agz.h (agz_get_samples_packed): {u32 game_id, i32 ply, i32 move, f32 value, i8 player, i8 pad[3],
    f32 policy[A], i8 state[2VS], i8 fstate[FS], pad to 16 B}.  A record written field by field with struct.pack must decode
    through shard.unpack_records (the consumer of the all-gather) — pins the layout on both sides of the exchange."""
    import struct
    from alphagpu_amd import shard
    g = ag.GameSpec(game, n, k)
    body = 20 + 4 * g.A + 2 * g.VS + g.FS
    assert g.rec_bytes == (body + 15) // 16 * 16                       # padded to a 16-B multiple
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read()
    assert "u32 game_id, i32 ply, i32 move, f32 value" in hdr and "f32 policy[A], i8 state[2VS], i8 fstate[FS]" in hdr
    rng = np.random.default_rng(1)
    recs, want = [], []
    for i in range(3):
        pol = rng.random(g.A).astype(np.float32)
        st = rng.integers(0, 2, 2 * g.VS).astype(np.int8)
        fs = rng.integers(-1, 2, g.FS).astype(np.int8)
        r = struct.pack("<IiifbBBB", 7000 + i, i, 3 + i, 0.5 * i, -1 if i & 1 else 1, 0, 0, 0) + pol.tobytes() + st.tobytes() + fs.tobytes()
        recs.append(r + bytes(g.rec_bytes - len(r)))
        want.append((7000 + i, i, 3 + i, 0.5 * i, -1 if i & 1 else 1, pol, st, fs))
    u = shard.unpack_records(np.frombuffer(b"".join(recs), np.uint8), 3, g)
    for i, (gid, ply, mv, val, pl, pol, st, fs) in enumerate(want):
        assert (u["game_id"][i], u["ply"][i], u["move"][i], u["value"][i], u["player"][i]) == (gid, ply, mv, val, pl)
        assert np.array_equal(u["policy"][i], pol) and np.array_equal(u["state"][i], st) and np.array_equal(u["fstate"][i], fs)


def test_snetwork2_accepts_flux_matrices_in_column_major_order():
    """Flux stores Dense(in,out).weight as an (out,in) column-major matrix: W[o + out*i].  A 2-D numpy (out,in) array must
    land in that memory order (not be silently transposed); 1-D input is taken as already in that order."""
    g = ag.GameSpec("gobang", 3, 3)
    H = 4
    W0 = np.arange(H * 18, dtype=np.float32).reshape(H, 18)             # W0[o, i]
    net = ag.SNetwork2(g, H, 1, W0=W0, Wres=np.zeros((1, H, H), np.float32), Wp=np.zeros((9, H), np.float32), Wv=np.zeros((1, H), np.float32))
    assert net.W0[2 + H * 5] == W0[2, 5]
    flat = ag.SNetwork2(g, H, 1, W0=net.W0.copy())
    assert np.array_equal(flat.W0, net.W0)
    with pytest.raises(ValueError):
        ag.SNetwork2(g, H, 1, W0=W0.T)


def test_wrappers_draw_a_fresh_seed_per_call():
    seeds = {M.fresh_seed() for _ in range(64)}
    assert len(seeds) == 64 and all(0 < s < 2 ** 63 for s in seeds)


def test_every_environment_switch_is_documented_in_the_header():
    """agz_create reads debug / A-B switches from the environment; include/agz.h lists every one of them (and nothing stale)."""
    csrc = os.path.join(ROOT, "alphagpu_amd", "csrc")
    used = set()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")):
            used |= set(re.findall(r'getenv\("(AGZ_[A-Z0-9_]+)"\)', open(os.path.join(csrc, f)).read()))
    header = open(os.path.join(ROOT, "include", "agz.h")).read()
    block = header[header.index("Environment switches read by agz_create"):]
    documented = set(re.findall(r"\b(AGZ_[A-Z0-9_]+)\b", block))
    assert used and used <= documented, sorted(used - documented)
    assert documented <= used, sorted(documented - used)


def test_push_packed_equals_push_generation_with_wrap_and_overflow():
    """PoolSample.push_packed (agz_unpack_records: host-only, straight into the ring's arrays, segment by segment across the end of the
    ring) == push_generation of the same samples — also when the write wraps and when one push brings more samples than the ring holds."""
    import alphagpu_amd as ag
    import oracle_lib as O
    from test_shard_gloo import _pack
    game = ag.GameSpec("gobang", 3, 3)
    og = O.make_game("gobang", 3, 3)
    s = O.selfplay(og, O.OracleNet(og, 16, 1), 40, 8, 1.5, 25, 3, 0)
    recs = _pack(s, game)
    n = s["n"]
    for length in (n + 50, n - 7, n // 2 + 3, n // 3):
        a, b = ag.PoolSample(game, length), ag.PoolSample(game, length)
        for _ in range(3):
            ia = a.push_packed(recs, n)
            ib = b.push_generation({k: s[k] for k in ("state", "policy", "player", "value", "fstate")})
            assert np.array_equal(ia, ib) and a.currentIndex == b.currentIndex and a.full == b.full
            for k in ("state", "policy", "player", "value", "fstate"):
                assert np.array_equal(getattr(a, k), getattr(b, k)), (length, k)
