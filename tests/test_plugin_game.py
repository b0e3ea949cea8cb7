"""A SIXTH game plugged into libagz from outside the library's sources (INTEGRATION.md "Adding a game"): tests/plugin/misere34.hpp is
compiled in with -DAGZ_EXTRA_GAME_HPP (the build recipe of __graft_entry__.build(): tests/plugin/libagz_misere34.so) — the plugin surface
the reference promises (README.md:70 "one bitboard for player one and one for player 2"; Gobang.jl:2,8-11,16-70: Position, canPlay, play,
isOver and the four constants) stays open: nothing in alphagpu_amd/csrc is edited to add the game.

CPU: the plugin library exports the whole C ABI and knows the game's constants; the stock library refuses the game kind.
GPU: the plugin's rules AS THE DEVICE RUNS THEM — breadth-first perft against a naive array model written from the rules — and a whole
self-play generation through the stock kernels instantiated for the new family (every move legal, every result the naive model's)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN_LIB = os.path.join(ROOT, "tests", "plugin", "libagz_misere34.so")


# ---- the rules, naively: 4 x 4, three in a row or a column LOSES, a full board is a draw -------------------------------------------------
def lines3():
    out = []
    for i2 in range(4):
        for i1 in range(2):
            out.append([4 * i2 + i1 + k for k in range(3)])        # down a column of the bitboard (i1 runs fastest)
    for i1 in range(4):
        for i2 in range(2):
            out.append([4 * (i2 + k) + i1 for k in range(3)])      # along a row
    return out


LINES = lines3()


def naive_over(board, mover):
    """board: 16 cells in {0, +1, -1}; mover: the colour that has just moved.  -> (over, winner)"""
    if any(all(board[c] == mover for c in ln) for ln in LINES):
        return True, -mover
    return (all(board), 0) if all(board) else (False, 0)


def naive_perft(depth):
    """(positions after exactly `depth` plies, [finished games with result +1, 0, -1 at any ply <= depth])"""
    level, term = [((0,) * 16, 1)], [0, 0, 0]
    for _ in range(depth):
        nxt = []
        for board, player in level:
            for a in range(16):
                if board[a]:
                    continue
                b = list(board); b[a] = player
                over, w = naive_over(b, player)
                if over:
                    term[{1: 0, 0: 1, -1: 2}[w]] += 1
                nxt.append((tuple(b), -player, over))
        level = [(b, p) for b, p, over in nxt if not over]
        last = len(nxt)
    return last, term


def _use_plugin_lib():
    import alphagpu_amd.lib as aglib
    aglib._LIB = None
    aglib.LIB_PATH = PLUGIN_LIB
    return aglib


@pytest.fixture
def plugin(monkeypatch):
    if not os.path.exists(PLUGIN_LIB):
        pytest.skip("tests/plugin/libagz_misere34.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    import alphagpu_amd.lib as aglib
    old = (aglib._LIB, aglib.LIB_PATH)
    _use_plugin_lib()
    yield aglib
    aglib._LIB, aglib.LIB_PATH = old


def test_plugin_library_exports_the_c_abi_and_knows_the_game(plugin):
    import alphagpu_amd as ag
    L = plugin.load_library()
    hdr = open(os.path.join(ROOT, "include", "agz.h")).read()
    for sym in sorted(set(re.findall(r"\b(agz_\w+)\s*\(", hdr))):
        assert hasattr(L, sym), sym
    g = ag.GameSpec("extra", 4, 3)
    assert (g.A, g.VS, g.FS, g.ML, g.max_plies) == (16, 16, 16, 16, 16) and g.rec_bytes == (20 + 4 * 16 + 2 * 16 + 16 + 15) // 16 * 16
    with pytest.raises(ValueError):
        ag.GameSpec("extra", 5, 3)                                 # the plugin's own parameter check
    assert ag.GameSpec("gobang", 9, 5).A == 81                     # the built-in games are all still there


def test_stock_library_refuses_the_extra_game_kind():
    import alphagpu_amd as ag
    with pytest.raises(ValueError):
        ag.GameSpec("extra", 4, 3)


def test_the_recipe_in_integration_md_names_what_the_plugin_defines():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    plug = open(os.path.join(ROOT, "tests", "plugin", "misere34.hpp")).read()
    for needle in ("AGZ_EXTRA_GAME_HPP", "Game<F_EXTRA, NC>", "extra_game_par", "AGZ_EXTRA_COMBOS", "AGZ_EXTRA_SHAPES", "tests/plugin/misere34.hpp"):
        assert needle in txt, needle
    for needle in ("Game<F_EXTRA, NC>", "extra_game_par", "AGZ_EXTRA_COMBOS", "AGZ_EXTRA_SHAPES", "canPlay", "play", "isOver"):
        assert needle in plug, needle


@pytest.mark.gpu
def test_plugin_rules_on_the_device_equal_the_naive_model(plugin):
    import alphagpu_amd as ag
    from alphagpu_amd.game import perft
    g = ag.GameSpec("extra", 4, 3)
    for depth in range(1, 6):
        nodes, term = perft(g, depth)
        want_nodes, want_term = naive_perft(depth)
        assert (nodes, term) == (want_nodes, want_term), (depth, nodes, term, want_nodes, want_term)
    assert perft(g, 5)[1][0] + perft(g, 5)[1][2] > 0               # somebody has lost by ply 5: the rule is live


@pytest.mark.gpu
@pytest.mark.parametrize("persist", ["0", "1"])
def test_plugin_game_plays_whole_generations_through_the_stock_kernels(plugin, persist, monkeypatch):
    """Search, ply loop and sample capture instantiated for the plugged-in family (one launch per ply, and the persistent form): every
    recorded move is legal, every game ends where the naive model says it ends, with the naive model's result."""
    monkeypatch.setenv("AGZ_PERSIST", persist)
    import alphagpu_amd as ag
    from alphagpu_amd import mcts_gpu as M
    g = ag.GameSpec("extra", 4, 3)
    net = ag.SNetwork2.random(g, 128, 2)
    n = 200
    with M.Engine(g, 64, 16, seed=3, nn_mode=M.NN_BF16, sample_capacity_games=n) as e:
        e.set_network(net)
        st = e.selfplay(n, 16, cpuct=1.5, tau_plies=25)
        assert e.search_form()[0].startswith("k_selfplay_small" if persist == "1" else "k_search_small"), e.search_form()
        s = e.samples()
    assert st["valid"] and st["wins"] + st["draws"] + st["losses"] == n
    results = {1: 0, 0: 0, -1: 0}
    for gid in np.unique(s["game_id"]):
        rows = np.nonzero(s["game_id"] == gid)[0]
        rows = rows[np.argsort(s["ply"][rows])]
        board, player, over = [0] * 16, 1, False
        for k, i in enumerate(rows):
            assert not over and s["ply"][i] == k and s["player"][i] == player
            planes = s["state"][i]                                 # side to move first
            assert [int(x) for x in planes[:16]] == [1 if c == player else 0 for c in board]
            assert [int(x) for x in planes[16:]] == [1 if c == -player else 0 for c in board]
            assert abs(float(s["policy"][i].sum()) - 1.0) < 1e-3 and all(s["policy"][i][a] == 0 for a in range(16) if board[a])
            a = int(s["move"][i])
            assert board[a] == 0, "illegal move recorded"
            board[a] = player
            over, w = naive_over(board, player)
            player = -player
        assert over, "the game's samples stop before the naive model says the game is over"
        results[w] += 1
        assert all(float(s["value"][i]) == (1 + w * int(s["player"][i])) / 2.0 for i in rows)
    assert (st["wins"], st["draws"], st["losses"]) == (results[1], results[0], results[-1])
