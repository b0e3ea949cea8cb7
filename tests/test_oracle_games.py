"""Pins the oracle's game plugins (Bitboard.jl, Gobang.jl, 4IARow.jl, Hex.jl, Reversi*.jl restatement)
with known answers that do NOT come from the reference (it has none):
published perft / game-tree counts and naive array models written from the rules of the games."""
import numpy as np
import pytest

import naive_games as NG
import oracle_lib as O


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(x) for x in O.philox([0] * 4, [0] * 2)] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in O.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_uniform_ranges():
    us = [O.lib().agzo_uniform_search(1, g, 0, 0, d) for g in range(200) for d in range(5)]
    assert min(us) > 0.0 and max(us) <= 1.0
    um = [O.lib().agzo_uniform_move(1, g, s) for g in range(200) for s in range(5)]
    assert min(um) >= 0.0 and max(um) < 1.0


def test_tictactoe_full_game_tree():
    """Gobang N=3,Nvict=3: 255168 complete games; 131184 won by X, 77904 by O, 46080 draws."""
    g = O.make_game("gobang", 3, 3)
    _, term = O.perft(g, O.pos_init(g), 9)
    assert term.tolist() == [131184, 46080, 77904]
    assert int(term.sum()) == 255168


def test_othello_perft():
    g = O.make_game("reversi8")
    p = O.pos_init(g)
    assert [O.perft(g, p, d)[0] for d in range(1, 9)] == [4, 12, 56, 244, 1396, 8200, 55092, 390216]


def test_connect4_perft():
    g = O.make_game("connect4")
    p = O.pos_init(g)
    # 7^d until a column can fill (7^7 - 7 at depth 7); depth 8 from the published perft table
    assert [O.perft(g, p, d)[0] for d in range(1, 9)] == [7, 49, 343, 2401, 16807, 117649, 823536, 5673234]


def _playout(g, naive, rng, check_bits=True, max_plies=400):
    p = O.pos_init(g)
    for ply in range(max_plies):
        fo, ro = O.is_over(g, p)
        fn, rn = naive.is_over()
        assert fo == fn, f"is_over flag mismatch at ply {ply}"
        if fo:
            assert ro == rn, "winner mismatch"
            return ply
        legal_o = [a for a in range(g.A) if O.can_play(g, p, a)]
        legal_n = [a for a in range(naive.A) if naive.can_play(a)]
        assert legal_o == legal_n, f"legal moves mismatch at ply {ply}"
        assert int(p.player) == naive.player
        if check_bits:
            me, op = naive.bits()
            assert O.bb_bits(p.bplayer, g.len) == me.astype(int).tolist()
            assert O.bb_bits(p.bopponent, g.len) == op.astype(int).tolist()
        a = legal_o[rng.integers(len(legal_o))]
        p = O.play(g, p, a)
        naive.play(a)
    raise AssertionError("game did not end")


@pytest.mark.parametrize("n,k", [(3, 3), (5, 4), (9, 5), (13, 5)])
def test_gobang_vs_naive(n, k):
    g = O.make_game("gobang", n, k)
    rng = np.random.default_rng(n * 100 + k)
    for _ in range(25 if n < 13 else 8):
        _playout(g, NG.NaiveLine(n, n, k, False), rng)


def test_connect4_vs_naive():
    g = O.make_game("connect4")
    rng = np.random.default_rng(4)
    for _ in range(60):
        _playout(g, NG.NaiveLine(6, 7, 4, True), rng)


@pytest.mark.parametrize("n", [6, 8])
def test_reversi_vs_naive(n):
    g = O.make_game("reversi8" if n == 8 else "reversi6")
    rng = np.random.default_rng(n)
    lens = [_playout(g, NG.NaiveReversi(n), rng) for _ in range(12)]
    assert max(lens) >= n * n - 6


def test_reversi6_perft_vs_naive():
    g = O.make_game("reversi6")

    def perft(nv, d):
        if d == 0:
            return 1
        if nv.is_over()[0]:
            return 0
        tot = 0
        for a in range(nv.A):
            if nv.can_play(a):
                c = NG.NaiveReversi(6)
                c.board = nv.board.copy()
                c.player = nv.player
                c.play(a)
                tot += perft(c, d - 1)
        return tot

    p = O.pos_init(g)
    assert [O.perft(g, p, d)[0] for d in range(1, 5)] == [perft(NG.NaiveReversi(6), d) for d in range(1, 5)]


@pytest.mark.parametrize("n", [3, 5, 9, 11])
def test_hex_vs_naive(n):
    """The reference's shift/AND/OR automaton (Hex.jl:54-67) is a Y-reduction; it must agree with plain
    flood-fill connectivity: the first player spans one axis, the second the other."""
    g = O.make_game("hex", n)
    rng = np.random.default_rng(n)
    # the first player's border stones sit in board column 1 (Hex.jl:27, startx[i,1]) => he spans the columns
    axis = 1
    for _ in range(20 if n <= 9 else 8):
        ply = _playout(g, NG.NaiveHex(n, axis), rng, check_bits=False)
        assert ply <= n * n                       # no draws in Hex: ends on or before a full board


def test_hex_full_board_has_exactly_one_winner():
    n = 7
    g = O.make_game("hex", n)
    rng = np.random.default_rng(77)
    for _ in range(30):
        order = rng.permutation(n * n)
        p = O.pos_init(g)
        over = False
        for a in order:
            p = O.play(g, p, int(a))
            f, r = O.is_over(g, p)
            if f:
                over = True
                assert r == -p.player
                break
        assert over


def test_bitboard_shifts_vs_array_model():
    rng = np.random.default_rng(5)
    for (kind, n, k) in (("gobang", 9, 5), ("gobang", 13, 5), ("connect4", 0, 0), ("hex", 9, 0), ("reversi8", 0, 0)):
        g = O.make_game(kind, n, k)
        R, Cc = g.d1, g.d2
        for _ in range(20):
            arr = rng.integers(0, 2, size=(R, Cc)).astype(np.uint8)
            bb = O.BB()
            flat = arr.T.reshape(-1)
            for i, v in enumerate(flat):
                if v:
                    bb.c[i >> 6] |= 1 << (i & 63)
            exp = {
                0: np.pad(arr, ((0, 0), (1, 0)))[:, :Cc],      # right: column c -> c+1
                1: np.pad(arr, ((0, 0), (0, 1)))[:, 1:],       # left
                2: np.pad(arr, ((1, 0), (0, 0)))[:R, :],       # down: row r -> r+1
                3: np.pad(arr, ((0, 1), (0, 0)))[1:, :],       # up
            }
            for op, e in exp.items():
                out = O.BB()
                O.lib().agzo_bb_shift(g, bb, op, out)
                assert O.bb_bits(out, g.len) == e.T.reshape(-1).astype(int).tolist(), (kind, op)


def test_julia_memory_image_roundtrip():
    import ctypes as C
    for kind, n, k, size in (("gobang", 9, 5, 104), ("hex", 9, 0, 104), ("reversi8", 0, 0, 152)):
        g = O.make_game(kind, n, k)
        p = O.pos_init(g)
        p = O.play(g, p, [a for a in range(g.A) if O.can_play(g, p, a)][0])
        img = O.pos_image(g, p)
        assert img.size == size
        meta = np.frombuffer(img[24:48].tobytes(), np.int64)
        assert meta.tolist() == [g.len, g.d1, g.d2]
        q = O.Pos()
        O.lib().agzo_pos_from_image(C.byref(g), img.ctypes.data_as(C.c_void_p), C.byref(q))
        assert bytes(q) == bytes(p)
