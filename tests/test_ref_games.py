"""The C oracle's game code against the second restatement of Bitboard.jl and the five game files (tests/ref_games.py, written from
the Julia text alone): random play-outs of every game — boards chunk for chunk, side to move, legality of EVERY action, the cached
legal set of Reversi, isOver flag and result, at every ply — plus the shift / edge operations of Bitboard.jl on random boards of every
geometry.  CPU only.  (The reference itself cannot run here — no Julia: two independent readings agreeing is the pin available.)"""
import numpy as np
import pytest

import oracle_lib as O
import ref_games as RG

CASES = [("gobang", 3, 3), ("gobang", 9, 5), ("gobang", 13, 5), ("gobang", 11, 4), ("connect4", 0, 0), ("hex", 5, 0), ("hex", 9, 0), ("hex", 12, 0),
         ("reversi8", 0, 0), ("reversi6", 0, 0)]


def same_boards(op, rp, reversi):
    ok = tuple(op.bplayer.c) == rp.bplayer.chunks and tuple(op.bopponent.c) == rp.bopponent.chunks and op.player == rp.player
    if reversi:
        ok = ok and tuple(op.legalplay.c) == rp.legalplay.chunks
    return ok


@pytest.mark.parametrize("kind,n,k", CASES)
def test_playouts_agree_with_the_oracle(kind, n, k):
    og, rg = O.make_game(kind, n, k), RG.make(kind, n, k)
    assert (og.A, og.VS, og.FS, og.ML) == (rg.maxActions, rg.VectorizedState, rg.FeatureSize, rg.maxLengthGame)
    rng = np.random.default_rng(hash((kind, n)) & 0xFFFF)
    results = {1: 0, 0: 0, -1: 0}
    games = 40 if og.A <= 81 else 16
    for _game in range(games):
        op, rp = O.pos_init(og), rg.start()
        assert same_boards(op, rp, kind.startswith("reversi"))
        for _ply in range(400):
            legal_o = [a for a in range(og.A) if O.can_play(og, op, a)]
            legal_r = [a for a in range(og.A) if rg.canPlay(rp, a + 1)]                # the reference's actions are 1-based
            assert legal_o == legal_r and legal_o, (kind, _ply)
            a = legal_o[int(rng.integers(len(legal_o)))]
            op, rp = O.play(og, op, a), rg.play(rp, a + 1)
            assert same_boards(op, rp, kind.startswith("reversi")), (kind, _ply, a)
            fo, ro = O.is_over(og, op)
            fr, rr = rg.isOver(rp)
            assert fo == bool(fr), (kind, _ply)
            if fo or kind == "reversi8":                                               # (Reversi 8x8 returns the sign product even while the game goes on, :121)
                assert ro == rr, (kind, _ply, ro, rr)
            if fo:
                results[ro] += 1
                break
        else:
            raise AssertionError("play-out does not end")
    assert sum(results.values()) == games


@pytest.mark.parametrize("d1,d2", [(3, 3), (6, 7), (8, 8), (6, 6), (9, 9), (10, 10), (13, 13), (12, 12), (11, 11)])
def test_bitboard_operations_agree_with_a_bool_array_model(d1, d2):
    """Bitboard.jl's <<, >>>, right, left, down, up, ~ as transliterated, against the obvious array model: cell [i1, i2] (1-based) is bit
    d1 (i2 - 1) + i1; right / left move a stone to the next / previous column, down / up to the next / previous row of its column and off
    the board at the edge.  Pins the transliteration itself (the C oracle's operations are pinned against it by the play-outs)."""
    rng = np.random.default_rng(d1 * 100 + d2)
    for _ in range(25):
        cells = rng.random((d1, d2)) < 0.4
        bb = RG.bitboard.new(d1, d2)
        for i1 in range(1, d1 + 1):
            for i2 in range(1, d2 + 1):
                if cells[i1 - 1, i2 - 1]:
                    bb = RG.setindex(bb, True, i1, i2)

        def arr(b):
            return np.array([[RG.getindex(b, i1, i2) for i2 in range(1, d2 + 1)] for i1 in range(1, d1 + 1)])
        assert np.array_equal(arr(bb), cells) and RG.num_bit(bb) == int(cells.sum())
        z = np.zeros_like(cells)
        r = z.copy(); r[:, 1:] = cells[:, :-1]
        l = z.copy(); l[:, :-1] = cells[:, 1:]                                          # noqa: E741
        dn = z.copy(); dn[1:, :] = cells[:-1, :]
        u = z.copy(); u[:-1, :] = cells[1:, :]
        assert np.array_equal(arr(RG.right(bb)), r) and np.array_equal(arr(RG.left(bb)), l)
        assert np.array_equal(arr(RG.down(bb)), dn) and np.array_equal(arr(RG.up(bb)), u)
        assert np.array_equal(arr(RG.bnot(bb)), ~cells) and RG.num_bit(RG.bnot(bb)) == d1 * d2 - int(cells.sum())
