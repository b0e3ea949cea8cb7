// misere34.hpp — a SIXTH game plugged into libagz from outside alphagpu_amd/csrc (INTEGRATION.md "Adding a game"; the worked example the
// plugin test builds: tests/test_plugin_game.py).  The same surface the reference's game files export (Gobang.jl:2,8-11,16-70:
// Position with bplayer / bopponent / player, canPlay, play, isOver, and the constants maxActions, VectorizedState, FeatureSize,
// maxLengthGame), written once for host and device.
//
// The game: 4 x 4 board, the players place stones in turn; whoever completes THREE of his stones in a row or a column (no diagonals)
// LOSES; a full board without such a line is a draw.  (Deliberately not a Gobang variant: the winner's sign is the other way round and
// the diagonals do not count, so a build that dispatched to F_LINE by mistake could not pass the test.)
#pragma once

namespace agz {

template <int NC> struct Game<F_EXTRA, NC> {
    static AGZ_HD bool canPlay(const GamePar&, const WPos<NC>& s, int a) { return !bb_get(s.p, a) && !bb_get(s.o, a); }
    static AGZ_HD WPos<NC> play(const GamePar&, const WPos<NC>& s, int a) {
        WPos<NC> r; r.p = s.o; r.o = bb_set(s.p, a); r.lg = bb_zero<NC>(); r.player = -s.player; r.aux = s.aux + 1;
        return r;
    }
    // s.o = the stones of the side that has just moved (as in Gobang.jl:37); r = the winner in absolute colours
    static AGZ_HD bool isOver(const GamePar& P, const WPos<NC>& s, int& r) {
        BB<NC> b = s.o;
        for (int j = 1; j < P.nvict; ++j) b = bb_and(b, bb_right(P, b));
        bool line = bb_any(b);
        b = s.o;
        for (int j = 1; j < P.nvict; ++j) b = bb_and(b, bb_down(P, b));
        line |= bb_any(b);
        r = line ? s.player : 0;                                   // the side that completed the line (-s.player) has lost
        return line || (bb_count(s.p) + bb_count(s.o) == P.len);
    }
};

inline int extra_game_par(int n, int nvict, GamePar& P) {
    if (n != 4 || nvict != 3) return -1;
    P.n = 4; P.nvict = 3; P.d1 = 4; P.d2 = 4; P.len = 16;
    P.A = P.VS = P.FS = P.ML = 16; P.max_plies = 16; P.start_player = 1; P.start_aux = 0;
    return 0;
}

}  // namespace agz

#define AGZ_EXTRA_COMBOS(X) X(F_EXTRA, 1, 1)      // ply kernels: one 64-action row, one 64-bit board chunk
#define AGZ_EXTRA_SHAPES(X) X(F_EXTRA, 1, 4)      // search kernels: one chunk, 4 actions per lane (8 lanes x 4 >= 16 actions)
