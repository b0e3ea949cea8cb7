// agz_games.hpp — game plugins (Position / canPlay / play / isOver) for host and gfx950 device code.
//
// Mirrors the plugin surface of the reference (Gobang.jl:16-70, 4IARow.jl:16-81, Hex.jl:16-67,
// Reversi8x8.jl:73-121, Reversi6x6.jl:73-121) over the 192-bit board of Bitboard.jl:5-205, with the same
// bit numbering: cell [i1,i2] (1-based) = bit d1*(i2-1)+(i1-1).  Board geometry is a runtime GamePar so one
// instantiation per (game kind, 64-bit chunks) serves every board size.  All game state is wave-uniform:
// on the device these functions run on scalar (SGPR) values.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define AGZ_HD __host__ __device__ __forceinline__

namespace agz {

enum { K_GOBANG = 0, K_CONNECT4 = 1, K_HEX = 2, K_REVERSI8 = 3, K_REVERSI6 = 4, K_EXTRA = 5 };
// code families
enum { F_LINE = 0, F_C4 = 1, F_HEX = 2, F_REV = 3, F_EXTRA = 4 };
// K_EXTRA / F_EXTRA: a game plugin compiled in from OUTSIDE this file — build with -DAGZ_EXTRA_GAME_HPP='"/path/to/mygame.hpp"'
// (INTEGRATION.md "Adding a game"; tests/plugin/misere34.hpp is the worked example).  The header is included below, after the
// bitboard operations and before make_game_par; it defines, in namespace agz:
//     template <int NC> struct Game<F_EXTRA, NC> { static AGZ_HD bool canPlay(const GamePar&, const WPos<NC>&, int a);
//                                                   static AGZ_HD WPos<NC> play(const GamePar&, const WPos<NC>&, int a);
//                                                   static AGZ_HD bool isOver(const GamePar&, const WPos<NC>&, int& r); };
//     inline int extra_game_par(int n, int nvict, GamePar& P);      // geometry, A / VS / FS / ML / max_plies, the start position; 0 = ok
//     #define AGZ_EXTRA_COMBOS(X)  X(F_EXTRA, NR, NC) ...           // (64-action rows, 64-bit board chunks) of the ply kernels
//     #define AGZ_EXTRA_SHAPES(X)  X(F_EXTRA, NC, KPL) ...          // (chunks, actions per lane: 4 | 8 | 12 | 16 | 24 with 8 KPL >= A) of the search kernels
// — the same three functions over the same Position fields (bplayer, bopponent, player: Gobang.jl:16-21) the reference's plugins export.

struct GamePar {
    int32_t kind, fam, n, nvict, d1, d2, len, A, VS, FS, ML, max_plies, NR, NC, pass_action, rev8;
    uint64_t lenmask[3];     // Bitboard.jl:33-41 _msk
    uint64_t keep_down[3];   // ~(cells with i1==1)   (Bitboard.jl:149-158)
    uint64_t keep_up[3];     // ~(cells with i1==d1)  (Bitboard.jl:165-174)
    uint64_t hex_row1[3];    // cells [1,k], k=4..N+1 (Hex.jl:60-62, j=1)
    uint64_t start_p[3], start_o[3], start_lg[3];
    int32_t start_player, start_aux;
};

// compact position record (80 B): what the engine stores per tree node
struct Pos {
    uint64_t p[3], o[3], lg[3];
    int8_t player, aux, pad[6];
};

template <int NC> struct BB { uint64_t c[NC]; };

template <int NC> AGZ_HD BB<NC> bb_zero() { BB<NC> r; for (int i = 0; i < NC; ++i) r.c[i] = 0; return r; }
template <int NC> AGZ_HD BB<NC> bb_and(BB<NC> a, BB<NC> b) { for (int i = 0; i < NC; ++i) a.c[i] &= b.c[i]; return a; }
template <int NC> AGZ_HD BB<NC> bb_or(BB<NC> a, BB<NC> b) { for (int i = 0; i < NC; ++i) a.c[i] |= b.c[i]; return a; }
template <int NC> AGZ_HD BB<NC> bb_xor(BB<NC> a, BB<NC> b) { for (int i = 0; i < NC; ++i) a.c[i] ^= b.c[i]; return a; }
template <int NC> AGZ_HD BB<NC> bb_not(const GamePar& P, BB<NC> a) { for (int i = 0; i < NC; ++i) a.c[i] = ~a.c[i] & P.lenmask[i]; return a; }
template <int NC> AGZ_HD bool bb_any(BB<NC> a) { uint64_t x = 0; for (int i = 0; i < NC; ++i) x |= a.c[i]; return x != 0; }
template <int NC> AGZ_HD int bb_count(BB<NC> a) {
    int n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < NC; ++i) n += __popcll(a.c[i]);
#else
    for (int i = 0; i < NC; ++i) n += __builtin_popcountll(a.c[i]);
#endif
    return n;
}
template <int NC> AGZ_HD bool bb_get(BB<NC> a, int bit) {
    uint64_t w = a.c[0];
    if (NC > 1 && bit >= 64) w = a.c[1];
    if (NC > 2 && bit >= 128) w = a.c[2];
    return (w >> (bit & 63)) & 1;
}
template <int NC> AGZ_HD BB<NC> bb_set(BB<NC> a, int bit) {
    uint64_t m = (uint64_t)1 << (bit & 63);
    int ch = bit >> 6;
    for (int i = 0; i < NC; ++i) a.c[i] |= (i == ch) ? m : 0;
    return a;
}
// Bitboard.jl:85-107 (<<) for 1 <= n <= 63, masked to len
template <int NC> AGZ_HD BB<NC> bb_shl(const GamePar& P, BB<NC> a, int n) {
    BB<NC> r;
    for (int i = NC - 1; i > 0; --i) r.c[i] = ((a.c[i] << n) | (a.c[i - 1] >> (64 - n))) & P.lenmask[i];
    r.c[0] = (a.c[0] << n) & P.lenmask[0];
    return r;
}
// Bitboard.jl:110-134 (>>>)
template <int NC> AGZ_HD BB<NC> bb_shr(const GamePar& P, BB<NC> a, int n) {
    BB<NC> r;
    for (int i = 0; i < NC - 1; ++i) r.c[i] = ((a.c[i] >> n) | (a.c[i + 1] << (64 - n))) & P.lenmask[i];
    r.c[NC - 1] = (a.c[NC - 1] >> n) & P.lenmask[NC - 1];
    return r;
}
template <int NC> AGZ_HD BB<NC> bb_right(const GamePar& P, BB<NC> a) { return bb_shl(P, a, P.d1); }   // :136
template <int NC> AGZ_HD BB<NC> bb_left(const GamePar& P, BB<NC> a) { return bb_shr(P, a, P.d1); }    // :142
template <int NC> AGZ_HD BB<NC> bb_down(const GamePar& P, BB<NC> a) {                                 // :146-160
    BB<NC> r = bb_shl(P, a, 1);
    for (int i = 0; i < NC; ++i) r.c[i] &= P.keep_down[i];
    return r;
}
template <int NC> AGZ_HD BB<NC> bb_up(const GamePar& P, BB<NC> a) {                                   // :162-176
    BB<NC> r = bb_shr(P, a, 1);
    for (int i = 0; i < NC; ++i) r.c[i] &= P.keep_up[i];
    return r;
}

// wave-uniform position in working form
template <int NC> struct WPos {
    BB<NC> p, o, lg;
    int player, aux;
};
template <int NC> AGZ_HD WPos<NC> unpack(const Pos& s) {
    WPos<NC> w;
    for (int i = 0; i < NC; ++i) { w.p.c[i] = s.p[i]; w.o.c[i] = s.o[i]; w.lg.c[i] = s.lg[i]; }
    w.player = s.player; w.aux = s.aux;
    return w;
}
template <int NC> AGZ_HD Pos pack(const WPos<NC>& w) {
    Pos s;
    for (int i = 0; i < 3; ++i) { s.p[i] = i < NC ? w.p.c[i] : 0; s.o[i] = i < NC ? w.o.c[i] : 0; s.lg[i] = i < NC ? w.lg.c[i] : 0; }
    s.player = (int8_t)w.player; s.aux = (int8_t)w.aux;
    for (int i = 0; i < 6; ++i) s.pad[i] = 0;
    return s;
}

// ------------------------------------------------------------------------------------------------
template <int FAM, int NC> struct Game;

// ---- k-in-a-row family: Gobang.jl / 4IARow.jl share isOver -------------------------------------
template <int NC> AGZ_HD bool line_is_over(const GamePar& P, const WPos<NC>& s, int& r) {          // Gobang.jl:36-70
    BB<NC> b = s.o;
    for (int j = 1; j < P.nvict; ++j) b = bb_and(b, bb_right(P, b));
    bool win = bb_any(b);
    b = s.o;
    for (int j = 1; j < P.nvict; ++j) b = bb_and(b, bb_down(P, b));
    win |= bb_any(b);
    b = s.o;
    for (int j = 1; j < P.nvict; ++j) b = bb_and(b, bb_down(P, bb_right(P, b)));
    win |= bb_any(b);
    b = s.o;
    for (int j = 1; j < P.nvict; ++j) b = bb_and(b, bb_left(P, bb_down(P, b)));
    win |= bb_any(b);
    r = win ? -s.player : 0;
    return win || (bb_count(s.p) + bb_count(s.o) == P.len);
}

template <int NC> struct Game<F_LINE, NC> {
    static AGZ_HD int cell(const GamePar&, const WPos<NC>&, int a) { return a; }
    static AGZ_HD bool canPlay(const GamePar&, const WPos<NC>& s, int a) {                         // Gobang.jl:25-27
        return !bb_get(s.p, a) && !bb_get(s.o, a);
    }
    static AGZ_HD WPos<NC> play(const GamePar&, const WPos<NC>& s, int a) {                         // Gobang.jl:30-33
        WPos<NC> r; r.p = s.o; r.o = bb_set(s.p, a); r.lg = bb_zero<NC>(); r.player = -s.player; r.aux = s.aux + 1;
        return r;
    }
    static AGZ_HD bool isOver(const GamePar& P, const WPos<NC>& s, int& r) { return line_is_over(P, s, r); }
};

template <int NC> struct Game<F_C4, NC> {
    static AGZ_HD bool canPlay(const GamePar& P, const WPos<NC>& s, int a) {                        // 4IARow.jl:25-27
        int b = P.d1 * a;                                                                           // cell [1, a+1]
        return !bb_get(s.p, b) && !bb_get(s.o, b);
    }
    static AGZ_HD WPos<NC> play(const GamePar& P, const WPos<NC>& s, int a) {                       // 4IARow.jl:30-44
        BB<NC> empty = bb_not(P, bb_or(s.p, s.o));
        int free_ = 1;
        bool go = true;
        for (int i = 1; i <= P.d1; ++i) {
            bool e = bb_get(empty, P.d1 * a + (i - 1));
            if (go && e) free_ = i; else go = false;
        }
        WPos<NC> r; r.p = s.o; r.o = bb_set(s.p, P.d1 * a + free_ - 1); r.lg = bb_zero<NC>();
        r.player = -s.player; r.aux = s.aux + 1;
        return r;
    }
    static AGZ_HD bool isOver(const GamePar& P, const WPos<NC>& s, int& r) { return line_is_over(P, s, r); }
};

// ---- Hex.jl -----------------------------------------------------------------------------------------
template <int NC> struct Game<F_HEX, NC> {
    static AGZ_HD int cell(const GamePar& P, int a) {                                               // Hex.jl:37-41
        int x = a / P.n, y = a - P.n * x + 1;
        return (P.n + 1) * (x + 1) + y;                                                             // 0-based bit
    }
    static AGZ_HD bool canPlay(const GamePar& P, const WPos<NC>& s, int a) {
        int b = cell(P, a);
        return !bb_get(s.p, b) && !bb_get(s.o, b);
    }
    static AGZ_HD WPos<NC> play(const GamePar& P, const WPos<NC>& s, int a) {                       // Hex.jl:45-51
        WPos<NC> r; r.p = s.o; r.o = bb_set(s.p, cell(P, a)); r.lg = bb_zero<NC>(); r.player = -s.player; r.aux = s.aux - 1;
        return r;
    }
    static AGZ_HD bool isOver(const GamePar& P, const WPos<NC>& s, int& r) {                        // Hex.jl:54-67
        BB<NC> a = s.o, row1;
        for (int i = 0; i < NC; ++i) row1.c[i] = P.hex_row1[i];
        for (int j = 1; j <= 2 * P.n - 2; ++j) {
            BB<NC> b = bb_up(P, a);
            BB<NC> c = bb_right(P, b);
            a = bb_down(P, bb_or(bb_and(a, bb_or(b, c)), bb_and(b, c)));
            if (s.player == 1) a = bb_or(a, row1);
            // next j drops cell [1, 3+j]
            int bit = P.d1 * (3 + j - 1);
            uint64_t m = ~((uint64_t)1 << (bit & 63));
            int ch = bit >> 6;
            for (int i = 0; i < NC; ++i) row1.c[i] &= (i == ch) ? m : ~(uint64_t)0;
        }
        r = -s.player;
        return bb_get(a, P.len - 1);                                                                // a[N+1,N+1]
    }
};

// ---- Reversi8x8.jl / Reversi6x6.jl ------------------------------------------------------------------
template <int NC, int D> AGZ_HD BB<NC> rev_dir(const GamePar& P, BB<NC> x) {
    // direction order of legalplay(): up, down, left, right, diaghg, diagbg, diaghd, diagbd (Reversi8x8.jl:16-39)
    if (D == 0) return bb_up(P, x);
    if (D == 1) return bb_down(P, x);
    if (D == 2) return bb_left(P, x);
    if (D == 3) return bb_right(P, x);
    if (D == 4) return bb_up(P, bb_left(P, x));
    if (D == 5) return bb_down(P, bb_left(P, x));
    if (D == 6) return bb_up(P, bb_right(P, x));
    return bb_down(P, bb_right(P, x));
}
template <int NC, int D> AGZ_HD BB<NC> rev_legal_dir(const GamePar& P, BB<NC> tj, BB<NC> ta, BB<NC> vide) { // :25-34
    BB<NC> moves = bb_zero<NC>();
    BB<NC> cand = bb_and(rev_dir<NC, D>(P, tj), ta);
    while (bb_any(cand)) {
        BB<NC> nx = rev_dir<NC, D>(P, cand);
        moves = bb_or(moves, bb_and(vide, nx));
        cand = bb_and(ta, nx);
    }
    return moves;
}
template <int NC> AGZ_HD BB<NC> rev_legal(const GamePar& P, BB<NC> tj, BB<NC> ta) {                  // :36-39
    BB<NC> vide = bb_and(bb_not(P, tj), bb_not(P, ta));
    BB<NC> m = rev_legal_dir<NC, 0>(P, tj, ta, vide);
    m = bb_or(m, rev_legal_dir<NC, 1>(P, tj, ta, vide));
    m = bb_or(m, rev_legal_dir<NC, 2>(P, tj, ta, vide));
    m = bb_or(m, rev_legal_dir<NC, 3>(P, tj, ta, vide));
    m = bb_or(m, rev_legal_dir<NC, 4>(P, tj, ta, vide));
    m = bb_or(m, rev_legal_dir<NC, 5>(P, tj, ta, vide));
    m = bb_or(m, rev_legal_dir<NC, 6>(P, tj, ta, vide));
    m = bb_or(m, rev_legal_dir<NC, 7>(P, tj, ta, vide));
    return m;
}
template <int NC, int D> AGZ_HD BB<NC> rev_flippar(const GamePar& P, BB<NC> tj, BB<NC> ta, BB<NC> play) { // :43-55
    BB<NC> cand = bb_and(rev_dir<NC, D>(P, play), ta);
    BB<NC> toflip = cand;
    while (bb_any(cand)) {
        cand = bb_and(ta, rev_dir<NC, D>(P, cand));
        toflip = bb_or(toflip, cand);
    }
    return bb_any(bb_and(rev_dir<NC, D>(P, toflip), tj)) ? toflip : bb_zero<NC>();
}
template <int NC> struct Game<F_REV, NC> {
    static AGZ_HD bool canPlay(const GamePar& P, const WPos<NC>& s, int a) {                        // Reversi8x8.jl:84-90
        if (a == P.pass_action) return !bb_any(s.lg);
        return bb_get(s.lg, a);
    }
    static AGZ_HD WPos<NC> play(const GamePar& P, const WPos<NC>& s, int a) {                       // :93-106
        WPos<NC> r;
        BB<NC> tj = s.p, ta = s.o;
        if (a != P.pass_action) {
            BB<NC> t = bb_set(bb_zero<NC>(), a);
            BB<NC> h = rev_flippar<NC, 0>(P, tj, ta, t);
            h = bb_or(h, rev_flippar<NC, 1>(P, tj, ta, t));
            h = bb_or(h, rev_flippar<NC, 2>(P, tj, ta, t));
            h = bb_or(h, rev_flippar<NC, 3>(P, tj, ta, t));
            h = bb_or(h, rev_flippar<NC, 4>(P, tj, ta, t));
            h = bb_or(h, rev_flippar<NC, 5>(P, tj, ta, t));
            h = bb_or(h, rev_flippar<NC, 6>(P, tj, ta, t));
            h = bb_or(h, rev_flippar<NC, 7>(P, tj, ta, t));
            tj = bb_xor(tj, h); ta = bb_xor(ta, h);
            tj = bb_set(tj, a);
        }
        r.p = ta; r.o = tj; r.lg = rev_legal(P, ta, tj); r.player = -s.player; r.aux = 0;
        return r;
    }
    static AGZ_HD bool isOver(const GamePar& P, const WPos<NC>& s, int& r) {                        // 8x8 :109-121, 6x6 :109-121
        bool over = !bb_any(s.lg) && !bb_any(rev_legal(P, s.o, s.p));
        int test = bb_count(s.p) - bb_count(s.o);
        int sgn = (test > 0) - (test < 0);
        r = (P.rev8 || over) ? sgn * s.player : 0;
        return over;
    }
};

}  // namespace agz
#ifdef AGZ_EXTRA_GAME_HPP
#include AGZ_EXTRA_GAME_HPP
#endif
#ifndef AGZ_EXTRA_COMBOS
#define AGZ_EXTRA_COMBOS(X)
#endif
#ifndef AGZ_EXTRA_SHAPES
#define AGZ_EXTRA_SHAPES(X)
#endif
namespace agz {

// ------------------------------------------------------------------------------------------------
// host-side construction of GamePar and start positions
inline int make_game_par(int kind, int n, int nvict, GamePar& P) {
    P = GamePar();
    P.kind = kind; P.pass_action = -1;
    switch (kind) {
    case K_GOBANG:
        if (n < 1 || n > 13 || nvict < 1) return -1;
        P.fam = F_LINE; P.n = n; P.nvict = nvict; P.d1 = n; P.d2 = n; P.len = n * n;
        P.A = P.VS = P.FS = P.ML = n * n; P.max_plies = n * n; P.start_player = 1; P.start_aux = 0; break;
    case K_CONNECT4:
        P.fam = F_C4; P.n = 6; P.nvict = 4; P.d1 = 6; P.d2 = 7; P.len = 42;
        P.A = 7; P.VS = P.FS = P.ML = 42; P.max_plies = 42; P.start_player = 1; P.start_aux = 1; break;
    case K_HEX:
        if (n < 2 || n > 12) return -1;
        P.fam = F_HEX; P.n = n; P.d1 = n + 1; P.d2 = n + 1; P.len = (n + 1) * (n + 1);
        P.VS = P.FS = P.len; P.A = P.ML = n * n; P.max_plies = n * n; P.start_player = 1; P.start_aux = n * n; break;
    case K_REVERSI8:
        P.fam = F_REV; P.n = 8; P.d1 = 8; P.d2 = 8; P.len = 64; P.VS = P.FS = 64; P.A = 65; P.ML = 70;
        P.max_plies = 128; P.pass_action = 64; P.rev8 = 1; P.start_player = 1; break;
    case K_REVERSI6:
        P.fam = F_REV; P.n = 6; P.d1 = 6; P.d2 = 6; P.len = 36; P.VS = P.FS = 36; P.A = 37; P.ML = 50;
        P.max_plies = 72; P.pass_action = 36; P.rev8 = 0; P.start_player = 1; break;
#ifdef AGZ_EXTRA_GAME_HPP
    case K_EXTRA:
        if (extra_game_par(n, nvict, P) != 0) return -1;
        P.kind = K_EXTRA; P.fam = F_EXTRA; break;
#endif
    default: return -1;
    }
    P.NR = (P.A + 63) / 64; P.NC = (P.len + 63) / 64;
    for (int i = 0; i < 3; ++i) { P.lenmask[i] = 0; P.keep_down[i] = ~(uint64_t)0; P.keep_up[i] = ~(uint64_t)0; }
    for (int b = 0; b < P.len; ++b) P.lenmask[b >> 6] |= (uint64_t)1 << (b & 63);
    for (int b = 0; b < P.len; b += P.d1) P.keep_down[b >> 6] &= ~((uint64_t)1 << (b & 63));
    for (int b = P.d1 - 1; b < P.len; b += P.d1) P.keep_up[b >> 6] &= ~((uint64_t)1 << (b & 63));
    auto setb = [](uint64_t* w, int b) { w[b >> 6] |= (uint64_t)1 << (b & 63); };
    auto idx = [&](int i1, int i2) { return P.d1 * (i2 - 1) + (i1 - 1); };
    if (kind == K_HEX) {
        for (int k = 4; k <= n + 1; ++k) setb(P.hex_row1, idx(1, k));
        for (int i = 3; i <= n + 1; ++i) { setb(P.start_p, idx(i, 1)); setb(P.start_o, idx(1, i)); }   // Hex.jl:22-35
    } else if (kind == K_REVERSI8) {                                                                  // Reversi8x8.jl:10-14
        setb(P.start_p, idx(4, 5)); setb(P.start_p, idx(5, 4)); setb(P.start_o, idx(5, 5)); setb(P.start_o, idx(4, 4));
    } else if (kind == K_REVERSI6) {                                                                  // Reversi6x6.jl:11-14
        setb(P.start_p, idx(4, 3)); setb(P.start_p, idx(3, 4)); setb(P.start_o, idx(3, 3)); setb(P.start_o, idx(4, 4));
    }
    if (P.fam == F_REV) {
        BB<1> tj, ta; tj.c[0] = P.start_p[0]; ta.c[0] = P.start_o[0];
        P.start_lg[0] = rev_legal<1>(P, tj, ta).c[0];
    }
    return 0;
}
inline Pos start_pos(const GamePar& P) {
    Pos s = Pos();
    for (int i = 0; i < 3; ++i) { s.p[i] = P.start_p[i]; s.o[i] = P.start_o[i]; s.lg[i] = P.start_lg[i]; }
    s.player = (int8_t)P.start_player; s.aux = (int8_t)P.start_aux;
    return s;
}

}  // namespace agz
