"""A SECOND restatement of the reference's game plugins and of Bitboard.jl — test infrastructure, written from the Julia text ALONE
(/root/reference/Bitboard.jl, Gobang.jl, 4IARow.jl, Hex.jl, Reversi8x8.jl, Reversi6x6.jl), without looking at the C oracle's
restatement (oracle/agz_oracle.c) or at the device code (alphagpu_amd/csrc/agz_games.hpp): same names, same 1-based indices, same
statement order, UInt64 chunks as Python ints masked to 64 bits (Julia: a shift by >= 64 gives 0).  Nothing here calls the oracle.

tests/test_ref_games.py requires that it agrees with the C oracle on random play-outs of every game (boards, side to move, legality of
every action, isOver flag and result at every ply) — a transcription error in either restatement shows up as a difference.  The
reference itself cannot run here (Julia + CUDA.jl), so this is the pin the oracle's game code can get; the external known answers
(TicTacToe 255 168 games, Othello and Connect-4 perft) pin both."""
from collections import namedtuple

M64 = (1 << 64) - 1


def _shl(x, n):          # UInt64 << n
    return (x << n) & M64 if 0 <= n < 64 else 0


def _shr(x, n):          # UInt64 >>> n
    return x >> n if 0 <= n < 64 else 0


# ---------------------------------------------------------------------------------------------------- Bitboard.jl
class bitboard:
    """struct bitboard{N}: chunks::NTuple{3,UInt64}, len::Int, dims::NTuple{N,Int} (Bitboard.jl:5-9)"""
    __slots__ = ("chunks", "len", "dims")

    def __init__(self, chunks, ln, dims):
        self.chunks, self.len, self.dims = tuple(chunks), ln, tuple(dims)

    @staticmethod
    def new(*dims):                                               # bitboard{N}(dims...) :14-25
        n = 1
        for d in dims:
            assert d >= 0
            n *= d
        assert n <= 192
        return bitboard((0, 0, 0), n, dims)

    def __eq__(self, o):
        return self.chunks == o.chunks and self.len == o.len and self.dims == o.dims


_msk64 = M64


def _div64(l):           # noqa: E741
    return l >> 6


def _mod64(l):           # noqa: E741
    return l & 63


def _msk_end(l):         # _msk64 >>> _mod64(-l)   :31       # noqa: E741
    return _shr(_msk64, _mod64(-l))


def _msk(bb):            # :33-41
    if bb.len <= 64:
        return (_msk_end(bb.len), 0, 0)
    elif bb.len <= 128:
        return (_msk64, _msk_end(bb.len), 0)
    else:
        return (_msk64, _msk64, _msk_end(bb.len))


def get_chunks_id(i):    # :45
    return _div64(i - 1) + 1, _mod64(i - 1)


def getindex(bb, i, i2=None):                                     # :47-57
    if i2 is not None:
        i = bb.dims[0] * (i2 - 1) + i
    c1, c2 = get_chunks_id(i)
    c = bb.chunks[c1 - 1]
    return (c & _shl(1, c2)) != 0


def setindex(bb, x, i, i2=None):                                  # :60-80
    if i2 is not None:
        i = bb.dims[0] * (i2 - 1) + i
    c1, c2 = get_chunks_id(i)
    u = _shl(1, c2)
    c = bb.chunks[c1 - 1]
    newc = (c | u) if x else (c & ~u & M64)
    if c1 == 1:
        return bitboard((newc, bb.chunks[1], bb.chunks[2]), bb.len, bb.dims)
    elif c1 == 2:
        return bitboard((bb.chunks[0], newc, bb.chunks[2]), bb.len, bb.dims)
    else:
        return bitboard((bb.chunks[0], bb.chunks[1], newc), bb.len, bb.dims)


def shl(bb, n):                                                   # <<  :85-107
    i1 = _div64(n)
    i2 = _mod64(n)
    x, y, z = bb.chunks
    if 1 <= i1 < 2:
        z = y
        y = x
        x = 0
    elif i1 >= 2:
        z = x
        x = 0
        y = x
    newx = _shl(x, n)
    headx = _shr(x, 64 - i2)
    heady = _shr(y, 64 - i2)
    newy = _shl(y, i2) | headx
    newz = _shl(z, i2) | heady
    mx, my, mz = _msk(bb)
    return bitboard((newx & mx, newy & my, newz & mz), bb.len, bb.dims)


def shr(bb, n):                                                   # >>>  :110-134
    i1 = _div64(n)
    i2 = _mod64(n)
    x, y, z = bb.chunks
    if 1 <= i1 < 2:
        y = z
        z = 0
        x = y
    elif i1 >= 2:
        x = z
        z = 0
        y = z
    newz = _shr(z, n)
    headz = _shl(z, 64 - i2)
    heady = _shl(y, 64 - i2)
    newy = _shr(y, i2) | headz
    newx = _shr(x, i2) | heady
    mx, my, mz = _msk(bb)
    return bitboard((newx & mx, newy & my, newz & mz), bb.len, bb.dims)


def right(bb):           # :136-139
    return shl(bb, bb.dims[0])


def left(bb):            # :142-145
    return shr(bb, bb.dims[0])


def _clear_cells(chunks, cells):
    x, y, z = chunks
    for i in cells:
        c1, c2 = get_chunks_id(i)
        if c1 == 1:
            x &= ~_shl(1, c2) & M64
        elif c1 == 2:
            y &= ~_shl(1, c2) & M64
        else:
            z &= ~_shl(1, c2) & M64
    return (x, y, z)


def down(bb):            # :147-161   for i in 1:size(bb)[1]:length(bb)
    dbb = shl(bb, 1)
    return bitboard(_clear_cells(dbb.chunks, range(1, bb.len + 1, bb.dims[0])), bb.len, bb.dims)


def up(bb):              # :163-177   for i in size(bb)[1]:size(bb)[1]:length(bb)
    dbb = shr(bb, 1)
    return bitboard(_clear_cells(dbb.chunks, range(bb.dims[0], bb.len + 1, bb.dims[0])), bb.len, bb.dims)


def num_bit(bb):         # :178-181
    return bin(bb.chunks[0]).count("1") + bin(bb.chunks[1]).count("1") + bin(bb.chunks[2]).count("1")


def bnot(bb):            # ~  :183-188
    x, y, z = bb.chunks
    mx, my, mz = _msk(bb)
    return bitboard(((~x) & mx, (~y) & my, (~z) & mz), bb.len, bb.dims)


def band(a, b):          # &
    return bitboard(tuple(p & q for p, q in zip(a.chunks, b.chunks)), a.len, a.dims)


def bor(a, b):           # |
    return bitboard(tuple(p | q for p, q in zip(a.chunks, b.chunks)), a.len, a.dims)


def bxor(a, b):          # ⊻
    return bitboard(tuple(p ^ q for p, q in zip(a.chunks, b.chunks)), a.len, a.dims)


def int8(x):             # Int8 arithmetic wraps
    x &= 0xFF
    return x - 256 if x >= 128 else x


# ---------------------------------------------------------------------------------------------------- the k-in-a-row test shared by
# Gobang.jl:36-70 and 4IARow.jl:47-81 (the two files carry the same text)
def _line_is_over(pos, Nvict, full):
    board = pos.bopponent
    for _j in range(1, Nvict):
        board = band(board, right(board))
    if num_bit(board) != 0:
        return True, int8(-pos.player)
    board = pos.bopponent
    for _j in range(1, Nvict):
        board = band(board, down(board))
    if num_bit(board) != 0:
        return True, int8(-pos.player)
    board = pos.bopponent
    for _j in range(1, Nvict):
        board = band(board, down(right(board)))
    if num_bit(board) != 0:
        return True, int8(-pos.player)
    board = pos.bopponent
    for _j in range(1, Nvict):
        board = band(board, left(down(board)))
    if num_bit(board) != 0:
        return True, int8(-pos.player)
    return num_bit(pos.bplayer) + num_bit(pos.bopponent) == full, 0


class GoBang:
    """Gobang.jl (module GoBang), N x N board, Nvict in a row"""
    Position = namedtuple("Position", "bplayer bopponent player round")

    def __init__(self, N, Nvict):
        self.N, self.Nvict, self.NN = N, Nvict, N * N
        self.VectorizedState = self.FeatureSize = self.maxActions = self.maxLengthGame = self.NN      # :8-11

    def start(self):                                              # Position() :23
        return self.Position(bitboard.new(self.N, self.N), bitboard.new(self.N, self.N), 1, 0)

    def canPlay(self, pos, col):                                  # :25-27
        return (not getindex(pos.bplayer, col)) and (not getindex(pos.bopponent, col))

    def play(self, pos, col):                                     # :30-33
        bplayer = setindex(pos.bplayer, True, col)
        return self.Position(pos.bopponent, bplayer, int8(-pos.player), int8(pos.round + 1))

    def isOver(self, pos):                                        # :36-70
        return _line_is_over(pos, self.Nvict, self.NN)


class FourIARow:
    """4IARow.jl (module FourIARow)"""
    Position = namedtuple("Position", "bplayer bopponent player round")
    Height, Width, Nvict = 6, 7, 4
    VectorizedState = FeatureSize = maxLengthGame = 42
    maxActions = 7

    def start(self):                                              # :23
        return self.Position(bitboard.new(self.Height, self.Width), bitboard.new(self.Height, self.Width), 1, 1)

    def canPlay(self, pos, col):                                  # :25-27
        return (not getindex(pos.bplayer, 1, col)) and (not getindex(pos.bopponent, 1, col))

    def play(self, pos, col):                                     # :30-44
        free = 1
        empty = bnot(bor(pos.bplayer, pos.bopponent))
        for i in range(1, self.Height + 1):
            if getindex(empty, i, col):
                free = i
            else:
                break
        c = self.Height * (col - 1) + free
        bplayer = setindex(pos.bplayer, True, c)
        return self.Position(pos.bopponent, bplayer, int8(-pos.player), int8(pos.round + 1))

    def isOver(self, pos):                                        # :47-81
        return _line_is_over(pos, self.Nvict, self.maxLengthGame)


class Hex:
    """Hex.jl (module Hex), N x N cells on an (N+1) x (N+1) bitboard"""
    Position = namedtuple("Position", "bplayer bopponent player lp")

    def __init__(self, N):
        self.N, self.NN = N, N * N
        self.VectorizedState = self.FeatureSize = (N + 1) * (N + 1)       # :7-8
        self.maxActions = self.maxLengthGame = self.NN
        empty = bitboard.new(N + 1, N + 1)
        startx, starto = empty, empty                             # init() :22-31
        for i in range(3, N + 2):
            startx = setindex(startx, True, i, 1)
            starto = setindex(starto, True, 1, i)
        self.startx, self.starto = startx, starto

    def start(self):                                              # :35
        return self.Position(self.startx, self.starto, 1, int8(self.NN))

    def _newcol(self, col):                                       # :38-40
        N = self.N
        x = (col - 1) // N
        y = col - N * x
        return (N + 1) * (x + 1) + y + 1

    def canPlay(self, pos, col):                                  # :37-42
        newcol = self._newcol(col)
        return (not getindex(pos.bplayer, newcol)) and (not getindex(pos.bopponent, newcol))

    def play(self, pos, col):                                     # :45-51
        bplayer = setindex(pos.bplayer, True, self._newcol(col))
        return self.Position(pos.bopponent, bplayer, int8(-pos.player), int8(pos.lp - 1))

    def isOver(self, pos):                                        # :54-67
        N = self.N
        a = pos.bopponent
        for j in range(1, 2 * N - 2 + 1):
            b = up(a)
            c = right(b)
            a = down(bor(band(a, bor(b, c)), band(b, c)))
            if pos.player == 1:
                for k in range(3 + j, N + 2):
                    a = setindex(a, True, 1, k)
        return getindex(a, N + 1, N + 1), int8(-pos.player)


class Reversi:
    """Reversi8x8.jl (module RevSix, 8 x 8) and Reversi6x6.jl (6 x 6): the same text up to the constants and the last lines of isOver"""
    Position = namedtuple("Position", "bplayer bopponent legalplay player")

    def __init__(self, size):
        assert size in (8, 6)
        self.size = size
        if size == 8:                                             # Reversi8x8.jl:5-14
            self.FeatureSize = self.VectorizedState = 64
            self.maxActions, self.maxLengthGame = 65, 70
            self.empty = bitboard.new(8, 8)
            start = setindex(self.empty, True, 4, 5)
            self.starto = setindex(start, True, 5, 4)
            start = setindex(self.empty, True, 5, 5)
            self.startp = setindex(start, True, 4, 4)
        else:                                                     # Reversi6x6.jl:6-14
            self.VectorizedState = self.FeatureSize = 36
            self.maxActions, self.maxLengthGame = 37, 50
            self.empty = bitboard.new(6, 6)
            start = setindex(self.empty, True, 4, 3)
            self.starto = setindex(start, True, 3, 4)
            start = setindex(self.empty, True, 3, 3)
            self.startp = setindex(start, True, 4, 4)
        self.lpstart = self.legalplay(self.starto, self.startp)   # :80
        self.passmove = self.maxActions

    # directions :17-23
    @staticmethod
    def diaghd(x):
        return up(right(x))

    @staticmethod
    def diaghg(x):
        return up(left(x))

    @staticmethod
    def diagbd(x):
        return down(right(x))

    @staticmethod
    def diagbg(x):
        return down(left(x))

    def legal_play(self, tabjoueur, tabadversaire, dir):          # :26-35      # noqa: A002
        tabvide = band(bnot(tabjoueur), bnot(tabadversaire))
        moves = self.empty
        candidats = band(dir(tabjoueur), tabadversaire)
        while num_bit(candidats) != 0:
            moves = bor(moves, band(tabvide, dir(candidats)))
            candidats = band(tabadversaire, dir(candidats))
        return moves

    def legalplay(self, tabjoueur, tabadversaire):                # :37-40
        r = self.empty
        for d in (up, down, left, right, self.diaghg, self.diagbg, self.diaghd, self.diagbd):
            r = bor(r, self.legal_play(tabjoueur, tabadversaire, d))
        return r

    def flippar(self, tabjoueur, tabadversaire, play, dir):       # :44-55      # noqa: A002
        candidats = band(dir(play), tabadversaire)
        toflip = candidats
        while num_bit(candidats) != 0:
            candidats = band(tabadversaire, dir(candidats))
            toflip = bor(toflip, candidats)
        if num_bit(band(dir(toflip), tabjoueur)) != 0:
            return toflip
        else:
            return self.empty

    def flip(self, tabjoueur, tabadversaire, play):               # :57-69
        test = setindex(self.empty, True, play)
        h = self.flippar(tabjoueur, tabadversaire, test, up)
        for d in (down, left, right, self.diaghd, self.diaghg, self.diagbd, self.diagbg):
            h = bor(h, self.flippar(tabjoueur, tabadversaire, test, d))
        return h

    def start(self):                                              # :82
        return self.Position(self.starto, self.startp, self.lpstart, 1)

    def canPlay(self, pos, c):                                    # :84-90
        if c == self.passmove:
            return num_bit(pos.legalplay) == 0
        else:
            return getindex(pos.legalplay, c)

    def play(self, pos, c):                                       # :93-106
        tabjoueur = pos.bplayer
        tabadversaire = pos.bopponent
        if c == self.passmove:
            moves = self.legalplay(tabadversaire, tabjoueur)
            return self.Position(pos.bopponent, pos.bplayer, moves, int8(-pos.player))
        h = self.flip(tabjoueur, tabadversaire, c)
        tabjoueur = bxor(tabjoueur, h)
        tabadversaire = bxor(tabadversaire, h)
        tabjoueur = setindex(tabjoueur, True, c)
        moves = self.legalplay(tabadversaire, tabjoueur)
        return self.Position(tabadversaire, tabjoueur, moves, int8(-pos.player))

    def isOver(self, pos):                                        # 8x8 :109-121 / 6x6 :109-121
        if self.size == 8:
            test = int8(num_bit(pos.bplayer) - num_bit(pos.bopponent))
            sgn = (test > 0) - (test < 0)
            return (num_bit(pos.legalplay) == 0 and num_bit(self.legalplay(pos.bopponent, pos.bplayer)) == 0), int8(sgn * pos.player)
        if num_bit(pos.legalplay) != 0 or num_bit(self.legalplay(pos.bopponent, pos.bplayer)) != 0:
            return False, 0
        else:
            test = int8(num_bit(pos.bplayer) - num_bit(pos.bopponent))
            if test > 0:
                return True, pos.player
            elif test == 0:
                return True, 0
            else:
                return True, int8(-pos.player)


def make(kind, n=0, nvict=0):
    """the plugin of a main*.jl: ('gobang', N, Nvict) | ('connect4') | ('hex', N) | ('reversi8') | ('reversi6')"""
    if kind == "gobang":
        return GoBang(n, nvict)
    if kind == "connect4":
        return FourIARow()
    if kind == "hex":
        return Hex(n)
    if kind == "reversi8":
        return Reversi(8)
    if kind == "reversi6":
        return Reversi(6)
    raise ValueError(kind)
