"""Naive 2-D array models of the five games, written from the RULES of the games (not from the
reference's bitboard code), used to cross-check the oracle's bitboard restatement.

Board cell (r, c) (0-based) corresponds to the reference's bitboard index [i1, i2] = [r+1, c+1],
i.e. bit d1*c + r (column-major, Bitboard.jl:54-57).
"""
import numpy as np


class NaiveLine:
    """k-in-a-row on an R x C board. gravity=True -> Connect4 (stones fall to the largest row index)."""

    def __init__(self, R, C, k, gravity):
        self.R, self.C, self.k, self.gravity = R, C, k, gravity
        self.board = np.zeros((R, C), np.int8)   # +1 first player, -1 second
        self.player = 1
        self.A = C if gravity else R * C

    def cell(self, a):
        if self.gravity:
            col = a
            r = -1
            for i in range(self.R):
                if self.board[i, col] == 0:
                    r = i
                else:
                    break
            return r, col
        return a % self.R, a // self.R

    def can_play(self, a):
        if self.gravity:
            return self.board[0, a] == 0
        r, c = self.cell(a)
        return self.board[r, c] == 0

    def play(self, a):
        r, c = self.cell(a)
        self.board[r, c] = self.player
        self.player = -self.player

    def is_over(self):
        """(flag, winner in absolute colours); only the side that just moved can have won."""
        who = -self.player
        B = self.board == who
        for dr, dc in ((0, 1), (1, 0), (1, 1), (1, -1)):
            for r in range(self.R):
                for c in range(self.C):
                    ok = True
                    for s in range(self.k):
                        rr, cc = r + dr * s, c + dc * s
                        if not (0 <= rr < self.R and 0 <= cc < self.C and B[rr, cc]):
                            ok = False
                            break
                    if ok:
                        return True, who
        return bool((self.board != 0).all()), 0

    def bits(self):
        me = (self.board == self.player)
        op = (self.board == -self.player)
        return me.T.reshape(-1), op.T.reshape(-1)       # column-major flattening


class NaiveHex:
    """Hex on an N x N rhombus.  Six neighbours: (+-1,0), (0,+-1), (-1,+1), (+1,-1).
    First player (+1) connects column 0 to column N-1 ... or rows: decided by `first_axis`."""

    def __init__(self, N, first_axis):
        self.N = N
        self.board = np.zeros((N, N), np.int8)
        self.player = 1
        self.first_axis = first_axis   # axis (0=rows, 1=cols) the FIRST player must span
        self.A = N * N

    def cell(self, a):
        # reference action c (1-based): x = (c-1) div N, y = c - N x ; board cell [i1, i2] = [y+1, x+2]
        x, y = a // self.N, a % self.N
        return y, x                    # (row, col) inside the playable N x N area

    def can_play(self, a):
        r, c = self.cell(a)
        return self.board[r, c] == 0

    def play(self, a):
        r, c = self.cell(a)
        self.board[r, c] = self.player
        self.player = -self.player

    def connected(self, who):
        axis = self.first_axis if who == 1 else 1 - self.first_axis
        N = self.N
        seen = np.zeros((N, N), bool)
        stack = []
        for i in range(N):
            r, c = (0, i) if axis == 0 else (i, 0)
            if self.board[r, c] == who:
                stack.append((r, c))
                seen[r, c] = True
        while stack:
            r, c = stack.pop()
            if (r if axis == 0 else c) == N - 1:
                return True
            for dr, dc in ((1, 0), (-1, 0), (0, 1), (0, -1), (-1, 1), (1, -1)):
                rr, cc = r + dr, c + dc
                if 0 <= rr < N and 0 <= cc < N and not seen[rr, cc] and self.board[rr, cc] == who:
                    seen[rr, cc] = True
                    stack.append((rr, cc))
        return False

    def is_over(self):
        who = -self.player
        return self.connected(who), who


class NaiveReversi:
    def __init__(self, N):
        self.N = N
        self.board = np.zeros((N, N), np.int8)
        h = N // 2
        # reference start (Reversi8x8.jl:10-14): player +1 owns [h, h+1] and [h+1, h] (1-based)
        self.board[h - 1, h] = 1
        self.board[h, h - 1] = 1
        self.board[h, h] = -1
        self.board[h - 1, h - 1] = -1
        self.player = 1
        self.A = N * N + 1

    def flips(self, who, r, c):
        if self.board[r, c] != 0:
            return []
        out = []
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                if dr == 0 and dc == 0:
                    continue
                line = []
                rr, cc = r + dr, c + dc
                while 0 <= rr < self.N and 0 <= cc < self.N and self.board[rr, cc] == -who:
                    line.append((rr, cc))
                    rr += dr
                    cc += dc
                if line and 0 <= rr < self.N and 0 <= cc < self.N and self.board[rr, cc] == who:
                    out += line
        return out

    def legal(self, who):
        return [(r, c) for c in range(self.N) for r in range(self.N) if self.flips(who, r, c)]

    def can_play(self, a):
        if a == self.N * self.N:
            return len(self.legal(self.player)) == 0
        r, c = a % self.N, a // self.N
        return len(self.flips(self.player, r, c)) > 0

    def play(self, a):
        if a != self.N * self.N:
            r, c = a % self.N, a // self.N
            for rr, cc in self.flips(self.player, r, c):
                self.board[rr, cc] = self.player
            self.board[r, c] = self.player
        self.player = -self.player

    def is_over(self):
        if self.legal(1) or self.legal(-1):
            return False, 0
        d = int((self.board == 1).sum()) - int((self.board == -1).sum())
        return True, (d > 0) - (d < 0)

    def bits(self):
        me = (self.board == self.player)
        op = (self.board == -self.player)
        return me.T.reshape(-1), op.T.reshape(-1)
