"""Game plugin constants (Gobang.jl:2,8-11 exports) resolved through agz_query_game (no GPU needed)."""
import ctypes as C

from . import lib as _lib

KINDS = {"gobang": 0, "connect4": 1, "hex": 2, "reversi8": 3, "reversi6": 4, "extra": 5}   # extra: a plugged-in game (INTEGRATION.md "Adding a game")


class GameSpec:
    """game = 'gobang' | 'connect4' | 'hex' | 'reversi8' | 'reversi6'; n, nvict as `const N, Nvict` (mainGobang.jl:24-26)."""

    def __init__(self, game, n=0, nvict=0):
        self.name = game
        self.kind = KINDS[game] if isinstance(game, str) else int(game)
        self.n, self.nvict = int(n), int(nvict)
        cfg = _lib.Config(game=self.kind, n=self.n, nvict=self.nvict, max_games=1, max_visits=1)
        info = _lib.GameInfo()
        rc = _lib.load_library().agz_query_game(C.byref(cfg), C.byref(info))
        if rc != 0:
            raise ValueError(f"unsupported game parameters {game} n={n} nvict={nvict}")
        self.info = info
        self.maxActions, self.VectorizedState = info.A, info.VS
        self.FeatureSize, self.maxLengthGame = info.FS, info.ML
        self.max_plies, self.pos_image_bytes, self.rec_bytes = info.max_plies, info.pos_image_bytes, info.rec_bytes

    A = property(lambda s: s.maxActions)
    VS = property(lambda s: s.VectorizedState)
    FS = property(lambda s: s.FeatureSize)
    ML = property(lambda s: s.maxLengthGame)


def perft(game, depth, device=0):
    """Perft of the game plugin as the DEVICE runs it (agz_perft): (positions after exactly `depth` plies, [games finished with
    result +1, 0, -1 at any ply <= depth]).  Known-answer test hook; needs a GPU."""
    cfg = _lib.Config(game=game.kind, n=game.n, nvict=game.nvict, max_games=1, max_visits=1, device=int(device))
    nodes, term = C.c_int64(0), (C.c_int64 * 3)()
    L = _lib.load_library()
    rc = L.agz_perft(C.byref(cfg), int(depth), C.byref(nodes), C.byref(term))
    if rc != 0:
        msg = L.agz_last_error(None)
        raise _lib.AgzError(rc, msg.decode() if msg else "agz_perft failed")
    return nodes.value, [term[0], term[1], term[2]]
