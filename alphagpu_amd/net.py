"""snetwork2 weight container (DenseNet.jl:279-286) in Flux's (out,in) column-major layout, i.e. what
convert_back(net) (DenseNet.jl:331-333) hands to mcts_gpu.mcts."""
import ctypes as C

import numpy as np

from . import lib as _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class SNetwork2:
    def __init__(self, game, n_filter, n_tower, W0=None, Wres=None, Wp=None, bp=None, Wv=None, bv=None):
        self.inp, self.H, self.T, self.A = 2 * game.VS, int(n_filter), int(n_tower), game.A
        z = lambda n: np.zeros(n, np.float32)  # noqa: E731

        def flat(w, rows):
            """1-D input: already Flux's memory order (W[o + out*i]).  2-D (out, in) input: flattened column-major, which is
            that order.  Wres may be (T, H, H): every block column-major."""
            w = np.asarray(w, np.float32)
            if w.ndim == 1:
                return np.ascontiguousarray(w)
            if w.ndim == 2:
                if w.shape[0] != rows:
                    raise ValueError(f"weight matrix must be (out={rows}, in), got {w.shape}")
                return np.ascontiguousarray(w.reshape(-1, order="F"))
            if w.ndim == 3:
                return np.ascontiguousarray(np.concatenate([b.reshape(-1, order="F") for b in w]))
            raise ValueError("weights must be 1-D (Flux memory order), 2-D (out, in) or 3-D (T, out, in)")
        self.W0 = z(self.H * self.inp) if W0 is None else flat(W0, self.H)
        self.Wres = z(max(self.T, 1) * self.H * self.H) if Wres is None else flat(Wres, self.H)
        self.Wp = z(self.A * self.H) if Wp is None else flat(Wp, self.A)
        self.bp = z(self.A) if bp is None else np.ascontiguousarray(bp, np.float32).reshape(-1)
        self.Wv = z(self.H) if Wv is None else flat(Wv, 1)
        self.bv = z(1) if bv is None else np.ascontiguousarray(bv, np.float32).reshape(-1)

    @classmethod
    def random(cls, game, n_filter, n_tower, seed=0x5EED):
        """ressimplesf(...) random init (DenseNet.jl:193-198): Flux glorot_uniform weights, zero biases."""
        net = cls(game, n_filter, n_tower)
        rc = _lib.load_library().agz_init_weights(seed, net.inp, net.H, net.T, net.A, _p(net.W0), _p(net.Wres),
                                                   _p(net.Wp), _p(net.bp), _p(net.Wv), _p(net.bv))
        if rc != 0:
            raise ValueError("agz_init_weights failed")
        return net

    def pointers(self):
        return [_p(x) for x in (self.W0, self.Wres, self.Wp, self.bp, self.Wv, self.bv)]
