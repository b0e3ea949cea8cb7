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
        self.W0 = z(self.H * self.inp) if W0 is None else np.ascontiguousarray(W0, np.float32).reshape(-1)
        self.Wres = z(max(self.T, 1) * self.H * self.H) if Wres is None else np.ascontiguousarray(Wres, np.float32).reshape(-1)
        self.Wp = z(self.A * self.H) if Wp is None else np.ascontiguousarray(Wp, np.float32).reshape(-1)
        self.bp = z(self.A) if bp is None else np.ascontiguousarray(bp, np.float32).reshape(-1)
        self.Wv = z(self.H) if Wv is None else np.ascontiguousarray(Wv, np.float32).reshape(-1)
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
