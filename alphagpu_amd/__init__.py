"""alphagpu_amd — MI355X-native batched AlphaZero self-play (host-side mirror of the reference's
mcts_gpu.jl module over the libagz C ABI).  There is no CPU fallback: every search entry point needs
libagz.so (HIP, gfx950) and a GPU, and fails loudly otherwise."""
from .lib import load_library, LibraryMissing, AgzError  # noqa: F401
from .game import GameSpec  # noqa: F401
from .net import SNetwork2  # noqa: F401
from .pool import PoolSample, Sample  # noqa: F401
from . import mcts_gpu  # noqa: F401
