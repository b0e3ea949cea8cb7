# first difference between a GPU generation and the oracle's (by game and ply): which field, which game, which ply
import sys, os
ROOT = os.environ.get("AGZ_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import alphagpu_amd.lib as aglib
if os.environ.get('AGZ_LIB'): aglib.LIB_PATH = os.environ['AGZ_LIB']
import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import common, oracle_lib as O
name, n, V, H, T, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
exact = os.environ.get("FUZZ_EXACT") == "1"
kind, nn, k = common.GAMES[name]
g, og = ag.GameSpec(kind, nn, k), O.make_game(kind, nn, k)
net, onet = ag.SNetwork2.random(g, H, T, 0x5EED + seed), O.OracleNet(og, H, T, 0x5EED + seed)
ref = O.selfplay(og, onet if exact else onet.bf16(), n, V, 1.5, 25, seed, 1000 * seed)
with M.Engine(g, n, V, seed=seed, game_id_base=1000 * seed, nn_mode=M.NN_EXACT if exact else M.NN_BF16, sample_capacity_games=n) as e:
    e.set_network(net)
    st = e.selfplay(n, V, cpuct=1.5, tau_plies=25)
    s = e.samples()
    print("form", e.search_form())
def index(d):
    return {(int(a), int(b)): i for i, (a, b) in enumerate(zip(d["game_id"], d["ply"]))}
ia, ib = index(s), index(ref)
bad = []
for key in sorted(set(ia) | set(ib), key=lambda x: (x[1], x[0])):
    if key not in ia or key not in ib:
        bad.append((key, "missing in " + ("gpu" if key not in ia else "oracle"))); continue
    i, j = ia[key], ib[key]
    for f in ("state", "player", "policy", "move"):
        a, b = np.asarray(s[f][i]), np.asarray(ref[f][j])
        if not np.array_equal(a.view(np.uint8) if a.ndim else a, b.view(np.uint8) if b.ndim else b):
            d = ""
            if f == "policy":
                w = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
                d = f" {len(w)} entries, first action {w[0]}: gpu {a[w[0]]!r} oracle {b[w[0]]!r}"
            bad.append((key, f + d)); break
    if len(bad) >= 4: break
print("samples", st["nsamples"], ref["n"])
for b in bad: print(b)
