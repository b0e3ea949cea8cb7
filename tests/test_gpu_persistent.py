"""GPU parity (-m gpu) of the PERSISTENT self-play kernels (alphagpu_amd/csrc/agz_selfplay_small.hpp): one launch per agz_selfplay /
agz_selfplay_chain call, every workgroup looping over the plies of its own games (search -> root policy -> move choice -> play / isOver ->
sample capture -> next game) without ever meeting the other workgroups.  Replaces the host-driven ply loop of mcts_gpu.jl:494-561.

Every game's samples must be, bit for bit, those of the oracle's LOCK-STEP generation over all the games (results are keyed by game id
and the game's own ply, never by slot or by the moment a game is played).  AGZ_PERSIST=1 forces the form at the small sizes the oracle
finishes in seconds (by default it serves calls with refilled slots on engines of more than 96 slots per CU: the benchmarked shape,
covered at full size in tests/test_gpu_scale_parity.py)."""
import numpy as np
import pytest

import alphagpu_amd as ag
from alphagpu_amd import mcts_gpu as M
import oracle_lib as O
from test_gpu_parity import spec, assert_same_bits

pytestmark = pytest.mark.gpu
KEYS = ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value")


def _nets(name, H=128, T=2):
    g, og = spec(name)
    return g, og, ag.SNetwork2.random(g, H, T), O.OracleNet(og, H, T).bf16()


@pytest.mark.parametrize("name,slots,ngames,V,narrow", [("gobang9", 24, 70, 16, None), ("gobang9", 200, 520, 8, None), ("connect4", 40, 150, 12, None),
                                                        ("connect4", 40, 150, 12, "-1"), ("hex5", 8, 40, 16, None), ("reversi6", 16, 50, 8, None),
                                                        ("tictactoe", 70, 300, 8, None), ("reversi8", 12, 30, 8, None)])
def test_persistent_selfplay_with_refilled_slots_equals_the_lockstep_oracle(name, slots, ngames, V, narrow, monkeypatch):
    """agz_selfplay(ngames > max_games) through k_selfplay_small: PoolSample order, W / D / L, plies — the oracle's lock-step generation."""
    monkeypatch.setenv("AGZ_PERSIST", "1")
    if narrow:
        monkeypatch.setenv("AGZ_NARROW", narrow)
    g, og, net, onet = _nets(name)
    ref = O.selfplay(og, onet, ngames, V, 1.5, 25, 9, 1000)
    with M.Engine(g, slots, V, seed=9, game_id_base=1000, nn_mode=M.NN_BF16, sample_capacity_games=ngames) as e:
        e.set_network(net)
        st = e.selfplay(ngames, V, cpuct=1.5, tau_plies=25)
        assert e.search_form()[0].startswith("k_selfplay_small"), e.search_form()
        if name == "connect4":
            assert ("G=4" in e.search_form()[0]) == (narrow is None), e.search_form()
        s = e.samples()
        assert st["valid"] and st["nsamples"] == ref["n"] and st["wins"] + st["draws"] + st["losses"] == ngames
        assert (st["wins"], st["draws"], st["losses"], st["total_plies"]) == (ref["wins"], ref["draws"], ref["losses"], ref["total_plies"])
        assert st["rollouts"] == V * ref["n"]                    # every (game, ply) was searched exactly once
        for key in KEYS:
            assert_same_bits(s[key], ref[key], key)
        # the engine is reusable: a lock-step generation afterwards (fewer games than slots), in the same form
        st2 = e.selfplay(slots // 2, V, cpuct=1.5, tau_plies=25)
        s2 = e.samples()
        # ... and plain searches work again once roots are set
        e.set_roots(None, L=slots)
        e.search(V, cpuct=1.5, training=True, step=0)
        assert (e.root_visits().sum(1) == V - 1).all()
    ref2 = O.selfplay(og, onet, slots // 2, V, 1.5, 25, 9, 1000)
    assert st2["valid"] and st2["nsamples"] == ref2["n"]
    for key in KEYS:
        assert_same_bits(s2[key], ref2[key], key + " (second call)")


@pytest.mark.parametrize("name,slots,N,V", [("gobang9", 24, 50, 16), ("connect4", 32, 90, 12), ("tictactoe", 64, 200, 8), ("hex5", 70, 150, 8)])
def test_persistent_chain_returns_the_oracles_games_call_by_call(name, slots, N, V, monkeypatch):
    """agz_selfplay_chain through k_selfplay_small: a call ends when ITS games are over (every workgroup looks at the device counter between
    two plies), the games of the next call that were started early stay in their slots — uncompacted — and the next launch goes on with
    them; slots that were left empty take waiting games when that launch starts.  Call by call the oracle's games of the same ids."""
    monkeypatch.setenv("AGZ_PERSIST", "1")
    g, og, net, onet = _nets(name)
    calls = [(N, N), (N, N // 2), (N // 2, 0)] if name != "connect4" else [(N, N), (N // 4, 0), (N // 2, N // 4), (N // 4, 0)]
    total = sum(n for n, _ in calls)
    ref = O.selfplay(og, onet, total, V, 1.5, 25, 9, 700)
    with M.Engine(g, slots, V, seed=9, game_id_base=700, nn_mode=M.NN_BF16, sample_capacity_games=2 * N + 7) as e:
        e.set_network(net)
        k0, rollouts = 0, 0
        for i, (n, nxt) in enumerate(calls):
            st = e.selfplay_chain(n, nxt, V, cpuct=1.5, tau_plies=25)
            assert e.search_form()[0].startswith("k_selfplay_small"), e.search_form()
            s = e.samples()
            sel = (ref["game_id"] >= 700 + k0) & (ref["game_id"] < 700 + k0 + n)
            assert st["valid"] and st["nsamples"] == int(sel.sum()) == len(s["ply"]), (i, st["nsamples"], int(sel.sum()))
            for key in KEYS:
                assert_same_bits(s[key], ref[key][sel], f"call {i}: {key}")
            first = sel & (ref["ply"] == 0)
            res = np.rint(2.0 * ref["value"][first] - 1.0).astype(int)
            assert (st["wins"], st["draws"], st["losses"]) == (int((res == 1).sum()), int((res == 0).sum()), int((res == -1).sum()))
            assert st["total_plies"] == int(sel.sum()) - n
            if nxt and i == 0:
                with pytest.raises(Exception):                   # games of the next call are in flight: their key must not change ...
                    e.set_seed(1234)
                with pytest.raises(Exception):                   # ... and a plain search must not run over their slots
                    e.search(V, cpuct=1.5, training=True, step=0)
            k0 += n
            rollouts += st["rollouts"]
        assert rollouts == V * len(ref["ply"])                   # the chain's work is the work of its games: nothing searched twice or dropped
        st = e.selfplay(slots, V, cpuct=1.5, tau_plies=25)       # a call of its own afterwards starts over
        s = e.samples()
    ref2 = O.selfplay(og, onet, slots, V, 1.5, 25, 9, 700)
    assert st["valid"] and st["nsamples"] == ref2["n"]
    for key in ("game_id", "ply", "move", "policy", "value"):
        assert_same_bits(s[key], ref2[key], key + " (call of its own after the chain)")


def test_persistent_and_ply_loop_forms_return_identical_records(monkeypatch):
    """The same call through both forms of the self-play loop: byte-identical packed records (what the all-gather ships)."""
    g, og, net, onet = _nets("gobang9")
    out = []
    for persist in ("0", "1"):
        monkeypatch.setenv("AGZ_PERSIST", persist)
        with M.Engine(g, 64, 16, seed=5, game_id_base=10, nn_mode=M.NN_BF16, sample_capacity_games=200) as e:
            e.set_network(net)
            st = e.selfplay(200, 16, cpuct=1.5, tau_plies=25)
            assert e.search_form()[0].startswith("k_selfplay_small") == (persist == "1")
            out.append((st["nsamples"], e.samples_packed_host().tobytes()))
    assert out[0] == out[1]


# ---- age classes: workgroups whose games are all old run node rows by the root's legal rank; games migrate between workgroups ------------
@pytest.mark.parametrize("name,slots,ngames,V,backlog", [("gobang9", 256, 520, 32, None), ("hex9", 128, 300, 32, None), ("gobang9", 192, 400, 32, "3")])
def test_persistent_selfplay_with_age_classes_equals_the_lockstep_oracle(name, slots, ngames, V, backlog, monkeypatch):
    """k_selfplay_small<..., AGE>: odd workgroups prefer old games (AGZ_AGE_CLASS=block), even ones hand a game that has reached ply
    A - 64 to the migration queue and start a new one; a workgroup whose games are all old searches with rows by legal rank and reads
    policy_final through the root's legal mask.  Which workgroup plays a game at which moment changes nothing: the oracle's lock-step
    generation, sample for sample.  The third case bounds the queue at three games (most old games stay where they are)."""
    monkeypatch.setenv("AGZ_PERSIST", "1")
    monkeypatch.setenv("AGZ_AGE_CLASS", "block")
    if backlog:
        monkeypatch.setenv("AGZ_AGE_BACKLOG", backlog)
    g, og, net, onet = _nets(name)
    ref = O.selfplay(og, onet, ngames, V, 1.5, 25, 9, 1000)
    with M.Engine(g, slots, V, seed=9, game_id_base=1000, nn_mode=M.NN_BF16, sample_capacity_games=ngames) as e:
        e.set_network(net)
        e.kernel_times(reset=True)
        st = e.selfplay(ngames, V, cpuct=1.5, tau_plies=25)
        assert e.search_form()[0].startswith("k_selfplay_small") and "AGE" in e.search_form()[0], e.search_form()
        searches, ranked, moved = e.age_stats()
        assert searches == ref["n"] and 0 < ranked < searches and moved > 0, (searches, ranked, moved)
        s = e.samples()
        assert st["valid"] and st["nsamples"] == ref["n"] and st["rollouts"] == V * ref["n"]
        assert (st["wins"], st["draws"], st["losses"], st["total_plies"]) == (ref["wins"], ref["draws"], ref["losses"], ref["total_plies"])
        for key in KEYS:
            assert_same_bits(s[key], ref[key], key)


def test_persistent_chain_with_age_classes_keeps_waiting_games_between_calls(monkeypatch):
    """A chain whose calls end while games wait in the migration queue: they are in flight like the games in slots, the next launch's
    empty slots and old-preferring workgroups pick them up, every call returns the oracle's games of its ids."""
    monkeypatch.setenv("AGZ_PERSIST", "1")
    monkeypatch.setenv("AGZ_AGE_CLASS", "block")
    g, og, net, onet = _nets("gobang9")
    slots, V, calls = 192, 32, [(210, 210), (210, 110), (110, 0)]
    ref = O.selfplay(og, onet, sum(n for n, _ in calls), V, 1.5, 25, 9, 700)
    with M.Engine(g, slots, V, seed=9, game_id_base=700, nn_mode=M.NN_BF16, sample_capacity_games=540) as e:
        e.set_network(net)
        e.kernel_times(reset=True)
        k0, rollouts = 0, 0
        for i, (n, nxt) in enumerate(calls):
            st = e.selfplay_chain(n, nxt, V, cpuct=1.5, tau_plies=25)
            s = e.samples()
            sel = (ref["game_id"] >= 700 + k0) & (ref["game_id"] < 700 + k0 + n)
            assert st["valid"] and st["nsamples"] == int(sel.sum()) == len(s["ply"]), (i, st["nsamples"], int(sel.sum()))
            for key in KEYS:
                assert_same_bits(s[key], ref[key][sel], f"call {i}: {key}")
            k0 += n
            rollouts += st["rollouts"]
        assert rollouts == V * len(ref["ply"])
        searches, ranked, moved = e.age_stats()
        assert 0 < ranked < searches and moved > 0, (searches, ranked, moved)


def test_network_tag_travels_with_every_sample(monkeypatch):
    """agz_set_network_tag: byte 17 of a packed record names the network that searched the ply — in a chain whose network changes between
    calls, the games a call starts early for the next one carry the OLD tag on their first plies and the new one afterwards."""
    monkeypatch.setenv("AGZ_PERSIST", "1")
    g, og, net, onet = _nets("gobang9")
    net2 = ag.SNetwork2.random(g, 128, 2, 77)
    with M.Engine(g, 64, 16, seed=3, nn_mode=M.NN_BF16, sample_capacity_games=200) as e:
        e.set_network(net); e.set_network_tag(5)
        e.selfplay_chain(100, 100, 16, cpuct=1.5, tau_plies=25)
        r1 = e.samples_packed_host().copy()
        e.set_network(net2); e.set_network_tag(6)
        e.selfplay_chain(100, 0, 16, cpuct=1.5, tau_plies=25)
        r2 = e.samples_packed_host().copy()
    assert (r1[:, 17] == 5).all()
    gid, ply, tag = r2[:, 0:4].copy().view(np.uint32)[:, 0], r2[:, 4:8].copy().view(np.int32)[:, 0], r2[:, 17]
    assert set(np.unique(tag)) == {5, 6}                          # some of the second call's games began under the first network
    for gme in np.unique(gid):                                    # within a game the tag never goes back
        t = tag[gid == gme][np.argsort(ply[gid == gme])]
        assert (np.diff(t.astype(int)) >= 0).all()


# ---- 512-wide trunks: k_selfplay_big ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,slots,ngames,V,big4", [("gobang9", 70, 160, 16, "0"), ("reversi8", 40, 90, 8, "0"), ("hex9", 24, 60, 16, "0"),
                                                      ("gobang9", 200, 330, 16, "1"), ("reversi8", 140, 230, 8, "1"), ("hex9", 24, 60, 16, "1"),
                                                      ("connect4", 140, 300, 8, "1"), ("reversi6", 70, 160, 8, "1"), ("hex5", 150, 320, 8, "1")])
def test_persistent_selfplay_with_a_wide_trunk_equals_the_lockstep_oracle(name, slots, ngames, V, big4, monkeypatch):
    """k_selfplay_big (the rollout loop of k_search_big's 64-game workgroups inside the persistent ply loop): refilled call, then a chain.
    big4: k_selfplay_big4 — ONE 128-game workgroup per CU, sixteen trees per wave on 4 lanes each, the network pass on 128 leaves (the default
    above 64 slots per CU; AGZ_BIG4=1 forces it at these sizes: a ragged last workgroup, a workgroup with idle waves)."""
    monkeypatch.setenv("AGZ_PERSIST", "1")
    monkeypatch.setenv("AGZ_BIG4", big4)
    g, og, net, onet = _nets(name, 512, 1)
    ref = O.selfplay(og, onet, ngames, V, 1.5, 25, 9, 1000)
    with M.Engine(g, slots, V, seed=9, game_id_base=1000, nn_mode=M.NN_BF16, sample_capacity_games=ngames) as e:
        e.set_network(net)
        st = e.selfplay(ngames, V, cpuct=1.5, tau_plies=25)
        assert e.search_form()[0].startswith("k_selfplay_big4" if big4 == "1" else "k_selfplay_big<"), e.search_form()
        s = e.samples()
        assert st["valid"] and st["nsamples"] == ref["n"] and st["rollouts"] == V * ref["n"]
        assert (st["wins"], st["draws"], st["losses"], st["total_plies"]) == (ref["wins"], ref["draws"], ref["losses"], ref["total_plies"])
        for key in KEYS:
            assert_same_bits(s[key], ref[key], key)
        # a chain of two calls on the same engine: ids start over at game_id_base
        n1 = ngames // 2
        k0 = 0
        for n, nxt in ((n1, ngames - n1), (ngames - n1, 0)):
            st = e.selfplay_chain(n, nxt, V, cpuct=1.5, tau_plies=25)
            s = e.samples()
            sel = (ref["game_id"] >= 1000 + k0) & (ref["game_id"] < 1000 + k0 + n)
            assert st["valid"] and st["nsamples"] == int(sel.sum())
            for key in KEYS:
                assert_same_bits(s[key], ref[key][sel], f"chain call at {k0}: {key}")
            k0 += n


# ---- the exchange step behind the C ABI (agz_comm_*: RCCL bound by libagz) ---------------------------------------------------------------
def test_rccl_exchange_through_the_c_abi_returns_the_engines_records():
    """One rank over RCCL through include/agz.h (agz_comm_create / agz_allgather_samples / _start / _wait / agz_comm_fetch_records):
    byte-identical to the engine's own packed records (agz_get_samples_packed) and to the torch.distributed form of the exchange — blocking
    form, pipelined form with an agreed count, and a count that is too small (the second collective inside _wait)."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from alphagpu_amd import shard
    g, og, net, onet = _nets("gobang9")
    with M.Engine(g, 64, 16, seed=5, game_id_base=10, nn_mode=M.NN_BF16, sample_capacity_games=200) as e:
        e.set_network(net)
        st = e.selfplay(200, 16, cpuct=1.5, tau_plies=25)
        n, rb = st["nsamples"], g.rec_bytes
        want = e.samples_packed_host().copy().reshape(-1)
        ex = shard.CommExchange(e, 0, 1, n + 100)
        try:
            parts, counts = ex.allgather()
            assert list(counts) == [n] and np.array_equal(parts[0], want), "blocking form"
            for send in (n + 50, n - 37, 1):                      # enough room; too small (second collective); far too small
                ex.start(units=1, send_records=send)
                parts, counts = ex.wait()
                assert list(counts) == [n] and np.array_equal(parts[0], want), f"pipelined form, {send} records agreed"
            # two in flight, then a third must be refused until one has been waited for
            ex.start(units=1, send_records=n); ex.start(units=1, send_records=n)
            with pytest.raises(RuntimeError):
                ex.start(units=1, send_records=n)
            for _ in range(2):
                parts, counts = ex.wait()
                assert np.array_equal(parts[0], want)
        finally:
            ex.close()
        # the torch.distributed form of the same exchange (backend "nccl" = RCCL), one rank
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        try:
            buf = torch.empty((n + 100) * rb, dtype=torch.uint8, device="cuda")
            assert e.samples_packed_into(buf.data_ptr(), n + 100) == n
            tparts, tcounts = shard.allgather_records(buf, n, rb)
            assert int(tcounts[0]) == n and np.array_equal(tparts[0].cpu().numpy(), want)
        finally:
            dist.destroy_process_group()


@pytest.mark.parametrize("persist", ["1", "0"])
def test_chain_with_a_different_network_per_call_equals_the_oracle_run_the_tags_describe(persist, monkeypatch):
    """What a chain MEANS when the network changes between calls (include/agz.h agz_selfplay_chain / agz_set_network_tag; the reference
    plays every game of a generation with one actor, selfplay.jl:34-56): three calls with three networks; every sample's tag names the
    network that searched it, and the whole chain equals, sample for sample, the oracle's lock-step run in which every (game, ply) is
    searched by the network its tag names (agzo_selfplay_tagged)."""
    monkeypatch.setenv("AGZ_PERSIST", persist)
    g, og = spec("gobang9")
    seeds = (0x5EED, 77, 78)
    gnets = [ag.SNetwork2.random(g, 128, 2, sd) for sd in seeds]
    onets = [O.OracleNet(og, 128, 2, sd).bf16() for sd in seeds]
    slots, V, N = 48, 16, 60
    calls = [(N, N), (N, N), (N, 0)]
    recs = []
    with M.Engine(g, slots, V, seed=9, game_id_base=700, nn_mode=M.NN_BF16, sample_capacity_games=2 * N) as e:
        for i, (n, nxt) in enumerate(calls):
            e.set_network(gnets[i]); e.set_network_tag(i)
            st = e.selfplay_chain(n, nxt, V, cpuct=1.5, tau_plies=25)
            assert st["valid"]
            recs.append(e.samples_packed_host().copy())
    total = sum(n for n, _ in calls)
    allr = np.concatenate(recs)
    gid = allr[:, 0:4].copy().view(np.uint32)[:, 0].astype(np.int64) - 700
    ply = allr[:, 4:8].copy().view(np.int32)[:, 0]
    tag = allr[:, 17]
    tags = np.zeros((total, g.max_plies), np.uint8)
    for k in range(total):
        m = gid == k
        t = np.zeros(g.max_plies, np.uint8)
        t[ply[m]] = tag[m]
        last = int(ply[m].max())
        t[last + 1:] = t[last]
        tags[k] = t
    assert len(np.unique(tag)) == 3 and ((tags[:, :1] != tags[:, 1:]).any(axis=1)).sum() > 0, "no game was searched by two networks"
    call_of = np.repeat(np.arange(len(calls)), [n for n, _ in calls])
    # a game is searched by the network of the call that returns it or — its first plies, or even all of them — by the one before
    assert (tags <= call_of[:, None]).all() and (tags >= call_of[:, None] - 1).all() and (tags[:, 0] < call_of).any()
    assert (np.diff(tags.astype(int), axis=1) >= 0).all()
    ref = O.selfplay(og, None, total, V, 1.5, 25, 9, 700, nets=onets, tags=tags)
    from alphagpu_amd import shard
    k0 = 0
    for i, (n, _) in enumerate(calls):
        got = shard.unpack_records(recs[i].reshape(-1), len(recs[i]), g)
        sel = (ref["game_id"] >= 700 + k0) & (ref["game_id"] < 700 + k0 + n)
        for key in ("game_id", "ply", "move", "player", "state", "fstate", "policy", "value"):
            assert_same_bits(got[key], ref[key][sel], f"call {i}: {key}")
        k0 += n


@pytest.mark.parametrize("persist", ["1", "0"])
def test_chain_may_start_games_beyond_the_next_call(persist, monkeypatch):
    """agz_selfplay_chain(ngames, next_ngames > the next call's size): short calls (a host loop of one generation per call) whose slots
    take games of the call AFTER the next one too; a game may be returned by a call two calls after the one that started it.  Every call
    returns the oracle's games of its ids."""
    monkeypatch.setenv("AGZ_PERSIST", persist)
    g, og, net, onet = _nets("gobang9")
    slots, V, N = 64, 16, 20                                     # calls of 20 games on 64 slots, 60 later games announced every time
    ncalls = 8
    ref = O.selfplay(og, onet, ncalls * N, V, 1.5, 25, 9, 700)
    with M.Engine(g, slots, V, seed=9, game_id_base=700, nn_mode=M.NN_BF16, sample_capacity_games=N + 3 * N) as e:
        e.set_network(net)
        rollouts = 0
        for i in range(ncalls):
            nxt = min(3 * N, (ncalls - 1 - i) * N)
            st = e.selfplay_chain(N, nxt, V, cpuct=1.5, tau_plies=25)
            s = e.samples()
            sel = (ref["game_id"] >= 700 + i * N) & (ref["game_id"] < 700 + (i + 1) * N)
            assert st["valid"] and st["nsamples"] == int(sel.sum()) == len(s["ply"]), (i, st["nsamples"], int(sel.sum()))
            for key in KEYS:
                assert_same_bits(s[key], ref[key][sel], f"call {i}: {key}")
            rollouts += st["rollouts"]
        assert rollouts == V * len(ref["ply"])
