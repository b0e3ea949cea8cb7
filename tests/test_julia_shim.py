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
    for sym in ("agz_comm_unique_id", "agz_comm_create", "agz_comm_destroy", "agz_allgather_samples", "agz_comm_fetch_records", "agz_unpack_records",
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
