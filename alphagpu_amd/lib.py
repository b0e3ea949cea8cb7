"""ctypes binding of libagz.so (include/agz.h)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AGZ_LIB_PATH") or os.path.join(_HERE, "libagz.so")   # (AGZ_LIB_PATH: an alternative build of the library, A/B measurements)
CSRC = os.path.join(_HERE, "csrc")
_LIB = None


class LibraryMissing(RuntimeError):
    pass


class AgzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libagz error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("game", C.c_int32), ("n", C.c_int32), ("nvict", C.c_int32), ("max_games", C.c_int32),
                ("max_visits", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint64),
                ("game_id_base", C.c_uint32), ("nn_mode", C.c_int32), ("sample_capacity_games", C.c_int32),
                ("reserved", C.c_int32 * 3)]


class GameInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("A", "VS", "FS", "ML", "max_plies", "pos_image_bytes", "rec_bytes", "reserved")]


class SelfplayStats(C.Structure):
    _fields_ = [("nsamples", C.c_int64), ("total_plies", C.c_int64), ("wins", C.c_int64), ("draws", C.c_int64),
                ("losses", C.c_int64), ("rollouts", C.c_int64), ("plies", C.c_int32), ("faults", C.c_int32),
                ("search_seconds", C.c_double), ("total_seconds", C.c_double)]


def build_library():
    """Compile libagz.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", CSRC])


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(f"{LIB_PATH} not found: build it with `make -C {CSRC}` "
                             "(alphagpu_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, f32p = C.c_void_p, C.c_void_p
    L.agz_query_game.argtypes = [C.POINTER(Config), C.POINTER(GameInfo)]
    L.agz_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.agz_destroy.argtypes = [vp]
    L.agz_destroy.restype = None
    L.agz_last_error.argtypes = [vp]
    L.agz_last_error.restype = C.c_char_p
    L.agz_get_info.argtypes = [vp, C.POINTER(GameInfo)]
    L.agz_set_network.argtypes = [vp, C.c_int, C.c_int] + [f32p] * 6
    L.agz_set_network_slot.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [f32p] * 6
    L.agz_init_weights.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int] + [f32p] * 6
    L.agz_set_roots.argtypes = [vp, vp, C.c_int, vp, C.c_int]
    L.agz_search.argtypes = [vp, C.c_int, C.c_float, C.c_int, C.c_uint32]
    L.agz_search_actor.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint32]
    L.agz_search_begin.argtypes = [vp, C.c_float, C.c_int, C.c_uint32]
    L.agz_rollout_select.argtypes = [vp, C.c_uint32, C.c_int]
    L.agz_rollout_eval.argtypes = [vp]
    L.agz_get_eval.argtypes = [vp, f32p, f32p]
    L.agz_inject_eval.argtypes = [vp, f32p, f32p]
    L.agz_get_logits.argtypes = [vp, f32p, f32p]
    L.agz_set_seed.argtypes = [vp, C.c_uint64]
    L.agz_get_search_form.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int]
    L.agz_rollout_expand_backup.argtypes = [vp]
    L.agz_search_end.argtypes = [vp]
    for n in ("policy", "batch", "leaf_batch", "root_visits", "root_q", "leaf", "node_count"):
        getattr(L, "agz_get_" + n).argtypes = [vp, vp]
    L.agz_get_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.agz_selfplay.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(SelfplayStats)]
    L.agz_selfplay_chain.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(SelfplayStats)]
    L.agz_duel.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int64 * 3)]
    L.agz_get_samples.argtypes = [vp] + [vp] * 8
    L.agz_get_samples_packed.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.agz_unpack_records.argtypes = [C.POINTER(GameInfo), vp, C.c_int64] + [vp] * 8
    L.agz_perft.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64 * 3)]
    L.agz_stream.argtypes = [vp]
    L.agz_stream.restype = vp
    L.agz_synchronize.argtypes = [vp]
    L.agz_get_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.agz_set_profiling.argtypes = [vp, C.c_int]
    L.agz_get_tree_busy_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.agz_get_nn_leaves.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.agz_set_network_tag.argtypes = [vp, C.c_uint32]
    i64p = C.POINTER(C.c_int64)
    L.agz_comm_unique_id.argtypes = [vp]
    L.agz_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int64, C.POINTER(vp)]
    L.agz_comm_destroy.argtypes = [vp]
    L.agz_comm_destroy.restype = None
    L.agz_comm_last_error.argtypes = [vp]
    L.agz_comm_last_error.restype = C.c_char_p
    L.agz_allgather_samples.argtypes = [vp, vp, i64p]
    L.agz_allgather_samples_start.argtypes = [vp, vp, C.c_int64]
    L.agz_allgather_samples_wait.argtypes = [vp, i64p, i64p]
    L.agz_comm_fetch_records.argtypes = [vp, C.c_int, vp, C.c_int64, C.c_int64]
    L.agz_comm_records_device.argtypes = [vp, C.c_int]
    L.agz_comm_records_device.restype = vp
    L.agz_get_age_stats.argtypes = [vp, C.POINTER(C.c_uint64 * 3)]
    if hasattr(L, "agz_comm_post_status") or not os.environ.get("AGZ_LIB_PATH"):   # (an A/B library named by AGZ_LIB_PATH may be a build of an earlier round)
        L.agz_get_samples_packed_async.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
        L.agz_comm_post_status.argtypes = [vp, C.c_int]
        L.agz_comm_get_statuses.argtypes = [vp, C.POINTER(C.c_int32)]
        L.agz_allgather_samples_status.argtypes = [vp, vp, C.c_int, i64p, C.POINTER(C.c_int32)]
    _LIB = L
    return L
