"""ctypes binding of libhnsw_mi355x.so (C ABI: include/hnsw_mi355x.h).

The library is the product: HIP kernels for gfx950 plus the host index.  If it is missing this
module raises -- there is no Python or CPU fallback for search.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HNSW_MI355X_LIB selects another build of the same library (e.g. the diagnostic stamps build)
LIB_PATH = os.environ.get("HNSW_MI355X_LIB") or os.path.join(_HERE, "libhnsw_mi355x.so")

VEC_QUANT8 = 0
VEC_F32 = 1
UINT32_MAX = 0xFFFFFFFF

OK = 0
ERR_BAD_DIM = -1
ERR_NAN_INPUT = -2
ERR_NODE_NOT_IN_GRAPH = -3
ERR_IO = -4
ERR_HIP = -5
ERR_RCCL = -6
ERR_OOM = -7
ERR_ARG = -8
ERR_EMPTY = -9
ERR_NO_DEVICE = -10
ERR_OVERFLOW = -11
ERR_SELF_CONNECTION = -12


class Params(C.Structure):
    _fields_ = [("ep", C.c_uint32), ("vec_kind", C.c_uint32), ("m", C.c_uint64), ("mmax", C.c_uint64),
                ("mmax0", C.c_uint64), ("ml", C.c_float), ("_pad", C.c_uint32), ("ef_cons", C.c_uint64),
                ("dim", C.c_uint64)]


class QueryStats(C.Structure):
    _fields_ = [("n_dist", C.c_uint32), ("n_exp", C.c_uint32), ("sum_deg", C.c_uint32),
                ("status", C.c_int32)]


class SnapshotDesc(C.Structure):
    """hnsw_snapshot_desc: the seven flat device arrays of the HBM snapshot + its scalar header"""
    _fields_ = [("bytes", C.c_uint64 * 7), ("ptr", C.c_void_p * 7), ("header", C.c_uint32 * 32)]


u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
vp = C.c_void_p

# every symbol include/hnsw_mi355x.h declares: name -> (restype, argtypes)
# int (*hnsw_allgather_fn)(void *ctx, uint64_t bytes_per_rank)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)

SYMBOLS = {
    "hnsw_last_error": (C.c_char_p, []),
    "hnsw_version": (C.c_char_p, []),
    "hnsw_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "hnsw_free": (None, [vp]),
    "hnsw_clone": (C.c_int, [vp, C.POINTER(vp)]),
    "hnsw_get_params": (C.c_int, [vp, C.POINTER(Params)]),
    "hnsw_set_ep": (C.c_int, [vp, C.c_uint32]),
    "hnsw_insert_bulk": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint32, C.c_int]),
    "hnsw_insert_bulk_levels": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint32, C.c_int, u8p]),
    "hnsw_insert_bulk_device": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint32, C.c_int, u8p]),
    "hnsw_sharded_slot_bytes": (C.c_uint64, [vp, C.c_uint32]),
    "hnsw_insert_bulk_sharded": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint32, C.c_int, u8p, C.c_uint32, C.c_uint32,
                                           vp, vp, C.c_uint64, ALLGATHER_FN, vp]),
    "hnsw_insert_vec": (C.c_int, [vp, f32p, u32p]),
    "hnsw_insert_vec_level": (C.c_int, [vp, f32p, C.c_int, u32p]),
    "hnsw_import_points": (C.c_int, [vp, f32p, C.c_uint64, u8p]),
    "hnsw_import_layer": (C.c_int, [vp, C.c_uint32, C.c_uint64, u32p, u64p, u32p]),
    "hnsw_search": (C.c_int, [vp, f32p, C.c_uint32, C.c_uint32, u32p, u32p]),
    "hnsw_search_batch": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint32, C.c_uint32, u32p, f32p, u32p,
                                    C.POINTER(QueryStats)]),
    "hnsw_search_batch_device": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]),
    "hnsw_search_batch_device_finish": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]),
    "hnsw_distance_batch": (C.c_int, [vp, f32p, u32p, C.c_uint64, f32p]),
    "hnsw_search_layer": (C.c_int, [vp, C.c_uint32, f32p, u32p, C.c_uint32, C.c_uint32, u32p, f32p, u32p,
                                    C.POINTER(QueryStats)]),
    "hnsw_brute_force": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint32, u32p, f32p]),
    "hnsw_brute_force_fast": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint32, u32p, f32p]),
    "hnsw_len": (C.c_uint64, [vp]),
    "hnsw_distance": (C.c_int, [vp, C.c_uint32, C.c_uint32, f32p]),
    "hnsw_get_vector": (C.c_int, [vp, C.c_uint32, f32p]),
    "hnsw_get_level": (C.c_int, [vp, C.c_uint32, u32p]),
    "hnsw_get_quant": (C.c_int, [vp, C.c_uint32, u8p, f32p, f32p]),
    "hnsw_layer_count": (C.c_uint32, [vp]),
    "hnsw_layer_nb_nodes": (C.c_uint64, [vp, C.c_uint32]),
    "hnsw_layer_m": (C.c_uint32, [vp, C.c_uint32]),
    "hnsw_layer_nodes": (C.c_int, [vp, C.c_uint32, u32p, C.c_uint64, u64p]),
    "hnsw_neighbors": (C.c_int, [vp, C.c_uint32, C.c_uint32, u32p, C.c_uint32, u32p]),
    "hnsw_export_layer": (C.c_int, [vp, C.c_uint32, u32p, u64p, u32p, u64p, u64p]),
    "hnsw_check_param_compliance": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "hnsw_save": (C.c_int, [vp, C.c_char_p]),
    "hnsw_load": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "hnsw_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "hnsw_set_device": (C.c_int, [vp, C.c_int]),
    "hnsw_upload": (C.c_int, [vp]),
    "hnsw_device_bytes": (C.c_int, [vp, u64p]),
    "hnsw_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "hnsw_get_stat": (C.c_int, [vp, C.c_char_p, u64p]),
    "hnsw_bench_search_threads": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, u32p,
                                            u32p, u64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hnsw_snapshot_describe": (C.c_int, [vp, C.POINTER(SnapshotDesc)]),
    "hnsw_snapshot_adopt": (C.c_int, [vp, C.POINTER(SnapshotDesc)]),
    "hnsw_snapshot_commit": (C.c_int, [vp]),
    "hnsw_bench_batch_threads": (C.c_int, [vp, f32p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_double)]),
    "hnsw_synth_rows": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, f32p, C.c_uint32]),
    "hnsw_draw_levels": (C.c_int, [C.c_uint32, C.c_uint64, u8p]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    soname as /opt/rocm's); whichever copy is loaded first serves both, and torch cannot see a GPU
    through a runtime it did not bring.  bench.py and the multi-GPU path use torch for streams and
    torch.distributed (RCCL), so when torch is installed its copy is mapped first -- without
    importing torch.  Without torch the library binds to the system ROCm as linked."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the native library; fail loudly when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hnsw_rs_amd/csrc` (hipcc, gfx950). There is no fallback path." % LIB_PATH)
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(L, name)  # AttributeError if the library does not export it
        except AttributeError:
            # an A/B build of an older commit (HNSW_MI355X_LIB) may lack the newest entry points; the product
            # library must export every one of them
            if os.environ.get("HNSW_MI355X_LIB"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class HnswError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hnsw_mi355x error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != OK:
        msg = lib().hnsw_last_error()
        raise HnswError(rc, msg.decode("utf-8", "replace") if msg else "")
