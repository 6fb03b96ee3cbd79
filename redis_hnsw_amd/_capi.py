"""ctypes view of include/hnsw_mi355x.h.  Loading fails loudly when the HIP
library has not been built: there is no Python / CPU fallback."""
import ctypes as C
import os

from . import build as _build

OK, ERR_DIM_MISMATCH, ERR_DUPLICATE, ERR_NOT_FOUND, ERR_DEVICE, ERR_INVALID, ERR_CAPACITY = range(7)


class Counters(C.Structure):
    _fields_ = [("n_dist", C.c_uint64), ("n_ids", C.c_uint64), ("n_expand", C.c_uint64), ("n_spill", C.c_uint64)]


class Info(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("m", C.c_uint32), ("m_max", C.c_uint32), ("m_max0", C.c_uint32),
                ("ef_construction", C.c_uint32), ("node_count", C.c_uint32), ("max_layer", C.c_uint32),
                ("enterpoint", C.c_int64), ("stride0", C.c_uint32), ("stride_upper", C.c_uint32),
                ("max_degree0", C.c_uint32), ("max_degree_upper", C.c_uint32), ("hbm_bytes", C.c_uint64),
                ("allocated_ids", C.c_uint32)]


class Replica(C.Structure):
    _fields_ = [("n", C.c_uint32), ("dim", C.c_uint32), ("upper_used", C.c_uint32), ("stride0", C.c_uint32),
                ("stride_upper", C.c_uint32), ("max_layer", C.c_uint32), ("max_degree0", C.c_uint32),
                ("max_degree_upper", C.c_uint32), ("n_dead", C.c_uint32), ("asymmetric", C.c_uint32), ("format", C.c_uint32),
                ("reserved", C.c_uint32), ("enterpoint", C.c_int64), ("vec_bytes", C.c_uint64), ("adj0_bytes", C.c_uint64),
                ("adj_upper_bytes", C.c_uint64), ("vec", C.c_void_p), ("adj0", C.c_void_p), ("adj_upper", C.c_void_p),
                ("upper_base", C.c_void_p), ("levels", C.c_void_p)]

    SCALARS = ("n", "dim", "upper_used", "stride0", "stride_upper", "max_layer", "max_degree0", "max_degree_upper", "n_dead",
               "asymmetric", "format", "enterpoint")


class Pipeline(C.Structure):
    _fields_ = [("lanes", C.c_uint32), ("overlap", C.c_int32), ("probe_ratio", C.c_float), ("priorities", C.c_uint32),
                ("chunk", C.c_uint32), ("min_batch", C.c_uint32), ("hw_queues_env", C.c_uint32)]


fp = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
H = C.c_void_p

# every symbol include/hnsw_mi355x.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "hnsw_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(H)]),
    "hnsw_destroy": (None, [H]),
    "hnsw_last_error": (C.c_char_p, [H]),
    "hnsw_add": (C.c_int, [H, fp, C.c_uint32, C.c_int32, u32p, u32p, C.c_uint32, u32p]),
    "hnsw_add_batch": (C.c_int, [H, fp, C.c_uint32, C.c_uint32, i32p, C.c_uint32]),
    "hnsw_delete": (C.c_int, [H, C.c_uint32, u32p, C.c_uint32, u32p]),
    "hnsw_search": (C.c_int, [H, fp, C.c_uint32, C.c_uint32, u32p, fp, u32p]),
    "hnsw_search_batch": (C.c_int, [H, fp, C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp, u32p]),
    "hnsw_search_batch_device": (C.c_int, [H, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "hnsw_pipeline_info": (C.c_int, [H, C.POINTER(Pipeline)]),
    "hnsw_replica_view": (C.c_int, [H, C.POINTER(Replica)]),
    "hnsw_replica_prepare": (C.c_int, [H, C.POINTER(Replica)]),
    "hnsw_replica_commit": (C.c_int, [H, C.POINTER(Replica), C.POINTER(C.c_uint8)]),
    "hnsw_get_tombstones": (C.c_int, [H, C.POINTER(C.c_uint8)]),
    "hnsw_import": (C.c_int, [H, C.c_uint32, fp, u32p, C.c_int64, C.c_uint32, C.POINTER(u64p), C.POINTER(u32p)]),
    "hnsw_get_info": (C.c_int, [H, C.POINTER(Info)]),
    "hnsw_get_levels": (C.c_int, [H, u32p]),
    "hnsw_get_level": (C.c_int, [H, C.c_uint32, u32p]),
    "hnsw_get_vector": (C.c_int, [H, C.c_uint32, fp]),
    "hnsw_get_neighbors": (C.c_int, [H, C.c_uint32, C.c_uint32, u32p, C.c_uint32, u32p]),
    "hnsw_layer_nnz": (C.c_int, [H, C.c_uint32, u64p]),
    "hnsw_export_layer": (C.c_int, [H, C.c_uint32, u64p, u32p]),
    "hnsw_serialize_size": (C.c_int, [H, u64p]),
    "hnsw_serialize": (C.c_int, [H, C.c_void_p, C.c_uint64, u64p]),
    "hnsw_deserialize": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(H)]),
    "hnsw_set_tuning": (C.c_int, [H, C.c_char_p, C.c_int64]),
    "hnsw_get_counters": (C.c_int, [H, C.POINTER(Counters), C.POINTER(Counters)]),
    "hnsw_reset_counters": (C.c_int, [H]),
    "hnsw_get_tie_counters": (C.c_int, [H, u64p]),
    "hnsw_last_search_kernel_ms": (C.c_int, [H, fp]),
    "hnsw_metric_pairs": (C.c_int, [C.c_int, fp, fp, C.c_uint32, C.c_uint32, fp]),
    # one process, several GPUs
    "hnsw_group_create": (C.c_int, [H, C.POINTER(C.c_int), C.c_uint32, C.c_uint64, C.POINTER(H)]),
    "hnsw_group_destroy": (None, [H]),
    "hnsw_group_last_error": (C.c_char_p, [H]),
    "hnsw_group_size": (C.c_uint32, [H]),
    "hnsw_group_member": (H, [H, C.c_uint32]),
    "hnsw_group_refresh": (C.c_int, [H]),
    "hnsw_group_search_batch": (C.c_int, [H, fp, C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp, u32p]),
    "hnsw_group_add": (C.c_int, [H, fp, C.c_uint32, C.c_int32, u32p, u32p, C.c_uint32, u32p]),
    "hnsw_group_delete": (C.c_int, [H, C.c_uint32, u32p, C.c_uint32, u32p]),
    "hnsw_group_add_batch": (C.c_int, [H, fp, C.c_uint32, C.c_uint32, i32p, C.c_uint32]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; if
    this library pulls in /opt/rocm's copies first and torch is imported later, torch's copy finds the device
    already claimed ("No HIP GPUs are available").  Loading torch's copy first (when torch is installed; torch
    itself is NOT imported) makes both resolve to the same runtime whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                      # torch's runtime is already in the process
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """dlopen redis_hnsw_amd/lib/libhnsw_mi355x.so and bind every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("HNSW_MI355X_LIB", _build.LIB_PATH)  # override: instrumented dev builds
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m redis_hnsw_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # More than two search streams only overlap if each has a hardware queue of its own: the HIP runtime's
    # default of 4 is shared with the null stream and the engine's stream (read once, when the runtime starts).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)  # AttributeError if the .so does not export it
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib
