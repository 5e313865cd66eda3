"""ctypes declarations for libpcpx.so (include/pcpx.h).  Fails loudly if the library is missing:
there is no Python or CPU fallback for the compute path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCPX_LIB") or os.path.join(HERE, "libpcpx.so")  # PCPX_LIB: tuning variants (tools/)

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)

ABI_VERSION = 5  # include/pcpx.h PCPX_ABI_VERSION
PCPX_OK = 0
PCPX_ERR_INVALID = -1
PCPX_ERR_DEVICE = -2
PCPX_ERR_ALLOC = -3
PCPX_ERR_CAPACITY = -4
PCPX_ERR_UNSUPPORTED = -5
PCPX_BUILD_USE_GRID = 1
PCPX_BUILD_COARSE_ORDER = 2
PCPX_BUILD_SHARD = 4
PCPX_BUILD_BORROW_CLOUD = 8
PCPX_BUILD_SHARD_RANGE = 16
UINT64_MAX = 0xFFFFFFFFFFFFFFFF


class BuildParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("grid_min", C.c_float * 3),
                ("grid_max", C.c_float * 3), ("shard_rank", C.c_uint32), ("shard_world", C.c_uint32),
                ("shard_k_hint", C.c_uint32), ("reserved", C.c_uint32), ("shard_first", C.c_uint64), ("shard_count", C.c_uint64)]


class Profile(C.Structure):
    _fields_ = [("launches", C.c_uint32 * 5), ("total_ms", C.c_float * 5)]


K_BUILD, K_KNN, K_NORMALS, K_RANGE, K_QUERY_PREP = range(5)


class PcpxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("pcpx status %d: %s" % (status, message))
        self.status = status


# name -> (restype, argtypes); every symbol declared in include/pcpx.h
SIGNATURES = {
    "pcpx_abi_version": (C.c_int, []),
    "pcpx_last_error": (C.c_char_p, []),
    "pcpx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pcpx_index_create": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(BuildParams), C.c_int, C.POINTER(C.c_void_p)]),
    "pcpx_index_create_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(BuildParams), C.c_int, C.c_void_p,
                                        C.POINTER(C.c_void_p)]),
    "pcpx_index_rebuild": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(BuildParams)]),
    "pcpx_index_rebuild_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(BuildParams)]),
    "pcpx_index_destroy": (None, [C.c_void_p]),
    "pcpx_index_size": (C.c_int, [C.c_void_p, u64p]),
    "pcpx_index_trim": (C.c_int, [C.c_void_p]),
    "pcpx_index_shard_info": (C.c_int, [C.c_void_p, u64p]),
    "pcpx_knn_self_curve_order_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p]),
    "pcpx_index_perm_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcpx_estimate_normals_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "pcpx_index_bbox": (C.c_int, [C.c_void_p, f32p]),
    "pcpx_bounding_box": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, f32p]),
    "pcpx_bounding_box_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]),
    "pcpx_knn_self": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcpx_knn_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "pcpx_knn_self_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "pcpx_knn_self_strided_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "pcpx_normals_knn_self_strided_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p,
                                                    C.c_void_p, C.c_void_p]),
    "pcpx_knn_group_costs_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p, C.c_uint64, u64p]),
    "pcpx_shard_cuts_by_cost": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, u64p]),
    "pcpx_debug_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "pcpx_debug_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "pcpx_debug_group_times": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p]),
    "pcpx_knn_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "pcpx_range_count_self": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "pcpx_range_count_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]),
    "pcpx_range_count_self_dev": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "pcpx_range_count_self_curve_order_dev": (C.c_int, [C.c_void_p, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "pcpx_range_lists_self_dev": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64, u64p]),
    "pcpx_range_sphere_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint64, C.c_void_p,
                                          C.c_void_p, C.c_uint64]),
    "pcpx_range_aabb_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
    "pcpx_normals_knn_self": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcpx_normals_knn_self_curve_order": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcpx_normals_knn_self_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "pcpx_tangent_planes_knn_self": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]),
    "pcpx_mean_knn_distance_self": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]),
    "pcpx_neighbourhoods_self_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    "pcpx_propagate_normal_orientations": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                     u64p]),
    "pcpx_propagate_normal_orientations_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                         C.c_int, C.c_void_p, u64p, C.POINTER(C.c_uint32)]),
    "pcpx_orient_normals_knn_self": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, u64p]),
    "pcpx_oriented_normals_knn_self": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, u64p]),
    "pcpx_normals_from_knn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                        C.c_void_p]),
    "pcpx_estimate_normal": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, f32p]),
    "pcpx_shard_range": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p]),
    "pcpx_bilateral_filter_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int, C.c_void_p]),
    "pcpx_bilateral_filter_normals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int, C.c_void_p]),
    "pcpx_bilateral_filter_points_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int,
                                                   C.c_void_p, C.c_void_p]),
    "pcpx_bilateral_filter_normals_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int,
                                                    C.c_void_p, C.c_void_p]),
    "pcpx_wlop": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int, C.c_int, C.c_void_p]),
    "pcpx_wlop_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_uint64, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p]),
    "pcpx_device_malloc": (C.c_int, [C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "pcpx_device_free": (None, [C.c_void_p, C.c_int]),
    "pcpx_device_trim": (C.c_int, [C.c_int]),
    "pcpx_device_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "pcpx_device_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]),
    "pcpx_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pcpx_comm_init_rank": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pcpx_comm_wrap": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pcpx_comm_destroy": (None, [C.c_void_p]),
    "pcpx_comm_allgather_boxes_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcpx_comm_global_grid_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, f32p]),
    "pcpx_index_synchronize": (C.c_int, [C.c_void_p]),
    "pcpx_debug_knn_stats": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, u64p, C.c_uint64]),
    "pcpx_debug_eps_test_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "pcpx_debug_sort_keys": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]),
    "pcpx_profile_begin": (C.c_int, [C.c_void_p]),
    "pcpx_profile_end": (C.c_int, [C.c_void_p, C.POINTER(Profile)]),
    "pcpx_kd_create": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "pcpx_kd_destroy": (None, [C.c_void_p]),
    "pcpx_kd_size": (C.c_uint64, [C.c_void_p]),
    "pcpx_kd_dims": (C.c_uint32, [C.c_void_p]),
    "pcpx_kd_knn_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcpx_kd_range_aabb_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, the soname libpcpx.so
    needs).  Device pointers and streams can only be exchanged with torch (bench.py, multi-GPU path) if
    both sit on ONE HIP runtime, so when torch is installed its copy is loaded first -- without
    importing torch -- and libpcpx.so then binds to it by soname.  PCPX_HIP_RUNTIME=system skips this."""
    if os.environ.get("PCPX_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # fall back to the RUNPATH copy under /opt/rocm


def load():
    """Load libpcpx.so; raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libpcpx.so is missing at %s: build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "There is no CPU fallback for the pcpx compute path." % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pcpx_abi_version() != ABI_VERSION:
        raise ImportError("libpcpx.so ABI version mismatch")
    _lib = lib
    return lib


def check(status):
    if status != PCPX_OK:
        msg = load().pcpx_last_error()
        raise PcpxError(status, msg.decode("utf-8", "replace") if msg else "")
