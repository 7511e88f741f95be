"""ctypes binding of include/vdb_flat.h.  There is no CPU path: if libvdbflat.so is missing
or no MI355X is visible, creating an index raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VDB_LIB names another build of the same ABI (tools/*.sh point it at libvdbflat_diag.so, the -DVDB_DIAG build with the
# ablation / A-B knobs); the default is the release library, which reads no environment itself
LIB_PATH = os.environ.get("VDB_LIB") or os.path.join(HERE, "libvdbflat.so")

OK, ERR_DIMENSION_MISMATCH, ERR_INVALID_VECTOR, ERR_NAN, ERR_DEVICE, ERR_INVALID_ARGUMENT, ERR_NOT_FOUND = range(7)

# every symbol include/vdb_flat.h declares
SYMBOLS = [
    "vdb_flat_create", "vdb_flat_destroy", "vdb_flat_create_sharded", "vdb_flat_shards", "vdb_flat_shard_len", "vdb_flat_set_exchange", "vdb_flat_shard_stats", "vdb_flat_add", "vdb_flat_add_bulk", "vdb_flat_add_bulk_device",
    "vdb_flat_load_vector_file", "vdb_flat_remove", "vdb_flat_get_vector", "vdb_flat_len", "vdb_flat_metric", "vdb_flat_dim",
    "vdb_flat_reserve", "vdb_flat_flush", "vdb_flat_search", "vdb_flat_search_batch",
    "vdb_flat_search_batch_device", "vdb_flat_search_batch_device_begin", "vdb_flat_search_batch_device_finish", "vdb_flat_search_batch_device_submit", "vdb_flat_search_batch_device_wait", "vdb_flat_distances_batch", "vdb_merge_topk_device", "vdb_merge_topk_packed_device", "vdb_flat_set_profile", "vdb_flat_last_stats", "vdb_flat_last_stats_ex", "vdb_flat_set_screen", "vdb_flat_set_wide", "vdb_flat_set_shadow", "vdb_flat_set_sample_cache", "vdb_flat_set_tiers", "vdb_flat_debug_screen_scores", "vdb_flat_debug_rows", "vdb_flat_debug_row_info", "vdb_flat_debug_last_thresholds", "vdb_flat_debug_cert_probe", "vdb_last_error",
    "vdb_abi_version", "vdb_build_arch",
    # include/vdb_hnsw.h
    "vdb_hnsw_create", "vdb_hnsw_destroy", "vdb_hnsw_add", "vdb_hnsw_add_bulk", "vdb_hnsw_remove", "vdb_hnsw_search_batch",
    "vdb_hnsw_len", "vdb_hnsw_metric", "vdb_hnsw_get_vector", "vdb_hnsw_neighbors", "vdb_hnsw_node_level",
    "vdb_hnsw_entry_point", "vdb_hnsw_stats", "vdb_hnsw_set_traversal", "vdb_hnsw_set_build", "vdb_hnsw_build_stats", "vdb_hnsw_build_times",
    # include/vdb_shard.h
    "vdb_shard_unique_id", "vdb_shard_group_create", "vdb_shard_group_destroy", "vdb_shard_group_rank", "vdb_shard_group_world",
    "vdb_shard_range", "vdb_flat_search_batch_sharded", "vdb_shard_group_last_stats",
]

_lib = None


def _preload_torch_hip():
    """PyTorch wheels ship their OWN libamdhip64.so / libhsa-runtime64.so (torch/lib, found by RPATH) and ask for them by
    file name, while this library asks for the SONAME libamdhip64.so.7.  If this library is loaded first it pulls in the
    ROCm installation's runtime, a later `import torch` then brings a second HIP + HSA runtime into the process, and the
    second one finds no GPU ("No HIP GPUs are available").  Loading torch's copy first -- without importing torch -- makes
    both resolve to that one runtime, whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    _preload_torch_hip()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python vectordb-from-scratch_amd/build.py "
            "or __graft_entry__.build()); this engine has no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, sz, u64 = c.c_void_p, c.c_size_t, c.c_uint64
    fp, u64p, szp, u32p = c.POINTER(c.c_float), c.POINTER(c.c_uint64), c.POINTER(c.c_size_t), c.POINTER(c.c_uint32)
    L.vdb_flat_create.argtypes = [c.c_int, c.c_int, c.POINTER(vp)]
    L.vdb_flat_destroy.argtypes = [vp]
    L.vdb_flat_destroy.restype = None
    L.vdb_flat_create_sharded.argtypes = [c.c_int, c.POINTER(c.c_int), sz, c.POINTER(vp)]
    L.vdb_flat_shards.argtypes = [vp]
    L.vdb_flat_shards.restype = sz
    L.vdb_flat_shard_len.argtypes = [vp, sz]
    L.vdb_flat_shard_len.restype = sz
    L.vdb_flat_set_exchange.argtypes = [vp, c.c_int]
    L.vdb_flat_shard_stats.argtypes = [vp, u64p]
    L.vdb_flat_add.argtypes = [vp, u64, fp, sz]
    L.vdb_flat_add_bulk.argtypes = [vp, u64p, u64, fp, sz, sz]
    L.vdb_flat_add_bulk_device.argtypes = [vp, u64p, u64, vp, sz, sz]
    L.vdb_flat_load_vector_file.argtypes = [vp, c.c_char_p, u64, szp]
    L.vdb_flat_remove.argtypes = [vp, u64]
    L.vdb_flat_get_vector.argtypes = [vp, u64, fp, sz, szp]
    L.vdb_flat_len.argtypes = [vp]
    L.vdb_flat_len.restype = sz
    L.vdb_flat_metric.argtypes = [vp]
    L.vdb_flat_dim.argtypes = [vp]
    L.vdb_flat_dim.restype = sz
    L.vdb_flat_reserve.argtypes = [vp, sz, sz]
    L.vdb_flat_flush.argtypes = [vp]
    L.vdb_flat_search.argtypes = [vp, fp, sz, sz, u64p, fp, szp]
    L.vdb_flat_search_batch.argtypes = [vp, fp, sz, sz, szp, sz, u64p, sz, sz, u64p, fp, szp]
    L.vdb_flat_search_batch_device.argtypes = [vp, vp, sz, sz, sz, vp, sz, vp, vp, vp, vp]
    L.vdb_flat_search_batch_device_begin.argtypes = [vp, vp, sz, sz, sz, vp, sz, vp, vp, vp, vp, vp]
    L.vdb_flat_search_batch_device_finish.argtypes = [vp, c.POINTER(c.c_int)]
    L.vdb_flat_search_batch_device_submit.argtypes = [vp, vp, sz, sz, sz, vp, sz, vp, vp, vp, vp, c.POINTER(c.c_int)]
    L.vdb_flat_search_batch_device_wait.argtypes = [vp, c.c_int]
    L.vdb_merge_topk_device.argtypes = [c.c_int, vp, vp, vp, sz, sz, sz, vp, vp, vp, vp]
    L.vdb_flat_distances_batch.argtypes = [vp, fp, sz, sz, szp, u64p, fp]
    L.vdb_merge_topk_packed_device.argtypes = [c.c_int, vp, sz, sz, sz, sz, vp, vp, vp, vp, vp]
    L.vdb_flat_last_stats.argtypes = [vp, u64p]
    L.vdb_flat_last_stats_ex.argtypes = [vp, u64p, sz]
    L.vdb_flat_set_screen.argtypes = [vp, c.c_int]
    L.vdb_flat_set_wide.argtypes = [vp, c.c_int]
    L.vdb_flat_set_tiers.argtypes = [vp, c.c_uint]
    L.vdb_flat_set_shadow.argtypes = [vp, c.c_int]
    L.vdb_flat_set_sample_cache.argtypes = [vp, c.c_int]
    L.vdb_flat_debug_screen_scores.argtypes = [vp, fp, sz, sz, c.c_int, fp, fp, c.POINTER(c.c_double)]
    L.vdb_flat_debug_rows.argtypes = [vp]
    L.vdb_flat_debug_last_thresholds.argtypes = [vp, fp, sz]
    L.vdb_flat_debug_rows.restype = sz
    L.vdb_flat_debug_row_info.argtypes = [vp, fp, sz]
    L.vdb_flat_debug_cert_probe.argtypes = [vp, u32p, fp, fp, sz, u32p]
    L.vdb_flat_set_profile.argtypes = [vp, c.c_int]
    L.vdb_last_error.argtypes = [c.c_char_p, sz, szp, szp]
    L.vdb_last_error.restype = None
    L.vdb_hnsw_create.argtypes = [c.c_int, sz, sz, sz, u64, c.c_int, c.POINTER(vp)]
    L.vdb_hnsw_destroy.argtypes = [vp]
    L.vdb_hnsw_destroy.restype = None
    L.vdb_hnsw_add.argtypes = [vp, u64, fp, sz, c.c_long]
    L.vdb_hnsw_add_bulk.argtypes = [vp, u64p, u64, fp, sz, sz]
    L.vdb_hnsw_remove.argtypes = [vp, u64]
    L.vdb_hnsw_search_batch.argtypes = [vp, fp, sz, sz, sz, sz, u64p, fp, szp]
    L.vdb_hnsw_len.argtypes = [vp]
    L.vdb_hnsw_len.restype = sz
    L.vdb_hnsw_metric.argtypes = [vp]
    L.vdb_hnsw_get_vector.argtypes = [vp, u64, fp, sz, szp]
    L.vdb_hnsw_neighbors.argtypes = [vp, u64, sz, u64p, sz]
    L.vdb_hnsw_neighbors.restype = c.c_long
    L.vdb_hnsw_node_level.argtypes = [vp, u64]
    L.vdb_hnsw_node_level.restype = c.c_long
    L.vdb_hnsw_entry_point.argtypes = [vp, u64p, szp]
    L.vdb_hnsw_stats.argtypes = [vp, u64p]
    L.vdb_hnsw_set_traversal.argtypes = [vp, c.c_int, sz]
    L.vdb_hnsw_set_build.argtypes = [vp, c.c_int]
    L.vdb_hnsw_build_stats.argtypes = [vp, u64p]
    L.vdb_hnsw_build_times.argtypes = [vp, c.POINTER(c.c_double)]
    L.vdb_shard_unique_id.argtypes = [c.c_char_p]
    L.vdb_shard_group_create.argtypes = [c.c_char_p, c.c_int, c.c_int, c.c_int, c.POINTER(vp)]
    L.vdb_shard_group_destroy.argtypes = [vp]
    L.vdb_shard_group_destroy.restype = None
    L.vdb_shard_group_rank.argtypes = [vp]
    L.vdb_shard_group_world.argtypes = [vp]
    L.vdb_shard_range.argtypes = [sz, c.c_int, c.c_int, szp, szp]
    L.vdb_shard_range.restype = None
    L.vdb_flat_search_batch_sharded.argtypes = [vp, vp, vp, sz, sz, sz, vp, sz, vp, vp, vp, vp]
    L.vdb_shard_group_last_stats.argtypes = [vp, u64p]
    L.vdb_abi_version.restype = c.c_int
    L.vdb_build_arch.restype = c.c_char_p
    _lib = L
    return L


def last_error():
    buf = ctypes.create_string_buffer(512)
    e, a = ctypes.c_size_t(), ctypes.c_size_t()
    lib().vdb_last_error(buf, 512, ctypes.byref(e), ctypes.byref(a))
    return buf.value.decode("utf-8", "replace"), e.value, a.value
