"""Row-sharded FlatIndex across the GPUs of one node (BASELINE config 3).

Two forms of the same exchange, both behind the C ABI:

* one process, several GPUs: `GpuFlatIndex(metric, devices=[...])` (vdb_flat_create_sharded, index.py) -- ONE index object
  that owns its shards; nothing of it lives in this file;

* one process per GPU: `ShardGroup` below, a thin ctypes wrapper of include/vdb_shard.h.  The exchange (RCCL all-gather of
  the packed partial top-k, merge kernel, status reduction) calls RCCL directly; nothing of it runs in Python.  Rank 0
  makes the unique id and hands it over by any side channel (`ShardGroup.from_torch_distributed` uses a torch.distributed
  broadcast, which is all torch does here).

(The torch.distributed restatement of the call pattern that the world-2 `gloo` tests drive lives with the tests:
tests/sharded_mirror.py.)

Rank r owns a contiguous block of database rows as an ordinary `GpuFlatIndex` with GLOBAL ids.  The payload is
B*k*12 bytes per rank: latency-bound.  A zero-norm row on any shard fails the batch on every rank, like the reference's
single loop (flat_index.rs:57-60).
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _ffi
from .error import IndexError_
from .index import _raise


def shard_range(n_rows, rank, world):
    """Contiguous row block of `rank`: [lo, hi).  Blocks differ by at most one row (vdb_shard_range)."""
    base, rem = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ------------------------------------------------------------------------------------------ the product path (C ABI)
class ShardGroup:
    """One rank of a sharded index: vdb_shard_group (an RCCL communicator + exchange buffers behind the C ABI)."""

    def __init__(self, unique_id, rank, world, device=0):
        self._L = _ffi.lib()
        self._h = ctypes.c_void_p()
        self.rank, self.device = int(rank), int(device)
        rc = self._L.vdb_shard_group_create(unique_id, int(rank), int(world), int(device), ctypes.byref(self._h))
        if rc:
            _raise(rc)

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        rc = _ffi.lib().vdb_shard_unique_id(buf)
        if rc:
            _raise(rc)
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, device, group=None):
        """Rank 0 creates the id, a torch.distributed broadcast hands it to the other ranks, every rank joins."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], rank, world, device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.vdb_shard_group_destroy(h)

    def world(self):
        """The rank count RCCL reports for the communicator."""
        return int(self._L.vdb_shard_group_world(self._h))

    def search_batch_device(self, index, q_ptr, nq, dim, k, out_ids_ptr, out_dists_ptr, out_counts_ptr, stream=0,
                            mask_ptr=0, mask_bits=0):
        """vdb_flat_search_batch_sharded: COLLECTIVE; raw device pointers; every rank ends with the global top-k."""
        rc = self._L.vdb_flat_search_batch_sharded(
            self._h, index._h, ctypes.c_void_p(q_ptr), int(nq), int(dim), int(k), ctypes.c_void_p(mask_ptr or None),
            int(mask_bits), ctypes.c_void_p(out_ids_ptr), ctypes.c_void_p(out_dists_ptr), ctypes.c_void_p(out_counts_ptr),
            ctypes.c_void_p(stream or None))
        if rc:
            _raise(rc)

    def last_stats(self):
        out = (ctypes.c_uint64 * 4)()
        self._L.vdb_shard_group_last_stats(self._h, out)
        return dict(zip(["collectives", "ranks", "local_pending", "host_ns"], [int(v) for v in out]))


def group_search(group, index, mask_ptr=0, mask_bits=0):
    """search(queries [B, d] CUDA tensor, k) -> (ids int64 [B, k], dists f32 [B, k], counts int32 [B]) through the C ABI
    group; the three output tensors are reused from call to call."""
    cache = {}

    def run(queries, k):
        B, d = queries.shape
        key = (B, k, queries.device)
        if key not in cache:
            cache[key] = (torch.empty((B, k), dtype=torch.int64, device=queries.device),
                          torch.empty((B, k), dtype=torch.float32, device=queries.device),
                          torch.empty((B,), dtype=torch.int32, device=queries.device))
        ids, dists, counts = cache[key]
        group.search_batch_device(index, queries.data_ptr(), B, d, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(),
                                  stream=_raw_stream(queries.device), mask_ptr=mask_ptr, mask_bits=mask_bits)
        return ids, dists, counts

    return run


def _raw_stream(dev):
    # the caller's current stream through the raw accessor (torch.cuda.current_stream builds a Stream object: ~10 us)
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return int(raw(dev.index if dev.index is not None else torch.cuda.current_device()))
    return torch.cuda.current_stream(dev).cuda_stream


# ------------------------------------------------------------------------------------------ the merge kernel alone
def merge_topk_hip(ids, dists, counts, k, stream=None):
    """The same merge on the GPU through the C ABI (vdb_merge_topk_device)."""
    W, B, kk = ids.shape
    assert kk == k and ids.is_cuda and ids.dtype == torch.int64 and dists.dtype == torch.float32
    counts = counts.to(torch.int32).contiguous()
    out_i = torch.empty((B, k), dtype=torch.int64, device=ids.device)
    out_d = torch.empty((B, k), dtype=torch.float32, device=ids.device)
    out_c = torch.empty((B,), dtype=torch.int32, device=ids.device)
    st = torch.cuda.current_stream(ids.device).cuda_stream if stream is None else stream
    rc = _ffi.lib().vdb_merge_topk_device(ids.device.index or 0, ctypes.c_void_p(ids.data_ptr()),
                                          ctypes.c_void_p(dists.data_ptr()), ctypes.c_void_p(counts.data_ptr()),
                                          W, B, k, ctypes.c_void_p(out_i.data_ptr()), ctypes.c_void_p(out_d.data_ptr()),
                                          ctypes.c_void_p(out_c.data_ptr()), ctypes.c_void_p(st))
    if rc:
        raise IndexError_(_ffi.last_error()[0])
    return out_i, out_d, out_c


def torch_group_search(index, mask_ptr=0, mask_bits=0, group=None):
    """The same exchange with torch.distributed as the transport: every rank searches its shard, three all-gathers (RCCL under
    the "nccl" backend) bring the partial top-k lists together, vdb_merge_topk_device merges them on the GPU.  bench.py falls
    back to this when the C-ABI shard group (include/vdb_shard.h: RCCL called from the library itself) cannot be created on
    every rank -- it costs three collectives and a blocking local search per batch instead of one collective, and says so in
    the line.  A rank whose local search fails still takes part in the collectives and raises afterwards; the others raise too."""
    import torch.distributed as dist
    local = gpu_local_search(index, mask_ptr=mask_ptr, mask_bits=mask_bits, reuse_outputs=True)
    cache = {}

    def run(queries, k):
        B = queries.shape[0]
        dev = queries.device
        W = dist.get_world_size(group)
        key = (B, k, dev)
        if key not in cache:
            cache[key] = (torch.empty((W, B, k), dtype=torch.int64, device=dev), torch.empty((W, B, k), dtype=torch.float32, device=dev),
                          torch.empty((W, B + 1), dtype=torch.int32, device=dev), torch.zeros((B + 1,), dtype=torch.int32, device=dev),
                          torch.zeros((B, k), dtype=torch.int64, device=dev), torch.zeros((B, k), dtype=torch.float32, device=dev))
        g_ids, g_d, g_c, mine_c, zero_i, zero_d = cache[key]
        err = None
        try:
            ids, dists, counts = local(queries, k)
            mine_c[:B] = counts
            mine_c[B] = 0
        except Exception as e:                                   # noqa: BLE001 -- re-raised below, after the collectives
            err, ids, dists = e, zero_i, zero_d
            mine_c.zero_()
            mine_c[B] = 1
        dist.all_gather_into_tensor(g_ids, ids.contiguous(), group=group)
        dist.all_gather_into_tensor(g_d, dists.contiguous(), group=group)
        dist.all_gather_into_tensor(g_c, mine_c, group=group)
        if err is not None:
            raise err
        if bool(g_c[:, B].any()):
            raise IndexError_("a shard on another rank failed the batch")
        return merge_topk_hip(g_ids, g_d, g_c[:, :B].contiguous(), k)
    return run


def gpu_local_search(index, mask_ptr=0, mask_bits=0, reuse_outputs=False):
    """local_search callable over a GpuFlatIndex with everything resident in HBM.  reuse_outputs=True
    returns the same three output tensors on every call (overwritten by the next search)."""
    cache = {}

    def run(queries, k, outs=None):
        B, d = queries.shape
        dev = queries.device
        key = (B, k, dev)
        if outs is not None:                    # caller-provided buffers (the packed exchange buffer)
            ids, dists, counts = outs
        elif reuse_outputs and key in cache:
            ids, dists, counts = cache[key]
        else:
            ids = torch.empty((B, k), dtype=torch.int64, device=dev)
            dists = torch.empty((B, k), dtype=torch.float32, device=dev)
            counts = torch.empty((B,), dtype=torch.int32, device=dev)
            if reuse_outputs:
                cache[key] = (ids, dists, counts)
        index.search_batch_device(queries.data_ptr(), B, d, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(),
                                  stream=_raw_stream(dev), mask_ptr=mask_ptr, mask_bits=mask_bits)
        return ids, dists, counts

    def begin(queries, k, outs, code_view):
        B, d = queries.shape
        ids, dists, counts = outs
        index.search_batch_device_begin(queries.data_ptr(), B, d, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(),
                                        code_ptr=code_view.data_ptr(), stream=_raw_stream(queries.device),
                                        mask_ptr=mask_ptr, mask_bits=mask_bits)

    run.begin = begin
    run.finish = index.search_batch_device_finish
    return run
