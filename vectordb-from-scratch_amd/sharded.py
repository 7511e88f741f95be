"""Row-sharded FlatIndex across the GPUs of one node (BASELINE config 3).

Two layers:

* `ShardGroup` -- the product path: a thin ctypes wrapper of include/vdb_shard.h.  The exchange (RCCL all-gather of the
  packed partial top-k, merge kernel, status reduction) lives behind the C ABI and calls RCCL directly; nothing of it runs
  in Python.  One process per GPU; rank 0 makes the unique id and hands it over by any side channel
  (`ShardGroup.from_torch_distributed` uses a torch.distributed broadcast, which is all torch does here).

* `ShardedSearcher` -- a mirror of the SAME call pattern over torch.distributed, so that the collective discipline can
  be tested with world-size-2 `gloo` process groups on CPU (tests/test_sharded_cpu.py): every rank performs the same
  number of collectives whatever happens locally; exchange 2 runs on all ranks iff exchange 1's reduced status says some
  rank was pending (decided from gathered data, never from local state); a rank whose local search failed -- in either
  half -- keeps sending zeroed results and its error code.  (the previous version skipped exchange 2 on a
  rank whose begin() raised and deadlocked the others.)

Rank r owns a contiguous block of database rows as an ordinary `GpuFlatIndex` with GLOBAL ids.  The payload is
B*k*12 bytes per rank: latency-bound.  A zero-norm row on any shard fails the batch on every rank, like the reference's
single loop (flat_index.rs:57-60).
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _ffi
from .error import DimensionMismatch, InvalidVector, NanDistance, IndexError_, VectorDbError
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


# ------------------------------------------------------------------------------------------ merges (test references)
def merge_topk_torch(ids, dists, counts, k):
    """Reference implementation of the exchange merge with torch ops (CPU or GPU tensors):
    ids/dists [W, B, k], counts [W, B] -> global top-k ascending by (distance, id)."""
    W, B, kk = ids.shape
    valid = torch.arange(kk, device=ids.device).view(1, 1, kk) < counts.view(W, B, 1)
    d = torch.where(valid, dists, torch.full_like(dists, float("inf"))).permute(1, 0, 2).reshape(B, W * kk)
    i = torch.where(valid, ids, torch.full_like(ids, torch.iinfo(torch.int64).max)).permute(1, 0, 2).reshape(B, W * kk)
    # stable two-key sort: by id first, then (stable) by distance
    o1 = torch.sort(i, dim=1, stable=True).indices
    d1, i1 = torch.gather(d, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.sort(d1, dim=1, stable=True).indices
    d2, i2 = torch.gather(d1, 1, o2), torch.gather(i1, 1, o2)
    total = counts.sum(0).clamp(max=k).to(torch.int32)
    return i2[:, :k].contiguous(), d2[:, :k].contiguous(), total


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


# ------------------------------------------------------------------------------------------ call-pattern mirror (tests)
_ERR_CODE = {DimensionMismatch: 1, InvalidVector: 2, NanDistance: 3}
_ERR_CLASS = {1: IndexError_, 2: InvalidVector, 3: NanDistance}
PENDING_HOST = 100          # VDB_PENDING_HOST
CODE_ERR_BASE = 1000        # vdb_shard.cpp: status word of a failed rank = 1000 + vdb_status (survives the MAX with 100)


class ShardedSearcher:
    """The call pattern of vdb_flat_search_batch_sharded (csrc/vdb_shard.cpp) over torch.distributed.

    local_search(queries [B, d] tensor, k[, outs]) -> (ids int64 [B, k], dists f32 [B, k], counts int32 [B]) on the
    shard this rank owns.  Optional attributes `begin(queries, k, outs, code_view)` / `finish()` give the two-half
    form (`gpu_local_search` below).  `collectives` counts the all-gathers of the last search.
    """

    def __init__(self, local_search, rank=0, world=1, group=None, merge=None):
        self.local_search, self.rank, self.world, self.group = local_search, rank, world, group
        self.merge = merge
        self._pack = self._gath = None
        self.collectives = 0

    def _buffers(self, B, k, device):
        """One int32 buffer per rank: ids (two words each) | distances (bit pattern) | counts | status word."""
        words = B * (3 * k + 1) + 1
        if self._pack is None or self._pack.numel() != words + (words & 1) or self._pack.device != device:
            self._pack = torch.zeros((words + (words & 1),), dtype=torch.int32, device=device)
            self._gath = torch.empty((self.world * self._pack.numel(),), dtype=torch.int32, device=device)
        pk = self._pack
        ids = pk[:2 * B * k].view(torch.int64).view(B, k)
        dists = pk[2 * B * k:3 * B * k].view(torch.float32).view(B, k)
        counts = pk[3 * B * k:3 * B * k + B]
        return pk, ids, dists, counts, words

    def _exchange(self, pk, B, k, ids):
        """ONE all-gather of the packed per-rank buffers + merge; returns (out, worst status): one host sync."""
        dist.all_gather_into_tensor(self._gath, pk, group=self.group)
        self.collectives += 1
        g = self._gath.view(self.world, pk.numel())
        words = B * (3 * k + 1) + 1
        if ids.is_cuda and self.merge is None:
            out_i = torch.empty((B, k), dtype=torch.int64, device=ids.device)
            out_d = torch.empty((B, k), dtype=torch.float32, device=ids.device)
            out_c = torch.empty((B + 1,), dtype=torch.int32, device=ids.device)      # [B] = worst status
            rc = _ffi.lib().vdb_merge_topk_packed_device(
                ids.device.index or 0, ctypes.c_void_p(self._gath.data_ptr()), self.world, pk.numel(), B, k,
                ctypes.c_void_p(out_i.data_ptr()), ctypes.c_void_p(out_d.data_ptr()), ctypes.c_void_p(out_c.data_ptr()),
                ctypes.c_void_p(out_c.data_ptr() + 4 * B), ctypes.c_void_p(torch.cuda.current_stream(ids.device).cuda_stream))
            if rc:
                raise IndexError_(_ffi.last_error()[0])
            return (out_i, out_d, out_c[:B]), int(out_c[B].item())
        g_ids = g[:, :2 * B * k].contiguous().view(torch.int64).view(self.world, B, k)
        g_d = g[:, 2 * B * k:3 * B * k].contiguous().view(torch.float32).view(self.world, B, k)
        g_cnt = g[:, 3 * B * k:3 * B * k + B].contiguous()
        merge = self.merge or merge_topk_torch
        return merge(g_ids, g_d, g_cnt, k), int(g[:, words - 1].max().item())

    def search_batch(self, queries, k):
        B = queries.shape[0]
        self.collectives = 0
        if self.world == 1:
            return self.local_search(queries, k)
        pk, ids, dists, counts, words = self._buffers(B, k, queries.device)
        outs = (ids, dists, counts)
        err, code = None, 0

        def failed(e):                         # zeroed results + 1000 + code: this rank stays in step with the others
            nonlocal err, code
            err, code = e, CODE_ERR_BASE + _ERR_CODE.get(type(e), 4)
            pk.zero_()
            pk[words - 1] = code

        # ---- first half of the local search
        begin = getattr(self.local_search, "begin", None)
        begun = False
        try:
            if begin is not None:
                begin(queries, k, outs, pk[words - 1:words])                 # the local search writes 0 / PENDING_HOST itself
                begun = True
            else:
                res = self.local_search(queries, k, outs)
                if res[0].data_ptr() != ids.data_ptr():                      # a local search that ignores `outs`
                    ids.copy_(res[0]); dists.copy_(res[1]); counts.copy_(res[2].to(torch.int32))
                pk[words - 1] = 0
        except VectorDbError as e:
            failed(e)
        # ---- exchange 1: ALWAYS, on every rank
        try:
            out, worst = self._exchange(pk, B, k, ids)
        except BaseException:
            if begun:                                                         # never leave the handle locked
                try:
                    self.local_search.finish()
                except VectorDbError:
                    pass
            raise
        # ---- second half of the local search
        if begun:
            try:
                self.local_search.finish()
            except VectorDbError as e:
                failed(e)
        # ---- exchange 2: on ALL ranks iff the REDUCED status of exchange 1 says some rank was pending
        if worst == PENDING_HOST:
            if err is None:
                pk[words - 1] = 0
            out, worst = self._exchange(pk, B, k, ids)
        if err is not None:                    # this rank's own error (its message) -- also when nobody else saw one
            raise err
        if worst == 0:
            return out
        rc = worst - CODE_ERR_BASE if worst >= CODE_ERR_BASE else 4
        raise _ERR_CLASS.get(rc, IndexError_)(f"a shard on another rank failed the batch (status {rc})")


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
