"""Row-sharded FlatIndex across the GPUs of one node.

One process per GPU.  Rank r owns a contiguous block of database rows as an ordinary
`GpuFlatIndex`; ids stay global.  A batched search is
    local search (fused MFMA kernel, exact re-rank)  ->  ONE all-gather of the partial top-k
    (ids, distances, counts) over RCCL/xGMI  ->  merge of world*k candidates per query,
and every rank ends with the global top-k.  The payload is B*k*12 bytes per rank, so the
exchange is latency-bound.  Error status is reduced (MAX) so that a zero-norm row on any shard
fails the whole batch on every rank, like the reference's single loop (flat_index.rs:57-60).

torch is used for device memory, streams and torch.distributed only.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _ffi
from .error import DimensionMismatch, InvalidVector, NanDistance, IndexError_, VectorDbError


def shard_range(n_rows, rank, world):
    """Contiguous row block of `rank`: [lo, hi).  Blocks differ by at most one row."""
    base, rem = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


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


_ERR_CODE = {DimensionMismatch: 1, InvalidVector: 2, NanDistance: 3}


class ShardedSearcher:
    """Drives one batched search over `world` row shards.

    local_search(queries [B, d] tensor, k) -> (ids int64 [B, k], dists f32 [B, k], counts int32 [B])
    on the shard this rank owns (normally `gpu_local_search(index)` below).
    """

    def __init__(self, local_search, rank=0, world=1, group=None, merge=None):
        self.local_search, self.rank, self.world, self.group = local_search, rank, world, group
        self.merge = merge
        self._pack = self._gath = None

    def _buffers(self, B, k, device):
        """One int32 buffer per rank holds ids (two words each), distances (bit pattern), counts and the
        status word; the local search writes its outputs straight into it, so the exchange is ONE
        all-gather with no packing copies."""
        words = B * (3 * k + 1) + 1
        if self._pack is None or self._pack.numel() != words + (words & 1) or self._pack.device != device:
            self._pack = torch.zeros((words + (words & 1),), dtype=torch.int32, device=device)
            self._gath = torch.empty((self.world * self._pack.numel(),), dtype=torch.int32, device=device)
        pk = self._pack
        ids = pk[:2 * B * k].view(torch.int64).view(B, k)
        dists = pk[2 * B * k:3 * B * k].view(torch.float32).view(B, k)
        counts = pk[3 * B * k:3 * B * k + B]
        return pk, ids, dists, counts, words

    PENDING_HOST = 100                     # VDB_PENDING_HOST: some rank's first tier left queries for the host to finish

    def _exchange(self, pk, B, k, ids):
        """ONE all-gather of the packed per-rank buffers + merge; returns (out, worst status) -- the only host sync."""
        dist.all_gather_into_tensor(self._gath, pk, group=self.group)
        g = self._gath.view(self.world, pk.numel())
        words = B * (3 * k + 1) + 1
        if ids.is_cuda and self.merge is None:
            # merge straight out of the gathered buffer; the kernel also reduces the status words
            out_i = torch.empty((B, k), dtype=torch.int64, device=ids.device)
            out_d = torch.empty((B, k), dtype=torch.float32, device=ids.device)
            out_c = torch.empty((B + 1,), dtype=torch.int32, device=ids.device)      # [B] = worst status
            rc = _ffi.lib().vdb_merge_topk_packed_device(
                ids.device.index or 0, ctypes.c_void_p(self._gath.data_ptr()), self.world, pk.numel(), B, k,
                ctypes.c_void_p(out_i.data_ptr()), ctypes.c_void_p(out_d.data_ptr()), ctypes.c_void_p(out_c.data_ptr()),
                ctypes.c_void_p(out_c.data_ptr() + 4 * B), ctypes.c_void_p(torch.cuda.current_stream(ids.device).cuda_stream))
            if rc:
                raise IndexError_(_ffi.last_error()[0])
            return (out_i, out_d, out_c[:B]), int(out_c[B].item())                   # the ONE host sync of the exchange
        g_ids = g[:, :2 * B * k].contiguous().view(torch.int64).view(self.world, B, k)
        g_d = g[:, 2 * B * k:3 * B * k].contiguous().view(torch.float32).view(self.world, B, k)
        g_cnt = g[:, 3 * B * k:3 * B * k + B].contiguous()
        merge = self.merge or merge_topk_torch
        return merge(g_ids, g_d, g_cnt, k), int(g[:, words - 1].max().item())

    def search_batch(self, queries, k):
        code = 0
        err = None
        B = queries.shape[0]
        if self.world == 1:
            return self.local_search(queries, k)
        pk, ids, dists, counts, words = self._buffers(B, k, queries.device)
        outs = (ids, dists, counts)
        # Two-half local search when the index offers it: the first tier is only ENQUEUED, its "needs the host" word lands
        # in the packed buffer on the device, and the exchange is enqueued right behind it -- one host sync per batch.
        begin = getattr(self.local_search, "begin", None)
        pending = False
        try:
            if begin is not None and queries.is_cuda:
                begin(queries, k, outs, pk[words - 1:words].data_ptr())
                pending = True
            else:
                res = self.local_search(queries, k, outs)
                if res[0].data_ptr() != ids.data_ptr():      # a local search that ignores `outs`
                    ids.copy_(res[0]); dists.copy_(res[1]); counts.copy_(res[2].to(torch.int32))
        except VectorDbError as e:            # keep the collective call pattern identical on every rank
            err, code = e, _ERR_CODE.get(type(e), 4)
            ids.zero_(); dists.zero_(); counts.zero_()
        if not pending:
            pk[words - 1] = code
        try:
            out, worst = self._exchange(pk, B, k, ids)
        except BaseException:
            if pending:                                       # never leave the handle locked behind a failed collective
                try:
                    self.local_search.finish()
                except VectorDbError:
                    pass
            raise
        if pending:
            try:
                self.local_search.finish()                    # fallback tiers for this rank's uncertified queries, errors
            except VectorDbError as e:
                err, code = e, _ERR_CODE.get(type(e), 4)
                ids.zero_(); dists.zero_(); counts.zero_()
            if worst == self.PENDING_HOST:                    # some rank rewrote its partial results: exchange again
                pk[words - 1] = code
                out, worst = self._exchange(pk, B, k, ids)
            elif err is not None:
                worst = max(worst, code)
        if worst:
            if err:
                raise err
            raise {1: IndexError_, 2: InvalidVector, 3: NanDistance}.get(worst, IndexError_)(
                "a shard on another rank failed the batch")
        return out


def gpu_local_search(index, mask_ptr=0, mask_bits=0, reuse_outputs=False):
    """local_search callable over a GpuFlatIndex with everything resident in HBM.  reuse_outputs=True
    returns the same three output tensors on every call (overwritten by the next search)."""
    cache = {}

    def _stream_of(dev):
        # the caller's current stream, through the raw accessor (torch.cuda.current_stream builds a Stream object: ~10 us)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return int(raw(dev.index if dev.index is not None else torch.cuda.current_device()))
        return torch.cuda.current_stream(dev).cuda_stream

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
                                  stream=_stream_of(dev), mask_ptr=mask_ptr, mask_bits=mask_bits)
        return ids, dists, counts

    def begin(queries, k, outs, code_ptr):
        B, d = queries.shape
        ids, dists, counts = outs
        index.search_batch_device_begin(queries.data_ptr(), B, d, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(),
                                        code_ptr=code_ptr, stream=_stream_of(queries.device),
                                        mask_ptr=mask_ptr, mask_bits=mask_bits)

    run.begin = begin
    run.finish = index.search_batch_device_finish
    return run
