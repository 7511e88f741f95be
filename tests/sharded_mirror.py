"""Test scaffolding: a torch.distributed restatement of the call pattern of vdb_flat_search_batch_sharded
(vectordb-from-scratch_amd/csrc/vdb_shard.cpp), so that its collective discipline can be driven with world-size-2 `gloo`
process groups on CPU (tests/test_sharded_cpu.py):

  * every rank performs the same number of collectives whatever happens locally;
  * growing the exchange buffers -- a rank-LOCAL allocation -- is agreed on through one fixed-size all-gather: either every
    rank grew, or every rank drops its buffers and fails the call (no rank is left waiting in exchange 1);
  * exchange 2 runs on all ranks iff exchange 1's reduced status says some rank was pending (decided from gathered data,
    never from local state);
  * a rank whose local search failed -- in either half -- keeps sending zeroed results and its error code.

This is NOT the product path (that is C++ behind the C ABI and needs RCCL and one GPU per rank); it lives here so that no
test scaffolding ships in the package.  `merge_topk_torch` is the reference implementation of the exchange merge.
"""
import ctypes

import torch
import torch.distributed as dist

from conftest import load_package

load_package()
from vectordb_from_scratch_amd import _ffi                                   # noqa: E402
from vectordb_from_scratch_amd.error import (DimensionMismatch, IndexError_, InvalidVector, NanDistance,  # noqa: E402
                                             VectorDbError)


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

    def __init__(self, local_search, rank=0, world=1, group=None, merge=None, alloc_fails=None):
        self.local_search, self.rank, self.world, self.group = local_search, rank, world, group
        self.merge = merge
        self._pack = self._gath = None
        self.collectives = 0
        self.votes = 0                          # growth votes of the last search (vdb_shard.cpp ensure_buffers)
        self.alloc_fails = alloc_fails          # test hook: callable(words) -> True when THIS rank's allocation is to fail

    def _buffers(self, B, k, device):
        """One int32 buffer per rank: ids (two words each) | distances (bit pattern) | counts | status word.
        Growth is a rank-local allocation: it is agreed on through ONE fixed-size all-gather (a word per rank); if any
        rank failed, every rank drops its buffers and the call fails everywhere -- nobody is left in exchange 1."""
        words = B * (3 * k + 1) + 1
        if self._pack is None or self._pack.numel() != words + (words & 1) or self._pack.device != device:
            self._pack = self._gath = None
            failed_here = bool(self.alloc_fails and self.alloc_fails(words))
            if not failed_here:
                self._pack = torch.zeros((words + (words & 1),), dtype=torch.int32, device=device)
                self._gath = torch.empty((self.world * self._pack.numel(),), dtype=torch.int32, device=device)
            vote = torch.tensor([1 if failed_here else 0], dtype=torch.int32, device=device)
            votes = torch.empty((self.world,), dtype=torch.int32, device=device)
            dist.all_gather_into_tensor(votes, vote, group=self.group)
            self.votes += 1
            bad = [i for i, v in enumerate(votes.tolist()) if v]
            if bad:
                self._pack = self._gath = None
                raise IndexError_(f"rank {bad[0]} could not allocate the exchange buffers; no rank searched")
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
        self.votes = 0
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


