/*
 * vdb_shard.h -- C ABI of the ROW-SHARDED FlatIndex search across the GPUs of one node (SURVEY.md 8(e),
 * BASELINE.json configs[2]: "FlatIndex 10M x 768 f32, dot product, batch=1024, k=100, index row-sharded across
 * 8 x MI355X with RCCL top-k merge").
 *
 * What it replaces in the reference: nothing is distributed there -- VectorStore::search_batch
 * (src/storage.rs:302-310) loops over ONE FlatIndex (src/flat_index.rs:52-65).  Rows are independent, so the
 * index shards by contiguous row blocks, one process (or thread) per GPU, each holding an ordinary
 * vdb_flat_index over its block with GLOBAL ids.  A batched search is
 *     local search on every rank (queries replicated)
 *  -> ONE all-gather of the packed partial results  ids u64[nq*k] | dists f32[nq*k] | counts u32[nq] | status
 *     (nq*k*12 bytes per rank: latency-bound, xGMI bandwidth does not enter)  -- RCCL, called directly from this
 *     library (librccl is dlopen'ed; nothing here goes through Python or torch.distributed)
 *  -> merge of world*k candidates per query by (distance, id) on every rank (exact distances are bit-identical
 *     across shards, so the merge is deterministic and every rank ends with the same global top-k).
 * Error semantics of the single loop are kept: a zero-norm row on ANY shard fails the batch on EVERY rank
 * (src/distance.rs:51-55, src/flat_index.rs:57-60) -- the status word travels in the same gather.
 *
 * Collective discipline: every rank performs the same number of collectives per call whatever happens locally.
 * The first tier of the local search is only enqueued; a device word tells whether any rank still needs its host
 * (uncertified queries, errors).  Exchange 1 always runs.  Exchange 2 runs on ALL ranks iff the reduced status of
 * exchange 1 says some rank rewrote its partial results -- a decision taken from the gathered data, identical
 * everywhere, never from local state (a rank whose local search failed still sends zeroed results and its error
 * code in both exchanges).  One host synchronisation per exchange.
 *
 * The Rust binding a maintainer would add is in INTEGRATION.md section 8.
 */
#ifndef VDB_SHARD_H
#define VDB_SHARD_H

#include <stddef.h>
#include <stdint.h>

#include "vdb_flat.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vdb_shard_group vdb_shard_group; /* one per rank: an RCCL communicator + exchange buffers */

#define VDB_SHARD_UNIQUE_ID_BYTES 128

/* ncclGetUniqueId: rank 0 creates the id; the caller hands the 128 bytes to the other ranks over its own side
 * channel (the Rust host: its control plane; bench.py: a torch.distributed broadcast; tests: a file). */
int vdb_shard_unique_id(unsigned char out[VDB_SHARD_UNIQUE_ID_BYTES]);

/* ncclCommInitRank on `device`: COLLECTIVE -- returns when all `world` ranks have called it with the same id.
 * world == 1 with id == NULL needs no RCCL: the group degenerates to the plain local search.  world == 1 WITH an id
 * builds a single-rank communicator and runs the full exchange path (how the path is tested on a one-GPU box). */
int vdb_shard_group_create(const unsigned char id[VDB_SHARD_UNIQUE_ID_BYTES], int rank, int world, int device,
                           vdb_shard_group **out);
void vdb_shard_group_destroy(vdb_shard_group *g);
int vdb_shard_group_rank(const vdb_shard_group *g);
/* the rank count RCCL reports for the communicator (ncclCommCount), not the number it was asked for */
int vdb_shard_group_world(const vdb_shard_group *g);

/* Contiguous row block [*lo, *hi) of `rank` when n_rows rows are dealt to `world` ranks (blocks differ by at most one row). */
void vdb_shard_range(size_t n_rows, int rank, int world, size_t *lo, size_t *hi);

/*
 * The sharded hot call.  COLLECTIVE: every rank calls it with the same queries, nq, dim and k and with the
 * vdb_flat_index of ITS row block (created on the group's device).  Arguments as vdb_flat_search_batch_device;
 * on return every rank holds the GLOBAL top-k: d_out_ids / d_out_dists [nq][k], d_out_counts [nq] = min(k, live
 * rows of all shards passing the mask).  nq * k * world <= ... see vdb_merge_topk_packed_device (world * k <= 2048).
 * `stream`: a hipStream_t of the group's device (NULL = the index's own stream).
 * Returns the worst status over all ranks; VDB_ERR_* raised by another rank's shard is reported here too.
 */
int vdb_flat_search_batch_sharded(vdb_shard_group *g, vdb_flat_index *local, const float *d_queries, size_t nq,
                                  size_t dim, size_t k, const uint64_t *d_id_mask, size_t mask_bits,
                                  uint64_t *d_out_ids, float *d_out_dists, uint32_t *d_out_counts, void *stream);

/* Counters of the last sharded search on this group: [0] collectives performed (1 or 2; 0 when world == 1),
 * [1] ranks in the communicator, [2] 1 when this rank's local search needed its host after the first tier,
 * [3] host clock of the call, ns. */
int vdb_shard_group_last_stats(const vdb_shard_group *g, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* VDB_SHARD_H */
