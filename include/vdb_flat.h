/*
 * vdb_flat.h -- C ABI of the MI355X-native brute-force kNN engine.
 *
 * This is the drop-in boundary for the reference's FlatIndex hot path
 * (Ricoledan/vectordb-from-scratch).  The reference has no FFI layer; its seam
 * is the Rust trait `Index` (src/index.rs:11-35) implemented by `FlatIndex`
 * (src/flat_index.rs:37-74) and consumed by `VectorStore<I: Index>`
 * (src/storage.rs:83, :116-127).  Every entry point below names the reference
 * interface it replaces.  The Rust binding a maintainer would add is in
 * INTEGRATION.md; a C++ mirror of the trait lives in
 * vectordb-from-scratch_amd/host/.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++ / torch types cross this boundary;
 *  - every function returns a vdb_status (0 = ok); no exception or panic
 *    crosses the boundary; vdb_last_error() gives the message and, for a
 *    dimension mismatch, the expected/actual pair (src/error.rs:12-13);
 *  - inputs are borrowed for the duration of the call; outputs are
 *    caller-allocated;
 *  - ids are the reference's `usize` internal ids, passed as uint64_t;
 *  - vectors are row-major little-endian f32, the byte layout of
 *    src/persistence/mmap.rs:77-84;
 *  - search entry points may be called concurrently from several threads on
 *    one handle (the server holds RwLock::read() around search,
 *    src/server/routes.rs:244,:342); add/remove are externally serialised by
 *    the caller's write lock (routes.rs:141,:210,:300).
 */
#ifndef VDB_FLAT_H
#define VDB_FLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/distance.rs:9-16  enum DistanceMetric */
typedef enum vdb_metric {
    VDB_METRIC_EUCLIDEAN = 0, /* sqrt(sum (a-b)^2)              distance.rs:37-44 */
    VDB_METRIC_COSINE = 1,    /* 1 - clamp(dot/(|a||b|), -1, 1) distance.rs:47-64 */
    VDB_METRIC_DOT = 2        /* -dot(a, b)                     distance.rs:31,:67-73 */
} vdb_metric;

/* src/error.rs:10-31  enum VectorDbError, as integer codes */
typedef enum vdb_status {
    VDB_OK = 0,
    VDB_ERR_DIMENSION_MISMATCH = 1, /* error.rs:12  DimensionMismatch{expected,actual} */
    VDB_ERR_INVALID_VECTOR = 2,     /* error.rs:18  InvalidVector (zero norm under Cosine, distance.rs:51-55) */
    VDB_ERR_NAN = 3,                /* the reference panics on a NaN distance (flat_index.rs:62); a panic cannot cross a C ABI */
    VDB_ERR_DEVICE = 4,             /* HIP runtime failure -> error.rs:30 IndexError(String) */
    VDB_ERR_INVALID_ARGUMENT = 5,   /* null pointer, bad metric, ... -> IndexError(String) */
    VDB_ERR_NOT_FOUND = 6           /* get_vector on an absent id (index.rs:23 returns None) */
} vdb_status;

typedef struct vdb_flat_index vdb_flat_index; /* opaque; owns all device and staging memory */

/* FlatIndex::new(metric)  src/flat_index.rs:19-25.  `device` is the HIP device ordinal. */
int vdb_flat_create(int metric, int device, vdb_flat_index **out);
void vdb_flat_destroy(vdb_flat_index *h);

/*
 * ONE index over several GPUs of one node, in ONE process -- the form the reference's seam needs: a server process holds a
 * single `VectorStore<I: Index>` behind one RwLock (src/storage.rs:83,:116-127, src/server/mod.rs:13-16), so the object that
 * implements `Index` must itself own the row shards (BASELINE.json north_star: "the index shards by database row across the
 * 8 GPUs of one node, each GPU producing a partial top-k that is merged after an RCCL all-gather over xGMI").
 *
 * The handle is an ordinary vdb_flat_index: every call below takes it.  Rows are dealt to the `n_devices` shards (one per
 * entry of `devices`; bulk loads in contiguous blocks, vdb_shard_range in vdb_shard.h; single adds to the shard that already
 * holds the id, else to the emptiest).  A batched search runs the local pipeline of every shard concurrently (one host
 * worker thread and one stream per shard), exchanges the packed partial top-k (ids | distances | counts | status,
 * nq*k*12 bytes per shard) and merges by (distance, id) on devices[0], where queries and outputs live.  Results are
 * identical to a single-GPU index over the same rows, bit for bit; errors keep the reference's semantics (a zero-norm row
 * on ANY shard fails the batch, src/flat_index.rs:57-60).
 *
 * Exchange (vdb_flat_set_exchange): VDB_EXCHANGE_RCCL -- in-process RCCL communicators (ncclCommInitAll), one grouped
 * ncclAllGather per batch; the default whenever the listed devices are distinct.  VDB_EXCHANGE_PEER -- each shard's stream
 * copies its packed block straight into devices[0]'s gather buffer (hipMemcpyPeerAsync over xGMI); the default when a device
 * is listed more than once (several shards on one GPU: how the multi-shard logic is tested on a one-GPU box -- RCCL refuses
 * two ranks on one device).
 *
 * Not available on a sharded handle: the two-half / ticket forms of the device search (begin/finish, submit/wait),
 * vdb_flat_distances_batch, the vdb_flat_debug_* probes and vdb_flat_search_batch_sharded (that one is the
 * process-per-GPU form of the same exchange, vdb_shard.h); they return VDB_ERR_INVALID_ARGUMENT.
 */
int vdb_flat_create_sharded(int metric, const int *devices, size_t n_devices, vdb_flat_index **out);
/* number of shards of the handle (1 for a plain vdb_flat_create handle) */
size_t vdb_flat_shards(const vdb_flat_index *h);
/* live rows of shard `shard` (sharded handle), or vdb_flat_len for a plain handle with shard == 0 */
size_t vdb_flat_shard_len(const vdb_flat_index *h, size_t shard);
enum { VDB_EXCHANGE_RCCL = 0, VDB_EXCHANGE_PEER = 1 };
int vdb_flat_set_exchange(vdb_flat_index *h, int mode);
/* Counters of the last search on a sharded handle: [0] exchanges performed (1, or 2 when a shard needed its host after the
 * first tier), [1] shards, [2] exchange mode used, [3] ranks RCCL reports for the communicators (0 in peer mode),
 * [4] host clock of the call (ns), [5] host clock until every shard's first tier was enqueued (ns). */
int vdb_flat_shard_stats(const vdb_flat_index *h, uint64_t out[8]);

/* Index::add(id, vector)  src/index.rs:13, src/flat_index.rs:38-41.
 * Copies the row.  An existing id is overwritten silently (HashMap::insert).
 * Like the reference there is NO dimension check at add: a row whose dimension
 * differs from the rows already stored is kept (host side) and makes later
 * searches fail with DimensionMismatch, exactly as distance.rs:21-26 does. */
int vdb_flat_add(vdb_flat_index *h, uint64_t id, const float *vector, size_t dim);

/* N adds in one call (what 1M calls of Index::add amount to, storage.rs:135-172).
 * rows is [n][dim] contiguous; ids[i] belongs to row i (ids == NULL: first_id + i). */
int vdb_flat_add_bulk(vdb_flat_index *h, const uint64_t *ids, uint64_t first_id, const float *rows,
                      size_t n, size_t dim);
/* Same, with `d_rows` already resident in this device's HBM (device-to-device copy). */
int vdb_flat_add_bulk_device(vdb_flat_index *h, const uint64_t *ids, uint64_t first_id,
                             const float *d_rows, size_t n, size_t dim);

/* Bulk-load the reference's mmap vector file (src/persistence/mmap.rs:13-15 header `[dim u32 LE][count u32 LE]`,
 * :77-84 body = row-major little-endian f32): the file is memory-mapped and handed to the device in
 * large chunks instead of `count` add() calls.  The file has no id column: row i gets id first_id + i.
 * *out_count receives the number of rows loaded. */
int vdb_flat_load_vector_file(vdb_flat_index *h, const char *path, uint64_t first_id, size_t *out_count);

/* Index::remove(id)  src/index.rs:16, src/flat_index.rs:43-46.  Absent id is VDB_OK. */
int vdb_flat_remove(vdb_flat_index *h, uint64_t id);

/* Index::get_vector(id)  src/index.rs:23.  Copies up to cap floats into out, sets *dim. */
int vdb_flat_get_vector(vdb_flat_index *h, uint64_t id, float *out, size_t cap, size_t *dim);

/* Index::len / Index::metric  src/index.rs:26-29.  vdb_flat_dim: dimension of the stored rows (0 if empty). */
size_t vdb_flat_len(const vdb_flat_index *h);
int vdb_flat_metric(const vdb_flat_index *h);
size_t vdb_flat_dim(const vdb_flat_index *h);

/* Pre-size the device row store (optional; avoids regrowth copies during a bulk build). */
int vdb_flat_reserve(vdb_flat_index *h, size_t rows, size_t dim);

/* Push staged adds/removes to the device now (otherwise done lazily by the next search). */
int vdb_flat_flush(vdb_flat_index *h);

/* Index::search(query, k)  src/index.rs:20, src/flat_index.rs:52-65.
 * out_ids/out_dists hold k entries; *out_count = min(k, len).  Ascending by
 * (distance, id): the reference's tie order is HashMap-random. */
int vdb_flat_search(vdb_flat_index *h, const float *query, size_t dim, size_t k, uint64_t *out_ids,
                    float *out_dists, size_t *out_count);

/*
 * The batched hot call.  Replaces the sequential loop of
 * VectorStore::search_batch (src/storage.rs:302-310) over FlatIndex::search,
 * with the per-query k of `&[(Vector, usize)]` (storage.rs:304).
 *   queries   [nq][dim] row-major f32 (host memory)
 *   ks        per-query k, or NULL to use `k` for every query
 *   id_mask   optional pre-filter: bit i (LSB-first in 64-bit words) set = id i
 *             eligible; ids >= mask_bits are not eligible; NULL = no filter.
 *             (BASELINE config 4; the reference post-filters a 3x over-fetch,
 *             storage.rs:249-290, whose result is a prefix of this one.)
 *   out_ids / out_dists   [nq][kstride], caller-allocated, kstride >= max k
 *   out_counts            [nq]: min(k_b, eligible rows)
 * The first failing query fails the whole batch (storage.rs:309).
 */
int vdb_flat_search_batch(vdb_flat_index *h, const float *queries, size_t nq, size_t dim,
                          const size_t *ks, size_t k, const uint64_t *id_mask, size_t mask_bits,
                          size_t kstride, uint64_t *out_ids, float *out_dists, size_t *out_counts);

/*
 * Same call with queries and outputs resident in this device's HBM and one k
 * for the whole batch; `stream` is a hipStream_t (NULL = the handle's own
 * stream).  d_id_mask is a device pointer (or NULL).  The call returns after
 * the results are complete on `stream` (it synchronises that stream once to
 * read the status word).  d_out_counts is uint32_t[nq].
 */
int vdb_flat_search_batch_device(vdb_flat_index *h, const float *d_queries, size_t nq, size_t dim,
                                 size_t k, const uint64_t *d_id_mask, size_t mask_bits,
                                 uint64_t *d_out_ids, float *d_out_dists, uint32_t *d_out_counts,
                                 void *stream);

/*
 * DistanceMetric::distance (src/distance.rs:20-33) for explicit (query, stored id) pairs, batched: the
 * candidate-list distance evaluations of an HNSW search (src/hnsw/graph.rs:155, :182 -- BASELINE config 5),
 * computed on the GPU in the reference's operation order (bit-identical to the CPU).  Query b is paired with
 * ids[offsets[b] .. offsets[b+1]); out_dists has offsets[nq] entries; an id that is not stored gives NaN.
 * Zero norm on either side under Cosine -> VDB_ERR_INVALID_VECTOR (distance.rs:51-55).
 */
int vdb_flat_distances_batch(vdb_flat_index *h, const float *queries, size_t nq, size_t dim,
                             const size_t *offsets, const uint64_t *ids, float *out_dists);

/*
 * Multi-GPU exchange step: merge `nparts` partial top-k lists per query
 * (gathered from the row shards with an RCCL all-gather) into the global
 * top-k, ascending by (distance, id).  All pointers are device pointers on
 * `device`; parts are laid out [nparts][nq][k] with counts [nparts][nq].
 */
int vdb_merge_topk_device(int device, const uint64_t *d_part_ids, const float *d_part_dists,
                          const uint32_t *d_part_counts, size_t nparts, size_t nq, size_t k,
                          uint64_t *d_out_ids, float *d_out_dists, uint32_t *d_out_counts,
                          void *stream);

/* The device-resident search in two halves, for callers that have more work to enqueue behind it (the multi-GPU
 * exchange of sharded.py): _begin runs every check and enqueues the FIRST tier on `stream` without synchronising the
 * host; *d_code (device word, may be NULL) receives 0 when every query was certified by that tier and no error status
 * was raised, else VDB_PENDING_HOST.  _finish waits for the stream, runs the fallback tiers for the queries that need
 * them (*changed = 1 when outputs were rewritten) and reports the errors a plain vdb_flat_search_batch_device call
 * would.  The handle stays locked between the two calls, which must come from the same thread; every _begin that
 * returned VDB_OK must be followed by one _finish. */
#define VDB_PENDING_HOST 100
int vdb_flat_search_batch_device_begin(vdb_flat_index *h, const float *d_queries, size_t nq, size_t dim, size_t k,
                                       const uint64_t *d_id_mask, size_t mask_bits, uint64_t *d_out_ids,
                                       float *d_out_dists, uint32_t *d_out_counts, int32_t *d_code, void *stream);
int vdb_flat_search_batch_device_finish(vdb_flat_index *h, int *changed);

/*
 * The device-resident search ASYNCHRONOUSLY, two batches in flight per handle (no reference counterpart: the reference's
 * batch loop, src/storage.rs:302-310, is synchronous; this is what a server that keeps the GPU busy calls instead).
 * _submit runs every check and enqueues the first tier on `stream` (NULL = a stream of the context it picked) and
 * returns a ticket WITHOUT waiting; _wait(ticket) waits for that search, runs its fallback tiers where needed and reports
 * the errors a plain vdb_flat_search_batch_device call would.  While batch i's HBM-bound pass runs, batch i+1's query
 * preparation and the host's turnaround are hidden behind it.  Rules: at most two tickets outstanding per handle (a third
 * submit fails); every ticket must be waited for exactly once; outputs are complete when _wait returns; add / remove /
 * flush and the host-pointer entry points are refused while a ticket is outstanding (the caller's read lock spans
 * submit .. wait, routes.rs:244,:342); a synchronous vdb_flat_search_batch_device call may run beside ONE outstanding ticket.
 */
int vdb_flat_search_batch_device_submit(vdb_flat_index *h, const float *d_queries, size_t nq, size_t dim, size_t k,
                                        const uint64_t *d_id_mask, size_t mask_bits, uint64_t *d_out_ids,
                                        float *d_out_dists, uint32_t *d_out_counts, void *stream, int *ticket);
int vdb_flat_search_batch_device_wait(vdb_flat_index *h, int ticket);

/* The same merge reading the all-gathered exchange buffer in place.  Each part is `words_per_part` int32
 * words (even): ids int64[nq*k] | dists f32[nq*k] | counts i32[nq] | status i32 | pad.  *d_out_status (may be
 * NULL) receives the maximum status word over the parts, so one host read tells whether any shard failed. */
int vdb_merge_topk_packed_device(int device, const int32_t *d_packed, size_t nparts, size_t words_per_part,
                                 size_t nq, size_t k, uint64_t *d_out_ids, float *d_out_dists,
                                 uint32_t *d_out_counts, uint32_t *d_out_status, void *stream);

/* Measurement hook for bench.py: when on, every search brackets its fused MFMA kernel launch
 * with HIP events on the launch stream and vdb_flat_last_stats()[7] reports the kernel's
 * duration in nanoseconds (summed over the launches of that search). */
int vdb_flat_set_profile(vdb_flat_index *h, int on);

/* Counters of the last search on this handle (diagnostics, tests, bench):
 *  [0] queries answered by the MFMA path   [1] queries re-done by the exact-scan fallback
 *  [2] candidate-pool overflows            [3] rows scanned by the fused kernel
 *  [4] sample rows used for the thresholds [5] k' (candidates kept per query)
 *  [6] uncertified queries                 [7] fused-kernel time of the search, ns (profiling on) */
int vdb_flat_last_stats(const vdb_flat_index *h, uint64_t out[8]);

/* The same counters followed by (n up to 16, the rest reads 0):
 *  [8] 1 when the bf16 screening tier answered the batch first, 0 when only the f32 MFMA tier ran
 *  [9] queries the screening tier could not certify and handed to the f32 MFMA tier
 *  [10] [11] [12] host clock of the call, ns: first tier enqueued / its flags on the host / return
 *  [13] queries answered by the re-threshold pass (a second screening pass whose thresholds are the score cuts that the
 *       k-th exact distances of the first pass imply; every key under the cut is re-ranked)
 *  [14] 1 when the screening pass read the bf16 shadow rows (vdb_flat_set_shadow)
 *  [15] 1 when the library is the DIAGNOSTICS build (-DVDB_DIAG) and one of its environment knobs is set: such a run
 *       is not covered by the exactness guarantee.  Always 0 in the release library, which reads no environment.
 * With the screening tier on, [4] [5] [7] describe ITS sample, k' and kernel time. */
int vdb_flat_last_stats_ex(const vdb_flat_index *h, uint64_t *out, size_t n);

/* Tier selection for indexes above 16384 rows (no reference counterpart; results are identical either way):
 *  1 (default): a first pass ranks every row on the bf16 matrix cores (HBM-bound), keeps k' candidates per
 *     query, re-ranks them with the reference's exact f32 arithmetic (distance.rs:37-73) and certifies the
 *     result with a rigorous bound on the bf16 rounding error; uncertified queries go to the f32 tier;
 *  0: the f32-input MFMA tier only (v_mfma_f32_32x32x2_f32, arithmetic-bound), then the exact scan.
 */
int vdb_flat_set_screen(vdb_flat_index *h, int mode);

/* Batches above 256 queries (VectorStore::search_batch takes any number, storage.rs:302-310; BASELINE config 3: 1024):
 *  1 (default): the screening pass serves 512 queries per fetch of the rows (128 rows x 512 queries per workgroup), so a batch
 *     of B queries reads the database ceil(B / 512) times;
 *  0: 256 queries per fetch (the kernel of batches up to 256), ceil(B / 256) reads.
 * No reference counterpart; results are identical either way, bit for bit. */
int vdb_flat_set_wide(vdb_flat_index *h, int on);

/*
 * Opt-in bf16 SHADOW of the rows for the screening pass (no reference counterpart; results are identical with and without
 * it).  on = 1: the index keeps, next to the f32 rows, their bf16 roundings (+50 % device memory: 2 bytes per element on top
 * of 4) -- exactly the values the screening kernel otherwise produces in registers -- and the filter pass streams THOSE:
 * half the HBM bytes per batch (csrc/kernels_fused_s16.hip).  The exact re-rank, the f32 tier and the exact scan keep
 * reading the f32 rows, so the returned ids and distances do not change by a bit.  Used when the padded row length is a
 * multiple of 64 elements (otherwise the f32-row pass runs as before).  Existing rows are converted by this call, later
 * adds at their flush.  on = 0 frees the shadow.  last_stats_ex()[14] = 1 when the last search used it.
 */
int vdb_flat_set_shadow(vdb_flat_index *h, int on);

/*
 * The screening tier's SAMPLE CACHE (on by default; no reference counterpart; results identical either way).  The tier
 * derives its per-query filter thresholds from the scores of S <= 65536 sample rows spread over the index.  With the cache
 * the index keeps a compact bf16 copy of exactly those rows (S * padded dimension * 2 bytes: 100 MB beside a 3 GB index;
 * S <= max(16384, rows / 8), so never more than an eighth of the f32 store),
 * rebuilt by the first search after rows were added, and the sample pass streams it instead of gathering the rows from
 * the f32 store: half the bytes, contiguous.  Same roundings, same MFMA order -> the same thresholds, bit for bit.
 * Used when the padded row length is a multiple of 64 elements.  on = 0 frees the copy and restores the f32 gather.
 */
int vdb_flat_set_sample_cache(vdb_flat_index *h, int on);

/* Test hook (no reference counterpart; results are identical whatever the flags): force the hand-over of queries to
 * the slower tiers so that every tier can be compared with every other on the same index.  Not read from the
 * environment -- the release library has no getenv on any path. */
#define VDB_TIERS_NO_RETHRESHOLD 1u /* skip the re-threshold pass: uncertified queries go straight to the f32 MFMA tier */
#define VDB_TIERS_FORCE_F32 2u      /* hand EVERY query the screening tier answered to the f32 MFMA tier as well */
#define VDB_TIERS_FORCE_EXACT 4u    /* hand every query to the exact scan */
#define VDB_TIERS_NO_DIRECT 8u      /* small indexes (<= 16384 rows), batches of <= 8 queries: the tiered pipeline instead of the direct exact scan */
int vdb_flat_set_tiers(vdb_flat_index *h, unsigned flags);

/*
 * Diagnostics of the screening tier's CERTIFICATE (tests/test_gpu_certificate.py).  The default tier ranks rows by
 * bf16-MFMA scores and returns exact f32 distances only because a bound on the score error proves that no excluded row
 * can enter the top k.  These entry points expose the inputs of that proof so that it can be tested
 * directly instead of end to end.  They do not alter any search.
 *
 * vdb_flat_debug_screen_scores: runs query preparation and the PRODUCTION filter kernel over every live row with
 * thresholds that let everything pass, and returns the ranking score of every (query, row) pair:
 *   out_scores [nq][rows uploaded] f32 (the NaN pattern 0xffffffff = no key: tombstoned row); nq <= 256;
 *   raw = 0: the scores the tier ranks by (Dot / Euclid: lower-bound scores, score - g_q * margin_row);
 *   raw = 1: the plain scores fma(acc, alpha, beta) (under Dot: -acc, the raw MFMA accumulator);
 *   raw = 2: the f32 MFMA TIER's scores (v_mfma_f32_32x32x2_f32; dense_scores_kernel over every row, bit-identical to the
 *            fused f32 kernel's); a following cert_probe then evaluates the f32 tier's certificate (eps_coef only);
 *   out_qinfo [nq][4]: exact-order |q|, the bound on |q - bf16(q)|, g_q, 0;
 *   out_consts [8]: eps_coef, c_acc, kappa, max |d|, max |d - bf16(d)|, max |d - bf16(d)|/|d|, 1 if lower-bound scores, ld.
 * vdb_flat_debug_rows: number of device rows.  vdb_flat_debug_row_info: out [rows][4] = exact-order |d|, alpha, beta, margin (0 under Cosine).
 * vdb_flat_debug_cert_probe: out[i] = the production certification test (kernels_aux.hip cert_test) for prepared query
 *   qi[i] of the last debug_screen_scores call, unexamined-row score bound T[i] and k-th exact distance ek[i].
 *   Soundness means: probe(T = score of row r, ek = exact distance of row r) is 0 for EVERY pair.
 */
int vdb_flat_debug_screen_scores(vdb_flat_index *h, const float *queries, size_t nq, size_t dim, int raw,
                                 float *out_scores, float *out_qinfo, double *out_consts);
size_t vdb_flat_debug_rows(const vdb_flat_index *h); /* device rows incl. tombstoned ones (row i = i-th row ever added since the last reset) */
int vdb_flat_debug_row_info(vdb_flat_index *h, float *out, size_t n_rows);
/* the per-query filter thresholds of the screening tier as the LAST search on this handle derived them from its sample pass (first nq queries) */
int vdb_flat_debug_last_thresholds(vdb_flat_index *h, float *out, size_t nq);
int vdb_flat_debug_cert_probe(vdb_flat_index *h, const uint32_t *qi, const float *T, const float *ek, size_t n,
                              uint32_t *out);

/* Thread-local message of the last failing call on this thread, plus the
 * DimensionMismatch pair (error.rs:12-13).  Any pointer may be NULL. */
void vdb_last_error(char *buf, size_t cap, size_t *expected, size_t *actual);

/* Library/ABI version and the GPU architecture the kernels were built for ("gfx950"). */
int vdb_abi_version(void);
const char *vdb_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* VDB_FLAT_H */
