/*
 * vdb_hnsw.h -- C ABI of the HNSW index with GPU-offloaded distance evaluation (BASELINE config 5:
 * "HnswIndex 1M x 768, m=16 ef_search=200 -- GPU offload of candidate-list distance evaluations only").
 *
 * What it replaces in the reference: HnswIndex / HnswGraph (src/hnsw/mod.rs:14-82, src/hnsw/graph.rs:75-425)
 * behind the same `Index` trait (src/index.rs:11-35).  The host side of this library owns the graph (node levels,
 * per-layer neighbour lists, entry point) and runs the reference's traversal (graph.rs:143-199 search_layer,
 * :244-342 insert, :345-381 remove, :386-412 search_knn) operation by operation, including the array layout of
 * Rust's BinaryHeap, which decides the order of equal-distance results (neighbor_queue.rs:102-106).  Every distance the
 * traversal asks for (graph.rs:155, :182, :224) is evaluated on the MI355X, in the reference's exact f32 operation
 * order (distance.rs:37-73).  Searches run DEVICE-RESIDENT by default (kernels_hnsw.hip): the graph is mirrored in HBM
 * and one workgroup per query walks it, with the reference's priority queues in LDS -- one launch per batch; what does
 * not fit that kernel (m > 19, ef > 1022, a walk that overflows its LDS structures, vdb_hnsw_set_traversal) is traversed on
 * the host with the candidate lists of ALL in-flight queries evaluated in one launch per traversal round.  Both give
 * the reference's results; there is no CPU distance path.
 *
 * Inserts are sequential by definition and make no GPU round trip per neighbour expansion.  Bulk inserts evaluate FRONTIERS
 * ONLY: per block of 64 inserts one launch of the device-resident walk (insert mode) evaluates what every insert's
 * search_layer asks for and records it -- one block AHEAD of the host, on the graph as of the end of the block before the one
 * being replayed; the host replays the inserts in order with the reference's algorithm, reading distances from the records and
 * asking the GPU again only where the real walk left the speculative one (vdb_hnsw_set_build, vdb_hnsw_build_stats).  Single adds and small batches scan the stored rows ahead of
 * the walk instead.  Every edge keeps the distance it was created with, so a prune (graph.rs:207-241) recomputes nothing.
 *
 * Not reproducible in the reference and fixed here: node levels come from StdRng::from_entropy() (graph.rs:101);
 * this index draws them from a seeded splitmix64 stream (`seed`), with the reference's formula (graph.rs:118-123).
 * Given the same seed and insertion order the graph and every search result are identical to the CPU restatement
 * (oracle/hnsw_oracle.c) -- ids, order and the bit patterns of the distances.
 *
 * Errors: the status codes of vdb_flat.h; messages through vdb_last_error().
 */
#ifndef VDB_HNSW_H
#define VDB_HNSW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vdb_hnsw_index vdb_hnsw_index;

/* HnswIndex::with_params(metric, HnswParams::new(m, ef_construction, ef_search))  (mod.rs:28-32, graph.rs:49-59);
 * m_max0 = 2m, ml = 1/ln(m), max_layers = 16.  HnswIndex::new is m = 16, ef_construction = 200, ef_search = 50. */
int vdb_hnsw_create(int metric, size_t m, size_t ef_construction, size_t ef_search, uint64_t seed, int device,
                    vdb_hnsw_index **out);
void vdb_hnsw_destroy(vdb_hnsw_index *h);

/* Index::add -> HnswGraph::insert (mod.rs:57-59, graph.rs:244-342).  `level` < 0: drawn from the seeded stream. */
int vdb_hnsw_add(vdb_hnsw_index *h, uint64_t id, const float *v, size_t dim, long level);
/* HnswIndex::build_batch (mod.rs:37-42): sequential inserts of rows [n][dim]; ids NULL: first_id + i. */
int vdb_hnsw_add_bulk(vdb_hnsw_index *h, const uint64_t *ids, uint64_t first_id, const float *rows, size_t n, size_t dim);
/* Index::remove -> HnswGraph::remove (graph.rs:345-381): absent id is ok. */
int vdb_hnsw_remove(vdb_hnsw_index *h, uint64_t id);

/* search_knn(query, k, ef) for a batch (graph.rs:386-412).  ef = 0: the index's ef_search; note that Index::search
 * of the reference always passes ef = 50 (mod.rs:71) and search_with_ef the caller's value (mod.rs:45-53).
 * Outputs [nq][k]; out_counts[b] <= k results, ascending by distance. */
int vdb_hnsw_search_batch(vdb_hnsw_index *h, const float *queries, size_t nq, size_t dim, size_t k, size_t ef,
                          uint64_t *out_ids, float *out_dists, size_t *out_counts);

/* Test hook (results are identical either way): host_only = 1 sends every search through the host traversal instead of
 * the device-resident walk; host_threads > 0 fixes its worker-thread count (0 = automatic).  Not read from the environment. */
int vdb_hnsw_set_traversal(vdb_hnsw_index *h, int host_only, size_t host_threads);

size_t vdb_hnsw_len(const vdb_hnsw_index *h);                              /* graph.rs:109-111 */
int vdb_hnsw_metric(const vdb_hnsw_index *h);
/* Index::get_vector (mod.rs:65-67): copies min(cap, dim) floats; VDB_ERR_NOT_FOUND for an absent id. */
int vdb_hnsw_get_vector(vdb_hnsw_index *h, uint64_t id, float *out, size_t cap, size_t *dim);

/* Graph inspection (tests: the graph must equal the CPU restatement's).  Neighbours of `id` at `layer`: returns
 * the count (copies at most cap ids), or -1 when the node or the layer does not exist. */
long vdb_hnsw_neighbors(const vdb_hnsw_index *h, uint64_t id, size_t layer, uint64_t *out, size_t cap);
long vdb_hnsw_node_level(const vdb_hnsw_index *h, uint64_t id);
int vdb_hnsw_entry_point(const vdb_hnsw_index *h, uint64_t *id, size_t *max_level);     /* returns 0 when empty */

/* Counters since creation: [0] distances evaluated on the GPU by the host-driven paths (inserts, host traversal),
 * [1] GPU launches (device searches, traversal rounds, row scans and prune batches), [2] traversal rounds of the last
 * HOST-traversed search_batch, [3] its distances, [4] queries answered by the device-resident search, [5] queries whose
 * device walk overflowed its LDS structures and that the host traversal re-ran. */
int vdb_hnsw_stats(const vdb_hnsw_index *h, uint64_t out[6]);

/* How bulk inserts (vdb_hnsw_add_bulk, 32 vectors or more) get their distances.  1 (default): FRONTIER ONLY -- a device walk per
 * insert evaluates what search_layer asks for (graph.rs:155, :182) on the graph as it was 64 + (its position in its block of
 * 64) inserts earlier, while the host replays the block before; the host replays the inserts in order with the reference's
 * algorithm and asks the GPU again only where the real walk left the speculative one.  0: the row-scan build (every stored row against every new vector, N^2 / 2 distances).  Same graph either way. */
int vdb_hnsw_set_build(vdb_hnsw_index *h, int frontier_only);
/* Counters of the builds since creation: [0] inserts of the frontier-only build, [1] distances its device walks evaluated,
 * [2] distances between vectors of one block and of the block before it, [3] distances evaluated after a miss (the real walk left the speculative one),
 * [4] such round trips, [5] walks whose record overflowed, [6] distances the host's inserts consumed (what the reference's
 * algorithm evaluates), [7] inserts of the row-scan build. */
int vdb_hnsw_build_stats(const vdb_hnsw_index *h, uint64_t out[8]);
/* Where the frontier-only builds spent their wall time, seconds since creation: [0] bringing the device mirror of the graph up to
 * date (once per block), [1] waiting for the block's device walks (they run a block ahead) and its window distances, [2] the host's replay of the inserts
 * (the reference's algorithm, distances read from the records), of which [3] were round trips for missed distances. */
int vdb_hnsw_build_times(const vdb_hnsw_index *h, double out[4]);

#ifdef __cplusplus
}
#endif
#endif /* VDB_HNSW_H */
