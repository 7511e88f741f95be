// vdb_hnsw.cpp -- host side of include/vdb_hnsw.h: the reference's HNSW graph (src/hnsw/graph.rs) kept on the
// host, every distance evaluated on the GPU (vdb_internal.h hooks of the device row store).
//
// Search: the reference walks one query at a time (graph.rs:386-412).  Here every query of a batch is a resumable
// traversal; a ROUND advances each of them to its next neighbour expansion (graph.rs:166-193), collects the
// unvisited neighbours of all of them as (query, row) pairs, evaluates the whole list in one launch, and feeds the
// distances back in list order.  Each query sees exactly the sequence of heap operations the reference performs, so
// results are identical to a CPU run on the same graph; only the waiting is shared.
// Insert: inserts are sequential by definition (each walks the graph the previous one left), and a GPU round trip per
// neighbour expansion (graph.rs:182: at most 32 distances) costs ~40 us -- 200 expansions would be 8 ms per insert.  So
// the distances are produced AHEAD of the walks, amortised over a chunk of inserts: one pass over the stored rows gives
// the exact distances of 16 new vectors at once (scan_rows_kernel; 128 vectors per chunk, the next chunk's scan and its
// transfer run while the host walks the current one), and the walks read them from pinned host memory -- no GPU round
// trip per insert at all.  Prune distances (graph.rs:207-241) are never recomputed: every edge keeps the distance it was
// created with (d(a,b) and d(b,a) are the same bits under all three metrics).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/vdb_flat.h"
#include "../../include/vdb_hnsw.h"
#include "vdb_internal.h"
#include "kernels.h"

namespace {

struct Nb { float d; uint64_t id; };                            // neighbor_queue.rs:7-11

// neighbor_queue.rs:37-43: distance.partial_cmp (unordered -> Equal), then id
inline int nb_cmp(const Nb& a, const Nb& b) {
    if (a.d < b.d) return -1;
    if (a.d > b.d) return 1;
    return a.id < b.id ? -1 : (a.id > b.id ? 1 : 0);
}

// Rust's std BinaryHeap (max-heap under SIGN * nb_cmp; SIGN = -1 is BinaryHeap<Reversed>, neighbor_queue.rs:47-60),
// with the standard library's element moves, so that the backing array -- what into_vec() returns -- is the same.
template <int SIGN> struct RustHeap {
    std::vector<Nb> v;
    static bool le(const Nb& a, const Nb& b) { return SIGN * nb_cmp(a, b) <= 0; }
    void sift_up(size_t start, size_t pos) {
        Nb e = v[pos];
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (le(e, v[parent])) break;
            v[pos] = v[parent];
            pos = parent;
        }
        v[pos] = e;
    }
    void push(Nb n) { v.push_back(n); sift_up(0, v.size() - 1); }
    bool pop(Nb& out) {
        if (v.empty()) return false;
        Nb item = v.back();
        v.pop_back();
        if (!v.empty()) {
            std::swap(item, v[0]);
            const size_t end = v.size();
            size_t pos = 0, child = 1;
            Nb e = v[0];
            while (end >= 2 && child <= end - 2) {                // sift_down_to_bottom
                if (le(v[child], v[child + 1])) ++child;
                v[pos] = v[child];
                pos = child;
                child = 2 * pos + 1;
            }
            if (child == end - 1) { v[pos] = v[child]; pos = child; }
            v[pos] = e;
            sift_up(0, pos);
        }
        out = item;
        return true;
    }
    size_t size() const { return v.size(); }
    bool empty() const { return v.empty(); }
    const Nb& top() const { return v[0]; }
    void clear() { v.clear(); }
};

// membership-only id set (graph.rs:151 HashSet<usize>): open addressing, grows by doubling.  A slot holds (epoch << 32 | id) --
// node ids are below 2^32 - 16 -- and only slots of the current epoch count as occupied, so clear() is one increment: a walk at
// 1M nodes visits ~6000 ids on each of up to 6 layers, and refilling (or re-growing) the table per layer showed in the build.
struct IdSet {
    std::vector<uint64_t> t;
    size_t used = 0;
    uint32_t epoch = 0;
    void clear() {
        if (t.empty()) t.assign(1024, 0ull);
        if (++epoch == 0) { std::fill(t.begin(), t.end(), 0ull); epoch = 1; }
        used = 0;
    }
    static size_t h(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; return (size_t)x; }
    bool insert(uint64_t id) {                                    // true when newly inserted
        if (t.empty()) clear();
        if (2 * (used + 1) > t.size()) {
            std::vector<uint64_t> old; old.swap(t);
            t.assign(old.size() * 2, 0ull); used = 0;
            const uint32_t e = epoch;
            for (uint64_t x : old) if ((uint32_t)(x >> 32) == e) insert(x & 0xffffffffull);
        }
        const uint64_t key = ((uint64_t)epoch << 32) | (id & 0xffffffffull);
        size_t m = t.size() - 1, i = h(id) & m;
        while ((uint32_t)(t[i] >> 32) == epoch) { if (t[i] == key) return false; i = (i + 1) & m; }
        t[i] = key; ++used;
        return true;
    }
};

struct Node {                                                     // graph.rs:63-72
    bool present = false;
    uint32_t level = 0, row = 0xffffffffu;
    std::vector<std::vector<uint64_t>> nbr;
    std::vector<std::vector<float>> nbr_d;                        // nbr_d[l][i] = distance(this node, nbr[l][i]) as evaluated when the edge was made
};

constexpr float F32_MAX = 3.40282347e+38f;
inline bool is_zero_norm_mark(float d) { uint32_t b; memcpy(&b, &d, 4); return b == vdb_internal::ZERO_NORM_MARK; }

}  // namespace

struct vdb_hnsw_index {
    vdb_flat_index* flat = nullptr;
    int metric = 0;
    size_t m = 16, m_max0 = 32, ef_construction = 200, ef_search = 50, max_layers = 16;
    double ml = 0;
    uint64_t rng = 0;
    std::vector<Node> nodes;                                      // indexed by id, like the reference's Vec<Option<HnswNode>>
    std::vector<uint32_t> row_of_id;                              // device row per id (0xffffffff: absent) -- 4 bytes per node: what the walks' inner loop reads instead of the 80-byte Node
    bool has_ep = false; uint64_t ep = 0; size_t max_level = 0;
    size_t count = 0, dim = 0;
    std::mutex mu;
    uint64_t stats[4] = {0, 0, 0, 0};
    // the graph mirrored in HBM for the device-resident search (kernels_hnsw.hip); rebuilt when the graph changed
    uint64_t graph_version = 1, mirror_version = 0;
    uint32_t *d_row_of = nullptr, *d_level = nullptr, *d_nbr0 = nullptr, *d_cnt0 = nullptr, *d_up_off = nullptr, *d_nbrU = nullptr, *d_cntU = nullptr;
    uint32_t *d_nbr0_row = nullptr, *d_nbrU_row = nullptr;
    uint64_t* d_out_ids = nullptr; float* d_out_dists = nullptr; uint32_t *d_out_counts = nullptr, *d_fail = nullptr;
    size_t out_cap = 0, out_nq_cap = 0;
    uint32_t mirror_ids = 0, stride0 = 0, strideU = 0, max_list = 0;
    uint64_t device_queries = 0, host_redone = 0;
    // batched insert scans: two mapped host matrices [SCAN_CHUNK][scan_ld] the scan kernel writes, a stream and events
    float* h_scan[2] = {nullptr, nullptr}; float* d_scan[2] = {nullptr, nullptr}; size_t scan_ld = 0;
    hipStream_t scan_stream = nullptr; hipEvent_t scan_ev[2] = {nullptr, nullptr};
    bool host_only = false; size_t host_threads = 0;              // vdb_hnsw_set_traversal
    // incremental mirror: capacity in node ids / pooled upper lists, the upper-list offset of every node, and the nodes whose
    // lists changed since the mirror was last brought up to date (inserts touch ~33 nodes each; a bulk build syncs per chunk)
    uint32_t cap_ids = 0, cap_upper = 0, n_upper_used = 0;
    std::vector<uint32_t> h_up_off; std::vector<uint8_t> dirty_flag; std::vector<uint32_t> dirty;
    bool mirror_full = true;                                      // the next sync rebuilds the whole mirror
    uint32_t* h_stage = nullptr; uint32_t* d_stage = nullptr; size_t stage_words = 0;      // mapped staging of the scatter records
    // speculative insert walks: per walk of a chunk its query row / level (host -> device) and its record (device -> host), mapped
    uint32_t *h_wq = nullptr, *d_wq = nullptr;                    // [2][WALKS]: rows, levels
    uint32_t *h_rec_row = nullptr, *d_rec_row = nullptr, *h_rec_cnt = nullptr, *d_rec_cnt = nullptr;
    float *h_rec_d = nullptr, *d_rec_d = nullptr;
    bool spec_build = true;                                       // vdb_hnsw_set_build: 0 = the row-scan build of round 2 (A/B, tests)
    uint64_t bstats[8] = {0};                                     // vdb_hnsw_build_stats
    double btimes[4] = {0, 0, 0, 0};                              // vdb_hnsw_build_times
    const Node* node(uint64_t id) const { return id < nodes.size() && nodes[id].present ? &nodes[id] : nullptr; }
    bool removed_any = false;                                     // no removal so far: every listed neighbour exists, the walks skip the presence check
    bool has(uint64_t id) const { return !removed_any || (id < row_of_id.size() && row_of_id[id] != 0xffffffffu); }
};

namespace {

using Graph = vdb_hnsw_index;

// One search_layer call (graph.rs:143-199) as a resumable state machine: next_request() runs the loop up to the
// next batch of distances it needs (entry points first, then one neighbour expansion at a time), feed() consumes them.
struct LayerSearch {
    const Graph* g = nullptr;
    size_t ef = 1, layer = 0;
    RustHeap<-1> cand;                                            // MinHeap: closest candidate on top
    RustHeap<+1> res;                                             // MaxHeap: furthest result on top
    IdSet visited;
    std::vector<uint64_t> pending;
    int stage = 0;                                                // 0: entry points not yet requested, 1: requested, 2: main loop
    bool zero_norm = false;

    void start(const Graph* g_, uint64_t ep, size_t ef_, size_t layer_) {
        g = g_; ef = ef_; layer = layer_;
        cand.clear(); res.clear(); visited.clear();
        pending.assign(1, ep);
        stage = 0; zero_norm = false;
    }
    // The walk is a chain of dependent cache misses at a million nodes (node -> its list table -> the layer's list): the node
    // most likely to be expanded NEXT -- the top of the candidate heap -- is pulled towards the cache one level per call, while
    // the current expansion's distances are fetched and folded.
    void prefetch_next(int depth) const {
        if (cand.empty()) return;
        const uint64_t id = cand.top().id;
        if (id >= g->nodes.size()) return;
        const Node& n = g->nodes[id];
        if (depth == 0) { __builtin_prefetch(&n); return; }
        if (!n.present || layer >= n.nbr.size()) return;
        if (depth == 1) { __builtin_prefetch(&n.nbr[layer]); return; }
        __builtin_prefetch(n.nbr[layer].data());
    }
    bool next_request() {
        if (stage == 0) { stage = 1; return true; }
        while (true) {
            Nb c;
            if (!cand.pop(c)) return false;
            const float furthest = res.empty() ? F32_MAX : res.top().d;
            if (c.d > furthest) return false;
            pending.clear();
            const Node* n = g->node(c.id);
            if (n && layer < n->nbr.size()) {
                prefetch_next(1);
                for (uint64_t nid : n->nbr[layer]) {
                    if (!visited.insert(nid)) continue;
                    if (!g->has(nid)) continue;                   // skip deleted nodes
                    pending.push_back(nid);
                }
            }
            if (!pending.empty()) { prefetch_next(2); return true; }
        }
    }
    void feed(const float* d) {
        for (size_t i = 0; i < pending.size(); ++i) {
            if (is_zero_norm_mark(d[i])) { zero_norm = true; return; }
            const Nb n{d[i], pending[i]};
            if (stage == 1) {
                visited.insert(n.id);
                cand.push(n);
                res.push(n);
            } else {
                const float furthest = res.empty() ? F32_MAX : res.top().d;
                if (n.d < furthest || res.size() < ef) {
                    if (n.id < g->nodes.size()) __builtin_prefetch(&g->nodes[n.id]);
                    cand.push(n);
                    res.push(n);
                    if (res.size() > ef) { Nb drop; res.pop(drop); }
                }
            }
        }
        stage = 2;
        prefetch_next(1);
    }
    std::vector<Nb> sorted() const {                              // into_sorted_vec: backing array, stable sort by distance
        std::vector<Nb> v = res.v;
        std::stable_sort(v.begin(), v.end(), [](const Nb& a, const Nb& b) { return a.d < b.d; });
        return v;
    }
};

// No C++ exception may cross the C ABI (std::bad_alloc from nodes.resize(id + 1) on a sparse id, vector growth in a search).
template <class F> int guarded(F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { return vdb_internal::set_error(VDB_ERR_DEVICE, "internal error: out of host memory"); }
    catch (const std::exception& e) { return vdb_internal::set_error(VDB_ERR_DEVICE, e.what()); }
    catch (...) { return vdb_internal::set_error(VDB_ERR_DEVICE, "internal error: unknown C++ exception"); }
}

#define HN_TRY(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return vdb_internal::set_error(VDB_ERR_DEVICE, hipGetErrorString(e_));       \
    } while (0)

int zero_norm_error() {
    return vdb_internal::set_error(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
}

// graph.rs:118-123
size_t level_from_unit(const Graph* g, double r) {
    double v = std::floor(-std::log(r) * g->ml);
    size_t level = (v >= 1.8446744073709552e19) ? (size_t)-1 : (v != v ? 0 : (size_t)v);
    return std::min(level, g->max_layers - 1);
}
double next_unit(Graph* g) {
    uint64_t z = (g->rng += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

void mark_dirty(Graph* g, uint64_t id) {
    if (id >= g->dirty_flag.size()) g->dirty_flag.resize(std::max<size_t>(id + 1, g->dirty_flag.size() * 2), 0);
    if (!g->dirty_flag[id]) { g->dirty_flag[id] = 1; g->dirty.push_back((uint32_t)id); }
}

// graph.rs:244-342.  Every distance of the insert comes from `fetch(pending ids, out distances)` -- the reference's
// metric.distance(vector, stored) at graph.rs:155 / :182 -- and `level` is the node's level, drawn by the caller in insert
// order (graph.rs:118-123, level_from_unit / next_unit).
template <class Fetch> int insert_node(Graph* g, uint64_t id, uint32_t row, size_t level, Fetch&& fetch) {
    if (id >= g->nodes.size()) g->nodes.resize(id + 1);
    if (id >= g->row_of_id.size()) g->row_of_id.resize(std::max<size_t>(id + 1, g->row_of_id.size() * 2), 0xffffffffu);
    g->graph_version++;
    Node& nd = g->nodes[id];
    nd = Node();
    nd.present = true; nd.level = (uint32_t)level; nd.row = row;
    g->row_of_id[id] = row;
    nd.nbr.assign(level + 1, {});
    nd.nbr_d.assign(level + 1, {});
    g->count++;
    mark_dirty(g, id);
    if (!g->has_ep) { g->has_ep = true; g->ep = id; g->max_level = level; return VDB_OK; }
    uint64_t ep_id = g->ep;
    const size_t cur_max = g->max_level;
    // the walk's state outlives the insert (inserts run under the graph's mutex; per thread, so two indexes do not share it): a
    // fresh one grew its visited set from 1024 slots to 16384 again -- four re-hashes -- for every insert of a large build
    static thread_local LayerSearch ls;
    static thread_local std::vector<float> d;
    auto run_layer = [&](size_t ef, size_t layer, std::vector<Nb>& out) -> int {
        ls.start(g, ep_id, ef, layer);
        while (ls.next_request()) {
            d.resize(ls.pending.size());
            int frc = fetch(ls.pending, d, layer);
            if (frc) return frc;
            g->bstats[6] += d.size();
            ls.feed(d.data());
            if (ls.zero_norm) return zero_norm_error();
        }
        out = ls.sorted();
        return VDB_OK;
    };
    int rc;
    static thread_local std::vector<Nb> nearest, scored;
    if (cur_max > level)
        for (size_t l = cur_max; l >= level + 1; --l) {
            if ((rc = run_layer(1, l, nearest))) return rc;
            if (!nearest.empty()) ep_id = nearest[0].id;
        }
    const size_t from = std::min(level, cur_max);
    for (size_t l = from;; --l) {
        const size_t m = l == 0 ? g->m_max0 : g->m;
        if ((rc = run_layer(g->ef_construction, l, nearest))) return rc;
        const size_t take = std::min(nearest.size(), m);          // select_neighbors_simple (graph.rs:202-204)
        {
            Node& me = g->nodes[id];
            me.nbr[l].resize(take); me.nbr_d[l].resize(take);
            for (size_t i = 0; i < take; ++i) { me.nbr[l][i] = nearest[i].id; me.nbr_d[l][i] = nearest[i].d; }
        }
        // bidirectional links (graph.rs:303-327); a list that grew beyond m is pruned at once (prune_neighbors, :207-241):
        // its entries scored by their distance to the list's owner -- the cached edge distances --, stable sort, truncate
        // (the `take` owners are scattered over the node array: three rounds of independent prefetches instead of 3 x take
        // dependent misses)
        for (size_t i = 0; i < take; ++i) __builtin_prefetch(&g->nodes[nearest[i].id]);
        for (size_t i = 0; i < take; ++i) {
            const Node& nb = g->nodes[nearest[i].id];
            if (nb.present && l < nb.nbr.size()) { __builtin_prefetch(&nb.nbr[l]); __builtin_prefetch(&nb.nbr_d[l]); }
        }
        for (size_t i = 0; i < take; ++i) {
            const Node& nb = g->nodes[nearest[i].id];
            if (nb.present && l < nb.nbr.size()) { __builtin_prefetch(nb.nbr[l].data()); __builtin_prefetch(nb.nbr_d[l].data()); }
        }
        for (size_t i = 0; i < take; ++i) {
            Node& nb = g->nodes[nearest[i].id];
            if (!nb.present || l >= nb.nbr.size()) continue;
            mark_dirty(g, nearest[i].id);
            nb.nbr[l].push_back(id);
            nb.nbr_d[l].push_back(nearest[i].d);                  // d(nb, new) == d(new, nb), bit for bit
            if (nb.nbr[l].size() <= m) continue;
            scored.clear();
            for (size_t t = 0; t < nb.nbr[l].size(); ++t)
                if (g->node(nb.nbr[l][t])) scored.push_back(Nb{is_zero_norm_mark(nb.nbr_d[l][t]) ? F32_MAX : nb.nbr_d[l][t], nb.nbr[l][t]});
            std::stable_sort(scored.begin(), scored.end(), [](const Nb& a, const Nb& b) { return a.d < b.d; });
            if (scored.size() > m) scored.resize(m);
            nb.nbr[l].resize(scored.size()); nb.nbr_d[l].resize(scored.size());
            for (size_t t = 0; t < scored.size(); ++t) { nb.nbr[l][t] = scored[t].id; nb.nbr_d[l][t] = scored[t].d; }
        }
        if (!nearest.empty()) ep_id = nearest[0].id;
        if (l == 0) break;
    }
    if (level > g->max_level) { g->ep = id; g->max_level = level; }
    return VDB_OK;
}

int sync_mirror(vdb_hnsw_index* g, hipStream_t s, size_t reserve_ids, size_t reserve_upper);
bool walks_supported(const Graph* g);
int build_speculative(Graph* g, const uint64_t* ids, uint64_t first_id, size_t n, const std::vector<uint32_t>& rowv,
                      const std::vector<size_t>& lev, size_t* done);

int add_rows(Graph* g, const uint64_t* ids, uint64_t first_id, const float* rows, size_t n, size_t dim, const long* levels, long level1) {
    if (n == 0) return VDB_OK;
    if (dim == 0) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "zero-dimensional vectors are not indexable");
    if (g->count == 0 && !g->has_ep) g->dim = dim;
    if (dim != g->dim) return vdb_internal::set_dim_error(g->dim, dim);
    // The graph is a Vec indexed by id in the reference too (graph.rs:78, :249-251 resize_with(id + 1)), so a huge sparse id
    // costs id + 1 slots there as here; the device mirror addresses nodes with 32 bits, so larger ids are refused up front
    // (and an allocation failure of the resize is caught at the boundary instead of unwinding through it).
    for (size_t i = 0; i < n; ++i)
        if ((ids ? ids[i] : first_id + i) >= 0xfffffff0ull)
            return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "HNSW node ids must be below 2^32 - 16");
    int rc;
    if ((rc = vdb_flat_add_bulk(g->flat, ids, first_id, rows, n, dim))) return rc;
    // When insert number i fails (zero-norm Cosine pair), the reference has stored node i without links and has not seen
    // the vectors after it (mod.rs:37-42 returns at the first error): the rows after i leave the row store again.
    auto rollback_after = [&](size_t i_fail) {                               // (size_t)-1: nothing was inserted
        for (size_t t = i_fail + 1; t < n; ++t) (void)vdb_flat_remove(g->flat, ids ? ids[t] : first_id + t);
    };
    if ((rc = vdb_flat_flush(g->flat))) { rollback_after((size_t)-1); return rc; }         // every row of the batch is on the device now
    vdb_internal::DeviceView dv;
    if ((rc = vdb_internal::device_view(g->flat, &dv))) { rollback_after((size_t)-1); return rc; }
    std::vector<uint32_t> rowv(n);
    for (size_t i = 0; i < n; ++i) {
        rowv[i] = vdb_internal::row_of(g->flat, ids ? ids[i] : first_id + i);
        if (rowv[i] == 0xffffffffu) { rollback_after((size_t)-1); return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "row was not stored (the same id twice in one batch?)"); }
    }
    // levels in insert order (graph.rs:118-123): drawn here, so that the walks that run AHEAD of the inserts know them.  When
    // insert i fails, the reference has drawn i + 1 levels: the generator is put back to that point.
    std::vector<size_t> lev(n);
    std::vector<uint64_t> rng_after(n);
    const uint64_t rng0 = g->rng;
    for (size_t i = 0; i < n; ++i) {
        const long li = levels ? levels[i] : level1;
        lev[i] = li >= 0 ? std::min<size_t>((size_t)li, g->max_layers - 1) : level_from_unit(g, next_unit(g));
        rng_after[i] = g->rng;
    }
    auto fail_at = [&](size_t i_fail, int code) { g->rng = i_fail < n ? rng_after[i_fail] : rng0; rollback_after(i_fail); return code; };
    size_t done = 0;
    // ---- the frontier-only build: a device walk per insert evaluates what search_layer asks for, the host replays (below)
    if (g->spec_build && !g->host_only && n >= 32 && walks_supported(g)) {
        rc = build_speculative(g, ids, first_id, n, rowv, lev, &done);
        if (rc) return fail_at(done, rc);
        return VDB_OK;
    }
    // ---- the row-scan build (single adds, small batches, shapes the walk kernel does not take): distances ahead of the walks --
    // chunk c's scan = the chunk's vectors (16 per launch) against device rows
    // [0, last row of the chunk), written by the kernel into one of two mapped host matrices; chunk c+1's scan is enqueued
    // before the host walks chunk c, so the passes over the rows and their transfer hide behind the sequential walks.
    constexpr size_t CHUNK = 128;
    const size_t need_ld = ((size_t)rowv[n - 1] + 1023) & ~(size_t)1023;
    if (!g->scan_stream) {
        HN_TRY(hipStreamCreateWithFlags(&g->scan_stream, hipStreamNonBlocking));
        HN_TRY(hipEventCreateWithFlags(&g->scan_ev[0], hipEventDisableTiming));
        HN_TRY(hipEventCreateWithFlags(&g->scan_ev[1], hipEventDisableTiming));
    }
    if (need_ld > g->scan_ld) {
        const size_t ld_new = std::max(need_ld, g->scan_ld + g->scan_ld / 2);
        for (int t = 0; t < 2; ++t) {
            if (g->h_scan[t]) (void)hipHostFree(g->h_scan[t]);
            g->h_scan[t] = nullptr;
            HN_TRY(hipHostMalloc((void**)&g->h_scan[t], CHUNK * ld_new * sizeof(float), hipHostMallocMapped));
            HN_TRY(hipHostGetDevicePointer((void**)&g->d_scan[t], g->h_scan[t], 0));
        }
        g->scan_ld = ld_new;
    }
    const size_t n_chunks = (n + CHUNK - 1) / CHUNK;
    auto enqueue_scan = [&](size_t c) -> int {
        const size_t c0 = c * CHUNK, nc = std::min(CHUNK, n - c0);
        const uint32_t n_scan = rowv[c0 + nc - 1];                           // rows stored before the chunk's LAST vector
        for (size_t q0 = 0; q0 < nc && n_scan; q0 += 16) {
            vdb::ScanRowsParams sp{};
            sp.rows = dv.rows; sp.ld = dv.ld; sp.dim = dv.dim; sp.nd = dv.nd; sp.metric = dv.metric; sp.mark = vdb_internal::ZERO_NORM_MARK;
            sp.nq = (uint32_t)std::min<size_t>(16, nc - q0);
            for (uint32_t j = 0; j < sp.nq; ++j) sp.qrow[j] = rowv[c0 + q0 + j];
            sp.n_scan = n_scan; sp.out = g->d_scan[c & 1] + q0 * g->scan_ld; sp.ldm = g->scan_ld;
            vdb::launch_scan_rows(sp, g->scan_stream);
            g->stats[0] += (uint64_t)sp.nq * n_scan; g->stats[1]++;
        }
        HN_TRY(hipGetLastError());
        HN_TRY(hipEventRecord(g->scan_ev[c & 1], g->scan_stream));
        return VDB_OK;
    };
    auto drain = [&]() { (void)hipStreamSynchronize(g->scan_stream); };
    if ((rc = enqueue_scan(0))) { drain(); return fail_at((size_t)-1, rc); }
    for (size_t c = 0; c < n_chunks; ++c) {
        if (c + 1 < n_chunks && (rc = enqueue_scan(c + 1))) { drain(); return fail_at(c * CHUNK - 1, rc); }
        if (hipEventSynchronize(g->scan_ev[c & 1]) != hipSuccess) { drain(); return fail_at(c * CHUNK - 1, vdb_internal::set_error(VDB_ERR_DEVICE, "scan failed")); }
        const size_t c0 = c * CHUNK, nc = std::min(CHUNK, n - c0);
        for (size_t i = 0; i < nc; ++i) {
            const uint64_t id = ids ? ids[c0 + i] : first_id + c0 + i;
            // every distance this insert can ask for: the new vector against the rows stored before it
            const float* scan = g->h_scan[c & 1] + i * g->scan_ld;
            auto fetch = [&](const std::vector<uint64_t>& pend, std::vector<float>& d, size_t) -> int {
                for (size_t t = 0; t < pend.size(); ++t) d[t] = scan[g->row_of_id[pend[t]]];
                return VDB_OK;
            };
            g->bstats[7]++;
            if ((rc = insert_node(g, id, rowv[c0 + i], lev[c0 + i], fetch))) { drain(); return fail_at(c0 + i, rc); }
        }
    }
    return VDB_OK;
}

void free_mirror(vdb_hnsw_index* g);

}  // namespace

extern "C" {

int vdb_hnsw_create(int metric, size_t m, size_t ef_construction, size_t ef_search, uint64_t seed, int device,
                    vdb_hnsw_index** out) {
    return guarded([&]() -> int {
    if (!out) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (m < 2) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "m must be >= 2");
    vdb_flat_index* flat = nullptr;
    int rc = vdb_flat_create(metric, device, &flat);
    if (rc) return rc;
    auto* g = new vdb_hnsw_index();
    g->flat = flat; g->metric = metric; g->m = m; g->m_max0 = 2 * m; g->ef_construction = ef_construction;
    g->ef_search = ef_search; g->ml = 1.0 / std::log((double)m); g->rng = seed;              // graph.rs:49-59
    *out = g;
    return VDB_OK;
    });
}

void vdb_hnsw_destroy(vdb_hnsw_index* g) {
    if (!g) return;
    free_mirror(g);
    if (g->scan_stream) { (void)hipStreamSynchronize(g->scan_stream); (void)hipStreamDestroy(g->scan_stream); }
    for (int t = 0; t < 2; ++t) { if (g->h_scan[t]) (void)hipHostFree(g->h_scan[t]); if (g->scan_ev[t]) (void)hipEventDestroy(g->scan_ev[t]); }
    vdb_flat_destroy(g->flat);
    delete g;
}

int vdb_hnsw_add(vdb_hnsw_index* g, uint64_t id, const float* v, size_t dim, long level) {
    return guarded([&]() -> int {
    if (!g || !v) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    return add_rows(g, &id, 0, v, 1, dim, nullptr, level);
    });
}

int vdb_hnsw_add_bulk(vdb_hnsw_index* g, const uint64_t* ids, uint64_t first_id, const float* rows, size_t n, size_t dim) {
    return guarded([&]() -> int {
    if (!g || (!rows && n)) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    return add_rows(g, ids, first_id, rows, n, dim, nullptr, -1);
    });
}

int vdb_hnsw_remove(vdb_hnsw_index* g, uint64_t id) {             // graph.rs:345-381
    return guarded([&]() -> int {
    if (!g) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null handle");
    std::lock_guard<std::mutex> lk(g->mu);
    if (!g->node(id)) return VDB_OK;
    g->graph_version++;
    g->mirror_full = true;                                         // lists that still name the node need its row cleared: rebuilt as a whole
    Node gone = std::move(g->nodes[id]);
    g->nodes[id] = Node();
    if (id < g->row_of_id.size()) g->row_of_id[id] = 0xffffffffu;
    g->removed_any = true;
    for (size_t l = 0; l < gone.nbr.size(); ++l)
        for (uint64_t nid : gone.nbr[l]) {
            if (nid >= g->nodes.size() || !g->nodes[nid].present || l >= g->nodes[nid].nbr.size()) continue;
            auto& lst = g->nodes[nid].nbr[l];
            auto& dst = g->nodes[nid].nbr_d[l];
            size_t w = 0;
            for (size_t t = 0; t < lst.size(); ++t)
                if (lst[t] != id) { lst[w] = lst[t]; dst[w] = dst[t]; ++w; }
            lst.resize(w); dst.resize(w);
        }
    g->count--;
    int rc = vdb_flat_remove(g->flat, id);
    if (g->has_ep && g->ep == id) {
        g->has_ep = false; g->max_level = 0;
        size_t best = 0;
        for (size_t i = 0; i < g->nodes.size(); ++i)            // max_by_key(level): the LAST of the equally maximal nodes
            if (g->nodes[i].present && (!g->has_ep || g->nodes[i].level >= best)) { g->has_ep = true; g->ep = i; best = g->nodes[i].level; }
        g->max_level = g->has_ep ? best : 0;
    }
    return rc;
    });
}

}  // extern "C"

namespace {

// host traversal (one GPU launch per round for the candidate lists of every query): the path for what the device-resident
// search does not take (m > 19, ef > 1022, overflowing queries, vdb_hnsw_set_traversal(h, 1, ..))
int search_host(vdb_hnsw_index* g, const float* queries, size_t nq, size_t dim, size_t k, size_t ef,
                uint64_t* out_ids, float* out_dists, size_t* out_counts) {
    int rc;
    if ((rc = vdb_internal::pairs_begin(g->flat, queries, nq, dim))) return rc;
    const size_t ef_actual = std::max(ef ? ef : g->ef_search, k);
    struct Q { LayerSearch ls; size_t layer; uint64_t ep; bool done; size_t off, n; };
    std::vector<Q> qs(nq);
    for (Q& q : qs) {
        q.layer = g->max_level; q.ep = g->ep; q.done = false; q.off = q.n = 0;
        q.ls.start(g, q.ep, q.layer >= 1 ? 1 : ef_actual, q.layer);
    }
    // The per-round host work (heap operations, neighbour scans of every in-flight query) is spread over worker
    // threads, each owning a contiguous block of queries; the GPU evaluates the round's pairs in ONE launch.
    unsigned hw = std::thread::hardware_concurrency();
    size_t T = std::min<size_t>({(size_t)(hw ? hw : 1), (size_t)16, std::max<size_t>(nq / 8, 1)});
    if (g->host_threads) T = g->host_threads;
    struct Local { std::vector<uint32_t> pq, pr; size_t base = 0; };
    std::vector<Local> loc(T);
    std::vector<uint32_t> pq, pr;
    std::vector<float> pd;
    std::atomic<int> arrived{0}, phase{0};
    std::atomic<bool> stop{false}, zero{false};
    int rc_eval = VDB_OK;
    auto barrier = [&]() {                                         // sense-reversing spin barrier over T threads
        const int ph = phase.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (int)T) {
            arrived.store(0, std::memory_order_relaxed);
            phase.store(ph + 1, std::memory_order_release);
        } else {
            while (phase.load(std::memory_order_acquire) == ph) { /* spin */ }
        }
    };
    auto request_phase = [&](size_t t) {
        Local& L = loc[t];
        L.pq.clear(); L.pr.clear();
        const size_t b0 = nq * t / T, b1 = nq * (t + 1) / T;
        for (size_t b = b0; b < b1; ++b) {
            Q& q = qs[b];
            q.n = 0;
            while (!q.done) {
                if (q.ls.next_request()) {
                    q.off = L.pq.size(); q.n = q.ls.pending.size();
                    for (uint64_t nid : q.ls.pending) { L.pq.push_back((uint32_t)b); L.pr.push_back(g->nodes[nid].row); }
                    break;
                }
                // layer finished (graph.rs:399-411)
                std::vector<Nb> r = q.ls.sorted();
                if (q.layer >= 1) {
                    if (!r.empty()) q.ep = r[0].id;
                    --q.layer;
                    q.ls.start(g, q.ep, q.layer >= 1 ? 1 : ef_actual, q.layer);
                } else {
                    const size_t cnt = std::min(r.size(), k);
                    for (size_t i = 0; i < cnt; ++i) { out_ids[b * k + i] = r[i].id; out_dists[b * k + i] = r[i].d; }
                    out_counts[b] = cnt;
                    q.done = true;
                }
            }
        }
    };
    auto feed_phase = [&](size_t t) {
        const size_t b0 = nq * t / T, b1 = nq * (t + 1) / T;
        for (size_t b = b0; b < b1; ++b) {
            Q& q = qs[b];
            if (q.done || !q.n) continue;
            q.ls.feed(pd.data() + loc[t].base + q.off);
            if (q.ls.zero_norm) zero.store(true, std::memory_order_relaxed);
        }
    };
    auto worker = [&](size_t t) {
        while (true) {
            request_phase(t);
            barrier();
            if (t == 0) {                                          // the round's pairs of every thread -> one launch
                size_t total = 0;
                for (Local& L : loc) { L.base = total; total += L.pq.size(); }
                if (total == 0) stop.store(true, std::memory_order_release);
                else {
                    pq.resize(total); pr.resize(total); pd.resize(total);
                    for (Local& L : loc)
                        if (!L.pq.empty()) {
                            memcpy(pq.data() + L.base, L.pq.data(), L.pq.size() * 4);
                            memcpy(pr.data() + L.base, L.pr.data(), L.pr.size() * 4);
                        }
                    rc_eval = vdb_internal::pairs_eval(g->flat, pq.data(), pr.data(), total, pd.data());
                    g->stats[0] += total; g->stats[1]++; g->stats[2]++; g->stats[3] += total;
                    if (rc_eval) stop.store(true, std::memory_order_release);
                }
            }
            barrier();
            if (stop.load(std::memory_order_acquire)) return;
            feed_phase(t);
            if (zero.load(std::memory_order_relaxed)) { /* every thread sees it after the next barrier */ }
            barrier();
            if (zero.load(std::memory_order_acquire)) return;
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < T; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (std::thread& th : pool) th.join();
    if (rc_eval) return rc_eval;
    if (zero.load()) return zero_norm_error();
    return VDB_OK;
}

void free_mirror(vdb_hnsw_index* g) {
    for (uint32_t** p : {&g->d_row_of, &g->d_level, &g->d_nbr0, &g->d_cnt0, &g->d_up_off, &g->d_nbrU, &g->d_cntU, &g->d_nbr0_row, &g->d_nbrU_row, &g->d_out_counts, &g->d_fail})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (g->d_out_ids) { (void)hipFree(g->d_out_ids); g->d_out_ids = nullptr; }
    if (g->d_out_dists) { (void)hipFree(g->d_out_dists); g->d_out_dists = nullptr; }
    g->out_cap = g->out_nq_cap = 0;
    g->mirror_version = 0;
    g->mirror_full = true; g->cap_ids = g->cap_upper = 0;
    for (void* hp : {(void*)g->h_stage, (void*)g->h_wq, (void*)g->h_rec_cnt, (void*)g->h_rec_row, (void*)g->h_rec_d}) if (hp) (void)hipHostFree(hp);
    g->h_stage = nullptr; g->stage_words = 0; g->h_wq = nullptr; g->h_rec_cnt = nullptr; g->h_rec_row = nullptr; g->h_rec_d = nullptr;
}

// Mirrors the graph into HBM (kernels.h HnswSearchParams): per node id its device row (0xffffffff = absent), level,
// layer-0 list (stride m_max0) and the offset of its upper-layer lists (stride m) in a pooled array; every list entry carries
// the device row of the neighbour beside its id.  A FULL rebuild allocates capacity beyond the present graph (reserve_*); after
// it, inserts are mirrored INCREMENTALLY: the nodes whose lists changed (mark_dirty) are packed into mapped staging memory and
// scattered by one small kernel -- a bulk build brings the mirror up to date once per chunk of inserts, not once per graph.
int sync_mirror(vdb_hnsw_index* g, hipStream_t s, size_t reserve_ids, size_t reserve_upper) {
    const uint32_t n = (uint32_t)g->nodes.size();
    const uint32_t stride0 = (uint32_t)g->m_max0, strideU = (uint32_t)g->m;
    auto row_now = [&](uint64_t x) -> uint32_t { const Node* t = g->node(x); return t ? t->row : 0xffffffffu; };
    // upper-list offsets of nodes that do not have one yet (new nodes of level >= 1)
    bool full = g->mirror_full || !g->d_row_of || n > g->cap_ids || stride0 != g->stride0 || strideU != g->strideU;
    if (!full) {
        if (g->h_up_off.size() < n) g->h_up_off.resize(n, 0xffffffffu);
        for (uint32_t id : g->dirty) {
            const Node& nd = g->nodes[id];
            if (!nd.present || nd.level == 0 || g->h_up_off[id] != 0xffffffffu) continue;
            if ((size_t)g->n_upper_used + nd.level > g->cap_upper) { full = true; break; }
            g->h_up_off[id] = g->n_upper_used;
            g->n_upper_used += nd.level;
        }
    }
    if (!full) {
        if (g->dirty.empty()) return VDB_OK;
        // ---- incremental: records of the dirty nodes
        size_t nU = 0;
        for (uint32_t id : g->dirty) if (g->nodes[id].present) nU += g->nodes[id].level;
        const size_t w0 = 4 + 2 * (size_t)stride0, wU = 1 + 2 * (size_t)strideU;
        const size_t words = g->dirty.size() * w0 + nU * wU;
        if (words > g->stage_words) {
            if (g->h_stage) (void)hipHostFree(g->h_stage);
            g->h_stage = g->d_stage = nullptr; g->stage_words = 0;
            const size_t cap = words + words / 2 + 4096;
            HN_TRY(hipHostMalloc((void**)&g->h_stage, cap * 4, hipHostMallocMapped));
            HN_TRY(hipHostGetDevicePointer((void**)&g->d_stage, g->h_stage, 0));
            g->stage_words = cap;
        }
        uint32_t* r0 = g->h_stage;
        uint32_t* rU = g->h_stage + g->dirty.size() * w0;
        uint32_t cU = 0;
        for (size_t t = 0; t < g->dirty.size(); ++t) {
            const uint32_t id = g->dirty[t];
            const Node& nd = g->nodes[id];
            uint32_t* r = r0 + t * w0;
            r[0] = id; r[1] = nd.present ? nd.row : 0xffffffffu; r[2] = nd.present ? nd.level : 0u;
            r[3] = nd.present && nd.level ? g->h_up_off[id] : 0u;
            for (uint32_t i = 0; i < stride0; ++i) {
                const bool in = nd.present && i < nd.nbr[0].size();
                r[4 + i] = in ? (uint32_t)nd.nbr[0][i] : 0xffffffffu;
                r[4 + stride0 + i] = in ? row_now(nd.nbr[0][i]) : 0xffffffffu;
            }
            if (nd.present)
                for (size_t l = 1; l < nd.nbr.size(); ++l) {
                    uint32_t* u = rU + (size_t)cU++ * wU;
                    u[0] = g->h_up_off[id] + (uint32_t)(l - 1);
                    for (uint32_t i = 0; i < strideU; ++i) {
                        const bool in = i < nd.nbr[l].size();
                        u[1 + i] = in ? (uint32_t)nd.nbr[l][i] : 0xffffffffu;
                        u[1 + strideU + i] = in ? row_now(nd.nbr[l][i]) : 0xffffffffu;
                    }
                }
            g->dirty_flag[id] = 0;
        }
        vdb::HnswScatterParams sp{g->d_stage, (uint32_t)g->dirty.size(), g->d_stage + g->dirty.size() * w0, cU,
                                  g->d_row_of, g->d_level, g->d_up_off, g->d_nbr0, g->d_nbr0_row, stride0, g->d_nbrU, g->d_nbrU_row, strideU};
        vdb::launch_hnsw_scatter(sp, s);
        HN_TRY(hipGetLastError());
        HN_TRY(hipStreamSynchronize(s));                               // the staging memory is reused by the next sync
        g->dirty.clear();
        g->mirror_ids = n;
        g->mirror_version = g->graph_version;
        return VDB_OK;
    }
    // ---- full rebuild, with room to grow
    uint32_t n_upper = 0;
    for (const Node& nd : g->nodes) if (nd.present) n_upper += nd.level;
    const uint32_t cap_ids = (uint32_t)std::min<size_t>(0xfffffff0ull, std::max<size_t>({(size_t)n + n / 4 + 1024, reserve_ids, (size_t)1}));
    const uint32_t cap_up = (uint32_t)std::min<size_t>(0xfffffff0ull, std::max<size_t>({(size_t)n_upper + n_upper / 4 + 1024, reserve_upper, (size_t)1}));
    std::vector<uint32_t> row_of(cap_ids, 0xffffffffu), level(cap_ids, 0), up_off(cap_ids, 0), nbr0((size_t)cap_ids * stride0, 0xffffffffu);
    std::vector<uint32_t> nbrU((size_t)cap_up * strideU, 0xffffffffu);
    std::vector<uint32_t> nbr0_row(nbr0.size(), 0xffffffffu), nbrU_row(nbrU.size(), 0xffffffffu);
    g->h_up_off.assign(n, 0xffffffffu);
    uint32_t off = 0;
    for (uint32_t id = 0; id < n; ++id) {
        const Node& nd = g->nodes[id];
        if (!nd.present) continue;
        if (nd.nbr[0].size() > stride0) return vdb_internal::set_error(VDB_ERR_DEVICE, "internal error: a layer-0 list exceeds m_max0");
        row_of[id] = nd.row; level[id] = nd.level; up_off[id] = off;
        if (nd.level) g->h_up_off[id] = off;
        for (size_t i = 0; i < nd.nbr[0].size(); ++i) { nbr0[(size_t)id * stride0 + i] = (uint32_t)nd.nbr[0][i]; nbr0_row[(size_t)id * stride0 + i] = row_now(nd.nbr[0][i]); }
        for (size_t l = 1; l < nd.nbr.size(); ++l) {
            if (nd.nbr[l].size() > strideU) return vdb_internal::set_error(VDB_ERR_DEVICE, "internal error: an upper list exceeds m");
            for (size_t i = 0; i < nd.nbr[l].size(); ++i) { nbrU[(size_t)(off + l - 1) * strideU + i] = (uint32_t)nd.nbr[l][i]; nbrU_row[(size_t)(off + l - 1) * strideU + i] = row_now(nd.nbr[l][i]); }
        }
        off += nd.level;
    }
    g->n_upper_used = off;
    for (uint32_t** p : {&g->d_row_of, &g->d_level, &g->d_nbr0, &g->d_cnt0, &g->d_up_off, &g->d_nbrU, &g->d_cntU, &g->d_nbr0_row, &g->d_nbrU_row})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    auto up = [&](uint32_t** dst, const std::vector<uint32_t>& v) -> int {
        HN_TRY(hipMalloc((void**)dst, std::max<size_t>(v.size(), 1) * 4));
        if (!v.empty()) HN_TRY(hipMemcpyAsync(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice, s));
        return VDB_OK;
    };
    int rc;
    if ((rc = up(&g->d_row_of, row_of)) || (rc = up(&g->d_level, level)) || (rc = up(&g->d_nbr0, nbr0)) ||
        (rc = up(&g->d_up_off, up_off)) || (rc = up(&g->d_nbrU, nbrU)) ||
        (rc = up(&g->d_nbr0_row, nbr0_row)) || (rc = up(&g->d_nbrU_row, nbrU_row)))
        return rc;
    HN_TRY(hipStreamSynchronize(s));                                   // the staging vectors go out of scope
    g->cap_ids = cap_ids; g->cap_upper = cap_up;
    g->mirror_ids = n; g->stride0 = stride0; g->strideU = strideU; g->max_list = std::max(stride0, strideU);
    g->mirror_version = g->graph_version;
    g->mirror_full = false;
    for (uint32_t id : g->dirty) if (id < g->dirty_flag.size()) g->dirty_flag[id] = 0;
    g->dirty.clear();
    return VDB_OK;
}
int upload_mirror(vdb_hnsw_index* g, hipStream_t s) {
    if (g->mirror_version == g->graph_version && !g->mirror_full) return VDB_OK;
    return sync_mirror(g, s, 0, 0);
}

bool walks_supported(const Graph* g) {
    return g->nodes.size() < 0xfffffff0ull &&
           vdb::hnsw_search_supported((uint32_t)g->dim, (uint32_t)std::min<size_t>(g->ef_construction, 0xffffffu), 1u, (uint32_t)std::max(g->m_max0, g->m) + 1);
}

// ------------------------------------------------------------------ the frontier-only build
// Round 2 answered every distance of an insert from a scan of ALL stored rows against the new vector: N^2 / 2 distances for N
// inserts (5.1e11 at 1M x 768), 70x what the reference's algorithm asks for -- search_layer evaluates only the neighbours of the
// nodes it expands (graph.rs:155, :182; ~3-7 thousand per insert).  Inserts are sequential by definition, and a GPU round trip
// per expansion is 8 ms per insert, so the scan was the way to have no round trip at all.  This build has none either, and
// evaluates only frontiers:
//   per chunk of WALKS inserts, ONE launch of the device-resident walk (kernels_hnsw.hip, insert mode) runs every insert's
//   own walk -- greedy descent above its level, search_layer(ef_construction) at and below -- on the graph AS OF THE CHUNK'S
//   START, and records every (row, distance) it evaluates into mapped host memory; one more launch gives the distances between
//   the chunk's own vectors.  Then the host replays the inserts IN ORDER with the reference's algorithm, operation by operation,
//   on the real graph: a distance is looked up in that insert's record (or the in-chunk matrix).  The real walk differs from
//   the speculative one only where an earlier insert of the same chunk changed a list on its way; a distance it then needs and
//   the record does not hold is a MISS, evaluated on the GPU at once (one small launch per expansion with misses).
// The graph is the sequential one by construction -- the replay IS the reference's insert, speculation only decides which
// distances are already there.  (tests/test_gpu_hnsw.py: node for node equal to the CPU restatement.)
constexpr uint32_t WALKS = 64, REC_CAP = 12288;           // (the walk's visited set holds 12288 nodes: a longer walk fails anyway)
// The walks run ONE BLOCK AHEAD of the replay, on their own stream: while the host replays block b (64 inserts, ~11 ms), the GPU
// walks block b + 1 on the graph as of the end of block b - 1 (~8 ms: a walk is a latency chain, 64 of them occupy 64 CUs and
// leave the rest to the replay's misses).  The host never waits for a walk launch; an insert's record is 64 + its position
// inserts stale -- distances to the vectors of the block in between come from the window matrix below, like those of the insert's
// own block.  The block is as small as the walk's latency allows: misses fall with the staleness (200k x 768: 2.1 round trips
// per insert with blocks of 128, 1.1 with 64, 0.83 with 48, where the host starts to wait for the walks).

int build_speculative(Graph* g, const uint64_t* ids, uint64_t first_id, size_t n, const std::vector<uint32_t>& rowv,
                      const std::vector<size_t>& lev, size_t* done) {
    int rc;
    *done = (size_t)-1;
    vdb_internal::DeviceView dv;
    if ((rc = vdb_internal::device_view(g->flat, &dv))) return rc;
    hipStream_t s = (hipStream_t)dv.stream;
    if (!g->h_wq) {                                              // two sets of everything a walk launch touches: block b + 1 is walked while b is replayed
        HN_TRY(hipHostMalloc((void**)&g->h_wq, 2 * 2 * WALKS * 4, hipHostMallocMapped));
        HN_TRY(hipHostGetDevicePointer((void**)&g->d_wq, g->h_wq, 0));
        HN_TRY(hipHostMalloc((void**)&g->h_rec_cnt, 2 * WALKS * 4, hipHostMallocMapped));
        HN_TRY(hipHostGetDevicePointer((void**)&g->d_rec_cnt, g->h_rec_cnt, 0));
        HN_TRY(hipHostMalloc((void**)&g->h_rec_row, (size_t)2 * WALKS * REC_CAP * 4, hipHostMallocMapped));
        HN_TRY(hipHostGetDevicePointer((void**)&g->d_rec_row, g->h_rec_row, 0));
        HN_TRY(hipHostMalloc((void**)&g->h_rec_d, (size_t)2 * WALKS * REC_CAP * 4, hipHostMallocMapped));
        HN_TRY(hipHostGetDevicePointer((void**)&g->d_rec_d, g->h_rec_d, 0));
    }
    if (!g->scan_stream) {
        HN_TRY(hipStreamCreateWithFlags(&g->scan_stream, hipStreamNonBlocking));
        HN_TRY(hipEventCreateWithFlags(&g->scan_ev[0], hipEventDisableTiming));
        HN_TRY(hipEventCreateWithFlags(&g->scan_ev[1], hipEventDisableTiming));
    }
    const hipStream_t sw = g->scan_stream;
    if (!g->d_fail || g->out_nq_cap < WALKS) {
        if (g->d_fail) (void)hipFree(g->d_fail);
        if (g->d_out_counts) (void)hipFree(g->d_out_counts);
        g->d_fail = g->d_out_counts = nullptr;
        HN_TRY(hipMalloc((void**)&g->d_fail, WALKS * 4));
        HN_TRY(hipMalloc((void**)&g->d_out_counts, WALKS * 4));
        if (g->d_out_ids) { (void)hipFree(g->d_out_ids); g->d_out_ids = nullptr; }
        if (g->d_out_dists) { (void)hipFree(g->d_out_dists); g->d_out_dists = nullptr; }
        g->out_cap = 0; g->out_nq_cap = WALKS;
    }
    // room in the mirror for the whole batch: ids up to the largest of the batch, upper lists for its levels
    size_t max_id = 0, up_need = 0;
    for (size_t i = 0; i < n; ++i) { max_id = std::max<size_t>(max_id, ids ? ids[i] : first_id + i); up_need += lev[i]; }
    size_t up_have = 0;
    for (const Node& nd : g->nodes) if (nd.present) up_have += nd.level;
    if (max_id + 1 > g->cap_ids || up_have + up_need > g->cap_upper) g->mirror_full = true;
    constexpr uint32_t TAB = 32768;                                 // hash slots of an insert's record (at most REC_CAP entries)
    std::vector<uint32_t> tab_row(TAB), fetch_miss_idx, miss_a, miss_b;
    std::vector<float> tab_d(TAB), win, miss_d;
    std::vector<uint32_t> pa, pb;
    auto id_of = [&](size_t i) { return ids ? ids[i] : first_id + i; };
    // a block's ids -> positions: a subtraction when they are consecutive (the usual case), a map otherwise
    struct BlockIds {
        uint64_t lo = ~0ull, hi = 0; bool dense = false; size_t nc = 0;
        std::unordered_map<uint64_t, uint32_t> pos;
        long find(uint64_t id) const {
            if (nc == 0 || id < lo || id > hi) return -1;
            if (dense) return (long)(id - lo);
            auto it = pos.find(id);
            return it == pos.end() ? -1 : (long)it->second;
        }
    } blk[2];
    auto describe = [&](BlockIds& B, size_t c0, size_t nc) {
        B.lo = ~0ull; B.hi = 0; B.nc = nc; B.pos.clear();
        for (size_t i = 0; i < nc; ++i) { B.lo = std::min<uint64_t>(B.lo, id_of(c0 + i)); B.hi = std::max<uint64_t>(B.hi, id_of(c0 + i)); }
        B.dense = nc && B.hi - B.lo + 1 == nc;
        for (size_t i = 0; i < nc && B.dense; ++i) B.dense = id_of(c0 + i) == B.lo + i;
        if (!B.dense) for (size_t i = 0; i < nc; ++i) B.pos[id_of(c0 + i)] = (uint32_t)i;
    };
    // no walk may be in flight when this function returns: it writes the mapped records and reads the mirror
    struct Drain { hipStream_t st; ~Drain() { (void)hipStreamSynchronize(st); } } drain{sw};
    bool walked[2] = {false, false};                               // did the block of this parity get a walk launch (a graph existed)?
    auto launch_walks = [&](size_t b) -> int {                     // the mirror is up to date and no walk is in flight
        const size_t c0 = b * WALKS, nc = std::min<size_t>(WALKS, n - c0);
        const uint32_t par = (uint32_t)(b & 1);
        walked[par] = g->has_ep;
        if (walked[par]) {
            uint32_t* wq = g->h_wq + (size_t)par * 2 * WALKS;
            for (size_t i = 0; i < nc; ++i) { wq[i] = rowv[c0 + i]; wq[WALKS + i] = (uint32_t)lev[c0 + i]; }
            vdb::HnswSearchParams hp{};
            hp.rows = dv.rows; hp.ld = dv.ld; hp.dim = dv.dim; hp.nd = dv.nd; hp.metric = dv.metric; hp.qp = nullptr; hp.qnorm = nullptr;
            hp.row_of = g->d_row_of; hp.level = g->d_level; hp.n_ids = g->mirror_ids; hp.nbr0 = g->d_nbr0; hp.nbr0_row = g->d_nbr0_row; hp.cnt0 = nullptr;
            hp.stride0 = g->stride0; hp.up_off = g->d_up_off; hp.nbrU = g->d_nbrU; hp.nbrU_row = g->d_nbrU_row; hp.cntU = nullptr; hp.strideU = g->strideU;
            hp.entry_point = (uint32_t)g->ep; hp.max_level = (uint32_t)g->max_level; hp.ef = (uint32_t)g->ef_construction; hp.k = 0;
            hp.out_ids = nullptr; hp.out_dists = nullptr; hp.out_counts = g->d_out_counts; hp.fail = g->d_fail; hp.status = dv.status;
            hp.qrow = g->d_wq + (size_t)par * 2 * WALKS; hp.qlevel = hp.qrow + WALKS;
            hp.rec_row = g->d_rec_row + (size_t)par * WALKS * REC_CAP; hp.rec_d = g->d_rec_d + (size_t)par * WALKS * REC_CAP;
            hp.rec_cnt = g->d_rec_cnt + (size_t)par * WALKS; hp.rec_cap = REC_CAP; hp.rec_zero_mark = vdb_internal::ZERO_NORM_MARK;
            vdb::launch_hnsw_search(hp, (uint32_t)nc, sw);
            HN_TRY(hipGetLastError());
            g->stats[1]++;
        }
        HN_TRY(hipEventRecord(g->scan_ev[par], sw));
        return VDB_OK;
    };
    auto clock = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const size_t n_blocks = (n + WALKS - 1) / WALKS;
    {
        const auto t0 = clock();
        if ((rc = sync_mirror(g, s, max_id + 1, up_have + up_need))) return rc;
        g->btimes[0] += secs(t0, clock());
        if ((rc = launch_walks(0))) return rc;
    }
    for (size_t b = 0; b < n_blocks; ++b) {
        const size_t c0 = b * WALKS, nc = std::min<size_t>(WALKS, n - c0);
        const uint32_t par = (uint32_t)(b & 1);
        const size_t p0 = b ? c0 - WALKS : 0, pn = b ? WALKS : 0;       // the block before this one (of this call)
        // ---- this block's walks (launched a block ago) are done; the next block's start on the graph as it is now
        const auto t_block = clock();
        HN_TRY(hipEventSynchronize(g->scan_ev[par]));
        const auto t_walked = clock();
        if (b + 1 < n_blocks) {
            if ((rc = sync_mirror(g, s, max_id + 1, up_have + up_need))) return rc;
            if ((rc = launch_walks(b + 1))) return rc;
        }
        const auto t_synced = clock();
        // ---- the window matrix: this block's vectors against each other (pair (i, j), j < i, at win[i (i - 1) / 2 + j]) and
        // against the block before (behind the triangle: (i, j) at win[tri_n + i * pn + j]) -- what no walk has seen
        describe(blk[par], c0, nc);
        if (b == 0) describe(blk[par ^ 1], 0, 0);
        const BlockIds& cur = blk[par]; const BlockIds& prev = blk[par ^ 1];
        const size_t tri_n = nc * (nc - 1) / 2;
        pa.clear(); pb.clear();
        for (size_t i = 1; i < nc; ++i) for (size_t j = 0; j < i; ++j) { pa.push_back(rowv[c0 + i]); pb.push_back(rowv[c0 + j]); }
        for (size_t i = 0; i < nc; ++i) for (size_t j = 0; j < pn; ++j) { pa.push_back(rowv[c0 + i]); pb.push_back(rowv[p0 + j]); }
        win.resize(pa.size());
        if (!pa.empty()) {
            if ((rc = vdb_internal::rows_eval(g->flat, pa.data(), pb.data(), pa.size(), win.data()))) return rc;
            g->stats[0] += pa.size(); g->stats[1]++; g->bstats[2] += pa.size();
        }
        const uint32_t* rec_cnt = g->h_rec_cnt + (size_t)par * WALKS;
        if (walked[par])
            for (size_t i = 0; i < nc; ++i) {
                g->bstats[1] += std::min<uint32_t>(rec_cnt[i], REC_CAP);
                g->stats[0] += std::min<uint32_t>(rec_cnt[i], REC_CAP);
                if (rec_cnt[i] > REC_CAP) g->bstats[5]++;
            }
        // ---- the authoritative replay, insert by insert
        const auto t_ready = clock();
        g->btimes[0] += secs(t_walked, t_synced);
        g->btimes[1] += secs(t_block, t_walked) + secs(t_synced, t_ready);
        struct ReplayClock {
            Graph* g; std::chrono::steady_clock::time_point t0;
            ~ReplayClock() { g->btimes[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
        } replay_clock{g, t_ready};
        for (size_t i = 0; i < nc; ++i) {
            const uint64_t id = id_of(c0 + i);
            const uint32_t my_row = rowv[c0 + i];
            // this insert's record as a hash table node id -> distance (ids, not rows: the replay's inner loop then touches nothing but
            // this table -- a per-distance lookup in the 4 MB id -> row array was a third of its time at 1M nodes)
            std::fill(tab_row.begin(), tab_row.end(), 0xffffffffu);
            const uint32_t cnt = walked[par] ? std::min<uint32_t>(rec_cnt[i], REC_CAP) : 0u;
            const uint32_t* rr = g->h_rec_row + ((size_t)par * WALKS + i) * REC_CAP;
            const float* rd = g->h_rec_d + ((size_t)par * WALKS + i) * REC_CAP;
            for (uint32_t t = 0; t < cnt; ++t) {
                uint32_t h = (rr[t] * 0x9e3779b1u) >> 17;
                while (tab_row[h] != 0xffffffffu && tab_row[h] != rr[t]) h = (h + 1) & (TAB - 1);
                tab_row[h] = rr[t]; tab_d[h] = rd[t];
            }
            auto fetch = [&](const std::vector<uint64_t>& pend, std::vector<float>& d, size_t) -> int {
                fetch_miss_idx.clear();
                for (size_t t = 0; t < pend.size(); ++t) {
                    const uint64_t nid = pend[t];
                    long cp = cur.find(nid);
                    if (cp >= 0) {                                       // a vector of this block, inserted before this one
                        const uint32_t a = (uint32_t)i, b2 = (uint32_t)cp;
                        d[t] = a > b2 ? win[(size_t)a * (a - 1) / 2 + b2] : win[(size_t)b2 * (b2 - 1) / 2 + a];
                        continue;
                    }
                    if ((cp = prev.find(nid)) >= 0) { d[t] = win[tri_n + i * pn + (size_t)cp]; continue; }     // of the block before
                    const uint32_t key = (uint32_t)nid;                  // (table key: node ids are below 2^32 - 16)
                    uint32_t h = (key * 0x9e3779b1u) >> 17;
                    while (tab_row[h] != 0xffffffffu && tab_row[h] != key) h = (h + 1) & (TAB - 1);
                    if (tab_row[h] == key) d[t] = tab_d[h];
                    else fetch_miss_idx.push_back((uint32_t)t);
                }
                // the real walk left the speculative one: evaluate what it asks for now.  (Tried: the neighbours of the missed nodes
                // in the same round trip -- 24 % fewer round trips at 200k x 768, but 9x the pairs and the host's bookkeeping for them:
                // 52.6 s against 47.8 s for the build.)
                if (!fetch_miss_idx.empty()) {
                    miss_a.assign(fetch_miss_idx.size(), my_row);
                    miss_b.resize(fetch_miss_idx.size());
                    miss_d.resize(fetch_miss_idx.size());
                    for (size_t t = 0; t < fetch_miss_idx.size(); ++t) miss_b[t] = g->row_of_id[pend[fetch_miss_idx[t]]];
                    const auto t_miss = std::chrono::steady_clock::now();
                    int r2 = vdb_internal::rows_eval(g->flat, miss_a.data(), miss_b.data(), miss_a.size(), miss_d.data());
                    g->btimes[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_miss).count();
                    if (r2) return r2;
                    for (size_t t = 0; t < fetch_miss_idx.size(); ++t) d[fetch_miss_idx[t]] = miss_d[t];
                    g->stats[0] += miss_a.size(); g->stats[1]++;
                    g->bstats[3] += miss_a.size(); g->bstats[4]++;
                }
                return VDB_OK;
            };
            g->bstats[0]++;
            if ((rc = insert_node(g, id, my_row, lev[c0 + i], fetch))) { *done = c0 + i; return rc; }
            *done = c0 + i;
        }
    }
    return VDB_OK;
}

// device-resident search of the whole batch in one launch; queries whose walk overflowed the kernel's LDS structures are
// listed in `redo`
int search_device(vdb_hnsw_index* g, const float* queries, size_t nq, size_t dim, size_t k, size_t ef_actual,
                  uint64_t* out_ids, float* out_dists, size_t* out_counts, std::vector<uint32_t>& redo) {
    int rc;
    if ((rc = vdb_internal::pairs_begin(g->flat, queries, nq, dim))) return rc;
    vdb_internal::DeviceView dv;
    if ((rc = vdb_internal::device_view(g->flat, &dv))) return rc;
    hipStream_t s = (hipStream_t)dv.stream;
    if ((rc = upload_mirror(g, s))) return rc;
    const size_t need = nq * std::max<size_t>(k, 1);
    if (need > g->out_cap || nq > g->out_nq_cap) {
        if (g->d_out_ids) (void)hipFree(g->d_out_ids);
        if (g->d_out_dists) (void)hipFree(g->d_out_dists);
        if (g->d_out_counts) (void)hipFree(g->d_out_counts);
        if (g->d_fail) (void)hipFree(g->d_fail);
        g->d_out_ids = nullptr; g->d_out_dists = nullptr; g->d_out_counts = g->d_fail = nullptr; g->out_cap = g->out_nq_cap = 0;
        HN_TRY(hipMalloc((void**)&g->d_out_ids, need * 8));
        HN_TRY(hipMalloc((void**)&g->d_out_dists, need * 4));
        HN_TRY(hipMalloc((void**)&g->d_out_counts, nq * 4));
        HN_TRY(hipMalloc((void**)&g->d_fail, nq * 4));
        g->out_cap = need; g->out_nq_cap = nq;
    }
    HN_TRY(hipMemsetAsync(dv.status, 0, 16, s));
    vdb::HnswSearchParams hp{};
    hp.rows = dv.rows; hp.ld = dv.ld; hp.dim = dv.dim; hp.nd = dv.nd; hp.metric = dv.metric; hp.qp = dv.qp; hp.qnorm = dv.qnorm;
    hp.row_of = g->d_row_of; hp.level = g->d_level; hp.n_ids = g->mirror_ids; hp.nbr0 = g->d_nbr0; hp.nbr0_row = g->d_nbr0_row; hp.cnt0 = g->d_cnt0;
    hp.stride0 = g->stride0; hp.up_off = g->d_up_off; hp.nbrU = g->d_nbrU; hp.nbrU_row = g->d_nbrU_row; hp.cntU = g->d_cntU; hp.strideU = g->strideU;
    hp.entry_point = (uint32_t)g->ep; hp.max_level = (uint32_t)g->max_level; hp.ef = (uint32_t)ef_actual; hp.k = (uint32_t)k;
    hp.out_ids = g->d_out_ids; hp.out_dists = g->d_out_dists; hp.out_counts = g->d_out_counts; hp.fail = g->d_fail; hp.status = dv.status;
    vdb::launch_hnsw_search(hp, (uint32_t)nq, s);
    HN_TRY(hipGetLastError());
    std::vector<uint32_t> cnt(nq), fail(nq);
    uint32_t status = 0;
    HN_TRY(hipMemcpyAsync(cnt.data(), g->d_out_counts, nq * 4, hipMemcpyDeviceToHost, s));
    HN_TRY(hipMemcpyAsync(fail.data(), g->d_fail, nq * 4, hipMemcpyDeviceToHost, s));
    HN_TRY(hipMemcpyAsync(&status, dv.status, 4, hipMemcpyDeviceToHost, s));
    if (k) {
        HN_TRY(hipMemcpyAsync(out_ids, g->d_out_ids, nq * k * 8, hipMemcpyDeviceToHost, s));
        HN_TRY(hipMemcpyAsync(out_dists, g->d_out_dists, nq * k * 4, hipMemcpyDeviceToHost, s));
    }
    HN_TRY(hipStreamSynchronize(s));
    g->stats[1]++;
    if (status & 2u) return zero_norm_error();                     // ST_ZERO_QUERY
    for (size_t b = 0; b < nq; ++b) {
        if (fail[b]) { redo.push_back((uint32_t)b); out_counts[b] = 0; }
        else out_counts[b] = cnt[b];
    }
    g->device_queries += nq - redo.size();
    return VDB_OK;
}

}  // namespace

extern "C" {

int vdb_hnsw_search_batch(vdb_hnsw_index* g, const float* queries, size_t nq, size_t dim, size_t k, size_t ef,
                          uint64_t* out_ids, float* out_dists, size_t* out_counts) {
    return guarded([&]() -> int {
    if (!g || (nq && (!queries || !out_counts || (k && (!out_ids || !out_dists)))))
        return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    for (size_t b = 0; b < nq; ++b) out_counts[b] = 0;
    g->stats[2] = g->stats[3] = 0;
    if (nq == 0 || !g->has_ep) return VDB_OK;                     // graph.rs:392-395: empty graph -> Ok(vec![])
    if (dim != g->dim) return vdb_internal::set_dim_error(dim, g->dim);   // distance.rs:21-26 on the first evaluation
    const size_t ef_actual = std::max(ef ? ef : g->ef_search, k);
    const bool on_device = !g->host_only && k > 0 && g->nodes.size() < 0xffffffffull &&
                           vdb::hnsw_search_supported((uint32_t)g->dim, (uint32_t)std::min<size_t>(ef_actual, 0xffffffu), (uint32_t)std::min<size_t>(k, 0xffffffu),
                                                      (uint32_t)std::max(g->m_max0, g->m) + 1);
    if (!on_device) return search_host(g, queries, nq, dim, k, ef, out_ids, out_dists, out_counts);
    int rc;
    std::vector<uint32_t> redo;
    if ((rc = search_device(g, queries, nq, dim, k, ef_actual, out_ids, out_dists, out_counts, redo))) return rc;
    if (!redo.empty()) {                                            // the walks that did not fit the kernel's LDS structures
        g->host_redone += redo.size();
        std::vector<float> q2(redo.size() * dim);
        for (size_t j = 0; j < redo.size(); ++j) memcpy(q2.data() + j * dim, queries + (size_t)redo[j] * dim, dim * sizeof(float));
        std::vector<uint64_t> i2(redo.size() * k);
        std::vector<float> d2(redo.size() * k);
        std::vector<size_t> c2(redo.size());
        if ((rc = search_host(g, q2.data(), redo.size(), dim, k, ef, i2.data(), d2.data(), c2.data()))) return rc;
        for (size_t j = 0; j < redo.size(); ++j) {
            memcpy(out_ids + (size_t)redo[j] * k, i2.data() + j * k, k * 8);
            memcpy(out_dists + (size_t)redo[j] * k, d2.data() + j * k, k * 4);
            out_counts[redo[j]] = c2[j];
        }
    }
    return VDB_OK;
    });
}

int vdb_hnsw_set_traversal(vdb_hnsw_index* g, int host_only, size_t host_threads) {
    return guarded([&]() -> int {
    if (!g) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null handle");
    std::lock_guard<std::mutex> lk(g->mu);
    g->host_only = host_only != 0;
    g->host_threads = std::min<size_t>(host_threads, 64);
    return VDB_OK;
    });
}

size_t vdb_hnsw_len(const vdb_hnsw_index* g) { return g ? g->count : 0; }
int vdb_hnsw_metric(const vdb_hnsw_index* g) { return g ? g->metric : -1; }

int vdb_hnsw_get_vector(vdb_hnsw_index* g, uint64_t id, float* out, size_t cap, size_t* dim) {
    return guarded([&]() -> int {
    if (!g) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null handle");
    std::lock_guard<std::mutex> lk(g->mu);
    if (!g->node(id)) return vdb_internal::set_error(VDB_ERR_NOT_FOUND, "Vector not found");
    return vdb_flat_get_vector(g->flat, id, out, cap, dim);
    });
}

long vdb_hnsw_neighbors(const vdb_hnsw_index* g, uint64_t id, size_t layer, uint64_t* out, size_t cap) {
    if (!g) return -1;
    const Node* n = g->node(id);
    if (!n || layer >= n->nbr.size()) return -1;
    for (size_t i = 0; i < n->nbr[layer].size() && i < cap; ++i) out[i] = n->nbr[layer][i];
    return (long)n->nbr[layer].size();
}
long vdb_hnsw_node_level(const vdb_hnsw_index* g, uint64_t id) {
    const Node* n = g ? g->node(id) : nullptr;
    return n ? (long)n->level : -1;
}
int vdb_hnsw_entry_point(const vdb_hnsw_index* g, uint64_t* id, size_t* max_level) {
    return guarded([&]() -> int {
    if (!g) return 0;
    if (id) *id = g->ep;
    if (max_level) *max_level = g->max_level;
    return g->has_ep ? 1 : 0;
    });
}
int vdb_hnsw_set_build(vdb_hnsw_index* g, int frontier_only) {
    return guarded([&]() -> int {
    if (!g) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null handle");
    std::lock_guard<std::mutex> lk(g->mu);
    g->spec_build = frontier_only != 0;
    return VDB_OK;
    });
}
int vdb_hnsw_build_stats(const vdb_hnsw_index* g, uint64_t out[8]) {
    return guarded([&]() -> int {
    if (!g || !out) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(out, g->bstats, sizeof(g->bstats));
    return VDB_OK;
    });
}
int vdb_hnsw_build_times(const vdb_hnsw_index* g, double out[4]) {
    return guarded([&]() -> int {
    if (!g || !out) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(out, g->btimes, sizeof(g->btimes));
    return VDB_OK;
    });
}
int vdb_hnsw_stats(const vdb_hnsw_index* g, uint64_t out[6]) {
    return guarded([&]() -> int {
    if (!g || !out) return vdb_internal::set_error(VDB_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(out, g->stats, sizeof(g->stats));
    out[4] = g->device_queries;
    out[5] = g->host_redone;
    return VDB_OK;
    });
}

}  // extern "C"
