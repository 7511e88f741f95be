/*
 * hnsw_oracle.c -- CPU restatement of the reference's HNSW index (src/hnsw/graph.rs,
 * src/hnsw/neighbor_queue.rs, src/hnsw/mod.rs).
 *
 * TEST INFRASTRUCTURE ONLY (see flat_oracle.c): the checker for the GPU-offloaded HNSW of
 * BASELINE config 5 ("GPU offload of candidate-list distance evaluations only, recall@10 vs
 * CPU HNSW").  Nothing in the product links or calls this file.
 *
 * Parity status: PINNED by the reference's own unit tests for this module, replayed in
 * tests/test_hnsw_oracle.py (graph.rs:436-538, neighbor_queue.rs:150-195, mod.rs:84-154) and
 * by the recall floors of tests/recall_test.rs:67-80.  Two things in the reference are
 * not reproducible and are fixed here by the build's own choice:
 *   - node levels come from StdRng::from_entropy() (graph.rs:101,119-123): unseeded.  The
 *     oracle (and the product) draw r from a seeded splitmix64 stream instead; the level
 *     formula floor(-ln(r) * ml), capped at max_layers-1, is the reference's;
 *   - nothing else: heap layouts, tie orders, pruning and deletion follow the reference
 *     operation by operation, including the array layout of Rust's std BinaryHeap (which
 *     decides the order of equal-distance results in into_sorted_vec, neighbor_queue.rs:102-106).
 *
 * Distances are flat_oracle.c's (vdbo_distance): the reference's f32 operation order.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int vdbo_distance(int metric, const float *a, size_t da, const float *b, size_t db, float *out);

#define H_OK 0
#define H_ERR_ARG 5

typedef struct { float distance; uint64_t id; } nb_t;           /* neighbor_queue.rs:7-11 */

/* neighbor_queue.rs:37-43  Ord for Neighbor: distance.partial_cmp (unordered -> Equal), then id */
static int nb_cmp(const nb_t *a, const nb_t *b) {
    if (a->distance < b->distance) return -1;
    if (a->distance > b->distance) return 1;
    if (a->id < b->id) return -1;
    if (a->id > b->id) return 1;
    return 0;
}

/* Rust std::collections::BinaryHeap<T> as a max-heap under `cmp`; sign = -1 gives BinaryHeap<Reversed>
 * (neighbor_queue.rs:47-60).  push = sift_up(0, old_len); pop = swap last into the root,
 * sift_down_to_bottom(0), sift_up.  The element moves are the std library's, so the backing array --
 * which into_vec() exposes -- has the same layout. */
typedef struct { nb_t *d; size_t len, cap; int sign; } heap_t;

static int hcmp(const heap_t *h, const nb_t *a, const nb_t *b) { return h->sign * nb_cmp(a, b); }
static void heap_init(heap_t *h, int sign) { h->d = NULL; h->len = h->cap = 0; h->sign = sign; }
static void heap_free(heap_t *h) { free(h->d); h->d = NULL; h->len = h->cap = 0; }
static void heap_sift_up(heap_t *h, size_t start, size_t pos) {
    nb_t elem = h->d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (hcmp(h, &elem, &h->d[parent]) <= 0) break;
        h->d[pos] = h->d[parent];
        pos = parent;
    }
    h->d[pos] = elem;
}
static void heap_push(heap_t *h, nb_t n) {
    if (h->len == h->cap) { h->cap = h->cap ? 2 * h->cap : 64; h->d = (nb_t *)realloc(h->d, h->cap * sizeof(nb_t)); }
    h->d[h->len] = n;
    heap_sift_up(h, 0, h->len++);
}
static int heap_pop(heap_t *h, nb_t *out) {
    if (!h->len) return 0;
    nb_t item = h->d[--h->len];
    if (h->len) {
        nb_t top = h->d[0];
        h->d[0] = item;
        item = top;
        /* sift_down_to_bottom(0) */
        size_t end = h->len, pos = 0;
        nb_t elem = h->d[0];
        size_t child = 1;
        while (end >= 2 && child <= end - 2) {
            if (hcmp(h, &h->d[child], &h->d[child + 1]) <= 0) child += 1;
            h->d[pos] = h->d[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { h->d[pos] = h->d[child]; pos = child; }
        h->d[pos] = elem;
        heap_sift_up(h, 0, pos);
    }
    *out = item;
    return 1;
}

/* stable insertion sort by distance only (neighbor_queue.rs:104, graph.rs:230: sort_by on partial_cmp,
 * unordered -> Equal; slice::sort_by is stable) */
static void sort_by_distance(nb_t *v, size_t n) {
    for (size_t i = 1; i < n; ++i) {
        nb_t x = v[i];
        size_t j = i;
        while (j > 0 && v[j - 1].distance > x.distance) { v[j] = v[j - 1]; --j; }
        v[j] = x;
    }
}

typedef struct {                       /* graph.rs:63-72 HnswNode */
    int present;
    float *vec;
    size_t level;
    uint64_t **nbr;                    /* nbr[l][0..cnt[l]) */
    size_t *cnt, *capn;
} node_t;

typedef struct {                       /* graph.rs:75-91 HnswGraph, graph.rs:19-59 HnswParams */
    node_t *nodes; size_t n_nodes;
    int has_ep; uint64_t entry_point; size_t max_level;
    size_t m, m_max0, ef_construction, ef_search, max_layers; double ml;
    int metric; size_t dim;
    uint64_t rng;
    size_t count;
    uint64_t n_dist;                   /* distance evaluations (statistics) */
} graph_t;

static double next_unit(graph_t *g) {  /* splitmix64 -> 53-bit uniform in [0,1)  (the build's seeded stand-in for rng.gen::<f64>()) */
    uint64_t z = (g->rng += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

/* graph.rs:118-123 random_level */
size_t vdbo_hnsw_level_from_unit(double r, double ml, size_t max_layers) {
    double v = floor(-log(r) * ml);
    size_t level = (v >= 1.8446744073709552e19 || v != v) ? (v != v ? 0 : (size_t)-1) : (size_t)v;   /* `as usize` saturates */
    return level < max_layers - 1 ? level : max_layers - 1;
}

void *vdbo_hnsw_create(int metric, size_t m, size_t ef_construction, size_t ef_search, uint64_t seed) {
    graph_t *g = (graph_t *)calloc(1, sizeof(graph_t));
    g->metric = metric; g->m = m; g->m_max0 = 2 * m; g->ef_construction = ef_construction; g->ef_search = ef_search;
    g->ml = 1.0 / log((double)m); g->max_layers = 16;                 /* graph.rs:50-58 */
    g->rng = seed;
    return g;
}
static void node_free(node_t *n) {
    if (!n->present) return;
    for (size_t l = 0; l <= n->level; ++l) free(n->nbr[l]);
    free(n->nbr); free(n->cnt); free(n->capn); free(n->vec);
    n->present = 0;
}
void vdbo_hnsw_destroy(void *h) {
    graph_t *g = (graph_t *)h;
    for (size_t i = 0; i < g->n_nodes; ++i) node_free(&g->nodes[i]);
    free(g->nodes); free(g);
}
size_t vdbo_hnsw_len(const void *h) { return ((const graph_t *)h)->count; }
uint64_t vdbo_hnsw_distance_evals(const void *h) { return ((const graph_t *)h)->n_dist; }
int vdbo_hnsw_entry_point(const void *h, uint64_t *ep, size_t *max_level) {
    const graph_t *g = (const graph_t *)h;
    if (ep) *ep = g->entry_point;
    if (max_level) *max_level = g->max_level;
    return g->has_ep;
}
/* neighbours of node `id` at `layer` (graph.rs:68); returns the count, or -1 if the node / layer does not exist */
long vdbo_hnsw_neighbors(const void *h, uint64_t id, size_t layer, uint64_t *out, size_t cap) {
    const graph_t *g = (const graph_t *)h;
    if (id >= g->n_nodes || !g->nodes[id].present || layer > g->nodes[id].level) return -1;
    const node_t *n = &g->nodes[id];
    for (size_t i = 0; i < n->cnt[layer] && i < cap; ++i) out[i] = n->nbr[layer][i];
    return (long)n->cnt[layer];
}
long vdbo_hnsw_node_level(const void *h, uint64_t id) {
    const graph_t *g = (const graph_t *)h;
    if (id >= g->n_nodes || !g->nodes[id].present) return -1;
    return (long)g->nodes[id].level;
}

/* graph.rs:126-131 distance(query, node) */
static int g_distance(graph_t *g, const float *q, size_t qdim, uint64_t node_id, float *out) {
    node_t *n = &g->nodes[node_id];
    if (!n->present) return 6;                                  /* IndexError("Node not found") */
    g->n_dist++;
    return vdbo_distance(g->metric, q, qdim, n->vec, g->dim, out);
}

/* a visited set keyed by node id (graph.rs:151 HashSet): membership only, so a byte map will do */
typedef struct { uint8_t *seen; uint64_t *touched; size_t n_touched, cap_touched; } visited_t;

/* graph.rs:143-199 search_layer (Algorithm 2).  Returns results.into_sorted_vec() in *out (malloc'ed). */
static int search_layer(graph_t *g, const float *q, size_t qdim, const uint64_t *ep, size_t n_ep, size_t ef, size_t layer,
                        nb_t **out, size_t *n_out) {
    uint8_t *seen = (uint8_t *)calloc(g->n_nodes ? g->n_nodes : 1, 1);
    heap_t cand, res;
    heap_init(&cand, -1);                                        /* MinHeap: closest candidate on top */
    heap_init(&res, +1);                                         /* MaxHeap: furthest result on top   */
    int rc = H_OK;
    for (size_t i = 0; i < n_ep; ++i) {
        float d;
        if ((rc = g_distance(g, q, qdim, ep[i], &d))) goto done;
        seen[ep[i]] = 1;
        nb_t n = {d, ep[i]};
        heap_push(&cand, n);
        heap_push(&res, n);
    }
    nb_t c;
    while (heap_pop(&cand, &c)) {
        float furthest = res.len ? res.d[0].distance : 3.40282347e+38f;          /* f32::MAX */
        if (c.distance > furthest) break;
        node_t *node = &g->nodes[c.id];
        if (node->present && layer <= node->level) {
            for (size_t i = 0; i < node->cnt[layer]; ++i) {
                uint64_t nid = node->nbr[layer][i];
                if (nid < g->n_nodes && seen[nid]) continue;
                if (nid < g->n_nodes) seen[nid] = 1;
                if (nid >= g->n_nodes || !g->nodes[nid].present) continue;        /* skip deleted nodes */
                float d;
                if ((rc = g_distance(g, q, qdim, nid, &d))) goto done;
                furthest = res.len ? res.d[0].distance : 3.40282347e+38f;
                if (d < furthest || res.len < ef) {
                    nb_t n = {d, nid};
                    heap_push(&cand, n);
                    heap_push(&res, n);
                    if (res.len > ef) { nb_t drop; heap_pop(&res, &drop); }
                }
            }
        }
    }
    *out = (nb_t *)malloc((res.len ? res.len : 1) * sizeof(nb_t));
    memcpy(*out, res.d, res.len * sizeof(nb_t));                 /* into_vec(): the heap's backing array */
    *n_out = res.len;
    sort_by_distance(*out, *n_out);                              /* neighbor_queue.rs:102-106 */
done:
    heap_free(&cand); heap_free(&res); free(seen);
    return rc;
}

static void list_push(node_t *n, size_t l, uint64_t id) {
    if (n->cnt[l] == n->capn[l]) { n->capn[l] = n->capn[l] ? 2 * n->capn[l] : 8; n->nbr[l] = (uint64_t *)realloc(n->nbr[l], n->capn[l] * sizeof(uint64_t)); }
    n->nbr[l][n->cnt[l]++] = id;
}

/* graph.rs:207-241 prune_neighbors */
static void prune_neighbors(graph_t *g, uint64_t node_id, size_t layer, size_t m) {
    node_t *node = &g->nodes[node_id];
    if (!node->present || layer > node->level) return;
    size_t n = node->cnt[layer], k = 0;
    nb_t *scored = (nb_t *)malloc((n ? n : 1) * sizeof(nb_t));
    for (size_t i = 0; i < n; ++i) {
        uint64_t nid = node->nbr[layer][i];
        if (nid >= g->n_nodes || !g->nodes[nid].present) continue;                /* filter_map drops deleted ids */
        float d;
        g->n_dist++;
        if (vdbo_distance(g->metric, node->vec, g->dim, g->nodes[nid].vec, g->dim, &d)) d = 3.40282347e+38f;   /* unwrap_or(f32::MAX) */
        scored[k].distance = d; scored[k].id = nid; ++k;
    }
    sort_by_distance(scored, k);
    if (k > m) k = m;
    for (size_t i = 0; i < k; ++i) node->nbr[layer][i] = scored[i].id;
    node->cnt[layer] = k;
    free(scored);
}

/* graph.rs:244-342 insert (Algorithm 1).  level < 0: draw it from the seeded stream. */
int vdbo_hnsw_insert(void *h, uint64_t id, const float *vec, size_t dim, long level_in) {
    graph_t *g = (graph_t *)h;
    if (g->count == 0 && !g->has_ep) g->dim = dim;               /* the flat store's first row fixes the dimension */
    size_t level = level_in >= 0 ? (size_t)level_in : vdbo_hnsw_level_from_unit(next_unit(g), g->ml, g->max_layers);
    if (id >= g->n_nodes) {
        g->nodes = (node_t *)realloc(g->nodes, (id + 1) * sizeof(node_t));
        memset(g->nodes + g->n_nodes, 0, (id + 1 - g->n_nodes) * sizeof(node_t));
        g->n_nodes = id + 1;
    }
    node_t *node = &g->nodes[id];
    node_free(node);                                             /* `self.nodes[id] = Some(node)` replaces whatever was there */
    node->present = 1; node->level = level;
    node->vec = (float *)malloc((dim ? dim : 1) * sizeof(float));
    memcpy(node->vec, vec, dim * sizeof(float));
    node->nbr = (uint64_t **)calloc(level + 1, sizeof(uint64_t *));
    node->cnt = (size_t *)calloc(level + 1, sizeof(size_t));
    node->capn = (size_t *)calloc(level + 1, sizeof(size_t));
    g->count++;
    if (!g->has_ep) { g->has_ep = 1; g->entry_point = id; g->max_level = level; return H_OK; }
    uint64_t ep_id = g->entry_point;
    size_t cur_max = g->max_level;
    int rc;
    /* phase 1: greedy descent from the top layer down to level+1 (ef = 1) */
    if (cur_max > level) {
        for (size_t l = cur_max; l >= level + 1; --l) {
            nb_t *nearest; size_t nn;
            if ((rc = search_layer(g, vec, dim, &ep_id, 1, 1, l, &nearest, &nn))) return rc;
            if (nn) ep_id = nearest[0].id;
            free(nearest);
            if (l == 0) break;
        }
    }
    /* phase 2: layers min(level, cur_max) .. 0 */
    size_t from = level < cur_max ? level : cur_max;
    for (size_t l = from;; --l) {
        size_t m = l == 0 ? g->m_max0 : g->m;
        nb_t *nearest; size_t nn;
        if ((rc = search_layer(g, vec, dim, &ep_id, 1, g->ef_construction, l, &nearest, &nn))) return rc;
        size_t take = nn < m ? nn : m;                            /* select_neighbors_simple: the first m (graph.rs:202-204) */
        node = &g->nodes[id];
        node->cnt[l] = 0;
        for (size_t i = 0; i < take; ++i) list_push(node, l, nearest[i].id);
        for (size_t i = 0; i < take; ++i) {
            node_t *nb = &g->nodes[nearest[i].id];
            if (nb->present && l <= nb->level) {
                list_push(nb, l, id);
                if (nb->cnt[l] > m) prune_neighbors(g, nearest[i].id, l, m);
            }
        }
        if (nn) ep_id = nearest[0].id;
        free(nearest);
        if (l == 0) break;
    }
    if (level > g->max_level) { g->entry_point = id; g->max_level = level; }
    return H_OK;
}

/* graph.rs:345-381 remove (lazy deletion) */
int vdbo_hnsw_remove(void *h, uint64_t id) {
    graph_t *g = (graph_t *)h;
    if (id >= g->n_nodes || !g->nodes[id].present) return H_OK;
    node_t *node = &g->nodes[id];
    for (size_t l = 0; l <= node->level; ++l)
        for (size_t i = 0; i < node->cnt[l]; ++i) {
            uint64_t nid = node->nbr[l][i];
            if (nid >= g->n_nodes || nid == id) continue;
            node_t *nb = &g->nodes[nid];
            if (!nb->present || l > nb->level) continue;
            size_t k = 0;
            for (size_t j = 0; j < nb->cnt[l]; ++j) if (nb->nbr[l][j] != id) nb->nbr[l][k++] = nb->nbr[l][j];    /* retain */
            nb->cnt[l] = k;
        }
    node_free(node);
    g->count--;
    if (g->has_ep && g->entry_point == id) {
        /* max_by_key(level): the LAST of the equally maximal elements */
        g->has_ep = 0; g->max_level = 0;
        size_t best = 0;
        for (size_t i = 0; i < g->n_nodes; ++i)
            if (g->nodes[i].present && (!g->has_ep || g->nodes[i].level >= best)) { g->has_ep = 1; g->entry_point = i; best = g->nodes[i].level; }
        g->max_level = g->has_ep ? best : 0;
    }
    return H_OK;
}

/* graph.rs:386-412 search_knn (Algorithm 5); HnswIndex::search uses ef = 50 (mod.rs:71) */
int vdbo_hnsw_search(void *h, const float *q, size_t dim, size_t k, size_t ef, uint64_t *out_ids, float *out_dists, size_t *out_count) {
    graph_t *g = (graph_t *)h;
    *out_count = 0;
    if (!g->has_ep) return H_OK;
    uint64_t ep_id = g->entry_point;
    int rc;
    for (size_t l = g->max_level; l >= 1; --l) {
        nb_t *nearest; size_t nn;
        if ((rc = search_layer(g, q, dim, &ep_id, 1, 1, l, &nearest, &nn))) return rc;
        if (nn) ep_id = nearest[0].id;
        free(nearest);
    }
    size_t ef_actual = ef > k ? ef : k;
    nb_t *res; size_t nr;
    if ((rc = search_layer(g, q, dim, &ep_id, 1, ef_actual, 0, &res, &nr))) return rc;
    if (nr > k) nr = k;
    for (size_t i = 0; i < nr; ++i) { out_ids[i] = res[i].id; out_dists[i] = res[i].distance; }
    *out_count = nr;
    free(res);
    return H_OK;
}

/* the heap behaviour itself, for the replay of neighbor_queue.rs:150-195: push all, then pop all (max-heap when sign > 0) */
void vdbo_heap_replay(int sign, const float *dists, const uint64_t *ids, size_t n, size_t bound, float *out_dists, uint64_t *out_ids, size_t *out_n) {
    heap_t hp; heap_init(&hp, sign);
    for (size_t i = 0; i < n; ++i) { nb_t x = {dists[i], ids[i]}; heap_push(&hp, x); if (bound && hp.len > bound) { nb_t d; heap_pop(&hp, &d); } }
    nb_t x; size_t k = 0;
    while (heap_pop(&hp, &x)) { out_dists[k] = x.distance; out_ids[k] = x.id; ++k; }
    *out_n = k;
    heap_free(&hp);
}
