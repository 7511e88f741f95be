// vdb_store.cpp -- the device-resident mirror of the reference's FlatIndex rows (src/flat_index.rs:12-50): host staging of
// single adds, upload at the next search, tombstones, rows of another dimension kept host-side.
//
// Device layout (all in HBM, one allocation each, grown by doubling):
//   rows     [cap][ld] f32   ld = dim rounded up to 32, zero padded (K stage of the MFMA kernel)
//   nd       [cap]     f32   exact-order row norm  (vector.rs:35-37)
//   alpha,beta [cap]   f32   ranking score = fma(dot, alpha, beta)
//   row_ids  [cap]     u64   device row -> reference internal id
//   live     [cap/32]  u32   tombstone bitmask (remove() clears a bit; rows are append-only)
#include <numeric>

#include "vdb_index.h"

namespace vdbi {

// ------------------------------------------------------------------ device store management
int grow(Index* ix, uint32_t need_rows) {
    if (need_rows <= ix->cap_rows) return VDB_OK;
    uint32_t cap = std::max<uint32_t>({need_rows, ix->cap_rows * 2u, 1024u});
    cap = round_up(cap, 256);
    float *rows = nullptr, *nd = nullptr, *al = nullptr, *be = nullptr, *mg = nullptr;
    uint64_t* ids = nullptr;
    uint32_t* lv = nullptr;
    size_t row_bytes = (size_t)ix->ld * sizeof(float);
    HIP_TRY(hipMalloc((void**)&rows, (size_t)cap * row_bytes));
    HIP_TRY(hipMalloc((void**)&nd, (size_t)cap * 4));
    HIP_TRY(hipMalloc((void**)&al, (size_t)cap * 4));
    HIP_TRY(hipMalloc((void**)&be, (size_t)cap * 4));
    if (ix->metric != vdb::COSINE) HIP_TRY(hipMalloc((void**)&mg, (size_t)cap * 4));
    HIP_TRY(hipMalloc((void**)&ids, (size_t)cap * 8));
    HIP_TRY(hipMalloc((void**)&lv, (size_t)cap / 8));
    hipStream_t s = ix->stream;
    uint32_t old = ix->n_uploaded;
    uint16_t* r16 = nullptr;
    if (ix->shadow) {
        HIP_TRY(hipMalloc((void**)&r16, (size_t)cap * ix->ld * 2));
        if (old && ix->d_rows16) HIP_TRY(hipMemcpyAsync(r16, ix->d_rows16, (size_t)old * ix->ld * 2, hipMemcpyDeviceToDevice, s));
        else if (old) vdb::launch_rows_to_bf16(ix->d_rows, r16, ix->ld, 0, old, s);     // no shadow yet: from the f32 rows, never left unset
        HIP_TRY(hipMemsetAsync((char*)r16 + (size_t)old * ix->ld * 2, 0, (size_t)(cap - old) * ix->ld * 2, s));
    }
    if (old) {
        HIP_TRY(hipMemcpyAsync(rows, ix->d_rows, (size_t)old * row_bytes, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(nd, ix->d_nd, (size_t)old * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(al, ix->d_alpha, (size_t)old * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(be, ix->d_beta, (size_t)old * 4, hipMemcpyDeviceToDevice, s));
        if (mg) HIP_TRY(hipMemcpyAsync(mg, ix->d_margin, (size_t)old * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(ids, ix->d_row_ids, (size_t)old * 8, hipMemcpyDeviceToDevice, s));
    }
    if (mg) HIP_TRY(hipMemsetAsync(mg + old, 0, (size_t)(cap - old) * 4, s));   // rows past the last one are staged by the kernels (ragged tile)
    // zero the rest of the row block: the [dim, ld) padding columns must read as 0
    HIP_TRY(hipMemsetAsync((char*)rows + (size_t)old * row_bytes, 0, (size_t)(cap - old) * row_bytes, s));
    HIP_TRY(hipMemsetAsync(lv, 0, (size_t)cap / 8, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (ix->d_rows) {
        (void)hipFree(ix->d_rows); (void)hipFree(ix->d_nd); (void)hipFree(ix->d_alpha);
        (void)hipFree(ix->d_beta); (void)hipFree(ix->d_row_ids); (void)hipFree(ix->d_live);
        if (ix->d_margin) (void)hipFree(ix->d_margin);
    }
    if (ix->d_rows16) (void)hipFree(ix->d_rows16);
    ix->d_rows16 = r16;
    ix->d_margin = mg;
    ix->d_rows = rows; ix->d_nd = nd; ix->d_alpha = al; ix->d_beta = be; ix->d_row_ids = ids; ix->d_live = lv;
    ix->cap_rows = cap;
    ix->live_dirty = true;
    return VDB_OK;
}

void free_store(Index* ix) {
    if (ix->d_rows) {
        (void)hipFree(ix->d_rows); (void)hipFree(ix->d_nd); (void)hipFree(ix->d_alpha);
        (void)hipFree(ix->d_beta); (void)hipFree(ix->d_row_ids); (void)hipFree(ix->d_live);
        if (ix->d_margin) (void)hipFree(ix->d_margin);
    }
    if (ix->d_rows16) (void)hipFree(ix->d_rows16);
    ix->d_rows16 = nullptr;
    if (ix->d_sample16) (void)hipFree(ix->d_sample16);
    ix->d_sample16 = nullptr; ix->sample16_cap = 0; ix->sample16_n = ix->sample16_S = 0;
    ix->d_margin = nullptr;
    ix->d_rows = ix->d_nd = ix->d_alpha = ix->d_beta = nullptr;
    ix->d_row_ids = nullptr; ix->d_live = nullptr;
    ix->cap_rows = 0;
}

// Reset to the empty state (keeps the handle, metric and workspace).
void reset_rows(Index* ix) {
    ix->row_ids.clear(); ix->live.clear(); ix->id2row.clear(); ix->pending.clear();
    ix->n_live = 0; ix->n_uploaded = 0; ix->dim = 0; ix->ld = 0; ix->ids_monotone = true;
    ix->zero_valid = false; ix->rank_valid = false; ix->live_dirty = false;
    free_store(ix);
    if (ix->d_scalars) (void)hipMemsetAsync(ix->d_scalars, 0, 32, ix->stream);
}

void kill_row(Index* ix, uint32_t row) {
    ix->live[row >> 5] &= ~(1u << (row & 31));
    --ix->n_live;
    ix->live_dirty = true;
    ix->zero_valid = false;
}

// Appends one primary-dimension row to the host staging area.
void append_row(Index* ix, uint64_t id, const float* v) {
    uint32_t row = ix->n_rows();
    if (row && id <= ix->row_ids.back()) ix->ids_monotone = false;
    ix->row_ids.push_back(id);
    if ((row >> 5) >= ix->live.size()) ix->live.push_back(0u);
    ix->live[row >> 5] |= 1u << (row & 31);
    ++ix->n_live;
    ix->id2row[id] = row;
    size_t off = ix->pending.size();
    ix->pending.resize(off + ix->ld, 0.0f);
    memcpy(ix->pending.data() + off, v, (size_t)ix->dim * sizeof(float));
    ix->live_dirty = true;
    ix->zero_valid = false;
    ix->rank_valid = false;
}

// When the last primary row is gone but rows of another dimension remain, the lowest-id
// such dimension becomes the primary one.
void promote_misfits(Index* ix) {
    if (ix->n_live != 0 || ix->misfits.empty()) return;
    uint64_t best = ~0ull;
    for (auto& kv : ix->misfits) best = std::min(best, kv.first);
    size_t nd = ix->misfits[best].size();
    reset_rows(ix);
    if (nd == 0) return;   // zero-length vectors stay host-side only
    ix->dim = (uint32_t)nd;
    ix->ld = round_up(ix->dim, vdb::KSTAGE);
    std::vector<uint64_t> ids;
    for (auto& kv : ix->misfits)
        if (kv.second.size() == nd) ids.push_back(kv.first);
    std::sort(ids.begin(), ids.end());
    for (uint64_t id : ids) {
        append_row(ix, id, ix->misfits[id].data());
        ix->misfits.erase(id);
    }
}

int remove_id(Index* ix, uint64_t id) {
    auto it = ix->id2row.find(id);
    if (it != ix->id2row.end()) {
        kill_row(ix, it->second);
        ix->id2row.erase(it);
        if (ix->n_live == 0) {
            if (ix->misfits.empty()) reset_rows(ix);
            else promote_misfits(ix);
        }
        return VDB_OK;
    }
    ix->misfits.erase(id);   // absent id is Ok(()) (flat_index.rs:43-46)
    return VDB_OK;
}

int add_one(Index* ix, uint64_t id, const float* v, size_t dim) {
    remove_id(ix, id);   // HashMap::insert overwrites (flat_index.rs:39)
    if (ix->n_live == 0 && ix->misfits.empty() && dim > 0) {
        if (ix->dim != dim) { reset_rows(ix); }
        ix->dim = (uint32_t)dim;
        ix->ld = round_up(ix->dim, vdb::KSTAGE);
    }
    if (dim == ix->dim && dim > 0) {
        append_row(ix, id, v);
    } else {
        ix->misfits[id] = std::vector<float>(v, v + dim);
        if (ix->n_live == 0) promote_misfits(ix);
    }
    return VDB_OK;
}

int flush(Index* ix) {
    hipStream_t s = ix->stream;
    uint32_t n = ix->n_rows();
    if (n > ix->n_uploaded) {
        int rc = grow(ix, n);
        if (rc) return rc;
        uint32_t first = ix->n_uploaded, cnt = n - first;
        HIP_TRY(hipMemcpyAsync(ix->d_rows + (size_t)first * ix->ld, ix->pending.data(),
                               (size_t)cnt * ix->ld * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(ix->d_row_ids + first, ix->row_ids.data() + first, (size_t)cnt * 8,
                               hipMemcpyHostToDevice, s));
        const MarginPlan mp = margin_plan(ix);
        vdb::RowStatsParams rp{ix->d_rows, ix->ld, ix->dim, first, n, ix->metric, ix->d_nd, ix->d_alpha,
                               ix->d_beta, ix->d_scalars, ix->d_margin, mp.m_e, mp.m_n, mp.m_b, mp.beta_shrink};
        vdb::launch_row_stats(rp, s);
        if (ix->d_rows16) vdb::launch_rows_to_bf16(ix->d_rows, ix->d_rows16, ix->ld, first, n, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));   // pending is host memory about to be released
        ix->pending.clear();
        ix->pending.shrink_to_fit();
        ix->n_uploaded = n;
        ix->zero_valid = false;
    }
    if (ix->live_dirty && ix->d_live && n) {
        HIP_TRY(hipMemcpyAsync(ix->d_live, ix->live.data(), ix->live.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        ix->live_dirty = false;
    }
    return VDB_OK;
}

int ensure_zero_count(Index* ix) {
    if (ix->zero_valid) return VDB_OK;
    hipStream_t s = ix->stream;
    HIP_TRY(hipMemsetAsync(ix->d_scalars + 1, 0, 4, s));
    vdb::launch_count_zero_live(ix->d_nd, ix->d_live, ix->n_uploaded, ix->d_scalars + 1, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&ix->zero_live, ix->d_scalars + 1, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    ix->zero_valid = true;
    return VDB_OK;
}

// id rank tables so that exact-scan keys order by (distance, id) even when ids were not
// appended in increasing order.
int ensure_ranks(Index* ix) {
    if (ix->ids_monotone || ix->rank_valid) return VDB_OK;
    uint32_t n = ix->n_rows();
    std::vector<uint32_t> order(n), rank(n);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return ix->row_ids[a] != ix->row_ids[b] ? ix->row_ids[a] < ix->row_ids[b] : a < b;
    });
    for (uint32_t r = 0; r < n; ++r) rank[order[r]] = r;
    int rc;
    if ((rc = ix->d_idrank.ensure(n)) || (rc = ix->d_rank2row.ensure(n))) return rc;
    HIP_TRY(hipMemcpyAsync(ix->d_idrank.p, rank.data(), (size_t)n * 4, hipMemcpyHostToDevice, ix->stream));
    HIP_TRY(hipMemcpyAsync(ix->d_rank2row.p, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->rank_valid = true;
    return VDB_OK;
}

}  // namespace vdbi
