// vdb_host.hpp -- C++ host-side mirror of the reference's interface for the FlatIndex path, over the
// C ABI of include/vdb_flat.h.  (The reference is Rust; this image has no Rust toolchain, so the host
// side above the C ABI is C++: same names, argument meaning and error behaviour.)
//
//   Vector            src/vector.rs:9-37
//   DistanceMetric    src/distance.rs:9-16
//   VectorDbError     src/error.rs:10-31
//   Index             src/index.rs:11-35   (+ the provided search_batch of SURVEY.md 8(b))
//   GpuFlatIndex      drop-in for FlatIndex, src/flat_index.rs:12-74
//   Metadata, MetadataFilter, SearchResult, VectorStore<I>   src/storage.rs:13-348
// Header-only; link with libvdbflat.so.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "vdb_flat.h"
#include "vdb_hnsw.h"

namespace vdb_host {

enum class DistanceMetric { Euclidean = 0, Cosine = 1, DotProduct = 2 };      // distance.rs:9-16

struct VectorDbError : std::runtime_error {                                     // error.rs:10-31
    enum Kind { DimensionMismatch, VectorNotFound, InvalidVector, IndexError } kind;
    size_t expected = 0, actual = 0;
    VectorDbError(Kind k, const std::string& m, size_t e = 0, size_t a = 0)
        : std::runtime_error(m), kind(k), expected(e), actual(a) {}
    static VectorDbError dimension_mismatch(size_t e, size_t a) {
        return {DimensionMismatch, "Dimension mismatch: expected " + std::to_string(e) + ", got " + std::to_string(a), e, a};
    }
};

class Vector {                                                                  // vector.rs:9-37
public:
    Vector() = default;
    explicit Vector(std::vector<float> d) : data_(std::move(d)) {}
    Vector(std::initializer_list<float> d) : data_(d) {}
    size_t dimension() const { return data_.size(); }
    const std::vector<float>& as_slice() const { return data_; }
    bool has_same_dimension(const Vector& o) const { return dimension() == o.dimension(); }
    bool operator==(const Vector& o) const { return data_ == o.data_; }
private:
    std::vector<float> data_;
};

using Neighbor = std::pair<size_t, float>;                                      // (internal id, distance)

class Index {                                                                   // index.rs:11-35
public:
    virtual ~Index() = default;
    virtual void add(size_t id, Vector v) = 0;
    virtual void remove(size_t id) = 0;
    virtual std::vector<Neighbor> search(const Vector& query, size_t k) const = 0;
    virtual const Vector* get_vector(size_t id) const = 0;
    virtual DistanceMetric metric() const = 0;
    virtual size_t len() const = 0;
    bool is_empty() const { return len() == 0; }                                // index.rs:32-34
    // provided method: default = the sequential loop of storage.rs:306-309
    virtual std::vector<std::vector<Neighbor>> search_batch(const std::vector<std::pair<Vector, size_t>>& qs) const {
        std::vector<std::vector<Neighbor>> out;
        for (auto& q : qs) out.push_back(search(q.first, q.second));
        return out;
    }
};

class GpuFlatIndex : public Index {                                             // replaces flat_index.rs:12-74
public:
    explicit GpuFlatIndex(DistanceMetric m, int device = 0) : metric_(m) {
        check(vdb_flat_create((int)m, device, &h_));
    }
    ~GpuFlatIndex() override { vdb_flat_destroy(h_); }
    GpuFlatIndex(const GpuFlatIndex&) = delete;
    GpuFlatIndex& operator=(const GpuFlatIndex&) = delete;

    void add(size_t id, Vector v) override {                                    // flat_index.rs:38-41
        check(vdb_flat_add(h_, id, v.as_slice().data(), v.dimension()));
        vectors_[id] = std::move(v);
    }
    void remove(size_t id) override {                                           // flat_index.rs:43-46
        check(vdb_flat_remove(h_, id));
        vectors_.erase(id);
    }
    const Vector* get_vector(size_t id) const override {                        // flat_index.rs:48-50
        auto it = vectors_.find(id);
        return it == vectors_.end() ? nullptr : &it->second;
    }
    std::vector<Neighbor> search(const Vector& q, size_t k) const override {    // flat_index.rs:52-65
        return search_batch({{q, k}}).at(0);
    }
    std::vector<std::vector<Neighbor>> search_batch(const std::vector<std::pair<Vector, size_t>>& qs) const override {
        return search_batch_masked(qs, nullptr, 0);
    }
    // BASELINE config 4: pre-filter as a bitmask over internal ids (bit i set = id i eligible)
    std::vector<std::vector<Neighbor>> search_batch_masked(const std::vector<std::pair<Vector, size_t>>& qs,
                                                           const uint64_t* id_mask, size_t mask_bits) const {
        std::vector<std::vector<Neighbor>> out(qs.size());
        if (qs.empty()) return out;
        const size_t dim = qs[0].first.dimension();
        bool ragged = false;
        for (auto& q : qs) ragged |= q.first.dimension() != dim;
        if (ragged) {                                                            // one call per query, as storage.rs:306-309
            for (size_t b = 0; b < qs.size(); ++b) out[b] = search_batch_masked({qs[b]}, id_mask, mask_bits)[0];
            return out;
        }
        std::vector<float> flat;
        std::vector<size_t> ks;
        size_t kmax = 1;
        for (auto& q : qs) {
            flat.insert(flat.end(), q.first.as_slice().begin(), q.first.as_slice().end());
            ks.push_back(q.second);
            kmax = std::max(kmax, q.second);
        }
        std::vector<uint64_t> ids(qs.size() * kmax);
        std::vector<float> ds(qs.size() * kmax);
        std::vector<size_t> cnt(qs.size());
        check(vdb_flat_search_batch(h_, flat.data(), qs.size(), dim, ks.data(), 0, id_mask, mask_bits, kmax, ids.data(),
                                    ds.data(), cnt.data()));
        for (size_t b = 0; b < qs.size(); ++b)
            for (size_t i = 0; i < cnt[b]; ++i) out[b].emplace_back((size_t)ids[b * kmax + i], ds[b * kmax + i]);
        return out;
    }
    DistanceMetric metric() const override { return metric_; }
    size_t len() const override { return vdb_flat_len(h_); }
    void add_bulk(const float* rows, size_t n, size_t dim, uint64_t first_id) {
        check(vdb_flat_add_bulk(h_, nullptr, first_id, rows, n, dim));
        for (size_t i = 0; i < n; ++i) vectors_[first_id + i] = Vector(std::vector<float>(rows + i * dim, rows + (i + 1) * dim));
    }
    vdb_flat_index* handle() const { return h_; }
    // no reference counterpart: tier selection (results are identical either way)
    void set_screen(int mode) { check(vdb_flat_set_screen(h_, mode)); }
    void set_tiers(unsigned flags) { check(vdb_flat_set_tiers(h_, flags)); }

private:
    static void check(int rc) {
        if (rc == VDB_OK) return;
        char buf[512];
        size_t e = 0, a = 0;
        vdb_last_error(buf, sizeof buf, &e, &a);
        switch (rc) {
        case VDB_ERR_DIMENSION_MISMATCH: throw VectorDbError::dimension_mismatch(e, a);
        case VDB_ERR_INVALID_VECTOR: throw VectorDbError(VectorDbError::InvalidVector, buf);
        default: throw VectorDbError(VectorDbError::IndexError, std::string("Index error: ") + buf);
        }
    }
    vdb_flat_index* h_ = nullptr;
    DistanceMetric metric_;
    std::unordered_map<size_t, Vector> vectors_;                                 // host copy for get_vector() borrows
};

struct HnswParams {                                                             // graph.rs:19-59
    size_t m = 16, ef_construction = 200, ef_search = 50;
    static HnswParams make(size_t m, size_t efc, size_t efs) { HnswParams p; p.m = m; p.ef_construction = efc; p.ef_search = efs; return p; }
};

// Drop-in for HnswIndex (src/hnsw/mod.rs:14-82) over include/vdb_hnsw.h: the library runs the reference's graph
// traversal, every distance is evaluated on the GPU.  `seed` replaces StdRng::from_entropy() (graph.rs:101).
class GpuHnswIndex : public Index {
public:
    explicit GpuHnswIndex(DistanceMetric m, HnswParams p = HnswParams(), uint64_t seed = 1, int device = 0) : metric_(m) {
        check(vdb_hnsw_create((int)m, p.m, p.ef_construction, p.ef_search, seed, device, &h_));
    }
    ~GpuHnswIndex() override { vdb_hnsw_destroy(h_); }
    GpuHnswIndex(const GpuHnswIndex&) = delete;
    GpuHnswIndex& operator=(const GpuHnswIndex&) = delete;
    GpuHnswIndex(GpuHnswIndex&& o) noexcept : h_(o.h_), metric_(o.metric_), vectors_(std::move(o.vectors_)) { o.h_ = nullptr; }

    void add(size_t id, Vector v) override {                                    // mod.rs:57-59
        check(vdb_hnsw_add(h_, id, v.as_slice().data(), v.dimension(), -1));
        vectors_[id] = std::move(v);
    }
    void remove(size_t id) override {                                           // mod.rs:61-63
        check(vdb_hnsw_remove(h_, id));
        vectors_.erase(id);
    }
    const Vector* get_vector(size_t id) const override {                        // mod.rs:65-67
        auto it = vectors_.find(id);
        return it == vectors_.end() ? nullptr : &it->second;
    }
    std::vector<Neighbor> search(const Vector& q, size_t k) const override { return search_with_ef(q, k, 50); }   // mod.rs:69-72
    std::vector<Neighbor> search_with_ef(const Vector& q, size_t k, size_t ef) const {                            // mod.rs:45-53
        std::vector<uint64_t> ids(std::max<size_t>(k, 1));
        std::vector<float> ds(ids.size());
        size_t n = 0;
        check(vdb_hnsw_search_batch(h_, q.as_slice().data(), 1, q.dimension(), k, ef, ids.data(), ds.data(), &n));
        std::vector<Neighbor> out;
        for (size_t i = 0; i < n; ++i) out.emplace_back((size_t)ids[i], ds[i]);
        return out;
    }
    // all queries walk the graph in lockstep: one GPU launch per traversal round for all their candidate lists
    std::vector<std::vector<Neighbor>> search_batch(const std::vector<std::pair<Vector, size_t>>& qs) const override {
        std::vector<std::vector<Neighbor>> out(qs.size());
        if (qs.empty()) return out;
        const size_t dim = qs[0].first.dimension();
        size_t kmax = 1;
        std::vector<float> flat;
        for (auto& q : qs) {
            if (q.first.dimension() != dim) return Index::search_batch(qs);     // ragged batch: the sequential loop
            flat.insert(flat.end(), q.first.as_slice().begin(), q.first.as_slice().end());
            kmax = std::max(kmax, q.second);
        }
        std::vector<uint64_t> ids(qs.size() * kmax);
        std::vector<float> ds(qs.size() * kmax);
        std::vector<size_t> cnt(qs.size());
        check(vdb_hnsw_search_batch(h_, flat.data(), qs.size(), dim, kmax, 50, ids.data(), ds.data(), cnt.data()));
        for (size_t b = 0; b < qs.size(); ++b)
            for (size_t i = 0; i < std::min(cnt[b], qs[b].second); ++i) out[b].emplace_back((size_t)ids[b * kmax + i], ds[b * kmax + i]);
        return out;
    }
    void build_batch(const std::vector<std::pair<size_t, Vector>>& vs) { for (auto& v : vs) add(v.first, v.second); }   // mod.rs:37-42
    DistanceMetric metric() const override { return metric_; }
    size_t len() const override { return vdb_hnsw_len(h_); }
    vdb_hnsw_index* handle() const { return h_; }

private:
    static void check(int rc) {
        if (rc == VDB_OK) return;
        char buf[512];
        size_t e = 0, a = 0;
        vdb_last_error(buf, sizeof buf, &e, &a);
        switch (rc) {
        case VDB_ERR_DIMENSION_MISMATCH: throw VectorDbError::dimension_mismatch(e, a);
        case VDB_ERR_INVALID_VECTOR: throw VectorDbError(VectorDbError::InvalidVector, buf);
        default: throw VectorDbError(VectorDbError::IndexError, std::string("Index error: ") + buf);
        }
    }
    vdb_hnsw_index* h_ = nullptr;
    DistanceMetric metric_;
    std::unordered_map<size_t, Vector> vectors_;
};

struct SearchResult { std::string id; float distance; };                        // storage.rs:13-16

class Metadata {                                                                // storage.rs:19-42
public:
    void insert(std::string k, std::string v) { fields_[std::move(k)] = std::move(v); }
    const std::string* get(const std::string& k) const {
        auto it = fields_.find(k);
        return it == fields_.end() ? nullptr : &it->second;
    }
    const std::map<std::string, std::string>& fields() const { return fields_; }
private:
    std::map<std::string, std::string> fields_;
};

struct MetadataFilter {                                                         // storage.rs:45-71
    enum Op { Eq, Ne, Exists, And, Or } op;
    std::string field, value;
    std::vector<MetadataFilter> filters;
    static MetadataFilter eq(std::string f, std::string v) { return {Eq, std::move(f), std::move(v), {}}; }
    static MetadataFilter ne(std::string f, std::string v) { return {Ne, std::move(f), std::move(v), {}}; }
    static MetadataFilter exists(std::string f) { return {Exists, std::move(f), "", {}}; }
    static MetadataFilter all(std::vector<MetadataFilter> fs) { return {And, "", "", std::move(fs)}; }
    static MetadataFilter any(std::vector<MetadataFilter> fs) { return {Or, "", "", std::move(fs)}; }
    bool matches(const Metadata& m) const {                                     // storage.rs:62-70
        const std::string* v = m.get(field);
        switch (op) {
        case Eq: return v && *v == value;
        case Ne: return !(v && *v == value);
        case Exists: return v != nullptr;
        case And: return std::all_of(filters.begin(), filters.end(), [&](const MetadataFilter& f) { return f.matches(m); });
        case Or: return std::any_of(filters.begin(), filters.end(), [&](const MetadataFilter& f) { return f.matches(m); });
        }
        return false;
    }
};

template <class I> class VectorStore {                                          // storage.rs:83-348
public:
    explicit VectorStore(std::unique_ptr<I> index) : index_(std::move(index)) {} // with_index, storage.rs:118-127

    void insert(const std::string& id, Vector v) { insert_with_metadata(id, std::move(v), Metadata()); }
    void insert_with_metadata(const std::string& id, Vector v, Metadata meta) {  // storage.rs:135-172
        const size_t dim = v.dimension();
        if (dimension_) {
            if (dim != *dimension_) throw VectorDbError::dimension_mismatch(*dimension_, dim);
        } else {
            dimension_ = dim;
        }
        auto old = id_to_internal_.find(id);
        if (old != id_to_internal_.end()) {
            index_->remove(old->second);
            metadata_.erase(old->second);
            internal_to_id_.erase(old->second);
        }
        const size_t internal = next_id_++;
        index_->add(internal, std::move(v));
        id_to_internal_[id] = internal;
        internal_to_id_[internal] = id;
        metadata_[internal] = std::move(meta);
    }
    Vector remove(const std::string& id) {                                       // VectorStore::delete, storage.rs:175-192
        auto it = id_to_internal_.find(id);
        if (it == id_to_internal_.end()) throw VectorDbError(VectorDbError::VectorNotFound, "Vector not found: " + id);
        const size_t internal = it->second;
        id_to_internal_.erase(it);
        const Vector* p = index_->get_vector(internal);
        Vector v = p ? *p : Vector();
        internal_to_id_.erase(internal);
        metadata_.erase(internal);
        index_->remove(internal);
        return v;
    }
    const Vector* get(const std::string& id) const {                             // storage.rs:195-198
        auto it = id_to_internal_.find(id);
        return it == id_to_internal_.end() ? nullptr : index_->get_vector(it->second);
    }
    size_t len() const { return index_->len(); }
    bool is_empty() const { return index_->is_empty(); }
    std::optional<size_t> dimension() const { return dimension_; }
    I& index() { return *index_; }

    std::vector<SearchResult> search(const Vector& q, size_t k) const {          // storage.rs:217-245
        if (is_empty()) return {};
        check_dim(q);
        return map(index_->search(q, k));
    }
    std::vector<SearchResult> search_with_filter(const Vector& q, size_t k, const MetadataFilter& f) const {
        if (is_empty()) return {};                                               // storage.rs:249-290
        check_dim(q);
        const size_t fetch_k = std::min(std::max(k * 3, k), len());
        return post_filter(index_->search(q, fetch_k), k, f);
    }
    std::vector<std::vector<SearchResult>> search_batch(const std::vector<std::pair<Vector, size_t>>& qs) const {
        std::vector<std::vector<SearchResult>> out(qs.size());                   // storage.rs:302-310, ONE index call
        if (is_empty()) return out;
        for (auto& q : qs) check_dim(q.first);
        auto res = index_->search_batch(qs);
        for (size_t b = 0; b < qs.size(); ++b) out[b] = map(res[b]);
        return out;
    }
    std::vector<std::vector<SearchResult>> search_batch_with_filter(const std::vector<std::pair<Vector, size_t>>& qs,
                                                                    const MetadataFilter& f) const {
        std::vector<std::vector<SearchResult>> out(qs.size());                   // storage.rs:313-322
        if (is_empty()) return out;
        std::vector<std::pair<Vector, size_t>> fetch;
        for (auto& q : qs) {
            check_dim(q.first);
            fetch.emplace_back(q.first, std::min(std::max(q.second * 3, q.second), len()));
        }
        auto res = index_->search_batch(fetch);
        for (size_t b = 0; b < qs.size(); ++b) out[b] = post_filter(res[b], qs[b].second, f);
        return out;
    }
    // the MetadataFilter compiled to a bitmask over internal ids (device pre-filter, BASELINE config 4)
    std::vector<uint64_t> compile_filter(const MetadataFilter& f, size_t* bits) const {
        *bits = std::max<size_t>(next_id_, 1);
        std::vector<uint64_t> mask((*bits + 63) / 64, 0);
        for (auto& kv : metadata_)
            if (f.matches(kv.second)) mask[kv.first >> 6] |= 1ull << (kv.first & 63);
        return mask;
    }

private:
    void check_dim(const Vector& q) const {
        if (dimension_ && q.dimension() != *dimension_) throw VectorDbError::dimension_mismatch(*dimension_, q.dimension());
    }
    std::vector<SearchResult> map(const std::vector<Neighbor>& rs) const {        // storage.rs:234-242
        std::vector<SearchResult> out;
        for (auto& r : rs) {
            auto it = internal_to_id_.find(r.first);
            if (it != internal_to_id_.end()) out.push_back({it->second, r.second});
        }
        return out;
    }
    std::vector<SearchResult> post_filter(const std::vector<Neighbor>& rs, size_t k, const MetadataFilter& f) const {
        std::vector<SearchResult> out;                                           // storage.rs:272-287
        for (auto& r : rs) {
            auto id = internal_to_id_.find(r.first);
            auto meta = metadata_.find(r.first);
            if (id == internal_to_id_.end() || meta == metadata_.end()) continue;
            if (f.matches(meta->second)) {
                out.push_back({id->second, r.second});
                if (out.size() == k) break;
            }
        }
        return out;
    }
    std::unique_ptr<I> index_;
    std::unordered_map<std::string, size_t> id_to_internal_;
    std::unordered_map<size_t, std::string> internal_to_id_;
    std::unordered_map<size_t, Metadata> metadata_;
    size_t next_id_ = 0;
    std::optional<size_t> dimension_;
};

inline VectorStore<GpuFlatIndex> make_store(DistanceMetric m, int device = 0) {  // VectorStore::new, storage.rs:99-101
    return VectorStore<GpuFlatIndex>(std::make_unique<GpuFlatIndex>(m, device));
}

}  // namespace vdb_host
