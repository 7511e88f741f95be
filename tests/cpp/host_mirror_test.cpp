// Replays the reference's own unit/integration tests for the FlatIndex path through the C++ host mirror
// (vdb_host.hpp) and the C ABI.  Inputs and expected results are those of the cited reference tests.
// Needs an MI355X; exits non-zero on the first failed check.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "vdb_host.hpp"
using namespace vdb_host;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

int main() {
    {   // src/flat_index.rs:81-93  test_flat_index_basic
        GpuFlatIndex ix(DistanceMetric::Euclidean);
        ix.add(0, Vector{1.f, 0.f, 0.f});
        ix.add(1, Vector{0.f, 1.f, 0.f});
        ix.add(2, Vector{1.f, 1.f, 0.f});
        auto r = ix.search(Vector{1.f, 0.f, 0.f}, 2);
        CHECK(r.size() == 2 && r[0].first == 0 && r[0].second < 1e-6f);
    }
    {   // src/flat_index.rs:96-114  get_vector / remove
        GpuFlatIndex ix(DistanceMetric::Euclidean);
        Vector v{1.f, 2.f, 3.f};
        ix.add(0, v);
        CHECK(ix.get_vector(0) && *ix.get_vector(0) == v && ix.get_vector(99) == nullptr);
        ix.add(1, Vector{0.f, 1.f, 0.f});
        CHECK(ix.len() == 2);
        ix.remove(0);
        CHECK(ix.len() == 1);
    }
    {   // src/distance.rs:81-133 via one-row indexes: the distance IS the search result
        auto dist = [](DistanceMetric m, Vector a, Vector b) {
            GpuFlatIndex ix(m);
            ix.add(0, b);
            return ix.search(a, 1).at(0).second;
        };
        CHECK(std::fabs(dist(DistanceMetric::Euclidean, {1, 2, 3}, {4, 5, 6}) - 5.196152f) < 1e-5f);
        CHECK(std::fabs(dist(DistanceMetric::Euclidean, {1, 2, 3}, {1, 2, 3})) < 1e-6f);
        CHECK(std::fabs(-dist(DistanceMetric::DotProduct, {1, 2, 3}, {4, 5, 6}) - 32.0f) < 1e-6f);
        CHECK(std::fabs(dist(DistanceMetric::Cosine, {1, 0, 0}, {1, 0, 0})) < 1e-6f);
        CHECK(std::fabs(dist(DistanceMetric::Cosine, {1, 0, 0}, {0, 1, 0}) - 1.0f) < 1e-6f);
        CHECK(std::fabs(dist(DistanceMetric::Cosine, {1, 0, 0}, {-1, 0, 0}) - 2.0f) < 1e-6f);
        try { dist(DistanceMetric::Euclidean, {1, 2}, {1, 2, 3}); CHECK(false); }   // distance.rs:136-143
        catch (const VectorDbError& e) { CHECK(e.kind == VectorDbError::DimensionMismatch && e.expected == 2 && e.actual == 3); }
    }
    {   // src/storage.rs:365-404  dimension consistency, delete, search, empty store
        auto st = make_store(DistanceMetric::Euclidean);
        CHECK(st.search(Vector{1, 2, 3}, 5).empty());
        st.insert("v1", Vector{1, 0, 0});
        try { st.insert("bad", Vector{1, 2}); CHECK(false); }
        catch (const VectorDbError& e) { CHECK(e.kind == VectorDbError::DimensionMismatch); }
        st.insert("v2", Vector{0, 1, 0});
        st.insert("v3", Vector{1, 1, 0});
        auto r = st.search(Vector{1, 0, 0}, 2);
        CHECK(r.size() == 2 && r[0].id == "v1" && std::fabs(r[0].distance) < 1e-6f);
        Vector gone = st.remove("v3");
        CHECK(gone == (Vector{1, 1, 0}) && st.len() == 2 && st.get("v3") == nullptr && st.get("v1") != nullptr);
        try { st.remove("nope"); CHECK(false); } catch (const VectorDbError& e) { CHECK(e.kind == VectorDbError::VectorNotFound); }
    }
    {   // src/storage.rs:578-630, :733-755  search_with_filter; :680-730 batch and batch + filter
        auto st = make_store(DistanceMetric::Euclidean);
        Metadata red, blue;
        red.insert("color", "red");
        blue.insert("color", "blue");
        st.insert_with_metadata("v1", Vector{1.f, 0.f, 0.f}, red);
        st.insert_with_metadata("v2", Vector{0.9f, 0.1f, 0.f}, blue);
        st.insert_with_metadata("v3", Vector{0.f, 1.f, 0.f}, red);
        auto f = MetadataFilter::eq("color", "red");
        auto r = st.search_with_filter(Vector{1, 0, 0}, 10, f);
        CHECK(r.size() == 2 && (r[0].id == "v1" || r[0].id == "v3") && (r[1].id == "v1" || r[1].id == "v3"));
        CHECK(st.search_with_filter(Vector{1, 0, 0}, 10, MetadataFilter::eq("color", "green")).empty());
        auto b = st.search_batch({{Vector{1, 0, 0}, 1}, {Vector{0, 1, 0}, 1}});
        CHECK(b.size() == 2 && b[0][0].id == "v1" && b[1][0].id == "v3");
        auto bf = st.search_batch_with_filter({{Vector{1, 0, 0}, 10}, {Vector{0.9f, 0.1f, 0}, 10}}, MetadataFilter::eq("color", "blue"));
        CHECK(bf[0].size() == 1 && bf[0][0].id == "v2" && bf[1].size() == 1 && bf[1][0].id == "v2");
        // device pre-filter: the reference's post-filter result is a prefix of it
        size_t bits = 0;
        auto mask = st.compile_filter(f, &bits);
        auto pre = st.index().search_batch_masked({{Vector{1, 0, 0}, 10}}, mask.data(), bits);
        CHECK(pre[0].size() == 2 && pre[0][0].first == 0 && pre[0][1].first == 2);
        // filters: storage.rs:456-575
        Metadata m;
        m.insert("color", "red");
        m.insert("size", "large");
        CHECK(MetadataFilter::ne("color", "blue").matches(m) && !MetadataFilter::ne("color", "red").matches(m));
        CHECK(MetadataFilter::exists("color").matches(m) && !MetadataFilter::exists("weight").matches(m));
        CHECK(MetadataFilter::all({MetadataFilter::eq("color", "red"), MetadataFilter::eq("size", "large")}).matches(m));
        CHECK(!MetadataFilter::any({MetadataFilter::eq("color", "green"), MetadataFilter::eq("color", "blue")}).matches(m));
    }
    {   // tests/integration_test.rs:6-47
        auto st = make_store(DistanceMetric::Euclidean);
        st.insert("v1", Vector{1, 0, 0});
        st.insert("v2", Vector{0, 1, 0});
        st.insert("v3", Vector{0, 0, 1});
        CHECK(st.len() == 3);
        auto r = st.search(Vector{1.f, 0.1f, 0.f}, 2);
        CHECK(r.size() == 2 && r[0].id == "v1");
        st.remove("v2");
        CHECK(st.len() == 2);
        for (auto m : {DistanceMetric::Euclidean, DistanceMetric::Cosine, DistanceMetric::DotProduct}) {
            auto s2 = make_store(m);
            s2.insert("v1", Vector{1, 2, 3});
            auto r2 = s2.search(Vector{1, 2, 3}, 1);
            CHECK(r2.size() == 1 && r2[0].id == "v1");
        }
    }
    {   // zero-norm row under Cosine fails every search (distance.rs:51-55)
        GpuFlatIndex ix(DistanceMetric::Cosine);
        ix.add(0, Vector{1, 0});
        ix.add(1, Vector{0, 0});
        try { ix.search(Vector{1, 1}, 1); CHECK(false); } catch (const VectorDbError& e) { CHECK(e.kind == VectorDbError::InvalidVector); }
    }
    {   // src/hnsw/mod.rs:88-108  HnswIndex through the trait; get_vector
        GpuHnswIndex ix(DistanceMetric::Euclidean);
        ix.add(0, Vector{1.f, 0.f, 0.f});
        ix.add(1, Vector{0.f, 1.f, 0.f});
        ix.add(2, Vector{1.f, 1.f, 0.f});
        auto r = ix.search(Vector{1.f, 0.f, 0.f}, 2);
        CHECK(r.size() == 2 && r[0].first == 0 && r[0].second < 1e-5f);
        CHECK(ix.get_vector(0) && *ix.get_vector(0) == (Vector{1.f, 0.f, 0.f}) && ix.get_vector(99) == nullptr);
    }
    {   // src/hnsw/graph.rs:488-537  search_knn, remove, remove the entry point
        GpuHnswIndex g(DistanceMetric::Euclidean, HnswParams::make(4, 32, 16));
        for (size_t i = 0; i < 5; ++i) g.add(i, Vector{(float)i, 0.f});
        auto r = g.search_with_ef(Vector{0.5f, 0.f}, 2, 16);
        CHECK(r.size() == 2 && ((r[0].first == 0 && r[1].first == 1) || (r[0].first == 1 && r[1].first == 0)));
        GpuHnswIndex h(DistanceMetric::Euclidean, HnswParams::make(4, 32, 16));
        h.add(0, Vector{1.f, 0.f});
        h.add(1, Vector{0.f, 1.f});
        h.remove(0);
        CHECK(h.len() == 1 && h.search_with_ef(Vector{0.f, 1.f}, 1, 16).at(0).first == 1);
        uint64_t ep = 0;
        h.add(2, Vector{1.f, 1.f});
        CHECK(vdb_hnsw_entry_point(h.handle(), &ep, nullptr) == 1);
        h.remove((size_t)ep);
        CHECK(h.len() == 1 && !h.search_with_ef(Vector{0.f, 1.f}, 1, 16).empty());
    }
    {   // src/hnsw/mod.rs:111-153  HnswIndex behind the VectorStore
        VectorStore<GpuHnswIndex> st(std::make_unique<GpuHnswIndex>(DistanceMetric::Euclidean, HnswParams::make(4, 32, 16)));
        st.insert("v1", Vector{1, 0, 0});
        st.insert("v2", Vector{0, 1, 0});
        st.insert("v3", Vector{0, 0, 1});
        auto r = st.search(Vector{1.f, 0.1f, 0.f}, 2);
        CHECK(r.size() == 2 && r[0].id == "v1");
        st.remove("v1");
        CHECK(st.len() == 2);
    }
    std::puts("host mirror ok");
    return 0;
}
