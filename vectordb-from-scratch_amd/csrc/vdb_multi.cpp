// vdb_multi.cpp -- ONE vdb_flat_index over several GPUs in ONE process (vdb_flat_create_sharded, include/vdb_flat.h).
//
// Why it exists: the reference's seam is a single in-process object -- `VectorStore<I: Index>` (src/storage.rs:83,:116-127)
// held by one server process behind one RwLock (src/server/mod.rs:13-16), batches arriving through
// VectorStore::search_batch (src/storage.rs:302-310).  BASELINE.json's north star shards the index by row across the 8 GPUs
// of a node "behind this repo's Index trait", so the object that implements `Index` has to own the shards itself.  (The
// process-per-GPU form of the same exchange is vdb_shard.cpp; bench.py's torchrun contract uses that one.)
//
// Shape: a parent handle with G children, each an ordinary single-GPU vdb_flat_index (vdb_store.cpp + vdb_search.cpp) on
// its own device, stream and host worker thread -- enqueueing a search costs ~40 us of host time per shard, so the shards
// are driven concurrently, never in a loop.  A batched search:
//     every shard: queries copied to its device (hipMemcpyPeerAsync), FIRST tier of the local pipeline enqueued, results
//                  written straight into the shard's packed block  ids u64[nq*k] | dists f32[nq*k] | counts u32[nq] | status
//     exchange 1:  RCCL mode: one grouped ncclAllGather over in-process communicators (ncclCommInitAll);
//                  peer mode: every shard's stream copies its block into devices[0]'s gather buffer
//     merge on devices[0] by (distance, id) into the caller's buffers + MAX of the status words; ONE host sync
//     every shard: second half of the local search (fallback tiers for the queries its first tier could not certify)
//     exchange 2 + merge, iff the reduced status said some shard rewrote its block
// Distances are exact and bit-identical across shards, so the merged result equals the single-GPU index's, bit for bit.
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <exception>
#include <functional>
#include <stdexcept>
#include <thread>
#include <unordered_set>

#include "../../include/vdb_shard.h"
#include "vdb_index.h"
#include "vdb_rccl.h"

namespace vdbi {

namespace {

constexpr uint32_t CODE_ERR_BASE = 1000;     // status word of a shard whose local search failed: 1000 + vdb_status (survives the MAX with VDB_PENDING_HOST)

inline void cpu_relax() { __builtin_ia32_pause(); }
inline uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// G - 1 worker threads, one per shard beyond the first (the caller's thread drives shard 0).  run(f) executes f(g) for
// every shard and returns when all are done.  Workers spin for a while after a job (a burst of searches finds them
// awake: a condition-variable wake-up costs 20-50 us, a C2 step on 8 GPUs is ~200 us) and then sleep.
class Gang {
public:
    ~Gang() { stop(); }
    void start(int n, std::function<void(int)> init) {
        n_ = n;
        for (int g = 1; g < n; ++g)
            th_.emplace_back([this, g, init] {
                init(g);
                uint64_t seen = 0;
                for (;;) {
                    const uint64_t t_idle = now_ns();
                    int it = 0;
                    while (epoch_.load(std::memory_order_acquire) == seen && !stop_.load(std::memory_order_acquire)) {
                        cpu_relax();
                        if ((++it & 255) == 0 && now_ns() - t_idle > SPIN_NS) {
                            std::unique_lock<std::mutex> lk(mu_);
                            cv_go_.wait(lk, [&] { return epoch_.load() != seen || stop_.load(); });
                        }
                    }
                    if (stop_.load(std::memory_order_acquire)) return;
                    seen = epoch_.load(std::memory_order_acquire);
                    try { (*job_)(g); } catch (...) { threw_.store(true, std::memory_order_release); }   // (re-raised by run() on the caller's thread)
                    if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                        std::lock_guard<std::mutex> lk(mu_);
                        cv_done_.notify_one();
                    }
                }
            });
    }
    void run(const std::function<void(int)>& f) {
        if (n_ > 1) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                job_ = &f;
                remaining_.store(n_ - 1, std::memory_order_release);
                epoch_.fetch_add(1, std::memory_order_acq_rel);
            }
            cv_go_.notify_all();
        }
        // an exception (std::bad_alloc from a message string, say) must not leave run() while workers still execute `f`: it is
        // held until every shard is done and re-raised here, where the ABI boundary's `guarded` turns it into a status
        std::exception_ptr ex;
        try { f(0); } catch (...) { ex = std::current_exception(); }
        if (n_ > 1) {
            const uint64_t t0 = now_ns();
            int it = 0;
            while (remaining_.load(std::memory_order_acquire) != 0) {
                cpu_relax();
                if ((++it & 255) == 0 && now_ns() - t0 > SPIN_NS) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_done_.wait(lk, [&] { return remaining_.load() == 0; });
                }
            }
        }
        if (ex) std::rethrow_exception(ex);
        if (threw_.exchange(false)) throw std::runtime_error("internal error: a shard worker raised a C++ exception");
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_.store(true);
        }
        cv_go_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
        th_.clear();
    }

private:
    static constexpr uint64_t SPIN_NS = 2000000;       // 2 ms
    int n_ = 1;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_go_, cv_done_;
    std::atomic<uint64_t> epoch_{0};
    std::atomic<int> remaining_{0};
    std::atomic<bool> stop_{false}, threw_{false};
    const std::function<void(int)>* job_ = nullptr;
};

}  // namespace

}  // namespace vdbi

struct vdb_multi {
    int G = 0, home = 0;
    std::vector<int> dev;
    std::vector<vdb_flat_index*> sh;
    bool distinct = true;                      // no device listed twice (RCCL needs that)
    int exchange = VDB_EXCHANGE_RCCL;
    std::vector<void*> comm;                   // in-process RCCL communicators, created at the first RCCL exchange
    int comm_world = 0;
    struct PerShard {
        hipStream_t stream = nullptr;
        hipEvent_t ev_done = nullptr;
        float* d_q = nullptr; size_t q_cap = 0;                 // the queries on this shard's device (shards off the home device)
        uint64_t* d_mask = nullptr; size_t mask_cap = 0;        // the id mask, likewise
        int32_t* d_pack = nullptr; size_t pack_words = 0;       // this shard's packed partial results
        int32_t* d_gath = nullptr; size_t gath_words = 0;       // [G][words]: RCCL receive buffer (every shard) / peer target (shard 0)
        int rc = VDB_OK; bool begun = false; int changed = 0;
        std::string msg; size_t e_exp = 0, e_act = 0;
    };
    std::vector<PerShard> ps;
    uint32_t* d_status = nullptr; uint32_t* h_status = nullptr;     // devices[0]
    // host-pointer entry point: staging on devices[0]
    vdbi::DevBuf<float> w_qin, w_outd; vdbi::DevBuf<uint64_t> w_outi, w_mask; vdbi::DevBuf<uint32_t> w_outc;
    uint64_t stats[8] = {0};
    vdbi::Gang gang;
};

namespace vdbi {

namespace {

void capture_error(vdb_multi::PerShard& p, int rc) {
    p.rc = rc;
    last_error(&p.msg, &p.e_exp, &p.e_act);
}
int report(const vdb_multi::PerShard& p) {
    if (p.rc == VDB_ERR_DIMENSION_MISMATCH) return fail_dim(p.e_exp, p.e_act);
    return fail(p.rc, "%s", p.msg.c_str());
}
// the first shard (lowest index) that failed decides the error of the call
int first_error(vdb_multi* M) {
    for (auto& p : M->ps) if (p.rc != VDB_OK) return report(p);
    return VDB_OK;
}

int nccl_fail(const char* what, int rc) {
    const vdb_rccl::Rccl* r = vdb_rccl::rccl();
    return fail(VDB_ERR_DEVICE, "RCCL %s failed: %d (%s)", what, rc, (r && r->error_string) ? r->error_string(rc) : "?");
}

int ensure_comms(vdb_multi* M) {
    if (!M->comm.empty()) return VDB_OK;
    const vdb_rccl::Rccl* r = vdb_rccl::rccl();
    if (!r) return fail(VDB_ERR_DEVICE, "%s", vdb_rccl::why());
    if (!M->distinct) return fail(VDB_ERR_INVALID_ARGUMENT, "the RCCL exchange needs distinct devices (a device is listed twice: use VDB_EXCHANGE_PEER)");
    std::vector<void*> comm((size_t)M->G, nullptr);
    int rc = r->comm_init_all(comm.data(), M->G, M->dev.data());
    if (rc) return nccl_fail("ncclCommInitAll", rc);
    int cnt = 0;
    rc = r->comm_count(comm[0], &cnt);
    if (rc || cnt != M->G) {
        for (void* c : comm) if (c) (void)r->comm_destroy(c);
        return rc ? nccl_fail("ncclCommCount", rc) : fail(VDB_ERR_DEVICE, "RCCL communicator has %d ranks, %d requested", cnt, M->G);
    }
    M->comm.swap(comm);
    M->comm_world = cnt;
    HIP_TRY(hipSetDevice(M->home));
    return VDB_OK;
}

// (re)allocates a device buffer of the shard on ITS device; the caller restores the current device
template <class T> int ensure_on(int device, T*& p, size_t& cap, size_t want) {
    if (want <= cap) return VDB_OK;
    HIP_TRY(hipSetDevice(device));
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t n = want + want / 2;
    HIP_TRY(hipMalloc((void**)&p, n * sizeof(T)));
    cap = n;
    return VDB_OK;
}

bool holds(Index* c, uint64_t id) {
    std::lock_guard<std::mutex> g(c->mu);
    return c->id2row.count(id) != 0 || c->misfits.count(id) != 0;
}
bool is_fresh(Index* c) {
    std::lock_guard<std::mutex> g(c->mu);
    return c->id2row.empty() && c->misfits.empty();
}

int add_one_routed(vdb_multi* M, uint64_t id, const float* v, size_t dim) {
    int target = -1;
    for (int g = 0; g < M->G && target < 0; ++g) if (holds(M->sh[g], id)) target = g;     // HashMap::insert overwrites (flat_index.rs:39): in place
    if (target < 0) {
        size_t best = ~(size_t)0;
        for (int g = 0; g < M->G; ++g) { const size_t l = vdb_flat_len(M->sh[g]); if (l < best) { best = l; target = g; } }
    }
    return vdb_flat_add(M->sh[target], id, v, dim);
}

// exchange of the packed blocks + merge on the home device + reduced status on the host: ONE host sync
int exchange(vdb_multi* M, size_t words, size_t nq, size_t k, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
             uint32_t* worst) {
    HIP_TRY(hipSetDevice(M->home));
    hipStream_t s0 = M->ps[0].stream;
    if (M->exchange == VDB_EXCHANGE_RCCL) {
        const vdb_rccl::Rccl* r = vdb_rccl::rccl();
        int rc = r->group_start();
        if (rc) return nccl_fail("ncclGroupStart", rc);
        for (int g = 0; g < M->G; ++g) {
            rc = r->all_gather(M->ps[g].d_pack, M->ps[g].d_gath, words * 4, /* ncclInt8 */ 0, M->comm[g], M->ps[g].stream);
            if (rc) { (void)r->group_end(); return nccl_fail("ncclAllGather", rc); }
        }
        rc = r->group_end();
        if (rc) return nccl_fail("ncclGroupEnd", rc);
        HIP_TRY(hipSetDevice(M->home));
    } else {
        // the copies were enqueued by the shards themselves (behind their searches); the home stream waits for them
        for (int g = 1; g < M->G; ++g) HIP_TRY(hipStreamWaitEvent(s0, M->ps[g].ev_done, 0));
    }
    vdb::launch_merge_packed(M->ps[0].d_gath, words, (uint32_t)M->G, (uint32_t)nq, (uint32_t)k, d_out_ids, d_out_dists, d_out_counts,
                             M->d_status, s0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(M->h_status, M->d_status, 4, hipMemcpyDeviceToHost, s0));
    HIP_TRY(hipStreamSynchronize(s0));
    *worst = *M->h_status;
    M->stats[0]++;
    return VDB_OK;
}

}  // namespace

// ------------------------------------------------------------------ lifetime
int multi_create(int metric, const int* devices, size_t n, vdb_flat_index** out) {
    if (!out) return fail(VDB_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (!devices || n == 0) return fail(VDB_ERR_INVALID_ARGUMENT, "at least one device is required");
    if (n > 64) return fail(VDB_ERR_INVALID_ARGUMENT, "at most 64 shards");
    if (metric < 0 || metric > 2) return fail(VDB_ERR_INVALID_ARGUMENT, "unknown metric %d", metric);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VDB_ERR_DEVICE, "no HIP device available: this engine has no CPU path");
    for (size_t g = 0; g < n; ++g)
        if (devices[g] < 0 || devices[g] >= ndev) return fail(VDB_ERR_INVALID_ARGUMENT, "device %d out of range (%d)", devices[g], ndev);
    auto* P = new vdb_flat_index();
    auto* M = new vdb_multi();
    P->multi = M;
    P->metric = metric;
    P->device = devices[0];
    M->G = (int)n; M->home = devices[0];
    M->dev.assign(devices, devices + n);
    for (size_t a = 0; a < n; ++a) for (size_t b = a + 1; b < n; ++b) if (devices[a] == devices[b]) M->distinct = false;
    M->exchange = M->distinct ? VDB_EXCHANGE_RCCL : VDB_EXCHANGE_PEER;
    M->ps.resize(n);
    auto bail = [&](int rc) { std::string m; size_t e = 0, a = 0; last_error(&m, &e, &a); multi_destroy(P); return rc == VDB_ERR_DIMENSION_MISMATCH ? fail_dim(e, a) : fail(rc, "%s", m.c_str()); };
    for (size_t g = 0; g < n; ++g) {
        vdb_flat_index* c = nullptr;
        int rc = vdb_flat_create(metric, devices[g], &c);
        if (rc) return bail(rc);
        M->sh.push_back(c);
        if (hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&M->ps[g].stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&M->ps[g].ev_done, hipEventDisableTiming) != hipSuccess)
            return bail(fail(VDB_ERR_DEVICE, "stream / event creation failed on device %d", devices[g]));
    }
    // direct xGMI copies between the home device and the others where the topology allows (otherwise the runtime stages them)
    for (size_t g = 0; g < n; ++g) {
        if (devices[g] == M->home) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[g], M->home) == hipSuccess && can) {
            (void)hipSetDevice(devices[g]);
            hipError_t e = hipDeviceEnablePeerAccess(M->home, 0);
            if (e != hipSuccess) (void)hipGetLastError();                // already enabled (e.g. by the host application): fine
        }
        if (hipDeviceCanAccessPeer(&can, M->home, devices[g]) == hipSuccess && can) {
            (void)hipSetDevice(M->home);
            hipError_t e = hipDeviceEnablePeerAccess(devices[g], 0);
            if (e != hipSuccess) (void)hipGetLastError();
        }
    }
    if (hipSetDevice(M->home) != hipSuccess || hipMalloc((void**)&M->d_status, 16) != hipSuccess ||
        hipHostMalloc((void**)&M->h_status, 16, hipHostMallocDefault) != hipSuccess)
        return bail(fail(VDB_ERR_DEVICE, "allocation failed on device %d", M->home));
    M->gang.start(M->G, [M](int g) { (void)hipSetDevice(M->dev[g]); });
    *out = P;
    return VDB_OK;
}

void multi_destroy(vdb_flat_index* P) {
    vdb_multi* M = P->multi;
    if (M) {
        M->gang.stop();
        const vdb_rccl::Rccl* r = M->comm.empty() ? nullptr : vdb_rccl::rccl();
        for (size_t g = 0; g < M->ps.size(); ++g) {
            auto& p = M->ps[g];
            (void)hipSetDevice(M->dev[g]);
            if (p.stream) (void)hipStreamSynchronize(p.stream);
            if (r && g < M->comm.size() && M->comm[g]) (void)r->comm_destroy(M->comm[g]);
            if (p.d_q) (void)hipFree(p.d_q);
            if (p.d_mask) (void)hipFree(p.d_mask);
            if (p.d_pack) (void)hipFree(p.d_pack);
            if (p.d_gath) (void)hipFree(p.d_gath);
            if (p.ev_done) (void)hipEventDestroy(p.ev_done);
            if (p.stream) (void)hipStreamDestroy(p.stream);
        }
        for (auto* c : M->sh) vdb_flat_destroy(c);
        (void)hipSetDevice(M->home);
        if (M->d_status) (void)hipFree(M->d_status);
        if (M->h_status) (void)hipHostFree(M->h_status);
        M->w_qin.release(); M->w_outd.release(); M->w_outi.release(); M->w_mask.release(); M->w_outc.release();
        delete M;
    }
    delete P;
    (void)hipGetLastError();                   // teardown is best effort: leave no stale error behind for the thread's next call
}

// ------------------------------------------------------------------ mutation and inspection
int multi_add(vdb_flat_index* P, uint64_t id, const float* v, size_t dim) {
    std::lock_guard<std::mutex> lk(P->mu);
    return add_one_routed(P->multi, id, v, dim);
}

int multi_add_bulk(vdb_flat_index* P, const uint64_t* ids, uint64_t first_id, const float* rows, size_t n, size_t dim, bool on_device) {
    vdb_multi* M = P->multi;
    std::lock_guard<std::mutex> lk(P->mu);
    if (n == 0) return VDB_OK;
    // the same id twice in ONE batch is last-wins (HashMap::insert, flat_index.rs:38-41): blocks on different shards cannot see
    // each other's ids, so such a batch takes the routed path row by row (host rows) or is refused (device rows)
    if (ids) {
        std::unordered_set<uint64_t> seen;
        seen.reserve(n * 2);
        bool dup = false;
        for (size_t i = 0; i < n && !dup; ++i) dup = !seen.insert(ids[i]).second;
        if (dup) {
            if (on_device) return fail(VDB_ERR_INVALID_ARGUMENT, "duplicate ids inside one device batch are not supported on a sharded handle");
            for (size_t i = 0; i < n; ++i) { int rc = add_one_routed(M, ids[i], rows + i * dim, dim); if (rc) return rc; }
            return VDB_OK;
        }
    }
    // ids that are already stored move to wherever their row lands now: drop the old rows first
    bool fresh = true;
    for (auto* c : M->sh) fresh = fresh && is_fresh(c);
    if (!fresh)
        for (size_t i = 0; i < n; ++i) {
            const uint64_t id = ids ? ids[i] : first_id + i;
            for (auto* c : M->sh) if (holds(c, id)) { int rc = vdb_flat_remove(c, id); if (rc) return rc; }
        }
    // contiguous blocks, one per shard (vdb_shard_range), loaded concurrently: G host-to-device streams instead of one
    for (auto& p : M->ps) { p.rc = VDB_OK; p.msg.clear(); }
    M->gang.run([&](int g) {
        size_t lo = 0, hi = 0;
        vdb_shard_range(n, g, M->G, &lo, &hi);
        if (hi == lo) return;
        int rc;
        if (!on_device) rc = vdb_flat_add_bulk(M->sh[g], ids ? ids + lo : nullptr, first_id + lo, rows + lo * dim, hi - lo, dim);
        else if (M->dev[g] == M->home) rc = vdb_flat_add_bulk_device(M->sh[g], ids ? ids + lo : nullptr, first_id + lo, rows + lo * dim, hi - lo, dim);
        else {
            // rows resident on the home device: staged through a buffer on the shard's device, 64 Mi floats at a time
            rc = VDB_OK;
            const size_t chunk_rows = std::max<size_t>(1, ((size_t)64 << 20) / std::max<size_t>(dim, 1));
            float* tmp = nullptr;
            if (hipSetDevice(M->dev[g]) != hipSuccess || hipMalloc((void**)&tmp, std::min(chunk_rows, hi - lo) * dim * 4) != hipSuccess)
                rc = fail(VDB_ERR_DEVICE, "staging allocation failed on device %d", M->dev[g]);
            for (size_t a = lo; a < hi && rc == VDB_OK; a += chunk_rows) {
                const size_t cnt = std::min(chunk_rows, hi - a);
                if (hipMemcpyPeerAsync(tmp, M->dev[g], rows + a * dim, M->home, cnt * dim * 4, M->ps[g].stream) != hipSuccess ||
                    hipStreamSynchronize(M->ps[g].stream) != hipSuccess)
                    rc = fail(VDB_ERR_DEVICE, "peer copy of rows to device %d failed", M->dev[g]);
                else rc = vdb_flat_add_bulk_device(M->sh[g], ids ? ids + a : nullptr, first_id + a, tmp, cnt, dim);
            }
            if (tmp) (void)hipFree(tmp);
        }
        if (rc) capture_error(M->ps[g], rc);
    });
    (void)hipSetDevice(M->home);
    return first_error(M);
}

int multi_remove(vdb_flat_index* P, uint64_t id) {
    std::lock_guard<std::mutex> lk(P->mu);
    for (auto* c : P->multi->sh) { int rc = vdb_flat_remove(c, id); if (rc) return rc; }     // absent id is Ok(()) on every shard
    return VDB_OK;
}

int multi_get_vector(vdb_flat_index* P, uint64_t id, float* out, size_t cap, size_t* dim) {
    std::lock_guard<std::mutex> lk(P->mu);
    for (auto* c : P->multi->sh) {
        int rc = vdb_flat_get_vector(c, id, out, cap, dim);
        if (rc != VDB_ERR_NOT_FOUND) return rc;
    }
    return fail(VDB_ERR_NOT_FOUND, "Vector not found: %llu", (unsigned long long)id);
}

size_t multi_len(const vdb_flat_index* P) {
    size_t n = 0;
    for (auto* c : P->multi->sh) n += vdb_flat_len(c);
    return n;
}
size_t multi_dim(const vdb_flat_index* P) {
    for (auto* c : P->multi->sh) { size_t d = vdb_flat_dim(c); if (d) return d; }
    return 0;
}

int multi_reserve(vdb_flat_index* P, size_t rows, size_t dim) {
    vdb_multi* M = P->multi;
    std::lock_guard<std::mutex> lk(P->mu);
    for (int g = 0; g < M->G; ++g) {
        size_t lo = 0, hi = 0;
        vdb_shard_range(rows, g, M->G, &lo, &hi);
        if (hi > lo) { int rc = vdb_flat_reserve(M->sh[g], hi - lo, dim); if (rc) return rc; }
    }
    return VDB_OK;
}

// f(child) on every shard, concurrently; the first failing shard's error is the call's
int multi_for_each(vdb_flat_index* P, const std::function<int(vdb_flat_index*)>& f) {
    vdb_multi* M = P->multi;
    std::lock_guard<std::mutex> lk(P->mu);
    for (auto& p : M->ps) { p.rc = VDB_OK; p.msg.clear(); }
    M->gang.run([&](int g) { int rc = f(M->sh[g]); if (rc) capture_error(M->ps[g], rc); });
    (void)hipSetDevice(M->home);
    return first_error(M);
}

// ------------------------------------------------------------------ the batched search
namespace {

int search_locked(vdb_flat_index* P, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_mask, size_t mask_bits,
                  uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, hipStream_t user_stream) {
    vdb_multi* M = P->multi;
    const uint64_t t0 = now_ns();
    M->stats[0] = 0; M->stats[1] = (uint64_t)M->G; M->stats[2] = (uint64_t)M->exchange; M->stats[3] = 0;
    auto done = [&](int rc) { M->stats[4] = now_ns() - t0; return rc; };
    if (nq == 0) return done(VDB_OK);
    HIP_TRY(hipSetDevice(M->home));
    // the caller's queries may have been produced on its stream: wait for it on the host (cheap when it is idle), so that
    // no shard stream needs an event of another device's stream in front of its copy
    if (user_stream) HIP_TRY(hipStreamSynchronize(user_stream));
    hipStream_t s0 = M->ps[0].stream;
    if (multi_len(P) == 0 || k == 0) {                               // storage.rs:218-220: empty store -> Ok(vec![]) before any check
        HIP_TRY(hipMemsetAsync(d_out_counts, 0, nq * 4, s0));
        HIP_TRY(hipStreamSynchronize(s0));
        return done(VDB_OK);
    }
    if ((size_t)M->G * k > 2048) return done(fail(VDB_ERR_INVALID_ARGUMENT, "shards * k = %zu exceeds the merge capacity of 2048", (size_t)M->G * k));
    if (nq > 0x3fffffffull) return done(fail(VDB_ERR_INVALID_ARGUMENT, "batch too large"));
    if (M->exchange == VDB_EXCHANGE_RCCL) { int rc = ensure_comms(M); if (rc) return done(rc); M->stats[3] = (uint64_t)M->comm_world; }
    const size_t nk = nq * k;
    size_t words = nq * (3 * k + 1) + 1;
    words += words & 1;
    const size_t mask_words = d_mask ? (mask_bits + 63) / 64 : 0;
    for (int g = 0; g < M->G; ++g) {
        auto& p = M->ps[g];
        int rc;
        if ((rc = ensure_on(M->dev[g], p.d_pack, p.pack_words, words))) return done(rc);
        if (M->exchange == VDB_EXCHANGE_RCCL || g == 0)
            if ((rc = ensure_on(M->dev[g], p.d_gath, p.gath_words, words * (size_t)M->G))) return done(rc);
        if (M->dev[g] != M->home) {
            if ((rc = ensure_on(M->dev[g], p.d_q, p.q_cap, nq * std::max<size_t>(dim, 1)))) return done(rc);
            if (mask_words && (rc = ensure_on(M->dev[g], p.d_mask, p.mask_cap, mask_words))) return done(rc);
        }
        p.rc = VDB_OK; p.begun = false; p.changed = 0; p.msg.clear();
    }
    HIP_TRY(hipSetDevice(M->home));

    // a shard's block = zeroed results + 1000 + code when its local search failed: the merge ignores it, the status carries the code
    auto send = [&](int g) {
        auto& p = M->ps[g];
        hipStream_t s = p.stream;
        int32_t* p_code = p.d_pack + 3 * nk + nq;
        hipError_t e = hipSuccess;
        if (p.rc != VDB_OK) {
            e = hipMemsetAsync(p.d_pack, 0, words * 4, s);
            if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)p_code, (int)(CODE_ERR_BASE + (uint32_t)p.rc), 1, s);
        }
        if (e == hipSuccess && M->exchange == VDB_EXCHANGE_PEER) {
            int32_t* dst = M->ps[0].d_gath + (size_t)g * words;
            if (M->dev[g] == M->home) e = hipMemcpyAsync(dst, p.d_pack, words * 4, hipMemcpyDeviceToDevice, s);
            else e = hipMemcpyPeerAsync(dst, M->home, p.d_pack, M->dev[g], words * 4, s);
            if (e == hipSuccess && g != 0) e = hipEventRecord(p.ev_done, s);
        }
        if (e != hipSuccess && p.rc == VDB_OK) capture_error(p, fail(VDB_ERR_DEVICE, "HIP error %d (%s) while sending shard %d's block", (int)e, hipGetErrorString(e), g));
    };

    // ---- first tier of every shard's local search, enqueued concurrently
    M->gang.run([&](int g) {
        auto& p = M->ps[g];
        hipStream_t s = p.stream;
        const float* q = d_q;
        const uint64_t* mk = d_mask;
        if (hipSetDevice(M->dev[g]) != hipSuccess) { capture_error(p, fail(VDB_ERR_DEVICE, "hipSetDevice(%d) failed", M->dev[g])); return; }
        if (M->dev[g] != M->home) {
            hipError_t e = dim ? hipMemcpyPeerAsync(p.d_q, M->dev[g], d_q, M->home, nq * dim * 4, s) : hipSuccess;
            if (e == hipSuccess && mask_words) e = hipMemcpyPeerAsync(p.d_mask, M->dev[g], d_mask, M->home, mask_words * 8, s);
            if (e != hipSuccess) capture_error(p, fail(VDB_ERR_DEVICE, "peer copy of the queries to device %d failed: %s", M->dev[g], hipGetErrorString(e)));
            q = p.d_q;
            mk = mask_words ? p.d_mask : nullptr;
        }
        if (p.rc == VDB_OK) {
            uint64_t* p_ids = reinterpret_cast<uint64_t*>(p.d_pack);
            float* p_dists = reinterpret_cast<float*>(p.d_pack + 2 * nk);
            uint32_t* p_counts = reinterpret_cast<uint32_t*>(p.d_pack + 3 * nk);
            int32_t* p_code = p.d_pack + 3 * nk + nq;
            int rc = vdb_flat_search_batch_device_begin(M->sh[g], q, nq, dim, k, mk, mask_bits, p_ids, p_dists, p_counts, p_code, s);
            if (rc) capture_error(p, rc);
            else p.begun = true;
        }
        send(g);
    });
    M->stats[5] = now_ns() - t0;

    // ---- exchange 1
    uint32_t worst = 0;
    int xrc = exchange(M, words, nq, k, d_out_ids, d_out_dists, d_out_counts, &worst);

    // ---- second half of every local search (never leave a child locked), and, when the reduced status says some shard
    // needed its host, the shard's block once more
    const bool again = xrc == VDB_OK && worst == VDB_PENDING_HOST;
    M->gang.run([&](int g) {
        auto& p = M->ps[g];
        if (hipSetDevice(M->dev[g]) != hipSuccess) return;
        if (p.begun) {
            int rc = vdb_flat_search_batch_device_finish(M->sh[g], &p.changed);
            if (rc) capture_error(p, rc);
        }
        if (again) {
            if (p.rc == VDB_OK) {
                int32_t* p_code = p.d_pack + 3 * nk + nq;
                if (hipMemsetD32Async((hipDeviceptr_t)p_code, 0, 1, p.stream) != hipSuccess) capture_error(p, fail(VDB_ERR_DEVICE, "hipMemsetD32Async failed"));
            }
            send(g);
        }
    });
    if (xrc) return done(xrc);
    if (again && (xrc = exchange(M, words, nq, k, d_out_ids, d_out_dists, d_out_counts, &worst))) return done(xrc);
    (void)hipSetDevice(M->home);
    // aggregate counters of the children (vdb_flat_last_stats_ex of the parent)
    memset(P->stats, 0, sizeof(P->stats));
    for (auto* c : M->sh) {
        uint64_t st[16];
        vdb_flat_last_stats_ex(c, st, 16);
        for (int i : {0, 1, 2, 3, 6, 9, 13}) P->stats[i] += st[i];
        for (int i : {4, 5, 7}) P->stats[i] = std::max(P->stats[i], st[i]);
        for (int i : {8, 14, 15}) P->stats[i] |= st[i];
    }
    P->stats[10] = M->stats[5];
    P->stats[12] = now_ns() - t0;
    int erc = first_error(M);
    if (erc) return done(erc);
    if (worst != 0) return done(fail(VDB_ERR_DEVICE, "sharded search ended with status %u and no failing shard", worst));
    return done(VDB_OK);
}

}  // namespace

int multi_search_device(vdb_flat_index* P, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_mask, size_t mask_bits,
                        uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, hipStream_t user_stream) {
    std::lock_guard<std::mutex> lk(P->mu);
    return search_locked(P, d_q, nq, dim, k, d_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts, user_stream);
}

int multi_search_host(vdb_flat_index* P, const float* queries, size_t nq, size_t dim, const size_t* ks, size_t k, const uint64_t* id_mask,
                      size_t mask_bits, size_t kstride, uint64_t* out_ids, float* out_dists, size_t* out_counts) {
    vdb_multi* M = P->multi;
    size_t kmax = k;
    if (ks) { kmax = 0; for (size_t b = 0; b < nq; ++b) kmax = std::max(kmax, ks[b]); }
    if (kmax > kstride) return fail(VDB_ERR_INVALID_ARGUMENT, "kstride %zu smaller than the largest k %zu", kstride, kmax);
    if (kmax && nq && (!out_ids || !out_dists)) return fail(VDB_ERR_INVALID_ARGUMENT, "null output");
    std::lock_guard<std::mutex> lk(P->mu);
    if (nq == 0) return VDB_OK;
    HIP_TRY(hipSetDevice(M->home));
    const size_t len = multi_len(P);
    const size_t kdev = std::min(kmax, std::max<size_t>(len, 1));      // Index::search returns at most len results
    hipStream_t s = M->ps[0].stream;
    int rc;
    if ((rc = M->w_qin.ensure(nq * std::max<size_t>(dim, 1)))) return rc;
    if ((rc = M->w_outi.ensure(nq * std::max<size_t>(kdev, 1)))) return rc;
    if ((rc = M->w_outd.ensure(nq * std::max<size_t>(kdev, 1)))) return rc;
    if ((rc = M->w_outc.ensure(nq))) return rc;
    if (dim) HIP_TRY(hipMemcpyAsync(M->w_qin.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
    const uint64_t* d_mask = nullptr;
    if (id_mask) {
        const size_t words = (mask_bits + 63) / 64;
        if ((rc = M->w_mask.ensure(std::max<size_t>(words, 1)))) return rc;
        if (words) HIP_TRY(hipMemcpyAsync(M->w_mask.p, id_mask, words * 8, hipMemcpyHostToDevice, s));
        d_mask = M->w_mask.p;
    }
    HIP_TRY(hipStreamSynchronize(s));                                  // the shards' streams read the staged queries
    if ((rc = search_locked(P, M->w_qin.p, nq, dim, kdev, d_mask, mask_bits, M->w_outi.p, M->w_outd.p, M->w_outc.p, nullptr))) return rc;
    std::vector<uint32_t> cnt(nq);
    std::vector<uint64_t> ids(nq * std::max<size_t>(kdev, 1));
    std::vector<float> ds(nq * std::max<size_t>(kdev, 1));
    HIP_TRY(hipMemcpyAsync(cnt.data(), M->w_outc.p, nq * 4, hipMemcpyDeviceToHost, s));
    if (kdev) {
        HIP_TRY(hipMemcpyAsync(ids.data(), M->w_outi.p, nq * kdev * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ds.data(), M->w_outd.p, nq * kdev * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t b = 0; b < nq; ++b) {
        const size_t kb = ks ? ks[b] : k;
        const size_t c = std::min<size_t>(cnt[b], kb);                 // per-query k: a prefix of the batch-wide result
        out_counts[b] = c;
        for (size_t i = 0; i < c; ++i) { out_ids[b * kstride + i] = ids[b * kdev + i]; out_dists[b * kstride + i] = ds[b * kdev + i]; }
    }
    return VDB_OK;
}

int multi_set_exchange(vdb_flat_index* P, int mode) {
    vdb_multi* M = P->multi;
    std::lock_guard<std::mutex> lk(P->mu);
    if (mode != VDB_EXCHANGE_RCCL && mode != VDB_EXCHANGE_PEER) return fail(VDB_ERR_INVALID_ARGUMENT, "mode must be VDB_EXCHANGE_RCCL or VDB_EXCHANGE_PEER");
    if (mode == VDB_EXCHANGE_RCCL && !M->distinct) return fail(VDB_ERR_INVALID_ARGUMENT, "the RCCL exchange needs distinct devices (a device is listed twice)");
    M->exchange = mode;
    return VDB_OK;
}

size_t multi_shards(const vdb_flat_index* P) { return (size_t)P->multi->G; }
size_t multi_shard_len(const vdb_flat_index* P, size_t g) { return g < (size_t)P->multi->G ? vdb_flat_len(P->multi->sh[g]) : 0; }
void multi_stats(const vdb_flat_index* P, uint64_t out[8]) { memcpy(out, P->multi->stats, sizeof(P->multi->stats)); }

}  // namespace vdbi
