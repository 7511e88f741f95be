// vdb_shard.cpp -- host side of include/vdb_shard.h: the row-sharded FlatIndex search of SURVEY.md 8(e) behind the
// C ABI, calling RCCL directly (dlopen: an RCCL already in the process -- PyTorch ships one -- is reused, otherwise
// librccl.so.1 of the ROCm installation is loaded).  One vdb_shard_group per rank = one RCCL communicator plus the packed
// exchange buffers; the local search is the ordinary vdb_flat_index pipeline (vdb_flat.cpp) in its two halves.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/vdb_flat.h"
#include "../../include/vdb_shard.h"
#include "kernels.h"
#include "vdb_internal.h"

namespace {

// ---- the six RCCL entry points this file needs (rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer,
// ncclUniqueId = 128 opaque bytes passed BY VALUE, ncclDataType_t ncclInt8 = 0)
struct NcclId { char internal[VDB_SHARD_UNIQUE_ID_BYTES]; };
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(void**, int, NcclId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_comm_count)(void*, int*);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);
struct Rccl {
    void* lib = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_count comm_count = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    char why[256] = {0};
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        // an RCCL that is already mapped (same process as PyTorch: torch/lib/librccl.so) is reused; two copies of a
        // 300 MB collective library in one process buy nothing
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : paths)
            if (!g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.lib) { snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl not found: %s", dlerror()); return; }
        g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.lib, "ncclGetUniqueId");
        g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.lib, "ncclCommInitRank");
        g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.lib, "ncclCommDestroy");
        g_rccl.comm_count = (fn_comm_count)dlsym(g_rccl.lib, "ncclCommCount");
        g_rccl.all_gather = (fn_all_gather)dlsym(g_rccl.lib, "ncclAllGather");
        g_rccl.error_string = (fn_error_string)dlsym(g_rccl.lib, "ncclGetErrorString");
        if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.comm_count || !g_rccl.all_gather) {
            snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks an expected symbol");
            g_rccl.lib = nullptr;
        }
    });
    return g_rccl.lib ? &g_rccl : nullptr;
}

int err(int code, const char* msg) { return vdb_internal::set_error(code, msg); }
int nccl_fail(const char* what, int rc) {
    char buf[320];
    const Rccl* r = rccl();
    snprintf(buf, sizeof(buf), "RCCL %s failed: %d (%s)", what, rc, (r && r->error_string) ? r->error_string(rc) : "?");
    return err(VDB_ERR_DEVICE, buf);
}

#define SH_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) { char b_[256]; snprintf(b_, sizeof(b_), "HIP error %d (%s): %s", (int)e_, hipGetErrorString(e_), #expr); return err(VDB_ERR_DEVICE, b_); } \
    } while (0)

constexpr uint32_t CODE_ERR_BASE = 1000;     // status word of a rank whose local search failed: 1000 + vdb_status (survives the MAX with VDB_PENDING_HOST)

template <class F> int guarded(F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { return err(VDB_ERR_DEVICE, "internal error: out of host memory"); }
    catch (...) { return err(VDB_ERR_DEVICE, "internal error: C++ exception"); }
}

}  // namespace

struct vdb_shard_group {
    int rank = 0, world = 1, device = 0, comm_world = 1;
    void* comm = nullptr;
    hipStream_t stream = nullptr;             // used when the caller passes no stream
    int32_t* d_pack = nullptr; int32_t* d_gath = nullptr; size_t pack_words = 0;
    uint32_t* d_status = nullptr; uint32_t* h_status = nullptr;
    std::mutex mu;
    uint64_t stats[4] = {0, 0, 0, 0};
};

namespace {

int ensure_buffers(vdb_shard_group* g, size_t words) {
    if (words <= g->pack_words) return VDB_OK;
    if (g->d_pack) (void)hipFree(g->d_pack);
    if (g->d_gath) (void)hipFree(g->d_gath);
    g->d_pack = g->d_gath = nullptr; g->pack_words = 0;
    const size_t cap = words + words / 2;
    SH_TRY(hipMalloc((void**)&g->d_pack, cap * 4));
    SH_TRY(hipMalloc((void**)&g->d_gath, cap * 4 * (size_t)g->world));
    g->pack_words = cap;
    return VDB_OK;
}

// all-gather of the packed per-rank buffers + merge into the caller's outputs + reduced status on the host (ONE sync)
int exchange(vdb_shard_group* g, size_t words, size_t nq, size_t k, uint64_t* d_out_ids, float* d_out_dists,
             uint32_t* d_out_counts, hipStream_t s, uint32_t* worst) {
    const Rccl* r = rccl();
    int rc = r->all_gather(g->d_pack, g->d_gath, words * 4, /* ncclInt8 */ 0, g->comm, s);
    if (rc) return nccl_fail("ncclAllGather", rc);
    vdb::launch_merge_packed(g->d_gath, words, (uint32_t)g->world, (uint32_t)nq, (uint32_t)k, d_out_ids, d_out_dists, d_out_counts,
                             g->d_status, s);
    SH_TRY(hipGetLastError());
    SH_TRY(hipMemcpyAsync(g->h_status, g->d_status, 4, hipMemcpyDeviceToHost, s));
    SH_TRY(hipStreamSynchronize(s));
    *worst = *g->h_status;
    g->stats[0]++;
    return VDB_OK;
}

}  // namespace

extern "C" {

int vdb_shard_unique_id(unsigned char out[VDB_SHARD_UNIQUE_ID_BYTES]) {
    if (!out) return err(VDB_ERR_INVALID_ARGUMENT, "out is null");
    const Rccl* r = rccl();
    if (!r) return err(VDB_ERR_DEVICE, g_rccl.why);
    NcclId id;
    int rc = r->get_unique_id(&id);
    if (rc) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(out, id.internal, VDB_SHARD_UNIQUE_ID_BYTES);
    return VDB_OK;
}

int vdb_shard_group_create(const unsigned char id[VDB_SHARD_UNIQUE_ID_BYTES], int rank, int world, int device,
                           vdb_shard_group** out) {
    return guarded([&]() -> int {
    if (!out) return err(VDB_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return err(VDB_ERR_INVALID_ARGUMENT, "rank / world out of range");
    if (world > 1 && !id) return err(VDB_ERR_INVALID_ARGUMENT, "a unique id is required for world > 1");
    SH_TRY(hipSetDevice(device));
    auto* g = new vdb_shard_group();
    g->rank = rank; g->world = world; g->device = device; g->comm_world = 1;
    auto fail_with = [&](int rc) { vdb_shard_group_destroy(g); return rc; };
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) return fail_with(err(VDB_ERR_DEVICE, "hipStreamCreate failed"));
    if (hipMalloc((void**)&g->d_status, 16) != hipSuccess || hipHostMalloc((void**)&g->h_status, 16, hipHostMallocDefault) != hipSuccess)
        return fail_with(err(VDB_ERR_DEVICE, "allocation failed"));
    if (world > 1 || id) {                                         // (world == 1 WITH an id: a single-rank communicator, full exchange path)
        const Rccl* r = rccl();
        if (!r) return fail_with(err(VDB_ERR_DEVICE, g_rccl.why));
        NcclId nid;
        memcpy(nid.internal, id, VDB_SHARD_UNIQUE_ID_BYTES);
        int rc = r->comm_init_rank(&g->comm, world, nid, rank);
        if (rc) return fail_with(nccl_fail("ncclCommInitRank", rc));
        int cnt = 0;
        rc = r->comm_count(g->comm, &cnt);
        if (rc) return fail_with(nccl_fail("ncclCommCount", rc));
        g->comm_world = cnt;
        if (cnt != world) return fail_with(err(VDB_ERR_DEVICE, "RCCL communicator has a different rank count than requested"));
    }
    g->stats[1] = (uint64_t)g->comm_world;
    *out = g;
    return VDB_OK;
    });
}

void vdb_shard_group_destroy(vdb_shard_group* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->comm) { const Rccl* r = rccl(); if (r) (void)r->comm_destroy(g->comm); }
    if (g->d_pack) (void)hipFree(g->d_pack);
    if (g->d_gath) (void)hipFree(g->d_gath);
    if (g->d_status) (void)hipFree(g->d_status);
    if (g->h_status) (void)hipHostFree(g->h_status);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

int vdb_shard_group_rank(const vdb_shard_group* g) { return g ? g->rank : -1; }
int vdb_shard_group_world(const vdb_shard_group* g) { return g ? g->comm_world : 0; }

void vdb_shard_range(size_t n_rows, int rank, int world, size_t* lo, size_t* hi) {
    if (world < 1) world = 1;
    const size_t base = n_rows / (size_t)world, rem = n_rows % (size_t)world, r = (size_t)(rank < 0 ? 0 : rank);
    const size_t l = r * base + (r < rem ? r : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (r < rem ? 1 : 0);
}

int vdb_flat_search_batch_sharded(vdb_shard_group* g, vdb_flat_index* local, const float* d_queries, size_t nq, size_t dim,
                                  size_t k, const uint64_t* d_id_mask, size_t mask_bits, uint64_t* d_out_ids, float* d_out_dists,
                                  uint32_t* d_out_counts, void* stream) {
    return guarded([&]() -> int {
    if (!g || !local || (nq && (!d_queries || !d_out_counts || (k && (!d_out_ids || !d_out_dists)))))
        return err(VDB_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    const auto t0 = std::chrono::steady_clock::now();
    g->stats[0] = 0; g->stats[2] = 0;
    auto done = [&](int rc) { g->stats[3] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); return rc; };
    if (!g->comm)
        return done(vdb_flat_search_batch_device(local, d_queries, nq, dim, k, d_id_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts, stream));
    // ---- argument checks that are identical on every rank come BEFORE any collective
    if (nq == 0) return done(VDB_OK);
    if ((size_t)g->world * k > 2048) return done(err(VDB_ERR_INVALID_ARGUMENT, "world * k exceeds the merge capacity of 2048"));
    if (nq > 0x3fffffffull) return done(err(VDB_ERR_INVALID_ARGUMENT, "batch too large"));
    SH_TRY(hipSetDevice(g->device));
    hipStream_t s = stream ? (hipStream_t)stream : g->stream;
    if (k == 0) {                                                  // (k is the same on every rank: no rank enters a collective)
        SH_TRY(hipMemsetAsync(d_out_counts, 0, nq * 4, s));
        SH_TRY(hipStreamSynchronize(s));
        return done(VDB_OK);
    }
    const size_t nk = nq * k;
    size_t words = nq * (3 * k + 1) + 1;
    words += words & 1;
    int rc;
    if ((rc = ensure_buffers(g, words))) return done(rc);
    // the local search writes straight into the packed buffer: ids | dists | counts | status word
    uint64_t* p_ids = reinterpret_cast<uint64_t*>(g->d_pack);
    float* p_dists = reinterpret_cast<float*>(g->d_pack + 2 * nk);
    uint32_t* p_counts = reinterpret_cast<uint32_t*>(g->d_pack + 3 * nk);
    int32_t* p_code = g->d_pack + 3 * nk + nq;
    auto send_error = [&](int code) -> int {                       // zeroed results + 1000 + code: keeps this rank in step
        SH_TRY(hipMemsetAsync(g->d_pack, 0, words * 4, s));
        SH_TRY(hipMemsetD32Async((hipDeviceptr_t)p_code, (int)(CODE_ERR_BASE + (uint32_t)code), 1, s));
        return VDB_OK;
    };
    // ---- first tier of the local search: enqueued only, its "needs the host" word lands in the buffer on the device
    int local_rc = vdb_flat_search_batch_device_begin(local, d_queries, nq, dim, k, d_id_mask, mask_bits, p_ids, p_dists, p_counts, p_code, s);
    const bool begun = local_rc == VDB_OK;
    char local_msg[512] = {0};
    size_t e_exp = 0, e_act = 0;
    if (!begun) { vdb_last_error(local_msg, sizeof(local_msg), &e_exp, &e_act); if ((rc = send_error(local_rc))) return done(rc); }
    // ---- exchange 1: ALWAYS, on every rank
    uint32_t worst = 0;
    if ((rc = exchange(g, words, nq, k, d_out_ids, d_out_dists, d_out_counts, s, &worst))) {
        if (begun) { int ch = 0; (void)vdb_flat_search_batch_device_finish(local, &ch); }      // never leave the handle locked
        return done(rc);
    }
    // ---- second half of the local search (fallback tiers for this rank's uncertified queries, its errors)
    if (begun) {
        int changed = 0;
        local_rc = vdb_flat_search_batch_device_finish(local, &changed);
        g->stats[2] = (uint64_t)(changed != 0);
        if (local_rc != VDB_OK) vdb_last_error(local_msg, sizeof(local_msg), &e_exp, &e_act);
    }
    // ---- exchange 2: on ALL ranks iff exchange 1's reduced status says some rank was pending -- decided from gathered
    // data, identical everywhere.  A rank that failed (in either half) sends zeroed results and its error code.
    if (worst == VDB_PENDING_HOST) {
        if (local_rc != VDB_OK) { if ((rc = send_error(local_rc))) return done(rc); }
        else SH_TRY(hipMemsetD32Async((hipDeviceptr_t)p_code, 0, 1, s));
        if ((rc = exchange(g, words, nq, k, d_out_ids, d_out_dists, d_out_counts, s, &worst))) return done(rc);
    } else if (local_rc != VDB_OK && worst < CODE_ERR_BASE) {
        worst = CODE_ERR_BASE + (uint32_t)local_rc;               // (cannot happen: a failing finish implies this rank was pending)
    }
    if (worst == 0) return done(VDB_OK);
    if (local_rc != VDB_OK) {                                       // this rank's own error: its message, its dimension pair
        if (local_rc == VDB_ERR_DIMENSION_MISMATCH) return done(vdb_internal::set_dim_error(e_exp, e_act));
        return done(err(local_rc, local_msg));
    }
    const int code = worst >= CODE_ERR_BASE ? (int)(worst - CODE_ERR_BASE) : VDB_ERR_DEVICE;
    char buf[160];
    snprintf(buf, sizeof(buf), "a shard on another rank failed the batch (status %d)", code);
    return done(err(code >= VDB_ERR_DIMENSION_MISMATCH && code <= VDB_ERR_NOT_FOUND ? code : VDB_ERR_DEVICE, buf));
    });
}

int vdb_shard_group_last_stats(const vdb_shard_group* g, uint64_t out[4]) {
    if (!g || !out) return err(VDB_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(out, g->stats, sizeof(g->stats));
    return VDB_OK;
}

}  // extern "C"
