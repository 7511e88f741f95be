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
#include "vdb_rccl.h"

namespace {

using vdb_rccl::NcclId;
using vdb_rccl::Rccl;
using namespace vdb_rccl;
Rccl g_rccl;
std::once_flag g_rccl_once;

}  // namespace

namespace vdb_rccl {
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
        g_rccl.comm_init_all = (fn_comm_init_all)dlsym(g_rccl.lib, "ncclCommInitAll");
        g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.lib, "ncclCommDestroy");
        g_rccl.comm_count = (fn_comm_count)dlsym(g_rccl.lib, "ncclCommCount");
        g_rccl.all_gather = (fn_all_gather)dlsym(g_rccl.lib, "ncclAllGather");
        g_rccl.group_start = (fn_group)dlsym(g_rccl.lib, "ncclGroupStart");
        g_rccl.group_end = (fn_group)dlsym(g_rccl.lib, "ncclGroupEnd");
        g_rccl.error_string = (fn_error_string)dlsym(g_rccl.lib, "ncclGetErrorString");
        if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_init_all || !g_rccl.comm_destroy || !g_rccl.comm_count ||
            !g_rccl.all_gather || !g_rccl.group_start || !g_rccl.group_end) {
            snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks an expected symbol");
            g_rccl.lib = nullptr;
        }
    });
    return g_rccl.lib ? &g_rccl : nullptr;
}
const char* why() { return g_rccl.why; }
}  // namespace vdb_rccl

namespace {

using vdb_rccl::rccl;

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
    // fixed-size exchange made at create time: one word per rank.  A rank-LOCAL failure that would otherwise keep a rank out
    // of a collective (growing the packed buffers) is first agreed on through these, which cannot fail to exist.
    int32_t* d_vote = nullptr; int32_t* d_votes = nullptr; int32_t* h_votes = nullptr;
    bool poisoned = false;                    // a local device failure left this rank unable to follow the protocol: every later call fails at once
    std::mutex mu;
    uint64_t stats[4] = {0, 0, 0, 0};
};

namespace {

void drop_buffers(vdb_shard_group* g) {
    if (g->d_pack) (void)hipFree(g->d_pack);
    if (g->d_gath) (void)hipFree(g->d_gath);
    g->d_pack = g->d_gath = nullptr; g->pack_words = 0;
}

// Growing the packed buffers is a rank-LOCAL allocation, and a rank that returned on its failure would leave the others
// blocked in exchange 1 for ever.  So growth is agreed on: every rank tries, then ONE all-gather of a word per rank (buffers
// made at create time) tells everybody whether everybody succeeded.  If not, EVERY rank drops its buffers and fails the call
// -- capacities stay identical on all ranks (`words` is, and so is the growth rule), which is what makes "does this call
// grow?" a rank-identical decision in the first place.
int ensure_buffers(vdb_shard_group* g, size_t words, hipStream_t s) {
    if (words <= g->pack_words) return VDB_OK;
    drop_buffers(g);
    const size_t cap = words + words / 2;
    int32_t vote = 0;
    if (hipMalloc((void**)&g->d_pack, cap * 4) != hipSuccess || hipMalloc((void**)&g->d_gath, cap * 4 * (size_t)g->world) != hipSuccess) {
        (void)hipGetLastError();
        vote = 1;
    }
    // (a failing memset / copy below means the device itself is gone: the all-gather is still CALLED, so that healthy ranks are
    // not left waiting for this one's participation)
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)g->d_vote, vote, 1, s);
    const Rccl* r = rccl();
    int rc = r->all_gather(g->d_vote, g->d_votes, 4, /* ncclInt8 */ 0, g->comm, s);
    if (rc) { drop_buffers(g); return nccl_fail("ncclAllGather (buffer growth vote)", rc); }
    if (e == hipSuccess) e = hipMemcpyAsync(g->h_votes, g->d_votes, (size_t)g->world * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int bad = e != hipSuccess ? g->rank : -1;
    for (int i = 0; i < g->world && bad < 0; ++i) if (g->h_votes[i]) bad = i;
    if (bad >= 0) {
        drop_buffers(g);
        char b_[160];
        snprintf(b_, sizeof(b_), "rank %d could not allocate the exchange buffers (%zu bytes per rank); no rank searched", bad, cap * 4);
        return err(VDB_ERR_DEVICE, b_);
    }
    g->pack_words = cap;
    return VDB_OK;
}

// all-gather of the packed per-rank buffers + merge into the caller's outputs + reduced status on the host (ONE sync).
// Two kinds of failure: the RCCL call itself (returned; nothing more can be done on this communicator) and a LOCAL HIP
// failure after the collective was enqueued (*local_rc; the reduced status is then unknown on this rank).
int exchange(vdb_shard_group* g, size_t words, size_t nq, size_t k, uint64_t* d_out_ids, float* d_out_dists,
             uint32_t* d_out_counts, hipStream_t s, uint32_t* worst, int* local_rc) {
    const Rccl* r = rccl();
    *local_rc = VDB_OK;
    int rc = r->all_gather(g->d_pack, g->d_gath, words * 4, /* ncclInt8 */ 0, g->comm, s);
    if (rc) return nccl_fail("ncclAllGather", rc);
    g->stats[0]++;
    vdb::launch_merge_packed(g->d_gath, words, (uint32_t)g->world, (uint32_t)nq, (uint32_t)k, d_out_ids, d_out_dists, d_out_counts,
                             g->d_status, s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(g->h_status, g->d_status, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        char b_[200];
        snprintf(b_, sizeof(b_), "HIP error %d (%s) after the all-gather (merge / status read)", (int)e, hipGetErrorString(e));
        *local_rc = err(VDB_ERR_DEVICE, b_);
        return VDB_OK;
    }
    *worst = *g->h_status;
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
    if (hipMalloc((void**)&g->d_status, 16) != hipSuccess || hipHostMalloc((void**)&g->h_status, 16, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&g->d_vote, 16) != hipSuccess || hipMalloc((void**)&g->d_votes, (size_t)world * 4 + 16) != hipSuccess ||
        hipHostMalloc((void**)&g->h_votes, (size_t)world * 4 + 16, hipHostMallocDefault) != hipSuccess)
        return fail_with(err(VDB_ERR_DEVICE, "allocation failed"));
    if (world > 1 || id) {                                         // (world == 1 WITH an id: a single-rank communicator, full exchange path)
        const Rccl* r = rccl();
        if (!r) return fail_with(err(VDB_ERR_DEVICE, vdb_rccl::why()));
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
    if (g->d_vote) (void)hipFree(g->d_vote);
    if (g->d_votes) (void)hipFree(g->d_votes);
    if (g->h_votes) (void)hipHostFree(g->h_votes);
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
    if (vdb_flat_shards(local) != 1) return err(VDB_ERR_INVALID_ARGUMENT, "the local index of a shard group must be a plain single-GPU handle");
    std::lock_guard<std::mutex> lk(g->mu);
    const auto t0 = std::chrono::steady_clock::now();
    g->stats[0] = 0; g->stats[2] = 0;
    auto done = [&](int rc) { g->stats[3] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); return rc; };
    if (!g->comm)
        return done(vdb_flat_search_batch_device(local, d_queries, nq, dim, k, d_id_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts, stream));
    // ---- argument checks that are identical on every rank come BEFORE any collective; nothing rank-LOCAL may return
    // between here and the last exchange (a rank that left early would leave the others blocked in ncclAllGather)
    if (nq == 0) return done(VDB_OK);
    if ((size_t)g->world * k > 2048) return done(err(VDB_ERR_INVALID_ARGUMENT, "world * k exceeds the merge capacity of 2048"));
    if (nq > 0x3fffffffull) return done(err(VDB_ERR_INVALID_ARGUMENT, "batch too large"));
    if (g->poisoned) return done(err(VDB_ERR_DEVICE, "an earlier device failure left this rank out of step with its group: destroy the group"));
    int local_rc = VDB_OK;                                           // this rank's own failure, kept until the exchanges are over
    char local_msg[512] = {0};
    size_t e_exp = 0, e_act = 0;
    auto note = [&](int rc_) { if (rc_ != VDB_OK && local_rc == VDB_OK) { local_rc = rc_; vdb_last_error(local_msg, sizeof(local_msg), &e_exp, &e_act); } };
    auto hip_note = [&](hipError_t e_, const char* what) {
        if (e_ == hipSuccess) return;
        char b_[256];
        snprintf(b_, sizeof(b_), "HIP error %d (%s): %s", (int)e_, hipGetErrorString(e_), what);
        note(err(VDB_ERR_DEVICE, b_));
    };
    hip_note(hipSetDevice(g->device), "hipSetDevice");
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
    if ((rc = ensure_buffers(g, words, s))) return done(rc);         // agreed on by all ranks (see there): every rank returns here, or none
    // the local search writes straight into the packed buffer: ids | dists | counts | status word
    uint64_t* p_ids = reinterpret_cast<uint64_t*>(g->d_pack);
    float* p_dists = reinterpret_cast<float*>(g->d_pack + 2 * nk);
    uint32_t* p_counts = reinterpret_cast<uint32_t*>(g->d_pack + 3 * nk);
    int32_t* p_code = g->d_pack + 3 * nk + nq;
    auto send_error = [&](int code) {                               // zeroed results + 1000 + code: keeps this rank in step
        hip_note(hipMemsetAsync(g->d_pack, 0, words * 4, s), "zeroing the packed block");
        hip_note(hipMemsetD32Async((hipDeviceptr_t)p_code, (int)(CODE_ERR_BASE + (uint32_t)code), 1, s), "writing the status word");
    };
    // ---- first tier of the local search: enqueued only, its "needs the host" word lands in the buffer on the device
    bool begun = false;
    if (local_rc == VDB_OK) {
        note(vdb_flat_search_batch_device_begin(local, d_queries, nq, dim, k, d_id_mask, mask_bits, p_ids, p_dists, p_counts, p_code, s));
        begun = local_rc == VDB_OK;
    }
    if (!begun) send_error(local_rc);
    // ---- exchange 1: ALWAYS, on every rank
    uint32_t worst = 0;
    int x_local = VDB_OK;
    if ((rc = exchange(g, words, nq, k, d_out_ids, d_out_dists, d_out_counts, s, &worst, &x_local))) {
        if (begun) { int ch = 0; (void)vdb_flat_search_batch_device_finish(local, &ch); }      // never leave the handle locked
        g->poisoned = true;                                          // the communicator itself failed
        return done(rc);
    }
    // ---- second half of the local search (fallback tiers for this rank's uncertified queries, its errors)
    if (begun) {
        int changed = 0;
        note(vdb_flat_search_batch_device_finish(local, &changed));
        g->stats[2] = (uint64_t)(changed != 0);
    }
    if (x_local != VDB_OK) {
        // This rank does not know the reduced status (its merge or status read failed), so it cannot know whether the others
        // go on to exchange 2.  It takes part in one BLINDLY -- enqueued, not waited for: if the others exchange again they
        // are not held up by this rank; if they do not, only this rank's stream is left with a collective nobody answers,
        // and the group refuses further use here.
        note(x_local);
        send_error(local_rc);
        const Rccl* r = rccl();
        (void)r->all_gather(g->d_pack, g->d_gath, words * 4, /* ncclInt8 */ 0, g->comm, s);
        g->poisoned = true;
        if (local_rc == VDB_ERR_DIMENSION_MISMATCH) return done(vdb_internal::set_dim_error(e_exp, e_act));
        return done(err(local_rc, local_msg));
    }
    // ---- exchange 2: on ALL ranks iff exchange 1's reduced status says some rank was pending -- decided from gathered
    // data, identical everywhere.  A rank that failed (in either half) sends zeroed results and its error code.
    if (worst == VDB_PENDING_HOST) {
        if (local_rc != VDB_OK) send_error(local_rc);
        else hip_note(hipMemsetD32Async((hipDeviceptr_t)p_code, 0, 1, s), "clearing the status word");
        if ((rc = exchange(g, words, nq, k, d_out_ids, d_out_dists, d_out_counts, s, &worst, &x_local))) { g->poisoned = true; return done(rc); }
        note(x_local);                                             // (the last collective of the call: a local failure after it holds nobody up)
    } else if (local_rc != VDB_OK && worst < CODE_ERR_BASE) {
        worst = CODE_ERR_BASE + (uint32_t)local_rc;               // (a failing finish implies this rank was pending; a local HIP failure lands here)
    }
    if (worst == 0 && local_rc == VDB_OK) return done(VDB_OK);
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
