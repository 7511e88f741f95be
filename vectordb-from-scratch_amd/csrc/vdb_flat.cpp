// vdb_flat.cpp -- the extern "C" shims of include/vdb_flat.h over the device-resident mirror of the reference's FlatIndex
// (src/flat_index.rs:12-74; vdb_store.cpp) and the search pipeline that drives the HIP kernels (vdb_search.cpp).  No CPU
// compute path exists in this library: every distance is produced on the GPU, and every entry point fails with
// VDB_ERR_DEVICE when no HIP device is usable.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "vdb_index.h"

namespace vdbi {

namespace {
thread_local std::string g_err;
thread_local size_t g_expected = 0, g_actual = 0;
}  // namespace

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int fail_dim(size_t expected, size_t actual) {
    g_expected = expected;
    g_actual = actual;
    // same text as error.rs:12
    return fail(VDB_ERR_DIMENSION_MISMATCH, "Dimension mismatch: expected %zu, got %zu", expected, actual);
}
int guard_fail(const char* what) { return fail(VDB_ERR_DEVICE, "internal error: %s", what); }
void last_error(std::string* msg, size_t* expected, size_t* actual) {
    if (msg) *msg = g_err;
    if (expected) *expected = g_expected;
    if (actual) *actual = g_actual;
}

}  // namespace vdbi

using namespace vdbi;

// =================================================================== C ABI
extern "C" {

int vdb_abi_version(void) { return 1; }
const char* vdb_build_arch(void) { return "gfx950"; }

void vdb_last_error(char* buf, size_t cap, size_t* expected, size_t* actual) {
    std::string msg;
    vdbi::last_error(&msg, expected, actual);
    if (buf && cap) {
        size_t n = std::min(cap - 1, msg.size());
        memcpy(buf, msg.data(), n);
        buf[n] = 0;
    }
}

int vdb_flat_create(int metric, int device, vdb_flat_index** out) {
    return guarded([&]() -> int {
    if (!out) return fail(VDB_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (metric < 0 || metric > 2) return fail(VDB_ERR_INVALID_ARGUMENT, "unknown metric %d", metric);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(VDB_ERR_DEVICE, "no HIP device available: this engine has no CPU path");
    if (device < 0 || device >= ndev) return fail(VDB_ERR_INVALID_ARGUMENT, "device %d out of range (%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VDB_ERR_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
    auto* ix = new vdb_flat_index();
    ix->metric = metric;
    ix->device = device;
    ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#ifdef VDB_DIAG
    {
        vdb_knobs& kn = ix->kn;
        auto num = [&](const char* name, uint32_t& dst) { if (const char* e = getenv(name)) { dst = (uint32_t)std::max(0, atoi(e)); kn.any = true; } };
        auto flag = [&](const char* name, bool& dst) { if (getenv(name)) { dst = true; kn.any = true; } };
        if (const char* e = getenv("VDB_EPS_SCALE")) { kn.eps_scale = atof(e); kn.any = true; }
        num("VDB_BF16_ABLATE", kn.bf16_ablate); num("VDB_FUSED_ABLATE", kn.fused_ablate);
        num("VDB_KT16", kn.kt16); num("VDB_SAMPLE16", kn.sample16); num("VDB_SAMPLE", kn.sample); num("VDB_KP_FIRST", kn.kp_first);
        flag("VDB_RR_DEPTH", kn.rr_depth); flag("VDB_SAMPLE_BLOCK", kn.sample_block);
        flag("VDB_FUSED_SHAPE4", kn.shape4); flag("VDB_FUSED_REGSTAGE", kn.regstage); flag("VDB_FUSED_DMA2", kn.dma2);
        if (const char* e = getenv("VDB_FUSED_PIPE")) { kn.fused_pipe = strcmp(e, "0") != 0; kn.any = true; }
    }
#endif
    if (hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ix;
        return fail(VDB_ERR_DEVICE, "hipStreamCreate failed");
    }
    if (hipMalloc((void**)&ix->d_scalars, 32) != hipSuccess || hipMemset(ix->d_scalars, 0, 32) != hipSuccess) {
        (void)hipStreamDestroy(ix->stream);
        delete ix;
        return fail(VDB_ERR_DEVICE, "hipMalloc failed");
    }
    ix->wsv = new Workspace[2];
    ix->cur = &ix->wsv[0];
    ix->wsv[0].stream = ix->stream;
    if (hipStreamCreateWithFlags(&ix->wsv[1].stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipFree(ix->d_scalars);
        (void)hipStreamDestroy(ix->stream);
        delete[] ix->wsv;
        delete ix;
        return fail(VDB_ERR_DEVICE, "hipStreamCreate failed");
    }
    *out = ix;
    return VDB_OK;
    });
}

void vdb_flat_destroy(vdb_flat_index* ix) {
    if (!ix) return;
    if (ix->multi) { multi_destroy(ix); return; }
    (void)hipSetDevice(ix->device);
    (void)hipStreamSynchronize(ix->stream);
    free_store(ix);
    if (ix->d_scalars) (void)hipFree(ix->d_scalars);
    ix->d_idrank.release(); ix->d_rank2row.release();
    for (int w = 0; ix->wsv && w < 2; ++w) {
        Workspace& W = ix->wsv[w];
        W.for_each_buffer([](auto& buf) { buf.release(); });
        if (W.h_flags) (void)hipHostFree(W.h_flags);
        if (W.h_dstat) (void)hipHostFree(W.h_dstat);
        if (W.h_io) (void)hipHostFree(W.h_io);
        if (w == 1 && W.stream) { (void)hipStreamSynchronize(W.stream); (void)hipStreamDestroy(W.stream); }
    }
    delete[] ix->wsv;
    if (ix->h_pairs) (void)hipHostFree(ix->h_pairs);
    if (ix->h_pout) (void)hipHostFree(ix->h_pout);
    if (ix->ev0) { (void)hipEventDestroy(ix->ev0); (void)hipEventDestroy(ix->ev1); }
    if (ix->ev_order) (void)hipEventDestroy(ix->ev_order);
    for (int t = 0; t < 2; ++t) if (ix->ev_pass[t]) (void)hipEventDestroy(ix->ev_pass[t]);
    (void)hipStreamDestroy(ix->stream);
    delete ix;
}

int vdb_flat_add(vdb_flat_index* ix, uint64_t id, const float* v, size_t dim) {
    return guarded([&]() -> int {
    if (!ix || (!v && dim)) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (ix->multi) return multi_add(ix, id, v, dim);
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc = set_device(ix);
    if (rc) return rc;
    return add_one(ix, id, v, dim);
    });
}

int vdb_flat_add_bulk(vdb_flat_index* ix, const uint64_t* ids, uint64_t first_id, const float* rows, size_t n,
                      size_t dim) {
    return guarded([&]() -> int {
    if (!ix || (!rows && n && dim)) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (ix->multi) return multi_add_bulk(ix, ids, first_id, rows, n, dim, false);
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc = set_device(ix);
    if (rc) return rc;
    if (ix->n_rows() + n > 0xfffffff0ull) return fail(VDB_ERR_INVALID_ARGUMENT, "more than 2^32 rows per index");
    ix->row_ids.reserve(ix->row_ids.size() + n);
    for (size_t i = 0; i < n; ++i) {
        rc = add_one(ix, ids ? ids[i] : first_id + i, rows + i * dim, dim);
        if (rc) return rc;
        // bound the host staging area: upload every 64 MiB
        if (ix->pending.size() * sizeof(float) >= (64u << 20)) {
            if ((rc = flush(ix))) return rc;
        }
    }
    return VDB_OK;
    });
}

int vdb_flat_add_bulk_device(vdb_flat_index* ix, const uint64_t* ids, uint64_t first_id, const float* d_rows,
                             size_t n, size_t dim) {
    return guarded([&]() -> int {
    if (!ix || (!d_rows && n)) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return VDB_OK;
    if (dim == 0) return fail(VDB_ERR_INVALID_ARGUMENT, "dim must be > 0");
    if (ix->multi) return multi_add_bulk(ix, ids, first_id, d_rows, n, dim, true);
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc = set_device(ix);
    if (rc) return rc;
    if (ix->n_rows() + n > 0xfffffff0ull) return fail(VDB_ERR_INVALID_ARGUMENT, "more than 2^32 rows per index");
    if (ix->n_live == 0 && ix->misfits.empty()) {
        if (ix->dim != dim) reset_rows(ix);
        ix->dim = (uint32_t)dim;
        ix->ld = round_up(ix->dim, vdb::KSTAGE);
    }
    if (dim != ix->dim) return fail_dim(ix->dim, dim);   // the device bulk path requires the index dimension
    // overwrite semantics for ids already present
    for (size_t i = 0; i < n; ++i) {
        uint64_t id = ids ? ids[i] : first_id + i;
        if (!ix->id2row.empty() || !ix->misfits.empty()) remove_id(ix, id);
    }
    if (ix->dim != dim) {   // remove_id may have emptied and reset the index
        ix->dim = (uint32_t)dim;
        ix->ld = round_up(ix->dim, vdb::KSTAGE);
    }
    if ((rc = flush(ix))) return rc;
    uint32_t first = ix->n_rows();
    if ((rc = grow(ix, first + (uint32_t)n))) return rc;
    hipStream_t s = ix->stream;
    HIP_TRY(hipMemcpy2DAsync(ix->d_rows + (size_t)first * ix->ld, (size_t)ix->ld * 4, d_rows, dim * 4, dim * 4, n,
                             hipMemcpyDeviceToDevice, s));
    ix->row_ids.reserve(first + n);
    ix->id2row.reserve(first + n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t id = ids ? ids[i] : first_id + i;
        uint32_t row = first + (uint32_t)i;
        if (row && id <= ix->row_ids.back()) ix->ids_monotone = false;
        ix->row_ids.push_back(id);
        if ((row >> 5) >= ix->live.size()) ix->live.push_back(0u);
        ix->live[row >> 5] |= 1u << (row & 31);
        ++ix->n_live;
        // the same id twice in ONE batch: HashMap::insert is last-wins (flat_index.rs:38-41) -- the earlier row of this
        // call dies (ids stored before the call were removed above)
        auto ins = ix->id2row.emplace(id, row);
        if (!ins.second) { kill_row(ix, ins.first->second); ins.first->second = row; }
    }
    HIP_TRY(hipMemcpyAsync(ix->d_row_ids + first, ix->row_ids.data() + first, n * 8, hipMemcpyHostToDevice, s));
    const MarginPlan mp = margin_plan(ix);
    vdb::RowStatsParams rp{ix->d_rows, ix->ld, ix->dim, first, first + (uint32_t)n, ix->metric, ix->d_nd,
                           ix->d_alpha, ix->d_beta, ix->d_scalars, ix->d_margin, mp.m_e, mp.m_n, mp.m_b, mp.beta_shrink};
    vdb::launch_row_stats(rp, s);
    if (ix->d_rows16) vdb::launch_rows_to_bf16(ix->d_rows, ix->d_rows16, ix->ld, first, first + (uint32_t)n, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    ix->n_uploaded = first + (uint32_t)n;
    ix->live_dirty = true;
    ix->zero_valid = false;
    ix->rank_valid = false;
    return VDB_OK;
    });
}

int vdb_flat_load_vector_file(vdb_flat_index* ix, const char* path, uint64_t first_id, size_t* out_count) {
    return guarded([&]() -> int {
    if (!ix || !path) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (out_count) *out_count = 0;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(VDB_ERR_INVALID_ARGUMENT, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) {
        close(fd);
        return fail(VDB_ERR_INVALID_ARGUMENT, "File too small for header");          // mmap.rs:52-54
    }
    void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(VDB_ERR_INVALID_ARGUMENT, "mmap of %s failed", path);
    const unsigned char* b = (const unsigned char*)map;
    auto le32 = [&](size_t o) { return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8) | ((uint32_t)b[o + 2] << 16) | ((uint32_t)b[o + 3] << 24); };
    const size_t dim = le32(0), count = le32(4);                                      // mmap.rs:161-172
    int rc = VDB_OK;
    if (dim == 0 && count) rc = fail(VDB_ERR_INVALID_ARGUMENT, "vector file with dimension 0");
    else if (dim > 16384) rc = fail(VDB_ERR_INVALID_ARGUMENT, "vector file dimension %zu exceeds the supported 16384", dim);
    // (count * dim * 4 can wrap a size_t -- both come from the file -- so the check divides instead)
    else if (dim && count > ((size_t)st.st_size - 8) / 4 / dim) rc = fail(VDB_ERR_INVALID_ARGUMENT, "vector file truncated: %zu rows of %zu floats do not fit %zu bytes", count, dim, (size_t)st.st_size);
    else if (count) {
        // the body starts at byte 8, so rows are 4-byte aligned; x86 is little-endian like the file
        rc = vdb_flat_add_bulk(ix, nullptr, first_id, (const float*)(b + 8), count, dim);
        if (rc == VDB_OK) rc = vdb_flat_flush(ix);
    }
    munmap(map, (size_t)st.st_size);
    if (rc == VDB_OK && out_count) *out_count = count;
    return rc;
    });
}

int vdb_flat_remove(vdb_flat_index* ix, uint64_t id) {
    return guarded([&]() -> int {
    if (!ix) return fail(VDB_ERR_INVALID_ARGUMENT, "null handle");
    if (ix->multi) return multi_remove(ix, id);
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc = set_device(ix);
    if (rc) return rc;
    return remove_id(ix, id);
    });
}

int vdb_flat_get_vector(vdb_flat_index* ix, uint64_t id, float* out, size_t cap, size_t* dim) {
    return guarded([&]() -> int {
    if (!ix) return fail(VDB_ERR_INVALID_ARGUMENT, "null handle");
    if (ix->multi) return multi_get_vector(ix, id, out, cap, dim);
    std::lock_guard<std::mutex> g(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    auto m = ix->misfits.find(id);
    if (m != ix->misfits.end()) {
        if (dim) *dim = m->second.size();
        if (out) memcpy(out, m->second.data(), std::min(cap, m->second.size()) * sizeof(float));
        return VDB_OK;
    }
    auto it = ix->id2row.find(id);
    if (it == ix->id2row.end()) return fail(VDB_ERR_NOT_FOUND, "Vector not found: %llu", (unsigned long long)id);
    if (dim) *dim = ix->dim;
    if (!out) return VDB_OK;
    size_t ncopy = std::min<size_t>(cap, ix->dim);
    uint32_t row = it->second;
    if (row >= ix->n_uploaded) {
        memcpy(out, ix->pending.data() + (size_t)(row - ix->n_uploaded) * ix->ld, ncopy * sizeof(float));
    } else {
        HIP_TRY(hipMemcpy(out, ix->d_rows + (size_t)row * ix->ld, ncopy * sizeof(float), hipMemcpyDeviceToHost));
    }
    return VDB_OK;
    });
}

size_t vdb_flat_len(const vdb_flat_index* ix) { return !ix ? 0 : ix->multi ? multi_len(ix) : ix->n_live + ix->misfits.size(); }
int vdb_flat_metric(const vdb_flat_index* ix) { return ix ? ix->metric : -1; }
size_t vdb_flat_dim(const vdb_flat_index* ix) { return !ix ? 0 : ix->multi ? multi_dim(ix) : ix->dim; }

int vdb_flat_reserve(vdb_flat_index* ix, size_t rows, size_t dim) {
    return guarded([&]() -> int {
    if (!ix || !dim) return fail(VDB_ERR_INVALID_ARGUMENT, "bad argument");
    if (ix->multi) return multi_reserve(ix, rows, dim);
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc = set_device(ix);
    if (rc) return rc;
    if (ix->n_live == 0 && ix->misfits.empty()) {
        if (ix->dim != dim) reset_rows(ix);
        ix->dim = (uint32_t)dim;
        ix->ld = round_up(ix->dim, vdb::KSTAGE);
    }
    if (dim != ix->dim) return fail_dim(ix->dim, dim);
    if (rows > 0xfffffff0ull) return fail(VDB_ERR_INVALID_ARGUMENT, "more than 2^32 rows per index");
    if ((rc = flush(ix))) return rc;
    return grow(ix, (uint32_t)rows);
    });
}

int vdb_flat_flush(vdb_flat_index* ix) {
    return guarded([&]() -> int {
    if (!ix) return fail(VDB_ERR_INVALID_ARGUMENT, "null handle");
    if (ix->multi) return multi_for_each(ix, [](vdb_flat_index* c) { return vdb_flat_flush(c); });
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc = set_device(ix);
    if (rc) return rc;
    if ((rc = flush(ix))) return rc;
    if (ix->metric == vdb::COSINE && ix->n_uploaded) return ensure_zero_count(ix);
    return VDB_OK;
    });
}

int vdb_flat_search_batch_device(vdb_flat_index* ix, const float* d_queries, size_t nq, size_t dim, size_t k,
                                 const uint64_t* d_id_mask, size_t mask_bits, uint64_t* d_out_ids,
                                 float* d_out_dists, uint32_t* d_out_counts, void* stream) {
    return guarded([&]() -> int {
    if (!ix || (nq && (!d_queries || !d_out_counts || (k && (!d_out_ids || !d_out_dists)))))
        return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (ix->multi) return multi_search_device(ix, d_queries, nq, dim, k, d_id_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts, (hipStream_t)stream);
    std::lock_guard<std::mutex> g(ix->mu);
    return search_device(ix, d_queries, nq, dim, k, d_id_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts,
                         (hipStream_t)stream);
    });
}

int vdb_flat_search_batch_device_begin(vdb_flat_index* ix, const float* d_queries, size_t nq, size_t dim, size_t k,
                                       const uint64_t* d_id_mask, size_t mask_bits, uint64_t* d_out_ids,
                                       float* d_out_dists, uint32_t* d_out_counts, int32_t* d_code, void* stream) {
    return guarded([&]() -> int {
    if (!ix || (nq && (!d_queries || !d_out_counts || (k && (!d_out_ids || !d_out_dists)))))
        return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (ix->multi) return refuse_multi("vdb_flat_search_batch_device_begin");
    ix->mu.lock();
    if (ix->begin_locked) { ix->mu.unlock(); return fail(VDB_ERR_INVALID_ARGUMENT, "a search is already pending on this handle"); }
    if (in_flight(ix)) { ix->mu.unlock(); return refuse_in_flight(); }
    ix->cur = &ix->wsv[0];
    // (the handle is locked by hand here: an exception must not skip the unlock below)
    int rc = guarded([&]() -> int { return search_part1(ix, d_queries, nq, dim, k, d_id_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts, (hipStream_t)stream, true); });
    if (rc == VDB_OK && d_code) {
        hipStream_t s = stream ? (hipStream_t)stream : ix->stream;
        // (a forced hand-over to the slower tiers -- vdb_flat_set_tiers, tests -- rewrites the outputs in _finish whatever the
        // first tier certified: the word says "pending" then, so that a caller exchanging partial results exchanges again)
        if (ix->cur->ctx.pending && (ix->tiers & (VDB_TIERS_FORCE_EXACT | VDB_TIERS_FORCE_F32))) {
            if (hipMemsetD32Async((hipDeviceptr_t)d_code, VDB_PENDING_HOST, 1, s) != hipSuccess) rc = fail(VDB_ERR_DEVICE, "hipMemsetD32Async failed");
        } else if (ix->cur->ctx.pending) vdb::launch_write_code(ix->cur->w_flags.p, d_code, s);
        else if (hipMemsetAsync(d_code, 0, 4, s) != hipSuccess) rc = fail(VDB_ERR_DEVICE, "hipMemsetAsync failed");
    }
    if (rc == VDB_OK && !stream) {
        // the work went to the handle's own (non-blocking) stream: whatever the caller enqueues next on the null stream
        // -- the exchange -- must wait for it
        if (!ix->ev_order && hipEventCreateWithFlags(&ix->ev_order, hipEventDisableTiming) != hipSuccess) rc = fail(VDB_ERR_DEVICE, "hipEventCreate failed");
        if (rc == VDB_OK && (hipEventRecord(ix->ev_order, ix->stream) != hipSuccess || hipStreamWaitEvent(nullptr, ix->ev_order, 0) != hipSuccess))
            rc = fail(VDB_ERR_DEVICE, "stream ordering failed");
    }
    if (rc) {
        if (ix->cur->ctx.pending) (void)hipStreamSynchronize(ix->cur->ctx.s);
        ix->cur->ctx.pending = false; ix->mu.unlock(); return rc;
    }
    ix->begin_locked = true;                                   // released by vdb_flat_search_batch_device_finish (same thread)
    return VDB_OK;
    });
}

int vdb_flat_search_batch_device_finish(vdb_flat_index* ix, int* changed) {
    return guarded([&]() -> int {
    if (!ix) return fail(VDB_ERR_INVALID_ARGUMENT, "null handle");
    if (ix->multi) return refuse_multi("vdb_flat_search_batch_device_finish");
    if (!ix->begin_locked) return fail(VDB_ERR_INVALID_ARGUMENT, "no search pending on this handle");
    int rc = guarded([&]() -> int { return search_part2(ix, changed); });
    publish_stats(ix);
    ix->cur->ctx.pending = false;
    ix->begin_locked = false;
    ix->mu.unlock();
    return rc;
    });
}

int vdb_flat_search_batch_device_submit(vdb_flat_index* ix, const float* d_queries, size_t nq, size_t dim, size_t k,
                                        const uint64_t* d_id_mask, size_t mask_bits, uint64_t* d_out_ids, float* d_out_dists,
                                        uint32_t* d_out_counts, void* stream, int* ticket) {
    return guarded([&]() -> int {
    if (!ix || !ticket || (nq && (!d_queries || !d_out_counts || (k && (!d_out_ids || !d_out_dists)))))
        return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (ix->multi) return refuse_multi("vdb_flat_search_batch_device_submit");
    *ticket = -1;
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->begin_locked) return fail(VDB_ERR_INVALID_ARGUMENT, "a search is pending between begin and finish");
    if (ix->profile) return fail(VDB_ERR_INVALID_ARGUMENT, "kernel profiling synchronises inside the search: not available for submitted searches");
    int slot = !ix->wsv[0].busy ? 0 : (!ix->wsv[1].busy ? 1 : -1);
    if (slot < 0) return fail(VDB_ERR_INVALID_ARGUMENT, "two searches are already in flight on this handle");
    ix->cur = &ix->wsv[slot];
    // (an exception must not leave ix->cur pointing at the ticket's workspace)
    int rc = guarded([&]() -> int { return search_part1(ix, d_queries, nq, dim, k, d_id_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts, (hipStream_t)stream); });
    if (rc) {
        // part 1 may have failed AFTER kernels were enqueued (a later allocation, a launch error): they could still write the
        // caller's buffers after this return -- wait for them, as _begin does
        hipStream_t s = stream ? (hipStream_t)stream : ix->cur->stream;
        if (s) (void)hipStreamSynchronize(s);
        if (ix->wsv[slot ^ 1].stream && !ix->wsv[slot ^ 1].busy) (void)hipStreamSynchronize(ix->wsv[slot ^ 1].stream);   // alternating passes of a large batch
        ix->cur->ctx.pending = false; ix->cur = &ix->wsv[0]; return rc;
    }
    ix->cur->busy = true;                       // (a search part 1 answered completely has pending = false: wait() returns at once)
    ix->cur = &ix->wsv[0];
    *ticket = slot;
    return VDB_OK;
    });
}

int vdb_flat_search_batch_device_wait(vdb_flat_index* ix, int ticket) {
    return guarded([&]() -> int {
    if (!ix || ticket < 0 || ticket > 1) return fail(VDB_ERR_INVALID_ARGUMENT, "bad ticket");
    if (ix->multi) return refuse_multi("vdb_flat_search_batch_device_wait");
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->wsv[ticket].busy) return fail(VDB_ERR_INVALID_ARGUMENT, "no submitted search behind this ticket");
    int rc = set_device(ix);
    if (rc) return rc;
    ix->cur = &ix->wsv[ticket];
    // (whatever part 2 does -- an exception from its host vectors included -- the ticket is retired: a workspace left busy
    // would refuse add / remove / flush on the handle for ever)
    rc = guarded([&]() -> int { return search_part2(ix, nullptr); });
    publish_stats(ix);
    ix->cur->ctx.pending = false;
    ix->cur->busy = false;
    ix->cur = &ix->wsv[0];
    return rc;
    });
}

int vdb_flat_search_batch(vdb_flat_index* ix, const float* queries, size_t nq, size_t dim, const size_t* ks,
                          size_t k, const uint64_t* id_mask, size_t mask_bits, size_t kstride, uint64_t* out_ids,
                          float* out_dists, size_t* out_counts) {
    return guarded([&]() -> int {
    if (!ix || (nq && (!queries || !out_counts))) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (ix->multi) return multi_search_host(ix, queries, nq, dim, ks, k, id_mask, mask_bits, kstride, out_ids, out_dists, out_counts);
    size_t kmax = k;
    if (ks) {
        kmax = 0;
        for (size_t b = 0; b < nq; ++b) kmax = std::max(kmax, ks[b]);
    }
    if (kmax > kstride) return fail(VDB_ERR_INVALID_ARGUMENT, "kstride %zu smaller than the largest k %zu", kstride, kmax);
    if (kmax && nq && (!out_ids || !out_dists)) return fail(VDB_ERR_INVALID_ARGUMENT, "null output");
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc = set_device(ix);
    if (rc) return rc;
    if (nq == 0) return VDB_OK;
    // Index::search returns at most len results; clamp before sizing device buffers
    size_t len = ix->n_live + ix->misfits.size();
    size_t kdev = std::min(kmax, std::max<size_t>(len, 1));
    hipStream_t s = ix->stream;
    // Small index, a few queries (BASELINE configs[0]: Index::search itself, one query): queries and results go through MAPPED
    // host memory -- the two kernels of the direct path read and write it in place, nothing is copied by the runtime
    if (direct_eligible(ix, ix->n_rows(), nq, kdev) && ix->misfits.empty() && dim == ix->dim && !id_mask) {
        const size_t qb = nq * dim * sizeof(float), ib = nq * kdev * sizeof(uint64_t), db = nq * kdev * sizeof(float), cb = nq * sizeof(uint32_t);
        const size_t o_i = (qb + 15) & ~(size_t)15, o_d = o_i + ib, o_c = o_d + ((db + 15) & ~(size_t)15);
        if ((rc = ensure_host_io(ix, o_c + cb))) return rc;
        Workspace* W = ix->cur;
        memcpy(W->h_io, queries, qb);
        rc = search_device(ix, reinterpret_cast<const float*>(W->d_h_io), nq, dim, kdev, nullptr, 0, reinterpret_cast<uint64_t*>(W->d_h_io + o_i),
                           reinterpret_cast<float*>(W->d_h_io + o_d), reinterpret_cast<uint32_t*>(W->d_h_io + o_c), nullptr);
        if (rc) return rc;
        W = &ix->wsv[0];                                           // (search_device leaves ix->cur at workspace 0, the one it used)
        const uint64_t* h_ids = reinterpret_cast<const uint64_t*>(W->h_io + o_i);
        const float* h_ds = reinterpret_cast<const float*>(W->h_io + o_d);
        const uint32_t* h_cnt = reinterpret_cast<const uint32_t*>(W->h_io + o_c);
        for (size_t b = 0; b < nq; ++b) {
            const size_t kb = ks ? ks[b] : k;
            const size_t c = std::min<size_t>(h_cnt[b], kb);       // per-query k: a prefix of the batch-wide result
            out_counts[b] = c;
            for (size_t i = 0; i < c; ++i) { out_ids[b * kstride + i] = h_ids[b * kdev + i]; out_dists[b * kstride + i] = h_ds[b * kdev + i]; }
        }
        return VDB_OK;
    }
    if ((rc = ix->cur->w_qin.ensure(nq * std::max<size_t>(dim, 1)))) return rc;
    if ((rc = ix->cur->w_outi.ensure(nq * std::max<size_t>(kdev, 1)))) return rc;
    if ((rc = ix->cur->w_outd.ensure(nq * std::max<size_t>(kdev, 1)))) return rc;
    if ((rc = ix->cur->w_outc.ensure(nq))) return rc;
    if (dim) HIP_TRY(hipMemcpyAsync(ix->cur->w_qin.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
    const uint64_t* d_mask = nullptr;
    if (id_mask) {
        size_t words = (mask_bits + 63) / 64;
        if ((rc = ix->cur->w_mask_ids.ensure(std::max<size_t>(words, 1)))) return rc;
        if (words) HIP_TRY(hipMemcpyAsync(ix->cur->w_mask_ids.p, id_mask, words * 8, hipMemcpyHostToDevice, s));
        d_mask = ix->cur->w_mask_ids.p;
    }
    rc = search_device(ix, ix->cur->w_qin.p, nq, dim, kdev, d_mask, mask_bits, ix->cur->w_outi.p, ix->cur->w_outd.p, ix->cur->w_outc.p,
                       nullptr);
    if (rc) return rc;
    std::vector<uint32_t> cnt(nq);
    std::vector<uint64_t> ids(nq * std::max<size_t>(kdev, 1));
    std::vector<float> ds(nq * std::max<size_t>(kdev, 1));
    HIP_TRY(hipMemcpyAsync(cnt.data(), ix->cur->w_outc.p, nq * 4, hipMemcpyDeviceToHost, s));
    if (kdev) {
        HIP_TRY(hipMemcpyAsync(ids.data(), ix->cur->w_outi.p, nq * kdev * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ds.data(), ix->cur->w_outd.p, nq * kdev * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t b = 0; b < nq; ++b) {
        size_t kb = ks ? ks[b] : k;
        size_t c = std::min<size_t>(cnt[b], kb);   // per-query k: a prefix of the batch-wide result
        out_counts[b] = c;
        for (size_t i = 0; i < c; ++i) {
            out_ids[b * kstride + i] = ids[b * kdev + i];
            out_dists[b * kstride + i] = ds[b * kdev + i];
        }
    }
    return VDB_OK;
    });
}

int vdb_flat_search(vdb_flat_index* ix, const float* query, size_t dim, size_t k, uint64_t* out_ids,
                    float* out_dists, size_t* out_count) {
    return guarded([&]() -> int {
    if (!out_count) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    return vdb_flat_search_batch(ix, query, 1, dim, nullptr, k, nullptr, 0, k, out_ids, out_dists, out_count);
    });
}

int vdb_merge_topk_device(int device, const uint64_t* d_part_ids, const float* d_part_dists,
                          const uint32_t* d_part_counts, size_t nparts, size_t nq, size_t k, uint64_t* d_out_ids,
                          float* d_out_dists, uint32_t* d_out_counts, void* stream) {
    return guarded([&]() -> int {
    if (!d_part_ids || !d_part_dists || !d_part_counts || !d_out_ids || !d_out_dists || !d_out_counts)
        return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (nparts * k > 2048) return fail(VDB_ERR_INVALID_ARGUMENT, "nparts*k = %zu exceeds 2048", nparts * k);
    HIP_TRY(hipSetDevice(device));
    vdb::launch_merge_parts(d_part_ids, d_part_dists, d_part_counts, (uint32_t)nparts, (uint32_t)nq, (uint32_t)k,
                            d_out_ids, d_out_dists, d_out_counts, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return VDB_OK;
    });
}

int vdb_flat_distances_batch(vdb_flat_index* ix, const float* queries, size_t nq, size_t dim, const size_t* offsets,
                             const uint64_t* ids, float* out_dists) {
    return guarded([&]() -> int {
    if (!ix || !offsets || (nq && !queries)) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (ix->multi) return refuse_multi("vdb_flat_distances_batch");
    const size_t total = offsets[nq];
    if (total && (!ids || !out_dists)) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc = set_device(ix);
    if (rc) return rc;
    if ((rc = flush(ix))) return rc;
    if (total == 0) return VDB_OK;
    if (total > 0xfffffff0ull || nq > 0x7fffffffull) return fail(VDB_ERR_INVALID_ARGUMENT, "too many pairs");
    // distance.rs:21-26: a stored row of another dimension fails the call
    for (size_t i = 0; i < total; ++i) {
        auto m = ix->misfits.find(ids[i]);
        if (m != ix->misfits.end() && m->second.size() != dim) return fail_dim(dim, m->second.size());
    }
    if (ix->n_live && ix->dim != dim) return fail_dim(dim, ix->dim);
    hipStream_t s = ix->stream;
    const uint32_t ld = ix->ld ? ix->ld : round_up((uint32_t)dim, vdb::KSTAGE);
    const uint32_t nq32 = (uint32_t)nq, bp = round_up(nq32, SUPER);
    std::vector<uint32_t> prow(total), pq(total);
    for (uint32_t q = 0; q < nq32; ++q)
        for (size_t i = offsets[q]; i < offsets[q + 1]; ++i) {
            auto it = ix->id2row.find(ids[i]);
            prow[i] = it == ix->id2row.end() ? 0xffffffffu : it->second;
            pq[i] = q;
        }
    if ((rc = ix->cur->w_qin.ensure(nq * dim))) return rc;
    if ((rc = ix->cur->w_qp.ensure((size_t)bp * ld))) return rc;
    if ((rc = ix->cur->w_qnorm.ensure(bp))) return rc;
    if ((rc = ix->cur->w_thr.ensure(bp))) return rc;
    if ((rc = ix->cur->w_flags.ensure(4))) return rc;
    if ((rc = ix->cur->w_rowmask.ensure(2 * total))) return rc;      // pair_row | pair_query
    if ((rc = ix->cur->w_outd.ensure(total))) return rc;
    HIP_TRY(hipMemsetAsync(ix->cur->w_flags.p, 0, 16, s));
    HIP_TRY(hipMemcpyAsync(ix->cur->w_qin.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ix->cur->w_rowmask.p, prow.data(), total * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ix->cur->w_rowmask.p + total, pq.data(), total * 4, hipMemcpyHostToDevice, s));
    vdb::QueryPrepParams qp{ix->cur->w_qin.p, (uint32_t)dim, nq32, ix->cur->w_qp.p, ld, bp, ix->cur->w_qnorm.p, ix->cur->w_thr.p, vdb::EUCLID,
                            ix->cur->w_flags.p, nullptr, nullptr, nullptr, 0.0f, nullptr, nullptr};   // metric EUCLID here: zero norms are judged per PAIR below
    vdb::launch_query_prep(qp, s);
    vdb::PairDistParams pp{ix->d_rows, ld, (uint32_t)dim, ix->cur->w_qp.p, ix->cur->w_qnorm.p, ix->d_nd, ix->cur->w_rowmask.p + total,
                           ix->cur->w_rowmask.p, (uint32_t)total, ix->metric, ix->cur->w_outd.p, ix->cur->w_flags.p};
    vdb::launch_pair_distances(pp, s);
    HIP_TRY(hipGetLastError());
    uint32_t st = 0;
    HIP_TRY(hipMemcpyAsync(out_dists, ix->cur->w_outd.p, total * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&st, ix->cur->w_flags.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (st & vdb::ST_ZERO_QUERY)
        return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
    return VDB_OK;   // a NaN distance is returned as NaN (only the sort in FlatIndex::search panics on it)
    });
}

int vdb_merge_topk_packed_device(int device, const int32_t* d_packed, size_t nparts, size_t words_per_part, size_t nq,
                                 size_t k, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                                 uint32_t* d_out_status, void* stream) {
    return guarded([&]() -> int {
    if (!d_packed || !d_out_ids || !d_out_dists || !d_out_counts)
        return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (nparts * k > 2048) return fail(VDB_ERR_INVALID_ARGUMENT, "nparts*k = %zu exceeds 2048", nparts * k);
    if ((words_per_part & 1) || words_per_part < nq * (3 * k + 1) + 1)
        return fail(VDB_ERR_INVALID_ARGUMENT, "words_per_part must be even and >= nq*(3k+1)+1");
    HIP_TRY(hipSetDevice(device));
    vdb::launch_merge_packed(d_packed, words_per_part, (uint32_t)nparts, (uint32_t)nq, (uint32_t)k, d_out_ids,
                             d_out_dists, d_out_counts, d_out_status, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return VDB_OK;
    });
}

int vdb_flat_set_profile(vdb_flat_index* ix, int on) {
    return guarded([&]() -> int {
    if (!ix) return fail(VDB_ERR_INVALID_ARGUMENT, "null handle");
    if (ix->multi) return multi_for_each(ix, [on](vdb_flat_index* c) { return vdb_flat_set_profile(c, on); });
    std::lock_guard<std::mutex> g(ix->mu);
    int rc = set_device(ix);
    if (rc) return rc;
    if (on && !ix->ev0) {
        HIP_TRY(hipEventCreate(&ix->ev0));
        HIP_TRY(hipEventCreate(&ix->ev1));
    }
    ix->profile = on != 0;
    return VDB_OK;
    });
}

int vdb_flat_last_stats(const vdb_flat_index* ix, uint64_t out[8]) {
    return guarded([&]() -> int {
    if (!ix || !out) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(out, ix->stats, 8 * sizeof(uint64_t));
    return VDB_OK;
    });
}

int vdb_flat_last_stats_ex(const vdb_flat_index* ix, uint64_t* out, size_t n) {
    return guarded([&]() -> int {
    if (!ix || !out) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    for (size_t i = 0; i < n; ++i) out[i] = i < 16 ? (ix->multi ? ix->stats[i] : ix->cur->stats[i]) : 0;
    return VDB_OK;
    });
}

// ------------------------------------------------------------------ certificate diagnostics (include/vdb_flat.h)
int vdb_flat_debug_screen_scores(vdb_flat_index* ix, const float* queries, size_t nq, size_t dim, int raw, float* out_scores,
                                 float* out_qinfo, double* out_consts) {
    return guarded([&]() -> int {
    if (ix && ix->multi) return refuse_multi("vdb_flat_debug_screen_scores");
    if (!ix || !queries || !out_scores || !nq) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (nq > SUPER) return fail(VDB_ERR_INVALID_ARGUMENT, "at most %u queries per call", SUPER);
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc;
    if ((rc = set_device(ix))) return rc;
    if ((rc = flush(ix))) return rc;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    if (!n) return fail(VDB_ERR_INVALID_ARGUMENT, "empty index");
    if (ix->dim != dim) return fail_dim(dim, ix->dim);
    if (!ix->misfits.empty()) return fail(VDB_ERR_INVALID_ARGUMENT, "rows of another dimension are stored");
    hipStream_t s = ix->stream;
    const uint32_t tile = vdb::fused_bf16_tile_rows();
    const uint32_t nblk = (n + tile - 1) / tile;
    const uint32_t n_wg = std::min<uint32_t>((uint32_t)ix->n_cu, nblk);
    const uint32_t n_sub = vdb::fused_bf16_subpools_per_query(n_wg);
    const uint32_t capl = 64u * ((nblk + n_wg - 1) / n_wg);            // every row of a workgroup's range fits its sub-pools
    const size_t pool_keys = (size_t)SUPER * n_sub * capl;
    if (pool_keys * 8 > ((size_t)6 << 30)) return fail(VDB_ERR_INVALID_ARGUMENT, "index too large for the score dump");
    if ((rc = ix->cur->w_qin.ensure(nq * dim))) return rc;
    if ((rc = ix->cur->w_qp.ensure((size_t)SUPER * ld))) return rc;
    if ((rc = ix->cur->w_qnorm.ensure(SUPER))) return rc;
    if ((rc = ix->cur->w_thr.ensure(SUPER))) return rc;
    if ((rc = ix->cur->w_qb.ensure((size_t)SUPER * ld))) return rc;
    if ((rc = ix->cur->w_qerr.ensure(SUPER))) return rc;
    if ((rc = ix->cur->w_qg.ensure(SUPER))) return rc;
    if ((rc = ix->cur->w_flags.ensure(4 + 3 * (size_t)SUPER))) return rc;
    if ((rc = ix->cur->w_pool.ensure(pool_keys))) return rc;
    if ((rc = ix->cur->w_subcnt.ensure((size_t)SUPER * n_sub))) return rc;
    if ((rc = ix->cur->w_dbg.ensure(nq * (size_t)n))) return rc;
    ix->cur->status_dirty = true;
    HIP_TRY(hipMemsetAsync(ix->cur->w_flags.p, 0, 16, s));
    HIP_TRY(hipMemcpyAsync(ix->cur->w_qin.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
    const bool lb = ix->d_margin && raw == 0;
    if (raw == 2) {
        // the f32 MFMA tier's scores: dense_scores_kernel over EVERY row -- the production kernel of indexes up to 16384 rows,
        // and bit-identical to the fused f32 kernel's scores by construction (same K order, same score expression)
        if ((size_t)nq * n > ((size_t)1 << 28)) return fail(VDB_ERR_INVALID_ARGUMENT, "index too large for the f32-tier score dump");
        if ((rc = ix->cur->w_dense.ensure((size_t)SUPER * n))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->cur->w_qin.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(ix->cur->w_flags.p, 0, 16, s));
        vdb::QueryPrepParams qp2{ix->cur->w_qin.p, (uint32_t)dim, (uint32_t)nq, ix->cur->w_qp.p, ld, SUPER, ix->cur->w_qnorm.p, ix->cur->w_thr.p, vdb::EUCLID,
                                 ix->cur->w_flags.p, nullptr, nullptr, nullptr, 0.0f, nullptr, nullptr};
        vdb::launch_query_prep(qp2, s);
        vdb::DenseParams dp{ix->d_rows, ld, n, ix->cur->w_qp.p, round_up((uint32_t)nq, 32), ix->d_alpha, ix->d_beta, ix->d_live, n, ix->cur->w_dense.p, n};
        vdb::launch_dense_scores(dp, s);
        HIP_TRY(hipGetLastError());
        std::vector<uint64_t> keys((size_t)nq * n);
        std::vector<float> qn2(nq);
        uint32_t sc2[8] = {0};
        HIP_TRY(hipMemcpyAsync(keys.data(), ix->cur->w_dense.p, keys.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(qn2.data(), ix->cur->w_qnorm.p, nq * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(sc2, ix->d_scalars, 32, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (size_t i = 0; i < keys.size(); ++i) {
            uint32_t bits = 0xffffffffu;                                    // no key (tombstoned row)
            if (keys[i] != vdb::EMPTY_KEY) { const float f = vdb::ordered_to_f32((uint32_t)(keys[i] >> 32)); memcpy(&bits, &f, 4); }
            memcpy(out_scores + i, &bits, 4);
        }
        if (out_qinfo)
            for (size_t q = 0; q < nq; ++q) { out_qinfo[4 * q] = qn2[q]; out_qinfo[4 * q + 1] = 0.0f; out_qinfo[4 * q + 2] = 0.0f; out_qinfo[4 * q + 3] = 0.0f; }
        if (out_consts) {
            auto f = [](uint32_t b) { float v; memcpy(&v, &b, 4); return (double)v; };
            out_consts[0] = eps_coef(ix); out_consts[1] = 0.0; out_consts[2] = 0.0; out_consts[3] = std::sqrt(f(sc2[0]));
            out_consts[4] = 0.0; out_consts[5] = 0.0; out_consts[6] = 0.0; out_consts[7] = (double)ld;
        }
        ix->cur->dbg_nq = (uint32_t)nq; ix->cur->dbg_lb = false; ix->cur->dbg_f32 = true;
        return VDB_OK;
    }
    ix->cur->dbg_f32 = false;
    vdb::QueryPrepParams qp{ix->cur->w_qin.p, (uint32_t)dim, (uint32_t)nq, ix->cur->w_qp.p, ld, SUPER, ix->cur->w_qnorm.p, ix->cur->w_thr.p, vdb::EUCLID,
                            ix->cur->w_flags.p, ix->cur->w_qb.p, ix->cur->w_qerr.p, ix->d_margin ? ix->cur->w_qg.p : nullptr, margin_plan(ix).kappa,
                            nullptr, nullptr};
    vdb::launch_query_prep(qp, s);
    std::vector<float> thr(nq, std::numeric_limits<float>::infinity());            // everything passes; padding queries keep -inf
    HIP_TRY(hipMemcpyAsync(ix->cur->w_thr.p, thr.data(), nq * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(ix->cur->w_dbg.p, 0xff, nq * (size_t)n * 4, s));            // NaN pattern = no key for this (query, row)
    vdb::FusedBf16Params fp{};
    fp.rows = ix->d_rows; fp.ld = ld; fp.n_rows = n; fp.qb = ix->cur->w_qb.p; fp.alpha = ix->d_alpha; fp.beta = ix->d_beta;
    fp.margin = lb ? ix->d_margin : nullptr; fp.qg = lb ? ix->cur->w_qg.p : nullptr;
    fp.rowmask = ix->d_live; fp.thr = ix->cur->w_thr.p; fp.pool = ix->cur->w_pool.p; fp.pool_cnt = ix->cur->w_subcnt.p; fp.capl = capl; fp.n_wg = n_wg;
    fp.scalars = ix->d_scalars; fp.qmax_bits = ix->cur->w_flags.p + 2;
    launch_filter_pass(ix, fp, s);                                                // the PRODUCTION filter pass (shadow rows if enabled)
    vdb::launch_pool_to_dense(ix->cur->w_pool.p, ix->cur->w_subcnt.p, n_sub, capl, (uint32_t)nq, n, ix->cur->w_dbg.p, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_scores, ix->cur->w_dbg.p, nq * (size_t)n * 4, hipMemcpyDeviceToHost, s));
    std::vector<float> qn(nq), qe(nq), qg(nq, 0.0f);
    uint32_t sc[8] = {0};
    HIP_TRY(hipMemcpyAsync(qn.data(), ix->cur->w_qnorm.p, nq * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(qe.data(), ix->cur->w_qerr.p, nq * 4, hipMemcpyDeviceToHost, s));
    if (ix->d_margin) HIP_TRY(hipMemcpyAsync(qg.data(), ix->cur->w_qg.p, nq * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(sc, ix->d_scalars, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (out_qinfo)
        for (size_t q = 0; q < nq; ++q) { out_qinfo[4 * q] = qn[q]; out_qinfo[4 * q + 1] = qe[q]; out_qinfo[4 * q + 2] = qg[q]; out_qinfo[4 * q + 3] = 0.0f; }
    if (out_consts) {
        auto f = [](uint32_t b) { float v; memcpy(&v, &b, 4); return (double)v; };
        const MarginPlan mp = margin_plan(ix);
        out_consts[0] = eps_coef(ix); out_consts[1] = c_acc_bf16(ix); out_consts[2] = mp.kappa;
        out_consts[3] = std::sqrt(f(sc[0])); out_consts[4] = std::sqrt(f(sc[2])); out_consts[5] = std::sqrt(f(sc[3]));   // max |d|, max |e_d|, max |e_d|/|d|
        out_consts[6] = lb ? 1.0 : 0.0; out_consts[7] = (double)ld;
    }
    ix->cur->dbg_nq = (uint32_t)nq; ix->cur->dbg_lb = lb;
    return VDB_OK;
    });
}

size_t vdb_flat_debug_rows(const vdb_flat_index* ix) { return (ix && !ix->multi) ? ix->row_ids.size() : 0; }

int vdb_flat_debug_last_thresholds(vdb_flat_index* ix, float* out, size_t nq) {
    return guarded([&]() -> int {
    if (ix && ix->multi) return refuse_multi("vdb_flat_debug_last_thresholds");
    if (!ix || !out) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc;
    if ((rc = set_device(ix))) return rc;
    if (nq > ix->cur->w_thr.n) return fail(VDB_ERR_INVALID_ARGUMENT, "more queries than the last search prepared");
    HIP_TRY(hipMemcpy(out, ix->cur->w_thr.p, nq * 4, hipMemcpyDeviceToHost));
    return VDB_OK;
    });
}

int vdb_flat_debug_row_info(vdb_flat_index* ix, float* out, size_t n_rows) {
    return guarded([&]() -> int {
    if (ix && ix->multi) return refuse_multi("vdb_flat_debug_row_info");
    if (!ix || !out) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc;
    if ((rc = set_device(ix))) return rc;
    if ((rc = flush(ix))) return rc;
    const size_t n = std::min<size_t>(n_rows, ix->n_uploaded);
    std::vector<float> a(n), b(n), c(n), d(n, 0.0f);
    if (n) {
        HIP_TRY(hipMemcpy(a.data(), ix->d_nd, n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(b.data(), ix->d_alpha, n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c.data(), ix->d_beta, n * 4, hipMemcpyDeviceToHost));
        if (ix->d_margin) HIP_TRY(hipMemcpy(d.data(), ix->d_margin, n * 4, hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < n; ++i) { out[4 * i] = a[i]; out[4 * i + 1] = b[i]; out[4 * i + 2] = c[i]; out[4 * i + 3] = d[i]; }
    return VDB_OK;
    });
}

int vdb_flat_debug_cert_probe(vdb_flat_index* ix, const uint32_t* qi, const float* T, const float* ek, size_t n, uint32_t* out) {
    return guarded([&]() -> int {
    if (ix && ix->multi) return refuse_multi("vdb_flat_debug_cert_probe");
    if (!ix || !qi || !T || !ek || !out) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    int rc;
    if ((rc = set_device(ix))) return rc;
    if (!ix->cur->dbg_nq) return fail(VDB_ERR_INVALID_ARGUMENT, "call vdb_flat_debug_screen_scores first");
    if (n == 0) return VDB_OK;
    if (n > 0x7fffffffull) return fail(VDB_ERR_INVALID_ARGUMENT, "too many probes");
    for (size_t i = 0; i < n; ++i)
        if (qi[i] >= ix->cur->dbg_nq) return fail(VDB_ERR_INVALID_ARGUMENT, "query index %u out of range", qi[i]);
    hipStream_t s = ix->stream;
    DevBuf<uint32_t> d_qi, d_out; DevBuf<float> d_T, d_ek;
    auto done = [&](int r) { d_qi.release(); d_out.release(); d_T.release(); d_ek.release(); return r; };
    if ((rc = d_qi.ensure(n)) || (rc = d_out.ensure(n)) || (rc = d_T.ensure(n)) || (rc = d_ek.ensure(n))) return done(rc);
    if (hipMemcpyAsync(d_qi.p, qi, n * 4, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(d_T.p, T, n * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_ek.p, ek, n * 4, hipMemcpyHostToDevice, s) != hipSuccess)
        return done(fail(VDB_ERR_DEVICE, "copy failed"));
    // the same parameter block the screening tier's re-rank gets (pass_bf16)
    vdb::RerankParams rp{};
    rp.metric = ix->metric; rp.eps_coef = eps_coef(ix); rp.nd2max_bits = ix->d_scalars; rp.qnorm = ix->cur->w_qnorm.p; rp.ld = ix->ld;
    rp.qerr = ix->cur->dbg_f32 ? nullptr : ix->cur->w_qerr.p; rp.c_acc = c_acc_bf16(ix); rp.lb_scores = ix->cur->dbg_lb ? 1u : 0u;   // dbg_f32: the f32 tier's parameter block (pass_f32)
    vdb::launch_cert_probe(rp, d_qi.p, d_T.p, d_ek.p, (uint32_t)n, d_out.p, s);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, d_out.p, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return done(fail(VDB_ERR_DEVICE, "cert probe failed"));
    return done(VDB_OK);
    });
}

int vdb_flat_set_sample_cache(vdb_flat_index* ix, int on) {
    return guarded([&]() -> int {
    if (!ix || on < 0 || on > 1) return fail(VDB_ERR_INVALID_ARGUMENT, "on must be 0 or 1");
    if (ix->multi) return multi_for_each(ix, [on](vdb_flat_index* c) { return vdb_flat_set_sample_cache(c, on); });
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    ix->sample_cache = on != 0;
    if (!on && ix->d_sample16) {
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(ix->d_sample16);
        ix->d_sample16 = nullptr; ix->sample16_cap = 0; ix->sample16_n = ix->sample16_S = 0;
    }
    return VDB_OK;
    });
}

int vdb_flat_set_shadow(vdb_flat_index* ix, int on) {
    return guarded([&]() -> int {
    if (!ix || on < 0 || on > 1) return fail(VDB_ERR_INVALID_ARGUMENT, "on must be 0 or 1");
    if (ix->multi) return multi_for_each(ix, [on](vdb_flat_index* c) { return vdb_flat_set_shadow(c, on); });
    std::lock_guard<std::mutex> g(ix->mu);
    if (in_flight(ix)) return refuse_in_flight();
    HIP_TRY(hipSetDevice(ix->device));
    int rc;
    if ((rc = flush(ix))) return rc;
    if (!on) {
        if (ix->d_rows16) { HIP_TRY(hipStreamSynchronize(ix->stream)); (void)hipFree(ix->d_rows16); }
        ix->d_rows16 = nullptr; ix->shadow = false;
        return VDB_OK;
    }
    if (!ix->d_rows16 && ix->cap_rows) {
        uint16_t* r16 = nullptr;
        HIP_TRY(hipMalloc((void**)&r16, (size_t)ix->cap_rows * ix->ld * 2));
        hipError_t e = hipMemsetAsync(r16, 0, (size_t)ix->cap_rows * ix->ld * 2, ix->stream);
        if (e == hipSuccess) { vdb::launch_rows_to_bf16(ix->d_rows, r16, ix->ld, 0, ix->n_uploaded, ix->stream); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        if (e != hipSuccess) { (void)hipFree(r16); return fail(VDB_ERR_DEVICE, "building the bf16 shadow failed: %s", hipGetErrorString(e)); }
        ix->d_rows16 = r16;                                       // only a COMPLETE shadow is ever visible to a search
    }
    ix->shadow = true;
    return VDB_OK;
    });
}

int vdb_flat_set_tiers(vdb_flat_index* ix, unsigned flags) {
    return guarded([&]() -> int {
    if (!ix || (flags & ~15u)) return fail(VDB_ERR_INVALID_ARGUMENT, "flags must be a combination of VDB_TIERS_*");
    if (ix->multi) return multi_for_each(ix, [flags](vdb_flat_index* c) { return vdb_flat_set_tiers(c, flags); });
    std::lock_guard<std::mutex> g(ix->mu);
    ix->tiers = flags;
    return VDB_OK;
    });
}

int vdb_flat_set_screen(vdb_flat_index* ix, int mode) {
    return guarded([&]() -> int {
    if (!ix || mode < 0 || mode > 1) return fail(VDB_ERR_INVALID_ARGUMENT, "mode must be 0 (f32 MFMA tier only) or 1 (bf16 screening tier first)");
    if (ix->multi) return multi_for_each(ix, [mode](vdb_flat_index* c) { return vdb_flat_set_screen(c, mode); });
    std::lock_guard<std::mutex> g(ix->mu);
    ix->screen = mode;
    return VDB_OK;
    });
}

int vdb_flat_set_wide(vdb_flat_index* ix, int on) {
    return guarded([&]() -> int {
    if (!ix || on < 0 || on > 1) return fail(VDB_ERR_INVALID_ARGUMENT, "on must be 0 or 1");
    if (ix->multi) return multi_for_each(ix, [on](vdb_flat_index* c) { return vdb_flat_set_wide(c, on); });
    std::lock_guard<std::mutex> g(ix->mu);
    ix->wide = on != 0;
    return VDB_OK;
    });
}

int vdb_flat_create_sharded(int metric, const int* devices, size_t n_devices, vdb_flat_index** out) {
    return guarded([&]() -> int { return multi_create(metric, devices, n_devices, out); });
}
size_t vdb_flat_shards(const vdb_flat_index* ix) { return !ix ? 0 : ix->multi ? multi_shards(ix) : 1; }
size_t vdb_flat_shard_len(const vdb_flat_index* ix, size_t shard) {
    return !ix ? 0 : ix->multi ? multi_shard_len(ix, shard) : (shard == 0 ? vdb_flat_len(ix) : 0);
}
int vdb_flat_set_exchange(vdb_flat_index* ix, int mode) {
    return guarded([&]() -> int {
    if (!ix) return fail(VDB_ERR_INVALID_ARGUMENT, "null handle");
    if (!ix->multi) return fail(VDB_ERR_INVALID_ARGUMENT, "not a sharded handle");
    return multi_set_exchange(ix, mode);
    });
}
int vdb_flat_shard_stats(const vdb_flat_index* ix, uint64_t out[8]) {
    return guarded([&]() -> int {
    if (!ix || !out) return fail(VDB_ERR_INVALID_ARGUMENT, "null argument");
    if (!ix->multi) return fail(VDB_ERR_INVALID_ARGUMENT, "not a sharded handle");
    multi_stats(ix, out);
    return VDB_OK;
    });
}

}  // extern "C"

// =================================================================== internal hooks (vdb_internal.h)
namespace vdb_internal {

static int ensure_pair_buffers(vdb_flat_index* ix, size_t n_pairs, size_t n_out) {
    if (n_pairs > ix->h_pairs_cap) {
        if (ix->h_pairs) (void)hipHostFree(ix->h_pairs);
        ix->h_pairs = nullptr; ix->h_pairs_cap = 0;
        size_t cap = std::max<size_t>(n_pairs + n_pairs / 2, 4096);
        HIP_TRY(hipHostMalloc((void**)&ix->h_pairs, cap * 2 * sizeof(uint32_t), hipHostMallocMapped));
        ix->h_pairs_cap = cap;
    }
    if (n_out > ix->h_pout_cap) {
        if (ix->h_pout) (void)hipHostFree(ix->h_pout);
        ix->h_pout = nullptr; ix->h_pout_cap = 0;
        size_t cap = std::max<size_t>(n_out + n_out / 2, 4096);
        HIP_TRY(hipHostMalloc((void**)&ix->h_pout, cap * sizeof(float), hipHostMallocMapped));
        ix->h_pout_cap = cap;
    }
    return VDB_OK;
}

int pairs_begin(vdb_flat_index* ix, const float* queries, size_t nq, size_t dim) {
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->wsv && (ix->wsv[0].busy || ix->wsv[1].busy)) return fail(VDB_ERR_INVALID_ARGUMENT, "a submitted search is still in flight on this handle");
    int rc;
    if ((rc = set_device(ix))) return rc;
    if ((rc = flush(ix))) return rc;
    ix->pairs_nq = 0;
    if (nq == 0) return VDB_OK;
    if (ix->n_live && ix->dim != dim) return fail_dim(dim, ix->dim);
    if (nq > 0x7fffffffull) return fail(VDB_ERR_INVALID_ARGUMENT, "too many queries");
    const uint32_t ld = ix->ld ? ix->ld : round_up((uint32_t)dim, vdb::KSTAGE);
    const uint32_t bp = round_up((uint32_t)nq, SUPER);
    hipStream_t s = ix->stream;
    if ((rc = ix->cur->w_qin.ensure(nq * dim))) return rc;
    if ((rc = ix->cur->w_qp.ensure((size_t)bp * ld))) return rc;
    if ((rc = ix->cur->w_qnorm.ensure(bp))) return rc;
    if ((rc = ix->cur->w_thr.ensure(bp))) return rc;
    if ((rc = ix->cur->w_flags.ensure(4))) return rc;
    HIP_TRY(hipMemcpyAsync(ix->cur->w_qin.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
    vdb::QueryPrepParams qp{ix->cur->w_qin.p, (uint32_t)dim, (uint32_t)nq, ix->cur->w_qp.p, ld, bp, ix->cur->w_qnorm.p, ix->cur->w_thr.p, vdb::EUCLID,
                            ix->cur->w_flags.p, nullptr, nullptr, nullptr, 0.0f, nullptr, nullptr};   // metric EUCLID: zero norms are judged per pair
    vdb::launch_query_prep(qp, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    ix->pairs_nq = (uint32_t)nq;
    return VDB_OK;
}

static int run_pair_eval(vdb_flat_index* ix, int mode, const uint32_t* a, const uint32_t* b, uint32_t q0, size_t n, float* out) {
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->wsv && (ix->wsv[0].busy || ix->wsv[1].busy)) return fail(VDB_ERR_INVALID_ARGUMENT, "a submitted search is still in flight on this handle");
    int rc;
    if ((rc = set_device(ix))) return rc;
    if (n == 0) return VDB_OK;
    if (n > 0xfffffff0ull) return fail(VDB_ERR_INVALID_ARGUMENT, "too many pairs");
    if (mode == 1 && (rc = flush(ix))) return rc;
    if ((rc = ensure_pair_buffers(ix, mode == 2 ? 1 : n, n))) return rc;
    uint32_t *d_pairs = nullptr; float* d_out = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void**)&d_pairs, ix->h_pairs, 0));
    HIP_TRY(hipHostGetDevicePointer((void**)&d_out, ix->h_pout, 0));
    if (mode != 2) {
        memcpy(ix->h_pairs, a, n * sizeof(uint32_t));
        memcpy(ix->h_pairs + n, b, n * sizeof(uint32_t));
    }
    vdb::PairEvalParams pp{};
    pp.rows = ix->d_rows; pp.ld = ix->ld; pp.dim = ix->dim; pp.nd = ix->d_nd; pp.qp = ix->cur->w_qp.p; pp.qnorm = ix->cur->w_qnorm.p;
    pp.a = d_pairs; pp.b = d_pairs + n; pp.n = (uint32_t)n; pp.q0 = q0; pp.mode = mode; pp.metric = ix->metric;
    pp.mark = ZERO_NORM_MARK; pp.out = d_out;
    // A SMALL request (an HNSW insert's misses: a dozen pairs) is waited for by watching the mapped result buffer instead of
    // synchronising the stream: the host fills it with a bit pattern no distance has, the kernel's stores land in host memory as
    // they retire, and the last one to change ends the wait -- the stream synchronisation's wake-up (~15-20 us) was a third of such
    // a round trip, and a 1M-row build makes a million of them.  (Bounded: after 2 ms the stream is synchronised after all.)
    constexpr uint32_t SENTINEL = 0xffc0fee1u;
    const bool poll = mode == 1 && n <= 256;
    volatile uint32_t* hw = reinterpret_cast<volatile uint32_t*>(ix->h_pout);
    if (poll) for (size_t i = 0; i < n; ++i) hw[i] = SENTINEL;
    vdb::launch_pair_eval(pp, ix->stream);
    HIP_TRY(hipGetLastError());
    bool landed = false;
    if (poll) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spin = 0; !landed; ++spin) {
            size_t i = 0;
            while (i < n && hw[i] != SENTINEL) ++i;
            landed = i == n;
            if (!landed && (spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    if (!landed) HIP_TRY(hipStreamSynchronize(ix->stream));
    memcpy(out, ix->h_pout, n * sizeof(float));
    return VDB_OK;
}

int pairs_eval(vdb_flat_index* ix, const uint32_t* pair_q, const uint32_t* pair_row, size_t n, float* out) {
    return run_pair_eval(ix, 0, pair_q, pair_row, 0, n, out);
}
int rows_eval(vdb_flat_index* ix, const uint32_t* row_a, const uint32_t* row_b, size_t n, float* out) {
    return run_pair_eval(ix, 1, row_a, row_b, 0, n, out);
}
int query_vs_rows(vdb_flat_index* ix, uint32_t q, uint32_t n_rows_, float* out) {
    if (n_rows_ > ix->n_uploaded) return fail(VDB_ERR_INVALID_ARGUMENT, "rows not uploaded");
    return run_pair_eval(ix, 2, nullptr, nullptr, q, n_rows_, out);
}
uint32_t row_of(vdb_flat_index* ix, uint64_t id) {
    std::lock_guard<std::mutex> g(ix->mu);
    auto it = ix->id2row.find(id);
    return it == ix->id2row.end() ? 0xffffffffu : it->second;
}
uint32_t n_rows(vdb_flat_index* ix) { return ix->n_rows(); }
int device_view(vdb_flat_index* ix, DeviceView* out) {
    std::lock_guard<std::mutex> g(ix->mu);
    int rc;
    if ((rc = set_device(ix))) return rc;
    if ((rc = ix->cur->w_flags.ensure(4))) return rc;
    out->rows = ix->d_rows; out->ld = ix->ld; out->dim = ix->dim; out->nd = ix->d_nd; out->qp = ix->cur->w_qp.p; out->qnorm = ix->cur->w_qnorm.p;
    out->metric = ix->metric; out->stream = (void*)ix->stream; out->status = ix->cur->w_flags.p;
    return VDB_OK;
}
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }
int set_dim_error(size_t expected, size_t actual) { return fail_dim(expected, actual); }

}  // namespace vdb_internal
