// vdb_index.h -- internal declarations shared by the host-side translation units of libvdbflat.so:
//   vdb_flat.cpp     the extern "C" shims of include/vdb_flat.h (argument checks, locking, exception guard)
//   vdb_store.cpp    the device-resident mirror of FlatIndex's rows (src/flat_index.rs:12-50): staging, upload, tombstones
//   vdb_cert.cpp     the coefficients of the "certified top-k" bounds and the tier plans
//   vdb_search.cpp   the tier scheduler: screening pass, re-threshold pass, f32 MFMA tier, exact scan
//   vdb_multi.cpp    ONE index over several GPUs in one process (vdb_flat_create_sharded)
// Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vdb_flat.h"
#include "kernels.h"
#include "vdb_internal.h"

namespace vdbi {

// ---- thread-local last error (vdb_last_error) and the exception guard of every extern "C" body
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail_dim(size_t expected, size_t actual);
int guard_fail(const char* what);
void last_error(std::string* msg, size_t* expected, size_t* actual);

// No C++ exception may cross the C ABI (ctypes, a Rust FFI caller: undefined behaviour or abort).  Every extern "C" entry
// point that can allocate runs its body through this.
template <class F> int guarded(F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { return guard_fail("out of host memory"); }
    catch (const std::exception& e) { return guard_fail(e.what()); }
    catch (...) { return guard_fail("unknown C++ exception"); }
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ::vdbi::fail(VDB_ERR_DEVICE, "HIP error %d (%s) at %s:%d: %s", (int)e_,      \
                                hipGetErrorString(e_), __FILE__, __LINE__, #expr);              \
    } while (0)

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t want) {
        if (want <= n) return VDB_OK;
        size_t cap = std::max(want, n + n / 2);
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        HIP_TRY(hipMalloc((void**)&p, cap * sizeof(T)));
        n = cap;
        return VDB_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }
inline uint32_t pow2_ceil(uint64_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

constexpr uint32_t SMALL_N = 16384;     // at or below: dense scores of every row, no fused pass
constexpr uint32_t SUPER = 256;         // queries per pipeline pass
constexpr uint32_t MAX_SELECT = 2048;   // select kernel capacity (kk)
constexpr uint32_t DIRECT_MAX_Q = 8;    // the direct exact path of small indexes takes batches up to this many queries

}  // namespace vdbi

struct Workspace {
    vdbi::DevBuf<float> w_qp, w_qnorm, w_thr, w_qin, w_outd, w_qerr, w_qg, w_dbg;
    uint32_t dbg_nq = 0; bool dbg_lb = false, dbg_f32 = false;             // vdb_flat_debug_screen_scores left this many prepared queries in the workspace
    vdbi::DevBuf<uint64_t> w_dense, w_samp, w_pool, w_cand, w_exact, w_exsel, w_mask_ids, w_outi;
    vdbi::DevBuf<uint32_t> w_cnt, w_rowmask, w_flags, w_outc, w_subcnt, w_depth;
    vdbi::DevBuf<uint16_t> w_qb;                                  // bf16 copy of the padded queries (screening tier)
    // compact block of the queries the screening tier could not certify (re-run by the f32 tier)
    vdbi::DevBuf<float> w2_qp, w2_qnorm, w2_thr, w2_outd, w2_qerr, w2_qg;
    vdbi::DevBuf<uint64_t> w2_outi, w2_cand;
    vdbi::DevBuf<uint16_t> w2_qb;
    vdbi::DevBuf<uint32_t> w2_outc, w2_flags, w2_qidx;
    uint32_t* h_flags = nullptr; size_t h_flags_n = 0;
    // the direct path of small indexes (search_direct): a device status word known to be zero between searches, and MAPPED host
    // memory -- the kernels write results and status straight into it (h_io: the host-pointer entry point's queries and outputs)
    vdbi::DevBuf<uint32_t> w_dstat; bool dstat_ready = false;
    uint32_t* h_dstat = nullptr; uint32_t* d_h_dstat = nullptr;       // [16] host view / device view
    char* h_io = nullptr; char* d_h_io = nullptr; size_t h_io_bytes = 0;
    bool status_dirty = true; uint32_t* status_buf = nullptr;   // device status block known to be zero?
    // a search between its two halves (search_part1 enqueues the first tier, search_part2 reads its flags and runs
    // the fallback tiers): vdb_flat_search_batch_device_begin / _finish keep the handle locked in between
    struct SearchCtx {
        bool pending = false;                               // part 2 still has to run
        uint32_t nq32 = 0, kp = 0, kp16 = 0;
        size_t k = 0;
        hipStream_t s = nullptr;
        const uint32_t* d_rowmask = nullptr;
        uint64_t* d_out_ids = nullptr; float* d_out_dists = nullptr; uint32_t* d_out_counts = nullptr;
        std::chrono::steady_clock::time_point t_entry;
    } ctx;
    uint64_t stats[16] = {0};
    hipStream_t stream = nullptr;                           // this context's own stream (used when the caller passes none)
    bool busy = false;                                      // submitted, not yet waited for
    template <class F> void for_each_buffer(F&& f) {
        f(w_qp); f(w_qnorm); f(w_thr); f(w_qin); f(w_outd); f(w_qerr); f(w_qg); f(w_dbg);
        f(w_dense); f(w_samp); f(w_pool); f(w_cand); f(w_exact); f(w_exsel); f(w_mask_ids); f(w_outi);
        f(w_cnt); f(w_rowmask); f(w_flags); f(w_outc); f(w_subcnt); f(w_depth); f(w_qb);
        f(w2_qp); f(w2_qnorm); f(w2_thr); f(w2_outd); f(w2_qerr); f(w2_qg); f(w2_outi); f(w2_cand); f(w2_qb);
        f(w2_outc); f(w2_flags); f(w2_qidx); f(w_dstat);
    }
};

// Diagnostic knobs: ablation switches, A/B kernel variants, scaled certificates, sample-size overrides.  Several of them
// VOID the exact-result guarantee, so they exist only in the diagnostics build (-DVDB_DIAG -> libvdbflat_diag.so,
// `make diag`), where vdb_flat_create reads them from the environment ONCE into the handle.  In the release library
// this struct is a set of constants and there is no getenv anywhere.
struct vdb_knobs {
    double eps_scale = 1.0;               // VDB_EPS_SCALE: scales both certification coefficients (0 = no margin!)
    uint32_t bf16_ablate = 0;             // VDB_BF16_ABLATE: phases of the screening kernel switched off (wrong results)
    uint32_t fused_ablate = 0;            // VDB_FUSED_ABLATE: the same for the f32 MFMA kernel
    uint32_t kt16 = 0, sample16 = 0;      // VDB_KT16 / VDB_SAMPLE16: threshold rank / sample size of the screening tier
    uint32_t sample = 0;                  // VDB_SAMPLE: sample size of the f32 tier
    uint32_t kp_first = 0;                // VDB_KP_FIRST: first re-rank round
    bool rr_depth = false;                // VDB_RR_DEPTH: print the re-rank depth distribution
    bool sample_block = false;            // VDB_SAMPLE_BLOCK: contiguous-block sampling
    bool shape4 = false, regstage = false, dma2 = false;   // VDB_FUSED_SHAPE4 / _REGSTAGE / _DMA2: A/B variants of the f32 kernel
    bool fused_pipe = true;               // VDB_FUSED_PIPE=0: unpipelined screening filter pass
    bool any = false;                     // some knob differs from its default -> last_stats_ex()[15] = 1
};

struct vdb_multi;                          // vdb_multi.cpp: the shards of a vdb_flat_create_sharded handle

struct vdb_flat_index {
    int metric = 0, device = 0;
    vdb_multi* multi = nullptr;           // non-null: this handle is the PARENT of a sharded index and owns nothing below but mu / stats
    vdb_knobs kn;
    uint32_t tiers = 0;                   // vdb_flat_set_tiers: VDB_TIERS_* bits (tier hand-over forced; results identical)
    hipStream_t stream = nullptr;
    int n_cu = 256;
    std::mutex mu;

    uint32_t dim = 0, ld = 0;             // primary dimension and padded row stride (floats)
    // host bookkeeping of the device rows
    std::vector<uint64_t> row_ids;
    std::vector<uint32_t> live;           // bit per row
    std::unordered_map<uint64_t, uint32_t> id2row;
    uint32_t n_live = 0;
    bool ids_monotone = true;
    // rows whose dimension differs from `dim` (reference add() has no check, flat_index.rs:38-41)
    std::unordered_map<uint64_t, std::vector<float>> misfits;
    // rows staged on the host, not yet uploaded: device rows [n_uploaded, row_ids.size())
    std::vector<float> pending;
    uint32_t n_uploaded = 0;
    bool live_dirty = false;

    // device store
    // compact bf16 copy of the screening tier's S sample rows (kernels_fused_s16.hip SAMPLE mode): +S*ld*2 bytes (3 % of a 1M-row
    // index), rebuilt when rows were added; the sample pass then streams 100 MB of contiguous bf16 instead of gathering 200 MB of
    // f32 rows.  Thresholds are identical (same roundings, same MFMA order).  vdb_flat_set_sample_cache(h, 0) turns it off.
    uint16_t* d_sample16 = nullptr; size_t sample16_cap = 0;            // capacity in elements
    uint32_t sample16_n = 0, sample16_S = 0;                            // what the copy was built for (rows uploaded, sample size)
    bool sample_cache = true;
    uint16_t* d_rows16 = nullptr;         // opt-in bf16 shadow of d_rows [cap_rows][ld] (vdb_flat_set_shadow), else null
    bool shadow = false;
    float* d_rows = nullptr; float* d_nd = nullptr; float* d_alpha = nullptr; float* d_beta = nullptr;
    float* d_margin = nullptr;            // [cap] per-row error margin of the screening tier's lower-bound scores (Dot / Euclid; null under Cosine)
    uint64_t* d_row_ids = nullptr; uint32_t* d_live = nullptr; uint32_t* d_scalars = nullptr;  // [0]=nd2max bits [1]=zero count [2],[3]=max bf16 rounding error of a row (abs^2, rel^2)
    uint32_t cap_rows = 0;
    bool zero_valid = false; uint32_t zero_live = 0;
    vdbi::DevBuf<uint32_t> d_idrank, d_rank2row; bool rank_valid = false;

    // search workspace: everything one search in flight owns.  Two of them, so that two batches can be in flight on two
    // streams (vdb_flat_search_batch_device_submit / _wait); every synchronous entry point uses the first.
    struct Workspace* cur = nullptr;                        // the context the search code below works in (set under the handle mutex)
    struct Workspace* wsv = nullptr;                        // [2]
    // mapped host memory for the pair hooks (vdb_internal.h): the kernel reads the pairs and writes the distances in place
    uint32_t* h_pairs = nullptr; float* h_pout = nullptr; size_t h_pairs_cap = 0, h_pout_cap = 0;
    uint32_t pairs_nq = 0;
    bool begin_locked = false;
    hipEvent_t ev_pass[2] = {nullptr, nullptr};             // fork / join of the alternating passes of a large batch (pass_bf16)
    hipEvent_t ev_order = nullptr;                          // orders the handle's stream before the null stream (search_batch_device_begin)
    int screen = 1;                                         // 1: bf16 screening tier first (default), 0: f32 MFMA tier only
    bool wide = true;                                       // batches above 256 queries: the 512-query filter kernel (vdb_flat_set_wide)
    uint64_t stats[16] = {0};                               // counters of the last COMPLETED search (copied from its context)
    bool profile = false; hipEvent_t ev0 = nullptr, ev1 = nullptr;

    uint32_t n_rows() const { return (uint32_t)row_ids.size(); }
    bool is_live(uint32_t r) const { return (live[r >> 5] >> (r & 31)) & 1u; }
};

namespace vdbi {

using Index = vdb_flat_index;

inline int set_device(const Index* ix) {
    HIP_TRY(hipSetDevice(ix->device));
    return VDB_OK;
}

// ---- vdb_store.cpp: the device row store
int grow(Index* ix, uint32_t need_rows);
void free_store(Index* ix);
void reset_rows(Index* ix);
void kill_row(Index* ix, uint32_t row);
int remove_id(Index* ix, uint64_t id);
int add_one(Index* ix, uint64_t id, const float* v, size_t dim);
int flush(Index* ix);
int ensure_zero_count(Index* ix);
int ensure_ranks(Index* ix);

// ---- vdb_cert.cpp: certificate coefficients and tier plans
float eps_coef(const Index* ix);
float c_acc_bf16(const Index* ix);
struct MarginPlan { float m_e = 0, m_n = 0, m_b = 0, kappa = 0, beta_shrink = 0; };
MarginPlan margin_plan(const Index* ix);
constexpr uint32_t BF16_MIN_ROWS = 65536;
struct Bf16Plan { uint32_t kp = 0, S = 0, shift = 0, kt = 0; };
Bf16Plan plan_bf16(const Index* ix, uint32_t n, size_t k);
uint32_t pick_kp(size_t k);

// ---- vdb_search.cpp: the tier scheduler
bool shadow_usable(const Index* ix);
void launch_filter_pass(Index* ix, vdb::FusedBf16Params& fp, hipStream_t s);
int search_part1(Index* ix, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_idmask,
                 size_t mask_bits, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                 hipStream_t user_stream, bool allow_alt = false);
int search_part2(Index* ix, int* changed);
bool direct_eligible(const Index* ix, size_t n_rows, size_t nq, size_t k);
int ensure_host_io(Index* ix, size_t bytes);
void publish_stats(Index* ix);
bool in_flight(const Index* ix);
int refuse_in_flight();
int search_device(Index* ix, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_idmask,
                  size_t mask_bits, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                  hipStream_t user_stream);

// ---- vdb_multi.cpp: one index over several GPUs in one process (the parent handle dispatches here)
int multi_create(int metric, const int* devices, size_t n, vdb_flat_index** out);
void multi_destroy(vdb_flat_index* P);
int multi_add(vdb_flat_index* P, uint64_t id, const float* v, size_t dim);
int multi_add_bulk(vdb_flat_index* P, const uint64_t* ids, uint64_t first_id, const float* rows, size_t n, size_t dim, bool on_device);
int multi_remove(vdb_flat_index* P, uint64_t id);
int multi_get_vector(vdb_flat_index* P, uint64_t id, float* out, size_t cap, size_t* dim);
size_t multi_len(const vdb_flat_index* P);
size_t multi_dim(const vdb_flat_index* P);
int multi_reserve(vdb_flat_index* P, size_t rows, size_t dim);
int multi_for_each(vdb_flat_index* P, const std::function<int(vdb_flat_index*)>& f);
int multi_search_device(vdb_flat_index* P, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_mask, size_t mask_bits,
                        uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, hipStream_t user_stream);
int multi_search_host(vdb_flat_index* P, const float* queries, size_t nq, size_t dim, const size_t* ks, size_t k, const uint64_t* id_mask,
                      size_t mask_bits, size_t kstride, uint64_t* out_ids, float* out_dists, size_t* out_counts);
int multi_set_exchange(vdb_flat_index* P, int mode);
size_t multi_shards(const vdb_flat_index* P);
size_t multi_shard_len(const vdb_flat_index* P, size_t g);
void multi_stats(const vdb_flat_index* P, uint64_t out[8]);
inline int refuse_multi(const char* what) { return fail(VDB_ERR_INVALID_ARGUMENT, "%s is not available on a sharded handle (vdb_flat_create_sharded)", what); }

}  // namespace vdbi
