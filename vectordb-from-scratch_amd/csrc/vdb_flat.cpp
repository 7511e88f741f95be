// vdb_flat.cpp -- host side of the C ABI declared in include/vdb_flat.h: the device-resident
// mirror of the reference's FlatIndex (src/flat_index.rs:12-74) and the search pipeline that
// drives the HIP kernels.  No CPU compute path exists here: every distance is produced on the
// GPU, and every entry point fails with VDB_ERR_DEVICE when no HIP device is usable.
//
// Device layout (all in HBM, one allocation each, grown by doubling):
//   rows     [cap][ld] f32   ld = dim rounded up to 32, zero padded (K stage of the MFMA kernel)
//   nd       [cap]     f32   exact-order row norm  (vector.rs:35-37)
//   alpha,beta [cap]   f32   ranking score = fma(dot, alpha, beta)
//   row_ids  [cap]     u64   device row -> reference internal id
//   live     [cap/32]  u32   tombstone bitmask (remove() clears a bit; rows are append-only)
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/vdb_flat.h"
#include "kernels.h"
#include "vdb_internal.h"

namespace {

thread_local std::string g_err;
thread_local size_t g_expected = 0, g_actual = 0;

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

int vdb_guard_fail(const char* what) { return fail(VDB_ERR_DEVICE, "internal error: %s", what); }

// No C++ exception may cross the C ABI (ctypes, a Rust FFI caller: undefined behaviour or abort).  Every extern "C" entry
// point that can allocate runs its body through this.
template <class F> int guarded(F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { return vdb_guard_fail("out of host memory"); }
    catch (const std::exception& e) { return vdb_guard_fail(e.what()); }
    catch (...) { return vdb_guard_fail("unknown C++ exception"); }
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(VDB_ERR_DEVICE, "HIP error %d (%s) at %s:%d: %s", (int)e_,              \
                        hipGetErrorString(e_), __FILE__, __LINE__, #expr);                      \
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

}  // namespace

struct Workspace {
    DevBuf<float> w_qp, w_qnorm, w_thr, w_qin, w_outd, w_qerr, w_qg, w_dbg;
    uint32_t dbg_nq = 0; bool dbg_lb = false, dbg_f32 = false;             // vdb_flat_debug_screen_scores left this many prepared queries in the workspace
    DevBuf<uint64_t> w_dense, w_samp, w_pool, w_cand, w_exact, w_exsel, w_mask_ids, w_outi;
    DevBuf<uint32_t> w_cnt, w_rowmask, w_flags, w_outc, w_subcnt, w_depth;
    DevBuf<uint16_t> w_qb;                                  // bf16 copy of the padded queries (screening tier)
    // compact block of the queries the screening tier could not certify (re-run by the f32 tier)
    DevBuf<float> w2_qp, w2_qnorm, w2_thr, w2_outd, w2_qerr, w2_qg;
    DevBuf<uint64_t> w2_outi, w2_cand;
    DevBuf<uint16_t> w2_qb;
    DevBuf<uint32_t> w2_outc, w2_flags, w2_qidx;
    uint32_t* h_flags = nullptr; size_t h_flags_n = 0;
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
        f(w2_outc); f(w2_flags); f(w2_qidx);
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

struct vdb_flat_index {
    int metric = 0, device = 0;
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
    DevBuf<uint32_t> d_idrank, d_rank2row; bool rank_valid = false;

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
    uint64_t stats[16] = {0};                               // counters of the last COMPLETED search (copied from its context)
    bool profile = false; hipEvent_t ev0 = nullptr, ev1 = nullptr;

    uint32_t n_rows() const { return (uint32_t)row_ids.size(); }
    bool is_live(uint32_t r) const { return (live[r >> 5] >> (r & 31)) & 1u; }
};

namespace {

using Index = vdb_flat_index;

int set_device(const Index* ix) {
    HIP_TRY(hipSetDevice(ix->device));
    return VDB_OK;
}

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

// Certification coefficient (DESIGN.md "certified top-k"): worst-case rounding bound of the
// MFMA fma chain plus the oracle's sequential fold, K = padded row length.  VDB_EPS_SCALE
// scales it (diagnostics only).
float eps_coef(const Index* ix) {
    const double u = 5.9604644775390625e-08;   // 2^-24
    double K = (double)ix->ld;
    double c;
    if (ix->metric == vdb::EUCLID) c = (K + 4.0) * u;
    else if (ix->metric == vdb::COSINE) c = (2.0 * K + 16.0) * u;
    else c = (2.0 * K + 2.0) * u;
    c *= 1.1;
    c *= ix->kn.eps_scale;                 // 1.0 outside the diagnostics build
    return (float)c;
}

// f32 accumulation inside the bf16 MFMAs (products of two bf16 are exact in f32): at most K 2^-22 |q||d|, 5 % margin.
// The operand-rounding part of the screening tier's error bound is evaluated per query in rerank_kernel from the
// known |q - bf16(q)| and the per-index maxima of |d - bf16(d)| (row_stats_kernel).
float c_acc_bf16(const Index* ix) {
    double c = (double)ix->ld * 2.384185791015625e-07 * 1.05;
    c *= ix->kn.eps_scale;                 // 1.0 outside the diagnostics build
    return (float)c;
}

// The LOCAL form of the screening tier's certificate (Dot / Euclid).  For a row d and a query q the ranking score differs
// from what the oracle's exact distance implies by at most
//     Dot:     1.01 (|e_q||d| + 1.004 |q||e_d|) + (c_acc + eps) |q||d|
//     Euclid:  2 x that with eps doubled, + eps |d|^2 (folded into the row's beta), + eps (|q|^2 + e_k^2) (query only, cert_test)
// with e_q = q - bf16(q), e_d = d - bf16(d).  Everything row-dependent is of the form |q| A_d + |e_q| B_d;
// with kappa = B_d / A_d of a typical row (relative bf16 rounding error 1e-3) it is bounded by g_q * M_d,
//     g_q = |q| + kappa |e_q|   (query_prep),      M_d = max(A_d, B_d / kappa)   (row_stats: ONE more constant per row),
// and the kernels rank by the lower-bound score  score - g_q M_d.  Any kappa > 0 is valid; this one makes the bound tight.
struct MarginPlan { float m_e = 0, m_n = 0, m_b = 0, kappa = 0, beta_shrink = 0; };
MarginPlan margin_plan(const Index* ix) {
    MarginPlan mp;
    if (ix->metric == vdb::COSINE) return mp;
    const double eps = (double)eps_coef(ix), cacc = (double)c_acc_bf16(ix);
    const double two = ix->metric == vdb::EUCLID ? 2.0 : 1.0;
    const double Ae = two * 1.01 * 1.004, An = two * (cacc + eps), Bn = two * 1.01;
    const double kappa = Bn / (Ae * 1.0e-3 + An);
    mp.m_e = (float)Ae; mp.m_n = (float)(An * 1.000001); mp.m_b = (float)(Bn / kappa * 1.000001); mp.kappa = (float)(kappa * 1.000001);
    mp.beta_shrink = ix->metric == vdb::EUCLID ? (float)eps : 0.0f;
    return mp;
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

// The filter pass of the screening tier: over the bf16 shadow rows when the index keeps them (vdb_flat_set_shadow) and the
// row pitch allows whole 128-byte row requests, else over the f32 rows.  Same scores either way, bit for bit.
bool shadow_usable(const Index* ix) { return ix->d_rows16 && ix->ld % 64 == 0; }
void launch_filter_pass(Index* ix, vdb::FusedBf16Params& fp, hipStream_t s) {
    if (shadow_usable(ix)) { fp.rows16 = ix->d_rows16; vdb::launch_fused_s16(fp, s); return; }
#ifdef VDB_DIAG
    if (!ix->kn.fused_pipe) { vdb::launch_fused_bf16(fp, s); return; }
#endif
    vdb::launch_fused_bf16p(fp, s);
}

// bf16 screening tier: the select delivers up to 256 candidates per query, sorted by score, and the re-rank goes
// through them adaptively (rerank_kernel): first round_up(k + 22, 32), then 32 more per round until the result is
// certified.  The filter threshold is the kt-th smallest of the M = S/64 group minima of an S-row sample (at least
// kt rows pass it); S = 2^s is sized so that about 2000 keys per query pass, and kt <= M/4 so that the kt smallest
// minima come from (nearly) distinct groups.
constexpr uint32_t BF16_MIN_ROWS = 65536;
struct Bf16Plan { uint32_t kp = 0, S = 0, shift = 0, kt = 0; };
Bf16Plan plan_bf16(const vdb_flat_index* ix, uint32_t n, size_t k) {
    Bf16Plan pl;
    if (n < BF16_MIN_ROWS || k > 112) return pl;
    // threshold rank: at least kt rows pass the filter, about kt * n / S are expected to (k = 10 at 1M rows: 16 -> ~244 keys per
    // query).  The re-rank certifies against the score of the first candidate it did NOT re-rank, so what the rank has to
    // provide is a pool a few times deeper than the first round (k + 38), not a margin: 16 instead of 32 halves the appends
    // of the filter pass (its epilogue's rare path, ~12 us per launch at config 2) at the same first-round certification
    const uint32_t want_kt = std::min<uint32_t>(128u, round_up((uint32_t)k + 6u, 16u));
    uint64_t S = std::min<uint64_t>(65536u, std::max<uint64_t>(16384u, pow2_ceil((uint64_t)n / 16u)));
    if (ix->kn.sample16) S = pow2_ceil(std::max(256u, ix->kn.sample16));
    while (S / 256u < want_kt && 2 * S <= n / 2) S *= 2;
    while (S > n) S /= 2;
    // threshold rank: enough for the first re-rank round; the pool (about N/S * kt keys) feeds the deeper rounds
    uint32_t kt = std::min<uint32_t>(want_kt, (uint32_t)(S / 256u));
    // ... and no deeper than that: one sample rank stands for n / S rows, so rank kt lets about kt * n / S keys through.  Aim at
    // ~6 k keys per query (k = 100 at 1.25M rows: rank 32 -> ~610 keys instead of rank 112 -> ~2100, of which the select kept
    // 512 anyway).  Below rank k + 1 the sample no longer GUARANTEES k candidates; it does not have to -- the re-rank refuses
    // to certify a result with fewer than k real rows and the query goes to the re-threshold pass (never observed: the pool
    // size varies by about +-18 % at rank 32).
    {
        const uint64_t per_rank = std::max<uint64_t>(1, (uint64_t)n / S);
        const uint32_t kt_pool = round_up((uint32_t)std::max<uint64_t>(16, (6ull * k + per_rank - 1) / per_rank), 16u);
        kt = std::min(kt, kt_pool);
    }
    if (ix->kn.kt16) kt = std::min<uint32_t>(ix->kn.kt16, (uint32_t)(S / 256u));
    if (kt < 16) return pl;
    pl.kp = k > 48 ? 512 : 256;                                  // candidates the select delivers (depth limit of the re-rank)
    pl.S = (uint32_t)S; pl.kt = kt;
    while ((1ull << pl.shift) < S) ++pl.shift;
    return pl;
}

uint32_t pick_kp(size_t k) {
    size_t want = k + std::max<size_t>(6, k / 5);
    if (want <= 32) return 32;
    if (want <= 64) return 64;
    if (want <= 128) return 128;
    return 0;   // exact-scan path
}

// ------------------------------------------------------------------ exact path for one query
int exact_one(Index* ix, hipStream_t s, uint32_t q, size_t k, const uint32_t* d_rowmask, uint64_t* d_out_ids,
              float* d_out_dists, uint32_t* d_out_count) {
    int rc;
    uint32_t n = ix->n_uploaded;
    if ((rc = ensure_ranks(ix))) return rc;
    if ((rc = ix->cur->w_exact.ensure(n))) return rc;
    if ((rc = ix->cur->w_exsel.ensure(MAX_SELECT + 8))) return rc;
    if ((rc = ix->cur->w_cnt.ensure(4 * SUPER + 16))) return rc;
    vdb::ExactScanParams ep{ix->d_rows, ix->ld, ix->dim, n, ix->cur->w_qp.p + (size_t)q * ix->ld, ix->cur->w_qnorm.p + q, ix->d_nd,
                            d_rowmask, ix->ids_monotone ? nullptr : ix->d_idrank.p, ix->metric, ix->cur->w_exact.p,
                            ix->cur->w_flags.p};
    vdb::launch_exact_scan(ep, s);
    uint32_t* cnt = ix->cur->w_cnt.p + 4 * SUPER;
    uint64_t* last = ix->cur->w_exsel.p + MAX_SELECT;      // largest key emitted so far (one u64 after the sort area)
    // k may be as large as the index: emit in chunks of MAX_SELECT, each chunk = the smallest keys
    // strictly above the previous chunk's last key
    for (size_t done = 0; done < k; done += MAX_SELECT) {
        uint32_t kk = (uint32_t)std::min<size_t>(MAX_SELECT, k - done);
        vdb::SelectParams sp{};
        sp.keys = ix->cur->w_exact.p; sp.stride = 0; sp.counts = nullptr; sp.n_fixed = n; sp.cap = n;
        sp.kk = kk; sp.out_keys = ix->cur->w_exsel.p; sp.out_stride = MAX_SELECT; sp.out_cnt = cnt;
        sp.out_thr = nullptr; sp.ovf = nullptr;
        sp.lo_excl = done ? last : nullptr; sp.out_last = last;
        vdb::launch_select(sp, 1, s);
        vdb::EmitParams em{ix->cur->w_exsel.p, MAX_SELECT, cnt, ix->ids_monotone ? nullptr : ix->d_rank2row.p,
                           ix->d_row_ids, d_out_ids + done, d_out_dists + done, d_out_count, kk, done ? 1u : 0u};
        vdb::launch_emit(em, s);
    }
    HIP_TRY(hipGetLastError());
    return VDB_OK;
}

// ------------------------------------------------------------------ tier: f32 MFMA scores + certified re-rank
// Runs the f32 pipeline for the nq queries whose padded rows start at qp (stride ld; the block
// must be readable and zero up to a multiple of 256 rows, thr = -inf in the padding), writing results for query j
// at out_*[j*k ..] and the certification / pool-overflow flags at d_cert[j] / d_ovf[j].
int pass_f32(Index* ix, hipStream_t s, const float* qp, const float* qnorm, float* thr, uint32_t nq, size_t k, uint32_t kp,
             const uint32_t* d_rowmask, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, uint32_t* d_cert,
             uint32_t* d_ovf, uint32_t* d_status) {
    int rc;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    const bool small = n <= SMALL_N;
    // Threshold sample size S: the fused pass keeps about n*kp/S keys per query, spread over 512 private
    // sub-pools of 64 slots and gathered into 16384 LDS slots by the select.  S is chosen so that this
    // expectation stays near 8000 or below (mean sub-pool fill <= 16), and the sample costs <= ~3 % of the
    // fused pass for k = 10.
    uint32_t S = n;
    if (!small) {
        uint64_t want = std::max<uint64_t>(n / 256u, (uint64_t)n * kp / 8000u);
        S = (uint32_t)std::min<uint64_t>(65536u, std::max<uint64_t>(2048u, pow2_ceil(want)));
        if (ix->kn.sample) S = std::min<uint32_t>(n, std::max(64u, ix->kn.sample));
    }
    // candidate pools: one private sub-pool per (query, row range, row part, lane half) of the fused kernel
    const uint32_t capl = 64;
    // sub-pools in one pass = queries * row ranges * row parts * 2 = 512 * n_cu for every kernel shape
    const size_t pass_subs = 512u * (size_t)ix->n_cu;
    if ((rc = ix->cur->w_dense.ensure((size_t)SUPER * S))) return rc;
    if ((rc = ix->cur->w_cand.ensure((size_t)SUPER * kp))) return rc;
    if (!small) {
        if ((rc = ix->cur->w_samp.ensure((size_t)SUPER * kp))) return rc;
        if ((rc = ix->cur->w_pool.ensure(pass_subs * capl))) return rc;
        if ((rc = ix->cur->w_subcnt.ensure(pass_subs))) return rc;
    }
    uint32_t* d_cnt_a = ix->cur->w_cnt.p;               // sample select counts
    uint32_t* d_cand_cnt = ix->cur->w_cnt.p + 2 * SUPER;
    if (!ix->cur->stats[8]) { ix->cur->stats[4] = S; ix->cur->stats[5] = kp; }
    const float eps = eps_coef(ix);

    for (uint32_t q0 = 0; q0 < nq; q0 += SUPER) {
        const uint32_t nb = std::min(SUPER, nq - q0);
        const uint32_t tiles = (nb + 31) / 32;
        // fused-kernel shape: 32 / 64 / 128 queries per workgroup; 2 workgroups per CU in flight
        const bool shape8 = !ix->kn.shape4;   // default: ONE 8-wave workgroup per CU, 256 queries share each fetched row tile (diagnostics: two 4-wave workgroups of 128 queries)
        const int nqt = (shape8 && tiles > 4) ? 8 : tiles > 2 ? 4 : (int)tiles;
        const uint32_t n_super = (tiles + nqt - 1) / nqt;          // workgroups along the query axis (1 or 2)
        const float* qp0 = qp + (size_t)q0 * ld;

        vdb::DenseParams dp{ix->d_rows, ld, n, qp0, round_up(nb, 32), ix->d_alpha, ix->d_beta, d_rowmask, S,
                            ix->cur->w_dense.p, S};
        vdb::launch_dense_scores(dp, s);

        vdb::SelectParams sp{};
        sp.keys = ix->cur->w_dense.p; sp.stride = S; sp.counts = nullptr; sp.n_fixed = S; sp.cap = S; sp.kk = kp;
        sp.out_stride = kp;
        if (small) {
            sp.out_keys = ix->cur->w_cand.p; sp.out_cnt = d_cand_cnt; sp.out_thr = nullptr; sp.ovf = nullptr;
            vdb::launch_select(sp, nb, s);
        } else {
            // thresholds: the sample's kp-th score (padding queries were given -inf by query_prep)
            sp.out_keys = ix->cur->w_samp.p; sp.out_cnt = d_cnt_a; sp.out_thr = thr + q0; sp.ovf = nullptr;
            vdb::launch_select(sp, nb, s);
            const uint32_t n_wg = std::min<uint32_t>((nqt == 8 ? 1u : 2u) * (uint32_t)ix->n_cu / n_super, (n + 31) / 32);
            const uint32_t n_sub = vdb::fused_subpools_per_query(nqt, n_wg);
            vdb::FusedParams fp{ix->d_rows, ld, n, qp, q0, ix->d_alpha, ix->d_beta, d_rowmask ? d_rowmask : ix->d_live,
                                thr, ix->cur->w_pool.p - (size_t)q0 * n_sub * capl,
                                ix->cur->w_subcnt.p - (size_t)q0 * n_sub, capl, n_wg,
                                ix->kn.fused_ablate};
            const bool prof = ix->profile && !ix->cur->stats[8];       // with the screening tier on, ITS kernel is the one timed
            if (prof) HIP_TRY(hipEventRecord(ix->ev0, s));
            // 256-query passes: LDS-DMA staging, 3-image ring with the barrier in mid-stage; smaller batches: the
            // register-staged 128/64/32-query shapes.  (Diagnostics build: the 2-image and register-staged A/B variants.)
#ifdef VDB_DIAG
            if (nqt == 8 && ix->kn.regstage) vdb::launch_fused(fp, nqt, n_super, s);
            else if (nqt == 8 && ix->kn.dma2) vdb::launch_fused_dma(fp, n_super, s);
            else
#endif
            if (nqt == 8) vdb::launch_fused_dma3(fp, n_super, s);
            else vdb::launch_fused(fp, nqt, n_super, s);
            if (prof) {
                // one super-tile per event pair: wait here so the pair can be reused (profiling mode only)
                HIP_TRY(hipEventRecord(ix->ev1, s));
                HIP_TRY(hipEventSynchronize(ix->ev1));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
                ix->cur->stats[7] += (uint64_t)((double)ms * 1e6);
            }
            ix->cur->stats[3] += n;
            vdb::SelectParams mp{};
            mp.keys = ix->cur->w_pool.p; mp.stride = 0; mp.counts = nullptr; mp.n_fixed = 0; mp.cap = 0;
            mp.sub_counts = ix->cur->w_subcnt.p; mp.n_sub = n_sub; mp.capl = capl;
            mp.kk = kp; mp.out_keys = ix->cur->w_cand.p; mp.out_stride = kp; mp.out_cnt = d_cand_cnt;
            mp.out_thr = nullptr; mp.ovf = d_ovf + q0; mp.summary = d_status + 1;
            vdb::launch_select(mp, nb, s);
        }
        vdb::RerankParams rp{};
        rp.rows = ix->d_rows; rp.ld = ld; rp.dim = ix->dim; rp.n_rows = n;
        rp.qp = qp0; rp.qnorm = qnorm + q0; rp.nd = ix->d_nd; rp.row_ids = ix->d_row_ids;
        rp.rowmask = d_rowmask; rp.cand = ix->cur->w_cand.p; rp.cand_stride = kp; rp.cand_cnt = d_cand_cnt; rp.kp = kp;
        rp.metric = ix->metric; rp.k = (uint32_t)k; rp.eps_coef = eps; rp.nd2max_bits = ix->d_scalars;
        rp.out_ids = d_out_ids + (size_t)q0 * k; rp.out_dists = d_out_dists + (size_t)q0 * k;
        rp.out_counts = d_out_counts + q0; rp.out_stride = (uint32_t)k; rp.cert = d_cert + q0; rp.status = d_status;
        rp.thr = small ? nullptr : thr + q0;
        vdb::launch_rerank(rp, nb, s);
    }
    return VDB_OK;
}

// ------------------------------------------------------------------ tier: bf16 screening + certified re-rank
// Same structure, with the scores of the HBM-bound bf16 kernel (kernels_fused_bf16.hip): group minima of a row
// sample -> per-query threshold -> one pass over all rows keeping the keys under the threshold -> the kp smallest
// keys -> exact re-rank, certified with the bf16 error bound.  Queries come from ix->cur->w_qp / w_qb / w_qnorm.
int pass_bf16(Index* ix, hipStream_t s, uint32_t nq, size_t k, const Bf16Plan& pl, const uint32_t* d_rowmask,
              uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts, uint32_t* d_cert, uint32_t* d_ovf,
              uint32_t* d_status, float* d_thr_next, bool allow_alt = false) {
    int rc;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    const uint32_t S = pl.S, kp = pl.kp, KT = pl.kt;
    const uint32_t M = vdb::fused_bf16_sample_groups(S);
    // private sub-pools of 256 slots: when the rows near a query are stored next to each other (data ordered by
    // cluster) most of the ~N*kt/S keys that pass land in ONE workgroup's four sub-pools; 4 x 256 slots hold about twice
    // the expected total, so that case stays on this tier instead of overflowing into the next.  The gather reads
    // counts and keys, never empty slots, so the capacity costs address space only (0.5 GB of workspace at 1M rows).
    const uint32_t capl = 256;
    const uint32_t n_wg = std::min<uint32_t>((uint32_t)ix->n_cu, (n + vdb::fused_bf16_tile_rows() - 1) / vdb::fused_bf16_tile_rows());
    const uint32_t n_sub = vdb::fused_bf16_subpools_per_query(n_wg);
    // A batch above 256 queries takes several passes (BASELINE config 3: four).  They are independent, so they ALTERNATE between
    // this context and the handle's other workspace and stream when that one is idle: the latency-bound tail of pass i (its
    // slowest re-rank workgroups, a few CUs) then runs beside the head of pass i+1 instead of in front of it.  The per-query
    // arrays (queries, thresholds, flags, outputs) are indexed by q0 and shared; only the pass-local buffers are doubled.
    Workspace* alt = nullptr;
    if (allow_alt && nq > SUPER && !ix->profile && !ix->kn.rr_depth) {
        Workspace* o = (ix->cur == &ix->wsv[0]) ? &ix->wsv[1] : &ix->wsv[0];
        if (!o->busy) alt = o;
    }
    Workspace* const Wv[2] = {ix->cur, alt ? alt : ix->cur};
    const hipStream_t Sv[2] = {s, alt ? alt->stream : s};
    for (int t = 0; t < (alt ? 2 : 1); ++t) {
        Workspace* w = Wv[t];
        if ((rc = w->w_dense.ensure((size_t)SUPER * M))) return rc;
        if ((rc = w->w_cand.ensure((size_t)SUPER * kp))) return rc;
        if ((rc = w->w_samp.ensure((size_t)SUPER * std::max(kp, KT)))) return rc;
        if ((rc = w->w_pool.ensure((size_t)SUPER * n_sub * capl))) return rc;
        if ((rc = w->w_subcnt.ensure((size_t)SUPER * n_sub))) return rc;
        if ((rc = w->w_cnt.ensure(4 * SUPER + 16))) return rc;
    }
    // the sample pass runs over the compact bf16 copy of the sample rows when the row pitch allows (rebuilt here, before the
    // passes fork onto two streams, when rows were added since it was made -- mutators are refused while a search is in flight,
    // so nobody else is reading it; with ANOTHER search in flight and a different key this one keeps the f32 gather)
    bool sample_copy = ix->sample_cache && ld % 64 == 0 && !ix->kn.sample_block;
    if (sample_copy && (ix->sample16_n != n || ix->sample16_S != S)) {
        Workspace* o = (ix->cur == &ix->wsv[0]) ? &ix->wsv[1] : &ix->wsv[0];
        if (o->busy) sample_copy = false;
        else {
            const size_t need = (size_t)S * ld;
            if (ix->sample16_cap < need) {
                if (ix->d_sample16) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(ix->d_sample16); ix->d_sample16 = nullptr; ix->sample16_cap = 0; }
                if (hipMalloc((void**)&ix->d_sample16, need * 2) == hipSuccess) ix->sample16_cap = need;
                else { (void)hipGetLastError(); ix->d_sample16 = nullptr; sample_copy = false; }   // no memory for the optional copy: f32 gather
            }
            if (sample_copy) {
                vdb::launch_sample_to_bf16(ix->d_rows, ld, n, S, pl.shift, ix->d_sample16, s);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(s));                    // once per change of the rows: later searches on OTHER streams read it
                ix->sample16_n = n; ix->sample16_S = S;
            }
        }
    }
    if (alt) {                                                   // the other stream starts behind query_prep and the row mask
        if (!ix->ev_pass[0]) {
            HIP_TRY(hipEventCreateWithFlags(&ix->ev_pass[0], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ix->ev_pass[1], hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(ix->ev_pass[0], s));
        HIP_TRY(hipStreamWaitEvent(Sv[1], ix->ev_pass[0], 0));
    }
    ix->cur->stats[4] = S;
    const float eps = eps_coef(ix);
    for (uint32_t q0 = 0, pass = 0; q0 < nq; q0 += SUPER, ++pass) {
        const uint32_t nb = std::min(SUPER, nq - q0);
        Workspace* const W = Wv[pass & 1];                        // pass-local buffers
        const hipStream_t s = Sv[pass & 1];                       // (shadows the caller's stream inside the loop)
        uint32_t* d_cnt_a = W->w_cnt.p;
        uint32_t* d_cand_cnt = W->w_cnt.p + 2 * SUPER;
        vdb::FusedBf16Params fp{};
        fp.rows = ix->d_rows; fp.ld = ld; fp.n_rows = n; fp.qb = ix->cur->w_qb.p + (size_t)q0 * ld;
        fp.alpha = ix->d_alpha; fp.beta = ix->d_beta; fp.rowmask = d_rowmask ? d_rowmask : ix->d_live;
        fp.margin = ix->d_margin; fp.qg = ix->d_margin ? ix->cur->w_qg.p + q0 : nullptr;
        fp.thr = ix->cur->w_thr.p + q0; fp.pool = W->w_pool.p; fp.pool_cnt = W->w_subcnt.p; fp.capl = capl; fp.n_wg = n_wg;
        fp.scalars = ix->d_scalars; fp.qmax_bits = d_status + 2;
        fp.ablate = ix->kn.bf16_ablate;
        fp.n_sample = S; fp.sample_shift = pl.shift;
        fp.sample_block = ix->kn.sample_block ? (n / (S / 256u)) : 0u; fp.minkeys = W->w_dense.p; fp.minkey_stride = M;
        if (sample_copy) {
            vdb::FusedBf16Params sp16 = fp;
            sp16.rows16 = ix->d_sample16;
            vdb::launch_sample_s16(sp16, s);
        } else vdb::launch_sample_bf16(fp, (uint32_t)ix->n_cu, s);

        vdb::SelectParams sp{};
        sp.keys = W->w_dense.p; sp.stride = M; sp.counts = nullptr; sp.n_fixed = M; sp.cap = M; sp.kk = KT;
        sp.out_stride = KT; sp.out_keys = W->w_samp.p; sp.out_cnt = d_cnt_a; sp.out_thr = ix->cur->w_thr.p + q0; sp.ovf = nullptr;
        if (ix->d_margin) { sp.shift_g = ix->cur->w_qg.p + q0; sp.shift_m_bits = ix->d_scalars + 4; }   // plain-score sample -> lower-bound units
        vdb::launch_thr_select(sp, nb, s);

        if (ix->profile) HIP_TRY(hipEventRecord(ix->ev0, s));
        launch_filter_pass(ix, fp, s);
        if (ix->profile) {
            HIP_TRY(hipEventRecord(ix->ev1, s));
            HIP_TRY(hipEventSynchronize(ix->ev1));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
            ix->cur->stats[7] += (uint64_t)((double)ms * 1e6);
        }
#ifdef VDB_DIAG
        // An ablated launch leaves wrong pools behind; if the step went on with them every query would fall through to the
        // slower tiers, and the extra milliseconds of f32 MFMA work change the clock the NEXT timed launch runs at (ablation
        // arms with broken results read 10-25 us low for that reason alone).  So the ablated launch is the timed one, and an
        // unablated launch (untimed) overwrites its pools: every arm of an A/B then runs the same step around the kernel.
        if (fp.ablate) { fp.ablate = 0; launch_filter_pass(ix, fp, s); ix->cur->stats[3] += n; }
#endif
        ix->cur->stats[3] += n;

        vdb::SelectParams mp{};
        mp.keys = W->w_pool.p; mp.stride = 0; mp.counts = nullptr; mp.n_fixed = 0; mp.cap = 0;
        mp.sub_counts = W->w_subcnt.p; mp.n_sub = n_sub; mp.capl = capl; mp.wg_major = 1;
        mp.kk = kp; mp.out_keys = W->w_cand.p; mp.out_stride = kp; mp.out_cnt = d_cand_cnt;
        mp.out_thr = nullptr; mp.ovf = d_ovf + q0; mp.summary = d_status + 1;
        vdb::launch_select(mp, nb, s);

        vdb::RerankParams rp{};
        rp.rows = ix->d_rows; rp.ld = ld; rp.dim = ix->dim; rp.n_rows = n;
        rp.qp = ix->cur->w_qp.p + (size_t)q0 * ld; rp.qnorm = ix->cur->w_qnorm.p + q0; rp.nd = ix->d_nd; rp.row_ids = ix->d_row_ids;
        rp.rowmask = d_rowmask; rp.cand = W->w_cand.p; rp.cand_stride = kp; rp.cand_cnt = d_cand_cnt; rp.kp = kp;
        rp.metric = ix->metric; rp.k = (uint32_t)k; rp.eps_coef = eps; rp.nd2max_bits = ix->d_scalars;
        rp.out_ids = d_out_ids + (size_t)q0 * k; rp.out_dists = d_out_dists + (size_t)q0 * k;
        rp.out_counts = d_out_counts + q0; rp.out_stride = (uint32_t)k; rp.cert = d_cert + q0; rp.status = d_status;
        rp.thr = ix->cur->w_thr.p + q0;
        rp.qerr = ix->cur->w_qerr.p + q0; rp.c_acc = c_acc_bf16(ix); rp.lb_scores = ix->d_margin ? 1u : 0u;
        rp.kp_first = round_up((uint32_t)k + 38u, 16u); rp.kp_step = 32;
        rp.thr_next = d_thr_next ? d_thr_next + q0 : nullptr;
        // diagnostics build: the first re-rank round overridden, the depth each query ended at printed
        if (ix->kn.kp_first) rp.kp_first = ix->kn.kp_first;
        const bool dump_depth = ix->kn.rr_depth;
        if (dump_depth) {
            if ((rc = W->w_depth.ensure(SUPER * 17))) return rc;      // depth[q], then 8 x 64-bit phase stamps per query
            rp.depth = W->w_depth.p;
        }
        vdb::launch_rerank(rp, nb, s);
        if (dump_depth) {
            std::vector<uint32_t> dep((size_t)SUPER * 17);
            HIP_TRY(hipMemcpyAsync(dep.data(), W->w_depth.p, dep.size() * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            // phase stamps (s_memrealtime, 100 MHz): 0 start, 1 query row in LDS, 2 round 1 staged+folded, 3 sorted, 4 depth decided, 5 last round folded, 6 end
            const uint64_t* st64 = reinterpret_cast<const uint64_t*>(dep.data() + SUPER);
            uint64_t t0 = ~0ull;
            for (uint32_t q = 0; q < nb; ++q) t0 = std::min(t0, st64[(size_t)q * 8]);
            double med[7];
            for (int ph = 0; ph < 7; ++ph) {
                std::vector<double> v(nb);
                for (uint32_t q = 0; q < nb; ++q) v[q] = (double)(st64[(size_t)q * 8 + ph] - t0) * 0.01;
                std::sort(v.begin(), v.end());
                med[ph] = v[nb / 2];
                fprintf(stderr, "[vdb] re-rank phase %d at us: min %.2f median %.2f p90 %.2f max %.2f\n", ph, v[0], v[nb / 2], v[(size_t)nb * 9 / 10], v[nb - 1]);
            }
            (void)med;
            std::sort(dep.begin(), dep.begin() + nb);
            fprintf(stderr, "[vdb] re-rank depth of %u queries: min %u  p25 %u  median %u  p75 %u  p95 %u  max %u  (first round %u)\n", nb,
                    dep[0], dep[nb / 4], dep[nb / 2], dep[(size_t)nb * 3 / 4], dep[(size_t)nb * 95 / 100], dep[nb - 1], rp.kp_first);
        }
    }
    if (alt) {                                                   // the caller's stream continues behind BOTH chains
        HIP_TRY(hipEventRecord(ix->ev_pass[1], Sv[1]));
        HIP_TRY(hipStreamWaitEvent(s, ix->ev_pass[1], 0));
    }
    return VDB_OK;
}

// ------------------------------------------------------------------ the batched search
// ------------------------------------------------------------------ tier 0b: the re-threshold pass
// For queries the screening tier re-ranked to its depth limit without a certificate, the k-th exact distance found so
// far still bounds the answer: rerank_kernel turned it into a score cut above which no row can enter the top k.  The
// queries are gathered into a compact block, the HBM-bound filter pass runs once more with those cuts as thresholds,
// and EVERY key that passes (up to 2048 per query) is re-ranked exactly.  Exact by construction; a query whose list does
// not fit (pool overflow, more than 2048 keys) keeps its flag and goes on to the next tier.
// todo: batch indices; cuts: their score cuts.  On return flags2 (host) holds cert / overflow per compact query.
int pass_rethreshold(Index* ix, hipStream_t s, const std::vector<uint32_t>& todo, const std::vector<float>& cuts, size_t k,
                     const uint32_t* d_rowmask, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                     uint32_t* d_status, std::vector<uint32_t>& flags2) {
    int rc;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    const uint32_t nf = (uint32_t)todo.size(), nfp = round_up(nf, SUPER);
    constexpr uint32_t KMAX = 2048;                              // keys re-ranked per query at most (select capacity)
    const uint32_t capl = 256;
    const uint32_t n_wg = std::min<uint32_t>((uint32_t)ix->n_cu, (n + vdb::fused_bf16_tile_rows() - 1) / vdb::fused_bf16_tile_rows());
    const uint32_t n_sub = vdb::fused_bf16_subpools_per_query(n_wg);
    if ((rc = ix->cur->w2_qp.ensure((size_t)nfp * ld))) return rc;
    if ((rc = ix->cur->w2_qnorm.ensure(nfp))) return rc;
    if ((rc = ix->cur->w2_thr.ensure(nfp))) return rc;
    if ((rc = ix->cur->w2_qerr.ensure(nfp))) return rc;
    if ((rc = ix->cur->w2_qg.ensure(nfp))) return rc;
    if ((rc = ix->cur->w2_qb.ensure((size_t)nfp * ld))) return rc;
    if ((rc = ix->cur->w2_outi.ensure((size_t)nf * k))) return rc;
    if ((rc = ix->cur->w2_outd.ensure((size_t)nf * k))) return rc;
    if ((rc = ix->cur->w2_outc.ensure(nf))) return rc;
    if ((rc = ix->cur->w2_flags.ensure(2 * (size_t)nf))) return rc;
    if ((rc = ix->cur->w2_qidx.ensure(nf))) return rc;
    if ((rc = ix->cur->w2_cand.ensure((size_t)SUPER * KMAX))) return rc;
    if ((rc = ix->cur->w_pool.ensure((size_t)SUPER * n_sub * capl))) return rc;
    if ((rc = ix->cur->w_subcnt.ensure((size_t)SUPER * n_sub))) return rc;
    uint32_t* d_cert2 = ix->cur->w2_flags.p;
    uint32_t* d_ovf2 = ix->cur->w2_flags.p + nf;
    HIP_TRY(hipMemcpyAsync(ix->cur->w2_qidx.p, todo.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
    vdb::launch_gather_queries(ix->cur->w_qp.p, ix->cur->w_qnorm.p, ld, ix->cur->w2_qidx.p, nf, nfp, ix->cur->w2_qp.p, ix->cur->w2_qnorm.p, ix->cur->w2_thr.p, s);
    // bf16 image, |q - bf16(q)| and zeroed flags of the compact block (the rows are already padded: dim = ld)
    vdb::QueryPrepParams qp{ix->cur->w2_qp.p, ld, nf, ix->cur->w2_qp.p, ld, nfp, ix->cur->w2_qnorm.p, ix->cur->w2_thr.p, vdb::EUCLID, d_status,
                            ix->cur->w2_qb.p, ix->cur->w2_qerr.p, ix->d_margin ? ix->cur->w2_qg.p : nullptr, margin_plan(ix).kappa, d_cert2, d_ovf2};
    vdb::launch_query_prep(qp, s);
    HIP_TRY(hipMemcpyAsync(ix->cur->w2_thr.p, cuts.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));   // padding queries keep -inf
    uint32_t* d_cand_cnt = ix->cur->w_cnt.p + 2 * SUPER;
    for (uint32_t q0 = 0; q0 < nf; q0 += SUPER) {
        const uint32_t nb = std::min(SUPER, nf - q0);
        vdb::FusedBf16Params fp{};
        fp.rows = ix->d_rows; fp.ld = ld; fp.n_rows = n; fp.qb = ix->cur->w2_qb.p + (size_t)q0 * ld;
        fp.alpha = ix->d_alpha; fp.beta = ix->d_beta; fp.rowmask = d_rowmask ? d_rowmask : ix->d_live;
        fp.margin = ix->d_margin; fp.qg = ix->d_margin ? ix->cur->w2_qg.p + q0 : nullptr;
        fp.thr = ix->cur->w2_thr.p + q0; fp.pool = ix->cur->w_pool.p; fp.pool_cnt = ix->cur->w_subcnt.p; fp.capl = capl; fp.n_wg = n_wg;
        fp.scalars = ix->d_scalars; fp.qmax_bits = d_status + 2;
        launch_filter_pass(ix, fp, s);
        ix->cur->stats[3] += n;
        vdb::SelectParams mp{};
        mp.keys = ix->cur->w_pool.p; mp.sub_counts = ix->cur->w_subcnt.p; mp.n_sub = n_sub; mp.capl = capl; mp.wg_major = 1;
        mp.kk = KMAX; mp.out_keys = ix->cur->w2_cand.p; mp.out_stride = KMAX; mp.out_cnt = d_cand_cnt;
        mp.ovf = d_ovf2 + q0; mp.summary = nullptr; mp.flag_truncation = 1;
        vdb::launch_select(mp, nb, s);
        vdb::RerankParams rp{};
        rp.rows = ix->d_rows; rp.ld = ld; rp.dim = ix->dim; rp.n_rows = n;
        rp.qp = ix->cur->w2_qp.p + (size_t)q0 * ld; rp.qnorm = ix->cur->w2_qnorm.p + q0; rp.nd = ix->d_nd; rp.row_ids = ix->d_row_ids;
        rp.rowmask = d_rowmask; rp.cand = ix->cur->w2_cand.p; rp.cand_stride = KMAX; rp.cand_cnt = d_cand_cnt; rp.kp = KMAX;
        rp.metric = ix->metric; rp.k = (uint32_t)k; rp.nd2max_bits = ix->d_scalars;
        rp.out_ids = ix->cur->w2_outi.p + (size_t)q0 * k; rp.out_dists = ix->cur->w2_outd.p + (size_t)q0 * k;
        rp.out_counts = ix->cur->w2_outc.p + q0; rp.out_stride = (uint32_t)k; rp.cert = d_cert2 + q0; rp.status = d_status;
        vdb::launch_rerank_all(rp, nb, s);
    }
    HIP_TRY(hipGetLastError());
    flags2.assign(2 * (size_t)nf, 0u);
    HIP_TRY(hipMemcpyAsync(flags2.data(), ix->cur->w2_flags.p, 2 * (size_t)nf * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // only the queries this pass answered completely are written back
    std::vector<uint32_t> good;
    for (uint32_t j = 0; j < nf; ++j) if (flags2[j] && !flags2[nf + j]) good.push_back(j);
    if (!good.empty()) {
        // scatter compact results j -> batch position todo[j] (the scatter kernel walks a (source, destination) list)
        std::vector<uint32_t> src_dst(2 * good.size());
        for (size_t i = 0; i < good.size(); ++i) { src_dst[i] = good[i]; src_dst[good.size() + i] = todo[good[i]]; }
        if ((rc = ix->cur->w2_qidx.ensure(2 * good.size()))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->cur->w2_qidx.p, src_dst.data(), src_dst.size() * 4, hipMemcpyHostToDevice, s));
        vdb::launch_scatter_results_list(ix->cur->w2_outi.p, ix->cur->w2_outd.p, ix->cur->w2_outc.p, ix->cur->w2_qidx.p, ix->cur->w2_qidx.p + good.size(),
                                         (uint32_t)good.size(), (uint32_t)k, d_out_ids, d_out_dists, d_out_counts, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
    }
    return VDB_OK;
}

// Part 1: checks, workspace, and the FIRST tier enqueued on the stream -- no host synchronisation unless the search is
// one of the cases answered completely here (empty store, k = 0, k too large for the MFMA tiers).
int search_part1(Index* ix, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_idmask,
                 size_t mask_bits, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                 hipStream_t user_stream, bool allow_alt = false) {
    int rc;
    ix->cur->ctx.pending = false;
    if ((rc = set_device(ix))) return rc;
    if ((rc = flush(ix))) return rc;
    if (nq == 0) return VDB_OK;
    // all launches of this search go to the caller's stream when one is given (so that the caller's
    // events bracket them); the workspace is protected by the handle mutex and the final sync
    hipStream_t s = user_stream ? user_stream : ix->cur->stream;
    memset(ix->cur->stats, 0, sizeof(ix->cur->stats));
    ix->cur->stats[14] = shadow_usable(ix) ? 1u : 0u;   // the screening pass reads the bf16 shadow rows
    ix->cur->stats[15] = ix->kn.any ? 1u : 0u;          // diagnostics build with a knob set: the run is NOT covered by the exactness guarantee
    const auto t_entry = std::chrono::steady_clock::now();
    auto since = [&]() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_entry).count(); };
    size_t total_rows = ix->n_live + ix->misfits.size();
    if (total_rows == 0 || k == 0) {   // storage.rs:218-220: empty store -> Ok(vec![]) before any check
        HIP_TRY(hipMemsetAsync(d_out_counts, 0, nq * 4, s));
        HIP_TRY(hipStreamSynchronize(s));
        return VDB_OK;
    }
    // distance.rs:21-26: the first row whose dimension differs from the query's fails the search
    if (ix->n_live && ix->dim != dim) return fail_dim(dim, ix->dim);
    for (auto& kv : ix->misfits)
        if (kv.second.size() != dim) return fail_dim(dim, kv.second.size());
    if (!ix->misfits.empty()) {
        // every stored row has the query's dimension but none is on the device (zero-length rows)
        return fail(VDB_ERR_INVALID_ARGUMENT, "zero-dimensional vectors are not searchable");
    }
    if (ix->metric == vdb::COSINE) {
        if ((rc = ensure_zero_count(ix))) return rc;
        if (ix->zero_live)   // distance.rs:51-55 aborts the whole search (flat_index.rs:57-60)
            return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
    }
    if (nq > 0x7fffffffull / 2 || k > 0x7fffffffull) return fail(VDB_ERR_INVALID_ARGUMENT, "batch too large");
    if (ix->dim > 16384) return fail(VDB_ERR_INVALID_ARGUMENT, "dimension %u exceeds the supported 16384", ix->dim);

    const uint32_t n = ix->n_uploaded;
    const uint32_t ld = ix->ld;
    const uint32_t nq32 = (uint32_t)nq;
    const uint32_t bp_all = round_up(nq32, SUPER);
    const uint32_t kp = pick_kp(k);

    // ---- workspace
    if ((rc = ix->cur->w_qp.ensure((size_t)bp_all * ld))) return rc;
    if ((rc = ix->cur->w_qnorm.ensure(bp_all))) return rc;
    if ((rc = ix->cur->w_thr.ensure(bp_all))) return rc;
    if ((rc = ix->cur->w_flags.ensure(4 + 3 * (size_t)nq32))) return rc;      // status block | cert | overflow | score cut per query
    if (ix->cur->h_flags_n < 4 + 3 * (size_t)nq32) {
        if (ix->cur->h_flags) (void)hipHostFree(ix->cur->h_flags);
        ix->cur->h_flags = nullptr;
        ix->cur->h_flags_n = 0;
        size_t want = 4 + 3 * (size_t)nq32 + 1024;
        HIP_TRY(hipHostMalloc((void**)&ix->cur->h_flags, want * 4, hipHostMallocDefault));
        ix->cur->h_flags_n = want;
    }
    uint32_t* d_status = ix->cur->w_flags.p;        // [0] status bits
    uint32_t* d_cert = ix->cur->w_flags.p + 4;      // [nq]
    uint32_t* d_ovf = d_cert + nq32;           // [nq]
    // the per-query flags are zeroed by query_prep; the 16-byte status block only needs a memset when the last
    // search left it set (or the buffer is new) -- one launch less at the head of every search
    const bool flags_by_prep = kp != 0 || (ix->screen && plan_bf16(ix, n, k).kp);
    if (!flags_by_prep) HIP_TRY(hipMemsetAsync(ix->cur->w_flags.p, 0, (4 + 3 * (size_t)nq32) * 4, s));
    else if (ix->cur->status_dirty || ix->cur->w_flags.p != ix->cur->status_buf) {
        HIP_TRY(hipMemsetAsync(ix->cur->w_flags.p, 0, 16, s));
        ix->cur->status_buf = ix->cur->w_flags.p;
    }
    ix->cur->status_dirty = true;                               // until a clean status word has been read back

    // ---- eligibility mask: tombstones, optionally AND the caller's id filter
    const uint32_t* d_rowmask = (ix->n_live == n) ? nullptr : ix->d_live;
    if (d_idmask) {
        if ((rc = ix->cur->w_rowmask.ensure((n + 31) / 32))) return rc;
        vdb::launch_build_rowmask(ix->d_row_ids, d_rowmask, d_idmask, mask_bits, n, ix->cur->w_rowmask.p, s);
        d_rowmask = ix->cur->w_rowmask.p;
    }

    // ---- queries: zero-padded copy + exact-order norms
    {
        uint16_t* qb = nullptr;
        if (ix->screen && plan_bf16(ix, n, k).kp) {
            if ((rc = ix->cur->w_qb.ensure((size_t)bp_all * ld))) return rc;
            if ((rc = ix->cur->w_qerr.ensure(bp_all))) return rc;
            if ((rc = ix->cur->w_qg.ensure(bp_all))) return rc;
            qb = ix->cur->w_qb.p;
        }
        vdb::QueryPrepParams qp{d_q, (uint32_t)dim, nq32, ix->cur->w_qp.p, ld, bp_all, ix->cur->w_qnorm.p, ix->cur->w_thr.p, ix->metric, d_status, qb,
                                ix->cur->w_qerr.p, (qb && ix->d_margin) ? ix->cur->w_qg.p : nullptr, margin_plan(ix).kappa,
                                flags_by_prep ? d_cert : nullptr, flags_by_prep ? d_ovf : nullptr};
        vdb::launch_query_prep(qp, s);
    }

    if (kp == 0 && !(ix->screen && plan_bf16(ix, n, k).kp)) {
        // large k: exact scan for every query
        HIP_TRY(hipMemcpyAsync(ix->cur->h_flags, ix->cur->w_flags.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (ix->cur->h_flags[0] & vdb::ST_ZERO_QUERY)
            return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
        for (uint32_t q = 0; q < nq32; ++q) {
            if ((rc = exact_one(ix, s, q, k, d_rowmask, d_out_ids + (size_t)q * k, d_out_dists + (size_t)q * k,
                                d_out_counts + q)))
                return rc;
        }
        ix->cur->stats[1] = nq32;
        HIP_TRY(hipMemcpyAsync(ix->cur->h_flags, ix->cur->w_flags.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (ix->cur->h_flags[0] & vdb::ST_NAN) return fail(VDB_ERR_NAN, "NaN distance (the reference panics here, flat_index.rs:62)");
        return VDB_OK;
    }

    // ---- tiers.  Large indexes: the bf16 screening tier first (HBM-bound pass), the queries it cannot certify
    // are re-run as a compact block by the f32 MFMA tier; whatever that cannot certify goes to the exact scan.
    const Bf16Plan pl16 = ix->screen ? plan_bf16(ix, n, k) : Bf16Plan{};
    const uint32_t kp16 = pl16.kp;
    if ((rc = ix->cur->w_cnt.ensure(4 * SUPER + 16))) return rc;
    if (kp16) {
        ix->cur->stats[8] = 1;
        ix->cur->stats[5] = kp16;
        if ((rc = pass_bf16(ix, s, nq32, k, pl16, d_rowmask, d_out_ids, d_out_dists, d_out_counts, d_cert, d_ovf, d_status,
                            reinterpret_cast<float*>(d_ovf + nq32), allow_alt)))
            return rc;
    } else {
        if ((rc = pass_f32(ix, s, ix->cur->w_qp.p, ix->cur->w_qnorm.p, ix->cur->w_thr.p, nq32, k, kp, d_rowmask, d_out_ids, d_out_dists,
                           d_out_counts, d_cert, d_ovf, d_status)))
            return rc;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ix->cur->h_flags, ix->cur->w_flags.p, (4 + 3 * (size_t)nq32) * 4, hipMemcpyDeviceToHost, s));
    ix->cur->stats[10] = since();                       // host time until everything of the first tier is enqueued, ns
    Workspace::SearchCtx& c = ix->cur->ctx;
    c.pending = true; c.nq32 = nq32; c.kp = kp; c.kp16 = kp16; c.k = k; c.s = s; c.d_rowmask = d_rowmask;
    c.d_out_ids = d_out_ids; c.d_out_dists = d_out_dists; c.d_out_counts = d_out_counts; c.t_entry = t_entry;
    return VDB_OK;
}

// Part 2: wait for the first tier, read its flags, run the fallback tiers for the queries it could not certify.
// *changed (may be null) tells whether outputs were rewritten after part 1's pass.
int search_part2(Index* ix, int* changed) {
    if (changed) *changed = 0;
    Workspace::SearchCtx& c = ix->cur->ctx;
    if (!c.pending) return VDB_OK;
    c.pending = false;
    int rc;
    const uint32_t nq32 = c.nq32, kp = c.kp, kp16 = c.kp16;
    const size_t k = c.k;
    hipStream_t s = c.s;
    const uint32_t* d_rowmask = c.d_rowmask;
    uint64_t* d_out_ids = c.d_out_ids; float* d_out_dists = c.d_out_dists; uint32_t* d_out_counts = c.d_out_counts;
    const uint32_t n = ix->n_uploaded, ld = ix->ld;
    uint32_t* d_status = ix->cur->w_flags.p;
    const auto t_entry = c.t_entry;
    auto since = [&]() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_entry).count(); };
    HIP_TRY(hipStreamSynchronize(s));
    ix->cur->stats[11] = since();                       // ... until the first tier's flags are on the host, ns
    uint32_t status = ix->cur->h_flags[0];
    if (status & vdb::ST_ZERO_QUERY)
        return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
    // vdb_flat_set_tiers: forced hand-over to the slower tiers (tests); every tier returns the same results
    const bool force_exact = (ix->tiers & VDB_TIERS_FORCE_EXACT) != 0;
    const bool force_f32 = (ix->tiers & VDB_TIERS_FORCE_F32) != 0;
    const bool no_rethr = (ix->tiers & VDB_TIERS_NO_RETHRESHOLD) != 0;
    std::vector<uint32_t> todo;
    for (uint32_t q = 0; q < nq32; ++q) {
        bool cert = ix->cur->h_flags[4 + q] != 0, ovf = ix->cur->h_flags[4 + nq32 + q] != 0;
        if (ovf) ++ix->cur->stats[2];
        if (!cert) ++ix->cur->stats[6];
        if (cert && !ovf && !force_exact && !(kp16 && force_f32)) continue;
        todo.push_back(q);
    }
    if (changed && !todo.empty()) *changed = 1;
    if (kp16 && !todo.empty() && !no_rethr && !force_exact && !force_f32) {
        // ---- tier 0b: queries with a known score cut get one more HBM-bound pass with that cut as the threshold
        const uint32_t* h_ovf = ix->cur->h_flags + 4 + nq32;
        const float* h_cut = reinterpret_cast<const float*>(ix->cur->h_flags + 4 + 2 * (size_t)nq32);
        std::vector<uint32_t> sel, rest;
        std::vector<float> cuts;
        for (uint32_t q : todo) {
            const float c = h_cut[q];
            if (!h_ovf[q] && c == c && std::isfinite(c)) { sel.push_back(q); cuts.push_back(c); }
            else rest.push_back(q);
        }
        if (!sel.empty()) {
            std::vector<uint32_t> fl;
            if ((rc = pass_rethreshold(ix, s, sel, cuts, k, d_rowmask, d_out_ids, d_out_dists, d_out_counts, d_status, fl))) return rc;
            const uint32_t nf = (uint32_t)sel.size();
            for (uint32_t j = 0; j < nf; ++j) {
                if (fl[j] && !fl[nf + j]) ++ix->cur->stats[13];
                else { rest.push_back(sel[j]); if (fl[nf + j]) ++ix->cur->stats[2]; }
            }
            std::sort(rest.begin(), rest.end());
        }
        todo.swap(rest);
    }
    if (kp16 && !todo.empty()) {
        // ---- second tier: the uncertified queries as one compact block through the f32 MFMA pipeline
        const uint32_t nf = (uint32_t)todo.size(), nfp = round_up(nf, SUPER);
        ix->cur->stats[9] = nf;
        if ((rc = ix->cur->w2_qp.ensure((size_t)nfp * ld))) return rc;
        if ((rc = ix->cur->w2_qnorm.ensure(nfp))) return rc;
        if ((rc = ix->cur->w2_thr.ensure(nfp))) return rc;
        if ((rc = ix->cur->w2_outi.ensure((size_t)nf * k))) return rc;
        if ((rc = ix->cur->w2_outd.ensure((size_t)nf * k))) return rc;
        if ((rc = ix->cur->w2_outc.ensure(nf))) return rc;
        if ((rc = ix->cur->w2_flags.ensure(2 * (size_t)nf))) return rc;
        if ((rc = ix->cur->w2_qidx.ensure(nf))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->cur->w2_qidx.p, todo.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(ix->cur->w2_flags.p, 0, 2 * (size_t)nf * 4, s));
        vdb::launch_gather_queries(ix->cur->w_qp.p, ix->cur->w_qnorm.p, ld, ix->cur->w2_qidx.p, nf, nfp, ix->cur->w2_qp.p, ix->cur->w2_qnorm.p,
                                   ix->cur->w2_thr.p, s);
        if (kp == 0) {
            // k too large for the f32 tier as well: straight to the exact scan (flags stay 0 = uncertified)
        } else {
            if ((rc = pass_f32(ix, s, ix->cur->w2_qp.p, ix->cur->w2_qnorm.p, ix->cur->w2_thr.p, nf, k, kp, d_rowmask, ix->cur->w2_outi.p,
                               ix->cur->w2_outd.p, ix->cur->w2_outc.p, ix->cur->w2_flags.p, ix->cur->w2_flags.p + nf, d_status)))
                return rc;
            vdb::launch_scatter_results(ix->cur->w2_outi.p, ix->cur->w2_outd.p, ix->cur->w2_outc.p, ix->cur->w2_qidx.p, nf, (uint32_t)k,
                                        d_out_ids, d_out_dists, d_out_counts, s);
        }
        HIP_TRY(hipGetLastError());
        std::vector<uint32_t> f2(2 * (size_t)nf + 4);
        HIP_TRY(hipMemcpyAsync(f2.data(), ix->cur->w2_flags.p, 2 * (size_t)nf * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(f2.data() + 2 * (size_t)nf, ix->cur->w_flags.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        status |= f2[2 * (size_t)nf];
        if (status & vdb::ST_ZERO_QUERY)
            return fail(VDB_ERR_INVALID_VECTOR, "Invalid vector: Cannot compute cosine distance with zero vector");
        std::vector<uint32_t> todo2;
        for (uint32_t j = 0; j < nf; ++j) {
            bool cert = f2[j] != 0, ovf = f2[nf + j] != 0;
            if (ovf) ++ix->cur->stats[2];
            if (cert && !ovf && !force_exact) continue;
            todo2.push_back(todo[j]);
        }
        todo.swap(todo2);
    }
    // ---- exact fallback for the queries the MFMA tiers could not certify.  Up to 8 of them share one pass
    // over the rows; a row survives for a query only if its exact distance is <= the k-th exact distance
    // the re-rank already found (a valid upper bound), so each query is left with a handful of keys.
    uint32_t n_fallback = 0;
    n_fallback = (uint32_t)todo.size();
    if (!todo.empty()) {
        const uint32_t cap = 32768;
        if ((rc = ensure_ranks(ix))) return rc;
        if ((rc = ix->cur->w_exact.ensure(std::max<size_t>((size_t)8 * cap, n)))) return rc;
        if ((rc = ix->cur->w_exsel.ensure((size_t)8 * MAX_SELECT + 8))) return rc;
        uint32_t* d_cnt8 = ix->cur->w_cnt.p + 3 * SUPER;            // [8] survivors per query, [8..16) select counts
        std::vector<uint32_t> dense;                             // queries whose bounded pass overflowed
        for (size_t g0 = 0; g0 < todo.size(); g0 += 8) {
            const uint32_t nqf = (uint32_t)std::min<size_t>(8, todo.size() - g0);
            HIP_TRY(hipMemsetAsync(d_cnt8, 0, 16 * 4, s));
            vdb::ExactMultiParams ep{};
            ep.rows = ix->d_rows; ep.ld = ld; ep.dim = ix->dim; ep.n_rows = n; ep.qp = ix->cur->w_qp.p; ep.qnorm = ix->cur->w_qnorm.p;
            ep.nd = ix->d_nd; ep.rowmask = d_rowmask; ep.idrank = ix->ids_monotone ? nullptr : ix->d_idrank.p;
            ep.metric = ix->metric; ep.nqf = nqf;
            for (uint32_t j = 0; j < nqf; ++j) ep.qidx[j] = todo[g0 + j];
            ep.prev_dists = d_out_dists; ep.prev_counts = d_out_counts; ep.k = (uint32_t)k;
            ep.keys = ix->cur->w_exact.p; ep.cap = cap; ep.cnt = d_cnt8; ep.status = d_status;
            vdb::launch_exact_multi(ep, s);
            uint32_t h_cnt[8];
            HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt8, nqf * 4, hipMemcpyDeviceToHost, s));
            vdb::SelectParams sp{};
            sp.keys = ix->cur->w_exact.p; sp.stride = cap; sp.counts = d_cnt8; sp.n_fixed = 0; sp.cap = cap; sp.kk = (uint32_t)k;
            sp.out_keys = ix->cur->w_exsel.p; sp.out_stride = MAX_SELECT; sp.out_cnt = d_cnt8 + 8;
            vdb::launch_select(sp, nqf, s);
            vdb::EmitMultiParams em{};
            em.keys = ix->cur->w_exsel.p; em.key_stride = MAX_SELECT; em.cnt = d_cnt8 + 8;
            em.rank2row = ix->ids_monotone ? nullptr : ix->d_rank2row.p; em.row_ids = ix->d_row_ids;
            em.out_ids = d_out_ids; em.out_dists = d_out_dists; em.out_count = d_out_counts; em.k = (uint32_t)k; em.nqf = nqf;
            for (uint32_t j = 0; j < nqf; ++j) em.qidx[j] = todo[g0 + j];
            vdb::launch_emit_multi(em, s);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
            for (uint32_t j = 0; j < nqf; ++j)
                if (h_cnt[j] > cap) dense.push_back(todo[g0 + j]);   // e.g. every row ties with the bound
        }
        for (uint32_t q : dense)
            if ((rc = exact_one(ix, s, q, k, d_rowmask, d_out_ids + (size_t)q * k, d_out_dists + (size_t)q * k,
                                d_out_counts + q)))
                return rc;
    }
    ix->cur->stats[12] = since();                       // whole call, ns
    ix->cur->stats[0] = nq32 - n_fallback;
    ix->cur->stats[1] = n_fallback;
    uint32_t st2 = status;
    if (n_fallback) {
        HIP_TRY(hipMemcpyAsync(ix->cur->h_flags, ix->cur->w_flags.p, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        st2 |= ix->cur->h_flags[0];
    }
    // status bits, the summary word, or a query norm beyond the kernels' no-NaN domain (word 2 is a running maximum: tame
    // values may stay, a wild one must not outlive its search)
    ix->cur->status_dirty = st2 != 0 || ix->cur->stats[6] != 0 || ix->cur->stats[2] != 0 || ix->cur->h_flags[2] > 0x53800000u;
    if (st2 & vdb::ST_NAN)
        return fail(VDB_ERR_NAN, "NaN distance (the reference panics here, flat_index.rs:62)");
    return VDB_OK;
}

void publish_stats(Index* ix) { memcpy(ix->stats, ix->cur->stats, sizeof(ix->stats)); }

// searches submitted and not yet waited for (vdb_flat_search_batch_device_submit): the row store must not change under them
bool in_flight(const Index* ix) { return ix->wsv && (ix->wsv[0].busy || ix->wsv[1].busy); }
int refuse_in_flight() { return fail(VDB_ERR_INVALID_ARGUMENT, "a submitted search is still in flight on this handle: wait for it first"); }

int search_device(Index* ix, const float* d_q, size_t nq, size_t dim, size_t k, const uint64_t* d_idmask,
                  size_t mask_bits, uint64_t* d_out_ids, float* d_out_dists, uint32_t* d_out_counts,
                  hipStream_t user_stream) {
    // a synchronous search takes a context no submitted search is using
    ix->cur = ix->wsv[0].busy ? &ix->wsv[1] : &ix->wsv[0];
    if (ix->cur->busy) return fail(VDB_ERR_INVALID_ARGUMENT, "two submitted searches are in flight on this handle: wait for one first");
    // (the other workspace may serve the alternating passes of a large batch: the handle mutex is held until part 2 is done,
    // so no submit can claim it meanwhile)
    int rc = search_part1(ix, d_q, nq, dim, k, d_idmask, mask_bits, d_out_ids, d_out_dists, d_out_counts, user_stream, !in_flight(ix));
    if (rc) { ix->cur->ctx.pending = false; publish_stats(ix); ix->cur = &ix->wsv[0]; return rc; }
    rc = search_part2(ix, nullptr);
    publish_stats(ix);
    ix->cur = &ix->wsv[0];
    return rc;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int vdb_abi_version(void) { return 1; }
const char* vdb_build_arch(void) { return "gfx950"; }

void vdb_last_error(char* buf, size_t cap, size_t* expected, size_t* actual) {
    if (buf && cap) {
        size_t n = std::min(cap - 1, g_err.size());
        memcpy(buf, g_err.data(), n);
        buf[n] = 0;
    }
    if (expected) *expected = g_expected;
    if (actual) *actual = g_actual;
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
    (void)hipSetDevice(ix->device);
    (void)hipStreamSynchronize(ix->stream);
    free_store(ix);
    if (ix->d_scalars) (void)hipFree(ix->d_scalars);
    ix->d_idrank.release(); ix->d_rank2row.release();
    for (int w = 0; ix->wsv && w < 2; ++w) {
        Workspace& W = ix->wsv[w];
        W.for_each_buffer([](auto& buf) { buf.release(); });
        if (W.h_flags) (void)hipHostFree(W.h_flags);
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

size_t vdb_flat_len(const vdb_flat_index* ix) { return ix ? ix->n_live + ix->misfits.size() : 0; }
int vdb_flat_metric(const vdb_flat_index* ix) { return ix ? ix->metric : -1; }
size_t vdb_flat_dim(const vdb_flat_index* ix) { return ix ? ix->dim : 0; }

int vdb_flat_reserve(vdb_flat_index* ix, size_t rows, size_t dim) {
    return guarded([&]() -> int {
    if (!ix || !dim) return fail(VDB_ERR_INVALID_ARGUMENT, "bad argument");
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
    ix->mu.lock();
    if (ix->begin_locked) { ix->mu.unlock(); return fail(VDB_ERR_INVALID_ARGUMENT, "a search is already pending on this handle"); }
    if (in_flight(ix)) { ix->mu.unlock(); return refuse_in_flight(); }
    ix->cur = &ix->wsv[0];
    // (the handle is locked by hand here: an exception must not skip the unlock below)
    int rc = guarded([&]() -> int { return search_part1(ix, d_queries, nq, dim, k, d_id_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts, (hipStream_t)stream, true); });
    if (rc == VDB_OK && d_code) {
        hipStream_t s = stream ? (hipStream_t)stream : ix->stream;
        if (ix->cur->ctx.pending) vdb::launch_write_code(ix->cur->w_flags.p, d_code, s);
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
    *ticket = -1;
    std::lock_guard<std::mutex> g(ix->mu);
    if (ix->begin_locked) return fail(VDB_ERR_INVALID_ARGUMENT, "a search is pending between begin and finish");
    if (ix->profile) return fail(VDB_ERR_INVALID_ARGUMENT, "kernel profiling synchronises inside the search: not available for submitted searches");
    int slot = !ix->wsv[0].busy ? 0 : (!ix->wsv[1].busy ? 1 : -1);
    if (slot < 0) return fail(VDB_ERR_INVALID_ARGUMENT, "two searches are already in flight on this handle");
    ix->cur = &ix->wsv[slot];
    int rc = search_part1(ix, d_queries, nq, dim, k, d_id_mask, mask_bits, d_out_ids, d_out_dists, d_out_counts, (hipStream_t)stream);
    if (rc) { ix->cur->ctx.pending = false; ix->cur = &ix->wsv[0]; return rc; }
    ix->cur->busy = true;                       // (a search part 1 answered completely has pending = false: wait() returns at once)
    ix->cur = &ix->wsv[0];
    *ticket = slot;
    return VDB_OK;
    });
}

int vdb_flat_search_batch_device_wait(vdb_flat_index* ix, int ticket) {
    return guarded([&]() -> int {
    if (!ix || ticket < 0 || ticket > 1) return fail(VDB_ERR_INVALID_ARGUMENT, "bad ticket");
    std::lock_guard<std::mutex> g(ix->mu);
    if (!ix->wsv[ticket].busy) return fail(VDB_ERR_INVALID_ARGUMENT, "no submitted search behind this ticket");
    int rc = set_device(ix);
    if (rc) return rc;
    ix->cur = &ix->wsv[ticket];
    rc = search_part2(ix, nullptr);
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
    for (size_t i = 0; i < n; ++i) out[i] = i < 16 ? ix->cur->stats[i] : 0;
    return VDB_OK;
    });
}

// ------------------------------------------------------------------ certificate diagnostics (include/vdb_flat.h)
int vdb_flat_debug_screen_scores(vdb_flat_index* ix, const float* queries, size_t nq, size_t dim, int raw, float* out_scores,
                                 float* out_qinfo, double* out_consts) {
    return guarded([&]() -> int {
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

size_t vdb_flat_debug_rows(const vdb_flat_index* ix) { return ix ? ix->row_ids.size() : 0; }

int vdb_flat_debug_last_thresholds(vdb_flat_index* ix, float* out, size_t nq) {
    return guarded([&]() -> int {
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
    if (!ix || (flags & ~7u)) return fail(VDB_ERR_INVALID_ARGUMENT, "flags must be a combination of VDB_TIERS_*");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->tiers = flags;
    return VDB_OK;
    });
}

int vdb_flat_set_screen(vdb_flat_index* ix, int mode) {
    return guarded([&]() -> int {
    if (!ix || mode < 0 || mode > 1) return fail(VDB_ERR_INVALID_ARGUMENT, "mode must be 0 (f32 MFMA tier only) or 1 (bf16 screening tier first)");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->screen = mode;
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
    vdb::launch_pair_eval(pp, ix->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ix->stream));
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
