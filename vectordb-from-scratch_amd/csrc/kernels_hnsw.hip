// kernels_hnsw.hip -- device-resident HNSW search (src/hnsw/graph.rs:143-199 search_layer, :386-412 search_knn):
// the graph of vdb_hnsw.cpp is mirrored in HBM and ONE workgroup walks it for one query, so a batch needs one launch
// instead of one launch + host round trip per traversal round.
//
// The walk is the reference's, operation by operation.  The two priority queues -- Rust's BinaryHeap with the standard
// library's element moves (neighbor_queue.rs; the backing array decides the order of equal distances in into_sorted_vec) --
// live in LDS; thread 0 pops the next candidate, WAVE 0 pushes an expansion's accepted neighbours (wave_sift_up /
// wave_pop_discard below: one lane per heap level, a ballot where a single lane would chase LDS latencies); the neighbour list
// of the popped candidate is scanned by the lanes of wave 0 (visited set = open-addressing hash in LDS, insertion order restored
// by a ballot compaction); the distances of the up to 33 unvisited neighbours are evaluated in the reference's exact f32
// operation order (distance.rs:37-73): their rows are staged through LDS in chunks by all waves (LDS-DMA, two chunks in
// flight), one lane per neighbour folds sequentially and carries its partial sum across chunks.  Anything that does not fit
// the LDS structures (candidate heap, visited set, ef) sets fail[q]; the host re-runs that query with its own traversal.
#include "kernels.h"


#pragma clang fp contract(off)

namespace vdb {

namespace {
constexpr uint32_t HT = 256;                  // threads per workgroup
constexpr uint32_t CAND_CAP = 4096;
constexpr uint32_t RES_CAP = 1024;            // ef + 1 <= RES_CAP
constexpr uint32_t VIS_CAP = 16384;           // hash slots; at most 3/4 are used
// the rows of an expansion are staged in chunks of `chunk` elements (64..256, chosen on the host so that TWO chunk buffers fit
// the LDS that the walk's own structures leave: hnsw_stage_plan); a staged row takes chunk + 4 floats
constexpr uint32_t MAXP = 40;                 // neighbours per expansion at most (m_max0 + 1 <= 40, i.e. m <= 19)

struct HNb { float d; uint32_t id; };
typedef __attribute__((address_space(3))) void* hn_lds_t;
typedef const __attribute__((address_space(1))) void* hn_glb_t;

__device__ __forceinline__ int nb_cmp(const HNb& a, const HNb& b) {      // neighbor_queue.rs:37-43
    if (a.d < b.d) return -1;
    if (a.d > b.d) return 1;
    return a.id < b.id ? -1 : (a.id > b.id ? 1 : 0);
}
// Rust BinaryHeap (max-heap under SIGN * nb_cmp) on an LDS array; n is the length.
template <int SIGN> __device__ __forceinline__ bool h_le(const HNb& a, const HNb& b) { return SIGN * nb_cmp(a, b) <= 0; }
template <int SIGN> __device__ void h_sift_up(HNb* v, uint32_t start, uint32_t pos) {
    HNb e = v[pos];
    while (pos > start) {
        uint32_t parent = (pos - 1) / 2;
        HNb pv = v[parent];
        if (h_le<SIGN>(e, pv)) break;
        v[pos] = pv;
        pos = parent;
    }
    v[pos] = e;
}
template <int SIGN> __device__ void h_push(HNb* v, uint32_t& n, HNb x) { v[n] = x; h_sift_up<SIGN>(v, 0, n); ++n; }
template <int SIGN> __device__ HNb h_pop(HNb* v, uint32_t& n) {       // n > 0
    HNb item = v[--n];
    if (n) {
        HNb top = v[0];
        const uint32_t end = n;
        uint32_t pos = 0, child = 1;
        HNb e = item;
        item = top;
        // (tried: two levels per LDS round trip -- the two children and their four children read together; 5 % SLOWER for the
        // whole walk, like v_pk_* products in the fold: the lane is issue-bound, not latency-bound, once its reads are batched)
        while (end >= 2 && child <= end - 2) {                          // sift_down_to_bottom
            HNb c0 = v[child], c1 = v[child + 1];
            if (h_le<SIGN>(c0, c1)) { ++child; c0 = c1; }
            v[pos] = c0;
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { v[pos] = v[child]; pos = child; }
        v[pos] = e;
        h_sift_up<SIGN>(v, 0, pos);
    }
    return item;
}
// ---- the same heap operations, executed by a whole WAVE.  A sift is a chain of dependent LDS round trips for a single lane
// (one per level); here every lane takes one level: the ancestors of a position are known from index arithmetic, so a push reads
// them all at once, one ballot finds where the element stops, and the shifts are independent writes.  A pop first lets every lane
// decide one inner node ("which child is larger") -- 64 nodes per ballot --, follows those bits from the root in scalar registers,
// and then needs one read of the path and one ballot for Rust's sift_down_to_bottom + sift_up pair.  Same comparisons, same
// final array as the single-lane code above (the order of equal distances depends on it); lane = 0..63, all lanes call.
__device__ __forceinline__ uint32_t h_anc(uint32_t pos, uint32_t k) { return ((pos + 1u) >> k) - 1u; }   // k-th ancestor of pos
template <int SIGN> __device__ __forceinline__ void wave_sift_up(HNb* v, uint32_t pos, HNb e, uint32_t lane) {
    const uint32_t L = 31u - (uint32_t)__builtin_clz(pos + 1u);             // depth of pos: ancestors 1..L
    bool stop = false;
    HNb P{0.f, 0u};
    if (lane >= 1 && lane <= L) { P = v[h_anc(pos, lane)]; stop = h_le<SIGN>(e, P); }
    const unsigned long long m = __ballot(stop);
    const uint32_t K = m ? (uint32_t)__builtin_ctzll(m) : L + 1u;           // the first ancestor the element does not pass (L + 1: none)
    if (lane >= 1 && lane < K) v[h_anc(pos, lane - 1u)] = P;
    if (lane == 0) v[h_anc(pos, K - 1u)] = e;
}
template <int SIGN> __device__ __forceinline__ void wave_push(HNb* v, uint32_t& n, HNb x, uint32_t lane) { wave_sift_up<SIGN>(v, n, x, lane); ++n; }
constexpr uint32_t WAVE_POP_MAX = 514;                                        // heaps up to this length: 4 ballots of inner nodes
template <int SIGN> __device__ __forceinline__ void wave_pop_discard(HNb* v, uint32_t& n, uint32_t lane) {   // 0 < n <= WAVE_POP_MAX
    const HNb e = v[n - 1u];
    const uint32_t end = --n;
    if (end == 0) return;
    // inner nodes with two children: i <= (end - 3) / 2; bit = 1: the right child is taken (h_le(left, right))
    const uint32_t n2 = end >= 3u ? (end - 3u) / 2u + 1u : 0u;
    unsigned long long mk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = 64u * r + lane;
        bool b = false;
        if (i < n2) { const HNb c0 = v[2u * i + 1u], c1 = v[2u * i + 2u]; b = h_le<SIGN>(c0, c1); }
        mk[r] = (64u * r < n2) ? __ballot(b) : 0ull;
    }
    // the path of sift_down_to_bottom from the root; lane j remembers path[j]
    uint32_t pos = 0, child = 1, D = 0, mypos = 0;
    while (end >= 2u && child <= end - 2u) {
        const unsigned long long mm = pos < 64u ? mk[0] : pos < 128u ? mk[1] : pos < 192u ? mk[2] : mk[3];
        child += (uint32_t)((mm >> (pos & 63u)) & 1ull);
        ++D; pos = child;
        if (lane == D) mypos = pos;
        child = 2u * pos + 1u;
    }
    if (child == end - 1u) { ++D; pos = child; if (lane == D) mypos = pos; }
    // after the shifts position path[j] holds old[path[j + 1]] and the hole is at path[D]; sift_up from there stops at the deepest j
    // with e <= old[path[j]]: e lands on path[J], path[0 .. J-1] take old[path[1 .. J]], the rest of the path keeps its values
    HNb mine{0.f, 0u};
    bool le_ = false;
    if (lane >= 1 && lane <= D) { mine = v[mypos]; le_ = h_le<SIGN>(e, mine); }
    const unsigned long long m = __ballot(le_);
    const uint32_t J = m ? 63u - (uint32_t)__builtin_clzll(m) : 0u;
    const uint32_t prev = (uint32_t)__shfl_up((int)mypos, 1);
    if (lane >= 1 && lane <= J) v[prev] = mine;
    if (lane == J) v[mypos] = e;
}
__device__ __forceinline__ uint32_t vis_hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
}  // namespace

__global__ __launch_bounds__(HT) void hnsw_search_kernel(HnswSearchParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t ldq = (p.dim + 3) & ~3u;
    float* sQ = reinterpret_cast<float*>(smem);                                   // [ldq]
    HNb* sCand = reinterpret_cast<HNb*>(sQ + ldq);                                // [CAND_CAP]
    HNb* sRes = sCand + CAND_CAP;                                                 // [RES_CAP]
    uint32_t* sVis = reinterpret_cast<uint32_t*>(sRes + RES_CAP);                 // [VIS_CAP]
    float* sStage = reinterpret_cast<float*>(sVis + VIS_CAP);                     // [2][stage_rows][chunk + 4]
    __shared__ uint32_t sPendId[MAXP], sPendRow[MAXP];
    __shared__ float sPendD[MAXP];
    __shared__ uint32_t sNP, sCont, sCur, sFail, sNVis, sZero, sNCand, sNRes, sRec;
    // the final stable sort's keys (ordered distance, position in the heap's backing array) live in the staging area, which is
    // idle by then (>= 8 KB: hnsw_stage_plan)
    uint32_t* sKeyD = reinterpret_cast<uint32_t*>(sStage);
    uint32_t* sKeyI = sKeyD + RES_CAP;

    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the query: a prepared query row, or -- insert walks -- a stored row of the index
    const float* qsrc = p.qrow ? p.rows + (size_t)p.qrow[q] * p.ld : p.qp + (size_t)q * p.ld;
    for (uint32_t i = tid; i < ldq; i += HT) sQ[i] = qsrc[i];
    if (tid == 0) { sFail = 0; sZero = 0; sRec = 0; }
    const float qn = p.qrow ? p.nd[p.qrow[q]] : p.qnorm[q];
    const uint32_t ins_level = p.qlevel ? p.qlevel[q] : 0xffffffffu;    // 0xffffffff: a search (ef = 1 above layer 0)
    __syncthreads();

    // distances of sPendRow[0..np) into sPendD, in the reference's operation order.
    // The rows come through LDS in chunks, TWO chunks in flight: while lane r folds chunk c of row r (a strictly sequential chain
    // of adds -- the reference's fold order), the DMA of chunk c + 1 is under way and that of chunk c + 2 is issued as soon as the
    // fold has freed its buffer.  (One chunk at a time cost a full memory latency per chunk: 3 x 4.4 us of the 25 us an
    // expansion took at 768 dimensions.)  The DMA is issued from inline asm and waited for with counted s_waitcnt, as in the
    // filter kernels: hipcc would put a vmcnt(0) in front of every barrier behind the builtin.
    const uint32_t CHK = p.chunk, CHSK = p.chunk + 4;
    float* const sBuf0 = sStage;
    float* const sBuf1 = sStage + (size_t)p.stage_rows * CHSK;
    auto eval_pending = [&](uint32_t np) {
        float s = 0.0f;
        const uint32_t d = p.dim;
        const uint32_t nch = (d + CHK - 1) / CHK;
        const uint32_t L = wv < np ? (np - wv + 3) / 4 : 0;         // DMA instructions of this wave per chunk (np <= MAXP: at most 10)
        auto issue = [&](uint32_t c) {
            const uint32_t c0 = c * CHK;
            const uint32_t cl = d - c0 < CHK ? d - c0 : CHK;        // elements of this chunk
            const uint32_t nv = (cl + 3) / 4;                       // float4 per row (rows are zero padded up to ld)
            float* const dst = (c & 1u) ? sBuf1 : sBuf0;
            // one LDS-DMA instruction moves a row's whole chunk (up to 64 lanes x 16 B); every row of the expansion is in flight at once
            for (uint32_t r = wv; r < np; r += HT / 64) {
                const float* gp = p.rows + (size_t)sPendRow[r] * p.ld + c0 + 4 * lane;
                const uint32_t la = (uint32_t)(uintptr_t)(hn_lds_t)(dst + r * CHSK);
                if (lane < nv)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(la), "v"((const void*)gp) : "memory", "m0");
            }
        };
        auto wait_outstanding = [&](uint32_t n) {                   // s_waitcnt takes an immediate
            switch (n) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // nothing older (the previous expansion's record stores) is counted below
        issue(0);
        if (nch > 1) issue(1);
        for (uint32_t c = 0; c < nch; ++c) {
            const uint32_t c0 = c * CHK;
            const uint32_t cl = d - c0 < CHK ? d - c0 : CHK;
            // this wave's pieces of chunk c have landed once only those of chunk c + 1 are outstanding; the barrier says so of every wave
            wait_outstanding(c + 1 < nch ? L : 0u);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tid < np) {
                const float* x = ((c & 1u) ? sBuf1 : sBuf0) + tid * CHSK;
                const float* a = sQ + c0;
                // The adds are one strictly sequential chain (the reference's fold order); everything else is kept off it: 16 elements
                // of both operands are read (eight 16-byte LDS reads in flight together) while the previous 16 are folded -- one
                // block at a time the lane waited a full LDS latency per four elements, 13 us per expansion at 768 dimensions.
                const uint32_t c16 = cl & ~15u;
                float4 av[4], xv[4], an[4], xn[4];
                if (c16) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { av[u] = *reinterpret_cast<const float4*>(a + 4 * u); xv[u] = *reinterpret_cast<const float4*>(x + 4 * u); }
                }
                for (uint32_t i = 0; i < c16; i += 16) {
                    if (i + 16 < c16) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { an[u] = *reinterpret_cast<const float4*>(a + i + 16 + 4 * u); xn[u] = *reinterpret_cast<const float4*>(x + i + 16 + 4 * u); }
                    }
                    if (p.metric == EUCLID) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float t0 = __fsub_rn(av[u].x, xv[u].x), t1 = __fsub_rn(av[u].y, xv[u].y), t2 = __fsub_rn(av[u].z, xv[u].z), t3 = __fsub_rn(av[u].w, xv[u].w);
                            const float p0 = __fmul_rn(t0, t0), p1 = __fmul_rn(t1, t1), p2 = __fmul_rn(t2, t2), p3 = __fmul_rn(t3, t3);
                            s = __fadd_rn(s, p0); s = __fadd_rn(s, p1); s = __fadd_rn(s, p2); s = __fadd_rn(s, p3);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float p0 = __fmul_rn(av[u].x, xv[u].x), p1 = __fmul_rn(av[u].y, xv[u].y), p2 = __fmul_rn(av[u].z, xv[u].z), p3 = __fmul_rn(av[u].w, xv[u].w);
                            s = __fadd_rn(s, p0); s = __fadd_rn(s, p1); s = __fadd_rn(s, p2); s = __fadd_rn(s, p3);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { av[u] = an[u]; xv[u] = xn[u]; }
                }
                if (p.metric == EUCLID) { for (uint32_t i = c16; i < cl; ++i) { float t = __fsub_rn(a[i], x[i]); s = __fadd_rn(s, __fmul_rn(t, t)); } }
                else { for (uint32_t i = c16; i < cl; ++i) s = __fadd_rn(s, __fmul_rn(a[i], x[i])); }
            }
            // the fold's reads of this buffer are done (their values are consumed): after the barrier it may be refilled
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (c + 2 < nch) issue(c + 2);
        }
        if (tid < np) {
            float dist;
            if (p.metric == EUCLID) dist = __builtin_sqrtf(s);
            else if (p.metric == DOT) dist = -s;
            else {
                const float xn = p.nd[sPendRow[tid]];
                if (qn == 0.0f || xn == 0.0f) { sZero = 1u; dist = p.rec_row ? __uint_as_float(p.rec_zero_mark) : 0.0f; }
                else {
                    float sim = __fdiv_rn(s, __fmul_rn(qn, xn));
                    if (sim < -1.0f) sim = -1.0f;
                    if (sim > 1.0f) sim = 1.0f;
                    dist = __fsub_rn(1.0f, sim);
                }
            }
            sPendD[tid] = dist;
        }
        __syncthreads();
        if (p.rec_row) {                                            // insert walks: every evaluated (row, distance), in order
            const uint32_t base = sRec;
            if (tid < np && base + tid < p.rec_cap) {
                p.rec_row[(size_t)q * p.rec_cap + base + tid] = sPendId[tid];      // the NODE ID: the host replay looks distances up by id
                p.rec_d[(size_t)q * p.rec_cap + base + tid] = sPendD[tid];
            }
            __syncthreads();
            if (tid == 0) sRec = base + np;
            __syncthreads();
        }
    };

    uint32_t ep = p.entry_point;
    const uint32_t ef_final = p.ef > p.k ? p.ef : p.k;
    for (int layer = (int)p.max_level; layer >= 0; --layer) {
        // search: ef = 1 above layer 0 (graph.rs:400-405); insert of a node of level L: ef = 1 above L, ef_construction at and below
        const uint32_t ef = p.qlevel ? ((uint32_t)layer > ins_level ? 1u : ef_final) : (layer >= 1 ? 1u : ef_final);
        // ---- search_layer(query, [ep], ef, layer)   (graph.rs:143-199)
        for (uint32_t i = tid; i < VIS_CAP; i += HT) sVis[i] = 0xffffffffu;
        if (tid == 0) {
            sNCand = 0; sNRes = 0; sNVis = 1;
            sPendId[0] = ep; sPendRow[0] = p.row_of[ep];
        }
        __syncthreads();
        if (tid == 0) sVis[vis_hash(ep) & (VIS_CAP - 1)] = ep;    // visited.insert(ep), after the clear has completed
        __syncthreads();
        eval_pending(1);
        if (tid == 0) {
            uint32_t nc = 0, nr = 0;
            HNb x{sPendD[0], ep};
            h_push<-1>(sCand, nc, x);
            h_push<+1>(sRes, nr, x);
            sNCand = nc; sNRes = nr;
        }
        __syncthreads();
        while (true) {
            // the candidate about to be popped is the top of the heap: it is LOOKED AT first, so that the pop itself (a single lane's
            // chain through the levels of a heap of a thousand entries) runs in wave 0 beside wave 1's read of that candidate's
            // neighbour list -- they touch different structures.  (graph.rs:170-176 pops, then compares with the furthest result
            // and returns: when the walk ends here the heap is not looked at again, so the pop is left out.)
            if (tid == 0) {
                uint32_t cont = 0;
                if (sNCand && !sFail && !sZero) {
                    const HNb c = sCand[0];
                    const float furthest = sNRes ? sRes[0].d : 3.40282347e+38f;
                    if (!(c.d > furthest)) { cont = 1; sCur = c.id; }
                }
                sCont = cont;
            }
            __syncthreads();
            if (!sCont) break;
            if (tid == 0) { uint32_t nc = sNCand; (void)h_pop<-1>(sCand, nc); sNCand = nc; }
            // ---- neighbours of sCur at this layer: visited filter, in list order
            const uint32_t cur = sCur;
            if (wv == 1) {
                // the whole list in ONE load per lane: lists are padded with 0xffffffff and carry the device row of each
                // neighbour (0xffffffff = deleted) beside its id, so no count / level / row lookup precedes the distances
                const uint32_t* lst; const uint32_t* lrow; uint32_t stride;
                if (layer == 0) { lst = p.nbr0 + (size_t)cur * p.stride0; lrow = p.nbr0_row + (size_t)cur * p.stride0; stride = p.stride0; }
                else {
                    const uint32_t li = p.up_off[cur] + (uint32_t)(layer - 1);
                    lst = p.nbrU + (size_t)li * p.strideU; lrow = p.nbrU_row + (size_t)li * p.strideU; stride = p.strideU;
                }
                if (stride > MAXP) { stride = MAXP; if (lane == 0) sFail = 1u; }
                bool keep = false;
                uint32_t nid = 0xffffffffu, row = 0xffffffffu;
                if (lane < stride) { nid = lst[lane]; row = lrow[lane]; }
                if (nid != 0xffffffffu) {
                    // visited.insert(nid): open addressing, CAS claims a slot; an equal key found = already visited
                    uint32_t h = vis_hash(nid) & (VIS_CAP - 1);
                    bool fresh = false;
                    for (uint32_t probe = 0; probe < VIS_CAP; ++probe) {
                        const uint32_t old = atomicCAS(&sVis[h], 0xffffffffu, nid);
                        if (old == 0xffffffffu) { fresh = true; break; }
                        if (old == nid) break;
                        h = (h + 1) & (VIS_CAP - 1);
                    }
                    if (fresh) {
                        atomicAdd(&sNVis, 1u);
                        keep = row != 0xffffffffu;                       // skip deleted nodes
                    }
                }
                const unsigned long long m = __ballot(keep);
                const uint32_t pos = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (keep) { sPendId[pos] = nid; sPendRow[pos] = row; }
                if (lane == 0) {
                    sNP = (uint32_t)__popcll(m);
                    if (sNVis > (VIS_CAP / 4) * 3) sFail = 1u;
                }
            }
            __syncthreads();
            const uint32_t np = sNP;
            if (np == 0) continue;
            eval_pending(np);
            if (wv == 0) {                                                   // one wave folds the new distances into the two heaps
                uint32_t nc = (uint32_t)__builtin_amdgcn_readfirstlane((int)sNCand), nr = (uint32_t)__builtin_amdgcn_readfirstlane((int)sNRes);
                // lane i holds neighbour i (np <= 40): the loop below reads them with v_readlane and keeps the furthest result in a
                // register -- two dependent LDS reads per neighbour, accepted or not, were half of this phase
                const uint32_t my_d = lane < np ? __float_as_uint(sPendD[lane]) : 0u, my_id = lane < np ? sPendId[lane] : 0u;
                float furthest = nr ? __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sRes[0].d))) : 3.40282347e+38f;
                for (uint32_t i = 0; i < np; ++i) {
                    const float dd = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)my_d, (int)i));
                    if (dd < furthest || nr < ef) {
                        if (nc >= CAND_CAP || nr >= RES_CAP) { if (lane == 0) sFail = 1u; break; }
                        const HNb x{dd, (uint32_t)__builtin_amdgcn_readlane((int)my_id, (int)i)};
                        wave_push<-1>(sCand, nc, x, lane);
                        wave_push<+1>(sRes, nr, x, lane);
                        if (nr > ef) {
                            if (nr <= WAVE_POP_MAX) wave_pop_discard<+1>(sRes, nr, lane);
                            else { if (lane == 0) (void)h_pop<+1>(sRes, nr); else --nr; }
                        }
                        furthest = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sRes[0].d)));
                    }
                }
                if (lane == 0) { sNCand = nc; sNRes = nr; }
            }
            __syncthreads();
        }
        if (sFail || sZero) break;
        // ---- results.into_sorted_vec(): stable sort of the heap's backing array by distance (bitonic on (distance, position))
        const uint32_t nr = sNRes;
        uint32_t P = 2;
        while (P < nr) P <<= 1;
        for (uint32_t i = tid; i < P; i += HT) {
            sKeyD[i] = i < nr ? f32_to_ordered(sRes[i].d) : 0xffffffffu;
            sKeyI[i] = i;
        }
        __syncthreads();
        for (uint32_t size = 2; size <= P; size <<= 1)
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t t = tid; t < P / 2; t += HT) {
                    uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                    bool up = ((lo & size) == 0);
                    uint32_t da = sKeyD[lo], db = sKeyD[hi], ia = sKeyI[lo], ib = sKeyI[hi];
                    bool gt = da > db || (da == db && ia > ib);
                    if (gt == up) { sKeyD[lo] = db; sKeyD[hi] = da; sKeyI[lo] = ib; sKeyI[hi] = ia; }
                }
                __syncthreads();
            }
        if (layer >= 1) {
            if (nr) ep = sRes[sKeyI[0]].id;                                  // nearest.first()
            __syncthreads();
        } else {
            const uint32_t cnt = nr < p.k ? nr : p.k;
            for (uint32_t i = tid; i < cnt; i += HT) {
                const HNb x = sRes[sKeyI[i]];
                p.out_ids[(size_t)q * p.k + i] = (uint64_t)x.id;
                p.out_dists[(size_t)q * p.k + i] = x.d;
            }
            if (tid == 0) p.out_counts[q] = cnt;
        }
    }
    if (tid == 0) {
        p.fail[q] = sFail ? 1u : 0u;
        if (p.rec_cnt) p.rec_cnt[q] = sRec;
        if (sZero && !p.rec_row) atomicOr(p.status, ST_ZERO_QUERY);
    }
}

// incremental mirror update: one 64-lane workgroup per record
__global__ __launch_bounds__(64) void hnsw_scatter_kernel(HnswScatterParams p) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b < p.n0) {
        const uint32_t* r = p.rec0 + (size_t)b * (4 + 2 * p.stride0);
        const uint32_t id = r[0];
        if (lane == 0) { p.row_of[id] = r[1]; p.level[id] = r[2]; p.up_off[id] = r[3]; }
        for (uint32_t i = lane; i < p.stride0; i += 64) {
            p.nbr0[(size_t)id * p.stride0 + i] = r[4 + i];
            p.nbr0_row[(size_t)id * p.stride0 + i] = r[4 + p.stride0 + i];
        }
    } else {
        const uint32_t* r = p.recU + (size_t)(b - p.n0) * (1 + 2 * p.strideU);
        const uint32_t li = r[0];
        for (uint32_t i = lane; i < p.strideU; i += 64) {
            p.nbrU[(size_t)li * p.strideU + i] = r[1 + i];
            p.nbrU_row[(size_t)li * p.strideU + i] = r[1 + p.strideU + i];
        }
    }
}

// Two staging buffers of stage_rows x (chunk + 4) floats in what the walk's own structures leave of the 160 KB: the largest chunk
// of 256 / 192 / 128 / 64 elements that fits (m = 16 at 768 dimensions: 33 rows x 192), not larger than the rows; 0 = does not fit.
static uint32_t hnsw_stage_plan(uint32_t dim, uint32_t stage_rows, size_t* bytes) {
    const uint32_t ldq = (dim + 3) & ~3u;
    const size_t fixed = (size_t)ldq * 4 + (size_t)CAND_CAP * 8 + (size_t)RES_CAP * 8 + (size_t)VIS_CAP * 4;
    const size_t limit = 160 * 1024 - 1024;                              // (static: the pending lists and a dozen words)
    if (stage_rows == 0) stage_rows = 1;
    if (fixed >= limit) return 0;
    const uint32_t cap = dim <= 64 ? 64u : dim <= 128 ? 128u : dim <= 192 ? 192u : 256u;
    for (uint32_t chunk : {256u, 192u, 128u, 64u}) {
        if (chunk > cap) continue;
        size_t st = (size_t)2 * stage_rows * (chunk + 4) * 4;
        if (st < (size_t)2 * RES_CAP * 4) st = (size_t)2 * RES_CAP * 4;  // the final sort's keys live there
        if (fixed + st <= limit) { if (bytes) *bytes = fixed + st; return chunk; }
    }
    return 0;
}
bool hnsw_search_supported(uint32_t dim, uint32_t ef, uint32_t k, uint32_t max_list) {
    return (ef > k ? ef : k) + 1 <= RES_CAP && max_list <= MAXP && hnsw_stage_plan(dim, max_list, nullptr) != 0;
}
void launch_hnsw_scatter(const HnswScatterParams& p, hipStream_t s) {
    if (!(p.n0 + p.nU)) return;
    hipLaunchKernelGGL(hnsw_scatter_kernel, dim3(p.n0 + p.nU), dim3(64), 0, s, p);
}
void launch_hnsw_search(const HnswSearchParams& p_in, uint32_t nq, hipStream_t s) {
    if (!nq) return;
    HnswSearchParams p = p_in;
    p.stage_rows = std::max(p.stride0, p.strideU);
    if (p.stage_rows > MAXP) p.stage_rows = MAXP;                        // (the kernel fails such a walk: hnsw_search_supported said no)
    size_t bytes = 0;
    p.chunk = hnsw_stage_plan(p.dim, p.stage_rows, &bytes);
    if (!p.chunk) return;                                                // unsupported shape: the caller checked hnsw_search_supported
    hipLaunchKernelGGL(hnsw_search_kernel, dim3(nq), dim3(HT), bytes, s, p);
}

}  // namespace vdb
