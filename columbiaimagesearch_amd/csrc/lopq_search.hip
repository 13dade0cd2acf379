// LOPQ index + batched search on MI355X.
//
// Replaces lopq/lopq/search.py: multisequence :13-82, get_result_quota :110-135,
// compute_distances (ADC) :137-177, search :179-224, LOPQSearcher (dict index) :310-382 -- for a
// whole batch of queries per call.
//
// Data layout in HBM
//   codes  [N][M] uint8   fine codes, cell-contiguous (CSR over the V*V coarse cells), inside a
//                         cell in insertion order (= the reference's per-cell list order)
//   ids    [N]    int64   caller ids, same order
//   loff   [V*V+1] int64  CSR offsets of the cells stored on THIS shard
//   gcount [V*V]  int64   item count of every cell over ALL shards (drives the quota cut-off)
//
// Pipeline per query batch (all on one stream):
//   PCA -> coarse distances (numpy order) -> per-split rank -> multisequence plan (count) ->
//   exclusive scan -> plan (emit work items + table list) -> ADC tables -> ADC scan + block
//   top-k -> per-query merge -> ids/dists.
#include <algorithm>
#include <unordered_set>

#include "lopq_model.h"

// ================================================================================================
// device structures
// ================================================================================================
struct WorkItem {
    int q;          // query index inside the batch
    int rank;       // multisequence visit rank of the cell
    int tab0, tab1; // indices of the two half tables
    int64_t start;  // first candidate (position in codes/ids)
    int len;        // candidates in this chunk
    int pos0;       // insertion position of the first candidate inside its cell
    int cell;       // c0 * V + c1
    int pad;
};

struct TabDesc {
    int q, split, cluster, pad;
};

struct PlanOut {  // per query
    int visited, n_items, ntab0, ntab1;
    int64_t ncand;
};

static __device__ __forceinline__ uint64_t f2bits(double d) { return (uint64_t)__double_as_longlong(d); }
static __device__ __forceinline__ uint64_t f2bits(float f) { return (uint64_t)__float_as_uint(f); }

// ================================================================================================
// kernels: coarse ranking and multisequence plan
// ================================================================================================
// Ascending order of the V coarse distances of one (query, split).  Distances are >= 0 so their
// bit patterns order like the values (NaN sorts last, as np.argsort does).  Ties -> lower index.
template <typename CT>
__global__ void k_rank(const CT* __restrict__ dist /* [2][nq][V] */, int nq, int V,
                       uint16_t* __restrict__ order /* [nq][2][V] */, CT* __restrict__ sorted /* [nq][2][V] */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* sb = reinterpret_cast<uint64_t*>(smem);
    const int q = blockIdx.x, s = blockIdx.y;
    const CT* d = dist + ((int64_t)s * nq + q) * V;
    for (int v = threadIdx.x; v < V; v += blockDim.x) sb[v] = f2bits(d[v]);
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const uint64_t mine = sb[v];
        int r = 0;
        for (int u = 0; u < V; ++u) {
            const uint64_t o = sb[u];
            r += (o < mine) || (o == mine && u < v);
        }
        order[((int64_t)q * 2 + s) * V + r] = (uint16_t)v;
        sorted[((int64_t)q * 2 + s) * V + r] = d[v];
    }
}

// One wave per query walks the multi-index exactly like lopq/lopq/search.py:58-82.  With two
// splits the traversed set is a Young diagram: t[i] cells taken in rank-row i; the reference's heap
// holds (i, t[i]) for rows with t[i] < V and (i == 0 or t[i-1] > t[i]) and pops the smallest
// (dist, i, j) with dist = d0[i] + d1[j] rounded in the coarse compute type.
template <typename CT, bool EMIT>
__global__ __launch_bounds__(64) void k_plan(const CT* __restrict__ sorted, const uint16_t* __restrict__ order,
                                             const int64_t* __restrict__ gcount, const int64_t* __restrict__ loff,
                                             int nq, int V, int64_t quota, int seg_max, PlanOut* __restrict__ plan,
                                             const int64_t* __restrict__ item_off, const int64_t* __restrict__ tab_off,
                                             WorkItem* __restrict__ items, TabDesc* __restrict__ tabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* t = reinterpret_cast<int*>(smem);  // [V]
    const int q = blockIdx.x;
    const int lane = threadIdx.x;
    const CT* d0 = sorted + ((int64_t)q * 2 + 0) * V;
    const CT* d1 = sorted + ((int64_t)q * 2 + 1) * V;
    const uint16_t* o0 = order + ((int64_t)q * 2 + 0) * V;
    const uint16_t* o1 = order + ((int64_t)q * 2 + 1) * V;
    for (int i = lane; i < V; i += 64) t[i] = 0;
    __syncthreads();
    int visited = 0, n_items = 0, max_i = -1, max_j = -1;
    int64_t retrieved = 0, ncand = 0;
    int rows = 1;  // rows [0, rows) can be on the frontier
    int64_t ibase = 0, tbase = 0;
    int ntab0 = 0;
    if (EMIT) {
        ibase = item_off[q];
        tbase = tab_off[q];
        ntab0 = plan[q].ntab0;
    }
    const int64_t total_cells = (int64_t)V * V;
    while ((int64_t)visited < total_cells) {
        // frontier minimum over rows, key = (dist bits, i, j)
        uint64_t bk = ~0ull;
        uint32_t bij = ~0u;
        for (int i = lane; i < rows; i += 64) {
            const int j = t[i];
            if (j >= V) continue;
            if (i > 0 && t[i - 1] <= j) continue;
            const CT dist = d0[i] + d1[j];
            const uint64_t kb = f2bits(dist);
            const uint32_t ij = ((uint32_t)i << 16) | (uint32_t)j;
            if (kb < bk || (kb == bk && ij < bij)) { bk = kb; bij = ij; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t ok = __shfl_xor(bk, off);
            const uint32_t oij = __shfl_xor(bij, off);
            if (ok < bk || (ok == bk && oij < bij)) { bk = ok; bij = oij; }
        }
        if (bij == ~0u) break;  // cannot happen before all cells are visited
        const int bi = (int)(bij >> 16), bj = (int)(bij & 0xffff);
        const int c0 = o0[bi], c1 = o1[bj];
        const int64_t cell = (int64_t)c0 * V + c1;
        const int64_t gc = gcount[cell];
        const int64_t ls = loff[cell];
        const int64_t ll = loff[cell + 1] - ls;
        if (ll > 0) {
            const int nch = (int)((ll + seg_max - 1) / seg_max);
            if (EMIT) {
                for (int ch = lane; ch < nch; ch += 64) {
                    WorkItem it;
                    it.q = q; it.rank = visited;
                    it.tab0 = (int)(tbase + bi);
                    it.tab1 = (int)(tbase + ntab0 + bj);
                    it.pos0 = ch * seg_max;
                    it.cell = (int)cell; it.pad = 0;
                    it.start = ls + (int64_t)ch * seg_max;
                    const int64_t rem = ll - (int64_t)ch * seg_max;
                    it.len = (int)(rem < seg_max ? rem : seg_max);
                    items[ibase + n_items + ch] = it;
                }
            }
            n_items += nch;
            ncand += ll;
            max_i = bi > max_i ? bi : max_i;
            max_j = bj > max_j ? bj : max_j;
        }
        visited += 1;
        retrieved += gc;
        __syncthreads();
        if (lane == 0) t[bi] = bj + 1;
        if (bi + 2 > rows) rows = (bi + 2 < V) ? bi + 2 : V;
        __syncthreads();
        if (retrieved >= quota) break;
    }
    if (!EMIT) {
        if (lane == 0) {
            PlanOut p;
            p.visited = visited; p.n_items = n_items; p.ntab0 = max_i + 1; p.ntab1 = max_j + 1; p.ncand = ncand;
            plan[q] = p;
        }
    } else {
        const int nt0 = plan[q].ntab0, nt1 = plan[q].ntab1;
        for (int i = lane; i < nt0 + nt1; i += 64) {
            TabDesc td;
            td.q = q; td.pad = 0;
            if (i < nt0) { td.split = 0; td.cluster = o0[i]; }
            else { td.split = 1; td.cluster = o1[i - nt0]; }
            tabs[tbase + i] = td;
        }
    }
}

// exclusive scans over the queries of one batch (single block); totals[0]=items, [1]=tables, [2]=cands
__global__ void k_plan_scan(const PlanOut* __restrict__ plan, int nq, int64_t* __restrict__ item_off,
                            int64_t* __restrict__ tab_off, int64_t* __restrict__ totals) {
    __shared__ int64_t s_items[256], s_tabs[256], s_cand[256];
    const int tid = threadIdx.x;
    const int per = (nq + 255) / 256;
    const int a = tid * per, b = (a + per < nq) ? a + per : nq;
    int64_t li = 0, lt = 0, lc = 0;
    for (int q = a; q < b; ++q) { li += plan[q].n_items; lt += plan[q].ntab0 + plan[q].ntab1; lc += plan[q].ncand; }
    s_items[tid] = li; s_tabs[tid] = lt; s_cand[tid] = lc;
    __syncthreads();
    if (tid == 0) {
        int64_t ri = 0, rt = 0, rc = 0;
        for (int k = 0; k < 256; ++k) {
            const int64_t xi = s_items[k], xt = s_tabs[k];
            s_items[k] = ri; s_tabs[k] = rt;
            ri += xi; rt += xt; rc += s_cand[k];
        }
        totals[0] = ri; totals[1] = rt; totals[2] = rc;
        item_off[nq] = ri; tab_off[nq] = rt;
    }
    __syncthreads();
    int64_t ri = s_items[tid], rt = s_tabs[tid];
    for (int q = a; q < b; ++q) {
        item_off[q] = ri; tab_off[q] = rt;
        ri += plan[q].n_items; rt += plan[q].ntab0 + plan[q].ntab1;
    }
}

// ================================================================================================
// kernel: ADC tables  (lopq/lopq/model.py:673-704 for one (query, split, coarse id))
// ================================================================================================
template <typename CT>
__global__ __launch_bounds__(256) void k_tables(const CT* __restrict__ X /* [nq][D] */, const CT* __restrict__ Cs,
                                                const double* __restrict__ Rt, const double* __restrict__ mus,
                                                const double* __restrict__ subs, const TabDesc* __restrict__ tabs,
                                                int V, int h, int w, int nf, int K, int D,
                                                double* __restrict__ T /* [ntab][nf][K] */, PwProg prog_w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* v = reinterpret_cast<double*>(smem);  // [h]
    double* px = v + h;                           // [h]
    const TabDesc td = tabs[blockIdx.x];
    const int s = td.split, c = td.cluster;
    const CT* x = X + (int64_t)td.q * D + s * h;
    const CT* Cc = Cs + ((int64_t)s * V + c) * h;
    const double* mu = mus + ((int64_t)s * V + c) * h;
    const double* R = Rt + ((int64_t)s * V + c) * h * h;
    for (int k = threadIdx.x; k < h; k += blockDim.x) {
        const CT res = x[k] - Cc[k];  // rounds in CT (float32 when both are float32), model.py:635
        v[k] = (double)res - mu[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h; i += blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < h; ++k) acc = fma(R[(int64_t)k * h + i], v[k], acc);
        px[i] = acc;
    }
    __syncthreads();
    double* out = T + (int64_t)blockIdx.x * nf * K;
    for (int e = threadIdx.x; e < nf * K; e += blockDim.x) {
        const int j = e / K, k = e % K;
        const double* sc = subs + ((int64_t)(s * nf + j) * K + k) * w;
        const double* f = px + j * w;
        auto elem = [&](int i) -> double { const double df = f[i] - sc[i]; return df * df; };
        out[e] = pw_sum<double>(prog_w, elem);
    }
}

// ================================================================================================
// block-wide bitonic sort of N (power of two) keys (a, b) with an optional payload, in LDS
// ================================================================================================
template <int N, int NT, bool PAY>
__device__ __forceinline__ void block_bitonic(uint64_t* ka, uint64_t* kb, int64_t* pay) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < N / 2; t += NT) {
                const int i = ((t / j) * 2 * j) + (t % j);
                const int p = i + j;
                const bool asc = ((i & k) == 0);
                const uint64_t a0 = ka[i], b0 = kb[i], a1 = ka[p], b1 = kb[p];
                const bool gt = (a0 > a1) || (a0 == a1 && b0 > b1);
                if (gt == asc) {
                    ka[i] = a1; kb[i] = b1; ka[p] = a0; kb[p] = b0;
                    if (PAY) { const int64_t x = pay[i]; pay[i] = pay[p]; pay[p] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// ================================================================================================
// kernel: ADC scan + block top-k   (lopq/lopq/search.py:166-175, :210-215)
// ================================================================================================
// One 256-thread block per work item (a chunk of one cell for one query).  The two half tables
// sit in LDS as float64 [M][K]; every candidate's distance is ((T0[f0] + T1[f1]) + ...) in
// float64, left to right, exactly the reference's sum().  Candidates whose key (dist, pos) beats
// the running limit-th best are appended to an LDS buffer; when the buffer could overflow it is
// sorted and cut back to `limit` entries.
template <int M>
__device__ __forceinline__ double adc_one(const uint8_t* __restrict__ codes, int64_t p, const double* __restrict__ T, int K) {
    double d;
    if constexpr (M == 4) {
        const uint32_t c = *reinterpret_cast<const uint32_t*>(codes + p * 4);
        d = T[c & 255];
        d = d + T[K + ((c >> 8) & 255)];
        d = d + T[2 * K + ((c >> 16) & 255)];
        d = d + T[3 * K + (c >> 24)];
    } else if constexpr (M == 8) {
        const uint2 c = *reinterpret_cast<const uint2*>(codes + p * 8);
        d = T[c.x & 255];
        d = d + T[K + ((c.x >> 8) & 255)];
        d = d + T[2 * K + ((c.x >> 16) & 255)];
        d = d + T[3 * K + (c.x >> 24)];
        d = d + T[4 * K + (c.y & 255)];
        d = d + T[5 * K + ((c.y >> 8) & 255)];
        d = d + T[6 * K + ((c.y >> 16) & 255)];
        d = d + T[7 * K + (c.y >> 24)];
    } else if constexpr (M == 16) {
        const uint4 c = *reinterpret_cast<const uint4*>(codes + p * 16);
        const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
        d = T[cw[0] & 255];
        d = d + T[K + ((cw[0] >> 8) & 255)];
        d = d + T[2 * K + ((cw[0] >> 16) & 255)];
        d = d + T[3 * K + (cw[0] >> 24)];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            d = d + T[(4 * q + 0) * K + (cw[q] & 255)];
            d = d + T[(4 * q + 1) * K + ((cw[q] >> 8) & 255)];
            d = d + T[(4 * q + 2) * K + ((cw[q] >> 16) & 255)];
            d = d + T[(4 * q + 3) * K + (cw[q] >> 24)];
        }
    } else {  // generic: M passed at run time through K's sibling argument (see caller)
        d = 0.0;
    }
    return d;
}

static __device__ __forceinline__ double adc_generic(const uint8_t* __restrict__ codes, int64_t p, int M,
                                                     const double* __restrict__ T, int K) {
    const uint8_t* c = codes + p * M;
    double d = T[c[0]];
    for (int j = 1; j < M; ++j) d = d + T[j * K + c[j]];
    return d;
}

template <int M /* 0 = generic */, int CAP, int U>
__global__ __launch_bounds__(256) void k_adc_scan(const WorkItem* __restrict__ items, const double* __restrict__ T,
                                                  const uint8_t* __restrict__ codes, const int64_t* __restrict__ ids,
                                                  int Mrt, int K, int limit, cis_hit* __restrict__ item_hits,
                                                  int* __restrict__ item_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* ka = reinterpret_cast<uint64_t*>(smem);  // [CAP] dist bits
    uint64_t* kb = ka + CAP;                           // [CAP] position inside the chunk
    double* tab = reinterpret_cast<double*>(kb + CAP); // [M][K]
    int& s_cnt = *reinterpret_cast<int*>(tab + Mrt * K);  // all LDS in the dynamic region (16-B aligned carve)
    const int tid = threadIdx.x;
    const WorkItem it = items[blockIdx.x];
    const int nf = Mrt / 2;
    {
        const double* t0 = T + (int64_t)it.tab0 * nf * K;
        const double* t1 = T + (int64_t)it.tab1 * nf * K;
        for (int e = tid; e < nf * K; e += 256) {
            tab[e] = t0[e];
            tab[nf * K + e] = t1[e];
        }
    }
    if (tid == 0) s_cnt = 0;
    uint64_t tau_a = ~0ull, tau_b = ~0ull;  // running limit-th best key (everything passes at first)
    __syncthreads();
    const int len = it.len;
    for (int base = 0; base < len; base += 256 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = base + u * 256 + tid;
            if (p < len) {
                double d;
                if constexpr (M == 0) d = adc_generic(codes, it.start + p, Mrt, tab, K);
                else d = adc_one<M>(codes, it.start + p, tab, K);
                const uint64_t a = f2bits(d);
                if (a < tau_a || (a == tau_a && (uint64_t)p < tau_b)) {
                    const int slot = atomicAdd(&s_cnt, 1);
                    ka[slot] = a;  // slot < CAP is guaranteed by the compaction rule below
                    kb[slot] = (uint64_t)p;
                }
            }
        }
        __syncthreads();
        const int cnt = s_cnt;
        if (cnt > CAP - 256 * U && base + 256 * U < len) {
            for (int e = cnt + tid; e < CAP; e += 256) { ka[e] = ~0ull; kb[e] = ~0ull; }
            __syncthreads();
            block_bitonic<CAP, 256, false>(ka, kb, nullptr);
            tau_a = ka[limit - 1];
            tau_b = kb[limit - 1];
            __syncthreads();
            if (tid == 0) s_cnt = limit;
            __syncthreads();
        }
    }
    int cnt = s_cnt;
    if (cnt > limit) {
        for (int e = cnt + tid; e < CAP; e += 256) { ka[e] = ~0ull; kb[e] = ~0ull; }
        __syncthreads();
        block_bitonic<CAP, 256, false>(ka, kb, nullptr);
        cnt = limit;
    }
    cis_hit* out = item_hits + (int64_t)blockIdx.x * limit;
    for (int e = tid; e < cnt; e += 256) {
        cis_hit hh;
        hh.dist = __longlong_as_double((long long)ka[e]);
        hh.visit_rank = (uint32_t)it.rank;
        hh.pos = (uint32_t)(it.pos0 + (int)kb[e]);
        hh.id = ids[it.start + (int64_t)kb[e]];
        hh.cell = it.cell; hh.reserved = 0;
        out[e] = hh;
    }
    if (tid == 0) item_n[blockIdx.x] = cnt;
}

// ================================================================================================
// kernel: per-query merge of ranked lists -> top `limit` by (dist, visit_rank, pos)
// ================================================================================================
// Lists of query q: entries src[lo .. hi) in groups: list l has `stride` slots of which cnt[l]
// are valid (cnt == nullptr: a slot is valid when id >= 0).  Used twice: (a) merging the work
// items of a query, (b) merging the per-shard partial results after the all-gather.
template <int CAPM>
__device__ void merge_lists(const cis_hit* __restrict__ src, const int* __restrict__ cnt, int64_t first_list,
                            int n_lists, int64_t list_stride /* distance between lists, in hits */,
                            int slots, int limit, uint64_t* ka, uint64_t* kb, int64_t* pay, int* s_n,
                            cis_hit* __restrict__ out_hits /* [limit] or null */, int64_t* __restrict__ out_ids,
                            double* __restrict__ out_dists, int* __restrict__ out_n, int32_t* __restrict__ out_cells,
                            uint32_t* __restrict__ out_pos) {
    const int tid = threadIdx.x;
    int have = 0;      // sorted survivors currently in [0, have)
    int l = 0, e = 0;  // cursor: list l, entry e (uniform over the block)
    // rounds: append up to CAPM - have entries, sort, keep `limit`.  pay = index of the hit in src.
    while (true) {
        int n = have;
        int room = CAPM - have;
        while (l < n_lists && room > 0) {
            const int64_t lbase = (first_list + l) * list_stride;
            const int valid = cnt ? cnt[first_list + l] : slots;
            const int take = (valid - e < room) ? (valid - e) : room;
            for (int x = tid; x < take; x += blockDim.x) {
                const cis_hit hh = src[lbase + e + x];
                const bool ok = hh.id >= 0;
                ka[n + x] = ok ? (uint64_t)__double_as_longlong(hh.dist) : ~0ull;
                kb[n + x] = ok ? (((uint64_t)hh.visit_rank << 32) | hh.pos) : ~0ull;
                pay[n + x] = ok ? (lbase + e + x) : -1;
            }
            n += take;
            room -= take;
            e += take;
            if (e >= valid) { ++l; e = 0; }
        }
        for (int x = n + tid; x < CAPM; x += blockDim.x) { ka[x] = ~0ull; kb[x] = ~0ull; pay[x] = -1; }
        __syncthreads();
        block_bitonic<CAPM, 256, true>(ka, kb, pay);
        have = n < limit ? n : limit;
        if (l >= n_lists) break;
    }
    // empty slots (id < 0) carry all-ones keys and therefore sit behind every real hit
    if (tid == 0) *s_n = 0;
    __syncthreads();
    int local = 0;
    for (int x = tid; x < have; x += blockDim.x) local += (pay[x] >= 0) ? 1 : 0;
    if (local) atomicAdd(s_n, local);
    __syncthreads();
    const int nv = *s_n;
    for (int x = tid; x < limit; x += blockDim.x) {
        cis_hit hh;
        if (x < nv) {
            hh = src[pay[x]];
        } else {
            hh.dist = __longlong_as_double(0x7ff0000000000000LL);
            hh.visit_rank = 0xffffffffu; hh.pos = 0xffffffffu; hh.id = -1; hh.cell = -1; hh.reserved = 0;
        }
        if (out_hits) out_hits[x] = hh;
        if (out_ids) {
            out_ids[x] = hh.id;
            out_dists[x] = (x < nv) ? hh.dist : __longlong_as_double(0x7ff8000000000000LL);
        }
        if (out_cells) out_cells[x] = hh.cell;
        if (out_pos) out_pos[x] = hh.pos;
    }
    if (tid == 0 && out_n) *out_n = nv;
}

template <int CAPM>
__global__ __launch_bounds__(256) void k_merge_items(const cis_hit* __restrict__ item_hits, const int* __restrict__ item_n,
                                                     const int64_t* __restrict__ item_off, int limit,
                                                     cis_hit* __restrict__ out_hits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* ka = reinterpret_cast<uint64_t*>(smem);
    uint64_t* kb = ka + CAPM;
    int64_t* pay = reinterpret_cast<int64_t*>(kb + CAPM);
    int* s_n = reinterpret_cast<int*>(pay + CAPM);
    const int q = blockIdx.x;
    const int64_t a = item_off[q], b = item_off[q + 1];
    merge_lists<CAPM>(item_hits, item_n, a, (int)(b - a), (int64_t)limit, limit, limit, ka, kb, pay, s_n,
                      out_hits + (int64_t)q * limit, nullptr, nullptr, nullptr, nullptr, nullptr);
}

template <int CAPM>
__global__ __launch_bounds__(256) void k_merge_parts(const cis_hit* __restrict__ parts /* [world][nq][limit] */, int world,
                                                     int nq, int limit, int64_t* __restrict__ out_ids,
                                                     double* __restrict__ out_dists, int* __restrict__ out_n,
                                                     int32_t* __restrict__ out_cells, uint32_t* __restrict__ out_pos) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* ka = reinterpret_cast<uint64_t*>(smem);
    uint64_t* kb = ka + CAPM;
    int64_t* pay = reinterpret_cast<int64_t*>(kb + CAPM);
    int* s_n = reinterpret_cast<int*>(pay + CAPM);
    const int q = blockIdx.x;
    // list w of query q starts at parts + (w*nq + q)*limit: first_list = q, distance between lists = nq*limit
    merge_lists<CAPM>(parts + (int64_t)q * limit, nullptr, 0, world, (int64_t)nq * limit, limit, limit, ka, kb, pay,
                      s_n, nullptr, out_ids + (int64_t)q * limit, out_dists + (int64_t)q * limit, out_n + q,
                      out_cells ? out_cells + (int64_t)q * limit : nullptr, out_pos ? out_pos + (int64_t)q * limit : nullptr);
}

__global__ void k_copy_visited(const PlanOut* __restrict__ plan, int nq, int32_t* __restrict__ visited) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) visited[q] = plan[q].visited;
}

// ================================================================================================
// host: index object
// ================================================================================================
struct PairHash {
    size_t operator()(const std::pair<int64_t, int64_t>& p) const {
        uint64_t x = (uint64_t)p.first * 0x9E3779B97F4A7C15ull ^ ((uint64_t)p.second + 0x7F4A7C15ull);
        x ^= x >> 31;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 29;
        return (size_t)x;
    }
};

struct cis_index {
    cis_model* m = nullptr;
    int V = 0, M = 0;
    int64_t ncells = 0;
    int rank = 0, world = 1;
    std::vector<int32_t> owner;  // empty: cell % world
    // host storage of THIS shard: CSR + pending appends (arrival order)
    std::vector<int64_t> csr_off;  // [ncells+1]
    std::vector<int64_t> csr_ids;
    std::vector<uint8_t> csr_fine;
    std::vector<int64_t> pend_ids;
    std::vector<int64_t> pend_cell;
    std::vector<uint8_t> pend_fine;
    std::vector<int64_t> gcount;  // [ncells] all shards
    std::unordered_set<std::pair<int64_t, int64_t>, PairHash> seen;  // (cell, id) of every dedup add
    bool had_plain_add = false;  // items were added without (cell, id) bookkeeping
    int64_t nb_indexed = 0;
    bool dirty = true;
    // device copy
    DevBuf d_codes, d_ids, d_loff, d_gcount;
    int64_t n_local = 0;
    // per-batch workspace
    DevBuf w_xp, w_cd, w_order, w_sorted, w_plan, w_off, w_items, w_tabs, w_T, w_hits, w_hitn, w_part, w_q,
        w_oids, w_odists, w_onf, w_ovis, w_ocell, w_opos;
    int64_t stats[4] = {0, 0, 0, 0};
    // optional stage timing (hipEvents on the launch stream)
    bool profiling = false;
    struct ProfRec { hipEvent_t ev[5]; bool has_scan; };
    std::vector<ProfRec> prof;
    double prof_ms[4] = {0, 0, 0, 0};
    int64_t prof_launches = 0;

    bool owns(int64_t cell) const {
        if (world <= 1) return true;
        if (!owner.empty()) return owner[cell] == rank;
        return (int)(cell % world) == rank;
    }
};

extern "C" int cis_index_create(cis_index** out, cis_model* m) {
    CIS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CIS_REQUIRE(m != nullptr, "model is NULL");
    cis_index* ix = new cis_index();
    ix->m = m;
    ix->V = m->V;
    ix->M = m->M;
    ix->ncells = (int64_t)m->V * m->V;
    ix->csr_off.assign(ix->ncells + 1, 0);
    ix->gcount.assign(ix->ncells, 0);
    *out = ix;
    return CIS_OK;
}

extern "C" void cis_index_destroy(cis_index* ix) {
    if (!ix) return;
    if (ix->m) (void)hipSetDevice(ix->m->device);
    DevBuf* bufs[] = {&ix->d_codes, &ix->d_ids, &ix->d_loff, &ix->d_gcount, &ix->w_xp, &ix->w_cd, &ix->w_order,
                      &ix->w_sorted, &ix->w_plan, &ix->w_off, &ix->w_items, &ix->w_tabs, &ix->w_T, &ix->w_hits,
                      &ix->w_hitn, &ix->w_part, &ix->w_q, &ix->w_oids, &ix->w_odists, &ix->w_onf, &ix->w_ovis,
                      &ix->w_ocell, &ix->w_opos};
    for (DevBuf* b : bufs) b->release();
    delete ix;
}

extern "C" int cis_index_set_shard(cis_index* ix, int rank, int world, const int32_t* owner) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad shard %d of %d", rank, world);
    CIS_REQUIRE(ix->nb_indexed == 0, "set_shard must be called on an empty index");
    ix->rank = rank;
    ix->world = world;
    ix->owner.clear();
    if (owner) {
        ix->owner.assign(owner, owner + ix->ncells);
        for (int64_t c = 0; c < ix->ncells; ++c)
            CIS_REQUIRE(owner[c] >= 0 && owner[c] < world, "owner[%lld]=%d out of range", (long long)c, owner[c]);
    }
    return CIS_OK;
}

extern "C" int64_t cis_index_size(cis_index* ix) { return ix ? ix->nb_indexed : 0; }

extern "C" int cis_index_add(cis_index* ix, const int64_t* ids, const uint16_t* coarse, const uint8_t* fine,
                             int64_t n, int dedup, int64_t* n_added) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(n >= 0 && (n == 0 || (ids && coarse && fine)), "NULL buffer");
    const int V = ix->V, M = ix->M, K = ix->m->K;
    for (int64_t i = 0; i < n; ++i)
        CIS_REQUIRE(coarse[2 * i] < V && coarse[2 * i + 1] < V, "item %lld: coarse code out of range (V=%d)",
                    (long long)i, V);
    if (K < 256)
        for (int64_t i = 0; i < n * M; ++i)
            CIS_REQUIRE(fine[i] < K, "fine code %d out of range (K=%d)", (int)fine[i], K);
    if (dedup && ix->had_plain_add) {
        cis_set_error("dedup add after plain (dedup=0) adds on the same index is not supported");
        return CIS_EUNSUPPORTED;
    }
    int64_t added = 0;
    try {
        for (int64_t i = 0; i < n; ++i) {
            const int64_t cell = (int64_t)coarse[2 * i] * V + coarse[2 * i + 1];
            if (dedup) {
                if (!ix->seen.insert(std::make_pair(cell, ids[i])).second) continue;
            }
            ix->gcount[cell] += 1;
            ++added;
            if (ix->owns(cell)) {
                ix->pend_ids.push_back(ids[i]);
                ix->pend_cell.push_back(cell);
                ix->pend_fine.insert(ix->pend_fine.end(), fine + i * M, fine + (i + 1) * M);
            }
        }
    } catch (const std::bad_alloc&) {
        cis_set_error("out of host memory while adding codes");
        return CIS_ENOMEM;
    }
    if (!dedup && n > 0) ix->had_plain_add = true;
    ix->nb_indexed += added;
    if (added) ix->dirty = true;
    if (n_added) *n_added = added;
    return CIS_OK;
}

// merge pending appends into the host CSR (stable: old items of a cell first, then new ones in
// arrival order) and refresh the device copy
static int index_sync(cis_index* ix) {
    if (!ix->dirty) return CIS_OK;
    CIS_TRY(cis_lazy_init());
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    const int M = ix->M;
    const int64_t nc = ix->ncells;
    const int64_t np = (int64_t)ix->pend_ids.size();
    try {
        if (np > 0) {
            std::vector<int64_t> add(nc + 1, 0);
            for (int64_t i = 0; i < np; ++i) add[ix->pend_cell[i] + 1] += 1;
            std::vector<int64_t> noff(nc + 1, 0);
            for (int64_t c = 0; c < nc; ++c) noff[c + 1] = noff[c] + (ix->csr_off[c + 1] - ix->csr_off[c]) + add[c + 1];
            const int64_t ntot = noff[nc];
            std::vector<int64_t> nids((size_t)ntot);
            std::vector<uint8_t> nfine((size_t)ntot * M);
            std::vector<int64_t> cur(nc);
            for (int64_t c = 0; c < nc; ++c) {
                const int64_t a = ix->csr_off[c], b = ix->csr_off[c + 1];
                if (b > a) {
                    memcpy(&nids[noff[c]], &ix->csr_ids[a], (size_t)(b - a) * sizeof(int64_t));
                    memcpy(&nfine[(size_t)noff[c] * M], &ix->csr_fine[(size_t)a * M], (size_t)(b - a) * M);
                }
                cur[c] = noff[c] + (b - a);
            }
            for (int64_t i = 0; i < np; ++i) {
                const int64_t p = cur[ix->pend_cell[i]]++;
                nids[p] = ix->pend_ids[i];
                memcpy(&nfine[(size_t)p * M], &ix->pend_fine[(size_t)i * M], M);
            }
            ix->csr_off.swap(noff);
            ix->csr_ids.swap(nids);
            ix->csr_fine.swap(nfine);
            std::vector<int64_t>().swap(ix->pend_ids);
            std::vector<int64_t>().swap(ix->pend_cell);
            std::vector<uint8_t>().swap(ix->pend_fine);
        }
    } catch (const std::bad_alloc&) {
        cis_set_error("out of host memory while building the cell index");
        return CIS_ENOMEM;
    }
    const int64_t nl = ix->csr_off[nc];
    ix->n_local = nl;
    CIS_TRY(ix->d_codes.reserve((size_t)(nl > 0 ? nl : 1) * M + 64));
    CIS_TRY(ix->d_ids.reserve((size_t)(nl > 0 ? nl : 1) * sizeof(int64_t)));
    CIS_TRY(ix->d_loff.reserve((size_t)(nc + 1) * sizeof(int64_t)));
    CIS_TRY(ix->d_gcount.reserve((size_t)nc * sizeof(int64_t)));
    if (nl > 0) {
        CIS_CHECK_HIP(hipMemcpy(ix->d_codes.p, ix->csr_fine.data(), (size_t)nl * M, hipMemcpyHostToDevice));
        CIS_CHECK_HIP(hipMemcpy(ix->d_ids.p, ix->csr_ids.data(), (size_t)nl * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    CIS_CHECK_HIP(hipMemcpy(ix->d_loff.p, ix->csr_off.data(), (size_t)(nc + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(ix->d_gcount.p, ix->gcount.data(), (size_t)nc * sizeof(int64_t), hipMemcpyHostToDevice));
    ix->dirty = false;
    return CIS_OK;
}

extern "C" int cis_index_get_cell(cis_index* ix, int c0, int c1, int64_t cap, int64_t* ids, uint8_t* fine, int64_t* n) {
    CIS_REQUIRE(ix != nullptr && n != nullptr, "NULL argument");
    CIS_REQUIRE(c0 >= 0 && c0 < ix->V && c1 >= 0 && c1 < ix->V, "cell (%d,%d) out of range", c0, c1);
    const int64_t cell = (int64_t)c0 * ix->V + c1;
    *n = ix->gcount[cell];
    if (cap <= 0 || !ix->owns(cell)) return CIS_OK;
    // host-only merge is enough here; the device copy is refreshed by the next search
    const int M = ix->M;
    std::vector<int64_t> tmp_ids;
    std::vector<uint8_t> tmp_fine;
    const int64_t a = ix->csr_off[cell], b = ix->csr_off[cell + 1];
    int64_t k = 0;
    for (int64_t p = a; p < b && k < cap; ++p, ++k) {
        if (ids) ids[k] = ix->csr_ids[p];
        if (fine) memcpy(fine + k * M, &ix->csr_fine[(size_t)p * M], M);
    }
    for (size_t i = 0; i < ix->pend_ids.size() && k < cap; ++i) {
        if (ix->pend_cell[i] != cell) continue;
        if (ids) ids[k] = ix->pend_ids[i];
        if (fine) memcpy(fine + k * M, &ix->pend_fine[i * M], M);
        ++k;
    }
    return CIS_OK;
}

extern "C" int cis_index_get_codes(cis_index* ix, const int32_t* cells, const uint32_t* pos, int64_t n, uint8_t* fine) {
    CIS_REQUIRE(ix != nullptr && (n == 0 || (cells && pos && fine)), "NULL argument");
    CIS_REQUIRE(ix->pend_ids.empty(), "index has unsynchronised adds; search first");
    const int M = ix->M;
    for (int64_t i = 0; i < n; ++i) {
        CIS_REQUIRE(cells[i] >= 0 && cells[i] < ix->ncells, "cell %d out of range", cells[i]);
        const int64_t a = ix->csr_off[cells[i]], b = ix->csr_off[cells[i] + 1];
        CIS_REQUIRE((int64_t)pos[i] < b - a, "item (%d, %u) is not stored on this shard", cells[i], pos[i]);
        memcpy(fine + i * M, &ix->csr_fine[(size_t)(a + pos[i]) * M], M);
    }
    return CIS_OK;
}

extern "C" int cis_index_set_profiling(cis_index* ix, int enable) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    ix->profiling = enable != 0;
    return CIS_OK;
}

extern "C" int cis_index_read_profile(cis_index* ix, double ms[4], int64_t* launches) {
    CIS_REQUIRE(ix != nullptr && ms != nullptr, "NULL argument");
    for (auto& r : ix->prof) {
        CIS_CHECK_HIP(hipEventSynchronize(r.ev[4]));
        for (int i = 0; i < 4; ++i) {
            float t = 0.f;
            CIS_CHECK_HIP(hipEventElapsedTime(&t, r.ev[i], r.ev[i + 1]));
            ix->prof_ms[i] += t;
        }
        if (r.has_scan) ix->prof_launches += 1;
        for (int i = 0; i < 5; ++i) (void)hipEventDestroy(r.ev[i]);
    }
    ix->prof.clear();
    for (int i = 0; i < 4; ++i) { ms[i] = ix->prof_ms[i]; ix->prof_ms[i] = 0; }
    if (launches) *launches = ix->prof_launches;
    ix->prof_launches = 0;
    return CIS_OK;
}

extern "C" int cis_index_last_stats(cis_index* ix, int64_t stats[4]) {
    CIS_REQUIRE(ix != nullptr && stats != nullptr, "NULL argument");
    for (int i = 0; i < 4; ++i) stats[i] = ix->stats[i];
    return CIS_OK;
}

// ================================================================================================
// host: search pipeline
// ================================================================================================
template <int M, int CAP, int U>
static void launch_scan(int64_t n_items, size_t lds, hipStream_t st, const WorkItem* items, const double* T,
                        const uint8_t* codes, const int64_t* ids, int Mrt, int K, int limit, cis_hit* hits, int* hitn) {
    hipLaunchKernelGGL((k_adc_scan<M, CAP, U>), dim3((unsigned)n_items), dim3(256), lds, st, items, T, codes, ids, Mrt,
                       K, limit, hits, hitn);
}

template <int CAP, int U>
static void launch_scan_m(int M, int64_t n_items, hipStream_t st, const WorkItem* items, const double* T,
                          const uint8_t* codes, const int64_t* ids, int K, int limit, cis_hit* hits, int* hitn) {
    const size_t lds = (size_t)CAP * 16 + (size_t)M * K * sizeof(double) + 16;
    switch (M) {
        case 4: launch_scan<4, CAP, U>(n_items, lds, st, items, T, codes, ids, M, K, limit, hits, hitn); break;
        case 8: launch_scan<8, CAP, U>(n_items, lds, st, items, T, codes, ids, M, K, limit, hits, hitn); break;
        case 16: launch_scan<16, CAP, U>(n_items, lds, st, items, T, codes, ids, M, K, limit, hits, hitn); break;
        default: launch_scan<0, CAP, U>(n_items, lds, st, items, T, codes, ids, M, K, limit, hits, hitn); break;
    }
}

static const int MAX_LIMIT = 3072;

// one sub-batch of queries (device pointers); writes ranked partial hits [nq][L] and visited [nq]
static int search_batch(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, int L, cis_hit* d_hits,
                        int32_t* d_visited, hipStream_t st) {
    cis_model* m = ix->m;
    const int V = m->V, D = m->D, K = m->K, M = m->M, h = m->h, nf = m->nf;
    cis_index::ProfRec pr;
    pr.has_scan = false;
    auto mark = [&](int i) -> int {
        if (!ix->profiling) return CIS_OK;
        CIS_CHECK_HIP(hipEventCreate(&pr.ev[i]));
        CIS_CHECK_HIP(hipEventRecord(pr.ev[i], st));
        return CIS_OK;
    };
    CIS_TRY(mark(0));
    // 1. LOPQ-space queries
    const void* xp = dQ;
    int xp_dtype = q_dtype;
    if (m->has_pca) {
        CIS_TRY(ix->w_xp.reserve((size_t)nq * D * sizeof(float)));
        CIS_TRY(cis_dev_apply_pca(m, dQ, q_dtype, nq, ix->w_xp.as<float>(), st));
        xp = ix->w_xp.p;
        xp_dtype = CIS_F32;
    }
    const void* xc;
    int ct;
    CIS_TRY(cis_dev_coarse_type(m, xp, xp_dtype, nq, &xc, &ct, st));
    const size_t csz = (ct == CIS_F32) ? 4 : 8;
    // 2. coarse distances, rank
    CIS_TRY(ix->w_cd.reserve((size_t)2 * nq * V * csz));
    CIS_TRY(ix->w_sorted.reserve((size_t)2 * nq * V * csz));
    CIS_TRY(ix->w_order.reserve((size_t)2 * nq * V * sizeof(uint16_t)));
    CIS_TRY(ix->w_plan.reserve((size_t)nq * sizeof(PlanOut)));
    CIS_TRY(ix->w_off.reserve((size_t)(2 * (nq + 1) + 4) * sizeof(int64_t)));
    int64_t* item_off = ix->w_off.as<int64_t>();
    int64_t* tab_off = item_off + (nq + 1);
    int64_t* totals = tab_off + (nq + 1);
    PlanOut* plan = ix->w_plan.as<PlanOut>();
    const int seg_max = nq >= 1024 ? (1 << 20) : (nq >= 64 ? 16384 : 4096);
    for (int s = 0; s < 2; ++s)
        CIS_TRY(cis_launch_sqdist(m, xc, ct, nq, s, (char*)ix->w_cd.p + (size_t)s * nq * V * csz, st));
    const size_t plan_lds = (size_t)V * sizeof(int);
    if (ct == CIS_F32) {
        hipLaunchKernelGGL(k_rank<float>, dim3(nq, 2), dim3(V <= 64 ? 64 : 256), (size_t)V * 8, st, ix->w_cd.as<float>(), nq, V,
                           ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>());
        hipLaunchKernelGGL((k_plan<float, false>), dim3(nq), dim3(64), plan_lds, st, ix->w_sorted.as<float>(),
                           ix->w_order.as<uint16_t>(), ix->d_gcount.as<int64_t>(), ix->d_loff.as<int64_t>(), nq, V, quota,
                           seg_max, plan, nullptr, nullptr, nullptr, nullptr);
    } else {
        hipLaunchKernelGGL(k_rank<double>, dim3(nq, 2), dim3(V <= 64 ? 64 : 256), (size_t)V * 8, st, ix->w_cd.as<double>(), nq,
                           V, ix->w_order.as<uint16_t>(), ix->w_sorted.as<double>());
        hipLaunchKernelGGL((k_plan<double, false>), dim3(nq), dim3(64), plan_lds, st, ix->w_sorted.as<double>(),
                           ix->w_order.as<uint16_t>(), ix->d_gcount.as<int64_t>(), ix->d_loff.as<int64_t>(), nq, V, quota,
                           seg_max, plan, nullptr, nullptr, nullptr, nullptr);
    }
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(256), 0, st, plan, nq, item_off, tab_off, totals);
    int64_t h_tot[3];
    CIS_CHECK_HIP(hipMemcpyAsync(h_tot, totals, sizeof(h_tot), hipMemcpyDeviceToHost, st));
    CIS_CHECK_HIP(hipStreamSynchronize(st));
    const int64_t n_items = h_tot[0], n_tabs = h_tot[1];
    ix->stats[0] += h_tot[2];
    ix->stats[1] += n_items;
    ix->stats[2] += n_tabs;
    CIS_REQUIRE(n_items < ((int64_t)1 << 31) && n_tabs < ((int64_t)1 << 31), "query batch too large");
    // 3. emit items + table list
    CIS_TRY(mark(1));  // the plan read-back above is part of the front end
    CIS_TRY(ix->w_items.reserve((size_t)(n_items + 1) * sizeof(WorkItem)));
    CIS_TRY(ix->w_tabs.reserve((size_t)(n_tabs + 1) * sizeof(TabDesc)));
    CIS_TRY(ix->w_T.reserve((size_t)(n_tabs + 1) * nf * K * sizeof(double)));
    CIS_TRY(ix->w_hits.reserve((size_t)(n_items + 1) * L * sizeof(cis_hit)));
    CIS_TRY(ix->w_hitn.reserve((size_t)(n_items + 1) * sizeof(int)));
    WorkItem* items = ix->w_items.as<WorkItem>();
    TabDesc* tabs = ix->w_tabs.as<TabDesc>();
    double* T = ix->w_T.as<double>();
    const size_t tab_lds = (size_t)2 * h * sizeof(double);
    if (ct == CIS_F32) {
        hipLaunchKernelGGL((k_plan<float, true>), dim3(nq), dim3(64), plan_lds, st, ix->w_sorted.as<float>(),
                           ix->w_order.as<uint16_t>(), ix->d_gcount.as<int64_t>(), ix->d_loff.as<int64_t>(), nq, V, quota,
                           seg_max, plan, item_off, tab_off, items, tabs);
        if (n_tabs > 0)
            hipLaunchKernelGGL(k_tables<float>, dim3((unsigned)n_tabs), dim3(256), tab_lds, st, (const float*)xc, m->d_Cs32,
                               m->d_Rt, m->d_mus, m->d_subs, tabs, V, h, m->w, nf, K, D, T, m->prog_w);
    } else {
        hipLaunchKernelGGL((k_plan<double, true>), dim3(nq), dim3(64), plan_lds, st, ix->w_sorted.as<double>(),
                           ix->w_order.as<uint16_t>(), ix->d_gcount.as<int64_t>(), ix->d_loff.as<int64_t>(), nq, V, quota,
                           seg_max, plan, item_off, tab_off, items, tabs);
        if (n_tabs > 0)
            hipLaunchKernelGGL(k_tables<double>, dim3((unsigned)n_tabs), dim3(256), tab_lds, st, (const double*)xc,
                               m->d_Cs64, m->d_Rt, m->d_mus, m->d_subs, tabs, V, h, m->w, nf, K, D, T, m->prog_w);
    }
    // 4. ADC scan + block top-k
    CIS_TRY(mark(2));
    if (n_items > 0) {
        pr.has_scan = true;
        const uint8_t* codes = ix->d_codes.as<uint8_t>();
        const int64_t* ids = ix->d_ids.as<int64_t>();
        cis_hit* hits = ix->w_hits.as<cis_hit>();
        int* hitn = ix->w_hitn.as<int>();
        if (L <= 512) launch_scan_m<1024, 2>(M, n_items, st, items, T, codes, ids, K, L, hits, hitn);
        else if (L <= 1024) launch_scan_m<2048, 4>(M, n_items, st, items, T, codes, ids, K, L, hits, hitn);
        else launch_scan_m<4096, 4>(M, n_items, st, items, T, codes, ids, K, L, hits, hitn);
        ix->stats[3] += 1;
    }
    // 5. per-query merge
    CIS_TRY(mark(3));
    {
        const cis_hit* hits = ix->w_hits.as<cis_hit>();
        const int* hitn = ix->w_hitn.as<int>();
        if (L <= 512)
            hipLaunchKernelGGL(k_merge_items<1024>, dim3(nq), dim3(256), (size_t)1024 * 24 + 16, st, hits, hitn, item_off, L, d_hits);
        else if (L <= 1024)
            hipLaunchKernelGGL(k_merge_items<2048>, dim3(nq), dim3(256), (size_t)2048 * 24 + 16, st, hits, hitn, item_off, L, d_hits);
        else
            hipLaunchKernelGGL(k_merge_items<4096>, dim3(nq), dim3(256), (size_t)4096 * 24 + 16, st, hits, hitn, item_off, L, d_hits);
    }
    hipLaunchKernelGGL(k_copy_visited, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, st, plan, nq, d_visited);
    CIS_CHECK_HIP(hipGetLastError());
    CIS_TRY(mark(4));
    if (ix->profiling) ix->prof.push_back(pr);
    return CIS_OK;
}

static int effective_limit(int64_t quota, int limit, int* L) {
    int64_t l = limit < 0 ? quota : limit;  // search.py:213-214
    if (l < 0) l = 0;
    if (l > MAX_LIMIT) {
        cis_set_error("limit=%lld exceeds the %d ranked results per query supported by this build", (long long)l, MAX_LIMIT);
        return CIS_EUNSUPPORTED;
    }
    *L = (int)l;
    return CIS_OK;
}

static const int QUERY_BATCH = 8192;

extern "C" int cis_index_search_partial_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota,
                                            int limit, cis_hit* d_hits, int32_t* d_visited, void* stream) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(q_dtype == CIS_F32 || q_dtype == CIS_F64, "q_dtype must be 4 or 8");
    CIS_REQUIRE(nq >= 0, "nq must be >= 0");
    int L;
    CIS_TRY(effective_limit(quota, limit, &L));
    CIS_TRY(index_sync(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 4; ++i) ix->stats[i] = 0;
    if (L == 0 || nq == 0) {
        // still report visited
        if (nq == 0) return CIS_OK;
    }
    const int Lk = L > 0 ? L : 1;
    for (int a = 0; a < nq; a += QUERY_BATCH) {
        const int bn = (nq - a < QUERY_BATCH) ? (nq - a) : QUERY_BATCH;
        const char* q = (const char*)dQ + (size_t)a * ix->m->D_in * q_dtype;
        if (L > 0) {
            CIS_TRY(search_batch(ix, q, q_dtype, bn, quota, L, d_hits + (int64_t)a * L, d_visited + a, st));
        } else {
            CIS_TRY(ix->w_part.reserve((size_t)bn * Lk * sizeof(cis_hit)));
            CIS_TRY(search_batch(ix, q, q_dtype, bn, quota, Lk, ix->w_part.as<cis_hit>(), d_visited + a, st));
        }
    }
    return CIS_OK;
}

static int merge_parts(const cis_hit* d_parts, int world, int nq, int L, int64_t* d_ids, double* d_dists,
                       int32_t* d_nf, int32_t* d_cells, uint32_t* d_pos, hipStream_t st) {
    if (nq == 0 || L == 0) return CIS_OK;
    if (L <= 512)
        hipLaunchKernelGGL(k_merge_parts<1024>, dim3(nq), dim3(256), (size_t)1024 * 24 + 16, st, d_parts, world, nq, L, d_ids, d_dists, d_nf, d_cells, d_pos);
    else if (L <= 1024)
        hipLaunchKernelGGL(k_merge_parts<2048>, dim3(nq), dim3(256), (size_t)2048 * 24 + 16, st, d_parts, world, nq, L, d_ids, d_dists, d_nf, d_cells, d_pos);
    else
        hipLaunchKernelGGL(k_merge_parts<4096>, dim3(nq), dim3(256), (size_t)4096 * 24 + 16, st, d_parts, world, nq, L, d_ids, d_dists, d_nf, d_cells, d_pos);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

extern "C" int cis_merge_hits_dev(const cis_hit* d_parts, int world, int nq, int limit, int64_t* d_ids,
                                  double* d_dists, int32_t* d_n_found, int32_t* d_cells, uint32_t* d_pos,
                                  void* stream) {
    CIS_REQUIRE(world >= 1 && nq >= 0 && limit >= 0 && limit <= MAX_LIMIT, "bad merge arguments");
    CIS_TRY(cis_lazy_init());
    return merge_parts(d_parts, world, nq, limit, d_ids, d_dists, d_n_found, d_cells, d_pos, (hipStream_t)stream);
}

extern "C" int cis_index_search_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, int limit,
                                    int64_t* d_ids, double* d_dists, int32_t* d_n_found, int32_t* d_visited,
                                    int32_t* d_cells, uint32_t* d_pos, void* stream) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    int L;
    CIS_TRY(effective_limit(quota, limit, &L));
    if (nq == 0) return CIS_OK;
    CIS_TRY(ix->w_part.reserve((size_t)nq * (L > 0 ? L : 1) * sizeof(cis_hit)));
    CIS_TRY(cis_index_search_partial_dev(ix, dQ, q_dtype, nq, quota, limit, ix->w_part.as<cis_hit>(), d_visited, stream));
    if (L == 0) {
        CIS_CHECK_HIP(hipMemsetAsync(d_n_found, 0, (size_t)nq * sizeof(int32_t), (hipStream_t)stream));
        return CIS_OK;
    }
    return merge_parts(ix->w_part.as<cis_hit>(), 1, nq, L, d_ids, d_dists, d_n_found, d_cells, d_pos, (hipStream_t)stream);
}

extern "C" int cis_index_search(cis_index* ix, const void* Q, int q_dtype, int nq, int64_t quota, int limit,
                                int64_t* ids, double* dists, int32_t* n_found, int32_t* visited, int32_t* cells,
                                uint32_t* pos) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(q_dtype == CIS_F32 || q_dtype == CIS_F64, "q_dtype must be 4 or 8");
    int L;
    CIS_TRY(effective_limit(quota, limit, &L));
    if (nq == 0) return CIS_OK;
    CIS_REQUIRE(Q && n_found && visited && (L == 0 || (ids && dists)), "NULL buffer");
    CIS_TRY(cis_lazy_init());
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    const size_t qbytes = (size_t)nq * ix->m->D_in * q_dtype;
    const int Lk = L > 0 ? L : 1;
    CIS_TRY(ix->w_q.reserve(qbytes));
    CIS_TRY(ix->w_oids.reserve((size_t)nq * Lk * sizeof(int64_t)));
    CIS_TRY(ix->w_odists.reserve((size_t)nq * Lk * sizeof(double)));
    CIS_TRY(ix->w_onf.reserve((size_t)nq * sizeof(int32_t)));
    CIS_TRY(ix->w_ovis.reserve((size_t)nq * sizeof(int32_t)));
    CIS_CHECK_HIP(hipMemcpy(ix->w_q.p, Q, qbytes, hipMemcpyHostToDevice));
    CIS_TRY(ix->w_ocell.reserve((size_t)nq * Lk * sizeof(int32_t)));
    CIS_TRY(ix->w_opos.reserve((size_t)nq * Lk * sizeof(uint32_t)));
    CIS_TRY(cis_index_search_dev(ix, ix->w_q.p, q_dtype, nq, quota, limit, ix->w_oids.as<int64_t>(),
                                 ix->w_odists.as<double>(), ix->w_onf.as<int32_t>(), ix->w_ovis.as<int32_t>(),
                                 ix->w_ocell.as<int32_t>(), ix->w_opos.as<uint32_t>(), nullptr));
    CIS_CHECK_HIP(hipDeviceSynchronize());
    if (L > 0) {
        CIS_CHECK_HIP(hipMemcpy(ids, ix->w_oids.p, (size_t)nq * L * sizeof(int64_t), hipMemcpyDeviceToHost));
        CIS_CHECK_HIP(hipMemcpy(dists, ix->w_odists.p, (size_t)nq * L * sizeof(double), hipMemcpyDeviceToHost));
        if (cells) CIS_CHECK_HIP(hipMemcpy(cells, ix->w_ocell.p, (size_t)nq * L * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (pos) CIS_CHECK_HIP(hipMemcpy(pos, ix->w_opos.p, (size_t)nq * L * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    CIS_CHECK_HIP(hipMemcpy(n_found, ix->w_onf.p, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost));
    CIS_CHECK_HIP(hipMemcpy(visited, ix->w_ovis.p, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost));
    return CIS_OK;
}
