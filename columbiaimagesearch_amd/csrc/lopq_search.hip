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
#include <atomic>
#include <chrono>
#include <mutex>
#include <algorithm>

#include "lopq_index.h"
#include "scan_common.h"

// ================================================================================================
// device structures
// ================================================================================================

static __device__ __forceinline__ uint64_t f2bits(double d) { return (uint64_t)__double_as_longlong(d); }
static __device__ __forceinline__ uint64_t f2bits(float f) { return (uint64_t)__float_as_uint(f); }

// ================================================================================================
// kernels: coarse ranking and multisequence plan
// ================================================================================================
// Ascending order of the V coarse distances of one (query, split).  Distances are >= 0 so their
// bit patterns order like the values (NaN sorts last, as np.argsort does).  Ties -> lower index.
// counters of the table groups are split GRP_SUB ways by query index: 16 k atomics on 32 addresses would serialise
static const int GRP_SUB = 32;
// words of the counters, cursors and bases of the table groups (+ 2, to an even count), then one 64-bit word per tile of k_group_bases
#define GRP_WORDS(V) (6 * (V) * GRP_SUB + 2)
#define GRP_TILES(V) ((2 * (V) * GRP_SUB + 1023) / 1024)

template <typename CT>
__global__ void k_rank(const CT* __restrict__ dist /* [2][nq][V] */, int nq, int V,
                       uint16_t* __restrict__ order /* [nq][2][V] */, CT* __restrict__ sorted /* [nq][2][V] */,
                       int* __restrict__ grp /* [4V]: tables per (split, cluster) and cursors, zeroed here for k_plan */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* sb = reinterpret_cast<uint64_t*>(smem);
    const int q = blockIdx.x, s = blockIdx.y;
    if (q == 0 && s == 0)
        for (int i = threadIdx.x; i < 4 * V * GRP_SUB; i += blockDim.x) grp[i] = 0;
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
                                             WorkItem* __restrict__ items, TabDesc* __restrict__ tabs,
                                             int* __restrict__ grp_cnt /* [2V] */, const int* __restrict__ grp_base /* [2V] */,
                                             int* __restrict__ grp_cur /* [2V] */, int* __restrict__ tab_order /* [n_tabs] */,
                                             const int* __restrict__ only /* null, or [nq]: walk only the flagged queries (k_plan_par's fallback) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* t = reinterpret_cast<int*>(smem);  // [V]
    const int q = blockIdx.x;
    const int lane = threadIdx.x;
    if (only && !only[q]) return;
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
        const int64_t ll = loff[(int64_t)V * V + 1 + cell] - ls;  // the used end (lend = loff + ncells + 1: cells keep insert slack behind their items)
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
        // tables per (split, cluster): k_tables handles the tables of one cluster together (one read of R[c])
        for (int i = lane; i < max_i + 1 + max_j + 1; i += 64) {
            const int g = i <= max_i ? (int)o0[i] : V + (int)o1[i - (max_i + 1)];
            atomicAdd(&grp_cnt[g * GRP_SUB + (q % GRP_SUB)], 1);
        }
    } else {
        const int nt0 = plan[q].ntab0, nt1 = plan[q].ntab1;
        for (int i = lane; i < nt0 + nt1; i += 64) {
            TabDesc td;
            td.q = q; td.pad = 0;
            if (i < nt0) { td.split = 0; td.cluster = o0[i]; }
            else { td.split = 1; td.cluster = o1[i - nt0]; }
            tabs[tbase + i] = td;
            const int g = (td.split * V + td.cluster) * GRP_SUB + (q % GRP_SUB);
            tab_order[grp_base[g] + atomicAdd(&grp_cur[g], 1)] = (int)(tbase + i);  // order inside a group does not matter
        }
    }
}

// Fused front end for small coarse codebooks (V <= 64: BASELINE configs C1-C5): the exact coarse distances of k_sqdist_rows2
// (lopq/lopq/search.py:39 -> lopq/lopq/utils.py:33-53 arithmetic: numpy's pairwise order, compute type CT), the ascending rank of
// k_rank (np.argsort, ties to the lower index) and the counting pass of the multisequence walk (k_plan<CT, false>) in ONE launch, one
// wave per query: the 2 V distances never leave the CU between the three steps (three launches and two round trips through L2 of
// the [2][nq][V] arrays before).  `sorted` / `order` still go to global memory for the emit pass and the tables.
template <typename CT>
__global__ __launch_bounds__(64) void k_front_small(const CT* __restrict__ X /* [nq][D] */, int D, int h, const CT* __restrict__ Cs /* [2][V][h] */,
                                                    PwProg prog, const int64_t* __restrict__ gcount, const int64_t* __restrict__ loff,
                                                    int nq, int V, int64_t quota, int seg_max, uint16_t* __restrict__ order /* [nq][2][V] */,
                                                    CT* __restrict__ sorted /* [nq][2][V] */, PlanOut* __restrict__ plan,
                                                    int* __restrict__ grp_cnt /* zeroed by the previous batch's k_plan_scan */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* sb = reinterpret_cast<uint64_t*>(smem);            // [2V] distance bits
    CT* sd = reinterpret_cast<CT*>(sb + 2 * V);                  // [2][V] ascending distances
    uint16_t* so = reinterpret_cast<uint16_t*>(sb + 4 * V);      // [2][V] cluster of every rank (8 bytes reserved per value of sd)
    int* t = reinterpret_cast<int*>(so + 2 * V);                 // [V] frontier
    const int q = blockIdx.x, lane = threadIdx.x;
    const CT* xr = X + (int64_t)q * D;
    for (int e = lane; e < 2 * V; e += 64) {
        const int s = e / V, c = e - s * V;
        const CT* x = xr + s * h;
        const CT* cc = Cs + ((int64_t)s * V + c) * h;
        auto elem = [&](int i) -> CT { const CT df = x[i] - cc[i]; return df * df; };
        const CT d = pw_sum<CT>(prog, elem);
        sb[e] = f2bits(d);
    }
    for (int i = lane; i < V; i += 64) t[i] = 0;
    __syncthreads();
    for (int e = lane; e < 2 * V; e += 64) {
        const int s = e / V, v = e - s * V;
        const uint64_t mine = sb[e];
        int r = 0;
        for (int u = 0; u < V; ++u) {
            const uint64_t o = sb[s * V + u];
            r += (o < mine) || (o == mine && u < v);
        }
        CT d;
        if constexpr (sizeof(CT) == 4) d = __uint_as_float((uint32_t)mine);
        else d = __longlong_as_double((long long)mine);
        sd[s * V + r] = d;
        so[s * V + r] = (uint16_t)v;
        order[((int64_t)q * 2 + s) * V + r] = (uint16_t)v;
        sorted[((int64_t)q * 2 + s) * V + r] = d;
    }
    __syncthreads();
    const CT* d0 = sd;
    const CT* d1 = sd + V;
    const uint16_t* o0 = so;
    const uint16_t* o1 = so + V;
    // the counting pass of k_plan (same frontier walk, inputs in LDS)
    int visited = 0, n_items = 0, max_i = -1, max_j = -1;
    int64_t retrieved = 0, ncand = 0;
    int rows = 1;
    const int64_t total_cells = (int64_t)V * V;
    while ((int64_t)visited < total_cells) {
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
        if (rows > 1) {  // wave-uniform: the first step has one frontier cell, in lane 0
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint64_t ok = __shfl_xor(bk, off);
                const uint32_t oij = __shfl_xor(bij, off);
                if (ok < bk || (ok == bk && oij < bij)) { bk = ok; bij = oij; }
            }
        } else {
            bij = (uint32_t)__builtin_amdgcn_readfirstlane((int)bij);
        }
        if (bij == ~0u) break;
        const int bi = (int)(bij >> 16), bj = (int)(bij & 0xffff);
        const int c0 = o0[bi], c1 = o1[bj];
        const int64_t cell = (int64_t)c0 * V + c1;
        const int64_t gc = gcount[cell];
        const int64_t ls = loff[cell];
        const int64_t ll = loff[(int64_t)V * V + 1 + cell] - ls;  // the used end (lend = loff + ncells + 1: cells keep insert slack behind their items)
        if (ll > 0) {
            n_items += (int)((ll + seg_max - 1) / seg_max);
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
    if (lane == 0) {
        PlanOut p;
        p.visited = visited; p.n_items = n_items; p.ntab0 = max_i + 1; p.ntab1 = max_j + 1; p.ncand = ncand;
        plan[q] = p;
    }
    for (int i = lane; i < max_i + 1 + max_j + 1; i += 64) {
        const int g = i <= max_i ? (int)o0[i] : V + (int)o1[i - (max_i + 1)];
        atomicAdd(&grp_cnt[g * GRP_SUB + (q % GRP_SUB)], 1);
    }
}

// The same plan for indexes with thousands of coarse clusters (the reference's release configurations use V = 2048 / 4096:
// millions of tiny cells, hundreds to thousands of cells per query at quota 10000), where one frontier step per visited
// cell is the whole cost of a search.  The multisequence order is the order of the sums s(i, j) = fl(d0[i] + d1[j]) (rank
// pairs, both lists ascending): the heap of lopq/lopq/search.py:58-82 holds (s, (i, j)) keys and a cell enters it when both
// its predecessors (i-1, j), (i, j-1) have been popped.  Those are componentwise smaller, so their keys are smaller too (s is
// monotone in i and j, the pair breaks ties): by induction everything with a smaller key is popped before a given cell, i.e.
// the heap's order IS the sorted order of the keys, ties included.  So, per query and with one workgroup:
//   1. bisection on the VALUE tau (bit patterns order like the non-negative sums): count of {s <= tau} = sum over rows of a
//      prefix length (binary search along the ascending d1), until about `target` cells are inside;
//   2. the cells {s <= tau} are enumerated into LDS with the global sizes of their cells and sorted by (s, i, j) (bitonic);
//   3. a prefix sum of the cell sizes in that order finds the quota cut (search.py:128-133); too few candidates inside ->
//      target * 4 and again;
//   4. a band that cannot be cut below what the workgroup sorts (thousands of equal sums), or more visited cells than the
//      list holds: the query is flagged and the frontier walk above (k_plan with `only`) handles it.
// The count pass leaves the visited (i, j) list in global memory for the emit pass.
#ifdef CIS_PLAN_DBG  // tools/build_variant.sh plandbg -DCIS_PLAN_DBG: probes / bands / cycles per phase of k_plan_par's count pass
__device__ unsigned long long g_plan_dbg[12];
#define PLAN_DBG(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_plan_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define PLAN_DBG(i, v) do { } while (0)
#endif
static const int PLAN_PAR_CAP = 2048;   // cells a workgroup enumerates and sorts per band
static const int PLAN_PAR_STAGE = 4096; // d0 / d1 staged in LDS: the kernel takes V <= 4096
static const int PLAN_NB_LOG = 10, PLAN_NB = 1 << PLAN_NB_LOG;  // buckets of a band's distribution sort

// all of d0 / d1 is staged in LDS (V <= PLAN_PAR_STAGE); read in place (no generic pointers to the LDS arrays)
#ifndef CIS_PLAN_WPE
#define CIS_PLAN_WPE 3
#endif
static const int PLAN_SP = 1024;        // ... of which the first PLAN_SP ranks of either list are staged in LDS (the rest is read in place)
template <typename CT> struct PlanKeyT { typedef uint64_t type; };
template <> struct PlanKeyT<float> { typedef uint32_t type; };
#define PL0(i) ((i) < SP ? s_d0[(i)] : d0[(i)])
#define PL1(i) ((i) < SP ? s_d1[(i)] : d1[(i)])

template <typename CT, bool EMIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CIS_PLAN_WPE))) void k_plan_par(const CT* __restrict__ sorted, const uint16_t* __restrict__ order,
                                                  const int64_t* __restrict__ gcount, const int64_t* __restrict__ loff,
                                                  int nq, int V, int64_t quota, int seg_max, PlanOut* __restrict__ plan,
                                                  const int64_t* __restrict__ item_off, const int64_t* __restrict__ tab_off,
                                                  WorkItem* __restrict__ items, TabDesc* __restrict__ tabs,
                                                  int* __restrict__ grp_cnt, const int* __restrict__ grp_base,
                                                  int* __restrict__ grp_cur, int* __restrict__ tab_order,
                                                  uint64_t* __restrict__ ent_list /* [nq][ent_cap][2]: the visited cells that hold anything, in visit
                                                  order: start (40 bits) | (i << 12 | j) << 40, then length | visit rank << 32 */,
                                                  int* __restrict__ fallback /* [nq] flags, then [nq] entries per query */, int ent_cap,
                                                  unsigned long long* __restrict__ hint /* null, or [2][2]: (cells visited, quota) summed over the
                                                  queries of the launches of either parity (count pass) */, int hint_slot) {
    __shared__ typename PlanKeyT<CT>::type s_key[PLAN_PAR_CAP];
    __shared__ uint32_t s_ij[PLAN_PAR_CAP];
    __shared__ uint32_t s_gc[PLAN_PAR_STAGE];  // row starts of the band (one per active row, <= V), then the cells' sizes (<= PLAN_PAR_CAP)
    // d0, d1: the first SP ranks of either (count pass).  A query of the release operating points touches a few hundred ranks; staging
    // all 2 x 4096 took 32 KB and held the kernel at two workgroups per CU -- it is bound by latency (barriers, dependent LDS and
    // global reads), three hide more of it.
    extern __shared__ __align__(16) unsigned char s_plan_dyn[];
    const int SP = V < PLAN_SP ? V : PLAN_SP;
    CT* s_d0 = reinterpret_cast<CT*>(s_plan_dyn);
    CT* s_d1 = s_d0 + SP;
    __shared__ int s_hist[PLAN_NB];  // the band's distribution sort: bucket sizes, then bucket starts
    __shared__ int64_t s_red[8];
    __shared__ int s_i[8];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long dbg_k0 = wall_clock64();
    (void)dbg_k0;
    const CT* d0 = sorted + ((int64_t)q * 2 + 0) * V;
    const CT* d1 = sorted + ((int64_t)q * 2 + 1) * V;
    const uint16_t* o0 = order + ((int64_t)q * 2 + 0) * V;
    const uint16_t* o1 = order + ((int64_t)q * 2 + 1) * V;
    uint64_t* ent = ent_list + (int64_t)q * ent_cap * 2;
    auto block_sum = [&](int64_t v) -> int64_t {  // every thread gets the sum
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if (lane == 0) s_red[wv] = v;
        __syncthreads();
        const int64_t t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        // the same value in every lane: say so (scalar registers, uniform branches on it)
        return ((int64_t)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t);
    };
    if constexpr (EMIT) {
        if (fallback[q]) return;  // the frontier walk emits this query
        const int n_ent = fallback[nq + q];
        const PlanOut pl = plan[q];
        const int64_t ibase = item_off[q], tbase = tab_off[q];
        // the ranks of either list that have a cell with candidates, as bits; a half table per set bit, numbered densely in rank order
        // (the count pass counted the same bits into ntab0 / ntab1)
        __shared__ uint32_t s_used[2 * PLAN_PAR_STAGE / 32];
        __shared__ uint16_t s_pre[2 * PLAN_PAR_STAGE / 32];
        constexpr int UW = PLAN_PAR_STAGE / 32;  // words per split
        s_used[tid] = 0u;
        __syncthreads();
        for (int idx = tid; idx < n_ent; idx += 256) {
            const uint64_t e0 = ent[2 * idx], e1 = ent[2 * idx + 1];
            if ((uint32_t)e1 > 0u) {
                const int bi = (int)(e0 >> 52), bj = (int)((e0 >> 40) & 0xfffu);
                atomicOr(&s_used[bi >> 5], 1u << (bi & 31));
                atomicOr(&s_used[UW + (bj >> 5)], 1u << (bj & 31));
            }
        }
        __syncthreads();
        {
            const int pc = __popc(s_used[tid]);
            int x = pc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            if (lane == 63) s_i[wv] = x;
            __syncthreads();
            s_pre[tid] = (uint16_t)(((wv & 1) ? s_i[wv - 1] : 0) + x - pc);  // waves 0, 1: split 0; waves 2, 3: split 1
        }
        __syncthreads();
        auto dense = [&](int split, int r) -> int {
            const int w = split * UW + (r >> 5);
            return (int)s_pre[w] + __popc(s_used[w] & ((1u << (r & 31)) - 1u));
        };
        // items: exclusive scan of the chunk counts of the listed cells, in visit order
        int run = 0;
        for (int b0 = 0; b0 < n_ent; b0 += 256) {
            const int idx = b0 + tid;
            int nch = 0, bi = 0, bj = 0, rank = 0;
            int64_t ls = 0, ll = 0;
            if (idx < n_ent) {
                const uint64_t e0 = ent[2 * idx], e1 = ent[2 * idx + 1];
                bi = (int)(e0 >> 52); bj = (int)((e0 >> 40) & 0xfffu);
                ls = (int64_t)(e0 & ((1ull << 40) - 1));
                ll = (int64_t)(uint32_t)e1;
                rank = (int)(e1 >> 32);
                nch = ll > 0 ? (int)((ll + seg_max - 1) / seg_max) : 0;
            }
            int x = nch;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            __syncthreads();
            if (lane == 63) s_i[wv] = x;
            __syncthreads();
            int base = run;
            for (int w = 0; w < wv; ++w) base += s_i[w];
            const int pos = base + x - nch;
            if (nch > 0) {
                const int64_t cell = (int64_t)o0[bi] * V + o1[bj];
                const int t0 = (int)tbase + dense(0, bi), t1 = (int)tbase + pl.ntab0 + dense(1, bj);
                for (int ch = 0; ch < nch; ++ch) {
                    WorkItem it;
                    it.q = q; it.rank = rank;
                    it.tab0 = t0;
                    it.tab1 = t1;
                    it.pos0 = ch * seg_max;
                    it.cell = (int)cell; it.pad = 0;
                    it.start = ls + (int64_t)ch * seg_max;
                    const int64_t rem = ll - (int64_t)ch * seg_max;
                    it.len = (int)(rem < seg_max ? rem : seg_max);
                    items[ibase + pos + ch] = it;
                }
            }
            run += s_i[0] + s_i[1] + s_i[2] + s_i[3];
        }
        // tables: a thread per rank of either list (a thread per word walking its bits ran the 16 tables of a V = 16 query one after the
        // other, each behind a load and an atomic's return)
        for (int x = tid; x < 2 * V; x += 256) {
            const int split = x >= V ? 1 : 0, r = x - split * V;
            if ((s_used[split * UW + (r >> 5)] >> (r & 31)) & 1u) {
                TabDesc td;
                td.q = q; td.pad = 0; td.split = split;
                td.cluster = split ? o1[r] : o0[r];
                const int ti = (int)tbase + (split ? pl.ntab0 : 0) + dense(split, r);
                tabs[ti] = td;
                const int g = (td.split * V + td.cluster) * GRP_SUB + (q % GRP_SUB);
                tab_order[grp_base[g] + atomicAdd(&grp_cur[g], 1)] = ti;
            }
        }
        return;
    } else {
        for (int i = tid; i < SP; i += 256) { s_d0[i] = d0[i]; s_d1[i] = d1[i]; }
        __syncthreads();
        // A thread owns the rows i = tid + 256 k and keeps three prefix lengths of each in registers: under the last band's tau
        // (`pprev`), under the bisection's lower end (`plo`: a tau below every later probe) and under its upper end (`phi`, exact).
        // A probe searches [plo, phi] only -- a step or two instead of log2 V -- and the band's enumeration needs no search at all.
        constexpr int KR = PLAN_PAR_STAGE / 256;
        int pprev[KR];
        uint32_t pbr[KR];  // plo | phi << 16 (prefix lengths <= 4096)
#pragma unroll
        for (int k = 0; k < KR; ++k) { pprev[k] = 0; pbr[k] = 0; }
        // first j of [lo_, hi_] with fl(a + d1[j]) > tau (hi_ when there is none below it)
        auto prefix_in = [&](CT a, uint64_t tau, int lo_, int hi_) -> int {
            while (lo_ < hi_) {
                const int mid = (lo_ + hi_) >> 1;
                if (f2bits((CT)(a + PL1(mid))) <= tau) lo_ = mid + 1;
                else hi_ = mid;
            }
            return lo_;
        };
        // the same over [lo_, V]: the staged part first
        auto prefix_from = [&](CT a, uint64_t tau, int lo_) -> int {
            if (SP < V && lo_ < SP) {
                if (f2bits((CT)(a + s_d1[SP - 1])) > tau) return prefix_in(a, tau, lo_, SP - 1);
                lo_ = SP;
            }
            return prefix_in(a, tau, lo_, V);
        };
        // rows with a cell under tau: first i with fl(d0[i] + d1[0]) > tau (every thread reads the same words: a uniform value)
        auto rows_under = [&](uint64_t tau) -> int {
            int lo_ = 0, hi_ = V;
            const CT b0 = s_d1[0];
            if (SP < V) {
                if (f2bits((CT)(s_d0[SP - 1] + b0)) > tau) hi_ = SP - 1;
                else lo_ = SP;
            }
            while (lo_ < hi_) {
                const int mid = (lo_ + hi_) >> 1;
                if (f2bits((CT)(PL0(mid) + b0)) <= tau) lo_ = mid + 1;
                else hi_ = mid;
            }
            return __builtin_amdgcn_readfirstlane(lo_);
        };
        // Bands of increasing tau: band b holds the cells with tau_{b-1} < s <= tau_b (at most PLAN_PAR_CAP of them), is
        // sorted on its own and appended to the visited list; the quota prefix sum carries over.
        PLAN_DBG(8, wall_clock64() - dbg_k0);
        bool fb = false, done = false;
        int visited = 0, ne_total = 0;
        int64_t cum = 0, c_prev = 0, target = 256;
        // first band: 0.6 x the cells a query of this quota visited in the previous launch (so that [target, 2 target] holds what most
        // queries need and ONE band is enumerated and sorted); 256 without a hint.  The hint only sizes the bands.
        if (hint && quota > 0) {
            const unsigned long long hv = hint[(hint_slot ^ 1) * 2], hq = hint[(hint_slot ^ 1) * 2 + 1];
            if (hq > 0) {
                const double t0 = 0.6 * (double)hv / (double)hq * (double)quota;
                target = t0 < 64.0 ? 64 : (t0 > (double)(PLAN_PAR_CAP / 2) ? PLAN_PAR_CAP / 2 : (int64_t)t0);
            }
        }
        bool have_prev = false;
        uint64_t tau_prev = 0;
        const int64_t all_cells = (int64_t)V * V;
        const uint64_t s_min = f2bits((CT)(PL0(0) + PL1(0)));
        if (quota <= 0) { target = 1; }  // the test follows the first append (search.py:131-132): one cell
        while (!done && !fb) {
            const int64_t left = all_cells - c_prev;
            if (left <= 0) break;  // every cell visited, quota not reached
            const int64_t want = target < left ? target : left;
            // tau with want <= #{tau_prev < s <= tau} <= 2 * want (or the smallest tau that reaches `want` when values repeat).
            // Upper end to start from: the a x a square of rank pairs lies under fl(d0[a-1] + d1[a-1]) (the sums are monotone in both
            // ranks), so with a * a >= c_prev + want that tau holds the band; it is ~2 x too large (the region under a tau is closer to
            // a triangle than a square), so a few probes finish -- and every probe sees few active rows (from s_max the first probes
            // searched all V rows).
            const uint64_t lo_key = have_prev ? tau_prev + 1 : s_min;
            uint64_t lo = lo_key, hi;
            {
                const int64_t need = c_prev + want;
                int64_t a = (int64_t)sqrt((double)need);
                while (a * a < need) ++a;
                while (a > 1 && (a - 1) * (a - 1) >= need) --a;
                if (a > V) a = V;
                hi = f2bits((CT)(PL0((int)a - 1) + PL1((int)a - 1)));
            }
            const long long dbg_t0 = wall_clock64();
            (void)dbg_t0;
            const int R0 = rows_under(hi);
            int64_t c_hi;
            {
                int64_t c = 0;
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    int ph = pprev[k];
                    if (k * 256 < R0) {
                        const int i = k * 256 + tid;
                        if (i < R0) ph = prefix_from(PL0(i), hi, pprev[k]);
                        c += ph;
                    }
                    pbr[k] = (uint32_t)pprev[k] | ((uint32_t)ph << 16);
                }
                c_hi = block_sum(c);
                PLAN_DBG(0, 1);
            }
            while (c_hi - c_prev > 2 * want && lo < hi) {
                const uint64_t mid = lo + ((hi - lo) >> 1);
                uint16_t pm[KR];
                int64_t c = 0;
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    pm[k] = (uint16_t)(pbr[k] & 0xffffu);
                    if (k * 256 < R0) {
                        const int i = k * 256 + tid;
                        if (i < R0) pm[k] = (uint16_t)prefix_in(PL0(i), mid, (int)(pbr[k] & 0xffffu), (int)(pbr[k] >> 16));
                        c += pm[k];
                    }
                }
                c = block_sum(c);
                PLAN_DBG(0, 1);
                if (c - c_prev >= want) {
                    hi = mid; c_hi = c;
#pragma unroll
                    for (int k = 0; k < KR; ++k) pbr[k] = (pbr[k] & 0xffffu) | ((uint32_t)pm[k] << 16);
                } else {
                    lo = mid + 1;
#pragma unroll
                    for (int k = 0; k < KR; ++k) pbr[k] = (pbr[k] & 0xffff0000u) | (uint32_t)pm[k];
                }
            }
            PLAN_DBG(1, 1);
            PLAN_DBG(2, wall_clock64() - dbg_t0);
            if (c_hi - c_prev > PLAN_PAR_CAP) { fb = true; break; }
            const int cnt = (int)(c_hi - c_prev);
            // enumerate the band.  (a) per row: cells [pprev, phi); s_gc[i] = first slot of the row | pprev << 12 (rows in order)
            const int rows = rows_under(hi);
            for (int x = tid; x < PLAN_NB; x += 256) s_hist[x] = 0;
            int run = 0;
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                if (k * 256 < rows) {
                    const int i = k * 256 + tid;
                    const int p = i < rows ? (int)(pbr[k] >> 16) - pprev[k] : 0;
                    int x = p;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const int y = __shfl_up(x, d);
                        if (lane >= d) x += y;
                    }
                    __syncthreads();
                    if (lane == 63) s_i[wv] = x;
                    __syncthreads();
                    int base = run;
                    for (int w = 0; w < wv; ++w) base += s_i[w];
                    if (i < rows) s_gc[i] = (uint32_t)(base + x - p) | ((uint32_t)pprev[k] << 12);
                    run += s_i[0] + s_i[1] + s_i[2] + s_i[3];
                }
            }
            __syncthreads();
            PLAN_DBG(5, wall_clock64() - dbg_t0);
            // (b) one thread per cell: row by binary search over the row starts (the LAST row whose start is <= e is the
            // one that holds e: empty rows share their start with the next row), then sum, rank pair and GLOBAL cell size
            constexpr int PER = PLAN_PAR_CAP / 256;
            // (the global reads of the thread's PER cells go out together, level by level -- cluster ids, then sizes: as
            // `if (e < cnt) { ... gcount[o0[i] * V + o1[j]] }` per cell they were 2 x PER round trips in a row, ~25 us per band)
            uint32_t eij[PER];
            uint64_t ekey[PER];
            int ei[PER], ej[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int e = r * 256 + tid;
                ei[r] = 0; ej[r] = 0; ekey[r] = 0; eij[r] = 0;
                if (e < cnt) {
                    int lo_ = 0, hi_ = rows;  // first row whose start is > e
                    while (lo_ < hi_) {
                        const int mid = (lo_ + hi_) >> 1;
                        if ((int)(s_gc[mid] & 0xfffu) <= e) lo_ = mid + 1;
                        else hi_ = mid;
                    }
                    const int i = lo_ - 1;
                    const uint32_t w = s_gc[i];
                    const int j = e - (int)(w & 0xfffu) + (int)(w >> 12);
                    ekey[r] = f2bits((CT)(PL0(i) + PL1(j)));
                    eij[r] = ((uint32_t)i << 16) | (uint32_t)j;
                    ei[r] = i; ej[r] = j;
                }
            }
            uint16_t ci[PER], cj[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) { ci[r] = o0[ei[r]]; cj[r] = o1[ej[r]]; }
            int64_t gg[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) gg[r] = gcount[(int64_t)ci[r] * V + cj[r]];
            // Sort by (s, i, j) as a distribution sort: the keys lie in (tau_prev, tau], a bucket is a slice of that range (the key
            // minus its lower end, shifted down to PLAN_NB values: monotone), a cell takes the next slot of its bucket (an LDS counter),
            // the buckets' sizes are scanned, and a cell's rank is its bucket's start + the cells of the bucket that order before it (a
            // cell or two per bucket; tied sums pile up in one bucket and are ordered by the rank pair there -- quadratic only in the
            // size of a tie group).  The bitonic network this replaces was 32-47 us of a band's ~80 us.
            int bk[PER], slot[PER];
            {
                const uint64_t range = hi - lo_key;
                const int nbits = range ? 64 - __builtin_clzll(range) : 0;
                const int shift = nbits > PLAN_NB_LOG ? nbits - PLAN_NB_LOG : 0;
#pragma unroll
                for (int r = 0; r < PER; ++r) {
                    const int e = r * 256 + tid;
                    bk[r] = 0; slot[r] = 0;
                    if (e < cnt) {
                        bk[r] = (int)((ekey[r] - lo_key) >> shift);
                        slot[r] = atomicAdd(&s_hist[bk[r]], 1);
                    }
                }
            }
            __syncthreads();
            {   // exclusive scan of the bucket sizes, in place (thread t: buckets [t * NBT, (t + 1) * NBT))
                constexpr int NBT = PLAN_NB / 256;
                int hb[NBT], sum = 0;
#pragma unroll
                for (int u = 0; u < NBT; ++u) { hb[u] = s_hist[tid * NBT + u]; sum += hb[u]; }
                int x = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int y = __shfl_up(x, d);
                    if (lane >= d) x += y;
                }
                if (lane == 63) s_i[wv] = x;
                __syncthreads();
                int base = x - sum;
                for (int w = 0; w < wv; ++w) base += s_i[w];
#pragma unroll
                for (int u = 0; u < NBT; ++u) { s_hist[tid * NBT + u] = base; base += hb[u]; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int e = r * 256 + tid;
                if (e < cnt) {
                    const int pos = s_hist[bk[r]] + slot[r];
                    s_key[pos] = (typename PlanKeyT<CT>::type)ekey[r]; s_ij[pos] = eij[r];
                }
            }
            __syncthreads();
            int rk[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int e = r * 256 + tid;
                rk[r] = 0;
                if (e < cnt) {
                    const int bb = s_hist[bk[r]], be = bk[r] + 1 < PLAN_NB ? s_hist[bk[r] + 1] : cnt;
                    int rank = bb;
                    for (int p = bb; p < be; ++p) {
                        const uint64_t k2 = (uint64_t)s_key[p];
                        const uint32_t i2 = s_ij[p];
                        rank += (k2 < ekey[r] || (k2 == ekey[r] && i2 < eij[r])) ? 1 : 0;
                    }
                    rk[r] = rank;
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int e = r * 256 + tid;
                if (e < cnt) {
                    s_key[rk[r]] = (typename PlanKeyT<CT>::type)ekey[r]; s_ij[rk[r]] = eij[r];
                    s_gc[rk[r]] = gg[r] > 0x7fffffffll ? 0x7fffffffu : (uint32_t)gg[r];
                }
            }
            __syncthreads();
            PLAN_DBG(6, wall_clock64() - dbg_t0);
            PLAN_DBG(7, wall_clock64() - dbg_t0);
            // quota cut inside this band: first position whose inclusive prefix of the cell sizes reaches the quota
            int64_t run64 = cum;
            int cut = -1;
            for (int b0 = 0; b0 < cnt && cut < 0; b0 += 256) {
                const int idx = b0 + tid;
                int64_t x = idx < cnt ? (int64_t)s_gc[idx] : 0;
                const int64_t own = x;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int64_t y = __shfl_up(x, d);
                    if (lane >= d) x += y;
                }
                __syncthreads();
                if (lane == 63) s_red[wv] = x;
                if (tid == 0) s_i[4] = 0x7fffffff;
                __syncthreads();
                int64_t base = run64;
                for (int w = 0; w < wv; ++w) base += s_red[w];
                const int64_t incl = base + x;
                if (idx < cnt && incl >= quota && incl - own < quota) atomicMin(&s_i[4], idx);
                const int64_t tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
                __syncthreads();
                if (s_i[4] != 0x7fffffff) cut = s_i[4];
                run64 += tot;
                __syncthreads();
            }
            PLAN_DBG(3, wall_clock64() - dbg_t0);
            PLAN_DBG(4, cnt);
            if (quota <= 0) cut = 0;
            const int take = cut >= 0 ? cut + 1 : cnt;
            if (ne_total + take > ent_cap) { fb = true; break; }
            // the cells of the band that hold anything (size over all shards > 0: the visited list is mostly empty cells at thousands
            // of coarse clusters) are appended to the query's list with their visit rank
            for (int b0 = 0; b0 < take; b0 += 256) {
                const int idx = b0 + tid;
                const int f = (idx < take && s_gc[idx] > 0u) ? 1 : 0;
                int x = f;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int y = __shfl_up(x, d);
                    if (lane >= d) x += y;
                }
                __syncthreads();
                if (lane == 63) s_i[wv] = x;
                __syncthreads();
                int base = ne_total;
                for (int w = 0; w < wv; ++w) base += s_i[w];
                if (f) {
                    const uint32_t ij = s_ij[idx];
                    const int64_t ep = (int64_t)(base + x - 1) * 2;
                    ent[ep] = (uint64_t)(((ij >> 16) << 12) | (ij & 0xfffu)) << 40;
                    ent[ep + 1] = (uint64_t)(uint32_t)(visited + idx) << 32;
                }
                ne_total += s_i[0] + s_i[1] + s_i[2] + s_i[3];
            }
            visited += take;
            if (cut >= 0) { done = true; break; }
            cum = run64;
            c_prev = c_hi;
            tau_prev = hi;
            have_prev = true;
#pragma unroll
            for (int k = 0; k < KR; ++k) pprev[k] = (int)(pbr[k] >> 16);
            // the next band: the cells the quota still needs at the candidates per cell seen so far, + 25 % (round 4).  Four times the
            // last target made the second band of a V = 2048 query 1024 ... 2048 cells when ~300 more were needed: the band's
            // enumeration and its sort (n log^2 n) were most of the count pass (tools/build_variant.sh plandbg -DCIS_PLAN_DBG).
            // The bands' boundaries do not change what is visited.
            {
                int64_t nxt = target * 4;
                if (cum > 0 && quota > cum) {
                    const int64_t need = ((quota - cum) * (int64_t)visited + cum - 1) / cum;
                    nxt = need + need / 4 + 16;
                }
                nxt = nxt < 64 ? 64 : nxt;
                target = nxt < PLAN_PAR_CAP / 2 ? nxt : PLAN_PAR_CAP / 2;
            }
            __syncthreads();
        }
        PLAN_DBG(9, wall_clock64() - dbg_k0);
        if (tid == 0) fallback[q] = fb ? 1 : 0;
        if (fb) return;
        if (tid == 0 && hint && quota > 0) {
            atomicAdd(&hint[hint_slot * 2], (unsigned long long)visited);
            atomicAdd(&hint[hint_slot * 2 + 1], (unsigned long long)quota);
        }
        __threadfence_block();
        __syncthreads();
        // the listed cells' own starts and lengths (this shard's), written back into the list: the emit pass reads nothing else.  The ranks
        // of either list that have a cell with candidates are bits (they alias the sort's bucket counters): a half table per set bit.
        constexpr int UW = PLAN_PAR_STAGE / 32;
        uint32_t* s_used = reinterpret_cast<uint32_t*>(s_hist);
        s_used[tid] = 0u;
        __syncthreads();
        int64_t n_items = 0, ncand = 0;
        for (int b0 = 0; b0 < ne_total; b0 += 4 * 256) {  // four cells per thread: their reads go out together, level by level
            uint64_t e0[4], e1[4];
            int64_t vc[4], l0[4], l1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = b0 + u * 256 + tid;
                const int64_t ep = (int64_t)(idx < ne_total ? idx : 0) * 2;
                e0[u] = ent[ep]; e1[u] = ent[ep + 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) vc[u] = (int64_t)o0[(int)(e0[u] >> 52)] * V + o1[(int)((e0[u] >> 40) & 0xfffu)];
#pragma unroll
            for (int u = 0; u < 4; ++u) { l0[u] = loff[vc[u]]; l1[u] = loff[(int64_t)V * V + 1 + vc[u]]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = b0 + u * 256 + tid;
                if (idx < ne_total) {
                    const int64_t ll = l1[u] - l0[u];
                    ent[(int64_t)idx * 2] = e0[u] | (uint64_t)l0[u];
                    ent[(int64_t)idx * 2 + 1] = e1[u] | (uint64_t)(uint32_t)(ll > 0 ? ll : 0);
                    if (ll > 0) {
                        const int bi = (int)(e0[u] >> 52), bj = (int)((e0[u] >> 40) & 0xfffu);
                        n_items += (ll + seg_max - 1) / seg_max;
                        ncand += ll;
                        atomicOr(&s_used[bi >> 5], 1u << (bi & 31));
                        atomicOr(&s_used[UW + (bj >> 5)], 1u << (bj & 31));
                    }
                }
            }
        }
        n_items = block_sum(n_items);
        ncand = block_sum(ncand);
        const uint32_t ubits = s_used[tid];
        const int64_t ntabs = block_sum(tid < UW ? (int64_t)__popc(ubits) : ((int64_t)__popc(ubits) << 32));
        PLAN_DBG(10, wall_clock64() - dbg_k0);
        if (tid == 0) {
            PlanOut p;
            p.visited = visited; p.n_items = (int)n_items; p.ntab0 = (int)(uint32_t)ntabs; p.ntab1 = (int)(ntabs >> 32); p.ncand = ncand;
            plan[q] = p;
            fallback[nq + q] = ne_total;
        }
        for (int x = tid; x < 2 * V; x += 256) {  // (a thread per rank: see the emit pass)
            const int split = x >= V ? 1 : 0, r = x - split * V;
            if ((s_used[split * UW + (r >> 5)] >> (r & 31)) & 1u) {
                const int g = split ? V + (int)o1[r] : (int)o0[r];
                atomicAdd(&grp_cnt[g * GRP_SUB + (q % GRP_SUB)], 1);
            }
        }
        PLAN_DBG(11, wall_clock64() - dbg_k0);
    }
}

// multisequence as a list: the first `max_cells` (dist, cell) pairs of every query, same frontier walk as k_plan
template <typename CT>
__global__ __launch_bounds__(64) void k_multiseq_list(const CT* __restrict__ sorted, const uint16_t* __restrict__ order, int V,
                                                      int max_cells, int32_t* __restrict__ cells, double* __restrict__ dists) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* t = reinterpret_cast<int*>(smem);
    const int q = blockIdx.x, lane = threadIdx.x;
    const CT* d0 = sorted + ((int64_t)q * 2 + 0) * V;
    const CT* d1 = sorted + ((int64_t)q * 2 + 1) * V;
    const uint16_t* o0 = order + ((int64_t)q * 2 + 0) * V;
    const uint16_t* o1 = order + ((int64_t)q * 2 + 1) * V;
    for (int i = lane; i < V; i += 64) t[i] = 0;
    __syncthreads();
    int rows = 1;
    for (int n = 0; n < max_cells; ++n) {
        uint64_t bk = ~0ull;
        uint32_t bij = ~0u;
        for (int i = lane; i < rows; i += 64) {
            const int j = t[i];
            if (j >= V) continue;
            if (i > 0 && t[i - 1] <= j) continue;
            const uint64_t kb = f2bits((CT)(d0[i] + d1[j]));
            const uint32_t ij = ((uint32_t)i << 16) | (uint32_t)j;
            if (kb < bk || (kb == bk && ij < bij)) { bk = kb; bij = ij; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint64_t ok = __shfl_xor(bk, off);
            const uint32_t oij = __shfl_xor(bij, off);
            if (ok < bk || (ok == bk && oij < bij)) { bk = ok; bij = oij; }
        }
        if (bij == ~0u) break;
        const int bi = (int)(bij >> 16), bj = (int)(bij & 0xffff);
        if (lane == 0) {
            cells[((int64_t)q * max_cells + n) * 2 + 0] = o0[bi];
            cells[((int64_t)q * max_cells + n) * 2 + 1] = o1[bj];
            dists[(int64_t)q * max_cells + n] = (double)(CT)(d0[bi] + d1[bj]);
        }
        __syncthreads();
        if (lane == 0) t[bi] = bj + 1;
        if (bi + 2 > rows) rows = (bi + 2 < V) ? bi + 2 : V;
        __syncthreads();
    }
}

// The table-group part of k_plan_scan for wide vocabularies: 2 V x 32 counters are 262144 words at V = 4096.  One tile of 1024 counters per
// workgroup (16-byte loads); a tile publishes its sum tagged with the batch's sequence number and takes as its base the sum of the tiles
// before it, each waited for by one thread -- every tile publishes before it waits, and tiles are dispatched in order, so nothing can wait
// for a tile that has not started.  (Round 4: one workgroup, 16384 counters per round, 0.46 ms at V = 4096 on the batch's critical path.)
static const int GROUP_TILE = 1024;
__global__ __launch_bounds__(256) void k_group_bases(int* __restrict__ grp_cnt, int* __restrict__ grp_base, int n_groups,
                                                     unsigned long long* __restrict__ agg /* [tiles] tag << 32 | sum */, uint32_t tag) {
    __shared__ int s_w[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, t = blockIdx.x;
    const int g = t * GROUP_TILE + tid * 4;
    int c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = g + i < n_groups ? grp_cnt[g + i] : 0;
    const int tot = c[0] + c[1] + c[2] + c[3];
    int x = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_w[wv] = x;
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(&agg[t], ((unsigned long long)tag << 32) | (unsigned long long)(uint32_t)(s_w[0] + s_w[1] + s_w[2] + s_w[3]),
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int pre = 0;
    for (int p = tid; p < t; p += 256) {
        unsigned long long v;
        do { v = __hip_atomic_load(&agg[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((uint32_t)(v >> 32) != tag);
        pre += (int)(uint32_t)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o);
    if (lane == 0) s_w[4 + wv] = pre;
    __syncthreads();
    int r = s_w[4] + s_w[5] + s_w[6] + s_w[7] + x - tot;
    for (int w = 0; w < wv; ++w) r += s_w[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (g + i < n_groups) {
            grp_base[g + i] = r;
            grp_cnt[g + i] = 0;              // as k_plan_scan: clean counters for the next batch, clean cursors for this batch's emit pass
            grp_cnt[n_groups + g + i] = 0;
        }
        r += c[i];
    }
}

// exclusive scans over the queries of one batch (single block of 1024 threads); totals[0]=items, [1]=tables, [2]=cands.
// Thread t owns queries t, t + 1024, ... (<= 8 rounds for a batch of 8192): every load and store of a wave is
// contiguous -- with eight consecutive queries per thread the wave touched 64 cache lines per instruction and this
// one-CU kernel took 21 us.  Round r, wave w: inclusive wave scans of all rounds at once, wave totals through LDS.
__global__ __launch_bounds__(1024) void k_plan_scan(const PlanOut* __restrict__ plan, int nq, int64_t* __restrict__ item_off,
                                                    int64_t* __restrict__ tab_off, int64_t* __restrict__ totals,
                                                    unsigned long long* __restrict__ qbound /* [nq] -> +inf */,
                                                    volatile int64_t* __restrict__ host_totals /* pinned, mapped */, int64_t seq,
                                                    int* __restrict__ grp_cnt /* read, then zeroed: the next batch's count pass finds it clean */,
                                                    int* __restrict__ grp_base, int n_groups,
                                                    unsigned long long* __restrict__ hint_zero /* null, or the two words k_plan_par's NEXT launch adds into */) {
    if (hint_zero && threadIdx.x == 0) { hint_zero[0] = 0ull; hint_zero[1] = 0ull; }
    constexpr int R = 8;  // rounds held in registers; more queries than 8192 take the slow tail loop below
    __shared__ int s_wi[R][16], s_wt[R][16];  // wave totals per round
    __shared__ int64_t s_cand[16];
    __shared__ int64_t s_base_i[R][16], s_base_t[R][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (wv == 15 && n_groups > 0) {  // exclusive scan of the table-group counters by one wave: 64 x 16 at a time, loads issued together
        // (n_groups == 0: k_group_bases did it -- thousands of coarse clusters)
        int run = 0;
        for (int g0 = 0; g0 < n_groups; g0 += 1024) {
            int c[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int g = g0 + lane * 16 + i;
                c[i] = g < n_groups ? grp_cnt[g] : 0;
            }
            int tot = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) tot += c[i];
            int x = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            int r = run + x - tot;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int g = g0 + lane * 16 + i;
                if (g < n_groups) {
                    grp_base[g] = r;
                    grp_cnt[g] = 0;              // counters: clean for the next batch's count pass (k_front_small does not zero them)
                    grp_cnt[n_groups + g] = 0;   // cursors (grp_cur = grp_cnt + n_groups): clean for this batch's emit pass
                }
                r += c[i];
            }
            run += __shfl(x, 63);
        }
    }
    int ni[R], nt[R], xi[R], xt[R];
    int64_t lc = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = r * 1024 + tid;
        const bool on = q < nq;
        const PlanOut pl = plan[on ? q : 0];
        ni[r] = on ? pl.n_items : 0;
        nt[r] = on ? pl.ntab0 + pl.ntab1 : 0;
        lc += on ? pl.ncand : 0;
        xi[r] = ni[r]; xt[r] = nt[r];
    }
    for (int q = R * 1024 + tid; q < nq; q += 1024) lc += plan[q].ncand;  // batches above 8192 queries (not used today)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int yi = __shfl_up(xi[r], d), yt = __shfl_up(xt[r], d);
            if (lane >= d) { xi[r] += yi; xt[r] += yt; }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) lc += __shfl_xor(lc, d);
    if (lane == 63) {
#pragma unroll
        for (int r = 0; r < R; ++r) { s_wi[r][wv] = xi[r]; s_wt[r][wv] = xt[r]; }
    }
    if (lane == 0) s_cand[wv] = lc;
    __syncthreads();
    if (wv == 0) {  // exclusive scan over the R x 16 (round, wave) totals: two consecutive entries per lane
        const int e0 = 2 * lane, e1 = 2 * lane + 1;
        const int a_i = (&s_wi[0][0])[e0], b_i = (&s_wi[0][0])[e1], a_t = (&s_wt[0][0])[e0], b_t = (&s_wt[0][0])[e1];
        int64_t xi2 = (int64_t)a_i + b_i, xt2 = (int64_t)a_t + b_t;
        const int64_t own_i = xi2, own_t = xt2;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t yi = __shfl_up(xi2, d), yt = __shfl_up(xt2, d);
            if (lane >= d) { xi2 += yi; xt2 += yt; }
        }
        (&s_base_i[0][0])[e0] = xi2 - own_i; (&s_base_i[0][0])[e1] = xi2 - own_i + a_i;
        (&s_base_t[0][0])[e0] = xt2 - own_t; (&s_base_t[0][0])[e1] = xt2 - own_t + a_t;
        int64_t ri = __shfl(xi2, 63), rt = __shfl(xt2, 63);
        int64_t rc = lane < 16 ? s_cand[lane] : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) rc += __shfl_xor(rc, d);
        if (lane == 0) {
            for (int q = R * 1024; q < nq; ++q) {  // queries beyond R * 1024 (slow path): sequential
                item_off[q] = ri; tab_off[q] = rt;
                qbound[q] = 0x7ff0000000000000ull;
                ri += plan[q].n_items; rt += plan[q].ntab0 + plan[q].ntab1;
            }
            totals[0] = ri; totals[1] = rt; totals[2] = rc;
            host_totals[0] = ri; host_totals[1] = rt; host_totals[2] = rc;  // straight into pinned host memory: no staged copy
            __threadfence_system();
            host_totals[3] = seq;  // the host polls this word (the totals above are visible before it)
            __threadfence_system();
            item_off[nq] = ri; tab_off[nq] = rt;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = r * 1024 + tid;
        if (q < nq) {
            item_off[q] = s_base_i[r][wv] + xi[r] - ni[r];
            tab_off[q] = s_base_t[r][wv] + xt[r] - nt[r];
            qbound[q] = 0x7ff0000000000000ull;
        }
    }
}

// one launch instead of three memsets: queue counters and per-cell counters to zero, slots to -1 (empty)
__global__ void k_slots_init(int* __restrict__ qctr16, int* __restrict__ cell_cnt, int ncells, int* __restrict__ slots,
                             int64_t n_slot_entries) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 64) qctr16[i] = 0;  // queue counters, slot counts, debug counters, the fall-back header of the sampled scan
    if (i < ncells) cell_cnt[i] = 0;
    if (i < n_slot_entries) slots[i] = -1;
}

// Sort key of a work item.  The slot list is cut into eight queues (one per XCD) by cell range; inside a queue the
// items of every query's FIRST visited cell come first, sorted by cell, then all the others, sorted by cell: the
// first cell usually holds the best candidates, so by the time the other cells of a query are scanned its bound
// (qbound) is already tight and they run the hot loop only.
static __device__ __forceinline__ int q8_begin(int x, int ncells) { return (int)(((int64_t)x * ncells + 7) / 8); }
// CH keys per (cell, first / other): the chunks of a cell longer than one chunk get their own slots (round 3; with one key per cell
// such items shared slots and ran as sub-slots, one chunk after the other, inside one workgroup)
static __device__ __forceinline__ int slot_key(const WorkItem& it, int ncells, int CH, int seg_max) {
    const int x = (int)(((int64_t)it.cell * 8) / ncells);
    const int b = q8_begin(x, ncells), sz = q8_begin(x + 1, ncells) - b;
    int ch = CH > 1 ? it.pos0 / seg_max : 0;
    ch = ch < CH ? ch : CH - 1;
    return (2 * b + (it.rank > 0 ? sz : 0) + (it.cell - b)) * CH + ch;
}

__global__ void k_item_hist(const WorkItem* __restrict__ items, int64_t n, int* __restrict__ cell_cnt, int ncells, int CH, int seg_max,
                            const int64_t* __restrict__ d_totals = nullptr /* the plan totals: n is a bound */) {
    if (d_totals) n = d_totals[0];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cell_cnt[slot_key(items[i], ncells, CH, seg_max)], 1);
}

// counts -> exclusive SLOT offsets per cell (a slot holds up to G items of one cell); the counts are
// reset to zero so that the scatter can reuse them as cursors.  *n_slots = total number of slots.
__global__ __launch_bounds__(1024) void k_cell_scan(int* __restrict__ cell_cnt, int* __restrict__ slot_off, int nkeys, int G,
                                                    int* __restrict__ n_slots, int* __restrict__ qstart /* [9] first slot of every queue */, int CH) {
    // exclusive scan of the slot counts in key order: rounds of 1024 consecutive keys (coalesced), wave scans + one LDS hop per
    // round (the first version gave every thread a run of consecutive keys and walked it load by load: 23 us at 8192 keys)
    __shared__ int s_w[16];
    __shared__ int s_tot;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int run = 0;
    for (int c0 = 0; c0 < nkeys; c0 += 1024) {
        const int c = c0 + tid;
        const int x = c < nkeys ? (cell_cnt[c] + G - 1) / G : 0;
        int inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(inc, d);
            if (lane >= d) inc += y;
        }
        __syncthreads();
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        int base = run, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int v = s_w[w];
            if (w < wv) base += v;
            tot += v;
        }
        if (c < nkeys) {
            slot_off[c] = base + inc - x;
            cell_cnt[c] = 0;
        }
        run += tot;
    }
    if (tid == 0) { *n_slots = run; s_tot = run; }
    __threadfence();
    __syncthreads();
    if (tid < 8) {  // queue x starts at the first key of its cell range
        const int k0 = 2 * q8_begin(tid, nkeys / (2 * CH)) * CH;
        qstart[tid] = k0 < nkeys ? __hip_atomic_load(&slot_off[k0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s_tot;
    }
    if (tid == 8) qstart[8] = s_tot;
}

// slots[(slot_off[cell] + r / G) * G + r % G] = item, r = arrival rank of the item inside its cell
__global__ void k_item_scatter(const WorkItem* __restrict__ items, int64_t n, const int* __restrict__ slot_off,
                               int* __restrict__ cursor, int G, int* __restrict__ slots, int ncells, int CH, int seg_max,
                               const int64_t* __restrict__ d_totals = nullptr /* the plan totals: n is a bound */) {
    if (d_totals) n = d_totals[0];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = slot_key(items[i], ncells, CH, seg_max);
    const int r = atomicAdd(&cursor[c], 1);
    slots[(slot_off[c] + r / G) * G + (r % G)] = (int)i;
}

// no sorting (huge V): slot i = item i alone
__global__ void k_identity_slots(int64_t n, int G, int* __restrict__ slots, int* __restrict__ n_slots, int* __restrict__ qstart) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        slots[i * G] = (int)i;
        for (int g = 1; g < G; ++g) slots[i * G + g] = -1;
    }
    if (i == 0) *n_slots = (int)n;
    if (i < 9) qstart[i] = (int)((n * i) / 8);  // equal eighths
}

// ================================================================================================
// kernel: ADC tables  (lopq/lopq/model.py:673-704 for one (query, split, coarse id))
// ================================================================================================
template <typename CT>
__global__ __launch_bounds__(256) void k_tables(const CT* __restrict__ X /* [nq][D] */, const CT* __restrict__ Cs,
                                                const double* __restrict__ Rt, const double* __restrict__ mus,
                                                const double* __restrict__ subs, const TabDesc* __restrict__ tabs,
                                                int V, int h, int w, int nf, int K, int D,
                                                double* __restrict__ T /* [ntab][nf][K] */, PwProg prog_w,
                                                double* __restrict__ px_out /* [ntab][h] or null */,
                                                const int* __restrict__ tab_order, int first) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* v = reinterpret_cast<double*>(smem);  // [h]
    double* px = v + h;                           // [h]
    const int tab = tab_order ? tab_order[first + blockIdx.x] : first + (int)blockIdx.x;
    const TabDesc td = tabs[tab];
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
    // px[i] = sum_k R[k][i] v[k] as ONE chain of fused multiply-adds, k ascending (round 4: the arithmetic of
    // v_mfma_f64_16x16x4_f64, which k_tables_group_mfma runs the grouped tables on -- tools/probes/mfma_f64_order.hip; the
    // reference's BLAS order is unspecified anyway).  Every route to px uses this order: the routes agree bit for bit.
    for (int i = threadIdx.x; i < h; i += blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < h; ++k) acc = fma(R[(int64_t)k * h + i], v[k], acc);
        px[i] = acc;
    }
    __syncthreads();
    if (px_out) {  // two-kernel path: the distance tables are built by k_tables_from_px
        for (int i = threadIdx.x; i < h; i += blockDim.x) px_out[(int64_t)tab * h + i] = px[i];
        return;
    }
    double* out = T + (int64_t)tab * nf * K;
    for (int e = threadIdx.x; e < nf * K; e += blockDim.x) {
        const int j = e / K, k = e % K;
        const double* sc = subs + ((int64_t)(s * nf + j) * K + k) * w;
        const double* f = px + j * w;
        auto elem = [&](int i) -> double { const double df = f[i] - sc[i]; return df * df; };
        out[e] = pw_sum<double>(prog_w, elem);
    }
}

// Second half of the table build for the common sub-vector widths: one block = (chunk of 64 table
// items, fine split j, coarse split z).  Thread k keeps sub-centroid (z, j, k) in registers and walks
// the chunk, so the 32 KB sub-quantizer is read once per 64 tables instead of once per table (the
// one-kernel version was bound by L2 reads of the sub-quantizers: 160 KB per table).
// The projection px = R[c] . ((x - C[c]) - mu[c]) for TB tables of ONE (split, cluster) group per block: the tables
// arrive grouped by cluster (tab_order), so the 8*h*h bytes of R[c] are read once per TB tables instead of once per
// table (k_tables is bound by exactly that L2 traffic).  Same arithmetic as k_tables, table by table: one chain of fused
// multiply-adds over ascending k.  h <= 256, two-kernel path only (px_out).  The vector-unit form: used when h % 16 != 0
// (or CIS_TABLES_VALU=1); k_tables_group_mfma below is the default.
template <typename CT, int TB>
__global__ __launch_bounds__(256) void k_tables_group(const CT* __restrict__ X, const CT* __restrict__ Cs,
                                                      const double* __restrict__ Rt, const double* __restrict__ mus,
                                                      const TabDesc* __restrict__ tabs, const int* __restrict__ tab_order,
                                                      int n_tabs, int V, int h, int D, double* __restrict__ px_out,
                                                      const int64_t* __restrict__ d_totals /* null, or the plan totals: n_tabs is a bound */,
                                                      float* __restrict__ px32_out /* null, or a float32 copy of px (k_tiny_select's prefilter) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (d_totals) {
        n_tabs = (int)d_totals[1];
        if ((int)blockIdx.x * TB >= n_tabs) return;
    }
    double* v = reinterpret_cast<double*>(smem);  // [TB][h]
    double* psum = v + TB * h;                     // [parts][TB][h]
    __shared__ int s_tab[TB];
    __shared__ int s_same;
    const int tid = threadIdx.x;
    const int first = blockIdx.x * TB;
    const int nt = (n_tabs - first < TB) ? (n_tabs - first) : TB;
    if (tid < TB) s_tab[tid] = tid < nt ? tab_order[first + tid] : -1;
    __syncthreads();
    const TabDesc td0 = tabs[s_tab[0]];
    if (tid == 0) {
        int same = 1;
        for (int t = 1; t < nt; ++t) {
            const TabDesc td = tabs[s_tab[t]];
            same &= (td.split == td0.split && td.cluster == td0.cluster);
        }
        s_same = same;
    }
    for (int e = tid; e < nt * h; e += 256) {
        const int t = e / h, k = e - t * h;
        const TabDesc td = tabs[s_tab[t]];
        const CT* x = X + (int64_t)td.q * D + td.split * h;
        const CT* Cc = Cs + ((int64_t)td.split * V + td.cluster) * h;
        const double* mu = mus + ((int64_t)td.split * V + td.cluster) * h;
        const CT res = x[k] - Cc[k];  // rounds in CT (float32 when both are float32), model.py:635
        v[t * h + k] = (double)res - mu[k];
    }
    __syncthreads();
    // one chain of fused multiply-adds per (table, output), k ascending (see k_tables): thread group g = tid / h takes the tables
    // t = g, g + parts, ...; sixteen elements of the thread's column of R in flight at a time
    const int parts = 256 / h;
    const int part = tid / h, i = tid - part * h;
    if (part < parts) {
        if (s_same) {
            const double* R = Rt + ((int64_t)td0.split * V + td0.cluster) * h * h;
            double acc[TB];
#pragma unroll
            for (int t = 0; t < TB; ++t) acc[t] = 0.0;
            for (int kb = 0; kb < h; kb += 16) {
                double rr[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) rr[u] = R[(int64_t)(kb + u < h ? kb + u : h - 1) * h + i];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (kb + u < h) {
#pragma unroll
                        for (int t = 0; t < TB; ++t)
                            if (t % parts == part) acc[t] = fma(rr[u], v[t * h + kb + u], acc[t]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TB; ++t)
                if (t % parts == part && t < nt) psum[t * h + i] = acc[t];
        } else {  // a block that straddles two groups: every table with its own R
            for (int t = part; t < nt; t += parts) {
                const TabDesc td = tabs[s_tab[t]];
                const double* R = Rt + ((int64_t)td.split * V + td.cluster) * h * h;
                double acc = 0.0;
                for (int k = 0; k < h; ++k) acc = fma(R[(int64_t)k * h + i], v[t * h + k], acc);
                psum[t * h + i] = acc;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < nt * h; e += 256) {
        const int t = e / h, ii = e - t * h;
        const double acc = psum[t * h + ii];
        px_out[(int64_t)s_tab[t] * h + ii] = acc;
        if (px32_out) px32_out[(int64_t)s_tab[t] * h + ii] = (float)acc;
    }
}

// The grouped projection on the float64 matrix cores (round 4).  The VALU form above reads one broadcast LDS operand per fused
// multiply-add and waits for it (195 v_fmac_f64, 140 ds_read, 185 s_waitcnt in its listing): 1.7 ms for the 1.2 M tables of a
// V = 2048 batch, 7 % of the float64 rate.  v_mfma_f64_16x16x4_f64 takes ONE operand pair per lane for 1024 multiply-adds and is, per
// output element, exactly the chain of fused multiply-adds over ascending k that k_tables computes (tools/probes/mfma_f64_order.hip:
// 512000 results, none differs) -- so this kernel changes no bit.  A block = up to 16 tables of one (split, cluster) group: A = R
// (rows = outputs i, 16 per tile; loaded straight from global memory, 128 contiguous bytes per k), B = the tables' residuals v
// (columns = tables; staged in LDS with a pitch of h + 4), wave w owns the output tiles w, w + 4, ...  h % 16 == 0.
template <typename CT, int CH /* MFMA steps whose A operands are in flight together: 16 when h % 64 == 0, else 4 */>
__global__ __launch_bounds__(256) void k_tables_group_mfma(const CT* __restrict__ X, const CT* __restrict__ Cs,
                                                           const double* __restrict__ Rt, const double* __restrict__ mus,
                                                           const TabDesc* __restrict__ tabs, const int* __restrict__ tab_order,
                                                           int n_tabs, int V, int h, int D, double* __restrict__ px_out,
                                                           const int64_t* __restrict__ d_totals, float* __restrict__ px32_out) {
    typedef double f64x4_t __attribute__((ext_vector_type(4)));
    constexpr int TB = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (d_totals) {
        n_tabs = (int)d_totals[1];
        if ((int)blockIdx.x * TB >= n_tabs) return;
    }
    const int pitch = h + 4;
    double* v = reinterpret_cast<double*>(smem);  // [TB][pitch]
    double* so = v + TB * pitch;                  // [TB][h + 2]: the results, so that a table's h outputs leave as one contiguous run
    const int opitch = h + 2;
    __shared__ int s_tab[TB];
    __shared__ TabDesc s_td[TB];
    __shared__ int s_first[TB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = blockIdx.x * TB;
    const int nt = (n_tabs - first < TB) ? (n_tabs - first) : TB;
    if (tid < TB) {
        const int tb = tab_order[first + (tid < nt ? tid : 0)];  // (columns past the block's tables: the first table again, masked below)
        s_tab[tid] = tid < nt ? tb : -1;
        s_td[tid] = tabs[tb];
    }
    __syncthreads();
    // s_first[t]: the first table of the block with table t's (split, cluster) -- a block that straddles groups runs one pass per group
    if (tid < TB) {
        int f = tid;
        for (int u = tid - 1; u >= 0; --u)
            if (s_td[u].split == s_td[tid].split && s_td[u].cluster == s_td[tid].cluster) f = u;
        s_first[tid] = tid < nt ? f : -1;
    }
    // residuals of the block's tables -> LDS; four elements per thread in flight (descriptors from LDS, loads unconditional)
    for (int e0 = tid; e0 < TB * h; e0 += 1024) {
        CT xv[4], cv[4];
        double mv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u < TB * h ? e0 + 256 * u : tid;
            const int t = e / h, k = e - t * h;
            const TabDesc td = s_td[t];
            xv[u] = X[(int64_t)td.q * D + td.split * h + k];
            cv[u] = Cs[((int64_t)td.split * V + td.cluster) * h + k];
            mv[u] = mus[((int64_t)td.split * V + td.cluster) * h + k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            if (e < TB * h) {
                const int t = e / h, k = e - t * h;
                const CT res = xv[u] - cv[u];  // rounds in CT (float32 when both are float32), model.py:635
                v[t * pitch + k] = t < nt ? (double)res - mv[u] : 0.0;  // columns past the block's tables: zeros (not stored)
            }
        }
    }
    __syncthreads();
    const int t = lane & 15, kq = lane >> 4;
    for (int t0 = 0; t0 < nt; ++t0) {
        if (s_first[t0] != t0) continue;  // (uniform) one pass per (split, cluster) group present in the block: one as a rule
        const TabDesc tdg = s_td[t0];
        const double* R = Rt + ((int64_t)tdg.split * V + tdg.cluster) * h * h;
        const bool mine = t < nt && s_first[t] == t0;  // columns of other groups are computed along and not stored
        for (int it = wave; it < h / 16; it += 4) {
            f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
            const double* Ra = R + (int64_t)kq * h + it * 16 + (lane & 15);  // A[row = i][k = 4 s + kq] = R[k][i]
            const double* vb = v + t * pitch + kq;                           // B[k = 4 s + kq][col = t] = v[t][k]
            for (int s0 = 0; s0 < h / 4; s0 += CH) {  // (h / 4) % CH == 0
                double ra[CH], rb[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) ra[u] = Ra[(int64_t)(s0 + u) * 4 * h];
#pragma unroll
                for (int u = 0; u < CH; ++u) rb[u] = vb[4 * (s0 + u)];
#pragma unroll
                for (int u = 0; u < CH; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[u], rb[u], acc, 0, 0, 0);
            }
            if (mine) {
                double* o = so + t * opitch + it * 16 + kq;  // result r: output i = it * 16 + kq + 4 r, table t
#pragma unroll
                for (int r = 0; r < 4; ++r) o[4 * r] = acc[r];
            }
        }
    }
    // (stored straight from the accumulators, every lane wrote four lone 8-byte values 32 bytes apart into sixteen different tables:
    // partial-sector writes, 1.2 M tables x 64 of them per V = 2048 batch)
    __syncthreads();
    for (int e = tid; e < nt * h; e += 256) {
        const int tt = e / h, i = e - tt * h;
        const double a = so[tt * opitch + i];
        px_out[(int64_t)s_tab[tt] * h + i] = a;
        if (px32_out) px32_out[(int64_t)s_tab[tt] * h + i] = (float)a;
    }
}

template <typename CT>
static void launch_tables(int64_t n_tabs, size_t tab_lds, hipStream_t st, const CT* X, const CT* Cs, const double* Rt,
                          const double* mus, const double* subs, const TabDesc* tabs, const int* tab_order, int V, int h, int w,
                          int nf, int K, int D, double* T, PwProg prog_w, double* px_out, const int64_t* d_totals = nullptr,
                          float* px32_out = nullptr) {
    constexpr int TB = 8;
    if (px_out && h <= 256 && h % 16 == 0 && !getenv("CIS_TABLES_UNGROUPED") && !getenv("CIS_TABLES_VALU")) {
        // the grouped projection on the float64 matrix cores (16 tables per block); CIS_TABLES_VALU=1: the vector form below
        const size_t lds = (size_t)16 * (h + 4 + h + 2) * sizeof(double);
        if (h % 64 == 0)
            hipLaunchKernelGGL((k_tables_group_mfma<CT, 16>), dim3((unsigned)ceil_div(n_tabs, 16)), dim3(256), lds, st, X, Cs, Rt, mus, tabs,
                               tab_order, (int)n_tabs, V, h, D, px_out, d_totals, px32_out);
        else
            hipLaunchKernelGGL((k_tables_group_mfma<CT, 4>), dim3((unsigned)ceil_div(n_tabs, 16)), dim3(256), lds, st, X, Cs, Rt, mus, tabs,
                               tab_order, (int)n_tabs, V, h, D, px_out, d_totals, px32_out);
    } else if (px_out && h <= 256 && 256 / h >= 1 && !getenv("CIS_TABLES_UNGROUPED")) {
        const size_t lds = (size_t)(TB * h + TB * h) * sizeof(double);
        hipLaunchKernelGGL((k_tables_group<CT, TB>), dim3((unsigned)ceil_div(n_tabs, TB)), dim3(256), lds, st, X, Cs, Rt, mus, tabs,
                           tab_order, (int)n_tabs, V, h, D, px_out, d_totals, px32_out);
    } else {
        hipLaunchKernelGGL(k_tables<CT>, dim3((unsigned)n_tabs), dim3(256), tab_lds, st, X, Cs, Rt, mus, subs, tabs, V, h, w, nf, K, D, T,
                           prog_w, px_out, (const int*)nullptr, 0);
    }
}

// TPB tables per block: 64 for batches (the sub-centroid a thread keeps in registers serves 64 tables), 8 when the batch has few
// tables -- a single exhaustive query has 32: two blocks per (sub-quantizer, split) then walked 16 of them one after the other, 19 us.
template <int W, int TPB = 64>
__global__ __launch_bounds__(256) void k_tables_from_px(const double* __restrict__ px /* [ntab][h] */,
                                                        TabDesc* __restrict__ tabs, int n_tabs,
                                                        const double* __restrict__ subs, int h, int nf, int K,
                                                        double* __restrict__ T /* [ntab][nf][K] */,
                                                        float* __restrict__ T32 /* [ntab][nf][K] float32 copy for the scan */,
                                                        const int64_t* __restrict__ d_totals /* null, or the plan totals: n_tabs is a bound */) {
#ifndef CIS_TABLES_SCALAR_PX  // (scalar loads of the projected residual instead of the LDS copy: measured 0.185 against 0.162 ms on C2)
    __shared__ double sf[TPB][W];
#endif
    __shared__ int ssplit[TPB];
    const int j = blockIdx.y, z = blockIdx.z, k = threadIdx.x;
    const int t0 = blockIdx.x * TPB;
    if (d_totals) {
        n_tabs = (int)d_totals[1];
        if (t0 >= n_tabs) return;
    }
    const int nt = (n_tabs - t0 < TPB) ? (n_tabs - t0) : TPB;
#ifndef CIS_TABLES_SCALAR_PX  // (scalar loads of the projected residual instead of the LDS copy: measured 0.185 against 0.162 ms on C2)
    for (int e = threadIdx.x; e < nt * W; e += 256) {
        const int t = e / W, i = e - t * W;
        sf[t][i] = px[(int64_t)(t0 + t) * h + j * W + i];
    }
#endif
    if (threadIdx.x < nt) ssplit[threadIdx.x] = tabs[t0 + threadIdx.x].split;
    __syncthreads();
    const bool on = k < K;  // all lanes stay in the loop: the per-table maximum below is a full-wave reduction
    double sc[W];
    const double* src = subs + ((int64_t)(z * nf + j) * K + (on ? k : 0)) * W;
#pragma unroll
    for (int i = 0; i < W; ++i) sc[i] = src[i];
    for (int t = 0; t < nt; ++t) {
        if (ssplit[t] != z) continue;
#ifndef CIS_TABLES_SCALAR_PX  // (scalar loads of the projected residual instead of the LDS copy: measured 0.185 against 0.162 ms on C2)
        const double* f = sf[t];
#else
        // the projected residual of table t is the same for every lane: a uniform address, i.e. scalar loads into SGPR operands
        // (it used to be staged in LDS and read back sixteen times per table by every wave)
        const double* f = px + (int64_t)(t0 + t) * h + j * W;
#endif
        auto elem = [&](int i) -> double { const double df = f[i] - sc[i]; return df * df; };
        const double v = pw_leaf<double>(elem, 0, W);
        const float v32 = (float)v;
        if (on) {
            // NON-TEMPORAL: the float64 copy (2/3 of the bytes this kernel writes) is read sparsely, by the merge, much later; written with
            // the default policy it pushed the float32 copy -- which the scan stages next -- and the codes out of L2 / Infinity Cache.
            // Round 6, same box, median of two runs each: C4 21.3 -> 21.8 M queries/s (scan 0.233 -> 0.230 ms), C2 19.7 -> 21.1 M
            // (scan 0.157 -> 0.152, merge 0.115 -> 0.105 ms).  -DCIS_TABLES_NO_NT: the default policy (A/B).
#ifndef CIS_TABLES_NO_NT
            __builtin_nontemporal_store(v, &T[((int64_t)(t0 + t) * nf + j) * K + k]);
#else
            T[((int64_t)(t0 + t) * nf + j) * K + k] = v;
#endif
            if (T32) T32[((int64_t)(t0 + t) * nf + j) * K + k] = v32;  // (null: the scans convert the float64 entries themselves, see tab_f4; stored non-temporal too: no gain)
        }
        // largest float32 entry of the table (entries are >= 0: the bit patterns order like the values), for the
        // fixed-point scan's scale (lopq_scan3.hip); TabDesc::pad was zeroed when the descriptor was written
        uint32_t b = on ? __float_as_uint(v32) : 0u, dummy = 0u;
        wave_minmax_step<1>(dummy, b); wave_minmax_step<2>(dummy, b); wave_minmax_step<4>(dummy, b);
        wave_minmax_step<8>(dummy, b); wave_minmax_step<16>(dummy, b); wave_minmax_step<32>(dummy, b);
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(&tabs[t0 + t].pad), b);
    }
}

// float32 copy of the tables for the configurations that do not go through k_tables_from_px
__global__ void k_tables_f32(const double* __restrict__ T, int64_t n, int nf, int K, float* __restrict__ T32, TabDesc* __restrict__ tabs) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index into T: (tab, j, k)
    if (e >= n) return;
    const float v = (float)T[e];
    T32[e] = v;
    atomicMax(reinterpret_cast<unsigned int*>(&tabs[e / ((int64_t)nf * K)].pad), __float_as_uint(v));  // see k_tables_from_px
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

// same network with the size chosen at run time (N a power of two): the merge sorts only as many
// slots as it actually filled
template <int NT, bool PAY>
__device__ __forceinline__ void block_bitonic_rt(uint64_t* ka, uint64_t* kb, int64_t* pay, int N) {
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

// k_rank for wide coarse vocabularies (production configs go up to V = 4096): the rank by counting above is O(V^2)
// per (query, split); here the (distance bits, centroid index) pairs are sorted in LDS -- the index as second key
// reproduces "first minimum wins" among equal distances.
template <typename CT>
__global__ __launch_bounds__(256) void k_rank_sort(const CT* __restrict__ dist /* [2][nq][V] */, int nq, int V, int Vp2,
                                                   uint16_t* __restrict__ order, CT* __restrict__ sorted, int* __restrict__ grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* ka = reinterpret_cast<uint64_t*>(smem);
    uint64_t* kb = ka + Vp2;
    const int q = blockIdx.x, s = blockIdx.y;
    if (q == 0 && s == 0)
        for (int i = threadIdx.x; i < 4 * V * GRP_SUB; i += blockDim.x) grp[i] = 0;
    const CT* d = dist + ((int64_t)s * nq + q) * V;
    if constexpr (sizeof(CT) == 4) {
        // float32 distances: (distance bits, index) is ONE 64-bit key -- half the LDS traffic of the pair sort below
        for (int v = threadIdx.x; v < Vp2; v += 256) ka[v] = v < V ? ((f2bits(d[v]) << 32) | (uint64_t)v) : ~0ull;
        __syncthreads();
        for (int k = 2; k <= Vp2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < (Vp2 >> 1); t += 256) {
                    const int a_ = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int b_ = a_ + j;
                    const uint64_t x = ka[a_], y = ka[b_];
                    if ((x > y) == ((a_ & k) == 0)) { ka[a_] = y; ka[b_] = x; }
                }
                __syncthreads();
            }
        }
        for (int r = threadIdx.x; r < V; r += 256) {
            const int v = (int)(uint32_t)ka[r];
            order[((int64_t)q * 2 + s) * V + r] = (uint16_t)v;
            sorted[((int64_t)q * 2 + s) * V + r] = d[v];
        }
        return;
    }
    for (int v = threadIdx.x; v < Vp2; v += 256) {
        ka[v] = v < V ? f2bits(d[v]) : ~0ull;
        kb[v] = v < V ? (uint64_t)v : ~0ull;
    }
    __syncthreads();
    block_bitonic_rt<256, false>(ka, kb, nullptr, Vp2);
    for (int r = threadIdx.x; r < V; r += 256) {
        const int v = (int)kb[r];
        order[((int64_t)q * 2 + s) * V + r] = (uint16_t)v;
        sorted[((int64_t)q * 2 + s) * V + r] = d[v];
    }
}

// k_rank_sort for float32 distances with the sort in REGISTERS (round 4): thread t owns the NPT consecutive elements t * NPT ..., so of
// the log2(N) (log2(N) + 1) / 2 compare-exchange stages of the bitonic network those with partner distance j < NPT stay inside a thread,
// those with j < 64 NPT are one 64-bit lane exchange inside a wave, and only the 3 (N = 2048) to 6 (N = 4096) stages across waves go
// through LDS with a barrier -- the LDS form above pays a barrier and four LDS accesses per element for every one of its 66 / 78 stages.
// Same keys (distance bits << 32 | centroid index), same order: identical output.
template <int NPT>
__global__ __launch_bounds__(256) void k_rank_sort_reg(const float* __restrict__ dist /* [2][nq][V] */, int nq, int V,
                                                       uint16_t* __restrict__ order, float* __restrict__ sorted, int* __restrict__ grp) {
    constexpr int N = 256 * NPT;
    __shared__ uint64_t sx[N];
    const int q = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    if (q == 0 && s == 0)
        for (int i = tid; i < 4 * V * GRP_SUB; i += 256) grp[i] = 0;
    const float* d = dist + ((int64_t)s * nq + q) * V;
    uint64_t key[NPT];
#pragma unroll
    for (int r = 0; r < NPT; ++r) {
        const int e = tid * NPT + r;
        key[r] = e < V ? ((f2bits(d[e < V ? e : 0]) << 32) | (uint64_t)e) : ~0ull;
    }
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64 * NPT) {  // across waves: through LDS
                __syncthreads();
#pragma unroll
                for (int r = 0; r < NPT; ++r) sx[tid * NPT + r] = key[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < NPT; ++r) {
                    const int e = tid * NPT + r;
                    const uint64_t o = sx[e ^ j];
                    const bool keep_min = ((e & j) == 0) == ((e & k) == 0);
                    key[r] = keep_min ? (o < key[r] ? o : key[r]) : (o > key[r] ? o : key[r]);
                }
            } else if (j >= NPT) {  // across lanes of the wave
                const int lj = j / NPT;
#pragma unroll
                for (int r = 0; r < NPT; ++r) {
                    const int e = tid * NPT + r;
                    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key[r], lj), hi = (uint32_t)__shfl_xor((int)(uint32_t)(key[r] >> 32), lj);
                    const uint64_t o = ((uint64_t)hi << 32) | lo;
                    const bool keep_min = ((e & j) == 0) == ((e & k) == 0);
                    key[r] = keep_min ? (o < key[r] ? o : key[r]) : (o > key[r] ? o : key[r]);
                }
            } else {  // inside the thread
#pragma unroll
                for (int r = 0; r < NPT; ++r) {
                    if ((r & j) == 0) {
                        const int e = tid * NPT + r;
                        const bool asc = (e & k) == 0;
                        const uint64_t a = key[r], b = key[r | j];
                        const bool sw = (a > b) == asc;
                        key[r] = sw ? b : a;
                        key[r | j] = sw ? a : b;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NPT; ++r) {
        const int e = tid * NPT + r;
        if (e < V) {
            order[((int64_t)q * 2 + s) * V + e] = (uint16_t)(uint32_t)key[r];
            sorted[((int64_t)q * 2 + s) * V + e] = __uint_as_float((uint32_t)(key[r] >> 32));
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
                                                  int Mrt, int K, int limit, int S, const int* __restrict__ item_flag,
                                                  cis_hit* __restrict__ item_hits, int* __restrict__ item_n) {
    if (item_flag && !item_flag[blockIdx.x]) return;  // only redo what the fast kernel gave up on
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
        __syncthreads();  // everybody has read the count before the next iteration's appends change it
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
    cis_hit* out = item_hits + (int64_t)blockIdx.x * S;
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
// kernel: ADC scan v2 -- float32 prefilter in LDS, barrier-free wave-private top-k on exact keys
// ================================================================================================
// Why a second kernel: v1 above is exact by construction (float64 tables and sums) but is bound by
// LDS bank conflicts (32 lanes gathering from one 256-entry table collide ~3.5-way), by float64
// adds and by workgroup barriers every iteration (rocprof: 59 % of wave cycles waiting).  v2 returns
// bit-identical results with a different division of labour:
//
//  * the two half tables sit in LDS as float32 in ENTRY-major order tab[k][j] (j fastest).  Bank =
//    (k*M + j) mod 32, so sub-quantizer j owns the 32/M banks {j, j+M, ...}.  Lane l walks the
//    sub-quantizers in a rotated order (step t -> table rot(l, t)), so at every step the 32 lanes of
//    an LDS lane group are spread evenly over all M tables: 32/M lanes into 32/M banks instead of
//    32 lanes into 32 banks;
//  * each candidate gets a float32 sum d32 (any order).  |d32 - d64| <= eps*d64 with
//    eps = 2*M*2^-24 (entries are >= 0: one rounding per converted entry, M-1 per add).  The hot
//    loop only REJECTS candidates with d32 > bound*(1+3eps), where `bound` is an exact float64
//    distance that at least `limit` already-seen candidates do not exceed; a rejected candidate is
//    therefore strictly worse than `limit` others and cannot be in the exact top-`limit`;
//  * everything that is not rejected is appended (position only) to a wave-private LDS region -- no
//    workgroup barrier in the loop.  When a region fills up its wave re-scores the new entries
//    EXACTLY (float64 table entries from global memory, summed left to right as search.py:173),
//    selects its own limit-th and ceil(limit/4)-th smallest exact keys (dist, pos) with register
//    bitonic sorts, publishes them, and keeps only entries that are within its own top-`limit` and
//    not above the block bound  min( max_w wt[w], min_w wl[w] )  built from the published values
//    (each is a distance that >= `limit` seen candidates do not exceed).  Exact ties -- thousands of
//    identical codes are common in real indexes -- are resolved on (dist, pos), never on float32;
//  * at the end every wave holds <= `limit` exact hits; they are written as cis_hit and the
//    per-query merge ranks them by (dist, visit_rank, pos).

#ifdef CIS_SCAN_COUNTERS
__device__ unsigned long long g_scan_ctr[16];  // compactions, rescored entries, exact-cut, second sorts, appended
#define CIS_CTR(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_scan_ctr[i], (unsigned long long)(v)); } while (0)
#define CIS_CLK() ((long long)__builtin_amdgcn_s_memtime())
#else
#define CIS_CTR(i, v) do { } while (0)
#define CIS_CLK() 0ll
#endif

struct ScanShared {  // one per query handled by the workgroup
    uint64_t wt[8];  // per wave: exact dist bits that >= ceil(limit/NW) of its candidates do not exceed
    uint64_t wl[8];  // per wave: exact dist bits that >= limit of its candidates do not exceed
    float bound_f;   // block bound rounded UP to float32 for the hot loop (refreshed at every compaction;
                     // a lost concurrent update only leaves it looser for a while)
    int pad0;
    int wcnt[8];     // survivors per wave at the end
    uint64_t ext;    // bound published by workgroups that scanned OTHER cells for the same query (qbound[q] at start)
};


static __device__ __forceinline__ float block_bound_f32(const ScanShared* sh) { return lds_ld(&sh->bound_f); }
template <int NW>
static __device__ __forceinline__ uint64_t block_bound_u64(const ScanShared* sh) {
    uint64_t t = lds_ld(&sh->wt[0]);
    uint64_t l = lds_ld(&sh->wl[0]);
#pragma unroll
    for (int i = 1; i < NW; ++i) {
        const uint64_t a = lds_ld(&sh->wt[i]), b = lds_ld(&sh->wl[i]);
        t = a > t ? a : t;
        l = b < l ? b : l;
    }
    const uint64_t e = lds_ld(&sh->ext);
    l = e < l ? e : l;
    return t < l ? t : l;
}

// exact float64 distance of one candidate: table entries from global memory, summed left to right
static __device__ __forceinline__ double adc64_global(const uint8_t* __restrict__ codes, int64_t p, int M, int K,
                                                      const double* __restrict__ t0, const double* __restrict__ t1) {
    const uint8_t* c = codes + p * M;
    const int nf = M / 2;
    double d = t0[c[0]];
    for (int j = 1; j < nf; ++j) d = d + t0[j * K + c[j]];
    for (int j = 0; j < nf; ++j) d = d + t1[j * K + c[nf + j]];
    return d;
}


// Region entries during the scan are (float32 distance << 32 | candidate position): ordered by (d32, pos) as
// plain integers, appended with one ds_write_b64.  Two compactions work on them:
//  * wave_compact_approx (in the loop): float32 only, nothing leaves the CU.  It finds a distance v that at least L
//    entries do not exceed (ballot bisection that stops as soon as the count is within [L, L+W]), keeps everything
//    up to v*(1+3eps) -- an entry above that is strictly worse, in exact arithmetic, than the L entries below v --
//    and publishes the bounds v*(1+2eps) >= the exact distances of those L (resp. ceil(L/NW)) entries.
//  * wave_compact_exact (fallback when ties make the float32 cut keep too much, and once at the end): fetches the
//    codes again, re-scores every entry in float64 (table entries summed left to right as search.py:173), cuts to
//    the exact top-L by (dist, pos) and resolves exact ties by region order.
// Region entries are always in increasing candidate position (appends are, and the compactions are stable), so
// among exactly equal distances "first in the region" == "smallest pos".  Wave-synchronous: no s_barrier inside.
template <int NR, int NW>
__device__ __forceinline__ int wave_compact_approx(uint64_t* rk, int cnt, int L, int Lw, int cap, float margin, double infl,
                                                   ScanShared* sh, int w) {
    const int lane = threadIdx.x & 63;
    CIS_CTR(0, 1);
    uint32_t hi[NR], pp[NR];
    bool keep[NR];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * 64 + lane;
        keep[r] = e < cnt;
        const uint64_t k = keep[r] ? rk[e] : ~0ull;
        hi[r] = (uint32_t)(k >> 32);
        pp[r] = (uint32_t)k;
        mn = (keep[r] && hi[r] < mn) ? hi[r] : mn;
        mx = (keep[r] && hi[r] > mx) ? hi[r] : mx;
    }
    wave_minmax_step<1>(mn, mx); wave_minmax_step<2>(mn, mx); wave_minmax_step<4>(mn, mx);
    wave_minmax_step<8>(mn, mx); wave_minmax_step<16>(mn, mx); wave_minmax_step<32>(mn, mx);
    mn = (uint32_t)__builtin_amdgcn_readfirstlane((int)mn);
    mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
    const uint64_t INF64 = 0x7ff0000000000000ull;
    // (1) v2 with #{d32 <= v2} in [Lw, Lw + W2] -> this wave's share of the block bound
    const int W2 = Lw >= 16 ? (Lw >> 3) : 1;
    uint32_t lo2 = mn, v2 = mx;
    while (cnt >= Lw && lo2 < v2) {  // wave-uniform
        const uint32_t p = lo2 + ((v2 - lo2) >> 1);
        int c = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) c += __popcll(__ballot(keep[r] && hi[r] <= p));
        if (c >= Lw) {
            v2 = p;
            if (c <= Lw + W2) break;
        } else {
            lo2 = p + 1;
        }
    }
    const uint64_t boundW = cnt >= Lw ? (uint64_t)__double_as_longlong((double)__uint_as_float(v2) * infl) : INF64;
    if (lane == 0) {
        if (boundW < lds_ld(&sh->wt[w])) lds_st(&sh->wt[w], boundW);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint64_t bound = block_bound_u64<NW>(sh);
    float bf = __double2float_ru(__longlong_as_double((long long)bound));
    uint32_t cut = __float_as_uint(bf * margin);  // the hot loop's test
    int c_thr = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) c_thr += __popcll(__ballot(keep[r] && hi[r] <= cut));
    int W = cap - L;
    W = W > 24 ? 24 : (W < 0 ? 0 : W);
    if (c_thr > L + W) {
        // (2) the block bound does not thin this wave out (the other waves lag, the good candidates sit in this
        // wave's stripes, or many distances are equal): cut to the wave's own top L.
        // v with #{d32 <= v} in [L, L+W], or the exact L-th smallest when ties prevent that
        uint32_t lo_ = mn, v = mx;
        while (lo_ < v) {
            const uint32_t p = lo_ + ((v - lo_) >> 1);
            int c = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) c += __popcll(__ballot(keep[r] && hi[r] <= p));
            if (c >= L) {
                v = p;
                if (c <= L + W) break;
            } else {
                lo_ = p + 1;
            }
        }
        const uint32_t vm = __float_as_uint(__double2float_ru((double)__uint_as_float(v) * (double)margin));
        int c_keep = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) c_keep += __popcll(__ballot(keep[r] && hi[r] <= vm));
        if (c_keep > cap) return -1;  // a crowd of (nearly) equal distances, e.g. duplicate codes: resolve exactly
        const uint64_t boundL = (uint64_t)__double_as_longlong((double)__uint_as_float(v) * infl);
        if (lane == 0) {
            if (boundL < lds_ld(&sh->wl[w])) lds_st(&sh->wl[w], boundL);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        bound = block_bound_u64<NW>(sh);
        bf = __double2float_ru(__longlong_as_double((long long)bound));
        const uint32_t thr = __float_as_uint(bf * margin);
        cut = vm < thr ? vm : thr;
    }
    if (lane == 0) {
        if (bf < lds_ld(&sh->bound_f)) lds_st(&sh->bound_f, bf);
    }
    int ncnt = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool kp = keep[r] && hi[r] <= cut;
        const unsigned long long m = __ballot(kp);
        const int idx = ncnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (kp) rk[idx] = ((uint64_t)hi[r] << 32) | pp[r];
        ncnt += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ncnt;
}

// FINAL: leaves exact float64 keys in rk and positions in rp (the output format); otherwise the survivors go back
// to the (float32 rounded up, position) form.  Returns the new entry count (<= L).
template <int M, int NR, int NW, bool FINAL>
__device__ __forceinline__ int wave_compact_exact(uint64_t* rk, uint32_t* rp, int cnt, int L, int Lw,
                                                  ScanShared* sh, int w, const uint8_t* __restrict__ codes, int64_t start,
                                                  int K, const double* __restrict__ t0, const double* __restrict__ t1,
                                                  uint32_t& dup_pos) {
    const int lane = threadIdx.x & 63;
    const uint64_t INF64 = 0x7ff0000000000000ull;
    CIS_CTR(1, 1);
    uint32_t hi[NR], lo[NR], pp[NR];
    bool keep[NR];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * 64 + lane;
        keep[r] = e < cnt;
        pp[r] = keep[r] ? (uint32_t)rk[e] : 0xffffffffu;
        hi[r] = 0xffffffffu;
        lo[r] = 0xffffffffu;
    }
    {
        // idle lanes of a row fetch candidate 0; rows past the end are skipped (wave-uniform)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r * 64 >= cnt) break;
            const CodeWords<M> cw = load_code<M>(codes, start + (keep[r] ? pp[r] : 0u));
            const uint64_t k = keep[r] ? (uint64_t)__double_as_longlong(adc64_words<M>(cw.w, K, t0, t1)) : ~0ull;
            hi[r] = (uint32_t)(k >> 32);
            lo[r] = (uint32_t)k;
            mn = (keep[r] && hi[r] < mn) ? hi[r] : mn;
            mx = (keep[r] && hi[r] > mx) ? hi[r] : mx;
        }
    }
    wave_minmax_step<1>(mn, mx); wave_minmax_step<2>(mn, mx); wave_minmax_step<4>(mn, mx);
    wave_minmax_step<8>(mn, mx); wave_minmax_step<16>(mn, mx); wave_minmax_step<32>(mn, mx);
    mn = (uint32_t)__builtin_amdgcn_readfirstlane((int)mn);
    mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
    uint32_t vhiL = mx;
    uint64_t boundL = INF64;
    if (cnt >= L) {  // cut to this wave's own exact top-L
        const uint32_t vhi = wave_kth_bisect<NR>(hi, keep, mn, mx, L);
        vhiL = vhi;
        boundL = hi_to_bound(vhi);
        int c_less = 0, g = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            c_less += __popcll(__ballot(keep[r] && hi[r] < vhi));
            g += __popcll(__ballot(keep[r] && hi[r] == vhi));
        }
        int need = L - c_less;  // members of the group {hi == vhi} to keep, 1 <= need <= g
        if (need >= g) {
#pragma unroll
            for (int r = 0; r < NR; ++r) keep[r] = keep[r] && hi[r] <= vhi;
        } else {
            // low word of the first group member in region order; are all members equal to it?
            uint32_t lo_first = 0;
            bool found = false, uniform = true;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool in_g = keep[r] && hi[r] == vhi;
                const unsigned long long m = __ballot(in_g);
                if (!found && m) {
                    lo_first = (uint32_t)__builtin_amdgcn_readlane((int)lo[r], __ffsll((long long)m) - 1);
                    found = true;
                }
                if (found) uniform = uniform && (__ballot(in_g && lo[r] != lo_first) == 0ull);
            }
            uint32_t vlo = lo_first;
            if (!uniform) {
                uint32_t t2[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) t2[r] = (keep[r] && hi[r] == vhi) ? lo[r] : 0xffffffffu;
                wave_bitonic_sort<NR>(t2);
                vlo = wave_kth<NR>(t2, need - 1);
#pragma unroll
                for (int r = 0; r < NR; ++r) need -= __popcll(__ballot(keep[r] && hi[r] == vhi && lo[r] < vlo));
            }
            // exact ties (hi, lo) == (vhi, vlo): the first `need` in region order have the smallest positions
            int seen = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool tie = keep[r] && hi[r] == vhi && lo[r] == vlo;
                const unsigned long long m = __ballot(tie);
                const int rank = seen + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                keep[r] = keep[r] && (hi[r] < vhi || (hi[r] == vhi && (lo[r] < vlo || (tie && rank < need))));
                // The cut fell inside a group of exactly equal distances.  Remember one member: every LATER
                // candidate with the same code has the same distance and a larger position, so it loses
                // against all L entries kept here and the hot loop may skip it (duplicate codes are common).
                if (seen == 0 && m) dup_pos = (uint32_t)__builtin_amdgcn_readlane((int)pp[r], __ffsll((long long)m) - 1);
                seen += __popcll(m);
            }
        }
    }
    const uint64_t boundW = (cnt >= Lw) ? hi_to_bound(wave_kth_bisect<NR>(hi, keep, mn, vhiL, Lw)) : INF64;
    if (lane == 0) {
        if (boundW < lds_ld(&sh->wt[w])) lds_st(&sh->wt[w], boundW);
        if (boundL < lds_ld(&sh->wl[w])) lds_st(&sh->wl[w], boundL);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const uint64_t bound = block_bound_u64<NW>(sh);
    if (lane == 0) {
        const float bf = __double2float_ru(__longlong_as_double((long long)bound));
        if (bf < lds_ld(&sh->bound_f)) lds_st(&sh->bound_f, bf);
    }
    int ncnt = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const uint64_t k = ((uint64_t)hi[r] << 32) | lo[r];
        const bool kp = keep[r] && (k <= bound);
        const unsigned long long m = __ballot(kp);
        const int idx = ncnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (kp) {
            if constexpr (FINAL) {
                rk[idx] = k;
                rp[idx] = pp[r];
            } else {
                rk[idx] = ((uint64_t)__float_as_uint(__double2float_ru(__longlong_as_double((long long)k))) << 32) | pp[r];
            }
        }
        ncnt += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ncnt;
}

// Drop (float32 distance, position) entries above the block bound, with the hot loop's test.  Stable.
template <int NR, int NW>
__device__ __forceinline__ int wave_filter_approx(uint64_t* rk, int cnt, float margin, const ScanShared* sh) {
    const int lane = threadIdx.x & 63;
    const uint64_t bound = block_bound_u64<NW>(sh);
    const uint32_t thr = __float_as_uint(__double2float_ru(__longlong_as_double((long long)bound)) * margin);
    uint64_t k[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * 64 + lane;
        k[r] = e < cnt ? rk[e] : ~0ull;
    }
    int ncnt = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool kp = (r * 64 + lane < cnt) && ((uint32_t)(k[r] >> 32) <= thr);
        const unsigned long long m = __ballot(kp);
        const int idx = ncnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (kp) rk[idx] = k[r];
        ncnt += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ncnt;
}

// Drop entries above the (final) block bound; entries already carry exact keys.  Stable.
template <int NR, int NW>
__device__ __forceinline__ int wave_filter(uint64_t* rk, uint32_t* rp, int cnt, const ScanShared* sh) {
    const int lane = threadIdx.x & 63;
    const uint64_t bound = block_bound_u64<NW>(sh);
    uint64_t k[NR];
    uint32_t pp[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * 64 + lane;
        k[r] = e < cnt ? rk[e] : ~0ull;
        pp[r] = e < cnt ? rp[e] : 0u;
    }
    int ncnt = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool kp = (r * 64 + lane < cnt) && (k[r] <= bound);
        const unsigned long long m = __ballot(kp);
        const int idx = ncnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (kp) { rk[idx] = k[r]; rp[idx] = pp[r]; }
        ncnt += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ncnt;
}


// float32 ADC sum of one candidate with the lane-rotated table order (see header comment)
template <int M>
__device__ __forceinline__ float adc32(const CodeWords<M>& c, const char* __restrict__ tab, const RotConsts<M>& rc) {
    constexpr int SH = (M == 4) ? 4 : (M == 8 ? 5 : 6);  // log2(M * 4 bytes)
    uint32_t D[(M + 3) / 4];
    if constexpr (M == 4) {
        D[0] = c.w[0];
    } else if constexpr (M == 8) {
        D[0] = rc.hsel ? c.w[1] : c.w[0];
        D[1] = rc.hsel ? c.w[0] : c.w[1];
    } else {
        const bool b0 = rc.hsel & 1, b1 = rc.hsel & 2;
        const uint32_t x01 = b0 ? c.w[1] : c.w[0], y01 = b0 ? c.w[0] : c.w[1];
        const uint32_t x23 = b0 ? c.w[3] : c.w[2], y23 = b0 ? c.w[2] : c.w[3];
        D[0] = b1 ? x23 : x01;
        D[1] = b1 ? y23 : y01;
        D[2] = b1 ? x01 : x23;
        D[3] = b1 ? y01 : y23;
    }
    float f[M];
#pragma unroll
    for (int t = 0; t < M; ++t) {
        const uint32_t k = __builtin_amdgcn_ubfe(D[t >> 2], rc.sh[t & 3], 8);
        const uint32_t addr = (k << SH) | rc.cj[t];
        f[t] = *reinterpret_cast<const float*>(tab + addr);
    }
    float acc;
    if constexpr (M == 4) {
        acc = (f[0] + f[1]) + (f[2] + f[3]);
    } else if constexpr (M == 8) {
        acc = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
    } else {
        acc = (((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]))) +
              (((f[8] + f[9]) + (f[10] + f[11])) + ((f[12] + f[13]) + (f[14] + f[15])));
    }
    return acc;
}

// float32 ADC sums of one candidate for G queries at once: the LDS table interleaves the queries,
// tab[k][j][g], so ONE ds_read (b32 for G=1, b64 for G=2) serves all G queries of the workgroup.
template <int M, int G>
__device__ __forceinline__ void adc32g(const CodeWords<M>& c, const char* __restrict__ tab, const RotConsts<M>& rc,
                                       float (&out)[G]) {
    constexpr int SH = ((M == 4) ? 4 : (M == 8 ? 5 : 6)) + (G == 2 ? 1 : 0);  // log2(M * G * 4 bytes)
    uint32_t D[(M + 3) / 4];
    if constexpr (M == 4) {
        D[0] = c.w[0];
    } else if constexpr (M == 8) {
        D[0] = rc.hsel ? c.w[1] : c.w[0];
        D[1] = rc.hsel ? c.w[0] : c.w[1];
    } else {
        const bool b0 = rc.hsel & 1, b1 = rc.hsel & 2;
        const uint32_t x01 = b0 ? c.w[1] : c.w[0], y01 = b0 ? c.w[0] : c.w[1];
        const uint32_t x23 = b0 ? c.w[3] : c.w[2], y23 = b0 ? c.w[2] : c.w[3];
        D[0] = b1 ? x23 : x01;
        D[1] = b1 ? y23 : y01;
        D[2] = b1 ? x01 : x23;
        D[3] = b1 ? y01 : y23;
    }
    if constexpr (G == 1) {
        float f[M];
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const uint32_t k = __builtin_amdgcn_ubfe(D[t >> 2], rc.sh[t & 3], 8);
            f[t] = *reinterpret_cast<const float*>(tab + ((k << SH) | rc.cj[t]));
        }
#pragma unroll
        for (int st = 1; st < M; st <<= 1)
#pragma unroll
            for (int t = 0; t < M; t += 2 * st) f[t] = f[t] + f[t + st];
        out[0] = f[0];
    } else {
        // the two queries' entries sit side by side: one ds_read_b64 per table entry, one packed add per tree node
        f32x2_t f[M];
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const uint32_t k = __builtin_amdgcn_ubfe(D[t >> 2], rc.sh[t & 3], 8);
            f[t] = *reinterpret_cast<const f32x2_t*>(tab + ((k << SH) | (rc.cj[t] << 1)));
        }
#pragma unroll
        for (int st = 1; st < M; st <<= 1)
#pragma unroll
            for (int t = 0; t < M; t += 2 * st) f[t] = f[t] + f[t + st];
        out[0] = f[0][0];
        out[1] = f[0][1];
    }
}

// adc32g for G = 2 or 4 queries in two halves, so that the table reads of the NEXT candidate row can be issued before
// the adds of the current one (the LDS pipe and the VALU then work at the same time inside one wave).  The G queries'
// entries sit side by side: one ds_read_b64 / ds_read_b128 per table entry, packed adds per tree node.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int G> struct AdcVec;
template <> struct AdcVec<2> { typedef f32x2_t type; };
template <> struct AdcVec<4> { typedef f32x4_t type; };

template <int M, int G>
__device__ __forceinline__ void adc32_issue(const CodeWords<M>& c, const char* __restrict__ tab, const RotConsts<M>& rc,
                                            typename AdcVec<G>::type (&f)[M]) {
    typedef typename AdcVec<G>::type VT;
    constexpr int LG = (G == 2) ? 1 : 2;
    constexpr int SH = ((M == 4) ? 4 : (M == 8 ? 5 : 6)) + LG;  // log2(M * G * 4 bytes)
    uint32_t D[(M + 3) / 4];
    if constexpr (M == 4) {
        D[0] = c.w[0];
    } else if constexpr (M == 8) {
        D[0] = rc.hsel ? c.w[1] : c.w[0];
        D[1] = rc.hsel ? c.w[0] : c.w[1];
    } else {
        const bool b0 = rc.hsel & 1, b1 = rc.hsel & 2;
        const uint32_t x01 = b0 ? c.w[1] : c.w[0], y01 = b0 ? c.w[0] : c.w[1];
        const uint32_t x23 = b0 ? c.w[3] : c.w[2], y23 = b0 ? c.w[2] : c.w[3];
        D[0] = b1 ? x23 : x01;
        D[1] = b1 ? y23 : y01;
        D[2] = b1 ? x01 : x23;
        D[3] = b1 ? y01 : y23;
    }
#pragma unroll
    for (int t = 0; t < M; ++t) {
        const uint32_t k = __builtin_amdgcn_ubfe(D[t >> 2], rc.sh[t & 3], 8);
        f[t] = *reinterpret_cast<const VT*>(tab + ((k << SH) | (rc.cj[t] << LG)));
    }
}

template <int M, int G>
__device__ __forceinline__ void adc32_reduce(typename AdcVec<G>::type (&f)[M], float (&out)[G]) {
#pragma unroll
    for (int st = 1; st < M; st <<= 1)
#pragma unroll
        for (int t = 0; t < M; t += 2 * st) f[t] = f[t] + f[t + st];
#pragma unroll
    for (int g = 0; g < G; ++g) out[g] = f[0][g];
}

// One workgroup (NW waves) scans one cell chunk for `ng` <= G queries that all visit it.
template <int M, int NR, int U, int G, int NW>
__device__ __forceinline__ void scan2_group(const WorkItem* __restrict__ items_all, const int (&item_idx)[G], int ng,
                                            const double* __restrict__ T, const float* __restrict__ T32,
                                            const uint8_t* __restrict__ codes,
                                            const int64_t* __restrict__ ids, int K, int L, int S, float margin,
                                            uint64_t* __restrict__ item_surv, int* __restrict__ item_n,
                                            unsigned long long* __restrict__ qbound, char* smem) {
    // region capacity: 8 entries short of the NR*64 keys a wave can hold in registers, so that the
    // G=2 / 4-wave layout (16 KB tables + 8 regions) stays under 40 KB and four workgroups share a CU
    constexpr int R = NR * 64 - 8;
    // item_idx[] is wave-uniform (the slot index went through readfirstlane): the work items are scalar loads and live in
    // scalar registers instead of 10 VGPRs each (they used to spill to scratch)
    // (field by field: a local array of the 48-byte structs was kept in scratch, 24 KB of stores per slot and workgroup)
    int it_tab0[G], it_tab1[G], it_q[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        it_tab0[g] = items_all[item_idx[g]].tab0;
        it_tab1[g] = items_all[item_idx[g]].tab1;
        it_q[g] = items_all[item_idx[g]].q;
    }
    const int it0_len = items_all[item_idx[0]].len;
    const int64_t it0_start = items_all[item_idx[0]].start;
    char* tab = smem;                                                              // [K][M][G] float32
    uint64_t* rk_all = reinterpret_cast<uint64_t*>(smem + (size_t)K * M * G * 4);  // [G][NW][R] (d32, pos) entries; exact keys at the end
    uint64_t* tr_all = rk_all + G * NW * R;                                        // [G][NW][64] scratch slots of the branch-free append
    ScanShared* sh = reinterpret_cast<ScanShared*>(tr_all + G * NW * 64);          // [G]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nf = M / 2;
    const long long clk_begin = CIS_CLK();
    long long clk_slow = 0, clk_comp = 0, clk_exact = 0, clk_final = 0;
    (void)clk_exact; (void)clk_final;
    int n_slow = 0, n_app = 0;
    (void)clk_begin; (void)clk_slow; (void)clk_comp; (void)n_slow; (void)n_app;
    const float INF = __int_as_float(0x7f800000);
    {
        // LDS tables from the float32 copies ([nf][K] per (query, half)): 16-byte loads (four consecutive k of one
        // sub-quantizer), all in flight together, then one ds_write per entry that carries both queries' values.
        float* tf = reinterpret_cast<float*>(tab);
        const int nvec = (nf * K) >> 2;  // K is a multiple of 4 for the supported shapes
        for (int e0 = 0; e0 < nvec; e0 += NW * 64) {
            const int e = e0 + tid;
            float4 v[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool on = (g < ng) && (e < nvec);  // an absent second query gets +inf tables: nothing ever passes
                const int eg = e < nvec ? e : 0;
                v[g][0] = tab_f4(T32, T, it_tab0[g], nf * K, eg);
                v[g][1] = tab_f4(T32, T, it_tab1[g], nf * K, eg);
                if (!on) { v[g][0] = make_float4(INF, INF, INF, INF); v[g][1] = v[g][0]; }
            }
            if (e < nvec) {
                const int j = (4 * e) / K, k0 = 4 * e - j * K;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float* dst = tf + ((k0 + c) * M + s2 * nf + j) * G;
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const float4 q = v[g][s2];
                            dst[g] = c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w));
                        }
                    }
                }
            }
        }
        if (tid < 8 * G) {
            const int g = tid >> 3, i = tid & 7;
            sh[g].wt[i] = 0x7ff0000000000000ull; sh[g].wl[i] = 0x7ff0000000000000ull;
            sh[g].wcnt[i] = 0;
            if (i == 0) {
                // A distance that >= L candidates of this query in already scanned cells do not exceed: candidates
                // above it are strictly worse than L others, whichever cell they are in.  Cells of one query are
                // scanned by different workgroups at different times (the slot list is sorted by cell), so the
                // later ones start with a tight bound instead of +inf.  Only the amount of work depends on timing.
#ifndef CIS_SCAN_NO_QBOUND
                int qg = it_q[0];  // g is a thread index here: select, do not index the register array
#pragma unroll
                for (int gg = 1; gg < G; ++gg) qg = (g == gg) ? it_q[gg] : qg;
                const unsigned long long e = (g < ng) ? __hip_atomic_load(&qbound[qg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                      : 0x7ff0000000000000ull;
#else
                const unsigned long long e = 0x7ff0000000000000ull;
#endif
                sh[g].ext = e;
                // an absent second query: negative bound, nothing ever passes (its tables are +inf as well)
                sh[g].bound_f = (g < ng) ? __double2float_ru(__longlong_as_double((long long)e)) : -1.0f;
            }
        }
    }
    __syncthreads();
    const long long clk_tab_end = CIS_CLK();
    (void)clk_tab_end;
    const RotConsts<M> rc = make_rot<M>(lane);
    const int Lw = (L + NW - 1) / NW;
    // wave-uniform by construction; tell the compiler so (scalar loop control, scalar tail test)
    const int len = __builtin_amdgcn_readfirstlane(it0_len);
    const int64_t start = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it0_start >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)it0_start);
    const int nit = (len + 64 * U - 1) / (64 * U);
    int cnt[G];
    constexpr double EPS32 = 2.0 * M * 5.9604644775390625e-8;  // float32 sum vs exact: |d32 - d64| <= EPS32 * d64
    const double infl = 1.0 + 2.0 * EPS32;
    CodeWords<M> dup[G];  // per query: a code whose later copies cannot enter this wave's top-L any more
    bool has_dup[G];
    bool any_dup = false;
#pragma unroll
    for (int g = 0; g < G; ++g) { cnt[g] = 0; has_dup[g] = false; dup[g] = CodeWords<M>(); }
#ifndef CIS_SCAN_PLAIN_LOADS
    // buffer descriptor over this chunk's codes, built from wave-uniform values only
    __amdgpu_buffer_rsrc_t rs;
    {
        const uint64_t cbase = (uint64_t)(uintptr_t)(codes + start * M);
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cbase);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cbase >> 32));
        const int nbytes = __builtin_amdgcn_readfirstlane(len * M);
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)bhi << 32) | blo), 0, nbytes, 0x00020000);
    }
#define CIS_LOAD_CODE(p) load_code_buf<M>(rs, (p))
#else
#define CIS_LOAD_CODE(p) load_code<M>(codes, start + ((p) < len ? (p) : len - 1))
#endif
    CodeWords<M> nxt[U];
    if (w < nit) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = CIS_LOAD_CODE(w * 64 * U + u * 64 + lane);
    }
    for (int iter = w; iter < nit; iter += NW) {
        const int base = iter * 64 * U;
        CodeWords<M> cur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        if (iter + NW < nit) {  // software prefetch: the next iteration's codes are in flight while this one computes
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = CIS_LOAD_CODE((iter + NW) * 64 * U + u * 64 + lane);  // tail lanes are masked below
        }
        float d[U][G];
#ifndef CIS_SCAN_NO_PIPELINE
        if constexpr (G >= 2) {
            typename AdcVec<G>::type fbuf[2][M];
            adc32_issue<M, G>(cur[0], tab, rc, fbuf[0]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u + 1 < U) adc32_issue<M, G>(cur[u + 1], tab, rc, fbuf[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                adc32_reduce<M, G>(fbuf[u & 1], d[u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#endif
        {
#pragma unroll
            for (int u = 0; u < U; ++u) adc32g<M, G>(cur[u], tab, rc, d[u]);
        }
        // Fast path: one compare + ballot per (candidate row, query), ONE branch if no mask is set.  Lanes past
        // the end of the chunk (last iteration only) get NaN distances, which never compare <=; the duplicate
        // exclusion is applied in the slow path only.
        if (base + 64 * U > len) {
            const float QNAN = __int_as_float(0x7fc00000);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int g = 0; g < G; ++g) d[u][g] = (base + u * 64 + lane < len) ? d[u][g] : QNAN;
        }
        unsigned long long pm[U][G];
        unsigned long long any = 0ull;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float thrm = block_bound_f32(&sh[g]) * margin;
#ifdef CIS_PROBE_HOTLOOP
            thrm = (margin > 100.f) ? thrm : -1.0f;  // probe: nothing passes, only the float32 scan runs
#endif
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pm[u][g] = __ballot(d[u][g] <= thrm);
                any |= pm[u][g];
            }
        }
        if (any_dup) {  // wave-uniform, rare: drop later copies of a code that already lost a tie-break (see wave_compact_exact)
            any = 0ull;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const unsigned long long dup_on = has_dup[g] ? ~0ull : 0ull;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < (M + 3) / 4; ++i) same = same && (cur[u].w[i] == dup[g].w[i]);
                    pm[u][g] &= ~(__ballot(same) & dup_on);
                    any |= pm[u][g];
                }
            }
        }
        if (any == 0ull) continue;  // the usual case: nothing in these 64*U candidates beats a bound
        const long long clk_s0 = CIS_CLK();
        ++n_slow;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g >= ng) break;
            uint64_t* rk = rk_all + (g * NW + w) * R;
            uint64_t* tr = tr_all + (g * NW + w) * 64;
            uint32_t* rp = nullptr;  // positions array of the exact compaction's final form: not used inside the loop
            int ntot = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) ntot += __popcll(pm[u][g]);
            if (ntot == 0) continue;  // only the other query has candidates in this iteration
            if (cnt[g] + ntot <= R) {
                // The usual case: everything fits.  Straight-line code, no branch per row: every lane stores, the
                // lanes that did not pass into a scratch slot of their own.
                const int trash = (int)(tr - rk) + lane;
                int c = cnt[g];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned long long m = pm[u][g];
                    const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, c));
                    int sel;
                    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(sel) : "v"(trash), "v"(idx), "s"(m));
                    rk[sel] = ((uint64_t)__float_as_uint(d[u][g]) << 32) | (uint32_t)(base + u * 64 + lane);
                    c += __popcll(m);
                }
                cnt[g] = c;
                n_app += ntot;
                continue;
            }
            // Append row after row while they fit; when one does not, compact (ONE call site per query, whatever U
            // is), re-test the rows not yet appended against the new bound and go on.  After a compaction
            // cnt <= R - 64, so the next row always fits and the loop ends after at most U compactions.
            int u0 = 0;
            while (true) {
                bool full = false;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned long long m = pm[u][g];
                    if (u < u0 || full || m == 0ull) continue;
                    const int n = __popcll(m);
                    if (cnt[g] + n > R) {
                        full = true;
                        u0 = u;
                        continue;
                    }
                    const int idx = cnt[g] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                    if ((m >> lane) & 1ull) rk[idx] = ((uint64_t)__float_as_uint(d[u][g]) << 32) | (uint32_t)(base + u * 64 + lane);
                    cnt[g] += n;
                    n_app += n;
                    u0 = u + 1;
                }
                if (!full) break;
                const long long clk_c0 = CIS_CLK();
                int c2 = wave_compact_approx<NR, NW>(rk, cnt[g], L, Lw, R - 64, margin, infl, &sh[g], w);
                if (c2 < 0) {
                    uint32_t dp = 0xffffffffu;
                    const long long clk_e0 = CIS_CLK();
                    c2 = wave_compact_exact<M, NR, NW, false>(rk, rp, cnt[g], L, Lw, &sh[g], w, codes, start, K, T + (int64_t)it_tab0[g] * nf * K,
                                                                  T + (int64_t)it_tab1[g] * nf * K, dp);
                    clk_exact += CIS_CLK() - clk_e0;
                    dp = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp);
                    if (dp != 0xffffffffu) {
                        dup[g] = load_code<M>(codes, start + (int64_t)dp);
                        has_dup[g] = true;
                        any_dup = true;
                    }
                }
                cnt[g] = c2;
                clk_comp += CIS_CLK() - clk_c0;
                const float thrm = block_bound_f32(&sh[g]) * margin;
                const unsigned long long dup_on = has_dup[g] ? ~0ull : 0ull;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < (M + 3) / 4; ++i) same = same && (cur[u].w[i] == dup[g].w[i]);
                    pm[u][g] &= __ballot(d[u][g] <= thrm) & ~(__ballot(same) & dup_on);
                }
            }
        }
        clk_slow += CIS_CLK() - clk_s0;
    }
    const long long clk_loop_end = CIS_CLK();
    (void)clk_loop_end;
    // End of the chunk.  Every wave cuts its region in float32 and publishes its bounds; after the barrier the
    // block bound is (about) the L-th smallest distance of the whole chunk, so only ~L/NW entries per wave survive
    // it.  The survivors -- a superset of the chunk's exact top L -- leave as (float32 distance, position) pairs;
    // k_merge_survivors re-scores them in float64 and ranks the query's candidates exactly.
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g >= ng) break;
        uint64_t* rk = rk_all + (g * NW + w) * R;
        if (cnt[g] > 0) cnt[g] = wave_compact_approx<NR, NW>(rk, cnt[g], L, Lw, NR * 64, margin, infl, &sh[g], w);
    }
    clk_final = CIS_CLK() - clk_loop_end;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g >= ng) break;
        uint64_t* rk = rk_all + (g * NW + w) * R;
        cnt[g] = wave_filter_approx<NR, NW>(rk, cnt[g], margin, &sh[g]);
        if (lane == 0) sh[g].wcnt[w] = cnt[g];
#ifndef CIS_SCAN_NO_QBOUND
        if (tid == 0) {
            const uint64_t b = block_bound_u64<NW>(&sh[g]);
            if (b < sh[g].ext) atomicMin(&qbound[it_q[g]], (unsigned long long)b);
        }
#endif
    }
    const long long clk_exact_end = CIS_CLK();
    (void)clk_exact_end;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g >= ng) break;
        const uint64_t* rk = rk_all + (g * NW + w) * R;
        int off = 0, total = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = sh[g].wcnt[i];
            off += (i < w) ? c : 0;
            total += c;
        }
        uint64_t* out = item_surv + (int64_t)item_idx[g] * S + off;  // S = NW * R >= total
        for (int e = lane; e < cnt[g]; e += 64) out[e] = rk[e];
        if (tid == 0) item_n[item_idx[g]] = total;
    }
#ifdef CIS_SCAN_COUNTERS
    CIS_CTR(3, n_slow);
    CIS_CTR(8, clk_exact);
    CIS_CTR(10, clk_exact_end - clk_loop_end);
    CIS_CTR(11, clk_tab_end - clk_begin);
    CIS_CTR(9, clk_final);
    CIS_CTR(4, n_app);
    CIS_CTR(5, clk_comp);
    CIS_CTR(6, clk_slow - clk_comp);
    CIS_CTR(7, CIS_CLK() - clk_begin);
    CIS_CTR(2, clk_loop_end - clk_begin);
#endif
}

// Persistent launch: (blocks per CU) x 256 workgroups pull SLOTS from eight queues, one per XCD.  A
// slot holds up to G work items of the same coarse cell (slots come from the cell-sorted item list);
// queue x owns the x-th eighth of the slot list, so the workgroups resident on one XCD (workgroup b
// runs on XCD b % 8 -- observed dispatch rule, used for speed only) stream the same few cells through
// that XCD's private L2.  A workgroup whose own queue is empty steals from the others, which removes
// the tail caused by unequal cell sizes.
template <int M, int NR, int U, int G, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NR == 16 ? 2 : (G == 4 ? 2 : ((M == 16 && G == 2) ? 3 : 4)), 4))) void k_adc_scan2(const WorkItem* __restrict__ items, const int* __restrict__ slots,
                                                       const int* __restrict__ n_slots_ptr, const double* __restrict__ T,
                                                       const float* __restrict__ T32,
                                                       const uint8_t* __restrict__ codes, const int64_t* __restrict__ ids,
                                                       int K, int L, int S, float margin,
                                                       int* __restrict__ queue_ctr /* [8], zeroed */,
                                                       uint64_t* __restrict__ item_surv, int* __restrict__ item_n,
                                                       unsigned long long* __restrict__ qbound /* [nq], +inf */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = NR * 64 - 8;
    int* s_next = reinterpret_cast<int*>(smem + (size_t)K * M * G * 4 + (size_t)G * NW * (R + 64) * 8 + G * sizeof(ScanShared));
    const int* qs = n_slots_ptr + 8;  // [9] queue starts, written by the slot builder
    const int home = blockIdx.x & 7;
    for (int a = 0; a < 8; ++a) {
        const int x = (home + a) & 7;
        const int qstart = qs[x];
        const int count = qs[x + 1] - qstart;
        while (true) {
            __syncthreads();  // previous slot fully written out; LDS may be reused
            if (threadIdx.x == 0) *s_next = atomicAdd(&queue_ctr[x], 1);
            __syncthreads();
            const int j = __builtin_amdgcn_readfirstlane(*s_next);  // wave-uniform: the slot and its work items stay in scalar registers
            if (j >= count) break;
            const int slot = qstart + j;
            int idx[G];
            int ng = 0;
            bool same = true;  // the items of a slot must cover the same chunk of the same cell; otherwise run them one by one
#pragma unroll
            for (int g = 0; g < G; ++g) {
                idx[g] = slots[slot * G + g];
                if (idx[g] >= 0) {
                    ng = g + 1;
                    same = same && items[idx[g]].start == items[idx[0]].start && items[idx[g]].len == items[idx[0]].len;
                } else {
                    idx[g] = idx[0];
                }
            }
            if constexpr (G >= 2) {
                if (!same) {
                    for (int g2 = 0; g2 < ng; ++g2) {
                        int oi[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            oi[g] = idx[0];
#pragma unroll
                            for (int gg = 1; gg < G; ++gg)
                                if (gg == g2) oi[g] = idx[gg];
                        }
                        if (g2 > 0) __syncthreads();
                        scan2_group<M, NR, U, G, NW>(items, oi, 1, T, T32, codes, ids, K, L, S, margin, item_surv, item_n, qbound, smem);
                    }
                    continue;
                }
            }
            scan2_group<M, NR, U, G, NW>(items, idx, ng, T, T32, codes, ids, K, L, S, margin, item_surv, item_n, qbound, smem);
        }
    }
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
        int ns = 64;  // sort only the next power of two above what was filled
        while (ns < n) ns <<= 1;
        for (int x = n + tid; x < ns; x += blockDim.x) { ka[x] = ~0ull; kb[x] = ~0ull; pay[x] = -1; }
        __syncthreads();
        block_bitonic_rt<256, true>(ka, kb, pay, ns);
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
                                                     const int64_t* __restrict__ item_off, int limit, int S,
                                                     cis_hit* __restrict__ out_hits, int64_t* __restrict__ out_ids,
                                                     double* __restrict__ out_dists, int* __restrict__ out_n,
                                                     int32_t* __restrict__ out_cells, uint32_t* __restrict__ out_pos) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* ka = reinterpret_cast<uint64_t*>(smem);
    uint64_t* kb = ka + CAPM;
    int64_t* pay = reinterpret_cast<int64_t*>(kb + CAPM);
    int* s_n = reinterpret_cast<int*>(pay + CAPM);
    const int q = blockIdx.x;
    const int64_t a = item_off[q], b = item_off[q + 1];
    const int64_t o = (int64_t)q * limit;
    merge_lists<CAPM>(item_hits, item_n, a, (int)(b - a), (int64_t)S, S, limit, ka, kb, pay, s_n,
                      out_hits ? out_hits + o : nullptr, out_ids ? out_ids + o : nullptr,
                      out_dists ? out_dists + o : nullptr, out_n ? out_n + q : nullptr,
                      out_cells ? out_cells + o : nullptr, out_pos ? out_pos + o : nullptr);
}

// The float32-prefilter scan hands over, per work item, the (float32 distance << 32 | position) pairs that survived
// its bounds: a superset of the item's exact top `limit`.  One WAVE per query (four queries per workgroup, no
// workgroup barrier: a query is a chain of dependent global loads, so what counts is how many queries a CU has in
// flight) re-scores them exactly -- the code from the index, float64 table entries summed left to right as
// search.py:173, one candidate per lane -- and ranks them by (dist, visit_rank, pos) with a bitonic sort in LDS.
static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ascending bitonic sort of N (power of two >= 64) 128-bit keys (ka, kb) by one wave
static __device__ __forceinline__ void wave_bitonic_lds(uint64_t* ka, uint64_t* kb, int N) {
    const int lane = threadIdx.x & 63;
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (N >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i + j;
                const bool asc = ((i & k) == 0);
                const uint64_t a0 = ka[i], b0 = kb[i], a1 = ka[p], b1 = kb[p];
                const bool gt = (a0 > a1) || (a0 == a1 && b0 > b1);
                if (gt == asc) { ka[i] = a1; kb[i] = b1; ka[p] = a0; kb[p] = b0; }
            }
            wave_lds_sync();
        }
    }
}

template <int CAPM, int MT /* 4, 8, 16, or 0 = any M */, int NEMAX /* 4 or 8: survivors per lane the fast path may hold */>
#ifndef CIS_MERGE_WPE
#define CIS_MERGE_WPE 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MT == 16 || MT == 0) ? 2 : CIS_MERGE_WPE, 8)))
void k_merge_survivors(const uint64_t* __restrict__ surv /* [n_items][S] */,
                                                         const int* __restrict__ item_n, const int64_t* __restrict__ item_off,
                                                         const WorkItem* __restrict__ items, const double* __restrict__ T,
                                                         const uint8_t* __restrict__ codes, const int64_t* __restrict__ ids,
                                                         int nq, int M, int K, int limit, int S,
                                                         cis_hit* __restrict__ out_hits, int64_t* __restrict__ out_ids,
                                                         double* __restrict__ out_dists, int* __restrict__ out_n,
                                                         int32_t* __restrict__ out_cells, uint32_t* __restrict__ out_pos,
                                                         const PlanOut* __restrict__ plan, int32_t* __restrict__ out_visited,
                                                         const float* __restrict__ item_slack /* null: the survivors' high words are float32
                                                         distances within eps of the exact ones (k_adc_scan2); else they are upper bounds of
                                                         the exact distances, at most item_slack[item] above them (k_adc_scan3) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wq;
    if (q >= nq) return;  // whole wave; nothing below synchronises across waves
    if (lane == 0 && out_visited) out_visited[q] = plan[q].visited;
    uint64_t* ka = reinterpret_cast<uint64_t*>(smem) + (size_t)wq * 2 * CAPM;
    uint64_t* kb = ka + CAPM;  // (visit_rank << 32) | position inside the cell
    const int64_t first = item_off[q];
    const int n_lists = (int)(item_off[q + 1] - first);
    const int nf = M / 2;
    int have = 0, l = 0, e = 0, total = 0;
    bool done_fast = false;
    if constexpr (MT != 0) {
        // Usual case: <= 4 lists, <= 256 survivors in all.  The survivors carry their float32 distances: find the value v
        // that `limit` of them do not exceed (ballot bisection in registers) and drop everything above v*(1+3eps) BEFORE
        // the exact re-scoring -- strictly worse, in exact arithmetic, than the `limit` entries below v (the scan's own
        // argument).  ~limit/130 of the table reads, and the sort runs on 128 keys instead of 256.
        int cntl[4] = {0, 0, 0, 0}, n_total = 0;
        if (n_lists <= 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < n_lists) cntl[i] = item_n[first + i];
                n_total += cntl[i];
            }
        }
        auto fast = [&](auto ne_tag) -> bool {
            constexpr int NE = decltype(ne_tag)::value;  // survivors per lane
            uint32_t hi[NE], pp[NE];
            int li[NE];
            bool valid[NE];
            uint32_t mn = 0xffffffffu, mx = 0u;
            // (the lane's NE survivors and their lists' scales are read together, from clamped addresses: as `valid ? surv[..] : ~0`
            // they were NE round trips one after the other at the head of every query's chain of dependent loads)
            uint64_t ent[NE];
            float ubv[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int x = lane + 64 * i;
                valid[i] = x < n_total;
                int lst = 0, off = x;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (lst == j && off >= cntl[j]) { off -= cntl[j]; lst = j + 1; }
                li[i] = lst;
                ent[i] = surv[valid[i] ? (first + lst) * (int64_t)S + off : first * (int64_t)S];
            }
            if (item_slack) {
#pragma unroll
                for (int i = 0; i < NE; ++i) ubv[i] = item_slack[2 * (first + (valid[i] ? li[i] : 0)) + 1];
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                hi[i] = valid[i] ? (uint32_t)(ent[i] >> 32) : 0xffffffffu;
                pp[i] = valid[i] ? (uint32_t)ent[i] : 0xffffffffu;
                if (item_slack && valid[i])  // k_adc_scan3 hands over the 16-bit sum: (sum + M) * ub >= the exact distance
                    hi[i] = __float_as_uint(__double2float_ru((double)(hi[i] + (uint32_t)MT) * (double)ubv[i]));
                mn = (valid[i] && hi[i] < mn) ? hi[i] : mn;
                mx = (valid[i] && hi[i] > mx) ? hi[i] : mx;
            }
            uint32_t thr = 0xffffffffu;
            float vub = __int_as_float(0x7f800000);  // scan3 survivors: a distance that `limit` of them do not exceed
            float sl[4] = {0.f, 0.f, 0.f, 0.f};
            if (n_total > limit) {
                wave_minmax_step<1>(mn, mx); wave_minmax_step<2>(mn, mx); wave_minmax_step<4>(mn, mx);
                wave_minmax_step<8>(mn, mx); wave_minmax_step<16>(mn, mx); wave_minmax_step<32>(mn, mx);
                mn = (uint32_t)__builtin_amdgcn_readfirstlane((int)mn);
                mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
                const uint32_t v = wave_kth_bisect<NE>(hi, valid, mn, mx, limit);
                if (item_slack) {
                    vub = __uint_as_float(v);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sl[i] = (i < n_lists) ? item_slack[2 * (first + i)] : 0.f;
                } else {
                    const float margin = 1.0f + 3.0f * (2.0f * (float)MT * 5.9604645e-8f);
                    thr = __float_as_uint(__double2float_ru((double)__uint_as_float(v) * (double)margin));
                }
            }
            int kept = 0;
            int idx[NE];
            bool keep[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                keep[i] = valid[i] && hi[i] <= thr;
                if (item_slack) {  // strictly worse than `limit` others only if even its lower bound is above vub
                    const float s_i = li[i] == 0 ? sl[0] : (li[i] == 1 ? sl[1] : (li[i] == 2 ? sl[2] : sl[3]));
                    keep[i] = valid[i] && ((double)__uint_as_float(hi[i]) - (double)s_i <= (double)vub);  // exact in float64
                }
                const unsigned long long m = __ballot(keep[i]);
                idx[i] = kept + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                kept += __popcll(m);
            }
            if (kept > CAPM) return false;  // a crowd of equal float32 distances: the general rounds below
            WorkItem its[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) its[i] = items[first + (i < n_lists ? i : 0)];
            // exact re-scoring of the kept entries, four per lane at a time: codes first, then 4 x M table entries in flight
#pragma unroll
            for (int i0 = 0; i0 < NE; i0 += 4) {
                int64_t startv[4];
                const double* t0v[4];
                const double* t1v[4];
                uint32_t rankv[4], pos0v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    WorkItem it = its[0];
#pragma unroll
                    for (int j = 1; j < 4; ++j)
                        if (li[i0 + i] == j) it = its[j];
                    startv[i] = it.start; rankv[i] = (uint32_t)it.rank; pos0v[i] = (uint32_t)it.pos0;
                    t0v[i] = T + (int64_t)it.tab0 * nf * K;
                    t1v[i] = T + (int64_t)it.tab1 * nf * K;
                }
                CodeWords<MT> cw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) cw[i] = load_code<MT>(codes, startv[i] + (keep[i0 + i] ? pp[i0 + i] : 0u));
                double dd[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) dd[i] = adc64_words<MT>(cw[i].w, K, t0v[i], t1v[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (keep[i0 + i]) {
                        ka[idx[i0 + i]] = (uint64_t)__double_as_longlong(dd[i]);
                        kb[idx[i0 + i]] = ((uint64_t)rankv[i] << 32) | (uint32_t)(pos0v[i] + pp[i0 + i]);
                    }
                }
            }
            int ns = 64;
            while (ns < kept) ns <<= 1;
            for (int x = kept + lane; x < ns; x += 64) { ka[x] = ~0ull; kb[x] = ~0ull; }
            wave_lds_sync();
            wave_bitonic_lds(ka, kb, ns);
            total = kept;
            return true;
        };
        if (n_lists >= 1 && n_lists <= 4) {
            if (n_total <= 256) done_fast = fast(std::integral_constant<int, 4>());
            else if (NEMAX >= 8 && n_total <= 512) {
                if constexpr (NEMAX >= 8) done_fast = fast(std::integral_constant<int, 8>());
            }
        }
    }
    while (!done_fast) {  // rounds: append up to CAPM - have entries, sort, keep `limit`
        int n = have;
        int room = CAPM - have;
        while (l < n_lists && room > 0) {
            const int valid = item_n[first + l];
            const int take = (valid - e < room) ? (valid - e) : room;
            if (take > 0) {
                const WorkItem it = items[first + l];
                const double* t0 = T + (int64_t)it.tab0 * nf * K;
                const double* t1 = T + (int64_t)it.tab1 * nf * K;
                const uint64_t* src = surv + (first + l) * (int64_t)S + e;
                if constexpr (MT == 0) {
                    for (int x = lane; x < take; x += 64) {
                        const uint32_t p = (uint32_t)src[x];
                        ka[n + x] = (uint64_t)__double_as_longlong(adc64_global(codes, it.start + p, M, K, t0, t1));
                        kb[n + x] = ((uint64_t)(uint32_t)it.rank << 32) | (uint32_t)(it.pos0 + (int)p);
                    }
                } else {
                    // four candidates per lane at a time: positions, then codes, then 4 x M table entries in flight together
                    for (int x0 = 0; x0 < take; x0 += 256) {
                        uint32_t p[4];
                        bool on[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int x = x0 + i * 64 + lane;
                            on[i] = x < take;
                            p[i] = on[i] ? (uint32_t)src[x] : 0u;
                        }
                        CodeWords<MT> cw[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) cw[i] = load_code<MT>(codes, it.start + p[i]);
                        double dd[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) dd[i] = adc64_words<MT>(cw[i].w, K, t0, t1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int x = x0 + i * 64 + lane;
                            if (on[i]) {
                                ka[n + x] = (uint64_t)__double_as_longlong(dd[i]);
                                kb[n + x] = ((uint64_t)(uint32_t)it.rank << 32) | (uint32_t)(it.pos0 + (int)p[i]);
                            }
                        }
                    }
                }
            }
            n += take;
            total += take;
            room -= take;
            e += take;
            if (e >= valid) { ++l; e = 0; }
        }
        int ns = 64;
        while (ns < n) ns <<= 1;
        for (int x = n + lane; x < ns; x += 64) { ka[x] = ~0ull; kb[x] = ~0ull; }
        wave_lds_sync();
        wave_bitonic_lds(ka, kb, ns);
        have = n < limit ? n : limit;
        if (l >= n_lists) break;
    }
    const int nv = total < limit ? total : limit;
    const int64_t o = (int64_t)q * limit;
    for (int x = lane; x < limit; x += 64) {
        cis_hit hh;
        hh.dist = __longlong_as_double(0x7ff0000000000000LL);
        hh.visit_rank = 0xffffffffu; hh.pos = 0xffffffffu; hh.id = -1; hh.cell = -1; hh.reserved = 0;
        if (x < nv) {
            const uint32_t rank = (uint32_t)(kb[x] >> 32), pos = (uint32_t)kb[x];
            // the work item this hit came from: same cell (visit rank), chunk that contains the position
            for (int li = 0; li < n_lists; ++li) {
                const WorkItem it = items[first + li];
                const uint32_t rel = pos - (uint32_t)it.pos0;
                if ((uint32_t)it.rank == rank && rel < (uint32_t)it.len) {
                    hh.dist = __longlong_as_double((long long)ka[x]);
                    hh.visit_rank = rank;
                    hh.pos = pos;
                    hh.id = ids[it.start + rel];
                    hh.cell = it.cell;
                    break;
                }
            }
        }
        if (out_hits) out_hits[o + x] = hh;
        if (out_ids) {
            out_ids[o + x] = hh.id;
            out_dists[o + x] = (x < nv) ? hh.dist : __longlong_as_double(0x7ff8000000000000LL);
        }
        if (out_cells) out_cells[o + x] = hh.cell;
        if (out_pos) out_pos[o + x] = hh.pos;
    }
    if (lane == 0 && out_n) out_n[q] = nv;
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
// host: index object -- storage, insert and the get_cell / get_codes readers are in lopq_index.hip
// ================================================================================================

extern "C" int cis_index_set_profiling(cis_index* ix, int enable) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    ix->profiling = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return CIS_OK;
}

// ---- device self test of the wave-level primitives (tests/test_lopq_hip_parity.py) --------------
template <int LJ>
__device__ int selftest_xor(uint32_t v) { return lane_xor<LJ>(v) != (uint32_t)__shfl_xor((int)v, LJ) ? 1 : 0; }

template <int NR>
__device__ int selftest_sort(uint32_t seed) {
    const int lane = threadIdx.x & 63;
    uint32_t k[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        uint32_t x = seed ^ (uint32_t)(lane * NR + r) * 2654435761u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        k[r] = x % 1000u;  // many duplicates
    }
    uint32_t mine[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) mine[r] = k[r];
    wave_bitonic_sort<NR>(k);
    int bad = 0;
    // sortedness: element e <= element e+1 (e = lane*NR + r)
#pragma unroll
    for (int r = 0; r + 1 < NR; ++r) bad += k[r] > k[r + 1];
    const uint32_t nxt = (uint32_t)__shfl_down((int)k[0], 1);
    if (lane < 63) bad += k[NR - 1] > nxt;
    // multiset preserved: compare sums and xors
    uint32_t s0 = 0, s1 = 0, x0 = 0, x1 = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) { s0 += mine[r]; s1 += k[r]; x0 ^= mine[r] * 40503u; x1 ^= k[r] * 40503u; }
    for (int o = 32; o > 0; o >>= 1) {
        s0 += (uint32_t)__shfl_xor((int)s0, o); s1 += (uint32_t)__shfl_xor((int)s1, o);
        x0 ^= (uint32_t)__shfl_xor((int)x0, o); x1 ^= (uint32_t)__shfl_xor((int)x1, o);
    }
    bad += (s0 != s1) + (x0 != x1);
    return bad;
}

__global__ void k_selftest(int* __restrict__ errors) {
    const uint32_t v = (uint32_t)threadIdx.x * 7919u + blockIdx.x * 104729u + 17u;
    int bad = selftest_xor<1>(v) + selftest_xor<2>(v) + selftest_xor<4>(v) + selftest_xor<8>(v) + selftest_xor<16>(v) +
              selftest_xor<32>(v);
    bad += selftest_sort<4>(blockIdx.x * 977u + 1u) + selftest_sort<8>(blockIdx.x * 31u + 5u);
    if (bad) atomicAdd(errors, bad);
}

#ifdef CIS_SCAN_COUNTERS
extern "C" int cis_debug_counters(unsigned long long* out, int reset) {
    CIS_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan_ctr), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        CIS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_scan_ctr), z, sizeof(z)));
    }
    return CIS_OK;
}
#endif

extern "C" int cis_selftest(int* n_errors) {
    CIS_REQUIRE(n_errors != nullptr, "NULL argument");
    CIS_TRY(cis_lazy_init());
    int* d = nullptr;
    CIS_CHECK_HIP(hipMalloc((void**)&d, sizeof(int)));
    CIS_CHECK_HIP(hipMemset(d, 0, sizeof(int)));
    hipLaunchKernelGGL(k_selftest, dim3(64), dim3(64), 0, nullptr, d);
    CIS_CHECK_HIP(hipMemcpy(n_errors, d, sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return CIS_OK;
}

extern "C" int cis_multisequence(const void* X, int x_dtype, const void* C0, const void* C1, int c_dtype, int64_t n, int V,
                                 int h, int max_cells, int32_t* cells, double* dists, int* dist_dtype) {
    CIS_REQUIRE((x_dtype == CIS_F32 || x_dtype == CIS_F64) && (c_dtype == CIS_F32 || c_dtype == CIS_F64), "dtype must be 4 or 8");
    CIS_REQUIRE(n >= 0 && V >= 1 && V <= 65535 && h >= 1 && max_cells >= 1 && (n == 0 || (X && C0 && C1 && cells && dists)),
                "bad arguments");
    if ((int64_t)max_cells > (int64_t)V * V) max_cells = V * V;
    const int ct = (x_dtype == CIS_F32 && c_dtype == CIS_F32) ? CIS_F32 : CIS_F64;
    if (dist_dtype) *dist_dtype = ct;
    if (n == 0) return CIS_OK;
    CIS_TRY(cis_lazy_init());
    const size_t csz = (size_t)ct;
    DevBuf bx, bc, bd, bs, bo, bcell, bdist;
    int rc = CIS_OK;
    auto done = [&](int r) { bx.release(); bc.release(); bd.release(); bs.release(); bo.release(); bcell.release(); bdist.release(); return r; };
    auto up = [&](const void* src, int dt, size_t cnt, DevBuf* b, size_t off_elems) -> int {
        if (dt == ct) { CIS_CHECK_HIP(hipMemcpy((char*)b->p + off_elems * csz, src, cnt * csz, hipMemcpyHostToDevice)); return CIS_OK; }
        std::vector<double> tmp(cnt);
        for (size_t i = 0; i < cnt; ++i) tmp[i] = (double)((const float*)src)[i];
        CIS_CHECK_HIP(hipMemcpy((char*)b->p + off_elems * csz, tmp.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
        return CIS_OK;
    };
    if ((rc = bx.reserve((size_t)n * 2 * h * csz)) != CIS_OK) return done(rc);
    if ((rc = bc.reserve((size_t)2 * V * h * csz)) != CIS_OK) return done(rc);
    if ((rc = up(X, x_dtype, (size_t)n * 2 * h, &bx, 0)) != CIS_OK) return done(rc);
    if ((rc = up(C0, c_dtype, (size_t)V * h, &bc, 0)) != CIS_OK) return done(rc);
    if ((rc = up(C1, c_dtype, (size_t)V * h, &bc, (size_t)V * h)) != CIS_OK) return done(rc);
    if ((rc = bd.reserve((size_t)2 * n * V * csz)) != CIS_OK) return done(rc);
    if ((rc = bs.reserve((size_t)2 * n * V * csz)) != CIS_OK) return done(rc);
    if ((rc = bo.reserve((size_t)2 * n * V * sizeof(uint16_t))) != CIS_OK) return done(rc);
    if ((rc = bcell.reserve((size_t)n * max_cells * 2 * sizeof(int32_t))) != CIS_OK) return done(rc);
    if ((rc = bdist.reserve((size_t)n * max_cells * sizeof(double))) != CIS_OK) return done(rc);
    DevBuf bgrp;
    if ((rc = bgrp.reserve((size_t)4 * V * GRP_SUB * sizeof(int))) != CIS_OK) return done(rc);
    for (int s = 0; s < 2; ++s)
        if ((rc = cis_launch_sqdist_generic(bx.p, ct, 2 * h, s * h, (char*)bc.p + (size_t)s * V * h * csz, n, V, h,
                                            (char*)bd.p + (size_t)s * n * V * csz, nullptr)) != CIS_OK) return done(rc);
    if (ct == CIS_F32) {
        hipLaunchKernelGGL(k_rank<float>, dim3((unsigned)n, 2), dim3(V <= 64 ? 64 : 256), (size_t)V * 8, nullptr, bd.as<float>(), (int)n, V,
                           bo.as<uint16_t>(), bs.as<float>(), bgrp.as<int>());
        hipLaunchKernelGGL(k_multiseq_list<float>, dim3((unsigned)n), dim3(64), (size_t)V * sizeof(int), nullptr, bs.as<float>(),
                           bo.as<uint16_t>(), V, max_cells, bcell.as<int32_t>(), bdist.as<double>());
    } else {
        hipLaunchKernelGGL(k_rank<double>, dim3((unsigned)n, 2), dim3(V <= 64 ? 64 : 256), (size_t)V * 8, nullptr, bd.as<double>(), (int)n, V,
                           bo.as<uint16_t>(), bs.as<double>(), bgrp.as<int>());
        hipLaunchKernelGGL(k_multiseq_list<double>, dim3((unsigned)n), dim3(64), (size_t)V * sizeof(int), nullptr, bs.as<double>(),
                           bo.as<uint16_t>(), V, max_cells, bcell.as<int32_t>(), bdist.as<double>());
    }
    hipError_t e = hipMemcpy(cells, bcell.p, (size_t)n * max_cells * 2 * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(dists, bdist.p, (size_t)n * max_cells * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { cis_set_error("hipMemcpy failed: %s", hipGetErrorString(e)); return done(CIS_EHIP); }
    return done(CIS_OK);
}

extern "C" int cis_index_set_scan_mode(cis_index* ix, int mode) {
    CIS_REQUIRE(ix != nullptr && mode >= 0 && mode <= 7, "bad scan mode");
    ix->force_stream = (mode == 6);
    ix->force_scan5 = (mode == 7);
    ix->force_exact_scan = (mode == 1);
    ix->force_prefilter_scan = (mode >= 2 && mode <= 5) || mode == 7;
    ix->force_scan2 = (mode == 2);
    ix->force_scan3 = (mode == 3 || mode == 4 || mode == 5 || mode == 7);
    ix->force_two_pass = mode == 3 ? 0 : (mode == 4 ? 1 : ((mode == 5 || mode == 7) ? 2 : -1));
    return CIS_OK;
}

extern "C" int cis_index_read_profile(cis_index* ix, double ms[5], int64_t* launches) {
    CIS_REQUIRE(ix != nullptr && ms != nullptr, "NULL argument");
    for (auto& r : ix->prof) {
        hipEvent_t last = r.ev[4] ? r.ev[4] : r.ev[3];
        if (last) CIS_CHECK_HIP(hipEventSynchronize(last));
        for (int i = 0; i < 4; ++i) {
            if (!r.ev[i] || !r.ev[i + 1]) continue;
            float t = 0.f;
            CIS_CHECK_HIP(hipEventElapsedTime(&t, r.ev[i], r.ev[i + 1]));
            ix->prof_ms[i] += t;
        }
        if (r.has_scan && r.ev[5] && r.ev[3]) {
            float t = 0.f;
            CIS_CHECK_HIP(hipEventElapsedTime(&t, r.ev[5], r.ev[3]));
            ix->prof_ms[4] += t;
            ix->prof_launches += 1;
        }
        for (int i = 0; i < 6; ++i)
            if (r.ev[i]) (void)hipEventDestroy(r.ev[i]);
    }
    ix->prof.clear();
    for (int i = 0; i < 5; ++i) { ms[i] = ix->prof_ms[i]; ix->prof_ms[i] = 0; }
    if (launches) *launches = ix->prof_launches;
    ix->prof_launches = 0;
    return CIS_OK;
}

extern "C" int cis_index_last_scan_kernel(cis_index* ix) { return ix ? ix->last_scan_kernel : 0; }

extern "C" int cis_index_stream_counters(cis_index* ix, int64_t counters[2]) {
    CIS_REQUIRE(ix != nullptr && counters != nullptr, "NULL argument");
    counters[0] = ix->stream_batches;
    counters[1] = ix->stream_fallbacks;
    return CIS_OK;
}

extern "C" int cis_index_last_stats(cis_index* ix, int64_t stats[4]) {
    CIS_REQUIRE(ix != nullptr && stats != nullptr, "NULL argument");
    if (ix->stats_pending_seq) {  // the last batch ran on bounds: its totals are in the pinned words once its plan has run
        if (__atomic_load_n(&ix->h_totals[3], __ATOMIC_ACQUIRE) != ix->stats_pending_seq) CIS_CHECK_HIP(hipDeviceSynchronize());
        ix->stats[0] += ix->h_totals[2];
        ix->stats[1] += ix->h_totals[0];
        ix->stats[2] += ix->h_totals[1];
        ix->stats_pending_seq = 0;
    }
    for (int i = 0; i < 4; ++i) stats[i] = ix->stats[i];
    return CIS_OK;
}

// ================================================================================================
// host: search pipeline
// ================================================================================================
template <int M, int CAP, int U>
static void launch_scan(int64_t n_items, size_t lds, hipStream_t st, const WorkItem* items, const double* T,
                        const uint8_t* codes, const int64_t* ids, int Mrt, int K, int limit, int S, const int* flag,
                        cis_hit* hits, int* hitn) {
    hipLaunchKernelGGL((k_adc_scan<M, CAP, U>), dim3((unsigned)n_items), dim3(256), lds, st, items, T, codes, ids, Mrt,
                       K, limit, S, flag, hits, hitn);
}

// exact float64 kernel (v1); flag != nullptr restricts it to the items the fast kernel flagged
template <int CAP, int U>
static void launch_scan_m(int M, int64_t n_items, hipStream_t st, const WorkItem* items, const double* T,
                          const uint8_t* codes, const int64_t* ids, int K, int limit, int S, const int* flag,
                          cis_hit* hits, int* hitn) {
    const size_t lds = (size_t)CAP * 16 + (size_t)M * K * sizeof(double) + 16;
    switch (M) {
        case 4: launch_scan<4, CAP, U>(n_items, lds, st, items, T, codes, ids, M, K, limit, S, flag, hits, hitn); break;
        case 8: launch_scan<8, CAP, U>(n_items, lds, st, items, T, codes, ids, M, K, limit, S, flag, hits, hitn); break;
        case 16: launch_scan<16, CAP, U>(n_items, lds, st, items, T, codes, ids, M, K, limit, S, flag, hits, hitn); break;
        default: launch_scan<0, CAP, U>(n_items, lds, st, items, T, codes, ids, M, K, limit, S, flag, hits, hitn); break;
    }
}

static void launch_scan_exact(int M, int64_t n_items, hipStream_t st, const WorkItem* items, const double* T,
                              const uint8_t* codes, const int64_t* ids, int K, int L, int S, const int* flag,
                              cis_hit* hits, int* hitn) {
    if (L <= 512) launch_scan_m<1024, 2>(M, n_items, st, items, T, codes, ids, K, L, S, flag, hits, hitn);
    else if (L <= 1024) launch_scan_m<2048, 4>(M, n_items, st, items, T, codes, ids, K, L, S, flag, hits, hitn);
    else launch_scan_m<4096, 4>(M, n_items, st, items, T, codes, ids, K, L, S, flag, hits, hitn);
}

// float32-prefilter kernel (v2): M in {4, 8, 16}, K <= 256, L <= 952 (a wave region of NR * 64 - 8 entries holds L + 64; 16 registers
// per lane above 440, one query per workgroup there: 43-51 KB of LDS, three workgroups per CU)
static const int SCAN2_MAX_LIMIT = 952;
static bool scan2_supported(int M, int K, int L) {
    return (M == 4 || M == 8 || M == 16) && K <= 256 && K % 4 == 0 && L >= 1 && L <= SCAN2_MAX_LIMIT;
}

struct Scan2Geom { int G, NW, U, S; size_t lds; };

// G = 2 queries per workgroup of 8 waves (one 2-query table set shared by 8 waves) when the batch is
// large enough to find pairs; G = 1 with 4 waves for small batches (latency mode).
static Scan2Geom scan2_geom(int M, int K, int L, int nq) {
    Scan2Geom g;
    const int NR = (L <= 184) ? 4 : (L <= 440 ? 8 : 16);
    g.G = (nq >= 64 && NR < 16) ? 2 : 1;  // pairs of queries per workgroup when the batch is large enough to find pairs
    g.NW = 4;
    g.U = (g.G == 2) ? 4 : 2;
    if (const char* e = getenv("CIS_SCAN_GEOM")) {  // experiments: "G,NW,U" out of the instantiated set
        int a = 0, b = 0, c = 0;
        if (NR < 16 && sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && ((a == 1 && b == 4) || (a == 2 && b == 4) || (a == 2 && b == 8 && c == 2) || (a == 2 && (b == 1 || b == 2) && c == 4) || (a == 4 && b == 4 && c == 4 && NR == 4)) && (c == 2 || c == 4)) {
            g.G = a; g.NW = b; g.U = c;
        }
    }
    g.S = g.NW * (NR * 64 - 8);  // survivor slots per work item: a full region per wave
    g.lds = (size_t)K * M * g.G * 4 + (size_t)g.G * g.NW * (NR * 64 - 8 + 64) * 8 + g.G * sizeof(ScanShared) + 16;
    return g;
}

template <int M, int NR, int G, int NW, int U>
static void launch_scan2_t(int64_t n_items, hipStream_t st, const WorkItem* items, const int* slots, const int* n_slots,
                           const double* T, const float* T32, const uint8_t* codes, const int64_t* ids, int K, int L, int S, size_t lds,
                           int* qctr, uint64_t* hits, int* hitn, unsigned long long* qbound) {
    const float eps = 2.0f * (float)M * 5.9604645e-8f;  // 2 * M * 2^-24
    const float margin = 1.0f + 3.0f * eps;
    const int per_cu = (int)(163840 / lds) < (32 / NW) ? (int)(163840 / lds) : (32 / NW);
    const int64_t resident = 256 * (per_cu < 1 ? 1 : per_cu);  // persistent grid: what the chip can hold
    const int64_t want = (n_items + G - 1) / G + 8;
    const unsigned grid = (unsigned)(want < resident ? ((want + 7) / 8) * 8 : resident);
    hipLaunchKernelGGL((k_adc_scan2<M, NR, U, G, NW>), dim3(grid), dim3(NW * 64), lds, st, items, slots, n_slots, T, T32, codes, ids,
                       K, L, S, margin, qctr, hits, hitn, qbound);
}

template <int M, int NR>
static void launch_scan2_mr(const Scan2Geom& g, int64_t n_items, hipStream_t st, const WorkItem* items, const int* slots,
                            const int* n_slots, const double* T, const float* T32, const uint8_t* codes, const int64_t* ids, int K, int L,
                            int* qctr, uint64_t* hits, int* hitn, unsigned long long* qbound) {
#define CIS_SCAN2_CASE(GG, WW, UU)                                                                                    \
    if (g.G == GG && g.NW == WW && g.U == UU) {                                                                       \
        launch_scan2_t<M, NR, GG, WW, UU>(n_items, st, items, slots, n_slots, T, T32, codes, ids, K, L, g.S, g.lds, qctr, hits, hitn, qbound); \
        return;                                                                                                       \
    }
    CIS_SCAN2_CASE(1, 4, 2)
    if constexpr (NR != 16) {  // limit 441 ... 952 (NR = 16): one query per workgroup only
        CIS_SCAN2_CASE(1, 4, 4)
        CIS_SCAN2_CASE(2, 4, 4)
        CIS_SCAN2_CASE(2, 4, 2)
        CIS_SCAN2_CASE(2, 8, 2)
        CIS_SCAN2_CASE(2, 1, 4)
        CIS_SCAN2_CASE(2, 2, 4)
    }
    if constexpr (NR == 4) { CIS_SCAN2_CASE(4, 4, 4) }  // four queries per workgroup: measured slower (72 KB of LDS -> 2 workgroups per CU)
#undef CIS_SCAN2_CASE
}

template <int M>
static void launch_scan2_m(const Scan2Geom& g, int64_t n_items, hipStream_t st, const WorkItem* items, const int* slots,
                           const int* n_slots, const double* T, const float* T32, const uint8_t* codes, const int64_t* ids, int K, int L,
                           int* qctr, uint64_t* hits, int* hitn, unsigned long long* qbound) {
    if (L <= 184) launch_scan2_mr<M, 4>(g, n_items, st, items, slots, n_slots, T, T32, codes, ids, K, L, qctr, hits, hitn, qbound);
    else if (L <= 440) launch_scan2_mr<M, 8>(g, n_items, st, items, slots, n_slots, T, T32, codes, ids, K, L, qctr, hits, hitn, qbound);
    else launch_scan2_mr<M, 16>(g, n_items, st, items, slots, n_slots, T, T32, codes, ids, K, L, qctr, hits, hitn, qbound);
}

static void launch_scan2(int M, const Scan2Geom& g, int64_t n_items, hipStream_t st, const WorkItem* items, const int* slots,
                         const int* n_slots, const double* T, const float* T32, const uint8_t* codes, const int64_t* ids, int K, int L,
                         int* qctr, uint64_t* hits, int* hitn, unsigned long long* qbound) {
    if (M == 4) launch_scan2_m<4>(g, n_items, st, items, slots, n_slots, T, T32, codes, ids, K, L, qctr, hits, hitn, qbound);
    else if (M == 8) launch_scan2_m<8>(g, n_items, st, items, slots, n_slots, T, T32, codes, ids, K, L, qctr, hits, hitn, qbound);
    else launch_scan2_m<16>(g, n_items, st, items, slots, n_slots, T, T32, codes, ids, K, L, qctr, hits, hitn, qbound);
}

// ---- large `limit` (above what the LDS top-k kernels hold): exact distance of EVERY candidate, stable segmented
// sort per query (the merge sort of csrc/lopq_sort.hip; rocPRIM until round 4).  Candidates are laid out in retrieval order (items of a query
// in visit order, positions ascending), so a stable sort on the distance alone yields the (dist, visit_rank, pos)
// ranking = the reference's stable sorted() over the retrieved list (search.py:210).
int cis_seg_sort_u64(void* temp, size_t* temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint64_t* vals_in,
                     uint64_t* vals_out, int64_t n, int nseg, const int64_t* seg_begin, const int64_t* seg_end, hipStream_t st);

__global__ void k_item_lens(const WorkItem* __restrict__ items, int64_t n, int64_t* __restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lens[i] = items[i].len;
}

__global__ void k_seg_begin(const int64_t* __restrict__ cand_start, const int64_t* __restrict__ item_off, int nq, int64_t n_items,
                            int64_t n_cand, int64_t* __restrict__ seg, unsigned long long* __restrict__ qmin,
                            unsigned long long* __restrict__ qmax) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nq) return;
    const int64_t it = item_off[q];
    seg[q] = (q == nq || it >= n_items) ? n_cand : cand_start[it];
    if (qmin && q < nq) { qmin[q] = ~0ull; qmax[q] = 0ull; }
}

// candidate layout for batches of many work items (hundreds of thousands: tiny cells): exclusive scan of the items' lengths
// over tiles of 4096 items -- tile sums, one workgroup scans them, tiles rescan with their offset -- then k_seg_begin.
// (k_cand_layout below does it all in one workgroup: the right thing for the usual few thousand items, 2 ms for a million.)
static const int CAND_TILE = 4096;
__global__ __launch_bounds__(256) void k_cand_tile_sum(const WorkItem* __restrict__ items, int64_t n_items, int64_t* __restrict__ tile_sums) {
    __shared__ int64_t s_w[4];
    const int64_t base = (int64_t)blockIdx.x * CAND_TILE;
    int64_t sum = 0;
    for (int e = threadIdx.x; e < CAND_TILE; e += 256)
        if (base + e < n_items) sum += items[base + e].len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(1024) void k_cand_tile_scan(int64_t* __restrict__ tile_sums, int64_t ntiles) {  // in place, exclusive
    __shared__ int64_t s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int64_t carry = 0;
    for (int64_t b0 = 0; b0 < ntiles; b0 += 1024) {
        const int64_t i = b0 + tid;
        const int64_t v = i < ntiles ? tile_sums[i] : 0;
        int64_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        __syncthreads();
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        int64_t base = carry, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) base += s_w[w]; tot += s_w[w]; }
        if (i < ntiles) tile_sums[i] = base + x - v;
        carry += tot;
    }
}

__global__ __launch_bounds__(256) void k_cand_tile_apply(const WorkItem* __restrict__ items, int64_t n_items,
                                                         const int64_t* __restrict__ tile_off, int64_t* __restrict__ cand_start) {
    __shared__ int64_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * CAND_TILE;
    int64_t carry = tile_off[blockIdx.x];
    for (int e0 = 0; e0 < CAND_TILE; e0 += 256) {
        const int64_t i = base + e0 + tid;
        const int64_t v = i < n_items ? items[i].len : 0;
        int64_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        __syncthreads();
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        int64_t b = carry;
        for (int w = 0; w < wv; ++w) b += s_w[w];
        if (i < n_items) cand_start[i] = b + x - v;
        carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
}

// candidate layout of the all-candidates path in one launch: exclusive scan of the work items' lengths (cand_start),
// the per-query segment starts (seg[q] = first candidate of query q, seg[nq] = n_cand) and the reset of the key ranges
__global__ __launch_bounds__(1024) void k_cand_layout(const WorkItem* __restrict__ items, int64_t n_items, const int64_t* __restrict__ item_off,
                                                      int nq, int64_t n_cand, int64_t* cand_start, int64_t* __restrict__ seg,
                                                      unsigned long long* __restrict__ qmin, unsigned long long* __restrict__ qmax,
                                                      const int64_t* __restrict__ d_totals /* null, or the plan totals (arguments are bounds) */) {
    __shared__ int64_t s_w[16];
    if (d_totals) { n_items = d_totals[0]; n_cand = d_totals[2]; }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t chunk = (n_items + 1023) / 1024;
    const int64_t lo = tid * chunk, hi = lo + chunk < n_items ? lo + chunk : n_items;
    int64_t sum = 0;
    for (int64_t i = lo; i < hi; ++i) sum += items[i].len;
    int64_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_w[wv] = x;
    __syncthreads();
    int64_t run = x - sum;
    for (int w = 0; w < wv; ++w) run += s_w[w];
    for (int64_t i = lo; i < hi; ++i) {
        __hip_atomic_store(&cand_start[i], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run += items[i].len;
    }
    __threadfence();
    __syncthreads();
    for (int q = tid; q <= nq; q += 1024) {
        const int64_t it = item_off[q];
        seg[q] = (q == nq || it >= n_items) ? n_cand : __hip_atomic_load(&cand_start[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (qmin && q < nq) { qmin[q] = ~0ull; qmax[q] = 0ull; }
    }
}

// per-query range of the keys (for the selection kernel): workgroup reduction in LDS, one pair of global atomics
__device__ __forceinline__ void publish_key_range(uint64_t mn, uint64_t mx, unsigned long long* __restrict__ qmin,
                                                  unsigned long long* __restrict__ qmax, int q) {
    __shared__ unsigned long long s_mm[2];
    if (threadIdx.x == 0) { s_mm[0] = ~0ull; s_mm[1] = 0ull; }
    __syncthreads();
    if (mn <= mx) {
        atomicMin(&s_mm[0], (unsigned long long)mn);
        atomicMax(&s_mm[1], (unsigned long long)mx);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_mm[0] <= s_mm[1]) {
        atomicMin(&qmin[q], s_mm[0]);
        atomicMax(&qmax[q], s_mm[1]);
    }
}

__global__ __launch_bounds__(256) void k_adc_all(const WorkItem* __restrict__ items, const int64_t* __restrict__ cand_start,
                                                 const double* __restrict__ T, const uint8_t* __restrict__ codes, int M, int K,
                                                 uint64_t* __restrict__ keys, uint64_t* __restrict__ vals,
                                                 unsigned long long* __restrict__ qmin, unsigned long long* __restrict__ qmax,
                                                 const int64_t* __restrict__ d_totals /* null, or the plan totals: gridDim.x is a bound */) {
    if (d_totals && (int64_t)blockIdx.x >= d_totals[0]) return;
    const WorkItem it = items[blockIdx.x];
    const int nf = M / 2;
    const double* t0 = T + (int64_t)it.tab0 * nf * K;
    const double* t1 = T + (int64_t)it.tab1 * nf * K;
    const int64_t o = cand_start[blockIdx.x];
    uint64_t mn = ~0ull, mx = 0ull;
    for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < it.len; p += gridDim.y * blockDim.x) {
        const uint64_t kk = (uint64_t)__double_as_longlong(adc64_global(codes, it.start + p, M, K, t0, t1));
        keys[o + p] = kk;
        mn = kk < mn ? kk : mn;
        mx = kk > mx ? kk : mx;
        if (vals) vals[o + p] = ((uint64_t)blockIdx.x << 32) | (uint32_t)p;
    }
    if (qmin) publish_key_range(mn, mx, qmin, qmax, it.q);
}

// the same with the item's two table halves staged in LDS (M in {4, 8, 16}, K <= 256): the lookups of a cell's candidates
// are random reads of 2 * nf * K float64 entries, served from LDS instead of L2
template <int MT>
__global__ __launch_bounds__(256) void k_adc_all_lds(const WorkItem* __restrict__ items, const int64_t* __restrict__ cand_start,
                                                     const double* __restrict__ T, const uint8_t* __restrict__ codes, int K,
                                                     uint64_t* __restrict__ keys, uint64_t* __restrict__ vals,
                                                     unsigned long long* __restrict__ qmin, unsigned long long* __restrict__ qmax,
                                                     const int64_t* __restrict__ d_totals /* null, or the plan totals: gridDim.x is a bound */) {
    extern __shared__ __align__(16) double adc_tab[];  // [MT][K]
    if (d_totals && (int64_t)blockIdx.x >= d_totals[0]) return;
    const WorkItem it = items[blockIdx.x];
    int nch = (it.len + 2047) / 2048;  // workgroups that share this item (<= gridDim.y): ~2048 candidates each at least
    nch = nch < 1 ? 1 : (nch > (int)gridDim.y ? (int)gridDim.y : nch);
    if ((int)blockIdx.y >= nch) return;
    constexpr int nf = MT / 2;
    const double* t0 = T + (int64_t)it.tab0 * nf * K;
    const double* t1 = T + (int64_t)it.tab1 * nf * K;
    for (int e = threadIdx.x; e < nf * K; e += 256) {
        adc_tab[e] = t0[e];
        adc_tab[nf * K + e] = t1[e];
    }
    __syncthreads();
    const double* l0 = adc_tab;
    const double* l1 = adc_tab + nf * K;
    const int64_t o = cand_start[blockIdx.x];
    const int step = nch * 256;
    int p = blockIdx.y * 256 + threadIdx.x;
    uint64_t mn = ~0ull, mx = 0ull;
    for (; p + step < it.len; p += 2 * step) {  // two candidates in flight
        const CodeWords<MT> ca = load_code<MT>(codes, it.start + p);
        const CodeWords<MT> cb = load_code<MT>(codes, it.start + p + step);
        const uint64_t ka = (uint64_t)__double_as_longlong(adc64_words<MT>(ca.w, K, l0, l1));
        const uint64_t kb = (uint64_t)__double_as_longlong(adc64_words<MT>(cb.w, K, l0, l1));
        keys[o + p] = ka;
        keys[o + p + step] = kb;
        const uint64_t lo = ka < kb ? ka : kb, hi = ka < kb ? kb : ka;
        mn = lo < mn ? lo : mn;
        mx = hi > mx ? hi : mx;
        if (vals) {
            vals[o + p] = ((uint64_t)blockIdx.x << 32) | (uint32_t)p;
            vals[o + p + step] = ((uint64_t)blockIdx.x << 32) | (uint32_t)(p + step);
        }
    }
    if (p < it.len) {
        const CodeWords<MT> ca = load_code<MT>(codes, it.start + p);
        const uint64_t ka = (uint64_t)__double_as_longlong(adc64_words<MT>(ca.w, K, l0, l1));
        keys[o + p] = ka;
        mn = ka < mn ? ka : mn;
        mx = ka > mx ? ka : mx;
        if (vals) vals[o + p] = ((uint64_t)blockIdx.x << 32) | (uint32_t)p;
    }
    if (qmin) publish_key_range(mn, mx, qmin, qmax, it.q);
}

// The same for indexes of tiny cells (thousands of coarse clusters: a query visits hundreds of cells of a few codes each):
// one workgroup row per QUERY walks the query's candidates as one flat range; a candidate finds its work item by binary
// search over the items' first-candidate positions (staged in LDS) and reads its table entries from global memory (the
// query's ~100 half tables are L2-resident).  A workgroup per work item would stage 16 KB of tables for 8 candidates.
static const int FLAT_ITEMS = 4096;  // work items of a query whose starts fit the LDS stage; above: searched in global memory
template <int MT>
__global__ __launch_bounds__(256) void k_adc_all_flat(const WorkItem* __restrict__ items, const int64_t* __restrict__ cand_start,
                                                      const int64_t* __restrict__ seg, const int64_t* __restrict__ item_off,
                                                      const double* __restrict__ T, const uint8_t* __restrict__ codes, int M, int K,
                                                      uint64_t* __restrict__ keys, uint64_t* __restrict__ vals,
                                                      unsigned long long* __restrict__ qmin, unsigned long long* __restrict__ qmax) {
    __shared__ int s_start[FLAT_ITEMS];
    const int q = blockIdx.x;
    const int64_t it0 = item_off[q], it1 = item_off[q + 1];
    const int64_t c0 = seg[q], c1 = seg[q + 1];
    const int ni = (int)(it1 - it0);
    const int64_t n = c1 - c0;
    const bool staged = ni <= FLAT_ITEMS;
    if (staged)
        for (int i = threadIdx.x; i < ni; i += 256) s_start[i] = (int)(cand_start[it0 + i] - c0);
    __syncthreads();
    const int nf = M / 2;
    uint64_t mn = ~0ull, mx = 0ull;
    for (int64_t c = (int64_t)blockIdx.y * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.y * 256) {
        int lo = 0, hi = ni;  // first item whose start is > c; the item before it holds c (empty items never exist)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const int64_t st_ = staged ? (int64_t)s_start[mid] : cand_start[it0 + mid] - c0;
            if (st_ <= c) lo = mid + 1;
            else hi = mid;
        }
        const int i = lo - 1;
        const WorkItem it = items[it0 + i];
        const int p = (int)(c - (staged ? (int64_t)s_start[i] : cand_start[it0 + i] - c0));
        const double* t0 = T + (int64_t)it.tab0 * nf * K;
        const double* t1 = T + (int64_t)it.tab1 * nf * K;
        double d;
        if constexpr (MT != 0) {
            const CodeWords<MT> cw = load_code<MT>(codes, it.start + p);
            d = adc64_words<MT>(cw.w, K, t0, t1);
        } else {
            d = adc64_global(codes, it.start + p, M, K, t0, t1);
        }
        const uint64_t kk = (uint64_t)__double_as_longlong(d);
        keys[c0 + c] = kk;
        mn = kk < mn ? kk : mn;
        mx = kk > mx ? kk : mx;
        if (vals) vals[c0 + c] = ((uint64_t)(it0 + i) << 32) | (uint32_t)p;
    }
    if (qmin) publish_key_range(mn, mx, qmin, qmax, q);
}

// Tiny cells, direct form: no distance tables at all.  With thousands of coarse clusters a query touches ~100 (split, cluster)
// pairs per 10 k candidates -- more table entries (100 x nf x K) than lookups (10 k x M), 8 KB of float64 per table written and
// gathered back at random (10 GB per 8192 queries at V = 2048).  Here a candidate's entry j is computed where it is needed:
// e_j = sum_i (px[tab][j w + i] - sub[j][code_j][i])^2 in predict_cluster's order -- the expression, operands and summation of
// k_tables_from_px, so the same bits -- with the sub-quantizers of a PHASE (JP of them, <= 128 KB of float64) resident in LDS and
// the projected residuals px (512 B per (query, cluster)) read through L2.  The running sum of search.py:173 crosses the phases
// through the key array: phase p adds its entries, left to right, to what phase p - 1 left.
template <int MT, int W, int JP>
__global__ __launch_bounds__(1024) void k_adc_direct(const WorkItem* __restrict__ items, const int64_t* __restrict__ cand_start,
                                                     const int64_t* __restrict__ seg, const int64_t* __restrict__ item_off,
                                                     const double* __restrict__ px, const double* __restrict__ subs,
                                                     const uint8_t* __restrict__ codes, int K, int h, int phase, int nq,
                                                     uint64_t* __restrict__ keys, uint64_t* __restrict__ vals,
                                                     unsigned long long* __restrict__ qmin, unsigned long long* __restrict__ qmax) {
    extern __shared__ __align__(16) double s_sub[];  // [JP][K][W], then the item starts of the current query
    int* s_start = reinterpret_cast<int*>(s_sub + (size_t)JP * K * W);
    constexpr int nf = MT / 2;
    constexpr int NPH = MT / JP;
    const int j0 = phase * JP;
    {
        const double2* src = reinterpret_cast<const double2*>(subs + (size_t)j0 * K * W);
        double2* dst = reinterpret_cast<double2*>(s_sub);
        for (int i = threadIdx.x; i < JP * K * W / 2; i += 1024) dst[i] = src[i];
    }
    const bool last = phase == NPH - 1;
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        const int64_t it0 = item_off[q], it1 = item_off[q + 1];
        const int64_t c0 = seg[q], c1 = seg[q + 1];
        const int ni = (int)(it1 - it0);
        const int64_t n = c1 - c0;
        const bool staged = ni <= FLAT_ITEMS;
        __syncthreads();  // the previous query's starts are no longer read (and the sub-quantizers are in place)
        if (staged)
            for (int i = threadIdx.x; i < ni; i += 1024) s_start[i] = (int)(cand_start[it0 + i] - c0);
        __syncthreads();
        uint64_t mn = ~0ull, mx = 0ull;
        for (int64_t c = threadIdx.x; c < n; c += 1024) {
            int lo = 0, hi = ni;  // first item whose start is > c; the item before it holds c
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int64_t st_ = staged ? (int64_t)s_start[mid] : cand_start[it0 + mid] - c0;
                if (st_ <= c) lo = mid + 1;
                else hi = mid;
            }
            const int i = lo - 1;
            const WorkItem it = items[it0 + i];
            const int p = (int)(c - (staged ? (int64_t)s_start[i] : cand_start[it0 + i] - c0));
            const CodeWords<MT> cw = load_code<MT>(codes, it.start + p);
            double d = phase == 0 ? 0.0 : __longlong_as_double((long long)keys[c0 + c]);
#pragma unroll
            for (int jj = 0; jj < JP; ++jj) {
                const int j = j0 + jj;
                const bool second = j >= nf;
                const double* f = px + (int64_t)(second ? it.tab1 : it.tab0) * h + (second ? j - nf : j) * W;
                const uint32_t code = (cw.w[j >> 2] >> (8 * (j & 3))) & 255u;
                const double* sc = s_sub + ((size_t)jj * K + code) * W;
                auto elem = [&](int e) -> double { const double df = f[e] - sc[e]; return df * df; };
                const double v = pw_leaf<double>(elem, 0, W);
                d = (phase == 0 && jj == 0) ? v : d + v;
            }
            const uint64_t kk = (uint64_t)__double_as_longlong(d);
            keys[c0 + c] = kk;
            if (last) {
                mn = kk < mn ? kk : mn;
                mx = kk > mx ? kk : mx;
                if (vals) vals[c0 + c] = ((uint64_t)(it0 + i) << 32) | (uint32_t)p;
            }
        }
        if (last && qmin) publish_key_range(mn, mx, qmin, qmax, q);
    }
}

// sub-quantizers per phase of the direct form: the largest divisor of M whose float64 centroids fit 128 KB of LDS; 0 = not served
static int direct_jp(int M, int K, int w) {
    if (!(M == 4 || M == 8 || M == 16) || K > 256 || K % 2 != 0 || !(w == 4 || w == 8 || w == 16 || w == 32)) return 0;
    return M < 64 / w ? M : 64 / w;  // sized for K = 256 (the instantiated set below)
}

template <int MT, int W, int JP>
static void launch_adc_direct_t(hipStream_t st, const WorkItem* items, const int64_t* cand_start, const int64_t* seg,
                                const int64_t* item_off, const double* px, const double* subs, const uint8_t* codes, int K, int h,
                                int nq, uint64_t* keys, uint64_t* vals, unsigned long long* qmin, unsigned long long* qmax) {
    const size_t lds = (size_t)JP * K * W * sizeof(double) + (size_t)FLAT_ITEMS * sizeof(int);
    const unsigned grid = (unsigned)(nq < 256 ? nq : 256);
    for (int ph = 0; ph < MT / JP; ++ph)
        hipLaunchKernelGGL((k_adc_direct<MT, W, JP>), dim3(grid), dim3(1024), lds, st, items, cand_start, seg, item_off, px, subs, codes, K, h,
                           ph, nq, keys, vals, qmin, qmax);
}

static bool launch_adc_direct(int M, int K, int w, hipStream_t st, const WorkItem* items, const int64_t* cand_start, const int64_t* seg,
                              const int64_t* item_off, const double* px, const double* subs, const uint8_t* codes, int h, int nq,
                              uint64_t* keys, uint64_t* vals, unsigned long long* qmin, unsigned long long* qmax) {
    const int jp = direct_jp(M, K, w);
#define CIS_DIRECT(MT, W, JP)                                                                                                     \
    if (M == MT && w == W && jp == JP) {                                                                                          \
        launch_adc_direct_t<MT, W, JP>(st, items, cand_start, seg, item_off, px, subs, codes, K, h, nq, keys, vals, qmin, qmax);  \
        return true;                                                                                                              \
    }
    // K = 256: 128 KB hold 64 / w sub-quantizers
    CIS_DIRECT(8, 16, 4) CIS_DIRECT(16, 16, 4) CIS_DIRECT(4, 16, 4) CIS_DIRECT(8, 8, 8) CIS_DIRECT(16, 8, 8) CIS_DIRECT(4, 8, 4)
    CIS_DIRECT(4, 32, 2) CIS_DIRECT(8, 32, 2) CIS_DIRECT(16, 32, 2) CIS_DIRECT(4, 4, 4) CIS_DIRECT(8, 4, 8) CIS_DIRECT(16, 4, 16)
#undef CIS_DIRECT
    return false;
}

static void launch_adc_all(int64_t n_items, hipStream_t st, const WorkItem* items, const int64_t* cand_start, const double* T,
                           const uint8_t* codes, int M, int K, uint64_t* keys, uint64_t* vals, unsigned long long* qmin,
                           unsigned long long* qmax, const int64_t* d_totals = nullptr, const int64_t* seg = nullptr,
                           const int64_t* item_off = nullptr, int nq = 0, bool flat = false) {
    if (flat && seg && item_off && nq > 0 && !d_totals) {
        const dim3 gf((unsigned)nq, 8);
        if (K <= 256 && M == 4) hipLaunchKernelGGL(k_adc_all_flat<4>, gf, dim3(256), 0, st, items, cand_start, seg, item_off, T, codes, M, K, keys, vals, qmin, qmax);
        else if (K <= 256 && M == 8) hipLaunchKernelGGL(k_adc_all_flat<8>, gf, dim3(256), 0, st, items, cand_start, seg, item_off, T, codes, M, K, keys, vals, qmin, qmax);
        else if (K <= 256 && M == 16) hipLaunchKernelGGL(k_adc_all_flat<16>, gf, dim3(256), 0, st, items, cand_start, seg, item_off, T, codes, M, K, keys, vals, qmin, qmax);
        else hipLaunchKernelGGL(k_adc_all_flat<0>, gf, dim3(256), 0, st, items, cand_start, seg, item_off, T, codes, M, K, keys, vals, qmin, qmax);
        return;
    }
    const dim3 g((unsigned)n_items, 8);
    const size_t lds = (size_t)M * K * sizeof(double);
    if (K <= 256 && M == 4) hipLaunchKernelGGL(k_adc_all_lds<4>, g, dim3(256), lds, st, items, cand_start, T, codes, K, keys, vals, qmin, qmax, d_totals);
    else if (K <= 256 && M == 8) hipLaunchKernelGGL(k_adc_all_lds<8>, g, dim3(256), lds, st, items, cand_start, T, codes, K, keys, vals, qmin, qmax, d_totals);
    else if (K <= 256 && M == 16) hipLaunchKernelGGL(k_adc_all_lds<16>, g, dim3(256), lds, st, items, cand_start, T, codes, K, keys, vals, qmin, qmax, d_totals);
    else hipLaunchKernelGGL(k_adc_all, g, dim3(256), 0, st, items, cand_start, T, codes, M, K, keys, vals, qmin, qmax, d_totals);
}

__global__ void k_emit_sorted(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals, const int64_t* __restrict__ seg,
                              const int* __restrict__ nsel, int64_t stride,
                              const WorkItem* __restrict__ items, const int64_t* __restrict__ ids, int nq, int limit,
                              cis_hit* __restrict__ out_hits, int64_t* __restrict__ out_ids, double* __restrict__ out_dists,
                              int* __restrict__ out_n, int32_t* __restrict__ out_cells, uint32_t* __restrict__ out_pos) {
    const int q = blockIdx.y;
    // segments: contiguous (seg[q] .. seg[q + 1]) or, after the selection kernel, nsel[q] entries at q * stride
    const int64_t a = nsel ? (int64_t)q * stride : seg[q], n = nsel ? (int64_t)nsel[q] : seg[q + 1] - a;
    const int nv = (int)(n < limit ? n : limit);
    const int64_t o = (int64_t)q * limit;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < limit; x += gridDim.x * blockDim.x) {
        cis_hit hh;
        hh.dist = __longlong_as_double(0x7ff0000000000000LL);
        hh.visit_rank = 0xffffffffu; hh.pos = 0xffffffffu; hh.id = -1; hh.cell = -1; hh.reserved = 0;
        if (x < nv) {
            const uint64_t v = vals[a + x];
            const WorkItem it = items[v >> 32];
            const uint32_t p = (uint32_t)v;
            hh.dist = __longlong_as_double((long long)keys[a + x]);
            hh.visit_rank = (uint32_t)it.rank;
            hh.pos = (uint32_t)it.pos0 + p;
            hh.id = ids[it.start + p];
            hh.cell = it.cell;
        }
        if (out_hits) out_hits[o + x] = hh;
        if (out_ids) {
            out_ids[o + x] = hh.id;
            out_dists[o + x] = (x < nv) ? hh.dist : __longlong_as_double(0x7ff8000000000000LL);
        }
        if (out_cells) out_cells[o + x] = hh.cell;
        if (out_pos) out_pos[o + x] = hh.pos;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && out_n) out_n[q] = nv;
}


// ---- `limit` between the float32-prefilter kernel's 440 and the candidate count: radix SELECT before any sort.
// k_adc_all has written the exact float64 distance of every candidate in retrieval order (keys; positive doubles
// order like their bit patterns).  One workgroup per query finds the `limit`-th smallest (distance, retrieval index)
// pair -- the cut of the reference's stable sorted()[:limit] (search.py:210-216):
//   A. min / max of the segment -> the bits all keys share are skipped;
//   B. most-significant-digit radix passes (11 bits, LDS histogram) narrow to the bin holding the limit-th key until
//      that bin has <= SEL_CAND keys (or every bit is fixed: a crowd of exact ties);
//   C. the bin's keys are sorted in LDS by (key, index) and the r-th is the cut; for a crowd of ties the cut index is
//      found by counting equal keys in retrieval order;
//   D. an ORDER-PRESERVING gather of everything <= cut (block prefix sums): exactly min(n, limit) pairs in retrieval
//      order.  limit <= 3072: sorted in LDS by (key, index) and written ranked; above: written in retrieval order for
//      the stable segmented sort, which now sees limit instead of n entries per query.
static const int SEL_THREADS = 1024;  // largest workgroup of the selection kernel (small batches); large batches use 512
static const int SEL_BITS = 11;
static const int SEL_CAND = 1024;
static const int SEL_PER = 4;  // consecutive keys per thread and chunk in the ordered passes

__device__ __forceinline__ int sel_wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive prefix of v over the workgroup and the total; sm: one int per wave (contains two barriers)
__device__ __forceinline__ int sel_block_excl_scan(int v, int* sm, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int inc = sel_wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) {
        const int x = sm[i];
        if (i < w) base += x;
        tot += x;
    }
    *total = tot;
    return base + inc - v;
}

__device__ __forceinline__ bool sel_pair_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
    return ka < kb || (ka == kb && ia < ib);
}

// ascending bitonic sort of n2 (power of two) (key, index) pairs in LDS by the whole workgroup
__device__ __forceinline__ void sel_block_bitonic(uint64_t* key, uint32_t* idx, int n2) {
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const uint64_t ka = key[lo], kb = key[hi];
                const uint32_t ia = idx[lo], ib = idx[hi];
                if (sel_pair_less(kb, ib, ka, ia) == up) {
                    key[lo] = kb; key[hi] = ka;
                    idx[lo] = ib; idx[hi] = ia;
                }
            }
        }
    __syncthreads();
}

// the candidate reference (work item << 32 | offset in the item's cell) of retrieval index g (global over the batch)
__device__ __forceinline__ uint64_t sel_val_of(int64_t g, const int64_t* __restrict__ cand_start, int64_t it_lo, int64_t it_hi) {
    while (it_hi - it_lo > 1) {  // last item with cand_start <= g
        const int64_t mid = (it_lo + it_hi) >> 1;
        if (cand_start[mid] <= g) it_lo = mid; else it_hi = mid;
    }
    return ((uint64_t)it_lo << 32) | (uint32_t)(g - cand_start[it_lo]);
}

struct SelShared {
    int bin, less, cnt, cn, on;
    unsigned int cut_idx;
    int wsum[SEL_THREADS / 64];
};

template <bool SORT_LDS, int NT>
__global__ __launch_bounds__(NT) void k_select_topl(const uint64_t* __restrict__ keys, const int64_t* __restrict__ seg,
                                                              const int64_t* __restrict__ cand_start, const int64_t* __restrict__ item_off,
                                                              const unsigned long long* __restrict__ qmin,
                                                              const unsigned long long* __restrict__ qmax, int64_t n_items, int L, int P2,
                                                              int64_t stride, uint64_t* __restrict__ sel_keys,
                                                              uint64_t* __restrict__ sel_vals, int* __restrict__ nsel,
                                                              int64_t* __restrict__ seg_b, int64_t* __restrict__ seg_e,
                                                              const int* __restrict__ only = nullptr /* ranked: flagged queries only */,
                                                              // INDIRECT (the streaming route, SORT_LDS only): the keys of query q are kcnt[q] (<= kstride) entries at
                                                              // q * kstride in arbitrary order, entry i is the candidate of retrieval index rid[q * kstride + i]
                                                              // (relative to seg[q]) -- ties are broken by THAT index, as the stable sort of search.py:210 does
                                                              const uint32_t* __restrict__ rid = nullptr, const int* __restrict__ kcnt = nullptr,
                                                              int64_t kstride = 0) {
    if (only && !only[blockIdx.x]) return;
    extern __shared__ __align__(16) unsigned char sel_lds[];
    // LDS: [okey P2][oidx P2] (SORT_LDS) | union { hist 2048 u32 ; ckey 1024 u64 + cidx 1024 u32 } | SelShared
    uint64_t* okey = reinterpret_cast<uint64_t*>(sel_lds);
    uint32_t* oidx = reinterpret_cast<uint32_t*>(okey + (SORT_LDS ? P2 : 0));
    unsigned char* un = reinterpret_cast<unsigned char*>(oidx + (SORT_LDS ? P2 : 0));
    unsigned int* hist = reinterpret_cast<unsigned int*>(un);
    uint64_t* ckey = reinterpret_cast<uint64_t*>(un);
    uint32_t* cidx = reinterpret_cast<uint32_t*>(ckey + SEL_CAND);
    SelShared* sh = reinterpret_cast<SelShared*>(un + SEL_CAND * 12);
    const int q = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int64_t fa = seg[q];                           // retrieval index of the query's first candidate
    const int64_t a = rid ? (int64_t)q * kstride : fa;   // where the query's keys start
    const int64_t n64 = rid ? (int64_t)(kcnt[q] < kstride ? kcnt[q] : (int)kstride) : seg[q + 1] - a;
    const unsigned int n = (unsigned int)n64;  // a query's candidates fit 32 bits (the batch total does)
    const uint64_t* __restrict__ k = keys + a;
    const uint32_t* __restrict__ ridq = rid ? rid + a : nullptr;
    auto idx_of = [&](unsigned int i) -> uint32_t { return ridq ? ridq[i] : i; };  // the tie-breaking index of key i
    const int nv = n64 < (int64_t)L ? (int)n64 : L;
    uint64_t cut_key = ~0ull;
    unsigned int cut_idx = 0xffffffffu;
    bool gathered = false;  // SORT_LDS: the list is already in LDS (unordered)
    if (n64 > (int64_t)L) {
        // A. range of the keys: published by the distance kernel
        const uint64_t mn = qmin[q], mx = qmax[q];
        int hi_shift = (mn == mx) ? 0 : 64 - __clzll((long long)(mn ^ mx));  // bits [0, hi_shift) differ somewhere
        uint64_t prefix = hi_shift >= 64 ? 0ull : (mn >> hi_shift);
        int r = L;            // rank (1-based) of the cut among the keys that share `prefix`
        unsigned int c = n;   // keys that share `prefix`
        // B. radix passes
        while (hi_shift > 0 && c > (unsigned)SEL_CAND) {
            const int bits = hi_shift < SEL_BITS ? hi_shift : SEL_BITS;
            const int shift = hi_shift - bits;
            const unsigned int mask = (1u << bits) - 1u;
            for (int b = tid; b < (1 << SEL_BITS); b += nt) hist[b] = 0;
            __syncthreads();
#pragma unroll 4
            for (unsigned int i = tid; i < n; i += nt) {
                const uint64_t v = k[i];
                if (hi_shift >= 64 || (v >> hi_shift) == prefix) atomicAdd(&hist[(unsigned int)(v >> shift) & mask], 1u);
            }
            __syncthreads();
            constexpr int per = (1 << SEL_BITS) / NT;
            unsigned int cb[per];
            int local = 0;
#pragma unroll
            for (int j = 0; j < per; ++j) { cb[j] = hist[tid * per + j]; local += (int)cb[j]; }
            int tot;
            int run = sel_block_excl_scan(local, sh->wsum, &tot);
#pragma unroll
            for (int j = 0; j < per; ++j) {
                if (run < r && r <= run + (int)cb[j]) { sh->bin = tid * per + j; sh->less = run; sh->cnt = (int)cb[j]; }
                run += (int)cb[j];
            }
            __syncthreads();
            r -= sh->less;
            c = (unsigned int)sh->cnt;
            prefix = (prefix << bits) | (uint64_t)sh->bin;
            hi_shift = shift;
            __syncthreads();
        }
        if (c <= (unsigned)SEL_CAND) {
            // C. the threshold bin's keys, ranked in LDS
            if (tid == 0) { sh->cn = 0; sh->on = 0; }
            __syncthreads();
#pragma unroll 4
            for (unsigned int i = tid; i < n; i += nt) {
                const uint64_t v = k[i];
                const uint64_t hv = hi_shift >= 64 ? 0ull : (v >> hi_shift);
                if (hv == prefix) {
                    const int j = atomicAdd(&sh->cn, 1);
                    ckey[j] = v;
                    cidx[j] = idx_of(i);
                } else if (SORT_LDS && hv < prefix) {  // below the threshold bin: in, whatever the order (ranked in LDS below)
                    const int j = atomicAdd(&sh->on, 1);
                    okey[j] = v;
                    oidx[j] = idx_of(i);
                }
            }
            __syncthreads();
            int c2 = 64;
            while (c2 < (int)c) c2 <<= 1;
            for (int j = (int)c + tid; j < c2; j += nt) { ckey[j] = ~0ull; cidx[j] = 0xffffffffu; }
            sel_block_bitonic(ckey, cidx, c2);
            cut_key = ckey[r - 1];
            cut_idx = cidx[r - 1];
            if (SORT_LDS) {  // the first r of the bin complete the list
                const int on = sh->on;
                for (int j = tid; j < r; j += nt) { okey[on + j] = ckey[j]; oidx[on + j] = cidx[j]; }
                gathered = true;
            }
            __syncthreads();
        } else if (rid) {
            // INDIRECT: a crowd of more than SEL_CAND exact ties at the cut is not resolved here -- the query reports no result and
            // the streaming route's verification sends it to the generic path
            if (tid == 0) nsel[q] = 0;
            return;
        } else {
            // a crowd of exact ties at the cut (every bit fixed): the first r of them in retrieval order
            cut_key = prefix;
            int base = 0;
            for (unsigned int i0 = 0; i0 < n; i0 += (unsigned)nt) {
                const unsigned int i = i0 + tid;
                const int f = (i < n && k[i] == cut_key) ? 1 : 0;
                int tot;
                const int ex = sel_block_excl_scan(f, sh->wsum, &tot);
                if (f && base + ex == r - 1) sh->cut_idx = i;
                base += tot;
                if (base >= r) break;
            }
            __syncthreads();
            cut_idx = sh->cut_idx;
            __syncthreads();
        }
    }
    // D. order-preserving gather of the pairs <= cut
    const int64_t it_lo = item_off[q];
    int64_t it_hi = item_off[q + 1];
    if (it_hi > n_items) it_hi = n_items;
    const int64_t ob = (int64_t)q * stride;
    int base = 0;
    if (!gathered)
    for (unsigned int i0 = 0; i0 < n && base < nv; i0 += (unsigned)(nt * SEL_PER)) {
        const unsigned int i = i0 + tid * SEL_PER;
        uint64_t v[SEL_PER];
        int f[SEL_PER], local = 0;
#pragma unroll
        for (int j = 0; j < SEL_PER; ++j) {
            v[j] = (i + j < n) ? k[i + j] : 0ull;
            f[j] = (i + j < n) && (v[j] < cut_key || (v[j] == cut_key && (ridq ? ridq[i + j] : i + j) <= cut_idx));
            local += f[j];
        }
        int tot;
        int o = base + sel_block_excl_scan(local, sh->wsum, &tot);
#pragma unroll
        for (int j = 0; j < SEL_PER; ++j)
            if (f[j]) {
                if (SORT_LDS) {
                    okey[o] = v[j];
                    oidx[o] = idx_of(i + j);
                } else {
                    sel_keys[ob + o] = v[j];
                    sel_vals[ob + o] = sel_val_of(a + i + j, cand_start, it_lo, it_hi);
                }
                ++o;
            }
        base += tot;
    }
    if (SORT_LDS) {
        __syncthreads();
        int n2 = 64;
        while (n2 < nv) n2 <<= 1;
        for (int j = nv + tid; j < n2; j += nt) { okey[j] = ~0ull; oidx[j] = 0xffffffffu; }
        sel_block_bitonic(okey, oidx, n2);
        for (int j = tid; j < nv; j += nt) {
            sel_keys[ob + j] = okey[j];
            sel_vals[ob + j] = sel_val_of(fa + oidx[j], cand_start, it_lo, it_hi);
        }
    }
    if (tid == 0) {
        nsel[q] = nv;
        if (seg_b) { seg_b[q] = ob; seg_e[q] = ob + nv; }
    }
}

static const int MAX_LDS_LIMIT = 3072;  // ranked results per query the LDS top-k kernels hold
static const int MAX_LIMIT = 1 << 24;  // with the sorted path: bounded by the workspace only

// how the all-candidates path ranks: select (k_select_topl) when it shrinks the sort, LDS-ranked for limit <= 3072
struct SelectPlan { bool select, sort_lds; int64_t stride; int p2; size_t lds; };
static SelectPlan select_plan(int L, int nq, int64_t n_cand) {
    SelectPlan sp;
    sp.stride = (int64_t)L < n_cand ? (int64_t)L : (n_cand > 0 ? n_cand : 1);
    sp.sort_lds = L <= MAX_LDS_LIMIT;
    sp.select = sp.sort_lds || (double)nq * (double)sp.stride <= 0.5 * (double)n_cand;
    if (getenv("CIS_NO_SELECT")) sp.select = sp.sort_lds = false;  // experiments: the full segmented sort
    if (!sp.select) sp.sort_lds = false;
    sp.p2 = 64;
    while (sp.p2 < L && sp.sort_lds) sp.p2 <<= 1;
    sp.lds = (sp.sort_lds ? (size_t)sp.p2 * 12 : 0) + (size_t)SEL_CAND * 12 + sizeof(SelShared) + 16;
    return sp;
}

// limit above the float32-prefilter kernel's 440: rank through the all-candidates path (the LDS top-k kernel of the exact
// scan holds up to 3072 but slows down steeply with limit)
// Largest batch at which the all-candidates path beats the float32-prefilter scan for limit <= 440 (measured with
// tools/bench_limits.py on the bench index, 40 k candidates per query): a small batch has too few (query, cell) work items
// to fill the persistent scan kernel, and its per-wave regions grow with limit, while scoring every candidate with one
// workgroup per 2048 of them and selecting per query costs the same 0.18-0.4 ms whatever the limit.
static int small_batch_nq(int L) {
    static const int forced = getenv("CIS_SMALL_NQ") ? atoi(getenv("CIS_SMALL_NQ")) : -1;  // experiments
    if (forced >= 0) return forced;
    if (L <= 184) return 63;
    if (L <= 300) return 256;
    return 512;
}

// thousands of coarse clusters: cells of a few codes each (the release configurations' V = 2048 / 4096)
static bool index_has_tiny_cells(const cis_index* ix) {
    return ix->nonempty_cells > 0 && ix->n_total / ix->nonempty_cells < 64;
}

static bool use_all_path(const cis_index* ix, int M, int K, int L, int nq) {
    if (L > MAX_LDS_LIMIT) return true;
    if (ix->force_exact_scan) return false;
    if (L > SCAN2_MAX_LIMIT) return true;
    // a scan workgroup per (query, cell) slot stages 16 KB of tables and a survivor list per work item: hopeless for cells of
    // a few codes -- every candidate's exact distance + a per-query select instead (unless a test forces a scan kernel)
    if (index_has_tiny_cells(ix) && !ix->force_prefilter_scan) return true;
    return !ix->force_prefilter_scan && nq <= small_batch_nq(L);
}

// ---- tiny cells, prefiltered (round 3): the release operating point (V = 2048 / 4096: a query visits hundreds of cells of a
// few codes each, ~10 k candidates for the 100 it returns) -----------------------------------------------------------------
// k_adc_direct computes the exact float64 distance of EVERY candidate (128 subtract-square-adds on operands gathered from LDS
// and L2: 5.8 of the 11.4 ms of an 8192-query batch) and k_select_topl then reads all keys back.  Here one workgroup per query
//   1. lays the query's candidates out in LDS (its work items, the owner item of every candidate, the half tables that have a
//      candidate at all, numbered densely);
//   2. estimates a cap on the limit-th best distance from 128 sampled candidates;
//   3. builds the query's distance tables as BYTES in LDS -- entry = min(255, floor(e / step)), step = cap / 256, e computed in
//      float32 from the projected residuals px (k_tables_group leaves a float32 copy; rows arrive through scalar loads) -- one
//      split and at most `tch` tables at a time, and adds every candidate's table bytes up (a lower bound of its distance in
//      steps: floor and min only round down);
//   4. histograms the sums: s* = the smallest sum that `limit` candidates reach; a candidate whose sum is <= 254 has no clamped
//      entry, so its distance is below (sum + M) steps, hence the limit-th best distance T < (s* + M) steps, and a candidate
//      can only be among the best `limit` if sum <= s* + M (+ 1 bin for the float32 rounding of e, which the cap test below
//      keeps under a sixteenth of a step per entry);
//   5. computes the exact float64 keys of those survivors only -- the expression, operands and summation order of
//      k_adc_direct, so the same bits -- ranks them by (key, retrieval position) by counting and writes the best `limit` in the
//      format of k_select_topl<true>.
// A query for which no bound comes out (fewer than `limit` candidates under the cap, survivors beyond the LDS list, more
// candidates than the layout holds, a cap too small for float32) gets every candidate's exact key written to the global key
// array and a flag; k_select_topl runs afterwards on the flagged queries alone.  Results are those of the exact path, bit for bit.
static const int TINY_NS = 128;     // sampled candidates per query (eight threads each)
static const int TINY_SMAX = 1024;  // survivors ranked in LDS
static const int TINY_NCMAX = 16384;
static const int TINY_TCH = 128;    // tables per chunk at most

struct TinyShared {
    int cnt, sstar, fb, nused[2];
    unsigned int cap_bits, amax_bits, cen_bits;
    int wsum[16];
};

template <int MT, int W>
__device__ __noinline__ void tiny_exact_all(const WorkItem* __restrict__ items, const int64_t* __restrict__ cand_start,
                                            const double* __restrict__ px, const double* __restrict__ subs,
                                            const uint8_t* __restrict__ codes, int h, int64_t it0, int ni, int64_t c0, int64_t n64,
                                            uint64_t* __restrict__ keys_fb,
                                            unsigned long long* __restrict__ qmin, unsigned long long* __restrict__ qmax, int q) {
    constexpr int NF = MT / 2, K = 256;
    uint64_t mn = ~0ull, mx = 0ull;
    for (int64_t c = threadIdx.x; c < n64; c += 1024) {
        int lo = 0, hi = ni;  // the item that holds candidate c
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cand_start[it0 + mid] - c0 <= c) lo = mid + 1; else hi = mid;
        }
        const int i = lo - 1;
        const WorkItem it = items[it0 + i];
        const int p = (int)(c - (cand_start[it0 + i] - c0));
        const CodeWords<MT> cw = load_code<MT>(codes, it.start + p);
        double d = 0.0;
        for (int j = 0; j < MT; ++j) {
            const bool second = j >= NF;
            const double* f = px + (int64_t)(second ? it.tab1 : it.tab0) * h + (second ? j - NF : j) * W;
            const uint32_t code = (cw.w[j >> 2] >> (8 * (j & 3))) & 255u;
            const double* sc = subs + ((size_t)j * K + code) * W;
            auto elem = [&](int e) -> double { const double df = f[e] - sc[e]; return df * df; };
            const double v = pw_leaf<double>(elem, 0, W);
            d = j == 0 ? v : d + v;
        }
        const uint64_t kk = (uint64_t)__double_as_longlong(d);
        keys_fb[c0 + c] = kk;
        mn = kk < mn ? kk : mn;
        mx = kk > mx ? kk : mx;
    }
    publish_key_range(mn, mx, qmin, qmax, q);
}

// Two rows of W float32 at wave-uniform addresses, through the scalar cache into SGPRs: one wait for both (the compiler keeps
// uniform loads of a kernel that also writes global memory on the vector path -- 1 KB of identical lanes per instruction)
typedef int tiny_i16v __attribute__((ext_vector_type(16)));
typedef int tiny_i8v __attribute__((ext_vector_type(8)));
typedef int tiny_i4v __attribute__((ext_vector_type(4)));
// rows per wait: a wave's table loop is bound by the latency of these loads (L2: the rows were just written), not by arithmetic
template <int W> struct TinyRows { static constexpr int U = W <= 16 ? 4 : 2; };
template <int W>
__device__ __forceinline__ void tiny_sload_rows(const float* const (&p)[TinyRows<W>::U], float (&f)[TinyRows<W>::U][W]) {
    if constexpr (W == 4) {
        tiny_i4v a, b, c, d;
        asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %5, 0x0\n\ts_load_dwordx4 %2, %6, 0x0\n\ts_load_dwordx4 %3, %7, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p[0]), "s"(p[1]), "s"(p[2]), "s"(p[3]) : "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[0][e] = __int_as_float(a[e]); f[1][e] = __int_as_float(b[e]); f[2][e] = __int_as_float(c[e]); f[3][e] = __int_as_float(d[e]); }
    } else if constexpr (W == 8) {
        tiny_i8v a, b, c, d;
        asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %5, 0x0\n\ts_load_dwordx8 %2, %6, 0x0\n\ts_load_dwordx8 %3, %7, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p[0]), "s"(p[1]), "s"(p[2]), "s"(p[3]) : "memory");
#pragma unroll
        for (int e = 0; e < 8; ++e) { f[0][e] = __int_as_float(a[e]); f[1][e] = __int_as_float(b[e]); f[2][e] = __int_as_float(c[e]); f[3][e] = __int_as_float(d[e]); }
    } else if constexpr (W == 16) {
        tiny_i16v a, b, c, d;
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %5, 0x0\n\ts_load_dwordx16 %2, %6, 0x0\n\ts_load_dwordx16 %3, %7, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p[0]), "s"(p[1]), "s"(p[2]), "s"(p[3]) : "memory");
#pragma unroll
        for (int e = 0; e < 16; ++e) { f[0][e] = __int_as_float(a[e]); f[1][e] = __int_as_float(b[e]); f[2][e] = __int_as_float(c[e]); f[3][e] = __int_as_float(d[e]); }
    } else {
        tiny_i16v a, b, c, d;
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %5, 0x0\n\ts_load_dwordx16 %3, %5, 0x40\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p[0]), "s"(p[1]) : "memory");
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            f[0][e] = __int_as_float(a[e]); f[0][16 + e] = __int_as_float(b[e]);
            f[1][e] = __int_as_float(c[e]); f[1][16 + e] = __int_as_float(d[e]);
        }
    }
}

// Touch the first dword of four rows (no wait): the lines are in the scalar cache when tiny_sload_rows asks for them an iteration
// later.  The destination registers stay reserved until then -- the loads land asynchronously -- by passing through tiny_keep.
__device__ __forceinline__ void tiny_prefetch4(const float* p0, const float* p1, const float* p2, const float* p3, int (&pf)[4]) {
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0"
                 : "=&s"(pf[0]), "=&s"(pf[1]), "=&s"(pf[2]), "=&s"(pf[3]) : "s"(p0), "s"(p1), "s"(p2), "s"(p3) : "memory");
}
__device__ __forceinline__ void tiny_keep(int (&pf)[4]) {  // placed right after the next wait: until here the registers are taken
    asm volatile("" : "+s"(pf[0]), "+s"(pf[1]), "+s"(pf[2]), "+s"(pf[3]));
}

struct TinyItem {       // a work item of the current query, staged in LDS
    uint32_t start;     // first candidate (position in codes/ids): the index holds fewer than 2^32 items, or the query is flagged
    uint16_t rel, len;  // first candidate inside the query's candidate range, candidates
    uint16_t t0, t1;    // the two half tables, counted from the query's first table of that split
};

template <int MT, int W>
__global__ __launch_bounds__(1024) void k_tiny_select(const WorkItem* __restrict__ items, const int64_t* __restrict__ cand_start,
                                                      const int64_t* __restrict__ seg, const int64_t* __restrict__ item_off,
                                                      const int64_t* __restrict__ tab_off, const PlanOut* __restrict__ plan,
                                                      const double* __restrict__ px, const double* __restrict__ subs,
                                                      const uint8_t* __restrict__ codes, int h, int nq, int L, int ncmax, int pool_bytes,
                                                      int64_t stride, uint64_t* __restrict__ sel_keys, uint64_t* __restrict__ sel_vals,
                                                      int* __restrict__ nsel, uint64_t* __restrict__ keys_fb,
                                                      unsigned long long* __restrict__ qmin, unsigned long long* __restrict__ qmax,
                                                      int* __restrict__ fbflag, const float* __restrict__ px32 /* [ntab][h]: float32 copy of px */,
                                                      unsigned int* __restrict__ dbg) {
    constexpr int NF = MT / 2, K = 256;
    constexpr int PAIRS = NF * K;
    constexpr int PPT = (PAIRS + 1023) / 1024;
    constexpr int CW = NF >= 4 ? NF / 4 : 1;
    extern __shared__ __align__(16) unsigned char tiny_lds[];
    uint16_t* own = reinterpret_cast<uint16_t*>(tiny_lds);
    uint16_t* sum = own + ncmax;
    unsigned char* pool = tiny_lds + (size_t)4 * ncmax;    // items | tables + px of a chunk, later the survivors' exact entries
    TinyItem* s_items = reinterpret_cast<TinyItem*>(pool);
    unsigned int* hist = reinterpret_cast<unsigned int*>(pool + pool_bytes);  // 512 bins
    float* samp = reinterpret_cast<float*>(hist + 512);                    // TINY_NS
    uint64_t* skey = reinterpret_cast<uint64_t*>(samp + TINY_NS);          // TINY_SMAX
    uint32_t* sidx = reinterpret_cast<uint32_t*>(skey + TINY_SMAX);        // TINY_SMAX
    TinyShared* sh = reinterpret_cast<TinyShared*>(sidx + TINY_SMAX);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __builtin_amdgcn_s_dcache_inv();  // the scalar cache may hold rows of an earlier batch at the addresses of px32
    long long t_last = dbg ? wall_clock64() : 0;
#define TINY_T(k_) do { if (dbg && tid == 0) { const long long now_ = wall_clock64(); atomicAdd(&dbg[4 + (k_)], (unsigned int)(now_ - t_last)); t_last = now_; } } while (0)
    // largest centroid component (float32 rounding bound), once per workgroup
    if (tid == 0) sh->cen_bits = 0u;
    __syncthreads();
    {
        float a = 0.f;
        for (int i = tid; i < MT * K * W; i += 1024) a = fmaxf(a, fabsf((float)subs[i]));
        atomicMax(&sh->cen_bits, __float_as_uint(a));
    }
    // the thread's (sub-quantizer, centroid) of BOTH splits stays in registers over the persistent loop when that is 32 floats or fewer
    // (M = 8, w = 16: the release shape; loading them per chunk cost 9 us of a 105 us query)
    constexpr bool KEEP = PPT * W <= 16;
    float cenk[2][KEEP ? PPT : 1][KEEP ? W : 1];
    if constexpr (KEEP) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int pp = 0; pp < PPT; ++pp) {
                const int pr = tid + pp * 1024;
                const double* sc = subs + ((size_t)(s2 * NF + (pr < PAIRS ? pr / K : 0)) * K + pr % K) * W;
#pragma unroll
                for (int e = 0; e < W; ++e) cenk[s2][pp][e] = (float)sc[e];
            }
    }
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        const int64_t it0 = item_off[q];
        const int ni = (int)(item_off[q + 1] - it0);
        const int64_t c0 = seg[q];
        const int64_t n64 = seg[q + 1] - c0;
        const int nv = n64 < (int64_t)L ? (int)n64 : L;
        const int64_t ob = (int64_t)q * stride;
        __syncthreads();  // the previous query's lists are no longer read
        TINY_T(0);
        if (n64 <= 0) {
            if (tid == 0) { nsel[q] = 0; fbflag[q] = 0; }
            continue;
        }
        // this query's share of the pool: its items first, the rest for tables (or the survivors' entries)
        const int ntt = plan[q].ntab0 + plan[q].ntab1;
        const int list_off = (ni * (int)sizeof(TinyItem) + 15) & ~15;           // uint16 used[ntt]: the half tables that have a candidate
        const int items_bytes = list_off + ((ntt * 2 + 15) & ~15);               // (items + lists: what stays for the whole query)
        uint16_t* used = reinterpret_cast<uint16_t*>(pool + list_off);          // split 0's list, then split 1's at used + nt0
        const int rest = pool_bytes - items_bytes;
        int tch = rest > 0 ? rest / (NF * K) : 0;
        tch = tch < TINY_TCH ? tch : TINY_TCH;
        int smax = rest > 0 ? rest / (8 * MT) : 0;
        smax = smax < TINY_SMAX ? smax : TINY_SMAX;
        uint8_t* s_tab = pool + items_bytes;
        double* e_lds = reinterpret_cast<double*>(pool + items_bytes);
        bool fall = n64 > (int64_t)ncmax || ni > 65535 || ntt > 4096 || tch < 4 || L > smax;
        const int n = (int)n64;
        const bool all_in = !fall && n <= smax && n <= (2 * L > 256 ? 2 * L : 256);  // few candidates: every one is ranked exactly
        int nsurv = 0;
        if (!fall) {
            if (tid == 0) { sh->cnt = 0; sh->sstar = 1 << 20; sh->amax_bits = 0u; sh->cap_bits = 0u; sh->fb = 0; }
            for (int b = tid; b < 512; b += 1024) hist[b] = 0u;
            // 1. the items, the half tables that have a candidate at all (155 of 198 at V = 2048: the multisequence visits empty
            //    cells too) numbered densely per split, then the owner item of every candidate
            const int64_t tbase = tab_off[q];
            const int nt0 = plan[q].ntab0;
            uint16_t* slot_of = reinterpret_cast<uint16_t*>(skey);  // [ntt], free until the survivors are ranked
            for (int i = tid; i < ntt; i += 1024) slot_of[i] = 0;
            for (int c = tid; c < n; c += 1024) sum[c] = 0;
            __syncthreads();
            for (int i = tid; i < ni; i += 1024) {
                const WorkItem it = items[it0 + i];
                TinyItem ti;
                ti.start = (uint32_t)it.start;
                ti.rel = (uint16_t)(cand_start[it0 + i] - c0);
                ti.len = (uint16_t)it.len;
                ti.t0 = (uint16_t)(it.tab0 - tbase);
                ti.t1 = (uint16_t)(it.tab1 - tbase - nt0);
                s_items[i] = ti;
                slot_of[ti.t0] = 1;
                slot_of[nt0 + ti.t1] = 1;
                if (it.start + it.len > (int64_t)0xffffffffll || it.len > 65535) sh->fb = 1;
            }
            __syncthreads();
            for (int s = 0; s < 2; ++s) {  // dense numbers: four entries per thread, one workgroup scan per split
                const int base = s ? nt0 : 0, nts = s ? ntt - nt0 : nt0;
                int fl[4], cnt = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = tid * 4 + u; fl[u] = i < nts ? (int)slot_of[base + i] : 0; cnt += fl[u]; }
                int tot;
                int run = sel_block_excl_scan(cnt, sh->wsum, &tot);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = tid * 4 + u;
                    if (i < nts) {
                        slot_of[base + i] = fl[u] ? (uint16_t)run : (uint16_t)0xffff;
                        if (fl[u]) used[base + run] = (uint16_t)i;
                        run += fl[u];
                    }
                }
                if (tid == 0) sh->nused[s] = tot;
                __syncthreads();
            }
            for (int i = tid; i < ni; i += 1024) {
                TinyItem ti = s_items[i];
                ti.t0 = slot_of[ti.t0];
                ti.t1 = slot_of[nt0 + ti.t1];
                s_items[i] = ti;
            }
            __syncthreads();
            for (int i = wv; i < ni; i += 16) {
                const TinyItem ti = s_items[i];
                for (int p = lane; p < (int)ti.len; p += 64) own[(int)ti.rel + p] = (uint16_t)i;
            }
            __syncthreads();
            if (sh->fb) fall = true;
            if (dbg && tid == 0) atomicAdd(&dbg[15], (unsigned int)(sh->nused[0] + sh->nused[1]));
            TINY_T(1);
        }
        if (!fall && !all_in) {
            const int64_t tbase = tab_off[q];
            const int nt0 = plan[q].ntab0;
            // 2. cap from the sample: the r-th smallest of TINY_NS sampled distances (float32 arithmetic: the cap is a heuristic,
            //    the bounds below hold for any value), r / TINY_NS ~ three times limit / n
            {
                const int s = tid >> 3, sub = tid & 7;
                const int c = (int)(((int64_t)s * n + (n >> 1)) / TINY_NS);
                const TinyItem ti = s_items[own[c]];
                const CodeWords<MT> cw = load_code<MT>(codes, (int64_t)ti.start + (c - (int)ti.rel));
                float acc = 0.f;
                for (int j = sub; j < MT; j += 8) {
                    const bool second = j >= NF;
                    const double* f = reinterpret_cast<const double*>(__builtin_assume_aligned(
                        px + (tbase + (second ? nt0 + (int)used[nt0 + ti.t1] : (int)used[ti.t0])) * h + (second ? j - NF : j) * W, 16));
                    const uint32_t code = (cw.w[j >> 2] >> (8 * (j & 3))) & 255u;
                    const double* sc = reinterpret_cast<const double*>(__builtin_assume_aligned(subs + ((size_t)j * K + code) * W, 16));
#pragma unroll
                    for (int e = 0; e < W; ++e) { const float df = (float)f[e] - (float)sc[e]; acc = fmaf(df, df, acc); }
                }
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                acc += __shfl_xor(acc, 4);
                if (sub == 0) samp[s] = acc;
            }
            __syncthreads();
            {
                const int s = tid >> 3, sub = tid & 7;
                const float v = samp[s];
                int less = 0;
                for (int k2 = sub * (TINY_NS / 8); k2 < (sub + 1) * (TINY_NS / 8); ++k2) {
                    const float u = samp[k2];
                    less += (u < v || (u == v && k2 < s)) ? 1 : 0;
                }
                less += __shfl_xor(less, 1);
                less += __shfl_xor(less, 2);
                less += __shfl_xor(less, 4);
                int r = (int)((3.0 * TINY_NS * (double)L) / (double)n) + 3;
                r = r > TINY_NS ? TINY_NS : r;
                if (sub == 0 && less == r - 1) sh->cap_bits = __float_as_uint(v);
            }
            __syncthreads();
            TINY_T(2);
            const float cap = __uint_as_float(sh->cap_bits);
            const float inv_step = cap > 0.f ? 256.0f / cap : 0.f;
            // 3. byte tables of a split, at most tch at a time; every candidate adds its entries up.  An entry is
            //    sum (px - c)^2 in float32: the thread keeps its (sub-quantizer, centroid) in registers, the px row of a table
            //    arrives through scalar loads (uniform over the wave) from the float32 copy of px that k_tables_group wrote
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int nts = sh->nused[s];            // tables of this split that have a candidate
                const int64_t tb = tbase + (s ? nt0 : 0);
                const uint16_t* lst = used + (s ? nt0 : 0);
                for (int tlo = 0; tlo < nts; tlo += tch) {
                    const int tc = nts - tlo < tch ? nts - tlo : tch;
                    __syncthreads();  // the previous chunk's tables are no longer read
                    TINY_T(3);
                    TINY_T(4);
#pragma unroll
                    for (int pp = 0; pp < PPT; ++pp) {
                        const int pr = tid + pp * 1024;
                        if (pr < PAIRS) {
                            const int jj = __builtin_amdgcn_readfirstlane(pr / K), k = pr % K;  // a wave shares its sub-quantizer
                            float cen[W];
                            if constexpr (KEEP) {
#pragma unroll
                                for (int e = 0; e < W; ++e) cen[e] = cenk[s][pp][e];
                            } else {
                                const double* sc = subs + ((size_t)(s * NF + jj) * K + k) * W;
#pragma unroll
                                for (int e = 0; e < W; ++e) cen[e] = (float)sc[e];
                            }
                            if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TINY_T(12); }
                            const float* frow = px32 + tb * h + jj * W;
                            uint8_t* trow = s_tab + jj * K + k;
                            constexpr int U = TinyRows<W>::U;
                            typedef float tiny_f2 __attribute__((ext_vector_type(2)));
                            int cur[U];  // table numbers of the next U rows: read from LDS one iteration ahead (under the scalar loads' wait)
#pragma unroll
                            for (int u = 0; u < U; ++u) cur[u] = __builtin_amdgcn_readfirstlane((int)lst[tlo + (u < tc ? u : tc - 1)]);
                            int pf[4] = {0, 0, 0, 0};
                            for (int t = 0; t < tc; t += U) {
                                const float* rp[U];
                                int tu[U], nxt[U];
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    tu[u] = t + u < tc ? t + u : tc - 1;
                                    rp[u] = frow + cur[u] * h;
                                    nxt[u] = (int)lst[tlo + (t + U + u < tc ? t + U + u : tc - 1)];
                                }
                                float f[U][W];
                                tiny_sload_rows<W>(rp, f);
                                if constexpr (U == 4) tiny_keep(pf);
#pragma unroll
                                for (int u = 0; u < U; ++u) cur[u] = __builtin_amdgcn_readfirstlane(nxt[u]);
                                if constexpr (U == 4) tiny_prefetch4(frow + cur[0] * h, frow + cur[1] * h, frow + cur[2] * h, frow + cur[3] * h, pf);
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    tiny_f2 acc = {0.f, 0.f};  // packed float32 math: even and odd components apart
#pragma unroll
                                    for (int e = 0; e < W; e += 2) {
                                        const tiny_f2 c2 = {cen[e], cen[e + 1]};
                                        const tiny_f2 x = {f[u][e], f[u][e + 1]};
                                        const tiny_f2 d = x - c2;
                                        acc = __builtin_elementwise_fma(d, d, acc);
                                    }
                                    trow[tu[u] * NF * K] = (uint8_t)(int)fminf((acc.x + acc.y) * inv_step, 255.0f);
                                }
                            }
                            if constexpr (U == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tiny_keep(pf); }
                        }
                    }
                    __syncthreads();
                    TINY_T(5);
                    for (int cb = tid; cb < n; cb += 4096) {  // four candidates per thread and round: their code loads overlap
                        int tl4[4];
                        uint32_t cw4[4][CW];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = cb + r * 1024;
                            tl4[r] = -1;
                            if (c < n) {
                                const TinyItem ti = s_items[own[c]];
                                const int tl = (s ? (int)ti.t1 : (int)ti.t0) - tlo;
                                if ((unsigned)tl < (unsigned)tc) {
                                    tl4[r] = tl;
                                    const uint8_t* cp = codes + ((int64_t)ti.start + (c - (int)ti.rel)) * MT;
                                    if constexpr (NF == 2) cw4[r][0] = *reinterpret_cast<const uint32_t*>(cp) >> (16 * s);
                                    else {
#pragma unroll
                                        for (int u = 0; u < CW; ++u) cw4[r][u] = reinterpret_cast<const uint32_t*>(cp + s * NF)[u];
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (tl4[r] >= 0) {
                                const int c = cb + r * 1024;
                                unsigned int acc = 0;
#pragma unroll
                                for (int jj = 0; jj < NF; ++jj)
                                    acc += s_tab[(tl4[r] * NF + jj) * K + ((cw4[r][jj >> 2] >> (8 * (jj & 3))) & 255u)];
                                sum[c] = (uint16_t)(sum[c] + acc);
                            }
                    }
                }
            }
            __syncthreads();
            TINY_T(6);
            // 4. histogram of the sums, s*, the survivors' bound
            for (int c = tid; c < n; c += 1024) {
                const int sv = sum[c];
                atomicAdd(&hist[sv < 511 ? sv : 511], 1u);
            }
            __syncthreads();
            {
                const int v = tid < 512 ? (int)hist[tid] : 0;
                int tot;
                const int ex = sel_block_excl_scan(v, sh->wsum, &tot);
                if (tid < 512) {
                    hist[tid] = (unsigned int)(ex + v);  // inclusive counts (each thread rewrites its own bin)
                    if (ex < L && L <= ex + v) sh->sstar = tid;
                }
            }
            __syncthreads();
            const int sstar = sh->sstar;
            // float32: a component difference carries an absolute error of 2^-23 amax_, amax_ = the largest operand -- |c| <= cmax_, and
            // where the entry is below the cap |px| < cmax_ + sqrt(cap) (a clamped entry only needs the RELATIVE accuracy it has);
            // an entry e < cap is off by at most 2 sqrt(W cap) 2^-23 amax_: under step / 16 = cap / 4096 when cap >= W amax_^2 / 2^20
            const float cmax_ = __uint_as_float(sh->cen_bits);
            const float amax_ = cmax_ + sqrtf(cap);
            if (sstar > 254 || !(cap >= (float)W * amax_ * amax_ * (1.0f / 1048576.0f))) fall = true;
            else {
                const int thr = sstar + MT + 1;
                nsurv = (int)hist[thr < 511 ? thr : 511];
                if (thr >= 511 || nsurv > smax) fall = true;
                else {
                    for (int c = tid; c < n; c += 1024)
                        if ((int)sum[c] <= thr) sidx[atomicAdd(&sh->cnt, 1)] = (uint32_t)c;
                }
            }
            if (dbg && tid == 0) {
                atomicAdd(&dbg[0], 1u);
                atomicAdd(&dbg[1], fall ? 1u : 0u);
                atomicAdd(&dbg[2], (unsigned int)nsurv);
                atomicAdd(&dbg[3], (unsigned int)(sstar > 254 ? 255 : sstar));
            }
            __syncthreads();
        } else if (all_in) {
            nsurv = n;
            for (int c = tid; c < n; c += 1024) sidx[c] = (uint32_t)c;
            __syncthreads();
        }
        if (fall) {
            // every candidate's exact key to the global key array; k_select_topl ranks this query
            tiny_exact_all<MT, W>(items, cand_start, px, subs, codes, h, it0, ni, c0, n64, keys_fb, qmin, qmax, q);
            if (tid == 0) fbflag[q] = 1;
            continue;
        }
        TINY_T(7);
        // 5. exact entries of the survivors, one thread per (survivor, sub-quantizer); then the running sum of search.py:173
        {
            const int64_t tbase = tab_off[q];
            const int nt0 = plan[q].ntab0;
            for (int pr = tid; pr < nsurv * MT; pr += 1024) {
                const int sv = pr / MT, j = pr % MT;
                const int c = (int)sidx[sv];
                const TinyItem ti = s_items[own[c]];
                const uint32_t code = codes[((int64_t)ti.start + (c - (int)ti.rel)) * MT + j];
                const bool second = j >= NF;
                const double* f = reinterpret_cast<const double*>(__builtin_assume_aligned(
                    px + (tbase + (second ? nt0 + (int)used[nt0 + ti.t1] : (int)used[ti.t0])) * h + (second ? j - NF : j) * W, 16));
                const double* sc = reinterpret_cast<const double*>(__builtin_assume_aligned(subs + ((size_t)j * K + code) * W, 16));
                auto elem = [&](int e) -> double { const double df = f[e] - sc[e]; return df * df; };
                e_lds[pr] = pw_leaf<double>(elem, 0, W);
            }
        }
        __syncthreads();
        TINY_T(8);
        for (int sv = tid; sv < nsurv; sv += 1024) {
            double d = e_lds[sv * MT];
#pragma unroll
            for (int j = 1; j < MT; ++j) d = d + e_lds[sv * MT + j];
            skey[sv] = (uint64_t)__double_as_longlong(d);
        }
        __syncthreads();
        TINY_T(9);
        // rank by counting: a survivor's place is the number of survivors before it in (key, retrieval position) order -- all lanes
        // read the same pair at a time (LDS broadcast), no barrier; the first `limit` places are the result
        for (int sv0 = 0; sv0 < nsurv; sv0 += 128) {  // eight threads per survivor, an eighth of the list each
            const int sv = sv0 + (tid >> 3), part = tid & 7;
            const bool on = sv < nsurv;
            const uint64_t kk = on ? skey[sv] : 0ull;
            const uint32_t cc = on ? sidx[sv] : 0u;
            const int per = (nsurv + 7) >> 3;
            const int j0 = part * per, j1 = j0 + per < nsurv ? j0 + per : nsurv;
            int place = 0;
#pragma unroll 4
            for (int j = j0; j < j1; ++j) place += sel_pair_less(skey[j], sidx[j], kk, cc) ? 1 : 0;
            place += __shfl_xor(place, 1);
            place += __shfl_xor(place, 2);
            place += __shfl_xor(place, 4);
            if (on && part == 0 && place < nv) {
                const int i = own[cc];
                sel_keys[ob + place] = kk;
                sel_vals[ob + place] = ((uint64_t)(it0 + i) << 32) | (uint32_t)((int)cc - (int)s_items[i].rel);
            }
        }
        if (tid == 0) { nsel[q] = nv; fbflag[q] = 0; }
        TINY_T(10);
    }
#undef TINY_T
}

static size_t tiny_lds_bytes(int ncmax, int pool_bytes) {
    return (size_t)4 * ncmax + (size_t)pool_bytes + 512 * 4 + TINY_NS * 4 + (size_t)TINY_SMAX * 12 + sizeof(TinyShared) + 16;
}

// bytes of the per-query pool (items, tables or survivors) next to the candidate layout; 0: the kernel does not serve this shape
static int tiny_pool(int M, int K, int w, int h, int L, int64_t ncmax_need, int* ncmax_out) {
    if (!(M == 4 || M == 8 || M == 16) || K != 256 || !(w == 4 || w == 8 || w == 16 || w == 32) || h != (M / 2) * w || h > 256) return 0;
    if (ncmax_need > TINY_NCMAX) return 0;
    int ncmax = 4096;
    while (ncmax < ncmax_need) ncmax += 2048;
    const size_t fixed = tiny_lds_bytes(ncmax, 0);
    const size_t budget = 160 * 1024 - 512;  // the static LDS of the helpers (key range, scans) comes on top
    if (fixed + 32768 > budget) return 0;
    const int pool = (int)((budget - fixed) & ~(size_t)15);
    // a query with a few hundred items must keep room for 2 * limit survivors' exact entries and a useful number of tables
    if ((pool - 8192) / (8 * M) < 2 * L || (pool - 8192) / ((M / 2) * 256) < 8) return 0;
    *ncmax_out = ncmax;
    return pool;
}

template <int MT, int W>
static void launch_tiny_t(hipStream_t st, const WorkItem* items, const int64_t* cand_start, const int64_t* seg, const int64_t* item_off,
                          const int64_t* tab_off, const PlanOut* plan, const double* px, const double* subs, const uint8_t* codes, int h, int nq,
                          int L, int ncmax, int tch, int64_t stride, uint64_t* sel_keys, uint64_t* sel_vals, int* nsel, uint64_t* keys_fb,
                          unsigned long long* qmin, unsigned long long* qmax, int* fbflag, float* px32_ws, unsigned int* dbg) {
    const size_t lds = tiny_lds_bytes(ncmax, tch);
    const unsigned grid = (unsigned)(nq < 256 ? nq : 256);
    hipLaunchKernelGGL((k_tiny_select<MT, W>), dim3(grid), dim3(1024), lds, st, items, cand_start, seg, item_off, tab_off, plan, px, subs, codes,
                       h, nq, L, ncmax, tch, stride, sel_keys, sel_vals, nsel, keys_fb, qmin, qmax, fbflag, px32_ws, dbg);
}

static bool launch_tiny(int M, int w, hipStream_t st, const WorkItem* items, const int64_t* cand_start, const int64_t* seg,
                        const int64_t* item_off, const int64_t* tab_off, const PlanOut* plan, const double* px, const double* subs,
                        const uint8_t* codes, int h, int nq, int L, int ncmax, int tch, int64_t stride, uint64_t* sel_keys, uint64_t* sel_vals,
                        int* nsel, uint64_t* keys_fb, unsigned long long* qmin, unsigned long long* qmax, int* fbflag, float* px32_ws, unsigned int* dbg) {
#define CIS_TINY(MT, W)                                                                                                            \
    if (M == MT && w == W) {                                                                                                       \
        launch_tiny_t<MT, W>(st, items, cand_start, seg, item_off, tab_off, plan, px, subs, codes, h, nq, L, ncmax, tch, stride,   \
                             sel_keys, sel_vals, nsel, keys_fb, qmin, qmax, fbflag, px32_ws, dbg);                                 \
        return true;                                                                                                               \
    }
    CIS_TINY(8, 16) CIS_TINY(16, 8) CIS_TINY(4, 32) CIS_TINY(16, 16) CIS_TINY(8, 32) CIS_TINY(8, 8) CIS_TINY(4, 16) CIS_TINY(16, 4)
#undef CIS_TINY
    return false;
}

// one sub-batch of queries (device pointers); writes ranked partial hits [nq][L] and visited [nq]
static const int CIS_RETRY_SMALLER = 1;  // internal: the batch does not fit the workspace budget, halve it

struct SearchOut {  // any of these may be null; all are [nq][L] except n_found / visited [nq]
    cis_hit* hits;
    int64_t* ids;
    double* dists;
    int32_t* n_found;
    int32_t* cells;
    uint32_t* pos;
    int32_t* visited;
    SearchOut at(int64_t q0, int L) const {
        SearchOut o = *this;
        if (o.hits) o.hits += q0 * L;
        if (o.ids) o.ids += q0 * L;
        if (o.dists) o.dists += q0 * L;
        if (o.cells) o.cells += q0 * L;
        if (o.pos) o.pos += q0 * L;
        if (o.n_found) o.n_found += q0;
        if (o.visited) o.visited += q0;
        return o;
    }
};

static int search_batch(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, int L, const SearchOut& out,
                        hipStream_t st) {
    cis_model* m = ix->m;
    const int V = m->V, D = m->D, K = m->K, M = m->M, h = m->h, nf = m->nf;
    const auto t_entry = std::chrono::steady_clock::now();
    ix->last_scan_kernel = 0;
    cis_index::ProfRec pr;
    pr.has_scan = false;
    for (int i = 0; i < 6; ++i) pr.ev[i] = nullptr;
    auto mark = [&](int i) -> int {
        if (!ix->profiling) return CIS_OK;
        if (ix->profiling == 1 && i != 5 && i != 3) return CIS_OK;  // level 1: only the pair around the scan kernel
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
        CIS_TRY(cis_dev_apply_pca(m, dQ, q_dtype, nq, ix->w_xp.as<float>(), st, &ix->w_y64));
        xp = ix->w_xp.p;
        xp_dtype = CIS_F32;
    }
    const void* xc;
    int ct;
    CIS_TRY(cis_dev_coarse_type(m, xp, xp_dtype, nq, &xc, &ct, st, &ix->w_x64));
    const size_t csz = (ct == CIS_F32) ? 4 : 8;
    // 2. coarse distances, rank
    CIS_TRY(ix->w_cd.reserve((size_t)2 * nq * V * csz));
    CIS_TRY(ix->w_sorted.reserve((size_t)2 * nq * V * csz));
    CIS_TRY(ix->w_order.reserve((size_t)2 * nq * V * sizeof(uint16_t)));
    CIS_TRY(ix->w_plan.reserve((size_t)nq * sizeof(PlanOut)));
    CIS_TRY(ix->w_off.reserve((size_t)(2 * (nq + 1) + 4 + nq) * sizeof(int64_t)));
    int64_t* item_off = ix->w_off.as<int64_t>();
    int64_t* tab_off = item_off + (nq + 1);
    int64_t* totals = tab_off + (nq + 1);
    unsigned long long* qbound = reinterpret_cast<unsigned long long*>(totals + 4);  // per query: cross-cell bound of the scan
    PlanOut* plan = ix->w_plan.as<PlanOut>();
    {
        const void* grp_before = ix->w_grp.p;
        CIS_TRY(ix->w_grp.reserve((size_t)(GRP_WORDS(V) + 2 * GRP_TILES(V)) * sizeof(int)));
        if (ix->w_grp.p != grp_before)  // fresh memory: the counters start clean (afterwards every k_plan_scan leaves them clean)
            CIS_CHECK_HIP(hipMemsetAsync(ix->w_grp.p, 0, ix->w_grp.cap, st));
    }
    int* grp_cnt = ix->w_grp.as<int>();         // [2V][GRP_SUB] tables per (split, cluster, query % GRP_SUB)
    int* grp_cur = grp_cnt + 2 * V * GRP_SUB;    // cursors
    int* grp_base = grp_cnt + 4 * V * GRP_SUB;   // exclusive scan
    // scan v3 (16-bit fixed-point tables, four queries per workgroup) for large batches; its region entries hold 16-bit
    // positions, so a chunk is at most 65536 candidates.  scan_mode 3 forces it for any batch size (tests).
    static const int env_scan = getenv("CIS_FORCE_SCAN") ? atoi(getenv("CIS_FORCE_SCAN")) : 0;  // A/B runs: 2 or 3
    // Automatic routing picks it where it measured faster than k_adc_scan2 (profiles/r02*): short cells (a batch that
    // visits about quota / (n_total / cells) + 1 cells per query, each shorter than 8192 codes) at M <= 8.
    const bool short_cells = ix->nonempty_cells > 0 && ix->n_total / ix->nonempty_cells < 8192 && M <= 8;
    // ... and long cells at M <= 8, limit <= 128 through its sampled single-pass form (k_adc_scan4 with a tenth of the rows as
    // the sample: profiles/r03m_*); CIS_SCAN_LONG=0 keeps them on k_adc_scan2
    static const int env_long = getenv("CIS_SCAN_LONG") ? atoi(getenv("CIS_SCAN_LONG")) : 1;
    // ... and at M = 16 with the saturating scale (round 4).  Its failure mode is expensive -- a slot whose scale missed runs again in
    // the streaming form, ~14x the time when most of them do -- so the index watches the fall-back count of its last such scan
    // (a pinned word the scan's stream writes; no wait, the value may be one batch old) and, when more than an eighth of the slots
    // fell back, serves the next 64 batches from k_adc_scan2 before it tries again.  CIS_S4_M16=0: always k_adc_scan2.
    static const int env_m16 = getenv("CIS_S4_M16") ? atoi(getenv("CIS_S4_M16")) : 1;
    if (M == 16 && env_m16 && ix->h_totals) {
        const int32_t* fb = reinterpret_cast<const int32_t*>(&ix->h_totals[4]);
        const int32_t n_s = __atomic_load_n(&fb[0], __ATOMIC_RELAXED), n_f = __atomic_load_n(&fb[1], __ATOMIC_RELAXED);
        if (ix->m16_holdoff > 0) --ix->m16_holdoff;
        else if (n_s > 0 && (int64_t)n_f * 8 > n_s) { ix->m16_holdoff = 64; ++ix->m16_backoffs; ix->h_totals[4] = 0; }
    }
    const bool long_cells = env_long != 0 && !short_cells && (M <= 8 || (env_m16 && ix->m16_holdoff == 0)) && L <= 128 && ix->ncells <= 65536;
    const bool use3 = !ix->force_exact_scan && scan3_supported(M, K, L) && !use_all_path(ix, M, K, L, nq) &&
                      (ix->force_scan3 || (nq >= 256 && !ix->force_scan2 && (env_scan == 3 || (env_scan == 0 && (short_cells || long_cells)))));
    // (Splitting a shard's cells into chunks so that a cell-sharded index at world = 8 fills the chip again was measured
    // and lost: 0.516 against 0.306 ms per partial search -- more survivors, colder bounds: profiles/r02d_shard_emulation.txt.)
    int seg_max = use3 ? (nq >= 64 ? 65536 : 4096) : (nq >= 1024 ? (1 << 20) : (nq >= 64 ? 16384 : 4096));
    // (Round 3 cut the long cells of a shard of four or more into chunks of 20480 candidates so that its few work items spread over
    // the chip: partial search 0.343 -> 0.307 ms alone on the device.  With three batches in flight the whole-cell chunks win -- emulated
    // rank 0 of world 8: 0.227 -> 0.163 ms per step, and 0.552 -> 0.479 ms for the routed protocol's full batches
    // (profiles/archive/r05b/r05_shards.txt) -- so the rule is gone; CIS_SEG_MAX=20480 brings it back for A/B runs.)
    // The HBM-streaming route (lopq_stream.hip): few queries, very many candidates each -- an exhaustive quota, or any quota on cells
    // of hundreds of thousands of codes.  Decided here from the bound of the candidates per query (the chunk size is part of the
    // plan); the exact count confirms it below.  Chunks of 65536 candidates, longer when the largest cell would need more than the
    // slot builder's 16 chunk keys (the items of a slot must be the SAME chunk of a cell).
    static const int64_t stream_min = getenv("CIS_STREAM_MIN") ? atoll(getenv("CIS_STREAM_MIN")) : 262144;  // candidates per query from which it pays
    static const int stream_nq = getenv("CIS_STREAM_NQ") ? atoi(getenv("CIS_STREAM_NQ")) : 16;               // 0: never
    const bool stream_auto = !ix->stream_off && !ix->force_exact_scan && !ix->force_prefilter_scan && !ix->force_scan2 && !ix->force_scan3 &&
                             nq <= stream_nq && L <= 440 && ix->world == 1 &&
                             ((quota < ix->n_total ? (quota < 0 ? 0 : quota) : ix->n_total) + ix->max_cell) >= stream_min;
    const bool stream_hint = (ix->force_stream || stream_auto) && !ix->stream_off && stream_supported(M, K, L) && (K % 4 == 0) &&
                             ix->ncells <= 65536 && !index_has_tiny_cells(ix);
    if (stream_hint) {
        // One chunk per cell (round 6): the stream kernel cuts the ROWS of all slots into equal ranges itself, so short chunks buy
        // nothing and cost plan, slot builder and candidate layout their work items (2398 -> 256 for the exhaustive query over 200 M
        // codes).  A chunk stays below 2^31 code bytes (32-bit buffer offsets).
        const int64_t cap_len = (((int64_t)1 << 31) - 65536) / M;
        int64_t want = ix->max_cell > 65536 ? ix->max_cell : 65536;
        want = want < cap_len ? want : cap_len;
        if (const char* e = getenv("CIS_STREAM_CHUNK")) want = atoi(e) > 0 ? atoi(e) : want;  // A/B runs
        seg_max = (int)(ceil_div(want, (int64_t)1024) * 1024);
        const int64_t need = ceil_div(ix->max_cell > 0 ? ix->max_cell : 1, (int64_t)16);   // (at most 16 chunk keys per cell)
        if (need > seg_max) seg_max = (int)ceil_div(need, (int64_t)1024) * 1024;
        if (ix->force_stream && getenv("CIS_STREAM_SEG")) seg_max = atoi(getenv("CIS_STREAM_SEG")) > 0 ? atoi(getenv("CIS_STREAM_SEG")) : seg_max;  // tests: short chunks on small fixtures
    }
    if (const char* e = getenv("CIS_SEG_MAX")) seg_max = atoi(e) > 0 ? atoi(e) : seg_max;  // A/B runs (tools/emulate_shard.py)
    // thousands of coarse clusters (the release configurations' V = 2048 / 4096): the plan is a per-query selection + sort
    // (k_plan_par) instead of one frontier step per visited cell; queries it cannot resolve fall back to the frontier walk
    const bool no_par_plan = getenv("CIS_NO_PAR_PLAN") != nullptr;
    // ... and few queries that visit MANY cells of a small vocabulary (the streaming route's exhaustive quota: all 256 cells of V = 16):
    // the frontier walk costs ~1.3 us per visited cell and runs twice (count, emit) -- 0.67 of the 1.25 ms of a single exhaustive
    // query over 200 M codes; the sort-based plan does the same cells in one band
    const bool stream_par = stream_hint && V >= 8 && ix->n_total > 0 &&
                            (double)(quota < 0 ? 0 : quota) * (double)ix->nonempty_cells >= 32.0 * (double)ix->n_total;
    const bool par_plan = ((V >= 128 || stream_par) && V <= PLAN_PAR_STAGE) && !no_par_plan;
    unsigned long long* plan_hint = (par_plan && !getenv("CIS_NO_PLAN_HINT")) ? ix->plan_hint_ptr() : nullptr;
    const int hint_slot = (int)((ix->plan_seq + 1) & 1);  // this batch's launches add into this parity and read the other
    int* plan_fb = nullptr;
    uint64_t* vis_list = nullptr;
    // cells WITH candidates per query the fast plan lists (16 bytes each; the empty cells it walks -- most of them at thousands of coarse
    // clusters, tens of thousands for an outlier query -- are not listed): 2 GB of lists per batch at most; past the cap the query goes
    // to the serial frontier walk, ~3 us per cell
    int64_t vis_cap64 = ((int64_t)1 << 31) / ((int64_t)(nq > 0 ? nq : 1) * 16);
    if (vis_cap64 < 4096) vis_cap64 = 4096;
    if (vis_cap64 > (1 << 20)) vis_cap64 = 1 << 20;
    if (vis_cap64 > (int64_t)V * V) vis_cap64 = (int64_t)V * V;
    const int vis_cap = (int)vis_cap64;
    if (par_plan) {
        CIS_TRY(ix->w_planfb.reserve((size_t)2 * nq * sizeof(int)));
        CIS_TRY(ix->w_vis.reserve((size_t)nq * vis_cap * 2 * sizeof(uint64_t)));
        plan_fb = ix->w_planfb.as<int>();
        vis_list = ix->w_vis.as<uint64_t>();
    }
    const size_t plan_lds = (size_t)V * sizeof(int);
    int Vp2 = 64;
    while (Vp2 < V) Vp2 <<= 1;
    static const bool no_fused_front = getenv("CIS_NO_FUSED_FRONT") != nullptr;
    const bool fused_front = V <= 64 && !par_plan && !no_fused_front;
    if (fused_front) {
        // coarse distances + rank + counting pass of the multisequence walk in one launch (k_front_small)
        const size_t flds = (size_t)V * (32 + 4 + 4);
        if (ct == CIS_F32)
            hipLaunchKernelGGL(k_front_small<float>, dim3(nq), dim3(64), flds, st, (const float*)xc, D, h, m->d_Cs32, m->prog_h, ix->gcount_ptr(),
                               ix->loff_ptr(), nq, V, quota, seg_max, ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>(), plan, grp_cnt);
        else
            hipLaunchKernelGGL(k_front_small<double>, dim3(nq), dim3(64), flds, st, (const double*)xc, D, h, m->d_Cs64, m->prog_h, ix->gcount_ptr(),
                               ix->loff_ptr(), nq, V, quota, seg_max, ix->w_order.as<uint16_t>(), ix->w_sorted.as<double>(), plan, grp_cnt);
    } else {
    CIS_TRY(cis_launch_sqdist_both(m, xc, ct, nq, ix->w_cd.p, st));
    if (ct == CIS_F32) {
        const bool sort_lds = getenv("CIS_RANK_SORT_LDS") != nullptr;  // the LDS form of the sort (A/B runs)
        if (V > 256 && Vp2 == 1024 && !sort_lds)
            hipLaunchKernelGGL(k_rank_sort_reg<4>, dim3(nq, 2), dim3(256), 0, st, ix->w_cd.as<float>(), nq, V, ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>(), grp_cnt);
        else if (V > 256 && Vp2 == 2048 && !sort_lds)
            hipLaunchKernelGGL(k_rank_sort_reg<8>, dim3(nq, 2), dim3(256), 0, st, ix->w_cd.as<float>(), nq, V, ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>(), grp_cnt);
        else if (V > 256 && Vp2 == 4096 && !sort_lds)
            hipLaunchKernelGGL(k_rank_sort_reg<16>, dim3(nq, 2), dim3(256), 0, st, ix->w_cd.as<float>(), nq, V, ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>(), grp_cnt);
        else if (V > 256 && Vp2 <= 4096)
            hipLaunchKernelGGL(k_rank_sort<float>, dim3(nq, 2), dim3(256), (size_t)Vp2 * 16, st, ix->w_cd.as<float>(), nq, V, Vp2,
                               ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>(), grp_cnt);
        else
            hipLaunchKernelGGL(k_rank<float>, dim3(nq, 2), dim3(V <= 64 ? 64 : 256), (size_t)V * 8, st, ix->w_cd.as<float>(), nq, V,
                               ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>(), grp_cnt);
        if (par_plan)
            hipLaunchKernelGGL((k_plan_par<float, false>), dim3(nq), dim3(256), (size_t)2 * (V < PLAN_SP ? V : PLAN_SP) * sizeof(float), st, ix->w_sorted.as<float>(), ix->w_order.as<uint16_t>(),
                               ix->gcount_ptr(), ix->loff_ptr(), nq, V, quota, seg_max, plan, nullptr, nullptr, nullptr,
                               nullptr, grp_cnt, nullptr, nullptr, nullptr, vis_list, plan_fb, vis_cap, plan_hint, hint_slot);
#ifdef CIS_PLAN_DBG
        if (par_plan) {
            unsigned long long h[12];
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_plan_dbg), sizeof(h));
            fprintf(stderr, "[cis] k_plan_par: %d queries, probes %.1f / query, bands %.2f / query, bisection %.1f us / query, after rows %.1f, after cells %.1f, after sort %.1f, after cut %.1f us / query (cumulative), cells per band %.0f (100 MHz clock)\n",
                    nq, (double)h[0] / nq, (double)h[1] / nq, (double)h[2] / nq / 100.0, (double)h[5] / nq / 100.0, (double)h[6] / nq / 100.0, (double)h[7] / nq / 100.0, (double)h[3] / nq / 100.0, h[1] ? (double)h[4] / (double)h[1] : 0.0);
            fprintf(stderr, "[cis] k_plan_par since kernel start: staged %.1f, bands done %.1f, visited pass %.1f, end %.1f us / query\n", (double)h[8] / nq / 100.0, (double)h[9] / nq / 100.0, (double)h[10] / nq / 100.0, (double)h[11] / nq / 100.0);
            memset(h, 0, sizeof(h));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_plan_dbg), h, sizeof(h));
        }
#endif
        hipLaunchKernelGGL((k_plan<float, false>), dim3(nq), dim3(64), plan_lds, st, ix->w_sorted.as<float>(),
                           ix->w_order.as<uint16_t>(), ix->gcount_ptr(), ix->loff_ptr(), nq, V, quota,
                           seg_max, plan, nullptr, nullptr, nullptr, nullptr, grp_cnt, nullptr, nullptr, nullptr, plan_fb);
    } else {
        if (V > 256 && Vp2 <= 4096)
            hipLaunchKernelGGL(k_rank_sort<double>, dim3(nq, 2), dim3(256), (size_t)Vp2 * 16, st, ix->w_cd.as<double>(), nq, V, Vp2,
                               ix->w_order.as<uint16_t>(), ix->w_sorted.as<double>(), grp_cnt);
        else
            hipLaunchKernelGGL(k_rank<double>, dim3(nq, 2), dim3(V <= 64 ? 64 : 256), (size_t)V * 8, st, ix->w_cd.as<double>(), nq,
                               V, ix->w_order.as<uint16_t>(), ix->w_sorted.as<double>(), grp_cnt);
        if (par_plan)
            hipLaunchKernelGGL((k_plan_par<double, false>), dim3(nq), dim3(256), (size_t)2 * (V < PLAN_SP ? V : PLAN_SP) * sizeof(double), st, ix->w_sorted.as<double>(), ix->w_order.as<uint16_t>(),
                               ix->gcount_ptr(), ix->loff_ptr(), nq, V, quota, seg_max, plan, nullptr, nullptr, nullptr,
                               nullptr, grp_cnt, nullptr, nullptr, nullptr, vis_list, plan_fb, vis_cap, plan_hint, hint_slot);
        hipLaunchKernelGGL((k_plan<double, false>), dim3(nq), dim3(64), plan_lds, st, ix->w_sorted.as<double>(),
                           ix->w_order.as<uint16_t>(), ix->gcount_ptr(), ix->loff_ptr(), nq, V, quota,
                           seg_max, plan, nullptr, nullptr, nullptr, nullptr, grp_cnt, nullptr, nullptr, nullptr, plan_fb);
    }
    }
    if (par_plan && getenv("CIS_DEBUG_PLAN")) {
        std::vector<int> fbh(nq);
        CIS_CHECK_HIP(hipMemcpyAsync(fbh.data(), plan_fb, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, st));
        CIS_CHECK_HIP(hipStreamSynchronize(st));
        int nfb = 0;
        for (int i = 0; i < nq; ++i) nfb += fbh[i] != 0;
        fprintf(stderr, "[cis] k_plan_par: %d of %d queries fall back to the frontier walk (quota %lld)\n", nfb, nq, (long long)quota);
    }
    if (!ix->h_totals) {
        CIS_CHECK_HIP(hipHostMalloc((void**)&ix->h_totals, 12 * sizeof(int64_t), hipHostMallocMapped | hipHostMallocCoherent));
        CIS_CHECK_HIP(hipHostGetDevicePointer((void**)&ix->d_h_totals, ix->h_totals, 0));
        ix->h_totals[3] = 0;
        ix->h_totals[4] = 0;  // (slots, fall-back slots) of the last sampled scan at M = 16: see m16_holdoff
        for (int i = 6; i < 12; ++i) ix->h_totals[i] = 0;  // [6] failed proofs, [7] overflowed lists, [8] sequence word of the streaming route
    }
    const int64_t seq = ++ix->plan_seq;
    const int n_groups = 2 * V * GRP_SUB;
    const bool groups_apart = n_groups > 8192;  // wide vocabularies: the group bases by their own launch (all 16 waves)
    if (groups_apart)
        hipLaunchKernelGGL(k_group_bases, dim3((n_groups + GROUP_TILE - 1) / GROUP_TILE), dim3(256), 0, st, grp_cnt, grp_base, n_groups,
                           reinterpret_cast<unsigned long long*>(grp_cnt + GRP_WORDS(V)), (uint32_t)(seq & 0x7fffffff) | 0x80000000u);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, plan, nq, item_off, tab_off, totals, qbound, ix->d_h_totals, seq, grp_cnt, grp_base, groups_apart ? 0 : n_groups,
                       plan_hint ? plan_hint + (hint_slot ^ 1) * 2 : nullptr);
    volatile int64_t* h_tot = ix->h_totals;
    // A small batch on the all-candidates path does not wait for the plan totals: the workspace is sized by upper bounds
    // (every query stops within quota + largest cell candidates, in at most `nonempty cells` cells) and the kernels
    // below read the real totals from device memory -- no host round trip in the middle of the batch.
    const int64_t* d_tot = nullptr;
    int64_t n_items = 0, n_tabs = 0, n_cand_all = 0;
    {
        static const bool no_bounds = getenv("CIS_NO_BOUNDS") != nullptr;
        // quota <= 0 still visits one cell (search.py:131-132: the test follows the first append)
        const int64_t q_eff = quota < 0 ? 0 : quota;
        const int64_t per_q = (q_eff < ix->n_total ? q_eff : ix->n_total) + ix->max_cell;
        const int64_t items_q = ix->nonempty_cells + per_q / seg_max + 2;
        const bool split_ok = (m->w == 4 || m->w == 8 || m->w == 16 || m->w == 32) && K <= 256;
        if (!no_bounds && !stream_hint && nq <= 64 && L <= MAX_LDS_LIMIT && split_ok && use_all_path(ix, M, K, L, nq) && items_q <= 4096 &&
            (double)nq * (double)per_q < 64.0e6) {
            d_tot = totals;
            n_items = (int64_t)nq * items_q;
            n_tabs = (int64_t)nq * 2 * V;
            n_cand_all = (int64_t)nq * per_q;
            ix->stats_pending_seq = seq;
        }
        // The streaming route when it is CERTAIN before the plan is known -- every query collects at least min(quota, n_total)
        // candidates (search.py:128-133 stops at the quota or at the end of the index), and that alone is past the route's threshold
        // (an exhaustive quota): the same bounds size the workspaces, the kernels read the real totals from device memory, and the
        // host does not stop in the middle of the batch (round 6: the read-back was a 35 us hole in a 0.5 ms exhaustive query).
        static const bool no_stream_bounds = getenv("CIS_STREAM_WAIT") != nullptr;   // A/B runs: the read-back as before
        if (!no_bounds && !no_stream_bounds && stream_hint && split_ok && items_q <= 65536 && (double)nq * (double)items_q < 4.0e6 &&
            (ix->force_stream || (q_eff < ix->n_total ? q_eff : ix->n_total) >= stream_min) && ix->n_total > 0 && nq <= 64) {
            d_tot = totals;
            n_items = (int64_t)nq * items_q;
            n_tabs = (int64_t)nq * 2 * V;
            n_cand_all = (int64_t)nq * per_q;
            ix->stats_pending_seq = seq;
        }
    }
    if (!d_tot) {
        // the plan totals size the rest of the batch: poll the pinned sequence word (a blocking stream synchronisation
        // wakes up tens of microseconds late); past 2 ms -- a stream busy with the caller's earlier work, or an error --
        // fall back to the blocking wait
        static const bool no_poll = getenv("CIS_NO_POLL") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        bool got = false;
        while (!no_poll) {
            if (__atomic_load_n(&ix->h_totals[3], __ATOMIC_ACQUIRE) == seq) { got = true; break; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
        if (!got) {
            CIS_CHECK_HIP(hipStreamSynchronize(st));
            CIS_REQUIRE(__atomic_load_n(&ix->h_totals[3], __ATOMIC_ACQUIRE) == seq, "plan totals did not arrive");
        }
        {   // CIS_HOST_TIMING=1: where the host spends a batch (stderr, every 400 batches): entry -> plan totals requested, the wait for them
            static const bool ht = getenv("CIS_HOST_TIMING") != nullptr;
            if (ht) {
                static std::atomic<long long> n_{0}, wait_ns{0}, front_ns{0};
                const auto t1 = std::chrono::steady_clock::now();
                wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
                front_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t0 - t_entry).count();
                if (++n_ % 400 == 0)
                    fprintf(stderr, "[cis] host timing over %lld batches: front-end enqueue %.1f us, wait for the plan totals %.1f us per batch\n", (long long)n_,
                            front_ns / 1e3 / n_, wait_ns / 1e3 / n_);
            }
        }
        n_items = h_tot[0]; n_tabs = h_tot[1]; n_cand_all = h_tot[2];
        ix->stats_pending_seq = 0;
    }
    CIS_REQUIRE(n_items < ((int64_t)1 << 31) && n_tabs < ((int64_t)1 << 31), "query batch too large");
    // tiny cells on the all-candidates path: entries computed per candidate from px (k_adc_direct), no tables
    const bool direct_elig = !d_tot && use_all_path(ix, M, K, L, nq) && index_has_tiny_cells(ix) && h <= 256 && direct_jp(M, K, m->w) > 0 &&
                             !getenv("CIS_TABLES_UNGROUPED") && !getenv("CIS_NO_DIRECT");
    const bool stream = stream_hint && n_items > 0 && ((m->w == 4 || m->w == 8 || m->w == 16 || m->w == 32) || (scan2_supported(M, K, L) && !ix->force_exact_scan)) &&
                        (ix->force_stream || d_tot != nullptr || n_cand_all / (nq > 0 ? nq : 1) >= stream_min);   // (d_tot: certain, see above)
    if (!stream)
    {
        // workspace budget: per-item hit lists and the float64 tables.  A batch that would need more (e.g. an
        // exhaustive quota: every query visits every cell) is split by the caller and planned again.
        const bool fast_ = scan2_supported(M, K, L) && !ix->force_exact_scan;
        const int64_t S_ = fast_ ? scan2_geom(M, K, L, nq).S : L;
        double need = (double)n_items * S_ * (fast_ ? sizeof(uint64_t) : sizeof(cis_hit)) + (double)n_tabs * nf * K * sizeof(double);
        if (use_all_path(ix, M, K, L, nq)) {  // every candidate's key, plus the selected pairs or the full sort's buffers
            const SelectPlan sp_ = select_plan(L, nq, n_cand_all);
            need = (sp_.select ? 8.0 * (double)n_cand_all + (sp_.sort_lds ? 16.0 : 32.0) * (double)nq * (double)sp_.stride : 32.0 * (double)n_cand_all) +
                   (direct_elig ? (double)n_tabs * h * sizeof(double) : (double)n_tabs * nf * K * sizeof(double));
        }
        // default 24 GB of the 288 GB: thousands of coarse clusters need ~1.6 MB of tables per query (V = 2048, quota 10000)
        const double budget = (getenv("CIS_WORKSPACE_GB") ? atof(getenv("CIS_WORKSPACE_GB")) : 24.0) * 1.0e9;
        if (need > budget && nq > 1) {
            ix->retry_fraction = 0.9 * budget / need;  // the caller plans again with this share of the batch
            return CIS_RETRY_SMALLER;
        }
    }
    if (!d_tot) {
        ix->stats[0] += n_cand_all;
        ix->stats[1] += n_items;
        ix->stats[2] += n_tabs;
    }
    // 3. emit items + table list
    CIS_TRY(mark(1));  // the plan read-back above is part of the front end
    CIS_TRY(ix->w_items.reserve((size_t)(n_items + 1) * sizeof(WorkItem)));
    CIS_TRY(ix->w_tabs.reserve((size_t)(n_tabs + 1) * sizeof(TabDesc)));
    CIS_TRY(ix->w_T.reserve(direct_elig ? 256 : (size_t)(n_tabs + 1) * nf * K * sizeof(double)));
    const bool big = use_all_path(ix, M, K, L, nq);  // ranked over all candidates' exact distances (below)
    const bool tiny_cells = index_has_tiny_cells(ix);
    const bool fast = scan2_supported(M, K, L) && !ix->force_exact_scan;
    const Scan2Geom geom = scan2_geom(M, K, L, nq);
    const Scan3Geom geom3 = scan3_geom(M, K, L, n_items > 0 ? n_cand_all / n_items : 0, ix->force_two_pass);
    const int S = fast ? (use3 ? geom3.S : geom.S) : L;  // hit slots per work item (fast kernels: a full region per wave)
    if (!big && !stream) CIS_TRY(ix->w_hits.reserve((size_t)(n_items + 1) * S * (fast ? sizeof(uint64_t) : sizeof(cis_hit))));
    CIS_TRY(ix->w_hitn.reserve((size_t)(n_items + 1) * 2 * sizeof(int)));
    if (use3) CIS_TRY(ix->w_slack.reserve((size_t)(n_items + 1) * 2 * sizeof(float)));
    WorkItem* items = ix->w_items.as<WorkItem>();
    TabDesc* tabs = ix->w_tabs.as<TabDesc>();
    CIS_TRY(ix->w_tord.reserve((size_t)(n_tabs + 1) * sizeof(int)));
    int* tab_order = ix->w_tord.as<int>();  // table indices grouped by (split, cluster)
    double* T = ix->w_T.as<double>();
    // tiny cells: the float32 copy of px for k_tiny_select lives here (no tables on that path)
    // The float32 copy of the tables is a third of what the tables kernel writes (8 + 4 bytes per entry: 402 MB per C2 batch, and that
    // kernel is bound by its writes); CIS_NO_T32=1 drops it -- the scans then convert the float64 entries while they stage them
    // (tab_f4: the same bits).  The tiny-cell path keeps its float32 px copy in this buffer.
    static const bool no_t32 = getenv("CIS_NO_T32") != nullptr && atoi(getenv("CIS_NO_T32")) != 0;
    const bool drop_t32 = no_t32 && !direct_elig && (m->w == 4 || m->w == 8 || m->w == 16 || m->w == 32) && K <= 256;
    if (!drop_t32) CIS_TRY(ix->w_T32.reserve(direct_elig ? (size_t)(n_tabs + 1) * h * sizeof(float) : (size_t)(n_tabs + 1) * nf * K * sizeof(float)));
    float* T32 = drop_t32 ? nullptr : ix->w_T32.as<float>();
    const size_t tab_lds = (size_t)(2 * h + (h < 256 ? 256 : 0)) * sizeof(double);
    const bool split_tables = (m->w == 4 || m->w == 8 || m->w == 16 || m->w == 32) && K <= 256;
    double* px_buf = nullptr;
    if (split_tables) {
        CIS_TRY(ix->w_px.reserve((size_t)(n_tabs + 1) * h * sizeof(double)));
        px_buf = ix->w_px.as<double>();
    }
    if (ct == CIS_F32) {
        if (par_plan)
            hipLaunchKernelGGL((k_plan_par<float, true>), dim3(nq), dim3(256), 0, st, ix->w_sorted.as<float>(), ix->w_order.as<uint16_t>(),
                               ix->gcount_ptr(), ix->loff_ptr(), nq, V, quota, seg_max, plan, item_off, tab_off, items,
                               tabs, nullptr, grp_base, grp_cur, tab_order, vis_list, plan_fb, vis_cap, nullptr, 0);
        hipLaunchKernelGGL((k_plan<float, true>), dim3(nq), dim3(64), plan_lds, st, ix->w_sorted.as<float>(),
                           ix->w_order.as<uint16_t>(), ix->gcount_ptr(), ix->loff_ptr(), nq, V, quota,
                           seg_max, plan, item_off, tab_off, items, tabs, nullptr, grp_base, grp_cur, tab_order, plan_fb);
        if (n_tabs > 0)
            launch_tables<float>(n_tabs, tab_lds, st, (const float*)xc, m->d_Cs32, m->d_Rt, m->d_mus, m->d_subs, tabs, tab_order, V, h,
                                 m->w, nf, K, D, T, m->prog_w, px_buf, d_tot, direct_elig ? T32 : nullptr);
    } else {
        if (par_plan)
            hipLaunchKernelGGL((k_plan_par<double, true>), dim3(nq), dim3(256), 0, st, ix->w_sorted.as<double>(), ix->w_order.as<uint16_t>(),
                               ix->gcount_ptr(), ix->loff_ptr(), nq, V, quota, seg_max, plan, item_off, tab_off, items,
                               tabs, nullptr, grp_base, grp_cur, tab_order, vis_list, plan_fb, vis_cap, nullptr, 0);
        hipLaunchKernelGGL((k_plan<double, true>), dim3(nq), dim3(64), plan_lds, st, ix->w_sorted.as<double>(),
                           ix->w_order.as<uint16_t>(), ix->gcount_ptr(), ix->loff_ptr(), nq, V, quota,
                           seg_max, plan, item_off, tab_off, items, tabs, nullptr, grp_base, grp_cur, tab_order, plan_fb);
        if (n_tabs > 0)
            launch_tables<double>(n_tabs, tab_lds, st, (const double*)xc, m->d_Cs64, m->d_Rt, m->d_mus, m->d_subs, tabs, tab_order, V,
                                  h, m->w, nf, K, D, T, m->prog_w, px_buf, d_tot, direct_elig ? T32 : nullptr);
    }
    const bool direct = direct_elig;
    if (direct) {
    } else if (split_tables && n_tabs > 0) {
        const bool few = n_tabs <= 1024;   // (d_tot: n_tabs is a bound, the kernel reads the real count)
        dim3 g((unsigned)ceil_div(n_tabs, few ? 8 : 64), (unsigned)nf, 2);
#define CIS_TFP(WW)                                                                                                                                      \
        if (few) hipLaunchKernelGGL((k_tables_from_px<WW, 8>), g, dim3(256), 0, st, px_buf, tabs, (int)n_tabs, m->d_subs, h, nf, K, T, T32, d_tot);      \
        else hipLaunchKernelGGL((k_tables_from_px<WW, 64>), g, dim3(256), 0, st, px_buf, tabs, (int)n_tabs, m->d_subs, h, nf, K, T, T32, d_tot)
        switch (m->w) {
            case 4: CIS_TFP(4); break;
            case 8: CIS_TFP(8); break;
            case 16: CIS_TFP(16); break;
            default: CIS_TFP(32); break;
        }
#undef CIS_TFP
    } else if (fast && n_tabs > 0) {
        const int64_t ne = n_tabs * nf * K;
        hipLaunchKernelGGL(k_tables_f32, dim3((unsigned)ceil_div(ne, 256)), dim3(256), 0, st, T, ne, nf, K, T32, tabs);
    }
    if (stream) {
        // 4''. the HBM-streaming route (lopq_stream.hip): sample -> threshold -> stream -> exact keys of the listed candidates ->
        // ranking (k_select_topl, ties by retrieval index) -> proof; a failed proof hands the batch to the generic path below
        CIS_TRY(mark(2));
        const uint8_t* codes = ix->codes_ptr();
        const int64_t* ids = ix->ids_ptr();
        // queries per slot: a pair costs the launch what one query costs (the codes are the bound), four cost ~1.45 of a pair
        // (profiles/r06_stream_probe.txt) -- worth it when the queries share their cells, i.e. the quota covers most of the index
        const bool shared_cells = quota >= ix->n_total / 2;
        const int G = (nq >= 3 && shared_cells && stream_max_group() >= 4) ? 4 : (nq >= 2 ? 2 : 1);
        const int cap = STREAM_CAP, B = STREAM_B;
        const SelectPlan sp = select_plan(L, nq, (int64_t)nq * cap);
        CIS_REQUIRE(sp.sort_lds, "streaming route: limit above the LDS-ranked range");
        // slots: the work items of one chunk of one cell, G per slot (the slot builder of the scan kernels)
        int64_t CH = ceil_div(ix->max_cell > 0 ? ix->max_cell : 1, (int64_t)seg_max);
        CH = CH < 1 ? 1 : (CH > 16 ? 16 : CH);
        const int64_t nkeys = 2 * ix->ncells * CH;
        const int64_t max_slots = (n_items + nkeys) / G + nkeys + 2;
        CIS_TRY(ix->w_order2.reserve((size_t)(64 + 2 * nkeys + 2 * max_slots * G) * sizeof(int)));
        int* qctr = ix->w_order2.as<int>();
        int* n_slots = qctr + 8;
        int* qstart = qctr + 16;
        int* cell_cnt = qctr + 64;
        int* slot_off = cell_cnt + nkeys;
        int* slots = slot_off + nkeys;
        if (G > 1) {
            const int64_t ninit = max_slots * G > nkeys ? max_slots * G : nkeys;
            hipLaunchKernelGGL(k_slots_init, dim3((unsigned)ceil_div(ninit < 16 ? 16 : ninit, 256)), dim3(256), 0, st, qctr, cell_cnt, (int)nkeys, slots, max_slots * G);
            hipLaunchKernelGGL(k_item_hist, dim3((unsigned)ceil_div(n_items, 256)), dim3(256), 0, st, items, n_items, cell_cnt, (int)ix->ncells, (int)CH, seg_max, d_tot);
            hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(1024), 0, st, cell_cnt, slot_off, (int)nkeys, G, n_slots, qstart, (int)CH);
            hipLaunchKernelGGL(k_item_scatter, dim3((unsigned)ceil_div(n_items, 256)), dim3(256), 0, st, items, n_items, slot_off, cell_cnt, G, slots, (int)ix->ncells, (int)CH, seg_max, d_tot);
        } else
            slots = nullptr;   // one query per slot: slot i = work item i (k_stream_prep)
        // workspace: candidate layout, lists, keys, ranked pairs; the sample buckets live in their own buffer (k_stream_tau leaves them
        // clean for the next batch; fresh memory is set to "empty" here)
        {
            const size_t need = (size_t)nq * B * sizeof(uint32_t);
            if (need > ix->w_bmin.cap) {
                CIS_TRY(ix->w_bmin.reserve(need > (size_t)16 * B * sizeof(uint32_t) ? need : (size_t)16 * B * sizeof(uint32_t)));
                CIS_CHECK_HIP(hipMemsetAsync(ix->w_bmin.p, 0xff, ix->w_bmin.cap, st));
            }
        }
        const size_t n_i64 = (size_t)(n_items + 1) + (size_t)3 * (nq + 2) + (size_t)(max_slots + 2) + (size_t)(max_slots + 1) * ((stream_slot_bytes() + 7) / 8);
        const size_t bytes = n_i64 * 8 + (size_t)(nq + 2) * 4 * 4 + (size_t)nq * cap * 4 + (size_t)nq * cap * 8 + (size_t)2 * nq * sp.stride * 8 + 1024;
        CIS_TRY(ix->w_hits.reserve(bytes));
        int64_t* cand_start = ix->w_hits.as<int64_t>();
        int64_t* seg = cand_start + (n_items + 1);
        unsigned long long* qmin = reinterpret_cast<unsigned long long*>(seg + (nq + 2));
        unsigned long long* qmax = qmin + (nq + 2);
        int64_t* rowoff = reinterpret_cast<int64_t*>(qmax + (nq + 2));     // [max_slots + 1]: rows of the slots before each (k_stream_prep)
        void* sdesc = rowoff + (max_slots + 2);                             // [max_slots] slot records (k_stream_prep)
        uint64_t* skeys = reinterpret_cast<uint64_t*>(rowoff + (max_slots + 2) + (size_t)(max_slots + 1) * ((stream_slot_bytes() + 7) / 8));   // [nq][cap]
        uint64_t* sel_keys = skeys + (size_t)nq * cap;                     // [nq][stride]
        uint64_t* sel_vals = sel_keys + (size_t)nq * sp.stride;
        uint32_t* surv = reinterpret_cast<uint32_t*>(sel_vals + (size_t)nq * sp.stride);  // [nq][cap]
        uint32_t* bmin = ix->w_bmin.as<uint32_t>();                        // [nq][B]
        int* cnt = reinterpret_cast<int*>(surv + (size_t)nq * cap);        // [nq + 2]
        int* nsel = cnt + (nq + 2);
        float* tau = reinterpret_cast<float*>(nsel + (nq + 2));
        int* status = reinterpret_cast<int*>(tau + (nq + 2));
        // candidate layout + slot records + row offsets + resets: one launch of one workgroup
        launch_stream_prep(st, items, n_items, item_off, nq, n_cand_all, slots, n_slots, G, M, cand_start, seg, qmin, qmax, cnt, status, rowoff, sdesc, d_tot);
        // sample: every SS-th row; the k-th smallest of the bucket minima lets about k * SS candidates of a query through -- aim at
        // ~max(4096, 16 limit) of them, with k >= 8 so that the count is stable (relative spread 1 / sqrt(k))
        const int64_t per_q = n_cand_all / nq;
        const int row = 64 * (16 / M);
        int64_t target = 4096 > 16 * L ? 4096 : 16 * L;
        if (const char* e = getenv("CIS_STREAM_TARGET")) target = atoll(e) > 0 ? atoll(e) : target;
        int64_t ss = per_q / row / 4096;            // ~4096 sampled rows per query
        ss = ss < 8 ? 8 : (ss > 4096 ? 4096 : ss);
        int64_t kth = target / ss;
        kth = kth < 8 ? 8 : (kth > B / 4 ? B / 4 : kth);
        if (ix->force_stream && getenv("CIS_STREAM_SS")) { ss = atoll(getenv("CIS_STREAM_SS")); kth = getenv("CIS_STREAM_K") ? atoll(getenv("CIS_STREAM_K")) : kth; }  // tests
        // a lane folds `flush` of its sampled rows into one bucket: ~4 B bucket writes per query (a query's sampled rows x 64 lanes / flush)
        int64_t flush = ceil_div(ceil_div(per_q, (int64_t)row * ss) * 64, (int64_t)4 * B);
        flush = flush < 1 ? 1 : flush;
        const int grid = stream_grid(M, G, K, ceil_div(n_cand_all, (int64_t)row) + n_items);
        launch_stream_scan(M, G, true, grid, st, sdesc, n_slots, rowoff, T32, T, codes, K, tau, bmin, B, (int)ss, (int)flush, surv, cnt, cap);
        launch_stream_tau(st, bmin, B, (int)kth, nq, tau);
        CIS_TRY(mark(5));
        pr.has_scan = true;
        ix->last_scan_kernel = 5;
        launch_stream_scan(M, G, false, grid, st, sdesc, n_slots, rowoff, T32, T, codes, K, tau, bmin, B, (int)ss, (int)flush, surv, cnt, cap);
        CIS_TRY(mark(3));
        launch_stream_keys(M, st, items, cand_start, seg, item_off, n_items, T, codes, K, surv, cnt, cap, nq, skeys, qmin, qmax);
        hipLaunchKernelGGL((k_select_topl<true, 1024>), dim3((unsigned)nq), dim3(1024), sp.lds, st, skeys, seg, cand_start, item_off, qmin, qmax, n_items, L, sp.p2,
                           sp.stride, sel_keys, sel_vals, nsel, (int64_t*)nullptr, (int64_t*)nullptr, (const int*)nullptr, surv, cnt, (int64_t)cap);
        const int64_t sseq = ++ix->stream_batches;
        launch_stream_finish(st, sel_keys, sel_vals, nsel, sp.stride, cnt, cap, seg, tau, nq, L, M, items, ids, plan, out.hits, out.ids, out.dists, out.n_found,
                             out.cells, out.pos, out.visited, status, ix->d_h_totals + 6, sseq);
        CIS_CHECK_HIP(hipGetLastError());
        CIS_TRY(mark(4));
        ix->stats[3] += 1;
        if (ix->profiling) ix->prof.push_back(pr);
        // the proof: two words behind a sequence number in pinned memory (this route serves scans of hundreds of microseconds and
        // more: waiting for their end costs the caller nothing it would not wait for anyway)
        {
            const auto t0 = std::chrono::steady_clock::now();
            bool got = false;
            while (!got) {
                if (__atomic_load_n(&ix->h_totals[8], __ATOMIC_ACQUIRE) == sseq) { got = true; break; }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
            }
            if (!got) {
                CIS_CHECK_HIP(hipStreamSynchronize(st));
                CIS_REQUIRE(__atomic_load_n(&ix->h_totals[8], __ATOMIC_ACQUIRE) == sseq, "the streaming route's status did not arrive");
            }
        }
        if (ix->h_totals[6] == 0 && ix->h_totals[7] == 0) return CIS_OK;
        // a failed proof (an unlucky sample) or an overflowed list (a crowd of equal codes): the generic path answers the batch
        ++ix->stream_fallbacks;
        if (getenv("CIS_STREAM_DEBUG"))
            fprintf(stderr, "[cis] streaming route: %lld failed proofs, %lld overflowed lists of %d queries -> generic path\n", (long long)ix->h_totals[6], (long long)ix->h_totals[7], nq);
        ix->stream_off = true;
        const int saved_stats3 = (int)ix->stats[3];
        for (int i = 0; i < 3; ++i) ix->stats[i] = 0;  // (the generic pass counts the batch again)
        (void)saved_stats3;
        const int rc = search_batch(ix, dQ, q_dtype, nq, quota, L, out, st);
        ix->stream_off = false;
        return rc;
    }
    if (big) {
        // 4'. every candidate's exact distance; then per query either a radix select of the `limit` best (ranked in LDS
        // for limit <= 3072, by a stable segmented sort of the selected pairs above) or, when `limit` is of the order
        // of the candidate count, the stable segmented sort of everything
        CIS_TRY(mark(2));
        const int64_t n_cand = n_cand_all;
        CIS_REQUIRE(n_cand < ((int64_t)1 << 32), "query batch too large for the sorted path");
        const uint8_t* codes = ix->codes_ptr();
        const int64_t* ids = ix->ids_ptr();
        const SelectPlan sp = select_plan(L, nq, n_cand);
        const int64_t n_sel = sp.select ? (int64_t)nq * sp.stride : 0;
        size_t sort_tmp = 0;
        if (!sp.sort_lds)
            CIS_TRY(cis_seg_sort_u64(nullptr, &sort_tmp, nullptr, nullptr, nullptr, nullptr, sp.select ? n_sel : n_cand, nq, nullptr, nullptr, st));
        const size_t tmp_bytes = (sort_tmp + 255) & ~(size_t)255;
        const size_t n_i64 = (size_t)2 * (n_items + 1) + (size_t)6 * (nq + 2);
        const size_t n_pairs = sp.select ? (size_t)(n_cand + 1) + (size_t)(sp.sort_lds ? 2 : 4) * (n_sel + 1) : (size_t)4 * (n_cand + 1);
        CIS_TRY(ix->w_hits.reserve(n_i64 * 8 + n_pairs * 8 + tmp_bytes + 256));
        int64_t* lens = ix->w_hits.as<int64_t>();
        int64_t* cand_start = lens + (n_items + 1);
        int64_t* seg = cand_start + (n_items + 1);
        int64_t* seg_b = seg + (nq + 2);
        int64_t* seg_e = seg_b + (nq + 2);
        int* nsel = reinterpret_cast<int*>(seg_e + (nq + 2));
        unsigned long long* qmin = reinterpret_cast<unsigned long long*>(seg_e + 2 * (nq + 2));
        unsigned long long* qmax = qmin + (nq + 2);
        uint64_t* keys_in = reinterpret_cast<uint64_t*>(qmax + (nq + 2));
        uint64_t* b1 = keys_in + (n_cand + 1);  // full sort: keys_out, vals_in, vals_out; select: sel_keys, sel_vals[, sorted copies]
        const size_t bl = sp.select ? (size_t)(n_sel + 1) : (size_t)(n_cand + 1);
        uint64_t* b2 = b1 + bl;
        uint64_t* b3 = b2 + bl;
        uint64_t* b4 = b3 + bl;
        void* tmp = reinterpret_cast<void*>(((uintptr_t)(sp.select ? (sp.sort_lds ? b3 : b3 + 2 * bl) : b4) + 255) & ~(uintptr_t)255);
        if (n_items > 16384 && !d_tot) {
            const int64_t ntiles = ceil_div(n_items, CAND_TILE);
            CIS_TRY(ix->w_tiles.reserve((size_t)(ntiles + 1) * sizeof(int64_t)));
            int64_t* tile_sums = ix->w_tiles.as<int64_t>();
            hipLaunchKernelGGL(k_cand_tile_sum, dim3((unsigned)ntiles), dim3(256), 0, st, items, n_items, tile_sums);
            hipLaunchKernelGGL(k_cand_tile_scan, dim3(1), dim3(1024), 0, st, tile_sums, ntiles);
            hipLaunchKernelGGL(k_cand_tile_apply, dim3((unsigned)ntiles), dim3(256), 0, st, items, n_items, tile_sums, cand_start);
            hipLaunchKernelGGL(k_seg_begin, dim3((unsigned)ceil_div(nq + 1, 256)), dim3(256), 0, st, cand_start, item_off, nq, n_items, n_cand, seg,
                               sp.select ? qmin : (unsigned long long*)nullptr, qmax);
        } else
        hipLaunchKernelGGL(k_cand_layout, dim3(1), dim3(1024), 0, st, items, n_items, item_off, nq, n_cand, cand_start, seg,
                           sp.select ? qmin : (unsigned long long*)nullptr, qmax, d_tot);
        const uint64_t *rk = nullptr, *rv = nullptr;  // ranked pairs
        if (!sp.select) {
            uint64_t *keys_out = b1, *vals_in = b2, *vals_out = b3;
            if (n_items > 0) {
                if (!(direct && launch_adc_direct(M, K, m->w, st, items, cand_start, seg, item_off, px_buf, m->d_subs, codes, h, nq, keys_in, vals_in, nullptr, nullptr)))
                    launch_adc_all(n_items, st, items, cand_start, T, codes, M, K, keys_in, vals_in, nullptr, nullptr, nullptr, seg, item_off, nq, tiny_cells);
                size_t b = tmp_bytes;
                CIS_TRY(cis_seg_sort_u64(tmp, &b, keys_in, keys_out, vals_in, vals_out, n_cand, nq, seg, seg + 1, st));
            }
            rk = keys_out; rv = vals_out;
        } else {
            uint64_t *sel_keys = b1, *sel_vals = b2;
            // tiny cells, limit a small share of the quota: byte-table prefilter + exact keys of the survivors in ONE kernel per
            // query (k_tiny_select); the queries it flags go through the exact kernels below
            int* fbflag = nullptr;
            // (measured at V = 2048, 8192 queries, limit 100: quota 10000 5.16 against 6.35 ms for the exact kernels, quota 1000 2.19 against
            // 1.36 -- the sample, the row copies and the sort are fixed costs per query: from 8192 candidates on)
            const int64_t tiny_min_quota = getenv("CIS_TINY_MIN_QUOTA") ? atoll(getenv("CIS_TINY_MIN_QUOTA")) : 8192;
            if (direct && sp.sort_lds && n_items > 0 && (int64_t)L * 8 <= (int64_t)quota && (int64_t)quota >= tiny_min_quota && !getenv("CIS_NO_TINY")) {
                int ncmax = 0;
                const int tch = tiny_pool(M, K, m->w, h, L, (int64_t)quota + ix->max_cell, &ncmax);  // bytes of the per-query pool
                if (tch > 0) {
                    fbflag = nsel + (nq + 2);
                    static const bool tiny_dbg = getenv("CIS_TINY_DEBUG") != nullptr;
                    unsigned int* dbg = nullptr;
                    if (tiny_dbg) {
                        CIS_TRY(ix->w_slack.reserve(128));
                        dbg = ix->w_slack.as<unsigned int>();
                        CIS_CHECK_HIP(hipMemsetAsync(dbg, 0, 96, st));
                    }
                    if (!launch_tiny(M, m->w, st, items, cand_start, seg, item_off, tab_off, plan, px_buf, m->d_subs, codes, h, nq, L, ncmax, tch,
                                     sp.stride, sel_keys, sel_vals, nsel, keys_in, qmin, qmax, fbflag, T32, dbg))
                        fbflag = nullptr;
                    else if (tiny_dbg) {
                        unsigned int hd[24];
                        CIS_CHECK_HIP(hipMemcpyAsync(hd, dbg, 96, hipMemcpyDeviceToHost, st));
                        CIS_CHECK_HIP(hipStreamSynchronize(st));
                        fprintf(stderr, "[tiny] queries %u  flagged %u  survivors/query %.1f  mean s* %.1f  (pool %d B, ncmax %d; tables/query %.1f, items/query %.1f)\n", hd[0], hd[1],
                                hd[0] ? (double)hd[2] / hd[0] : 0.0, hd[0] ? (double)hd[3] / hd[0] : 0.0, tch, ncmax, (double)n_tabs / nq, (double)n_items / nq);
                        // 100 MHz ticks of thread 0 per phase, summed over all queries: us per query
                        fprintf(stderr, "[tiny] us/query: top %.1f layout %.1f sample %.1f lookups %.1f rows %.1f tables %.1f last lookups %.1f hist+gather %.1f exact %.1f sums+sort %.1f out %.1f\n",
                                hd[4] / 100.0 / nq, hd[5] / 100.0 / nq, hd[6] / 100.0 / nq, hd[7] / 100.0 / nq, hd[8] / 100.0 / nq, hd[9] / 100.0 / nq,
                                hd[10] / 100.0 / nq, hd[11] / 100.0 / nq, hd[12] / 100.0 / nq, hd[13] / 100.0 / nq, hd[14] / 100.0 / nq);
                        fprintf(stderr, "[tiny] half tables with a candidate: %.1f per query; centroid loads %.1f us/query\n", (double)hd[15] / nq, hd[16] / 100.0 / nq);
                    }
                }
            }
            if (fbflag) {
            } else
            if (n_items > 0 && !(direct && launch_adc_direct(M, K, m->w, st, items, cand_start, seg, item_off, px_buf, m->d_subs, codes, h, nq, keys_in, nullptr, qmin, qmax)))
                launch_adc_all(n_items, st, items, cand_start, T, codes, M, K, keys_in, nullptr, qmin, qmax, d_tot, seg, item_off, nq, tiny_cells);
            if (sp.sort_lds) {
                // fewer queries than CUs: one large workgroup per query walks its keys faster; else two 512-thread ones per CU
                if (nq <= 256)
                    hipLaunchKernelGGL((k_select_topl<true, 1024>), dim3((unsigned)nq), dim3(1024), sp.lds, st, keys_in, seg, cand_start, item_off,
                                       qmin, qmax, n_items, L, sp.p2, sp.stride, sel_keys, sel_vals, nsel, (int64_t*)nullptr, (int64_t*)nullptr,
                                       (const int*)fbflag);
                else
                    hipLaunchKernelGGL((k_select_topl<true, 512>), dim3((unsigned)nq), dim3(512), sp.lds, st, keys_in, seg, cand_start, item_off,
                                       qmin, qmax, n_items, L, sp.p2, sp.stride, sel_keys, sel_vals, nsel, (int64_t*)nullptr, (int64_t*)nullptr,
                                       (const int*)fbflag);
                rk = sel_keys; rv = sel_vals;
            } else {
                uint64_t *srt_keys = b3, *srt_vals = b3 + bl;
                if (nq <= 256)
                    hipLaunchKernelGGL((k_select_topl<false, 1024>), dim3((unsigned)nq), dim3(1024), sp.lds, st, keys_in, seg, cand_start, item_off,
                                       qmin, qmax, n_items, L, sp.p2, sp.stride, sel_keys, sel_vals, nsel, seg_b, seg_e, (const int*)nullptr);
                else
                    hipLaunchKernelGGL((k_select_topl<false, 512>), dim3((unsigned)nq), dim3(512), sp.lds, st, keys_in, seg, cand_start, item_off,
                                       qmin, qmax, n_items, L, sp.p2, sp.stride, sel_keys, sel_vals, nsel, seg_b, seg_e, (const int*)nullptr);
                size_t b = tmp_bytes;
                CIS_TRY(cis_seg_sort_u64(tmp, &b, sel_keys, srt_keys, sel_vals, srt_vals, n_sel, nq, seg_b, seg_e, st));
                rk = srt_keys; rv = srt_vals;
            }
        }
        CIS_TRY(mark(3));
        hipLaunchKernelGGL(k_emit_sorted, dim3((unsigned)ceil_div(L, 1024) < 64 ? (unsigned)ceil_div(L, 1024) : 64, (unsigned)nq), dim3(256), 0, st,
                           rk, rv, seg, sp.select ? nsel : (const int*)nullptr, sp.stride, items, ids, nq, L, out.hits, out.ids, out.dists,
                           out.n_found, out.cells, out.pos);
        if (out.visited)
            hipLaunchKernelGGL(k_copy_visited, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, st, plan, nq, out.visited);
        CIS_CHECK_HIP(hipGetLastError());
        CIS_TRY(mark(4));
        if (ix->profiling) ix->prof.push_back(pr);
        return CIS_OK;
    }
    // 4. ADC scan + block top-k
    CIS_TRY(mark(2));
    if (n_items > 0) {
        pr.has_scan = true;
        const uint8_t* codes = ix->codes_ptr();
        const int64_t* ids = ix->ids_ptr();
        cis_hit* hits = ix->w_hits.as<cis_hit>();
        int* hitn = ix->w_hitn.as<int>();
        if (fast) {
            // slot list: work items grouped by coarse cell (counting sort; skipped for huge V), G per slot
            const bool sort_items = ix->ncells <= 65536;
            // k_adc_scan5 (round 5): where the sampled form k_adc_scan4 would run -- one threshold per query for the whole batch
            // instead of one per slot, eight queries per slot.  MEASURED AND LEFT OFF (profiles/r05g_*, r05h_*: C4 sample 64 + thresholds 17 +
            // main pass 200 + check 12 us against 231 us for the whole of k_adc_scan4, and a merge of 134 instead of 76 us for the longer
            // lists; both pipes ~40 % busy at three workgroups per CU: the loop is bound by latency, not by instructions).  CIS_SCAN5=1
            // routes large batches to it, scan mode 7 forces it (tests: every search test passes on it).
            static const int env_s5 = getenv("CIS_SCAN5") ? atoi(getenv("CIS_SCAN5")) : 0;
            const bool use5 = use3 && geom3.two_pass == 2 && scan5_supported(M, K, L) && (ix->force_scan5 || (env_s5 != 0 && !ix->force_scan3 && L <= 128));
            const int G = use5 ? 8 : (use3 ? geom3.G : geom.G);
            // chunks per cell that get their own slot keys: what the largest cell needs (all shards' sizes bound this shard's)
            int64_t CH = use3 ? ceil_div(ix->max_cell > 0 ? ix->max_cell : 1, (int64_t)seg_max) : 1;
            CH = CH < 1 ? 1 : (CH > 16 ? 16 : CH);
            const int64_t nkeys = 2 * ix->ncells * CH;
            const int64_t max_slots = sort_items ? (n_items + nkeys) / G + nkeys + 2 : n_items;
            CIS_TRY(ix->w_order2.reserve((size_t)(64 + 2 * nkeys + 2 * max_slots * G) * sizeof(int)));
            int* qctr = ix->w_order2.as<int>();  // [8] queue counters, [8] n_slots (first), [9] queue starts, [32] fall-back header (scan3)
            int* n_slots = qctr + 8;
            int* qstart = qctr + 16;
            int* fhdr = qctr + 32;
            int* cell_cnt = qctr + 64;
            int* slot_off = cell_cnt + nkeys;
            int* slots = slot_off + nkeys;
            int* fslots = slots + max_slots * G;
            if (sort_items) {
                const int64_t ninit = max_slots * G > nkeys ? max_slots * G : nkeys;
                hipLaunchKernelGGL(k_slots_init, dim3((unsigned)ceil_div(ninit < 16 ? 16 : ninit, 256)), dim3(256), 0, st, qctr,
                                   cell_cnt, (int)nkeys, slots, max_slots * G);
                hipLaunchKernelGGL(k_item_hist, dim3((unsigned)ceil_div(n_items, 256)), dim3(256), 0, st, items, n_items, cell_cnt,
                                   (int)ix->ncells, (int)CH, seg_max);
                hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(1024), 0, st, cell_cnt, slot_off, (int)nkeys, G, n_slots, qstart, (int)CH);
                hipLaunchKernelGGL(k_item_scatter, dim3((unsigned)ceil_div(n_items, 256)), dim3(256), 0, st, items, n_items,
                                   slot_off, cell_cnt, G, slots, (int)ix->ncells, (int)CH, seg_max);
            } else {
                CIS_CHECK_HIP(hipMemsetAsync(qctr, 0, 64 * sizeof(int), st));
                hipLaunchKernelGGL(k_identity_slots, dim3((unsigned)ceil_div(n_items < 9 ? 9 : n_items, 256)), dim3(256), 0, st, n_items, G,
                                   slots, n_slots, qstart);
            }
            CIS_TRY(mark(5));
            ix->last_scan_kernel = use5 ? 6 : (use3 ? (geom3.two_pass == 2 ? 4 : 3) : 2);  // 4: the sampled single-pass form k_adc_scan4 does the work (k_adc_scan3 only its fall-back slots)
            if (use5) {
                CIS_TRY(ix->w_s5.reserve(scan5_workspace_bytes(nq)));
                launch_scan5(M, geom3, n_items, nq, st, items, tabs, slots, n_slots, plan, T, T32, codes, K, L, qctr, ix->w_hits.as<uint64_t>(), hitn,
                             ix->w_slack.as<float>(), qbound, fhdr, fslots, ix->w_s5.p, nullptr);
            } else
            if (use3) {
                Scan3Geom g3 = geom3;
                // M = 16 on the sampled form: the saturating scale of k_adc_scan4 (sums of the near candidates at this fraction of the entry cap)
                const float sat16 = getenv("CIS_S4_SAT") ? (float)atof(getenv("CIS_S4_SAT")) : 0.75f;
                if (M == 16 && g3.two_pass == 2) g3.sat = sat16;
                launch_scan3(M, g3, n_items, st, items, tabs, slots, n_slots, T, T32, codes, K, L, qctr, ix->w_hits.as<uint64_t>(), hitn, ix->w_slack.as<float>(), qbound, fhdr, fslots);
                if (g3.sat > 0.f && ix->h_totals)  // (slots, fall-back slots) for the back-off above
                    CIS_CHECK_HIP(hipMemcpyAsync(&ix->h_totals[4], qctr + 9, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
            }
            else
                launch_scan2(M, geom, n_items, st, items, slots, n_slots, T, T32, codes, ids, K, L, qctr, ix->w_hits.as<uint64_t>(), hitn, qbound);
        }
        else {
            CIS_TRY(mark(5));
            ix->last_scan_kernel = 1;
            launch_scan_exact(M, n_items, st, items, T, codes, ids, K, L, S, nullptr, hits, hitn);
        }
        ix->stats[3] += 1;
    }
    // 5. per-query merge
    CIS_TRY(mark(3));
    {
        const cis_hit* hits = ix->w_hits.as<cis_hit>();
        const int* hitn = ix->w_hitn.as<int>();
        if (fast) {
            // survivors of the float32 scan: exact re-scoring + ranking (limit <= 440 here)
            const uint64_t* surv = ix->w_hits.as<uint64_t>();
            const uint8_t* codes = ix->codes_ptr();
            const int64_t* ids = ix->ids_ptr();
            // several lists per query (short cells): the variant whose fast path holds 512 survivors per query
            const bool many = n_items > nq + nq / 4;
#define CIS_MERGE_SURV(CAP, MT)                                                                                              \
    do {                                                                                                                     \
        if (many)                                                                                                            \
            hipLaunchKernelGGL((k_merge_survivors<CAP, MT, 8>), dim3((unsigned)ceil_div(nq, 4)), dim3(256), (size_t)4 * CAP * 16, st, \
                               surv, hitn, item_off, items, T, codes, ids, nq, M, K, L, S, out.hits, out.ids, out.dists,     \
                               out.n_found, out.cells, out.pos, plan, out.visited, use3 ? ix->w_slack.as<float>() : (const float*)nullptr);                            \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_merge_survivors<CAP, MT, 4>), dim3((unsigned)ceil_div(nq, 4)), dim3(256), (size_t)4 * CAP * 16, st, \
                               surv, hitn, item_off, items, T, codes, ids, nq, M, K, L, S, out.hits, out.ids, out.dists,     \
                               out.n_found, out.cells, out.pos, plan, out.visited, use3 ? ix->w_slack.as<float>() : (const float*)nullptr);                            \
    } while (0)
#define CIS_MERGE_SURV_M(CAP)                                                                          \
    do {                                                                                               \
        if (M == 4) CIS_MERGE_SURV(CAP, 4); else if (M == 8) CIS_MERGE_SURV(CAP, 8); else CIS_MERGE_SURV(CAP, 16); \
    } while (0)
            if (L <= 128) CIS_MERGE_SURV_M(256);
            else if (L <= 256) CIS_MERGE_SURV_M(512);
            else if (L <= 440) CIS_MERGE_SURV_M(1024);
            else CIS_MERGE_SURV_M(2048);
#undef CIS_MERGE_SURV_M
#undef CIS_MERGE_SURV
        } else if (L <= 512)
            hipLaunchKernelGGL(k_merge_items<1024>, dim3(nq), dim3(256), (size_t)1024 * 24 + 16, st, hits, hitn, item_off, L, S, out.hits,
                               out.ids, out.dists, out.n_found, out.cells, out.pos);
        else if (L <= 1024)
            hipLaunchKernelGGL(k_merge_items<2048>, dim3(nq), dim3(256), (size_t)2048 * 24 + 16, st, hits, hitn, item_off, L, S, out.hits,
                               out.ids, out.dists, out.n_found, out.cells, out.pos);
        else
            hipLaunchKernelGGL(k_merge_items<4096>, dim3(nq), dim3(256), (size_t)4096 * 24 + 16, st, hits, hitn, item_off, L, S, out.hits,
                               out.ids, out.dists, out.n_found, out.cells, out.pos);
    }
    if (out.visited && !fast)  // the survivor merge writes `visited` itself
        hipLaunchKernelGGL(k_copy_visited, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, st, plan, nq, out.visited);
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

static int search_all(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, int L, const SearchOut& out,
                      hipStream_t st) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(q_dtype == CIS_F32 || q_dtype == CIS_F64, "q_dtype must be 4 or 8");
    CIS_REQUIRE(nq >= 0, "nq must be >= 0");
    CIS_REQUIRE(!ix->orphaned, "this view's base index was destroyed: close views before their base");
    CIS_TRY(cis_index_ready(ix->base ? ix->base : ix));
    ix->sync_from_base();  // a view reads the storage of its base
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    for (int i = 0; i < 4; ++i) ix->stats[i] = 0;
    ix->stats_pending_seq = 0;
    // a batch that did not fit the workspace is planned again smaller; the size that fitted is remembered for the next
    // call with the same quota (the plan of a rejected batch is wasted front-end time)
    int batch = (ix->batch_hint > 0 && ix->batch_hint_quota == quota) ? ix->batch_hint : QUERY_BATCH;
    for (int a = 0; a < nq;) {
        const int bn = (nq - a < batch) ? (nq - a) : batch;
        const char* q = (const char*)dQ + (size_t)a * ix->m->D_in * q_dtype;
        int rc;
        if (L > 0) {
            rc = search_batch(ix, q, q_dtype, bn, quota, L, out.at(a, L), st);
        } else {  // limit 0: only `visited` is defined
            SearchOut o{};
            o.visited = out.visited ? out.visited + a : nullptr;
            CIS_TRY(ix->w_part.reserve((size_t)bn * sizeof(cis_hit)));
            o.hits = ix->w_part.as<cis_hit>();
            rc = search_batch(ix, q, q_dtype, bn, quota, 1, o, st);
        }
        if (rc == CIS_RETRY_SMALLER) {
            int nb = (int)((double)bn * ix->retry_fraction);
            if (nb > bn / 2 && bn / 2 >= 64) nb = nb / 64 * 64;
            if (nb >= bn) nb = bn / 2;
            batch = nb > 1 ? nb : 1;
            ix->batch_hint = batch;
            ix->batch_hint_quota = quota;
            continue;
        }
        CIS_TRY(rc);
        a += bn;
    }
    return CIS_OK;
}

// ---- routed cell-sharded search (round 5): which ranks own the cells a query visits ---------------------------------------------
// The all-gather protocol hands every rank the whole batch: projection, cell ranking and walk are done `world` times over.  Routed,
// a query's HOME rank (1 / world of the batch each) walks the multisequence against the cell sizes of the whole index -- the same
// walk as k_plan (lopq/lopq/search.py:58-82, :128-133), no items -- and notes the owner of every non-empty visited cell; only those
// ranks (one or two at V = 16) receive the query (columbiaimagesearch_amd/distributed.py: RoutedSearcher).
template <typename CT>
__global__ __launch_bounds__(64) void k_plan_owners(const CT* __restrict__ sorted, const uint16_t* __restrict__ order,
                                                    const int64_t* __restrict__ gcount, const int32_t* __restrict__ owner, int world,
                                                    int nq, int V, int64_t quota, unsigned long long* __restrict__ mask,
                                                    int32_t* __restrict__ visited_out) {
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
    int visited = 0, rows = 1;
    int64_t retrieved = 0;
    unsigned long long mk = 0ull;
    const int64_t total_cells = (int64_t)V * V;
    while ((int64_t)visited < total_cells) {
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
        if (bij == ~0u) break;
        const int bi = (int)(bij >> 16), bj = (int)(bij & 0xffff);
        const int64_t cell = (int64_t)o0[bi] * V + o1[bj];
        const int64_t gc = gcount[cell];
        if (gc > 0) mk |= 1ull << (owner ? owner[cell] : (int)(cell % world));
        visited += 1;
        retrieved += gc;
        __syncthreads();
        if (lane == 0) t[bi] = bj + 1;
        if (bi + 2 > rows) rows = (bi + 2 < V) ? bi + 2 : V;
        __syncthreads();
        if (retrieved >= quota) break;
    }
    if (lane == 0) {
        mask[q] = mk;
        if (visited_out) visited_out[q] = visited;
    }
}

extern "C" int cis_index_query_owners_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, uint64_t* d_mask,
                                          int32_t* d_visited, void* stream) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(q_dtype == CIS_F32 || q_dtype == CIS_F64, "q_dtype must be 4 or 8");
    CIS_REQUIRE(nq >= 0 && (nq == 0 || (dQ && d_mask)), "NULL buffer");
    CIS_REQUIRE(!ix->orphaned, "this view's base index was destroyed: close views before their base");
    CIS_REQUIRE(ix->world >= 1 && ix->world <= 64, "owner masks hold 64 ranks");
    if (nq == 0) return CIS_OK;
    CIS_TRY(cis_index_ready(ix->base ? ix->base : ix));
    ix->sync_from_base();
    cis_model* m = ix->m;
    CIS_CHECK_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    const int V = m->V, D = m->D;
    CIS_REQUIRE(V <= 4096, "owner walk: V <= 4096");
    const void* xp = dQ;
    int xp_dtype = q_dtype;
    if (m->has_pca) {
        CIS_TRY(ix->w_xp.reserve((size_t)nq * D * sizeof(float)));
        CIS_TRY(cis_dev_apply_pca(m, dQ, q_dtype, nq, ix->w_xp.as<float>(), st, &ix->w_y64));
        xp = ix->w_xp.p;
        xp_dtype = CIS_F32;
    }
    const void* xc;
    int ct;
    CIS_TRY(cis_dev_coarse_type(m, xp, xp_dtype, nq, &xc, &ct, st, &ix->w_x64));
    const size_t csz = (ct == CIS_F32) ? 4 : 8;
    CIS_TRY(ix->w_cd.reserve((size_t)2 * nq * V * csz));
    CIS_TRY(ix->w_sorted.reserve((size_t)2 * nq * V * csz));
    CIS_TRY(ix->w_order.reserve((size_t)2 * nq * V * sizeof(uint16_t)));
    {
        const void* grp_before = ix->w_grp.p;
        CIS_TRY(ix->w_grp.reserve((size_t)(GRP_WORDS(V) + 2 * GRP_TILES(V)) * sizeof(int)));
        if (ix->w_grp.p != grp_before) CIS_CHECK_HIP(hipMemsetAsync(ix->w_grp.p, 0, ix->w_grp.cap, st));
    }
    int* grp_cnt = ix->w_grp.as<int>();  // the rank kernels leave the table-group counters zeroed, as every search expects to find them
    CIS_TRY(cis_launch_sqdist_both(m, xc, ct, nq, ix->w_cd.p, st));
    int Vp2 = 64;
    while (Vp2 < V) Vp2 <<= 1;
    const cis_index* own = ix->base ? ix->base : ix;  // a view reads the owner table of its base
    const int32_t* d_owner = own->owner.empty() ? nullptr : own->d_owner.as<int32_t>();
    CIS_REQUIRE(own->owner.empty() || d_owner != nullptr, "owner table not on the device");
    if (ct == CIS_F32) {
        if (V > 256) hipLaunchKernelGGL(k_rank_sort<float>, dim3(nq, 2), dim3(256), (size_t)Vp2 * 16, st, ix->w_cd.as<float>(), nq, V, Vp2,
                                        ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>(), grp_cnt);
        else hipLaunchKernelGGL(k_rank<float>, dim3(nq, 2), dim3(V <= 64 ? 64 : 256), (size_t)V * 8, st, ix->w_cd.as<float>(), nq, V,
                                ix->w_order.as<uint16_t>(), ix->w_sorted.as<float>(), grp_cnt);
        hipLaunchKernelGGL(k_plan_owners<float>, dim3(nq), dim3(64), (size_t)V * sizeof(int), st, ix->w_sorted.as<float>(), ix->w_order.as<uint16_t>(),
                           ix->gcount_ptr(), d_owner, ix->world, nq, V, quota, (unsigned long long*)d_mask, d_visited);
    } else {
        if (V > 256) hipLaunchKernelGGL(k_rank_sort<double>, dim3(nq, 2), dim3(256), (size_t)Vp2 * 16, st, ix->w_cd.as<double>(), nq, V, Vp2,
                                        ix->w_order.as<uint16_t>(), ix->w_sorted.as<double>(), grp_cnt);
        else hipLaunchKernelGGL(k_rank<double>, dim3(nq, 2), dim3(V <= 64 ? 64 : 256), (size_t)V * 8, st, ix->w_cd.as<double>(), nq, V,
                                ix->w_order.as<uint16_t>(), ix->w_sorted.as<double>(), grp_cnt);
        hipLaunchKernelGGL(k_plan_owners<double>, dim3(nq), dim3(64), (size_t)V * sizeof(int), st, ix->w_sorted.as<double>(), ix->w_order.as<uint16_t>(),
                           ix->gcount_ptr(), d_owner, ix->world, nq, V, quota, (unsigned long long*)d_mask, d_visited);
    }
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

// Routing tables of a home rank: slot[d][i] = the row of query i in the buffer that goes to rank d (-1: not sent), in query order;
// cnt[d] = rows used (at most cap; *overflow = 1 when a destination would need more).  One workgroup per destination.
__global__ __launch_bounds__(1024) void k_route_slots(const unsigned long long* __restrict__ mask, int nq, int cap, int32_t* __restrict__ slot,
                                                      int32_t* __restrict__ cnt, int32_t* __restrict__ overflow) {
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nq; i0 += 1024) {
        const int i = i0 + tid;
        const bool f = i < nq && ((mask[i] >> d) & 1ull);
        const unsigned long long b = __builtin_amdgcn_ballot_w64(f);
        const int before = __builtin_popcountll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wv] = __builtin_popcountll(b);
        __syncthreads();
        int wbase = s_base;
        for (int w = 0; w < wv; ++w) wbase += s_w[w];
        if (i < nq) {
            const int pos = wbase + before;
            slot[(int64_t)d * nq + i] = (f && pos < cap) ? pos : -1;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = s_base;
            for (int w = 0; w < 16; ++w) tot += s_w[w];
            s_base = tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        cnt[d] = s_base < cap ? s_base : cap;
        if (s_base > cap) atomicExch(overflow, 1);
    }
}

__global__ void k_route_rows(const uint32_t* __restrict__ q, int nq, int W /* 32-bit words per row */, const int32_t* __restrict__ slot, int cap,
                             uint32_t* __restrict__ out) {
    const int i = blockIdx.x, d = blockIdx.y;
    const int sl = slot[(int64_t)d * nq + i];
    if (sl < 0) return;
    const uint32_t* src = q + (int64_t)i * W;
    uint32_t* dst = out + ((int64_t)d * cap + sl) * W;
    for (int k = threadIdx.x; k < W; k += blockDim.x) dst[k] = src[k];
}

extern "C" int cis_route_queries_dev(const void* d_q, int nq, int row_bytes, const uint64_t* d_mask, int world, int cap, void* d_out_q,
                                     int32_t* d_slot, int32_t* d_cnt, int32_t* d_overflow, void* stream) {
    CIS_REQUIRE(nq >= 0 && row_bytes > 0 && row_bytes % 4 == 0 && world >= 1 && world <= 64 && cap >= 1, "route: sizes out of range");
    CIS_REQUIRE(d_slot && d_cnt && d_overflow && (nq == 0 || (d_q && d_mask && d_out_q)), "NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    const int W = row_bytes / 4;
    CIS_CHECK_HIP(hipMemsetAsync(d_overflow, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_route_slots, dim3(world), dim3(1024), 0, st, (const unsigned long long*)d_mask, nq, cap, d_slot, d_cnt, d_overflow);
    if (nq > 0) hipLaunchKernelGGL(k_route_rows, dim3(nq, world), dim3(W >= 256 ? 256 : 64), 0, st, (const uint32_t*)d_q, nq, W, d_slot, cap, (uint32_t*)d_out_q);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

// Merge tables of the routed search's return trip: the list of home query i from rank d is row base[d] + slot[d][i] of the returned
// buffer (L records per row, ranked, valid hits first).  off = the row's first record, cnt = its valid hits (0: rank d was not asked).
struct RouteBase { int64_t v[64]; };
__global__ void k_routed_tables(const int32_t* __restrict__ slot, int world, int nq, RouteBase base, const cis_hit* __restrict__ hits, int L,
                                int64_t* __restrict__ off, int32_t* __restrict__ cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)world * nq) return;
    const int d = (int)(t / nq);
    const int sl = slot[t];
    if (sl < 0) { off[t] = 0; cnt[t] = 0; return; }
    const int64_t row = base.v[d] + sl;
    const cis_hit* h = hits + row * L;
    int lo = 0, hi = L;  // first empty slot (id < 0): the valid hits are a prefix
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (h[mid].id >= 0) lo = mid + 1; else hi = mid;
    }
    off[t] = row * L;
    cnt[t] = lo;
}

extern "C" int cis_routed_merge_tables_dev(const int32_t* d_slot, int world, int nq, const int64_t* h_base, const cis_hit* d_hits, int L,
                                           int64_t* d_off, int32_t* d_cnt, void* stream) {
    CIS_REQUIRE(world >= 1 && world <= 64 && nq >= 0 && L >= 0, "routed merge tables: sizes out of range");
    CIS_REQUIRE(nq == 0 || (d_slot && h_base && d_off && d_cnt && (L == 0 || d_hits)), "NULL buffer");
    if (nq == 0) return CIS_OK;
    RouteBase b;
    for (int d = 0; d < 64; ++d) b.v[d] = d < world ? h_base[d] : 0;
    const int64_t n = (int64_t)world * nq;
    hipLaunchKernelGGL(k_routed_tables, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, d_slot, world, nq, b, d_hits, L, d_off, d_cnt);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

extern "C" int cis_index_search_partial_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota,
                                            int limit, cis_hit* d_hits, int32_t* d_visited, void* stream) {
    int L;
    CIS_TRY(effective_limit(quota, limit, &L));
    SearchOut o{};
    o.hits = d_hits;
    o.visited = d_visited;
    return search_all(ix, dQ, q_dtype, nq, quota, L, o, (hipStream_t)stream);
}

// exclusive scan of the per-query hit counts (single block) and the packing of the valid row prefixes
__global__ __launch_bounds__(1024) void k_pack_scan(const int32_t* __restrict__ cnt, int nq, int64_t* __restrict__ off,
                                                    int64_t* __restrict__ total) {
    __shared__ int64_t s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int64_t run = 0;  // all queries before this block-sized chunk
    for (int base = 0; base < nq; base += 1024) {
        const int q = base + tid;
        const int64_t c = q < nq ? cnt[q] : 0;
        int64_t x = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        int64_t wp = 0, all = 0;
        for (int k = 0; k < 16; ++k) { const int64_t y = s_w[k]; if (k < wv) wp += y; all += y; }
        if (q < nq) off[q] = run + wp + x - c;
        run += all;
        __syncthreads();
    }
    if (tid == 0) *total = run;
}

__global__ void k_pack_hits(const cis_hit* __restrict__ dense /* [nq][L] */, const int32_t* __restrict__ cnt,
                            const int64_t* __restrict__ off, int nq, int L, cis_hit* __restrict__ packed) {
    const int q = blockIdx.x;
    const int c = cnt[q];
    const int64_t o = off[q];
    for (int x = threadIdx.x; x < c; x += blockDim.x) packed[o + x] = dense[(int64_t)q * L + x];
}

extern "C" int cis_index_search_partial_packed_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, int limit,
                                                   cis_hit* d_packed, int32_t* d_cnt, int64_t* d_off, int64_t* d_total,
                                                   int32_t* d_visited, void* stream) {
    int L;
    CIS_TRY(effective_limit(quota, limit, &L));
    CIS_REQUIRE(ix != nullptr && d_cnt && d_off && d_total && (L == 0 || d_packed), "NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    if (nq == 0 || L == 0) {
        CIS_CHECK_HIP(hipMemsetAsync(d_total, 0, sizeof(int64_t), st));
        if (nq > 0) {
            CIS_CHECK_HIP(hipMemsetAsync(d_cnt, 0, (size_t)nq * sizeof(int32_t), st));
            CIS_CHECK_HIP(hipMemsetAsync(d_off, 0, (size_t)nq * sizeof(int64_t), st));
        }
        if (nq == 0) return CIS_OK;
    }
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    CIS_TRY(ix->w_part.reserve((size_t)nq * (L > 0 ? L : 1) * sizeof(cis_hit)));
    SearchOut o{};
    o.hits = ix->w_part.as<cis_hit>();
    o.n_found = d_cnt;
    o.visited = d_visited;
    CIS_TRY(search_all(ix, dQ, q_dtype, nq, quota, L, o, st));
    if (L == 0) return CIS_OK;
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, d_cnt, nq, d_off, d_total);
    hipLaunchKernelGGL(k_pack_hits, dim3(nq), dim3(64), 0, st, ix->w_part.as<cis_hit>(), d_cnt, d_off, nq, L, d_packed);
    CIS_CHECK_HIP(hipGetLastError());
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

// Records of shard w for query q that really ARRIVED: with the fixed-size exchange a shard that held more than `stride` records was
// cut there (the overflow flag tells the caller to repeat the exchange); the merge must not read past the cut.
static __device__ __forceinline__ int arrived(const int32_t* __restrict__ cnt, const int64_t* __restrict__ off, int64_t stride, int w, int nq, int q) {
    const int64_t o = off[(int64_t)w * nq + q];
    const int64_t room = stride - o;
    const int c = cnt[(int64_t)w * nq + q];
    if (stride == 0) return c;  // one flat buffer, absolute offsets, nothing was cut (the routed search's return trip)
    return room <= 0 ? 0 : (c < room ? c : (int)room);
}

// Merge of PACKED per-shard hit lists: shard w contributed parts[w*stride + off[w*nq+q] .. + cnt[w*nq+q]) for query q
// (its valid hits only, in query order).  One wave per query; same ranking key as everywhere: (dist, visit_rank, pos).
template <int CAPM, int WPB /* waves (= queries) per workgroup */>
__global__ __launch_bounds__(WPB * 64) void k_merge_packed(const cis_hit* __restrict__ parts, int world, int64_t stride,
                                                      const int64_t* __restrict__ off, const int32_t* __restrict__ cnt, int nq,
                                                      int limit, int64_t* __restrict__ out_ids, double* __restrict__ out_dists,
                                                      int* __restrict__ out_n, int32_t* __restrict__ out_cells,
                                                      uint32_t* __restrict__ out_pos) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int q = blockIdx.x * WPB + wq;
    if (q >= nq) return;
    uint64_t* ka = reinterpret_cast<uint64_t*>(smem) + (size_t)wq * 3 * CAPM;
    uint64_t* kb = ka + CAPM;
    uint64_t* pay = kb + CAPM;  // index of the hit in parts
    // A query whose hits all come from ONE shard (the rule with few coarse clusters: a V = 16 query visits one or two cells, and a
    // cell lives on one shard): that list arrives ranked, so it is copied -- no LDS, no sort.  The merge then costs what the number
    // of non-empty lists costs, not what the number of shards does.
    {
        int nonempty = 0, lone = 0, lone_n = 0;
        for (int w = 0; w < world; ++w) {
            const int v = arrived(cnt, off, stride, w, nq, q);
            if (v > 0) { ++nonempty; lone = w; lone_n = v; }
        }
        if (nonempty <= 1) {
            const int nv1 = lone_n < limit ? lone_n : limit;
            const int64_t base = nonempty ? (int64_t)lone * stride + off[(int64_t)lone * nq + q] : 0;
            const int64_t o1 = (int64_t)q * limit;
            for (int x = lane; x < limit; x += 64) {
                int64_t id = -1;
                double dist = __longlong_as_double(0x7ff8000000000000LL);
                int32_t cell = -1;
                uint32_t pos = 0xffffffffu;
                if (x < nv1) {
                    const cis_hit hh = parts[base + x];
                    id = hh.id; dist = hh.dist; cell = hh.cell; pos = hh.pos;
                }
                out_ids[o1 + x] = id;
                out_dists[o1 + x] = dist;
                if (out_cells) out_cells[o1 + x] = cell;
                if (out_pos) out_pos[o1 + x] = pos;
            }
            if (lane == 0 && out_n) out_n[q] = nv1;
            return;
        }
    }
    int have = 0, l = 0, e = 0, total = 0;
    while (true) {
        int n = have;
        int room = CAPM - have;
        while (l < world && room > 0) {
            const int valid = arrived(cnt, off, stride, l, nq, q);
            const int take = (valid - e < room) ? (valid - e) : room;
            const int64_t base = (int64_t)l * stride + off[(int64_t)l * nq + q] + e;
            for (int x = lane; x < take; x += 64) {
                const cis_hit hh = parts[base + x];
                ka[n + x] = (uint64_t)__double_as_longlong(hh.dist);
                kb[n + x] = ((uint64_t)hh.visit_rank << 32) | hh.pos;
                pay[n + x] = (uint64_t)(base + x);
            }
            n += take; total += take; room -= take; e += take;
            if (e >= valid) { ++l; e = 0; }
        }
        int ns = 64;
        while (ns < n) ns <<= 1;
        for (int x = n + lane; x < ns; x += 64) { ka[x] = ~0ull; kb[x] = ~0ull; pay[x] = ~0ull; }
        wave_lds_sync();
        // bitonic sort with payload
        for (int k = 2; k <= ns; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < (ns >> 1); t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int p = i + j;
                    const bool asc = ((i & k) == 0);
                    const uint64_t a0 = ka[i], b0 = kb[i], a1 = ka[p], b1 = kb[p];
                    const bool gt = (a0 > a1) || (a0 == a1 && b0 > b1);
                    if (gt == asc) {
                        ka[i] = a1; kb[i] = b1; ka[p] = a0; kb[p] = b0;
                        const uint64_t y = pay[i]; pay[i] = pay[p]; pay[p] = y;
                    }
                }
                wave_lds_sync();
            }
        }
        have = n < limit ? n : limit;
        if (l >= world) break;
    }
    const int nv = total < limit ? total : limit;
    const int64_t o = (int64_t)q * limit;
    for (int x = lane; x < limit; x += 64) {
        int64_t id = -1;
        double dist = __longlong_as_double(0x7ff8000000000000LL);
        int32_t cell = -1;
        uint32_t pos = 0xffffffffu;
        if (x < nv) {
            const cis_hit hh = parts[pay[x]];
            id = hh.id; dist = hh.dist; cell = hh.cell; pos = hh.pos;
        }
        out_ids[o + x] = id;
        out_dists[o + x] = dist;
        if (out_cells) out_cells[o + x] = cell;
        if (out_pos) out_pos[o + x] = pos;
    }
    if (lane == 0 && out_n) out_n[q] = nv;
}

// Any limit (above the 3072 records a wave ranks in LDS): every shard's list arrives ranked by (dist, visit_rank, pos), and the
// keys of different shards never tie (a cell lives on one shard), so a record's place in the merged ranking is its index in its
// own list plus, for every other list, the number of records with a smaller key -- binary searches, no sort.  One workgroup per
// query; records past `limit` are dropped, unused slots padded like every other route (-1 / NaN).
__global__ __launch_bounds__(256) void k_merge_packed_ranked(const cis_hit* __restrict__ parts, int world, int64_t stride,
                                                             const int64_t* __restrict__ off, const int32_t* __restrict__ cnt, int nq, int limit,
                                                             int64_t* __restrict__ out_ids, double* __restrict__ out_dists, int32_t* __restrict__ out_n,
                                                             int32_t* __restrict__ out_cells, uint32_t* __restrict__ out_pos) {
    const int q = blockIdx.x;
    __shared__ int s_tot;
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < world; ++w) t += arrived(cnt, off, stride, w, nq, q);
        s_tot = t;
    }
    __syncthreads();
    const int total = s_tot;
    const int64_t o = (int64_t)q * limit;
    auto less = [](const cis_hit& a, const cis_hit& b) -> bool {
        const uint64_t da = (uint64_t)__double_as_longlong(a.dist), db = (uint64_t)__double_as_longlong(b.dist);
        if (da != db) return da < db;  // non-negative doubles order like their bit patterns
        if (a.visit_rank != b.visit_rank) return a.visit_rank < b.visit_rank;
        return a.pos < b.pos;
    };
    for (int w = 0; w < world; ++w) {
        const cis_hit* lst = parts + (int64_t)w * stride + off[(int64_t)w * nq + q];
        const int n = arrived(cnt, off, stride, w, nq, q);
        for (int a = threadIdx.x; a < n; a += blockDim.x) {
            const cis_hit e = lst[a];
            int64_t rank = a;
            for (int w2 = 0; w2 < world && rank < limit; ++w2) {
                if (w2 == w) continue;
                const cis_hit* l2 = parts + (int64_t)w2 * stride + off[(int64_t)w2 * nq + q];
                int lo = 0, hi = arrived(cnt, off, stride, w2, nq, q);
                while (lo < hi) {  // records of list w2 with a smaller key
                    const int mid = (lo + hi) >> 1;
                    if (less(l2[mid], e)) lo = mid + 1;
                    else hi = mid;
                }
                rank += lo;
            }
            if (rank < limit) {
                out_ids[o + rank] = e.id;
                out_dists[o + rank] = e.dist;
                if (out_cells) out_cells[o + rank] = e.cell;
                if (out_pos) out_pos[o + rank] = e.pos;
            }
        }
    }
    const int nv = total < limit ? total : limit;
    for (int x = nv + threadIdx.x; x < limit; x += blockDim.x) {
        out_ids[o + x] = -1;
        out_dists[o + x] = __longlong_as_double(0x7ff8000000000000LL);
        if (out_cells) out_cells[o + x] = -1;
        if (out_pos) out_pos[o + x] = 0xffffffffu;
    }
    if (threadIdx.x == 0 && out_n) out_n[q] = nv;
}

// Offsets of the packed exchange on the device: cnt_all [world][nq] (what the counts all-gather delivered) -> off [world][nq] =
// exclusive scan of a shard's counts over the queries, totals[w], and *overflow = 1 when a shard holds more records than the fixed
// stride of the payload all-gather (the caller then repeats the exchange with the exact stride).  Replaces a torch.cumsum + a host
// read per batch (round 3).  One workgroup per shard.
__global__ __launch_bounds__(1024) void k_exchange_offsets(const int32_t* __restrict__ cnt_all, int nq, int64_t stride, int64_t* __restrict__ off,
                                                           int64_t* __restrict__ totals, int32_t* __restrict__ overflow) {
    __shared__ int64_t s_w[16];
    __shared__ int64_t s_run;
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int32_t* c = cnt_all + (int64_t)w * nq;
    int64_t* o = off + (int64_t)w * nq;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024) {
        const int q = q0 + tid;
        const int64_t v = q < nq ? (int64_t)c[q] : 0;
        int64_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wv] = x;
        __syncthreads();
        int64_t base = s_run;
        for (int k = 0; k < wv; ++k) base += s_w[k];
        if (q < nq) o[q] = base + x - v;
        __syncthreads();
        if (tid == 1023) s_run = base + x;
        __syncthreads();
    }
    if (tid == 0) {
        totals[w] = s_run;
        if (s_run > stride) atomicExch(overflow, 1);
    }
}

extern "C" int cis_exchange_offsets_dev(const int32_t* d_cnt_all, int world, int nq, int64_t stride, int64_t* d_off, int64_t* d_totals,
                                        int32_t* d_overflow, void* stream) {
    CIS_REQUIRE(world >= 1 && nq >= 0 && stride >= 0, "bad exchange arguments");
    CIS_REQUIRE(d_cnt_all && d_off && d_totals && d_overflow, "NULL buffer");
    CIS_TRY(cis_lazy_init());
    hipStream_t st = (hipStream_t)stream;
    CIS_CHECK_HIP(hipMemsetAsync(d_overflow, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(k_exchange_offsets, dim3((unsigned)world), dim3(1024), 0, st, d_cnt_all, nq, stride, d_off, d_totals, d_overflow);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

extern "C" int cis_merge_packed_dev(const cis_hit* d_parts, int world, int64_t stride, const int64_t* d_off,
                                    const int32_t* d_cnt, int nq, int limit, int64_t* d_ids, double* d_dists,
                                    int32_t* d_n_found, int32_t* d_cells, uint32_t* d_pos, void* stream) {
    CIS_REQUIRE(world >= 1 && nq >= 0 && limit >= 0 && limit <= MAX_LIMIT && stride >= 0, "bad merge arguments");
    CIS_REQUIRE(nq == 0 || limit == 0 || (d_parts && d_off && d_cnt && d_ids && d_dists), "NULL buffer");
    CIS_TRY(cis_lazy_init());
    if (nq == 0 || limit == 0) return CIS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g((unsigned)ceil_div(nq, 4));
    if (limit <= 128)
        hipLaunchKernelGGL((k_merge_packed<256, 4>), g, dim3(256), (size_t)4 * 3 * 256 * 8, st, d_parts, world, stride, d_off, d_cnt, nq, limit,
                           d_ids, d_dists, d_n_found, d_cells, d_pos);
    else if (limit <= 512)
        hipLaunchKernelGGL((k_merge_packed<1024, 4>), g, dim3(256), (size_t)4 * 3 * 1024 * 8, st, d_parts, world, stride, d_off, d_cnt, nq,
                           limit, d_ids, d_dists, d_n_found, d_cells, d_pos);
    else if (limit <= 3072)  // one wave per workgroup with 96 KB of LDS: 4096 keys per round, `limit` of them carried over
        hipLaunchKernelGGL((k_merge_packed<4096, 1>), dim3((unsigned)nq), dim3(64), (size_t)3 * 4096 * 8, st, d_parts, world, stride, d_off,
                           d_cnt, nq, limit, d_ids, d_dists, d_n_found, d_cells, d_pos);
    else  // any limit: places by binary search in the other shards' ranked lists
        hipLaunchKernelGGL(k_merge_packed_ranked, dim3((unsigned)nq), dim3(256), 0, st, d_parts, world, stride, d_off, d_cnt, nq, limit,
                           d_ids, d_dists, d_n_found, d_cells, d_pos);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

// ---- exact re-ranking with resident features (searcher_lopqhbase.py:864-912): true L2 distance of a query to the
// original features of its first `L` results.  One wave per (query, result); arithmetic in the feature dtype like
// np.linalg.norm(normed_feat - res_fts[pos]) (float32 features -> float32 distance), returned as float64.
template <typename T>
__global__ __launch_bounds__(256) void k_rerank(const T* __restrict__ feats, int64_t n_feats, int D, const T* __restrict__ Q,
                                                const int64_t* __restrict__ rows, int64_t n_pairs, int L,
                                                double* __restrict__ dists) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;
    const int64_t r = rows[pair];
    if (r < 0 || r >= n_feats) {  // feature not resident: the caller keeps the ADC distance (reference :889-893)
        if (lane == 0) dists[pair] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    const T* x = feats + r * D;
    const T* q = Q + (pair / L) * D;
    T acc = (T)0;
    for (int i = lane; i < D; i += 64) {
        const T df = q[i] - x[i];
        acc = fma(df, df, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = acc + __shfl_xor(acc, o);
    if (lane == 0) dists[pair] = (double)(T)sqrt(acc);
}

extern "C" int cis_rerank_dev(const void* d_feats, int f_dtype, int64_t n_feats, int D, const void* d_q, int nq,
                              const int64_t* d_rows, int L, double* d_dists, void* stream) {
    CIS_REQUIRE(f_dtype == CIS_F32 || f_dtype == CIS_F64, "f_dtype must be 4 or 8");
    CIS_REQUIRE(n_feats >= 0 && D > 0 && nq >= 0 && L >= 0, "bad re-ranking arguments");
    if (nq == 0 || L == 0) return CIS_OK;
    CIS_REQUIRE(d_feats && d_q && d_rows && d_dists, "NULL buffer");
    CIS_TRY(cis_lazy_init());
    const int64_t n_pairs = (int64_t)nq * L;
    const dim3 g((unsigned)ceil_div(n_pairs, 4));
    hipStream_t st = (hipStream_t)stream;
    if (f_dtype == CIS_F32)
        hipLaunchKernelGGL(k_rerank<float>, g, dim3(256), 0, st, (const float*)d_feats, n_feats, D, (const float*)d_q, d_rows, n_pairs, L, d_dists);
    else
        hipLaunchKernelGGL(k_rerank<double>, g, dim3(256), 0, st, (const double*)d_feats, n_feats, D, (const double*)d_q, d_rows, n_pairs, L, d_dists);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

extern "C" int cis_merge_hits_dev(const cis_hit* d_parts, int world, int nq, int limit, int64_t* d_ids,
                                  double* d_dists, int32_t* d_n_found, int32_t* d_cells, uint32_t* d_pos,
                                  void* stream) {
    CIS_REQUIRE(world >= 1 && nq >= 0 && limit >= 0 && limit <= MAX_LDS_LIMIT, "bad merge arguments (limit <= 3072)");
    CIS_TRY(cis_lazy_init());
    return merge_parts(d_parts, world, nq, limit, d_ids, d_dists, d_n_found, d_cells, d_pos, (hipStream_t)stream);
}

extern "C" int cis_index_search_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, int limit,
                                    int64_t* d_ids, double* d_dists, int32_t* d_n_found, int32_t* d_visited,
                                    int32_t* d_cells, uint32_t* d_pos, void* stream) {
    int L;
    CIS_TRY(effective_limit(quota, limit, &L));
    if (nq == 0) return CIS_OK;
    SearchOut o{};
    o.ids = d_ids; o.dists = d_dists; o.n_found = d_n_found; o.cells = d_cells; o.pos = d_pos; o.visited = d_visited;
    if (L == 0 && d_n_found) CIS_CHECK_HIP(hipMemsetAsync(d_n_found, 0, (size_t)nq * sizeof(int32_t), (hipStream_t)stream));
    return search_all(ix, dQ, q_dtype, nq, quota, L, o, (hipStream_t)stream);
}

// ---- host-pointer entry points: asynchronous form + pinned memory ------------------------------------------------------------
// The reference's callers hold their queries and want their results in HOST memory (searcher_lopqhbase.py:849-857).  Rounds 1-4 moved
// them with blocking hipMemcpy on the null stream around the search and a device-wide synchronisation: pageable copies are staged page
// by page, nothing overlapped, and a batch through this door ran at 31 % of the resident rate.  Now every handle owns a stream:
// cis_index_search_async enqueues copy-in, search and copy-out on it and returns as soon as the search's own launches are queued (the
// plan read-back in the middle of a large batch still waits ~0.1 ms); cis_index_search_wait blocks until the results have landed.
// With the buffers in pinned memory (cis_host_alloc) the copies are DMA transfers that overlap the searches of the other handles --
// views of one index (cis_index_create_view) give several batches in flight.
static std::mutex g_copy_mu;
static hipStream_t g_copy_stream[64] = {nullptr};
static int cis_copy_stream(int device, hipStream_t* out) {
    std::lock_guard<std::mutex> lk(g_copy_mu);
    const int d = device & 63;
    if (!g_copy_stream[d]) CIS_CHECK_HIP(hipStreamCreateWithFlags(&g_copy_stream[d], hipStreamNonBlocking));
    *out = g_copy_stream[d];
    return CIS_OK;
}

// Copy-outs ahead of their wait (round 6).  cis_index_search_wait used to enqueue its handle's copy-out and block for it: 13 MB of
// results of a C4 batch are 0.25 ms during which the calling thread launched nothing -- with the plan read-back of the next launch that
// made the host the bottleneck of the host-facing path (0.57 ms per step against 0.375 ms resident).  Now every call that holds the copy
// stream's lock looks at the OTHER handles with a search in flight: where the search has finished (hipEventQuery) the copy-out goes onto
// the copy stream there and then, and runs while the caller launches its own batch; the owner's wait finds it under way or landed.
static std::vector<cis_index*> g_host_pending;  // guarded by g_copy_mu: search enqueued, copy-out not yet

static hipError_t host_copy_out_locked(cis_index* ix, hipStream_t cp) {
    const cis_index::HostOut& o = ix->h_out;
    const int nq = o.nq, L = o.L;
    hipError_t e = hipSuccess;
    auto cpy = [&](void* dst, const void* src, size_t bytes) { if (e == hipSuccess && dst) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cp); };
    if (nq > 0) {
        if (L > 0) {
            cpy(o.ids, ix->w_oids.p, (size_t)nq * L * sizeof(int64_t));
            cpy(o.dists, ix->w_odists.p, (size_t)nq * L * sizeof(double));
            cpy(o.cells, ix->w_ocell.p, (size_t)nq * L * sizeof(int32_t));
            cpy(o.pos, ix->w_opos.p, (size_t)nq * L * sizeof(uint32_t));
        }
        cpy(o.n_found, ix->w_onf.p, (size_t)nq * sizeof(int32_t));
        cpy(o.visited, ix->w_ovis.p, (size_t)nq * sizeof(int32_t));
    }
    if (e == hipSuccess) e = hipEventRecord(ix->h_ev_done, cp);
    if (e == hipSuccess) ix->h_out_enqueued = true;
    return e;
}

static void host_pump_locked(int device, hipStream_t cp, const cis_index* self) {
    for (size_t i = 0; i < g_host_pending.size();) {
        cis_index* o = g_host_pending[i];
        if (o != self && o->m->device == device && hipEventQuery(o->h_ev_out) == hipSuccess && host_copy_out_locked(o, cp) == hipSuccess) {
            g_host_pending[i] = g_host_pending.back();
            g_host_pending.pop_back();
        } else {
            ++i;
        }
    }
    (void)hipGetLastError();  // (hipEventQuery's hipErrorNotReady is not an error of this call)
}

void cis_host_forget(cis_index* ix) {
    std::lock_guard<std::mutex> lk(g_copy_mu);
    g_host_pending.erase(std::remove(g_host_pending.begin(), g_host_pending.end(), ix), g_host_pending.end());
}

extern "C" int cis_host_alloc(void** out, size_t bytes) {
    CIS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CIS_TRY(cis_lazy_init());
    hipError_t e = hipHostMalloc(out, bytes > 0 ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        cis_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        *out = nullptr;
        return CIS_ENOMEM;
    }
    return CIS_OK;
}

extern "C" void cis_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int cis_index_search_wait(cis_index* ix);

extern "C" int cis_index_search_async(cis_index* ix, const void* Q, int q_dtype, int nq, int64_t quota, int limit,
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
    // (the copy stream is made before the first handle's stream: the GPU dispatches from four hardware pipes, streams are dealt onto them
    // in the order they first submit work, and two compute streams on one pipe do not overlap -- with the copy stream first, the fourth
    // handle's stream shares a pipe with it and not with the first handle's: tools/r06_queue_probe.py)
    hipStream_t cp = nullptr;
    CIS_TRY(cis_copy_stream(ix->m->device, &cp));
    if (!ix->h_stream) {
        CIS_CHECK_HIP(hipStreamCreateWithFlags(&ix->h_stream, hipStreamNonBlocking));
        CIS_CHECK_HIP(hipEventCreateWithFlags(&ix->h_ev_in, hipEventDisableTiming));
        CIS_CHECK_HIP(hipEventCreateWithFlags(&ix->h_ev_out, hipEventDisableTiming));
        CIS_CHECK_HIP(hipEventCreateWithFlags(&ix->h_ev_done, hipEventDisableTiming));
    }
    hipStream_t st = ix->h_stream;
    // Every copy of every handle goes through ONE copy stream per device: a copy-in and a copy-out that run at the same time collapse
    // on this platform (measured with pinned memory: 52-56 GB/s in either direction alone, 11 GB/s combined when both run --
    // profiles/archive/r05b/r05_pcie_probe.txt), so the copies are serialised among themselves and overlap only the searches.
    if (ix->h_pending) CIS_TRY(cis_index_search_wait(ix));  // one batch in flight per handle: its buffers are this handle's workspaces
    const size_t qbytes = (size_t)nq * ix->m->D_in * q_dtype;
    const int Lk = L > 0 ? L : 1;
    CIS_TRY(ix->w_q.reserve(qbytes));
    CIS_TRY(ix->w_oids.reserve((size_t)nq * Lk * sizeof(int64_t)));
    CIS_TRY(ix->w_odists.reserve((size_t)nq * Lk * sizeof(double)));
    CIS_TRY(ix->w_onf.reserve((size_t)nq * sizeof(int32_t)));
    CIS_TRY(ix->w_ovis.reserve((size_t)nq * sizeof(int32_t)));
    CIS_TRY(ix->w_ocell.reserve((size_t)nq * Lk * sizeof(int32_t)));
    CIS_TRY(ix->w_opos.reserve((size_t)nq * Lk * sizeof(uint32_t)));
    {
        std::lock_guard<std::mutex> lk(g_copy_mu);  // (enqueue order on the shared stream: a handle's copy and its event stay adjacent)
        CIS_CHECK_HIP(hipMemcpyAsync(ix->w_q.p, Q, qbytes, hipMemcpyHostToDevice, cp));
        CIS_CHECK_HIP(hipEventRecord(ix->h_ev_in, cp));
        // finished searches of the other handles: their results leave BEHIND this copy-in (4 MB against 13 MB: the launch below blocks
        // on this batch's plan read-back, which waits for the copy-in)
        host_pump_locked(ix->m->device, cp, ix);
    }
    // h_pending is set LAST, with this batch's h_out in place: an error exit in between must not leave the flag set over the previous
    // call's h_out (whose host buffers may be gone) -- cis_index_search_wait / cis_index_destroy would copy results into them.  Every
    // error exit below drains the stream (the copy-in may still be reading Q) and leaves the handle idle.
    struct Guard {
        cis_index* ix; hipStream_t st; bool armed;
        ~Guard() { if (armed) { (void)hipStreamSynchronize(st); ix->h_pending = false; ix->h_out = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}; } }
    } guard{ix, st, true};
    CIS_CHECK_HIP(hipStreamWaitEvent(st, ix->h_ev_in, 0));
    int rc = cis_index_search_dev(ix, ix->w_q.p, q_dtype, nq, quota, limit, ix->w_oids.as<int64_t>(),
                                  ix->w_odists.as<double>(), ix->w_onf.as<int32_t>(), ix->w_ovis.as<int32_t>(),
                                  ix->w_ocell.as<int32_t>(), ix->w_opos.as<uint32_t>(), st);
    if (rc != CIS_OK) return rc;
    CIS_CHECK_HIP(hipEventRecord(ix->h_ev_out, st));
    // the copy-out is enqueued by cis_index_search_wait, once the search has finished: enqueued here it would sit at the head of the
    // shared copy stream, waiting for the search, with every later copy-in of the other handles stuck behind it
    ix->h_out = {ids, dists, n_found, visited, cells, pos, nq, L};
    ix->h_out_enqueued = false;
    ix->h_pending = true;
    guard.armed = false;
    {
        // (no copy-outs from here: one enqueued now would be in the next call's copy-in's way -- that call's launch blocks on its plan
        // read-back, the read-back waits for the copy-in, and the copy stream is first in, first out)
        std::lock_guard<std::mutex> lk(g_copy_mu);
        g_host_pending.push_back(ix);
    }
    return CIS_OK;
}

extern "C" int cis_index_search_wait(cis_index* ix) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    if (ix->h_stream && ix->h_pending) {
        CIS_CHECK_HIP(hipSetDevice(ix->m->device));
        ix->h_pending = false;
        hipStream_t cp = nullptr;
        CIS_TRY(cis_copy_stream(ix->m->device, &cp));
        bool enq;
        {
            std::lock_guard<std::mutex> lk(g_copy_mu);
            enq = ix->h_out_enqueued;   // another handle's call may have put this handle's copy-out on the copy stream already
        }
        if (!enq) {
            CIS_CHECK_HIP(hipEventSynchronize(ix->h_ev_out));
            std::lock_guard<std::mutex> lk(g_copy_mu);
            if (!ix->h_out_enqueued) {
                g_host_pending.erase(std::remove(g_host_pending.begin(), g_host_pending.end(), ix), g_host_pending.end());
                CIS_CHECK_HIP(host_copy_out_locked(ix, cp));
            }
            host_pump_locked(ix->m->device, cp, ix);
        }
        CIS_CHECK_HIP(hipEventSynchronize(ix->h_ev_done));
    }
    return CIS_OK;
}

extern "C" int cis_index_search(cis_index* ix, const void* Q, int q_dtype, int nq, int64_t quota, int limit,
                                int64_t* ids, double* dists, int32_t* n_found, int32_t* visited, int32_t* cells,
                                uint32_t* pos) {
    CIS_TRY(cis_index_search_async(ix, Q, q_dtype, nq, quota, limit, ids, dists, n_found, visited, cells, pos));
    return cis_index_search_wait(ix);
}
