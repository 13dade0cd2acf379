// ADC scan v3: 16-bit fixed-point tables, four queries per workgroup, 4-byte region entries.
//
// Replaces the inner loop of lopq/lopq/search.py:137-177 (compute_distances: dist = sum_i T[i][fine_i]) for large
// query batches.  Same division of labour as the float32-prefilter kernel (k_adc_scan2, lopq_search.hip): the scan
// only REJECTS candidates that are provably outside the exact top `limit`; k_merge_survivors re-scores the survivors
// in float64 in the reference's order and ranks them by (dist, visit_rank, pos).  What changes is the arithmetic of
// the prefilter, because k_adc_scan2 is bound by LDS reads and VALU issue, not by HBM (profiles/r01i_*):
//
//  * The two half tables of a (query, cell) pair are quantised when they are staged in LDS:
//        q[j][k] = trunc(T32[j][k] * qinv),   qinv = cap / max(T32) * (1 - 2^-20),  cap = floor(65535 / M)
//    so that a candidate's sum s = sum_j q[j][code_j] fits 16 bits and brackets the exact float64 distance d:
//        s <= d * qinv * (1 + 2^-23)        and        d * qinv < (s + M) * (1 + 2^-22).
//    Integer sums have no accumulation error, so the bracket is tighter than float16 and costs half the LDS bytes
//    and half the adds of float32: entry (k, j) holds the values of FOUR queries in 8 bytes (tab[k][j][g] uint16),
//    one ds_read_b64 serves four queries and two v_pk_add_u16 add them.
//  * Every bound that leaves a wave is an exact-distance upper bound in float64 bits, as in k_adc_scan2 (wt/wl per
//    wave, `ext` from other cells of the query via qbound[]): "at least `limit` candidates seen so far do not exceed
//    it".  A bound B becomes the integer threshold thr = floor(B * qinv * (1 + 2^-22)) + 1; s > thr implies
//    d >= s / (qinv (1 + 2^-23)) > B: strictly worse than `limit` others.  A quantised value v that >= n entries do
//    not exceed becomes the published bound (v + M) * (1 + 2^-21) / qinv >= their exact distances.
//  * Region entries are (s << 16 | position in the chunk): 4 bytes, ordered by (s, pos) as integers; chunks are at
//    most 65536 candidates.  Four queries x four waves x 312 entries = 20 KB next to 16 KB of tables (M = 8): four
//    workgroups per CU at G = 4, which the float32 layout could not reach (72 KB).
//  * Crowds of equal sums (duplicate codes: the integer cut cannot thin them) fall back to the exact compaction:
//    codes fetched again, float64 distances summed left to right (search.py:173), cut on (dist, pos), one code of
//    the tie group remembered so that its later copies are skipped in the hot loop -- as in k_adc_scan2.
//
// Survivors leave as (float(s) << 32 | pos) 8-byte pairs, the format k_merge_survivors reads; their high words are
// only ordered inside one work item (each item has its own scale), so the merge re-scores all of them (cut = 0).
#include "scan_common.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) + __builtin_bit_cast(u16x2_t, b)));
}
// per 16-bit half: max(a - b, 0)
static __device__ __forceinline__ uint32_t pk_subsat_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}

// per 16-bit half: min(a, b)
static __device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}

static const int S3G = 4;  // queries per workgroup

#ifdef CIS_S3_COUNTERS
__device__ unsigned long long g_s3_ctr[16];  // cycle counters of wave 0 of every workgroup (tools/debug_counters3.py)
#define S3_CLK() ((long long)__builtin_amdgcn_s_memtime())
#define S3_CTR(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_s3_ctr[i], (unsigned long long)(v)); } while (0)
extern "C" int cis_debug_counters3(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s3_ctr), 16 * sizeof(unsigned long long)) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_s3_ctr), z, sizeof(z)) != hipSuccess) return -2;
    }
    return 0;
}
#else
#define S3_CLK() 0ll
#define S3_CTR(i, v) do { } while (0)
#endif

struct Scan3Shared {  // one per query handled by the workgroup
    uint64_t wt[8];   // per wave: exact-distance bound (float64 bits) that >= ceil(limit/NW) of its candidates do not exceed
    uint64_t wl[8];   // per wave: ... that >= limit of its candidates do not exceed
    uint64_t ext;     // bound published by workgroups that scanned OTHER cells/chunks of the query (qbound[q] at start)
    double inv_up;    // qinv * (1 + 2^-22): exact bound -> integer threshold
    double ub;        // (1 + 2^-21) / qinv: quantised value (+ M) -> exact-distance upper bound
    uint32_t thr;     // the hot loop's integer threshold (refreshed at every compaction; a lost concurrent update only
                      // leaves it looser for a while)
    int wcnt[8];      // survivors per wave at the end
    int pad;
};

static __device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NW>
static __device__ __forceinline__ uint64_t block_bound3(const Scan3Shared* sh) {
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

struct QScale { double inv_up, ub; };  // a query's scales (Scan3Shared), read where a compaction needs them
static __device__ __forceinline__ QScale load_scale(const Scan3Shared* sh) {
    QScale q;
    q.inv_up = sh->inv_up;
    q.ub = sh->ub;
    return q;
}

static __device__ __forceinline__ uint32_t bound_to_thr(uint64_t bound_bits, double inv_up) {
    if (bound_bits >= 0x7ff0000000000000ull) return 65534u;
    const double x = __longlong_as_double((long long)bound_bits) * inv_up;
    return x >= 65533.0 ? 65534u : (uint32_t)x + 1u;
}
static __device__ __forceinline__ uint64_t val_to_bound(uint32_t v, int M, double ub) {
    return (uint64_t)__double_as_longlong((double)(v + (uint32_t)M) * ub);  // ub carries the inflation (1 + 2^-21)
}

// In-loop compaction on the 16-bit sums only (wave-synchronous, nothing leaves the CU): see wave_compact_approx of
// k_adc_scan2 -- same two steps, integer margins.  Returns the new count, or -1 when a crowd of equal sums keeps
// more than `cap` entries (resolve exactly).
template <int M, int NR, int NW>
__device__ __forceinline__ int wave_compact3(uint32_t* rk, int cnt, int L, int Lw, int cap, const QScale& qs, Scan3Shared* sh, int w) {
    const int lane = threadIdx.x & 63;
    uint32_t ent[NR], s[NR];
    bool keep[NR];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * 64 + lane;
        keep[r] = e < cnt;
        ent[r] = keep[r] ? rk[e] : 0xffffffffu;
        s[r] = ent[r] >> 16;
        mn = (keep[r] && s[r] < mn) ? s[r] : mn;
        mx = (keep[r] && s[r] > mx) ? s[r] : mx;
    }
    wave_minmax_step<1>(mn, mx); wave_minmax_step<2>(mn, mx); wave_minmax_step<4>(mn, mx);
    wave_minmax_step<8>(mn, mx); wave_minmax_step<16>(mn, mx); wave_minmax_step<32>(mn, mx);
    mn = (uint32_t)__builtin_amdgcn_readfirstlane((int)mn);
    mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
    const uint64_t INF64 = 0x7ff0000000000000ull;
    // (1) v2 with #{s <= v2} in [Lw, Lw + W2] -> this wave's share of the block bound
    const int W2 = Lw >= 16 ? (Lw >> 3) : 1;
    uint32_t lo2 = mn, v2 = mx;
    while (cnt >= Lw && lo2 < v2) {  // wave-uniform
        const uint32_t p = lo2 + ((v2 - lo2) >> 1);
        int c = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) c += __popcll(__ballot(keep[r] && s[r] <= p));
        if (c >= Lw) {
            v2 = p;
            if (c <= Lw + W2) break;
        } else {
            lo2 = p + 1;
        }
    }
    const uint64_t boundW = cnt >= Lw ? val_to_bound(v2, M, qs.ub) : INF64;
    if (lane == 0) {
        if (boundW < lds_ld(&sh->wt[w])) lds_st(&sh->wt[w], boundW);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t thr = bound_to_thr(block_bound3<NW>(sh), qs.inv_up);
    uint32_t cut = thr;
    int c_thr = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) c_thr += __popcll(__ballot(keep[r] && s[r] <= cut));
    int W = cap - L;
    W = W > 24 ? 24 : (W < 0 ? 0 : W);
    if (c_thr > L + W) {
        // (2) the block bound does not thin this wave out: cut to the wave's own top L.  v with #{s <= v} in [L, L+W],
        // or the L-th smallest sum when ties prevent that; everything up to v + M stays (an entry above that is
        // strictly worse, in exact arithmetic, than the L entries that do not exceed v).
        uint32_t lo_ = mn, v = mx;
        while (lo_ < v) {
            const uint32_t p = lo_ + ((v - lo_) >> 1);
            int c = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) c += __popcll(__ballot(keep[r] && s[r] <= p));
            if (c >= L) {
                v = p;
                if (c <= L + W) break;
            } else {
                lo_ = p + 1;
            }
        }
        const uint32_t vm = v + (uint32_t)M;
        int c_keep = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) c_keep += __popcll(__ballot(keep[r] && s[r] <= vm));
        if (c_keep > cap) return -1;  // a crowd of (nearly) equal sums, e.g. duplicate codes: resolve exactly
        const uint64_t boundL = val_to_bound(v, M, qs.ub);
        if (lane == 0) {
            if (boundL < lds_ld(&sh->wl[w])) lds_st(&sh->wl[w], boundL);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        thr = bound_to_thr(block_bound3<NW>(sh), qs.inv_up);
        cut = vm < thr ? vm : thr;
    }
    if (lane == 0) {
        if (thr < lds_ld(&sh->thr)) lds_st(&sh->thr, thr);
    }
    int ncnt = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool kp = keep[r] && s[r] <= cut;
        const unsigned long long m = __ballot(kp);
        const int idx = ncnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (kp) rk[idx] = ent[r];
        ncnt += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ncnt;
}

// Exact compaction (crowds of equal sums): every entry is re-scored in float64 -- code from the index, table entries
// from global memory summed left to right (search.py:173) -- the cut is exact on (dist, pos), ties by region order
// (= position order: appends are in increasing position and the compactions are stable).  Survivors keep their
// 16-bit sums.  Returns the new count (<= L).  dup_pos: a member of a tie group the cut went through (its later
// copies lose against all L entries kept here), or 0xffffffff.
template <int M, int NR, int NW>
__device__ __forceinline__ int wave_compact3_exact(uint32_t* rk, int cnt, int L, int Lw, const QScale& qs, Scan3Shared* sh, int w,
                                                   const uint8_t* __restrict__ codes, int64_t start, int K,
                                                   const double* __restrict__ t0, const double* __restrict__ t1, uint32_t& dup_pos) {
    const int lane = threadIdx.x & 63;
    const uint64_t INF64 = 0x7ff0000000000000ull;
    uint32_t hi[NR], lo[NR], ent[NR];
    bool keep[NR];
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * 64 + lane;
        keep[r] = e < cnt;
        ent[r] = keep[r] ? rk[e] : 0xffffffffu;
        hi[r] = 0xffffffffu;
        lo[r] = 0xffffffffu;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (r * 64 >= cnt) break;  // wave-uniform
        const CodeWords<M> cw = load_code<M>(codes, start + (keep[r] ? (ent[r] & 0xffffu) : 0u));
        const uint64_t k = keep[r] ? (uint64_t)__double_as_longlong(adc64_words<M>(cw.w, K, t0, t1)) : ~0ull;
        hi[r] = (uint32_t)(k >> 32);
        lo[r] = (uint32_t)k;
        mn = (keep[r] && hi[r] < mn) ? hi[r] : mn;
        mx = (keep[r] && hi[r] > mx) ? hi[r] : mx;
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
            int seen = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool tie = keep[r] && hi[r] == vhi && lo[r] == vlo;
                const unsigned long long m = __ballot(tie);
                const int rank = seen + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                keep[r] = keep[r] && (hi[r] < vhi || (hi[r] == vhi && (lo[r] < vlo || (tie && rank < need))));
                if (seen == 0 && m) dup_pos = (uint32_t)__builtin_amdgcn_readlane((int)(ent[r] & 0xffffu), __ffsll((long long)m) - 1);
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
    const uint64_t bound = block_bound3<NW>(sh);
    if (lane == 0) {
        const uint32_t thr = bound_to_thr(bound, qs.inv_up);
        if (thr < lds_ld(&sh->thr)) lds_st(&sh->thr, thr);
    }
    int ncnt = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const uint64_t k = ((uint64_t)hi[r] << 32) | lo[r];
        const bool kp = keep[r] && (k <= bound);
        const unsigned long long m = __ballot(kp);
        const int idx = ncnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (kp) rk[idx] = ent[r];
        ncnt += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ncnt;
}

// Drop entries above the block bound (hot loop's test).  Stable.
template <int NR, int NW>
__device__ __forceinline__ int wave_filter3(uint32_t* rk, int cnt, const QScale& qs, const Scan3Shared* sh) {
    const int lane = threadIdx.x & 63;
    const uint32_t thr = bound_to_thr(block_bound3<NW>(sh), qs.inv_up);
    uint32_t k[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * 64 + lane;
        k[r] = e < cnt ? rk[e] : 0xffffffffu;
    }
    int ncnt = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const bool kp = (r * 64 + lane < cnt) && ((k[r] >> 16) <= thr);
        const unsigned long long m = __ballot(kp);
        const int idx = ncnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (kp) rk[idx] = k[r];
        ncnt += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ncnt;
}

// one candidate's code words in this lane's rotated order (make_rot): D[t >> 2] holds the byte used at step t
template <int M>
__device__ __forceinline__ void rot_words(const CodeWords<M>& c, const RotConsts<M>& rc, uint32_t (&D)[(M + 3) / 4]) {
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
}

// The pipeline unit of the hot loop: OCT table reads (ds_read_b64, four queries each) of one candidate -- steps
// [o*OCT, (o+1)*OCT) of the lane-rotated sub-quantizer order -- and their sum.  M = 16 is two units per candidate, so
// that the register footprint of the reads in flight is the same for every M.
template <int M, int OCT>
__device__ __forceinline__ void adc16_issue(const uint32_t (&D)[(M + 3) / 4], int o, const char* __restrict__ tab,
                                            const RotConsts<M>& rc, u32x2_t (&f)[OCT]) {
    constexpr int SH = ((M == 4) ? 2 : (M == 8 ? 3 : 4)) + 3;  // log2(M * 8 bytes): one k-row of the table
#pragma unroll
    for (int i = 0; i < OCT; ++i) {
        const int t = o * OCT + i;
        const uint32_t k = __builtin_amdgcn_ubfe(D[t >> 2], rc.sh[t & 3], 8);
        f[i] = *reinterpret_cast<const u32x2_t*>(tab + ((k << SH) | (rc.cj[t] << 1)));
    }
}

// two independent chains (queries 0-1 and 2-3), interleaved: no back-to-back dependent packed adds
template <int OCT>
__device__ __forceinline__ u32x2_t adc16_sum(const u32x2_t (&f)[OCT]) {
    uint32_t a0 = f[0][0], a1 = f[0][1];
#pragma unroll
    for (int i = 1; i < OCT; ++i) {
        a0 = pk_add_u16(a0, f[i][0]);
        a1 = pk_add_u16(a1, f[i][1]);
    }
    u32x2_t r;
    r[0] = a0; r[1] = a1;
    return r;
}

static __device__ __forceinline__ uint32_t ld16(const uint16_t* p) { return *reinterpret_cast<const volatile uint16_t*>(p); }
static __device__ __forceinline__ void st16(uint16_t* p, uint32_t v) { *reinterpret_cast<volatile uint16_t*>(p) = (uint16_t)v; }

// One workgroup (NW waves) scans one cell chunk (<= 65536 candidates) for `ng` <= 4 queries that all visit it.
template <int M, int NR, int U, int NW>
__device__ __forceinline__ void scan3_group(const WorkItem* __restrict__ items, const TabDesc* __restrict__ tabs,
                                            const int (&item_idx)[S3G], int ng,
                                            const double* __restrict__ T, const float* __restrict__ T32,
                                            const uint8_t* __restrict__ codes, int K, int L, int S,
                                            uint64_t* __restrict__ item_surv, int* __restrict__ item_n, float* __restrict__ item_slack,
                                            unsigned long long* __restrict__ qbound, char* smem, bool two_pass) {
    constexpr int G = S3G;
    constexpr int R = NR * 64 - 8;
    constexpr int nf = M / 2;
    constexpr int OCT = M >= 8 ? 8 : 4, NU = M / OCT;
    constexpr uint32_t CAP = 65535u / M;
    char* tab = smem;                                                              // [K][M][G] uint16
    uint32_t* rk_all = reinterpret_cast<uint32_t*>(smem + (size_t)K * M * G * 2);  // [G][NW][R] (s << 16 | pos)
    Scan3Shared* sh = reinterpret_cast<Scan3Shared*>(rk_all + G * NW * R);         // [G]
    uint16_t* thr1 = reinterpret_cast<uint16_t*>(sh + G);                          // [G] hot-loop thresholds + 1, packed like the sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long c0 = S3_CLK();
    (void)c0;
    // item_idx[] is wave-uniform (scalar registers); item fields are (re)loaded through scalar loads where they are used
    const WorkItem it0 = items[item_idx[0]];
    {
        int tab0[G], tab1[G];
        float qinv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            tab0[g] = items[item_idx[g]].tab0; tab1[g] = items[item_idx[g]].tab1;
            // largest entry of the query's two half tables (k_tables_from_px leaves it in TabDesc::pad, float32 bits)
            float mxT = fmaxf(__int_as_float(tabs[tab0[g]].pad), __int_as_float(tabs[tab1[g]].pad));
            mxT = fmaxf(mxT, 1e-30f);
            // qinv = cap / max * (1 - 2^-20): T32 * qinv, rounded, stays below cap (no clamping, so the bracket holds);
            // a table that holds inf / NaN gets qinv = 0 (all sums 0: everything survives to the exact re-scoring)
            float qi = ((float)CAP / mxT) * (1.0f - 9.5367431640625e-7f);
            qi = (mxT < 3.0e38f) ? qi : 0.0f;
            qi = (qi < 3.0e38f) ? qi : 3.0e38f;
            qinv[g] = qi;
        }
        if (tid < G) {
            const int g = tid;
            float qi = qinv[0];
#pragma unroll
            for (int gg = 1; gg < G; ++gg) qi = (g == gg) ? qinv[gg] : qi;
            const double inv_up = (double)qi * (1.0 + 2.384185791015625e-7);
            const double ub = qi > 0.0f ? (1.0 + 4.76837158203125e-7) / (double)qi : __longlong_as_double(0x7ff0000000000000LL);
            sh[g].inv_up = inv_up;
            sh[g].ub = ub;
            // A distance that >= L candidates of this query in already scanned cells do not exceed (see k_adc_scan2)
            const unsigned long long e = (g < ng) ? __hip_atomic_load(&qbound[items[item_idx[g]].q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                  : 0x7ff0000000000000ull;
            sh[g].ext = e;
            // an absent query: threshold 0 (thr1 = 0: the saturated difference is never non-zero) and maximal table entries
            uint32_t thr = (g < ng) ? bound_to_thr(e, inv_up) : 0u;
#ifdef CIS_S3_PROBE_NOPASS
            thr = 0u;  // probe: nothing passes, only the fixed-point scan runs
            if (true) { sh[g].thr = 0u; thr1[g] = 0; } else
#endif
            { sh[g].thr = thr; thr1[g] = (uint16_t)((g < ng) ? thr + 1u : 0u); }
            // what the merge may assume about a survivor: exact distance >= (its float32 upper bound) - slack
            if (g < ng) {
                item_slack[2 * (int64_t)item_idx[g] + 0] = __double2float_ru(ub * ((double)M + 0.1));
                item_slack[2 * (int64_t)item_idx[g] + 1] = __double2float_ru(ub);  // the merge turns a survivor's sum s into the bound (s + M) * ub
            }
        }
        if (tid < 8 * G) {
            const int g = tid >> 3, i = tid & 7;
            sh[g].wt[i] = 0x7ff0000000000000ull; sh[g].wl[i] = 0x7ff0000000000000ull;
            sh[g].wcnt[i] = 0;
        }
        // Tables: float32 copies ([nf][K] per (query, half)) -> 16-bit entries -> LDS, one 8-byte store per (k, j) that
        // carries the four queries' values; NW*64 float4 per half and round, eight 16-byte loads in flight per thread.
        const int nvec = (nf * K) >> 2;  // float4 per half table; K is a multiple of 4
        uint32_t* tw = reinterpret_cast<uint32_t*>(tab);
#pragma unroll 1
        for (int e = tid; e < nvec; e += NW * 64) {
            float4 v[G][2];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v[g][0] = tab_f4(T32, T, tab0[g], nf * K, e);
                v[g][1] = tab_f4(T32, T, tab1[g], nf * K, e);
            }
            const int j = (4 * e) / K, k0 = 4 * e - j * K;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    uint32_t qv[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float4 q = v[g][s2];
                        const float x = c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w));
                        qv[g] = (g < ng) ? (uint32_t)(x * qinv[g]) : CAP;  // truncation: a lower bound of x * qinv
                        qv[g] = qv[g] > CAP ? CAP : qv[g];                 // (NaN / garbage guard; never taken for finite tables)
                    }
                    u32x2_t pk;
                    pk[0] = qv[0] | (qv[1] << 16);
                    pk[1] = qv[2] | (qv[3] << 16);
                    *reinterpret_cast<u32x2_t*>(tw + (((k0 + c) * M + s2 * nf + j) << 1)) = pk;
                }
            }
        }
    }
    S3_CTR(1, S3_CLK() - c0);
    __syncthreads();
    S3_CTR(3, S3_CLK() - c0);
    const RotConsts<M> rc = make_rot<M>(lane);
    const int Lw = (L + NW - 1) / NW;
    const int len = __builtin_amdgcn_readfirstlane(it0.len);
    const int64_t start = ((int64_t)__builtin_amdgcn_readfirstlane((int)(it0.start >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)it0.start);
#ifdef CIS_S3_PROBE_NOLOOP
    const int nit = 0;  // probe: prologue + epilogue only
#else
    const int nit = (len + 64 * U - 1) / (64 * U);
#endif
    int cnt[G];
    CodeWords<M> dup[G];  // per query: a code whose later copies cannot enter this wave's top-L any more
    bool has_dup[G];
    bool any_dup = false;
#pragma unroll
    for (int g = 0; g < G; ++g) { cnt[g] = 0; has_dup[g] = false; dup[g] = CodeWords<M>(); }
    // buffer descriptor over this chunk's codes (wave-uniform): 32-bit offsets, positions past the end read zero
    __amdgpu_buffer_rsrc_t rs;
    {
        const uint64_t cbase = (uint64_t)(uintptr_t)(codes + start * M);
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cbase);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cbase >> 32));
        const int nbytes = __builtin_amdgcn_readfirstlane(len * M);
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)bhi << 32) | blo), 0, nbytes, 0x00020000);
    }
    // ---- two-pass mode (short chunks): no running bound, no regions, no selections --------------------------------------
    // Pass 1 adds every candidate's sum into a 512-bin histogram per query (LDS atomics; the histograms use the memory of the
    // regions).  One wave per query then finds the first bin whose cumulative count reaches L: every candidate up to that bin's
    // upper edge is kept -- at least L of them, so the edge is a valid bound (published for the query's other cells), and at
    // most L plus one bin's population.  Pass 2 computes the sums again and appends what is under the fixed threshold straight
    // to the work item's survivor list (one LDS atomic per survivor).  A chunk of a few thousand candidates spends its time in
    // the per-wave selections of the streaming form (one when a region first fills, one at the end, per wave and query); here
    // it pays the table gathers twice instead.  A crowd -- more candidates under the edge than the list holds (thousands of
    // equal sums: duplicate codes) -- falls through to the streaming form below, which resolves ties exactly.
    if (two_pass) {
        constexpr int NB = 512, BSH = 7;  // bins of 128 sums
        static_assert(G * NB <= G * NW * R, "the histograms live in the region memory");
        uint32_t* hist = rk_all;          // [G][NB]
        int* s_flag = reinterpret_cast<int*>(thr1 + 4);  // crowd flag (the 8 bytes after the packed thresholds)
        for (int e = tid; e < G * NB; e += NW * 64) hist[e] = 0u;
        if (tid == 0) *s_flag = 0;
        __syncthreads();
        auto sums_of = [&](const CodeWords<M> (&cur)[U], u32x2_t (&d)[U]) {
            u32x2_t fbuf[2][OCT];
            uint32_t D[2][(M + 3) / 4];
            rot_words<M>(cur[0], rc, D[0]);
            adc16_issue<M, OCT>(D[0], 0, tab, rc, fbuf[0]);
#pragma unroll
            for (int q = 0; q < U * NU; ++q) {
                const int u = q / NU, o = q % NU;
                if (q + 1 < U * NU) {
                    const int u1 = (q + 1) / NU, o1 = (q + 1) % NU;
                    if (o1 == 0) rot_words<M>(cur[u1], rc, D[u1 & 1]);
                    adc16_issue<M, OCT>(D[u1 & 1], o1, tab, rc, fbuf[(q + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x2_t part = adc16_sum<OCT>(fbuf[q & 1]);
                if (o == 0) {
                    d[u] = part;
                } else {
                    d[u][0] = pk_add_u16(d[u][0], part[0]);
                    d[u][1] = pk_add_u16(d[u][1], part[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {   // pass 1
            CodeWords<M> nx[U];
            if (w < nit) {
#pragma unroll
                for (int u = 0; u < U; ++u) nx[u] = load_code_buf<M>(rs, w * 64 * U + u * 64 + lane);
            }
            for (int iter = w; iter < nit; iter += NW) {
                const int base = iter * 64 * U;
                CodeWords<M> cur[U];
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nx[u];
                if (iter + NW < nit) {
#pragma unroll
                    for (int u = 0; u < U; ++u) nx[u] = load_code_buf<M>(rs, (iter + NW) * 64 * U + u * 64 + lane);
                }
                u32x2_t d[U];
                sums_of(cur, d);
                const bool tail = base + 64 * U > len;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!tail || base + u * 64 + lane < len) {
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const uint32_t sg = (g & 1) ? (d[u][g >> 1] >> 16) : (d[u][g >> 1] & 0xffffu);
                            atomicAdd(&hist[g * NB + (sg >> BSH)], 1u);
                        }
                    }
                }
            }
        }
        S3_CTR(11, S3_CLK() - c0);
        __syncthreads();
        S3_CTR(12, S3_CLK() - c0);
        // thresholds: wave w serves the queries g = w, w + NW, ...
        for (int g = w; g < G; g += NW) {
            uint32_t c[8];
            int own = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { c[i] = hist[g * NB + lane * 8 + i]; own += (int)c[i]; }
            int x = own;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                const int y = __shfl_up(x, dd);
                if (lane >= dd) x += y;
            }
            const unsigned long long reach = __ballot(x >= L);
            uint32_t thr = 65534u;
            int n_le = __builtin_amdgcn_readlane(x, 63);  // all candidates
            bool have_bound = false;
            if (g < ng && reach != 0ull) {
                const int fl = __ffsll((long long)reach) - 1;
                int cum = x - own, bin = lane * 8, cnt_le = 0;
                bool hit = false;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    cum += (int)c[i];
                    if (!hit && cum >= L) { hit = true; bin = lane * 8 + i; cnt_le = cum; }
                }
                const int bsel = __builtin_amdgcn_readlane(bin, fl);
                n_le = __builtin_amdgcn_readlane(cnt_le, fl);
                const uint32_t edge = ((uint32_t)(bsel + 1) << BSH) - 1u;
                thr = edge < 65534u ? edge : 65534u;
                have_bound = true;
            }
            if (lane == 0) {
                // L candidates have a sum <= thr, i.e. an exact distance below (thr + M)(1 + eps) / qinv -- and a candidate with an
                // exact distance below THAT can carry any sum up to thr + M + 1 (a sum brackets its distance only to within M
                // units: s <= d qinv (1 + 2^-23), d qinv < (s + M)(1 + 2^-22)).  So the collection threshold is thr + M + 1, not
                // thr: with the edge itself the last ranks of a query could go to a candidate a few units further out (found by
                // the parity fuzz: one query in ~20 k, rank `limit` only).  M + 1 < one bin: the count under the collection
                // threshold is at most the next bin's population more.
                uint32_t keep = thr, n_keep = (uint32_t)n_le;
                if (have_bound && thr < 65534u) {
                    keep = thr + (uint32_t)M + 1u;
                    keep = keep < 65534u ? keep : 65534u;
                    const uint32_t nb = (thr + 1u) >> BSH;  // the bin right after the edge
                    if (nb < (uint32_t)NB) n_keep += hist[g * NB + nb];
                }
#ifdef CIS_S3_OLD_EDGE  // test-sensitivity builds only: the collection threshold before the fix
                keep = thr;
#endif
                const uint32_t t_ext = lds_ld(&sh[g].thr);  // from the query's other cells (or 0 for an absent query)
                const uint32_t t = keep < t_ext ? keep : t_ext;
                sh[g].thr = t;
                thr1[g] = (uint16_t)((g < ng) ? t + 1u : 0u);
                sh[g].wcnt[0] = 0;
                if (g < ng && n_keep > (uint32_t)R) *s_flag = 1;  // possibly more survivors than ONE wave's region holds: streaming form
                // at least L candidates of this chunk have a sum <= thr: an exact-distance bound for the whole query
                sh[g].wt[0] = have_bound ? val_to_bound(thr, M, sh[g].ub) : 0x7ff0000000000000ull;
            }
        }
        __syncthreads();
        S3_CTR(13, S3_CLK() - c0);
        if (*s_flag == 0) {
            if (tid < G && tid < ng) {
                const uint64_t b = sh[tid].wt[0];
                if (b < sh[tid].ext) atomicMin(&qbound[items[item_idx[tid]].q], (unsigned long long)b);
            }
            // pass 2: the histograms are dead, their memory is the wave-private regions again.  At most R candidates of a query
            // are under its threshold (checked above), so no region can overflow and nothing is ever compacted.
            const u32x2_t tpk = *reinterpret_cast<const volatile u32x2_t*>(thr1);
            const uint32_t s01 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tpk[0]);
            const uint32_t s23 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tpk[1]);
            int cnt2[G];
#pragma unroll
            for (int g = 0; g < G; ++g) cnt2[g] = 0;
            CodeWords<M> nx[U];
            if (w < nit) {
#pragma unroll
                for (int u = 0; u < U; ++u) nx[u] = load_code_buf<M>(rs, w * 64 * U + u * 64 + lane);
            }
            for (int iter = w; iter < nit; iter += NW) {
                const int base = iter * 64 * U;
                CodeWords<M> cur[U];
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nx[u];
                if (iter + NW < nit) {
#pragma unroll
                    for (int u = 0; u < U; ++u) nx[u] = load_code_buf<M>(rs, (iter + NW) * 64 * U + u * 64 + lane);
                }
                u32x2_t d[U];
                sums_of(cur, d);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t x = pk_subsat_u16(tpk[0], d[u][0]) | pk_subsat_u16(tpk[1], d[u][1]);
                    unsigned long long am = __ballot(x != 0u);
                    const int n = len - base - u * 64;
                    if (n < 64) am &= n <= 0 ? 0ull : ((1ull << n) - 1ull);
                    if (am == 0ull) continue;  // scalar branch
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const uint32_t sg = (g & 1) ? (d[u][g >> 1] >> 16) : (d[u][g >> 1] & 0xffffu);
                        const uint32_t t1 = (g & 1) ? ((g >> 1) ? s23 >> 16 : s01 >> 16) : ((g >> 1) ? s23 & 0xffffu : s01 & 0xffffu);
                        const unsigned long long m = __ballot(sg < t1) & am;
                        if (m == 0ull) continue;
                        uint32_t* rk = rk_all + (g * NW + w) * R;
                        const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, cnt2[g]));
                        if (((m >> lane) & 1ull) && idx < R) rk[idx] = (sg << 16) | (uint32_t)(base + u * 64 + lane);
                        cnt2[g] += __popcll(m);
                    }
                }
            }
            S3_CTR(14, S3_CLK() - c0);
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (lane == 0 && g < ng) sh[g].wcnt[w] = cnt2[g] < R ? cnt2[g] : R;
            __syncthreads();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g >= ng) break;
                const uint32_t* rk = rk_all + (g * NW + w) * R;
                int off = 0, total = 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const int c = sh[g].wcnt[i];
                    off += (i < w) ? c : 0;
                    total += c;
                }
                const int mine = cnt2[g] < R ? cnt2[g] : R;
                uint64_t* out = item_surv + (int64_t)item_idx[g] * S + off;
                for (int e = lane; e < mine; e += 64) {
                    const uint32_t x = rk[e];
                    out[e] = ((uint64_t)(x >> 16) << 32) | (x & 0xffffu);
                }
                if (tid == 0) item_n[item_idx[g]] = total;
            }
            S3_CTR(7, S3_CLK() - c0);
            S3_CTR(0, 1);
            return;
        }
        // crowd: the streaming form starts from the bounds of the query's other cells again
        if (tid < G) {
            const int g = tid;
            const uint32_t t = (g < ng) ? bound_to_thr(sh[g].ext, sh[g].inv_up) : 0u;
            sh[g].thr = t;
            thr1[g] = (uint16_t)((g < ng) ? t + 1u : 0u);
            sh[g].wt[0] = 0x7ff0000000000000ull;
            sh[g].wcnt[0] = 0;
        }
        __syncthreads();
    }
    CodeWords<M> nxt[U];
    if (w < nit) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = load_code_buf<M>(rs, w * 64 * U + u * 64 + lane);
    }
    for (int iter = w; iter < nit; iter += NW) {
        const int base = iter * 64 * U;
        CodeWords<M> cur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        if (iter + NW < nit) {  // software prefetch: the next iteration's codes are in flight while this one computes
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = load_code_buf<M>(rs, (iter + NW) * 64 * U + u * 64 + lane);
        }
        u32x2_t d[U];
        {
            // U * NU units; the reads of unit q+1 are issued before the adds of unit q
            u32x2_t fbuf[2][OCT];
            uint32_t D[2][(M + 3) / 4];
            rot_words<M>(cur[0], rc, D[0]);
            adc16_issue<M, OCT>(D[0], 0, tab, rc, fbuf[0]);
#pragma unroll
            for (int q = 0; q < U * NU; ++q) {
                const int u = q / NU, o = q % NU;
                if (q + 1 < U * NU) {
                    const int u1 = (q + 1) / NU, o1 = (q + 1) % NU;
                    if (o1 == 0) rot_words<M>(cur[u1], rc, D[u1 & 1]);
                    adc16_issue<M, OCT>(D[u1 & 1], o1, tab, rc, fbuf[(q + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x2_t part = adc16_sum<OCT>(fbuf[q & 1]);
                if (o == 0) {
                    d[u] = part;
                } else {
                    d[u][0] = pk_add_u16(d[u][0], part[0]);
                    d[u][1] = pk_add_u16(d[u][1], part[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // thresholds + 1 of the four queries, packed like the sums: (thr + 1) - s, saturated at 0, is non-zero iff s <= thr
        const u32x2_t tpk = *reinterpret_cast<const volatile u32x2_t*>(thr1);
        unsigned long long anym[U];
        unsigned long long any = 0ull;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t x = pk_subsat_u16(tpk[0], d[u][0]) | pk_subsat_u16(tpk[1], d[u][1]);
            anym[u] = __ballot(x != 0u);
        }
        if (base + 64 * U > len) {  // last iteration: lanes past the end of the chunk (wave-uniform masks)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = len - base - u * 64;
                anym[u] &= n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) any |= anym[u];
        if (any == 0ull) continue;  // nothing in these 64*U candidates beats a bound
        const uint32_t s01 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tpk[0]);
        const uint32_t s23 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tpk[1]);
        unsigned long long pm[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (anym[u] == 0ull) {  // scalar branch: passes are sparse, most rows of an iteration have none
#pragma unroll
                for (int g = 0; g < G; ++g) pm[u][g] = 0ull;
                continue;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t sg = (g & 1) ? (d[u][g >> 1] >> 16) : (d[u][g >> 1] & 0xffffu);
                const uint32_t t1 = (g & 1) ? ((g >> 1) ? s23 >> 16 : s01 >> 16) : ((g >> 1) ? s23 & 0xffffu : s01 & 0xffffu);
                pm[u][g] = __ballot(sg < t1) & anym[u];
                if (any_dup) {  // wave-uniform, rare: later copies of a code that already lost a tie-break
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < (M + 3) / 4; ++i) same = same && (cur[u].w[i] == dup[g].w[i]);
                    pm[u][g] &= ~(__ballot(same) & (has_dup[g] ? ~0ull : 0ull));
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g >= ng) break;
            uint32_t* rk = rk_all + (g * NW + w) * R;
            int ntot = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) ntot += __popcll(pm[u][g]);
            if (ntot == 0) continue;
            auto entry = [&](int u) -> uint32_t {
                const uint32_t sg = (g & 1) ? (d[u][g >> 1] >> 16) : (d[u][g >> 1] & 0xffffu);
                return (sg << 16) | (uint32_t)(base + u * 64 + lane);
            };
            if (cnt[g] + ntot <= R) {  // everything fits: only the rows that have a pass touch the region
                int c = cnt[g];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned long long m = pm[u][g];
                    if (m == 0ull) continue;
                    const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, c));
                    if ((m >> lane) & 1ull) rk[idx] = entry(u);
                    c += __popcll(m);
                }
                cnt[g] = c;
                continue;
            }
            // Append row after row while they fit; when one does not, compact, re-test the rows not yet appended
            // against the new threshold and go on (after a compaction cnt <= R - 64, so the next row always fits).
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
                    if ((m >> lane) & 1ull) rk[idx] = entry(u);
                    cnt[g] += n;
                    u0 = u + 1;
                }
                if (!full) break;
                const QScale qsg = load_scale(&sh[g]);
                int c2 = wave_compact3<M, NR, NW>(rk, cnt[g], L, Lw, R - 64, qsg, &sh[g], w);
                if (c2 < 0) {
                    uint32_t dp = 0xffffffffu;
                    const WorkItem itg = items[item_idx[g]];
                    c2 = wave_compact3_exact<M, NR, NW>(rk, cnt[g], L, Lw, qsg, &sh[g], w, codes, start, K,
                                                        T + (int64_t)itg.tab0 * nf * K, T + (int64_t)itg.tab1 * nf * K, dp);
                    dp = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp);
                    if (dp != 0xffffffffu) {
                        dup[g] = load_code<M>(codes, start + (int64_t)dp);
                        has_dup[g] = true;
                        any_dup = true;
                    }
                }
                cnt[g] = c2;
                const uint32_t thg = lds_ld(&sh[g].thr);
                if (lane == 0) st16(&thr1[g], thg + 1u);  // the hot loop's copy (a stale overwrite only leaves it looser for a while)
                const unsigned long long dup_on = has_dup[g] ? ~0ull : 0ull;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t sg = (g & 1) ? (d[u][g >> 1] >> 16) : (d[u][g >> 1] & 0xffffu);
                    bool same = true;
#pragma unroll
                    for (int i = 0; i < (M + 3) / 4; ++i) same = same && (cur[u].w[i] == dup[g].w[i]);
                    pm[u][g] &= __ballot(sg <= thg) & ~(__ballot(same) & dup_on);
                }
            }
        }
    }
    S3_CTR(4, S3_CLK() - c0);
    // End of the chunk: every wave cuts its region and publishes its bounds; after the barrier the block bound is
    // (about) the L-th smallest distance of the whole chunk, so only ~L/NW entries per wave survive it.
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g >= ng) break;
        uint32_t* rk = rk_all + (g * NW + w) * R;
        // cap = NR*64 >= cnt: the cut never reports a crowd here (ties beyond L + M stay; the merge resolves them exactly)
        if (cnt[g] > 0) cnt[g] = wave_compact3<M, NR, NW>(rk, cnt[g], L, Lw, NR * 64, load_scale(&sh[g]), &sh[g], w);
    }
    S3_CTR(5, S3_CLK() - c0);
    if constexpr (NW > 1) __syncthreads();
    S3_CTR(6, S3_CLK() - c0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g >= ng) break;
        uint32_t* rk = rk_all + (g * NW + w) * R;
        const QScale qsg = load_scale(&sh[g]);
        if constexpr (NW > 1) cnt[g] = wave_filter3<NR, NW>(rk, cnt[g], qsg, &sh[g]);  // one wave: its own cut was the block bound
        if (lane == 0) sh[g].wcnt[w] = cnt[g];
        if (tid == 0) {
            const uint64_t b = block_bound3<NW>(&sh[g]);
            if (b < sh[g].ext) atomicMin(&qbound[items[item_idx[g]].q], (unsigned long long)b);
        }
    }
    if constexpr (NW > 1) __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g >= ng) break;
        const uint32_t* rk = rk_all + (g * NW + w) * R;
        int off = 0, total = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = (NW > 1) ? sh[g].wcnt[i] : cnt[g];
            off += (i < w) ? c : 0;
            total += c;
        }
        // a survivor leaves as (16-bit sum << 32 | position); the merge turns the sum into an upper bound of the exact distance
        uint64_t* out = item_surv + (int64_t)item_idx[g] * S + off;  // S = NW * R >= total
        for (int e = lane; e < cnt[g]; e += 64) {
            const uint32_t x = rk[e];
            out[e] = ((uint64_t)(x >> 16) << 32) | (x & 0xffffu);
        }
        if (tid == 0) item_n[item_idx[g]] = total;
    }
    S3_CTR(7, S3_CLK() - c0);
    S3_CTR(0, 1);
}

// Persistent launch over the slot queues, as k_adc_scan2: (workgroups per CU) x 256 workgroups pull slots (<= 4 work
// items of one cell chunk) from eight queues, one per XCD; a workgroup whose own queue is empty steals.
template <int M, int NR, int U, int NW, int WPE>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_adc_scan3(
    const WorkItem* __restrict__ items, const TabDesc* __restrict__ tabs, const int* __restrict__ slots,
    const int* __restrict__ n_slots_ptr, const double* __restrict__ T, const float* __restrict__ T32,
    const uint8_t* __restrict__ codes, int K, int L, int S, int* __restrict__ queue_ctr /* [8], zeroed */,
    uint64_t* __restrict__ item_surv, int* __restrict__ item_n, float* __restrict__ item_slack,
    unsigned long long* __restrict__ qbound /* [nq], +inf */, int two_pass, int single_queue /* queue 0 holds everything: [qs[0], qs[1]) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int G = S3G;
    constexpr int R = NR * 64 - 8;
    int* s_next = reinterpret_cast<int*>(smem + (size_t)K * M * G * 2 + (size_t)G * NW * R * 4 + G * sizeof(Scan3Shared) + 32);
    const int* qs = n_slots_ptr + 8;  // [9] queue starts, written by the slot builder
    const int home = blockIdx.x & 7;
    const long long k0 = S3_CLK();
    (void)k0;
    for (int a = 0; a < 8; ++a) {
        const int x = (home + a) & 7;
        const int qstart = qs[x];
        const int count = single_queue ? (x == 0 ? qs[1] - qstart : 0) : qs[x + 1] - qstart;
        if (count <= 0) continue;  // an empty queue costs no atomic and no barrier (the fall-back launch usually finds nothing at all)
        while (true) {
            const long long q0 = S3_CLK();
            (void)q0;
            int j;
            if constexpr (NW > 1) {
                __syncthreads();  // previous slot fully written out; LDS may be reused
                if (threadIdx.x == 0) *s_next = atomicAdd(&queue_ctr[x], 1);
                __syncthreads();
                j = __builtin_amdgcn_readfirstlane(*s_next);  // wave-uniform: everything derived from it stays scalar
            } else {
                int t = 0;
                if (threadIdx.x == 0) t = atomicAdd(&queue_ctr[x], 1);
                j = __builtin_amdgcn_readfirstlane(t);
            }
            S3_CTR(8, S3_CLK() - q0);
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
                    scan3_group<M, NR, U, NW>(items, tabs, oi, 1, T, T32, codes, K, L, S, item_surv, item_n, item_slack, qbound, smem, two_pass != 0);
                }
                continue;
            }
            scan3_group<M, NR, U, NW>(items, tabs, idx, ng, T, T32, codes, K, L, S, item_surv, item_n, item_slack, qbound, smem, two_pass != 0);
        }
    }
    S3_CTR(9, S3_CLK() - k0);
    S3_CTR(10, 1);
}

// ---- sampled single-pass form (scan mode 5; automatic for short chunks at M <= 8, limit <= 128) ---------------------------
// On short chunks (BASELINE config C2: 1M vectors, 3.9 k candidates per cell, ~3 cells per query) the forms above spend their
// time around the table gathers, not in them (profiles/r03a_c2_probes.txt: 0.183 of 0.271 ms remain with the loops removed;
// every phase of a slot is a few microseconds of latency and the SIMDs are ~1/3 busy): at 128 registers and 37 KB of LDS a CU
// holds four workgroups, too few to hide a slot's chain of dependent loads and its barriers, and the two-pass form gathers every
// candidate twice.  This form is built for occupancy -- 80 registers, 25 KB of LDS: six workgroups = 24 waves per CU -- and does
// less per slot:
//   * its slots are known in advance (static schedule: runs of 32 consecutive slots -- about one cell -- round-robin over the
//     XCDs, a workgroup takes every LW-th slot of its XCD): one lane per slot resolves the descriptor chain slot -> items ->
//     table descriptors for eight of the workgroup's slots at once, into LDS;
//   * the collection threshold comes from a SAMPLE of <= 16 rows of 64 candidates spread over the chunk: the waves write the
//     sample's sums to LDS, wave g takes query g's k_s-th smallest by ballot bisection in registers (no histogram, no LDS
//     atomics); k_s - 4 sqrt(k_s) >= limit x sampled fraction, so that the whole chunk holds >= limit candidates under that
//     value tau but for a ~1e-4 tail.  A query whose other cells already published a bound (qbound) at least as tight skips
//     the sample's verdict: that bound is valid by itself;
//   * every candidate is gathered ONCE; sums up to tau + M + 1 are appended to one list per query (one 4-lane LDS atomic per
//     row of 64 candidates reserves the places of all four queries);
//   * what the sample only made likely is VERIFIED: wave g loads query g's list (<= 504 entries, eight registers per lane),
//     counts the sums <= tau -- fewer than `limit`, or a list that overflowed, puts the slot on the fall-back list, which
//     k_adc_scan3's two-pass form (-> streaming form for crowds of equal sums) works off right after this kernel; the result
//     therefore never depends on the sample -- and cuts the list to the exact limit-th smallest sum + M + 1 (the bracket
//     argument of the two-pass form) before it leaves the CU, so the merge sees ~limit survivors per item.
// Measured on C2 (profiles/r03f..h): 0.268 -> 0.205 ms per 8192 queries; sample of 8 rows with 376-entry lists and k - 3 sqrt(k):
// the same time but ~60 of 6289 slots through the fall-back; here none.  Probes of this kernel: nothing passing 0.146, no main
// pass 0.135, no sample pass either 0.120 ms -- what is left is staging 32 KB of float32 tables per slot (201 MB per batch).
// (A second attempt inside the kernel -- exact two-level histogram threshold -- and a dynamic share of slots were built and
// measured slower, 0.22 - 0.24 ms: the extra code costs the common path more than the fall-back launch costs.)
struct Slot4 {  // one slot's descriptor chain, resolved once (LDS)
    int item[S3G], tab0[S3G], tab1[S3G], q[S3G];
    float qinv[S3G];
    int start_lo, start_hi, len, ng;  // ng < 0: the slot's items do not cover one chunk: they run chunk by chunk (sub-slots)
    int ist_lo[S3G], ist_hi[S3G], ilen[S3G];  // every item's own chunk (read for such slots only)
};
struct Scan4Shared {  // per query
    double inv_up, ub;
    uint64_t ext;
    uint32_t tau;     // >= limit candidates are expected (sample) to have a sum <= tau
    int verify;       // the collection threshold rests on the sample: count before trusting it
    int cnt;          // entries in the query's list
    int q;
};

// sums of UU rows of 64 candidates (software pipelined: the table reads of unit q+1 are issued before the adds of unit q)
template <int M, int UU, int OCT = (M >= 8 ? 8 : 4)>
__device__ __forceinline__ void adc16_rows(const CodeWords<M> (&cur)[UU], const char* __restrict__ tab, const RotConsts<M>& rc,
                                           u32x2_t (&d)[UU]) {
    constexpr int NU = M / OCT;
    u32x2_t fbuf[2][OCT];
    uint32_t D[2][(M + 3) / 4];
    rot_words<M>(cur[0], rc, D[0]);
    adc16_issue<M, OCT>(D[0], 0, tab, rc, fbuf[0]);
#pragma unroll
    for (int q = 0; q < UU * NU; ++q) {
        const int u = q / NU, o = q % NU;
        if (q + 1 < UU * NU) {
            const int u1 = (q + 1) / NU, o1 = (q + 1) % NU;
            if (o1 == 0) rot_words<M>(cur[u1], rc, D[u1 & 1]);
            adc16_issue<M, OCT>(D[u1 & 1], o1, tab, rc, fbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const u32x2_t part = adc16_sum<OCT>(fbuf[q & 1]);
        if (o == 0) {
            d[u] = part;
        } else {
            d[u][0] = pk_add_u16(d[u][0], part[0]);
            d[u][1] = pk_add_u16(d[u][1], part[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// A barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence on every address space (s_waitcnt
// vmcnt(0) before s_barrier: it also waits for every global load and store in flight).  Inside a slot the waves of a
// workgroup exchange data through LDS only, so lgkmcnt(0) is the whole fence.
static __device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef CIS_S4_U
#define CIS_S4_U 2
#endif
#ifndef CIS_S4_UL
#define CIS_S4_UL 2  // rows per iteration of the long-chunk form
#endif
#ifndef CIS_S4_OCT
#define CIS_S4_OCT 4
#endif
#ifndef CIS_S4_WPE
#define CIS_S4_WPE 6
#endif
static const int S4_OCT = CIS_S4_OCT;  // table reads per pipeline unit of the main pass (4: 16 registers of reads in flight)
#ifndef CIS_S4_DEFER
#define CIS_S4_DEFER 1  // the main pass records positions only; the per-query split runs on a second gather of the recorded candidates
#endif
#ifndef CIS_S4_NW_LONG
#define CIS_S4_NW_LONG 4   // waves per slot, long chunks (8: measured slower on C4, 0.275 - 0.32 against 0.243 ms: profiles/r04f_ab.txt)
#endif
#ifndef CIS_S4_NW_SHORT
#define CIS_S4_NW_SHORT 4  // waves per slot, short chunks
#endif
#ifndef CIS_S4_WPE8
#define CIS_S4_WPE8 6      // waves per SIMD the 8-wave long-chunk variant is compiled for (80 registers: three workgroups per CU)
#endif
#ifndef CIS_S4_GLISTS
#define CIS_S4_GLISTS 0    // 1: long chunks keep their per-query lists in global memory (26 KB of LDS per workgroup) -- measured and lost, see GLISTS below
#endif
#ifndef CIS_S4_WPE_LONG
#define CIS_S4_WPE_LONG 6  // waves per SIMD the long-chunk variant is compiled for (80 registers) when its lists are in global memory
#endif
#ifndef CIS_S4_PF
#define CIS_S4_PF 2  // iterations of a wave that its code rows travel ahead
#endif
#ifndef CIS_S4_LCAP
#define CIS_S4_LCAP 504
#endif
#ifndef CIS_S4_NS
#define CIS_S4_NS 16
#endif
#ifndef CIS_S4_Z
#define CIS_S4_Z 4.0f
#endif
static const int S4_DS = 8;              // slot descriptors resolved per round
static const int S4_LCAP = CIS_S4_LCAP;  // entries of a query's list
static const int S4_NS = CIS_S4_NS;      // sample rows per chunk (64 sums per query each)

static const int S4_LCAP_LONG = 1016;    // list entries per query for long chunks (sixteen registers per lane in the verification)

static size_t scan4_lds(int M, int K, int lcap, int nw) {
    const bool glists = lcap > 504 && CIS_S4_DEFER != 0 && CIS_S4_GLISTS != 0;
    const size_t lists = glists ? (size_t)nw * ((32 / nw) * 128) * 2 : (size_t)S3G * lcap * 4, samp = glists ? 0 : (size_t)S3G * S4_NS * 64 * 2;
    return (size_t)K * M * S3G * 2 + (lists > samp ? lists : samp) + S3G * sizeof(Scan4Shared) + 32 + (size_t)S3G * nw * 4 + (S4_DS + 1) * sizeof(Slot4);
}

template <int M, int U, int NW, int WPE, int LCAPT>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_adc_scan4(
    const WorkItem* __restrict__ items, const TabDesc* __restrict__ tabs, const int* __restrict__ slots,
    const int* __restrict__ n_slots_ptr, const float* __restrict__ T32 /* null: converted from T */, const double* __restrict__ T,
    const uint8_t* __restrict__ codes, int K, int L, int S,
    int* __restrict__ dbg /* [2]: slots, fallbacks */, int* __restrict__ fhdr /* fall-back list: [17] = count (queue 0 of a scan3 header) */,
    int* __restrict__ fslots, uint64_t* __restrict__ item_surv, int* __restrict__ item_n, float* __restrict__ item_slack,
    unsigned long long* __restrict__ qbound, float zq /* sample rank margin: k - zq sqrt(k) */, int frac_den /* sample one row in frac_den */,
    int nsx /* most sample rows per chunk */, int dyn /* slots from a counter instead of the static schedule */,
    float sat /* 0, or the saturating scale: a strided sample of the chunk puts the sums of the near candidates at sat * CAP */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int G = S3G;
    // NW waves gather (sample, main pass, second gather); waves 0 .. G-1 also serve one query each in the threshold and verification
    // phases.  NW = 8 (round 4): a slot's latency, not the chip's throughput, bounds the launch -- a slot alone on the chip takes as long
    // as one among 1024 (tools/nq_sweep.py) -- so twice the waves per slot halve the main pass, and three such workgroups per CU are
    // 24 waves against 16
    static_assert(NW >= G && NW % G == 0 && NW <= 8, "waves 0 .. G-1 serve one query each in the threshold and verification phases");
    static_assert(M == 4 || M == 8 || M == 16, "float4s of the half tables are dealt to the threads in rounds of 256");
    constexpr int nf = M / 2;
    constexpr uint32_t CAP = 65535u / M;
    constexpr int LCAP = LCAPT, NS = S4_NS, NRV = (LCAP + 63) / 64;
    // GLISTS (round 4, long chunks with the deferred split): the per-query lists live in the items' survivor rows in GLOBAL memory
    // (written by the second gather, read back from L2 by the verification: ~1600 entries per slot), so LDS holds the tables and the
    // 16-bit position lists only -- 26 KB per workgroup, six workgroups per CU instead of four: a slot is bound by latency, and 46 % of
    // it is serial per-query work during which only other workgroups can use the SIMDs.
    // MEASURED (profiles/r04_scan4_glists_ab.txt, C4, one batch at a time): 128 registers / 4 workgroups per CU 0.237 ms (= the LDS lists,
    // 0.232-0.238), 96 registers / 5 per CU 0.256, 80 registers / 6 per CU 0.310 -- the register budget of the higher occupancies costs
    // more (48 / 71 spilled registers) than the extra workgroups bring.  Off by default; results identical either way.
    constexpr bool GLISTS = (LCAPT > 504) && (CIS_S4_DEFER != 0) && (CIS_S4_GLISTS != 0);
    constexpr size_t LIST_B_FULL = (size_t)G * LCAP * 4 > (size_t)G * NS * 64 * 2 ? (size_t)G * LCAP * 4 : (size_t)G * NS * 64 * 2;
    constexpr size_t LIST_B = GLISTS ? (size_t)NW * (((LCAPT > 504 ? 32 : 16) / NW) * 128) * 2 : LIST_B_FULL;
    char* tab = smem;                                                            // [K][M][G] uint16
    uint32_t* lists = reinterpret_cast<uint32_t*>(smem + (size_t)K * M * G * 2);  // [G][LCAP] (sum << 16 | position)
    uint16_t* samp = reinterpret_cast<uint16_t*>(lists);                         // [G][NS * 64] the sample's sums (before the main pass)
    Scan4Shared* sh = reinterpret_cast<Scan4Shared*>(reinterpret_cast<char*>(lists) + LIST_B);
    uint16_t* thr1 = reinterpret_cast<uint16_t*>(sh + G);  // [G] collection thresholds + 1, packed like the sums
    int* s_flag = reinterpret_cast<int*>(thr1 + 4);
    int* s_cnt = s_flag + 1;                               // [G] list cursors (the 4-lane atomic of the main pass)
    int* wcnt = reinterpret_cast<int*>(reinterpret_cast<char*>(thr1) + 32);  // [G][NW] entries of the wave-private lists (long chunks)
    Slot4* sd = reinterpret_cast<Slot4*>(reinterpret_cast<char*>(thr1) + 32 + G * NW * 4);
#ifdef CIS_S4_SHARED_LISTS  // short chunks with one list per query and an LDS atomic per row: 0.176 against 0.172 ms on C2
    constexpr bool WLISTS = LCAPT > 504;
#else
    constexpr bool WLISTS = true;  // every wave appends to its own quarter of a query's list (no LDS atomic in the loop)
#endif
    constexpr int WCAP = LCAPT / NW;
    // deferred split: 16-bit positions per wave in the list memory (PCAP_LDS of them fit), of which NRP registers' worth are used
    constexpr int PCAP_LDS = (int)(LIST_B / (NW * 2));
    constexpr int NRP = (LCAPT > 504 ? 32 : 16) / NW;      // registers of packed positions per lane in the second gather
    constexpr int PCAP = NRP * 128 < PCAP_LDS ? NRP * 128 : PCAP_LDS;
    static_assert(PCAP % 2 == 0 && (PCAP_LDS * 2) % 4 == 0, "position lists are read back as 32-bit pairs");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = n_slots_ptr[0];
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int LW = gridDim.x >> 3;
    // (Balanced rounds -- the same slots over ceil(slots / rounds) workgroups, the rest leaving at once -- were measured and lost, 0.300
    // against 0.245 ms on C4: the workgroups that leave are the ones the dispatcher placed last, so whole CUs go idle; and three
    // rounds at three workgroups per CU cost what two full rounds and a sparse third do: profiles/r04h_ab.txt.)
    // the workgroup's k-th slot: local index li = lb + k * LW on its XCD -> run (li / 32) * 8 + xcd, place li % 32
    auto slot_of = [&](int k) -> int {
        const int li = lb + k * LW;
        const int j = (((li >> 5) * 8 + xcd) << 5) | (li & 31);
        return j < total ? j : -1;
    };
    const RotConsts<M> rc = make_rot<M>(lane);
    const int nvec = (nf * K) >> 2;  // float4 per half table (<= 256 = threads)
    const long long k0 = S3_CLK();
    (void)k0;
    for (int kb = 0;; kb += S4_DS) {
        int dyn_j = -1;
        if (LCAPT > 504 && dyn) {
            // long chunks: a slot lasts ~100 us and slots differ (cells longer than a chunk run twice): the next slot comes from
            // a counter instead of the static schedule, one per round
            // (dyn == 1: one counter for all; dyn == 2: one queue per XCD holding the runs of 32 slots -- about one cell -- that the
            // static schedule gives that XCD, so that a cell still streams through ONE L2; an XCD whose queue is empty steals)
            __syncthreads();
            if (tid == 0) {
                int jn = -1;
                if (dyn == 1) {
                    jn = atomicAdd(&dbg[6], 1);
                    jn = jn < total ? jn : -1;
                } else {
                    int* qx = dbg - 9;  // the slot header's eight queue counters (zeroed with it; the streaming kernels' own use of them is another launch)
                    for (int a2 = 0; a2 < 8 && jn < 0; ++a2) {
                        const int x = (xcd + a2) & 7;
                        const int li = atomicAdd(&qx[x], 1);
                        const int jj = (((li >> 5) * 8 + x) << 5) | (li & 31);
                        if (jj < total) jn = jj;
                    }
                }
                *s_flag = jn;
            }
            __syncthreads();
            dyn_j = *s_flag;
            if (dyn_j < 0) break;
        } else {
            if (slot_of(kb) < 0) break;  // wave-uniform
        }
        __syncthreads();             // the previous round's descriptors and lists are dead
        if (tid < S4_DS) {           // one lane per slot walks the chain slot -> items -> table descriptors
            Slot4 d;
            const int j = (LCAPT > 504 && dyn) ? (tid == 0 ? dyn_j : -1) : slot_of(kb + tid);
            d.ng = 0; d.len = 0; d.start_lo = d.start_hi = 0;
#pragma unroll
            for (int g = 0; g < G; ++g) { d.item[g] = 0; d.tab0[g] = 0; d.tab1[g] = 0; d.q[g] = 0; d.qinv[g] = 0.f; d.ist_lo[g] = d.ist_hi[g] = d.ilen[g] = 0; }
            if (j >= 0) {
                int idx[G];
                int ng = 0;
                bool same = true;
#pragma unroll
                for (int g = 0; g < G; ++g) idx[g] = slots[j * G + g];
                const int64_t st0 = items[idx[0]].start;
                const int len0 = items[idx[0]].len;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (idx[g] >= 0) ng = g + 1;
                    else idx[g] = idx[0];
                    const WorkItem* it = &items[idx[g]];
                    same = same && it->start == st0 && it->len == len0;
                    const int t0 = it->tab0, t1 = it->tab1;
                    d.item[g] = idx[g]; d.tab0[g] = t0; d.tab1[g] = t1; d.q[g] = it->q;
                    d.ist_lo[g] = (int)(uint32_t)it->start; d.ist_hi[g] = (int)(it->start >> 32); d.ilen[g] = it->len;
                    float mxT = fmaxf(__int_as_float(tabs[t0].pad), __int_as_float(tabs[t1].pad));
                    mxT = fmaxf(mxT, 1e-30f);
                    float qi = ((float)CAP / mxT) * (1.0f - 9.5367431640625e-7f);  // as scan3_group: T32 * qinv stays below cap
                    qi = (mxT < 3.0e38f) ? qi : 0.0f;
                    qi = (qi < 3.0e38f) ? qi : 3.0e38f;
                    d.qinv[g] = qi;
                }
                d.start_lo = (int)(uint32_t)st0; d.start_hi = (int)(st0 >> 32); d.len = len0;
                d.ng = same ? ng : -ng;
            }
            sd[tid] = d;
        }
        __syncthreads();
        for (int kk = 0; kk < S4_DS; ++kk) {
            const Slot4* d = &sd[kk];
            const int ngs0 = __builtin_amdgcn_readfirstlane(d->ng);
            if (ngs0 == 0) break;  // no more slots
            // Items of different chunks in one slot (a cell longer than a chunk: its chunks share the slot key): the slot runs as
            // sub-slots, one per chunk, each with the items of that chunk moved to the front of a scratch descriptor.
            int remaining = ngs0 < 0 ? ((1 << (-ngs0)) - 1) : 0;
            do {
            if (ngs0 < 0) {
                Slot4* ds = &sd[S4_DS];
                if (tid == 0) {
                    const Slot4* src = &sd[kk];
                    int g0 = 0;
                    while (!((remaining >> g0) & 1)) ++g0;
                    int n = 0, rest = remaining;
                    for (int h = g0; h < G; ++h) {
                        if (((remaining >> h) & 1) && src->ist_lo[h] == src->ist_lo[g0] && src->ist_hi[h] == src->ist_hi[g0] && src->ilen[h] == src->ilen[g0]) {
                            ds->item[n] = src->item[h]; ds->tab0[n] = src->tab0[h]; ds->tab1[n] = src->tab1[h]; ds->q[n] = src->q[h];
                            ds->qinv[n] = src->qinv[h];
                            ++n;
                            rest &= ~(1 << h);
                        }
                    }
                    for (int h = n; h < G; ++h) {
                        ds->item[h] = ds->item[0]; ds->tab0[h] = ds->tab0[0]; ds->tab1[h] = ds->tab1[0]; ds->q[h] = ds->q[0]; ds->qinv[h] = ds->qinv[0];
                    }
                    ds->start_lo = src->ist_lo[g0]; ds->start_hi = src->ist_hi[g0]; ds->len = src->ilen[g0];
                    ds->ng = n;
                    ds->ist_lo[0] = rest;  // what is left for the next sub-slot
                }
                lds_barrier();
                d = ds;
                remaining = __builtin_amdgcn_readfirstlane(ds->ist_lo[0]);
            }
            const int ngs = __builtin_amdgcn_readfirstlane(d->ng);
            const long long c0 = S3_CLK();
            (void)c0;
            const int ng = ngs;
            bool fail = false;
            if (!fail) {
                float qinv[G];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    qinv[g] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, d->qinv[g])));
                const int len = __builtin_amdgcn_readfirstlane(d->len);
                const int64_t start = ((int64_t)__builtin_amdgcn_readfirstlane(d->start_hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(d->start_lo);
                if (sat > 0.0f) {
                    // SATURATING scale (M = 16, round 4).  Scaled by the tables' LARGEST entries, sixteen 12-bit entries leave the sums of
                    // the near candidates ~100 units, and the + M + 1 slack of the brackets is then 17 % of the cut: 84 % of the lists
                    // overflowed.  Instead wave g takes 64 candidates of the chunk at a stride, sums them in float32 from the global
                    // tables, and scales query g so that the FOURTH smallest of them for a long chunk (the ~6 % quantile: never one of
                    // the limit best by bad luck alone; a later one for short chunks) lands at sat * CAP; far entries SATURATE at CAP.
                    // Measured on C3: no slot falls back for sat in 0.5 .. 0.95, all do from 1.1 (profiles/r04v_m16.txt).  A saturated entry keeps every sum a lower
                    // bound of d * qinv (truncation and the clamp only round down), so rejecting s > thr stays exact; the other
                    // direction -- "the candidates with s <= v have d * qinv < v + M" -- needs unsaturated entries, which s < CAP
                    // guarantees: thresholds and cuts at or above CAP send the slot to the forms that need no such scale.
                    if (w < G) {
                        const int g = w;
                        if (g < ng) {  // wave-uniform
                            const int t0 = __builtin_amdgcn_readfirstlane(d->tab0[g]), t1 = __builtin_amdgcn_readfirstlane(d->tab1[g]);
                            const int64_t Ta = (int64_t)t0 * nf * K, Tb = (int64_t)t1 * nf * K;
                            const int c = len >= 64 ? (int)(((int64_t)lane * len) >> 6) : lane;
                            uint32_t key = 0x7f7fffffu;
                            if (c < len) {
                                const CodeWords<M> cw = load_code<M>(codes, start + c);
                                float ds = 0.f;
#pragma unroll
                                for (int j = 0; j < M / 2; ++j) ds += tab_f1(T32, T, Ta + j * K + ((cw.w[j >> 2] >> (8 * (j & 3))) & 255u));
#pragma unroll
                                for (int j = 0; j < M / 2; ++j) ds += tab_f1(T32, T, Tb + j * K + ((cw.w[(M / 2 + j) >> 2] >> (8 * ((M / 2 + j) & 3))) & 255u));
                                key = __float_as_uint(fmaxf(ds, 0.f));  // (entries are >= 0: the bit patterns order like the values)
                            }
                            // rank of the sample that stands for "just above the limit-th candidate": 4 + the sample's share of the limit
                            // best (64 L / len: short chunks keep a larger part of themselves)
                            int rk = 4 + (int)((64 * (int64_t)L + len - 1) / len);
                            rk = rk > 48 ? 48 : rk;
                            uint32_t mn = key;
                            for (int r = 0; r < rk; ++r) {
                                uint32_t mx = 0u;
                                mn = key;
                                wave_minmax_step<1>(mn, mx); wave_minmax_step<2>(mn, mx); wave_minmax_step<4>(mn, mx);
                                wave_minmax_step<8>(mn, mx); wave_minmax_step<16>(mn, mx); wave_minmax_step<32>(mn, mx);
                                if (r + 1 < rk) {
                                    const uint64_t hit = __ballot(key == mn);
                                    if (lane == (int)__builtin_ctzll(hit)) key = 0x7f7fffffu;
                                }
                            }
                            const float d4 = __uint_as_float(mn);
                            const float qs = (sat * (float)CAP) / fmaxf(d4, 1e-30f);
                            const float q0 = d->qinv[g];
                            if (lane == 0 && q0 > 0.0f && qs > q0 && qs < 3.0e38f && d4 < 1.0e38f) const_cast<Slot4*>(d)->qinv[g] = qs;
                        }
                    }
                    lds_barrier();
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        qinv[g] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, d->qinv[g])));
                }
                // ---- tables -> 16-bit entries -> LDS (scan3_group's arithmetic) --------------------------------------------------
                // (the thread index is laundered per slot: the staging loop's per-thread offsets are loop invariants of the slot loop,
                // and hoisted they sat in ~20 registers across the hot loop -- spilled, and a spill reload in front of the hot loop makes the
                // compiler wait for ALL outstanding loads at every list store: the code rows then never travel ahead)
                int tid_st = tid;
                asm volatile("" : "+v"(tid_st));
                for (int vt = tid_st; vt < nvec; vt += NW * 64) {
                    // thread -> (sub-quantizer j = vt % nf, four consecutive k): the lanes of a store group spread over the nf
                    // sub-quantizers (a run of consecutive k per lane group put all sixteen lanes on one bank pair: the entry
                    // stride of k is M * 8 bytes)
                    const int j = vt & (nf - 1), kq = vt / nf, k0 = 4 * kq;
                    const int vidx = j * (K >> 2) + kq;  // float4 index inside a half table [nf][K]
                    float4 pv[G][2];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int t0 = __builtin_amdgcn_readfirstlane(d->tab0[g]), t1 = __builtin_amdgcn_readfirstlane(d->tab1[g]);
                        pv[g][0] = tab_f4(T32, T, t0, nf * K, vidx);
                        pv[g][1] = tab_f4(T32, T, t1, nf * K, vidx);
                    }
                    uint32_t* tw = reinterpret_cast<uint32_t*>(tab);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            uint32_t qv[G];
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                const float4 q = pv[g][s2];
                                const float x = c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w));
                                qv[g] = (g < ng) ? (uint32_t)(x * qinv[g]) : CAP;  // truncation: a lower bound of x * qinv
                                qv[g] = qv[g] > CAP ? CAP : qv[g];
                            }
                            u32x2_t pk;
                            pk[0] = qv[0] | (qv[1] << 16);
                            pk[1] = qv[2] | (qv[3] << 16);
                            *reinterpret_cast<u32x2_t*>(tw + (((k0 + c) * M + s2 * nf + j) << 1)) = pk;
                        }
                    }
                }
                if (tid >= 64 && tid < 64 + G) {
                    const int g = tid - 64;
                    float qi = qinv[0];
#pragma unroll
                    for (int gg = 1; gg < G; ++gg) qi = (g == gg) ? qinv[gg] : qi;
                    const double inv_up = (double)qi * (1.0 + 2.384185791015625e-7);
                    const double ub = qi > 0.0f ? (1.0 + 4.76837158203125e-7) / (double)qi : __longlong_as_double(0x7ff0000000000000LL);
                    const int item = d->item[g];
                    const int q = d->q[g];
                    sh[g].inv_up = inv_up;
                    sh[g].ub = ub;
                    sh[g].q = q;
                    sh[g].ext = (g < ng) ? __hip_atomic_load(&qbound[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7ff0000000000000ull;
                    sh[g].cnt = 0;
                    s_cnt[g] = 0;
                    if (g < ng) {
                        item_slack[2 * (int64_t)item + 0] = __double2float_ru(ub * ((double)M + 0.1));
                        item_slack[2 * (int64_t)item + 1] = __double2float_ru(ub);
                    }
                }
                if (tid == 0) *s_flag = 0;
                __amdgpu_buffer_rsrc_t rs;
                {
                    const uint64_t cbase = (uint64_t)(uintptr_t)(codes + start * M);
                    const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cbase);
                    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cbase >> 32));
                    rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)bhi << 32) | blo), 0, len * M, 0x00020000);
                }
                const int nrows = (len + 63) >> 6;
                constexpr bool MULTI = LCAPT > 504;  // long chunks: a tenth of the rows as the sample, bucket minima instead of the sums
                if constexpr (!MULTI) {
                    // sample rows: t * nrows / ns for t < ns = min(nrows, NS) (all rows when the chunk has <= NS of them)
                    const int ns = nrows < NS ? nrows : NS;
                    // the sample's code rows travel while the tables are staged
                    constexpr int SPW = NS / NW;  // sample rows per wave
                    static_assert(NS % (2 * NW) == 0, "sample rows come in pairs per wave");
                    CodeWords<M> sc[SPW];
                    int srow[SPW];
#pragma unroll
                    for (int i = 0; i < SPW; ++i) {
                        const int t = SPW * w + i;
                        srow[i] = t < ns ? (int)(((int64_t)t * nrows) / ns) : nrows;
                        sc[i] = load_code_buf<M>(rs, srow[i] * 64 + lane);  // rows past the chunk read zeros (masked below)
                    }
                    S3_CTR(11, S3_CLK() - c0);
                    lds_barrier();  // B1: tables and scales visible
                    S3_CTR(2, S3_CLK() - c0);
                    // ---- sample pass: wave w computes sample rows 2w, 2w+1; an absent sample leaves 0xffff -------------------------------
                    {
                        u32x2_t dd[SPW];
#if defined(CIS_S4_PROBE) && CIS_S4_PROBE >= 3
                        for (int u = 0; u < SPW; ++u) { dd[u][0] = 0xffffffffu; dd[u][1] = 0xffffffffu; }  // probe: no sample pass
#else
                        adc16_rows<M, SPW, S4_OCT>(sc, tab, rc, dd);
#endif
#pragma unroll
                        for (int u = 0; u < SPW; ++u) {
                            const int t = SPW * w + u;
                            const bool ok = (t < ns) && (srow[u] * 64 + lane < len);
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                const uint32_t sg = (g & 1) ? (dd[u][g >> 1] >> 16) : (dd[u][g >> 1] & 0xffffu);
#if defined(CIS_S4_PROBE) && CIS_S4_PROBE >= 3
                                samp[g * (NS * 64) + t * 64 + lane] = (uint16_t)0xffffu;
#else
                                samp[g * (NS * 64) + t * 64 + lane] = (uint16_t)(ok ? sg : 0xffffu);
#endif
                            }
                        }
                    }
                    lds_barrier();  // B2
                    S3_CTR(3, S3_CLK() - c0);
                    // ---- thresholds: wave g serves query g ---------------------------------------------------------------------------------
                    if (w < G) {
                        const int g = w;
                        uint32_t sv[NS];
                        int nsam = 0;
#pragma unroll
                        for (int r = 0; r < NS; ++r) {
                            sv[r] = samp[g * (NS * 64) + r * 64 + lane];
                            nsam += __popcll(__ballot(sv[r] != 0xffffu));
                        }
                        // sample rank: k - z sqrt(k) >= L * nsam / len (all rows sampled: the L-th smallest itself, exact)
                        const bool exact = ns == nrows;
                        int ksv = L;
                        if (!exact) {
                            const float need = (float)L * (float)nsam / (float)len;
                            const float hz = 0.5f * zq;
                            const float rt = hz + sqrtf(hz * hz + need);  // root of k - z sqrt(k) = need
                            ksv = (int)(rt * rt) + 1;
                        }
                        uint32_t tau = 65534u;
                        bool have_bound = false;
                        if (g < ng && nsam >= ksv) {  // wave-uniform
                            uint32_t lo = 0u, hi = 65534u;
                            while (lo < hi) {
                                const uint32_t p = lo + ((hi - lo) >> 1);
                                int c = 0;
#pragma unroll
                                for (int r = 0; r < NS; ++r) c += __popcll(__ballot(sv[r] <= p));  // (absent samples are 0xffff > p)
                                if (c >= ksv) hi = p;
                                else lo = p + 1;
                            }
                            tau = lo;
                            have_bound = tau < (sat > 0.f ? CAP - (uint32_t)M - 2u : 65534u);  // (saturating scale: the bracket needs sums below CAP)
                        }
                        if (lane == 0) {
                            uint32_t keep = 65534u;
                            if (have_bound) {
                                keep = tau + (uint32_t)M + 1u;  // a sum brackets its distance to within M units (two-pass form above)
                                keep = keep < 65534u ? keep : 65534u;
                            }
                            const uint32_t t_ext = (g < ng) ? bound_to_thr(sh[g].ext, sh[g].inv_up) : 0u;
                            const uint32_t t = keep < t_ext ? keep : t_ext;
                            thr1[g] = (uint16_t)((g < ng) ? t + 1u : 0u);
#if defined(CIS_S4_PROBE) && CIS_S4_PROBE == 1
                            thr1[g] = 0;  // probe: nothing passes
#endif
                            sh[g].tau = tau;
                            // keep < t_ext: the threshold rests on the sample (exact when every row was sampled, else to be verified)
                            sh[g].verify = (have_bound && keep < t_ext && !exact) ? 1 : 0;
#if defined(CIS_S4_PROBE)
                            sh[g].verify = 0;
#endif
                        }
                    }
                } else {
                    // Sample rows: t * nrows / ns for t < ns; ns = about one row in frac_den, at least NS and at most S4_NSX rows, all
                    // rows when the chunk has no more than that.  The waves keep, per lane and query, the MINIMUM of the sample sums
                    // they computed: 4 x 64 bucket minima per query instead of the sums themselves.  count(buckets <= v) <=
                    // count(samples <= v), so the k-th smallest bucket minimum is an upper bound of the k-th smallest sample sum:
                    // a threshold taken from the buckets is never tighter than the one the whole sample would give, costs one packed
                    // min per row and a bisection over four registers, and needs 2 KB of LDS whatever the sample size.
                    int ns = nrows / frac_den;
                    ns = ns < NS ? NS : ns;
                    ns = ns > nsx ? nsx : ns;
                    ns = ns < nrows ? ns : nrows;
                    constexpr int SPW = 4;  // sample rows per wave and round
                    const int rounds = (ns + NW * SPW - 1) / (NW * SPW);
                    // the first round's code rows travel while the tables are staged
                    CodeWords<M> sc[SPW];
                    int srow[SPW];
#pragma unroll
                    for (int i = 0; i < SPW; ++i) {
                        const int t = SPW * w + i;
                        srow[i] = t < ns ? (int)(((int64_t)t * nrows) / ns) : nrows;
                        sc[i] = load_code_buf<M>(rs, srow[i] * 64 + lane);  // rows past the chunk read zeros (masked below)
                    }
                    S3_CTR(11, S3_CLK() - c0);
                    lds_barrier();  // B1: tables and scales visible
                    S3_CTR(2, S3_CLK() - c0);
                    // ---- sample pass --------------------------------------------------------------------------------------------------------
                    {
                        uint32_t bm0 = 0xffffffffu, bm1 = 0xffffffffu;
#pragma unroll 1
                        for (int r = 0; r < rounds; ++r) {
                            CodeWords<M> nsc[SPW];
                            int nrow[SPW];
                            if constexpr (MULTI) {
                                if (r + 1 < rounds) {
#pragma unroll
                                    for (int i = 0; i < SPW; ++i) {
                                        const int t = ((r + 1) * NW + w) * SPW + i;
                                        nrow[i] = t < ns ? (int)(((int64_t)t * nrows) / ns) : nrows;
                                        nsc[i] = load_code_buf<M>(rs, nrow[i] * 64 + lane);
                                    }
                                }
                            }
                            u32x2_t dd[SPW];
#if defined(CIS_S4_PROBE) && CIS_S4_PROBE >= 3
                            for (int u = 0; u < SPW; ++u) { dd[u][0] = 0xffffffffu; dd[u][1] = 0xffffffffu; }  // probe: no sample pass
#else
                            adc16_rows<M, SPW, S4_OCT>(sc, tab, rc, dd);
#endif
#pragma unroll
                            for (int u = 0; u < SPW; ++u) {
                                const bool ok = srow[u] < nrows && (srow[u] * 64 + lane < len);
                                bm0 = pk_min_u16(bm0, ok ? dd[u][0] : 0xffffffffu);
                                bm1 = pk_min_u16(bm1, ok ? dd[u][1] : 0xffffffffu);
                            }
                            if constexpr (MULTI) {
                                if (r + 1 < rounds) {
#pragma unroll
                                    for (int i = 0; i < SPW; ++i) { sc[i] = nsc[i]; srow[i] = nrow[i]; }
                                }
                            }
                        }
                        u32x2_t bm;
                        bm[0] = bm0; bm[1] = bm1;
                        reinterpret_cast<u32x2_t*>(samp)[w * 64 + lane] = bm;  // [NW][64] x four queries
                    }
                    lds_barrier();  // B2
                    S3_CTR(3, S3_CLK() - c0);
                    // ---- thresholds: wave g serves query g ---------------------------------------------------------------------------------
                    if (w < G) {
                        const int g = w;
                        uint32_t sv[NW];
#pragma unroll
                        for (int r = 0; r < NW; ++r) {
                            const u32x2_t x = reinterpret_cast<const volatile u32x2_t*>(samp)[r * 64 + lane];
                            const uint32_t xw = (g >> 1) ? x[1] : x[0];
                            sv[r] = (g & 1) ? (xw >> 16) : (xw & 0xffffu);
                        }
                        // every sampled row is full but the chunk's last one, which is sampled only when all rows are
                        const bool all = ns == nrows;
                        const int nsam = all ? len : ns * 64;
                        // sample rank: k - z sqrt(k) >= L * nsam / len (all rows sampled: the L-th smallest bucket is a valid bound)
                        int ksv = L;
                        if (!all) {
                            const float need = (float)L * (float)nsam / (float)len;
                            const float hz = 0.5f * zq;
                            const float rt = hz + sqrtf(hz * hz + need);  // root of k - z sqrt(k) = need
                            ksv = (int)(rt * rt) + 1;
                        }
                        uint32_t tau = 65534u;
                        bool have_bound = false;
                        if (g < ng && nsam >= ksv) {  // wave-uniform
                            uint32_t lo = 0u, hi = 65534u;
                            while (lo < hi) {
                                const uint32_t p = lo + ((hi - lo) >> 1);
                                int c = 0;
#pragma unroll
                                for (int r = 0; r < NW; ++r) c += __popcll(__ballot(sv[r] <= p));  // (empty buckets are 0xffff > p)
                                if (c >= ksv) hi = p;
                                else lo = p + 1;
                            }
                            tau = lo;
                            have_bound = tau < (sat > 0.f ? CAP - (uint32_t)M - 2u : 65534u);  // (saturating scale: the bracket needs sums below CAP)
                        }
                        if (lane == 0) {
                            uint32_t keep = 65534u;
                            if (have_bound) {
                                keep = tau + (uint32_t)M + 1u;  // a sum brackets its distance to within M units (two-pass form above)
                                keep = keep < 65534u ? keep : 65534u;
                            }
                            const uint32_t t_ext = (g < ng) ? bound_to_thr(sh[g].ext, sh[g].inv_up) : 0u;
                            const uint32_t t = keep < t_ext ? keep : t_ext;
                            thr1[g] = (uint16_t)((g < ng) ? t + 1u : 0u);
#if defined(CIS_S4_PROBE) && CIS_S4_PROBE == 1
                            thr1[g] = 0;  // probe: nothing passes
#endif
                            sh[g].tau = tau;
                            // keep < t_ext: the threshold rests on the sample (valid by itself when every row was sampled, else to be verified)
                            sh[g].verify = (have_bound && keep < t_ext && !all) ? 1 : 0;
#if defined(CIS_S4_PROBE)
                            sh[g].verify = 0;
#endif
                        }
                    }
                }
                lds_barrier();  // B3: thresholds set, the sample is dead (its memory is the lists from here on)
                S3_CTR(4, S3_CLK() - c0);
                // ---- main pass: every candidate once ------------------------------------------------------------------------------------
                {
                    const u32x2_t tpk = *reinterpret_cast<const volatile u32x2_t*>(thr1);
                    const uint32_t s01 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tpk[0]);
                    const uint32_t s23 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tpk[1]);
#if defined(CIS_S4_PROBE) && CIS_S4_PROBE >= 2
                    const int nit = 0;  // probe: no main pass
#else
                    const int nit = (len + 64 * U - 1) / (64 * U);
#endif
                    // code rows travel PF iterations of this wave ahead of their use (a ring of PF register sets, the loop unrolled PF
                    // times so that the sets are static): one iteration ahead left the wave waiting at the end of most iterations --
                    // an iteration is ~100 instructions, a code row comes from L2 / Infinity Cache / HBM
                    constexpr int PF = CIS_S4_PF;
                    CodeWords<M> ring[PF][U];
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
#pragma unroll
                        for (int u = 0; u < U; ++u) ring[k][u] = load_code_buf<M>(rs, (w + k * NW) * 64 * U + u * 64 + lane);  // past the chunk: zeros, never consumed
                        // (the sets are requested in the order the loop consumes them: entering the loop with set 0 as the NEWEST request would
                        // make the wait at the loop head vmcnt(0) on every trip)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    int wcur[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) wcur[g] = 0;
#if CIS_S4_DEFER
                    // Deferred split (round 4).  ~94 % of the rows of a long chunk carry a lane that passes for SOME query (four queries x 64
                    // lanes against a threshold that lets ~1 % through), and the per-query ballots / prefix counts / stores of such a row cost
                    // about as many instructions as its gathers (profiles/r03zz_c4_sq_pmc.csv: 1.8x the bare loop).  The loop now only
                    // records WHICH candidates passed -- one 16-bit position per passing lane, one ballot, in a wave-private list that lives
                    // where the sample was -- and the ~2-3 % recorded candidates are gathered a second time afterwards (the tables are still
                    // in LDS), where the per-query split runs on 64 passing lanes per row instead of one or two.
                    uint16_t* plist = reinterpret_cast<uint16_t*>(lists) + (size_t)w * PCAP_LDS;
                    int pcur = 0;
#endif
                    for (int iter0 = w; iter0 < nit; iter0 += NW * PF) {
#pragma unroll
                      for (int k = 0; k < PF; ++k) {
                        // (no early exit inside the unrolled group: an iteration past the chunk reads zeros and its rows are masked out
                        // below -- at most PF - 1 idle iterations per wave and slot, and the register sets keep their names)
                        const int iter = iter0 + k * NW;
                        const int base = iter * 64 * U;
                        u32x2_t dd[U];
                        adc16_rows<M, U, S4_OCT>(ring[k], tab, rc, dd);
                        // the set's next rows are requested as soon as its words have been consumed (the loads target the same registers:
                        // no copies).  Unconditional -- past the chunk the descriptor returns zeros -- so that every path through the
                        // loop has the same number of loads in flight and the wait before a set's use is vmcnt((PF - 1) * U), not vmcnt(0)
#pragma unroll
                        for (int u = 0; u < U; ++u) ring[k][u] = load_code_buf<M>(rs, (iter + PF * NW) * 64 * U + u * 64 + lane);
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t xx = pk_subsat_u16(tpk[0], dd[u][0]) | pk_subsat_u16(tpk[1], dd[u][1]);
                            unsigned long long am = __ballot(xx != 0u);
                            const int n = len - base - u * 64;
                            if (n < 64) am &= n <= 0 ? 0ull : ((1ull << n) - 1ull);
                            if (am == 0ull) continue;  // scalar branch
#if CIS_S4_DEFER
                            if constexpr (WLISTS) {
                                const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, pcur));
                                if (((am >> lane) & 1ull) && idx < PCAP) plist[idx] = (uint16_t)(base + u * 64 + lane);
                                pcur += __popcll(am);
                                continue;
                            }
#endif
                            if constexpr (WLISTS) {
#ifdef CIS_S4_SCALAR_APPEND  // measured on C4: 0.341 ms against 0.307 ms for the ballot form below (the scalar chain serialises)
                                // A row that gets here carries one passing lane as a rule (the threshold lets ~1 % through): the lanes
                                // are taken one by one on the scalar unit -- two readlanes, four scalar compares, a uniform LDS store per
                                // pass -- instead of four ballots and prefix counts over the whole wave.
                                while (am != 0ull) {
                                    const int l = __builtin_ctzll(am);
                                    am &= am - 1ull;
                                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)dd[u][0], l);
                                    const uint32_t d1 = (uint32_t)__builtin_amdgcn_readlane((int)dd[u][1], l);
                                    const uint32_t pos = (uint32_t)(base + u * 64 + l);
#pragma unroll
                                    for (int g = 0; g < G; ++g) {
                                        const uint32_t dw = (g >> 1) ? d1 : d0;
                                        const uint32_t sg = (g & 1) ? (dw >> 16) : (dw & 0xffffu);
                                        const uint32_t t1 = (g & 1) ? ((g >> 1) ? s23 >> 16 : s01 >> 16) : ((g >> 1) ? s23 & 0xffffu : s01 & 0xffffu);
                                        if (sg < t1) {  // scalar
                                            if (wcur[g] < WCAP) lists[(g * NW + w) * WCAP + wcur[g]] = (sg << 16) | pos;  // every lane stores the same word
                                            wcur[g] += 1;
                                        }
                                    }
                                }
#else
#pragma unroll
                                for (int g = 0; g < G; ++g) {
                                    const uint32_t sg = (g & 1) ? (dd[u][g >> 1] >> 16) : (dd[u][g >> 1] & 0xffffu);
                                    const uint32_t t1 = (g & 1) ? ((g >> 1) ? s23 >> 16 : s01 >> 16) : ((g >> 1) ? s23 & 0xffffu : s01 & 0xffffu);
                                    const unsigned long long mg = __ballot(sg < t1) & am;
                                    if (mg == 0ull) continue;
                                    const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(mg >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mg, wcur[g]));
                                    if (((mg >> lane) & 1ull) && idx < WCAP) lists[(g * NW + w) * WCAP + idx] = (sg << 16) | (uint32_t)(base + u * 64 + lane);
                                    wcur[g] += __popcll(mg);
                                }
#endif
                                continue;
                            }
                            unsigned long long m[G];
                            int mine = 0;  // lane g: the places query g needs
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                const uint32_t sg = (g & 1) ? (dd[u][g >> 1] >> 16) : (dd[u][g >> 1] & 0xffffu);
                                const uint32_t t1 = (g & 1) ? ((g >> 1) ? s23 >> 16 : s01 >> 16) : ((g >> 1) ? s23 & 0xffffu : s01 & 0xffffu);
                                m[g] = __ballot(sg < t1) & am;
                                mine = (lane == g) ? __popcll(m[g]) : mine;
                            }
                            int got = 0;
                            if (lane < G) got = atomicAdd(&s_cnt[lane], mine);  // one LDS atomic reserves the places of all four queries
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                if (m[g] == 0ull) continue;
                                const int b0 = __builtin_amdgcn_readlane(got, g);
                                const uint32_t sg = (g & 1) ? (dd[u][g >> 1] >> 16) : (dd[u][g >> 1] & 0xffffu);
                                const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(m[g] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[g], b0));
                                if (((m[g] >> lane) & 1ull) && idx < LCAP) lists[g * LCAP + idx] = (sg << 16) | (uint32_t)(base + u * 64 + lane);
                            }
                        }
                      }
                    }
                    S3_CTR(13, S3_CLK() - c0);
#if CIS_S4_DEFER
                    if constexpr (WLISTS) {
                        // the recorded positions move to registers (two per register), then the list memory becomes the per-query lists
                        const bool pover = pcur > PCAP;  // more passing candidates than the position list holds: the slot falls back
                        const int pn = pover ? 0 : pcur;
                        uint32_t pp[NRP];
#pragma unroll
                        for (int r = 0; r < NRP; ++r)
                            pp[r] = (2 * (r * 64) < pn) ? reinterpret_cast<const volatile uint32_t*>(plist)[r * 64 + lane] : 0u;
                        lds_barrier();  // B3b: every wave holds its positions; nobody still writes a position list
                        uint32_t* grow[G];  // GLISTS: the items' survivor rows as 32-bit list memory (S 8-byte entries = 2 S list entries >= LCAP)
#pragma unroll
                        for (int g = 0; g < G; ++g)
                            grow[g] = reinterpret_cast<uint32_t*>(item_surv + (int64_t)__builtin_amdgcn_readfirstlane(d->item[g]) * S);
                        if (pn > 0) {
                            auto pos_of = [&](int it) -> uint32_t {  // iteration it takes half (it & 1) of register it >> 1
                                uint32_t v = pp[0];
#pragma unroll
                                for (int r = 1; r < NRP; ++r) v = ((it >> 1) == r) ? pp[r] : v;
                                return (it & 1) ? (v >> 16) : (v & 0xffffu);
                            };
                            auto on_of = [&](int it) -> bool { return 2 * (((it >> 1) * 64) + lane) + (it & 1) < pn; };
                            const int nit2 = 2 * ((pn + 127) >> 7);  // iterations: halves of ceil(pn / 128) registers
                            CodeWords<M> nxc[1];
                            uint32_t npos = pos_of(0);
                            nxc[0] = load_code_buf<M>(rs, on_of(0) ? (int)npos : len);  // past the chunk: zeros, masked below
#pragma unroll 1
                            for (int it = 0; it < nit2; ++it) {
                                CodeWords<M> cc[1];
                                cc[0] = nxc[0];
                                const uint32_t pos = npos;
                                const bool on = on_of(it);
                                if (it + 1 < nit2) {
                                    npos = pos_of(it + 1);
                                    nxc[0] = load_code_buf<M>(rs, on_of(it + 1) ? (int)npos : len);
                                }
                                u32x2_t d1[1];
                                adc16_rows<M, 1, S4_OCT>(cc, tab, rc, d1);
                                const unsigned long long am2 = __ballot(on);
#pragma unroll
                                for (int g = 0; g < G; ++g) {
                                    const uint32_t sg = (g & 1) ? (d1[0][g >> 1] >> 16) : (d1[0][g >> 1] & 0xffffu);
                                    const uint32_t t1 = (g & 1) ? ((g >> 1) ? s23 >> 16 : s01 >> 16) : ((g >> 1) ? s23 & 0xffffu : s01 & 0xffffu);
                                    const unsigned long long mg = __ballot(sg < t1) & am2;
                                    if (mg == 0ull) continue;
                                    const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(mg >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mg, wcur[g]));
                                    if (((mg >> lane) & 1ull) && idx < WCAP) {
                                        if constexpr (GLISTS) grow[g][w * WCAP + idx] = (sg << 16) | pos;
                                        else lists[(g * NW + w) * WCAP + idx] = (sg << 16) | pos;
                                    }
                                    wcur[g] += __popcll(mg);
                                }
                            }
                        }
                        if (pover) wcur[0] = WCAP + 1;  // reported like an overflowing quarter (the verification below sends the slot to the fall-back)
                    }
#endif
                    if constexpr (WLISTS) {
                        if (lane < G) {
                            int c = wcur[0];
#pragma unroll
                            for (int g = 1; g < G; ++g) c = (lane == g) ? wcur[g] : c;
                            wcnt[lane * NW + w] = c;
                        }
                    }
                }
                S3_CTR(12, S3_CLK() - c0);
                if constexpr (GLISTS) __syncthreads();  // B4: lists complete -- in global memory: the barrier also waits for the stores (vmcnt)
                else lds_barrier();  // B4: lists complete
                S3_CTR(5, S3_CLK() - c0);
                // ---- verification + cut + write-out: wave g serves query g ------------------------------------------------------------
                bool bad = false;
                if (w < ng) {
                    const int g = w;
                    int tot = 0;
                    int wtot[NW];
                    int lane_v = lane;  // laundered: the verification's per-lane offsets are not to live across the hot loop
                    asm volatile("" : "+v"(lane_v));
                    if constexpr (WLISTS) {
#pragma unroll
                        for (int r = 0; r < NW; ++r) {
                            wtot[r] = wcnt[g * NW + r];
                            bad = bad || wtot[r] > WCAP;  // a wave's quarter overflowed
                            tot += wtot[r];
                        }
                    } else {
                        tot = s_cnt[g];
                        bad = tot > LCAP;  // the list overflowed (a crowd of equal sums, or a threshold far too loose)
                    }
                    if (bad && lane_v == 0) atomicAdd(&dbg[2], 1);
                    if (!bad) {
                        uint32_t ent[NRV];
                        bool val[NRV];
#pragma unroll
                        for (int r = 0; r < NRV; ++r) {
                            if constexpr (WLISTS) {
                                constexpr int RW = NRV / NW;  // registers per wave quarter
                                const int e = (r % RW) * 64 + lane_v;
                                val[r] = e < wtot[r / RW];
                                if constexpr (GLISTS) {  // written by the other waves of this workgroup: read past the L1 (device scope)
                                    const uint32_t* gr = reinterpret_cast<const uint32_t*>(item_surv + (int64_t)__builtin_amdgcn_readfirstlane(d->item[g]) * S);
                                    ent[r] = val[r] ? __hip_atomic_load(gr + (r / RW) * WCAP + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                                } else {
                                    ent[r] = val[r] ? lists[(g * NW + r / RW) * WCAP + e] : 0xffffffffu;
                                }
                            } else {
                                val[r] = r * 64 + lane_v < tot;
                                ent[r] = val[r] ? lists[g * LCAP + r * 64 + lane_v] : 0xffffffffu;
                            }
                        }
                        const uint32_t tau = sh[g].tau;
                        if (sh[g].verify) {
                            int nle = 0;
#pragma unroll
                            for (int r = 0; r < NRV; ++r) nle += __popcll(__ballot(val[r] && (ent[r] >> 16) <= tau));
                            bad = nle < L;
                            if (bad && lane_v == 0) atomicAdd(&dbg[3], 1);
                        }
                        if (!bad) {
                            uint32_t cut = 65535u;
                            if (tot >= L) {
                                // v = the L-th smallest collected sum = the chunk's L-th smallest (everything up to the collection
                                // threshold is in the list); sums above v + M + 1 are strictly worse than L candidates
                                uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
                                for (int r = 0; r < NRV; ++r) {
                                    const uint32_t s = ent[r] >> 16;
                                    mn = (val[r] && s < mn) ? s : mn;
                                    mx = (val[r] && s > mx) ? s : mx;
                                }
                                wave_minmax_step<1>(mn, mx); wave_minmax_step<2>(mn, mx); wave_minmax_step<4>(mn, mx);
                                wave_minmax_step<8>(mn, mx); wave_minmax_step<16>(mn, mx); wave_minmax_step<32>(mn, mx);
                                uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)mn), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)mx);
                                // hi always has >= L sums at or below it (it starts at the largest collected sum and tot >= L); the search stops
                                // as soon as that count is within L + 16: any such value serves as v (a step costs NRV ballots, the exact L-th
                                // smallest ~12 steps, this ~5; the merge cuts the few extra survivors before re-scoring)
                                while (lo < hi) {  // wave-uniform bisection on the 16-bit value
                                    const uint32_t p = lo + ((hi - lo) >> 1);
                                    int c = 0;
#pragma unroll
                                    for (int r = 0; r < NRV; ++r) c += __popcll(__ballot(val[r] && (ent[r] >> 16) <= p));
                                    if (c >= L) {
                                        hi = p;
                                        if (c <= L + 16) break;
                                    } else {
                                        lo = p + 1;
                                    }
                                }
                                cut = hi + (uint32_t)M + 1u;
                                if (sat > 0.f && cut >= CAP) {  // saturating scale: the L best are not all below CAP -- no valid bracket from this scale
                                    bad = true;
                                    if (lane_v == 0) atomicAdd(&dbg[4], 1);
                                }
                                const uint64_t b = val_to_bound(hi, M, sh[g].ub);  // >= L candidates of this chunk do not exceed it
                                if (lane_v == 0 && !bad && b < sh[g].ext) atomicMin(&qbound[sh[g].q], (unsigned long long)b);
                            } else if (sat > 0.f) {
                                // fewer than L collected: everything leaves; with the saturating scale every survivor must still be below CAP
                                // (the merge turns a survivor's sum into an upper bound of its distance, which a saturated entry would break)
                                bool sat = false;
#pragma unroll
                                for (int r = 0; r < NRV; ++r) sat = sat || (__ballot(val[r] && (ent[r] >> 16) + (uint32_t)M + 1u >= CAP) != 0ull);
                                if (sat) {
                                    bad = true;
                                    if (lane_v == 0) atomicAdd(&dbg[4], 1);
                                }
                            }
                            const int item = __builtin_amdgcn_readfirstlane(d->item[g]);
                            uint64_t* out = item_surv + (int64_t)item * S;
                            if constexpr (LCAP > 504) {  // an item's survivor row holds S entries: a crowd of sums at the cut goes to the forms that handle crowds
                                int nk = 0;
#pragma unroll
                                for (int r = 0; r < NRV; ++r) nk += __popcll(__ballot(val[r] && (ent[r] >> 16) <= cut));
                                if (nk > S) {
                                    bad = true;
                                    if (lane_v == 0) atomicAdd(&dbg[4], 1);
                                }
                            }
                            if (!bad) {
                                int kept = 0;
#pragma unroll
                                for (int r = 0; r < NRV; ++r) {
                                    const bool kp = val[r] && (ent[r] >> 16) <= cut;
                                    const unsigned long long mk = __ballot(kp);
                                    const int idx = kept + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
                                    if (kp) out[idx] = ((uint64_t)(ent[r] >> 16) << 32) | (ent[r] & 0xffffu);
                                    kept += __popcll(mk);
                                }
                                if (lane_v == 0) item_n[item] = kept;
                            }
                        }
                    }
                    if (bad && lane_v == 0) *s_flag = 1;
                }
                lds_barrier();  // B5: the lists have been read, the verdict is in
                S3_CTR(6, S3_CLK() - c0);
                fail = *s_flag != 0;
            }
            S3_CTR(0, 1);
            if (tid == 0) {
                atomicAdd(&dbg[0], 1);
                if (ngs0 < 0) atomicAdd(&dbg[5], 1);
                if (fail) {
                    // the sample misjudged the chunk, or a crowd of equal sums: the (sub-)slot goes to the forms that need no sample
                    // (their output replaces whatever this slot wrote)
                    atomicAdd(&dbg[1], 1);
                    const int f = atomicAdd(&fhdr[17], 1);
#pragma unroll
                    for (int g = 0; g < G; ++g) fslots[f * G + g] = g < ng ? d->item[g] : -1;
                }
            }
            } while (remaining != 0);
        }
    }
    S3_CTR(9, S3_CLK() - k0);
    S3_CTR(10, 1);
}

// ---- k_adc_scan5 (round 5): ONE threshold per query for the whole batch, eight queries per slot ------------------------------
// k_adc_scan4 spends 46 % of a slot in serial per-query phases (sample, threshold bisection, second gather, verification: one wave per
// query while the others idle) and issues 2.3 x the instructions of its gather loop.  The streaming route
// (lopq_stream.hip) showed the alternative: take the threshold out of the slot.
//   1. k_adc_scan5<SAMPLE>  every slot stages its tables and gathers one row in `ss`; a lane folds the smallest upper bound
//                           (sum + M) * ub it saw into one of B buckets of its query (atomicMin on float bits);
//   2. k_scan5_tau          tau[q] = the k-th smallest bucket minimum, k = P x sampled fraction: about P (~2.4 x limit) candidates of
//                           the query lie under it (count(buckets <= v) <= count(samples <= v): never tighter than the sample's own);
//   3. k_adc_scan5          the slot's only phases are staging and gathering: thresholds thr_g = floor(tau[q_g] * qinv_g (1 + 2^-22)) + 1
//                           -- a candidate with d <= tau has sum <= d qinv (1 + 2^-23) < thr (truncated, saturating entries only round
//                           sums DOWN) -- the loop records the positions where ANY of the eight queries passes (one packed compare,
//                           one ballot), a second gather of those ~5 % splits them per query, counts, reserves room in the items'
//                           survivor rows with one atomic per (wave, query) and writes (sum << 32 | position): k_merge_survivors'
//                           input.  It also counts, per query, the listed candidates whose UPPER bound (sum + M) ub is <= tau;
//   4. k_scan5_check        the proof: that count >= min(limit, candidates) -- then the limit-th smallest upper bound is <= tau, every
//                           candidate of the true top `limit` has d <= tau and was listed (ties included) -- and no row or position list
//                           overflowed.  Slots with a query that fails go to the fall-back list k_adc_scan3 works off (as k_adc_scan4's).
// Eight queries per slot (ds_read_b128: 8 x 16-bit entries; four packed adds per gather): a code row is fetched, rotated and its
// bytes extracted once for eight queries instead of four, and a 16-lane read group meets two lanes per sub-quantizer instead of four.
static const int S5G = 8;
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

template <int M, int OCT>
__device__ __forceinline__ void adc16x8_issue(const uint32_t (&D)[(M + 3) / 4], int o, const char* __restrict__ tab, const RotConsts<M>& rc,
                                              u32x4v (&f)[OCT]) {
    constexpr int SH = ((M == 4) ? 2 : (M == 8 ? 3 : 4)) + 4;  // log2(M * 16 bytes): one k-row of the table
#pragma unroll
    for (int i = 0; i < OCT; ++i) {
        const int t = o * OCT + i;
        const uint32_t k = __builtin_amdgcn_ubfe(D[t >> 2], rc.sh[t & 3], 8);
        f[i] = *reinterpret_cast<const u32x4v*>(tab + ((k << SH) | (rc.cj[t] << 2)));
    }
}

template <int OCT>
__device__ __forceinline__ u32x4v adc16x8_sum(const u32x4v (&f)[OCT]) {
    u32x4v a = f[0];
#pragma unroll
    for (int i = 1; i < OCT; ++i) {
        a[0] = pk_add_u16(a[0], f[i][0]);
        a[1] = pk_add_u16(a[1], f[i][1]);
        a[2] = pk_add_u16(a[2], f[i][2]);
        a[3] = pk_add_u16(a[3], f[i][3]);
    }
    return a;
}

// sums of UU rows of 64 candidates for eight queries (software pipelined like adc16_rows)
template <int M, int UU, int OCT>
__device__ __forceinline__ void adc16x8_rows(const CodeWords<M> (&cur)[UU], const char* __restrict__ tab, const RotConsts<M>& rc, u32x4v (&d)[UU]) {
    constexpr int NU = M / OCT;
    u32x4v fbuf[2][OCT];
    uint32_t D[2][(M + 3) / 4];
    rot_words<M>(cur[0], rc, D[0]);
    adc16x8_issue<M, OCT>(D[0], 0, tab, rc, fbuf[0]);
#pragma unroll
    for (int q = 0; q < UU * NU; ++q) {
        const int u = q / NU, o = q % NU;
        if (q + 1 < UU * NU) {
            const int u1 = (q + 1) / NU, o1 = (q + 1) % NU;
            if (o1 == 0) rot_words<M>(cur[u1], rc, D[u1 & 1]);
            adc16x8_issue<M, OCT>(D[u1 & 1], o1, tab, rc, fbuf[(q + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const u32x4v part = adc16x8_sum<OCT>(fbuf[q & 1]);
        if (o == 0) {
            d[u] = part;
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) d[u][x] = pk_add_u16(d[u][x], part[x]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct Slot5 {
    int item[S5G], tab0[S5G], tab1[S5G], q[S5G];
    float qinv[S5G];
    uint32_t thr1[S5G], ok1[S5G];  // keep sum < thr1; sum < ok1: the candidate's upper bound is <= tau
    float ubf[S5G];                // (1 + 2^-21) / qinv, rounded up
    int start_lo, start_hi, len, ng, same;
};

static __device__ __forceinline__ uint32_t s5_sum_of(const u32x4v& d, int g) {  // query g's 16-bit sum
    const uint32_t w = d[g >> 1];
    return (g & 1) ? (w >> 16) : (w & 0xffffu);
}

static const int S5_PCAP = 1024;  // recorded positions per wave and slot

static size_t scan5_lds(int M, int K) { return (size_t)K * M * S5G * 2 + (size_t)4 * S5_PCAP * 2 + sizeof(Slot5) + 64; }

template <int M, bool SAMPLE, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_adc_scan5(
    const WorkItem* __restrict__ items, const TabDesc* __restrict__ tabs, const int* __restrict__ slots, const int* __restrict__ n_slots_ptr,
    const float* __restrict__ T32 /* null: converted from T */, const double* __restrict__ T, const uint8_t* __restrict__ codes, int K, int S,
    const float* __restrict__ tau, uint32_t* __restrict__ bmin, int B, int* __restrict__ nsamp, int ss,
    uint64_t* __restrict__ item_surv, int* __restrict__ item_n, float* __restrict__ item_slack, int* __restrict__ cnt_ok, int* __restrict__ ovf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int G = S5G, NW = 4, nf = M / 2, U = 2, PF = 2, OCT = 4;
    constexpr uint32_t CAP = 65535u / M;
    char* tab = smem;                                                                  // [K][M][G] uint16
    uint16_t* plist_all = reinterpret_cast<uint16_t*>(smem + (size_t)K * M * G * 2);    // [NW][S5_PCAP]
    Slot5* sd = reinterpret_cast<Slot5*>(reinterpret_cast<char*>(plist_all) + (size_t)NW * S5_PCAP * 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = n_slots_ptr[0];
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, LW = gridDim.x >> 3;
    const RotConsts<M> rc = make_rot<M>(lane);
    const int nvec = (nf * K) >> 2;  // float4 per half table
    for (int kk = 0;; ++kk) {
        // the workgroup's kk-th slot: runs of 32 consecutive slots (a few cells) stay on one XCD, as in k_adc_scan4
        const int li = lb + kk * LW;
        const int j = (((li >> 5) * 8 + xcd) << 5) | (li & 31);
        if ((li >> 5) * 256 >= total + 256) break;   // past every run (uniform)
        if (j >= total) continue;
        __syncthreads();  // the previous slot's tables, lists and descriptor are dead
        if (tid < G) {    // lane g resolves item g of the slot
            const int g = tid;
            int idx = slots[j * G + g];
            const int idx0 = slots[j * G];
            const bool on = idx >= 0;
            idx = on ? idx : idx0;
            const WorkItem it = items[idx];
            const WorkItem it0 = items[idx0];
            float mxT = fmaxf(__int_as_float(tabs[it.tab0].pad), __int_as_float(tabs[it.tab1].pad));
            mxT = fmaxf(mxT, 1e-30f);
            float qi = ((float)CAP / mxT) * (1.0f - 9.5367431640625e-7f);  // as k_adc_scan4: T32 * qinv stays below cap
            qi = (mxT < 3.0e38f) ? qi : 0.0f;
            qi = (qi < 3.0e38f) ? qi : 3.0e38f;
            sd->item[g] = idx; sd->tab0[g] = it.tab0; sd->tab1[g] = it.tab1; sd->q[g] = on ? it.q : -1;
            sd->qinv[g] = qi;
            const double inv_up = (double)qi * (1.0 + 2.384185791015625e-7);
            const double ub = qi > 0.0f ? (1.0 + 4.76837158203125e-7) / (double)qi : __longlong_as_double(0x7ff0000000000000LL);
            sd->ubf[g] = __double2float_ru(ub);
            uint32_t t1 = 0u, o1 = 0u;
            if (on && !SAMPLE) {
                const float tq = tau[it.q];
                if (!(tq < 3.0e38f)) { t1 = 65535u; o1 = 65535u; }   // no threshold: everything is listed (and counts)
                else {
                    const double x = (double)tq * inv_up;
                    t1 = x >= 65533.0 ? 65535u : (uint32_t)x + 2u;    // keep sum <= floor(x) + 1 (bound_to_thr), as sum < t1
                    const double y = (double)tq / ((double)sd->ubf[g]) - (double)M - 1.0;  // (sum + M) ubf <= tau  <=  sum <= y
                    o1 = y <= 0.0 ? 0u : (y >= 65533.0 ? 65535u : (uint32_t)y + 1u);      // counted: sum < o1
                }
            }
            sd->thr1[g] = t1; sd->ok1[g] = o1;
            if (on && !SAMPLE) {
                item_slack[2 * (int64_t)idx + 0] = __double2float_ru(ub * ((double)M + 0.1));
                item_slack[2 * (int64_t)idx + 1] = __double2float_ru(ub);
            }
            // all items of a slot are the SAME chunk of one cell (the slot key); a slot that mixes chunks (a cell of more than 16
            // chunks) is not scanned here: its queries are flagged and take the fall-back
            const bool same = it.start == it0.start && it.len == it0.len;
            const unsigned long long sm = __ballot(same || !on);
            if (g == 0) {
                sd->start_lo = (int)(uint32_t)it0.start; sd->start_hi = (int)(it0.start >> 32); sd->len = it0.len;
                sd->same = ((sm & 0xffull) == 0xffull) ? 1 : 0;
            }
            const unsigned long long om = __ballot(on);
            if (g == 0) sd->ng = 64 - __builtin_clzll((om & 0xffull) | 0ull) ;
        }
        __syncthreads();
        const int ng = __builtin_amdgcn_readfirstlane(sd->ng);
        const int len = __builtin_amdgcn_readfirstlane(sd->len);
        const int64_t start = ((int64_t)__builtin_amdgcn_readfirstlane(sd->start_hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(sd->start_lo);
        if (!__builtin_amdgcn_readfirstlane(sd->same)) {
            if (!SAMPLE && tid < ng && sd->q[tid] >= 0) ovf[sd->q[tid]] = 1;
            continue;
        }
        // ---- tables -> 16-bit entries -> LDS: thread -> (sub-quantizer j = vt % nf, four consecutive k), half tables one after the other
        {
            float qinv[G];
#pragma unroll
            for (int g = 0; g < G; ++g) qinv[g] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sd->qinv[g])));
            int tid_st = tid;
            asm volatile("" : "+v"(tid_st));
            for (int vt = tid_st; vt < nvec; vt += 256) {
                const int jq = vt & (nf - 1), kq = vt / nf, k0 = 4 * kq;
                const int vidx = jq * (K >> 2) + kq;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float4 pv[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int tb = __builtin_amdgcn_readfirstlane(s2 ? sd->tab1[g] : sd->tab0[g]);
                        pv[g] = tab_f4(T32, T, tb, nf * K, vidx);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        uint32_t qv[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const float x = c == 0 ? pv[g].x : (c == 1 ? pv[g].y : (c == 2 ? pv[g].z : pv[g].w));
                            qv[g] = (g < ng) ? (uint32_t)(x * qinv[g]) : CAP;  // truncation: a lower bound of x * qinv
                            qv[g] = qv[g] > CAP ? CAP : qv[g];
                        }
                        u32x4v pk;
                        pk[0] = qv[0] | (qv[1] << 16); pk[1] = qv[2] | (qv[3] << 16); pk[2] = qv[4] | (qv[5] << 16); pk[3] = qv[6] | (qv[7] << 16);
                        *reinterpret_cast<u32x4v*>(tab + ((size_t)((k0 + c) * M + s2 * nf + jq) << 4)) = pk;
                    }
                }
            }
        }
        __amdgpu_buffer_rsrc_t rs;
        {
            const uint64_t cbase = (uint64_t)(uintptr_t)(codes + start * M);
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cbase);
            const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cbase >> 32));
            rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)bhi << 32) | blo), 0, len * M, 0x00020000);
        }
        const int nrows = (len + 63) >> 6;
        if constexpr (SAMPLE) {
            // every ss-th row (all rows of a chunk shorter than that: at least one row per wave where there is one)
            lds_barrier();
            const int step = nrows >= 4 * ss ? ss : 1;
            u32x4v bm;
            bm[0] = bm[1] = bm[2] = bm[3] = 0xffffffffu;
            int nval = 0;
            for (int r0 = w * step; r0 < nrows; r0 += 4 * step * U) {
                CodeWords<M> sc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) sc[u] = load_code_buf<M>(rs, (r0 + u * 4 * step) * 64 + lane);  // past the chunk: zeros, masked below
                u32x4v dd[U];
                adc16x8_rows<M, U, OCT>(sc, tab, rc, dd);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = (r0 + u * 4 * step) * 64 + lane < len;
                    nval += __popcll(__ballot(ok));
#pragma unroll
                    for (int x = 0; x < 4; ++x) bm[x] = pk_min_u16(bm[x], ok ? dd[u][x] : 0xffffffffu);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < ng) {
                    const int q = __builtin_amdgcn_readfirstlane(sd->q[g]);
                    const uint32_t sg = s5_sum_of(bm, g);
                    if (sg != 0xffffu) {
                        const float ubv = (float)(sg + (uint32_t)M) * sd->ubf[g] * 1.0000002f;  // >= the candidate's exact distance
                        atomicMin(&bmin[(int64_t)q * B + (((unsigned)j * 131u + (unsigned)tid) & (unsigned)(B - 1))], __float_as_uint(ubv));
                    }
                    if (lane == 0 && nval > 0) atomicAdd(&nsamp[q], nval);
                }
            }
            continue;
        } else {
        lds_barrier();  // tables and the descriptor's thresholds visible
        u32x4v tpk;  // the eight thresholds, packed like the sums
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const uint32_t a = sd->thr1[2 * x], b = sd->thr1[2 * x + 1];
            tpk[x] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((a & 0xffffu) | (b << 16)));
        }
        // ---- main pass: every candidate once, the positions where ANY query passes are recorded ---------------------------------
        uint16_t* plist = plist_all + (size_t)w * S5_PCAP;
        int pcur = 0;
        {
            const int nit = (len + 64 * U - 1) / (64 * U);
            CodeWords<M> ring[PF][U];
#pragma unroll
            for (int k = 0; k < PF; ++k) {
#pragma unroll
                for (int u = 0; u < U; ++u) ring[k][u] = load_code_buf<M>(rs, (w + k * NW) * 64 * U + u * 64 + lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int iter0 = w; iter0 < nit; iter0 += NW * PF) {
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const int iter = iter0 + k * NW;
                    const int base = iter * 64 * U;
                    u32x4v dd[U];
                    adc16x8_rows<M, U, OCT>(ring[k], tab, rc, dd);
#pragma unroll
                    for (int u = 0; u < U; ++u) ring[k][u] = load_code_buf<M>(rs, (iter + PF * NW) * 64 * U + u * 64 + lane);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t xx = (pk_subsat_u16(tpk[0], dd[u][0]) | pk_subsat_u16(tpk[1], dd[u][1])) |
                                            (pk_subsat_u16(tpk[2], dd[u][2]) | pk_subsat_u16(tpk[3], dd[u][3]));
                        unsigned long long am = __ballot(xx != 0u);
                        const int n = len - base - u * 64;
                        if (n < 64) am &= n <= 0 ? 0ull : ((1ull << n) - 1ull);
                        if (am == 0ull) continue;  // scalar branch
                        const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, pcur));
                        if (((am >> lane) & 1ull) && idx < S5_PCAP) plist[idx] = (uint16_t)(base + u * 64 + lane);
                        pcur += __popcll(am);
                    }
                }
            }
        }
        // ---- second gather of the recorded candidates: per-query split, counts, room, write-out --------------------------------------
        const bool pover = pcur > S5_PCAP;
        const int pn = pover ? 0 : pcur;
        if (pover) {
            if (lane < ng) ovf[sd->q[lane]] = 1;
        }
        if (pn > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own position stores
            const int nit2 = (pn + 63) >> 6;
            int cnt[G], okc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { cnt[g] = 0; okc[g] = 0; }
            uint32_t thr1[G], ok1[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                thr1[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sd->thr1[g]);
                ok1[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sd->ok1[g]);
            }
            for (int it = 0; it < nit2; ++it) {   // pass A: counts
                const bool on = it * 64 + lane < pn;
                const uint32_t pos = on ? (uint32_t)ld16(&plist[it * 64 + lane]) : (uint32_t)len;
                CodeWords<M> cc[1];
                cc[0] = load_code_buf<M>(rs, (int)pos);
                u32x4v d1[1];
                adc16x8_rows<M, 1, OCT>(cc, tab, rc, d1);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t sg = s5_sum_of(d1[0], g);
                    cnt[g] += __popcll(__ballot(on && sg < thr1[g]));
                    okc[g] += __popcll(__ballot(on && sg < ok1[g]));
                }
            }
            // room in the items' survivor rows: lane g reserves for query g (one atomic per wave and query), counts the proven ones
            int mine = cnt[0], mok = okc[0];
#pragma unroll
            for (int g = 1; g < G; ++g) { mine = (lane == g) ? cnt[g] : mine; mok = (lane == g) ? okc[g] : mok; }
            int base_l = 0;
            if (lane < ng && mine > 0) {
                base_l = atomicAdd(&item_n[sd->item[lane]], mine);
                if (base_l + mine > S) ovf[sd->q[lane]] = 1;
                if (mok > 0) atomicAdd(&cnt_ok[sd->q[lane]], mok);
            }
            int wbase[G];
#pragma unroll
            for (int g = 0; g < G; ++g) wbase[g] = __builtin_amdgcn_readlane(base_l, g);
            for (int it = 0; it < nit2; ++it) {   // pass B: the entries
                const bool on = it * 64 + lane < pn;
                const uint32_t pos = on ? (uint32_t)ld16(&plist[it * 64 + lane]) : (uint32_t)len;
                CodeWords<M> cc[1];
                cc[0] = load_code_buf<M>(rs, (int)pos);
                u32x4v d1[1];
                adc16x8_rows<M, 1, OCT>(cc, tab, rc, d1);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (g >= ng) continue;  // uniform
                    const uint32_t sg = s5_sum_of(d1[0], g);
                    const bool pass = on && sg < thr1[g];
                    const unsigned long long mg = __ballot(pass);
                    if (mg == 0ull) continue;
                    const int idx = __builtin_amdgcn_mbcnt_hi((unsigned)(mg >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mg, wbase[g]));
                    if (pass && idx < S) item_surv[(int64_t)__builtin_amdgcn_readfirstlane(sd->item[g]) * S + idx] = ((uint64_t)sg << 32) | pos;
                    wbase[g] += __popcll(mg);
                }
            }
        }
        }
    }
}

// tau[q] = the k-th smallest of the query's B bucket minima (float bits), k = P x sampled fraction; +inf when the query is short or
// the sample thin (everything is listed then).  One wave per query; also clears the query's counters for the main pass.
template <int PER>
__global__ __launch_bounds__(256) void k_scan5_tau(const uint32_t* __restrict__ bmin, int B, const int* __restrict__ nsamp, const PlanOut* __restrict__ plan,
                                                   int nq, int P, float* __restrict__ tau) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= nq) return;
    uint32_t v[PER];
    bool ok[PER];
    uint32_t lo = 0xffffffffu, hi = 0u;
    int fin = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = bmin[(int64_t)q * B + i * 64 + lane];
        ok[i] = v[i] < 0x7f800000u;
        lo = (ok[i] && v[i] < lo) ? v[i] : lo;
        hi = (ok[i] && v[i] > hi) ? v[i] : hi;
        fin += __popcll(__ballot(ok[i]));
    }
    const int64_t ncand = plan[q].ncand;
    const int ns = nsamp[q];
    float t = __uint_as_float(0x7f800000u);
    if (ns > 0 && ncand > 2 * (int64_t)P) {
        int k = (int)(((int64_t)P * ns + ncand - 1) / ncand);
        k = k < 4 ? 4 : k;
        if (k <= fin && k <= B / 2) {
            wave_minmax_step<1>(lo, hi); wave_minmax_step<2>(lo, hi); wave_minmax_step<4>(lo, hi);
            wave_minmax_step<8>(lo, hi); wave_minmax_step<16>(lo, hi); wave_minmax_step<32>(lo, hi);
            lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
            hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
            t = __uint_as_float(wave_kth_bisect<PER>(v, ok, lo, hi, k));
        }
    }
    if (lane == 0) tau[q] = t;
}

// the proof (see the header comment of k_adc_scan5): a thread per slot; a slot with a query that fails goes to the fall-back list
__global__ void k_scan5_check(const int* __restrict__ slots, const int* __restrict__ n_slots_ptr, const WorkItem* __restrict__ items,
                              const PlanOut* __restrict__ plan, const int* __restrict__ cnt_ok, const int* __restrict__ ovf, int L,
                              int* __restrict__ fhdr, int* __restrict__ fslots, int* __restrict__ item_n, int* __restrict__ dbg) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_slots_ptr[0]) return;
    int bad[S5G], nb = 0;
#pragma unroll
    for (int g = 0; g < S5G; ++g) {
        const int idx = slots[j * S5G + g];
        if (idx < 0) continue;
        const int q = items[idx].q;
        const int64_t nc = plan[q].ncand;
        const int need = nc < (int64_t)L ? (int)nc : L;
        if (ovf[q] != 0 || cnt_ok[q] < need) bad[nb++] = idx;
    }
    if (nb == 0) return;
    atomicAdd(&dbg[1], 1);
    for (int b0 = 0; b0 < nb; b0 += S3G) {  // fall-back slots hold S3G items
        const int f = atomicAdd(&fhdr[17], 1);
#pragma unroll
        for (int g = 0; g < S3G; ++g) {
            fslots[f * S3G + g] = b0 + g < nb ? bad[b0 + g] : -1;
            if (b0 + g < nb) item_n[bad[b0 + g]] = 0;  // whatever the slot wrote is replaced
        }
    }
}

__global__ void k_scan5_init(uint32_t* __restrict__ bmin, int64_t n_b, int* __restrict__ ints, int n_ints) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_b) bmin[i] = 0x7f800000u;
    if (i < n_ints) ints[i] = 0;
}

bool scan5_supported(int M, int K, int L) { return (M == 4 || M == 8) && K <= 256 && K % 4 == 0 && L >= 1 && L <= 440; }
static const int S5_B = 512;
size_t scan5_workspace_bytes(int nq) { return (size_t)nq * S5_B * 4 + (size_t)nq * 4 * 4 + 256; }

template <int M, int NR>
static void launch_scan5_t(const Scan3Geom& g, int64_t n_items, int nq, hipStream_t st, const WorkItem* items, const TabDesc* tabs, const int* slots,
                           const int* n_slots, const PlanOut* plan, const double* T, const float* T32, const uint8_t* codes, int K, int L, int* qctr,
                           uint64_t* hits, int* hitn, float* slack, unsigned long long* qbound, int* fhdr, int* fslots, void* ws, hipEvent_t ev_main) {
    uint32_t* bmin = reinterpret_cast<uint32_t*>(ws);
    int* ints = reinterpret_cast<int*>(bmin + (size_t)nq * S5_B);
    int* nsamp = ints;
    int* cnt_ok = ints + nq;
    int* ovf = ints + 2 * nq;
    float* tau = reinterpret_cast<float*>(ints + 3 * nq);
    const int64_t n_b = (int64_t)nq * S5_B;
    hipLaunchKernelGGL(k_scan5_init, dim3((unsigned)((n_b + 255) / 256)), dim3(256), 0, st, bmin, n_b, ints, 3 * nq);
    (void)hipMemsetAsync(hitn, 0, (size_t)(n_items + 1) * sizeof(int), st);
    constexpr int WPE = 4;
    const size_t lds = scan5_lds(M, K);
    const int by_lds = (int)(163840 / lds);
    const int per_cu = by_lds < WPE ? by_lds : WPE;
    const unsigned grid = 256u * (unsigned)(per_cu < 1 ? 1 : per_cu);
    static const int P_env = getenv("CIS_S5_P") ? atoi(getenv("CIS_S5_P")) : 0;
    static const int ss_env = getenv("CIS_S5_SS") ? atoi(getenv("CIS_S5_SS")) : 4;
    const int P = P_env > 0 ? P_env : (L < 100 ? 240 : (int)(2.4 * L));
    hipLaunchKernelGGL((k_adc_scan5<M, true, WPE>), dim3(grid), dim3(256), lds, st, items, tabs, slots, n_slots, T32, T, codes, K, g.S, (const float*)nullptr, bmin,
                       S5_B, nsamp, ss_env, hits, hitn, slack, cnt_ok, ovf);
    hipLaunchKernelGGL(k_scan5_tau<S5_B / 64>, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, bmin, S5_B, nsamp, plan, nq, P, tau);
    if (ev_main) (void)hipEventRecord(ev_main, st);
    hipLaunchKernelGGL((k_adc_scan5<M, false, WPE>), dim3(grid), dim3(256), lds, st, items, tabs, slots, n_slots, T32, T, codes, K, g.S, tau, bmin, S5_B, nsamp,
                       ss_env, hits, hitn, slack, cnt_ok, ovf);
    const int64_t max_slots = n_items + 8;  // (an upper bound of the slot count: the kernel reads the real one)
    hipLaunchKernelGGL(k_scan5_check, dim3((unsigned)((max_slots + 255) / 256)), dim3(256), 0, st, slots, n_slots, items, plan, cnt_ok, ovf, L, fhdr, fslots, hitn,
                       qctr + 9);
    if (getenv("CIS_SCAN5_DEBUG")) {
        int h[6] = {0, 0, 0, 0, 0, 0};
        if (hipMemcpyAsync(h, qctr + 9, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
            fprintf(stderr, "[cis] k_adc_scan5: %d slots to the fall-back list\n", h[1]);
    }
    // the slots it could not settle: k_adc_scan3's two-pass form for short chunks, its streaming form for long ones (as after k_adc_scan4)
    constexpr int U = 4, NW = 4;
    constexpr int WPE3 = 4;
    hipLaunchKernelGGL((k_adc_scan3<M, NR, U, NW, WPE3>), dim3(256), dim3(NW * 64), g.lds, st, items, tabs, fslots, fhdr + 8, T, T32, codes, K, L, g.S, fhdr, hits, hitn,
                       slack, qbound, g.long_chunks ? 0 : 1, 1);
}

void launch_scan5(int M, const Scan3Geom& g, int64_t n_items, int nq, hipStream_t st, const WorkItem* items, const TabDesc* tabs, const int* slots,
                  const int* n_slots, const PlanOut* plan, const double* T, const float* T32, const uint8_t* codes, int K, int L, int* qctr, uint64_t* hits,
                  int* hitn, float* slack, unsigned long long* qbound, int* fhdr, int* fslots, void* ws, hipEvent_t ev_main) {
    if (M == 4) {
        if (L <= 184) launch_scan5_t<4, 4>(g, n_items, nq, st, items, tabs, slots, n_slots, plan, T, T32, codes, K, L, qctr, hits, hitn, slack, qbound, fhdr, fslots, ws, ev_main);
        else launch_scan5_t<4, 8>(g, n_items, nq, st, items, tabs, slots, n_slots, plan, T, T32, codes, K, L, qctr, hits, hitn, slack, qbound, fhdr, fslots, ws, ev_main);
    } else {
        if (L <= 184) launch_scan5_t<8, 4>(g, n_items, nq, st, items, tabs, slots, n_slots, plan, T, T32, codes, K, L, qctr, hits, hitn, slack, qbound, fhdr, fslots, ws, ev_main);
        else launch_scan5_t<8, 8>(g, n_items, nq, st, items, tabs, slots, n_slots, plan, T, T32, codes, K, L, qctr, hits, hitn, slack, qbound, fhdr, fslots, ws, ev_main);
    }
}

bool scan3_supported(int M, int K, int L) {
    return (M == 4 || M == 8 || M == 16) && K <= 256 && K % 4 == 0 && L >= 1 && L <= 440;
}

// NW = waves per workgroup: every wave of a workgroup pays one compaction per query when its region first fills and
// one at the end of the chunk, so short chunks want few waves (less of the chunk's time goes to selection), long ones
// four (more waves share one 16 KB table set).
Scan3Geom scan3_geom(int M, int K, int L, int64_t avg_chunk, int force_two_pass) {
    Scan3Geom g;
    const int NR = (L <= 184) ? 4 : 8;
    g.G = S3G;
    g.NW = 4;  // measured (profiles/r02c_scan3_nw.txt): 2 and 1 waves lose more to latency than they save in selections
    (void)avg_chunk;
    g.U = 4;
    // short chunks (a few thousand candidates): the two-pass form; CIS_SCAN3_TWOPASS=0/1 overrides (A/B runs)
    // 2 = the sampled single-pass form (k_adc_scan4), 1 = the two-pass form; CIS_SCAN3_TWOPASS=0/1/2 overrides (A/B runs)
    g.two_pass = (avg_chunk > 0 && avg_chunk < 12288 && M <= 8) ? 1 : 0;
    // the sampled form's lists hold 376 sums per query: chunks of a few thousand candidates at limit <= 128 (a looser sample
    // threshold overflows them and the slot is scanned again by the two-pass form)
    if (g.two_pass == 1 && L <= 128 && avg_chunk < 6144) g.two_pass = 2;
    // long chunks (tens of thousands of candidates): the sampled form as well, with a tenth of the rows as the sample, lists
    // of 1016 entries (the sample threshold lets ~4 L candidates through) and a wider rank margin (a list that fails its
    // verification costs a whole streaming-form slot in the fall-back launch)
    g.long_chunks = avg_chunk >= 6144 ? 1 : 0;
    // the sampled form at M = 16: with the tables' largest entries as the scale the + M + 1 slack was ~17 % of the cut and 84 % of the
    // lists overflowed (8.99 against 0.60 ms for k_adc_scan2); with the SATURATING scale of k_adc_scan4 (Scan3Geom::sat) no slot of C3
    // falls back and the scan takes 0.474 ms (profiles/r04v_m16.txt).  CIS_S4_M16=0 keeps M = 16 on k_adc_scan2.
    static const int s4_m16 = getenv("CIS_S4_M16") ? atoi(getenv("CIS_S4_M16")) : 1;
    if (g.two_pass == 0 && g.long_chunks && (M <= 8 || s4_m16) && L <= 128) g.two_pass = 2;
    if (const char* e = getenv("CIS_SCAN3_TWOPASS")) g.two_pass = atoi(e);
    if (force_two_pass >= 0) g.two_pass = force_two_pass;  // scan modes 3 / 4 / 5 (tests)
    if (g.two_pass == 2 && M > 8 && !s4_m16 && force_two_pass != 2) g.two_pass = g.long_chunks ? 0 : 1;
    if (g.two_pass == 1 && g.long_chunks && force_two_pass < 0) g.two_pass = 0;  // the histogram atomics of the two-pass form contend on long chunks
    if (g.two_pass < 0 || g.two_pass > 2) g.two_pass = 0;
    g.S = g.NW * (NR * 64 - 8);
    g.lds = (size_t)K * M * g.G * 2 + (size_t)g.G * g.NW * (NR * 64 - 8) * 4 + g.G * sizeof(Scan3Shared) + 32 + 16;
    return g;
}

template <int M, int NR, int NW>
static void launch_scan3_t(const Scan3Geom& g, int64_t n_items, hipStream_t st, const WorkItem* items, const TabDesc* tabs,
                           const int* slots, const int* n_slots, const double* T, const float* T32, const uint8_t* codes, int K,
                           int L, int* qctr, uint64_t* hits, int* hitn, float* slack, unsigned long long* qbound, int* fhdr, int* fslots) {
    // waves per SIMD the kernel is compiled for (register budget): what the LDS footprint lets a CU hold anyway
    constexpr int U = 4;  // (U = 2 at 5 waves per SIMD measured slower: 0.535 against 0.497 ms on c4)
    constexpr int WPE = M == 16 ? (NW == 4 ? 3 : 2) : (NW == 4 ? 4 : (NW == 2 ? 3 : 2));
    const int by_lds = (int)(163840 / g.lds), by_waves = (WPE * 4) / NW;
    const int per_cu = by_lds < by_waves ? by_lds : by_waves;
    const int64_t resident = 256 * (per_cu < 1 ? 1 : per_cu);  // persistent grid: what the chip can hold
    const int64_t want = (n_items + g.G - 1) / g.G + 8;
    const unsigned grid = (unsigned)(want < resident ? ((want + 7) / 8) * 8 : resident);
    if constexpr (NW == 4) {
        if (g.two_pass == 2) {
            // k_adc_scan4, then the slots it could not settle (normally none) through this kernel's two-pass form: the fall-back
            // list is a second slot header (fhdr: queue counters [0..7], queue starts [16..24] of which only [17] = count is used)
            // read per call (A/B runs and the fall-back tests set them between calls)
            const float z_short = getenv("CIS_S4_ZS") ? (float)atof(getenv("CIS_S4_ZS")) : CIS_S4_Z;
            // long chunks, measured on C4 (profiles/r03m_scan_long.txt): z 4 / 4.5 / 5 -> 0.326 / 0.332 / 0.335 ms, no verification
            // failure in 100 k lists at z = 4; static schedule 0.301 against 0.326 ms with slots from a counter; one row in 8 / 10 /
            // 16 as the sample -> 0.321 / 0.326 / 0.45 ms; at most 128 sample rows (64: the 65536-candidate chunks overflow their lists)
            const float z_long = getenv("CIS_S4_ZL") ? (float)atof(getenv("CIS_S4_ZL")) : 4.5f;
            const int frac_long = getenv("CIS_S4_FRAC") ? atoi(getenv("CIS_S4_FRAC")) : 8;
            const int nsx_long = getenv("CIS_S4_NSX") ? atoi(getenv("CIS_S4_NSX")) : 128;
            const int dyn_long = getenv("CIS_S4_DYN") ? atoi(getenv("CIS_S4_DYN")) : 0;
            int* dbg4 = qctr + 9;
            // waves per slot (NW4) and waves per SIMD the variant is compiled for (WPS): 4 x 64 threads at 4 (long chunks) / 6 (short) waves
            // per SIMD were rounds 2-3; 8 waves per slot halve a slot's main pass, and three such workgroups per CU are 24 waves
            const int nw4_long = getenv("CIS_S4_NWL") ? atoi(getenv("CIS_S4_NWL")) : CIS_S4_NW_LONG;
            const int nw4_short = getenv("CIS_S4_NWS") ? atoi(getenv("CIS_S4_NWS")) : CIS_S4_NW_SHORT;
            auto grid_of = [&](size_t lds4, int wps, int nw4) -> unsigned {
                const int by_lds4 = (int)(163840 / lds4), by_waves4 = (wps * 4) / nw4;
                int per_cu4 = by_lds4 < by_waves4 ? by_lds4 : by_waves4;
                if (const char* e = getenv("CIS_S4_PER_CU")) per_cu4 = atoi(e) > 0 && atoi(e) < per_cu4 ? atoi(e) : per_cu4;  // A/B: room for other batches' kernels
                const int64_t resident4 = 256 * (per_cu4 < 1 ? 1 : per_cu4);
                return (unsigned)(want < resident4 ? ((want + 7) / 8) * 8 : resident4);
            };
            if (!g.long_chunks) {
                constexpr int WPE4 = (M == 16) ? 4 : CIS_S4_WPE;  // (M = 16: 41 KB of LDS hold three workgroups per CU anyway)
                if (nw4_short == 8 && M != 16) {
                    const size_t lds4 = scan4_lds(M, K, S4_LCAP, 8);
                    hipLaunchKernelGGL((k_adc_scan4<M, CIS_S4_U, 8, WPE4, S4_LCAP>), dim3(grid_of(lds4, WPE4, 8)), dim3(8 * 64), lds4, st, items, tabs, slots,
                                       n_slots, T32, T, codes, K, L, g.S, dbg4, fhdr, fslots, hits, hitn, slack, qbound, z_short, 1 << 20, S4_NS, 0, g.sat);
                } else {
                    const size_t lds4 = scan4_lds(M, K, S4_LCAP, NW);
                    hipLaunchKernelGGL((k_adc_scan4<M, CIS_S4_U, NW, WPE4, S4_LCAP>), dim3(grid_of(lds4, WPE4, NW)), dim3(NW * 64), lds4, st, items, tabs, slots,
                                       n_slots, T32, T, codes, K, L, g.S, dbg4, fhdr, fslots, hits, hitn, slack, qbound, z_short, 1 << 20, S4_NS, 0, g.sat);
                }
            } else {
                constexpr int WPE4 = (M == 16) ? 3 : ((CIS_S4_DEFER != 0 && CIS_S4_GLISTS != 0) ? CIS_S4_WPE_LONG : 4);  // (M = 16: 49 KB of LDS = three workgroups per CU, 168 registers)
                if (nw4_long == 8 && M != 16) {
                    constexpr int WPE8 = CIS_S4_WPE8;
                    const size_t lds4 = scan4_lds(M, K, S4_LCAP_LONG, 8);
                    hipLaunchKernelGGL((k_adc_scan4<M, CIS_S4_UL, 8, WPE8, S4_LCAP_LONG>), dim3(grid_of(lds4, WPE8, 8)), dim3(8 * 64), lds4, st, items, tabs, slots,
                                       n_slots, T32, T, codes, K, L, g.S, dbg4, fhdr, fslots, hits, hitn, slack, qbound, z_long, frac_long, nsx_long, dyn_long, g.sat);
                } else {
                    const size_t lds4 = scan4_lds(M, K, S4_LCAP_LONG, NW);
                    hipLaunchKernelGGL((k_adc_scan4<M, CIS_S4_UL, NW, WPE4, S4_LCAP_LONG>), dim3(grid_of(lds4, WPE4, NW)), dim3(NW * 64), lds4, st, items, tabs, slots,
                                       n_slots, T32, T, codes, K, L, g.S, dbg4, fhdr, fslots, hits, hitn, slack, qbound, z_long, frac_long, nsx_long, dyn_long, g.sat);
                }
            }
            if (getenv("CIS_SCAN4_DEBUG")) {  // diagnosis: how many slots the sample misjudged (blocks)
                int h[6] = {0, 0, 0, 0, 0, 0};
                if (hipMemcpyAsync(h, qctr + 9, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
                    fprintf(stderr, "[cis] k_adc_scan4: %d slots, %d to the fall-back list (lists: %d overflowed, %d failed the verification, %d crowds at the cut; %d slots with items of different chunks)\n",
                            h[0], h[1], h[2], h[3], h[4], h[5]);
            }
            // the slots it could not settle: the two-pass form for short chunks, the streaming form for long ones
            hipLaunchKernelGGL((k_adc_scan3<M, NR, U, NW, WPE>), dim3(256), dim3(NW * 64), g.lds, st, items, tabs, fslots, fhdr + 8, T, T32,
                               codes, K, L, g.S, fhdr, hits, hitn, slack, qbound, g.long_chunks ? 0 : 1, 1);
            return;
        }
    }
    hipLaunchKernelGGL((k_adc_scan3<M, NR, U, NW, WPE>), dim3(grid), dim3(NW * 64), g.lds, st, items, tabs, slots, n_slots, T, T32,
                       codes, K, L, g.S, qctr, hits, hitn, slack, qbound, g.two_pass, 0);
}

void launch_scan3(int M, const Scan3Geom& g, int64_t n_items, hipStream_t st, const WorkItem* items, const TabDesc* tabs,
                  const int* slots, const int* n_slots, const double* T, const float* T32, const uint8_t* codes, int K, int L,
                  int* qctr, uint64_t* hits, int* hitn, float* slack, unsigned long long* qbound, int* fhdr, int* fslots) {
#define CIS_S3_NW(MM, RR) launch_scan3_t<MM, RR, 4>(g, n_items, st, items, tabs, slots, n_slots, T, T32, codes, K, L, qctr, hits, hitn, slack, qbound, fhdr, fslots)
#define CIS_S3(MM)                   \
    do {                             \
        if (L <= 184) CIS_S3_NW(MM, 4); \
        else CIS_S3_NW(MM, 8);       \
    } while (0)
    if (M == 4) CIS_S3(4);
    else if (M == 8) CIS_S3(8);
    else CIS_S3(16);
#undef CIS_S3
#undef CIS_S3_NW
}
