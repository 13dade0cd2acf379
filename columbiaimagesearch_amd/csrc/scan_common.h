// Device helpers shared by the ADC scan kernels (lopq_search.hip: float32-prefilter scan; lopq_scan3.hip: 16-bit
// fixed-point scan): work-item layout, wave-level primitives on the VALU only, code loads, exact re-scoring.
#pragma once
#include "lopq_model.h"

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

// value of lane (l ^ LJ) for every lane l, on the VALU only (DPP / permlane swaps): the LDS pipe is the
// scan's bottleneck, so the in-register sorts must not use ds_bpermute.
template <int LJ>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (LJ == 1) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    } else if constexpr (LJ == 2) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    } else if constexpr (LJ == 4) {
        const int a = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);      // row_half_mirror: l ^ 7
        return (uint32_t)__builtin_amdgcn_update_dpp(0, a, 0x1B, 0xF, 0xF, false);        // quad_perm [3,2,1,0]: ^ 3
    } else if constexpr (LJ == 8) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8 == l ^ 8 in a row of 16
    } else if constexpr (LJ == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // .x = even rows twice, .y = odd rows twice
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        static_assert(LJ == 32, "lane_xor: LJ must be a power of two below 64");
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // .x = low half twice, .y = high half twice
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}

template <int NR, int KK, int J>
__device__ __forceinline__ void bitonic_step(uint32_t (&k)[NR]) {
    const int lane = threadIdx.x & 63;
    if constexpr (J < NR) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if ((r & J) == 0) {
                const bool asc = (((lane * NR + r) & KK) == 0);
                const uint32_t a = k[r], b = k[r | J];
                const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
                k[r] = asc ? lo : hi;
                k[r | J] = asc ? hi : lo;
            }
        }
    } else {
        constexpr int LJ = J / NR;
        const bool lower = ((lane & LJ) == 0);
        const bool asc = (((lane * NR) & KK) == 0);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const uint32_t o = lane_xor<LJ>(k[r]);
            const uint32_t mn = k[r] < o ? k[r] : o, mx = k[r] < o ? o : k[r];
            k[r] = (lower == asc) ? mn : mx;
        }
    }
}

template <int NR, int KK, int J>
__device__ __forceinline__ void bitonic_merge(uint32_t (&k)[NR]) {
    bitonic_step<NR, KK, J>(k);
    if constexpr (J > 1) bitonic_merge<NR, KK, J / 2>(k);
}

template <int NR, int KK>
__device__ __forceinline__ void bitonic_levels(uint32_t (&k)[NR]) {
    if constexpr (KK > 2) bitonic_levels<NR, KK / 2>(k);
    bitonic_merge<NR, KK, KK / 2>(k);
}

// ascending sort of the NR*64 keys of a wave, element e = lane*NR + r
template <int NR>
__device__ __forceinline__ void wave_bitonic_sort(uint32_t (&k)[NR]) {
    bitonic_levels<NR, NR * 64>(k);
}

template <int LJ>
__device__ __forceinline__ void wave_minmax_step(uint32_t& mn, uint32_t& mx) {
    const uint32_t a = lane_xor<LJ>(mn), b = lane_xor<LJ>(mx);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
}

// Smallest v with  #{valid keys <= v} >= target  == the target-th smallest key (1-based), found by
// bisection on the value range with ballots: ~4 VALU compares per probe instead of a ~650-instruction
// register sort.  lo/hi must bracket the answer (lo = min key, hi = max key is always fine).
template <int NR>
__device__ __forceinline__ uint32_t wave_kth_bisect(const uint32_t (&key)[NR], const bool (&valid)[NR], uint32_t lo,
                                                    uint32_t hi, int target) {
    while (lo < hi) {  // wave-uniform
        const uint32_t p = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) c += __popcll(__ballot(valid[r] && key[r] <= p));
        if (c >= target) hi = p;
        else lo = p + 1;
    }
    return lo;
}

template <int NR>
__device__ __forceinline__ uint32_t wave_kth(const uint32_t (&k)[NR], int i) {  // i-th smallest after the sort
    uint32_t v = k[0];
#pragma unroll
    for (int r = 1; r < NR; ++r)
        if ((i % NR) == r) v = k[r];
    return (uint32_t)__builtin_amdgcn_readlane((int)v, i / NR);
}

static __device__ __forceinline__ float lds_ld(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ uint64_t lds_ld(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void lds_st(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void lds_st(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// exact float64 distance from the code words (little-endian bytes = fine codes 0..M-1)
template <int M>
static __device__ __forceinline__ double adc64_words(const uint32_t (&cw)[(M + 3) / 4], int K, const double* __restrict__ t0,
                                                     const double* __restrict__ t1) {
    constexpr int nf = M / 2;
    double f[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const uint32_t c = (cw[j >> 2] >> (8 * (j & 3))) & 255u;
        f[j] = (j < nf) ? t0[j * K + c] : t1[(j - nf) * K + c];   // (non-temporal loads here: the merge 0.104 -> 0.119 ms on C2, round 6)
    }
    double d = f[0];
#pragma unroll
    for (int j = 1; j < M; ++j) d = d + f[j];
    return d;
}

// hi word -> the largest float64 bit pattern with that hi word (a distance no smaller than any
// distance whose bits start with `vhi`); infinities stay infinite
static __device__ __forceinline__ uint64_t hi_to_bound(uint32_t vhi) {
    return vhi >= 0x7ff00000u ? 0x7ff0000000000000ull : (((uint64_t)vhi << 32) | 0xffffffffull);
}

// one candidate's code as 32-bit words (little-endian bytes = fine codes 0..M-1)
template <int M>
struct CodeWords { uint32_t w[(M + 3) / 4]; };

template <int M>
__device__ __forceinline__ CodeWords<M> load_code(const uint8_t* __restrict__ codes, int64_t p) {
    CodeWords<M> c;
    if constexpr (M == 4) {
        c.w[0] = *reinterpret_cast<const uint32_t*>(codes + p * 4);
    } else if constexpr (M == 8) {
        const uint2 v = *reinterpret_cast<const uint2*>(codes + p * 8);
        c.w[0] = v.x; c.w[1] = v.y;
    } else {
        const uint4 v = *reinterpret_cast<const uint4*>(codes + p * 16);
        c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
    }
    return c;
}

template <int M>
struct RotConsts {
    uint32_t sh[4];   // bit offset of the byte used at sub-step tq inside the selected dword
    uint32_t cj[M];   // (table index << 2) for step t
    uint32_t hsel;    // which dword this lane starts with
};

template <int M>
__device__ __forceinline__ RotConsts<M> make_rot(int lane) {
    RotConsts<M> rc;
    const int r = lane & (M - 1);
    const int h = r >> 2, q = r & 3;
    rc.hsel = (uint32_t)h;
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) rc.sh[tq] = 8u * (uint32_t)((q + tq) & 3);
#pragma unroll
    for (int t = 0; t < M; ++t) {
        const int th = t >> 2, tq = t & 3;
        const int j = ((h ^ th) << 2) | ((q + tq) & 3);
        rc.cj[t] = (uint32_t)j << 2;
#ifdef CIS_SCAN_OPAQUE_CJ
        asm volatile("" : "+v"(rc.cj[t]));  // keep the M offsets in M registers (else the compiler re-derives half of them per use)
#endif
    }
    return rc;
}

// the same through a buffer descriptor that covers exactly the chunk being scanned: one 32-bit offset per
// load instead of 64-bit address arithmetic, and positions past the end of the chunk read as zero
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int M>
__device__ __forceinline__ CodeWords<M> load_code_buf(__amdgpu_buffer_rsrc_t rs, int p) {
    CodeWords<M> c;
    if constexpr (M == 4) {
        c.w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, p * 4, 0, 0);
    } else if constexpr (M == 8) {
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, p * 8, 0, 0);
        c.w[0] = v[0]; c.w[1] = v[1];
    } else {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, p * 16, 0, 0);
        c.w[0] = v[0]; c.w[1] = v[1]; c.w[2] = v[2]; c.w[3] = v[3];
    }
    return c;
}

// Four consecutive float32 table entries (float4 number e4 of half table `tab`, nfK entries per half table): from the float32 copy when
// the batch has one, else converted from the float64 tables (round 5: the copy is a third of the tables kernel's writes -- 402 MB per C2
// batch -- and (float)T is the value the copy holds, so the scans see the same bits either way).
static __device__ __forceinline__ float4 tab_f4(const float* __restrict__ T32, const double* __restrict__ T, int64_t tab, int nfK, int e4) {
    if (T32) return reinterpret_cast<const float4*>(T32 + tab * nfK)[e4];   // (non-temporal loads here: no effect, round 6 same-box A/B)
    const double2* p = reinterpret_cast<const double2*>(T + tab * nfK) + 2 * e4;
    const double2 a = p[0], b = p[1];
    return make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
}
static __device__ __forceinline__ float tab_f1(const float* __restrict__ T32, const double* __restrict__ T, int64_t idx) {
    return T32 ? T32[idx] : (float)T[idx];
}

// ---- ADC scan v3 (lopq_scan3.hip): 16-bit fixed-point tables, four queries per workgroup ---------------------------
struct Scan3Geom { int G, NW, U, S, two_pass, long_chunks; size_t lds;
                   float sat = 0.f; };  // > 0: the sampled form's SATURATING scale (M = 16) -- see k_adc_scan4
bool scan3_supported(int M, int K, int L);
Scan3Geom scan3_geom(int M, int K, int L, int64_t avg_chunk /* candidates per work item of the batch */,
                     int force_two_pass /* -1: by chunk length, 0: streaming form, 1: two-pass form */);
void launch_scan3(int M, const Scan3Geom& g, int64_t n_items, hipStream_t st, const WorkItem* items, const TabDesc* tabs,
                  const int* slots, const int* n_slots, const double* T, const float* T32, const uint8_t* codes, int K, int L,
                  int* qctr, uint64_t* hits, int* hitn, float* slack, unsigned long long* qbound,
                  int* fhdr /* [32] zeroed: fall-back slot header of the sampled form */, int* fslots /* [n_slots * G] */);

// ---- the HBM-streaming scan (lopq_stream.hip): few queries, very many candidates each --------------------------------------
static const int STREAM_B = 4096;     // buckets of sample minima per query (k_stream_tau: 1024 threads x 4; the k-th smallest of them, k <= B / 4, bounds the k-th smallest sample)
static const int STREAM_CAP = 16384;  // listed candidates per query at most
bool stream_supported(int M, int K, int L);
int stream_grid(int M, int G, int K, int64_t max_rows);
int stream_max_group();  // queries per slot at most (1, 2 or 4 are instantiated)
size_t stream_slot_bytes();  // one record per slot (lopq_stream.hip: StreamSlot), written by launch_stream_init
void launch_stream_prep(hipStream_t st, const WorkItem* items, int64_t n_items, const int64_t* item_off, int nq, int64_t n_cand,
                        const int* slots /* null: slot i = work item i alone */, int* n_slots, int G, int M, int64_t* cand_start, int64_t* seg,
                        unsigned long long* qmin, unsigned long long* qmax, int* cnt, int* status, int64_t* rowoff /* [n_slots + 1]: rows of the slots before each */,
                        void* desc /* [max_slots] records */, const int64_t* d_totals /* null, or the plan totals: n_items and n_cand are bounds */);
void launch_stream_scan(int M, int G, bool sample, int grid, hipStream_t st, const void* desc, const int* n_slots, const int64_t* rowoff,
                        const float* T32, const double* T, const uint8_t* codes, int K, const float* tau,
                        uint32_t* bmin, int B, int sample_stride, int flush /* sampled rows a lane folds into one bucket */, uint32_t* surv, int* cnt, int cap);
void launch_stream_tau(hipStream_t st, uint32_t* bmin /* read, then reset */, int B, int k, int nq, float* tau);
void launch_stream_keys(int M, hipStream_t st, const WorkItem* items, const int64_t* cand_start, const int64_t* seg, const int64_t* item_off,
                        int64_t n_items, const double* T, const uint8_t* codes, int K, const uint32_t* surv, const int* cnt, int cap, int nq,
                        uint64_t* keys, unsigned long long* qmin, unsigned long long* qmax);
void launch_stream_finish(hipStream_t st, const uint64_t* sel_keys, const uint64_t* sel_vals, const int* nsel, int64_t stride, const int* cnt, int cap,
                          const int64_t* seg, const float* tau, int nq, int L, int M, const WorkItem* items, const int64_t* ids, const PlanOut* plan,
                          cis_hit* out_hits, int64_t* out_ids, double* out_dists, int* out_n, int32_t* out_cells, uint32_t* out_pos, int32_t* out_visited,
                          int* status, int64_t* status_host_dev, int64_t seq);

// ---- k_adc_scan5 (lopq_scan3.hip): one threshold per query for the whole batch, eight queries per slot -----------------------
bool scan5_supported(int M, int K, int L);
size_t scan5_workspace_bytes(int nq);
void launch_scan5(int M, const Scan3Geom& g, int64_t n_items, int nq, hipStream_t st, const WorkItem* items, const TabDesc* tabs, const int* slots,
                  const int* n_slots, const PlanOut* plan, const double* T, const float* T32, const uint8_t* codes, int K, int L, int* qctr, uint64_t* hits,
                  int* hitn, float* slack, unsigned long long* qbound, int* fhdr, int* fslots, void* ws, hipEvent_t ev_main);
