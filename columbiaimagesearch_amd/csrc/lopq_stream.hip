// The HBM-streaming ADC scan: few queries, very many candidates each (an exhaustive quota = N run, SURVEY.md section 8(d)'s
// "full scan, roofline run", or any quota on an index whose cells hold hundreds of thousands of codes).
//
// Reference: lopq/lopq/search.py:128-133 (whole cells until the quota), :137-177 (compute_distances: dist = sum of M table
// entries), :210-216 (stable sorted()[:limit]).  In this regime every code byte is needed ONCE per query (pair), the codes
// exceed L2 and the Infinity Cache, and the bound is the HBM stream itself: algorithmic bytes = candidates x M per launch
// (per pair of queries when two share a launch), and they are also the physical minimum.
//
// The batch kernels (k_adc_scan2 / 3 / 4) keep a running top-`limit` per work item and hand its survivors to a one-wave-per-query
// merge: with ~50 000 work items per query that merge took 40 ms of the 42 ms of a single exhaustive query over 200 M codes.
// Here nothing is ranked inside the stream (one launch each; k_stream_prep lays the batch out first: candidate offsets, one record per
// slot, the rows of the slots before each):
//   1. k_adc_stream<SAMPLE>: every SS-th row of every chunk; a lane folds the MINIMUM float32 distance of a block of its sampled rows
//      into one of B buckets per query (atomicMin; the bucket is a hash of (slot, row block, thread)).  Any partition of the samples
//      into buckets gives count(buckets <= v) <= count(samples <= v), so the k-th smallest bucket minimum is an upper bound of the
//      k-th smallest sample distance;
//   2. k_stream_tau: per query, tau = the k-th smallest bucket minimum -- about k x SS candidates of the whole query lie below it
//      (k is chosen so that this is a few thousand, ~20 x limit); it resets the buckets it reads;
//   3. k_adc_stream: the stream.  Codes arrive as 16 bytes per lane, the float32 tables of the slot's (<= G) queries sit in LDS
//      entry-major, REPLICATED so that a gather meets no bank conflict, with the sub-quantizers rotated over the lanes; a candidate
//      costs M gathers + M adds and ONE compare against tau; the few that pass append their retrieval index to the query's list (one
//      atomic each);
//   4. k_stream_keys: the exact float64 distance of every listed candidate (left-to-right sum of the float64 entries: the bits of
//      every other route), k_select_topl ranks them by (distance, retrieval index), k_stream_finish PROVES the result and writes it:
//          let B = the limit-th smallest exact distance in the list (the list holds >= limit candidates, or every candidate);
//          |d32 - d64| <= eps d64 (eps = 2 M 2^-24: M rounded entries, M - 1 rounded adds), so a candidate of the true top
//          `limit` has d64 <= B, hence d32 <= B (1 + eps); if B (1 + 2 eps) <= tau it was listed.  All candidates with d64 <= B are
//          then in the list, ties included, and ranking the list by (d64, retrieval index) is the reference's stable sort.
//      A query whose proof fails (sample unlucky), whose list overflowed (a crowd of equal codes) or whose cut sits in more exact ties
//      than k_select_topl ranks is flagged in pinned memory; the host then answers the batch through the generic path.  The result
//      never depends on the sample.
#include "scan_common.h"

// Geometry of the stream (round 6; every choice below is a measurement of tools/probes/stream_probe.hip on 200 M x 8-byte codes,
// profiles/r06_stream_probe.txt -- the round-5 kernel is that probe's "slots round-robin" form):
//   * the ROWS (one 16-byte load per lane = 1 KB per wave) of all slots are cut into one equal range per workgroup: the launch ends
//     together whatever the number and the length of the slots (round 5 dealt whole slots: 2.3 rounds, the last a third full);
//   * ONE load per wave and iteration, the next iteration's requested before the current is used, NON-TEMPORAL (each byte is used once:
//     the bare read of the same structure goes from 0.74-0.78 to 0.84 of 8 TB/s with it, and only with one load per wave);
//   * the tables REPLICATED in LDS so that a gather meets no bank conflict (32 / (M G) copies of an entry row, rows of 128 bytes: the
//     32 lanes of a read group own distinct banks whatever their bytes), 32 KB per workgroup, staged with 16-byte stores;
//   * eight waves per workgroup, two workgroups per CU: half the stagings of four-wave workgroups, and a CU never idles on one
//     workgroup's barrier.
#ifndef CIS_STREAM_NW
#define CIS_STREAM_NW 8     // waves per workgroup
#endif
#ifndef CIS_STREAM_PER_CU
#define CIS_STREAM_PER_CU 2 // resident workgroups per CU
#endif
#ifndef CIS_STREAM_AUX
#define CIS_STREAM_AUX 2    // cache policy of the code loads: 2 = nt
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// copies of an entry row: what makes a gather conflict-free -- a ds_read_b32 (G = 1) serves 32 lanes per cycle, ds_read_b64 / b128
// (G = 2 / 4) measured best with 16 lanes' worth ((copy, sub-quantizer) pairs distinct inside a 16-lane group); rows of 128 or 256 bytes
constexpr int stream_copies(int M, int G) { return ((G == 1 ? 32 : 16) / M) > 0 ? ((G == 1 ? 32 : 16) / M) : 1; }

// What the stream needs to know of a slot, in one record (one scalar load instead of slots -> items -> cand_start / seg hops at every piece)
struct StreamSlot {
    int64_t start;      // first candidate (position in codes)
    int len;            // candidates of the chunk; 0: empty slot
    int pad;
    int q[4];           // query of each item, -1: none
    int tab0[4], tab1[4];
    uint32_t rbase[4];  // retrieval index of the chunk's first candidate, relative to the query's first candidate
};

// Everything between the plan and the stream in ONE launch of one workgroup (round 6; before: k_cand_layout, k_stream_init and -- for one
// query per slot -- the four launches of the slot builder, ~5 us each for a few hundred work items):
//   A. cand_start[i] = candidates of the work items before i (retrieval order), seg[q] = the query's first candidate, key ranges reset
//      (what k_cand_layout does on the all-candidates path);
//   B. a StreamSlot per slot; slots == null: slot i = work item i alone (G = 1 needs no grouping);
//   C. rowoff[s] = rows of the slots before s (a slot's rows = ceil(len / ROW)), rowoff[n_slots] = all rows;
//   D. the list counters and status words reset.  (The sample buckets are left clean by their reader, k_stream_tau.)
// Work items by the thousand are walked in rounds of 1024: a batch of this route has few (one chunk per visited cell and query).
__device__ __forceinline__ int64_t prep_block_excl_scan(int64_t x, int64_t* s_w, int64_t* total) {  // two barriers
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t y = __shfl_up(inc, d);
        if (lane >= d) inc += y;
    }
    __syncthreads();
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int64_t v = s_w[w];
        if (w < wv) base += v;
        tot += v;
    }
    *total = tot;
    return base + inc - x;
}

__global__ __launch_bounds__(1024) void k_stream_prep(const WorkItem* __restrict__ items, int64_t n_items, const int64_t* __restrict__ item_off, int nq, int64_t n_cand,
                                                      const int* __restrict__ slots, int* __restrict__ n_slots, int G, int row,
                                                      int64_t* cand_start, int64_t* seg, unsigned long long* __restrict__ qmin, unsigned long long* __restrict__ qmax,
                                                      int* __restrict__ cnt, int* __restrict__ status, int64_t* __restrict__ rowoff, StreamSlot* __restrict__ desc,
                                                      const int64_t* __restrict__ d_totals /* null, or the plan totals (n_items and n_cand are bounds) */) {
    __shared__ int64_t s_w[16];
    const int tid = threadIdx.x;
    if (d_totals) { n_items = d_totals[0]; n_cand = d_totals[2]; }
    // A. (cand_start and seg are written and read by this workgroup through agent-scope atomics: no stale L1 line, as in k_cand_layout)
    int64_t run = 0;
    for (int64_t i0 = 0; i0 < n_items; i0 += 1024) {
        const int64_t i = i0 + tid;
        const int64_t x = i < n_items ? (int64_t)items[i].len : 0;
        int64_t tot;
        const int64_t ex = prep_block_excl_scan(x, s_w, &tot);
        if (i < n_items) __hip_atomic_store(&cand_start[i], run + ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run += tot;
    }
    __threadfence();
    __syncthreads();
    for (int q = tid; q <= nq; q += 1024) {
        const int64_t it = item_off[q];
        const int64_t v = (q == nq || it >= n_items) ? n_cand : __hip_atomic_load(&cand_start[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&seg[q], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q < nq) { qmin[q] = ~0ull; qmax[q] = 0ull; cnt[q] = 0; }
    }
    if (tid < 4) status[tid] = 0;
    __threadfence();
    __syncthreads();
    // B + C.
    const int ns = slots ? *n_slots : (int)n_items;
    run = 0;
    for (int s0 = 0; s0 < ns; s0 += 1024) {
        const int sl = s0 + tid;
        int64_t x = 0;
        if (sl < ns) {
            StreamSlot d;
            d.start = 0; d.len = 0; d.pad = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ii = slots ? (g < G ? slots[(int64_t)sl * G + g] : -1) : (g == 0 ? sl : -1);
                d.q[g] = -1; d.tab0[g] = -2; d.tab1[g] = -2; d.rbase[g] = 0u;
                if (ii >= 0) {
                    const WorkItem it = items[ii];
                    if (g == 0) { d.start = it.start; d.len = it.len; }
                    d.q[g] = it.q; d.tab0[g] = it.tab0; d.tab1[g] = it.tab1;
                    d.rbase[g] = (uint32_t)(__hip_atomic_load(&cand_start[ii], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                                            __hip_atomic_load(&seg[it.q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                }
            }
            desc[sl] = d;
            x = (d.len + row - 1) / row;
        }
        int64_t tot;
        const int64_t ex = prep_block_excl_scan(x, s_w, &tot);
        if (sl < ns) rowoff[sl] = run + ex;
        run += tot;
    }
    if (tid == 0) {
        rowoff[ns] = run;
        if (!slots) *n_slots = ns;
    }
}

// Persistent workgroups, one equal range of rows each.  A range spans PIECES of consecutive slots; a slot = one chunk of one cell for
// up to G queries (the slot builder of lopq_search.hip groups the work items of a cell chunk).  SAMPLE: every sample_stride-th row
// of every slot (rows counted inside the slot, so the sample does not depend on the ranges).
template <int M, int G, bool SAMPLE>
__global__ __launch_bounds__(64 * CIS_STREAM_NW) void k_adc_stream(const StreamSlot* __restrict__ desc, const int* __restrict__ n_slots,
                                                    const int64_t* __restrict__ rowoff,
                                                    const float* __restrict__ T32 /* null: converted from T */, const double* __restrict__ T,
                                                    const uint8_t* __restrict__ codes, int K,
                                                    const float* __restrict__ tau, uint32_t* __restrict__ bmin, int B, int sample_stride, int flush,
                                                    uint32_t* __restrict__ surv, int* __restrict__ cnt, int cap) {
    constexpr int NW = CIS_STREAM_NW;
    constexpr int nf = M / 2;
    constexpr int CPL = 16 / M;        // candidates per lane and 16-byte load
    constexpr int ROW = 64 * CPL;      // candidates per wave and load
    constexpr int R = stream_copies(M, G);
    constexpr int ROWB = R * M * G * 4;  // bytes of an entry's row: R copies of the M sub-quantizers' entries of the G queries
    constexpr int ROWSH = ROWB == 256 ? 8 : (ROWB == 128 ? 7 : (ROWB == 64 ? 6 : 5));
    static_assert((1 << ROWSH) == ROWB, "rows of 32 .. 256 bytes");
    extern __shared__ __align__(16) float s_tab[];  // [K][R][M][G]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ns = *n_slots;
    const RotConsts<M> rc = make_rot<M>(lane);
    // LDS byte address of sub-quantizer j(t, lane)'s entry in row 0: the tables' base + (copy * M + j) * G * 4.  Entry addresses are
    // integers all the way (bit-field extract, shift-add, ds_read): pointer arithmetic on s_tab costs an add of the base per gather.
    const uint32_t tab_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)s_tab;
    uint32_t cjb[M];
#pragma unroll
    for (int t = 0; t < M; ++t) {
        cjb[t] = tab_base + (((uint32_t)(lane & 31) / (uint32_t)M) % (uint32_t)R * (uint32_t)M + (rc.cj[t] >> 2)) * (uint32_t)(G * 4);
        asm volatile("" : "+v"(cjb[t]));  // M registers for the whole kernel (else re-derived per use)
    }
    int cur0[G], cur1[G];  // the half tables in LDS
#pragma unroll
    for (int g = 0; g < G; ++g) { cur0[g] = -1; cur1[g] = -1; }
    float mn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) mn[g] = __uint_as_float(0x7f800000u);

    const int64_t total = rowoff[ns];
    int64_t r_lo = total * blockIdx.x / gridDim.x;
    const int64_t r_hi = total * (blockIdx.x + 1) / gridDim.x;
    if (r_lo >= r_hi) return;
    int s;
    {   // the last slot with rowoff[s] <= r_lo (empty slots in front of it have the same offset: skipped by the search)
        int lo = 0, hi = ns;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rowoff[mid] <= r_lo) lo = mid; else hi = mid;
        }
        s = lo;
    }
    for (; r_lo < r_hi; ++s) {
        const int64_t sb = rowoff[s], se = rowoff[s + 1];
        if (se <= r_lo) continue;  // (an empty slot)
        const int ra = (int)(r_lo - sb);
        const int rb = (int)((r_hi < se ? r_hi : se) - sb);
        r_lo = sb + rb;
        // ---- rows [ra, rb) of slot s -------------------------------------------------------------------------------------------------
        const StreamSlot* sp = desc + s;   // uniform: scalar loads
        const int len = __builtin_amdgcn_readfirstlane(sp->len);
        const int64_t start = sp->start;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(codes + start * M), 0, len * M, 0x00020000);
        const int ss = SAMPLE ? sample_stride : 1;
        const int rstep = NW * ss;
        const int rfirst = (ra + ss - 1) / ss * ss + wv * ss;
        // rows past the piece belong to another workgroup (or to no one): an offset past the descriptor returns zeros without a memory access
        auto request = [&](int r) -> u32x4_t {
            const int off = r < rb ? (r * ROW + lane * CPL) * M : 0x7ffffff0;
            return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, CIS_STREAM_AUX);
        };
        u32x4_t cw = request(rfirst), cn;  // the piece's first rows travel while its tables are staged
        int qg[G], t0g[G], t1g[G];
        uint32_t rbase[G];   // retrieval index of the chunk's first candidate, relative to the query's first candidate
        float tg[G];
        bool any_new = false;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            qg[g] = sp->q[g];
            const bool on = qg[g] >= 0;
            rbase[g] = sp->rbase[g];
            tg[g] = (on && !SAMPLE) ? tau[qg[g]] : -1.0f;
            t0g[g] = sp->tab0[g];
            t1g[g] = sp->tab1[g];
            any_new = any_new || t0g[g] != cur0[g] || t1g[g] != cur1[g];
        }
        if (any_new) {   // uniform over the workgroup
            __syncthreads();     // the previous piece's readers are done with the tables
            // entry k by thread k: its M x G values (coalesced reads of T32[tab][j][k]), then the row's R copies as 16-byte stores, the
            // copy order rotated by k so that the lanes of a store group spread over the banks
            for (int k = tid; k < K; k += 64 * NW) {
                float v[M * G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int64_t u0 = (int64_t)t0g[g] * nf * K, u1 = (int64_t)t1g[g] * nf * K;
#pragma unroll
                    for (int j = 0; j < M; ++j) v[j * G + g] = t0g[g] >= 0 ? (j < nf ? tab_f1(T32, T, u0 + j * K + k) : tab_f1(T32, T, u1 + (j - nf) * K + k)) : 0.f;
                }
                char* rowp = reinterpret_cast<char*>(s_tab) + (size_t)k * ROWB;
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const int cc = (c + k) & (R - 1);
#pragma unroll
                    for (int x = 0; x < M * G / 4; ++x) {
                        const f32x4_t q4 = {v[4 * x], v[4 * x + 1], v[4 * x + 2], v[4 * x + 3]};
                        *reinterpret_cast<f32x4_t*>(rowp + cc * (M * G * 4) + 16 * x) = q4;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) { cur0[g] = t0g[g]; cur1[g] = t1g[g]; }
            __syncthreads();
        }
        unsigned blk = 0u;
        for (int r0 = rfirst; r0 < rb; r0 += rstep) {
            cn = request(r0 + rstep);  // in flight while this iteration gathers
            float d[CPL][G];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                uint32_t w[(M + 3) / 4];
#pragma unroll
                for (int x = 0; x < (M + 3) / 4; ++x) w[x] = cw[c * ((M + 3) / 4) + x];
                // the lane's rotation: which dword holds the byte of step t (M = 8: two candidates of swapped dwords)
                uint32_t wsel[(M + 3) / 4];
                if constexpr (M == 4) wsel[0] = w[0];
                else if constexpr (M == 8) { wsel[0] = rc.hsel ? w[1] : w[0]; wsel[1] = rc.hsel ? w[0] : w[1]; }
                else {
#pragma unroll
                    for (int th = 0; th < 4; ++th) {
                        const uint32_t hs = rc.hsel ^ (uint32_t)th;
                        wsel[th] = hs == 0 ? w[0] : (hs == 1 ? w[1] : (hs == 2 ? w[2] : w[3]));
                    }
                }
#pragma unroll
                for (int t = 0; t < M; ++t) {
                    const int th = t >> 2, tq = t & 3;
                    // byte of sub-quantizer j(t, lane), then the LDS address of entry (byte, j): v_bfe_u32 + v_lshl_add_u32
                    const uint32_t addr = (__builtin_amdgcn_ubfe(wsel[th], rc.sh[tq], 8u) << ROWSH) + cjb[t];
                    const __attribute__((address_space(3))) char* ep = (const __attribute__((address_space(3))) char*)(uintptr_t)addr;
                    if constexpr (G == 1) {
                        const float e = *reinterpret_cast<const __attribute__((address_space(3))) float*>(ep);
                        d[c][0] = t == 0 ? e : d[c][0] + e;
                    } else if constexpr (G == 2) {
                        const f32x2_t e = *reinterpret_cast<const __attribute__((address_space(3))) f32x2_t*>(ep);
                        d[c][0] = t == 0 ? e[0] : d[c][0] + e[0];
                        d[c][1] = t == 0 ? e[1] : d[c][1] + e[1];
                    } else {
                        static_assert(G == 1 || G == 2 || G == 4, "one, two or four queries per slot");
                        const f32x4_t e = *reinterpret_cast<const __attribute__((address_space(3))) f32x4_t*>(ep);
#pragma unroll
                        for (int g = 0; g < 4; ++g) d[c][g] = t == 0 ? e[g] : d[c][g] + e[g];
                    }
                }
            }
            if (SAMPLE) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const bool valid = r0 * ROW + lane * CPL + c < len;
#pragma unroll
                    for (int g = 0; g < G; ++g) mn[g] = (valid && d[c][g] < mn[g]) ? d[c][g] : mn[g];
                }
                // a lane folds `flush` sampled rows into one bucket (any partition of the samples into buckets keeps the bound of the
                // header comment; the more non-empty buckets, the tighter): a query that meets ONE long chunk still fills its buckets
                const int itn = r0 / rstep;   // counted inside the slot: ranges of different workgroups fill different buckets
                blk = (unsigned)s * 40503u + (unsigned)(itn / flush);
                if ((itn + 1) % flush == 0) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (qg[g] >= 0 && f2u(mn[g]) < 0x7f800000u)
                            atomicMin(&bmin[(int64_t)qg[g] * B + ((((blk * (64u * NW) + (unsigned)tid) * 2654435761u) >> 12) & (unsigned)(B - 1))], f2u(mn[g]));
                        mn[g] = __uint_as_float(0x7f800000u);
                    }
                }
            } else {
                bool any = false;
#pragma unroll
                for (int c = 0; c < CPL; ++c)
#pragma unroll
                    for (int g = 0; g < G; ++g) any = any || d[c][g] <= tg[g];
                if (__builtin_amdgcn_ballot_w64(any) != 0ull) {  // rare: a few thousand candidates of hundreds of millions pass
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const int p = r0 * ROW + lane * CPL + c;
#pragma unroll
                        for (int g = 0; g < G; ++g)
                            if (p < len && d[c][g] <= tg[g]) {
                                const int j = atomicAdd(&cnt[qg[g]], 1);
                                if (j < cap) surv[(int64_t)qg[g] * cap + j] = rbase[g] + (uint32_t)p;
                            }
                    }
                }
            }
            cw = cn;
        }
        if (SAMPLE) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (qg[g] >= 0 && f2u(mn[g]) < 0x7f800000u)
                    atomicMin(&bmin[(int64_t)qg[g] * B + ((((blk * (64u * NW) + (unsigned)tid) * 2654435761u) >> 12) & (unsigned)(B - 1))], f2u(mn[g]));
                mn[g] = __uint_as_float(0x7f800000u);
            }
        }
    }
}

// tau[q] = (an upper bound within 2^-16 of the range of) the k-th smallest of the query's B bucket minima; +inf when fewer than k buckets
// were touched.  One workgroup of 1024 threads per query, B / 1024 values per thread.  Non-negative floats order like their bits: a
// two-level radix SELECT on (value - min) -- a 256-bin LDS histogram of the leading eight bits of the range, the bin that holds the
// k-th value, then the next eight bits inside that bin -- and tau = the upper edge of the final sub-bin (round 5: sixteen halvings, a
// barrier pair each: 20 us; fifteen pivots per round x four: 17-20 us, all compares; this form: two histogram passes).
// Every value at or below the k-th smallest lies at or below that edge, so tau admits a few more candidates than the exact k-th
// smallest would, which only lengthens the list; the proof does not care where tau came from.
// The buckets are RESET here, by their only reader: the next batch's sample pass finds them clean (the index memsets them when it
// allocates them; values >= 0x7f800000 are empty).
static __device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v) {   // on the VALU (DPP / permlane swaps)
    v += lane_xor<1>(v);
    v += lane_xor<2>(v);
    v += lane_xor<4>(v);
    v += lane_xor<8>(v);
    v += lane_xor<16>(v);
    v += lane_xor<32>(v);
    return v;
}
static __device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v) {
    uint32_t o;
    o = lane_xor<1>(v); v = o < v ? o : v;
    o = lane_xor<2>(v); v = o < v ? o : v;
    o = lane_xor<4>(v); v = o < v ? o : v;
    o = lane_xor<8>(v); v = o < v ? o : v;
    o = lane_xor<16>(v); v = o < v ? o : v;
    o = lane_xor<32>(v); v = o < v ? o : v;
    return v;
}
static __device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v) {
    uint32_t o;
    o = lane_xor<1>(v); v = o > v ? o : v;
    o = lane_xor<2>(v); v = o > v ? o : v;
    o = lane_xor<4>(v); v = o > v ? o : v;
    o = lane_xor<8>(v); v = o > v ? o : v;
    o = lane_xor<16>(v); v = o > v ? o : v;
    o = lane_xor<32>(v); v = o > v ? o : v;
    return v;
}

// the bin of a 256-bin histogram (four bins per lane of wave 0) where the running count, started at `before`, first reaches k;
// *below = the count of the bins before it.  Every lane returns the same pair.
static __device__ __forceinline__ int tau_pick_bin(const int* __restrict__ hist, int before, int k, int* below) {
    const int lane = threadIdx.x & 63;
    const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
    int inc = c0 + c1 + c2 + c3;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(inc, d);
        if (lane >= d) inc += y;
    }
    const int ex = before + inc - (c0 + c1 + c2 + c3);   // count of everything before this lane's four bins
    int bin = -1, bel = 0;
    if (ex < k && ex + c0 + c1 + c2 + c3 >= k) {          // exactly one lane
        int run = ex;
        bin = 4 * lane; bel = run;
        if (run + c0 < k) { run += c0; bin = 4 * lane + 1; bel = run;
            if (run + c1 < k) { run += c1; bin = 4 * lane + 2; bel = run;
                if (run + c2 < k) { run += c2; bin = 4 * lane + 3; bel = run; } } }
    }
    const unsigned long long m = __ballot(bin >= 0);
    const int src = m ? __builtin_ctzll(m) : 0;
    *below = __shfl(bel, src);
    return __shfl(bin, src);
}

template <int PER>
__global__ __launch_bounds__(1024) void k_stream_tau(uint32_t* __restrict__ bmin, int B, int k, float* __restrict__ tau) {
    __shared__ int s_h1[256], s_h2[256];
    __shared__ uint32_t s_mm[2];
    __shared__ int s_sel[4];   // finite buckets; bin of the first level; count below it
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    uint32_t v[PER];
    uint32_t lo = 0xffffffffu, hi = 0u;
    int nfin = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        uint32_t* bp = bmin + (int64_t)q * B + i * 1024 + tid;
        v[i] = *bp;
        *bp = 0x7f800000u;
        if (v[i] < 0x7f800000u) { hi = v[i] > hi ? v[i] : hi; lo = v[i] < lo ? v[i] : lo; ++nfin; }
        else v[i] = 0xffffffffu;
    }
    if (tid == 0) { s_mm[0] = 0xffffffffu; s_mm[1] = 0u; s_sel[0] = 0; }
    if (tid < 256) { s_h1[tid] = 0; s_h2[tid] = 0; }
    __syncthreads();
    lo = wave_min_dpp(lo);
    hi = wave_max_dpp(hi);
    nfin = (int)wave_sum_dpp((uint32_t)nfin);
    if (lane == 0) {
        atomicMin(&s_mm[0], lo);
        atomicMax(&s_mm[1], hi);
        atomicAdd(&s_sel[0], nfin);
    }
    __syncthreads();
    lo = s_mm[0];
    hi = s_mm[1];
    if (s_sel[0] < k) {   // uniform over the workgroup
        if (tid == 0) tau[q] = __uint_as_float(0x7f800000u);
        return;
    }
    const uint32_t span = hi - lo;
    const int bits = span ? 32 - __builtin_clz(span) : 0;   // (v - lo) < 2^bits
    const int sh1 = bits > 8 ? bits - 8 : 0, sh2 = sh1 > 8 ? sh1 - 8 : 0;
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (v[i] != 0xffffffffu) atomicAdd(&s_h1[(v[i] - lo) >> sh1], 1);
    __syncthreads();
    if (tid < 64) {
        int below;
        const int b1 = tau_pick_bin(s_h1, 0, k, &below);
        if (tid == 0) { s_sel[1] = b1; s_sel[2] = below; }
    }
    __syncthreads();
    const uint32_t b1 = (uint32_t)s_sel[1];
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (v[i] != 0xffffffffu && ((v[i] - lo) >> sh1) == b1) atomicAdd(&s_h2[((v[i] - lo) >> sh2) & ((1u << (sh1 - sh2)) - 1u)], 1);
    __syncthreads();
    if (tid < 64) {
        int below;
        const int b2 = sh1 > sh2 ? tau_pick_bin(s_h2, s_sel[2], k, &below) : 0;
        if (tid == 0) {
            // upper edge of sub-bin (b1, b2): the largest offset whose leading bits are (b1 << (sh1 - sh2)) | b2
            const uint64_t edge = ((((uint64_t)b1 << (sh1 - sh2)) | (uint64_t)b2) + 1ull) << sh2;
            const uint64_t t = (uint64_t)lo + edge - 1ull;
            tau[q] = __uint_as_float(t < (uint64_t)hi ? (uint32_t)t : hi);
        }
    }
}

// exact float64 key of every listed candidate + the key range of the query (for k_select_topl)
template <int M>
__global__ __launch_bounds__(256) void k_stream_keys(const WorkItem* __restrict__ items, const int64_t* __restrict__ cand_start,
                                                     const int64_t* __restrict__ seg, const int64_t* __restrict__ item_off, int64_t n_items,
                                                     const double* __restrict__ T, const uint8_t* __restrict__ codes, int K,
                                                     const uint32_t* __restrict__ surv, const int* __restrict__ cnt, int cap,
                                                     uint64_t* __restrict__ keys, unsigned long long* __restrict__ qmin,
                                                     unsigned long long* __restrict__ qmax) {
    __shared__ unsigned long long s_mm[2];
    constexpr int nf = M / 2;
    const int q = blockIdx.y;
    int n = cnt[q];
    n = n < cap ? n : cap;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    if (threadIdx.x == 0) { s_mm[0] = ~0ull; s_mm[1] = 0ull; }
    __syncthreads();
    if (j < n) {
        const int64_t g = seg[q] + (int64_t)surv[(int64_t)q * cap + j];
        int64_t lo = item_off[q], hi = item_off[q + 1];
        hi = hi < n_items ? hi : n_items;
        while (hi - lo > 1) {  // last item with cand_start <= g
            const int64_t mid = (lo + hi) >> 1;
            if (cand_start[mid] <= g) lo = mid; else hi = mid;
        }
        const WorkItem it = items[lo];
        const CodeWords<M> cw = load_code<M>(codes, it.start + (g - cand_start[lo]));
        const uint64_t kk = (uint64_t)__double_as_longlong(adc64_words<M>(cw.w, K, T + (int64_t)it.tab0 * nf * K, T + (int64_t)it.tab1 * nf * K));
        keys[(int64_t)q * cap + j] = kk;
        atomicMin(&s_mm[0], (unsigned long long)kk);
        atomicMax(&s_mm[1], (unsigned long long)kk);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_mm[0] <= s_mm[1]) {
        atomicMin(&qmin[q], s_mm[0]);
        atomicMax(&qmax[q], s_mm[1]);
    }
}

// The proof of the header comment, the ranked output and the visited counts of a query, one workgroup per query (round 6: one launch for
// k_stream_verify + k_emit_sorted + k_copy_visited).  status[0] = queries whose proof failed, status[1] = lists that overflowed,
// status[2] = workgroups done; the last one lands the words in pinned host memory (status_host) behind a sequence number.
__global__ __launch_bounds__(256) void k_stream_finish(const uint64_t* __restrict__ sel_keys, const uint64_t* __restrict__ sel_vals, const int* __restrict__ nsel,
                                                       int64_t stride, const int* __restrict__ cnt, int cap, const int64_t* __restrict__ seg, const float* __restrict__ tau,
                                                       int nq, int L, double eps, const WorkItem* __restrict__ items, const int64_t* __restrict__ ids,
                                                       const PlanOut* __restrict__ plan, cis_hit* __restrict__ out_hits, int64_t* __restrict__ out_ids,
                                                       double* __restrict__ out_dists, int* __restrict__ out_n, int32_t* __restrict__ out_cells,
                                                       uint32_t* __restrict__ out_pos, int32_t* __restrict__ out_visited,
                                                       int* __restrict__ status, volatile int64_t* __restrict__ status_host, int64_t seq) {
    const int q = blockIdx.x;
    const int nv_all = nsel[q];
    const int nv = nv_all < L ? nv_all : L;
    const int64_t a = (int64_t)q * stride, o = (int64_t)q * L;
    for (int x = threadIdx.x; x < L; x += blockDim.x) {
        cis_hit hh;
        hh.dist = __longlong_as_double(0x7ff0000000000000LL);
        hh.visit_rank = 0xffffffffu; hh.pos = 0xffffffffu; hh.id = -1; hh.cell = -1; hh.reserved = 0;
        if (x < nv) {
            const uint64_t v = sel_vals[a + x];
            const WorkItem it = items[v >> 32];
            const uint32_t p = (uint32_t)v;
            hh.dist = __longlong_as_double((long long)sel_keys[a + x]);
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
    if (threadIdx.x == 0) {
        if (out_n) out_n[q] = nv;
        if (out_visited) out_visited[q] = plan[q].visited;
        const int64_t ncand = seg[q + 1] - seg[q];
        const int c = cnt[q];
        int bad = 0, over = 0;
        if (c > cap) over = 1;
        else {
            // k_select_topl ranks nothing (nsel = 0) when more than its tie capacity of exact ties sit at the cut: the generic path resolves
            // those (first ties in retrieval order).  Checked BEFORE the every-candidate-listed shortcut -- a short query whose every
            // candidate was listed can be such a crowd (round-5 advice: it returned n_found = 0 with no fall-back).
            const int64_t want = (int64_t)L < ncand ? (int64_t)L : ncand;
            if ((int64_t)nv_all != want) bad = 1;
            else if ((int64_t)c != ncand) {   // (c == ncand: every candidate was listed -- tau = +inf, or a short query)
                // c < ncand and L ranked: B (1 + 2 eps) <= tau proves that every candidate at or below the cut was listed
                const double Bd = __longlong_as_double((long long)sel_keys[a + nv_all - 1]);
                if (!(nv_all == L && Bd * (1.0 + 2.0 * eps) <= (double)tau[q])) bad = 1;
            }
        }
        if (bad) atomicAdd(&status[0], 1);
        if (over) atomicAdd(&status[1], 1);
        __threadfence();
        const int done = atomicAdd(&status[2], 1) + 1;
        if (done == nq) {
            __threadfence();
            status_host[0] = __hip_atomic_load(&status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            status_host[1] = __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            status_host[2] = seq;
            __threadfence_system();
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
bool stream_supported(int M, int K, int L) { return (M == 4 || M == 8 || M == 16) && K <= 256 && L >= 1 && L <= 1024; }

size_t stream_lds(int M, int K, int G) { return (size_t)K * stream_copies(M, G) * M * G * sizeof(float); }

int stream_max_group() { return 4; }

template <int M, int G, bool SAMPLE>
static void launch_stream_t(int grid, hipStream_t st, const StreamSlot* desc, const int* n_slots, const int64_t* rowoff, const float* T32,
                            const double* T, const uint8_t* codes, int K, const float* tau, uint32_t* bmin, int B,
                            int sample_stride, int flush, uint32_t* surv, int* cnt, int cap) {
    hipLaunchKernelGGL((k_adc_stream<M, G, SAMPLE>), dim3((unsigned)grid), dim3(64 * CIS_STREAM_NW), stream_lds(M, K, G), st, desc, n_slots, rowoff, T32, T,
                       codes, K, tau, bmin, B, sample_stride, flush < 1 ? 1 : flush, surv, cnt, cap);
}

void launch_stream_scan(int M, int G, bool sample, int grid, hipStream_t st, const void* desc_, const int* n_slots, const int64_t* rowoff,
                        const float* T32, const double* T, const uint8_t* codes, int K, const float* tau,
                        uint32_t* bmin, int B, int sample_stride, int flush, uint32_t* surv, int* cnt, int cap) {
    const StreamSlot* desc = static_cast<const StreamSlot*>(desc_);
#define CIS_STREAM(MM, GG)                                                                                                              \
    if (M == MM && G == GG) {                                                                                                           \
        if (sample) launch_stream_t<MM, GG, true>(grid, st, desc, n_slots, rowoff, T32, T, codes, K, tau, bmin, B, sample_stride, flush, surv, cnt, cap);  \
        else launch_stream_t<MM, GG, false>(grid, st, desc, n_slots, rowoff, T32, T, codes, K, tau, bmin, B, sample_stride, flush, surv, cnt, cap);        \
        return;                                                                                                                         \
    }
    CIS_STREAM(8, 1) CIS_STREAM(8, 2) CIS_STREAM(8, 4) CIS_STREAM(4, 1) CIS_STREAM(4, 2) CIS_STREAM(4, 4) CIS_STREAM(16, 1) CIS_STREAM(16, 2) CIS_STREAM(16, 4)
#undef CIS_STREAM
}

size_t stream_slot_bytes() { return sizeof(StreamSlot); }

// workgroups of the persistent launch: CIS_STREAM_PER_CU per CU (two 8-wave workgroups: measured best, see the head of the file),
// fewer when the batch has fewer rows than that
int stream_grid(int M, int G, int K, int64_t max_rows) {
    (void)M; (void)G; (void)K;
    int64_t g = (int64_t)256 * CIS_STREAM_PER_CU;
    if (const char* e = getenv("CIS_STREAM_GRID")) g = atoll(e) > 0 ? atoll(e) : g;  // experiments
    const int64_t by_rows = ceil_div(max_rows < 1 ? 1 : max_rows, (int64_t)CIS_STREAM_NW);
    g = g < by_rows ? g : by_rows;
    return (int)(g < 1 ? 1 : g);
}

void launch_stream_prep(hipStream_t st, const WorkItem* items, int64_t n_items, const int64_t* item_off, int nq, int64_t n_cand, const int* slots, int* n_slots,
                        int G, int M, int64_t* cand_start, int64_t* seg, unsigned long long* qmin, unsigned long long* qmax, int* cnt, int* status, int64_t* rowoff,
                        void* desc, const int64_t* d_totals) {
    hipLaunchKernelGGL(k_stream_prep, dim3(1), dim3(1024), 0, st, items, n_items, item_off, nq, n_cand, slots, n_slots, G, 64 * (16 / M), cand_start, seg, qmin, qmax,
                       cnt, status, rowoff, static_cast<StreamSlot*>(desc), d_totals);
}

void launch_stream_tau(hipStream_t st, uint32_t* bmin, int B, int k, int nq, float* tau) {
    // B = STREAM_B = 1024 threads x PER
    hipLaunchKernelGGL(k_stream_tau<STREAM_B / 1024>, dim3((unsigned)nq), dim3(1024), 0, st, bmin, B, k, tau);
}

void launch_stream_keys(int M, hipStream_t st, const WorkItem* items, const int64_t* cand_start, const int64_t* seg, const int64_t* item_off,
                        int64_t n_items, const double* T, const uint8_t* codes, int K, const uint32_t* surv, const int* cnt, int cap, int nq,
                        uint64_t* keys, unsigned long long* qmin, unsigned long long* qmax) {
    const dim3 g((unsigned)ceil_div(cap, 256), (unsigned)nq);
    if (M == 4) hipLaunchKernelGGL(k_stream_keys<4>, g, dim3(256), 0, st, items, cand_start, seg, item_off, n_items, T, codes, K, surv, cnt, cap, keys, qmin, qmax);
    else if (M == 8) hipLaunchKernelGGL(k_stream_keys<8>, g, dim3(256), 0, st, items, cand_start, seg, item_off, n_items, T, codes, K, surv, cnt, cap, keys, qmin, qmax);
    else hipLaunchKernelGGL(k_stream_keys<16>, g, dim3(256), 0, st, items, cand_start, seg, item_off, n_items, T, codes, K, surv, cnt, cap, keys, qmin, qmax);
}

void launch_stream_finish(hipStream_t st, const uint64_t* sel_keys, const uint64_t* sel_vals, const int* nsel, int64_t stride, const int* cnt, int cap,
                          const int64_t* seg, const float* tau, int nq, int L, int M, const WorkItem* items, const int64_t* ids, const PlanOut* plan,
                          cis_hit* out_hits, int64_t* out_ids, double* out_dists, int* out_n, int32_t* out_cells, uint32_t* out_pos, int32_t* out_visited,
                          int* status, int64_t* status_host_dev, int64_t seq) {
    const double eps = 2.0 * M * 5.9604644775390625e-08;  // 2 M 2^-24
    hipLaunchKernelGGL(k_stream_finish, dim3((unsigned)nq), dim3(256), 0, st, sel_keys, sel_vals, nsel, stride, cnt, cap, seg, tau, nq, L, eps, items, ids, plan,
                       out_hits, out_ids, out_dists, out_n, out_cells, out_pos, out_visited, status, status_host_dev, seq);
}
