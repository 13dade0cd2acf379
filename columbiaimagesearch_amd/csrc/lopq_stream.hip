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
// Here nothing is ranked inside the stream:
//   1. k_adc_stream<SAMPLE>: every SS-th row of every chunk; each lane keeps the MINIMUM float32 distance it saw and folds it into
//      one of B buckets per query (atomicMin).  count(buckets <= v) <= count(samples <= v), so the k-th smallest bucket minimum is
//      an upper bound of the k-th smallest sample distance;
//   2. k_stream_tau: per query, tau = the k-th smallest bucket minimum -- about k x SS candidates of the whole query lie below it
//      (k is chosen so that this is a few thousand, ~20 x limit);
//   3. k_adc_stream: the stream.  Codes arrive as 16 bytes per lane with several loads in flight, the float32 tables of the slot's
//      (<= G) queries sit in LDS entry-major with the sub-quantizers rotated over the lanes (RotConsts: 2.1-way bank conflicts
//      instead of 3.5), a candidate costs M gathers + M adds and ONE compare against tau; the few that pass append their
//      retrieval index to the query's list (one atomic each);
//   4. k_stream_keys: the exact float64 distance of every listed candidate (left-to-right sum of the float64 entries: the bits of
//      every other route), k_select_topl ranks them by (distance, retrieval index), k_stream_verify PROVES the result:
//          let B = the limit-th smallest exact distance in the list (the list holds >= limit candidates, or every candidate);
//          |d32 - d64| <= eps d64 (eps = 2 M 2^-24: M rounded entries, M - 1 rounded adds), so a candidate of the true top
//          `limit` has d64 <= B, hence d32 <= B (1 + eps); if B (1 + 2 eps) <= tau it was listed.  All candidates with d64 <= B are
//          then in the list, ties included, and ranking the list by (d64, retrieval index) is the reference's stable sort.
//      A query whose proof fails (sample unlucky), or whose list overflowed (a crowd of equal codes), is flagged in pinned
//      memory; the host then answers the batch through the generic path.  The result never depends on the sample.
#include "scan_common.h"

#ifndef CIS_STREAM_RING
#define CIS_STREAM_RING 1   // 1: the next iteration's code rows are requested before the current ones are used (0: requested and awaited per iteration)
#endif
#ifndef CIS_STREAM_U
#define CIS_STREAM_U 0      // 16-byte loads per lane and iteration; 0: two for one query per slot and for the sample pass, four for a pair
#endif                      // (measured on 200 M codes, profiles/r05_experiments.txt: 324 / 333 / 339 us for ring + 2, ring + 4, no ring + 4)
#ifndef CIS_STREAM_NW
#define CIS_STREAM_NW 4     // waves per workgroup = per slot: the workgroup reads NW KB of contiguous codes per load round
#endif
#ifndef CIS_STREAM_REPL
#define CIS_STREAM_REPL 0
#endif

static __device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

__global__ void k_stream_init(uint32_t* __restrict__ bmin, int64_t n_b, int* __restrict__ cnt, int nq, int* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_b) bmin[i] = 0x7f800000u;  // +inf
    if (i < nq) cnt[i] = 0;
    if (i < 4) status[i] = 0;
}

// One workgroup (four waves) per slot at a time, persistent over the slots with a stride of the grid.  A slot = one chunk of one
// cell for up to G queries (the slot builder of lopq_search.hip groups the work items of a cell chunk).
template <int M, int G, bool SAMPLE>
__global__ __launch_bounds__(64 * CIS_STREAM_NW) void k_adc_stream(const WorkItem* __restrict__ items, const int* __restrict__ slots, const int* __restrict__ n_slots,
                                                    const float* __restrict__ T32 /* null: converted from T */, const double* __restrict__ T,
                                                    const uint8_t* __restrict__ codes, int K,
                                                    const int64_t* __restrict__ cand_start, const int64_t* __restrict__ seg,
                                                    const float* __restrict__ tau, uint32_t* __restrict__ bmin, int B, int sample_stride,
                                                    uint32_t* __restrict__ surv, int* __restrict__ cnt, int cap) {
    constexpr int nf = M / 2;
    constexpr int CPL = 16 / M;        // candidates per lane and 16-byte load
    constexpr int ROW = 64 * CPL;      // candidates per wave and load
    constexpr int U = CIS_STREAM_U > 0 ? CIS_STREAM_U : ((SAMPLE || G == 1) ? 2 : 4);  // loads per wave and iteration (the ring doubles what is in flight)
    // Tables in LDS, entry-major, the sub-quantizers rotated over the lanes.  Measured (profiles/r05f_c4x_*): SQ_LDS_BANK_CONFLICT /
    // SQ_LDS_IDX_ACTIVE = 0.66 -- the worst of the eight 4-lane groups of a read is 2.9-way, not the 2.1-way of one group -- and the LDS
    // pipe 0.72 busy at 0.60 of 8 TB/s.  CIS_STREAM_REPL=1 REPLICATES the tables so that the gathers meet no conflict at all (a row of 128
    // bytes per k holds R copies of the M entries; lane l uses copy (l % 32) / M, so the 32 lanes of a read group own distinct banks
    // whatever their k): built, bit-identical, and slower -- see the macro.
    // MEASURED (profiles/r05g_*): 357 us against 333-339 us per exhaustive launch over 200 M codes -- the conflicts go (the LDS pipe was
    // 0.72 busy, not saturated), four times the staging and 32 KB per workgroup cost more: the stream waits on HBM.
    // CIS_STREAM_REPL = 1: as many copies as a 128-byte row holds (no conflict at all); = 2: TWO copies (rows of 64 bytes: lanes l and
    // l + 16 of a read group share a copy and a sub-quantizer, so a read is two passes instead of ~2.9, at 16 KB per workgroup)
    constexpr int RMAX = (32 / G) / M > 0 ? (32 / G) / M : 1;
    constexpr int R = CIS_STREAM_REPL == 1 ? RMAX : (CIS_STREAM_REPL == 2 ? (RMAX >= 2 ? 2 : 1) : 1);    // copies
    constexpr int ROWSH = (R * M * G * 4 == 128) ? 7 : (R * M * G * 4 == 64 ? 6 : (R * M * G * 4 == 32 ? 5 : 4));  // log2(row bytes)
    static_assert(R >= 1 && (1 << ROWSH) == R * M * G * 4, "rows of 16 .. 128 bytes");
    extern __shared__ __align__(16) float s_tab[];  // [K][R][M][G]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ns = *n_slots;
    const RotConsts<M> rc = make_rot<M>(lane);
    uint32_t cjb[M];  // byte offset of sub-quantizer j(t, lane) inside an entry row: j * G * 4
#pragma unroll
    for (int t = 0; t < M; ++t) {
        cjb[t] = (((uint32_t)(lane & 31) / (uint32_t)M) % (uint32_t)R * (uint32_t)M + (rc.cj[t] >> 2)) * (uint32_t)(G * 4);
        asm volatile("" : "+v"(cjb[t]));  // M registers for the whole kernel (else re-derived per use)
    }
    for (int s = blockIdx.x; s < ns; s += gridDim.x) {
        int ii[G];
#pragma unroll
        for (int g = 0; g < G; ++g) ii[g] = __builtin_amdgcn_readfirstlane(slots[(int64_t)s * G + g]);
        if (ii[0] < 0) continue;
        const WorkItem it0 = items[ii[0]];
        const int len = __builtin_amdgcn_readfirstlane(it0.len);
        const int64_t start = it0.start;
        int qg[G];
        uint32_t rbase[G];   // retrieval index of the chunk's first candidate, relative to the query's first candidate
        float tg[G];
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(codes + start * M), 0, len * M, 0x00020000);
        const int rows = (len + ROW - 1) / ROW;
        const int rstep = SAMPLE ? CIS_STREAM_NW * sample_stride : CIS_STREAM_NW;
        const int rfirst = wv * (SAMPLE ? sample_stride : 1);
        // past the chunk the descriptor returns zeros (no memory access); such candidates are masked below
        auto request = [&](int r0, u32x4_t(&dst)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) dst[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((r0 + u * rstep) * ROW + lane * CPL) * M, 0, 0);
        };
        u32x4_t cw[U], cn[U];
        if (CIS_STREAM_RING) request(rfirst, cw);  // the slot's first rows travel while its tables are staged
        __syncthreads();     // the previous slot's readers are done with the tables
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool on = ii[g] >= 0;
            const WorkItem it = items[on ? ii[g] : ii[0]];
            qg[g] = on ? it.q : -1;
            rbase[g] = on ? (uint32_t)(cand_start[ii[g]] - seg[it.q]) : 0u;
            tg[g] = (on && !SAMPLE) ? tau[it.q] : -1.0f;
            // tables: thread t stages entries k = t, t + 256, ... of every sub-quantizer (coalesced reads of T32[tab][j][k])
            const int64_t t0 = (int64_t)it.tab0 * nf * K, t1 = (int64_t)it.tab1 * nf * K;
            for (int k = tid; k < K; k += 64 * CIS_STREAM_NW) {
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const float e = on ? (j < nf ? tab_f1(T32, T, t0 + j * K + k) : tab_f1(T32, T, t1 + (j - nf) * K + k)) : 0.f;
#pragma unroll
                    for (int c = 0; c < R; ++c) s_tab[(((size_t)k * R + c) * M + j) * G + g] = e;
                }
            }
        }
        __syncthreads();
        float mn[G];
#pragma unroll
        for (int g = 0; g < G; ++g) mn[g] = __uint_as_float(0x7f800000u);
        for (int r0 = rfirst; r0 < rows; r0 += rstep * U) {
            // The ring: the rows of iteration i + 1 are requested before those of iteration i are used, so a wave has U .. 2 U kilobytes
            // on their way at any time instead of none while it gathers (round 5: without it ~40 % of the waves had requests in
            // flight, 4.8-5.0 TB/s = what ~8 MB in flight sustain at the loaded latency).
            if (CIS_STREAM_RING) request(r0 + rstep * U, cn);
            else request(r0, cw);
            // all U * CPL candidates' distances first (their gathers overlap), the rare appends afterwards
            float d[U * CPL][G];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    uint32_t w[(M + 3) / 4];
#pragma unroll
                    for (int x = 0; x < (M + 3) / 4; ++x) w[x] = cw[u][c * ((M + 3) / 4) + x];
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
                        // byte of sub-quantizer j(t, lane), then the byte address of entry (byte, j): two VALU instructions
                        uint32_t byte, addr;
                        asm("v_bfe_u32 %0, %1, %2, 8" : "=v"(byte) : "v"(wsel[th]), "v"(rc.sh[tq]));
                        if constexpr (ROWSH == 7) asm("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(addr) : "v"(byte), "v"(cjb[t]));
                        else if constexpr (ROWSH == 6) asm("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(addr) : "v"(byte), "v"(cjb[t]));
                        else if constexpr (ROWSH == 5) asm("v_lshl_add_u32 %0, %1, 5, %2" : "=v"(addr) : "v"(byte), "v"(cjb[t]));
                        else asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(addr) : "v"(byte), "v"(cjb[t]));
                        const char* ep = reinterpret_cast<const char*>(s_tab) + addr;
                        if constexpr (G == 1) {
                            const float e = *reinterpret_cast<const float*>(ep);
                            d[u * CPL + c][0] = t == 0 ? e : d[u * CPL + c][0] + e;
                        } else {
                            const f32x2_t e = *reinterpret_cast<const f32x2_t*>(ep);
                            d[u * CPL + c][0] = t == 0 ? e[0] : d[u * CPL + c][0] + e[0];
                            d[u * CPL + c][1] = t == 0 ? e[1] : d[u * CPL + c][1] + e[1];
                        }
                    }
                }
            }
            if (SAMPLE) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const bool valid = (r0 + u * rstep) * ROW + lane * CPL + c < len;
#pragma unroll
                        for (int g = 0; g < G; ++g) mn[g] = (valid && d[u * CPL + c][g] < mn[g]) ? d[u * CPL + c][g] : mn[g];
                    }
            } else {
                bool any = false;
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
#pragma unroll
                        for (int g = 0; g < G; ++g) any = any || d[u * CPL + c][g] <= tg[g];
                if (__builtin_amdgcn_ballot_w64(any) != 0ull) {  // rare: a few thousand candidates of hundreds of millions pass
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int c = 0; c < CPL; ++c) {
                            const int p = (r0 + u * rstep) * ROW + lane * CPL + c;
#pragma unroll
                            for (int g = 0; g < G; ++g)
                                if (p < len && d[u * CPL + c][g] <= tg[g]) {
                                    const int j = atomicAdd(&cnt[qg[g]], 1);
                                    if (j < cap) surv[(int64_t)qg[g] * cap + j] = rbase[g] + (uint32_t)p;
                                }
                        }
                }
            }
            if (CIS_STREAM_RING) {
#pragma unroll
                for (int u = 0; u < U; ++u) cw[u] = cn[u];
            }
        }
        if (SAMPLE) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (qg[g] >= 0 && f2u(mn[g]) < 0x7f800000u)
                    atomicMin(&bmin[(int64_t)qg[g] * B + (((unsigned)s * (64u * CIS_STREAM_NW) + (unsigned)tid) & (unsigned)(B - 1))], f2u(mn[g]));
        }
    }
}

// tau[q] = the k-th smallest of the query's B bucket minima (bisection on the bit patterns: non-negative floats order like their
// bits); +inf when fewer than k buckets were touched.  One workgroup of 1024 threads per query, B / 1024 values per thread.
template <int PER>
__global__ __launch_bounds__(1024) void k_stream_tau(const uint32_t* __restrict__ bmin, int B, int k, float* __restrict__ tau) {
    __shared__ int s_c[2][16];
    __shared__ uint32_t s_mm[2];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t v[PER];
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = bmin[(int64_t)q * B + i * 1024 + tid];
        lo = v[i] < lo ? v[i] : lo;
        if (v[i] < 0x7f800000u) hi = v[i] > hi ? v[i] : hi;
    }
    if (tid == 0) { s_mm[0] = 0xffffffffu; s_mm[1] = 0u; }
    __syncthreads();
    atomicMin(&s_mm[0], lo);
    atomicMax(&s_mm[1], hi);
    __syncthreads();
    lo = s_mm[0];
    hi = s_mm[1];
    int it = 0;
    bool enough = true;
    {   // finite buckets >= k ?
        int c = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) c += v[i] < 0x7f800000u ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) s_c[0][wv] = c;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += s_c[0][w];
        enough = tot >= k;
        it = 1;
    }
    if (!enough) {
        if (tid == 0) tau[q] = __uint_as_float(0x7f800000u);
        return;
    }
    // (hi always has >= k buckets at or below it.  Sixteen halvings of [min, max] of the bucket minima -- floats of one or two
    // binades -- leave an interval of ~1e-5 of the value: tau = hi then admits a few more candidates than the exact k-th smallest
    // would, which only lengthens the list; the proof does not care where tau came from)
    for (int step = 0; step < 16 && lo < hi; ++step) {  // uniform over the workgroup
        const uint32_t p = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) c += v[i] <= p ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) s_c[it & 1][wv] = c;
        __syncthreads();  // (two buffers: a wave that runs ahead writes the other one)
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += s_c[it & 1][w];
        if (tot >= k) hi = p; else lo = p + 1;
        ++it;
    }
    if (tid == 0) tau[q] = __uint_as_float(hi);
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

// The proof of the header comment, per query.  status[0] = queries whose proof failed, status[1] = lists that overflowed;
// the words land in pinned host memory (status_host) behind a sequence number.
__global__ void k_stream_verify(const uint64_t* __restrict__ sel_keys, const int* __restrict__ nsel, int64_t stride, const int* __restrict__ cnt, int cap,
                                const int64_t* __restrict__ seg, const float* __restrict__ tau, int nq, int L, double eps,
                                int* __restrict__ status, volatile int64_t* __restrict__ status_host, int64_t seq) {
    __shared__ int s_bad[2];
    if (threadIdx.x == 0) { s_bad[0] = 0; s_bad[1] = 0; }
    __syncthreads();
    for (int q = threadIdx.x; q < nq; q += blockDim.x) {
        const int64_t ncand = seg[q + 1] - seg[q];
        const int c = cnt[q];
        if (c > cap) { atomicAdd(&s_bad[1], 1); continue; }
        if ((int64_t)c == ncand) continue;  // every candidate was listed (tau = +inf, or a short query)
        const int nv = nsel[q];
        bool ok = nv == L;                    // (c < ncand and fewer than L listed: the threshold was too tight)
        if (ok) {
            const double Bd = __longlong_as_double((long long)sel_keys[(int64_t)q * stride + nv - 1]);
            ok = Bd * (1.0 + 2.0 * eps) <= (double)tau[q];
        }
        if (!ok) atomicAdd(&s_bad[0], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        status[0] = s_bad[0];
        status[1] = s_bad[1];
        status_host[0] = s_bad[0];
        status_host[1] = s_bad[1];
        __threadfence_system();
        status_host[2] = seq;
        __threadfence_system();
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
bool stream_supported(int M, int K, int L) { return (M == 4 || M == 8 || M == 16) && K <= 256 && L >= 1 && L <= 1024; }

size_t stream_lds(int M, int K, int G) {
    const int rmax = (32 / G) / M > 0 ? (32 / G) / M : 1;
    const int R = CIS_STREAM_REPL == 1 ? rmax : (CIS_STREAM_REPL == 2 ? (rmax >= 2 ? 2 : 1) : 1);
    return (size_t)K * R * M * G * sizeof(float);
}

template <int M, int G, bool SAMPLE>
static void launch_stream_t(int grid, hipStream_t st, const WorkItem* items, const int* slots, const int* n_slots, const float* T32, const double* T,
                            const uint8_t* codes, int K, const int64_t* cand_start, const int64_t* seg, const float* tau, uint32_t* bmin, int B,
                            int sample_stride, uint32_t* surv, int* cnt, int cap) {
    hipLaunchKernelGGL((k_adc_stream<M, G, SAMPLE>), dim3((unsigned)grid), dim3(64 * CIS_STREAM_NW), stream_lds(M, K, G), st, items, slots, n_slots, T32, T, codes, K,
                       cand_start, seg, tau, bmin, B, sample_stride, surv, cnt, cap);
}

void launch_stream_scan(int M, int G, bool sample, int grid, hipStream_t st, const WorkItem* items, const int* slots, const int* n_slots,
                        const float* T32, const double* T, const uint8_t* codes, int K, const int64_t* cand_start, const int64_t* seg, const float* tau,
                        uint32_t* bmin, int B, int sample_stride, uint32_t* surv, int* cnt, int cap) {
#define CIS_STREAM(MM, GG)                                                                                                              \
    if (M == MM && G == GG) {                                                                                                           \
        if (sample) launch_stream_t<MM, GG, true>(grid, st, items, slots, n_slots, T32, T, codes, K, cand_start, seg, tau, bmin, B,         \
                                                  sample_stride, surv, cnt, cap);                                                       \
        else launch_stream_t<MM, GG, false>(grid, st, items, slots, n_slots, T32, T, codes, K, cand_start, seg, tau, bmin, B,               \
                                            sample_stride, surv, cnt, cap);                                                             \
        return;                                                                                                                         \
    }
    CIS_STREAM(8, 1) CIS_STREAM(8, 2) CIS_STREAM(4, 1) CIS_STREAM(4, 2) CIS_STREAM(16, 1) CIS_STREAM(16, 2)
#undef CIS_STREAM
}

// workgroups of the persistent launch: what the chip holds at once (registers, LDS: the occupancy API), at most one per slot
int stream_grid(int M, int G, int K, int64_t max_slots) {
    static int per_cu[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    const int mi = M == 4 ? 0 : (M == 8 ? 1 : 2), gi = G - 1;
    if (per_cu[mi][gi] == 0) {
        int nb = 0;
        hipError_t e = hipErrorUnknown;
        const size_t lds = stream_lds(M, 256, G);
#define CIS_OCC(MM, GG) if (M == MM && G == GG) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_adc_stream<MM, GG, false>, 64 * CIS_STREAM_NW, lds);
        CIS_OCC(4, 1) CIS_OCC(4, 2) CIS_OCC(8, 1) CIS_OCC(8, 2) CIS_OCC(16, 1) CIS_OCC(16, 2)
#undef CIS_OCC
        per_cu[mi][gi] = (e == hipSuccess && nb > 0) ? (nb > 8 ? 8 : nb) : 4;
    }
    (void)K;
    int64_t g = (int64_t)256 * per_cu[mi][gi];
    if (const char* e = getenv("CIS_STREAM_GRID")) g = atoll(e) > 0 ? atoll(e) : g;  // experiments
    g = g < max_slots ? g : max_slots;
    return (int)(g < 1 ? 1 : g);
}

void launch_stream_init(hipStream_t st, uint32_t* bmin, int64_t n_b, int* cnt, int nq, int* status) {
    const int64_t n = n_b > nq ? n_b : nq;
    hipLaunchKernelGGL(k_stream_init, dim3((unsigned)ceil_div(n < 4 ? 4 : n, 256)), dim3(256), 0, st, bmin, n_b, cnt, nq, status);
}

void launch_stream_tau(hipStream_t st, const uint32_t* bmin, int B, int k, int nq, float* tau) {
    // B is 16384 (PER = 16)
    hipLaunchKernelGGL(k_stream_tau<16>, dim3((unsigned)nq), dim3(1024), 0, st, bmin, B, k, tau);
}

void launch_stream_keys(int M, hipStream_t st, const WorkItem* items, const int64_t* cand_start, const int64_t* seg, const int64_t* item_off,
                        int64_t n_items, const double* T, const uint8_t* codes, int K, const uint32_t* surv, const int* cnt, int cap, int nq,
                        uint64_t* keys, unsigned long long* qmin, unsigned long long* qmax) {
    const dim3 g((unsigned)ceil_div(cap, 256), (unsigned)nq);
    if (M == 4) hipLaunchKernelGGL(k_stream_keys<4>, g, dim3(256), 0, st, items, cand_start, seg, item_off, n_items, T, codes, K, surv, cnt, cap, keys, qmin, qmax);
    else if (M == 8) hipLaunchKernelGGL(k_stream_keys<8>, g, dim3(256), 0, st, items, cand_start, seg, item_off, n_items, T, codes, K, surv, cnt, cap, keys, qmin, qmax);
    else hipLaunchKernelGGL(k_stream_keys<16>, g, dim3(256), 0, st, items, cand_start, seg, item_off, n_items, T, codes, K, surv, cnt, cap, keys, qmin, qmax);
}

void launch_stream_verify(hipStream_t st, const uint64_t* sel_keys, const int* nsel, int64_t stride, const int* cnt, int cap, const int64_t* seg,
                          const float* tau, int nq, int L, int M, int* status, int64_t* status_host_dev, int64_t seq) {
    const double eps = 2.0 * M * 5.9604644775390625e-08;  // 2 M 2^-24
    hipLaunchKernelGGL(k_stream_verify, dim3(1), dim3(256), 0, st, sel_keys, nsel, stride, cnt, cap, seg, tau, nq, L, eps, status, status_host_dev, seq);
}
