// Stable segmented sort of (float64 distance bits, candidate reference) pairs, one segment per query: the ranking of the large-`limit`
// search path (limit above what the LDS top-k kernels hold, or the reference's default limit = quota: lopq/lopq/search.py:210-216 is
// the stable sorted() of ALL retrieved candidates).  Rounds 1-4 called rocPRIM's segmented radix sort here; this is the library's own
// (round 5): a merge sort, ONE workgroup per segment --
//   1. tiles of 4096 pairs are sorted in LDS (bitonic network on (key, position in the segment): the position makes every pair unique,
//      so the order is the stable one whatever the network does with equal keys);
//   2. runs are merged pairwise, doubling, between two global buffers (they stay in L2): a chunk of 4096 outputs finds its two input
//      ranges by a merge-path search, stages them in LDS, and every element computes its output place = its place in its own range +
//      the number of elements of the other range in front of it (binary search in LDS) -- no serial merge anywhere;
//   3. the values follow their positions in one gather at the end.
// Cost per pair: one LDS sort + ceil(log2(n / 4096)) passes of 12 bytes read and written.  A batch of 8192 queries at limit = quota =
// 10000 (three tiles, two passes per segment) is what it is sized for; a single segment of millions of pairs is served correctly, by
// one workgroup (the exhaustive quota with limit = None: seconds of the reference's time either way).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

#include "common.h"

static const int SS_T = 4096;  // pairs per LDS tile / outputs per merge chunk

static __device__ __forceinline__ bool ss_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) { return ka < kb || (ka == kb && ia < ib); }

__global__ __launch_bounds__(1024) void k_seg_sort(const uint64_t* __restrict__ keys_in, uint64_t* __restrict__ keys_out,
                                                   const uint64_t* __restrict__ vals_in, uint64_t* __restrict__ vals_out,
                                                   const int64_t* __restrict__ seg_begin, const int64_t* __restrict__ seg_end,
                                                   uint64_t* __restrict__ tkeys, uint32_t* __restrict__ tidx0, uint32_t* __restrict__ tidx1) {
    __shared__ uint64_t s_k[SS_T];
    __shared__ uint32_t s_i[SS_T];
    const int tid = threadIdx.x;
    const int64_t a = seg_begin[blockIdx.x];
    const int64_t n64 = seg_end[blockIdx.x] - a;
    if (n64 <= 0) return;
    const uint32_t n = (uint32_t)n64;
    int P = 0;  // merge passes
    while (((uint64_t)SS_T << P) < n) ++P;
    // buffers: K[0] = keys_out, K[1] = tkeys; I[0] = tidx0, I[1] = tidx1 (all at the segment's own positions).  Pass p reads side
    // (P - p) & 1 and writes the other, so that the last pass writes side 0 = keys_out; the tile sort writes side P & 1.
    uint64_t* K[2] = {keys_out + a, tkeys + a};
    uint32_t* I[2] = {tidx0 + a, tidx1 + a};
    // ---- 1. tiles ---------------------------------------------------------------------------------------------------------------
    {
        uint64_t* dk = K[P & 1];
        uint32_t* di = I[P & 1];
        for (uint32_t t0 = 0; t0 < n; t0 += SS_T) {
            const uint32_t m = n - t0 < (uint32_t)SS_T ? n - t0 : (uint32_t)SS_T;
            int n2 = 64;
            while ((uint32_t)n2 < m) n2 <<= 1;
            __syncthreads();
            for (int x = tid; x < n2; x += 1024) {
                s_k[x] = (uint32_t)x < m ? keys_in[a + t0 + x] : ~0ull;
                s_i[x] = (uint32_t)x < m ? t0 + (uint32_t)x : 0xffffffffu;
            }
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    __syncthreads();
                    for (int t = tid; t < (n2 >> 1); t += 1024) {
                        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                        const bool up = (lo & k) == 0;
                        const uint64_t ka = s_k[lo], kb = s_k[hi];
                        const uint32_t ia = s_i[lo], ib = s_i[hi];
                        if (ss_less(kb, ib, ka, ia) == up) { s_k[lo] = kb; s_k[hi] = ka; s_i[lo] = ib; s_i[hi] = ia; }
                    }
                }
            __syncthreads();
            for (uint32_t x = tid; x < m; x += 1024) { dk[t0 + x] = s_k[x]; di[t0 + x] = s_i[x]; }
        }
    }
    // ---- 2. merge passes -------------------------------------------------------------------------------------------------------
    for (int p = 0; p < P; ++p) {
        __threadfence_block();
        __syncthreads();
        const uint64_t* sk = K[(P - p) & 1];
        const uint32_t* si = I[(P - p) & 1];
        uint64_t* dk = K[(P - p + 1) & 1];
        uint32_t* di = I[(P - p + 1) & 1];
        const uint64_t R = (uint64_t)SS_T << p;
        for (uint64_t base = 0; base < n; base += 2 * R) {
            const uint32_t na = (uint32_t)(n - base < R ? n - base : R);
            const uint32_t nb = (uint32_t)(n - base - na < R ? n - base - na : R);
            const uint64_t* ak = sk + base;
            const uint32_t* ai = si + base;
            const uint64_t* bk = ak + na;
            const uint32_t* bi = ai + na;
            if (nb == 0) {  // an odd run at the end: carried over
                for (uint32_t x = tid; x < na; x += 1024) { dk[base + x] = ak[x]; di[base + x] = ai[x]; }
                continue;
            }
            // how many elements of A are among the first d outputs of the pair (merge path; A before B on a tie cannot happen: pairs are unique)
            auto corank = [&](uint32_t d) -> uint32_t {
                uint32_t lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ss_less(ak[mid], ai[mid], bk[d - 1 - mid], bi[d - 1 - mid])) lo = mid + 1; else hi = mid;
                }
                return lo;
            };
            for (uint32_t d0 = 0; d0 < na + nb; d0 += SS_T) {
                const uint32_t d1 = d0 + SS_T < na + nb ? d0 + SS_T : na + nb;
                const uint32_t i0 = corank(d0), i1 = corank(d1);
                const uint32_t j0 = d0 - i0, j1 = d1 - i1;
                const uint32_t la = i1 - i0, lb = j1 - j0;
                __syncthreads();  // the previous chunk's readers are done with the stage
                for (uint32_t x = tid; x < la + lb; x += 1024) {
                    s_k[x] = x < la ? ak[i0 + x] : bk[j0 + (x - la)];
                    s_i[x] = x < la ? ai[i0 + x] : bi[j0 + (x - la)];
                }
                __syncthreads();
                for (uint32_t x = tid; x < la + lb; x += 1024) {
                    const uint64_t kx = s_k[x];
                    const uint32_t ix = s_i[x];
                    uint32_t lo, hi;  // elements of the OTHER range in front of this one
                    if (x < la) { lo = la; hi = la + lb; } else { lo = 0; hi = la; }
                    const uint32_t org = lo;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (ss_less(s_k[mid], s_i[mid], kx, ix)) lo = mid + 1; else hi = mid;
                    }
                    const uint32_t rank = (x < la ? x : x - la) + (lo - org);
                    dk[base + d0 + rank] = kx;
                    di[base + d0 + rank] = ix;
                }
            }
        }
    }
    // ---- 3. the values follow -----------------------------------------------------------------------------------------------------
    __threadfence_block();
    __syncthreads();
    const uint32_t* fi = I[0];
    for (uint32_t x = tid; x < n; x += 1024) vals_out[a + x] = vals_in[a + fi[x]];
}

// Two-call convention: temp == nullptr -> only *temp_bytes is set.  keys_in / vals_in are not modified; segments must not overlap.
int cis_seg_sort_u64(void* temp, size_t* temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint64_t* vals_in,
                     uint64_t* vals_out, int64_t n, int nseg, const int64_t* seg_begin, const int64_t* seg_end, hipStream_t st) {
    CIS_REQUIRE(n >= 0 && n < ((int64_t)1 << 32) && nseg >= 0, "segmented sort: size out of range");
    const size_t need = (size_t)(n + 1) * 16 + 64;
    if (temp == nullptr) {
        *temp_bytes = need;
        return CIS_OK;
    }
    CIS_REQUIRE(*temp_bytes >= need, "segmented sort: temporary storage too small");
    if (nseg == 0 || n == 0) return CIS_OK;
    uint64_t* tkeys = reinterpret_cast<uint64_t*>(temp);
    uint32_t* tidx0 = reinterpret_cast<uint32_t*>(tkeys + (n + 1));
    uint32_t* tidx1 = tidx0 + (n + 1);
    hipLaunchKernelGGL(k_seg_sort, dim3((unsigned)nseg), dim3(1024), 0, st, keys_in, keys_out, vals_in, vals_out, seg_begin, seg_end, tkeys, tidx0, tidx1);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}
