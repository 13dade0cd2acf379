// LOPQ index storage on MI355X: the HBM-resident, cell-contiguous index and its DEVICE-SIDE insert.
//
// Replaces lopq/lopq/search.py:325-382 (LOPQSearcher.add_codes / get_cell: a dict of per-cell lists, first (cell, id)
// wins, insertion order inside a cell) and the refresh loop that feeds it
// (cufacesearch/cufacesearch/searcher/searcher_lopqhbase.py:743-758).
//
// Data layout in HBM (one CellStore): codes [n][M] uint8, ids [n] int64, both cell-contiguous (CSR over the V*V coarse
// cells, `loff`), inside a cell in insertion order; cmax [V*V] = 1 + the largest id of the cell.  There is no host copy.
//
// An insert of n items (host arrays are uploaded first; device arrays are used where they are) is a stable merge, all
// kernels on one stream:
//   k_ins_keys      item -> key (its cell) / dead (invalid code, or a cell of another shard), id or -1
//   radix passes    stable LSD sort of (key, arrival index) -- k_rs_hist / scan / k_rs_scatter, 8 bits per pass
//   k_ins_gather    ids in sorted order
//   k_ins_dedup     first (cell, id) wins (search.py:349-364): an item is dropped when its id is already stored in the
//                   cell (looked up only when id <= cmax[cell]: ids that grow with time never scan) or when an earlier
//                   item of the batch has the same cell and id; all lanes of a wave walk the same cell -> broadcast loads
//   scan            exclusive prefix of the accepted flags
//   k_ins_counts    per cell: items after the insert, room of the rebuilt cell (items + slack); scan -> new offsets,
//                   global cell sizes += accepted
//   k_ins_move      old items to their new places (a cell moves as a block)
//   k_ins_scatter   accepted items behind the old items of their cell, in arrival order: place = old offset of the next cell +
//                   accepted items before it
// The two generations of the arrays swap.  Cost: one read + one write of the shard's index (24 B per item at M = 16)
// plus O(n) for the batch -- 10M items x M = 8: ~0.1 ms; the host does nothing per item and keeps no (cell, id) set.
//
// IN-PLACE inserts (round 4).  The reference appends to a per-cell Python list (search.py:349-364): O(batch).  A cell here owns
// [loff[c], loff[c+1]) of the arrays but uses only [loff[c], lend[c]) of it (lend = loff + ncells + 1, same buffer): the rebuild above
// leaves cnt / 8 + a few items of slack behind every cell, and a batch whose accepted items all fit their cells' slack is
// written behind the cells' last items by two kernels (k_ins_place, k_ins_commit) that touch O(batch) bytes -- no move, no
// generation swap.  A batch that does not fit raises a flag BEFORE anything becomes visible (items land beyond lend, lend is
// advanced by the commit kernel only when the flag is clear) and takes the rebuild, which renews every cell's slack:
// geometric growth, amortised O(1) per item.  Same first-wins / insertion-order semantics on both routes.
//
// A cell-sharded index that is handed EVERY item with dedup (cis_index_add on all ranks -- tests and small set-ups; the
// production form is the routed insert of columbiaimagesearch_amd/distributed.py, where an owner sees only its cells)
// keeps the ids of the other shards' cells in a second, id-only store so that their duplicates are recognised too.
#include <algorithm>

#include "lopq_index.h"

// ================================================================================================
// device-wide exclusive scan (tile sums -> scan of the sums -> add)
// ================================================================================================
static const int SCAN_T = 256, SCAN_PER = 8, SCAN_TILE = SCAN_T * SCAN_PER;

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// exclusive scan of one value per thread over a block of SCAN_T threads; returns the block total in *total
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T* sh /* [SCAN_T / 64 + 1] */, T* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T inc = wave_incl_scan(v);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_T / 64; ++i) {
        const T s = sh[i];
        base += (i < w) ? s : (T)0;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(SCAN_T) void k_scan_tiles(const TI* __restrict__ in, TO* __restrict__ out, TO* __restrict__ sums, int64_t n) {
    __shared__ TO sh[SCAN_T / 64 + 1];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_PER;
    TO v[SCAN_PER];
    TO s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        v[i] = (base + i < n) ? (TO)in[base + i] : (TO)0;
        s += v[i];
    }
    TO tot;
    TO run = block_excl_scan<TO>(s, sh, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the tile sums in place, the grand total to total[0] (and total2[0], may be null)
template <typename TO>
__global__ __launch_bounds__(SCAN_T) void k_scan_sums(TO* __restrict__ sums, int64_t m, TO* __restrict__ total, int64_t* __restrict__ total2) {
    __shared__ TO sh[SCAN_T / 64 + 1];
    TO carry = 0;
    for (int64_t b = 0; b < m; b += SCAN_T) {
        const int64_t i = b + threadIdx.x;
        const TO v = i < m ? sums[i] : (TO)0;
        TO tot;
        const TO e = block_excl_scan<TO>(v, sh, &tot);
        if (i < m) sums[i] = carry + e;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        if (total) total[0] = carry;
        if (total2) total2[0] = (int64_t)carry;
    }
}

template <typename TO>
__global__ __launch_bounds__(SCAN_T) void k_scan_add(TO* __restrict__ out, const TO* __restrict__ sums, int64_t n) {
    const TO add = sums[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_PER;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i)
        if (base + i < n) out[base + i] += add;
}

// out[i] = sum of in[0 .. i); total (device, may be null) = sum of all; sums: scratch of ceil(n / SCAN_TILE) + 1 elements
template <typename TI, typename TO>
static void dev_exclusive_scan(const TI* in, TO* out, int64_t n, TO* sums, TO* total, int64_t* total2, hipStream_t st) {
    const int64_t m = n > 0 ? ceil_div(n, SCAN_TILE) : 0;
    if (m > 0) hipLaunchKernelGGL((k_scan_tiles<TI, TO>), dim3((unsigned)m), dim3(SCAN_T), 0, st, in, out, sums, n);
    hipLaunchKernelGGL((k_scan_sums<TO>), dim3(1), dim3(SCAN_T), 0, st, sums, m, total, total2);
    if (m > 1) hipLaunchKernelGGL((k_scan_add<TO>), dim3((unsigned)m), dim3(SCAN_T), 0, st, out, (const TO*)sums, n);
}

// ================================================================================================
// stable LSD radix sort of (key uint32, value uint32), 8 bits per pass, one wave per tile of 2048 items
// ================================================================================================
static const int RS_ROWS = 32, RS_TILE = RS_ROWS * 64;

__global__ __launch_bounds__(64) void k_rs_hist(const uint32_t* __restrict__ key, int64_t n, int shift, uint32_t* __restrict__ H, int nblk) {
    __shared__ uint32_t h[256];
    for (int i = threadIdx.x; i < 256; i += 64) h[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ROWS; ++r) {
        const int64_t i = base + r * 64 + threadIdx.x;
        if (i < n) atomicAdd(&h[(key[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 256; d += 64) H[(int64_t)d * nblk + blockIdx.x] = h[d];
}

// H holds the exclusive scan of the digit-major histogram: H[d * nblk + b] = first output place of tile b's digit d
__global__ __launch_bounds__(64) void k_rs_scatter(const uint32_t* __restrict__ key, const uint32_t* __restrict__ val, int64_t n, int shift,
                                                   const uint32_t* __restrict__ H, int nblk, uint32_t* __restrict__ key_out,
                                                   uint32_t* __restrict__ val_out) {
    __shared__ uint32_t base[256];
    for (int d = threadIdx.x; d < 256; d += 64) base[d] = H[(int64_t)d * nblk + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ROWS; ++r) {
        const int64_t i = b0 + r * 64 + lane;
        const bool ok = i < n;
        const uint32_t k = ok ? key[i] : 0u;
        const uint32_t v = ok ? val[i] : 0u;
        const uint32_t d = (k >> shift) & 255u;
        // lanes with the same digit (8 ballots), rows in order, lanes in order: stable
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0));
        const uint32_t start = base[d];
        __syncthreads();
        if (ok) {
            key_out[start + before] = k;
            val_out[start + before] = v;
            if (before == 0) base[d] = start + (uint32_t)__popcll(peers);
        }
        __syncthreads();
    }
}

// ================================================================================================
// insert kernels
// ================================================================================================
// statistics words (device memory; copied into pinned host memory when the host needs them)
enum { INS_ACC_OWN = 0, INS_INVALID = 1, INS_NTOTAL = 2, INS_MAXCELL = 3, INS_NONEMPTY = 4, INS_ERR = 5, INS_ACC_GHOST = 6,
       INS_REMOTE_PLAIN = 7, INS_OVERFLOW = 8, INS_WORDS = 9 };

__device__ __forceinline__ bool dev_owns(int64_t cell, const int32_t* __restrict__ owner, int rank, int world) {
    if (world <= 1) return true;
    return owner ? owner[cell] == rank : (int)(cell % world) == rank;
}

// sel 0: the own store (items of this shard's cells; items of other shards are counted when dedup == 0)
// sel 1: the ghost store (valid items of the other shards' cells)
__global__ void k_ins_keys(const int64_t* __restrict__ ids, const uint16_t* __restrict__ coarse, const uint8_t* __restrict__ fine,
                           int64_t n, int V, int M, int K, const int32_t* __restrict__ owner, int rank, int world, int sel,
                           int dedup, uint32_t* __restrict__ key, uint32_t* __restrict__ val, int64_t* __restrict__ idv,
                           int64_t* __restrict__ gcount, int64_t* __restrict__ stats) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c0 = coarse[2 * i], c1 = coarse[2 * i + 1];
    const int64_t id = ids[i];
    bool valid = c0 < V && c1 < V && id >= 0;
    if (valid && K < 256) {
        for (int j = 0; j < M; ++j) valid = valid && fine[i * M + j] < K;
    }
    uint32_t k = 0;
    int64_t keep = -1;
    if (!valid) {
        if (sel == 0) atomicAdd((unsigned long long*)&stats[INS_INVALID], 1ull);
    } else {
        const int64_t cell = (int64_t)c0 * V + c1;
        const bool mine = dev_owns(cell, owner, rank, world);
        if (mine == (sel == 0)) {
            k = (uint32_t)cell;
            keep = id;
        } else if (sel == 0 && !dedup) {
            atomicAdd((unsigned long long*)&gcount[cell], 1ull);
            atomicAdd((unsigned long long*)&stats[INS_REMOTE_PLAIN], 1ull);
        }
    }
    key[i] = k;
    val[i] = (uint32_t)i;
    idv[i] = keep;
}

__global__ void k_ins_gather(const int64_t* __restrict__ idv, const uint32_t* __restrict__ perm, int64_t n, int64_t* __restrict__ sid) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) sid[j] = idv[perm[j]];
}

__device__ __forceinline__ int64_t lower_bound_u32(const uint32_t* __restrict__ a, int64_t n, uint32_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// first (cell, id) wins (search.py:349-364).  One WAVE per item (round 3: one thread walked the whole cell -- 39 k ids at 10M
// vectors, 4.9 ms per 256-item batch whose ids lie below the cell's maximum): the lanes stride over the ids stored in the cell
// (skipped when id >= cmax[cell]: ids that grow with time never scan) and over the earlier items of the batch in the same cell.
__global__ __launch_bounds__(256) void k_ins_dedup(const uint32_t* __restrict__ key, const int64_t* __restrict__ sid, int64_t n, int dedup,
                                                   const int64_t* __restrict__ loff, const int64_t* __restrict__ lend,
                                                   const int64_t* __restrict__ old_ids, const unsigned long long* __restrict__ cmaxp,
                                                   uint32_t* __restrict__ acc) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n) return;
    const int64_t id = sid[j];
    uint32_t a = id >= 0 ? 1u : 0u;
    if (a && dedup) {
        const uint32_t c = key[j];
        bool dup = false;
        if ((unsigned long long)id < cmaxp[c]) {  // ids above everything stored in the cell cannot be there
            const int64_t b = loff[c], e = lend[c];
            for (int64_t p0 = b; p0 < e && !dup; p0 += 64) {
                const int64_t p = p0 + lane;
                dup = __ballot(p < e && old_ids[p] == id) != 0ull;
            }
        }
        if (!dup) {
            const int64_t first = lower_bound_u32(key, n, c);  // the batch is sorted by cell: earlier items of this cell are [first, j)
            for (int64_t j0 = first; j0 < j && !dup; j0 += 64) {
                const int64_t jj = j0 + lane;
                dup = __ballot(jj < j && sid[jj] == id) != 0ull;
            }
        }
        a = dup ? 0u : 1u;
    }
    if (lane == 0) acc[j] = a;
}

// thread c in [0, ncells): items of the cell after this insert and the room the rebuilt cell gets (used part + slack);
// a0[c] = accepted items of the cells before c in the sorted batch; the global cell sizes take the accepted items
__global__ void k_ins_counts(const uint32_t* __restrict__ key, const uint32_t* __restrict__ apre, const uint32_t* __restrict__ total,
                             int64_t n, int64_t ncells, const int64_t* __restrict__ loff, const int64_t* __restrict__ lend,
                             int64_t* __restrict__ cnt_new, int64_t* __restrict__ cap_new, uint32_t* __restrict__ a0v,
                             int64_t* __restrict__ gcount, int slack_const) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const uint32_t tot = total[0];
    const int64_t l0 = lower_bound_u32(key, n, (uint32_t)c);
    const int64_t a0 = l0 < n ? apre[l0] : tot;
    const int64_t l1 = lower_bound_u32(key, n, (uint32_t)c + 1u);
    const int64_t a1 = l1 < n ? apre[l1] : tot;
    const int64_t cn = (lend[c] - loff[c]) + (a1 - a0);
    cnt_new[c] = cn;
    // (an empty cell gets room too, or the first item of a cell would always force a rebuild: the full constant with few cells, one
    // slot each with millions of them -- V = 4096 has 16 M cells, most of them empty)
    cap_new[c] = slack_const < 0 ? cn : cn + (cn >> 3) + ((cn > 0 || ncells <= 65536) ? slack_const : 1);
    a0v[c] = (uint32_t)a0;
    if (a1 > a0) gcount[c] += a1 - a0;
}

// noff[0 .. ncells] is the exclusive scan of cap_new (written by the scan); the used ends follow it in the same buffer
__global__ void k_ins_ends(const int64_t* __restrict__ noff, const int64_t* __restrict__ cnt_new, int64_t ncells, int64_t* __restrict__ nend) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncells) nend[c] = noff[c] + cnt_new[c];
}

// old items to their new places: a cell moves as a block (thread per SLOT of the old layout; the cell by binary search in the
// old offsets; slots of a cell's slack hold nothing)
template <int MW /* code words per item, 0: bytes */>
__global__ void k_ins_move(const int64_t* __restrict__ loff, const int64_t* __restrict__ lend, const int64_t* __restrict__ noff,
                           int64_t ncells, int64_t cap_bound, const int64_t* __restrict__ ids, const uint8_t* __restrict__ codes, int M,
                           int64_t* __restrict__ ids_new, uint8_t* __restrict__ codes_new) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cap_bound || p >= loff[ncells]) return;
    int64_t lo = 0, hi = ncells;  // largest c with loff[c] <= p
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (loff[mid] <= p) lo = mid;
        else hi = mid;
    }
    if (p >= lend[lo]) return;
    const int64_t dst = p + (noff[lo] - loff[lo]);
    ids_new[dst] = ids[p];
    if (codes) {
        if constexpr (MW > 0) {
            const uint32_t* s = reinterpret_cast<const uint32_t*>(codes) + p * MW;
            uint32_t* d = reinterpret_cast<uint32_t*>(codes_new) + dst * MW;
#pragma unroll
            for (int i = 0; i < MW; ++i) d[i] = s[i];
        } else {
            for (int i = 0; i < M; ++i) codes_new[dst * M + i] = codes[p * M + i];
        }
    }
}

// many tiny cells (thousands of coarse clusters): thread per cell
__global__ void k_ins_move_cells(const int64_t* __restrict__ loff, const int64_t* __restrict__ lend, const int64_t* __restrict__ noff,
                                 int64_t ncells, const int64_t* __restrict__ ids, const uint8_t* __restrict__ codes, int M,
                                 int64_t* __restrict__ ids_new, uint8_t* __restrict__ codes_new) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const int64_t a = loff[c], b = lend[c], d = noff[c] - a;
    for (int64_t p = a; p < b; ++p) {
        ids_new[p + d] = ids[p];
        if (codes)
            for (int i = 0; i < M; ++i) codes_new[(p + d) * M + i] = codes[p * M + i];
    }
}

// Place of accepted item j (sorted by cell, then arrival): behind the items the cell already holds and behind the accepted
// items of the same cell before it -- base[c] + used[c] + (apre[j] - apre[first item of the cell in the batch]).
//   rebuild (INPLACE = false): base = the new offsets, used = the old used length (the moved items)
//   in place (INPLACE = true): base = the offsets, used ends = lend; an item that would land beyond the cell's room raises
//   stats[INS_OVERFLOW] and writes nothing; nothing it writes is visible before k_ins_commit advances lend.
// The cell's largest id: one atomic per run of equal cells in a wave (the items are sorted by cell: a wave usually holds one
// cell, and 10M items hammering 256 addresses one by one took 22 ms).
template <bool INPLACE>
__global__ void k_ins_scatter(const uint32_t* __restrict__ key, const uint32_t* __restrict__ perm, const int64_t* __restrict__ sid,
                              const uint32_t* __restrict__ acc, const uint32_t* __restrict__ apre, int64_t n,
                              const int64_t* __restrict__ loff_old, const int64_t* __restrict__ lend_old,
                              const int64_t* __restrict__ noff, const uint32_t* __restrict__ a0v, const uint8_t* __restrict__ fine, int M,
                              int64_t* __restrict__ ids_new, uint8_t* __restrict__ codes_new, unsigned long long* __restrict__ cmaxp,
                              int64_t* __restrict__ stats) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = j < n && acc[j];
    const uint32_t c = on ? key[j] : 0xffffffffu;
    unsigned long long v = 0ull;
    if (on) {
        int64_t dst;
        bool fits = true;
        if constexpr (INPLACE) {
            const int64_t l0 = lower_bound_u32(key, n, c);
            dst = lend_old[c] + (int64_t)(apre[j] - apre[l0]);
            fits = dst < loff_old[c + 1];
            if (!fits) stats[INS_OVERFLOW] = 1;
        } else {
            dst = noff[c] + (lend_old[c] - loff_old[c]) + (int64_t)(apre[j] - a0v[c]);
        }
        const int64_t id = sid[j];
        if (fits) {
            ids_new[dst] = id;
            if (codes_new) {
                const int64_t i = perm[j];
                for (int b = 0; b < M; ++b) codes_new[dst * M + b] = fine[i * M + b];
            }
        }
        v = (unsigned long long)id + 1ull;
    }
    // segmented maximum over runs of equal cells inside the wave (keys are sorted, so runs are contiguous lanes)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(v, o);
        const uint32_t cu = __shfl_up(c, o);
        if (lane >= o && cu == c && u > v) v = u;
    }
    const uint32_t cn = __shfl_down(c, 1);
    if (on && (lane == 63 || cn != c)) atomicMax(&cmaxp[c], v);  // the last lane of a run holds the run's maximum
}

// in-place insert, second kernel: the accepted items of a cell become visible -- lend and the global cell size advance by the
// cell's accepted count -- unless some item of the batch did not fit (then the rebuild takes the whole batch)
__global__ void k_ins_commit(const uint32_t* __restrict__ key, const uint32_t* __restrict__ acc, const uint32_t* __restrict__ apre, int64_t n,
                             int64_t* __restrict__ lend, int64_t* __restrict__ gcount, const int64_t* __restrict__ stats) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || stats[INS_OVERFLOW] != 0) return;
    const uint32_t c = key[j];
    if (j + 1 < n && key[j + 1] == c) return;  // the last item of the cell's run commits the run
    const int64_t l0 = lower_bound_u32(key, n, c);
    const int64_t a = (int64_t)(apre[j] + acc[j]) - (int64_t)apre[l0];
    if (a > 0) {
        lend[c] += a;
        gcount[c] += a;
    }
}

// ---- small batches (round 5): a 256-item insert was twelve launches of 5-7 us each behind each other (keys, four radix passes, gather,
// dedup, three scan kernels, scatter, commit, memset, statistics).  Up to 1024 items ONE workgroup does the keys, a stable sort by
// cell in LDS and the gather (k_ins_small_sort), k_ins_dedup stays what it is (a wave per item walks a cell), and ONE workgroup does
// the prefix sums, the in-place scatter, the commit and the statistics of the global cell sizes (k_ins_small_place).  Same arrays,
// same order of the accepted items, same statistics words as the general path, which remains for larger batches and for the rebuild.
static const int INS_SMALL = 1024;

__global__ __launch_bounds__(1024) void k_ins_small_sort(const int64_t* __restrict__ ids, const uint16_t* __restrict__ coarse,
                                                         const uint8_t* __restrict__ fine, int n, int V, int M, int K,
                                                         const int32_t* __restrict__ owner, int rank, int world, int sel, int dedup,
                                                         uint32_t* __restrict__ skey, uint32_t* __restrict__ perm, int64_t* __restrict__ idv,
                                                         int64_t* __restrict__ sid, int64_t* __restrict__ gcount, int64_t* __restrict__ stats) {
    __shared__ uint64_t s_k[INS_SMALL];
    __shared__ int64_t s_id[INS_SMALL];
    const int i = threadIdx.x;
    uint32_t k = 0xffffffffu;  // padding sorts behind every item
    if (i < n) {
        const int c0 = coarse[2 * i], c1 = coarse[2 * i + 1];
        const int64_t id = ids[i];
        bool valid = c0 < V && c1 < V && id >= 0;
        if (valid && K < 256)
            for (int j = 0; j < M; ++j) valid = valid && fine[(int64_t)i * M + j] < K;
        int64_t keep = -1;
        k = 0;
        if (!valid) {
            if (sel == 0) atomicAdd((unsigned long long*)&stats[INS_INVALID], 1ull);
        } else {
            const int64_t cell = (int64_t)c0 * V + c1;
            const bool mine = dev_owns(cell, owner, rank, world);
            if (mine == (sel == 0)) {
                k = (uint32_t)cell;
                keep = id;
            } else if (sel == 0 && !dedup) {
                atomicAdd((unsigned long long*)&gcount[cell], 1ull);
                atomicAdd((unsigned long long*)&stats[INS_REMOTE_PLAIN], 1ull);
            }
        }
        s_id[i] = keep;
        idv[i] = keep;
    }
    s_k[i] = ((uint64_t)k << 32) | (uint32_t)i;  // the position makes every key unique: the network's order IS the stable order
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    for (int kk = 2; kk <= n2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            __syncthreads();
            if (i < (n2 >> 1)) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                const bool up = (lo & kk) == 0;
                const uint64_t a = s_k[lo], b = s_k[hi];
                if ((b < a) == up) { s_k[lo] = b; s_k[hi] = a; }
            }
        }
    __syncthreads();
    if (i < n) {
        const uint64_t v = s_k[i];
        const uint32_t src = (uint32_t)v;
        skey[i] = (uint32_t)(v >> 32);
        perm[i] = src;
        sid[i] = s_id[src];
    }
}

// One workgroup: apre = exclusive scan of acc (total behind it, added to stats[acc_word]); the in-place scatter of k_ins_scatter<true>;
// if nothing overflowed, the commit of k_ins_commit; the statistics of the global cell sizes (ncells <= STAT_CELLS: in here, else the
// caller launches k_gcount_stats).
__global__ __launch_bounds__(1024) void k_ins_small_place(const uint32_t* __restrict__ key, const uint32_t* __restrict__ perm,
                                                          const int64_t* __restrict__ sid, const uint32_t* __restrict__ acc,
                                                          uint32_t* __restrict__ apre, int n, int acc_word, const int64_t* __restrict__ loff,
                                                          int64_t* __restrict__ lend, const uint8_t* __restrict__ fine, int M,
                                                          int64_t* __restrict__ ids_cur, uint8_t* __restrict__ codes_cur,
                                                          unsigned long long* __restrict__ cmaxp, int64_t* __restrict__ gcount, int64_t ncells,
                                                          int do_stats, int64_t* __restrict__ stats) {
    __shared__ uint32_t s_pre[INS_SMALL + 1];
    __shared__ uint32_t s_key[INS_SMALL];
    __shared__ uint32_t s_w[17];
    __shared__ int s_over;
    __shared__ long long s_sum[16], s_max[16], s_ne[16];
    const int j = threadIdx.x, lane = j & 63, wv = j >> 6;
    if (j == 0) s_over = 0;
    const uint32_t a = j < n ? acc[j] : 0u;
    const uint32_t c = j < n ? key[j] : 0xffffffffu;
    s_key[j] = c;
    // exclusive scan of the 0 / 1 flags: ballot inside the wave, the waves' totals through LDS
    const unsigned long long b = __builtin_amdgcn_ballot_w64(a != 0u);
    const uint32_t before = (uint32_t)__builtin_popcountll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wv] = (uint32_t)__builtin_popcountll(b);
    __syncthreads();
    uint32_t wbase = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wv) wbase += s_w[w];
        total += s_w[w];
    }
    const uint32_t pre = wbase + before;
    s_pre[j] = pre;
    if (j < n) apre[j] = pre;
    if (j == 0) {
        apre[n] = total;  // `total`: one word behind the prefix array
        stats[acc_word] += (int64_t)total;
    }
    __syncthreads();
    // first item of this item's cell in the sorted batch (n <= 1024: a binary search in LDS)
    int l0 = 0;
    {
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_key[mid] < c) lo = mid + 1; else hi = mid;
        }
        l0 = lo;
    }
    const bool on = j < n && a != 0u;
    unsigned long long v = 0ull;
    if (on) {
        const int64_t dst = lend[c] + (int64_t)(pre - s_pre[l0]);
        const bool fits = dst < loff[c + 1];
        if (!fits) { s_over = 1; stats[INS_OVERFLOW] = 1; }
        const int64_t id = sid[j];
        if (fits) {
            ids_cur[dst] = id;
            if (codes_cur) {
                const int64_t i = perm[j];
                for (int bb = 0; bb < M; ++bb) codes_cur[dst * M + bb] = fine[i * M + bb];
            }
        }
        v = (unsigned long long)id + 1ull;
    }
    // the cell's largest id: segmented maximum over the runs of equal cells inside the wave.  Rejected items take no part (as in
    // k_ins_scatter, where they carry no cell): a run that ENDS with a rejected item must still publish its maximum.
    const uint32_t cm = on ? c : 0xffffffffu;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_up(v, o);
        const uint32_t cu = __shfl_up(cm, o);
        if (lane >= o && cu == cm && u > v) v = u;
    }
    const uint32_t cn = __shfl_down(cm, 1);
    if (on && (lane == 63 || cn != cm)) atomicMax(&cmaxp[c], v);
    __threadfence();   // the items are in place before a cell's used end moves
    __syncthreads();
    if (s_over != 0) return;  // nothing became visible: the caller's rebuild takes the whole batch
    if (j < n && (j + 1 >= n || s_key[j + 1] != c)) {  // the last item of a cell's run commits the run
        const int64_t add = (int64_t)(pre + a) - (int64_t)s_pre[l0];
        if (add > 0) {
            lend[c] += add;
            gcount[c] += add;
        }
    }
    if (!do_stats) return;
    __threadfence();
    __syncthreads();
    long long sum = 0, mx = 0, ne = 0;
    for (int64_t cc = j; cc < ncells; cc += 1024) {
        const long long g = gcount[cc];
        sum += g;
        mx = g > mx ? g : mx;
        ne += g > 0;
    }
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        const long long m2 = __shfl_xor(mx, o);
        mx = m2 > mx ? m2 : mx;
        ne += __shfl_xor(ne, o);
    }
    if (lane == 0) { s_sum[wv] = sum; s_max[wv] = mx; s_ne[wv] = ne; }
    __syncthreads();
    if (j == 0) {
        for (int w = 1; w < 16; ++w) { sum += s_sum[w]; mx = s_max[w] > mx ? s_max[w] : mx; ne += s_ne[w]; }
        stats[INS_NTOTAL] = sum; stats[INS_MAXCELL] = mx; stats[INS_NONEMPTY] = ne;
    }
}

// statistics of the global cell-size table: total, largest cell, non-empty cells (atomics into zeroed words)
__global__ __launch_bounds__(256) void k_gcount_stats(const int64_t* __restrict__ gcount, int64_t ncells, int64_t* __restrict__ stats) {
    __shared__ long long s_sum[4], s_max[4], s_ne[4];
    long long sum = 0, mx = 0, ne = 0;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncells; c += (int64_t)gridDim.x * blockDim.x) {
        const long long g = gcount[c];
        sum += g;
        mx = g > mx ? g : mx;
        ne += g > 0;
    }
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        const long long m2 = __shfl_xor(mx, o);
        mx = m2 > mx ? m2 : mx;
        ne += __shfl_xor(ne, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[w] = sum; s_max[w] = mx; s_ne[w] = ne; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { sum += s_sum[i]; mx = s_max[i] > mx ? s_max[i] : mx; ne += s_ne[i]; }
        atomicAdd((unsigned long long*)&stats[INS_NTOTAL], (unsigned long long)sum);
        atomicMax((unsigned long long*)&stats[INS_MAXCELL], (unsigned long long)mx);
        atomicAdd((unsigned long long*)&stats[INS_NONEMPTY], (unsigned long long)ne);
    }
}

// delta[c] = gcount[c] - delta[c] (delta held the sizes before the insert)
__global__ void k_cell_delta(const int64_t* __restrict__ gcount, int64_t* __restrict__ delta, int64_t ncells) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncells) delta[c] = gcount[c] - delta[c];
}

__global__ void k_add_remote(int64_t* __restrict__ gcount, const int64_t* __restrict__ delta, int64_t ncells,
                             const int32_t* __restrict__ owner, int rank, int world, int64_t* __restrict__ stats) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const int64_t d = delta[c];
    if (d < 0) { stats[INS_ERR] = 1; return; }
    if (d == 0 || dev_owns(c, owner, rank, world)) return;
    gcount[c] += d;
}

__global__ void k_get_codes(const int32_t* __restrict__ cells, const uint32_t* __restrict__ pos, int64_t n, int64_t ncells,
                            const int64_t* __restrict__ loff, const uint8_t* __restrict__ codes, int M, uint8_t* __restrict__ out,
                            int64_t* __restrict__ stats) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t c = cells[i];
    if (c < 0 || c >= ncells) { stats[INS_ERR] = 1 + i; return; }
    const int64_t a = loff[c], b = loff[ncells + 1 + c];  // (the used end: lend = loff + ncells + 1)
    if ((int64_t)pos[i] >= b - a) { stats[INS_ERR] = 1 + i; return; }
    for (int j = 0; j < M; ++j) out[i * M + j] = codes[(a + pos[i]) * M + j];
}

// records of the routed insert (columbiaimagesearch_amd/distributed.py): id (8 B) | coarse (2 x uint16) | fine (M B)
__global__ void k_route_keys(const uint16_t* __restrict__ coarse, int64_t n, int V, const int32_t* __restrict__ owner, int world,
                             uint32_t* __restrict__ key, uint32_t* __restrict__ val, int64_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c0 = coarse[2 * i], c1 = coarse[2 * i + 1];
    uint32_t dst = 0;
    if (c0 < V && c1 < V) {  // codes that are out of range travel to rank 0, whose insert reports them
        const int64_t cell = (int64_t)c0 * V + c1;
        dst = (uint32_t)(owner ? owner[cell] : (int)(cell % world));
    }
    key[i] = dst;
    val[i] = (uint32_t)i;
    atomicAdd((unsigned long long*)&counts[dst], 1ull);
}

__global__ void k_route_pack(const uint32_t* __restrict__ perm, const int64_t* __restrict__ ids, const uint16_t* __restrict__ coarse,
                             const uint8_t* __restrict__ fine, int64_t n, int M, uint8_t* __restrict__ rec) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t i = perm[j];
    uint8_t* r = rec + j * (12 + M);
    const int64_t id = ids[i];
    for (int b = 0; b < 8; ++b) r[b] = (uint8_t)((uint64_t)id >> (8 * b));
    const uint16_t c0 = coarse[2 * i], c1 = coarse[2 * i + 1];
    r[8] = (uint8_t)c0; r[9] = (uint8_t)(c0 >> 8); r[10] = (uint8_t)c1; r[11] = (uint8_t)(c1 >> 8);
    for (int b = 0; b < M; ++b) r[12 + b] = fine[i * M + b];
}

__global__ void k_route_unpack(const uint8_t* __restrict__ rec, int64_t n, int M, int64_t* __restrict__ ids, uint16_t* __restrict__ coarse,
                               uint8_t* __restrict__ fine) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint8_t* r = rec + j * (12 + M);
    uint64_t id = 0;
    for (int b = 0; b < 8; ++b) id |= (uint64_t)r[b] << (8 * b);
    ids[j] = (int64_t)id;
    coarse[2 * j] = (uint16_t)(r[8] | (r[9] << 8));
    coarse[2 * j + 1] = (uint16_t)(r[10] | (r[11] << 8));
    for (int b = 0; b < M; ++b) fine[j * M + b] = r[12 + b];
}

// ================================================================================================
// host: index object
// ================================================================================================
static inline unsigned grid_for(int64_t n, int t) { return (unsigned)(n > 0 ? ceil_div(n, t) : 1); }

static int store_init(cis_index* ix, CellStore& s, bool with_codes) {
    if (s.init) return CIS_OK;
    const int64_t nc = ix->ncells;
    s.with_codes = with_codes;
    for (int g = 0; g < 2; ++g) CIS_TRY(s.loff[g].reserve((size_t)(2 * nc + 1) * sizeof(int64_t)));  // starts [nc + 1], used ends [nc]
    CIS_TRY(s.cmax.reserve((size_t)nc * sizeof(int64_t)));
    CIS_TRY(s.ids[0].reserve(256));
    if (with_codes) CIS_TRY(s.codes[0].reserve(256));
    CIS_CHECK_HIP(hipMemset(s.loff[0].p, 0, (size_t)(2 * nc + 1) * sizeof(int64_t)));
    CIS_CHECK_HIP(hipMemset(s.cmax.p, 0, (size_t)nc * sizeof(int64_t)));
    s.cur = 0;
    s.n = 0;
    s.init = true;
    return CIS_OK;
}

int cis_index_ready(cis_index* ix) {
    CIS_REQUIRE(ix->base == nullptr, "this handle is a search view (cis_index_create_view): inserts and reads go to the base index");
    if (ix->own.init && ix->h_ins) return CIS_OK;
    CIS_TRY(cis_lazy_init());
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    CIS_TRY(store_init(ix, ix->own, true));
    if (!ix->d_gcount.p) {
        CIS_TRY(ix->d_gcount.reserve((size_t)ix->ncells * sizeof(int64_t)));
        CIS_CHECK_HIP(hipMemset(ix->d_gcount.p, 0, (size_t)ix->ncells * sizeof(int64_t)));
    }
    if (!ix->d_plan_hint.p) {  // k_plan_par's band-size hint (cells visited per unit of quota, the two launch parities)
        CIS_TRY(ix->d_plan_hint.reserve(4 * sizeof(unsigned long long)));
        CIS_CHECK_HIP(hipMemset(ix->d_plan_hint.p, 0, 4 * sizeof(unsigned long long)));
    }
    if (!ix->owner.empty() && !ix->d_owner.p) {
        CIS_TRY(ix->d_owner.reserve((size_t)ix->ncells * sizeof(int32_t)));
        CIS_CHECK_HIP(hipMemcpy(ix->d_owner.p, ix->owner.data(), (size_t)ix->ncells * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (!ix->h_ins) {
        CIS_TRY(ix->d_stats.reserve(INS_WORDS * sizeof(int64_t)));
        CIS_CHECK_HIP(hipMemset(ix->d_stats.p, 0, INS_WORDS * sizeof(int64_t)));
        CIS_CHECK_HIP(hipHostMalloc((void**)&ix->h_ins, INS_WORDS * sizeof(int64_t), hipHostMallocDefault));
        for (int i = 0; i < INS_WORDS; ++i) ix->h_ins[i] = 0;
    }
    return CIS_OK;
}

extern "C" int cis_index_create(cis_index** out, cis_model* m) {
    CIS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CIS_REQUIRE(m != nullptr, "model is NULL");
    cis_index* ix = new cis_index();
    ix->m = m;
    ix->V = m->V;
    ix->M = m->M;
    ix->ncells = (int64_t)m->V * m->V;
    *out = ix;
    return CIS_OK;
}

extern "C" int cis_index_create_view(cis_index** out, cis_index* base) {
    CIS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CIS_REQUIRE(base != nullptr, "base index is NULL");
    CIS_REQUIRE(base->base == nullptr, "a view of a view: create views from the base index");
    cis_index* ix = new cis_index();
    ix->m = base->m;
    ix->V = base->V;
    ix->M = base->M;
    ix->base = base;
    ix->force_exact_scan = base->force_exact_scan; ix->force_scan2 = base->force_scan2; ix->force_scan3 = base->force_scan3;
    ix->force_two_pass = base->force_two_pass; ix->force_prefilter_scan = base->force_prefilter_scan; ix->force_stream = base->force_stream; ix->force_scan5 = base->force_scan5;
    ix->sync_from_base();
    base->views.push_back(ix);
    *out = ix;
    return CIS_OK;
}

extern "C" void cis_index_destroy(cis_index* ix) {
    if (!ix) return;
    if (ix->m) (void)hipSetDevice(ix->m->device);
    if (ix->base) {  // a view leaves its base's list
        std::vector<cis_index*>& v = ix->base->views;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == ix) { v.erase(v.begin() + i); break; }
    }
    // a base destroyed before its views: the views keep their own workspaces but lose the storage -- they are marked and every
    // later search through them is refused (no dangling pointer is ever followed)
    for (cis_index* v : ix->views) { v->base = nullptr; v->orphaned = true; }
    ix->views.clear();
    cis_host_forget(ix);
    if (ix->h_stream) {
        if (ix->h_pending && ix->h_ev_done) (void)hipEventSynchronize(ix->h_ev_done);
        (void)hipStreamSynchronize(ix->h_stream);
        (void)hipStreamDestroy(ix->h_stream);
        ix->h_stream = nullptr;
        for (hipEvent_t e : {ix->h_ev_in, ix->h_ev_out, ix->h_ev_done})
            if (e) (void)hipEventDestroy(e);
    }
    DevBuf* bufs[] = {&ix->d_gcount, &ix->d_plan_hint, &ix->d_owner, &ix->w_xp, &ix->w_cd, &ix->w_order,
                      &ix->w_sorted, &ix->w_plan, &ix->w_off, &ix->w_items, &ix->w_tabs, &ix->w_T, &ix->w_hits,
                      &ix->w_hitn, &ix->w_slack, &ix->w_planfb, &ix->w_vis, &ix->w_tiles, &ix->w_part, &ix->w_q, &ix->w_oids, &ix->w_odists, &ix->w_onf, &ix->w_ovis,
                      &ix->w_ocell, &ix->w_opos, &ix->w_order2, &ix->w_px, &ix->w_T32, &ix->w_grp, &ix->w_tord, &ix->w_y64, &ix->w_x64,
                      &ix->wi_key[0], &ix->wi_key[1], &ix->wi_val[0], &ix->wi_val[1], &ix->wi_hist, &ix->wi_sid, &ix->wi_acc,
                      &ix->w_s5, &ix->w_bmin, &ix->wi_apre, &ix->wi_tmp, &ix->wi_in_ids, &ix->wi_in_coarse, &ix->wi_in_fine, &ix->wi_scan, &ix->wi_cnt, &ix->wi_cnt2, &ix->d_stats};
    for (DevBuf* b : bufs) b->release();
    ix->own.release();
    ix->ghost.release();
    if (ix->h_totals) (void)hipHostFree(ix->h_totals);
    if (ix->h_ins) (void)hipHostFree(ix->h_ins);
    delete ix;
}

extern "C" int cis_index_set_shard(cis_index* ix, int rank, int world, const int32_t* owner) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad shard %d of %d", rank, world);
    CIS_REQUIRE(ix->nb_indexed == 0, "set_shard must be called on an empty index");
    ix->rank = rank;
    ix->world = world;
    ix->owner.clear();
    ix->d_owner.release();
    if (owner) {
        ix->owner.assign(owner, owner + ix->ncells);
        for (int64_t c = 0; c < ix->ncells; ++c)
            CIS_REQUIRE(owner[c] >= 0 && owner[c] < world, "owner[%lld]=%d out of range", (long long)c, owner[c]);
    }
    return CIS_OK;
}

extern "C" int64_t cis_index_size(cis_index* ix) { return ix ? ix->nb_indexed : 0; }

static int key_bits(int64_t ncells) {
    int b = 1;
    while (((int64_t)1 << b) < ncells) ++b;
    return b;
}

// stable sort of (key, index) for the first n entries of wi_key[0] / wi_val[0]; returns which generation holds the result
static int radix_sort_pairs(cis_index* ix, int64_t n, int bits, hipStream_t st, int* gen_out) {
    const int nblk = (int)ceil_div(n, RS_TILE);
    const int64_t hn = (int64_t)256 * nblk;
    CIS_TRY(ix->wi_hist.reserve((size_t)hn * sizeof(uint32_t)));
    CIS_TRY(ix->wi_scan.reserve((size_t)(ceil_div(hn, SCAN_TILE) + 2) * sizeof(int64_t) + 64));
    int gen = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        uint32_t* k0 = ix->wi_key[gen].as<uint32_t>();
        uint32_t* v0 = ix->wi_val[gen].as<uint32_t>();
        uint32_t* k1 = ix->wi_key[1 - gen].as<uint32_t>();
        uint32_t* v1 = ix->wi_val[1 - gen].as<uint32_t>();
        uint32_t* H = ix->wi_hist.as<uint32_t>();
        hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(64), 0, st, (const uint32_t*)k0, n, shift, H, nblk);
        dev_exclusive_scan<uint32_t, uint32_t>(H, H, hn, ix->wi_scan.as<uint32_t>(), nullptr, nullptr, st);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(64), 0, st, (const uint32_t*)k0, (const uint32_t*)v0, n, shift,
                           (const uint32_t*)H, nblk, k1, v1);
        gen = 1 - gen;
    }
    *gen_out = gen;
    return CIS_OK;
}

// slack items behind a rebuilt cell, next to an eighth of its size (few cells: room for several batches; thousands of coarse
// clusters: millions of tiny cells, two items each are already 32 B x 16 M); CIS_INSERT_SLACK=-1: tight packing (rounds 1-3)
static int slack_const(const cis_index* ix) {
    static const int env = getenv("CIS_INSERT_SLACK") ? atoi(getenv("CIS_INSERT_SLACK")) : -2;
    if (env != -2) return env;
    return ix->ncells <= 65536 ? 32 : 2;
}
static const int64_t INPLACE_MAX = 65536;  // larger batches go straight to the rebuild (they would not fit the slack anyway)

// one stable merge of n device-resident items into store `s` (sel 0: own, 1: ghost); accepted count -> stats word `acc_word`.
// Synchronises `st` (the statistics words are read back) and adds the accepted count to s.n.
static int store_merge(cis_index* ix, CellStore& s, int sel, const int64_t* d_ids, const uint16_t* d_coarse, const uint8_t* d_fine,
                       int64_t n, int dedup, int acc_word, hipStream_t st) {
    const int M = ix->M, V = ix->V, K = ix->m->K;
    const int64_t nc = ix->ncells;
    CIS_REQUIRE(n < ((int64_t)1 << 31), "at most 2^31 - 1 items per insert call");
    ix->stats_fresh = false;  // this merge changes the global cell sizes
    for (int g = 0; g < 2; ++g) {
        CIS_TRY(ix->wi_key[g].reserve((size_t)n * sizeof(uint32_t)));
        CIS_TRY(ix->wi_val[g].reserve((size_t)n * sizeof(uint32_t)));
    }
    CIS_TRY(ix->wi_tmp.reserve((size_t)n * sizeof(int64_t)));   // ids (or -1) in arrival order
    CIS_TRY(ix->wi_sid.reserve((size_t)n * sizeof(int64_t)));   // ... in sorted order
    CIS_TRY(ix->wi_acc.reserve((size_t)n * sizeof(uint32_t)));
    CIS_TRY(ix->wi_apre.reserve((size_t)(n + 2) * sizeof(uint32_t)));
    CIS_TRY(ix->wi_scan.reserve((size_t)(ceil_div(std::max<int64_t>(std::max<int64_t>(n, nc + 1), (int64_t)256 * ceil_div(n, RS_TILE)), SCAN_TILE) + 2) * sizeof(int64_t) + 64));
    const int32_t* d_owner = ix->d_owner.as<int32_t>();
    int64_t* stats = ix->d_stats.as<int64_t>();
    int64_t* gcount = ix->d_gcount.as<int64_t>();
    static const bool no_small = getenv("CIS_INS_NO_SMALL") != nullptr;  // A/B and tests: the general path for every batch size
    const bool small = n > 0 && n <= INS_SMALL && !no_small;
    int gen = 0;
    int64_t* sid = ix->wi_sid.as<int64_t>();
    if (small) {
        gen = 1;
        hipLaunchKernelGGL(k_ins_small_sort, dim3(1), dim3(1024), 0, st, d_ids, d_coarse, d_fine, (int)n, V, M, K, d_owner, ix->rank, ix->world, sel,
                           dedup, ix->wi_key[1].as<uint32_t>(), ix->wi_val[1].as<uint32_t>(), ix->wi_tmp.as<int64_t>(), sid, gcount, stats);
    } else {
        hipLaunchKernelGGL(k_ins_keys, dim3(grid_for(n, 256)), dim3(256), 0, st, d_ids, d_coarse, d_fine, n, V, M, K, d_owner, ix->rank,
                           ix->world, sel, dedup, ix->wi_key[0].as<uint32_t>(), ix->wi_val[0].as<uint32_t>(), ix->wi_tmp.as<int64_t>(),
                           gcount, stats);
        CIS_TRY(radix_sort_pairs(ix, n, key_bits(nc), st, &gen));
    }
    const uint32_t* skey = ix->wi_key[gen].as<uint32_t>();
    const uint32_t* perm = ix->wi_val[gen].as<uint32_t>();
    uint32_t* acc = ix->wi_acc.as<uint32_t>();
    uint32_t* apre = ix->wi_apre.as<uint32_t>();
    uint32_t* total = apre + n;  // one word behind the prefix array
    int64_t* loff = s.loff[s.cur].as<int64_t>();
    int64_t* lend = loff + nc + 1;
    unsigned long long* cmaxp = s.cmax.as<unsigned long long>();
    if (!small) hipLaunchKernelGGL(k_ins_gather, dim3(grid_for(n, 256)), dim3(256), 0, st, (const int64_t*)ix->wi_tmp.as<int64_t>(), perm, n, sid);
    hipLaunchKernelGGL(k_ins_dedup, dim3(grid_for(n, 4)), dim3(256), 0, st, skey, (const int64_t*)sid, n, dedup, (const int64_t*)loff,
                       (const int64_t*)lend, (const int64_t*)s.ids[s.cur].as<int64_t>(), (const unsigned long long*)cmaxp, acc);
    const int slack = slack_const(ix);
    int64_t* gsink = gcount;  // (the id-only store of the other shards' cells counts into the global cell sizes as well)
    const bool small_place = small && slack >= 0 && s.n > 0;
    if (!small_place) dev_exclusive_scan<uint32_t, uint32_t>(acc, apre, n, ix->wi_scan.as<uint32_t>(), total, stats + acc_word, st);
    // ---- in place, up to 1024 items: prefix sums, scatter, commit and statistics in one workgroup (three launches per insert) ----
    if (small_place) {
        uint8_t* codes_cur = s.with_codes ? s.codes[s.cur].as<uint8_t>() : nullptr;
        const int do_stats = nc <= 65536 ? 1 : 0;
        hipLaunchKernelGGL(k_ins_small_place, dim3(1), dim3(1024), 0, st, skey, perm, (const int64_t*)sid, (const uint32_t*)acc, apre, (int)n, acc_word,
                           (const int64_t*)loff, lend, d_fine, M, s.ids[s.cur].as<int64_t>(), codes_cur, cmaxp, gsink, nc, do_stats, stats);
        if (!do_stats) {
            CIS_CHECK_HIP(hipMemsetAsync(stats + INS_NTOTAL, 0, 3 * sizeof(int64_t), st));
            hipLaunchKernelGGL(k_gcount_stats, dim3((unsigned)std::min<int64_t>(ceil_div(nc, 256), 1024)), dim3(256), 0, st, (const int64_t*)gcount, nc, stats);
        }
        CIS_CHECK_HIP(hipGetLastError());
        CIS_CHECK_HIP(hipMemcpyAsync(ix->h_ins, ix->d_stats.p, INS_WORDS * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        CIS_CHECK_HIP(hipStreamSynchronize(st));
        if (ix->h_ins[INS_OVERFLOW] == 0) {
            s.n += ix->h_ins[acc_word];
            if (sel == 0) ix->n_inplace += 1;
            ix->n_total = ix->h_ins[INS_NTOTAL];
            ix->max_cell = ix->h_ins[INS_MAXCELL];
            ix->nonempty_cells = ix->h_ins[INS_NONEMPTY];
            ix->nb_indexed = ix->n_total;
            ix->stats_fresh = true;
            return CIS_OK;
        }
        CIS_CHECK_HIP(hipMemsetAsync(stats + INS_OVERFLOW, 0, sizeof(int64_t), st));
    }
    // ---- in place: the accepted items behind their cells' last items, if every one of them fits its cell's slack ----
    if (!small_place && slack >= 0 && s.n > 0 && n <= INPLACE_MAX) {
        uint8_t* codes_cur = s.with_codes ? s.codes[s.cur].as<uint8_t>() : nullptr;
        hipLaunchKernelGGL((k_ins_scatter<true>), dim3(grid_for(n, 256)), dim3(256), 0, st, skey, perm, (const int64_t*)sid, (const uint32_t*)acc,
                           (const uint32_t*)apre, n, (const int64_t*)loff, (const int64_t*)lend, (const int64_t*)nullptr, (const uint32_t*)nullptr,
                           d_fine, M, s.ids[s.cur].as<int64_t>(), codes_cur, cmaxp, stats);
        hipLaunchKernelGGL(k_ins_commit, dim3(grid_for(n, 256)), dim3(256), 0, st, skey, (const uint32_t*)acc, (const uint32_t*)apre, n, lend,
                           gsink, (const int64_t*)stats);
        // the statistics of the global cell sizes in the same read-back (valid when nothing overflowed): one host round trip per batch
        CIS_CHECK_HIP(hipMemsetAsync(stats + INS_NTOTAL, 0, 3 * sizeof(int64_t), st));
        hipLaunchKernelGGL(k_gcount_stats, dim3((unsigned)std::min<int64_t>(ceil_div(nc, 256), 1024)), dim3(256), 0, st, (const int64_t*)gcount, nc, stats);
        CIS_CHECK_HIP(hipGetLastError());
        CIS_CHECK_HIP(hipMemcpyAsync(ix->h_ins, ix->d_stats.p, INS_WORDS * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        CIS_CHECK_HIP(hipStreamSynchronize(st));
        if (ix->h_ins[INS_OVERFLOW] == 0) {
            s.n += ix->h_ins[acc_word];
            if (sel == 0) ix->n_inplace += 1;
            ix->n_total = ix->h_ins[INS_NTOTAL];
            ix->max_cell = ix->h_ins[INS_MAXCELL];
            ix->nonempty_cells = ix->h_ins[INS_NONEMPTY];
            ix->nb_indexed = ix->n_total;
            ix->stats_fresh = true;
            return CIS_OK;
        }
        // some cell is full: nothing became visible; the rebuild below takes the whole batch and renews every cell's slack
        CIS_CHECK_HIP(hipMemsetAsync(stats + INS_OVERFLOW, 0, sizeof(int64_t), st));
    }
    // ---- rebuild: every cell moves to its place in the other generation, the accepted items behind it ----
    const int nxt = 1 - s.cur;
    const int64_t items_bound = s.n + n;
    const int64_t cap_new_bound = slack < 0 ? items_bound : items_bound + items_bound / 8 + (int64_t)slack * nc + 64;
    CIS_TRY(s.ids[nxt].reserve((size_t)(cap_new_bound > 0 ? cap_new_bound : 1) * sizeof(int64_t)));
    if (s.with_codes) CIS_TRY(s.codes[nxt].reserve((size_t)(cap_new_bound > 0 ? cap_new_bound : 1) * M + 64));
    CIS_TRY(ix->wi_cnt.reserve((size_t)(2 * nc + 2) * sizeof(int64_t) + (size_t)(nc + 1) * sizeof(uint32_t)));
    int64_t* cnt_new = ix->wi_cnt.as<int64_t>();
    int64_t* cap_new = cnt_new + nc + 1;
    uint32_t* a0v = reinterpret_cast<uint32_t*>(cap_new + nc + 1);
    int64_t* noff = s.loff[nxt].as<int64_t>();
    int64_t* nend = noff + nc + 1;
    hipLaunchKernelGGL(k_ins_counts, dim3(grid_for(nc, 256)), dim3(256), 0, st, skey, (const uint32_t*)apre, (const uint32_t*)total, n, nc,
                       (const int64_t*)loff, (const int64_t*)lend, cnt_new, cap_new, a0v, gsink, slack);
    dev_exclusive_scan<int64_t, int64_t>(cap_new, noff, nc, ix->wi_scan.as<int64_t>(), noff + nc, nullptr, st);
    hipLaunchKernelGGL(k_ins_ends, dim3(grid_for(nc, 256)), dim3(256), 0, st, (const int64_t*)noff, (const int64_t*)cnt_new, nc, nend);
    const uint8_t* ocodes = s.with_codes ? s.codes[s.cur].as<uint8_t>() : nullptr;
    uint8_t* ncodes = s.with_codes ? s.codes[nxt].as<uint8_t>() : nullptr;
    if (s.n > 0) {
        if (s.n / nc < 16 && nc >= 65536) {
            hipLaunchKernelGGL(k_ins_move_cells, dim3(grid_for(nc, 256)), dim3(256), 0, st, (const int64_t*)loff, (const int64_t*)lend, (const int64_t*)noff, nc,
                               (const int64_t*)s.ids[s.cur].as<int64_t>(), ocodes, M, s.ids[nxt].as<int64_t>(), ncodes);
        } else {
#define CIS_MOVE(MW)                                                                                                             \
    hipLaunchKernelGGL((k_ins_move<MW>), dim3(grid_for(s.cap, 256)), dim3(256), 0, st, (const int64_t*)loff, (const int64_t*)lend, (const int64_t*)noff, nc, s.cap, \
                       (const int64_t*)s.ids[s.cur].as<int64_t>(), ocodes, M, s.ids[nxt].as<int64_t>(), ncodes)
            if (M == 4) CIS_MOVE(1);
            else if (M == 8) CIS_MOVE(2);
            else if (M == 16) CIS_MOVE(4);
            else if (M == 32) CIS_MOVE(8);
            else CIS_MOVE(0);
#undef CIS_MOVE
        }
    }
    hipLaunchKernelGGL((k_ins_scatter<false>), dim3(grid_for(n, 256)), dim3(256), 0, st, skey, perm, (const int64_t*)sid, (const uint32_t*)acc,
                       (const uint32_t*)apre, n, (const int64_t*)loff, (const int64_t*)lend, (const int64_t*)noff, (const uint32_t*)a0v, d_fine, M,
                       s.ids[nxt].as<int64_t>(), ncodes, cmaxp, stats);
    CIS_CHECK_HIP(hipGetLastError());
    CIS_CHECK_HIP(hipMemcpyAsync(ix->h_ins, ix->d_stats.p, INS_WORDS * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    CIS_CHECK_HIP(hipStreamSynchronize(st));
    s.n += ix->h_ins[acc_word];
    if (sel == 0) ix->n_rebuild += 1;
    s.cap = cap_new_bound;  // (an upper bound of the new layout's extent: the move kernel's grid; the true extent is loff[ncells])
    s.cur = nxt;
    return CIS_OK;
}

static int stats_zero(cis_index* ix, hipStream_t st) {
    CIS_CHECK_HIP(hipMemsetAsync(ix->d_stats.p, 0, INS_WORDS * sizeof(int64_t), st));
    return CIS_OK;
}

// copies the statistics words to the host and waits for the stream
static int stats_fetch(cis_index* ix, hipStream_t st) {
    CIS_CHECK_HIP(hipMemcpyAsync(ix->h_ins, ix->d_stats.p, INS_WORDS * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    CIS_CHECK_HIP(hipStreamSynchronize(st));
    return CIS_OK;
}

static int refresh_stats(cis_index* ix, hipStream_t st) {
    int64_t* stats = ix->d_stats.as<int64_t>();
    CIS_CHECK_HIP(hipMemsetAsync(stats + INS_NTOTAL, 0, 3 * sizeof(int64_t), st));
    const int64_t nc = ix->ncells;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(nc, 256), 1024);
    hipLaunchKernelGGL(k_gcount_stats, dim3(grid), dim3(256), 0, st, (const int64_t*)ix->d_gcount.as<int64_t>(), nc, stats);
    CIS_TRY(stats_fetch(ix, st));
    ix->n_total = ix->h_ins[INS_NTOTAL];
    ix->max_cell = ix->h_ins[INS_MAXCELL];
    ix->nonempty_cells = ix->h_ins[INS_NONEMPTY];
    ix->nb_indexed = ix->n_total;
    return CIS_OK;
}

static const int64_t DEDUP_CHUNK = 262144;       // bounds the quadratic part of the in-batch duplicate test
static const int64_t PLAIN_CHUNK = (int64_t)1 << 28;

// device arrays in, merged on `st`; synchronises `st` before returning (the accepted counts are read back)
static int index_add_dev(cis_index* ix, const int64_t* d_ids, const uint16_t* d_coarse, const uint8_t* d_fine, int64_t n, int dedup,
                         int64_t* n_added, int64_t* n_invalid, int64_t* d_cell_delta, hipStream_t st) {
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    if (n_added) *n_added = 0;
    if (n_invalid) *n_invalid = 0;
    if (d_cell_delta) {
        if (n == 0) CIS_CHECK_HIP(hipMemsetAsync(d_cell_delta, 0, (size_t)ix->ncells * sizeof(int64_t), st));
        else CIS_CHECK_HIP(hipMemcpyAsync(d_cell_delta, ix->d_gcount.p, (size_t)ix->ncells * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    }
    if (n == 0) return CIS_OK;
    const bool sharded = ix->world > 1;
    if (dedup && ix->had_plain_remote) {
        cis_set_error("dedup add after plain (dedup=0) adds that counted other shards' cells is not supported on a sharded index");
        return CIS_EUNSUPPORTED;
    }
    const int64_t chunk = dedup ? DEDUP_CHUNK : PLAIN_CHUNK;
    int64_t added = 0, invalid = 0;
    for (int64_t a = 0; a < n; a += chunk) {
        const int64_t bn = std::min(chunk, n - a);
        ix->stats_fresh = false;
        CIS_TRY(stats_zero(ix, st));
        CIS_TRY(store_merge(ix, ix->own, 0, d_ids + a, d_coarse + 2 * a, d_fine + a * ix->M, bn, dedup, INS_ACC_OWN, st));
        if (sharded && dedup) {
            // items of the other shards' cells: recognised as duplicates through the id-only store.  Only when such items
            // exist (the routed insert hands an owner its own cells only and never pays for this).
            const int64_t seen = ix->h_ins[INS_ACC_OWN] + ix->h_ins[INS_INVALID];  // (store_merge left the statistics words in h_ins)
            added += ix->h_ins[INS_ACC_OWN];
            invalid += ix->h_ins[INS_INVALID];
            bool foreign = seen < bn;  // some item was neither accepted nor invalid: a duplicate, or a foreign cell
            if (foreign) {
                CIS_TRY(store_init(ix, ix->ghost, false));
                CIS_TRY(store_merge(ix, ix->ghost, 1, d_ids + a, d_coarse + 2 * a, d_fine + a * ix->M, bn, 1, INS_ACC_GHOST, st));
                added += ix->h_ins[INS_ACC_GHOST];
            }
        } else {
            added += ix->h_ins[INS_ACC_OWN] + ix->h_ins[INS_REMOTE_PLAIN];
            invalid += ix->h_ins[INS_INVALID];
            if (ix->h_ins[INS_REMOTE_PLAIN] > 0) ix->had_plain_remote = true;
        }
    }
    ix->n_local = ix->own.n;
    if (d_cell_delta)
        hipLaunchKernelGGL(k_cell_delta, dim3(grid_for(ix->ncells, 256)), dim3(256), 0, st, (const int64_t*)ix->d_gcount.as<int64_t>(),
                           d_cell_delta, ix->ncells);
    if (!ix->stats_fresh) CIS_TRY(refresh_stats(ix, st));  // (an in-place batch read them back with its own statistics)
    if (n_added) *n_added = added;
    if (n_invalid) *n_invalid = invalid;
    return CIS_OK;
}

extern "C" int cis_index_add_dev(cis_index* ix, const int64_t* d_ids, const uint16_t* d_coarse, const uint8_t* d_fine, int64_t n,
                                 int dedup, int64_t* n_added, int64_t* n_invalid, int64_t* d_cell_delta, void* stream) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(n >= 0 && (n == 0 || (d_ids && d_coarse && d_fine)), "NULL buffer");
    return index_add_dev(ix, d_ids, d_coarse, d_fine, n, dedup, n_added, n_invalid, d_cell_delta, (hipStream_t)stream);
}

extern "C" int cis_index_add(cis_index* ix, const int64_t* ids, const uint16_t* coarse, const uint8_t* fine,
                             int64_t n, int dedup, int64_t* n_added) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(n >= 0 && (n == 0 || (ids && coarse && fine)), "NULL buffer");
    const int V = ix->V, M = ix->M, K = ix->m->K;
    // host arrays are checked before anything changes (the device entry point skips and counts bad items instead)
    for (int64_t i = 0; i < n; ++i)
        CIS_REQUIRE(coarse[2 * i] < V && coarse[2 * i + 1] < V, "item %lld: coarse code out of range (V=%d)",
                    (long long)i, V);
    if (K < 256)
        for (int64_t i = 0; i < n * M; ++i)
            CIS_REQUIRE(fine[i] < K, "fine code %d out of range (K=%d)", (int)fine[i], K);
    for (int64_t i = 0; i < n; ++i) CIS_REQUIRE(ids[i] >= 0, "item %lld: ids must be >= 0", (long long)i);
    if (n_added) *n_added = 0;
    if (n == 0) return CIS_OK;
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    CIS_TRY(ix->wi_in_ids.reserve((size_t)n * sizeof(int64_t)));
    CIS_TRY(ix->wi_in_coarse.reserve((size_t)n * 2 * sizeof(uint16_t)));
    CIS_TRY(ix->wi_in_fine.reserve((size_t)n * M));
    CIS_CHECK_HIP(hipMemcpy(ix->wi_in_ids.p, ids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(ix->wi_in_coarse.p, coarse, (size_t)n * 2 * sizeof(uint16_t), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(ix->wi_in_fine.p, fine, (size_t)n * M, hipMemcpyHostToDevice));
    int64_t invalid = 0;
    return index_add_dev(ix, ix->wi_in_ids.as<int64_t>(), ix->wi_in_coarse.as<uint16_t>(), ix->wi_in_fine.as<uint8_t>(), n, dedup,
                         n_added, &invalid, nullptr, nullptr);
}

// Cell-sharded insert with routed codes (columbiaimagesearch_amd/distributed.py:add_codes_routed): a rank is handed only
// the codes of the cells it owns; the sizes of the other cells -- which drive the quota cut of every query on every
// rank (search.py:128-133) -- arrive as per-cell increments summed over the owners.
extern "C" int cis_index_insert_counters(cis_index* ix, int64_t counters[2]) {
    CIS_REQUIRE(ix != nullptr && counters != nullptr, "NULL argument");
    counters[0] = ix->n_inplace;
    counters[1] = ix->n_rebuild;
    return CIS_OK;
}

extern "C" int cis_index_cell_counts(cis_index* ix, int64_t* counts) {
    CIS_REQUIRE(ix != nullptr && counts != nullptr, "NULL argument");
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    CIS_CHECK_HIP(hipMemcpy(counts, ix->d_gcount.p, (size_t)ix->ncells * sizeof(int64_t), hipMemcpyDeviceToHost));
    return CIS_OK;
}

extern "C" int cis_index_cell_counts_dev(cis_index* ix, int64_t* d_counts, void* stream) {
    CIS_REQUIRE(ix != nullptr && d_counts != nullptr, "NULL argument");
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    CIS_CHECK_HIP(hipMemcpyAsync(d_counts, ix->d_gcount.p, (size_t)ix->ncells * sizeof(int64_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CIS_OK;
}

static int add_remote_dev(cis_index* ix, const int64_t* d_delta, hipStream_t st) {
    CIS_TRY(stats_zero(ix, st));
    hipLaunchKernelGGL(k_add_remote, dim3(grid_for(ix->ncells, 256)), dim3(256), 0, st, ix->d_gcount.as<int64_t>(), d_delta, ix->ncells,
                       (const int32_t*)ix->d_owner.as<int32_t>(), ix->rank, ix->world, ix->d_stats.as<int64_t>());
    CIS_TRY(refresh_stats(ix, st));
    CIS_REQUIRE(ix->h_ins[INS_ERR] == 0, "negative per-cell count");
    return CIS_OK;
}

extern "C" int cis_index_add_remote_counts_dev(cis_index* ix, const int64_t* d_delta, void* stream) {
    CIS_REQUIRE(ix != nullptr && d_delta != nullptr, "NULL argument");
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    return add_remote_dev(ix, d_delta, (hipStream_t)stream);
}

extern "C" int cis_index_add_remote_counts(cis_index* ix, const int64_t* delta) {
    CIS_REQUIRE(ix != nullptr && delta != nullptr, "NULL argument");
    for (int64_t c = 0; c < ix->ncells; ++c) CIS_REQUIRE(delta[c] >= 0, "negative count for cell %lld", (long long)c);
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    CIS_TRY(ix->wi_tmp.reserve((size_t)ix->ncells * sizeof(int64_t)));
    CIS_CHECK_HIP(hipMemcpy(ix->wi_tmp.p, delta, (size_t)ix->ncells * sizeof(int64_t), hipMemcpyHostToDevice));
    return add_remote_dev(ix, ix->wi_tmp.as<int64_t>(), nullptr);
}

extern "C" int cis_index_get_cell(cis_index* ix, int c0, int c1, int64_t cap, int64_t* ids, uint8_t* fine, int64_t* n) {
    CIS_REQUIRE(ix != nullptr && n != nullptr, "NULL argument");
    CIS_REQUIRE(c0 >= 0 && c0 < ix->V && c1 >= 0 && c1 < ix->V, "cell (%d,%d) out of range", c0, c1);
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    const int64_t cell = (int64_t)c0 * ix->V + c1;
    int64_t g = 0, ab[2] = {0, 0};
    CIS_CHECK_HIP(hipMemcpy(&g, ix->d_gcount.as<int64_t>() + cell, sizeof(int64_t), hipMemcpyDeviceToHost));
    *n = g;
    if (cap <= 0) return CIS_OK;
    *n = 0;  // with a buffer: the number of items copied (a cell of another shard has none here)
    if (!ix->owns(cell)) return CIS_OK;
    CIS_CHECK_HIP(hipMemcpy(&ab[0], ix->loff_ptr() + cell, sizeof(int64_t), hipMemcpyDeviceToHost));
    CIS_CHECK_HIP(hipMemcpy(&ab[1], ix->loff_ptr() + ix->ncells + 1 + cell, sizeof(int64_t), hipMemcpyDeviceToHost));  // the used end (lend)
    const int64_t k = std::min(cap, ab[1] - ab[0]);
    *n = k;
    if (k > 0) {
        if (ids) CIS_CHECK_HIP(hipMemcpy(ids, ix->ids_ptr() + ab[0], (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (fine) CIS_CHECK_HIP(hipMemcpy(fine, ix->codes_ptr() + ab[0] * ix->M, (size_t)k * ix->M, hipMemcpyDeviceToHost));
    }
    return CIS_OK;
}

extern "C" int cis_index_get_codes(cis_index* ix, const int32_t* cells, const uint32_t* pos, int64_t n, uint8_t* fine) {
    CIS_REQUIRE(ix != nullptr && (n == 0 || (cells && pos && fine)), "NULL argument");
    if (n == 0) return CIS_OK;
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    const int M = ix->M;
    CIS_TRY(ix->wi_key[0].reserve((size_t)n * sizeof(int32_t)));
    CIS_TRY(ix->wi_val[0].reserve((size_t)n * sizeof(uint32_t)));
    CIS_TRY(ix->wi_in_fine.reserve((size_t)n * M));
    CIS_CHECK_HIP(hipMemcpy(ix->wi_key[0].p, cells, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(ix->wi_val[0].p, pos, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    CIS_TRY(stats_zero(ix, nullptr));
    hipLaunchKernelGGL(k_get_codes, dim3(grid_for(n, 256)), dim3(256), 0, nullptr, (const int32_t*)ix->wi_key[0].as<int32_t>(),
                       (const uint32_t*)ix->wi_val[0].as<uint32_t>(), n, ix->ncells, ix->loff_ptr(), ix->codes_ptr(), M,
                       ix->wi_in_fine.as<uint8_t>(), ix->d_stats.as<int64_t>());
    CIS_TRY(stats_fetch(ix, nullptr));
    CIS_CHECK_HIP(hipMemcpy(fine, ix->wi_in_fine.p, (size_t)n * M, hipMemcpyDeviceToHost));
    if (ix->h_ins[INS_ERR] != 0) {
        const int64_t i = ix->h_ins[INS_ERR] - 1;
        cis_set_error("item (%d, %u) is not stored on this shard", cells[i], pos[i]);
        return CIS_EINVAL;
    }
    return CIS_OK;
}

// ---- routed insert on device arrays ------------------------------------------------------
// Records of 12 + M bytes grouped by the rank that owns their cell, in arrival order inside a group (so that the
// per-cell insertion order after the all-to-all is that of a single index); d_counts [world] = records per rank.
extern "C" int cis_index_route_pack_dev(cis_index* ix, const int64_t* d_ids, const uint16_t* d_coarse, const uint8_t* d_fine,
                                        int64_t n, uint8_t* d_records, int64_t* d_counts, void* stream) {
    CIS_REQUIRE(ix != nullptr && d_counts != nullptr, "NULL argument");
    CIS_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && (n == 0 || (d_ids && d_coarse && d_fine && d_records)), "bad arguments");
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    hipStream_t st = (hipStream_t)stream;
    CIS_CHECK_HIP(hipMemsetAsync(d_counts, 0, (size_t)ix->world * sizeof(int64_t), st));
    if (n == 0) return CIS_OK;
    for (int g = 0; g < 2; ++g) {
        CIS_TRY(ix->wi_key[g].reserve((size_t)n * sizeof(uint32_t)));
        CIS_TRY(ix->wi_val[g].reserve((size_t)n * sizeof(uint32_t)));
    }
    hipLaunchKernelGGL(k_route_keys, dim3(grid_for(n, 256)), dim3(256), 0, st, d_coarse, n, ix->V, (const int32_t*)ix->d_owner.as<int32_t>(),
                       ix->world, ix->wi_key[0].as<uint32_t>(), ix->wi_val[0].as<uint32_t>(), d_counts);
    int gen = 0;
    CIS_TRY(radix_sort_pairs(ix, n, key_bits(ix->world), st, &gen));
    hipLaunchKernelGGL(k_route_pack, dim3(grid_for(n, 256)), dim3(256), 0, st, (const uint32_t*)ix->wi_val[gen].as<uint32_t>(), d_ids,
                       d_coarse, d_fine, n, ix->M, d_records);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

extern "C" int cis_index_add_records_dev(cis_index* ix, const uint8_t* d_records, int64_t n, int dedup, int64_t* n_added,
                                         int64_t* n_invalid, int64_t* d_cell_delta, void* stream) {
    CIS_REQUIRE(ix != nullptr, "index is NULL");
    CIS_REQUIRE(n >= 0 && (n == 0 || d_records), "NULL buffer");
    if (n_added) *n_added = 0;
    if (n_invalid) *n_invalid = 0;
    CIS_TRY(cis_index_ready(ix));
    CIS_CHECK_HIP(hipSetDevice(ix->m->device));
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return index_add_dev(ix, nullptr, nullptr, nullptr, 0, dedup, n_added, n_invalid, d_cell_delta, st);
    CIS_TRY(ix->wi_in_ids.reserve((size_t)n * sizeof(int64_t)));
    CIS_TRY(ix->wi_in_coarse.reserve((size_t)n * 2 * sizeof(uint16_t)));
    CIS_TRY(ix->wi_in_fine.reserve((size_t)n * ix->M));
    hipLaunchKernelGGL(k_route_unpack, dim3(grid_for(n, 256)), dim3(256), 0, st, d_records, n, ix->M, ix->wi_in_ids.as<int64_t>(),
                       ix->wi_in_coarse.as<uint16_t>(), ix->wi_in_fine.as<uint8_t>());
    return index_add_dev(ix, ix->wi_in_ids.as<int64_t>(), ix->wi_in_coarse.as<uint16_t>(), ix->wi_in_fine.as<uint8_t>(), n, dedup,
                         n_added, n_invalid, d_cell_delta, st);
}

// ---- featsio.normfeatB64encode's normalisation on device rows (cufacesearch/cufacesearch/featurizer/featsio.py:13-22) ----
// x[r] /= ||x[r]|| in the rows' dtype: the squared norm is accumulated in float64 and rounded to the dtype (numpy's own
// float32 dot product has an unspecified order; ours is within its rounding), the division is the dtype's.  Zero rows stay
// zero (the reference would emit NaNs for them).  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void k_l2_normalize(T* __restrict__ x, int64_t n, int d) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & 63;
    T* row = x + r * d;
    double s = 0.0;
    for (int i = lane; i < d; i += 64) {
        const double v = (double)row[i];
        s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const T nrm = (T)sqrt(s);
    if (nrm > (T)0) {
        for (int i = lane; i < d; i += 64) row[i] = row[i] / nrm;
    }
}

extern "C" int cis_l2_normalize_dev(void* d_x, int dtype, int64_t n, int d, void* stream) {
    CIS_REQUIRE(dtype == CIS_F32 || dtype == CIS_F64, "dtype must be 4 or 8");
    CIS_REQUIRE(n >= 0 && d >= 1 && (n == 0 || d_x), "bad arguments");
    if (n == 0) return CIS_OK;
    CIS_TRY(cis_lazy_init());
    if (dtype == CIS_F32) hipLaunchKernelGGL((k_l2_normalize<float>), dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, (float*)d_x, n, d);
    else hipLaunchKernelGGL((k_l2_normalize<double>), dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, (double*)d_x, n, d);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}
