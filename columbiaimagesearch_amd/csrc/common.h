// Shared host/device helpers for libcis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/cis_hip.h"

// ---- error plumbing: nothing throws or aborts across the C ABI --------------------------------
void cis_set_error(const char* fmt, ...);
int cis_lazy_init();         // selects the device chosen by cis_set_device(); CIS_ENODEVICE if none
int cis_current_device();

#define CIS_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            cis_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                          __LINE__);                                                          \
            return (_e == hipErrorOutOfMemory) ? CIS_ENOMEM : CIS_EHIP;                       \
        }                                                                                     \
    } while (0)

#define CIS_TRY(expr)              \
    do {                           \
        int _r = (expr);           \
        if (_r != CIS_OK) return _r; \
    } while (0)

#define CIS_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            cis_set_error(__VA_ARGS__); \
            return CIS_EINVAL;          \
        }                               \
    } while (0)

// ---- grow-only device buffer -------------------------------------------------------------------
// Every growth is a hipFree (which waits for the device) + hipMalloc: tens of milliseconds when it happens inside a timed loop.
// The counters let a harness prove that its timed region allocated nothing (cis_alloc_stats; bench.py reports them per leg).
extern long long g_cis_allocs, g_cis_alloc_bytes;
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return CIS_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            cis_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return CIS_ENOMEM;
        }
        cap = want;
        __atomic_add_fetch(&g_cis_allocs, 1, __ATOMIC_RELAXED);
        __atomic_add_fetch(&g_cis_alloc_bytes, (long long)want, __ATOMIC_RELAXED);
        // Test hook: CIS_POISON_ALLOC=1 fills every new workspace with 0xff bytes (NaN / -1): nothing may rely on what fresh memory
        // holds (the runtime hands back blocks this process freed earlier, with their contents -- see k_nchw3_to_nhwc).
        static const bool poison = getenv("CIS_POISON_ALLOC") != nullptr && atoi(getenv("CIS_POISON_ALLOC")) != 0;
        // (the fill runs on the null stream; the library's own streams are non-blocking: wait for it before anybody writes the block)
        if (poison && (hipMemset(p, 0xff, want) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) { cis_set_error("hipMemset (poison) failed"); return CIS_ENOMEM; }
        return CIS_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- numpy pairwise summation as a tiny stack program ---------------------------------------
// numpy's add.reduce over a contiguous axis sums n <= 128 elements with 8 interleaved
// accumulators and recursively halves longer runs (SURVEY.md section 8a item 7;
// oracle/lopq_oracle.py:np_pairwise_sum is the executable statement of the order).  The
// recursion is flattened on the host into a postfix program: op >= 0 pushes the sum of leaf
// `op`, op == -1 pops two values and pushes their sum (left + right).
#define CIS_PW_MAX_LEAVES 64
struct PwProg {
    int n;        // total elements
    int n_leaves; // 1 when n <= 128
    int n_ops;
    int16_t leaf_start[CIS_PW_MAX_LEAVES];
    int16_t leaf_len[CIS_PW_MAX_LEAVES];
    int8_t ops[2 * CIS_PW_MAX_LEAVES];
};
int cis_build_pwprog(int n, PwProg* out);  // CIS_EUNSUPPORTED when n needs > 64 leaves

#ifdef __HIPCC__
// Sum of elem(lo) .. elem(lo+n-1), n <= 128, in numpy's leaf order.  T is float or double; the
// file is compiled with -ffp-contract=off so every add/mul below rounds on its own.
template <typename T, class F>
__device__ __forceinline__ T pw_leaf(F elem, int lo, int n) {
    if (n < 8) {
        T res = (T)0;
        for (int i = 0; i < n; ++i) res = res + elem(lo + i);
        return res;
    }
    T r0 = elem(lo + 0), r1 = elem(lo + 1), r2 = elem(lo + 2), r3 = elem(lo + 3);
    T r4 = elem(lo + 4), r5 = elem(lo + 5), r6 = elem(lo + 6), r7 = elem(lo + 7);
    int i = 8;
    const int n8 = n - (n % 8);
    for (; i < n8; i += 8) {
        r0 = r0 + elem(lo + i + 0);
        r1 = r1 + elem(lo + i + 1);
        r2 = r2 + elem(lo + i + 2);
        r3 = r3 + elem(lo + i + 3);
        r4 = r4 + elem(lo + i + 4);
        r5 = r5 + elem(lo + i + 5);
        r6 = r6 + elem(lo + i + 6);
        r7 = r7 + elem(lo + i + 7);
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + elem(lo + i);
    return res;
}

template <typename T, class F>
__device__ __forceinline__ T pw_sum(const PwProg& P, F elem) {
    if (P.n_leaves == 1) return pw_leaf<T>(elem, 0, P.n);
    T stack[8];  // depth <= log2(64 leaves) + 1
    int sp = 0;
    for (int o = 0; o < P.n_ops; ++o) {
        const int op = P.ops[o];
        if (op >= 0) {
            stack[sp++] = pw_leaf<T>(elem, P.leaf_start[op], P.leaf_len[op]);
        } else {
            const T b = stack[--sp];
            const T a = stack[--sp];
            stack[sp++] = a + b;
        }
    }
    return stack[0];
}
#endif  // __HIPCC__
