// LOPQ model on MI355X: PCA projection, coarse assignment, local rotation, fine encoding.
//
// Replaces the arithmetic of lopq/lopq/model.py (predict :543-561/:980-1003, apply_PCA :961-978,
// predict_coarse :563-573, project :604-641, predict_fine :575-602, reconstruct :643-671) and
// lopq/lopq/utils.py:33-53 (predict_cluster) for whole batches.
//
// Exactness rules (SURVEY.md section 7 "hard parts"):
//  * every squared distance that feeds an argmin is computed as the reference does: (x - c)
//    rounded, squared rounded, summed in numpy's pairwise order (common.h), in float32 when both
//    operands are float32 and float64 otherwise; argmin takes the first minimum.  Compiled with
//    -ffp-contract=off so nothing is fused behind our back;
//  * the rotation R[c].(r - mu[c]) and the PCA product are float64 dot products whose summation
//    order inside BLAS is unspecified in the reference; here they are k-ascending fma chains.
#include <stdarg.h>

#include <mutex>

#include "lopq_model.h"

// ================================================================================================
// library plumbing
// ================================================================================================
static thread_local std::string g_err;
static int g_device = 0;
static bool g_inited = false;
static std::mutex g_init_mu;

void cis_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char* cis_last_error(void) { return g_err.c_str(); }
extern "C" int cis_version(void) { return 100; }

long long g_cis_allocs = 0, g_cis_alloc_bytes = 0;  // DevBuf::reserve (common.h)
extern "C" int cis_alloc_stats(int64_t* n_allocs, int64_t* n_bytes) {
    if (n_allocs) *n_allocs = (int64_t)__atomic_load_n(&g_cis_allocs, __ATOMIC_RELAXED);
    if (n_bytes) *n_bytes = (int64_t)__atomic_load_n(&g_cis_alloc_bytes, __ATOMIC_RELAXED);
    return CIS_OK;
}

extern "C" int cis_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int cis_set_device(int device) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    g_device = device;
    g_inited = false;
    return CIS_OK;
}

int cis_current_device() { return g_device; }

int cis_lazy_init() {
    std::lock_guard<std::mutex> lk(g_init_mu);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        cis_set_error("no HIP device visible: libcis_hip.so needs an MI355X (gfx950); there is no CPU fallback");
        return CIS_ENODEVICE;
    }
    if (g_device < 0 || g_device >= n) {
        cis_set_error("device %d out of range (%d visible)", g_device, n);
        return CIS_EINVAL;
    }
    CIS_CHECK_HIP(hipSetDevice(g_device));
    if (!g_inited) {
        hipDeviceProp_t p;
        CIS_CHECK_HIP(hipGetDeviceProperties(&p, g_device));
        if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
            cis_set_error("device %d is %s; this library is built for gfx950 only", g_device, p.gcnArchName);
            return CIS_ENODEVICE;
        }
        g_inited = true;
    }
    return CIS_OK;
}

static void pw_rec(int lo, int n, PwProg* P, bool* ok) {
    if (n <= 128) {
        if (P->n_leaves >= CIS_PW_MAX_LEAVES) {
            *ok = false;
            return;
        }
        P->leaf_start[P->n_leaves] = (int16_t)lo;
        P->leaf_len[P->n_leaves] = (int16_t)n;
        P->ops[P->n_ops++] = (int8_t)P->n_leaves;
        P->n_leaves++;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    pw_rec(lo, n2, P, ok);
    if (!*ok) return;
    pw_rec(lo + n2, n - n2, P, ok);
    if (!*ok) return;
    P->ops[P->n_ops++] = -1;
}

int cis_build_pwprog(int n, PwProg* out) {
    memset(out, 0, sizeof(*out));
    out->n = n;
    bool ok = true;
    pw_rec(0, n, out, &ok);
    if (!ok || n > 32000) {
        cis_set_error("vector length %d needs more than %d summation leaves", n, CIS_PW_MAX_LEAVES);
        return CIS_EUNSUPPORTED;
    }
    return CIS_OK;
}

// ================================================================================================
// kernels
// ================================================================================================
template <typename TI>
__global__ void k_to_f64(const TI* __restrict__ in, double* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (double)in[i];
}

// Y[n][D] = (X - mu) . P, float64, (16 TM) x 64 tile per 256-thread block, TM x 4 outputs per thread (TM = 4: 64 rows;
// TM = 2: 32 rows, twice the workgroups -- a batch of 8192 x 128 is only 256 tiles of 64 x 64, one per CU, with nothing to
// hide the load latency of its eight K slabs behind).
template <typename TX, bool SUBF32, int TM>
__global__ __launch_bounds__(256) void k_pca_gemm(const TX* __restrict__ X, const double* __restrict__ mu,
                                                  const double* __restrict__ P, double* __restrict__ Y,
                                                  int64_t n, int D_in, int D) {
    constexpr int BM = 16 * TM;
    __shared__ double sA[16][BM + 1];  // [k][row]
    __shared__ double sB[16][64];      // [k][col]
    const int tid = threadIdx.x;
    const int tr = tid / 16, tc = tid % 16;  // thread micro-tile origin: rows tr*TM.., cols tc*4..
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int col0 = blockIdx.y * 64;
    double acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    // register prefetch: the global loads of slab k0+16 are in flight while slab k0 is multiplied
    double ra[TM], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < TM; ++e) {
            const int idx = tid + e * 256;
            const int r = idx / 16, k = idx % 16;
            double v = 0.0;
            if (row0 + r < n && k0 + k < D_in) {
                if constexpr (SUBF32) {
                    const float df = (float)X[(row0 + r) * D_in + k0 + k] - (float)mu[k0 + k];  // float32 - float32
                    v = (double)df;
                } else {
                    v = (double)X[(row0 + r) * D_in + k0 + k] - mu[k0 + k];
                }
            }
            ra[e] = v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int k = idx / 64, c = idx % 64;
            double v = 0.0;
            if (k0 + k < D_in && col0 + c < D) v = P[(int64_t)(k0 + k) * D + col0 + c];
            rb[e] = v;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < D_in; k0 += 16) {
#pragma unroll
        for (int e = 0; e < TM; ++e) {
            const int idx = tid + e * 256;
            sA[idx % 16][idx / 16] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            sB[idx / 64][idx % 64] = rb[e];
        }
        __syncthreads();
        if (k0 + 16 < D_in) fetch(k0 + 16);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            double a[TM], b[4];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = sA[k][tr * TM + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[k][tc * 4 + j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t r = row0 + tr * TM + i;
        if (r >= n) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = col0 + tc * 4 + j;
            if (c < D) Y[r * D + c] = acc[i][j];
        }
    }
}

// The same product on the float64 matrix cores: v_mfma_f64_16x16x4_f64, 64 x 64 block tile, four waves of 32 x 32 (2 x 2 MFMA
// tiles), K advances 16 per LDS stage with the next stage's operands prefetched into registers.  Operand layout: lane l gives
// A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15]; result reg r of lane l is C[row = (l >> 4) + 4 r][col = l & 15].
// The register-tiled kernel above is LDS-bound (eight 8-byte reads per sixteen FMAs: 17-23 TFLOP/s); here a wave reads four
// operands per four MFMAs (4 x 2048 flop).
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <typename TX, bool SUBF32, int BK>
__global__ __launch_bounds__(256) void k_pca_gemm_mfma(const TX* __restrict__ X, const double* __restrict__ mu,
                                                       const double* __restrict__ P, double* __restrict__ Y,
                                                       int64_t n, int D_in, int D) {
    constexpr int BM = 64;
    constexpr int QK = BK / 4;            // quads of consecutive k per row and stage
    constexpr int NA = 64 * QK / 256;     // X quads per thread and stage (2 at BK = 32, 1 at BK = 16)
    constexpr int NB = BK * 32 / 256;     // P column pairs per thread and stage
    __shared__ double sA[2][BK][BM + 2];  // [stage][k][row]
    __shared__ double sB[2][BK][64 + 2];  // [stage][k][col]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int col0 = blockIdx.y * 64;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
    // a stage is 64 rows x 32 k of X (four consecutive k per thread and fetch: one 16-byte load for float32 rows) and
    // 32 k x 64 columns of P (two consecutive columns per thread and fetch)
    double ra[NA][4], rb[NB][2];
    const bool vec4 = (D_in % 4 == 0);
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;            // 64 * QK quads: row = idx / QK, k quad = idx % QK
            const int r = idx / QK, kq = (idx % QK) * 4;
            const bool ron = row0 + r < n;
            float xf[4];
            double xd[4];
            bool on[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { on[c] = ron && k0 + kq + c < D_in; xf[c] = 0.f; xd[c] = 0.0; }
            if constexpr (sizeof(TX) == 4) {
                if (vec4 && on[3]) {
                    const float4 q = *reinterpret_cast<const float4*>(X + (row0 + r) * D_in + k0 + kq);
                    xf[0] = q.x; xf[1] = q.y; xf[2] = q.z; xf[3] = q.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (on[c]) xf[c] = (float)X[(row0 + r) * D_in + k0 + kq + c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) xd[c] = (double)xf[c];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (on[c]) xd[c] = (double)X[(row0 + r) * D_in + k0 + kq + c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double v = 0.0;
                if (on[c]) {
                    if constexpr (SUBF32) v = (double)(xf[c] - (float)mu[k0 + kq + c]);  // float32 - float32
                    else v = xd[c] - mu[k0 + kq + c];
                }
                ra[e][c] = v;
            }
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;            // BK * 32 pairs: k = idx / 32, column pair = idx % 32
            const int k = idx >> 5, c = (idx & 31) * 2;
            rb[e][0] = 0.0; rb[e][1] = 0.0;
            if (k0 + k < D_in) {
                if (col0 + c + 1 < D && (D % 2 == 0)) {
                    const double2 q = *reinterpret_cast<const double2*>(P + (int64_t)(k0 + k) * D + col0 + c);
                    rb[e][0] = q.x; rb[e][1] = q.y;
                } else {
                    if (col0 + c < D) rb[e][0] = P[(int64_t)(k0 + k) * D + col0 + c];
                    if (col0 + c + 1 < D) rb[e][1] = P[(int64_t)(k0 + k) * D + col0 + c + 1];
                }
            }
        }
    };
    auto stash = [&](int st) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            const int r = idx / QK, kq = (idx % QK) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) sA[st][kq + c][r] = ra[e][c];
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;
            const int k = idx >> 5, c = (idx & 31) * 2;
            sB[st][k][c] = rb[e][0];
            sB[st][k][c + 1] = rb[e][1];
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int st = 0;
    for (int k0 = 0; k0 < D_in; k0 += BK) {
        const bool more = k0 + BK < D_in;
        if (more) fetch(k0 + BK);
#pragma unroll
        for (int k = 0; k < BK; k += 4) {
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[st][k + (lane >> 4)][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sB[st][k + (lane >> 4)][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(st ^ 1);
        __syncthreads();
        st ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r;
                const int c = col0 + wn * 32 + j * 16 + (lane & 15);
                if (row < n && c < D) Y[row * D + c] = acc[i][j][r];
            }
}

// The same product with 128 x 128 block tiles for passes that fill the chip (the encode's 65536-row passes at 4096 input dims:
// 137 GFLOP each): a wave owns 64 x 64 = 4 x 4 MFMA tiles, so a k-step of four reads eight operands for sixteen MFMAs (the 64 x 64
// form: four for four) and every staged element of X and P feeds twice as many products.  Same instruction and the same k order per
// output element as the form above -- bit-identical results.
template <typename TX, bool SUBF32>
__global__ __launch_bounds__(256) void k_pca_gemm_mfma128(const TX* __restrict__ X, const double* __restrict__ mu,
                                                          const double* __restrict__ P, double* __restrict__ Y,
                                                          int64_t n, int D_in, int D) {
    constexpr int BK = 16, BM = 128, BN = 128;
    __shared__ double sA[2][BK][BM + 2];  // [stage][k][row]
    __shared__ double sB[2][BK][BN + 2];  // [stage][k][col]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
    double ra[2][4], rb[4][2];
    const bool vec4 = (D_in % 4 == 0);
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + e * 256;            // 128 rows x 4 quads of consecutive k
            const int r = idx >> 2, kq = (idx & 3) * 4;
            const bool ron = row0 + r < n;
            float xf[4];
            double xd[4];
            bool on[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { on[c] = ron && k0 + kq + c < D_in; xf[c] = 0.f; xd[c] = 0.0; }
            if constexpr (sizeof(TX) == 4) {
                if (vec4 && on[3]) {
                    const float4 q = *reinterpret_cast<const float4*>(X + (row0 + r) * D_in + k0 + kq);
                    xf[0] = q.x; xf[1] = q.y; xf[2] = q.z; xf[3] = q.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (on[c]) xf[c] = (float)X[(row0 + r) * D_in + k0 + kq + c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) xd[c] = (double)xf[c];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (on[c]) xd[c] = (double)X[(row0 + r) * D_in + k0 + kq + c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double v = 0.0;
                if (on[c]) {
                    if constexpr (SUBF32) v = (double)(xf[c] - (float)mu[k0 + kq + c]);  // float32 - float32
                    else v = xd[c] - mu[k0 + kq + c];
                }
                ra[e][c] = v;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;            // 16 k x 64 column pairs
            const int k = idx >> 6, c = (idx & 63) * 2;
            rb[e][0] = 0.0; rb[e][1] = 0.0;
            if (k0 + k < D_in) {
                if (col0 + c + 1 < D && (D % 2 == 0)) {
                    const double2 q = *reinterpret_cast<const double2*>(P + (int64_t)(k0 + k) * D + col0 + c);
                    rb[e][0] = q.x; rb[e][1] = q.y;
                } else {
                    if (col0 + c < D) rb[e][0] = P[(int64_t)(k0 + k) * D + col0 + c];
                    if (col0 + c + 1 < D) rb[e][1] = P[(int64_t)(k0 + k) * D + col0 + c + 1];
                }
            }
        }
    };
    auto stash = [&](int st) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = tid + e * 256;
            const int r = idx >> 2, kq = (idx & 3) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) sA[st][kq + c][r] = ra[e][c];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int k = idx >> 6, c = (idx & 63) * 2;
            sB[st][k][c] = rb[e][0];
            sB[st][k][c + 1] = rb[e][1];
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int st = 0;
    for (int k0 = 0; k0 < D_in; k0 += BK) {
        const bool more = k0 + BK < D_in;
        if (more) fetch(k0 + BK);
#pragma unroll
        for (int k = 0; k < BK; k += 4) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[st][k + (lane >> 4)][wm * 64 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[st][k + (lane >> 4)][wn * 64 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(st ^ 1);
        __syncthreads();
        st ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + wm * 64 + i * 16 + (lane >> 4) + 4 * r;
                const int c = col0 + wn * 64 + j * 16 + (lane & 15);
                if (row < n && c < D) Y[row * D + c] = acc[i][j][r];
            }
}

// The two MFMA forms above with a fetch that is ONLY loads (round 4).  Their fetch lambdas branch on the tile borders and convert as
// they load, and the compiler answers with one `s_waitcnt vmcnt(0)` per load inside the k loop (build/lopq_model.s: global_load ->
// wait -> v_cvt, ten round trips in a row per stage): the "prefetch" was a chain of exposed latencies, hidden only by the CU's other
// workgroup (2.89 ms per 62500 x 4096 -> 256 pass = 0.61 of the float64 matrix peak).  Here every address is clamped into the
// arrays, the loads are unconditional 16-byte loads into registers that nothing touches before the MFMAs of the stage are issued,
// and the border masks, the centring and the conversions happen when the registers go to LDS.  Needs D_in % 4 == 0 and D % 2 == 0
// (the host keeps the forms above for other shapes).  Same instruction, same k order per output element: bit-identical results.
template <typename TX, bool SUBF32, int BM, int BK>
__global__ __launch_bounds__(256) void k_pca_gemm_mfma_pf(const TX* __restrict__ X, const double* __restrict__ mu,
                                                          const double* __restrict__ P, double* __restrict__ Y,
                                                          int64_t n, int D_in, int D) {
    constexpr int BN = BM, TW = BM / 32;     // a wave owns (BM / 2) x (BN / 2) = TW x TW MFMA tiles
    constexpr int QK = BK / 4;               // quads of consecutive k per row and stage
    constexpr int NA = BM * QK / 256;        // X quads per thread and stage
    constexpr int NB = BK * (BN / 2) / 256;  // P column pairs per thread and stage
    static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");
    __shared__ double sA[2][BK][BM + 2];     // [stage][k][row]
    __shared__ double sB[2][BK][BN + 2];     // [stage][k][col]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;
    f64x4 acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
    float4 xf4[NA];       // float32 rows: one 16-byte load per quad
    double2 xd2[NA][2];   // float64 rows: two
    double2 mu2[NA][2];
    double2 pb2[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            const int r = idx / QK, kq = (idx % QK) * 4;
            int64_t rr = row0 + r;
            rr = rr < n ? rr : n - 1;
            int kc = k0 + kq;
            kc = kc <= D_in - 4 ? kc : D_in - 4;
            if constexpr (sizeof(TX) == 4) {
                xf4[e] = *reinterpret_cast<const float4*>(X + rr * D_in + kc);   // (non-temporal: C3 +0.3 %, inside the spread -- round 6)
            } else {
                xd2[e][0] = *reinterpret_cast<const double2*>(X + rr * D_in + kc);
                xd2[e][1] = *reinterpret_cast<const double2*>(X + rr * D_in + kc + 2);
            }
            mu2[e][0] = *reinterpret_cast<const double2*>(mu + kc);
            mu2[e][1] = *reinterpret_cast<const double2*>(mu + kc + 2);
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;
            const int k = idx / (BN / 2), c = (idx % (BN / 2)) * 2;
            int kc = k0 + k;
            kc = kc < D_in ? kc : D_in - 1;
            int cc = col0 + c;
            cc = cc <= D - 2 ? cc : D - 2;
            pb2[e] = *reinterpret_cast<const double2*>(P + (int64_t)kc * D + cc);
        }
    };
    auto stash = [&](int st, int k0) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            const int r = idx / QK, kq = (idx % QK) * 4;
            const bool on = row0 + r < n && k0 + kq < D_in;  // whole quads: D_in % 4 == 0
            const double m4[4] = {mu2[e][0].x, mu2[e][0].y, mu2[e][1].x, mu2[e][1].y};
            double v[4];
            if constexpr (sizeof(TX) == 4) {
                const float x4[4] = {xf4[e].x, xf4[e].y, xf4[e].z, xf4[e].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (SUBF32) v[c] = (double)(x4[c] - (float)m4[c]);  // float32 - float32
                    else v[c] = (double)x4[c] - m4[c];
                }
            } else {
                const double x4[4] = {(double)xd2[e][0].x, (double)xd2[e][0].y, (double)xd2[e][1].x, (double)xd2[e][1].y};
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = x4[c] - m4[c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) sA[st][kq + c][r] = on ? v[c] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;
            const int k = idx / (BN / 2), c = (idx % (BN / 2)) * 2;
            const bool on = k0 + k < D_in && col0 + c < D;  // column pairs: D and col0 are even
            sB[st][k][c] = on ? pb2[e].x : 0.0;
            sB[st][k][c + 1] = on ? pb2[e].y : 0.0;
        }
    };
    fetch(0);
    stash(0, 0);
    __syncthreads();
    int st = 0;
    for (int k0 = 0; k0 < D_in; k0 += BK) {
        const bool more = k0 + BK < D_in;
        if (more) fetch(k0 + BK);
#pragma unroll
        for (int k = 0; k < BK; k += 4) {
            double a[TW], b[TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) a[i] = sA[st][k + (lane >> 4)][wm * (BM / 2) + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < TW; ++j) b[j] = sB[st][k + (lane >> 4)][wn * (BN / 2) + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(st ^ 1, k0 + BK);
        __syncthreads();
        st ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + wm * (BM / 2) + i * 16 + (lane >> 4) + 4 * r;
                const int c = col0 + wn * (BN / 2) + j * 16 + (lane & 15);
                if (row < n && c < D) Y[row * D + c] = acc[i][j][r];
            }
}

// Row L2 renormalisation (numpy: sqrt(add.reduce(y*y, axis=1)), then y / norm) and float32 cast.
// One 64-lane wave per row would break the summation order, so each thread owns a row.
__global__ void k_pca_finish(const double* __restrict__ Y, float* __restrict__ out, int64_t n, int D,
                             int renorm, PwProg prog) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double* y = Y + r * D;
    double nrm = 1.0;
    if (renorm) {
        auto sq = [&](int i) -> double { const double v = y[i]; return v * v; };
        nrm = sqrt(pw_sum<double>(prog, sq));
    }
    for (int i = 0; i < D; ++i) out[r * D + i] = (float)(renorm ? (y[i] / nrm) : y[i]);
}

// The same for 8 <= D <= 128, D % 8 == 0 (the usual descriptor sizes): numpy's pairwise sum is then ONE block of eight
// interleaved accumulators, r[j] = a[j] + a[8+j] + a[16+j] + ..., combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)).
// Eight lanes own the eight accumulators of a row (same additions in the same order, IEEE addition is commutative),
// so the loads and stores of a row are contiguous instead of one strided row per thread.
__global__ __launch_bounds__(64) void k_pca_finish8(const double* __restrict__ Y, float* __restrict__ out, int64_t n, int D) {
    const int j = threadIdx.x & 7;
    const int64_t r = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 3);
    const bool on = r < n;
    const double* y = Y + (on ? r : 0) * D;
    double v[16];  // D <= 128: the row's values stay in registers between the norm and the division (one wave per 8 rows)
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (i * 8 < D) ? y[i * 8 + j] : 0.0;
    double acc = v[0] * v[0];
#pragma unroll
    for (int i = 1; i < 16; ++i)
        if (i * 8 < D) acc = acc + v[i] * v[i];
    acc = acc + __shfl_xor(acc, 1);
    acc = acc + __shfl_xor(acc, 2);
    acc = acc + __shfl_xor(acc, 4);
    const double nrm = sqrt(acc);
    if (!on) return;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i * 8 < D) out[r * D + i * 8 + j] = (float)(v[i] / nrm);
}

// out[r][c] = sum_i (X[r][xoff+i] - C[c][i])^2 in numpy's order; 16 rows x 16 centroids / block.
template <typename T>
__global__ __launch_bounds__(256) void k_sqdist_rows(const T* __restrict__ X, int64_t ldx, int xoff,
                                                     const T* __restrict__ C, int64_t n, int ncent, int d,
                                                     T* __restrict__ out, PwProg prog, const int* __restrict__ only_if) {
    if (only_if && *only_if == 0) return;
    // the grid may be smaller than the tile count (predicated launches keep it small: they normally exit at once)
    for (int cb = blockIdx.y; cb * 16 < ncent; cb += gridDim.y)
        for (int64_t rb = blockIdx.x; rb * 16 < n; rb += gridDim.x) {
            const int c = cb * 16 + (threadIdx.x % 16);
            const int64_t r = rb * 16 + (threadIdx.x / 16);
            if (r >= n || c >= ncent) continue;
            const T* x = X + r * ldx + xoff;
            const T* cc = C + (int64_t)c * d;
            auto elem = [&](int i) -> T { const T df = x[i] - cc[i]; return df * df; };
            out[r * ncent + c] = pw_sum<T>(prog, elem);
        }
}

// First-minimum argmin over each row (numpy argmin tie rule).
template <typename T, typename TO>
__global__ void k_argmin_rows(const T* __restrict__ dist, int64_t n, int ncent, TO* __restrict__ out,
                              int ostride, int ooff, const int* __restrict__ only_if) {
    if (only_if && *only_if == 0) return;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const T* d = dist + r * ncent;
    T best = d[0];
    int bi = 0;
    for (int c = 1; c < ncent; ++c) {
        const T v = d[c];
        if (v < best) { best = v; bi = c; }
    }
    out[r * ostride + ooff] = (TO)bi;
}

// Coarse assignment when there are hundreds or thousands of clusters (the release configurations: V = 2048 / 4096): computing
// every one of the n x V distances in numpy's summation order is a 4 ms pass per 64 k vectors and split.  Here the matrix
// cores produce dt[c] = |c|^2 - 2 x.c in float64 (v_mfma_f64_16x16x4_f64; the vectors of a wave live in registers as the A
// operand, 64 centroids x 64 dims per LDS stage as B), a row keeps every centroid whose dt is within `2 slack` of the running
// minimum in a wave-private LDS list (about ln(V/64)+1 entries per row), and only the listed pairs are evaluated exactly
// as predict_cluster does (lopq/lopq/utils.py:33-53: ((x - C)**2).sum(axis=1) in the compute type, first minimum wins).
// slack = eps_rel (|x|^2 + max_c |c|^2) bounds |dt + |x|^2 - numpy's value| for every centroid, so the numpy argmin is always in the list:
// numpy's value d_c* <= d_c for all c  =>  dt_c* <= dt_cmin + 2 slack.  A list that overflows raises `fallback` and the exact
// kernels (predicated on that flag) redo the pass.
template <typename T, int KS /* h padded to 4*KS */>
__global__ __launch_bounds__(256) void k_coarse_mfma(const T* __restrict__ X, int64_t ldx, int h, const T* __restrict__ Call,
                                                     const double* __restrict__ cnorm /* [2][V] + [2] maxima */, int64_t n, int V,
                                                     uint16_t* __restrict__ out, PwProg prog, double eps_rel,
                                                     int* __restrict__ fallback) {
    constexpr int KC = (KS * 4 < 64) ? KS * 4 : 64;  // dims per LDS stage
    constexpr int NKC = (KS * 4) / KC;
    constexpr int CAP = 512;
    __shared__ double sB[2][64][KC + 2];  // [stage][centroid][k]
    __shared__ double sCn[2][64];
    __shared__ uint32_t sList[4][CAP];
    __shared__ double sXn[4][16];
    __shared__ unsigned long long sBest[4][16];
    __shared__ uint32_t sBestC[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.y;
    const int xoff = split * h;
    const T* C = Call + (size_t)split * V * h;
    const double* cn = cnorm + (size_t)split * V;
    const double cn_max = cnorm[2 * V + split];
    const int64_t row0 = (int64_t)blockIdx.x * 64 + wave * 16;
    // A operand: lane holds x[row = lane & 15][k = 4 step + (lane >> 4)]
    double a[KS];
    {
        const int64_t r = row0 + (lane & 15);
        const T* x = X + r * ldx + xoff;
        double sq = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 4 * s + (lane >> 4);
            a[s] = (r < n && k < h) ? (double)x[k] : 0.0;
            sq = fma(a[s], a[s], sq);
        }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        if (lane < 16) {
            sXn[wave][lane] = sq;
            sBest[wave][lane] = ~0ull;
            sBestC[wave][lane] = 0xffffffffu;
        }
    }
    __syncthreads();
    double slack2[4], lmin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        slack2[r] = 2.0 * eps_rel * (sXn[wave][(lane >> 4) + 4 * r] + cn_max);
        lmin[r] = __builtin_inf();
    }
    // stage fetch: 64 centroids x KC dims = 64 * KC / 4 quads of consecutive k, 256 threads
    constexpr int QPC = KC / 4;               // quads per centroid
    constexpr int NQ = 64 * QPC / 256;        // quads per thread (4 at KC = 64)
    static_assert(NQ >= 1, "stage too small");
    double rb[NQ][4];
    double rcn = 0.0;
    auto fetch = [&](int t, int kc) {
#pragma unroll
        for (int e = 0; e < NQ; ++e) {
            const int idx = tid + e * 256;
            const int c = t * 64 + idx / QPC, k = kc * KC + (idx % QPC) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) rb[e][q] = (c < V && k + q < h) ? (double)C[(size_t)c * h + k + q] : 0.0;
        }
        if (kc == 0 && tid < 64) rcn = (t * 64 + tid < V) ? cn[t * 64 + tid] : __builtin_inf();
    };
    auto stash = [&](int st, int kc) {
#pragma unroll
        for (int e = 0; e < NQ; ++e) {
            const int idx = tid + e * 256;
            double* d = &sB[st][idx / QPC][(idx % QPC) * 4];
            d[0] = rb[e][0]; d[1] = rb[e][1]; d[2] = rb[e][2]; d[3] = rb[e][3];
        }
        if (kc == 0 && tid < 64) sCn[st][tid] = rcn;
    };
    const int ntiles = (V + 63) / 64;
    const int nstages = ntiles * NKC;
    int wcnt = 0;        // entries in this wave's list (wave-uniform)
    bool over = false;
    f64x4 acc[4];
    fetch(0, 0);
    stash(0, 0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        const int sidx = t * NKC + kc, st = sidx & 1;
        const bool more = sidx + 1 < nstages;
        if (more) fetch((sidx + 1) / NKC, (sidx + 1) % NKC);
        if (kc == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] = 0.0;
        }
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            const double av = a[kc * (KC / 4) + s];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double bv = sB[st][j * 16 + (lane & 15)][4 * s + (lane >> 4)];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[j], 0, 0, 0);
            }
        }
        if (kc == NKC - 1) {
            const int cst = (sidx - kc) & 1;  // the stage that carried this tile's norms
            double dt[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double cnv = sCn[cst][j * 16 + (lane & 15)];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dt[j][r] = fma(-2.0, acc[j][r], cnv);
                    lmin[r] = fmin(lmin[r], dt[j][r]);
                }
            }
            unsigned mask = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double mrow = lmin[r];
                mrow = fmin(mrow, __shfl_xor(mrow, 1));
                mrow = fmin(mrow, __shfl_xor(mrow, 2));
                mrow = fmin(mrow, __shfl_xor(mrow, 4));
                mrow = fmin(mrow, __shfl_xor(mrow, 8));
                lmin[r] = mrow;
                const double thr = mrow + slack2[r];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (dt[j][r] <= thr) mask |= 1u << (j * 4 + r);
            }
            if (__ballot(mask != 0) != 0ull && !over) {
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const bool p = (mask >> b) & 1u;
                    const unsigned long long bal = __ballot(p);
                    if (bal == 0ull) continue;
                    const int cnt = __popcll(bal);
                    if (wcnt + cnt > CAP) { over = true; break; }
                    if (p) {
                        const int pos = wcnt + __popcll(bal & ((1ull << lane) - 1ull));
                        const int row = (lane >> 4) + 4 * (b & 3), c = t * 64 + (b >> 2) * 16 + (lane & 15);
                        sList[wave][pos] = ((uint32_t)row << 16) | (uint32_t)c;
                    }
                    wcnt += cnt;
                }
            }
        }
        if (more) stash(st ^ 1, (sidx + 1) % NKC);
        __syncthreads();
      }
    }
    if (over) {
        if (lane == 0) atomicOr(fallback, 1);  // the exact kernels redo the pass
        wcnt = 0;
    }
    // exact re-check of the listed pairs, numpy's order and type; first minimum per row
    auto exact = [&](uint32_t e) -> double {
        const int64_t r = row0 + (e >> 16);
        const T* x = X + r * ldx + xoff;
        const T* cc = C + (size_t)(e & 0xffffu) * h;
        auto elem = [&](int i) -> T { const T df = x[i] - cc[i]; return df * df; };
        return (double)pw_sum<T>(prog, elem);
    };
    for (int i = lane; i < wcnt; i += 64) {
        const uint32_t e = sList[wave][i];
        if (row0 + (e >> 16) >= n) continue;
        const double v = exact(e);
        if (v == v) atomicMin(&sBest[wave][e >> 16], (unsigned long long)__double_as_longlong(v));
    }
    __syncthreads();
    for (int i = lane; i < wcnt; i += 64) {
        const uint32_t e = sList[wave][i];
        if (row0 + (e >> 16) >= n) continue;
        const double v = exact(e);
        if ((unsigned long long)__double_as_longlong(v) == sBest[wave][e >> 16]) atomicMin(&sBestC[wave][e >> 16], e & 0xffffu);
    }
    __syncthreads();
    if (lane < 16 && row0 + lane < n && !over) {
        const uint32_t c = sBestC[wave][lane];
        out[(row0 + lane) * 2 + split] = (uint16_t)(c == 0xffffffffu ? 0u : c);
    }
}

// The same for the float32 compute type (float32 vectors and centroids: what the reference's k-means returns for float32
// training data) on the float32 matrix cores, v_mfma_f32_32x32x2_f32: twice the float64 rate, half the LDS bytes.  numpy's
// value is itself a float32 sum here, so the slack is of the order of 2^-17 (|x|^2 + |c|^2) whatever produces dt; a wave
// owns 32 vectors (A operand: lane l holds x[row = l & 31][k = 2 s + (l >> 5)]), a stage is 64 centroids x 64 dims stored
// [centroid][k parity][k / 2] with a row stride of 65 words (conflict-free operand reads), the result register r of lane l is
// row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KS /* h padded to 2*KS, a multiple of 32 */>
__global__ __launch_bounds__(256) void k_coarse_mfma32(const float* __restrict__ X, int64_t ldx, int h, const float* __restrict__ Call,
                                                       const double* __restrict__ cnorm, int64_t n, int V,
                                                       uint16_t* __restrict__ out, PwProg prog, double eps_rel,
                                                       int* __restrict__ fallback) {
    constexpr int KC = 64;             // dims per LDS stage
    constexpr int NKC = (2 * KS) / KC;
    constexpr int CAP = 1024;
    __shared__ float sB[2][64][65];
    __shared__ float sCn[2][64];
    __shared__ uint32_t sList[4][CAP];
    __shared__ float sXn[4][32];
    __shared__ unsigned long long sBest[4][32];
    __shared__ uint32_t sBestC[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.y;
    const int xoff = split * h;
    const float* C = Call + (size_t)split * V * h;
    const double* cn = cnorm + (size_t)split * V;
    const double cn_max = cnorm[2 * V + split];
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
    float a[KS];
    {
        const int64_t r = row0 + (lane & 31);
        const float* x = X + r * ldx + xoff;
        double sq = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 2 * s + (lane >> 5);
            a[s] = (r < n && k < h) ? x[k] : 0.f;
            sq = fma((double)a[s], (double)a[s], sq);
        }
        sq += __shfl_xor(sq, 32);
        if (lane < 32) {
            sXn[wave][lane] = (float)(2.0 * eps_rel * (sq + cn_max) * 1.0000002);  // the row's 2 x slack, rounded up
            sBest[wave][lane] = ~0ull;
            sBestC[wave][lane] = 0xffffffffu;
        }
    }
    __syncthreads();
    float slack2[16], lmin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        slack2[r] = sXn[wave][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
        lmin[r] = __builtin_inff();
    }
    // stage fetch: 64 centroids x 16 quads of consecutive k = 1024 quads, four per thread
    float rb[4][4];
    float rcn = 0.f;
    auto fetch = [&](int t, int kc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int c = t * 64 + (idx >> 4), k = kc * KC + (idx & 15) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) rb[e][q] = (c < V && k + q < h) ? C[(size_t)c * h + k + q] : 0.f;
        }
        if (kc == 0 && tid < 64) rcn = (t * 64 + tid < V) ? (float)cn[t * 64 + tid] : __builtin_inff();
    };
    auto stash = [&](int st, int kc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            float* d = &sB[st][idx >> 4][0];
            const int s0 = (idx & 15) * 2;  // k = 4 (idx & 15) + q  ->  parity q & 1, index s0 + (q >> 1)
            d[s0] = rb[e][0]; d[32 + s0] = rb[e][1]; d[s0 + 1] = rb[e][2]; d[32 + s0 + 1] = rb[e][3];
        }
        if (kc == 0 && tid < 64) sCn[st][tid] = rcn;
    };
    const int ntiles = (V + 63) / 64;
    const int nstages = ntiles * NKC;
    int wcnt = 0;
    bool over = false;
    f32x16 acc[2];
    fetch(0, 0);
    stash(0, 0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        const int sidx = t * NKC + kc, st = sidx & 1;
        const bool more = sidx + 1 < nstages;
        if (more) fetch((sidx + 1) / NKC, (sidx + 1) % NKC);
        if (kc == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const float av = a[kc * (KC / 2) + s];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bv = sB[st][j * 32 + (lane & 31)][(lane >> 5) * 32 + s];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
            }
        }
        if (kc == NKC - 1) {
            const int cst = (sidx - kc) & 1;
            float dt[2][16];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float cnv = sCn[cst][j * 32 + (lane & 31)];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dt[j][r] = fmaf(-2.f, acc[j][r], cnv);
                    lmin[r] = fminf(lmin[r], dt[j][r]);
                }
            }
            unsigned mask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float mrow = lmin[r];
                mrow = fminf(mrow, __shfl_xor(mrow, 1));
                mrow = fminf(mrow, __shfl_xor(mrow, 2));
                mrow = fminf(mrow, __shfl_xor(mrow, 4));
                mrow = fminf(mrow, __shfl_xor(mrow, 8));
                mrow = fminf(mrow, __shfl_xor(mrow, 16));
                lmin[r] = mrow;
                const float thr = mrow + slack2[r];
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (dt[j][r] <= thr) mask |= 1u << (j * 16 + r);
            }
            if (__ballot(mask != 0) != 0ull && !over) {
#pragma unroll
                for (int b = 0; b < 32; ++b) {
                    const bool p = (mask >> b) & 1u;
                    const unsigned long long bal = __ballot(p);
                    if (bal == 0ull) continue;
                    const int cnt = __popcll(bal);
                    if (wcnt + cnt > CAP) { over = true; break; }
                    if (p) {
                        const int pos = wcnt + __popcll(bal & ((1ull << lane) - 1ull));
                        const int r = b & 15;
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), c = t * 64 + (b >> 4) * 32 + (lane & 31);
                        sList[wave][pos] = ((uint32_t)row << 16) | (uint32_t)c;
                    }
                    wcnt += cnt;
                }
            }
        }
        if (more) stash(st ^ 1, (sidx + 1) % NKC);
        __syncthreads();
      }
    }
    if (over) {
        if (lane == 0) atomicOr(fallback, 1);  // the exact kernels redo the pass
        wcnt = 0;
    }
    auto exact = [&](uint32_t e) -> double {
        const int64_t r = row0 + (e >> 16);
        const float* x = X + r * ldx + xoff;
        const float* cc = C + (size_t)(e & 0xffffu) * h;
        auto elem = [&](int i) -> float { const float df = x[i] - cc[i]; return df * df; };
        return (double)pw_sum<float>(prog, elem);
    };
    for (int i = lane; i < wcnt; i += 64) {
        const uint32_t e = sList[wave][i];
        if (row0 + (e >> 16) >= n) continue;
        const double v = exact(e);
        if (v == v) atomicMin(&sBest[wave][e >> 16], (unsigned long long)__double_as_longlong(v));
    }
    __syncthreads();
    for (int i = lane; i < wcnt; i += 64) {
        const uint32_t e = sList[wave][i];
        if (row0 + (e >> 16) >= n) continue;
        const double v = exact(e);
        if ((unsigned long long)__double_as_longlong(v) == sBest[wave][e >> 16]) atomicMin(&sBestC[wave][e >> 16], e & 0xffffu);
    }
    __syncthreads();
    if (lane < 32 && row0 + lane < n && !over) {
        const uint32_t c = sBestC[wave][lane];
        out[(row0 + lane) * 2 + split] = (uint16_t)(c == 0xffffffffu ? 0u : c);
    }
}

// Fine codes in one pass (predict_fine, lopq/lopq/model.py:575-602 -> predict_cluster, lopq/lopq/utils.py:33-53): a thread
// owns one (vector, sub-quantizer), keeps its w projected values in registers, walks the K sub-centroids staged in LDS
// (every lane reads the same address: broadcast) and keeps the first minimum.  Every distance is summed exactly as numpy
// does for n = w <= 128 (pw_leaf); nothing but the code leaves the kernel -- the n x K distance matrix of the two-kernel
// path (k_sqdist_rows + k_argmin_rows) is 134 MB per sub-quantizer and chunk.
template <int W>
__global__ __launch_bounds__(256) void k_fine_codes(const double* __restrict__ proj /* [n][D] */, int D,
                                                    const double* __restrict__ subs /* [M][K][W] */, int64_t n, int K, int M,
                                                    uint8_t* __restrict__ fine /* [n][M] */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sc = reinterpret_cast<double*>(smem);  // [K][W]
    const int j = blockIdx.y;
    const double* src = subs + (size_t)j * K * W;
    for (int e = threadIdx.x; e < K * W; e += 256) sc[e] = src[e];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    double x[W];
#pragma unroll
    for (int i = 0; i < W; ++i) x[i] = proj[r * D + j * W + i];
    double best = 0.0;
    int bi = 0;
    for (int k = 0; k < K; ++k) {
        const double* c = sc + k * W;
        auto elem = [&](int i) -> double { const double df = x[i] - c[i]; return df * df; };
        const double dd = pw_leaf<double>(elem, 0, W);
        if (k == 0 || dd < best) { best = dd; bi = k; }
    }
    fine[r * M + j] = (uint8_t)bi;
}

// Fine codes on the float32 matrix cores with an exact re-check (round 3; the scheme of k_coarse_mfma32).
// dt[k] = |c_k|^2 - 2 x.c_k for the K sub-centroids of sub-quantizer j comes from v_mfma_f32_32x32x2_f32 with the CENTROIDS as
// the A operand (rows) and the wave's 32 vectors as the B operand (columns): lane l ends up with 16 of the 32 centroids of a
// tile for vector l & 31, so the minimum over centroids is in-lane plus one exchange with lane l ^ 32.  |dt + |x|^2 - d| <=
// slack = (W + 8) 2^-24 (|x|^2 + max|c|^2) bounds the float32 path (inputs rounded to float32, W-term float32 dot product,
// the final fma) against numpy's float64 value d of predict_cluster (lopq/lopq/utils.py:33-53), so every centroid within
// 2 x slack of the row's minimum (a first pass of the same products) is listed -- numpy's argmin always is.  A row with ONE
// listed centroid is done; rows with several (near-ties, duplicate centroids) are evaluated exactly: float64, numpy's pairwise
// order (pw_leaf), first minimum wins.
// A wave whose list overflows (hundreds of identical sub-centroids) evaluates its 32 rows against all centroids itself.
template <int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fine_mfma(
                                                   const double* __restrict__ proj /* [n][D] */, int D,
                                                   const double* __restrict__ subs /* [M][K][W] */, int64_t n, int K, int M,
                                                   uint8_t* __restrict__ fine /* [n][M] */, int reps) {
    // score[k] = x.c_k - |c_k|^2 / 2 = -dt[k] / 2: the norm rides in the product as one more k-step (A = |c|^2 / 2, B = -1), so the
    // vector work per tile is sixteen max / compare instructions and nothing else
    constexpr int KS = W / 2 + 1;  // MFMA steps (two dims each; the last one carries the norm)
    constexpr int PITCH = W + 3;   // odd: the 32 centroid rows of an operand read fall on 32 banks
    constexpr int CAP = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KP = (K + 31) & ~31;
    float* sA = reinterpret_cast<float*>(smem);              // [KP][PITCH] float32 sub-centroids, |c|^2 / 2, 0 (rows >= K: zeros, 3e38)
    uint32_t* sList = reinterpret_cast<uint32_t*>(sA + (size_t)KP * PITCH); // [4][CAP] (row << 16 | centroid)
    unsigned long long* sBest = reinterpret_cast<unsigned long long*>(sList + 4 * CAP);  // [4][32]
    uint32_t* sBestC = reinterpret_cast<uint32_t*>(sBest + 4 * 32);                       // [4][32]
    uint32_t* sCnt = sBestC + 4 * 32;                                                     // [4][32] listed candidates per row
    float* sCmax = reinterpret_cast<float*>(sCnt + 4 * 32);                               // [4] partial maxima of |c|^2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.y;
    const double* src = subs + (size_t)j * K * W;
    float cmx = 0.f;
    for (int k = tid; k < KP; k += 256) {
        double sq = 0.0;
        // (loads first, from a clamped row, then the mask: `k < K ? src[..] : 0` became a branch and a full wait per element)
        const int kc = k < K ? k : K - 1;
        double row[W];
#pragma unroll
        for (int i = 0; i < W; i += 2) {
            const double2 q = *reinterpret_cast<const double2*>(src + (size_t)kc * W + i);  // W even, rows 16-byte aligned
            row[i] = q.x; row[i + 1] = q.y;
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const double v = k < K ? row[i] : 0.0;
            sA[k * PITCH + i] = (float)v;
            sq = fma(v, v, sq);
        }
        sA[k * PITCH + W] = k < K ? 0.5f * (float)sq : 3.0e38f;
        sA[k * PITCH + W + 1] = 0.f;
        cmx = fmaxf(cmx, k < K ? (float)sq : 0.f);
    }
    for (int o = 32; o > 0; o >>= 1) cmx = fmaxf(cmx, __shfl_xor(cmx, o));
    if (lane == 0) sCmax[wave] = cmx;
    __syncthreads();
    const float cmax = fmaxf(fmaxf(sCmax[0], sCmax[1]), fmaxf(sCmax[2], sCmax[3]));
    // the staged codebook serves `reps` groups of 128 vectors (staging it costs as much as one group's matrix products)
    for (int rep = 0; rep < reps; ++rep) {
    const int64_t row0 = ((int64_t)blockIdx.x * reps + rep) * 128 + wave * 32;
    if (row0 >= n) break;
    const int64_t r = row0 + (lane & 31);
    float a[KS];
    double xsq = 0.0;
    {
        const double* x = proj + r * D + j * W;
#pragma unroll
        for (int s2 = 0; s2 < KS - 1; ++s2) {
            const double v = r < n ? x[2 * s2 + (lane >> 5)] : 0.0;
            a[s2] = (float)v;
            xsq = fma(v, v, xsq);
        }
        a[KS - 1] = (lane >> 5) ? 0.f : -1.f;
        xsq += __shfl_xor(xsq, 32);
    }
    if (lane < 32) {
        sBest[wave * 32 + lane] = ~0ull;
        sBestC[wave * 32 + lane] = 0xffffffffu;
        sCnt[wave * 32 + lane] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // |score + (d - |x|^2) / 2| <= slack / 2, slack = (W + 8) 2^-24 (|x|^2 + max |c|^2): inputs rounded to float32 (2 x 2^-24 per
    // product), W + 1 float32 additions of the accumulation (2^-24 of the partial sums each), |c|^2 rounded
    const float slack1 = (float)((double)(W + 8) * 5.9604644775390625e-8 * (xsq + (double)cmax) * 1.0000002);  // = 2 x slack / 2
    int wcnt = 0;
    bool over = false;
    const int ntiles = KP / 32;
    // the matrix products of tile t + 1 are issued before the vector work on tile t: the matrix pipe runs under it
    auto tile_products = [&](int t) -> f32x16 {
        f32x16 c;
#pragma unroll
        for (int q = 0; q < 16; ++q) c[q] = 0.f;
        const float* arow = sA + (size_t)(t * 32 + (lane & 31)) * PITCH + (lane >> 5);
        float av[KS];
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) av[s2] = arow[2 * s2];
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], a[s2], c, 0, 0, 0);
        return c;
    };
    // pass 1: the row's best score over all sub-centroids
    float rmax = -__builtin_inff();
    f32x16 acc_next = tile_products(0);
    for (int t = 0; t < ntiles; ++t) {
        const f32x16 acc = acc_next;
        if (t + 1 < ntiles) acc_next = tile_products(t + 1);
#pragma unroll
        for (int q = 0; q < 16; ++q) rmax = fmaxf(rmax, acc[q]);
    }
    rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
    // pass 2: the same products again (bit-identical), every centroid within 2 x slack of the best is listed -- one per row
    // as a rule, and a row with a single candidate needs no exact evaluation at all
    const float thr = rmax - slack1;
    acc_next = tile_products(0);
    for (int t = 0; t < ntiles; ++t) {
        const f32x16 acc = acc_next;
        if (t + 1 < ntiles) acc_next = tile_products(t + 1);
        unsigned mask = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (acc[q] >= thr) mask |= 1u << q;
        // lanes with candidates append them one bit at a time (one iteration as a rule)
        while (!over) {
            const bool p = mask != 0u;
            const unsigned long long bal = __ballot(p);
            if (bal == 0ull) break;
            const int cnt = __popcll(bal);
            if (wcnt + cnt > CAP) { over = true; break; }
            if (p) {
                const int q = __builtin_ctz(mask);
                mask &= mask - 1u;
                const int pos = wcnt + __popcll(bal & ((1ull << lane) - 1ull));
                const int c = t * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                sList[wave * CAP + pos] = ((uint32_t)(lane & 31) << 16) | (uint32_t)c;
                atomicAdd(&sCnt[wave * 32 + (lane & 31)], 1u);
            }
            wcnt += cnt;
        }
    }
    auto exact = [&](int row, int c) -> double {
        const double* x = proj + (row0 + row) * D + j * W;
        const double* cc = src + (size_t)c * W;
        auto elem = [&](int i) -> double { const double df = x[i] - cc[i]; return df * df; };
        return pw_leaf<double>(elem, 0, W);
    };
    if (over) {
        // every centroid for the wave's rows: lane l takes row l & 31 and the centroids of its half, in index order
        double best = 0.0;
        int bi = -1;
        if (row0 + (lane & 31) < n) {
            const int half = (K + 1) / 2, k0 = (lane >> 5) * half, k1 = (k0 + half < K) ? k0 + half : K;
            for (int k = k0; k < k1; ++k) {
                const double dd = exact(lane & 31, k);
                if (bi < 0 || dd < best) { best = dd; bi = k; }
            }
        }
        const double ob = __shfl_xor(best, 32);
        const int obi = __shfl_xor(bi, 32);
        if (lane < 32 && row0 + lane < n) {
            // the upper half's centroid wins only with a strictly smaller distance (first minimum)
            const int code = (obi >= 0 && (bi < 0 || ob < best)) ? obi : bi;
            fine[(row0 + lane) * M + j] = (uint8_t)code;
        }
        continue;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rows with ONE listed centroid are done; the others (near-ties, duplicate centroids) are evaluated exactly: float64,
    // numpy's order, first minimum
    for (int i = lane; i < wcnt; i += 64) {
        const uint32_t e = sList[wave * CAP + i];
        if (sCnt[wave * 32 + (e >> 16)] == 1u) sBestC[wave * 32 + (e >> 16)] = e & 0xffffu;
    }
    bool any_multi = false;
    for (int i = lane; i < wcnt; i += 64) any_multi = any_multi || sCnt[wave * 32 + (sList[wave * CAP + i] >> 16)] > 1u;
    if (__ballot(any_multi) != 0ull) {
        for (int i = lane; i < wcnt; i += 64) {
            const uint32_t e = sList[wave * CAP + i];
            if (row0 + (e >> 16) >= n || sCnt[wave * 32 + (e >> 16)] <= 1u) continue;
            const double v = exact((int)(e >> 16), (int)(e & 0xffffu));
            atomicMin(&sBest[wave * 32 + (e >> 16)], (unsigned long long)__double_as_longlong(v));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < wcnt; i += 64) {
            const uint32_t e = sList[wave * CAP + i];
            if (row0 + (e >> 16) >= n || sCnt[wave * 32 + (e >> 16)] <= 1u) continue;
            const double v = exact((int)(e >> 16), (int)(e & 0xffffu));
            if ((unsigned long long)__double_as_longlong(v) == sBest[wave * 32 + (e >> 16)]) atomicMin(&sBestC[wave * 32 + (e >> 16)], e & 0xffffu);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 32 && row0 + lane < n) {
        const uint32_t c = sBestC[wave * 32 + lane];
        fine[(row0 + lane) * M + j] = (uint8_t)(c == 0xffffffffu ? 0u : c);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    }
}

// ---- grouping rows by coarse cluster so that a tile of 64 vectors shares one rotation ----------
struct ProjTile { int split, cluster, start, count; };

// Workgroup-aggregated: a histogram of the block's rows in LDS first, then one global atomic per non-empty bin (65536
// rows hammering 2V counters one by one serialise in the L2).  V <= 1024 uses the LDS path.
__global__ __launch_bounds__(256) void k_group_hist(const uint16_t* __restrict__ coarse, int64_t n, int V, int* __restrict__ counts) {
    extern __shared__ int s_bins[];  // [2V] or nothing
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (V > 1024) {
        if (r < n) {
            atomicAdd(&counts[coarse[r * 2 + 0]], 1);
            atomicAdd(&counts[V + coarse[r * 2 + 1]], 1);
        }
        return;
    }
    for (int i = threadIdx.x; i < 2 * V; i += 256) s_bins[i] = 0;
    __syncthreads();
    if (r < n) {
        atomicAdd(&s_bins[coarse[r * 2 + 0]], 1);
        atomicAdd(&s_bins[V + coarse[r * 2 + 1]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * V; i += 256)
        if (s_bins[i]) atomicAdd(&counts[i], s_bins[i]);
}

// single block: exclusive scans over the 2V (split, cluster) bins; emits the tile descriptors.  A thread owns a contiguous
// range of bins (row offsets restart at the second split, tile numbers run on); with more bins than threads every thread
// emits the tiles of its own bins, with few large bins the block emits them together.
__global__ __launch_bounds__(256) void k_group_scan(const int* __restrict__ counts, int V, int* __restrict__ offsets,
                                                    int* __restrict__ cursor, ProjTile* __restrict__ tiles, int* __restrict__ n_tiles,
                                                    int tile_rows) {
    extern __shared__ int s_tbase[];  // [2V] first tile of every bin
    __shared__ int s_scan[3][256];
    const int tid = threadIdx.x;
    const int nb = 2 * V, per = (nb + 255) / 256;
    const int b0 = (tid * per < nb) ? tid * per : nb, b1 = (b0 + per < nb) ? b0 + per : nb;
    int loc[3] = {0, 0, 0};  // rows of split 0, rows of split 1, tiles
    for (int b = b0; b < b1; ++b) {
        const int cnt = counts[b];
        loc[b < V ? 0 : 1] += cnt;
        loc[2] += (cnt + tile_rows - 1) / tile_rows;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) s_scan[q][tid] = loc[q];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        int v[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = (tid >= d) ? s_scan[q][tid - d] : 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) s_scan[q][tid] += v[q];
        __syncthreads();
    }
    int off[2] = {s_scan[0][tid] - loc[0], s_scan[1][tid] - loc[1]};
    int t = s_scan[2][tid] - loc[2];
    if (tid == 255) *n_tiles = s_scan[2][255];
    const bool own = nb > 256;
    for (int b = b0; b < b1; ++b) {
        const int cnt = counts[b], s = b < V ? 0 : 1;
        offsets[b] = off[s];
        cursor[b] = 0;
        s_tbase[b] = t;
        if (own) {
            for (int r = 0; r < cnt; r += tile_rows) {
                ProjTile pt;
                pt.split = s; pt.cluster = b - s * V; pt.start = off[s] + r;
                pt.count = (cnt - r < tile_rows) ? (cnt - r) : tile_rows;
                tiles[t++] = pt;
            }
        } else {
            t += (cnt + tile_rows - 1) / tile_rows;
        }
        off[s] += cnt;
    }
    if (own) return;
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int cnt = counts[b], nbt = (cnt + tile_rows - 1) / tile_rows;
        for (int i = tid; i < nbt; i += 256) {
            ProjTile pt;
            pt.split = b / V; pt.cluster = b - pt.split * V; pt.start = offsets[b] + i * tile_rows;
            pt.count = (cnt - i * tile_rows < tile_rows) ? (cnt - i * tile_rows) : tile_rows;
            tiles[s_tbase[b] + i] = pt;
        }
    }
}

__global__ __launch_bounds__(256) void k_group_scatter(const uint16_t* __restrict__ coarse, int64_t n, int V,
                                                       const int* __restrict__ offsets, int* __restrict__ cursor,
                                                       int* __restrict__ perm /* [2][n] */) {
    extern __shared__ int s_bins[];  // [2V] counts, then [2V] reserved bases
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (V > 1024) {
        if (r >= n) return;
        for (int s = 0; s < 2; ++s) {
            const int c = coarse[r * 2 + s];
            const int p = offsets[s * V + c] + atomicAdd(&cursor[s * V + c], 1);
            perm[(int64_t)s * n + p] = (int)r;
        }
        return;
    }
    // rank inside the block through LDS atomics, one global atomic per non-empty bin reserves the block's range
    int* s_base = s_bins + 2 * V;
    for (int i = threadIdx.x; i < 2 * V; i += 256) s_bins[i] = 0;
    __syncthreads();
    int c[2] = {0, 0}, lr[2] = {0, 0};
    if (r < n) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            c[s] = s * V + coarse[r * 2 + s];
            lr[s] = atomicAdd(&s_bins[c[s]], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * V; i += 256) s_base[i] = s_bins[i] ? atomicAdd(&cursor[i], s_bins[i]) : 0;
    __syncthreads();
    if (r < n) {
#pragma unroll
        for (int s = 0; s < 2; ++s) perm[(int64_t)s * n + offsets[c[s]] + s_base[c[s]] + lr[s]] = (int)r;
    }
}

// out[row][s*h + i] = sum_k R[s][c][i][k] * ((double)(x[row][s*h+k] - C[s][c][k]) - mu[s][c][k])
// for a tile of <= 64 rows of one (split, cluster); grid.y tiles the h outputs by 64.
template <typename CT>
__global__ __launch_bounds__(256) void k_project_tiles(const CT* __restrict__ X, const CT* __restrict__ Cs,
                                                       const double* __restrict__ Rt, const double* __restrict__ mus,
                                                       const ProjTile* __restrict__ tiles, const int* __restrict__ n_tiles,
                                                       const int* __restrict__ perm, int64_t n, int V, int h, int D,
                                                       double* __restrict__ out) {
    if ((int)blockIdx.x >= *n_tiles) return;
    const ProjTile pt = tiles[blockIdx.x];
    __shared__ double sA[16][64 + 1];  // [k][vec]
    __shared__ double sB[16][64];      // [k][i]
    __shared__ int srow[64];
    const int tid = threadIdx.x;
    const int tr = tid / 16, tc = tid % 16;
    const int i0 = blockIdx.y * 64;
    if (tid < 64) srow[tid] = (tid < pt.count) ? perm[(int64_t)pt.split * n + pt.start + tid] : -1;
    __syncthreads();
    const CT* Cc = Cs + ((int64_t)pt.split * V + pt.cluster) * h;
    const double* mu = mus + ((int64_t)pt.split * V + pt.cluster) * h;
    const double* R = Rt + ((int64_t)pt.split * V + pt.cluster) * h * h;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    // The operands of stage k0 + 16 are fetched (loads only: clamped addresses, nothing converted) while stage k0 is multiplied, and
    // go to LDS afterwards with the border masks (round 4; before, `if (inside) load` gave one exposed round trip per load group and
    // the loads of a stage started only after the previous stage's products).  Same products in the same order: bit-identical.
    CT xa[4], ca[4];
    double ma[4], rb[4];
    const int row_any = srow[0];  // a tile holds at least one row
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int v = idx / 16, k = idx % 16;
            const int row = srow[v];
            const int rowc = row >= 0 ? row : row_any;
            const int kc = k0 + k < h ? k0 + k : h - 1;
            xa[e] = X[(int64_t)rowc * D + pt.split * h + kc];
            ca[e] = Cc[kc];
            ma[e] = mu[kc];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int k = idx / 64, i = idx % 64;
            const int kc = k0 + k < h ? k0 + k : h - 1;
            const int ic = i0 + i < h ? i0 + i : h - 1;
            rb[e] = R[(int64_t)kc * h + ic];
        }
    };
    auto stash = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int v = idx / 16, k = idx % 16;
            const CT res = xa[e] - ca[e];  // rounds in CT
            const double val = (double)res - ma[e];
            sA[k][v] = (srow[v] >= 0 && k0 + k < h) ? val : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int k = idx / 64, i = idx % 64;
            sB[k][i] = (k0 + k < h && i0 + i < h) ? rb[e] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < h; k0 += 16) {
        stash(k0);
        __syncthreads();
        if (k0 + 16 < h) fetch(k0 + 16);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[k][tr * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[k][tc * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow[tr * 4 + i];
        if (row < 0) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = i0 + tc * 4 + j;
            if (col < h) out[(int64_t)row * D + pt.split * h + col] = acc[i][j];
        }
    }
}

// The same tile on the float64 matrix cores (round 4).  k_project_tiles reads two LDS operands per four fused multiply-adds and is a
// single chain over ascending k per output -- exactly what v_mfma_f64_16x16x4_f64 computes per element (tools/probes/mfma_f64_order.hip),
// so this kernel changes no bit.  A = R (rows = outputs i: wave w owns the 16 outputs blockIdx.y * 64 + 16 w ..., loaded straight from
// global memory), B = the tile's centred residuals (columns = the tile's <= 64 vectors, four column tiles per wave; staged in LDS with a
// pitch of KC + 4), K in chunks of KC = 4 CH (h % KC == 0).
template <typename CT, int CH>
__global__ __launch_bounds__(256) void k_project_tiles_mfma(const CT* __restrict__ X, const CT* __restrict__ Cs,
                                                            const double* __restrict__ Rt, const double* __restrict__ mus,
                                                            const ProjTile* __restrict__ tiles, const int* __restrict__ n_tiles,
                                                            const int* __restrict__ perm, int64_t n, int V, int h, int D,
                                                            double* __restrict__ out) {
    if ((int)blockIdx.x >= *n_tiles) return;
    constexpr int KC = 4 * CH, PITCH = KC + 4;
    const ProjTile pt = tiles[blockIdx.x];
    __shared__ double sV[64][PITCH];  // [vector][k of the chunk]
    __shared__ int srow[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) srow[tid] = (tid < pt.count) ? perm[(int64_t)pt.split * n + pt.start + tid] : -1;
    __syncthreads();
    const CT* Cc = Cs + ((int64_t)pt.split * V + pt.cluster) * h;
    const double* mu = mus + ((int64_t)pt.split * V + pt.cluster) * h;
    const double* R = Rt + ((int64_t)pt.split * V + pt.cluster) * h * h;
    const int row_any = srow[0];
    const int i0 = blockIdx.y * 64 + wave * 16;  // this wave's outputs
    const int kq = lane >> 4, lc = lane & 15;
    f64x4 acc[4];
#pragma unroll
    for (int vt = 0; vt < 4; ++vt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[vt][r] = 0.0;
    constexpr int NE = 64 * KC / 256;  // staged elements per thread and chunk
    for (int kc = 0; kc < h; kc += KC) {
        CT xa[NE], ca[NE];
        double ma[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {  // loads only (a row that is not there: any row of the tile, masked below)
            const int idx = tid + e * 256;
            const int v = idx / KC, k = idx % KC;
            const int row = srow[v];
            xa[e] = X[(int64_t)(row >= 0 ? row : row_any) * D + pt.split * h + kc + k];
            ca[e] = Cc[kc + k];
            ma[e] = mu[kc + k];
        }
        double ra[CH];
        if (i0 < h) {
#pragma unroll
            for (int u = 0; u < CH; ++u) ra[u] = R[(int64_t)(kc + 4 * u + kq) * h + i0 + lc];  // A[row = i][k] = R[k][i]
        }
        __syncthreads();  // the previous chunk's operands have been read
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + e * 256;
            const int v = idx / KC, k = idx % KC;
            const CT res = xa[e] - ca[e];  // rounds in CT
            sV[v][k] = srow[v] >= 0 ? (double)res - ma[e] : 0.0;
        }
        __syncthreads();
        if (i0 < h) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
#pragma unroll
                for (int vt = 0; vt < 4; ++vt)
                    acc[vt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[u], sV[vt * 16 + lc][4 * u + kq], acc[vt], 0, 0, 0);
            }
        }
    }
    if (i0 < h) {
#pragma unroll
        for (int vt = 0; vt < 4; ++vt) {
            const int row = srow[vt * 16 + lc];
            if (row < 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(int64_t)row * D + pt.split * h + i0 + kq + 4 * r] = acc[vt][r];  // result r: output i0 + kq + 4 r
        }
    }
}

// reconstruct: x[s*h + k] = (sum_i R[c][i][k] * sx[i] + mu[c][k]) + C[c][k]   (model.py:662-669)
__global__ void k_reconstruct(const uint16_t* __restrict__ coarse, const uint8_t* __restrict__ fine,
                              const double* __restrict__ Rs, const double* __restrict__ mus,
                              const double* __restrict__ Cs64, const double* __restrict__ subs, int64_t n, int V,
                              int M, int K, int h, int w, double* __restrict__ out) {
    const int64_t item = blockIdx.x;
    const int s = blockIdx.y;
    const int nf = M / 2;
    const int c = coarse[item * 2 + s];
    const double* R = Rs + ((int64_t)s * V + c) * h * h;
    for (int k = threadIdx.x; k < h; k += blockDim.x) {
        double acc = 0.0;
        for (int i = 0; i < h; ++i) {
            const int j = i / w;
            const int f = fine[item * M + s * nf + j];
            const double sx = subs[((int64_t)(s * nf + j) * K + f) * w + (i % w)];
            acc = fma(R[(int64_t)i * h + k], sx, acc);
        }
        const double r = acc + mus[((int64_t)s * V + c) * h + k];
        out[item * (2 * h) + s * h + k] = r + Cs64[((int64_t)s * V + c) * h + k];
    }
}

// ================================================================================================
// host side
// ================================================================================================
template <typename T>
static int upload(T** dst, const T* src, size_t count) {
    CIS_CHECK_HIP(hipMalloc((void**)dst, count * sizeof(T)));
    CIS_CHECK_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return CIS_OK;
}

extern "C" void cis_model_destroy(cis_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    void* ptrs[] = {m->d_Cs32, m->d_Cs64, m->d_Rs, m->d_Rt, m->d_mus, m->d_subs, m->d_P, m->d_pmu, m->d_cnorm, m->d_flag};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    DevBuf* bufs[] = {&m->ws_xp, &m->ws_x64, &m->ws_y64, &m->ws_dist, &m->ws_proj, &m->ws_group,
                      &m->ws_in, &m->ws_out0, &m->ws_out1};
    for (DevBuf* b : bufs) b->release();
    delete m;
}

extern "C" int cis_model_create(cis_model** out, int D_in, int D, int V, int M, int K, int coarse_dtype,
                                const void* Cs, const double* Rs, const double* mus, const double* subs,
                                const double* pca_P, const double* pca_mu, int pca_mu_dtype, int renorm) {
    CIS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CIS_REQUIRE(Cs && Rs && mus && subs, "model parameters must not be NULL (fit the model first)");
    CIS_REQUIRE(D > 0 && D % 2 == 0, "D=%d must be a positive even number", D);
    CIS_REQUIRE(M >= 2 && M % 2 == 0 && D % M == 0, "M=%d must be even and divide D=%d", M, D);
    CIS_REQUIRE(V >= 1 && K >= 1, "V and K must be positive");
    CIS_REQUIRE(coarse_dtype == CIS_F32 || coarse_dtype == CIS_F64, "coarse_dtype must be 4 or 8");
    CIS_REQUIRE((pca_P == nullptr) == (pca_mu == nullptr), "pca_P and pca_mu must both be given or both NULL");
    CIS_REQUIRE(pca_P != nullptr || D_in == D, "without PCA D_in (%d) must equal D (%d)", D_in, D);
    if (K > 256 || V > 4096) {
        cis_set_error("K=%d > 256 or V=%d > 4096 is not supported by this build", K, V);
        return CIS_EUNSUPPORTED;
    }
    CIS_TRY(cis_lazy_init());
    cis_model* m = new cis_model();
    m->device = cis_current_device();
    m->D_in = D_in; m->D = D; m->V = V; m->M = M; m->K = K;
    m->h = D / 2; m->w = D / M; m->nf = M / 2;
    m->coarse_f32 = (coarse_dtype == CIS_F32);
    m->has_pca = (pca_P != nullptr);
    m->renorm = renorm != 0;
    m->pca_mu_f32 = m->has_pca && pca_mu_dtype == CIS_F32;
    int rc = CIS_OK;
    auto fail = [&](int r) { cis_model_destroy(m); return r; };
    if ((rc = cis_build_pwprog(m->h, &m->prog_h)) != CIS_OK) return fail(rc);
    if ((rc = cis_build_pwprog(m->w, &m->prog_w)) != CIS_OK) return fail(rc);
    if ((rc = cis_build_pwprog(m->D, &m->prog_D)) != CIS_OK) return fail(rc);
    const size_t nC = (size_t)2 * V * m->h;
    std::vector<double> c64(nC);
    if (m->coarse_f32) {
        const float* c = (const float*)Cs;
        for (size_t i = 0; i < nC; ++i) c64[i] = (double)c[i];
        if ((rc = upload(&m->d_Cs32, c, nC)) != CIS_OK) return fail(rc);
    } else {
        memcpy(c64.data(), Cs, nC * sizeof(double));
    }
    if ((rc = upload(&m->d_Cs64, c64.data(), nC)) != CIS_OK) return fail(rc);
    {
        std::vector<double> cn((size_t)2 * V + 2, 0.0);
        for (int s = 0; s < 2; ++s)
            for (int c = 0; c < V; ++c) {
                double a = 0.0;
                for (int i = 0; i < m->h; ++i) { const double v = c64[((size_t)s * V + c) * m->h + i]; a += v * v; }
                cn[(size_t)s * V + c] = a;
                if (a > cn[(size_t)2 * V + s]) cn[(size_t)2 * V + s] = a;
            }
        if ((rc = upload(&m->d_cnorm, cn.data(), cn.size())) != CIS_OK) return fail(rc);
        const int zero[2] = {0, 0};
        if ((rc = upload(&m->d_flag, zero, 2)) != CIS_OK) return fail(rc);
    }
    const size_t nR = (size_t)2 * V * m->h * m->h;
    if ((rc = upload(&m->d_Rs, Rs, nR)) != CIS_OK) return fail(rc);
    {
        std::vector<double> rt(nR);
        const size_t hh = (size_t)m->h * m->h;
        for (size_t c = 0; c < (size_t)2 * V; ++c)
            for (int i = 0; i < m->h; ++i)
                for (int k = 0; k < m->h; ++k) rt[c * hh + (size_t)k * m->h + i] = Rs[c * hh + (size_t)i * m->h + k];
        if ((rc = upload(&m->d_Rt, rt.data(), nR)) != CIS_OK) return fail(rc);
    }
    if ((rc = upload(&m->d_mus, mus, nC)) != CIS_OK) return fail(rc);
    if ((rc = upload(&m->d_subs, subs, (size_t)M * K * m->w)) != CIS_OK) return fail(rc);
    if (m->has_pca) {
        if ((rc = upload(&m->d_P, pca_P, (size_t)D_in * D)) != CIS_OK) return fail(rc);
        if ((rc = upload(&m->d_pmu, pca_mu, (size_t)D_in)) != CIS_OK) return fail(rc);
    }
    *out = m;
    return CIS_OK;
}


static inline int grid1(int64_t n, int bs) { return (int)ceil_div(n, bs); }

// apply_PCA of a HANDFUL of rows in one launch (round 6: the reference's callers search one feature per call, and the two launches of the
// batched form -- a 64-row MFMA tile for one row, then the finish pass -- were 14 us of a 120-170 us call).  One workgroup per row; thread c
// owns output column c.  Bit-identical to the batched form by construction: the float64 MFMA is, per output element, ONE chain of fused
// multiply-adds over ascending k from a zero accumulator (tools/probes/mfma_f64_order.hip), which is this loop; the centring follows
// k_pca_gemm_mfma_pf's stash (float32 - float32 when both are float32), the norm k_pca_finish8's order (numpy's eight interleaved
// accumulators, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))), the division and the float32 cast are the same expressions.
template <typename TX, bool SUBF32>
__global__ __launch_bounds__(128) void k_pca_small(const TX* __restrict__ X, const double* __restrict__ mu, const double* __restrict__ P,
                                                   float* __restrict__ out, int D_in /* <= 512 */, int D /* <= 128, multiple of 8 */) {
    __shared__ double s_a[512];
    __shared__ double s_y[128];
    __shared__ double s_nrm;
    const int tid = threadIdx.x;
    const TX* x = X + (int64_t)blockIdx.x * D_in;
    for (int k = tid; k < D_in; k += 128) {
        if constexpr (sizeof(TX) == 4) {
            if constexpr (SUBF32) s_a[k] = (double)((float)x[k] - (float)mu[k]);
            else s_a[k] = (double)x[k] - mu[k];
        } else {
            s_a[k] = (double)x[k] - mu[k];
        }
    }
    __syncthreads();
    if (tid < D) {
        double acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < D_in; ++k) acc = fma(s_a[k], P[(int64_t)k * D + tid], acc);
        s_y[tid] = acc;
    }
    __syncthreads();
    if (tid < 64) {   // (every 8-lane group of the wave computes the same norm; lane 0 publishes it)
        const int j = tid & 7;
        double acc = s_y[j] * s_y[j];
        for (int i = 1; i * 8 < D; ++i) acc = acc + s_y[i * 8 + j] * s_y[i * 8 + j];
        acc = acc + __shfl_xor(acc, 1);
        acc = acc + __shfl_xor(acc, 2);
        acc = acc + __shfl_xor(acc, 4);
        if (tid == 0) s_nrm = sqrt(acc);
    }
    __syncthreads();
    if (tid < D) out[(int64_t)blockIdx.x * D + tid] = (float)(s_y[tid] / s_nrm);
}

int cis_dev_apply_pca(cis_model* m, const void* dX, int x_dtype, int64_t n, float* d_out, hipStream_t st, DevBuf* ws_y) {
    CIS_REQUIRE(m->has_pca, "model has no PCA parameters");
    if (n == 0) return CIS_OK;
    static const bool no_small = getenv("CIS_PCA_NO_SMALL") != nullptr;   // A/B runs and tests: the batched form for every n
    if (!no_small && n <= 8 && m->renorm && m->D >= 8 && m->D <= 128 && m->D % 8 == 0 && m->D_in <= 512) {
        if (x_dtype == CIS_F32 && m->pca_mu_f32)
            hipLaunchKernelGGL((k_pca_small<float, true>), dim3((unsigned)n), dim3(128), 0, st, (const float*)dX, m->d_pmu, m->d_P, d_out, m->D_in, m->D);
        else if (x_dtype == CIS_F32)
            hipLaunchKernelGGL((k_pca_small<float, false>), dim3((unsigned)n), dim3(128), 0, st, (const float*)dX, m->d_pmu, m->d_P, d_out, m->D_in, m->D);
        else
            hipLaunchKernelGGL((k_pca_small<double, false>), dim3((unsigned)n), dim3(128), 0, st, (const double*)dX, m->d_pmu, m->d_P, d_out, m->D_in, m->D);
        CIS_CHECK_HIP(hipGetLastError());
        return CIS_OK;
    }
    DevBuf* wy = ws_y ? ws_y : &m->ws_y64;
    CIS_TRY(wy->reserve((size_t)n * m->D * sizeof(double)));
    double* Y = wy->as<double>();
    // 64-row tiles when they fill the chip twice over, 32-row tiles otherwise
    const bool small = ceil_div(n, 64) * ceil_div(m->D, 64) < 512;
    dim3 g((unsigned)ceil_div(n, small ? 32 : 64), (unsigned)ceil_div(m->D, 64));
    // wide inputs (the 4096-d DeepSentibank features): the float64 matrix cores; CIS_PCA_GEMM=valu keeps the register-tiled kernel
    const char* pca_env = getenv("CIS_PCA_GEMM");
    const bool pca_valu = pca_env && !strcmp(pca_env, "valu");
    const bool pca_force = pca_env && !strcmp(pca_env, "mfma");
    const bool use_mfma = !pca_valu && (m->D_in >= 128 || pca_force);
    dim3 gm((unsigned)ceil_div(n, 64), (unsigned)ceil_div(m->D, 64));
    // K 16 per stage = four workgroups per CU: pays once the grid is more than one round of the 32-per-stage form (measured on
    // the 4096 -> 256 product: 12500 rows 7.5 -> 8.3 M vectors/s of encode, 8192 rows 0.61 -> 0.64 ms)
    const bool bk16 = getenv("CIS_PCA_BK") ? atoi(getenv("CIS_PCA_BK")) == 16 : (int64_t)gm.x * gm.y > 640;
    // 128 x 128 tiles once they fill the chip twice over (CIS_PCA_TILE=64 keeps the 64 x 64 form)
    const dim3 gm128((unsigned)ceil_div(n, 128), (unsigned)ceil_div(m->D, 128));
    const int tile_env = getenv("CIS_PCA_TILE") ? atoi(getenv("CIS_PCA_TILE")) : 0;
    const bool big_tiles = use_mfma && m->D >= 128 && tile_env != 64 && ((int64_t)gm128.x * gm128.y >= 512 || tile_env == 128);
    // the loads-only fetch (k_pca_gemm_mfma_pf) where whole quads of k and pairs of columns exist; CIS_PCA_PF=0: the forms before it
    const bool pf = m->D_in % 4 == 0 && m->D_in >= 4 && m->D % 2 == 0 && m->D >= 2 && !(getenv("CIS_PCA_PF") && atoi(getenv("CIS_PCA_PF")) == 0);
#define CIS_PCA_LAUNCH(TX, SUB, XP)                                                                                              \
    do {                                                                                                                          \
        if (big_tiles && pf) hipLaunchKernelGGL((k_pca_gemm_mfma_pf<TX, SUB, 128, 16>), gm128, dim3(256), 0, st, XP, m->d_pmu, m->d_P, Y, n, m->D_in, m->D); \
        else if (use_mfma && bk16 && pf) hipLaunchKernelGGL((k_pca_gemm_mfma_pf<TX, SUB, 64, 16>), gm, dim3(256), 0, st, XP, m->d_pmu, m->d_P, Y, n, m->D_in, m->D); \
        else if (use_mfma && pf) hipLaunchKernelGGL((k_pca_gemm_mfma_pf<TX, SUB, 64, 32>), gm, dim3(256), 0, st, XP, m->d_pmu, m->d_P, Y, n, m->D_in, m->D); \
        else if (big_tiles) hipLaunchKernelGGL((k_pca_gemm_mfma128<TX, SUB>), gm128, dim3(256), 0, st, XP, m->d_pmu, m->d_P, Y, n, m->D_in, m->D); \
        else if (use_mfma && bk16) hipLaunchKernelGGL((k_pca_gemm_mfma<TX, SUB, 16>), gm, dim3(256), 0, st, XP, m->d_pmu, m->d_P, Y, n, m->D_in, m->D); \
        else if (use_mfma) hipLaunchKernelGGL((k_pca_gemm_mfma<TX, SUB, 32>), gm, dim3(256), 0, st, XP, m->d_pmu, m->d_P, Y, n, m->D_in, m->D); \
        else if (small) hipLaunchKernelGGL((k_pca_gemm<TX, SUB, 2>), g, dim3(256), 0, st, XP, m->d_pmu, m->d_P, Y, n, m->D_in, m->D); \
        else hipLaunchKernelGGL((k_pca_gemm<TX, SUB, 4>), g, dim3(256), 0, st, XP, m->d_pmu, m->d_P, Y, n, m->D_in, m->D);       \
    } while (0)
    if (x_dtype == CIS_F32 && m->pca_mu_f32) CIS_PCA_LAUNCH(float, true, (const float*)dX);
    else if (x_dtype == CIS_F32) CIS_PCA_LAUNCH(float, false, (const float*)dX);
    else CIS_PCA_LAUNCH(double, false, (const double*)dX);
#undef CIS_PCA_LAUNCH
    if (m->renorm && m->D >= 8 && m->D <= 128 && m->D % 8 == 0)
        hipLaunchKernelGGL(k_pca_finish8, dim3((unsigned)ceil_div(n, 8)), dim3(64), 0, st, Y, d_out, n, m->D);
    else
        hipLaunchKernelGGL(k_pca_finish, dim3(grid1(n, 64)), dim3(64), 0, st, Y, d_out, n, m->D, m->renorm ? 1 : 0, m->prog_D);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

int cis_dev_coarse_type(cis_model* m, const void* d_xp, int xp_dtype, int64_t n, const void** xc, int* ct,
                        hipStream_t st, DevBuf* ws_x) {
    if (xp_dtype == CIS_F32 && m->coarse_f32) {
        *xc = d_xp;
        *ct = CIS_F32;
        return CIS_OK;
    }
    *ct = CIS_F64;
    if (xp_dtype == CIS_F64) {
        *xc = d_xp;
        return CIS_OK;
    }
    DevBuf* wx = ws_x ? ws_x : &m->ws_x64;
    CIS_TRY(wx->reserve((size_t)n * m->D * sizeof(double)));
    if (n > 0)
        hipLaunchKernelGGL(k_to_f64<float>, dim3(2048), dim3(256), 0, st, (const float*)d_xp, wx->as<double>(),
                           n * m->D);
    *xc = wx->p;
    return CIS_OK;
}

// both coarse halves in one launch: blockIdx.z = split; out = [2][n][V]
template <typename T>
__global__ __launch_bounds__(256) void k_sqdist_rows2(const T* __restrict__ X, int64_t ldx, int h, const T* __restrict__ C,
                                                      int64_t n, int ncent, T* __restrict__ out, PwProg prog) {
    const int s = blockIdx.z;
    const int c = blockIdx.y * 16 + (threadIdx.x % 16);
    const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x / 16);
    if (r >= n || c >= ncent) return;
    const T* x = X + r * ldx + s * h;
    const T* cc = C + ((int64_t)s * ncent + c) * h;
    auto elem = [&](int i) -> T { const T df = x[i] - cc[i]; return df * df; };
    out[((int64_t)s * n + r) * ncent + c] = pw_sum<T>(prog, elem);
}

// The same for many centroids (V >= 64, h = 32 or 64: one summation leaf): a thread keeps its row in registers, a
// workgroup of 256 rows walks 64 centroids staged in LDS (every lane reads the same address: broadcast), four results
// leave as one 16-byte store.  The element order is pw_leaf's, so the values are the ones of the kernel above -- which
// reads both operands through strided 4-byte global loads (1.2 ms per 8192 queries at V = 2048).
template <typename T, int H>
__global__ __launch_bounds__(256) void k_sqdist_rows2_reg(const T* __restrict__ X, int64_t ldx, const T* __restrict__ C,
                                                          int64_t n, int ncent, T* __restrict__ out) {
    constexpr int CT = 64;
    __shared__ __align__(16) T sC[CT][H];
    const int s = blockIdx.z;
    const int c0 = blockIdx.y * CT;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nc = (ncent - c0 < CT) ? (ncent - c0) : CT;
    {
        const T* src = C + ((int64_t)s * ncent + c0) * H;
        for (int e = threadIdx.x; e < nc * H; e += 256) (&sC[0][0])[e] = src[e];
    }
    T x[H];
    {
        const T* xr = X + (r < n ? r : 0) * ldx + s * H;
#pragma unroll
        for (int i = 0; i < H; ++i) x[i] = xr[i];
    }
    __syncthreads();
    if (r >= n) return;
    T* o = out + ((int64_t)s * n + r) * ncent + c0;
    const bool vec = (ncent % 4 == 0) && sizeof(T) == 4;
    for (int cq = 0; cq < nc; cq += 4) {
        T v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (cq + k < nc) ? cq + k : nc - 1;
            const T* cc = sC[c];
            auto elem = [&](int i) -> T { const T df = x[i] - cc[i]; return df * df; };
            v[k] = pw_leaf<T>(elem, 0, H);
        }
        if (vec && cq + 3 < nc) {
            *reinterpret_cast<float4*>(o + cq) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cq + k < nc) o[cq + k] = v[k];
        }
    }
}

int cis_launch_sqdist_both(cis_model* m, const void* xc, int ct, int64_t n, void* out, hipStream_t st) {
    if (n == 0) return CIS_OK;
    // (the row must fit the register file next to eight accumulators: float32 up to h = 64, float64 up to h = 32)
    if (m->V >= 64 && ((ct == CIS_F32 && (m->h == 32 || m->h == 64)) || (ct != CIS_F32 && m->h == 32)) && !getenv("CIS_SQDIST_PLAIN")) {
        const dim3 gr((unsigned)ceil_div(n, 256), (unsigned)ceil_div(m->V, 64), 2);
#define CIS_SQ_REG(T, H, CP) hipLaunchKernelGGL((k_sqdist_rows2_reg<T, H>), gr, dim3(256), 0, st, (const T*)xc, (int64_t)m->D, CP, n, m->V, (T*)out)
        if (ct == CIS_F32) {
            if (m->h == 32) CIS_SQ_REG(float, 32, m->d_Cs32); else CIS_SQ_REG(float, 64, m->d_Cs32);
        } else {
            CIS_SQ_REG(double, 32, m->d_Cs64);
        }
#undef CIS_SQ_REG
        CIS_CHECK_HIP(hipGetLastError());
        return CIS_OK;
    }
    dim3 g((unsigned)ceil_div(n, 16), (unsigned)ceil_div(m->V, 16), 2);
    if (ct == CIS_F32)
        hipLaunchKernelGGL(k_sqdist_rows2<float>, g, dim3(256), 0, st, (const float*)xc, (int64_t)m->D, m->h, m->d_Cs32, n, m->V,
                           (float*)out, m->prog_h);
    else
        hipLaunchKernelGGL(k_sqdist_rows2<double>, g, dim3(256), 0, st, (const double*)xc, (int64_t)m->D, m->h, m->d_Cs64, n, m->V,
                           (double*)out, m->prog_h);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

int cis_launch_sqdist(cis_model* m, const void* xc, int ct, int64_t n, int split, void* out, hipStream_t st) {
    if (n == 0) return CIS_OK;
    dim3 g((unsigned)ceil_div(n, 16), (unsigned)ceil_div(m->V, 16));
    if (ct == CIS_F32)
        hipLaunchKernelGGL(k_sqdist_rows<float>, g, dim3(256), 0, st, (const float*)xc, (int64_t)m->D, split * m->h,
                           m->d_Cs32 + (size_t)split * m->V * m->h, n, m->V, m->h, (float*)out, m->prog_h, (const int*)nullptr);
    else
        hipLaunchKernelGGL(k_sqdist_rows<double>, g, dim3(256), 0, st, (const double*)xc, (int64_t)m->D, split * m->h,
                           m->d_Cs64 + (size_t)split * m->V * m->h, n, m->V, m->h, (double*)out, m->prog_h, (const int*)nullptr);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

int cis_launch_sqdist_generic(const void* X, int ct, int64_t ldx, int xoff, const void* C, int64_t n, int ncent, int d,
                              void* out, hipStream_t st) {
    if (n == 0) return CIS_OK;
    PwProg prog;
    CIS_TRY(cis_build_pwprog(d, &prog));
    dim3 g((unsigned)ceil_div(n, 16), (unsigned)ceil_div(ncent, 16));
    if (ct == CIS_F32)
        hipLaunchKernelGGL(k_sqdist_rows<float>, g, dim3(256), 0, st, (const float*)X, ldx, xoff, (const float*)C, n, ncent, d,
                           (float*)out, prog, (const int*)nullptr);
    else
        hipLaunchKernelGGL(k_sqdist_rows<double>, g, dim3(256), 0, st, (const double*)X, ldx, xoff, (const double*)C, n, ncent,
                           d, (double*)out, prog, (const int*)nullptr);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

// one temporary device array: uploads `n` elements of src_dtype widened to the compute type ct
static int upload_as(const void* src, int src_dtype, int ct, size_t n, DevBuf* buf) {
    CIS_TRY(buf->reserve(n * (size_t)ct + 16));
    if (src_dtype == ct) {
        CIS_CHECK_HIP(hipMemcpy(buf->p, src, n * (size_t)ct, hipMemcpyHostToDevice));
    } else {  // float32 -> float64 is exact
        std::vector<double> tmp(n);
        const float* f = (const float*)src;
        for (size_t i = 0; i < n; ++i) tmp[i] = (double)f[i];
        CIS_CHECK_HIP(hipMemcpy(buf->p, tmp.data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    return CIS_OK;
}

extern "C" int cis_predict_cluster(const void* X, int x_dtype, const void* C, int c_dtype, int64_t n, int ncent, int d,
                                   uint32_t* out) {
    CIS_REQUIRE((x_dtype == CIS_F32 || x_dtype == CIS_F64) && (c_dtype == CIS_F32 || c_dtype == CIS_F64), "dtype must be 4 or 8");
    CIS_REQUIRE(n >= 0 && ncent >= 1 && d >= 1 && (n == 0 || (X && C && out)), "bad arguments");
    if (n == 0) return CIS_OK;
    CIS_TRY(cis_lazy_init());
    const int ct = (x_dtype == CIS_F32 && c_dtype == CIS_F32) ? CIS_F32 : CIS_F64;  // numpy promotion
    DevBuf bx, bc, bd, bo;
    int rc = CIS_OK;
    auto done = [&](int r) { bx.release(); bc.release(); bd.release(); bo.release(); return r; };
    if ((rc = upload_as(X, x_dtype, ct, (size_t)n * d, &bx)) != CIS_OK) return done(rc);
    if ((rc = upload_as(C, c_dtype, ct, (size_t)ncent * d, &bc)) != CIS_OK) return done(rc);
    if ((rc = bd.reserve((size_t)n * ncent * ct)) != CIS_OK) return done(rc);
    if ((rc = bo.reserve((size_t)n * sizeof(uint32_t))) != CIS_OK) return done(rc);
    if ((rc = cis_launch_sqdist_generic(bx.p, ct, d, 0, bc.p, n, ncent, d, bd.p, nullptr)) != CIS_OK) return done(rc);
    if (ct == CIS_F32)
        hipLaunchKernelGGL((k_argmin_rows<float, uint32_t>), dim3(grid1(n, 256)), dim3(256), 0, nullptr, bd.as<float>(), n, ncent,
                           bo.as<uint32_t>(), 1, 0, (const int*)nullptr);
    else
        hipLaunchKernelGGL((k_argmin_rows<double, uint32_t>), dim3(grid1(n, 256)), dim3(256), 0, nullptr, bd.as<double>(), n,
                           ncent, bo.as<uint32_t>(), 1, 0, (const int*)nullptr);
    hipError_t e = hipMemcpy(out, bo.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { cis_set_error("hipMemcpy failed: %s", hipGetErrorString(e)); return done(CIS_EHIP); }
    return done(CIS_OK);
}

// Coarse assignment for a few clusters (V < 256: no matrix-core prefilter): a thread keeps its row half in registers, walks
// the V centroids of the split staged in LDS (broadcast reads) and keeps the first minimum -- distances in numpy's order
// (pw_leaf: h <= 128 is one summation leaf) and compute type, exactly as k_sqdist_rows + k_argmin_rows produce them, without
// the n x V distance matrix and the second launch (2 x 38 + 2 x 5 us per 65536 vectors at V = 16 -> one launch).
template <typename T, int H>
__global__ __launch_bounds__(256) void k_coarse_assign_reg(const T* __restrict__ X, int64_t ldx, const T* __restrict__ C,
                                                           int64_t n, int ncent, uint16_t* __restrict__ out /* [n][2] */) {
    constexpr int CT = 64;
    __shared__ __align__(16) T sC[CT][H];
    const int s = blockIdx.y;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    T x[H];
    {
        const T* xr = X + (r < n ? r : 0) * ldx + s * H;
#pragma unroll
        for (int i = 0; i < H; ++i) x[i] = xr[i];
    }
    T best = (T)0;
    int bi = -1;
    for (int c0 = 0; c0 < ncent; c0 += CT) {
        const int nc = (ncent - c0 < CT) ? (ncent - c0) : CT;
        __syncthreads();
        const T* src = C + ((int64_t)s * ncent + c0) * H;
        for (int e = threadIdx.x; e < nc * H; e += 256) (&sC[0][0])[e] = src[e];
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
            const T* cc = sC[c];
            auto elem = [&](int i) -> T { const T df = x[i] - cc[i]; return df * df; };
            const T v = pw_leaf<T>(elem, 0, H);
            if (bi < 0 || v < best) { best = v; bi = c0 + c; }
        }
    }
    if (r < n) out[r * 2 + s] = (uint16_t)bi;
}

// coarse ids of n LOPQ-space vectors (already in compute type ct) -> d_coarse [n][2]
static int dev_predict_coarse(cis_model* m, const void* xc, int ct, int64_t n, uint16_t* d_coarse, hipStream_t st) {
    if (n == 0) return CIS_OK;
    CIS_TRY(m->ws_dist.reserve((size_t)n * (m->V > m->K ? m->V : m->K) * sizeof(double)));
    // many clusters: matrix-core prefilter + exact re-check of the few listed pairs; the exact kernels below then only run
    // (device-side predicate) when a candidate list overflowed
    const int coarse_mode = getenv("CIS_COARSE") ? atoi(getenv("CIS_COARSE")) : -1;  // 0 exact kernels, 1 prefilter
    const bool pre = (coarse_mode >= 1 || (coarse_mode != 0 && m->V >= 256)) && m->h <= 128 && m->V <= 65535;  // 2: float64 cores for float32 too
    const int* only_if = nullptr;
    if (!pre && m->V <= 1024 && ((ct == CIS_F32 && (m->h == 32 || m->h == 64)) || (ct != CIS_F32 && m->h == 32)) && !getenv("CIS_SQDIST_PLAIN")) {
        const dim3 gr((unsigned)ceil_div(n, 256), 2);
        if (ct == CIS_F32) {
            if (m->h == 32) hipLaunchKernelGGL((k_coarse_assign_reg<float, 32>), gr, dim3(256), 0, st, (const float*)xc, (int64_t)m->D, m->d_Cs32, n, m->V, d_coarse);
            else hipLaunchKernelGGL((k_coarse_assign_reg<float, 64>), gr, dim3(256), 0, st, (const float*)xc, (int64_t)m->D, m->d_Cs32, n, m->V, d_coarse);
        } else {
            hipLaunchKernelGGL((k_coarse_assign_reg<double, 32>), gr, dim3(256), 0, st, (const double*)xc, (int64_t)m->D, m->d_Cs64, n, m->V, d_coarse);
        }
        CIS_CHECK_HIP(hipGetLastError());
        return CIS_OK;
    }
    if (pre) {
        CIS_CHECK_HIP(hipMemsetAsync(m->d_flag, 0, sizeof(int), st));
        const dim3 gp((unsigned)ceil_div(n, 64), 2);
        // |dt + |x|^2 - numpy's value| <= (h + 6) u (|x| + |c|)^2 for the exact sum in the compute type (u = 2^-24 / 2^-53) plus
        // (h + 4) 2^-53 (|x|^2 + |c|^2) for the float64 matrix-core form; (|x| + |c|)^2 <= 2 (|x|^2 + |c|^2); margin x4
        const double u = (ct == CIS_F32) ? ldexp(1.0, -24) : ldexp(1.0, -53);
        const double eps_rel = 4.0 * (2.0 * (m->h + 6) * u + (m->h + 4) * ldexp(1.0, -53));
#define CIS_COARSE_LAUNCH(T, KS, C)                                                                                          \
    hipLaunchKernelGGL((k_coarse_mfma<T, KS>), gp, dim3(256), 0, st, (const T*)xc, (int64_t)m->D, m->h, C, m->d_cnorm, n, m->V, \
                       d_coarse, m->prog_h, eps_rel, m->d_flag)
        if (ct == CIS_F32 && coarse_mode != 2) {
            // float32 matrix cores; the slack adds the float32 product's own error, (h + 4) 2^-24 (|x|^2 + |c|^2)
            const double eps32 = 4.0 * (2.0 * (m->h + 6) + (m->h + 4)) * ldexp(1.0, -24);
            const dim3 gp32((unsigned)ceil_div(n, 128), 2);
            if (m->h <= 64)
                hipLaunchKernelGGL((k_coarse_mfma32<32>), gp32, dim3(256), 0, st, (const float*)xc, (int64_t)m->D, m->h, m->d_Cs32, m->d_cnorm, n,
                                   m->V, d_coarse, m->prog_h, eps32, m->d_flag);
            else
                hipLaunchKernelGGL((k_coarse_mfma32<64>), gp32, dim3(256), 0, st, (const float*)xc, (int64_t)m->D, m->h, m->d_Cs32, m->d_cnorm, n,
                                   m->V, d_coarse, m->prog_h, eps32, m->d_flag);
        } else if (ct == CIS_F32) {
            if (m->h <= 32) CIS_COARSE_LAUNCH(float, 8, m->d_Cs32);
            else if (m->h <= 64) CIS_COARSE_LAUNCH(float, 16, m->d_Cs32);
            else CIS_COARSE_LAUNCH(float, 32, m->d_Cs32);
        } else {
            if (m->h <= 32) CIS_COARSE_LAUNCH(double, 8, m->d_Cs64);
            else if (m->h <= 64) CIS_COARSE_LAUNCH(double, 16, m->d_Cs64);
            else CIS_COARSE_LAUNCH(double, 32, m->d_Cs64);
        }
#undef CIS_COARSE_LAUNCH
        only_if = m->d_flag;
    }
    dim3 g((unsigned)ceil_div(n, 16), (unsigned)ceil_div(m->V, 16));
    if (pre) {
        if (g.x > 512) g.x = 512;
        if (g.y > 8) g.y = 8;
    }
    for (int s = 0; s < 2; ++s) {
        if (ct == CIS_F32) {
            float* dist = m->ws_dist.as<float>();
            hipLaunchKernelGGL(k_sqdist_rows<float>, g, dim3(256), 0, st, (const float*)xc, (int64_t)m->D, s * m->h,
                               m->d_Cs32 + (size_t)s * m->V * m->h, n, m->V, m->h, dist, m->prog_h, only_if);
            hipLaunchKernelGGL((k_argmin_rows<float, uint16_t>), dim3(grid1(n, 256)), dim3(256), 0, st, dist, n, m->V,
                               d_coarse, 2, s, only_if);
        } else {
            double* dist = m->ws_dist.as<double>();
            hipLaunchKernelGGL(k_sqdist_rows<double>, g, dim3(256), 0, st, (const double*)xc, (int64_t)m->D, s * m->h,
                               m->d_Cs64 + (size_t)s * m->V * m->h, n, m->V, m->h, dist, m->prog_h, only_if);
            hipLaunchKernelGGL((k_argmin_rows<double, uint16_t>), dim3(grid1(n, 256)), dim3(256), 0, st, dist, n, m->V,
                               d_coarse, 2, s, only_if);
        }
    }
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

// local projection of n vectors given their coarse ids -> d_proj [n][D] float64
static int dev_project(cis_model* m, const void* xc, int ct, int64_t n, const uint16_t* d_coarse, double* d_proj,
                       hipStream_t st) {
    if (n == 0) return CIS_OK;
    CIS_REQUIRE(n < (int64_t)1 << 30, "batch too large for one projection pass");
    const int V = m->V;
    const int64_t max_tiles = 2 * (ceil_div(n, 64) + V);
    // layout of ws_group: counts[2V] offsets[2V] cursor[2V] n_tiles[1] pad | perm[2n] | tiles[max_tiles]
    const size_t ints = (size_t)6 * V + 4 + (size_t)2 * n;
    CIS_TRY(m->ws_group.reserve(ints * sizeof(int) + (size_t)max_tiles * sizeof(ProjTile) + 64));
    int* counts = m->ws_group.as<int>();
    int* offsets = counts + 2 * V;
    int* cursor = offsets + 2 * V;
    int* n_tiles = cursor + 2 * V;
    int* perm = n_tiles + 4;
    ProjTile* tiles = reinterpret_cast<ProjTile*>(perm + 2 * n);
    CIS_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)2 * V * sizeof(int), st));
    const size_t bins_lds = V <= 1024 ? (size_t)4 * V * sizeof(int) : 0;
    hipLaunchKernelGGL(k_group_hist, dim3(grid1(n, 256)), dim3(256), bins_lds, st, d_coarse, n, V, counts);
    hipLaunchKernelGGL(k_group_scan, dim3(1), dim3(256), (size_t)2 * V * sizeof(int), st, counts, V, offsets, cursor, tiles,
                       n_tiles, 64);
    hipLaunchKernelGGL(k_group_scatter, dim3(grid1(n, 256)), dim3(256), bins_lds, st, d_coarse, n, V, offsets, cursor, perm);
    dim3 g((unsigned)max_tiles, (unsigned)ceil_div(m->h, 64));
    // the tile product on the float64 matrix cores where h is a multiple of 16 (CIS_PROJECT_VALU=1: the vector form)
    const bool pm = m->h % 16 == 0 && !getenv("CIS_PROJECT_VALU");
    const bool pm64 = pm && m->h % 64 == 0;
#define CIS_PROJ(KERN, TT, XP, CP) hipLaunchKernelGGL(KERN, g, dim3(256), 0, st, (const TT*)XP, CP, m->d_Rt, m->d_mus, tiles, n_tiles, perm, n, V, m->h, m->D, d_proj)
    if (ct == CIS_F32) {
        if (pm64) CIS_PROJ((k_project_tiles_mfma<float, 16>), float, xc, m->d_Cs32);
        else if (pm) CIS_PROJ((k_project_tiles_mfma<float, 4>), float, xc, m->d_Cs32);
        else CIS_PROJ(k_project_tiles<float>, float, xc, m->d_Cs32);
    } else {
        if (pm64) CIS_PROJ((k_project_tiles_mfma<double, 16>), double, xc, m->d_Cs64);
        else if (pm) CIS_PROJ((k_project_tiles_mfma<double, 4>), double, xc, m->d_Cs64);
        else CIS_PROJ(k_project_tiles<double>, double, xc, m->d_Cs64);
    }
#undef CIS_PROJ
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

static int dev_fine_from_proj(cis_model* m, const double* d_proj, int64_t n, uint8_t* d_fine, hipStream_t st) {
    if (n == 0) return CIS_OK;
    // CIS_FINE=0: every (vector, sub-centroid) pair exactly on the VALU (k_fine_codes, rounds 1-2); default: matrix-core
    // prefilter + exact re-check of the listed pairs
    const bool fine_mfma = !(getenv("CIS_FINE") && atoi(getenv("CIS_FINE")) == 0);
    if (fine_mfma && m->K <= 256 && (m->w == 4 || m->w == 8 || m->w == 16 || m->w == 32) && !getenv("CIS_ENCODE_TWO_KERNEL")) {
        // groups of 128 vectors per workgroup: as many as keep >= ~4 workgroups per CU in the launch
        int reps = (int)(ceil_div(n, 128) * m->M / 1024);
        reps = reps < 1 ? 1 : (reps > 8 ? 8 : reps);
        const dim3 gf((unsigned)ceil_div(n, (int64_t)128 * reps), (unsigned)m->M);
        const int KP = (m->K + 31) & ~31;
        const size_t lds = (size_t)KP * (m->w + 3) * 4 + 4 * 512 * 4 + 4 * 32 * 8 + 2 * 4 * 32 * 4 + 16;
        switch (m->w) {
            case 4: hipLaunchKernelGGL(k_fine_mfma<4>, gf, dim3(256), lds, st, d_proj, m->D, m->d_subs, n, m->K, m->M, d_fine, reps); break;
            case 8: hipLaunchKernelGGL(k_fine_mfma<8>, gf, dim3(256), lds, st, d_proj, m->D, m->d_subs, n, m->K, m->M, d_fine, reps); break;
            case 16: hipLaunchKernelGGL(k_fine_mfma<16>, gf, dim3(256), lds, st, d_proj, m->D, m->d_subs, n, m->K, m->M, d_fine, reps); break;
            default: hipLaunchKernelGGL(k_fine_mfma<32>, gf, dim3(256), lds, st, d_proj, m->D, m->d_subs, n, m->K, m->M, d_fine, reps); break;
        }
        CIS_CHECK_HIP(hipGetLastError());
        return CIS_OK;
    }
    if (m->K <= 256 && (m->w == 4 || m->w == 8 || m->w == 16 || m->w == 32) && !getenv("CIS_ENCODE_TWO_KERNEL")) {
        const dim3 gf((unsigned)ceil_div(n, 256), (unsigned)m->M);
        const size_t lds = (size_t)m->K * m->w * sizeof(double);
        switch (m->w) {
            case 4: hipLaunchKernelGGL(k_fine_codes<4>, gf, dim3(256), lds, st, d_proj, m->D, m->d_subs, n, m->K, m->M, d_fine); break;
            case 8: hipLaunchKernelGGL(k_fine_codes<8>, gf, dim3(256), lds, st, d_proj, m->D, m->d_subs, n, m->K, m->M, d_fine); break;
            case 16: hipLaunchKernelGGL(k_fine_codes<16>, gf, dim3(256), lds, st, d_proj, m->D, m->d_subs, n, m->K, m->M, d_fine); break;
            default: hipLaunchKernelGGL(k_fine_codes<32>, gf, dim3(256), lds, st, d_proj, m->D, m->d_subs, n, m->K, m->M, d_fine); break;
        }
        CIS_CHECK_HIP(hipGetLastError());
        return CIS_OK;
    }
    CIS_TRY(m->ws_dist.reserve((size_t)n * (m->V > m->K ? m->V : m->K) * sizeof(double)));
    double* dist = m->ws_dist.as<double>();
    dim3 g((unsigned)ceil_div(n, 16), (unsigned)ceil_div(m->K, 16));
    for (int j = 0; j < m->M; ++j) {
        hipLaunchKernelGGL(k_sqdist_rows<double>, g, dim3(256), 0, st, d_proj, (int64_t)m->D, j * m->w,
                           m->d_subs + (size_t)j * m->K * m->w, n, m->K, m->w, dist, m->prog_w, (const int*)nullptr);
        hipLaunchKernelGGL((k_argmin_rows<double, uint8_t>), dim3(grid1(n, 256)), dim3(256), 0, st, dist, n, m->K,
                           d_fine, m->M, j, (const int*)nullptr);
    }
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

static const int64_t ENCODE_CHUNK = 1 << 16;

extern "C" int cis_encode_dev(cis_model* m, const void* dX, int x_dtype, int64_t n, uint16_t* d_coarse,
                              uint8_t* d_fine, void* stream) {
    CIS_REQUIRE(m != nullptr, "model is NULL");
    CIS_REQUIRE(x_dtype == CIS_F32 || x_dtype == CIS_F64, "x_dtype must be 4 or 8");
    CIS_REQUIRE(n >= 0, "n must be >= 0");
    CIS_CHECK_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    for (int64_t a = 0; a < n; a += ENCODE_CHUNK) {
        const int64_t cn = (n - a < ENCODE_CHUNK) ? (n - a) : ENCODE_CHUNK;
        const char* xin = (const char*)dX + (size_t)a * m->D_in * x_dtype;
        const void* xp = xin;
        int xp_dtype = x_dtype;
        if (m->has_pca) {
            CIS_TRY(m->ws_xp.reserve((size_t)cn * m->D * sizeof(float)));
            CIS_TRY(cis_dev_apply_pca(m, xin, x_dtype, cn, m->ws_xp.as<float>(), st));
            xp = m->ws_xp.p;
            xp_dtype = CIS_F32;
        }
        const void* xc;
        int ct;
        CIS_TRY(cis_dev_coarse_type(m, xp, xp_dtype, cn, &xc, &ct, st));
        CIS_TRY(dev_predict_coarse(m, xc, ct, cn, d_coarse + a * 2, st));
        CIS_TRY(m->ws_proj.reserve((size_t)cn * m->D * sizeof(double)));
        CIS_TRY(dev_project(m, xc, ct, cn, d_coarse + a * 2, m->ws_proj.as<double>(), st));
        CIS_TRY(dev_fine_from_proj(m, m->ws_proj.as<double>(), cn, d_fine + a * m->M, st));
    }
    return CIS_OK;
}

// ---- host-pointer wrappers ---------------------------------------------------------------------
struct HostIO {
    cis_model* m;
    hipStream_t st = nullptr;
    int in(const void* src, size_t bytes, void** dptr) {
        CIS_TRY(m->ws_in.reserve(bytes ? bytes : 1));
        if (bytes) CIS_CHECK_HIP(hipMemcpyAsync(m->ws_in.p, src, bytes, hipMemcpyHostToDevice, st));
        *dptr = m->ws_in.p;
        return CIS_OK;
    }
};

#define CIS_ENTER(m)                                   \
    CIS_REQUIRE((m) != nullptr, "model is NULL");      \
    CIS_CHECK_HIP(hipSetDevice((m)->device));

extern "C" int cis_encode(cis_model* m, const void* X, int x_dtype, int64_t n, uint16_t* coarse, uint8_t* fine) {
    CIS_ENTER(m);
    CIS_REQUIRE(x_dtype == CIS_F32 || x_dtype == CIS_F64, "x_dtype must be 4 or 8");
    CIS_REQUIRE(n >= 0 && (n == 0 || (X && coarse && fine)), "NULL buffer");
    if (n == 0) return CIS_OK;
    HostIO io{m};
    void* dX;
    CIS_TRY(io.in(X, (size_t)n * m->D_in * x_dtype, &dX));
    CIS_TRY(m->ws_out0.reserve((size_t)n * 2 * sizeof(uint16_t)));
    CIS_TRY(m->ws_out1.reserve((size_t)n * m->M));
    CIS_TRY(cis_encode_dev(m, dX, x_dtype, n, m->ws_out0.as<uint16_t>(), m->ws_out1.as<uint8_t>(), nullptr));
    CIS_CHECK_HIP(hipMemcpy(coarse, m->ws_out0.p, (size_t)n * 2 * sizeof(uint16_t), hipMemcpyDeviceToHost));
    CIS_CHECK_HIP(hipMemcpy(fine, m->ws_out1.p, (size_t)n * m->M, hipMemcpyDeviceToHost));
    return CIS_OK;
}

extern "C" int cis_apply_pca(cis_model* m, const void* X, int x_dtype, int64_t n, float* out) {
    CIS_ENTER(m);
    CIS_REQUIRE(x_dtype == CIS_F32 || x_dtype == CIS_F64, "x_dtype must be 4 or 8");
    CIS_REQUIRE(m->has_pca, "model has no PCA parameters");
    if (n == 0) return CIS_OK;
    HostIO io{m};
    void* dX;
    CIS_TRY(io.in(X, (size_t)n * m->D_in * x_dtype, &dX));
    CIS_TRY(m->ws_xp.reserve((size_t)n * m->D * sizeof(float)));
    CIS_TRY(cis_dev_apply_pca(m, dX, x_dtype, n, m->ws_xp.as<float>(), nullptr));
    CIS_CHECK_HIP(hipMemcpy(out, m->ws_xp.p, (size_t)n * m->D * sizeof(float), hipMemcpyDeviceToHost));
    return CIS_OK;
}

extern "C" int cis_predict_coarse(cis_model* m, const void* X, int x_dtype, int64_t n, uint16_t* coarse) {
    CIS_ENTER(m);
    CIS_REQUIRE(x_dtype == CIS_F32 || x_dtype == CIS_F64, "x_dtype must be 4 or 8");
    if (n == 0) return CIS_OK;
    HostIO io{m};
    void* dX;
    CIS_TRY(io.in(X, (size_t)n * m->D * x_dtype, &dX));
    const void* xc; int ct;
    CIS_TRY(cis_dev_coarse_type(m, dX, x_dtype, n, &xc, &ct, nullptr));
    CIS_TRY(m->ws_out0.reserve((size_t)n * 2 * sizeof(uint16_t)));
    CIS_TRY(dev_predict_coarse(m, xc, ct, n, m->ws_out0.as<uint16_t>(), nullptr));
    CIS_CHECK_HIP(hipMemcpy(coarse, m->ws_out0.p, (size_t)n * 2 * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return CIS_OK;
}

static int check_coarse_host(cis_model* m, const uint16_t* coarse, int64_t n) {
    for (int64_t i = 0; i < 2 * n; ++i)
        CIS_REQUIRE(coarse[i] < m->V, "coarse code %d out of range (V=%d)", (int)coarse[i], m->V);
    return CIS_OK;
}

extern "C" int cis_project(cis_model* m, const void* X, int x_dtype, int64_t n, const uint16_t* coarse, double* out) {
    CIS_ENTER(m);
    CIS_REQUIRE(x_dtype == CIS_F32 || x_dtype == CIS_F64, "x_dtype must be 4 or 8");
    if (n == 0) return CIS_OK;
    CIS_TRY(check_coarse_host(m, coarse, n));
    HostIO io{m};
    void* dX;
    CIS_TRY(io.in(X, (size_t)n * m->D * x_dtype, &dX));
    const void* xc; int ct;
    CIS_TRY(cis_dev_coarse_type(m, dX, x_dtype, n, &xc, &ct, nullptr));
    CIS_TRY(m->ws_out0.reserve((size_t)n * 2 * sizeof(uint16_t)));
    CIS_CHECK_HIP(hipMemcpyAsync(m->ws_out0.p, coarse, (size_t)n * 2 * sizeof(uint16_t), hipMemcpyHostToDevice, nullptr));
    CIS_TRY(m->ws_proj.reserve((size_t)n * m->D * sizeof(double)));
    CIS_TRY(dev_project(m, xc, ct, n, m->ws_out0.as<uint16_t>(), m->ws_proj.as<double>(), nullptr));
    CIS_CHECK_HIP(hipMemcpy(out, m->ws_proj.p, (size_t)n * m->D * sizeof(double), hipMemcpyDeviceToHost));
    return CIS_OK;
}

extern "C" int cis_predict_fine(cis_model* m, const void* X, int x_dtype, int64_t n, const uint16_t* coarse,
                                uint8_t* fine) {
    CIS_ENTER(m);
    CIS_REQUIRE(x_dtype == CIS_F32 || x_dtype == CIS_F64, "x_dtype must be 4 or 8");
    if (n == 0) return CIS_OK;
    CIS_TRY(check_coarse_host(m, coarse, n));
    HostIO io{m};
    void* dX;
    CIS_TRY(io.in(X, (size_t)n * m->D * x_dtype, &dX));
    const void* xc; int ct;
    CIS_TRY(cis_dev_coarse_type(m, dX, x_dtype, n, &xc, &ct, nullptr));
    CIS_TRY(m->ws_out0.reserve((size_t)n * 2 * sizeof(uint16_t)));
    CIS_CHECK_HIP(hipMemcpyAsync(m->ws_out0.p, coarse, (size_t)n * 2 * sizeof(uint16_t), hipMemcpyHostToDevice, nullptr));
    CIS_TRY(m->ws_proj.reserve((size_t)n * m->D * sizeof(double)));
    CIS_TRY(dev_project(m, xc, ct, n, m->ws_out0.as<uint16_t>(), m->ws_proj.as<double>(), nullptr));
    CIS_TRY(m->ws_out1.reserve((size_t)n * m->M));
    CIS_TRY(dev_fine_from_proj(m, m->ws_proj.as<double>(), n, m->ws_out1.as<uint8_t>(), nullptr));
    CIS_CHECK_HIP(hipMemcpy(fine, m->ws_out1.p, (size_t)n * m->M, hipMemcpyDeviceToHost));
    return CIS_OK;
}

extern "C" int cis_subquantizer_distances(cis_model* m, const void* X, int x_dtype, int64_t n,
                                          const uint16_t* coarse, double* tables) {
    CIS_ENTER(m);
    CIS_REQUIRE(x_dtype == CIS_F32 || x_dtype == CIS_F64, "x_dtype must be 4 or 8");
    if (n == 0) return CIS_OK;
    CIS_TRY(check_coarse_host(m, coarse, n));
    HostIO io{m};
    void* dX;
    CIS_TRY(io.in(X, (size_t)n * m->D * x_dtype, &dX));
    const void* xc; int ct;
    CIS_TRY(cis_dev_coarse_type(m, dX, x_dtype, n, &xc, &ct, nullptr));
    CIS_TRY(m->ws_out0.reserve((size_t)n * 2 * sizeof(uint16_t)));
    CIS_CHECK_HIP(hipMemcpyAsync(m->ws_out0.p, coarse, (size_t)n * 2 * sizeof(uint16_t), hipMemcpyHostToDevice, nullptr));
    CIS_TRY(m->ws_proj.reserve((size_t)n * m->D * sizeof(double)));
    CIS_TRY(dev_project(m, xc, ct, n, m->ws_out0.as<uint16_t>(), m->ws_proj.as<double>(), nullptr));
    // tables[r][j][k] = squared distance of projected sub-vector j to sub-centroid k
    CIS_TRY(m->ws_dist.reserve((size_t)n * (m->V > m->K ? m->V : m->K) * sizeof(double)));
    double* dist = m->ws_dist.as<double>();
    dim3 g((unsigned)ceil_div(n, 16), (unsigned)ceil_div(m->K, 16));
    for (int j = 0; j < m->M; ++j) {
        hipLaunchKernelGGL(k_sqdist_rows<double>, g, dim3(256), 0, nullptr, m->ws_proj.as<double>(), (int64_t)m->D,
                           j * m->w, m->d_subs + (size_t)j * m->K * m->w, n, m->K, m->w, dist, m->prog_w, (const int*)nullptr);
        CIS_CHECK_HIP(hipMemcpy2D(tables + (size_t)j * m->K, (size_t)m->M * m->K * sizeof(double), dist,
                                  (size_t)m->K * sizeof(double), (size_t)m->K * sizeof(double), (size_t)n,
                                  hipMemcpyDeviceToHost));
    }
    return CIS_OK;
}

extern "C" int cis_reconstruct(cis_model* m, const uint16_t* coarse, const uint8_t* fine, int64_t n, double* out) {
    CIS_ENTER(m);
    if (n == 0) return CIS_OK;
    CIS_TRY(check_coarse_host(m, coarse, n));
    for (int64_t i = 0; i < n * m->M; ++i)
        CIS_REQUIRE(fine[i] < m->K, "fine code %d out of range (K=%d)", (int)fine[i], m->K);
    CIS_TRY(m->ws_out0.reserve((size_t)n * 2 * sizeof(uint16_t)));
    CIS_TRY(m->ws_out1.reserve((size_t)n * m->M));
    CIS_TRY(m->ws_proj.reserve((size_t)n * m->D * sizeof(double)));
    CIS_CHECK_HIP(hipMemcpy(m->ws_out0.p, coarse, (size_t)n * 2 * sizeof(uint16_t), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(m->ws_out1.p, fine, (size_t)n * m->M, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_reconstruct, dim3((unsigned)n, 2), dim3(128), 0, nullptr, m->ws_out0.as<uint16_t>(),
                       m->ws_out1.as<uint8_t>(), m->d_Rs, m->d_mus, m->d_Cs64, m->d_subs, n, m->V, m->M, m->K, m->h,
                       m->w, m->ws_proj.as<double>());
    CIS_CHECK_HIP(hipGetLastError());
    CIS_CHECK_HIP(hipMemcpy(out, m->ws_proj.p, (size_t)n * m->D * sizeof(double), hipMemcpyDeviceToHost));
    return CIS_OK;
}
