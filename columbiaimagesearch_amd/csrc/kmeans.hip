// Lloyd iterations for LOPQ training on the GPU.  The reference trains its coarse and fine
// codebooks with scikit-learn k-means (lopq/lopq/model.py:339-437); results of k-means depend on initialisation and
// library version, so this is judged by distortion, not bit parity.  float32 arithmetic.
//   k_km_assign: one thread per point, centroids staged in LDS (k * d <= 7680 floats), squared distances by fma,
//                first minimum; the workgroup adds its points into LDS sums (ds_add_f32), then flushes the non-empty
//                rows with one global atomic per (centroid, dimension) -- a persistent grid keeps those few.
//   k_km_update: centroid = sum / count (an empty cluster keeps its centroid).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/cis_hip.h"
#include "common.h"

__global__ __launch_bounds__(256) void k_km_assign(const float* __restrict__ X, int64_t n, int d, int k,
                                                   const float* __restrict__ C, int* __restrict__ assign,
                                                   float* __restrict__ sums /* [k][d] */, int* __restrict__ counts /* [k] */,
                                                   double* __restrict__ inertia) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);  // [k][d] centroids
    float* ss = sc + (size_t)k * d;               // [k][d] sums of this block
    int* scnt = reinterpret_cast<int*>(ss + (size_t)k * d);
    for (int e = threadIdx.x; e < k * d; e += 256) { sc[e] = C[e]; ss[e] = 0.f; }
    for (int e = threadIdx.x; e < k; e += 256) scnt[e] = 0;
    __syncthreads();
    double local = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        const float* x = X + r * d;
        float best = 3.4e38f;
        int bi = 0;
        for (int c = 0; c < k; ++c) {
            const float* cc = sc + (size_t)c * d;
            float acc = 0.f;
            for (int i = 0; i < d; ++i) {
                const float df = x[i] - cc[i];
                acc = fmaf(df, df, acc);
            }
            if (acc < best) { best = acc; bi = c; }
        }
        if (assign) assign[r] = bi;
        local += (double)best;
        float* dst = ss + (size_t)bi * d;
        for (int i = 0; i < d; ++i) atomicAdd(&dst[i], x[i]);
        atomicAdd(&scnt[bi], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < k * d; e += 256)
        if (scnt[e / d] > 0) atomicAdd(&sums[e], ss[e]);
    for (int e = threadIdx.x; e < k; e += 256)
        if (scnt[e] > 0) atomicAdd(&counts[e], scnt[e]);
    // block reduction of the inertia
    __shared__ double s_in[256];
    s_in[threadIdx.x] = local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_in[threadIdx.x] += s_in[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(inertia, s_in[0]);
}

__global__ void k_km_update(float* __restrict__ C, float* __restrict__ sums, int* __restrict__ counts, int k, int d) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * d) return;
    const int c = counts[e / d];
    if (c > 0) C[e] = sums[e] / (float)c;
}

// X [n][d] float32 (host), centroids [k][d] float32: initial values in, final values out.  assign [n] (or NULL) and
// *inertia describe the assignment to the RETURNED centroids (one extra assignment pass after the last update).
extern "C" int cis_kmeans(const float* X, int64_t n, int d, int k, int iters, float* centroids, int32_t* assign, double* inertia) {
    CIS_REQUIRE(X && centroids && n > 0 && d > 0 && k > 0 && iters >= 0, "bad k-means arguments");
    CIS_REQUIRE((size_t)k * d <= 7680, "k * d must be <= 7680 (centroids and block sums live in LDS)");
    CIS_TRY(cis_lazy_init());
    float *dX = nullptr, *dC = nullptr, *dS = nullptr;
    int *dA = nullptr, *dN = nullptr;
    double* dI = nullptr;
    auto done = [&](int rc) {
        if (dX) (void)hipFree(dX); if (dC) (void)hipFree(dC); if (dS) (void)hipFree(dS);
        if (dA) (void)hipFree(dA); if (dN) (void)hipFree(dN); if (dI) (void)hipFree(dI);
        return rc;
    };
#define KM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cis_set_error("%s: %s", #call, hipGetErrorString(e_)); return done(CIS_EHIP); } } while (0)
    KM_HIP(hipMalloc((void**)&dX, (size_t)n * d * sizeof(float)));
    KM_HIP(hipMalloc((void**)&dC, (size_t)k * d * sizeof(float)));
    KM_HIP(hipMalloc((void**)&dS, (size_t)k * d * sizeof(float)));
    KM_HIP(hipMalloc((void**)&dN, (size_t)k * sizeof(int)));
    KM_HIP(hipMalloc((void**)&dI, sizeof(double)));
    if (assign) KM_HIP(hipMalloc((void**)&dA, (size_t)n * sizeof(int)));
    KM_HIP(hipMemcpy(dX, X, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice));
    KM_HIP(hipMemcpy(dC, centroids, (size_t)k * d * sizeof(float), hipMemcpyHostToDevice));
    const size_t lds = (size_t)2 * k * d * sizeof(float) + (size_t)k * sizeof(int);
    const unsigned grid = (unsigned)(ceil_div(n, 256) < 1024 ? ceil_div(n, 256) : 1024);
    for (int it = 0; it <= iters; ++it) {
        KM_HIP(hipMemsetAsync(dS, 0, (size_t)k * d * sizeof(float), nullptr));
        KM_HIP(hipMemsetAsync(dN, 0, (size_t)k * sizeof(int), nullptr));
        KM_HIP(hipMemsetAsync(dI, 0, sizeof(double), nullptr));
        const bool last = it == iters;
        hipLaunchKernelGGL(k_km_assign, dim3(grid), dim3(256), lds, nullptr, dX, n, d, k, dC, last ? dA : nullptr, dS, dN, dI);
        if (!last) hipLaunchKernelGGL(k_km_update, dim3((unsigned)ceil_div((int64_t)k * d, 256)), dim3(256), 0, nullptr, dC, dS, dN, k, d);
    }
    KM_HIP(hipGetLastError());
    KM_HIP(hipMemcpy(centroids, dC, (size_t)k * d * sizeof(float), hipMemcpyDeviceToHost));
    if (assign) KM_HIP(hipMemcpy(assign, dA, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    if (inertia) KM_HIP(hipMemcpy(inertia, dI, sizeof(double), hipMemcpyDeviceToHost));
#undef KM_HIP
    return done(CIS_OK);
}
