// LOPQ training on the GPU beyond the k-means steps: the accumulations the reference does
// with per-sample Python loops -- np.outer sums for the PCA covariance (lopq/lopq/model.py:263-267) and for the per-cluster
// residual covariances (:142-155), and the per-cluster projection of the residuals (:209-234) -- as float64 tiled products.
// The eigendecompositions stay on the host (LAPACK, V matrices of h x h): they are O(V h^3) against O(n h^2) here.
//   k_gram_groups:    G[g] = sum_{r in group g} x_r x_r^T (upper 64x64 tiles, mirrored), s[g] = sum x_r; rows sorted by group
//   k_project_groups: y_r = (x_r - mu[g]) . R[g]^T for the rows of group g
// float64 fused multiply-adds, k-ascending inside a 16-row LDS stage; results differ from numpy's BLAS by summation order
// only (tests: 1e-12 relative).  Training is judged by distortion, not bit parity.
#include "common.h"

// grid (tiles_j, tiles_i, groups); block 256 = 16 x 16 threads, 4 x 4 outputs each (64 x 64 tile); only tiles with
// j0 >= i0 compute, the mirror is written by the same block
__global__ __launch_bounds__(256) void k_gram_groups(const double* __restrict__ X, const int64_t* __restrict__ goff /* [groups+1] */,
                                                     int d, double* __restrict__ G /* [groups][d][d] */, double* __restrict__ S /* [groups][d] */) {
    const int g = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    if (j0 < i0) return;
    __shared__ double sa[16][64 + 1], sb[16][64 + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t r0 = goff[g], r1 = goff[g + 1];
    double acc[4][4], csum[4] = {0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int64_t r = r0; r < r1; r += 16) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int rr = e >> 6, c = e & 63;
            const bool on = r + rr < r1;
            sa[rr][c] = (on && i0 + c < d) ? X[(r + rr) * d + i0 + c] : 0.0;
            sb[rr][c] = (on && j0 + c < d) ? X[(r + rr) * d + j0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[q] = sa[rr][ty * 4 + q]; b[q] = sb[rr][tx * 4 + q]; }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], b[q], acc[p][q]);
            if (i0 == 0 && ty == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) csum[q] += b[q];
            }
        }
        __syncthreads();
    }
    double* Gg = G + (int64_t)g * d * d;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + ty * 4 + p, j = j0 + tx * 4 + q;
            if (i < d && j < d) {
                Gg[(int64_t)i * d + j] = acc[p][q];
                Gg[(int64_t)j * d + i] = acc[p][q];
            }
        }
    if (S && i0 == 0 && ty == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j0 + tx * 4 + q < d) S[(int64_t)g * d + j0 + tx * 4 + q] = csum[q];
    }
}

// grid (tiles of 64 outputs, ceil(max rows of a group / 64), groups): y[r][o] = sum_k (x[r][k] - mu[g][k]) * R[g][o][k]
__global__ __launch_bounds__(256) void k_project_groups(const double* __restrict__ X, const int64_t* __restrict__ goff, int d,
                                                        const double* __restrict__ R /* [groups][d][d] */,
                                                        const double* __restrict__ mu /* [groups][d] */, double* __restrict__ Y,
                                                        int64_t tile_base /* first row tile of this launch (groups above 4M rows take several) */) {
    const int g = blockIdx.z, o0 = blockIdx.x * 64;
    const int64_t r0 = goff[g] + (tile_base + (int64_t)blockIdx.y) * 64, r1 = goff[g + 1];
    if (r0 >= r1) return;
    __shared__ double sx[16][64 + 1], sr[16][64 + 1];  // [k][row], [k][output]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    const double* Rg = R + (int64_t)g * d * d;
    const double* mg = mu + (int64_t)g * d;
    for (int k0 = 0; k0 < d; k0 += 16) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int kk = e & 15, c = e >> 4;  // consecutive threads walk k: contiguous in X rows and R rows
            const bool kon = k0 + kk < d;
            sx[kk][c] = (kon && r0 + c < r1) ? X[(r0 + c) * d + k0 + kk] - mg[k0 + kk] : 0.0;
            sr[kk][c] = (kon && o0 + c < d) ? Rg[(int64_t)(o0 + c) * d + k0 + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[q] = sx[kk][ty * 4 + q]; b[q] = sr[kk][tx * 4 + q]; }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], b[q], acc[p][q]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = r0 + ty * 4 + p;
            const int o = o0 + tx * 4 + q;
            if (r < r1 && o < d) Y[r * d + o] = acc[p][q];
        }
}

struct TrainBufs {
    double *dX = nullptr, *dG = nullptr, *dS = nullptr, *dR = nullptr, *dM = nullptr, *dY = nullptr;
    int64_t* dOff = nullptr;
    ~TrainBufs() {
        for (void* p : {(void*)dX, (void*)dG, (void*)dS, (void*)dR, (void*)dM, (void*)dY, (void*)dOff})
            if (p) (void)hipFree(p);
    }
};

extern "C" int cis_train_gram(const double* X, int64_t n, int d, const int64_t* group_off, int groups, double* G, double* S) {
    CIS_REQUIRE(X && group_off && G && n >= 0 && d > 0 && groups > 0, "bad arguments");
    CIS_REQUIRE(group_off[0] == 0 && group_off[groups] == n, "group offsets must cover [0, n)");
    CIS_TRY(cis_lazy_init());
    TrainBufs b;
    const size_t gx = (size_t)(n > 0 ? n : 1) * d * sizeof(double), gg = (size_t)groups * d * d * sizeof(double);
    CIS_CHECK_HIP(hipMalloc((void**)&b.dX, gx));
    CIS_CHECK_HIP(hipMalloc((void**)&b.dG, gg));
    CIS_CHECK_HIP(hipMalloc((void**)&b.dS, (size_t)groups * d * sizeof(double)));
    CIS_CHECK_HIP(hipMalloc((void**)&b.dOff, (size_t)(groups + 1) * sizeof(int64_t)));
    if (n > 0) CIS_CHECK_HIP(hipMemcpy(b.dX, X, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(b.dOff, group_off, (size_t)(groups + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    const unsigned t = (unsigned)ceil_div(d, 64);
    for (int g0 = 0; g0 < groups; g0 += 32768) {  // gridDim.z limit
        const int ng = groups - g0 < 32768 ? groups - g0 : 32768;
        hipLaunchKernelGGL(k_gram_groups, dim3(t, t, (unsigned)ng), dim3(256), 0, nullptr, b.dX, b.dOff + g0, d,
                           b.dG + (int64_t)g0 * d * d, b.dS + (int64_t)g0 * d);
    }
    CIS_CHECK_HIP(hipGetLastError());
    CIS_CHECK_HIP(hipMemcpy(G, b.dG, gg, hipMemcpyDeviceToHost));
    if (S) CIS_CHECK_HIP(hipMemcpy(S, b.dS, (size_t)groups * d * sizeof(double), hipMemcpyDeviceToHost));
    return CIS_OK;
}

extern "C" int cis_train_project(const double* X, int64_t n, int d, const int64_t* group_off, int groups, const double* R,
                                 const double* mu, double* Y) {
    CIS_REQUIRE(X && group_off && R && mu && Y && n >= 0 && d > 0 && groups > 0, "bad arguments");
    CIS_REQUIRE(group_off[0] == 0 && group_off[groups] == n, "group offsets must cover [0, n)");
    if (n == 0) return CIS_OK;
    CIS_TRY(cis_lazy_init());
    TrainBufs b;
    CIS_CHECK_HIP(hipMalloc((void**)&b.dX, (size_t)n * d * sizeof(double)));
    CIS_CHECK_HIP(hipMalloc((void**)&b.dY, (size_t)n * d * sizeof(double)));
    CIS_CHECK_HIP(hipMalloc((void**)&b.dR, (size_t)groups * d * d * sizeof(double)));
    CIS_CHECK_HIP(hipMalloc((void**)&b.dM, (size_t)groups * d * sizeof(double)));
    CIS_CHECK_HIP(hipMalloc((void**)&b.dOff, (size_t)(groups + 1) * sizeof(int64_t)));
    CIS_CHECK_HIP(hipMemcpy(b.dX, X, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(b.dR, R, (size_t)groups * d * d * sizeof(double), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(b.dM, mu, (size_t)groups * d * sizeof(double), hipMemcpyHostToDevice));
    CIS_CHECK_HIP(hipMemcpy(b.dOff, group_off, (size_t)(groups + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    int64_t max_rows = 0;
    for (int g = 0; g < groups; ++g) max_rows = group_off[g + 1] - group_off[g] > max_rows ? group_off[g + 1] - group_off[g] : max_rows;
    const int64_t row_tiles = ceil_div(max_rows, 64);
    for (int g0 = 0; g0 < groups; g0 += 32768) {
        const int ng = groups - g0 < 32768 ? groups - g0 : 32768;
        for (int64_t t0 = 0; t0 < row_tiles; t0 += 65535) {  // the grid's y extent holds 65535 tiles = 4M rows of a group
            const int64_t nt = row_tiles - t0 < 65535 ? row_tiles - t0 : 65535;
            hipLaunchKernelGGL(k_project_groups, dim3((unsigned)ceil_div(d, 64), (unsigned)nt, (unsigned)ng), dim3(256), 0, nullptr,
                               b.dX, b.dOff + g0, d, b.dR + (int64_t)g0 * d * d, b.dM + (int64_t)g0 * d, b.dY, t0);
        }
    }
    CIS_CHECK_HIP(hipGetLastError());
    CIS_CHECK_HIP(hipMemcpy(Y, b.dY, (size_t)n * d * sizeof(double), hipMemcpyDeviceToHost));
    return CIS_OK;
}
