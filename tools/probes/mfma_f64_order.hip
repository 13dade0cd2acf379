// Probe: is v_mfma_f64_16x16x4_f64 one sequential chain of fused multiply-adds over k (k = 0, 1, 2, 3 into the accumulator)?
// Compares the MFMA's 16 x 16 results over K = 64 (16 instructions) with fma chains in float64, bit for bit, on random operands.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probes/mfma_f64_order.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A /* [16][64] row i, k */, const double* B /* [64][16] k, col t */, double* D /* [16][16] */) {
    const int l = threadIdx.x;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int s = 0; s < 16; ++s) {
        const double a = A[(l & 15) * 64 + 4 * s + (l >> 4)];
        const double b = B[(4 * s + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
int main() {
    const int trials = 2000;
    std::vector<double> A(16 * 64), B(64 * 16), D(256);
    double *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dD, 256 * 8);
    srand(7);
    long bad_seq = 0, bad_rev = 0, bad_pair = 0, total = 0;
    for (int t = 0; t < trials; ++t) {
        for (auto& x : A) x = (rand() / (double)RAND_MAX - 0.5) * ((t & 1) ? 1e3 : 1.0);
        for (auto& x : B) x = (rand() / (double)RAND_MAX - 0.5);
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int c = 0; c < 16; ++c) {
                double seq = 0.0;
                for (int kk = 0; kk < 64; ++kk) seq = __builtin_fma(A[i * 64 + kk], B[kk * 16 + c], seq);
                double rev = 0.0;  // within each group of four: k = 3, 2, 1, 0
                for (int s = 0; s < 16; ++s) for (int kk = 3; kk >= 0; --kk) rev = __builtin_fma(A[i * 64 + 4 * s + kk], B[(4 * s + kk) * 16 + c], rev);
                double pr = 0.0;   // each group of four summed on its own, then added
                for (int s = 0; s < 16; ++s) { double g = 0.0; for (int kk = 0; kk < 4; ++kk) g = __builtin_fma(A[i * 64 + 4 * s + kk], B[(4 * s + kk) * 16 + c], g); pr += g; }
                const double d = D[i * 16 + c];
                bad_seq += memcmp(&d, &seq, 8) != 0;
                bad_rev += memcmp(&d, &rev, 8) != 0;
                bad_pair += memcmp(&d, &pr, 8) != 0;
                ++total;
            }
    }
    printf("results %ld: differ from the sequential fma chain %ld, from the reversed-in-group chain %ld, from group sums %ld\n", total, bad_seq, bad_rev, bad_pair);
    return 0;
}
