// Issue rate of the VALU instructions the ADC scans are made of (wave64, gfx950): cycles per wave-instruction per SIMD at full occupancy.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o tools/probes/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 977u + i * 131u;
    uint32_t sh = (threadIdx.x & 3) * 8, c = threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(sh));
                else if (OP == 1) asm volatile("v_lshl_add_u32 %0, %0, 7, %1" : "+v"(a[i]) : "v"(c));
                else if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
                else if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                else if (OP == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                else if (OP == 7) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
                else if (OP == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(uint64_t*)&a[i & ~1]) : "v"(*(uint64_t*)&a[(i & ~1)]));
                else if (OP == 9) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            }
        }
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; ++i) x ^= a[i];
    if (x == 0x12345u) out[0] = x;
}
template <int OP>
int run(const char* name, uint32_t* out) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4000, grid = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters, 12345u);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r) best = ms < best ? ms : best;
    }
    const double instr_per_simd = (double)iters * 32 * 8;    // 32 instructions per iteration and wave, 8 waves per SIMD
    printf("%-16s %.3f ms  %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz (%.2f at 2.1)\n", name, best, best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.4,
           best * 1e6 / instr_per_simd * 2.1);
    return 0;
}
int main() {
    uint32_t* out; CHECK(hipMalloc(&out, 64));
    run<5>("v_fma_f32", out); run<2>("v_add_f32", out); run<0>("v_bfe_u32", out); run<1>("v_lshl_add_u32", out); run<3>("v_add_u32", out); run<4>("v_cndmask_b32", out);
    run<6>("v_and_b32", out); run<7>("v_lshrrev_b32", out); run<8>("v_pk_add_f32", out); run<9>("v_min_u32", out);
    return 0;
}
