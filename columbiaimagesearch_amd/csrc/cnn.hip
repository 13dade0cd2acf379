// DeepSentibank (CaffeNet) forward pass on MI355X.
//
// Replaces the caffe arithmetic behind SentiBankPyCaffeImgFeaturizer.featurize
// (cufacesearch/cufacesearch/featurizer/sbpycaffe_img_featurizer.py:137-154; network
// cufacesearch/cufacesearch/featurizer/data/pycaffe_sentibank.prototxt:1-212) for whole batches:
// input = what preprocess_img (:113-134) hands to the net (N x 3 x 227 x 227 float32, BGR, mean
// subtracted), output = blobs['fc7'] after the in-place ReLU (:154), N x 4096 float32.
//
// Convolutions and fully connected layers are implicit GEMMs on the float32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact float32 multiply-add, 157 TFLOP/s peak): M = output pixels,
// N = output channels of one group, K = KH*KW*IC/group; the im2col operand is gathered straight into
// LDS, never materialised.  Activations are kept channels-last (NHWC) between layers so that the K
// axis of the gather and the across-channel LRN window are contiguous.  Bias + ReLU are fused into the
// GEMM epilogue; pooling uses caffe's ceil-mode output size with windows clipped to the input.
#include <algorithm>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvDesc {
    int N, H, W, C;          // input: images, height, width, total channels
    int OH, OW, OC;          // output
    int KH, KW, stride, pad, groups;
    int ICg, OCg, K;         // per group; K = KH*KW*ICg, k = (ky*KW + kx)*ICg + ic
    int64_t sN, sC, sH, sW;  // input strides in elements (NCHW for the first layer, NHWC afterwards)
    int relu;
    int kx_fastest;          // K ordering of the packed weights: 0: k = (ky*KW+kx)*ICg+ic, 1: k = (ic*KH+ky)*KW+kx (NCHW input)
    int splitk = 1;          // > 1 (groups == 1, vector path only): blockIdx.z owns a K range and writes raw partial sums
    int64_t part_stride = 0; // elements between the partial outputs of consecutive K ranges
    int runq = 0;            // > 0: "row-run" gather of the 3-channel first layer (vector path): K is cut per kernel row into runq
                             // float4 of contiguous NHWC input (KW*C floats, zero-weight padded), k = ky*4*runq + (kx*C + ic)
    const float* res = nullptr;  // residual input [npix][resC] added before the ReLU (channels >= resC get nothing): dlib add_prev
    int resC = 0;
    int run_planes = 0;      // > 0 (with runq): the runs are taken from the caller's NCHW planes directly -- run r = ic * KH + ky is the KW
                             // contiguous floats of channel plane ic, row iy0 + ky, from column ix0 (runq float4, the tail meets zero
                             // weights), k = r * 4 * runq + kx; run_planes = ICg * KH runs; in_elems = floats in the input (the last
                             // float4 of the last run would end past it: it is loaded one to three floats earlier and shifted)
    int64_t in_elems = 0;
};

// C[pixel][oc] = sum_k A[pixel][k] * Wp[k][oc] + bias[oc]; block tile BM x BN, 256 threads = 4 waves laid
// out WAVES_M x WAVES_N, each wave WM x WN MFMA tiles of 32x32.  K advances 16 at a time through two LDS
// stages: the next stage's operands are fetched into registers before the MFMAs of the current stage and
// written to the other LDS buffer after them (one barrier per stage).
// VEC = channels-last input with ICg % 16 == 0: a stage covers 16 consecutive input channels of ONE kernel tap,
// so every thread fetches float4s along the channel axis and the tap decode is wave-uniform.
// MODE: 0 = scalar gather, 1 = VEC, 2 = VEC with the row-run gather of the 3-channel first layers (ConvDesc::runq), 3 = the run gather
// over NCHW planes (ConvDesc::run_planes) -- a template
// parameter since round 4: as a run-time branch inside the fetch it made the compiler emit both gathers with a wait between them.
template <int WM, int WN, int WAVES_M, int WAVES_N, int MODE>
__global__ __launch_bounds__(256, 4) void k_conv_igemm(const float* __restrict__ in, const float* __restrict__ Wp,
                                                    const float* __restrict__ bias, float* __restrict__ out, ConvDesc d) {
    constexpr bool VEC = MODE != 0, RUN = MODE >= 2, PLANES = MODE == 3;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, BK = 16;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A_VEC = (BM * 4 + 255) / 256;        // float4 fetches per thread per stage (VEC)
    constexpr int A_SCL = (BM * BK + 255) / 256;       // scalar fetches per thread per stage (!VEC)
    constexpr int B_VEC = (BK * (BN / 4) + 255) / 256; // float4 fetches of packed weights
    __shared__ float As[2][BK][LDA];  // [stage][k][pixel]
    __shared__ float Bs[2][BK][LDB];  // [stage][k][oc]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = d.splitk > 1 ? 0 : blockIdx.z;
    const int ksplit = d.splitk > 1 ? blockIdx.z : 0;
    const int64_t npix = (int64_t)d.N * d.OH * d.OW;
    const int64_t pix0 = (int64_t)blockIdx.x * BM;
    const int oc0 = blockIdx.y * BN;  // inside the group
    // ---- the pixels this thread gathers are the same for every stage: decode them once ----
    constexpr int NPX = VEC ? A_VEC : A_SCL;
    int iy0[NPX], ix0[NPX], mloc[NPX];
    int64_t pbase[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int m = VEC ? ((tid >> 2) + 64 * i) : ((tid >> 4) + 16 * i);
        mloc[i] = m;
        const int64_t p = pix0 + m;
        if (m < BM && p < npix) {
            const int ox = (int)(p % d.OW);
            const int oy = (int)((p / d.OW) % d.OH);
            const int64_t img = p / ((int64_t)d.OW * d.OH);
            iy0[i] = oy * d.stride - d.pad;
            ix0[i] = ox * d.stride - d.pad;
            pbase[i] = img * d.sN + (int64_t)g * d.ICg * d.sC;
        } else {
            iy0[i] = -(1 << 28);  // every tap falls outside -> zeros
            ix0[i] = -(1 << 28);
            pbase[i] = 0;
        }
    }
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* Wg = Wp + (int64_t)g * d.K * d.OCg;
    const int nkt_all = (d.K + BK - 1) / BK;
    const int nkt_per = (nkt_all + d.splitk - 1) / d.splitk;
    const int kt_begin = ksplit * nkt_per;
    const int nkt = (kt_begin + nkt_per < nkt_all) ? kt_begin + nkt_per : nkt_all;  // one past this block's last K tile
    // tap decode state: VEC -> of the stage (uniform); !VEC -> of this thread's k = k0 + (tid & 15)
    int ky = 0, kx = 0, ic = 0;
    if (VEC && kt_begin > 0) {  // start of this block's K range
        const int k0 = kt_begin * BK;
        ic = k0 % d.ICg;
        const int t = k0 / d.ICg;
        kx = t % d.KW;
        ky = t / d.KW;
    }
    if (!VEC) {
        if (d.kx_fastest) {
            kx = tid & 15;
            while (kx >= d.KW) { kx -= d.KW; if (++ky == d.KH) { ky = 0; ++ic; } }
        } else {
            ic = tid & 15;
            while (ic >= d.ICg) { ic -= d.ICg; if (++kx == d.KW) { kx = 0; ++ky; } }
        }
    }
    int run_r = 0, run_jq = 0, run_ic = 0, run_ky = 0;  // RUN: this thread's float4 of the stage = quad run_jq of run run_r = (run_ic, run_ky)
    if constexpr (RUN) {
        const int kq0 = kt_begin * 4 + (tid & 3);
        run_r = kq0 / d.runq; run_jq = kq0 - run_r * d.runq;
        run_ic = run_r / d.KH; run_ky = run_r - run_ic * d.KH;
    }
    float4 ra[VEC ? A_VEC : 1];
    float rs[VEC ? 1 : A_SCL];
    float4 rb[B_VEC];
    bool okA[VEC ? A_VEC : 1], okB[B_VEC];
    int shA[PLANES ? A_VEC : 1];
    // The fetch of the vector path is ONLY loads (round 4): every address is clamped into the arrays, the loads are unconditional and
    // nothing touches their registers before the stage's MFMAs are issued; what is padding, past the tile or past K is zeroed when
    // the registers go to LDS (okA / okB).  With `if (inside) load` the compiler built a branch per load and waited for each load
    // before the next (build/cnn.s: global_load -> s_waitcnt vmcnt(0) -> global_load ...): two to five exposed round trips per stage.
    auto fetch = [&](int kt) {
        const int k0 = kt * BK;
        if constexpr (VEC) {
            if constexpr (RUN) {
                // this thread's float4 along K: run r (kernel row ky of the NHWC rows, or (channel plane, kernel row) of the NCHW planes),
                // quad jq of that run's contiguous floats (no padding: pad == 0)
                // (the run decode is carried from stage to stage: as kq / runq and r / KH per stage the two integer divisions were a
                // fifth of the first layer's instructions)
                const int r = run_r, jq = run_jq, ric = run_ic, rky = run_ky;
                const int nruns = PLANES ? d.run_planes : d.KH;
                run_jq += 4;
                while (run_jq >= d.runq) {
                    run_jq -= d.runq; ++run_r;
                    if (++run_ky == d.KH) { run_ky = 0; ++run_ic; }
                }
#pragma unroll
                for (int i = 0; i < A_VEC; ++i) {
                    const int iy = iy0[i] + rky;
                    okA[i] = mloc[i] < BM && r < nruns && iy >= 0 && iy < d.H;
                    const int iyc = iy < 0 ? 0 : (iy < d.H ? iy : d.H - 1);
                    const int ixc = ix0[i] < 0 ? 0 : ix0[i];
                    if constexpr (PLANES) {
                        const int64_t off = pbase[i] + (int64_t)(r < nruns ? ric : 0) * d.sC + (int64_t)(iyc * (int)d.sH + ixc + jq * 4);
                        const int64_t over = off + 4 - d.in_elems;  // 1 ... 3 for the last runs of the last plane, else <= 0
                        const int sh = over > 0 ? (int)over : 0;
                        shA[i] = sh;  // applied when the registers go to LDS
                        ra[i] = *reinterpret_cast<const float4*>(in + (off - sh));
                    } else {
                        ra[i] = *reinterpret_cast<const float4*>(in + pbase[i] + (int64_t)iyc * d.sH + (int64_t)ixc * d.sW + jq * 4);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < A_VEC; ++i) {
                    const int iy = iy0[i] + ky, ix = ix0[i] + kx;
                    okA[i] = mloc[i] < BM && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
                    const int iyc = iy < 0 ? 0 : (iy < d.H ? iy : d.H - 1);
                    const int ixc = ix < 0 ? 0 : (ix < d.W ? ix : d.W - 1);
                    ra[i] = *reinterpret_cast<const float4*>(in + pbase[i] + (int64_t)iyc * d.sH + (int64_t)ixc * d.sW + ic + (tid & 3) * 4);
                }
            }
        } else {
            const bool kvalid = (k0 + (tid & 15)) < d.K;
#pragma unroll
            for (int i = 0; i < A_SCL; ++i) {
                const int iy = iy0[i] + ky, ix = ix0[i] + kx;
                rs[i] = 0.f;
                if (kvalid && mloc[i] < BM && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W)
                    rs[i] = in[pbase[i] + (int64_t)ic * d.sC + (int64_t)iy * d.sH + (int64_t)ix * d.sW];
            }
        }
#pragma unroll
        for (int i = 0; i < B_VEC; ++i) {
            constexpr int LIM = BK * (BN / 4);
            const int idx = tid + 256 * i;
            const int idc = idx < LIM ? idx : LIM - 1;
            const int kl = idc / (BN / 4), n4 = (idc % (BN / 4)) * 4;
            okB[i] = idx < LIM && k0 + kl < d.K && oc0 + n4 < d.OCg;
            if constexpr (VEC) {
                const int kc = k0 + kl < d.K ? k0 + kl : d.K - 1;
                const int nc = oc0 + n4 < d.OCg ? oc0 + n4 : d.OCg - 4;  // (OCg % 4 == 0: the float4 rows of the packed weights)
                rb[i] = *reinterpret_cast<const float4*>(Wg + (int64_t)kc * d.OCg + nc);
            } else {
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (okB[i]) rb[i] = *reinterpret_cast<const float4*>(Wg + (int64_t)(k0 + kl) * d.OCg + oc0 + n4);
            }
        }
        // advance the tap decode to the next stage
        if (!VEC && d.kx_fastest) {
            kx += BK;
            while (kx >= d.KW) { kx -= d.KW; if (++ky == d.KH) { ky = 0; ++ic; } }
        } else {
            ic += BK;
            while (ic >= d.ICg) { ic -= d.ICg; if (++kx == d.KW) { kx = 0; ++ky; } }
        }
    };
    auto stash = [&](int st) {
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < A_VEC; ++i) {
                if (mloc[i] < BM) {
                    const int kq = (tid & 3) * 4;
                    const bool ok = okA[i];
                    float4 v = ra[i];
                    if constexpr (PLANES) {  // a load that was moved back from the end of the input: its floats move down, zeros follow
                        const int sh = shA[i];
                        v = make_float4(sh == 0 ? v.x : (sh == 1 ? v.y : (sh == 2 ? v.z : v.w)), sh == 0 ? v.y : (sh == 1 ? v.z : (sh == 2 ? v.w : 0.f)),
                                        sh == 0 ? v.z : (sh == 1 ? v.w : 0.f), sh == 0 ? v.w : 0.f);
                    }
                    As[st][kq + 0][mloc[i]] = ok ? v.x : 0.f; As[st][kq + 1][mloc[i]] = ok ? v.y : 0.f;
                    As[st][kq + 2][mloc[i]] = ok ? v.z : 0.f; As[st][kq + 3][mloc[i]] = ok ? v.w : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_SCL; ++i)
                if (mloc[i] < BM) As[st][tid & 15][mloc[i]] = rs[i];
        }
#pragma unroll
        for (int i = 0; i < B_VEC; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BK * (BN / 4)) {
                const int kl = idx / (BN / 4), n4 = (idx % (BN / 4)) * 4;
                const bool ok = okB[i];
                Bs[st][kl][n4 + 0] = ok ? rb[i].x : 0.f; Bs[st][kl][n4 + 1] = ok ? rb[i].y : 0.f;
                Bs[st][kl][n4 + 2] = ok ? rb[i].z : 0.f; Bs[st][kl][n4 + 3] = ok ? rb[i].w : 0.f;
            }
        }
    };
    if (kt_begin < nkt) {
        fetch(kt_begin);
        stash(kt_begin & 1);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) fetch(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = As[cur][kk + (lane >> 5)][(wm * WM + i) * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = Bs[cur][kk + (lane >> 5)][(wn * WN + j) * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) stash(cur ^ 1);
        __syncthreads();
    }
    // ---- epilogue: C/D layout col = lane&31 (oc), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) ----
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int ocl = oc0 + (wn * WN + j) * 32 + (lane & 31);
        if (ocl >= d.OCg) continue;
        const float bv = bias ? bias[g * d.OCg + ocl] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t p = pix0 + (wm * WM + i) * 32 + row;
                if (p < npix) {
                    if (d.splitk > 1) {
                        out[(int64_t)ksplit * d.part_stride + p * d.OC + ocl] = acc[i][j][r];  // raw partial sum
                    } else {
                        float v = acc[i][j][r] + bv;
                        if (d.res && ocl < d.resC) v += d.res[p * d.resC + ocl];
                        if (d.relu) v = v > 0.f ? v : 0.f;
                        out[p * d.OC + g * d.OCg + ocl] = v;
                    }
                }
            }
        }
    }
}

// s_waitcnt vmcnt(n) for a compile-time n that is only known after unrolling (the immediate must be a literal)
static __device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Direct 3x3 / stride 1 / pad 1 convolution with C input = C output channels on a channels-last activation (the residual
// branches of the dlib net at 35x35x32 and 17x17x64: half of its multiply-adds).  The implicit GEMM above gathers every
// input pixel nine times from global memory, once per tap, and transposes it into LDS each time.  Here the input of a
// workgroup's output range reaches LDS ONCE and the nine taps are nine shifted views of it:
//   positions are numbered on a grid with one shared zero column per row and one shared zero row per image,
//   q = (img*(H+1) + r)*(W+1) + c, so that the input of output q for tap (ky, kx) is position q + (ky-1)*(W+1) + (kx-1)
//   for EVERY q (the borders read the zero column / zero row) -- a pure shift of the flattened index.  The grid exists
//   only inside the kernel: the fill decodes q and reads the ordinary NHWC activation, the epilogue writes valid q only.
// LDS: A window [BM + 2(W+2)][C+2] floats (pitch 2*odd: the 32 positions of a ds_read_b64 lane group fall on 32 distinct
// bank pairs), weights of one tap [C/2][C][2] double-buffered (a lane's two k values adjacent).  One ds_read_b64 feeds two
// v_mfma_f32_32x32x2_f32: lane half h holds k = kk+2h and kk+2h+1 of a 4-deep k step, A and B paired alike.
// Accumulation order per output: taps in (ky, kx) order, channels ascending in 4-deep steps -- the same for every pixel,
// batch size and tile position.
template <int C, int H, int W, int WM, int NWM, int NWN>
__global__ __launch_bounds__(NWM * NWN * 64) void k_conv3x3_direct(const float* __restrict__ in, const float* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ out, int N,
                                                           const float* __restrict__ res, int resC, int relu) {
    constexpr int P = W + 1, IMGQ = (H + 1) * P, HALO = P + 1;
    constexpr int NW = NWM * NWN, BM = NWM * WM * 32, NPOS = BM + 2 * HALO, LDC = C + 2, WN = C / 32 / NWN, NT = NW * 64;
    static_assert(WN * NWN * 32 == C, "output channels split evenly over the waves");
    constexpr int C4 = C / 4;
    extern __shared__ float smem[];
    float* As = smem;                    // [NPOS][LDC]
    float* Bs = smem + NPOS * LDC;       // [2][C/2][C][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int wm = wave / NWN, wn = wave % NWN;  // wave tile: WM*32 positions x WN*32 output channels
    const int64_t Qtot = (int64_t)N * IMGQ;
    const int64_t q0 = (int64_t)blockIdx.x * BM;
    // ---- the first WD taps of weights -> registers; Wp holds the paired layout [tap][C/2][C][2] ----
    constexpr int WQ = C * C / 4;                          // float4 per tap
    constexpr int WIT = (WQ + NT - 1) / NT;
    // The fetch is inline assembly: as a C++ load the compiler sinks it below the tap's MFMAs, next to the LDS write that
    // consumes it, and every tap then waits out a global-memory round trip.  The matching wait is wait_vmcnt().
    constexpr int WD = 3;  // taps of weights in flight: one tap of MFMAs (0.4 - 1.7 us) does not cover a loaded L2 round trip
    f32x4 wr[WD][WIT];
#define CIS_WFETCH(tap_)                                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < WIT; ++i_) {                                                               \
        const int idx_ = tid + NT * i_;                                                                                \
        if (WQ % NT == 0 || idx_ < WQ)                                                                                 \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wr[(tap_) % WD][i_]) : "v"(Wp + (int64_t)(tap_) * C * C + idx_ * 4) : "memory"); \
    }
#define CIS_WSTASH(tap_)                                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < WIT; ++i_) {                                                               \
        const int idx_ = tid + NT * i_;                                                                                \
        if (WQ % NT == 0 || idx_ < WQ) *reinterpret_cast<f32x4*>(Bs + ((tap_) & 1) * (C * C) + idx_ * 4) = wr[(tap_) % WD][i_]; \
    }
#pragma unroll
    for (int t = 0; t < WD; ++t) { CIS_WFETCH(t) }
    // ---- A window: positions q0 - HALO ... q0 + BM + HALO; every load of the thread in flight before the first LDS write ----
    constexpr int AIT = (NPOS * C4 + NT - 1) / NT;
    {
        float4 v[AIT];
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int idx = tid + NT * i;
            const int j = idx / C4, f = idx % C4;
            const int64_t q = q0 - HALO + j;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NPOS * C4 && q >= 0 && q < Qtot) {
                const int img = (int)(q / IMGQ), rem = (int)(q - (int64_t)img * IMGQ);
                const int r = rem / P, c = rem - r * P;
                if (r < H && c < W) v[i] = *reinterpret_cast<const float4*>(in + (((int64_t)img * H + r) * W + c) * C + f * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int idx = tid + NT * i;
            if (idx < NPOS * C4) {
                const int j = idx / C4, f = idx % C4;
                float2* dst = reinterpret_cast<float2*>(As + j * LDC + f * 4);
                dst[0] = make_float2(v[i].x, v[i].y);
                dst[1] = make_float2(v[i].z, v[i].w);
            }
        }
    }
    wait_vmcnt(0);
    CIS_WSTASH(0)
    __syncthreads();
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* Abase = As + (HALO + wm * WM * 32 + l31) * LDC + 2 * h;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap + WD < 9) { CIS_WFETCH(tap + WD) }  // its ring slot held this tap's weights, already in LDS
        __builtin_amdgcn_sched_barrier(0);
        const int ky = tap / 3, kx = tap - 3 * ky;
        const float* At = Abase + ((ky - 1) * P + (kx - 1)) * LDC;
        const float* Bt = Bs + (tap & 1) * (C * C) + (h * C + wn * WN * 32 + l31) * 2;
        float2 a[2][WM], b[2][WN];
        auto operands = [&](int kk, float2(&av)[WM], float2(&bw)[WN]) {
#pragma unroll
            for (int i = 0; i < WM; ++i) av[i] = *reinterpret_cast<const float2*>(At + i * 32 * LDC + kk);
#pragma unroll
            for (int j = 0; j < WN; ++j) bw[j] = *reinterpret_cast<const float2*>(Bt + ((kk / 2) * C + j * 32) * 2);
        };
        operands(0, a[0], b[0]);
#pragma unroll
        for (int s = 0; s < C / 4; ++s) {
            if (s + 1 < C / 4) operands((s + 1) * 4, a[(s + 1) & 1], b[(s + 1) & 1]);  // next step's reads before this step's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i].x, b[s & 1][j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i].y, b[s & 1][j].y, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < 9) {
            // loads return in order: tap + 1 has landed once only the younger taps' loads are outstanding
            wait_vmcnt(WIT * ((tap + WD < 8 ? tap + WD : 8) - (tap + 1)));
            CIS_WSTASH(tap + 1)  // the other buffer: its readers finished before the previous barrier
            __syncthreads();
        }
    }
    // ---- epilogue: col = lane&31 (oc), row = (r&3) + 8*(r>>2) + 4*h (position); residual loads issued together ----
    float bv[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) bv[j] = bias[(wn * WN + j) * 32 + l31];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        int64_t pp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t q = q0 + (wm * WM + i) * 32 + row;
            pp[r] = -1;
            if (q < Qtot) {
                const int img = (int)(q / IMGQ), rem = (int)(q - (int64_t)img * IMGQ);
                const int y = rem / P, x = rem - y * P;
                if (y < H && x < W) pp[r] = ((int64_t)img * H + y) * W + x;
            }
        }
        if (res) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int oc = (wn * WN + j) * 32 + l31;
                if (oc < resC) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = pp[r] >= 0 ? res[pp[r] * resC + oc] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += bv[j] + rv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += bv[j];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += bv[j];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (pp[r] < 0) continue;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                float v = acc[i][j][r];
                if (relu) v = v > 0.f ? v : 0.f;
                out[pp[r] * C + (wn * WN + j) * 32 + l31] = v;
            }
        }
    }
}

#undef CIS_WFETCH
#undef CIS_WSTASH

template <int C, int H, int W, int WM, int NWM, int NWN>
static int launch_conv3x3_direct(const float* in, const float* w, const float* b, float* out, int N, const float* res, int resC,
                                 int relu, hipStream_t st) {
    constexpr int P = W + 1, BM = NWM * WM * 32, NPOS = BM + 2 * (P + 1);
    constexpr size_t lds = ((size_t)NPOS * (C + 2) + 2 * C * C) * sizeof(float);
    // the dynamic-LDS attribute is a per-device property of the function: one bit per device, set by whichever thread launches there
    // first (setting it twice is harmless, so the race of two first launches needs no lock)
    static std::atomic<uint64_t> attr_set{0};
    auto kern = k_conv3x3_direct<C, H, W, WM, NWM, NWN>;
    int dev = 0;
    CIS_CHECK_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        CIS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    const int64_t Qtot = (int64_t)N * (H + 1) * P;
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(Qtot, BM)), dim3(NWM * NWN * 64), lds, st, in, w, b, out, N, res, resC, relu);
    return CIS_OK;
}

// max pooling 3x3 stride 2 on NHWC, caffe output size ceil((H-3)/2)+1, windows clipped to the input
__global__ void k_maxpool_nhwc(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C, int OH,
                               int OW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * OH * OW * C;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int ox = (int)((i / C) % OW);
    const int oy = (int)((i / ((int64_t)C * OW)) % OH);
    const int64_t n = i / ((int64_t)C * OW * OH);
    float m = -3.402823466e38f;
    for (int dy = 0; dy < 3; ++dy) {
        const int y = oy * 2 + dy;
        if (y >= H) break;
        for (int dx = 0; dx < 3; ++dx) {
            const int x = ox * 2 + dx;
            if (x >= W) break;
            const float v = in[((n * H + y) * W + x) * C + c];
            m = v > m ? v : m;
        }
    }
    out[i] = m;
}

// the same, four channels per thread (C % 4 == 0): 16-byte loads and stores
__global__ void k_maxpool_nhwc_v4(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C, int OH,
                                  int OW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    if (i >= total) return;
    const int c = (int)(i % C4) * 4;
    const int ox = (int)((i / C4) % OW);
    const int oy = (int)((i / ((int64_t)C4 * OW)) % OH);
    const int64_t n = i / ((int64_t)C4 * OW * OH);
    float4 m = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = oy * 2 + dy;
        if (y >= H) break;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int x = ox * 2 + dx;
            if (x >= W) break;
            const float4 v = *reinterpret_cast<const float4*>(in + ((n * H + y) * W + x) * C + c);
            m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y; m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
        }
    }
    *reinterpret_cast<float4*>(out + (((n * OH + oy) * OW + ox) * C + c)) = m;
}

// LRN across channels on NHWC: b = a / (1 + alpha/size * sum_{|c'-c| <= size/2} a^2)^beta
__global__ void k_lrn_nhwc(const float* __restrict__ in, float* __restrict__ out, int64_t npix, int C, int size,
                           float alpha, float beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * C) return;
    const int c = (int)(i % C);
    const float* px = in + (i - c);
    const int half = size / 2;
    float s = 0.f;
    for (int cc = (c - half < 0 ? 0 : c - half); cc <= (c + half >= C ? C - 1 : c + half); ++cc) s = fmaf(px[cc], px[cc], s);
    out[i] = px[c] * powf(1.0f + (alpha / (float)size) * s, -beta);
}

// max pool 3x3/2 (as above) followed by the across-channel LRN, one block per output pixel, one thread per
// channel: the pooled pixel goes through LDS, so the intermediate blob is never written
__global__ void k_maxpool_lrn_nhwc(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C,
                                   int OH, int OW, int size, float alpha, float beta) {
    extern __shared__ float sp[];  // [C]
    const int64_t pix = blockIdx.x;
    const int ox = (int)(pix % OW);
    const int oy = (int)((pix / OW) % OH);
    const int64_t n = pix / ((int64_t)OW * OH);
    const int half = size / 2;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float m = -3.402823466e38f;
        for (int dy = 0; dy < 3; ++dy) {
            const int y = oy * 2 + dy;
            if (y >= H) break;
            for (int dx = 0; dx < 3; ++dx) {
                const int x = ox * 2 + dx;
                if (x >= W) break;
                const float v = in[((n * H + y) * W + x) * C + c];
                m = v > m ? v : m;
            }
        }
        sp[c] = m;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int cc = (c - half < 0 ? 0 : c - half); cc <= (c + half >= C ? C - 1 : c + half); ++cc) s = fmaf(sp[cc], sp[cc], s);
        out[pix * C + c] = sp[c] * powf(1.0f + (alpha / (float)size) * s, -beta);
    }
}

// The same with four channels per thread (16-byte loads and stores) and several output pixels per block;
// x^-0.75 as rsqrt(x) * rsqrt(sqrt(x)) (the network's beta), powf otherwise.
__global__ __launch_bounds__(256) void k_maxpool_lrn_nhwc_v4(const float* __restrict__ in, float* __restrict__ out, int N, int H,
                                                             int W, int C, int OH, int OW, int size, float alpha, float beta,
                                                             int ppb /* pixels per block */, int xcd_remap) {
    extern __shared__ float sp[];  // [ppb][C]
    const int tpp = C >> 2;        // threads per pixel
    const int lp = threadIdx.x / tpp, lc = (threadIdx.x - lp * tpp) << 2;
    const int64_t total = (int64_t)N * OH * OW;
    // Workgroups go to the 8 XCDs round-robin and every XCD has its own L2: with blockIdx.x as the tile number, the output rows that
    // share an input row (3 x 3 windows, stride 2) sit on different XCDs and every input row is fetched 2.25 times.  The grid is a
    // multiple of 8; XCD x works on the x-th eighth of the tiles, consecutive tiles in consecutive slots.
    const unsigned per_xcd = gridDim.x >> 3;
    const int64_t tile = xcd_remap ? (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    const int64_t pix = tile * ppb + lp;
    const bool on = lp < ppb && pix < total;
    if (on) {
        const int ox = (int)(pix % OW);
        const int oy = (int)((pix / OW) % OH);
        const int64_t n = pix / ((int64_t)OW * OH);
        float4 m = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int y = oy * 2 + dy;
            if (y >= H) break;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int x = ox * 2 + dx;
                if (x >= W) break;
                const float4 v = *reinterpret_cast<const float4*>(in + ((n * H + y) * W + x) * C + lc);
                m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;
                m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
            }
        }
        *reinterpret_cast<float4*>(sp + lp * C + lc) = m;
    }
    __syncthreads();
    if (!on) return;
    const int half = size / 2;
    const float* row = sp + lp * C;
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lc + i;
        float s = 0.f;
        for (int cc = (c - half < 0 ? 0 : c - half); cc <= (c + half >= C ? C - 1 : c + half); ++cc) s = fmaf(row[cc], row[cc], s);
        const float base = 1.0f + (alpha / (float)size) * s;
        const float sc = (beta == 0.75f) ? rsqrtf(base) * rsqrtf(sqrtf(base)) : powf(base, -beta);
        r[i] = row[c] * sc;
    }
    *reinterpret_cast<float4*>(out + pix * C + lc) = make_float4(r[0], r[1], r[2], r[3]);
}

// NCHW [n][3][H][W] -> NHWC rows of `pitch` floats (W*3 rounded up to a multiple of 4: every row starts 16-byte aligned)
__global__ void k_nchw3_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int64_t n_rows /* n*H */, int H, int W, int pitch) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, x)
    if (i >= n_rows * W) return;
    const int x = (int)(i % W);
    const int64_t row = i / W;
    const int64_t img = row / H;
    const int y = (int)(row - img * H);
    const float* p = in + (img * 3 * H + y) * W + x;
    float* o = out + row * pitch + x * 3;
    o[0] = p[0]; o[1] = p[(int64_t)H * W]; o[2] = p[(int64_t)2 * H * W];
    // the pad floats of the row and the 16 floats behind the last row are READ by the first convolution's 36-float runs (against
    // zero weights): they must be finite whatever the memory held before -- NaN bytes left there by an earlier owner of the memory
    // gave 0 * NaN = NaN, clamped to 0 by the ReLU: descriptors off by 0.5 % (found in round 4, tests: CIS_CNN_POISON)
    if (x == W - 1)
        for (int k = W * 3; k < pitch; ++k) out[row * pitch + k] = 0.f;
    if (i == 0)
        for (int k = 0; k < 16; ++k) out[n_rows * pitch + k] = 0.f;
}

// ---- dlib face ResNet helpers (NHWC) ---------------------------------------------------------------
// input_rgb_image_sized: (pixel - mean_rgb) / 256
__global__ void k_normalize_rgb(const float* __restrict__ in, float* __restrict__ out, int64_t npix, float m0, float m1, float m2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 3) return;
    const int c = (int)(i % 3);
    out[i] = (in[i] - (c == 0 ? m0 : (c == 1 ? m1 : m2))) * (1.0f / 256.0f);
}

// the same into 4-float pixels (R, G, B, 0): every pixel is one aligned float4, so the first convolution's kernel rows are
// runs of KW float4 (row-run gather of k_conv_igemm)
__global__ void k_normalize_rgb4(const float* __restrict__ in, float* __restrict__ out, int64_t npix, float m0, float m1, float m2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float r = in[i * 3], g = in[i * 3 + 1], b = in[i * 3 + 2];
    *reinterpret_cast<float4*>(out + i * 4) = make_float4((r - m0) * (1.0f / 256.0f), (g - m1) * (1.0f / 256.0f), (b - m2) * (1.0f / 256.0f), 0.f);
}

// The dlib net's first layer as a direct convolution (round 3): 7 x 7, stride 2, 150 x 150 x 3 -> 72 x 72 x 32, normalisation of the
// raw RGB chip, bias (dlib's affine layer is folded into the weights) and ReLU in the same kernel.
//   * GEMM view: M = 32 consecutive output pixels of one image (p = oy * 72 + ox; 72 * 72 = 162 tiles exactly), N = the 32 output
//     channels, K = 7 * 7 * 3 = 147 (+1 zero) = 74 steps of v_mfma_f32_32x32x2_f32.
//   * The WEIGHTS are the B operand and stay in registers for the whole kernel: lane l holds W[k = 2 s + (l >> 5)][oc = l & 31] for
//     every step s -- 74 registers, loaded once.
//   * A workgroup (four waves = four tiles = 128 pixels) stages the input rows its pixels touch -- at most 11 rows of 150 x 3 floats,
//     normalised on the way in ((px - mean) / 256) -- in LDS as they are in memory ([row][x][c]: 450 floats per row).  The A operand of
//     lane (pixel m, k half) for step s is window[base_m + (k / 21) * 450 + k % 21], k = 2 s + half: base_m is one register, the
//     rest an immediate offset of the ds_read (the odd k of a pair sits one float further, 430 further where a kernel row ends).
//     Pixels are 6 floats apart (stride 2 x 3 channels): the 32 lanes of a read group fall on 16 banks, a 2-way conflict, against 64
//     cycles of matrix pipe per step.
// k_conv_igemm ran this layer on (R, G, B, 0) pixels with K = 196 in 254 us per 256 chips (0.31 of the f32 MFMA peak counting the real
// 147-term products) behind a 25 us normalisation pass.
template <int TPW /* tiles per wave: the staged window and the weight registers serve 4 TPW tiles */>
__global__ __launch_bounds__(256) void k_conv7x7s2_direct(const float* __restrict__ in /* [n][150][150][3] raw RGB */,
                                                          const float* __restrict__ wpk /* [196][32]: k = ky * 28 + kx * 4 + c */,
                                                          const float* __restrict__ bias /* [32] */, float* __restrict__ out /* [n][72][72][32] */,
                                                          int n, float m0, float m1, float m2) {
    constexpr int IW = 150, OW = 72, ROWF = IW * 3, NT = OW * OW / 32;  // 162 tiles per image
    constexpr int KSTEPS = 74;
    constexpr int GT = 4 * TPW;                                  // tiles per workgroup
    constexpr int GROUPS = (NT + GT - 1) / GT;                   // workgroups per image
    constexpr int MAXROWS = 2 * ((GT * 32 + OW - 1) / OW) + 7;   // input rows a group's pixels can touch
    __shared__ float win[MAXROWS * ROWF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.x / GROUPS, grp = blockIdx.x % GROUPS;
    const int tile0 = grp * GT;
    const int ntile = (NT - tile0 < GT) ? (NT - tile0) : GT;
    const int p0 = tile0 * 32, p1 = p0 + ntile * 32 - 1;
    const int oy0 = p0 / OW, oy1 = p1 / OW;
    const int nrows = 2 * (oy1 - oy0) + 7;  // input rows 2 oy0 .. 2 oy1 + 6
    // weights -> registers (B operand)
    float w[KSTEPS];
    {
        const int oc = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int s2 = 0; s2 < KSTEPS; ++s2) {
            const int k = 2 * s2 + kh;             // (ky, kx, c) = (k / 21, (k % 21) / 3, k % 3)
            const int ky = k / 21, rem = k - ky * 21, kx = rem / 3, cc = rem - kx * 3;
            w[s2] = k < 147 ? wpk[(ky * 28 + kx * 4 + cc) * 32 + oc] : 0.f;
        }
    }
    // window: rows 2 oy0 ... of the raw chip, normalised
    {
        const float* src = in + ((int64_t)img * IW + 2 * oy0) * ROWF;
        const int tot = nrows * ROWF;
        for (int e = tid; e < tot; e += 256) {
            const int cc = e % 3;
            const float mean = cc == 0 ? m0 : (cc == 1 ? m1 : m2);
            win[e] = (src[e] - mean) * (1.0f / 256.0f);
        }
    }
    __syncthreads();
    const float bv = bias[lane & 31];
    for (int tw = wave; tw < ntile; tw += 4) {
        const int p = p0 + tw * 32 + (lane & 31);
        const int oy = p / OW, ox = p - oy * OW;
        const float* a0 = win + (2 * (oy - oy0)) * ROWF + 6 * ox + (lane >> 5);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < KSTEPS; ++s2) {
            const int k = 2 * s2;
            const int off = (k / 21) * ROWF + (k % 21);            // folds to an immediate
            const bool row_end = (k % 21) == 20;                   // the odd element of the pair starts the next kernel row
            float av = row_end ? a0[off + (lane >> 5) * (ROWF - 21)] : a0[off];
            if (k + 1 >= 147) av = (lane >> 5) ? 0.f : av;        // the padding term: its address is past the window (0 x garbage may be NaN)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w[s2], acc, 0, 0, 0);
        }
        // epilogue: bias + ReLU; register q of lane l is pixel (q & 3) + 8 (q >> 2) + 4 (l >> 5), channel l & 31
        float* o = out + ((int64_t)img * OW * OW + p0 + tw * 32) * 32 + (lane & 31);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            const float v = acc[q] + bv;
            o[row * 32] = v > 0.f ? v : 0.f;
        }
    }
}

// First layer + max_pool<3,3,2,2> in one kernel (round 5).  The separate pool pass read the 170 MB the convolution had just written
// (44 us per 256 chips behind a 135 us convolution whose own stores were a third of its time).  A workgroup now owns THREE pooled rows of
// one image = the seven convolution rows 2 Y0 .. 2 Y0 + 6 (504 pixels: sixteen 32-pixel tiles, the last one three quarters used; one
// row in seven is computed by two workgroups), keeps the four accumulators of each wave in registers until every wave has finished with
// the staged input window, writes bias + ReLU'd convolution values OVER the window (7 x 72 x 32 floats = 64.5 KB: two workgroups per
// CU), and pools from there: only the 35 x 35 x 32 pooled map leaves the CU (40 MB per 256 chips instead of 170 MB out + 170 MB in).
// Same products in the same order as k_conv7x7s2_direct, and max is exact: bit-identical to the two-kernel route.
__global__ __launch_bounds__(256) void k_conv7x7s2_pool(const float* __restrict__ in /* [n][150][150][3] raw RGB */,
                                                        const float* __restrict__ wpk /* [196][32]: k = ky * 28 + kx * 4 + c */,
                                                        const float* __restrict__ bias /* [32] */, float* __restrict__ out /* [n][35][35][32] */,
                                                        int n, float m0, float m1, float m2) {
    constexpr int IW = 150, OW = 72, PW = 35, ROWF = IW * 3;
    constexpr int KSTEPS = 74, GROUPS = 12, CONVR = 7;
    __shared__ float buf[CONVR * OW * 32];   // first the input window (<= 19 rows of 450 floats), then the convolution rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.x / GROUPS, grp = blockIdx.x % GROUPS;
    const int Y0 = 3 * grp;
    const int npool = PW - Y0 < 3 ? PW - Y0 : 3;
    const int nconv = 2 * npool + 1, r0 = 2 * Y0;
    const int nrows = 2 * (nconv - 1) + 7;   // input rows 2 r0 .. 2 (r0 + nconv - 1) + 6
    float w[KSTEPS];
    {
        const int oc = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int s2 = 0; s2 < KSTEPS; ++s2) {
            const int k = 2 * s2 + kh;
            const int ky = k / 21, rem = k - ky * 21, kx = rem / 3, cc = rem - kx * 3;
            w[s2] = k < 147 ? wpk[(ky * 28 + kx * 4 + cc) * 32 + oc] : 0.f;
        }
    }
    {
        const float* src = in + ((int64_t)img * IW + 2 * r0) * ROWF;
        const int tot = nrows * ROWF;
        for (int e = tid; e < tot; e += 256) {
            const int cc = e % 3;
            const float mean = cc == 0 ? m0 : (cc == 1 ? m1 : m2);
            buf[e] = (src[e] - mean) * (1.0f / 256.0f);
        }
    }
    __syncthreads();
    const float bv = bias[lane & 31];
    f32x16 acc[4];   // tiles wave, wave + 4, wave + 8, wave + 12
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int p = (wave + 4 * t) * 32 + (lane & 31);   // pixel of the group's 512 (rows past nconv read in-bounds garbage: discarded below)
        const int oy = p / OW, ox = p - oy * OW;
        const float* a0 = buf + (2 * oy) * ROWF + 6 * ox + (lane >> 5);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < KSTEPS; ++s2) {
            const int k = 2 * s2;
            const int off = (k / 21) * ROWF + (k % 21);
            const bool row_end = (k % 21) == 20;
            float av = row_end ? a0[off + (lane >> 5) * (ROWF - 21)] : a0[off];
            if (k + 1 >= 147) av = (lane >> 5) ? 0.f : av;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w[s2], acc[t], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done with the window
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int p = (wave + 4 * t) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            const float v = acc[t][q] + bv;
            if (p < nconv * OW) buf[p * 32 + (lane & 31)] = v > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    const int tot = npool * PW * 32;
    float* o = out + ((int64_t)img * PW + Y0) * PW * 32;
    for (int e = tid; e < tot; e += 256) {
        const int c = e & 31, x = (e >> 5) % PW, y = (e >> 5) / PW;
        const float* q = buf + ((2 * y) * OW + 2 * x) * 32 + c;
        float mx = q[0];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float v = q[(dy * OW + dx) * 32];
                mx = v > mx ? v : mx;
            }
        o[e] = mx;
    }
}

// avg_pool<2,2,2,2>, no padding, output floor((H-2)/2)+1
__global__ void k_avgpool2_nhwc(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C, int OH, int OW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * OH * OW * C) return;
    const int c = (int)(i % C);
    const int ox = (int)((i / C) % OW);
    const int oy = (int)((i / ((int64_t)C * OW)) % OH);
    const int64_t n = i / ((int64_t)C * OW * OH);
    const float* p = in + ((n * H + 2 * oy) * W + 2 * ox) * C + c;
    out[i] = ((p[0] + p[C]) + (p[(int64_t)W * C] + p[(int64_t)W * C + C])) * 0.25f;
}

// add_prev + relu: out = relu(a + b) where a, b are zero-padded (bottom/right, channels) to the larger shape
__global__ void k_add_relu_pad(const float* __restrict__ a, int AH, int AW, int AC, const float* __restrict__ b, int BH, int BW,
                               int BC, float* __restrict__ out, int N, int OH, int OW, int OC) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * OH * OW * OC) return;
    const int c = (int)(i % OC);
    const int x = (int)((i / OC) % OW);
    const int y = (int)((i / ((int64_t)OC * OW)) % OH);
    const int64_t n = i / ((int64_t)OC * OW * OH);
    float v = 0.f;
    if (y < AH && x < AW && c < AC) v += a[((n * AH + y) * AW + x) * AC + c];
    if (y < BH && x < BW && c < BC) v += b[((n * BH + y) * BW + x) * BC + c];
    out[i] = v > 0.f ? v : 0.f;
}

// avg_pool_everything: [N][H*W][C] -> [N][C]
__global__ void k_global_avgpool(const float* __restrict__ in, float* __restrict__ out, int N, int HW, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * C) return;
    const int c = (int)(i % C);
    const int64_t n = i / C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += in[(n * HW + p) * C + c];
    out[i] = s / (float)HW;
}

// The tail of the dlib net in one launch (round 5).  The last residual block works on 4 x 4 x 256 maps: its first convolution
// (3 x 3 / 2, no padding) leaves ONE pixel per chip, so its second convolution (3 x 3, padding 1, on a 1 x 1 map) only ever meets its
// centre tap -- a 256 x 256 product per chip that ran as a K = 2304 implicit GEMM with eight zero taps -- the skip branch is the 2 x 2
// average pool of the block's input, add_prev pads the 1 x 1 branch to 2 x 2, then come ReLU, the global average and fc_no_bias<128>.
// Six launches (convolution + split-K reduction, pool, add, global pool, fc: 51 us of a 1.99 ms forward for 0.1 % of its arithmetic)
// become one: a workgroup takes IMG chips, thread = output channel, the chips' 256-vectors sit in LDS.
//   t1  [n][256]        first convolution's output (bias + ReLU applied)
//   x   [n][4][4][256]  the block's input (skip branch)
//   wb  [9 * 256][256]  second convolution, packed [k][oc], k = (ky * 3 + kx) * 256 + ic: rows 4 * 256 .. 5 * 256 are the centre tap
//   wfc [256][128]
template <int IMG>
__global__ __launch_bounds__(256) void k_dlib_tail(const float* __restrict__ t1, const float* __restrict__ x, const float* __restrict__ wb,
                                                   const float* __restrict__ bb, const float* __restrict__ wfc, float* __restrict__ feats, int n) {
    __shared__ float s_in[IMG][256];
    __shared__ float s_g[IMG][256];
    const int oc = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * IMG;
#pragma unroll
    for (int i = 0; i < IMG; ++i) s_in[i][oc] = i0 + i < n ? t1[(i0 + i) * 256 + oc] : 0.f;
    __syncthreads();
    float acc[IMG];
#pragma unroll
    for (int i = 0; i < IMG; ++i) acc[i] = bb[oc];
    const float* w = wb + (size_t)4 * 256 * 256 + oc;
    for (int c0 = 0; c0 < 256; c0 += 32) {  // 32 weights of the thread's column in flight (8 left every round trip exposed: 20 us per launch)
        float wv[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) wv[c] = w[(size_t)(c0 + c) * 256];
#pragma unroll
        for (int c = 0; c < 32; ++c)
#pragma unroll
            for (int i = 0; i < IMG; ++i) acc[i] = fmaf(s_in[i][c0 + c], wv[c], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < IMG; ++i) {
        float g = 0.f;
        if (i0 + i < n) {
            const float* p = x + ((i0 + i) * 16) * 256 + oc;
            float sum = 0.f;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const float* q = p + ((2 * py) * 4 + 2 * px) * 256;
                    const float sk = ((q[0] + q[256]) + (q[4 * 256] + q[5 * 256])) * 0.25f;   // avg_pool<2,2,2,2> (k_avgpool2_nhwc's order)
                    float v = sk + ((py == 0 && px == 0) ? acc[i] : 0.f);                    // add_prev: the 1 x 1 branch zero-padded to 2 x 2
                    v = v > 0.f ? v : 0.f;
                    sum += v;
                }
            g = sum / 4.0f;                                                                // avg_pool_everything
        }
        s_g[i][oc] = g;
    }
    __syncthreads();
    if (oc < 128) {
        float o[IMG];
#pragma unroll
        for (int i = 0; i < IMG; ++i) o[i] = 0.f;
        for (int c0 = 0; c0 < 256; c0 += 32) {
            float wv[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) wv[c] = wfc[(size_t)(c0 + c) * 128 + oc];
#pragma unroll
            for (int c = 0; c < 32; ++c)
#pragma unroll
                for (int i = 0; i < IMG; ++i) o[i] = fmaf(s_g[i][c0 + c], wv[c], o[i]);
        }
#pragma unroll
        for (int i = 0; i < IMG; ++i)
            if (i0 + i < n) feats[(i0 + i) * 128 + oc] = o[i];
    }
}

// ================================================================================================
// host
// ================================================================================================
struct LayerW {
    float* d_w = nullptr;  // packed [g][K][OCg]
    float* d_b = nullptr;
    float* d_w2 = nullptr; // 3x3 layers with IC == OC == C served by k_conv3x3_direct: [tap][C/2][C][2] (a k pair adjacent)
};

// activations and split-K partial sums of one forward in flight
struct CnnWs {
    DevBuf act0, act1, act2, act3, part;
    void release() { act0.release(); act1.release(); act2.release(); act3.release(); part.release(); }
};
static const int kMaxParts = 4;

// A captured forward (hipGraph) of the dlib network (CIS_CNN_GRAPH=1; see cis_cnn_forward_dev for why it is not the default): ~50 launches
// per chain, two chains per batch of 256 -- on a slow host the enqueue is longer than the kernels.  Keyed by what the launches bake in:
// input, batch size, output, parts, the CIS_CNN_* switches; valid while the workspaces it was captured on are the handle's.
static const int kCnnGraphs = 4;
struct CnnGraph {
    const float* in = nullptr;
    float* out = nullptr;
    int n = 0, parts = 0;
    uint64_t env = 0, used = 0;
    void* wsp[kMaxParts][5] = {};
    hipGraphExec_t exec = nullptr;
    bool seen = false;
};

struct cis_cnn {
    int arch = 0, device = 0;
    LayerW conv[5], fc[2];        // DeepSentibank
    std::vector<LayerW> dl;       // dlib ResNet: conv0, then (a, b) per block, then fc (bias-free); affine layers folded in
    // A batch is cut into up to kMaxParts contiguous parts that run the whole forward concurrently on the handle's own streams,
    // each with its own workspace: a launch of the 256-item batch lasts 60-80 us and spends a fifth of it in the ramp and tail of
    // its workgroup rounds, which the other part's launches fill (dlib, batch 256: 2.16 -> 2.03 ms with two parts).
    CnnWs ws[kMaxParts];
    hipStream_t ps[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_in = nullptr, ev_done[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};
    DevBuf in_buf, out_buf;
    // views (round 5): a handle that shares the weights of its base and owns workspaces, streams and events -- several BATCHES in flight
    // on the caller's streams (consecutive forwards are in different layers at any time and fill each other's workgroup rounds:
    // dlib 0.47-0.52 -> 0.60, DeepSentibank 0.61 -> 0.68 of the f32 MFMA peak at batch 256, profiles/archive/r05b/r05_experiments.txt)
    cis_cnn* base = nullptr;
    std::vector<cis_cnn*> views;
    bool orphaned = false;      // a view whose base was destroyed first: every call fails, nothing dangles
    int parts_override = 0;     // > 0: parts of a batch that run concurrently (views and their base: 1 -- whole batches overlap instead)
    CnnGraph graphs[kCnnGraphs];
    uint64_t graph_clock = 0;
    hipStream_t gs = nullptr;   // capture stream
    bool graph_off = false;     // a capture failed on this handle: launches from here on
};

// dlib anet_type block plan: (in channels, out channels, down-sampling block)
struct DlibBlock { int cin, cout, down; };
static const DlibBlock kDlibBlocks[14] = {{32, 32, 0}, {32, 32, 0}, {32, 32, 0}, {32, 64, 1}, {64, 64, 0}, {64, 64, 0}, {64, 64, 0},
                                          {64, 128, 1}, {128, 128, 0}, {128, 128, 0}, {128, 256, 1}, {256, 256, 0}, {256, 256, 0},
                                          {256, 256, 1}};

static const int kConvCfg[5][5] = {  // OC, kernel, stride, pad, groups   (prototxt :7-16,:47-58,:88-98,:105-116,:123-134)
    {96, 11, 4, 0, 1}, {256, 5, 1, 2, 2}, {384, 3, 1, 1, 1}, {384, 3, 1, 1, 2}, {256, 3, 1, 1, 2}};
static const bool kPoolAfter[5] = {true, true, false, false, true};
static const bool kLrnAfter[5] = {true, true, false, false, false};

extern "C" int cis_cnn_feat_dim(int arch) { return arch == 1 ? 4096 : (arch == 2 ? 128 : 0); }

static std::mutex g_cnn_views_mu;  // guards base <-> view links

extern "C" void cis_cnn_destroy(cis_cnn* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    {
        std::lock_guard<std::mutex> lk(g_cnn_views_mu);
        if (c->base) {  // a view: unlink; the weights are the base's
            auto& v = c->base->views;
            v.erase(std::remove(v.begin(), v.end(), c), v.end());
        }
        for (cis_cnn* v : c->views) { v->orphaned = true; v->base = nullptr; }  // views that outlive their base answer CIS_EINVAL
        c->views.clear();
    }
    if (c->base || c->orphaned) {  // weights are not this handle's to free (an orphan's pointers died with its base)
        for (auto& l : c->conv) l = LayerW();
        for (auto& l : c->fc) l = LayerW();
        c->dl.clear();
    }
    for (auto& l : c->conv) { if (l.d_w) (void)hipFree(l.d_w); if (l.d_b) (void)hipFree(l.d_b); if (l.d_w2) (void)hipFree(l.d_w2); }
    for (auto& l : c->fc) { if (l.d_w) (void)hipFree(l.d_w); if (l.d_b) (void)hipFree(l.d_b); }
    for (auto& l : c->dl) { if (l.d_w) (void)hipFree(l.d_w); if (l.d_b) (void)hipFree(l.d_b); if (l.d_w2) (void)hipFree(l.d_w2); }
    for (auto& g : c->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (c->gs) (void)hipStreamDestroy(c->gs);
    for (auto& w : c->ws) w.release();
    for (auto& s : c->ps) if (s) (void)hipStreamDestroy(s);
    for (auto& e : c->ev_done) if (e) (void)hipEventDestroy(e);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    c->in_buf.release(); c->out_buf.release();
    delete c;
}

static int upload_f(float** dst, const float* src, size_t n) {
    CIS_CHECK_HIP(hipMalloc((void**)dst, n * sizeof(float)));
    CIS_CHECK_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return CIS_OK;
}

// conv OIHW + bias with the following per-channel affine (gamma, beta) folded in, packed [K][OC], k = (ky*KW+kx)*IC+ic
static int pack_conv_affine(LayerW* L, const float* w, const float* b, const float* gam, const float* bet, int OC, int IC, int k,
                            int ICp = 0 /* > IC: input pixels padded to ICp channels (zero weights) */) {
    if (ICp < IC) ICp = IC;
    const int K = k * k * ICp;
    std::vector<float> packed((size_t)K * OC, 0.f), bias(OC);
    for (int o = 0; o < OC; ++o) {
        const float g = gam ? gam[o] : 1.f;
        bias[o] = g * (b ? b[o] : 0.f) + (bet ? bet[o] : 0.f);
        for (int ic = 0; ic < IC; ++ic)
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx)
                    packed[((size_t)(ky * k + kx) * ICp + ic) * OC + o] = g * w[(((size_t)o * IC + ic) * k + ky) * k + kx];
    }
    CIS_TRY(upload_f(&L->d_w, packed.data(), packed.size()));
    CIS_TRY(upload_f(&L->d_b, bias.data(), bias.size()));
    if (k == 3 && IC == OC && ICp == IC && (OC == 32 || OC == 64)) {
        std::vector<float> paired(packed.size());
        for (int t = 0; t < 9; ++t)
            for (int ic = 0; ic < IC; ++ic)
                for (int o = 0; o < OC; ++o)
                    paired[(((size_t)t * (IC / 2) + ic / 2) * OC + o) * 2 + (ic & 1)] = packed[((size_t)t * IC + ic) * OC + o];
        CIS_TRY(upload_f(&L->d_w2, paired.data(), paired.size()));
    }
    return CIS_OK;
}

static int cnn_create_dlib(cis_cnn** out, const float* const* tensors, int n_tensors) {
    CIS_REQUIRE(tensors != nullptr && n_tensors == 117, "the dlib face ResNet needs 117 tensors (conv0 w,b,gamma,beta; 14 blocks x 2 x (w,b,gamma,beta); fc)");
    for (int i = 0; i < 117; ++i) CIS_REQUIRE(tensors[i] != nullptr, "tensor %d is NULL", i);
    CIS_TRY(cis_lazy_init());
    cis_cnn* c = new cis_cnn();
    c->arch = 2;
    c->device = cis_current_device();
    c->dl.resize(1 + 28 + 1);
    int rc = pack_conv_affine(&c->dl[0], tensors[0], tensors[1], tensors[2], tensors[3], 32, 3, 7, 4);  // (R, G, B, 0) pixels
    for (int i = 0; i < 14 && rc == CIS_OK; ++i) {
        const float* const* t = tensors + 4 + 8 * i;
        rc = pack_conv_affine(&c->dl[1 + 2 * i], t[0], t[1], t[2], t[3], kDlibBlocks[i].cout, kDlibBlocks[i].cin, 3);
        if (rc == CIS_OK) rc = pack_conv_affine(&c->dl[2 + 2 * i], t[4], t[5], t[6], t[7], kDlibBlocks[i].cout, kDlibBlocks[i].cout, 3);
    }
    if (rc == CIS_OK) {  // fc_no_bias<128>: [128][256] -> packed [256][128]
        std::vector<float> packed((size_t)256 * 128);
        for (int o = 0; o < 128; ++o)
            for (int k = 0; k < 256; ++k) packed[(size_t)k * 128 + o] = tensors[116][(size_t)o * 256 + k];
        rc = upload_f(&c->dl[29].d_w, packed.data(), packed.size());
    }
    if (rc != CIS_OK) { cis_cnn_destroy(c); return rc; }
    *out = c;
    return CIS_OK;
}

extern "C" int cis_cnn_create(cis_cnn** out, int arch, const float* const* tensors, int n_tensors) {
    CIS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    if (arch == 2) return cnn_create_dlib(out, tensors, n_tensors);
    if (arch != 1) {
        cis_set_error("arch %d: CIS_CNN_SENTIBANK (1) and CIS_CNN_DLIB_RESNET (2) are built", arch);
        return CIS_EUNSUPPORTED;
    }
    CIS_REQUIRE(tensors != nullptr && n_tensors == 14, "DeepSentibank needs 14 tensors (conv1..conv5, fc6, fc7: weight, bias)");
    for (int i = 0; i < 14; ++i) CIS_REQUIRE(tensors[i] != nullptr, "tensor %d is NULL", i);
    CIS_TRY(cis_lazy_init());
    cis_cnn* c = new cis_cnn();
    c->arch = arch;
    c->device = cis_current_device();
    int rc = CIS_OK;
    auto fail = [&](int r) { cis_cnn_destroy(c); return r; };
    int C = 3, hw = 227;
    for (int l = 0; l < 5; ++l) {
        const int OC = kConvCfg[l][0], k = kConvCfg[l][1], s = kConvCfg[l][2], p = kConvCfg[l][3], g = kConvCfg[l][4];
        const int ICg = C / g, OCg = OC / g, K = k * k * ICg;
        const float* w = tensors[2 * l];  // caffe OIHW: [OC][ICg][k][k]
        if (l == 0) {
            // first layer: row-run K order for the vector gather over NHWC rows: k = ky * RUNP + (kx * 3 + ic), RUNP = 36
            // (33 real taps of a kernel row + 3 zero weights), K' = 11 * 36
            const int RUNP = ((k * C + 3) / 4) * 4, Kp = k * RUNP;
            std::vector<float> pr((size_t)Kp * OC, 0.f);
            for (int o = 0; o < OC; ++o)
                for (int ic = 0; ic < C; ++ic)
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx)
                            pr[(size_t)(ky * RUNP + kx * C + ic) * OC + o] = w[(((size_t)o * C + ic) * k + ky) * k + kx];
            if ((rc = upload_f(&c->conv[l].d_w, pr.data(), pr.size())) != CIS_OK) return fail(rc);
            if ((rc = upload_f(&c->conv[l].d_b, tensors[2 * l + 1], OC)) != CIS_OK) return fail(rc);
            // the same layer over the caller's NCHW planes (no re-layout pass): k = (ic * 11 + ky) * 12 + kx, kx = 11 a zero weight
            const int RUNW = ((k + 3) / 4) * 4, Kq = C * k * RUNW;
            std::vector<float> pq((size_t)Kq * OC, 0.f);
            for (int o = 0; o < OC; ++o)
                for (int ic = 0; ic < C; ++ic)
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx)
                            pq[(size_t)((ic * k + ky) * RUNW + kx) * OC + o] = w[(((size_t)o * C + ic) * k + ky) * k + kx];
            if ((rc = upload_f(&c->conv[l].d_w2, pq.data(), pq.size())) != CIS_OK) return fail(rc);
            hw = (hw + 2 * p - k) / s + 1;
            C = OC;
            if (kPoolAfter[l]) hw = (hw - 3 + 1) / 2 + 1;
            continue;
        }
        std::vector<float> packed((size_t)g * K * OCg);
        for (int gi = 0; gi < g; ++gi)
            for (int o = 0; o < OCg; ++o)
                for (int ic = 0; ic < ICg; ++ic)
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx)
                            packed[((size_t)gi * K + (size_t)(ky * k + kx) * ICg + ic) * OCg + o] =
                                w[(((size_t)(gi * OCg + o) * ICg + ic) * k + ky) * k + kx];
        if ((rc = upload_f(&c->conv[l].d_w, packed.data(), packed.size())) != CIS_OK) return fail(rc);
        if ((rc = upload_f(&c->conv[l].d_b, tensors[2 * l + 1], OC)) != CIS_OK) return fail(rc);
        hw = (hw + 2 * p - k) / s + 1;
        C = OC;
        if (kPoolAfter[l]) hw = (hw - 3 + 1) / 2 + 1;  // ceil((hw-3)/2)+1
    }
    // fc6 consumes pool5 flattened CHW (c*36 + y*6 + x) in caffe; ours is HWC -> permute the K axis once
    const int fin6 = C * hw * hw;  // 9216
    {
        const float* w = tensors[10];  // [4096][9216]
        std::vector<float> packed((size_t)fin6 * 4096);
        for (int o = 0; o < 4096; ++o)
            for (int ch = 0; ch < C; ++ch)
                for (int y = 0; y < hw; ++y)
                    for (int x = 0; x < hw; ++x)
                        packed[((size_t)((y * hw + x) * C + ch)) * 4096 + o] = w[(size_t)o * fin6 + (size_t)ch * hw * hw + y * hw + x];
        if ((rc = upload_f(&c->fc[0].d_w, packed.data(), packed.size())) != CIS_OK) return fail(rc);
        if ((rc = upload_f(&c->fc[0].d_b, tensors[11], 4096)) != CIS_OK) return fail(rc);
    }
    {
        const float* w = tensors[12];  // [4096][4096] -> [k][o]
        std::vector<float> packed((size_t)4096 * 4096);
        for (int o = 0; o < 4096; ++o)
            for (int k = 0; k < 4096; ++k) packed[(size_t)k * 4096 + o] = w[(size_t)o * 4096 + k];
        if ((rc = upload_f(&c->fc[1].d_w, packed.data(), packed.size())) != CIS_OK) return fail(rc);
        if ((rc = upload_f(&c->fc[1].d_b, tensors[13], 4096)) != CIS_OK) return fail(rc);
    }
    *out = c;
    return CIS_OK;
}

extern "C" int cis_cnn_create_view(cis_cnn** out, cis_cnn* base) {
    CIS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    CIS_REQUIRE(base != nullptr, "base is NULL");
    CIS_REQUIRE(!base->orphaned, "the base of this handle was destroyed");
    cis_cnn* root = base->base ? base->base : base;  // a view of a view is a view of the base
    cis_cnn* c = new cis_cnn();
    c->arch = root->arch;
    c->device = root->device;
    for (int i = 0; i < 5; ++i) c->conv[i] = root->conv[i];
    for (int i = 0; i < 2; ++i) c->fc[i] = root->fc[i];
    c->dl = root->dl;
    c->base = root;
    c->parts_override = 1;
    {
        std::lock_guard<std::mutex> lk(g_cnn_views_mu);
        root->views.push_back(c);
        root->parts_override = 1;  // batches overlap through the views from now on, not parts of one batch
    }
    // the base's part streams are idle from now on: give their hardware queues back (see cis_cnn_forward_dev)
    (void)hipSetDevice(root->device);
    for (int p = 0; p < kMaxParts; ++p)
        if (root->ps[p]) {
            (void)hipStreamSynchronize(root->ps[p]);
            (void)hipStreamDestroy(root->ps[p]);
            root->ps[p] = nullptr;
        }
    *out = c;
    return CIS_OK;
}

// out = relu?(bias + part[0] + part[1] + ...) in that order (deterministic), four outputs per thread
// pool_w > 0: the residual branch is avg_pool<2,2,2,2> of a [N][2 OH'..][pool_w][resC] map taken on the fly (dlib's down blocks: the
// pooled skip was a launch and a round trip of its own) -- pool_oh / pool_ow = the OUTPUT's spatial size, pool_h / pool_w the input's
__global__ void k_splitk_reduce(const float* __restrict__ part, int splitk, int64_t part_stride, const float* __restrict__ bias,
                                float* __restrict__ out, int64_t n4, int OC, int relu, const float* __restrict__ res, int resC,
                                int pool_h = 0, int pool_w = 0, int pool_oh = 0, int pool_ow = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int oc = (int)((i * 4) % OC);
    float4 a = bias ? *reinterpret_cast<const float4*>(bias + oc) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < splitk; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)s * part_stride + i * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (res && oc < resC && pool_w > 0) {
        const int64_t pix = (i * 4) / OC;
        const int ox = (int)(pix % pool_ow), oy = (int)((pix / pool_ow) % pool_oh);
        const int64_t nn = pix / ((int64_t)pool_ow * pool_oh);
        const float* p = res + ((nn * pool_h + 2 * oy) * pool_w + 2 * ox) * resC + oc;
        const float4 p0 = *reinterpret_cast<const float4*>(p), p1 = *reinterpret_cast<const float4*>(p + resC);
        const float4 p2 = *reinterpret_cast<const float4*>(p + (int64_t)pool_w * resC), p3 = *reinterpret_cast<const float4*>(p + (int64_t)pool_w * resC + resC);
        a.x += ((p0.x + p1.x) + (p2.x + p3.x)) * 0.25f; a.y += ((p0.y + p1.y) + (p2.y + p3.y)) * 0.25f;
        a.z += ((p0.z + p1.z) + (p2.z + p3.z)) * 0.25f; a.w += ((p0.w + p1.w) + (p2.w + p3.w)) * 0.25f;
    } else
    if (res && oc < resC) {  // residual branch (resC % 4 == 0 channels, zero-padded above)
        const float4 v = *reinterpret_cast<const float4*>(res + ((i * 4) / OC) * resC + oc);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (relu) { a.x = a.x > 0.f ? a.x : 0.f; a.y = a.y > 0.f ? a.y : 0.f; a.z = a.z > 0.f ? a.z : 0.f; a.w = a.w > 0.f ? a.w : 0.f; }
    *reinterpret_cast<float4*>(out + i * 4) = a;
}

template <int WM, int WN, int WAVES_M, int WAVES_N>
static void launch_conv_cfg(const ConvDesc& d, const float* in, const float* w, const float* b, float* out, hipStream_t st) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    const int64_t npix = (int64_t)d.N * d.OH * d.OW;
    dim3 g((unsigned)ceil_div(npix, BM), (unsigned)ceil_div(d.OCg, BN), (unsigned)(d.splitk > 1 ? d.splitk : d.groups));
    const bool vec = (d.sC == 1 && d.ICg % 16 == 0 && d.C % 4 == 0) || d.runq > 0;
    if (d.runq > 0 && d.run_planes > 0) {
        if constexpr (WM == 1 && WN == 3) hipLaunchKernelGGL((k_conv_igemm<WM, WN, WAVES_M, WAVES_N, 3>), g, dim3(256), 0, st, in, w, b, out, d);
        else cis_set_error("plane-run gather: built for the 128 x 96 tile only");
    } else if (d.runq > 0) hipLaunchKernelGGL((k_conv_igemm<WM, WN, WAVES_M, WAVES_N, 2>), g, dim3(256), 0, st, in, w, b, out, d);
    else if (vec) hipLaunchKernelGGL((k_conv_igemm<WM, WN, WAVES_M, WAVES_N, 1>), g, dim3(256), 0, st, in, w, b, out, d);
    else hipLaunchKernelGGL((k_conv_igemm<WM, WN, WAVES_M, WAVES_N, 0>), g, dim3(256), 0, st, in, w, b, out, d);
}

static void launch_conv(const ConvDesc& d, const float* in, const float* w, const float* b, float* out, hipStream_t st) {
    const int64_t npix = (int64_t)d.N * d.OH * d.OW;
    const int z = d.splitk > 1 ? d.splitk : d.groups;
    auto blocks = [&](int bm, int bn) { return ceil_div(npix, bm) * ceil_div(d.OCg, bn) * z; };
    const int64_t want = 2 * 256;  // at least two workgroups per CU, else a smaller tile
    if (d.OCg <= 32) launch_conv_cfg<1, 1, 4, 1>(d, in, w, b, out, st);                // 128 x 32 tiles
    else if (npix <= 2048) {
        // fc layers.  32 x 128 tiles fill the chip at any batch, but every 32 rows stream the whole weight matrix again (batch 256: 8 x
        // 151 MB for fc6); from 64 rows up 64 x 128 tiles halve that (the tile shape does not change a sum's order: same bits)
        const char* e = getenv("CIS_CNN_FC_TILE");
        const int t = e ? atoi(e) : (npix >= 64 ? 1 : 0);
        if (t == 1) launch_conv_cfg<1, 2, 2, 2>(d, in, w, b, out, st);
        else if (t == 2) launch_conv_cfg<2, 2, 2, 2>(d, in, w, b, out, st);
        else if (t == 3) launch_conv_cfg<1, 1, 2, 2>(d, in, w, b, out, st);
        else if (t == 4) launch_conv_cfg<2, 1, 2, 2>(d, in, w, b, out, st);
        else launch_conv_cfg<1, 1, 1, 4>(d, in, w, b, out, st);
    }
    else if (d.OCg <= 64) {
        if (blocks(128, 64) >= want) launch_conv_cfg<2, 1, 2, 2>(d, in, w, b, out, st);   // 128 x 64 tiles
        else launch_conv_cfg<1, 1, 2, 2>(d, in, w, b, out, st);                           // 64 x 64
    } else if (d.OCg % 96 == 0 && d.OCg % 128 != 0) launch_conv_cfg<1, 3, 4, 1>(d, in, w, b, out, st);  // 96, 192: 128 x 96
    else if (blocks(128, 128) >= want) launch_conv_cfg<2, 2, 2, 2>(d, in, w, b, out, st);  // 128 x 128
    else if (blocks(128, 64) >= want) launch_conv_cfg<2, 1, 2, 2>(d, in, w, b, out, st);   // 128 x 64
    else launch_conv_cfg<1, 1, 2, 2>(d, in, w, b, out, st);                               // 64 x 64
}

static ConvDesc nhwc_conv(int n, int H, int W, int C, int OC, int k, int stride, int pad, int relu) {
    ConvDesc d;
    d.N = n; d.H = H; d.W = W; d.C = C; d.OC = OC; d.KH = d.KW = k; d.stride = stride; d.pad = pad; d.groups = 1;
    d.OH = (H + 2 * pad - k) / stride + 1;
    d.OW = (W + 2 * pad - k) / stride + 1;
    d.ICg = C; d.OCg = OC; d.K = k * k * C;
    d.sN = (int64_t)H * W * C; d.sC = 1; d.sH = (int64_t)W * C; d.sW = C;
    d.relu = relu; d.kx_fastest = 0;
    return d;
}

// A convolution whose output has too few tiles to fill the chip (the deep, spatially small layers of the dlib net: 16-256
// workgroups walking K = 1152 ... 2304 serially) splits K over blockIdx.z and adds the partial sums in a fixed order,
// together with bias, residual branch and ReLU (k_splitk_reduce).
static int conv_fill_chip(CnnWs* ws, ConvDesc d, const float* in, const LayerW& L, float* out, hipStream_t st) {
    const float* w = L.d_w;
    const float* b = L.d_b;
    const int64_t npix = (int64_t)d.N * d.OH * d.OW;
    // the 3x3 / stride 1 / pad 1 layers with as many output as input channels at the two large spatial sizes: direct kernel
    const bool no_direct = getenv("CIS_CNN_NO_DIRECT") != nullptr;  // read per call: A/B runs in one process
    if (!no_direct && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.groups == 1 && d.C == d.OC && d.sC == 1 && b != nullptr && L.d_w2 != nullptr &&
        d.runq == 0 && d.splitk == 1) {
        const char* cfg_s = getenv("CIS_CNN_DIRECT_CFG");  // tile sweep (tools/check_dlib_direct.py)
        const int cfg = cfg_s ? atoi(cfg_s) : 0;
#define CIS_DIRECT(C_, H_, W_, WM_, NWM_, NWN_) return launch_conv3x3_direct<C_, H_, W_, WM_, NWM_, NWN_>(in, L.d_w2, b, out, d.N, d.res, d.resC, d.relu, st)
        if (d.C == 32 && d.H == 35 && d.W == 35) {
            if (cfg == 1) CIS_DIRECT(32, 35, 35, 1, 8, 1);
            if (cfg == 2) CIS_DIRECT(32, 35, 35, 2, 4, 1);
            CIS_DIRECT(32, 35, 35, 1, 4, 1);
        }
        if (d.C == 64 && d.H == 17 && d.W == 17) {
            if (cfg == 1) CIS_DIRECT(64, 17, 17, 1, 2, 2);
            if (cfg == 2) CIS_DIRECT(64, 17, 17, 1, 4, 1);
            CIS_DIRECT(64, 17, 17, 1, 4, 2);
        }
#undef CIS_DIRECT
    }
    const bool vec = d.sC == 1 && d.ICg % 16 == 0 && d.C % 4 == 0 && d.groups == 1 && d.OC % 4 == 0 && d.runq == 0;
    const int64_t tiles = npix <= 2048 ? ceil_div(npix, 32) * ceil_div((int64_t)d.OCg, 128) : ceil_div(npix, 64) * ceil_div((int64_t)d.OCg, 64);
    const int nkt = d.K / 16;
    int splitk = 1;
    static const bool off = getenv("CIS_CNN_NO_SPLITK") != nullptr;
    while (!off && vec && d.OCg > 32 && tiles * splitk < 512 && splitk < 32 && nkt / (splitk * 2) >= 6) splitk *= 2;
    // the 128- and 256-channel maps (8x8 ... 3x3 per chip): 128 x 128 tiles (the kernel's most efficient shape) with K split until
    // the chip is full, instead of 64 x 64 tiles: 70 -> 66 us and 74 -> 64 us per layer at batch 256 (CIS_CNN_NO_BIGTILE: old choice)
    if (!off && vec && !getenv("CIS_CNN_NO_BIGTILE") && d.OCg >= 128 && d.OCg % 128 == 0 && npix > 2048) {
        const int64_t t128 = ceil_div(npix, 128) * (d.OCg / 128);
        const char* te = getenv("CIS_CNN_SPLIT_TARGET");
        const int64_t target = te ? atoi(te) : 512;
        int s = 1;
        while (t128 * s < target && s < 32 && nkt / (s * 2) >= 6) s *= 2;
        if (t128 * s >= 512) splitk = s;
    }
    if (splitk == 1) {
        launch_conv(d, in, w, b, out, st);
        return CIS_OK;
    }
    CIS_TRY(ws->part.reserve((size_t)splitk * npix * d.OC * sizeof(float)));
    float* part = ws->part.as<float>();
    ConvDesc dp = d;
    dp.splitk = splitk;
    dp.part_stride = npix * d.OC;
    dp.res = nullptr; dp.resC = 0;
    launch_conv(dp, in, w, nullptr, part, st);
    const int64_t n4 = npix * d.OC / 4;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, st, part, splitk, dp.part_stride, b, out, n4, d.OC,
                       d.relu, d.res, d.resC);
    return CIS_OK;
}

// Test hook (tests/test_cnn_hip_parity.py): CIS_CNN_POISON=<mask> fills the workspaces (bit 0..4: act0, act1, act2, act3, part) with
// 0xff bytes (NaN) before a forward.  A forward must not read what it has not written: results do not change under any mask.
static int cnn_poison(CnnWs* ws, hipStream_t st) {
    const char* e = getenv("CIS_CNN_POISON");
    if (!e) return CIS_OK;
    const int mask = atoi(e);
    DevBuf* b[5] = {&ws->act0, &ws->act1, &ws->act2, &ws->act3, &ws->part};
    for (int i = 0; i < 5; ++i)
        if (((mask >> i) & 1) && b[i]->p) CIS_CHECK_HIP(hipMemsetAsync(b[i]->p, 0xff, b[i]->cap, st));
    return CIS_OK;
}

// dlib face ResNet: d_in = [n][150][150][3] float32 RGB 0..255 (aligned chips), d_feats = [n][128]
static int cnn_forward_dlib(cis_cnn* c, CnnWs* ws, const float* d_in, int n, float* d_feats, hipStream_t st) {
    const size_t big = (size_t)n * 72 * 72 * 32;  // largest activation: first convolution's output
    CIS_TRY(ws->act0.reserve(big * sizeof(float)));
    CIS_TRY(ws->act1.reserve(big * sizeof(float)));
    CIS_TRY(ws->act2.reserve(big * sizeof(float)));
    CIS_TRY(ws->act3.reserve(big * sizeof(float)));
    CIS_TRY(cnn_poison(ws, st));
    float* A = ws->act0.as<float>();
    float* B = ws->act1.as<float>();
    float* T1 = ws->act2.as<float>();
    float* T2 = ws->act3.as<float>();
    auto grid = [](int64_t total) { return dim3((unsigned)ceil_div(total, 256)); };
    bool pooled = false;
    if (!getenv("CIS_CNN_NO_DIRECT7") && !getenv("CIS_CNN_NO_POOL7")) {
        // first layer AND the max pool behind it in one kernel (k_conv7x7s2_pool, round 5): the pooled 35 x 35 x 32 map goes to A
        hipLaunchKernelGGL(k_conv7x7s2_pool, dim3((unsigned)n * 12), dim3(256), 0, st, d_in, (const float*)c->dl[0].d_w, (const float*)c->dl[0].d_b, A, n,
                           122.782f, 117.001f, 104.298f);
        pooled = true;
    } else
    if (!getenv("CIS_CNN_NO_DIRECT7")) {
        // first layer: direct 7 x 7 / 2 convolution, normalisation + bias + ReLU fused (k_conv7x7s2_direct)
        // tiles per wave (the staged window and the weight registers serve 4 x that many tiles): 1 / 2 / 4 -> 161 / 141 / 133 us per 256 chips, forward 1.97 / 1.92 / 1.90 ms (profiles/r03r_dlib_first_layer.txt)
        const int tpw = getenv("CIS_CNN_D7_TPW") ? atoi(getenv("CIS_CNN_D7_TPW")) : 4;
        if (tpw == 1)
            hipLaunchKernelGGL(k_conv7x7s2_direct<1>, dim3((unsigned)n * 41), dim3(256), 0, st, d_in, (const float*)c->dl[0].d_w,
                               (const float*)c->dl[0].d_b, B, n, 122.782f, 117.001f, 104.298f);
        else if (tpw == 4)
            hipLaunchKernelGGL(k_conv7x7s2_direct<4>, dim3((unsigned)n * 11), dim3(256), 0, st, d_in, (const float*)c->dl[0].d_w,
                               (const float*)c->dl[0].d_b, B, n, 122.782f, 117.001f, 104.298f);
        else
            hipLaunchKernelGGL(k_conv7x7s2_direct<2>, dim3((unsigned)n * 21), dim3(256), 0, st, d_in, (const float*)c->dl[0].d_w,
                               (const float*)c->dl[0].d_b, B, n, 122.782f, 117.001f, 104.298f);
    } else {
        hipLaunchKernelGGL(k_normalize_rgb4, grid((int64_t)n * 150 * 150), dim3(256), 0, st, d_in, A, (int64_t)n * 150 * 150, 122.782f,
                           117.001f, 104.298f);
        ConvDesc d0 = nhwc_conv(n, 150, 150, 3, 32, 7, 2, 0, 1);
        // (R, G, B, 0) pixels: a kernel row is a run of 7 aligned float4, k = ky * 28 + kx * 4 + c (weights packed to match)
        d0.sN = (int64_t)150 * 150 * 4; d0.sH = 150 * 4; d0.sW = 4;
        d0.runq = 7;
        d0.K = 7 * 28;
        launch_conv(d0, A, c->dl[0].d_w, c->dl[0].d_b, B, st);  // 72 x 72 x 32, affine folded, relu
    }
    int H = (72 - 3) / 2 + 1, W = H, C = 32;                  // max_pool<3,3,2,2>: 35
    if (!pooled) hipLaunchKernelGGL(k_maxpool_nhwc_v4, grid((int64_t)n * H * W * C / 4), dim3(256), 0, st, B, A, n, 72, 72, C, H, W);
    float* x = A;      // current activation
    float* other = B;  // free buffer for the next activation
    for (int i = 0; i < 14; ++i) {
        const DlibBlock& b = kDlibBlocks[i];
        const int s = b.down ? 2 : 1, p = b.down ? 0 : 1;
        if (i == 13 && H == 4 && W == 4 && C == 256 && b.cout == 256 && b.down && !getenv("CIS_CNN_NO_TAIL")) {
            // the last block's second convolution, skip branch, add_prev, ReLU, global average and the fully connected layer: one launch
            // (k_dlib_tail; CIS_CNN_NO_TAIL=1 keeps the six launches it replaces)
            ConvDesc da = nhwc_conv(n, 4, 4, 256, 256, 3, 2, 0, 1);
            CIS_TRY(conv_fill_chip(ws, da, x, c->dl[27], T1, st));  // [n][1][1][256]
            hipLaunchKernelGGL(k_dlib_tail<2>, dim3((unsigned)ceil_div(n, 2)), dim3(256), 0, st, (const float*)T1, (const float*)x, (const float*)c->dl[28].d_w,
                               (const float*)c->dl[28].d_b, (const float*)c->dl[29].d_w, d_feats, n);
            CIS_CHECK_HIP(hipGetLastError());
            return CIS_OK;
        }
        ConvDesc da = nhwc_conv(n, H, W, C, b.cout, 3, s, p, 1);
        CIS_TRY(conv_fill_chip(ws, da, x, c->dl[1 + 2 * i], T1, st));
        ConvDesc db = nhwc_conv(n, da.OH, da.OW, b.cout, b.cout, 3, 1, 1, 0);
        int SH = H, SW = W, SC = C;
        if (b.down) { SH = (H - 2) / 2 + 1; SW = (W - 2) / 2 + 1; }
        const int OH = db.OH > SH ? db.OH : SH, OW = db.OW > SW ? db.OW : SW, OC = b.cout > SC ? b.cout : SC;
        // add_prev + relu inside the second convolution's epilogue when both branches have the output's spatial size
        // (the residual may have fewer channels); the two smallest down blocks grow spatially and keep the separate pass
        const bool fuse = db.OH == OH && db.OW == OW && SH == OH && SW == OW && OC == b.cout;
        const float* skip = x;
        if (b.down) {
            // the pooled skip branch goes to T2 when fused (T1 still feeds the second convolution), else to T1 afterwards
            if (fuse) {
                hipLaunchKernelGGL(k_avgpool2_nhwc, grid((int64_t)n * SH * SW * C), dim3(256), 0, st, x, T2, n, H, W, C, SH, SW);
                skip = T2;
            }
        }
        if (fuse) {
            db.res = skip; db.resC = SC; db.relu = 1;
            CIS_TRY(conv_fill_chip(ws, db, T1, c->dl[2 + 2 * i], other, st));
        } else {
            CIS_TRY(conv_fill_chip(ws, db, T1, c->dl[2 + 2 * i], T2, st));
            if (b.down) {
                hipLaunchKernelGGL(k_avgpool2_nhwc, grid((int64_t)n * SH * SW * C), dim3(256), 0, st, x, T1, n, H, W, C, SH, SW);
                skip = T1;  // T1 is free again: conv b has consumed it (same stream)
            }
            hipLaunchKernelGGL(k_add_relu_pad, grid((int64_t)n * OH * OW * OC), dim3(256), 0, st, T2, db.OH, db.OW, b.cout, skip, SH, SW, SC,
                               other, n, OH, OW, OC);
        }
        float* t = x; x = other; other = t;
        H = OH; W = OW; C = OC;
    }
    hipLaunchKernelGGL(k_global_avgpool, grid((int64_t)n * C), dim3(256), 0, st, x, other, n, H * W, C);
    ConvDesc df = nhwc_conv(n, 1, 1, C, 128, 1, 1, 0, 0);
    launch_conv(df, other, c->dl[29].d_w, nullptr, d_feats, st);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

static int cnn_forward_sentibank(cis_cnn* c, CnnWs* ws, const float* d_nchw, int n, float* d_feats, hipStream_t st);

extern char** environ;
// the CIS_CNN_* switches of the environment as one word (they are read per call, so that one process can compare routes: a captured
// forward belongs to the switches it was captured under)
static uint64_t cnn_env_signature() {
    uint64_t h = 1469598103934665603ull;
    for (char** e = environ; e && *e; ++e) {
        if (strncmp(*e, "CIS_CNN_", 8) != 0) continue;
        for (const char* p = *e; *p; ++p) h = (h ^ (unsigned char)*p) * 1099511628211ull;
        h = (h ^ 0xffu) * 1099511628211ull;
    }
    return h;
}

// The forward of (input, n, output, parts) as a graph launch on the caller's stream.  First sight of a key: nothing (the caller enqueues
// the launches, which also sizes the workspaces); second sight: the same enqueue is captured on the handle's own streams -- the parts fork
// from and join the capture stream through the events the launch path uses -- and instantiated; from then on one hipGraphLaunch.
// *handled = false: the caller goes on with the launches (not captured yet, or the capture failed: the handle stops trying).
static int cnn_forward_graph(cis_cnn* c, const float* d_in, int n, float* d_feats, int parts, size_t in_item, size_t out_item, hipStream_t st,
                             bool* handled) {
    *handled = false;
    const uint64_t env = cnn_env_signature();
    CnnGraph* g = nullptr;
    CnnGraph* lru = &c->graphs[0];
    for (auto& e : c->graphs) {
        if (e.seen && e.in == d_in && e.out == d_feats && e.n == n && e.parts == parts && e.env == env) { g = &e; break; }
        if (e.used < lru->used) lru = &e;
    }
    if (!g) {  // first sight: remember the key in the least recently used slot
        if (lru->exec) { (void)hipGraphExecDestroy(lru->exec); lru->exec = nullptr; }
        lru->in = d_in; lru->out = d_feats; lru->n = n; lru->parts = parts; lru->env = env; lru->seen = true;
        lru->used = ++c->graph_clock;
        return CIS_OK;
    }
    g->used = ++c->graph_clock;
    auto same_ws = [&]() {
        for (int p = 0; p < parts; ++p) {
            const void* now[5] = {c->ws[p].act0.p, c->ws[p].act1.p, c->ws[p].act2.p, c->ws[p].act3.p, c->ws[p].part.p};
            for (int i = 0; i < 5; ++i) if (g->wsp[p][i] != now[i]) return false;
        }
        return true;
    };
    if (g->exec && !same_ws()) { (void)hipGraphExecDestroy(g->exec); g->exec = nullptr; }  // a larger batch grew a workspace since
    if (!g->exec) {
        if (!c->gs && hipStreamCreateWithFlags(&c->gs, hipStreamNonBlocking) != hipSuccess) { c->graph_off = true; return CIS_OK; }
        bool ok = true;
        if (parts > 1) {
            if (!c->ev_in) ok = ok && hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) == hipSuccess;
            for (int p = 1; p < parts && ok; ++p) {
                if (!c->ps[p]) ok = ok && hipStreamCreateWithFlags(&c->ps[p], hipStreamNonBlocking) == hipSuccess;
                if (!c->ev_done[p]) ok = ok && hipEventCreateWithFlags(&c->ev_done[p], hipEventDisableTiming) == hipSuccess;
            }
        }
        if (!ok || hipStreamBeginCapture(c->gs, hipStreamCaptureModeRelaxed) != hipSuccess) { c->graph_off = true; (void)hipGetLastError(); return CIS_OK; }
        int rc = CIS_OK;
        if (parts <= 1) {
            rc = cnn_forward_dlib(c, &c->ws[0], d_in, n, d_feats, c->gs);
        } else {
            ok = hipEventRecord(c->ev_in, c->gs) == hipSuccess;
            for (int p = 1; p < parts && ok && rc == CIS_OK; ++p) {
                const int lo = (int)((int64_t)n * p / parts), hi = (int)((int64_t)n * (p + 1) / parts);
                ok = hipStreamWaitEvent(c->ps[p], c->ev_in, 0) == hipSuccess;
                if (!ok) break;
                rc = cnn_forward_dlib(c, &c->ws[p], d_in + lo * in_item, hi - lo, d_feats + lo * out_item, c->ps[p]);
                ok = hipEventRecord(c->ev_done[p], c->ps[p]) == hipSuccess && hipStreamWaitEvent(c->gs, c->ev_done[p], 0) == hipSuccess;
            }
            if (ok && rc == CIS_OK) rc = cnn_forward_dlib(c, &c->ws[0], d_in, (int)((int64_t)n / parts), d_feats, c->gs);
        }
        hipGraph_t graph = nullptr;
        const hipError_t ee = hipStreamEndCapture(c->gs, &graph);
        if (!ok || rc != CIS_OK || ee != hipSuccess || !graph ||
            hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            g->exec = nullptr;
            c->graph_off = true;
            (void)hipGetLastError();
            return CIS_OK;  // (the launches: they report what is wrong, if anything is)
        }
        (void)hipGraphDestroy(graph);
        for (int p = 0; p < kMaxParts; ++p) {
            const void* now[5] = {c->ws[p].act0.p, c->ws[p].act1.p, c->ws[p].act2.p, c->ws[p].act3.p, c->ws[p].part.p};
            for (int i = 0; i < 5; ++i) g->wsp[p][i] = const_cast<void*>(now[i]);
        }
    }
    CIS_CHECK_HIP(hipGraphLaunch(g->exec, st));
    *handled = true;
    return CIS_OK;
}


extern "C" int cis_cnn_forward_dev(cis_cnn* c, const float* d_nchw, int n, float* d_feats, void* stream) {
    CIS_REQUIRE(c != nullptr, "cnn is NULL");
    CIS_REQUIRE(n >= 0, "n must be >= 0");
    if (n == 0) return CIS_OK;
    CIS_CHECK_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    // parts of the batch in flight together (CIS_CNN_PARTS=1: one launch chain on the caller's stream)
    // measured (dlib): two parts at 256 chips 2.16 -> 2.03 ms; four parts lose (the host enqueues the parts' ~50 launches one part after
    // the other, so the last part starts late); 1024 chips: launches are long enough alone; DeepSentibank: +2 %, left alone
    CIS_REQUIRE(!c->orphaned, "this view's base was destroyed: close views before their base");
    int parts = (c->arch == 2 && n >= 128 && n <= 512) ? 2 : 1;
    if (c->parts_override > 0) parts = c->parts_override;
    if (const char* e = getenv("CIS_CNN_PARTS")) parts = atoi(e);
    if (parts > kMaxParts) parts = kMaxParts;
    if (parts > n) parts = n;
    const size_t in_item = c->arch == 2 ? (size_t)150 * 150 * 3 : (size_t)3 * 227 * 227, out_item = c->arch == 2 ? 128 : 4096;
    static const bool threaded = getenv("CIS_CNN_THREADS") && atoi(getenv("CIS_CNN_THREADS")) != 0;
    // CIS_CNN_GRAPH=1: the dlib forward as a captured graph.  Off by default -- measured on ROCm 7.2 (profiles/r06_cnn_graph_ab.txt): the two
    // half-batch chains of a graph do not overlap (2.15 ms against 1.71 ms as launches), three views in flight 1.58 against 1.44 ms.
    const char* ge = getenv("CIS_CNN_GRAPH");
    const bool graphs_on = ge && atoi(ge) != 0;
    if (graphs_on && c->arch == 2 && !c->graph_off && !threaded && parts >= 1) {
        bool handled = false;
        const int rc = cnn_forward_graph(c, d_nchw, n, d_feats, parts, in_item, out_item, st, &handled);
        if (handled) return rc;
    }
    if (parts <= 1)
        return c->arch == 2 ? cnn_forward_dlib(c, &c->ws[0], d_nchw, n, d_feats, st) : cnn_forward_sentibank(c, &c->ws[0], d_nchw, n, d_feats, st);
    // Part 0 runs on the caller's stream, parts 1.. on the handle's own (made for the parts in use only).  The GPU dispatches from four
    // hardware pipes and a process's streams are dealt onto them in the order they first submit work (tools/r06_queue_probe.py): two
    // streams on one pipe do not overlap, and a stream that holds a wait blocks its pipe for the others on it.  With every part on a
    // stream of the handle and the caller's stream holding the joins, a part that shared the caller's pipe ran after the other part
    // (2.5 instead of 1.7 ms for 256 dlib chips); now one stream instead of two can collide, and a collision costs the overlap only.
    if (!c->ev_in) CIS_CHECK_HIP(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    for (int p = 1; p < parts; ++p) {
        if (!c->ps[p]) CIS_CHECK_HIP(hipStreamCreateWithFlags(&c->ps[p], hipStreamNonBlocking));
        if (!c->ev_done[p]) CIS_CHECK_HIP(hipEventCreateWithFlags(&c->ev_done[p], hipEventDisableTiming));
    }
    CIS_CHECK_HIP(hipEventRecord(c->ev_in, st));  // the parts start after the caller's earlier work ...
    // A failure inside the loop must not leave parts that are already running unfenced: every part that was started is still
    // recorded and waited for on the caller's stream, then the first error is returned.
    int rc = CIS_OK;
    hipError_t herr = hipSuccess;
    // CIS_CNN_THREADS=1: every part is enqueued from its own host thread (a part is ~50 launches: enqueued one part after the other,
    // the last part starts late); default off, measured in profiles/r03y_cnn_parts.txt
    if (threaded) {
        if (!c->ps[0]) CIS_CHECK_HIP(hipStreamCreateWithFlags(&c->ps[0], hipStreamNonBlocking));
        if (!c->ev_done[0]) CIS_CHECK_HIP(hipEventCreateWithFlags(&c->ev_done[0], hipEventDisableTiming));
        int rcs[kMaxParts] = {0, 0, 0, 0};
        for (int p = 0; p < parts && herr == hipSuccess; ++p) herr = hipStreamWaitEvent(c->ps[p], c->ev_in, 0);
        if (herr == hipSuccess) {
            auto run = [&](int p) {
                (void)hipSetDevice(c->device);
                const int lo = (int)((int64_t)n * p / parts), hi = (int)((int64_t)n * (p + 1) / parts);
                rcs[p] = c->arch == 2 ? cnn_forward_dlib(c, &c->ws[p], d_nchw + lo * in_item, hi - lo, d_feats + lo * out_item, c->ps[p])
                                      : cnn_forward_sentibank(c, &c->ws[p], d_nchw + lo * in_item, hi - lo, d_feats + lo * out_item, c->ps[p]);
            };
            std::thread th[kMaxParts];
            for (int p = 1; p < parts; ++p) th[p] = std::thread(run, p);
            run(0);
            for (int p = 1; p < parts; ++p) th[p].join();
            for (int p = 0; p < parts; ++p) {
                if (rcs[p] != CIS_OK && rc == CIS_OK) rc = rcs[p];
                hipError_t e = hipEventRecord(c->ev_done[p], c->ps[p]);
                if (e == hipSuccess) e = hipStreamWaitEvent(st, c->ev_done[p], 0);
                if (e != hipSuccess) { (void)hipStreamSynchronize(c->ps[p]); herr = e; }
            }
            if (rc != CIS_OK) { cis_set_error("a part of the CNN forward failed (code %d)", rc); return rc; }
        }
        CIS_CHECK_HIP(herr);
        return CIS_OK;
    }
    int started = 1;  // parts 1 .. started - 1 were enqueued
    for (int p = 1; p < parts && rc == CIS_OK && herr == hipSuccess; ++p) {
        const int lo = (int)((int64_t)n * p / parts), hi = (int)((int64_t)n * (p + 1) / parts);
        herr = hipStreamWaitEvent(c->ps[p], c->ev_in, 0);
        if (herr != hipSuccess) break;  // nothing of this part was enqueued
        rc = c->arch == 2 ? cnn_forward_dlib(c, &c->ws[p], d_nchw + lo * in_item, hi - lo, d_feats + lo * out_item, c->ps[p])
                          : cnn_forward_sentibank(c, &c->ws[p], d_nchw + lo * in_item, hi - lo, d_feats + lo * out_item, c->ps[p]);
        started = p + 1;
    }
    if (rc == CIS_OK && herr == hipSuccess) {
        const int hi = (int)((int64_t)n / parts);
        rc = c->arch == 2 ? cnn_forward_dlib(c, &c->ws[0], d_nchw, hi, d_feats, st) : cnn_forward_sentibank(c, &c->ws[0], d_nchw, hi, d_feats, st);
    }
    for (int p = 1; p < started; ++p) {  // ... and the caller's later work waits for every part
        hipError_t e = hipEventRecord(c->ev_done[p], c->ps[p]);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, c->ev_done[p], 0);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(c->ps[p]);  // the fence could not be placed: drain the part before reporting
            if (herr == hipSuccess) herr = e;
        }
    }
    if (rc != CIS_OK) return rc;
    CIS_CHECK_HIP(herr);
    return CIS_OK;
}

static int cnn_forward_sentibank(cis_cnn* c, CnnWs* ws, const float* d_nchw, int n, float* d_feats, hipStream_t st) {
    // largest activation: conv1 output n x 55 x 55 x 96
    const size_t act_elems = (size_t)n * 55 * 55 * 96;
    CIS_TRY(ws->act0.reserve(act_elems * sizeof(float)));
    CIS_TRY(ws->act1.reserve(act_elems * sizeof(float)));
    const char* sk_e = getenv("CIS_CNN_FC_SPLITK");  // measurement switch: the order of a sum depends on it
    const int fc_splitk = sk_e ? atoi(sk_e) : 8;
    CIS_REQUIRE(fc_splitk >= 1 && fc_splitk <= 32, "CIS_CNN_FC_SPLITK out of range");
    CIS_TRY(ws->act2.reserve((size_t)fc_splitk * n * 4096 * sizeof(float)));  // split-K partial sums of the fc layers
    CIS_TRY(cnn_poison(ws, st));
    float* bufs[2] = {ws->act0.as<float>(), ws->act1.as<float>()};
    const float* cur = d_nchw;
    int which = 0;
    int C = 3, H = 227, W = 227;
    bool nchw = true;
    for (int l = 0; l < 5; ++l) {
        ConvDesc d;
        d.N = n; d.H = H; d.W = W; d.C = C;
        d.OC = kConvCfg[l][0]; d.KH = d.KW = kConvCfg[l][1]; d.stride = kConvCfg[l][2]; d.pad = kConvCfg[l][3];
        d.groups = kConvCfg[l][4];
        d.OH = (H + 2 * d.pad - d.KH) / d.stride + 1;
        d.OW = (W + 2 * d.pad - d.KW) / d.stride + 1;
        d.ICg = C / d.groups; d.OCg = d.OC / d.groups; d.K = d.KH * d.KW * d.ICg;
        if (nchw) { d.sN = (int64_t)C * H * W; d.sC = (int64_t)H * W; d.sH = W; d.sW = 1; }
        else { d.sN = (int64_t)H * W * C; d.sC = 1; d.sH = (int64_t)W * C; d.sW = C; }
        d.relu = 1;
        d.kx_fastest = 0;
        float* o = bufs[which];
        const float* wl = c->conv[l].d_w;
        if (l == 0 && !getenv("CIS_CNN_NHWC_FIRST")) {
            // first layer straight from the caller's NCHW planes (round 5): a (pixel, channel, kernel row) is a run of 11 contiguous
            // floats, gathered as three 16-byte loads (4-byte aligned: the planes' rows are 227 floats); the re-layout pass below
            // (57 us per 256 images, a 158 MB round trip) is gone.  K = 3 * 11 * 12 = 396 as before.
            d.runq = (d.KW + 3) / 4;                 // 3 float4 per run
            d.run_planes = C * d.KH;                 // 33 runs
            d.K = d.run_planes * d.runq * 4;         // 396
            d.in_elems = (int64_t)n * C * H * W;
            wl = c->conv[l].d_w2;
        } else if (l == 0) {
            // first layer: the NCHW batch is re-laid as NHWC rows (pitch 684 floats) once, then every (pixel, kernel row)
            // is a run of 33 contiguous floats gathered with aligned 16-byte loads instead of 363 scalar loads per pixel
            const int pitch = ((W * 3 + 3) / 4) * 4;
            CIS_TRY(ws->act3.reserve(((size_t)n * H * pitch + 16) * sizeof(float)));
            float* xt = ws->act3.as<float>();
            hipLaunchKernelGGL(k_nchw3_to_nhwc, dim3((unsigned)ceil_div((int64_t)n * H * W, 256)), dim3(256), 0, st, cur, xt,
                               (int64_t)n * H, H, W, pitch);
            d.sN = (int64_t)H * pitch; d.sC = 1; d.sH = pitch; d.sW = 3;
            d.runq = ((d.KW * 3 + 3) / 4);      // 9 float4 per kernel row
            d.K = d.KH * d.runq * 4;            // 396
            cur = xt;
        }
        launch_conv(d, cur, wl, c->conv[l].d_b, o, st);
        cur = o; which ^= 1; nchw = false;
        H = d.OH; W = d.OW; C = d.OC;
        if (kPoolAfter[l] && kLrnAfter[l]) {
            const int OH = (H - 3 + 1) / 2 + 1, OW = (W - 3 + 1) / 2 + 1;
            float* po = bufs[which];
            if (C % 4 == 0 && C <= 1024) {
                const int tpp = C / 4, ppb = 256 / tpp;
                const int remap = getenv("CIS_CNN_NO_XCD_REMAP") ? 0 : 1;
                const int64_t tiles = ceil_div((int64_t)n * OH * OW, ppb);
                hipLaunchKernelGGL(k_maxpool_lrn_nhwc_v4, dim3((unsigned)(remap ? ceil_div(tiles, 8) * 8 : tiles)), dim3(256),
                                   (size_t)ppb * C * sizeof(float), st, cur, po, n, H, W, C, OH, OW, 5, 1e-4f, 0.75f, ppb, remap);
            } else {
                const int threads = C <= 64 ? 64 : (C <= 128 ? 128 : 256);
                hipLaunchKernelGGL(k_maxpool_lrn_nhwc, dim3((unsigned)((int64_t)n * OH * OW)), dim3(threads), (size_t)C * sizeof(float), st,
                                   cur, po, n, H, W, C, OH, OW, 5, 1e-4f, 0.75f);
            }
            cur = po; which ^= 1; H = OH; W = OW;
        } else if (kPoolAfter[l]) {
            const int OH = (H - 3 + 1) / 2 + 1, OW = (W - 3 + 1) / 2 + 1;
            float* po = bufs[which];
            const int64_t total = (int64_t)n * OH * OW * C;
            if (C % 4 == 0) hipLaunchKernelGGL(k_maxpool_nhwc_v4, dim3((unsigned)ceil_div(total / 4, 256)), dim3(256), 0, st, cur, po, n, H, W, C, OH, OW);
            else hipLaunchKernelGGL(k_maxpool_nhwc, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, cur, po, n, H, W, C, OH, OW);
            cur = po; which ^= 1; H = OH; W = OW;
        }
        if (kLrnAfter[l] && !kPoolAfter[l]) {
            float* lo = bufs[which];
            const int64_t npix = (int64_t)n * H * W;
            hipLaunchKernelGGL(k_lrn_nhwc, dim3((unsigned)ceil_div(npix * C, 256)), dim3(256), 0, st, cur, lo, npix, C, 5, 1e-4f, 0.75f);
            cur = lo; which ^= 1;
        }
    }
    // fc6, fc7 as 1x1 convolutions over a 1x1 image with C = flattened features
    int fin = C * H * W;
    for (int l = 0; l < 2; ++l) {
        ConvDesc d;
        d.N = n; d.H = 1; d.W = 1; d.C = fin; d.OH = 1; d.OW = 1; d.OC = 4096; d.KH = d.KW = 1; d.stride = 1; d.pad = 0;
        d.groups = 1; d.ICg = fin; d.OCg = 4096; d.K = fin;
        d.sN = fin; d.sC = 1; d.sH = fin; d.sW = fin;
        d.relu = 1;
        d.kx_fastest = 0;
        float* o = (l == 1) ? d_feats : bufs[which];
        // few output tiles (n x 4096) and a long K: split K over eight blocks per tile (round 5; four before) so that every CU holds several
        // workgroups (the K loop is a chain of dependent loads), then add the partial sums in a fixed order
        const int splitk = (fin % 16 == 0 && (int64_t)n * 4096 % 4 == 0 && n <= 4096) ? fc_splitk : 1;
        if (splitk > 1) {
            d.splitk = splitk;
            d.part_stride = (int64_t)n * 4096;
            float* part = ws->act2.as<float>();
            launch_conv(d, cur, c->fc[l].d_w, nullptr, part, st);
            const int64_t n4 = (int64_t)n * 4096 / 4;
            hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, st, part, splitk, d.part_stride,
                               c->fc[l].d_b, o, n4, 4096, 1, (const float*)nullptr, 0);
        } else {
            launch_conv(d, cur, c->fc[l].d_w, c->fc[l].d_b, o, st);
        }
        cur = o; which ^= 1; fin = 4096;
    }
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

extern "C" int cis_cnn_forward(cis_cnn* c, const float* nchw, int n, float* feats) {
    CIS_REQUIRE(c != nullptr, "cnn is NULL");
    CIS_REQUIRE(!c->orphaned, "this view's base was destroyed: close views before their base");
    CIS_REQUIRE(n >= 0 && (n == 0 || (nchw && feats)), "NULL buffer");
    if (n == 0) return CIS_OK;
    CIS_CHECK_HIP(hipSetDevice(c->device));
    const int fdim = cis_cnn_feat_dim(c->arch);
    const size_t in_bytes = (size_t)n * (c->arch == 2 ? 3 * 150 * 150 : 3 * 227 * 227) * sizeof(float);
    CIS_TRY(c->in_buf.reserve(in_bytes));
    CIS_TRY(c->out_buf.reserve((size_t)n * fdim * sizeof(float)));
    CIS_CHECK_HIP(hipMemcpy(c->in_buf.p, nchw, in_bytes, hipMemcpyHostToDevice));
    CIS_TRY(cis_cnn_forward_dev(c, c->in_buf.as<float>(), n, c->out_buf.as<float>(), nullptr));
    CIS_CHECK_HIP(hipMemcpy(feats, c->out_buf.p, (size_t)n * fdim * sizeof(float), hipMemcpyDeviceToHost));
    return CIS_OK;
}
