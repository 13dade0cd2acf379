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
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvDesc {
    int N, H, W, C;          // input: images, height, width, total channels
    int OH, OW, OC;          // output
    int KH, KW, stride, pad, groups;
    int ICg, OCg, K;         // per group; K = KH*KW*ICg, k = (ky*KW + kx)*ICg + ic
    int64_t sN, sC, sH, sW;  // input strides in elements (NCHW for the first layer, NHWC afterwards)
    int relu;
};

// C[pixel][oc] = sum_k A[pixel][k] * Wp[k][oc] + bias[oc]; block tile BM x BN, 256 threads = 4 waves laid
// out WAVES_M x WAVES_N, each wave WM x WN MFMA tiles of 32x32.
template <int WM, int WN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void k_conv_igemm(const float* __restrict__ in, const float* __restrict__ Wp,
                                                    const float* __restrict__ bias, float* __restrict__ out, ConvDesc d) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, BK = 16;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    __shared__ float As[BK][LDA];  // [k][pixel]
    __shared__ float Bs[BK][LDB];  // [k][oc]
    __shared__ int s_iy0[BM], s_ix0[BM];
    __shared__ int64_t s_base[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int g = blockIdx.z;
    const int64_t npix = (int64_t)d.N * d.OH * d.OW;
    const int64_t pix0 = (int64_t)blockIdx.x * BM;
    const int oc0 = blockIdx.y * BN;  // inside the group
    for (int m = tid; m < BM; m += 256) {
        const int64_t p = pix0 + m;
        if (p < npix) {
            const int ox = (int)(p % d.OW);
            const int oy = (int)((p / d.OW) % d.OH);
            const int64_t img = p / ((int64_t)d.OW * d.OH);
            s_iy0[m] = oy * d.stride - d.pad;
            s_ix0[m] = ox * d.stride - d.pad;
            s_base[m] = img * d.sN + (int64_t)g * d.ICg * d.sC;
        } else {
            s_iy0[m] = -(1 << 28);  // every tap falls outside -> zeros
            s_ix0[m] = -(1 << 28);
            s_base[m] = 0;
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
    __syncthreads();
    for (int k0 = 0; k0 < d.K; k0 += BK) {
        // ---- A tile: im2col gather, lanes along k (contiguous input channels in NHWC) ----
        {
            const int kl = tid & 15;
            const int k = k0 + kl;
            int ky = 0, kx = 0, ic = 0;
            const bool kvalid = k < d.K;
            if (kvalid) {
                ic = k % d.ICg;
                const int t = k / d.ICg;
                kx = t % d.KW;
                ky = t / d.KW;
            }
#pragma unroll
            for (int i = 0; i < BM / 16; ++i) {
                const int m = (tid >> 4) + 16 * i;
                const int iy = s_iy0[m] + ky, ix = s_ix0[m] + kx;
                float v = 0.f;
                if (kvalid && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W)
                    v = in[s_base[m] + (int64_t)ic * d.sC + (int64_t)iy * d.sH + (int64_t)ix * d.sW];
                As[kl][m] = v;
            }
        }
        // ---- B tile: packed weights [k][oc], float4 along oc ----
        for (int idx = tid; idx < BK * (BN / 4); idx += 256) {
            const int kl = idx / (BN / 4), n4 = (idx % (BN / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + kl < d.K && oc0 + n4 < d.OCg) v = *reinterpret_cast<const float4*>(Wg + (int64_t)(k0 + kl) * d.OCg + oc0 + n4);
            Bs[kl][n4 + 0] = v.x; Bs[kl][n4 + 1] = v.y; Bs[kl][n4 + 2] = v.z; Bs[kl][n4 + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = As[kk + (lane >> 5)][(wm * WM + i) * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = Bs[kk + (lane >> 5)][(wn * WN + j) * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: C/D layout col = lane&31 (oc), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) ----
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int ocl = oc0 + (wn * WN + j) * 32 + (lane & 31);
        if (ocl >= d.OCg) continue;
        const float bv = bias[g * d.OCg + ocl];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t p = pix0 + (wm * WM + i) * 32 + row;
                if (p < npix) {
                    float v = acc[i][j][r] + bv;
                    if (d.relu) v = v > 0.f ? v : 0.f;
                    out[p * d.OC + g * d.OCg + ocl] = v;
                }
            }
        }
    }
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

// ================================================================================================
// host
// ================================================================================================
struct LayerW {
    float* d_w = nullptr;  // packed [g][K][OCg]
    float* d_b = nullptr;
};

struct cis_cnn {
    int arch = 0, device = 0;
    LayerW conv[5], fc[2];
    DevBuf act0, act1, in_buf, out_buf;
};

static const int kConvCfg[5][5] = {  // OC, kernel, stride, pad, groups   (prototxt :7-16,:47-58,:88-98,:105-116,:123-134)
    {96, 11, 4, 0, 1}, {256, 5, 1, 2, 2}, {384, 3, 1, 1, 1}, {384, 3, 1, 1, 2}, {256, 3, 1, 1, 2}};
static const bool kPoolAfter[5] = {true, true, false, false, true};
static const bool kLrnAfter[5] = {true, true, false, false, false};

extern "C" int cis_cnn_feat_dim(int arch) { return arch == 1 ? 4096 : 0; }

extern "C" void cis_cnn_destroy(cis_cnn* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto& l : c->conv) { if (l.d_w) (void)hipFree(l.d_w); if (l.d_b) (void)hipFree(l.d_b); }
    for (auto& l : c->fc) { if (l.d_w) (void)hipFree(l.d_w); if (l.d_b) (void)hipFree(l.d_b); }
    c->act0.release(); c->act1.release(); c->in_buf.release(); c->out_buf.release();
    delete c;
}

static int upload_f(float** dst, const float* src, size_t n) {
    CIS_CHECK_HIP(hipMalloc((void**)dst, n * sizeof(float)));
    CIS_CHECK_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return CIS_OK;
}

extern "C" int cis_cnn_create(cis_cnn** out, int arch, const float* const* tensors, int n_tensors) {
    CIS_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    if (arch != 1) {
        cis_set_error("arch %d: only CIS_CNN_SENTIBANK (1) is built", arch);
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

static void launch_conv(const ConvDesc& d, const float* in, const float* w, const float* b, float* out, hipStream_t st) {
    const int64_t npix = (int64_t)d.N * d.OH * d.OW;
    if (d.OCg % 96 == 0 && d.OCg % 128 != 0) {  // 96, 192: 128 x 96 tiles
        dim3 g((unsigned)ceil_div(npix, 128), (unsigned)ceil_div(d.OCg, 96), (unsigned)d.groups);
        hipLaunchKernelGGL((k_conv_igemm<1, 3, 4, 1>), g, dim3(256), 0, st, in, w, b, out, d);
    } else {
        dim3 g((unsigned)ceil_div(npix, 128), (unsigned)ceil_div(d.OCg, 128), (unsigned)d.groups);
        hipLaunchKernelGGL((k_conv_igemm<2, 2, 2, 2>), g, dim3(256), 0, st, in, w, b, out, d);
    }
}

extern "C" int cis_cnn_forward_dev(cis_cnn* c, const float* d_nchw, int n, float* d_feats, void* stream) {
    CIS_REQUIRE(c != nullptr, "cnn is NULL");
    CIS_REQUIRE(n >= 0, "n must be >= 0");
    if (n == 0) return CIS_OK;
    CIS_CHECK_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    // largest activation: conv1 output n x 55 x 55 x 96
    const size_t act_elems = (size_t)n * 55 * 55 * 96;
    CIS_TRY(c->act0.reserve(act_elems * sizeof(float)));
    CIS_TRY(c->act1.reserve(act_elems * sizeof(float)));
    float* bufs[2] = {c->act0.as<float>(), c->act1.as<float>()};
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
        float* o = bufs[which];
        launch_conv(d, cur, c->conv[l].d_w, c->conv[l].d_b, o, st);
        cur = o; which ^= 1; nchw = false;
        H = d.OH; W = d.OW; C = d.OC;
        if (kPoolAfter[l]) {
            const int OH = (H - 3 + 1) / 2 + 1, OW = (W - 3 + 1) / 2 + 1;
            float* po = bufs[which];
            const int64_t total = (int64_t)n * OH * OW * C;
            hipLaunchKernelGGL(k_maxpool_nhwc, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, cur, po, n, H, W, C, OH, OW);
            cur = po; which ^= 1; H = OH; W = OW;
        }
        if (kLrnAfter[l]) {
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
        float* o = (l == 1) ? d_feats : bufs[which];
        launch_conv(d, cur, c->fc[l].d_w, c->fc[l].d_b, o, st);
        cur = o; which ^= 1; fin = 4096;
    }
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

extern "C" int cis_cnn_forward(cis_cnn* c, const float* nchw, int n, float* feats) {
    CIS_REQUIRE(c != nullptr, "cnn is NULL");
    CIS_REQUIRE(n >= 0 && (n == 0 || (nchw && feats)), "NULL buffer");
    if (n == 0) return CIS_OK;
    CIS_CHECK_HIP(hipSetDevice(c->device));
    const size_t in_bytes = (size_t)n * 3 * 227 * 227 * sizeof(float);
    CIS_TRY(c->in_buf.reserve(in_bytes));
    CIS_TRY(c->out_buf.reserve((size_t)n * 4096 * sizeof(float)));
    CIS_CHECK_HIP(hipMemcpy(c->in_buf.p, nchw, in_bytes, hipMemcpyHostToDevice));
    CIS_TRY(cis_cnn_forward_dev(c, c->in_buf.as<float>(), n, c->out_buf.as<float>(), nullptr));
    CIS_CHECK_HIP(hipMemcpy(feats, c->out_buf.p, (size_t)n * 4096 * sizeof(float), hipMemcpyDeviceToHost));
    return CIS_OK;
}
