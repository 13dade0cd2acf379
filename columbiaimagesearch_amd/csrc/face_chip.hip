// Aligned face chips on the MI355X: the image side of dlib's compute_face_descriptor(img, shape)
// (cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:103-105: the 68-landmark shape -> get_face_chip_details(shape, 150, 0.25)
// -> extract_image_chip -> the network).  The landmark predictor stays the caller's; the similarity transform of the landmarks onto
// dlib's mean face (get_face_chip_details) is a few dozen float64 operations per face and runs on the host
// (columbiaimagesearch_amd/featurizer/face_chip.py); the two image passes run here:
//   k_pyr_down2      dlib's pyramid_down<2> on an RGB uint8 image: separable 5-tap (1 4 6 4 1) filter in integer arithmetic, every
//                    other row and column kept, sum / 256 truncated -- extract_image_chips builds these levels when the face region is
//                    more than twice the chip, so that the bilinear sampling below never degenerates to point sampling;
//   k_extract_chips  one thread per chip pixel: the affine map chip (c, r) -> source (three corner correspondences, as
//                    find_affine_transform of extract_image_chips gives it), dlib's interpolate_bilinear on the four neighbours in
//                    float64 ((1-tb)((1-lr) tl + lr tr) + tb((1-lr) bl + lr br)), stored as uint8 like assign_pixel(unsigned
//                    char&, double) does: clamped to 0..255 and TRUNCATED (static_cast, no + 0.5 -- round 5; rounds 3-4 rounded to
//                    nearest, a bias of up to one grey level against dlib on half of the pixels), pixels whose neighbourhood leaves the image black; written as float32 0..255, the input format of
//                    cis_cnn_forward for CIS_CNN_DLIB_RESNET.
// dlib is a third-party dependency of the reference with an unpinned version (cufacesearch/cufacesearch/requirements.txt:3) and is not
// installed here: this restates its published image_transforms/interpolation.h / image_pyramid.h; parity is unpinned.
#include "common.h"
#include "cis_hip.h"

__global__ void k_pyr_down2_rows(const uint8_t* __restrict__ in, int nr, int nc, int* __restrict__ tmp /* [nr][tc][3] */, int tc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nr * tc) return;
    const int r = (int)(i / tc), c = (int)(i % tc);
    const uint8_t* p = in + ((int64_t)r * nc + 2 * c) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
        tmp[i * 3 + ch] = (int)p[ch] + 4 * (int)p[3 + ch] + 6 * (int)p[6 + ch] + 4 * (int)p[9 + ch] + (int)p[12 + ch];
}

__global__ void k_pyr_down2_cols(const int* __restrict__ tmp, int nr, int tc, uint8_t* __restrict__ out /* [(nr-3)/2][tc][3] */) {
    const int orows = (nr - 3) / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)orows * tc) return;
    const int dr = (int)(i / tc), c = (int)(i % tc);
    const int r = 2 * dr + 2;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int t = tmp[((int64_t)(r - 2) * tc + c) * 3 + ch] + 4 * tmp[((int64_t)(r - 1) * tc + c) * 3 + ch] +
                      6 * tmp[((int64_t)r * tc + c) * 3 + ch] + 4 * tmp[((int64_t)(r + 1) * tc + c) * 3 + ch] +
                      tmp[((int64_t)(r + 2) * tc + c) * 3 + ch];
        out[i * 3 + ch] = (uint8_t)(t / 256);
    }
}

// maps [n][6]: source = (m0 + m1 c + m2 r, m3 + m4 c + m5 r) for chip pixel (column c, row r)
__global__ void k_extract_chips(const uint8_t* __restrict__ img, int nr, int nc, const double* __restrict__ maps, int n, int size,
                                float* __restrict__ chips /* [n][size][size][3] */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)size * size;
    if (i >= (int64_t)n * per) return;
    const int k = (int)(i / per);
    const int r = (int)((i % per) / size), c = (int)(i % size);
    const double* m = maps + (int64_t)k * 6;
    const double x = m[0] + m[1] * (double)c + m[2] * (double)r;
    const double y = m[3] + m[4] * (double)c + m[5] * (double)r;
    const double fx = floor(x), fy = floor(y);
    const long left = (long)fx, top = (long)fy, right = left + 1, bottom = top + 1;
    float* o = chips + i * 3;
    if (!(left >= 0 && top >= 0 && right < nc && bottom < nr)) {  // dlib: interpolation outside the image -> background (black)
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
        return;
    }
    const double lr = x - (double)left, tb = y - (double)top;
    const uint8_t* tl = img + ((int64_t)top * nc + left) * 3;
    const uint8_t* tr = tl + 3;
    const uint8_t* bl = tl + (int64_t)nc * 3;
    const uint8_t* br = bl + 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const double v = (1.0 - tb) * ((1.0 - lr) * (double)tl[ch] + lr * (double)tr[ch]) + tb * ((1.0 - lr) * (double)bl[ch] + lr * (double)br[ch]);
        o[ch] = (float)(unsigned char)(v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v));  // pixel.h: assign(unsigned char&, double) clamps, then static_cast
    }
}

// d_img [nr][nc][3] uint8 RGB on the device -> d_out [(nr-3)/2][(nc-3)/2][3]; d_tmp: nr * ((nc-3)/2) * 3 ints of scratch
extern "C" int cis_pyramid_down2_dev(const uint8_t* d_img, int nr, int nc, uint8_t* d_out, int* d_tmp, void* stream) {
    CIS_REQUIRE(d_img && d_out && d_tmp, "NULL buffer");
    CIS_REQUIRE(nr > 8 && nc > 8, "pyramid_down<2> needs an image of more than 8 x 8 pixels");
    CIS_TRY(cis_lazy_init());
    hipStream_t st = (hipStream_t)stream;
    const int tc = (nc - 3) / 2, orows = (nr - 3) / 2;
    hipLaunchKernelGGL(k_pyr_down2_rows, dim3((unsigned)ceil_div((int64_t)nr * tc, 256)), dim3(256), 0, st, d_img, nr, nc, d_tmp, tc);
    hipLaunchKernelGGL(k_pyr_down2_cols, dim3((unsigned)ceil_div((int64_t)orows * tc, 256)), dim3(256), 0, st, (const int*)d_tmp, nr, tc, d_out);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}

// n chips of size x size pixels out of ONE image (or pyramid level) on the device; d_maps [n][6] float64 affine maps chip -> source
extern "C" int cis_extract_chips_dev(const uint8_t* d_img, int nr, int nc, const double* d_maps, int n, int size, float* d_chips, void* stream) {
    CIS_REQUIRE(n >= 0 && size >= 1 && nr >= 1 && nc >= 1, "bad arguments");
    if (n == 0) return CIS_OK;
    CIS_REQUIRE(d_img && d_maps && d_chips, "NULL buffer");
    CIS_TRY(cis_lazy_init());
    hipLaunchKernelGGL(k_extract_chips, dim3((unsigned)ceil_div((int64_t)n * size * size, 256)), dim3(256), 0, (hipStream_t)stream, d_img, nr, nc,
                       d_maps, n, size, d_chips);
    CIS_CHECK_HIP(hipGetLastError());
    return CIS_OK;
}
