// Internal view of the LOPQ model handle shared by lopq_model.hip and lopq_search.hip.
#pragma once
#include "common.h"

struct cis_model {
    int D_in = 0, D = 0, V = 0, M = 0, K = 0;
    int h = 0, w = 0, nf = 0;  // D/2, D/M, M/2
    bool coarse_f32 = false;   // coarse centroids were given as float32
    bool has_pca = false, renorm = false;
    bool pca_mu_f32 = false;   // the caller's pca_mu was float32: x - mu rounds in float32 for float32 x
    int device = 0;

    float* d_Cs32 = nullptr;   // [2][V][h]   (only when coarse_f32)
    double* d_Cs64 = nullptr;  // [2][V][h]   exact widening of Cs
    double* d_Rs = nullptr;    // [2][V][h][h]  R[c][i][k]
    double* d_Rt = nullptr;    // [2][V][h][h]  transposed: Rt[c][k][i] = R[c][i][k]
    double* d_mus = nullptr;   // [2][V][h]
    double* d_subs = nullptr;  // [M][K][w]
    double* d_P = nullptr;     // [D_in][D]
    double* d_pmu = nullptr;   // [D_in]
    double* d_cnorm = nullptr; // [2][V] squared norms of the coarse centroids, then the maximum per split [2]
    int* d_flag = nullptr;     // [2] overflow flag of the coarse prefilter (k_coarse_mfma), one per encode pass parity

    PwProg prog_h, prog_w, prog_D;  // numpy summation order over h, w and D elements

    // scratch for encode / piecewise calls (calls on one handle are serialised by the caller)
    DevBuf ws_xp, ws_x64, ws_y64, ws_dist, ws_proj, ws_group, ws_in, ws_out0, ws_out1;
};

// Device-pointer building blocks used by both translation units (all asynchronous on `st`).
// xp: LOPQ-space vectors [n][D] of xp_dtype.  Returns in *xc a pointer to the same vectors in the
// coarse compute type (float when both xp and Cs are float32, else double; may alias xp or
// m->ws_x64) and the compute type in *ct (CIS_F32 / CIS_F64).
int cis_dev_apply_pca(cis_model* m, const void* dX, int x_dtype, int64_t n, float* d_out, hipStream_t st,
                      DevBuf* ws_y = nullptr /* float64 product before the finish pass; default: the model's own (one caller at a time) */);
int cis_dev_coarse_type(cis_model* m, const void* d_xp, int xp_dtype, int64_t n, const void** xc, int* ct,
                        hipStream_t st,
                        DevBuf* ws_x = nullptr /* widened copy; default: the model's own */);
// squared distances of n compute-type vectors to the V coarse centroids of `split`, numpy order:
// out [n][V] of the compute type.
int cis_launch_sqdist(cis_model* m, const void* xc, int ct, int64_t n, int split, void* out, hipStream_t st);
int cis_launch_sqdist_both(cis_model* m, const void* xc, int ct, int64_t n, void* out /* [2][n][V] */, hipStream_t st);
// generic: out[r][c] = numpy-order squared distance of X[r][xoff .. xoff+d) to C[c][0..d), both of compute type ct
int cis_launch_sqdist_generic(const void* X, int ct, int64_t ldx, int xoff, const void* C, int64_t n, int ncent, int d,
                              void* out, hipStream_t st);
