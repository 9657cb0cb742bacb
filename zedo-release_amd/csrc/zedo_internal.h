// Internal declarations shared by the translation units of libzedo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace zedo {

constexpr int HID = 1024;    // hidden width of ScoreModelFC_Adv (reference run/opt_main.py:35)
constexpr int EMB = 512;     // time-embedding width (run/opt_main.py:36)
constexpr int XLD = 64;      // padded row length of the pose state: J*3 = 51 floats + 13 zeros
constexpr int ROW_PAD = 256; // rows of every activation buffer are padded to a multiple of this
constexpr int NLAYER = 5;    // hidden layers with a time bias: pre + 2 blocks x 2

enum Epilogue : int {
    EPI_GN_SILU = 0,      // out = SiLU(GroupNorm32(acc + bias))
    EPI_GN_SILU_RES = 1,  // out = out + SiLU(GroupNorm32(acc + bias))          (h = h + h2)
    EPI_SDE = 2,          // out = a*out + c*(acc + bias)                        (x' = a x + c eps)
    EPI_BIAS = 3,         // out = acc + bias
    EPI_BIAS_SILU = 4,    // out = SiLU(acc + bias)
};

// One dense layer out[M][N] (+epilogue) = X[M][K] . W[N][K]^T ; all row-major fp32, K contiguous.
struct LayerArgs {
    const float *X;   // [Mp][ldx]
    const float *W;   // [N][ldw]   (torch.nn.Linear.weight layout, K zero-padded to ldw)
    const float *bias;   // [N]
    const float *gamma;  // [N]  GroupNorm weight (GN epilogues)
    const float *beta;   // [N]  GroupNorm bias
    float *out;          // [Mp][ldo]
    int ldx, ldw, ldo;
    int K;               // multiple of 32
    int N;               // multiple of the tile's BN
    int Mp;              // multiple of the tile's BM
    float sde_a, sde_c;  // EPI_SDE
};

hipError_t launch_layer(const LayerArgs &a, int epilogue, hipStream_t st);

// geometry kernels (zedo_geom.hip)
hipError_t launch_pack_rows(const float *x, float *xpad, int B, int Bp, int D, hipStream_t st);
hipError_t launch_unpack_rows(const float *xpad, float *x, int B, int D, hipStream_t st);
hipError_t launch_reproj_prepare(const float *uv, const float *K, const float *conf, int N, int J, float *geom,
                                 float *conf_clamped, hipStream_t st);
hipError_t launch_reproj_grad(const float *x, const float *geom, float *T, int solve_T, float *g, int B, int N,
                              int J, long long row_offset, hipStream_t st);
hipError_t launch_reproj_step_padded(float *xpad, const float *geom, float *T, int solve_T, int B, int N,
                                     long long row0, hipStream_t st);
hipError_t launch_posemb(const float *t, int S, int Sp, float label_scale, float *pe, hipStream_t st);
hipError_t launch_add_bias_rows(float *b_sum, const float *b1, const float *b2, int n, hipStream_t st);
hipError_t launch_ipo_fit(const float *x0, const float *uv, const float *K, const int *d_keylist, int k,
                          int axes_mask, float ipo_T, float min_scale, float max_scale, int iters,
                          double normaliser, float *R, float *T, float *q, float *scale, float *state, int it_begin,
                          int B, int N, int J, long long row_offset, hipStream_t st);
hipError_t launch_rotate_init(const float *x0, const float *R, float *x, int B, int N, int J, long long row_offset,
                              hipStream_t st);
hipError_t launch_min_mpjpe(const float *pred, const double *gt, int B, int N, int J, long long row_offset,
                            int procrustes, double *err, double *best, int *best_h, hipStream_t st);

}  // namespace zedo
