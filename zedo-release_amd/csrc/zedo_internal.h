// Internal declarations shared by the translation units of libzedo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace zedo {

// ---- per-device launch state ----------------------------------------------------------------------------------
// The dense kernels ask for more dynamic LDS than the 64 KB default, which must be allowed once per kernel AND per
// device; the CU count that drives the tile split is a property of the device too.  Both are cached per device index
// below MAX_DEVICES; a device index beyond that is NOT aliased onto another device's slot: it pays the (cheap) query /
// attribute call every launch.  The flags are idempotent (a race sets the attribute twice).
constexpr int MAX_DEVICES = 16;
inline int device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    return (dev >= 0 && dev < MAX_DEVICES) ? dev : -1;
}
inline hipError_t allow_lds(const void *kern, size_t lds, std::atomic<bool> *done /* [MAX_DEVICES] */) {
    const int slot = device_slot();
    if (slot < 0 || !done[slot].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (slot >= 0) done[slot].store(true, std::memory_order_release);
    }
    return hipSuccess;
}
inline int num_cus() {
    static std::atomic<int> n[MAX_DEVICES];
    const int slot = device_slot();
    int v = slot >= 0 ? n[slot].load(std::memory_order_relaxed) : 0;
    if (!v) {
        int dev = 0;
        hipDeviceProp_t p;
        (void)hipGetDevice(&dev);
        v = (hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
        if (slot >= 0) n[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}

constexpr int HID = 1024;    // hidden width of ScoreModelFC_Adv (reference run/opt_main.py:35)
constexpr int EMB = 512;     // time-embedding width (run/opt_main.py:36)
constexpr int XLD = 64;      // padded row length of the pose state: J*3 = 51 floats + 13 zeros
constexpr int ROW_PAD = 256; // rows of the schedule tables are padded to a multiple of this (one round of 32-row tiles)
constexpr int BATCH_PAD = 64; // pose rows of a call are padded to a multiple of this: the smallest row tile every layer accepts
constexpr int NLAYER = 5;    // hidden layers with a time bias: pre + 2 blocks x 2

enum Epilogue : int {
    EPI_GN_SILU = 0,      // out = SiLU(GroupNorm32(acc + bias))
    EPI_GN_SILU_RES = 1,  // out = out + SiLU(GroupNorm32(acc + bias))          (h = h + h2)
    EPI_SDE = 2,          // out = a*out + c*(acc + bias)                        (x' = a x + c eps)
    EPI_BIAS = 3,         // out = acc + bias
    EPI_BIAS_SILU = 4,    // out = SiLU(acc + bias)
    EPI_PARTIAL = 5,      // out = acc of ONE K quarter, no bias (post_dense on small batches: see LayerArgs::scratch)
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
    int kzero8;          // K == 64 only: columns k = 56..63 of X and W are zero padding (their MFMAs are skipped)
    long long *clk;      // diagnostic (may be null): workgroup 0 writes {shader cycles, 100 MHz wall ticks} it spent in the tile
    // EPI_SDE only, optional (rp_geom != nullptr): the reprojection correction of the NEXT loop iteration
    // (gradient_field_gen + "denoise_x += joint_gradient", run/opt_main.py:203-208) applied to the freshly updated rows
    // while they are still in LDS, instead of a separate launch that reads and rewrites them.
    const float *rp_geom;   // [N][17][8]
    float *rp_T;            // [rows][3], row 0 = first row of this launch
    int rp_solve, rp_B, rp_N;   // least-squares T?, valid rows of this launch, poses
    long long rp_row0;      // global row index of row 0 of this launch
    // post_dense (N == XLD) sums its K = 1024 products as FOUR quarter chains q0..q3 (k in [256 q, 256 q + 256), each an fma
    // chain from zero) combined as ((q0 + q1) + q2) + q3 - in every launch shape, so that results do not depend on it:
    // large batches fold the quarters inside one tile, batches of up to POST_SPLIT_ROWS rows give every quarter its own
    // workgroup (4x the workgroups on a launch that otherwise fills 1/9 of the chip with 512-MFMA dependent chains) and
    // combine them in post_reduce_kernel.  scratch: [Mp/32][4][32][64] floats for the quarter sums (small batches only).
    float *scratch;
};
constexpr int POST_SPLIT_ROWS = 2048;     // measured: 886 rows 22.7 -> 16.3 us per step, 4 096 rows equal, 6 400 rows 22.8 -> 25.7

hipError_t launch_layer(const LayerArgs &a, int epilogue, hipStream_t st);

// One hidden layer on the fp16 matrix pipe (zedo_gemm16.hip): out = epilogue(X . W^T * unscale + bias), X and W in the
// split-fp16 planes format, three 32x32x16 fp16 MFMAs (hl, lh, hh) per 16-k block, fp32 accumulation.
struct Layer16Args {
    // planes are K-BLOCK-MAJOR (zedo_tile.h): [K/16][ld rows][2 planes][16], i.e. the 64 bytes of (block kb, row r) at (kb * ld + r) * 64
    const uint16_t *X;      // planes [K/16][ldx][2][16], rows [0, Mp) used
    int ldx, ldo;           // rows per k block of the X buffer / per 16-channel group of the out and res buffers (>= Mp; the workspace's row count)
    const uint16_t *W;      // planes [K/16][N][2][16] of W * 2^wshift
    const float *bias, *gamma, *beta;   // [N]
    float unscale;          // 2^-wshift (exact)
    const uint16_t *res;    // EPI_GN_SILU_RES: residual planes [N/16][ldo][2][16]; may be the output buffer (in place)
    void *out;              // planes [N/16][ldo][2][16] (out_f32 == 0) or fp32 [Mp][N] (out_f32 != 0): the same 4 N bytes per row
    int out_f32;
    int K, N, Mp;           // K % 64 == 0, N % 128 == 0 (or N == 64: post_dense), Mp % 64 == 0
    long long *clk;         // diagnostic (may be null), as in LayerArgs
    // pre_dense (K == 64): X == nullptr and the operand is the fp32 pose state itself, split by the kernel
    const float *Xf32;      // [Mp][64]
    // post_dense (N == 64, EPI_SDE / EPI_BIAS): fp32 pose state updated in place (EPI_SDE) or eps written to `out` as fp32
    // [Mp][64] (EPI_BIAS); the optional fused reprojection of the next iteration as in LayerArgs
    float *xio;             // [Mp][64]
    float sde_a, sde_c;
    const float *rp_geom;
    float *rp_T;
    int rp_solve, rp_B, rp_N;
    long long rp_row0;
};
hipError_t launch_layer16(const Layer16Args &a, int epilogue, hipStream_t st);
// fp32 [rows][cols] (row stride ld floats) * scale -> planes [cols/16][ldr][2][16] (k-block-major; ldr >= rows); cols % 16 == 0
hipError_t launch_split_planes(const float *src, int rows, int cols, int ld, float scale, uint16_t *dst, int ldr, hipStream_t st);
hipError_t probe_mfma_peak(int iters, double *tflops, double *shader_ghz, hipStream_t st, bool f16 = false);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// geom[n][j] = 8 floats: { r_x, r_y, W, 0,  rhat_x, rhat_y, rhat_z, 0 }   (zedo_reproj_prepare)
constexpr int GEOM_F = 8;

// gradient_field_gen for one pose row held in registers (simple_zeroshot_opt.py:73-109)
// Weighted least squares for T, centred closed form of the 3x3 normal equations of :73-92:
//   minimise sum_j W_j [(-T_x + r_xj T_z - b_xj)^2 + (-T_y + r_yj T_z - b_yj)^2],  b = x_xy - x_z r_xy
//   T_z = sum W[(r_x-rbar_x)(b_x-bbar_x) + (r_y-rbar_y)(b_y-bbar_y)] / sum W[(r_x-rbar_x)^2 + (r_y-rbar_y)^2]
//   T_xy = rbar_xy T_z - bbar_xy ;  T <- -T if T_z < 0 (:93)
template <int J>
__device__ __forceinline__ void reproj_row(const float *x, const float *__restrict__ gp, float *T, bool solve,
                                           float *g) {
    if (solve) {
        float sw = 0.f, srx = 0.f, sry = 0.f, sbx = 0.f, sby = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(gp + j * GEOM_F);
            const float bx = x[3 * j] - x[3 * j + 2] * a[0], by = x[3 * j + 1] - x[3 * j + 2] * a[1];
            sw += a[2]; srx += a[2] * a[0]; sry += a[2] * a[1]; sbx += a[2] * bx; sby += a[2] * by;
        }
        const float iw = 1.0f / sw;
        const float mrx = srx * iw, mry = sry * iw, mbx = sbx * iw, mby = sby * iw;
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(gp + j * GEOM_F);
            const float bx = x[3 * j] - x[3 * j + 2] * a[0], by = x[3 * j + 1] - x[3 * j + 2] * a[1];
            const float dx = a[0] - mrx, dy = a[1] - mry;
            num += a[2] * (dx * (bx - mbx) + dy * (by - mby));
            den += a[2] * (dx * dx + dy * dy);
        }
        float tz = num / den;
        float tx = mrx * tz - mbx, ty = mry * tz - mby;
        if (tz < 0.f) { tx = -tx; ty = -ty; tz = -tz; }
        T[0] = tx; T[1] = ty; T[2] = tz;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const f32x4 rh = *reinterpret_cast<const f32x4 *>(gp + j * GEOM_F + 4);
        const float px = x[3 * j] + T[0], py = x[3 * j + 1] + T[1], pz = x[3 * j + 2] + T[2];
        const float d = px * rh[0] + py * rh[1] + pz * rh[2];
        g[3 * j] = d * rh[0] - px;
        g[3 * j + 1] = d * rh[1] - py;
        g[3 * j + 2] = d * rh[2] - pz;
    }
}



// geometry kernels (zedo_geom.hip)
hipError_t launch_pack_rows(const float *x, float *xpad, int B, int Bp, int D, hipStream_t st);
hipError_t launch_unpack_rows(const float *xpad, float *x, int B, int D, hipStream_t st);
hipError_t launch_reproj_prepare(const float *uv, const float *K, const float *conf, int N, int J, float *geom,
                                 float *conf_clamped, hipStream_t st);
hipError_t launch_reproj_grad(const float *x, const float *geom, float *T, int solve_T, float *g, int B, int N,
                              int J, long long row_offset, hipStream_t st);
// the second half of post_dense on small batches: x' = a x + c (((q0 + q1) + q2) + q3 + bias) on the padded state (sde != 0)
// or eps = ((q0 + q1) + q2) + q3 + bias -> eps_out [Bp][64] (sde == 0), then - geom != nullptr - the reprojection correction of
// the next iteration on rows < B, exactly as the fused epilogue of the large-batch tile does it
hipError_t launch_post_reduce(float *xpad, const float *partial, const float *bias, float sde_a, float sde_c, int sde,
                              float *eps_out, const float *geom, float *T, int solve_T, int B, int Bp, int N, long long row0,
                              hipStream_t st);
hipError_t launch_reproj_step_padded(float *xpad, const float *geom, float *T, int solve_T, int B, int N,
                                     long long row0, hipStream_t st);
hipError_t launch_posemb(const float *t, int S, int Sp, float label_scale, float *pe, hipStream_t st);
hipError_t launch_ipo_fit(const float *x0, const float *uv, const float *K, const int *h_keylist, int k,
                          int axes_mask, float ipo_T, float min_scale, float max_scale, int iters,
                          double normaliser, float *R, float *T, float *q, float *scale, float *state, int it_begin,
                          int B, int N, int J, long long row_offset, hipStream_t st);
hipError_t launch_rotate_init(const float *x0, const float *R, float *x, int B, int N, int J, long long row_offset,
                              hipStream_t st);
hipError_t launch_pose_min(const double *err, int B, int N, long long row_offset, double *best, int *best_h, hipStream_t st);
hipError_t launch_reproj_degenerate(const float *geom, int N, int J, int *d_count, hipStream_t st);
hipError_t launch_min_mpjpe(const float *pred, const double *gt, int B, int N, int J, long long row_offset,
                            int procrustes, double *err, double *best, int *best_h, hipStream_t st);

}  // namespace zedo
