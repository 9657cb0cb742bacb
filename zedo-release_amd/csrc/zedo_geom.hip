// Geometry side of the ZeDO loop for gfx950: reprojection correction (gradient_field_gen),
// the in-register IPO fit, and the small packing / table kernels around them.
// These kernels are HBM/latency bound (about 1.5 kFLOP and a few hundred bytes per row):
// global traffic is coalesced through LDS tiles, each lane then owns one pose row.
#include "zedo_internal.h"

#include <atomic>
#include <cstdlib>

namespace zedo {

// f32x4, GEOM_F and reproj_row<J>: zedo_internal.h

// ------------------------------------------------------------------------------------------
// pose rows [B][D] <-> padded rows [Bp][XLD] (pad columns and pad rows are zero)
// ------------------------------------------------------------------------------------------
__global__ void pack_rows_kernel(const float *__restrict__ x, float *__restrict__ xpad, int B, int Bp, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Bp * XLD) return;
    const int r = (int)(i / XLD), c = (int)(i % XLD);
    xpad[i] = (r < B && c < D) ? x[(size_t)r * D + c] : 0.0f;
}

__global__ void unpack_rows_kernel(const float *__restrict__ xpad, float *__restrict__ x, int B, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * D) return;
    const int r = (int)(i / D), c = (int)(i % D);
    x[i] = xpad[(size_t)r * XLD + c];
}

hipError_t launch_pack_rows(const float *x, float *xpad, int B, int Bp, int D, hipStream_t st) {
    const size_t n = (size_t)Bp * XLD;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xpad, B, Bp, D);
    return hipGetLastError();
}
hipError_t launch_unpack_rows(const float *xpad, float *x, int B, int D, hipStream_t st) {
    const size_t n = (size_t)B * D;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xpad, x, B, D);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// step-invariant rays (reference simple_zeroshot_opt.py:61-71,99 and conf clamp :64-66)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv3x3(const float *K, double *Ki) {
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * Bc + c * C;
    const double id = 1.0 / det;
    Ki[0] = A * id;  Ki[1] = -(b * i - c * h) * id;  Ki[2] = (b * f - c * e) * id;
    Ki[3] = Bc * id; Ki[4] = (a * i - c * g) * id;   Ki[5] = -(a * f - c * d) * id;
    Ki[6] = C * id;  Ki[7] = -(a * h - b * g) * id;  Ki[8] = (a * e - b * d) * id;
}

__global__ void reproj_prepare_kernel(const float *__restrict__ uv, const float *__restrict__ K,
                                      const float *__restrict__ conf, int N, int J, float *__restrict__ geom,
                                      float *__restrict__ conf_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * J) return;
    const int n = i / J;
    double Ki[9];
    inv3x3(K + (size_t)n * 9, Ki);
    const double u = uv[(size_t)i * 2], v = uv[(size_t)i * 2 + 1];
    const double rz = Ki[6] * u + Ki[7] * v + Ki[8];
    const double rx = (Ki[0] * u + Ki[1] * v + Ki[2]) / rz;
    const double ry = (Ki[3] * u + Ki[4] * v + Ki[5]) / rz;
    const double rn = 1.0 / sqrt(rx * rx + ry * ry + 1.0);
    float W = 1.0f;
    if (conf) {
        float c = conf[i];
        if (c > 1.0f) c = 1.0f;      // NaN stays NaN, exactly like the masked assignment of the reference
        if (c < 1e-4f) c = 1e-4f;
        if (conf_out) conf_out[i] = c;
        const float w = c * c;       // rows of A and b are scaled by conf^2 (:85-88) ...
        W = w * w;                   // ... so the normal equations weight each residual by conf^4
#ifdef ZEDO_MUT_CONF2       // tools/mutation_check.py only
        W = w;
#endif
    }
    float *g = geom + (size_t)i * GEOM_F;
    g[0] = (float)rx; g[1] = (float)ry; g[2] = W; g[3] = 0.0f;
    g[4] = (float)(rx * rn); g[5] = (float)(ry * rn); g[6] = (float)rn; g[7] = 0.0f;
}

hipError_t launch_reproj_prepare(const float *uv, const float *K, const float *conf, int N, int J, float *geom,
                                 float *conf_clamped, hipStream_t st) {
    const int n = N * J;
    hipLaunchKernelGGL(reproj_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, st, uv, K, conf, N, J, geom,
                       conf_clamped);
    return hipGetLastError();
}

// The denominator of reproj_row's T_z for every pose - it depends on the rays and weights only - with the same fp32
// operations in the same order; poses where it is exactly zero have a singular normal matrix (zedo_reproj_degenerate).
__global__ void reproj_degenerate_kernel(const float *__restrict__ geom, int N, int J, int *__restrict__ count) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *gp = geom + (size_t)n * J * GEOM_F;
    float sw = 0.f, srx = 0.f, sry = 0.f;
    for (int j = 0; j < J; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(gp + j * GEOM_F);
        sw += a[2]; srx += a[2] * a[0]; sry += a[2] * a[1];
    }
    const float iw = 1.0f / sw;
    const float mrx = srx * iw, mry = sry * iw;
    float den = 0.f;
    for (int j = 0; j < J; ++j) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(gp + j * GEOM_F);
        const float dx = a[0] - mrx, dy = a[1] - mry;
        den += a[2] * (dx * dx + dy * dy);
    }
    if (den == 0.f) atomicAdd(count, 1);
}

hipError_t launch_reproj_degenerate(const float *geom, int N, int J, int *d_count, hipStream_t st) {
    hipLaunchKernelGGL(reproj_degenerate_kernel, dim3((N + 255) / 256), dim3(256), 0, st, geom, N, J, d_count);
    return hipGetLastError();
}

// reproj_row<J> (gradient_field_gen for one pose row held in registers) lives in zedo_internal.h: the fused
// post_dense + reprojection + pre_dense kernel of zedo_gemm.hip runs the same source.

// Standalone surface op: x [B][J*3] -> g [B][J*3] (+T).  128 rows per workgroup, tile staged through
// LDS with unit-stride global accesses; odd row stride (51) keeps the per-lane row reads conflict free.
template <int J>
__global__ __launch_bounds__(128) void reproj_grad_kernel(const float *__restrict__ x, const float *__restrict__ geom,
                                                          float *__restrict__ T, int solve, float *__restrict__ gout,
                                                          int B, int N, long long row_offset) {
    constexpr int D = J * 3, R = 128;
    __shared__ float sx[R * D];
    const int row0 = blockIdx.x * R, tid = threadIdx.x;
    const int rows = min(R, B - row0);
    const float *src = x + (size_t)row0 * D;
    for (int i = tid; i < rows * D; i += R) sx[i] = src[i];
    __syncthreads();
    float xr[D], gr[D], Tr[3];
    if (tid < rows) {
        const int b = row0 + tid;
#pragma unroll
        for (int c = 0; c < D; ++c) xr[c] = sx[tid * D + c];
        Tr[0] = T[(size_t)b * 3]; Tr[1] = T[(size_t)b * 3 + 1]; Tr[2] = T[(size_t)b * 3 + 2];
        const int n = (int)((row_offset + b) % N);
        reproj_row<J>(xr, geom + (size_t)n * J * GEOM_F, Tr, solve != 0, gr);
        if (solve) { T[(size_t)b * 3] = Tr[0]; T[(size_t)b * 3 + 1] = Tr[1]; T[(size_t)b * 3 + 2] = Tr[2]; }
    }
    __syncthreads();
    if (tid < rows) {
#pragma unroll
        for (int c = 0; c < D; ++c) sx[tid * D + c] = gr[c];
    }
    __syncthreads();
    float *dst = gout + (size_t)row0 * D;
    for (int i = tid; i < rows * D; i += R) dst[i] = sx[i];
}

hipError_t launch_reproj_grad(const float *x, const float *geom, float *T, int solve_T, float *g, int B, int N,
                              int J, long long row_offset, hipStream_t st) {
    if (J != 17) return hipErrorInvalidValue;
    hipLaunchKernelGGL(reproj_grad_kernel<17>, dim3((B + 127) / 128), dim3(128), 0, st, x, geom, T, solve_T, g, B, N,
                       row_offset);
    return hipGetLastError();
}

// Fused-loop variant on the padded state: xpad[row][64] += g, in place.  64 rows (one wavefront) per workgroup:
// the 16 KB tile moves as 16-byte, fully coalesced accesses through LDS (row stride 68 floats so
// that each lane's ds_read_b128 / ds_write_b128 of its own row is bank-conflict free).
template <int J>
__global__ __launch_bounds__(64) void reproj_step_kernel(float *__restrict__ xpad, const float *__restrict__ geom,
                                                         float *__restrict__ T, int solve, int B, int N,
                                                         long long row_offset) {
    constexpr int D = J * 3, R = BATCH_PAD, LD = XLD + 4, NV = (D + 3) / 4;
    __shared__ __attribute__((aligned(16))) float sx[R * LD];
    const int row0 = blockIdx.x * R, tid = threadIdx.x;
    float *base = xpad + (size_t)row0 * XLD;  // Bp is a multiple of BATCH_PAD: the whole tile exists
#pragma unroll
    for (int it = 0; it < XLD / 4; ++it) {
        const int idx = it * R + tid, r = idx >> 4, c4 = idx & 15;
        *reinterpret_cast<f32x4 *>(sx + r * LD + c4 * 4) = *reinterpret_cast<const f32x4 *>(base + (size_t)idx * 4);
    }
    __syncthreads();
    const int b = row0 + tid;
    if (b < B) {
        float xr[NV * 4], gr[D], Tr[3];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(sx + tid * LD + v * 4);
            xr[4 * v] = t[0]; xr[4 * v + 1] = t[1]; xr[4 * v + 2] = t[2]; xr[4 * v + 3] = t[3];
        }
        Tr[0] = T[(size_t)b * 3]; Tr[1] = T[(size_t)b * 3 + 1]; Tr[2] = T[(size_t)b * 3 + 2];
        const int n = (int)((row_offset + b) % N);
        reproj_row<J>(xr, geom + (size_t)n * J * GEOM_F, Tr, solve != 0, gr);
        if (solve) { T[(size_t)b * 3] = Tr[0]; T[(size_t)b * 3 + 1] = Tr[1]; T[(size_t)b * 3 + 2] = Tr[2]; }
#pragma unroll
        for (int c = 0; c < D; ++c) xr[c] += gr[c];  // denoise_x += joint_gradient (run/opt_main.py:208)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            f32x4 t = {xr[4 * v], xr[4 * v + 1], xr[4 * v + 2], xr[4 * v + 3]};
            *reinterpret_cast<f32x4 *>(sx + tid * LD + v * 4) = t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < XLD / 4; ++it) {
        const int idx = it * R + tid, r = idx >> 4, c4 = idx & 15;
        *reinterpret_cast<f32x4 *>(base + (size_t)idx * 4) = *reinterpret_cast<const f32x4 *>(sx + r * LD + c4 * 4);
    }
}

// Second half of post_dense on small batches (zedo_gemm.hip, EPI_PARTIAL): the four K-quarter sums of every row are
// combined in the one order every launch shape uses - ((q0 + q1) + q2) + q3 - then bias, then either the SDE update of
// the padded state (x' = a x + c eps: "o = acc + b; o *= c; x' = fma(a, x, o)", the statements of the fused epilogue) or
// eps itself, then - geom given - the reprojection correction of the next iteration, one lane per row like
// reproj_step_kernel.  64 rows per workgroup; the state tile moves through LDS as coalesced 16-byte accesses.
template <int J>
__global__ __launch_bounds__(64) void post_reduce_kernel(float *__restrict__ xpad, const float *__restrict__ partial,
                                                         const float *__restrict__ bias, float sde_a, float sde_c, int sde,
                                                         float *__restrict__ eps_out, const float *__restrict__ geom,
                                                         float *__restrict__ T, int solve, int B, int N, long long row_offset) {
    constexpr int D = J * 3, R = BATCH_PAD, LD = XLD + 4, NV = (D + 3) / 4;
    __shared__ __attribute__((aligned(16))) float sx[R * LD];
    __shared__ __attribute__((aligned(16))) float sb[XLD];
    const int row0 = blockIdx.x * R, tid = threadIdx.x;
    float *base = (sde ? xpad : eps_out) + (size_t)row0 * XLD;  // Bp is a multiple of BATCH_PAD: the whole tile exists
    sb[tid] = bias[tid];
    if (sde) {
#pragma unroll
        for (int it = 0; it < XLD / 4; ++it) {
            const int idx = it * R + tid, r = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4 *>(sx + r * LD + c4 * 4) = *reinterpret_cast<const f32x4 *>(base + (size_t)idx * 4);
        }
    }
    __syncthreads();
    const int b = row0 + tid;
    {
        // quarter sums of row b: partial[(b / 32) * 4 + q][b % 32][64]
        const float *p0 = partial + ((size_t)(b >> 5) * 4 * 32 + (b & 31)) * XLD;
        float *xr = sx + tid * LD;
#pragma unroll
        for (int v = 0; v < XLD / 4; ++v) {
            const f32x4 q0 = *reinterpret_cast<const f32x4 *>(p0 + v * 4);
            const f32x4 q1 = *reinterpret_cast<const f32x4 *>(p0 + 32 * XLD + v * 4);
            const f32x4 q2 = *reinterpret_cast<const f32x4 *>(p0 + 2 * 32 * XLD + v * 4);
#ifdef ZEDO_MUT_POST_Q3       // tools/mutation_check.py only
            const f32x4 q3 = q2;
#else
            const f32x4 q3 = *reinterpret_cast<const f32x4 *>(p0 + 3 * 32 * XLD + v * 4);
#endif
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(sb + v * 4);
            f32x4 x4 = {0.f, 0.f, 0.f, 0.f};
            if (sde) x4 = *reinterpret_cast<const f32x4 *>(xr + v * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = ((q0[e] + q1[e]) + q2[e]) + q3[e];
                t = t + bb[e];
                if (sde) { t = t * sde_c; t = __builtin_fmaf(sde_a, x4[e], t); }
                o[e] = t;
            }
            *reinterpret_cast<f32x4 *>(xr + v * 4) = o;
        }
    }
    if (geom != nullptr && b < B) {
        float xr[NV * 4], gr[D], Tr[3];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(sx + tid * LD + v * 4);
            xr[4 * v] = t[0]; xr[4 * v + 1] = t[1]; xr[4 * v + 2] = t[2]; xr[4 * v + 3] = t[3];
        }
        Tr[0] = T[(size_t)b * 3]; Tr[1] = T[(size_t)b * 3 + 1]; Tr[2] = T[(size_t)b * 3 + 2];
        const int n = (int)((row_offset + b) % N);
        reproj_row<J>(xr, geom + (size_t)n * J * GEOM_F, Tr, solve != 0, gr);
        if (solve) { T[(size_t)b * 3] = Tr[0]; T[(size_t)b * 3 + 1] = Tr[1]; T[(size_t)b * 3 + 2] = Tr[2]; }
#pragma unroll
        for (int c = 0; c < D; ++c) xr[c] += gr[c];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            f32x4 t = {xr[4 * v], xr[4 * v + 1], xr[4 * v + 2], xr[4 * v + 3]};
            *reinterpret_cast<f32x4 *>(sx + tid * LD + v * 4) = t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < XLD / 4; ++it) {
        const int idx = it * R + tid, r = idx >> 4, c4 = idx & 15;
        *reinterpret_cast<f32x4 *>(base + (size_t)idx * 4) = *reinterpret_cast<const f32x4 *>(sx + r * LD + c4 * 4);
    }
}

hipError_t launch_post_reduce(float *xpad, const float *partial, const float *bias, float sde_a, float sde_c, int sde,
                              float *eps_out, const float *geom, float *T, int solve_T, int B, int Bp, int N, long long row0,
                              hipStream_t st) {
    if (Bp % BATCH_PAD || !partial || !bias || (sde ? !xpad : !eps_out)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(post_reduce_kernel<17>, dim3(Bp / BATCH_PAD), dim3(BATCH_PAD), 0, st, xpad, partial, bias, sde_a, sde_c, sde,
                       eps_out, geom, T, solve_T, B, N > 0 ? N : 1, row0);
    return hipGetLastError();
}

hipError_t launch_reproj_step_padded(float *xpad, const float *geom, float *T, int solve_T, int B, int N,
                                     long long row0, hipStream_t st) {
    hipLaunchKernelGGL(reproj_step_kernel<17>, dim3((B + BATCH_PAD - 1) / BATCH_PAD), dim3(BATCH_PAD), 0, st, xpad, geom, T,
                       solve_T, B, N, row0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// sinusoidal timestep embedding (model.py:81-95) for labels = 999 t  (utils.py:762)
// ------------------------------------------------------------------------------------------
__global__ void posemb_kernel(const float *__restrict__ t, int S, int Sp, float label_scale, float *__restrict__ pe) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Sp * EMB) return;
    const int s = i / EMB, c = i % EMB;
    float o = 0.0f;
    if (s < S) {
        constexpr int half = EMB / 2;
        const float emb = (float)(-9.210340371976184 / (half - 1));  // -log(10000)/(half-1), rounded like the fp32 mul
        const int k = c < half ? c : c - half;
        const float freq = expf((float)k * emb);
        const float arg = (t[s] * label_scale) * freq;
        o = c < half ? sinf(arg) : cosf(arg);
    }
    pe[i] = o;
}

hipError_t launch_posemb(const float *t, int S, int Sp, float label_scale, float *pe, hipStream_t st) {
    const int n = Sp * EMB;
    hipLaunchKernelGGL(posemb_kernel, dim3((n + 255) / 256), dim3(256), 0, st, t, S, Sp, label_scale, pe);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// IPO: RotOpt + L1 reprojection loss + Adam, all iterations in registers
// (run/opt_main.py:177-195, simple_zeroshot_opt.py:8-31, utils.py:59-88)
// ------------------------------------------------------------------------------------------
struct AdamP {
    float p, m, v;
    __device__ __forceinline__ void step(float g, float step_size, float bc2_sqrt) {
        // torch.optim.Adam single-tensor update: lerp, mul+addcmul, sqrt/bc2_sqrt + eps, addcdiv
        m = m + (g - m) * 0.1f;                       // 1 - beta1, beta1 = 0.9
        v = v * 0.999f + 0.001f * g * g;              // beta2 = 0.999
        const float denom = sqrtf(v) / bc2_sqrt + 1e-8f;
        p = p + (-step_size * m) / denom;
    }
};

__device__ __forceinline__ float sgnf(float e) { return (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f); }

constexpr int IPO_KMAX = 17;
constexpr int IPO_TABLE = 2048;      // Adam bias-correction terms of the first IPO_TABLE iterations (the reference runs 500)

struct IpoKeys { int j[IPO_KMAX]; };   // IPO_keylist travels as a kernel argument: no device buffer, no sync

// step_size = lr / (1 - beta1^t) and sqrt(1 - beta2^t) of torch.optim.Adam for t = 1 .. IPO_TABLE, formed on the host by
// the statements the kernel itself uses beyond the table (running double products, one division, one square root:
// correctly rounded on both sides, so the table changes no bit) - two double-precision long-latency operations per
// iteration that every lane of the fit would otherwise repeat.  The table is part of the code object: the Makefile runs
// gen_adam_table.cpp (those statements, on the build host) into zedo_adam_table.inc as exact hexadecimal float literals.  No
// upload at run time: the first zedo_ipo_fit of a device neither blocks nor races with a fit on another stream or host thread,
// and is legal under stream capture (rounds 1-4 copied the table to the symbol inside the first fit).
__constant__ float c_adam_step[IPO_TABLE] = {
#define ZEDO_ADAM_STEP
#include "zedo_adam_table.inc"
#undef ZEDO_ADAM_STEP
};
__constant__ float c_adam_bc2s[IPO_TABLE] = {
#define ZEDO_ADAM_BC2S
#include "zedo_adam_table.inc"
#undef ZEDO_ADAM_BC2S
};

// Sum over the 32 lanes of a half-wave in ONE fixed order: four pairing levels inside each 16-lane row as DPP operand
// modifiers of the add itself (row_mirror: i <-> 15 - i, row_half_mirror: i <-> 7 - i, quad_perm xor 2, quad_perm xor 1 - no
// cross-lane instruction, no LDS-pipe latency), then the two rows of the half-wave by one ds_swizzle (xor 16).  Every level
// pairs lanes SYMMETRICALLY and addition commutes, so both partners form the same bits and all 32 lanes end with the same
// value: the optimiser state stays replicated without a broadcast.  (First version of round 4: five ds_swizzle levels -
// 50 LDS-pipe round trips per iteration were most of the iteration's latency.)
template <int M>
__device__ __forceinline__ float swz_xor(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (M << 10) | 0x1F));
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int C>      // every lane of a 32-lane group reads lane C of its group (and-mask 0, or-mask C)
__device__ __forceinline__ float bcast(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (C << 5)));
}
__device__ __forceinline__ float half_sum(float v) {
    v = v + dpp<0x140>(v);      // row_mirror
    v = v + dpp<0x141>(v);      // row_half_mirror
    v = v + dpp<0x4E>(v);       // quad_perm [2,3,0,1]
    v = v + dpp<0xB1>(v);       // quad_perm [1,0,3,2]
    v = v + swz_xor<16>(v);
    return v;
}

// ---- pieces shared by the two IPO kernels (the SAME statements, so that both produce the same bits) -------------------
struct IpoRot { float R00, R01, R02, R10, R11, R12, R20, R21, R22, Tx, Ty, Tz, ts; };

// R = I + ts * Q(q) (quaternion_to_matrix, utils.py:59-88) and T = T0 * clamp(scale) (simple_zeroshot_opt.py:26-31)
__device__ __forceinline__ IpoRot ipo_rot(float r, float i, float j, float kk, float scp, float min_s, float max_s, const float (&T0)[3]) {
    IpoRot o;
    const float n2 = r * r + i * i + j * j + kk * kk;
    const float ts = 2.0f / n2;
    o.ts = ts;
    o.R00 = 1.f - ts * (j * j + kk * kk); o.R01 = ts * (i * j - kk * r); o.R02 = ts * (i * kk + j * r);
    o.R10 = ts * (i * j + kk * r); o.R11 = 1.f - ts * (i * i + kk * kk); o.R12 = ts * (j * kk - i * r);
    o.R20 = ts * (i * kk - j * r); o.R21 = ts * (j * kk + i * r); o.R22 = 1.f - ts * (i * i + j * j);
    const float scc = fminf(fmaxf(scp, min_s), max_s);
    o.Tx = T0[0] * scc; o.Ty = T0[1] * scc; o.Tz = T0[2] * scc;
    return o;
}

// One key joint: forward, L1 sign, backward to the camera-frame point; its ten contributions to the gradient sums
// t[0] -> d/d scale, t[1..9] -> G00 G01 G02 G10 G11 G12 G20 G21 G22 (G = sum_j g_p x^T)
__device__ __forceinline__ void ipo_joint_terms(const float (&K)[9], const float (&T0)[3], const IpoRot &o, float x, float y, float z,
                                                float cu, float cv, float inv_norm, float (&t)[10]) {
    const float px = o.R00 * x + o.R01 * y + o.R02 * z + o.Tx;
    const float py = o.R10 * x + o.R11 * y + o.R12 * z + o.Ty;
    const float pz = o.R20 * x + o.R21 * y + o.R22 * z + o.Tz;
    const float w0 = K[0] * px + K[1] * py + K[2] * pz;
    const float w1 = K[3] * px + K[4] * py + K[5] * pz;
    const float w2 = K[6] * px + K[7] * py + K[8] * pz;
    const float gu = sgnf(w0 / w2 - cu) * inv_norm;   // d mean|e| / du
    const float gv = sgnf(w1 / w2 - cv) * inv_norm;
    const float gw0 = gu / w2, gw1 = gv / w2;
    const float gw2 = -gu * ((w0 / w2) / w2) - gv * ((w1 / w2) / w2);  // torch div backward form
    const float gpx = K[0] * gw0 + K[3] * gw1 + K[6] * gw2;       // K^T g_w
    const float gpy = K[1] * gw0 + K[4] * gw1 + K[7] * gw2;
    const float gpz = K[2] * gw0 + K[5] * gw1 + K[8] * gw2;
    t[0] = gpx * T0[0] + gpy * T0[1] + gpz * T0[2];
    t[1] = gpx * x; t[2] = gpx * y; t[3] = gpx * z;
    t[4] = gpy * x; t[5] = gpy * y; t[6] = gpy * z;
    t[7] = gpz * x; t[8] = gpz * y; t[9] = gpz * z;
}

// gradients of the four quaternion components from the summed G (dL/d two_s = <G, Q>, d two_s / d q_c = -two_s^2 q_c)
__device__ __forceinline__ void ipo_quat_grads(const float (&S)[10], float r, float i, float j, float kk, float ts, float &gr, float &gi,
                                               float &gj, float &gk) {
    const float G00 = S[1], G01 = S[2], G02 = S[3], G10 = S[4], G11 = S[5], G12 = S[6], G20 = S[7], G21 = S[8], G22 = S[9];
    const float gts = G00 * -(j * j + kk * kk) + G01 * (i * j - kk * r) + G02 * (i * kk + j * r) +
                      G10 * (i * j + kk * r) + G11 * -(i * i + kk * kk) + G12 * (j * kk - i * r) +
                      G20 * (i * kk - j * r) + G21 * (j * kk + i * r) + G22 * -(i * i + j * j);
    const float dts = -gts * ts * ts;
    gr = ts * (-G01 * kk + G02 * j + G10 * kk - G12 * i - G20 * j + G21 * i) + dts * r;
    gi = ts * (G01 * j + G02 * kk + G10 * j - 2.f * G11 * i - G12 * r + G20 * kk + G21 * r - 2.f * G22 * i) + dts * i;
    gj = ts * (-2.f * G00 * j + G01 * i + G02 * r + G10 * i + G12 * kk - G20 * r + G21 * kk - 2.f * G22 * j) + dts * j;
    gk = ts * (-2.f * G00 * kk - G01 * r + G02 * i + G10 * r - 2.f * G11 * kk + G12 * j + G20 * i + G21 * j) + dts * kk;
}

// T0 = ipo_T * normalise(Kinv [u0 v0 1])  (opt_main.py:177-179); joint 0 is the pelvis
__device__ __forceinline__ void ipo_T0(const float (&K)[9], double u, double v, float ipo_T, float (&T0)[3]) {
    double Ki[9];
    inv3x3(K, Ki);
    const double tx = Ki[0] * u + Ki[1] * v + Ki[2], ty = Ki[3] * u + Ki[4] * v + Ki[5], tz = Ki[6] * u + Ki[7] * v + Ki[8];
    const double in = (double)ipo_T / sqrt(tx * tx + ty * ty + tz * tz);
    T0[0] = (float)(tx * in); T0[1] = (float)(ty * in); T0[2] = (float)(tz * in);
}

__device__ __forceinline__ void ipo_adam_terms(int ta, double &b1p, double &b2p, float &step_size, float &bc2s) {
    b1p *= 0.9; b2p *= 0.999;
    if (ta < IPO_TABLE) { step_size = c_adam_step[ta]; bc2s = c_adam_bc2s[ta]; }
    else { step_size = (float)(0.1 / (1.0 - b1p)); bc2s = (float)sqrt(1.0 - b2p); }
}

__device__ __forceinline__ void ipo_store(const AdamP &qr, const AdamP &qi, const AdamP &qj, const AdamP &qk, const AdamP &sc, const float (&T0)[3],
                                          float min_s, float max_s, int b, float *Rout, float *Tout, float *qout, float *sout, float *stp) {
    const float r = qr.p, i = qi.p, j = qj.p, kk = qk.p;
    const float ts = 2.0f / (r * r + i * i + j * j + kk * kk);
    float *Ro = Rout + (size_t)b * 9;
    Ro[0] = 1.f - ts * (j * j + kk * kk); Ro[1] = ts * (i * j - kk * r); Ro[2] = ts * (i * kk + j * r);
    Ro[3] = ts * (i * j + kk * r); Ro[4] = 1.f - ts * (i * i + kk * kk); Ro[5] = ts * (j * kk - i * r);
    Ro[6] = ts * (i * kk - j * r); Ro[7] = ts * (j * kk + i * r); Ro[8] = 1.f - ts * (i * i + j * j);
    const float scc = fminf(fmaxf(sc.p, min_s), max_s);
    Tout[(size_t)b * 3] = T0[0] * scc; Tout[(size_t)b * 3 + 1] = T0[1] * scc; Tout[(size_t)b * 3 + 2] = T0[2] * scc;
    if (qout) { qout[(size_t)b * 4] = r; qout[(size_t)b * 4 + 1] = i; qout[(size_t)b * 4 + 2] = j; qout[(size_t)b * 4 + 3] = kk; }
    if (sout) sout[b] = sc.p;
    if (stp) {
        stp[0] = qr.p; stp[1] = qi.p; stp[2] = qj.p; stp[3] = qk.p; stp[4] = sc.p;
        stp[5] = qr.m; stp[6] = qi.m; stp[7] = qj.m; stp[8] = qk.m; stp[9] = sc.m;
        stp[10] = qr.v; stp[11] = qi.v; stp[12] = qj.v; stp[13] = qk.v; stp[14] = sc.v;
    }
}

// One pose-hypothesis row per 32-lane HALF-WAVE, one key joint per lane (k <= 17 of the 32 lanes carry a joint, the
// others contribute exact zeros): an iteration is one joint's forward / backward (~190 instructions with its six IEEE
// divisions) + ten half-wave sums + one Adam update, instead of k joints in sequence on one lane - the 500-iteration
// chain of the reference's 17-joint 3DPW key list shortens ~6x (3.2 ms -> 0.49 ms for any batch that does not fill the
// chip).  Large batches run ipo_row_kernel below: the same statements and the same pairing tree on one lane per row.
// The sums over joints have ONE order (half_sum) whatever the batch, shard, launch or kernel: rows are bit-identical.
__global__ __launch_bounds__(64) void ipo_kernel(const float *__restrict__ x0, const float *__restrict__ uv,
                                                  const float *__restrict__ Kmat, const IpoKeys keylist,
                                                  int k, int axes_mask, float ipo_T, float min_s, float max_s,
                                                  int iters, float inv_norm, float *__restrict__ Rout,
                                                  float *__restrict__ Tout, float *__restrict__ qout,
                                                  float *__restrict__ sout, float *__restrict__ state,
                                                  int it_begin, double b1p0, double b2p0, int B, int N, int J,
                                                  long long row_offset) {
    __shared__ int s_kl[32];
    const int lane = threadIdx.x, jl = lane & 31;
    if (lane < 32) s_kl[lane] = lane < k ? keylist.j[lane] : keylist.j[0];
    __syncthreads();
    const int b_raw = blockIdx.x * 2 + (lane >> 5);
    const bool row_ok = b_raw < B;
    const int b = row_ok ? b_raw : B - 1;     // a half-wave without a row shadows the last row (uniform control flow) and stores nothing
#ifdef ZEDO_MUT_IPO_JOINT     // tools/mutation_check.py only: the last joint of a 17-joint key list is lost
    const bool jact = jl < k && jl < 16;
#else
    const bool jact = jl < k;
#endif
    const long long gb = row_offset + b;
    const int n = (int)(gb % N), h = (int)(gb / N);
    const int jn = s_kl[jl];
    const float *xh = x0 + ((size_t)h * J + jn) * 3;
    const float x = xh[0], y = xh[1], z = xh[2];
    const float cu = uv[((size_t)n * J + jn) * 2], cv = uv[((size_t)n * J + jn) * 2 + 1];
    float K[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) K[i] = Kmat[(size_t)n * 9 + i];
    float T0[3];
    ipo_T0(K, uv[(size_t)n * J * 2], uv[(size_t)n * J * 2 + 1], ipo_T, T0);
    AdamP qr{1.f, 0.f, 0.f}, qi{0.f, 0.f, 0.f}, qj{0.f, 0.f, 0.f}, qk{0.f, 0.f, 0.f}, sc{1.f, 0.f, 0.f};
    const bool ax = axes_mask & 1, ay = axes_mask & 2, az = axes_mask & 4;
    // resumable fit: state[b] = (p[5], exp_avg[5], exp_avg_sq[5]) in the order (rot_vect, x, y, z, scale) after
    // it_begin iterations; b1p0 / b2p0 = beta^it_begin.  it_begin == 0 starts from RotOpt's initial values.
    float *stp = state ? state + (size_t)b * 15 : nullptr;
    if (stp && it_begin > 0) {
        qr = AdamP{stp[0], stp[5], stp[10]}; qi = AdamP{stp[1], stp[6], stp[11]}; qj = AdamP{stp[2], stp[7], stp[12]};
        qk = AdamP{stp[3], stp[8], stp[13]}; sc = AdamP{stp[4], stp[9], stp[14]};
    }
    double b1p = b1p0, b2p = b2p0;
    // parameter c's (value, exp_avg, exp_avg_sq) live in lane c of the half-wave; lanes >= 5 carry a dummy
    AdamP mine;
    mine.p = jl == 0 ? qr.p : jl == 1 ? qi.p : jl == 2 ? qj.p : jl == 3 ? qk.p : sc.p;
    mine.m = jl == 0 ? qr.m : jl == 1 ? qi.m : jl == 2 ? qj.m : jl == 3 ? qk.m : sc.m;
    mine.v = jl == 0 ? qr.v : jl == 1 ? qi.v : jl == 2 ? qj.v : jl == 3 ? qk.v : sc.v;
    for (int it = 0; it < iters; ++it) {
        const float r = qr.p, i = qi.p, j = qj.p, kk = qk.p;
        const IpoRot o = ipo_rot(r, i, j, kk, sc.p, min_s, max_s, T0);
        float t[10], S[10];
        ipo_joint_terms(K, T0, o, x, y, z, cu, cv, inv_norm, t);       // this lane's joint
#pragma unroll
        for (int e = 0; e < 10; ++e) S[e] = half_sum(jact ? t[e] : 0.f);   // lanes without a joint: exact +0
        float gr, gi, gj, gk;
        ipo_quat_grads(S, r, i, j, kk, o.ts, gr, gi, gj, gk);
        const float gs = (sc.p >= min_s && sc.p <= max_s) ? S[0] : 0.f;   // clamp backward
        float step_size, bc2s;
        ipo_adam_terms(it_begin + it, b1p, b2p, step_size, bc2s);
        // The five Adam updates are the same statements on different data: lane c of the half-wave (c = 0..4 <-> rot_vect,
        // x, y, z, scale) carries parameter c's moments and takes its step, the new value is broadcast back (ds_swizzle
        // with and-mask 0: every lane of the 32-lane group reads lane c).  One update's IEEE sqrt and two divisions per
        // iteration instead of five - the same operations on the same values as updating all five in every lane.
        {
            const float g_c = jl == 0 ? gr : jl == 1 ? gi : jl == 2 ? gj : jl == 3 ? gk : gs;
            const bool upd = jl == 0 || (jl == 1 && ax) || (jl == 2 && ay) || (jl == 3 && az) || jl == 4;
            AdamP nxt = mine;
            nxt.step(g_c, step_size, bc2s);
            if (upd) mine = nxt;
            qr.p = bcast<0>(mine.p); qi.p = bcast<1>(mine.p); qj.p = bcast<2>(mine.p); qk.p = bcast<3>(mine.p); sc.p = bcast<4>(mine.p);
        }
    }
    // moments back from their lanes (the state record / resume interface is per row)
    qr.m = bcast<0>(mine.m); qi.m = bcast<1>(mine.m); qj.m = bcast<2>(mine.m); qk.m = bcast<3>(mine.m); sc.m = bcast<4>(mine.m);
    qr.v = bcast<0>(mine.v); qi.v = bcast<1>(mine.v); qj.v = bcast<2>(mine.v); qk.v = bcast<3>(mine.v); sc.v = bcast<4>(mine.v);
    if (row_ok && jl == 0) ipo_store(qr, qi, qj, qk, sc, T0, min_s, max_s, b, Rout, Tout, qout, sout, stp);
}

// ---- the same fit on ONE LANE per row, for batches that fill the chip --------------------------------------------------
// Above ~17 000 rows (17-joint key list; ~4 000 rows with the 3-joint list) the half-wave kernel, bound by VALU issue
// and repeating every per-row statement in 32 lanes, is the slower one; here a lane walks its row's K key joints itself and combines their ten
// terms in the pairing tree half_sum spells with cross-lane moves - slots 0..31, slot j = key joint j, exact +0 beyond K:
//   a[i] = t[i] + t[15 - i], b[i] = a[i] + a[7 - i], c[i] = b[i] + b[i ^ 2], d = c[0] + c[1] per 16-slot row; sum = d(row 0) + d(row 1)
// (what lanes 0 and 16 of a half-wave compute; every other lane computes the same bits, addition being commutative).
// K is a template parameter (one instantiation per key-list length 1 .. 17) so that the tree unrolls over registers.
constexpr int IPO_ROW_TB = 128;
template <int KJ, int S>
__device__ __forceinline__ void ipo_slot(const float (&K)[9], const float (&T0)[3], const IpoRot &o, const float (&x)[KJ], const float (&y)[KJ],
                                         const float (&z)[KJ], const float *cu, const float *cv, float inv_norm, float (&t)[10]) {
#ifdef ZEDO_MUT_IPO_JOINT     // tools/mutation_check.py only: the last joint of a 17-joint key list is lost (both kernels)
    constexpr bool live = S < KJ && S < 16;
#else
    constexpr bool live = S < KJ;
#endif
    if constexpr (live) {
        ipo_joint_terms(K, T0, o, x[S], y[S], z[S], cu[S * IPO_ROW_TB], cv[S * IPO_ROW_TB], inv_norm, t);
    } else {
#pragma unroll
        for (int e = 0; e < 10; ++e) t[e] = 0.f;
    }
}
#define IPO_SLOT_ARGS K, T0, o, x, y, z, cu, cv, inv_norm
#define IPO_SLOT_PARAMS const float (&K)[9], const float (&T0)[3], const IpoRot &o, const float (&x)[KJ], const float (&y)[KJ], \
                        const float (&z)[KJ], const float *cu, const float *cv, float inv_norm
template <int KJ, int BASE, int I>
__device__ __forceinline__ void ipo_pair_a(IPO_SLOT_PARAMS, float (&a)[10]) {
    float u[10], v[10];
    ipo_slot<KJ, BASE + I>(IPO_SLOT_ARGS, u);
    ipo_slot<KJ, BASE + 15 - I>(IPO_SLOT_ARGS, v);
#pragma unroll
    for (int e = 0; e < 10; ++e) a[e] = u[e] + v[e];
}
template <int KJ, int BASE, int I>
__device__ __forceinline__ void ipo_pair_b(IPO_SLOT_PARAMS, float (&b)[10]) {
    float u[10], v[10];
    ipo_pair_a<KJ, BASE, I>(IPO_SLOT_ARGS, u);
    ipo_pair_a<KJ, BASE, 7 - I>(IPO_SLOT_ARGS, v);
#pragma unroll
    for (int e = 0; e < 10; ++e) b[e] = u[e] + v[e];
}
template <int KJ, int BASE>
__device__ __forceinline__ void ipo_row_sum(IPO_SLOT_PARAMS, float (&d)[10]) {
    float b0[10], b1[10], b2[10], b3[10];
    ipo_pair_b<KJ, BASE, 0>(IPO_SLOT_ARGS, b0);
    ipo_pair_b<KJ, BASE, 2>(IPO_SLOT_ARGS, b2);
#pragma unroll
    for (int e = 0; e < 10; ++e) b0[e] = b0[e] + b2[e];                  // c[0]
    ipo_pair_b<KJ, BASE, 1>(IPO_SLOT_ARGS, b1);
    ipo_pair_b<KJ, BASE, 3>(IPO_SLOT_ARGS, b3);
#pragma unroll
    for (int e = 0; e < 10; ++e) d[e] = b0[e] + (b1[e] + b3[e]);         // c[0] + c[1]
}

template <int KJ>
__global__ __launch_bounds__(IPO_ROW_TB) void ipo_row_kernel(const float *__restrict__ x0, const float *__restrict__ uv,
                                                             const float *__restrict__ Kmat, const IpoKeys keylist, int axes_mask,
                                                             float ipo_T, float min_s, float max_s, int iters, float inv_norm,
                                                             float *__restrict__ Rout, float *__restrict__ Tout,
                                                             float *__restrict__ qout, float *__restrict__ sout,
                                                             float *__restrict__ state, int it_begin, double b1p0, double b2p0,
                                                             int B, int N, int J, long long row_offset) {
    __shared__ float s_cu[KJ][IPO_ROW_TB], s_cv[KJ][IPO_ROW_TB];
    const int tid = threadIdx.x;
    const int b = blockIdx.x * IPO_ROW_TB + tid;
    if (b >= B) return;
    const long long gb = row_offset + b;
    const int n = (int)(gb % N), h = (int)(gb / N);
    float x[KJ], y[KJ], z[KJ];
#pragma unroll
    for (int jj = 0; jj < KJ; ++jj) {
        const int jn = keylist.j[jj];
        const float *xh = x0 + ((size_t)h * J + jn) * 3;
        x[jj] = xh[0]; y[jj] = xh[1]; z[jj] = xh[2];
        s_cu[jj][tid] = uv[((size_t)n * J + jn) * 2];
        s_cv[jj][tid] = uv[((size_t)n * J + jn) * 2 + 1];
    }
    const float *cu = &s_cu[0][tid], *cv = &s_cv[0][tid];
    float K[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) K[i] = Kmat[(size_t)n * 9 + i];
    float T0[3];
    ipo_T0(K, uv[(size_t)n * J * 2], uv[(size_t)n * J * 2 + 1], ipo_T, T0);
    AdamP qr{1.f, 0.f, 0.f}, qi{0.f, 0.f, 0.f}, qj{0.f, 0.f, 0.f}, qk{0.f, 0.f, 0.f}, sc{1.f, 0.f, 0.f};
    const bool ax = axes_mask & 1, ay = axes_mask & 2, az = axes_mask & 4;
    float *stp = state ? state + (size_t)b * 15 : nullptr;
    if (stp && it_begin > 0) {
        qr = AdamP{stp[0], stp[5], stp[10]}; qi = AdamP{stp[1], stp[6], stp[11]}; qj = AdamP{stp[2], stp[7], stp[12]};
        qk = AdamP{stp[3], stp[8], stp[13]}; sc = AdamP{stp[4], stp[9], stp[14]};
    }
    double b1p = b1p0, b2p = b2p0;
    for (int it = 0; it < iters; ++it) {
        const float r = qr.p, i = qi.p, j = qj.p, kk = qk.p;
        const IpoRot o = ipo_rot(r, i, j, kk, sc.p, min_s, max_s, T0);
        float d0[10], d1[10], S[10];
        ipo_row_sum<KJ, 0>(IPO_SLOT_ARGS, d0);
        ipo_row_sum<KJ, 16>(IPO_SLOT_ARGS, d1);
#pragma unroll
        for (int e = 0; e < 10; ++e) S[e] = d0[e] + d1[e];
        float gr, gi, gj, gk;
        ipo_quat_grads(S, r, i, j, kk, o.ts, gr, gi, gj, gk);
        const float gs = (sc.p >= min_s && sc.p <= max_s) ? S[0] : 0.f;   // clamp backward
        float step_size, bc2s;
        ipo_adam_terms(it_begin + it, b1p, b2p, step_size, bc2s);
        qr.step(gr, step_size, bc2s);
        if (ax) qi.step(gi, step_size, bc2s);
        if (ay) qj.step(gj, step_size, bc2s);
        if (az) qk.step(gk, step_size, bc2s);
        sc.step(gs, step_size, bc2s);
    }
    ipo_store(qr, qi, qj, qk, sc, T0, min_s, max_s, b, Rout, Tout, qout, sout, stp);
}
#undef IPO_SLOT_ARGS
#undef IPO_SLOT_PARAMS

hipError_t launch_ipo_fit(const float *x0, const float *uv, const float *K, const int *h_keylist, int k,
                          int axes_mask, float ipo_T, float min_scale, float max_scale, int iters,
                          double normaliser, float *R, float *T, float *q, float *scale, float *state, int it_begin,
                          int B, int N, int J, long long row_offset, hipStream_t st) {
    if (k < 1 || k > IPO_KMAX) return hipErrorInvalidValue;
    double b1p0 = 1.0, b2p0 = 1.0;
    for (int i = 0; i < it_begin; ++i) { b1p0 *= 0.9; b2p0 *= 0.999; }   // the same running products the kernel forms
    IpoKeys keys{};
    for (int i = 0; i < k; ++i) keys.j[i] = h_keylist[i];
    // Which kernel: the half-wave kernel (latency: 0.49 ms) until the batch fills the chip, the lane-per-row kernel above that
    // (measured, profiles/ipo_kernels_r04.txt: the half-wave kernel takes 0.49 ms up to ~2 000 rows and then grows by 0.13 ms per
    // 1 000 rows whatever the key list; the row kernel takes 2.57 ms with 17 joints and 0.72 ms with 3 up to 65 536 rows).  Both produce the same bits (one pairing tree), so
    // the choice may depend on the LOCAL row count without breaking shard invariance.  ZEDO_IPO_KERNEL=half|row pins it.
    static const char *pin = getenv("ZEDO_IPO_KERNEL");
    // crossover in rows per CU of the CURRENT device.  Measured on 256 CUs: 17 408 rows (68 per CU) with 17 key joints, 4 096 (16 per CU)
    // with 3; in between the row kernel's time is linear in the number of key joints (0.32 + 0.13 k ms against the half-wave kernel's
    // 0.49 ms + 0.13 ms per 1 000 rows beyond 2 000): ~4 k + 4 rows per CU.  Round 6: EVERY key-list length 1 .. 17 has its lane-per-row
    // instantiation (until round 5 only the shipped 3 and 17: a custom ZeDO.IPO_keylist at configs[3]'s 3.5 M-row shard stayed on the
    // half-wave kernel, +0.46 s per pass).
    bool row = B >= num_cus() * (k == 17 ? 68 : 4 * k + 4);
    if (pin && pin[0] == 'h') row = false;
    if (pin && pin[0] == 'r') row = true;
    const float inv_norm = (float)(1.0 / normaliser);
    if (row) {
        const dim3 grid((B + IPO_ROW_TB - 1) / IPO_ROW_TB), block(IPO_ROW_TB);
#define ZEDO_IPO_ROW_CASE(KJ)                                                                                                                   \
        case KJ:                                                                                                                                \
            hipLaunchKernelGGL(ipo_row_kernel<KJ>, grid, block, 0, st, x0, uv, K, keys, axes_mask, ipo_T, min_scale, max_scale, iters, inv_norm, R, T, \
                               q, scale, state, it_begin, b1p0, b2p0, B, N, J, row_offset);                                                    \
            break;
        switch (k) {
            ZEDO_IPO_ROW_CASE(1) ZEDO_IPO_ROW_CASE(2) ZEDO_IPO_ROW_CASE(3) ZEDO_IPO_ROW_CASE(4) ZEDO_IPO_ROW_CASE(5) ZEDO_IPO_ROW_CASE(6)
            ZEDO_IPO_ROW_CASE(7) ZEDO_IPO_ROW_CASE(8) ZEDO_IPO_ROW_CASE(9) ZEDO_IPO_ROW_CASE(10) ZEDO_IPO_ROW_CASE(11) ZEDO_IPO_ROW_CASE(12)
            ZEDO_IPO_ROW_CASE(13) ZEDO_IPO_ROW_CASE(14) ZEDO_IPO_ROW_CASE(15) ZEDO_IPO_ROW_CASE(16) ZEDO_IPO_ROW_CASE(17)
        }
#undef ZEDO_IPO_ROW_CASE
        return hipGetLastError();
    }
    hipLaunchKernelGGL(ipo_kernel, dim3((B + 1) / 2), dim3(64), 0, st, x0, uv, K, keys, k,
                       axes_mask, ipo_T, min_scale, max_scale, iters, inv_norm, R, T, q, scale, state,
                       it_begin, b1p0, b2p0, B, N, J, row_offset);
    return hipGetLastError();
}

// x[b] = R[b] . x0[h]  (run/opt_main.py:201)
__global__ void rotate_init_kernel(const float *__restrict__ x0, const float *__restrict__ R, float *__restrict__ x,
                                   int B, int N, int J, long long row_offset) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * J) return;
    const int b = (int)(i / J), j = (int)(i % J);
    const int h = (int)((row_offset + b) / N);
    const float *r = R + (size_t)b * 9;
    const float *p = x0 + ((size_t)h * J + j) * 3;
    const float a = p[0], c = p[1], d = p[2];
    x[i * 3] = r[0] * a + r[1] * c + r[2] * d;
    x[i * 3 + 1] = r[3] * a + r[4] * c + r[5] * d;
    x[i * 3 + 2] = r[6] * a + r[7] * c + r[8] * d;
}

hipError_t launch_rotate_init(const float *x0, const float *R, float *x, int B, int N, int J, long long row_offset,
                              hipStream_t st) {
    const size_t n = (size_t)B * J;
    hipLaunchKernelGGL(rotate_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x0, R, x, B, N, J,
                       row_offset);
    return hipGetLastError();
}

}  // namespace zedo
