// Dense layers of the ZeDO score network (reference lib/algorithms/advanced/model.py:264-291) as
// exact-fp32 MFMA GEMMs with fused epilogues, written for gfx950 (MI355X) only.
//
//   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] + bias[n] )
//
// Mapping to v_mfma_f32_32x32x2_f32 (D[i][j] += A[i][k] B[k][j], 64 lanes, 16 accumulators):
//   i = output CHANNEL (rows of W), j = BATCH ROW (rows of X).  The product is computed
//   "transposed" on purpose: in the C/D layout lane l holds, for batch row j = l&31, the 16
//   channels (r&3) + 8*(r>>2) + 4*(l>>5) of one 32-channel block - and GroupNorm(32, 1024)
//   normalises exactly such blocks of 32 consecutive channels per row.  The group statistics
//   therefore need 16 in-register adds plus ONE exchange with lane l^32, and each lane ends up
//   with 4 runs of 4 consecutive channels -> 16-byte stores into the row-major activation.
//
// Both operands are K-contiguous in HBM (torch Linear weight [out][in], activations [row][in]),
// so both LDS tiles are [rows][BK] and every lane feeds 4 consecutive MFMAs from one
// ds_read_b128 (lane half kh supplies k = 8*kg + 4*kh + e for MFMA e; A and B agree on that
// order, so the sum over k is complete).  Row stride BK+4 floats makes the b128 reads and
// writes bank-conflict free.
//
// Pipeline: register-staged global->LDS double buffering, one barrier per 32-wide K tile
// (4096 MFMA cycles per wave between barriers at the 128x128 tile), 2 workgroups per CU so that
// one workgroup's epilogue/barrier bubbles hide under the other's MFMAs.
#include "zedo_internal.h"

namespace zedo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

__device__ __forceinline__ float silu_f(float y) { return y / (1.0f + expf(-y)); }

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM *WN * 64) void layer_kernel(LayerArgs a) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int TJ = TM / 32, TI = TN / 32;
    constexpr int RPP = NT / 8;  // tile rows covered per pass of 16-byte loads
    constexpr int LA = BN / RPP, LB = BM / RPP;
    static_assert(TM % 32 == 0 && TN % 32 == 0 && LA >= 1 && LB >= 1, "tile shape");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                    // [2][BN][LDS_LD]  W tile
    float *Bs = smem + 2 * BN * LDS_LD;  // [2][BM][LDS_LD]  X tile

    // XCD-aware, bijective block -> tile map: the hardware places block b on XCD b % 8; give every
    // XCD a contiguous range of tiles so that the column tiles of one row tile share one L2.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int ncol = a.N / BN;
    const int m0 = (lid / ncol) * BM, n0 = (lid % ncol) * BN;

    const int tid = threadIdx.x;
    const int chunk = tid & 7, lrow = tid >> 3;
    const float *Wg = a.W + (size_t)(n0 + lrow) * a.ldw + chunk * 4;
    const float *Xg = a.X + (size_t)(m0 + lrow) * a.ldx + chunk * 4;

    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, kh = lane >> 5;

    f32x4 ra[LA], rb[LB];
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < LA; ++p) ra[p] = *reinterpret_cast<const f32x4 *>(Wg + (size_t)p * RPP * a.ldw + kt * BK);
#pragma unroll
        for (int p = 0; p < LB; ++p) rb[p] = *reinterpret_cast<const f32x4 *>(Xg + (size_t)p * RPP * a.ldx + kt * BK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < LA; ++p)
            *reinterpret_cast<f32x4 *>(As + (buf * BN + lrow + p * RPP) * LDS_LD + chunk * 4) = ra[p];
#pragma unroll
        for (int p = 0; p < LB; ++p)
            *reinterpret_cast<f32x4 *>(Bs + (buf * BM + lrow + p * RPP) * LDS_LD + chunk * 4) = rb[p];
    };

    const int KT = a.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
        const float *Ab = As + (buf * BN + wn * TN + li) * LDS_LD + kh * 4;
        const float *Bb = Bs + (buf * BM + wm * TM + li) * LDS_LD + kh * 4;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            f32x4 af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * LDS_LD + kg * 8);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * LDS_LD + kg * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue: lane = (batch row j = li, channel half kh) -----------------
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int cbase = n0 + wn * TN + i * 32 + 4 * kh;  // channel of accumulator r: cbase + (r&3) + 8*(r>>2)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = m0 + wm * TM + j * 32 + li;
            float *orow = a.out + (size_t)m * a.ldo + cbase;
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + cbase + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = acc[i][j][4 * g + e] + b4[e];
            }
            if constexpr (EPI == EPI_GN_SILU || EPI == EPI_GN_SILU_RES) {
                // GroupNorm(32 groups of 32 channels), biased variance, eps 1e-5 (model.py:116,145,150)
                float s = 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) s += v[e];
                s += __shfl_xor(s, 32);
                const float mean = s * (1.0f / 32.0f);
                float qs = 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    v[e] -= mean;
                    qs += v[e] * v[e];
                }
                qs += __shfl_xor(qs, 32);
                const float rstd = 1.0f / sqrtf(qs * (1.0f / 32.0f) + 1e-5f);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 ga = *reinterpret_cast<const f32x4 *>(a.gamma + cbase + 8 * g);
                    const f32x4 be = *reinterpret_cast<const f32x4 *>(a.beta + cbase + 8 * g);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = silu_f(v[4 * g + e] * rstd * ga[e] + be[e]);
                    if constexpr (EPI == EPI_GN_SILU_RES) {
                        const f32x4 h = *reinterpret_cast<const f32x4 *>(orow + 8 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += h[e];
                    }
                    *reinterpret_cast<f32x4 *>(orow + 8 * g) = o;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
                    if constexpr (EPI == EPI_SDE) {
                        const f32x4 x = *reinterpret_cast<const f32x4 *>(orow + 8 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = a.sde_a * x[e] + a.sde_c * v[4 * g + e];
                    } else if constexpr (EPI == EPI_BIAS_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = silu_f(v[4 * g + e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = v[4 * g + e];
                    }
                    *reinterpret_cast<f32x4 *>(orow + 8 * g) = o;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
static hipError_t launch_cfg(const LayerArgs &a, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
    if (a.Mp % BM || a.N % BN || a.K % BK) return hipErrorInvalidValue;
    auto kern = layer_kernel<BM, BN, WM, WN, EPI>;
    static bool attr_done = false;  // per instantiation; benign race (idempotent call)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int nwg = (a.Mp / BM) * (a.N / BN);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_layer(const LayerArgs &a, int epilogue, hipStream_t st) {
    if (a.N == XLD) {  // post_dense: 51 (padded to 64) output channels, one column tile
        switch (epilogue) {
            case EPI_SDE: return launch_cfg<128, 64, 4, 1, EPI_SDE>(a, st);
            case EPI_BIAS: return launch_cfg<128, 64, 4, 1, EPI_BIAS>(a, st);
        }
        return hipErrorInvalidValue;
    }
    switch (epilogue) {
        case EPI_GN_SILU: return launch_cfg<128, 128, 2, 2, EPI_GN_SILU>(a, st);
        case EPI_GN_SILU_RES: return launch_cfg<128, 128, 2, 2, EPI_GN_SILU_RES>(a, st);
        case EPI_BIAS: return launch_cfg<128, 128, 2, 2, EPI_BIAS>(a, st);
        case EPI_BIAS_SILU: return launch_cfg<128, 128, 2, 2, EPI_BIAS_SILU>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace zedo
