// Device helpers shared by the two dense-layer translation units (zedo_gemm.hip: exact-fp32 MFMA tiles; zedo_gemm16.hip:
// split-fp16 tiles): the LDS-DMA issue helper and the GroupNorm / SiLU / SDE epilogue arithmetic.  gfx950 only.
#pragma once
#include "zedo_internal.h"

namespace zedo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- split-fp16 activation format ("planes") --------------------------------------------------------------------
// An fp32 value a is carried as two fp16 pieces, a = h + l + O(2^-24 |a|): h = fp16(a), l = fp16(a - h), both round to
// nearest even (11 significant bits each, signed: 23 bits of a).  A row of C channels is stored as
//     P[row][C/16][2 planes][16] fp16        (64 bytes per 16 channels: 32 B of h, 32 B of l; 4 bytes per element)
// which is exactly the operand layout the 32x32x16 fp16 MFMA tiles of zedo_gemm16.hip stream through the LDS.
// fp16 tops out at 65504: activations of this network are SiLU(GroupNorm(.)) outputs and their residual sums, bounded
// by (|gamma| sqrt(31) + |beta|) per layer - O(10) - and are stored unscaled; pieces below 6.1e-5 are fp16 denormals,
// which the gfx950 matrix pipe honours (tools/ubench/ubench_f16x3.hip probes it): absolute error <= 3e-8 per element.
__device__ __forceinline__ void split_f16x8(const f32x4 &v0, const f32x4 &v1, f16x8 &h, f16x8 &l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h0 = (_Float16)v0[e], h1 = (_Float16)v1[e];
        h[e] = h0; h[4 + e] = h1;
        l[e] = (_Float16)(v0[e] - (float)h0); l[4 + e] = (_Float16)(v1[e] - (float)h1);
    }
}
// the value the planes stand for: h + l is exact in fp32 (the pieces do not overlap)
__device__ __forceinline__ float join_f16(_Float16 h, _Float16 l) { return (float)h + (float)l; }


// global -> LDS DMA of 64 x 16 bytes: source = wave-uniform 64-bit base + per-lane 32-bit byte offset, destination =
// wave-uniform LDS byte address + lane*16.  Inline asm because hipcc materialises base + zext(offset) with a 64-bit
// VALU add per instruction inside the K loop, and VALU issue time is matrix-pipe time here.  M0 (the LDS
// destination) is compiler-reserved: it is saved and restored inside the statement.  hipcc does not count this load
// in its vmcnt bookkeeping - every consumer below sits behind an explicit s_waitcnt vmcnt + barrier.
__device__ __forceinline__ void dma16(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(sbase), "v"(voff), "s"(lds_byte_addr)
                 : "memory");
}

// N (1, 2 or 4) such DMAs of CONSECUTIVE 1-KB pieces - source and LDS destination both advance by 1 KB per piece - behind ONE M0
// write: the instruction's 13-bit immediate offset is added to the global AND the LDS address (checked on the device by
// tools/ubench/ubench_dma_group.hip, which also measured the burst of a 24-MFMA block: six separate dma16 cost the issuing wave
// 39 / 31 cycles each at 1 / 2 waves per SIMD, 4 + 2 behind two M0 writes 20 / 18).
// (The non-temporal cache policy for the activation tiles was measured in round 5 and is off: +3 % on a stand-alone layer that re-reads the
// same planes, -3 % in the product, where a layer's input was just written by the launch before it; profiles/f16x3_designs_r05.txt.)
template <int N>
__device__ __forceinline__ void dma16n(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    static_assert(N == 1 || N == 2 || N == 4, "pieces per M0 write (offsets up to 3072 fit the 13-bit signed field)");
    unsigned keep;
    if constexpr (N == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\tglobal_load_lds_dwordx4 %2, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %1 offset:2048\n\tglobal_load_lds_dwordx4 %2, %1 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
    else if constexpr (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\tglobal_load_lds_dwordx4 %2, %1 offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
}

__device__ __forceinline__ float silu_fast(float y) {
    // y * sigmoid(y) with hardware exp2 / rcp (each <= 1 ulp): |rel err| <= ~3e-7
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.44269504088896340736f));
}

// Per-element part of the epilogue for one 32(channel) x 32(row) accumulator tile; lane = (batch row li,
// channel half kh), o[4g+e] belongs to channel cbase + 8g + e.  The residual / previous-x term is added by
// the caller from the LDS stage (EPI_GN_SILU_RES: o += h;  EPI_SDE: o += sde_a * x).
// SCALED (split-fp16 tiles, whose W operand carries a power-of-two scale): acc * acc_scale + bias in one fma.
// PACKED = false (split-fp16 tiles): the same arithmetic on single floats.  Beside a co-resident wave's fp16 MFMAs a plain
// VALU instruction issues for free while a packed-fp32 one costs ~9 matrix-pipe cycles (profiles/coissue_f16_r03.txt) - the
// opposite of the exact-fp32 regime, where every VALU issue slot is matrix-pipe time and packing halves the slots.  The
// results are bit-identical either way (same operations, same order; -ffp-contract=off).
template <int EPI, bool SCALED = false, bool PACKED = true>
__device__ __forceinline__ void epilogue_values(const f32x16 &acc, const f32x4 (&b4)[4], const f32x4 (&ga)[4],
                                                const f32x4 (&be)[4], float sde_c, float (&o)[16], float acc_scale = 1.0f) {
    if constexpr ((EPI == EPI_GN_SILU || EPI == EPI_GN_SILU_RES) && !PACKED) {
        float p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float b = b4[k >> 2][k & 3];
            if constexpr (SCALED) p[k] = __builtin_fmaf(acc[k], acc_scale, b);
            else p[k] = acc[k] + b;
        }
        // the same summation tree as the packed form: two lanes of pairs (even / odd elements), then their sum
        float se = p[0], so = p[1];
#pragma unroll
        for (int k = 1; k < 8; ++k) { se += p[2 * k]; so += p[2 * k + 1]; }
        float s = se + so;
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 32.0f);
        float qe = 0.0f, qo = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            p[2 * k] -= mean; p[2 * k + 1] -= mean;
            qe = __builtin_fmaf(p[2 * k], p[2 * k], qe);
            qo = __builtin_fmaf(p[2 * k + 1], p[2 * k + 1], qo);
        }
        float qs = qe + qo;
        qs += __shfl_xor(qs, 32);
#ifdef ZEDO_MUT_GN_EPS
        constexpr float GN_EPS_S = 2e-5f;
#else
        constexpr float GN_EPS_S = 1e-5f;
#endif
        const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(qs, 1.0f / 32.0f, GN_EPS_S));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float y = __builtin_fmaf(p[k], rstd * ga[k >> 2][k & 3], be[k >> 2][k & 3]);
            const float d = __builtin_amdgcn_exp2f(y * -1.44269504088896340736f) + 1.0f;
            o[k] = y * __builtin_amdgcn_rcpf(d);
        }
        return;
    }
    if constexpr (EPI == EPI_GN_SILU || EPI == EPI_GN_SILU_RES) {
        // GroupNorm(32 groups of 32 channels, biased variance, eps 1e-5: model.py:116,145,150) then SiLU:
        //   o = acc + bias;  mean = sum32(o)/32;  o -= mean;  rstd = rsq(sum32(o^2)/32 + 1e-5);
        //   y = o * (rstd * gamma) + beta;  out = y / (1 + exp(-y))
        // written on float pairs so that it maps to v_pk_add / v_pk_mul / v_pk_fma_f32 (two results per VALU issue
        // slot): VALU issue time is matrix-pipe time for the co-resident workgroup, and this epilogue is the largest
        // non-MFMA item of the layer.
        f32x2 p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 a2 = {acc[2 * k], acc[2 * k + 1]};
            const f32x2 b2 = {b4[k >> 1][2 * (k & 1)], b4[k >> 1][2 * (k & 1) + 1]};
            if constexpr (SCALED) p[k] = __builtin_elementwise_fma(a2, (f32x2){acc_scale, acc_scale}, b2);
            else p[k] = a2 + b2;
        }
        f32x2 s2 = p[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) s2 += p[k];
        float s = s2.x + s2.y;
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 32.0f);
        const f32x2 m2 = {mean, mean};
        f32x2 q2 = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            p[k] -= m2;
            q2 = __builtin_elementwise_fma(p[k], p[k], q2);
        }
        float qs = q2.x + q2.y;
        qs += __shfl_xor(qs, 32);
#ifdef ZEDO_MUT_GN_EPS      // tools/mutation_check.py only: a deliberately wrong constant that the parity suite must catch
        constexpr float GN_EPS = 2e-5f;
#else
        constexpr float GN_EPS = 1e-5f;
#endif
        const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(qs, 1.0f / 32.0f, GN_EPS));   // one fma, spelled out (-ffp-contract=off)
        const f32x2 r2 = {rstd, rstd}, c2 = {-1.44269504088896340736f, -1.44269504088896340736f}, one2 = {1.0f, 1.0f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 g2 = {ga[k >> 1][2 * (k & 1)], ga[k >> 1][2 * (k & 1) + 1]};
            const f32x2 e2 = {be[k >> 1][2 * (k & 1)], be[k >> 1][2 * (k & 1) + 1]};
            const f32x2 y = __builtin_elementwise_fma(p[k], r2 * g2, e2);
            const f32x2 t = y * c2;
            const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + one2;
            const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            const f32x2 v = y * r;
            o[2 * k] = v.x;
            o[2 * k + 1] = v.y;
        }
        return;
    }
    if constexpr (EPI == EPI_PARTIAL) {        // one K quarter of post_dense, raw: bias and the update follow in post_reduce_kernel
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = acc[e];
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * g + e] = acc[4 * g + e] + b4[g][e];
    if constexpr (EPI == EPI_SDE) {            // x' = a x + c eps  (sampling.py:185-190 folded)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] *= sde_c;
    } else if constexpr (EPI == EPI_BIAS_SILU) {      // built once per schedule: IEEE exp / divide
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = o[e] / (1.0f + expf(-o[e]));
    }
}


}  // namespace zedo
