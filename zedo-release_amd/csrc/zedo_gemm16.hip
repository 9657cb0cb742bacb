// Hidden layers of the ZeDO score network (reference lib/algorithms/advanced/model.py:271-288) on the fp16 matrix pipe of
// gfx950 at fp32-level accuracy ("f16x3", the opt-in math mode ZEDO_MATH_F16X3; the default stays exact fp32 MFMA,
// zedo_gemm.hip).
//
//   out[m][n] = epilogue( unscale * sum_k X[m][k] * Ws[n][k] + bias[n] ),     Ws = W * 2^wshift
//
// Both operands travel as TWO fp16 pieces per element, a = ah + al + O(2^-24 |a|) (zedo_tile.h: the "planes" format,
// 4 bytes per element like fp32), and a 16-deep k block of a 32x32 output tile costs three v_mfma_f32_32x32x16_f16:
//     al.bh + ah.bl + ah.bh                     (al.bl <= 2^-24 relative is dropped)
// with fp32 accumulation inside the MFMA.  Per product the error is that of ONE fp32 rounding - what a single step of the
// exact-fp32 fma chain commits - and the sum over k takes 64 block additions instead of 1024 chained roundings: measured
// max |y - y_fp64| of a whole GroupNorm + SiLU layer 1.4e-6 against 2.3e-6 for the exact-fp32 kernel on the same data
// (tools/ubench/ubench_f16x3.hip).  fp16 has 5 exponent bits: W carries a per-layer power-of-two scale (max |w| -> [2^13,
// 2^14)) undone exactly in the epilogue's first fma; activations are O(1..10) and unscaled; the matrix pipe honours fp16
// denormals (probed), so tiny pieces cost an absolute 3e-8, not a flush.
//
// What bounds it (MI355X; tools/ubench/ubench_f16x3.hip, ubench_vmem_issue.hip, ubench_dma_layout.hip, ubench_epi_pattern.hip;
// profiles/f16x3_designs_r05.txt): three fp16 MFMAs are 96 matrix-pipe cycles against 512 for the eight exact-fp32 MFMAs of the same
// block, but (i) under a dense fp16 MFMA stream power management holds the shader clock at 1.5-1.85 GHz (2.36 for the fp32 kernel;
// the busier the pipe the lower): the MFMA floor of a 50k-row layer is 165-198 us, not 127; (ii) the operands still cross L2 -> LDS
// at 4 bytes per element: the DMA stream of the 128 x 256 tiles alone takes 91 us, the stores alone 31 us, the epilogue's VALU
// ~50 us per SIMD, and they overlap only partly.  A vector-memory instruction stalls the wave that issues it (an LDS-DMA ~70
// cycles, 100-130 of MFMA issue when the issuer is the MFMA wave itself), not the other waves of its SIMD (round 5; round 3 had
// read its co-issue table as "VMEM serialises with MFMA").  Plain VALU beside an fp16 MFMA is free; the epilogue is scalar fp32
// (no v_pk_*: those cost ~9 cycles each here, the file is built with -fno-slp-vectorize).  Address patterns matter as much as
// counts: planes are k-block-major so that every DMA instruction reads 1 KB of contiguous memory, and the epilogue writes every
// 64-byte row piece with four adjacent lanes (see the write-out below).
//
// Mapping, LDS layout and loop are the exact-fp32 kernel's (zedo_gemm.hip) with 64-byte LDS rows:
//   i = output CHANNEL (rows of W, MFMA A operand), j = BATCH ROW (rows of X, B operand); lane (li, kh) holds k = 8 kh .. 8 kh + 7
//   of row li: one ds_read_b128 per plane; chunk c = 2 plane + kh of LDS row r sits at position c ^ ((r >> 2) & 3), which
//   makes every 16-lane group of a ds_read_b128 hit 16 distinct slots (SQ_LDS_BANK_CONFLICT = 0, measured).
#include "zedo_internal.h"
#include "zedo_tile.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#ifdef ZEDO_UBENCH      // the harness (tools/ubench/ubench_gemm16.hip) compiles this file with per-workgroup timeline marks
#include "../../tools/ubench/zedo_tile_hooks.inc"
#else
#define TL_MARK(var)
#define TL_FLUSH(t0, t1, t2)
#endif

namespace zedo {

// fp32 [rows][cols] (row stride ld) * scale -> planes [cols/16][ldr][2][16] (k-block-major, zedo_tile.h)
__global__ void split_planes_kernel(const float *__restrict__ src, int rows, int cols, int ld, float scale,
                                    uint16_t *__restrict__ dst, int ldr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread = 8 consecutive columns
    const int per_row = cols / 8;
    if (i >= (size_t)rows * per_row) return;
    const int r = (int)(i / per_row), c = (int)(i % per_row) * 8;
    const float *s = src + (size_t)r * ld + c;
    f32x4 v0 = *reinterpret_cast<const f32x4 *>(s), v1 = *reinterpret_cast<const f32x4 *>(s + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v0[e] *= scale; v1[e] *= scale; }
    f16x8 h, l;
    split_f16x8(v0, v1, h, l);
    char *d = reinterpret_cast<char *>(dst) + ((size_t)(c >> 4) * ldr + r) * 64 + ((c >> 3) & 1) * 16;
    *reinterpret_cast<f16x8 *>(d) = h;
    *reinterpret_cast<f16x8 *>(d + 32) = l;
}

hipError_t launch_split_planes(const float *src, int rows, int cols, int ld, float scale, uint16_t *dst, int ldr, hipStream_t st) {
    if (cols % 16 || ld % 4 || ldr < rows) return hipErrorInvalidValue;
    const size_t n = (size_t)rows * (cols / 8);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, rows, cols, ld, scale, dst, ldr);
    return hipGetLastError();
}


// One BM x BN output tile.  WM x WN waves, each TM x TN = (BM/WM) x (BN/WN); ring of NBUF 16-k blocks (2 for the big tile,
// whose co-resident workgroups cover each other's DMA latency; 4 for the small tiles, which run at the end of a launch
// or in small batches with the CU to themselves: a block's MFMAs (6 x 32 cycles) are far shorter than the DMA latency).
// XF32 (pre_dense: K = 64 = NBUF blocks, everything resident): X arrives as fp32 rows [Mp][64] (the padded pose state) and is
// split into the LDS slots by the workgroup itself - 32 values per thread, once - instead of by LDS-DMA.
// EPI_SDE / EPI_BIAS (post_dense: BN = 64 = the padded pose row): the epilogue of zedo_gemm.hip's post_dense - bias, the SDE
// update x' = a x + c eps on the fp32 pose state, the next iteration's reprojection correction on the rows in the LDS.
template <int BM, int BN, int WM, int WN, int EPI, int NBUF, int XF32 = 0>
__device__ __forceinline__ void layer16_tile(const Layer16Args &a, const int m0, const int n0) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int TM = BM / WM, TN = BN / WN, TJ = TM / 32, TI = TN / 32;
    constexpr int RB = 64, CPR = 4;                                   // bytes / 16-byte chunks per LDS row per k block
    constexpr int IA = BN * CPR / 64 / NW, IB = BM * CPR / 64 / NW, IPW = IA + IB;    // DMA instructions (1 KB) per wave per block
    static_assert(TM % 32 == 0 && TN % 32 == 0 && IA >= 1 && IB >= 1 && (BN * CPR) % (64 * NW) == 0 && (BM * CPR) % (64 * NW) == 0, "tile");
    constexpr int SLOT = (BN + BM) * RB;                              // one ring slot: [BN rows of W][BM rows of X]
    constexpr int SR = WM * 32;                                       // epilogue stage rows per phase
    static_assert(NBUF >= 2 && (NBUF - 1) * IPW <= 63, "ring depth (6-bit vmcnt)");
    constexpr int RING_B = NBUF * SLOT, STAGE_B = SR * BN * 4, BODY_B = RING_B > STAGE_B ? RING_B : STAGE_B;

    extern __shared__ __attribute__((aligned(16))) char smem16[];
    float *Ps = reinterpret_cast<float *>(smem16 + BODY_B);           // [3][BN] bias | gamma | beta

    TL_MARK(tl0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN, li = lane & 31, kh = lane >> 5;
    // planes are k-block-major: block kb of row r at ((kb * ld + r) * 64 bytes: the rows of a block are contiguous, a 1-KB DMA
    // instruction reads 1 KB of memory (16 rows x 64 bytes; round 5 - with [row][k/16] rows 4 KB apart the same stream ran at 0.6x)
    const size_t wkb = (size_t)a.N * RB, xkb = (size_t)a.ldx * RB;   // bytes between k blocks of W / X
    const char *Wbase = reinterpret_cast<const char *>(a.W) + (size_t)n0 * RB;
    const char *Xbase = reinterpret_cast<const char *>(a.X) + (size_t)m0 * RB;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem16;

    // DMA instruction p of this wave moves the 16-byte chunks g = (wid * I + p) * 64 + lane of the tile's block: LDS row
    // g / 4, position g % 4, which holds source chunk (g % 4) ^ swz(row)
    // A wave's pieces of one operand are consecutive: piece p is rows (wid * I + p) * 16 .. + 15, i.e. source offset AND LDS offset of
    // piece 0 + 1024 p (the swizzle term (r >> 2) & 3 = (lane / 16) & 3 does not depend on p): they go out behind one M0 write (dma16n).
    static_assert((IA == 1 || IA == 2 || IA == 4) && (IB == 1 || IB == 2 || IB == 4), "pieces per wave per operand");
    unsigned woff0, xoff0;
    { const int g = wid * IA * 64 + lane, r = g / CPR; woff0 = (unsigned)(r * RB + (((g % CPR) ^ ((r >> 2) & 3)) * 16)); }
    { const int g = wid * IB * 64 + lane, r = g / CPR; xoff0 = (unsigned)(r * RB + (((g % CPR) ^ ((r >> 2) & 3)) * 16)); }
    auto dma = [&](int kb, int slot) {
        const char *wk = Wbase + (size_t)kb * wkb, *xk = Xbase + (size_t)kb * xkb;
        dma16n<IA>(wk, woff0, lds0 + slot * SLOT + wid * IA * 1024);
        if constexpr (!XF32) dma16n<IB>(xk, xoff0, lds0 + slot * SLOT + BN * RB + wid * IB * 1024);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragments {A.h, A.l, B.h, B.l} of one block, two register sets (the next block's reads are issued in front of this
    // block's MFMAs and pinned there: left alone the compiler sinks them to their first use)
    f16x8 fa[2][TI][2], fb[2][TJ][2];
    const int fs = (li >> 2) & 3;                  // tile bases are multiples of 32 rows: swz(row) == swz(li)
    auto fread = [&](int set, int slot) {
        const char *As = smem16 + slot * SLOT + (wn * TN + li) * RB;
        const char *Bs = smem16 + slot * SLOT + BN * RB + (wm * TM + li) * RB;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa[set][i][pl] = *reinterpret_cast<const f16x8 *>(As + i * 32 * RB + (((pl * 2 + kh) ^ fs) * 16));
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[set][j][pl] = *reinterpret_cast<const f16x8 *>(Bs + j * 32 * RB + (((pl * 2 + kh) ^ fs) * 16));
    };
    // product-major: consecutive MFMAs go to different accumulators; the small terms first.  The ORDER (lh, hl, hh per
    // block, blocks ascending) is the same for every tile shape: two launches of the same rows agree bit for bit.
    auto mma = [&](int set) {
#ifndef ZEDO_MUT_F16_DROP_LH     // tools/mutation_check.py only: the low pieces of W never meet the high pieces of X (W as plain fp16)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][1], fb[set][j][0], acc[i][j], 0, 0, 0);
#endif
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][0], fb[set][j][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][0], fb[set][j][0], acc[i][j], 0, 0, 0);
    };

    constexpr bool GN = (EPI == EPI_GN_SILU || EPI == EPI_GN_SILU_RES);
    if (tid < BN / 4) {       // epilogue parameters -> LDS once
        *reinterpret_cast<f32x4 *>(Ps + tid * 4) = *reinterpret_cast<const f32x4 *>(a.bias + n0 + tid * 4);
        if constexpr (GN) {
            *reinterpret_cast<f32x4 *>(Ps + BN + tid * 4) = *reinterpret_cast<const f32x4 *>(a.gamma + n0 + tid * 4);
            *reinterpret_cast<f32x4 *>(Ps + 2 * BN + tid * 4) = *reinterpret_cast<const f32x4 *>(a.beta + n0 + tid * 4);
        }
    }
    const int KB = a.K / 16;                       // multiple of NBUF (checked at launch): the ring slot is a compile-time constant
    // hidden layers: waves inside the k loop outrank the co-resident workgroup's epilogue waves in the SIMD's arbitration
    // (measured 337 -> 334 us per layer; the other way round - epilogue above loop - 340 -> 345; on the thin layers nothing)
    constexpr bool LOOP_PRIO = (EPI == EPI_GN_SILU || EPI == EPI_GN_SILU_RES) && !XF32;
    if constexpr (LOOP_PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int t = 0; t < NBUF; ++t) dma(min(t, KB - 1), t);
    if constexpr (XF32) {
        // the fp32 pose rows of this tile -> fp16 pieces in the X part of the NBUF (= K / 16) slots: 8 consecutive k per item
        // = one 16-byte chunk of the h plane and one of the l plane
        static_assert(NBUF == 4, "XF32: K = 64 resident");
        for (int q = tid; q < BM * 8; q += NT) {
            const int r = q >> 3, c8 = q & 7;
            const float *src = a.Xf32 + (size_t)(m0 + r) * XLD + c8 * 8;
            f16x8 h, l;
            split_f16x8(*reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4), h, l);
            char *dst = smem16 + (c8 >> 1) * SLOT + BN * RB + r * RB;
            const int sw = (r >> 2) & 3, kh8 = c8 & 1;
            *reinterpret_cast<f16x8 *>(dst + ((kh8 ^ sw) * 16)) = h;
            *reinterpret_cast<f16x8 *>(dst + (((2 + kh8) ^ sw) * 16)) = l;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * IPW) : "memory");      // block 0 landed
    }
    __syncthreads();
    fread(0, 0);
#ifdef ZEDO_MUT_F16_XLOW0         // tools/mutation_check.py only: the low pieces of the activations are lost in the first of the 64 k blocks
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) fb[0][j][1][e] = (_Float16)0.0f;
#endif
    TL_MARK(tl1)
    //   block kb in slot kb % NBUF, its fragments in set kb & 1:
    //       vmcnt((NBUF-2) blocks); barrier   <- every wave has read block kb (its fragments are in registers), block kb+1 has landed
    //       DMA(block kb+NBUF -> slot of kb);  read(block kb+1) -> the other set;  MFMA(block kb)
    // slots and fragment sets are both back at 0 after U blocks: unrolled by U every LDS offset is an instruction immediate
    constexpr int U = NBUF % 2 == 0 ? NBUF : 2 * NBUF;
    auto step = [&](int kb, int slot, int set) __attribute__((always_inline)) {
        // hipcc does not count the LDS-DMA in its vmcnt bookkeeping: wait explicitly
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * IPW) : "memory");
        __syncthreads();
        if constexpr (!XF32) dma(min(kb + NBUF, KB - 1), slot);     // branch-free: past the end it refills a slot nobody reads again
        fread(set ^ 1, (slot + 1) % NBUF);
        __builtin_amdgcn_sched_barrier(0);
        mma(set);
        __builtin_amdgcn_sched_barrier(0);
    };
    int kb0 = 0;
    for (; kb0 + U <= KB; kb0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step(kb0 + u, u % NBUF, u & 1);
    }
    if constexpr (NBUF % 2 != 0) {      // odd ring (3): K / 16 is a multiple of 4, not of 6 - the last 2 or 4 blocks (uniform branches)
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (kb0 + u < KB) step(kb0 + u, u % NBUF, u & 1);
    }
    // even rings have no tail: K / 16 is a multiple of the ring depth - launch_layer16 refuses any other K for every instantiation
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();           // drain the trailing DMA before the ring becomes the epilogue stage
    TL_MARK(tl2)
    if constexpr (LOOP_PRIO) __builtin_amdgcn_s_setprio(0);

    // ---- epilogue: GroupNorm + SiLU on the accumulators, staged through the LDS row-wise (chunk c of stage row sr at
    //      position c ^ (sr & 7), as in zedo_gemm.hip), then per thread 16 consecutive channels of a row: [residual from
    //      its planes +] either 64 bytes of fp32 or 32 + 32 bytes of fp16 pieces - the same 64 bytes of the row either way
    if constexpr (GN) {
        constexpr int CG = BN / 16;              // 16-channel groups per stage row
        float *S = reinterpret_cast<float *>(smem16);
        // planes out (and residual): 16-channel group cg of row r at ((n0 / 16 + cg) * ldo + m0 + r) * 64 bytes; fp32 out: row-major [Mp][N]
        const bool pl_out = !a.out_f32;
        const size_t orow = pl_out ? 64 : (size_t)a.N * 4, ogrp = pl_out ? (size_t)a.ldo * 64 : 64;
        char *obase = reinterpret_cast<char *>(a.out) + (pl_out ? ((size_t)(n0 >> 4) * a.ldo + m0) * 64 : (size_t)m0 * a.N * 4 + (size_t)n0 * 4);
        const char *rbase = reinterpret_cast<const char *>(a.res) + ((size_t)(n0 >> 4) * a.ldo + m0) * 64;
        const size_t rgrp = (size_t)a.ldo * 64;
        // (row, 16-channel group) items: in round `it` wave `wid` owns stage rows it * NT / CG + wid * 64 / CG .. + 64 / CG - 1 and all CG
        // groups; within the wave lane = (row & 3) | (group & 3) << 2 | the rest: the 16 lanes of one LDS pass then hit 16 different
        // 16-byte slots of the swizzled stage rows (slot = (4 group + q) ^ (row & 7); with lane = group | row << 4, the map until
        // round 5, they hit four - SQ_LDS_BANK_CONFLICT 5.6 M cycles per launch once the write-back below doubled those accesses)
        constexpr int ITEMS = (SR * CG + NT - 1) / NT;             // items per thread
        static_assert((SR * CG) % NT == 0 && (CG == 4 || CG == 8 || CG == 16), "write-out shape");
        constexpr int RPW = 64 / CG;                               // rows per wave per round
        const int item_cg = ((lane >> 2) & 3) + 4 * ((lane >> 4) % (CG / 4));
        const int item_r = wid * RPW + (lane & 3) + 4 * ((lane >> 4) / (CG / 4));
        auto item_row = [&](int it) { return it * (NT / CG) + item_r; };
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            // residual (EPI_GN_SILU_RES): ALL of this thread's loads of the phase, issued BEFORE the phase's GroupNorm / SiLU arithmetic
            // (round 5: they used to follow it, one exposed memory round trip per phase - the residual layers were 40 us per launch
            // slower than the plain ones); they are consumed after the stage barrier.  `res` may alias `out`: the rows of phase j are
            // not written before phase j's own stores, which follow these loads in program order.
            f16x8 rres[EPI == EPI_GN_SILU_RES ? ITEMS : 1][4];
            if constexpr (EPI == EPI_GN_SILU_RES) {
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const int sr = item_row(it), cg = item_cg;
                    const int grow = (sr >> 5) * TM + j * 32 + (sr & 31);
                    const size_t off = (size_t)grow * 64 + (size_t)cg * rgrp;
#pragma unroll
                    for (int q = 0; q < 4; ++q) rres[it][q] = *reinterpret_cast<const f16x8 *>(rbase + off + 16 * q);
                }
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                float o[16];
                f32x4 b4[4], ga[4], be[4];
                const float *pc = Ps + wn * TN + i * 32 + 4 * kh;   // channel of accumulator r: + (r&3) + 8*(r>>2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    b4[g] = *reinterpret_cast<const f32x4 *>(pc + 8 * g);
                    ga[g] = *reinterpret_cast<const f32x4 *>(pc + BN + 8 * g);
                    be[g] = *reinterpret_cast<const f32x4 *>(pc + 2 * BN + 8 * g);
                }
                // hidden layers: scalar VALU (free beside the other workgroups' fp16 MFMAs); pre_dense has 12 MFMAs per tile and is
                // bound by this very arithmetic: packed fp32 halves its instruction count
                epilogue_values<EPI_GN_SILU, true, XF32 != 0>(acc[i][j], b4, ga, be, 0.f, o, a.unscale);
                float *srow = S + (wm * 32 + li) * BN;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = (wn * TN + i * 32 + 8 * g + 4 * kh) >> 2;
                    const f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(srow + ((c ^ (li & 7)) << 2)) = v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int sr = item_row(it), cg = item_cg;
                const int grow = (sr >> 5) * TM + j * 32 + (sr & 31);
                const size_t off = (size_t)grow * orow + (size_t)cg * ogrp;
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4 *>(S + sr * BN + (((cg * 4 + q) ^ (sr & 7)) << 2));
                if constexpr (EPI == EPI_GN_SILU_RES) {            // h = h + h2 (model.py:288): h from its planes, exactly h + l
                    const f16x8 &rh0 = rres[it][0], &rh1 = rres[it][1], &rl0 = rres[it][2], &rl1 = rres[it][3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[0][e] += join_f16(rh0[e], rl0[e]);     v[1][e] += join_f16(rh0[4 + e], rl0[4 + e]);
                        v[2][e] += join_f16(rh1[e], rl1[e]);     v[3][e] += join_f16(rh1[4 + e], rl1[4 + e]);
                    }
                }
                if (a.out_f32) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(obase + off + 16 * q) = v[q];
                } else {
                    // the item's 64 bytes of planes {h 0-7, h 8-15, l 0-7, l 8-15} go back into its own four stage slots ...
                    f16x8 h0, l0, h1, l1;
                    split_f16x8(v[0], v[1], h0, l0);
                    split_f16x8(v[2], v[3], h1, l1);
                    float *sp = S + sr * BN;
                    *reinterpret_cast<f16x8 *>(sp + (((cg * 4 + 0) ^ (sr & 7)) << 2)) = h0;
                    *reinterpret_cast<f16x8 *>(sp + (((cg * 4 + 1) ^ (sr & 7)) << 2)) = h1;
                    *reinterpret_cast<f16x8 *>(sp + (((cg * 4 + 2) ^ (sr & 7)) << 2)) = l0;
                    *reinterpret_cast<f16x8 *>(sp + (((cg * 4 + 3) ^ (sr & 7)) << 2)) = l1;
                }
            }
            if (!a.out_f32) {
                // ... and leave with FOUR LANES PER 64-BYTE ROW PIECE: instruction n of a wave writes 4 rows x 4 channel groups, each group's
                // rows 256 contiguous bytes.  (Round 5, tools/ubench/ubench_epi_pattern.hip: with the planes k-block-major the old map -
                // one lane = one row piece, instruction q = its bytes 16 q .. 16 q + 15 - wrote 64 separate 16-byte pieces per instruction
                // and the stores of one layer alone took 66 us; this map 31 us.)  Wave-local: item (it, lane) of the loop above covers
                // rows it * NT / CG + wid * 64 / CG .. and all CG groups, the same rows this wave moves out here; LDS operations of one
                // wave execute in order, so no barrier separates the write-back from these reads.
                const int q = lane & 3, c4 = (lane >> 2) & 3, r4 = lane >> 4;
#pragma unroll
                for (int n = 0; n < ITEMS * 4; ++n) {
                    const int it = n >> 2, blk = n & 3;
                    const int cg = c4 + 4 * (blk % (CG / 4)), sr = it * (NT / CG) + wid * RPW + r4 + 4 * (blk / (CG / 4));
                    const int grow = (sr >> 5) * TM + j * 32 + (sr & 31);
                    const f16x8 piece = *reinterpret_cast<const f16x8 *>(S + sr * BN + (((cg * 4 + q) ^ (sr & 7)) << 2));
                    *reinterpret_cast<f16x8 *>(obase + (size_t)cg * ogrp + (size_t)grow * 64 + 16 * q) = piece;
                }
            }
            if (j + 1 < TJ) __syncthreads();
        }
    }
    else {
        // ---- post_dense: out = acc * unscale + bias [, x' = a x + c out, next reprojection] on the fp32 pose rows [Mp][64]
        static_assert(BN == XLD, "post_dense tile: one column tile of 64 (= padded pose row)");
        constexpr int CPRW = BN / 4;             // 16-byte chunks per stage row
        float *S = reinterpret_cast<float *>(smem16);
        float *xbase = a.xio + (size_t)m0 * XLD;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const float *pc = Ps + wn * TN + i * 32 + 4 * kh;
                float *srow = S + (wm * 32 + li) * BN;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(pc + 8 * g);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = __builtin_fmaf(acc[i][j][4 * g + e], a.unscale, b[e]);
                        if constexpr (EPI == EPI_SDE) v[e] *= a.sde_c;
                    }
                    const int c = (wn * TN + i * 32 + 8 * g + 4 * kh) >> 2;
                    *reinterpret_cast<f32x4 *>(srow + ((c ^ (li & 7)) << 2)) = v;
                }
            }
            __syncthreads();
            if constexpr (EPI == EPI_SDE) {
                // x' = a x + [c eps]: one fma per element, x read row-wise (coalesced) from the pose state
                for (int qi = tid; qi < SR * CPRW; qi += NT) {
                    const int sr = qi / CPRW, c = qi % CPRW;
                    const int grow = (sr >> 5) * TM + j * 32 + (sr & 31);
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(xbase + (size_t)grow * XLD + c * 4);
                    f32x4 *slot = reinterpret_cast<f32x4 *>(S + sr * BN + ((c ^ (sr & 7)) << 2));
                    f32x4 v = *slot;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(a.sde_a, x[e], v[e]);
                    *slot = v;
                }
                __syncthreads();
                // reprojection correction of the next iteration on the updated rows, one lane per row, straight from the
                // stage; same source (reproj_row) as the exact-fp32 path and the stand-alone kernel
                if (a.rp_geom != nullptr) {
                    constexpr int NV = (17 * 3 + 3) / 4;
                    if (tid < SR) {
                        const int sr = tid;
                        const int b = m0 + (sr >> 5) * TM + j * 32 + (sr & 31);
                        if (b < a.rp_B) {
                            float xr[NV * 4], gr[17 * 3], Tr[3];
                            float *srow = S + sr * BN;
#pragma unroll
                            for (int v = 0; v < NV; ++v) {
                                const f32x4 t = *reinterpret_cast<const f32x4 *>(srow + ((v ^ (sr & 7)) << 2));
                                xr[4 * v] = t[0]; xr[4 * v + 1] = t[1]; xr[4 * v + 2] = t[2]; xr[4 * v + 3] = t[3];
                            }
                            Tr[0] = a.rp_T[(size_t)b * 3]; Tr[1] = a.rp_T[(size_t)b * 3 + 1]; Tr[2] = a.rp_T[(size_t)b * 3 + 2];
                            const int n = (int)((a.rp_row0 + b) % a.rp_N);
                            reproj_row<17>(xr, a.rp_geom + (size_t)n * 17 * GEOM_F, Tr, a.rp_solve != 0, gr);
                            if (a.rp_solve) { a.rp_T[(size_t)b * 3] = Tr[0]; a.rp_T[(size_t)b * 3 + 1] = Tr[1]; a.rp_T[(size_t)b * 3 + 2] = Tr[2]; }
#pragma unroll
                            for (int c = 0; c < 17 * 3; ++c) xr[c] += gr[c];
#pragma unroll
                            for (int v = 0; v < NV; ++v) {
                                const f32x4 t = {xr[4 * v], xr[4 * v + 1], xr[4 * v + 2], xr[4 * v + 3]};
                                *reinterpret_cast<f32x4 *>(srow + ((v ^ (sr & 7)) << 2)) = t;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            float *obase = (EPI == EPI_SDE ? a.xio : reinterpret_cast<float *>(a.out)) + (size_t)m0 * XLD;
            for (int qi = tid; qi < SR * CPRW; qi += NT) {
                const int sr = qi / CPRW, c = qi % CPRW;
                const int grow = (sr >> 5) * TM + j * 32 + (sr & 31);
                *reinterpret_cast<f32x4 *>(obase + (size_t)grow * XLD + c * 4) = *reinterpret_cast<const f32x4 *>(S + sr * BN + ((c ^ (sr & 7)) << 2));
            }
            if (j + 1 < TJ) __syncthreads();
        }
    }
    TL_FLUSH(tl0, tl1, tl2)
}

// block -> tile, XCD aware (the hardware places block b on XCD b % 8): every XCD gets a contiguous range of tiles so that
// the column tiles of one row tile share one L2 (same map as zedo_gemm.hip)
template <int BM, int BN, int WM, int WN, int EPI, int NBUF, int XF32 = 0>
__device__ __forceinline__ void layer16_body(const Layer16Args &a, const int bid, const int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int ncol = a.N / BN;
    layer16_tile<BM, BN, WM, WN, EPI, NBUF, XF32>(a, (lid / ncol) * BM, (lid % ncol) * BN);
}

// One launch, two tile shapes (as layer_pair_kernel of zedo_gemm.hip): workgroups [0, nbig) run BIG_M x BIG_N tiles on the rows
// that fill whole rounds of the chip, the rest 64x64 tiles on the remainder rows, which back-fill CUs as the big tiles
// drain (64x64: the tail of the launch is one small tile's latency, and a 64x64 tile's is the shortest).  Either count may be zero.
// The big tile is 128 rows x 256 channels (four waves of 64 x 128, 230 registers, two workgroups per CU, 67 KB of LDS): every wave
// issues its own LDS-DMA (6 per k block) between its MFMAs, which costs it 100-130 cycles of MFMA issue each while the
// co-resident workgroup's wave on the same SIMD keeps the pipe busy; the tile shape sets how many there are per MFMA:
// LDS-DMA / MFMA instructions = 43 (BM + BN) / (BM BN) x ... = 0.68 for 128x128, 0.51 for 128x256 (measured 341 -> 313 us per
// layer, profiles/ubench_f16x3_tiles_r03.txt), 0.34 for 256x256 (291 us in the microbenchmark; eight waves, one workgroup per
// CU - tried in the product with the remainder as a launch of its own: 0.354 ms per layer against 0.335, the epilogue of the
// only resident workgroup is fully exposed: not adopted; so were round 5's loader-wave and tile ping-pong designs).
// BIG_N = 128: batches between 2 048 and 8 192 rows take 128x128 tiles (three workgroups per CU) - see launch_layer16.
constexpr int BIG_M = 128;
constexpr int BIG_NBUF16 = 3;     // ring depth of the 128 x 256 tile: 3 x 24 KB = 72 KB (the 64 KB epilogue stage fits inside), 2 workgroups per CU = 150 KB
constexpr int MID_NBUF16 = 2;     // ring depth of the 128 x 128 tile (batches of 2 048 - 8 192 rows, three workgroups per CU)
template <int EPI, int BIG_N>
__global__ __launch_bounds__(256, BIG_N == 256 ? 2 : 3) void layer16_pair_kernel(Layer16Args big, Layer16Args small, int nbig) {
    long long c0 = 0, w0 = 0;
    const bool probe = big.clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    if (probe) { c0 = clock64(); w0 = wall_clock64(); }
    if ((int)blockIdx.x < nbig) layer16_body<BIG_M, BIG_N, 2, 2, EPI, BIG_N == 256 ? BIG_NBUF16 : MID_NBUF16>(big, blockIdx.x, nbig);
    else layer16_body<64, 64, 2, 2, EPI, 4>(small, (int)blockIdx.x - nbig, (int)gridDim.x - nbig);
    if (probe) { big.clk[0] = clock64() - c0; big.clk[1] = wall_clock64() - w0; }
}

// The thin layers: pre_dense (51 -> 1024: 64x128 tiles, the four k blocks resident, X split from the fp32 pose state) and
// post_dense (1024 -> 51: 64x64 tiles on an 8-deep ring - a block is 3 MFMAs per wave, 96 cycles: only a deep ring keeps the
// DMA round trip off the critical path -, SDE / bias epilogue on the fp32 pose state).
template <int EPI>
__global__ __launch_bounds__(256, 3) void layer16_pre_kernel(Layer16Args a) {
    layer16_body<64, 128, 2, 2, EPI, 4, 1>(a, blockIdx.x, gridDim.x);
}
template <int EPI>
__global__ __launch_bounds__(256, 2) void layer16_post_kernel(Layer16Args a) {   // 2: the fused reprojection keeps ~180 values per lane live (184 registers in zedo_gemm.hip too); no scratch
    layer16_body<64, 64, 2, 2, EPI, 8, 0>(a, blockIdx.x, gridDim.x);
}

// Small batches (up to 2048 rows: BASELINE configs[0] / [1]): a hidden layer is then a chain of 64 dependent k blocks per
// workgroup and the chip is mostly empty; 64x64 tiles double the number of workgroups and halve each one's MFMA + DMA
// issue time per block.  Same product order per element as every other shape.
template <int EPI>
__global__ __launch_bounds__(256, 4) void layer16_small_kernel(Layer16Args a) {
    layer16_body<64, 64, 2, 2, EPI, 4, 0>(a, blockIdx.x, gridDim.x);
}

#ifdef ZEDO_UBENCH      // round 5's tile ping-pong experiment (not part of the library): see the file
#include "../../tools/ubench/zedo_gemm16_tp.inc"
#endif

constexpr int MAX_DEVICES16 = MAX_DEVICES;      // per-device launch state: allow_lds / num_cus of zedo_internal.h

template <class K>
static hipError_t launch_thin16(K kern, std::atomic<bool> *attr_done, size_t lds, int grid, const Layer16Args &a, hipStream_t st) {
    if (hipError_t e = allow_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int EPI, int BIG_N>
static hipError_t launch_pair16(const Layer16Args &big, const Layer16Args &small, hipStream_t st) {
    constexpr size_t ring_big = (BIG_N == 256 ? BIG_NBUF16 : MID_NBUF16) * (BIG_M + BIG_N) * 64, stage_big = (size_t)64 * BIG_N * 4, par_big = 3 * BIG_N * sizeof(float);
    constexpr size_t ring_small = 4 * (64 + 64) * 64, stage_small = (size_t)64 * 64 * 4, par_small = 3 * 64 * sizeof(float);
    constexpr size_t lds_big = (ring_big > stage_big ? ring_big : stage_big) + par_big;
    constexpr size_t lds_small = (ring_small > stage_small ? ring_small : stage_small) + par_small;
    constexpr size_t lds = lds_big > lds_small ? lds_big : lds_small;      // each tile shape finds its parameter block behind ITS body
    auto kern = layer16_pair_kernel<EPI, BIG_N>;
    static std::atomic<bool> attr_done[MAX_DEVICES16];
    if (hipError_t e = allow_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const int nbig = (big.Mp / BIG_M) * (big.N / BIG_N), nsmall = (small.Mp / 64) * (small.N / 64);
    if (nbig + nsmall == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(nbig + nsmall), dim3(256), lds, st, big, small, nbig);
    return hipGetLastError();
}

static Layer16Args rows_of16(const Layer16Args &a, int row0, int rows) {
    Layer16Args b = a;
    b.X = a.X + (size_t)row0 * 32;                 // uint16 units: a row of a k block is 64 bytes; ldx / ldo stay
    b.out = reinterpret_cast<char *>(a.out) + (a.out_f32 ? (size_t)row0 * a.N * 4 : (size_t)row0 * 64);
    if (a.res) b.res = a.res + (size_t)row0 * 32;
    b.Mp = rows;
    return b;
}

hipError_t launch_layer16(const Layer16Args &a, int epilogue, hipStream_t st) {
    // K / 16 must be a multiple of the ring depth of EVERY tile the launch may pick (the prologue fills NBUF slots with blocks
    // 0 .. NBUF - 1 and the ring slot of block kb is kb % NBUF at compile time): 4-deep rings (64x64 tiles, pre_dense) K % 64, the
    // 8-deep ring of post_dense K % 128; the 2- and 3-deep rings of the big tiles are covered by K % 64 (3: guarded tail in the loop)
    if (a.Mp <= 0 || a.Mp % 64 || a.K % 64 || !a.W) return hipErrorInvalidValue;
    if (a.N == XLD && a.K % (8 * 16)) return hipErrorInvalidValue;
    if ((a.X && a.ldx < a.Mp) || (a.out && !a.out_f32 && a.N != XLD && a.ldo < a.Mp)) return hipErrorInvalidValue;   // planes operands carry their row count
    if (a.Xf32) {               // pre_dense
        if (a.K != XLD || a.N % 128 || epilogue != EPI_GN_SILU || !a.out) return hipErrorInvalidValue;
        static std::atomic<bool> done[MAX_DEVICES16];
        constexpr size_t lds = (size_t)4 * (128 + 64) * 64 + 3 * 128 * sizeof(float);
        return launch_thin16(layer16_pre_kernel<EPI_GN_SILU>, done, lds, (a.Mp / 64) * (a.N / 128), a, st);
    }
    if (a.N == XLD) {           // post_dense
        if (!a.X) return hipErrorInvalidValue;
        constexpr size_t lds = (size_t)8 * (64 + 64) * 64 + 3 * 64 * sizeof(float);
        if (epilogue == EPI_SDE && a.xio) {
            static std::atomic<bool> done[MAX_DEVICES16];
            return launch_thin16(layer16_post_kernel<EPI_SDE>, done, lds, a.Mp / 64, a, st);
        }
        if (epilogue == EPI_BIAS && a.out) {
            static std::atomic<bool> done[MAX_DEVICES16];
            return launch_thin16(layer16_post_kernel<EPI_BIAS>, done, lds, a.Mp / 64, a, st);
        }
        return hipErrorInvalidValue;
    }
    if (a.N % 256 || !a.X || !a.out) return hipErrorInvalidValue;
    if (epilogue == EPI_GN_SILU_RES && !a.res) return hipErrorInvalidValue;
    if (a.Mp <= 2048) {
        constexpr size_t lds = (size_t)4 * (64 + 64) * 64 + 3 * 64 * sizeof(float);
        const int grid = (a.Mp / 64) * (a.N / 64);
        if (epilogue == EPI_GN_SILU) {
            static std::atomic<bool> done[MAX_DEVICES16];
            return launch_thin16(layer16_small_kernel<EPI_GN_SILU>, done, lds, grid, a, st);
        }
        if (epilogue == EPI_GN_SILU_RES) {
            static std::atomic<bool> done[MAX_DEVICES16];
            return launch_thin16(layer16_small_kernel<EPI_GN_SILU_RES>, done, lds, grid, a, st);
        }
        return hipErrorInvalidValue;
    }
    const int cus = num_cus();            // of the CURRENT device (cached per device)
    // big tiles on the rows that fill whole rounds of 2 workgroups per CU; the remainder (and every batch smaller than
    // one round) on 64x64 tiles: finer tiles spread a short launch over more CUs
    // Which rows get which tile (cycles of a CU: a 128x256 tile ~74k alone, a 128x128 tile ~41k, a 64x64 tile ~14.5k):
    //   * whole rounds of 2 x 128x256 tiles per CU: big tiles;
    //   * what is left after the whole rounds (or the whole batch, if it is smaller than a round): big tiles too from 5 120 rows
    //     up (one more, partly filled round of big tiles then beats ~14 cycles per row of 64x64 tiles), 64x64 tiles below;
    //   * a batch below 8 192 rows: 128x128 tiles on every whole 128 rows (more workgroups than 128x256, fewer DMA bytes than
    //     64x64), 64x64 for a last 64-row strip.  Measured per hidden layer: 6 350 rows 55 us (128x128) / 61 (128x256) / 68
    //     (64x64); 12 700 rows 118 / 96 / 125.
    const int per_round = cus * 2 * BIG_M / (a.N / 256);
    const int rows_whole = (a.Mp / per_round) * per_round;
    const int rest = a.Mp - rows_whole;
    if (rows_whole == 0 && rest < 8192) {
        const int rows_mid = (a.Mp / BIG_M) * BIG_M;
        const Layer16Args mid = rows_of16(a, 0, rows_mid), tail = rows_of16(a, rows_mid, a.Mp - rows_mid);
        switch (epilogue) {
            case EPI_GN_SILU: return launch_pair16<EPI_GN_SILU, 128>(mid, tail, st);
            case EPI_GN_SILU_RES: return launch_pair16<EPI_GN_SILU_RES, 128>(mid, tail, st);
        }
        return hipErrorInvalidValue;
    }
    const int rows_big = rest >= 5120 ? rows_whole + (rest / BIG_M) * BIG_M : rows_whole;
    const Layer16Args big = rows_of16(a, 0, rows_big), small = rows_of16(a, rows_big, a.Mp - rows_big);
    switch (epilogue) {
        case EPI_GN_SILU: return launch_pair16<EPI_GN_SILU, 256>(big, small, st);
        case EPI_GN_SILU_RES: return launch_pair16<EPI_GN_SILU_RES, 256>(big, small, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace zedo
