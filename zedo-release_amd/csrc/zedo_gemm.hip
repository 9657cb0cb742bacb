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
// so both LDS tiles are [rows][32 floats] and every lane feeds 4 consecutive MFMAs from one
// ds_read_b128 (lane half kh supplies k = 8*kg + 4*kh + e for MFMA e; A and B agree on that
// order, so the sum over k is complete).
//
// What shapes this kernel (measured on MI355X, tools/ubench/ubench_coissue.hip and
// tools/ubench/ubench_gemm.hip, numbers in DESIGN.md / profiles/):
//   * v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 cycles per instruction per SIMD,
//     155.9 TFLOP/s sustained) and, while one is pending on a SIMD, VALU and VMEM instructions
//     of the co-resident waves do not issue (LDS and scalar instructions do).  Every cycle spent
//     issuing VALU / global-load instructions is therefore lost to the matrix pipe: about 55 cycles
//     per global_load_dwordx4 with its address add, 40 per ds_write_b128, 6-10 per VALU op.
//   * Hence tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, no
//     ds_write pass), per-lane addresses are loop invariant (scalar base advanced with scalar adds,
//     issued from inline asm), and the GroupNorm/SiLU epilogue uses v_exp/v_rcp/v_rsq (<= 2 ulp) on
//     float pairs (v_pk_add/mul/fma_f32).
//   * LDS rows are unpadded 128-byte lines (the DMA destination is lane-linear by construction);
//     bank conflicts are avoided by XOR-swizzling the 16-byte chunk index with (row & 7) on the
//     SOURCE address and on the fragment read.
//   * Software pipeline (SCHED template flag): fragments double-buffered in registers with every
//     ds_read pinned (sched_barrier) in front of the MFMA group that hides it - left alone, hipcc
//     sinks the reads to their first use, merges the two fragment sets and exposes the LDS latency
//     16 times per K tile; ONE barrier per K tile; tile kt+2 is DMA'd into the buffer of tile kt
//     behind the barrier that proves every read of tile kt complete (two LDS buffers suffice), one
//     DMA instruction per four MFMAs over the next two groups instead of a burst: a burst of
//     8 x 1 KB per wave queues on the CU's 64 B/clk vector-memory path and stalls the wave's MFMA issue.
//   * Tile shape: 128x128 per workgroup on 16-deep K tiles (64-byte LDS rows, 32 KB ring), THREE workgroups per CU
//     (round 2; round 1: 32-deep K tiles, two per CU).  One workgroup alone runs its K loop at ~98 % of the pipe; what
//     the others buy is cover for prologue and epilogue.  Four waves (64x64 each) for the plain layers, eight waves
//     (64x32 each, 74 registers, six per SIMD) for the residual layers, whose epilogue waits on a residual DMA.
//     Larger tiles (256x128, 256x256) and four-fragment-set schedules are in the ubench variant table below; all
//     measured equal or slower (profiles/ubench_gemm_*).
//   * Tile quantisation: rows that do not fill a whole round of 3 workgroups x 256 CUs run as small
//     tiles in the SAME launch (layer_pair_kernel): they back-fill CUs as the last big tiles drain.
//   * Where the remaining ~8 % goes (ablations, profiles/ubench_gemm_49152_r01.txt): no epilogue
//     -5.2 %, no in-loop DMA -1.7 %, neither: 150.8 TFLOP/s = 96.8 % of the measured MFMA peak.  The
//     epilogue's ~550 VALU instructions per wave cannot issue under the partner's MFMAs, so it
//     stretches from 4.6 us (alone) to ~27 us and a CU has both workgroups inside the K loop only
//     ~45 % of the time (tools/ubench/timeline_stats.py).
#include "zedo_internal.h"
#include "zedo_tile.h"

#include <atomic>
#include <cstdlib>

#ifdef ZEDO_UBENCH      // the harness (tools/ubench/ubench_gemm.hip) compiles this file with per-workgroup timeline marks
#include "../../tools/ubench/zedo_tile_hooks.inc"
#else
#define TL_MARK(var)
#define TL_FLUSH(t0, t1, t2)
#endif

namespace zedo {

// NBUF = depth of the LDS tile ring (2 for the big tile, whose ring already fills the LDS; 3-4 for the small
// tiles, whose iterations are shorter than the DMA latency).
// BK = K depth of one LDS tile: 32 (128-byte rows) or 16 (64-byte rows: half the LDS, so that three 128x128
// workgroups fit on a CU).
// SCHED (instruction placement inside the K loop): bit 0 = pin every fragment read in front of the MFMA group
// that hides it (the compiler otherwise sinks the ds_reads to their first use and exposes the LDS latency);
// bit 1 = issue the DMA of a tile one instruction at a time between the MFMAs of the two groups that follow the
// barrier instead of as one burst right behind it.
// KSKIP (K == NBUF * BK only, i.e. the ring holds all of K): the last KSKIP fragment groups (8 k each) of the last
// tile are known to be zero in both operands and their MFMAs are not issued - pre_dense: K = 51 padded to 64, the
// group k = 56..63 is padding on both sides; skipping exact zeros leaves every sum bit-identical.
// KQ > 1 (post_dense): the K loop is cut into KQ equal parts, each summed as its own fma chain from zero and the parts
// combined left to right - ((q0 + q1) + q2) + q3 - the order the split launch of small batches produces (EPI_PARTIAL
// tiles + post_reduce_kernel), so that a row's result does not depend on the launch shape.
// out_m0: row of a.out the tile's first row is written to (== m0 except for EPI_PARTIAL tiles, which write quarter sums).
template <int BM, int BN, int WM, int WN, int EPI, int NBUF = 2, int BK = 32, int SCHED = 0, int KSKIP = 0, int KQ = 1>
__device__ __forceinline__ void layer_tile(const LayerArgs &a, const int m0, const int n0, const int out_m0) {
    constexpr int NW = WM * WN;
    constexpr int CPR = BK / 4;                         // 16-byte chunks per tile row (8 or 4)
    constexpr int RPD = 64 / CPR;                       // tile rows moved by one DMA instruction (8 or 16)
    constexpr int KG = BK / 8;                          // fragment groups (8 k each) per tile
    static_assert(BK == 32 || BK == 16, "BK");
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int TJ = TM / 32, TI = TN / 32;
    constexpr int IA = BN / RPD / NW, IB = BM / RPD / NW;   // DMA instructions (1 KB each) per wave per tile
    constexpr int IPW = IA + IB;                        // DMA instructions per wave per tile
    static_assert(TM % 32 == 0 && TN % 32 == 0 && IA >= 1 && IB >= 1 && (BN / RPD) % NW == 0 && (BM / RPD) % NW == 0, "tile");
    static_assert(NBUF >= 2 && (NBUF - 1) * IPW <= 63, "ring depth vs the 6-bit vmcnt");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                    // [NBUF][BN][32]  W tile, chunk-swizzled
    float *Bs = smem + NBUF * BN * BK;   // [NBUF][BM][32]  X tile, chunk-swizzled
    // the epilogue stages WM*32 rows x BN channels in the (then free) tile ring; where that is larger than the ring
    // itself (64-row tiles on a 16-deep ring) the parameter block simply sits behind the larger of the two
    constexpr int RING_F = NBUF * (BM + BN) * BK, STAGE_F = WM * 32 * BN, BODY_F = RING_F > STAGE_F ? RING_F : STAGE_F;
    float *Ps = smem + BODY_F;   // [3][BN]  bias | gamma | beta of this tile's channels (epilogue)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, kh = lane >> 5;

    // ---- DMA addressing: instruction p of this wave fills LDS rows [(wid*I + p)*RPD, +RPD); lane -> (row, pos);
    //      LDS position `pos` of a row holds source chunk pos ^ swz(row), swz(r) = r & 7 (128-byte rows) or
    //      (r >> 2) & 3 (64-byte rows): both make a ds_read_b128 of 16 consecutive rows hit 16 distinct slots
    //      Addresses are kept as (wave-uniform 64-bit base, advanced with scalar adds) + (loop-invariant 32-bit
    //      per-lane byte offset) so that the loads use the SGPR-base form and need no VALU address math.
    const int drow = lane / CPR, dpos = lane % CPR;
    const int dswz = (BK == 32) ? (drow & 7) : ((drow >> 2) & 3);
    const char *Wbase = reinterpret_cast<const char *>(a.W + (size_t)(n0 + wid * IA * RPD) * a.ldw);
    const char *Xbase = reinterpret_cast<const char *>(a.X + (size_t)(m0 + wid * IB * RPD) * a.ldx);
    const unsigned wlane = (unsigned)(drow * a.ldw + (dpos ^ dswz) * 4) * 4u;
    const unsigned xlane = (unsigned)(drow * a.ldx + (dpos ^ dswz) * 4) * 4u;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)smem;   // LDS byte address of As
    auto dma_one = [&](int kt, int buf, int p) {       // instruction p (compile-time after unrolling) of a tile: W rows first, then X rows
        if (p < IA)
            dma16(Wbase + (size_t)kt * (BK * 4) + (size_t)p * (RPD * 4) * a.ldw, wlane,
                  lds0 + (unsigned)(((buf * BN + (wid * IA + p) * RPD) * BK) * 4));
        else
            dma16(Xbase + (size_t)kt * (BK * 4) + (size_t)(p - IA) * (RPD * 4) * a.ldx, xlane,
                  lds0 + (unsigned)(((NBUF * BN + buf * BM + (wid * IB + (p - IA)) * RPD) * BK) * 4));
    };
    auto dma = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < IPW; ++p) dma_one(kt, buf, p);
    };

    // ---- fragment reads: k-chunk c = 2*kg + kh of row i lives at position c ^ swz(i); tile bases are
    //      multiples of 32 rows, so swz(i) == swz(li) and the offsets are loop invariant
    const int fswz = (BK == 32) ? (li & 7) : ((li >> 2) & 3);
    int foff[KG];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) foff[kg] = ((2 * kg + kh) ^ fswz) * 4;
    const float *Ab0 = As + (wn * TN + li) * BK;
    const float *Bb0 = Bs + (wm * TM + li) * BK;
    f32x4 fa0[TI], fb0[TJ], fa1[TI], fb1[TJ];
    auto fread = [&](f32x4(&fa)[TI], f32x4(&fb)[TJ], int buf, int kg) {
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(Ab0 + (buf * BN + i * 32) * BK + foff[kg]);
#pragma unroll
        for (int j = 0; j < TJ; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(Bb0 + (buf * BM + j * 32) * BK + foff[kg]);
    };
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    auto mma = [&](const f32x4(&fa)[TI], const f32x4(&fb)[TJ]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    };
    // one MFMA group with DMA instructions [p0, p0 + cnt) of tile `kt` (ring slot `buf`) spread between its four sub-groups
    constexpr int H1 = (IPW + 1) / 2, H2 = IPW - H1;      // issued behind the barrier / in the first group of the next iteration
    auto mma_dma = [&](const f32x4(&fa)[TI], const f32x4(&fb)[TJ], int kt, int buf, int p0, int cnt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < cnt && q * 4 / cnt == e) {         // cnt <= 4: DMA q goes in front of sub-group q*4/cnt
                    dma_one(kt, buf, p0 + q);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(SCHED >= 0 && SCHED <= 3, "SCHED: bit 0 = pinned reads, bit 1 = spread DMA");
    static_assert(!(SCHED & 2) || H1 <= 4, "spread DMA: at most 4 instructions per MFMA group");

    //   iteration kt (tile kt in ring slot kt % NBUF; F0 = fragments (kt, kg 0) already in registers):
    //       F1 = read(kt,1); MFMA(F0);  F0 = read(kt,2); MFMA(F1);  F1 = read(kt,3); MFMA(F0)
    //       vmcnt((NBUF-2)*IPW); barrier   <- all waves: reads of tile kt complete, DMA of tile kt+1 landed
    //       DMA(tile kt+NBUF -> slot of tile kt);  F0 = read(kt+1, 0);  MFMA(F1)
    TL_MARK(tl0)
    // (s_setprio around prologue / epilogue was measured: no effect - their starvation behind the partner's
    //  pending MFMAs is structural, not a priority matter.)
    // epilogue parameters -> LDS once (LDS reads are free next to MFMAs, VMEM loads are not)
    if (EPI != EPI_PARTIAL && tid < BN / 4) {
        *reinterpret_cast<f32x4 *>(Ps + tid * 4) = *reinterpret_cast<const f32x4 *>(a.bias + n0 + tid * 4);
        if constexpr (EPI == EPI_GN_SILU || EPI == EPI_GN_SILU_RES) {
            *reinterpret_cast<f32x4 *>(Ps + BN + tid * 4) = *reinterpret_cast<const f32x4 *>(a.gamma + n0 + tid * 4);
            *reinterpret_cast<f32x4 *>(Ps + 2 * BN + tid * 4) = *reinterpret_cast<const f32x4 *>(a.beta + n0 + tid * 4);
        }
    }
    const int KT = a.K / BK;
    f32x16 qsum[KQ > 1 ? TI : 1][KQ > 1 ? TJ : 1];       // KQ > 1: ((q0 + q1) + q2) + ... so far
    const int KTQ = KT / KQ;                              // tiles per part (a multiple of NBUF: checked at launch)
#pragma unroll
    for (int t = 0; t < NBUF; ++t) dma(min(t, KT - 1), t);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * IPW) : "memory");   // tile 0 landed
    __syncthreads();
    fread(fa0, fb0, 0, 0);
    TL_MARK(tl1)
    // KT % NBUF == 0 (checked at launch): unrolling by NBUF makes the ring slot a compile-time constant, so
    // every LDS offset folds into an instruction immediate (no VALU address math inside the loop).
    for (int kt0 = 0; kt0 < KT; kt0 += NBUF) {
#pragma unroll
      for (int buf = 0; buf < NBUF; ++buf) {
        const int kt = kt0 + buf;
        const int nxt = (buf + 1 == NBUF) ? 0 : buf + 1;
        const int prv = (buf == 0) ? NBUF - 1 : buf - 1;
        fread(fa1, fb1, buf, 1);
        if constexpr (SCHED & 1) __builtin_amdgcn_sched_barrier(0);
        if constexpr (SCHED & 2) mma_dma(fa0, fb0, min(kt - 1 + NBUF, KT - 1), prv, H1, H2);   // second half of the tile begun behind the last barrier
        else mma(fa0, fb0);
        if constexpr (KG == 4) {
            fread(fa0, fb0, buf, 2);
            if constexpr (SCHED & 1) __builtin_amdgcn_sched_barrier(0);
            mma(fa1, fb1);
            if constexpr (SCHED & 1) __builtin_amdgcn_sched_barrier(0);
            fread(fa1, fb1, buf, 3);
            if constexpr (SCHED & 1) __builtin_amdgcn_sched_barrier(0);
            mma(fa0, fb0);
        }
        const bool skip_last = KSKIP == 1 && KG == 4 && buf == NBUF - 1;   // group 3 of the last tile is all zeros (folds: buf is unrolled)
        // hipcc (ROCm 7.2) emits only lgkmcnt(0) before this barrier: it does not count the outstanding LDS-DMA,
        // so wait explicitly until tile kt+1 has landed (the NBUF-2 younger tiles may stay in flight).
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * IPW) : "memory");
        __syncthreads();
        // Branch-free on purpose (one basic block, so these issue under the MFMAs below): past the last
        // tile the DMA refills a buffer nobody reads again and the fragment read fetches unused values.
        if (skip_last) {
            // nothing left to issue: the tile after the last one does not exist
        } else if constexpr (SCHED & 2) {
            fread(fa0, fb0, nxt, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_dma(fa1, fb1, min(kt + NBUF, KT - 1), buf, 0, H1);
        } else {
            dma(min(kt + NBUF, KT - 1), buf);
            fread(fa0, fb0, nxt, 0);
            if constexpr (SCHED & 1) __builtin_amdgcn_sched_barrier(0);
            mma(fa1, fb1);
            if constexpr (SCHED & 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (KQ > 1) {
          // every MFMA of tiles kt0 .. kt0 + NBUF - 1 is issued, none of the next tile: a part ends exactly here
          if ((kt0 + NBUF) % KTQ == 0) {
              const bool first = (kt0 + NBUF == KTQ);
#pragma unroll
              for (int i = 0; i < TI; ++i)
#pragma unroll
                  for (int j = 0; j < TJ; ++j)
#pragma unroll
                      for (int e = 0; e < 16; ++e) {
                          qsum[i][j][e] = first ? acc[i][j][e] : qsum[i][j][e] + acc[i][j][e];
                          acc[i][j][e] = 0.0f;
                      }
          }
      }
    }
    if constexpr (KQ > 1) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = qsum[i][j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // drain the trailing (unused) DMA before any wave of the workgroup may exit
    TL_MARK(tl2)

    // ---------------- epilogue: all global traffic goes through the (now free) LDS as 1 KB wave accesses ----
    // Direct stores from the accumulator layout would be 16 bytes per lane at a 4 KB row stride: 4x the VMEM
    // instructions, and VMEM issue time is matrix-pipe time for the next workgroup's MFMAs (see header).
    // Phase j stages rows {wm*TM + j*32 + li} x BN channels; 16-byte chunk c of stage row sr sits at position
    // c ^ (sr & 7), which makes the per-lane ds_write_b128 and the row-wise read-out both conflict free and
    // lets the residual (EPI_GN_SILU_RES: h, EPI_SDE: x) arrive by LDS-DMA in the same layout.
    {
        constexpr int NT = NW * 64;
        constexpr int SR = WM * 32;          // stage rows per phase
        constexpr int CPRW = BN / 4;         // 16-byte chunks per stage row
        constexpr int RPI = 64 / CPRW;       // stage rows filled by one DMA instruction
        static_assert(CPRW <= 64 && 64 % CPRW == 0 && SR % (RPI * NW) == 0 && (SR * CPRW) % NT == 0, "stage shape");
        static_assert(SR * BN == STAGE_F, "stage size");
        constexpr bool HAS_RES = (EPI == EPI_GN_SILU_RES || EPI == EPI_SDE);
        float *S = smem;
        float *obase = a.out + (size_t)out_m0 * a.ldo + n0;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            if constexpr (HAS_RES) {
                constexpr int NI = SR / RPI / NW;   // DMA instructions per wave
#pragma unroll
                for (int p = 0; p < NI; ++p) {
                    const int idx = wid * NI + p;                         // wave-uniform
                    const int sr = idx * RPI + lane / CPRW, pos = lane % CPRW;
                    const int grow = (sr >> 5) * TM + j * 32 + (sr & 31);
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(obase) +
                                                                         ((unsigned)grow * (unsigned)a.ldo + (unsigned)((pos ^ (sr & 7)) * 4)) * 4u),
                        (__attribute__((address_space(3))) void *)(S + idx * RPI * BN), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                float o[16];
                f32x4 b4[4], ga[4], be[4];
                const float *pc = Ps + wn * TN + i * 32 + 4 * kh;   // channel of accumulator r: + (r&3) + 8*(r>>2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    b4[g] = *reinterpret_cast<const f32x4 *>(pc + 8 * g);
                    if constexpr (EPI == EPI_GN_SILU || EPI == EPI_GN_SILU_RES) {
                        ga[g] = *reinterpret_cast<const f32x4 *>(pc + BN + 8 * g);
                        be[g] = *reinterpret_cast<const f32x4 *>(pc + 2 * BN + 8 * g);
                    }
                }
                epilogue_values<EPI>(acc[i][j], b4, ga, be, a.sde_c, o);
                float *srow = S + (wm * 32 + li) * BN;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = (wn * TN + i * 32 + 8 * g + 4 * kh) >> 2;
                    f32x4 *slot = reinterpret_cast<f32x4 *>(srow + ((c ^ (li & 7)) << 2));
                    f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                    if constexpr (EPI == EPI_GN_SILU_RES) {          // h = h + h2 (model.py:288)
                        const f32x4 h = *slot;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += h[e];
                    } else if constexpr (EPI == EPI_SDE) {
                        const f32x4 x = *slot;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(a.sde_a, x[e], v[e]);   // x' = a x + [c (eps)]: one fma
                    }
                    *slot = v;
                }
            }
            __syncthreads();
            if constexpr (EPI == EPI_SDE && BN == XLD) {
                // reprojection correction of the next iteration on the updated rows, one lane per row, straight from the
                // stage (chunk c of stage row sr sits at position c ^ (sr & 7)); same source as reproj_step_kernel
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
#pragma unroll
            for (int pass = 0; pass < SR * CPRW / NT; ++pass) {
                const int qi = pass * NT + tid;
                const int sr = qi / CPRW, c = qi % CPRW;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(S + sr * BN + ((c ^ (sr & 7)) << 2));
                const int grow = (sr >> 5) * TM + j * 32 + (sr & 31);
                // wave-uniform base + 32-bit byte offset (a tile spans < 4 GB): the store takes its SGPR-base form and the
                // offset is one add per store instead of a 64-bit multiply-add chain (VALU issue time is matrix-pipe time)
                const unsigned off = ((unsigned)grow * (unsigned)a.ldo + (unsigned)(c * 4)) * 4u;
                *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(obase) + off) = v;
            }
            if (j + 1 < TJ) __syncthreads();
        }
    }
    TL_FLUSH(tl0, tl1, tl2)
}

// One tile per workgroup: block index -> tile.
template <int BM, int BN, int WM, int WN, int EPI, int NBUF = 2, int BK = 32, int SCHED = 0, int KSKIP = 0, int KQ = 1>
__device__ __forceinline__ void layer_body(const LayerArgs &a, const int bid, const int nwg) {
    // XCD-aware, bijective block -> tile map: the hardware places block b on XCD b % 8; give every
    // XCD a contiguous range of tiles so that the column tiles of one row tile share one L2.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int ncol = a.N / BN;
    if constexpr (EPI == EPI_PARTIAL) {
        // one K quarter per workgroup: block lid -> (row tile lid / 4, quarter lid % 4); a.K is the quarter's depth;
        // the quarter sums of a row tile land next to each other in a.out = [tile][quarter][BM][BN]
        LayerArgs b = a;
        const int tile = lid >> 2, q = lid & 3;
        b.X = a.X + (size_t)q * a.K;
        b.W = a.W + (size_t)q * a.K;
        layer_tile<BM, BN, WM, WN, EPI, NBUF, BK, SCHED, KSKIP, 1>(b, tile * BM, 0, lid * BM);
        return;
    }
    layer_tile<BM, BN, WM, WN, EPI, NBUF, BK, SCHED, KSKIP, KQ>(a, (lid / ncol) * BM, (lid % ncol) * BN, (lid / ncol) * BM);
}

template <int BM, int BN, int WM, int WN, int EPI, int NBUF = 2, int BK = 32, int SCHED = 0, int WPE = 1, int KSKIP = 0, int KQ = 1>
__global__ __launch_bounds__(WM *WN * 64, WPE) void layer_kernel(LayerArgs a) {
    layer_body<BM, BN, WM, WN, EPI, NBUF, BK, SCHED, KSKIP, KQ>(a, blockIdx.x, gridDim.x);
}

// One launch, two tile shapes: workgroups [0, nbig) run 128x128 tiles on the rows that fill whole rounds of the
// chip, workgroups [nbig, grid) run 64x64 (four-wave layers; until round 3: 32x128) / 64x128 (eight-wave layers) tiles on the
// remainder rows.  Workgroups are dispatched in order, so the
// small tiles start as CUs run out of big tiles and fill the tail of the launch instead of costing a separate,
// latency-bound launch (40 us -> ~26 us per layer at 50 750 rows).  Both shapes of a launch use the same block size.
// SCHED of the tile shapes (A/B/A/B-measured in rounds 1-4, profiles/ubench_gemm_*; the alternatives lived behind -D knobs until round 5):
// 128x128 on 32-deep K tiles: pinned reads + spread DMA; the 16-deep product tiles: pinned reads; remainder tiles: compiler's order.
constexpr int SCHED_BIG = 3, SCHED_SMALL = 0, SCHED_THIN = 3, SCHED_BK16 = 1;

// W8 = 0 (plain layers): 4 waves per workgroup (64x64 per wave, THREE workgroups per CU on the 16-deep ring: 0.5-0.9 % faster than two on
// 32-deep tiles; remainder in 64x64 tiles); W8 = 1 (residual layers): 8 waves (64x32 per wave, 74 registers, six waves per SIMD; remainder
// in 64x128 tiles) - measured faster for the residual epilogue, whose residual DMA + wait has more co-resident waves to hide behind.
constexpr int PAIR_BK = 16, PLAIN_WGS = 3, PLAIN_WPE = 2;
template <int EPI, int W8>
// (a waves-per-SIMD bound >= 2 also makes hipcc keep the accumulators in VGPRs: no v_accvgpr_read/write, -0.6 %)
__global__ __launch_bounds__(W8 ? 512 : 256, W8 ? 6 : PLAIN_WGS) void layer_pair_kernel(LayerArgs big, LayerArgs small, int nbig) {
    // diagnostic: the shader clock this launch really runs at (power management differs box to box and with the load)
    long long c0 = 0, w0 = 0;
    const bool probe = big.clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    if (probe) { c0 = clock64(); w0 = wall_clock64(); }
    if constexpr (W8) {
        if ((int)blockIdx.x < nbig) layer_body<128, 128, 2, 4, EPI, 2, PAIR_BK, SCHED_BK16>(big, blockIdx.x, nbig);
        else layer_body<64, 128, 2, 4, EPI, 2, 32, SCHED_SMALL>(small, (int)blockIdx.x - nbig, (int)gridDim.x - nbig);
    } else {
        if ((int)blockIdx.x < nbig) layer_body<128, 128, 2, 2, EPI, 2, PAIR_BK, SCHED_BK16>(big, blockIdx.x, nbig);
        // remainder rows on 64x64 tiles (round 4; rounds 1-3: 32x128): the same number of tiles and the same 512-MFMA chain per wave,
        // 64 + 64 instead of 32 + 128 operand rows per K tile = 16 instead of 20 LDS-DMA instructions (6 350 rows: 505.8 -> 496.3 ms
        // per pass, 50 750 rows: 3107.8 -> 3099.9 ms, A/B/A/B on one box; bit-identical)
        else layer_body<64, 64, 2, 2, EPI, 2, 32, SCHED_SMALL>(small, (int)blockIdx.x - nbig, (int)gridDim.x - nbig);
    }
    if (probe) { big.clk[0] = clock64() - c0; big.clk[1] = wall_clock64() - w0; }
}

// (per-device launch state: allow_lds / num_cus of zedo_internal.h)

template <int EPI, int W8>
static hipError_t launch_pair(const LayerArgs &big, const LayerArgs &small, hipStream_t st) {
    constexpr int SM = 64, SN = W8 ? 128 : 64;             // remainder tile rows / columns
    constexpr size_t lds_big = ((size_t)2 * (128 + 128) * PAIR_BK + 3 * 128) * sizeof(float);
    constexpr size_t lds_small = ((size_t)2 * (SM + SN) * 32 + 3 * SN) * sizeof(float);
    constexpr size_t lds = lds_big > lds_small ? lds_big : lds_small;
    if (big.Mp % 128 || small.Mp % SM || big.N % 128 || big.K % 64) return hipErrorInvalidValue;
    auto kern = layer_pair_kernel<EPI, W8>;
    static std::atomic<bool> attr_done[MAX_DEVICES];      // per instantiation and per device
    if (hipError_t e = allow_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const int nbig = (big.Mp / 128) * (big.N / 128), nsmall = (small.Mp / SM) * (small.N / SN);
    hipLaunchKernelGGL(kern, dim3(nbig + nsmall), dim3(W8 ? 512 : 256), lds, st, big, small, nbig);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int EPI, int NBUF = 2, int BK = 32, int SCHED = 0, int WPE = 1, int KSKIP = 0, int KQ = 1>
static hipError_t launch_cfg(const LayerArgs &a, hipStream_t st) {
    constexpr size_t ring_f = (size_t)NBUF * (BM + BN) * BK, stage_f = (size_t)WM * 32 * BN;
    constexpr size_t lds = ((ring_f > stage_f ? ring_f : stage_f) + 3 * BN) * sizeof(float);
    if (a.Mp <= 0 || a.Mp % BM || a.N % BN || a.K % (BK * NBUF * KQ)) return hipErrorInvalidValue;
    if (KSKIP && a.K != BK * NBUF) return hipErrorInvalidValue;
    auto kern = layer_kernel<BM, BN, WM, WN, EPI, NBUF, BK, SCHED, WPE, KSKIP, KQ>;
    static std::atomic<bool> attr_done[MAX_DEVICES];      // per instantiation and per device
    if (hipError_t e = allow_lds(reinterpret_cast<const void *>(kern), lds, attr_done); e != hipSuccess) return e;
    const int nwg = (a.Mp / BM) * (a.N / BN) * (EPI == EPI_PARTIAL ? 4 : 1);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), lds, st, a);
    return hipGetLastError();
}

static LayerArgs rows_of(const LayerArgs &a, int row0, int rows) {
    LayerArgs b = a;
    b.X = a.X + (size_t)row0 * a.ldx;
    b.out = a.out + (size_t)row0 * a.ldo;
    b.Mp = rows;
    return b;
}

// Launches that do not fill the chip with 128x128 tiles (small and mid-size batches, schedule tables).  A CU's time
// is ~6 us of prologue + epilogue plus the MFMA time of the tiles it ends up with, so what matters is how evenly the
// tiles spread over the 256 CUs: finer tiles balance better, coarser tiles issue fewer DMAs per MFMA.  The choice
// below minimises  rounds * 6 us + ceil(tiles / CUs) * tile_time  over the three shapes (model checked against
// tools/ubench/ubench_gemm for 512..8192 rows: within 5 %); up to 256 tiles the 32-row shape on a 4-deep ring wins
// on latency alone.
template <int EPI>
static hipError_t launch_small(const LayerArgs &a, hipStream_t st) {
    const int ncol = a.N / 128, cus = num_cus();
    // up to one tile per CU: 64x64 tiles (four waves of 32x32, the same 512-MFMA chain per wave as a 32x128 tile and the same
    // number of workgroups) stream 64 + 64 operand rows per K tile instead of 32 + 128: 16 LDS-DMA instructions instead of 20,
    // each worth ~60 matrix-pipe cycles on the issuing SIMD - 886 rows: 19.9 -> 19.3 us per layer, 99.2 -> 96.8 ms per
    // 1000-step pass (round 4; bit-identical).
    if (a.Mp % 64 == 0 && (a.Mp / 64) * (a.N / 64) <= cus)
        return launch_cfg<64, 64, 2, 2, EPI, 4, 32, SCHED_THIN & 1>(a, st);
    if (a.Mp % 32 == 0 && (a.Mp / 32) * ncol <= cus) return launch_cfg<32, 128, 1, 4, EPI, 4, 32, SCHED_THIN & 1>(a, st);
    auto cost = [&](int bm, int slots_per_cu, double tile_us) {
        if (a.Mp % bm) return 1e30;
        const long tiles = (long)(a.Mp / bm) * ncol;
        const long rounds = (tiles + (long)cus * slots_per_cu - 1) / ((long)cus * slots_per_cu);
        return rounds * 6.0 + (double)((tiles + cus - 1) / cus) * tile_us;
    };
    const double kdepth = a.K / 1024.0;                               // tile times below are for K = 1024
    const double c32 = cost(32, 3, 14.5 * kdepth), c64 = cost(64, 3, 28.2 * kdepth), c128 = cost(128, 2, 55.1 * kdepth);
    if (c32 <= c64 && c32 <= c128) return launch_cfg<32, 128, 1, 4, EPI, 2, 32, SCHED_THIN>(a, st);
    if (c64 <= c128) return launch_cfg<64, 128, 2, 4, EPI, 2, 32, SCHED_THIN>(a, st);
    return launch_cfg<128, 128, 2, 2, EPI, 2, 32, SCHED_BIG, PLAIN_WPE>(a, st);
}

// N == 1024 or 512 (hidden / embedding width): 128x128 tiles on 16-deep K tiles, three workgroups per CU, on the rows that
// fill whole rounds of the chip; the remainder rows as small tiles in the same launch (launch_pair).  734 / 739 us per
// plain / residual layer at 50 752 rows (145 / 144 TFLOP/s); 256x256 tiles (one workgroup per CU): 719 / 748 us at 49 152.
template <int EPI>
static hipError_t launch_wide(const LayerArgs &a, hipStream_t st) {
    // K <= 64 (pre_dense): almost no MFMA work per output, the layer is bound by writing the activation:
    // many small co-resident workgroups overlap their stores, one big tile per CU cannot.
    // (measured at 50 750 rows: 64x128 82 us; 32x128 92 us; 128x128 91 us; 64x256 108 us)
    if (a.K <= 64) {
        // a.kzero8: the caller vouches that k >= K - 8 is zero in X and W (pre_dense: 51 real inputs)
        if (a.K == 64 && a.kzero8) return launch_cfg<64, 128, 2, 4, EPI, 2, 32, SCHED_THIN, 1, 1>(a, st);
        return launch_cfg<64, 128, 2, 4, EPI, 2, 32, SCHED_THIN>(a, st);
    }
    constexpr int WG_PER_CU = 3;                                          // both pair kernels: three 128x128 workgroups per CU
    constexpr int W8 = (EPI == EPI_GN_SILU_RES) ? 1 : 0;
    const int per_round = num_cus() * WG_PER_CU * 128 / (a.N / 128);      // rows covered by one full round of 128x128 tiles
    int rows_big = (a.Mp / per_round) * per_round;
    if (rows_big == 0) {
        // Less than one round of resident workgroups (mid-size batches, strong-scaling shards): a single tile shape
        // quantises badly - 6 400 rows are 400 128x128 tiles on 256 CUs, i.e. two tiles on 144 CUs and one on the rest
        // (124 us where 85 would do).  Instead: 128x128 tiles on as many rows as give every CU the SAME number of them
        // (whole multiples of one tile per CU), the rest as small tiles dispatched last in the same launch, which spread
        // 2-3 per CU behind the big ones.  Estimated with the tile times of launch_small; taken when it beats the best
        // single shape.  (Bit-identical either way: every tile shape issues the same products in the same order.)
        const int one_each = num_cus() * 128 / (a.N / 128);               // rows that put one 128x128 tile on every CU
        const int mix_big = (a.Mp / one_each) * one_each;
        constexpr int SMR = (W8 ? 64 : 32);
        if (mix_big > 0 && a.Mp > mix_big && (a.Mp - mix_big) % SMR == 0) {
            const double kd = a.K / 1024.0;
            const long small_tiles = (long)((a.Mp - mix_big) / SMR) * (a.N / 128);
            const double est_mix = (mix_big / one_each) * 55.1 * kd + (double)((small_tiles + num_cus() - 1) / num_cus()) * (W8 ? 28.2 : 13.9) * kd + 6.0;
            auto single = [&](int bm, int slots, double t) {
                if (a.Mp % bm) return 1e30;
                const long tiles = (long)(a.Mp / bm) * (a.N / 128);
                return (double)((tiles + (long)num_cus() * slots - 1) / ((long)num_cus() * slots)) * 6.0 + (double)((tiles + num_cus() - 1) / num_cus()) * t * kd;
            };
            const double est_single = fmin(fmin(single(32, 3, 14.5), single(64, 3, 28.2)), single(128, 2, 55.1));
            if (est_mix < est_single) rows_big = mix_big;
        }
    }
    const int rows_small = a.Mp - rows_big;                       // multiple of 64 (BATCH_PAD)
    if (rows_small > 0 && rows_big > 0) {
        // eight-wave workgroups bring 64-row remainder tiles along; a short remainder finishes sooner as 32-row tiles
        if (W8 && rows_small >= 1536) return launch_pair<EPI, W8>(rows_of(a, 0, rows_big), rows_of(a, rows_big, rows_small), st);
        return launch_pair<EPI, 0>(rows_of(a, 0, rows_big), rows_of(a, rows_big, rows_small), st);
    }
    // one shape only: whole rounds of big tiles and nothing else (the pair kernel with no small tile), or a batch below one round
    if (rows_big > 0) return launch_pair<EPI, W8>(rows_of(a, 0, rows_big), rows_of(a, rows_big, 0), st);
    return launch_small<EPI>(a, st);
}

hipError_t launch_layer(const LayerArgs &a, int epilogue, hipStream_t st) {
    if (a.N == XLD) {  // post_dense: 51 (padded to 64) output channels, one column tile
        // bandwidth bound (reads the 4 KB activation row once): 64-row tiles, two workgroups per CU, 4-deep ring
        // (measured 70.6 us vs 75.7 us for 128-row tiles and 88 us for 32-row tiles at 50 750 rows)
        // K is summed as four quarter chains combined left to right in EVERY shape (LayerArgs::scratch): up to
        // POST_SPLIT_ROWS rows one workgroup per (32-row tile, quarter) + post_reduce_kernel - a 32x64 tile is a 512-MFMA
        // dependent chain on 2 of a CU's 4 SIMDs, 18.9 us whatever the batch; four 128-MFMA chains on 4x the CUs take a
        // third of that - above it the quarters are folded inside one 64x64 tile (the chip is full there).
        if (epilogue != EPI_SDE && epilogue != EPI_BIAS) return hipErrorInvalidValue;
        if (a.K % (4 * 32 * 4)) return hipErrorInvalidValue;
        if (a.Mp <= POST_SPLIT_ROWS && a.Mp % 32 == 0 && a.scratch) {
            LayerArgs p = a;
            p.K = a.K / 4; p.out = a.scratch; p.ldo = XLD;
            if (hipError_t e = launch_cfg<32, 64, 1, 2, EPI_PARTIAL, 4, 32, SCHED_THIN>(p, st); e != hipSuccess) return e;
            const bool sde = epilogue == EPI_SDE;
            return launch_post_reduce(sde ? a.out : nullptr, a.scratch, a.bias, a.sde_a, a.sde_c, sde ? 1 : 0, sde ? nullptr : a.out,
                                      sde ? a.rp_geom : nullptr, a.rp_T, a.rp_solve, a.rp_B, a.Mp, a.rp_N, a.rp_row0, st);
        }
        const bool small = a.Mp <= 8192 && a.Mp % 32 == 0;
        switch (epilogue) {
            case EPI_SDE: return small ? launch_cfg<32, 64, 1, 2, EPI_SDE, 4, 32, SCHED_THIN, 1, 0, 4>(a, st) : launch_cfg<64, 64, 2, 2, EPI_SDE, 4, 32, SCHED_THIN, 1, 0, 4>(a, st);
            case EPI_BIAS: return small ? launch_cfg<32, 64, 1, 2, EPI_BIAS, 4, 32, SCHED_THIN, 1, 0, 4>(a, st) : launch_cfg<64, 64, 2, 2, EPI_BIAS, 4, 32, SCHED_THIN, 1, 0, 4>(a, st);
        }
        return hipErrorInvalidValue;
    }
    if (a.N % 128) return hipErrorInvalidValue;
    switch (epilogue) {
        case EPI_GN_SILU: return launch_wide<EPI_GN_SILU>(a, st);
        case EPI_GN_SILU_RES: return launch_wide<EPI_GN_SILU_RES>(a, st);
        case EPI_BIAS: return launch_wide<EPI_BIAS>(a, st);
        case EPI_BIAS_SILU: return launch_wide<EPI_BIAS_SILU>(a, st);
    }
    return hipErrorInvalidValue;
}

// ---- diagnostic: what this box's matrix pipe sustains right now (clock / power state differ box to box) ----------
__global__ __launch_bounds__(256) void mfma_probe_kernel(float *out, int iters, long long *clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.0f;
    long long t0 = 0, w0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t0 = clock64(); w0 = wall_clock64(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = clock64() - t0; clk[1] = wall_clock64() - w0; }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// The fp16 twin: a bare v_mfma_f32_32x32x16_f16 stream (what the split-fp16 hidden layers issue, zedo_gemm16.hip), two waves per SIMD,
// four independent accumulators per wave, operands that differ per lane and per instruction (two alternating fragment sets of
// pseudo-random fp16 values: the power a matrix pipe draws, and with it the clock power management grants, depends on the bits that
// toggle) - no LDS, no vector memory.  -> what THIS box's fp16 matrix pipe sustains at the clock it is granted under a pure MFMA load:
// the attainable ceiling that the 2.5 PFLOP/s datasheet figure (quoted at 2.4 GHz) is not.
__global__ __launch_bounds__(256) void mfma16_probe_kernel(float *out, int iters, long long *clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8 fa[4], fb[4];
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 17u);
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u; fa[i][e] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 4096.0f));
            h = h * 1664525u + 1013904223u; fb[i][e] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 4096.0f));
        }
    long long t0 = 0, w0 = 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t0 = clock64(); w0 = wall_clock64(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s0 = (u & 1) * 2;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s0], fb[s0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s0], fb[s0 + 1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s0 + 1], fb[s0], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s0 + 1], fb[s0 + 1], acc[3], 0, 0, 0);
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = clock64() - t0; clk[1] = wall_clock64() - w0; }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

hipError_t probe_mfma_peak(int iters, double *tflops, double *shader_ghz, hipStream_t st, bool f16) {
    const int blocks = num_cus() * 2;          // 2 blocks of 4 waves per CU: 2 waves per SIMD, dependent chains hidden
    float *d_out = nullptr;
    long long *d_clk = nullptr, clk[2] = {0, 0};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    hipError_t e = hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_clk, 16);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) {
        if (f16) hipLaunchKernelGGL(mfma16_probe_kernel, dim3(blocks), dim3(256), 0, st, d_out, iters / 4, d_clk);   // warm the clocks
        else hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, st, d_out, iters / 4, d_clk);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    if (e == hipSuccess) {
        if (f16) hipLaunchKernelGGL(mfma16_probe_kernel, dim3(blocks), dim3(256), 0, st, d_out, iters, d_clk);
        else hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, st, d_out, iters, d_clk);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        const double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * (f16 ? 16 : 2);     // 4 waves x iters x 16 MFMAs of 32x32xK
        if (tflops) *tflops = flop / ms / 1e9;
        if (shader_ghz) *shader_ghz = clk[1] > 0 ? (double)clk[0] / ((double)clk[1] / 100e6) / 1e9 : 0.0;   // wall_clock64 ticks at 100 MHz
    } else {
        (void)hipStreamSynchronize(st);        // nothing of this probe may still run when its buffers go away
    }
    // one way out: whatever was created is released
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    (void)hipFree(d_clk);
    return e;
}

#ifdef ZEDO_UBENCH      // variant table of the harness (tools/ubench/ubench_gemm.hip): not part of the library
#include "../../tools/ubench/zedo_gemm_variants.inc"
#endif

}  // namespace zedo
