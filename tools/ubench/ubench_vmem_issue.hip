// What does an LDS-DMA instruction cost the fp16 MFMA stream of a CU - and does it depend on WHO issues it and WHEN?
// (round 5; follows tools/ubench/ubench_coissue.hip, which measured "MFMA time + VMEM time" with one filler wave on EVERY SIMD)
//
//   mode 0   waves 0-3 (one per SIMD): NM x v_mfma_f32_32x32x16_f16; waves 4-7: loader waves, selected by `mask`, share
//            NF 1-KB LDS-DMA instructions between them (mask 15 = the old experiment, mask 1 = ONE loader wave per CU)
//   mode 1   no loader waves; every MFMA wave issues one LDS-DMA per G of its own MFMAs, all four waves at the same MFMA
//            index (stagger 0) or at index simd * G / 4 (stagger 1: the CU sees one DMA per G/4 MFMA slots instead of four at once)
//   mode 2   as mode 1, but the DMAs of a 24-MFMA block are all issued by the wave whose turn it is (block % 4): concentrated issue
// Per wave: {begin, end} of s_memtime and HW_ID; the host prints per-SIMD MFMA-wave cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ubench_vmem_issue.hip -o tools/ubench/ubench_vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
}

constexpr int WINDOW_MAX = 1 << 20;  // bytes of source reserved per CU; `window` (power of two <= this) is what a CU cycles through:
                                     // 64 KB per CU = 2 MB per XCD stays in the 4 MB L2 and misses the 32 KB L1; 1 MB per CU streams from MALL / HBM

template <int MODE, int G>
__global__ __launch_bounds__(512) void vk(const char *gsrc, long long *rec, int nm, int nf, int mask, int stagger, int window, int pace) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const char *src = gsrc + (size_t)blockIdx.x * WINDOW_MAX;
    const unsigned wmask = (unsigned)window - 1u;
    __syncthreads();
    const long long t0 = clock64();
    float sink = 0.f;
    if (wid < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        f16x8 ha, hb;
        for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(lane * 1e-3f); hb[e] = (_Float16)(1.0f + blockIdx.x); }
        if constexpr (MODE == 0) {
            for (int it = 0; it < nm / 16; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[u & 3], 0, 0, 0);
            }
        } else if constexpr (MODE == 1) {
            const int simd = (hwid >> 4) & 3;
            const int ph = __builtin_amdgcn_readfirstlane(stagger ? (simd * G) / 4 : 0);
            unsigned o = 0;
            for (int it = 0; it < nm / G; ++it) {
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[u & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (u == ph) {
                        dma16(src + o, (unsigned)lane * 16u, lds0 + (wid * 8 + (it & 7)) * 1024);
                        o = (o + 4096u) & wmask;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int simd = (hwid >> 4) & 3;
            unsigned o = 0;
            constexpr int PER = 4 * 24 / G;          // DMAs of a 24-MFMA block, all from one wave
            constexpr int EVERY = 24 / PER > 0 ? 24 / PER : 1;
            for (int it = 0; it < nm / 24; ++it) {
                const bool mine = __builtin_amdgcn_readfirstlane((it & 3) == simd);
#pragma unroll
                for (int u = 0; u < 24; ++u) {
                    acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[u & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (mine && (u % EVERY) == 0 && u / EVERY < PER) {
                        dma16(src + o, (unsigned)lane * 16u, lds0 + (wid * 8 + (u & 7)) * 1024);
                        o = (o + 4096u) & wmask;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (mine) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) sink += acc[i][e];
    } else if (MODE == 0 && ((mask >> (wid - 4)) & 1)) {
        const int nload = __builtin_popcount(mask & 15);
        const int mine = nf / nload;
        unsigned o = (wid - 4) * 1024u;
        for (int it = 0; it < mine / 8; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dma16(src + o, (unsigned)lane * 16u, lds0 + (32 + (wid - 4) * 8 + u) * 1024);
                o = (o + 4096u) & wmask;
                for (int p = 0; p < pace; ++p) asm volatile("s_nop 15");      // 16 idle cycles each: the loader is PACED, not back-pressured
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    if (lane == 0) {
        long long *r = rec + ((size_t)blockIdx.x * 8 + wid) * 4;
        r[0] = t0; r[1] = t1; r[2] = hwid; r[3] = (long long)sink;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Res { float ms; double mfma_cyc[4]; double load_cyc[4]; int simd_of_wave[8]; };

template <int MODE, int G>
Res run(const char *gsrc, long long *rec, int nm, int nf, int mask, int stagger, int window = 1 << 16, int pace = 0) {
    constexpr size_t LDS = 96 * 1024;     // one workgroup per CU
    auto kern = vk<MODE, G>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS, 0, gsrc, rec, nm, nf, mask, stagger, window, pace);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS, 0, gsrc, rec, nm, nf, mask, stagger, window, pace);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    Res r{}; CK(hipEventElapsedTime(&r.ms, e0, e1)); r.ms /= 10;
    std::vector<long long> h(256 * 8 * 4);
    CK(hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost));
    double cnt[4] = {0, 0, 0, 0}, lcnt[4] = {0, 0, 0, 0};
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 8; ++w) {
            const long long *p = &h[((size_t)b * 8 + w) * 4];
            const int simd = (int)((p[2] >> 4) & 3);
            if (b == 0) r.simd_of_wave[w] = simd;
            if (w < 4) { r.mfma_cyc[simd] += (double)(p[1] - p[0]); cnt[simd] += 1; }
            else if (MODE == 0 && ((mask >> (w - 4)) & 1)) { r.load_cyc[simd] += (double)(p[1] - p[0]); lcnt[simd] += 1; }
        }
    for (int s = 0; s < 4; ++s) { if (cnt[s]) r.mfma_cyc[s] /= cnt[s]; if (lcnt[s]) r.load_cyc[s] /= lcnt[s]; }
    return r;
}

int main() {
    char *gsrc; long long *rec;
    CK(hipMalloc(&gsrc, (size_t)256 * WINDOW_MAX)); CK(hipMemset(gsrc, 0, (size_t)256 * WINDOW_MAX));
    CK(hipMalloc(&rec, 256 * 8 * 4 * 8));
    const int nm = 24 * 1024;          // MFMAs per wave: 786 432 cycles per SIMD at 32 each
    const Res base = run<0, 4>(gsrc, rec, nm, 0, 0, 0);
    printf("waves -> SIMD (block 0): "); for (int w = 0; w < 8; ++w) printf("%d ", base.simd_of_wave[w]); printf("\n");
    printf("MFMA alone: %.3f ms; cycles per MFMA by SIMD: %.2f %.2f %.2f %.2f\n", base.ms, base.mfma_cyc[0] / nm, base.mfma_cyc[1] / nm, base.mfma_cyc[2] / nm, base.mfma_cyc[3] / nm);
    const double base_c = base.mfma_cyc[0] / nm;
    for (int window : {1 << 14, 1 << 16, 1 << 20}) {
        printf("\n==== source window per CU: %d KB (%s) ====\n", window >> 10, window <= (1 << 14) ? "L1-resident" : window <= (1 << 17) ? "L2-resident, L1-missing" : "streams from MALL / HBM");
        const int nf = nm;             // as many 1-KB DMAs per CU as one wave issues MFMAs: one DMA per 32 cycles of a CU if it kept up
        printf("mode 0: %d LDS-DMA per CU from separate loader waves (`mask` bit i = wave 4 + i), `pace` x 16 idle cycles after each DMA\n", nf);
        for (int mask : {15, 1, 3}) {
            for (int pace : {0, 2, 6, 12}) {
                const Res alone = run<0, 4>(gsrc, rec, 0, nf, mask, 0, window, pace);
                const Res both = run<0, 4>(gsrc, rec, nm, nf, mask, 0, window, pace);
                const double lsum = both.load_cyc[0] + both.load_cyc[1] + both.load_cyc[2] + both.load_cyc[3];
                const int nl = __builtin_popcount(mask);
                printf("  mask %2d pace %2d: loaders alone %.3f ms | both %.3f ms | loader-wave cycles per own DMA (both) %.1f = one DMA per %.1f cycles of the CU | MFMA-wave cycles per MFMA by SIMD %.2f %.2f %.2f %.2f | MFMA cycles lost per DMA of the CU: %.1f\n",
                       mask, pace, alone.ms, both.ms, lsum / nf, lsum / nf / nl, both.mfma_cyc[0] / nm, both.mfma_cyc[1] / nm, both.mfma_cyc[2] / nm, both.mfma_cyc[3] / nm,
                       ((both.mfma_cyc[0] + both.mfma_cyc[1] + both.mfma_cyc[2] + both.mfma_cyc[3]) / 4 - base.mfma_cyc[0]) * 4 / nf);
            }
        }
        printf("mode 1: every MFMA wave issues one LDS-DMA per G of its own MFMAs (in phase / staggered by SIMD)\n");
        auto m1 = [&](auto gtag) {
            constexpr int GG = decltype(gtag)::value;
            for (int st : {0, 1}) {
                const Res r = run<1, GG>(gsrc, rec, nm, 0, 0, st, window, 0);
                const double per = (r.mfma_cyc[0] + r.mfma_cyc[1] + r.mfma_cyc[2] + r.mfma_cyc[3]) / 4 / nm;
                printf("  G = %2d stagger %d: %.3f ms; cycles per MFMA %.2f -> %.1f cycles per DMA\n", GG, st, r.ms, per, (per - base_c) * GG);
            }
        };
        m1(std::integral_constant<int, 4>{}); m1(std::integral_constant<int, 8>{}); m1(std::integral_constant<int, 12>{});
        printf("mode 2: the same DMA count, but a 24-MFMA block's DMAs all come from ONE wave (turns rotate over the SIMDs)\n");
        {
            const Res r4 = run<2, 4>(gsrc, rec, nm, 0, 0, 0, window, 0), r8 = run<2, 8>(gsrc, rec, nm, 0, 0, 0, window, 0), r12 = run<2, 12>(gsrc, rec, nm, 0, 0, 0, window, 0);
            for (auto pr : {std::make_pair(4, r4), std::make_pair(8, r8), std::make_pair(12, r12)}) {
                const Res &r = pr.second;
                const double per = (r.mfma_cyc[0] + r.mfma_cyc[1] + r.mfma_cyc[2] + r.mfma_cyc[3]) / 4 / nm;
                printf("  G = %2d: %.3f ms; cycles per MFMA %.2f -> %.1f cycles per DMA\n", pr.first, r.ms, per, (per - base_c) * pr.first);
            }
        }
    }
    return 0;
}
