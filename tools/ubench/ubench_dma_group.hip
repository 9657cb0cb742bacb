// Does an LDS-DMA burst get cheaper when its instructions share ONE M0 write?  (round 5)
// zedo_tile.h's dma16 is five instructions per 1-KB DMA: save M0, write M0, s_nop, global_load_lds_dwordx4, restore M0.  The instruction's
// 13-bit immediate offset is applied to BOTH the global address and the LDS address, and with k-block-major planes a wave's pieces of
// one k block are contiguous in memory AND in the ring slot: a burst of up to four can share one M0 (offsets 0 / 1024 / 2048 / 3072).
//   part 1  correctness of that reading of the offset (the data of piece p must land at LDS base + 1024 p)
//   part 2  a wave per SIMD (x WAVES) running the product's block shape - 24 fp16 MFMAs, then its six DMAs as a burst - with the burst
//           written (a) as six dma16, (b) as 4 + 2 sharing two M0 writes, (c) no DMA; cycles per block per wave
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ubench_dma_group.hip -o tools/ubench/ubench_dma_group
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
}
template <int N> __device__ __forceinline__ void dma16_group(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    static_assert(N >= 1 && N <= 4, "13-bit signed offset");
    if constexpr (N == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\tglobal_load_lds_dwordx4 %2, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %1 offset:2048\n\tglobal_load_lds_dwordx4 %2, %1 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
    else if constexpr (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\tglobal_load_lds_dwordx4 %2, %1 offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
    else dma16(sbase, voff, lds_byte_addr);
}

__global__ void check_kernel(const char *src, unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192 / 4; i += 64) reinterpret_cast<unsigned *>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    dma16_group<4>(src, (unsigned)lane * 16u, lds0 + 2048);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 8192 / 4; i += 64) out[i] = reinterpret_cast<unsigned *>(lds)[i];
}

template <int VAR, int WAVES>
__global__ __launch_bounds__(WAVES * 256) void blk_kernel(const char *gsrc, long long *rec, int nblk, unsigned window) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const char *src = gsrc + (size_t)blockIdx.x * (1 << 20);
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8 ha, hb;
    for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(lane * 1e-3f); hb[e] = (_Float16)(1.0f + blockIdx.x); }
    __syncthreads();
    const long long t0 = clock64();
    unsigned o = wid * 6144u;
    for (int b = 0; b < nblk; ++b) {
        const unsigned dst = lds0 + (wid * 12 + (b & 1) * 6) * 1024;
        if constexpr (VAR == 0) {
#pragma unroll
            for (int p = 0; p < 6; ++p) dma16(src + o + p * 1024, (unsigned)lane * 16u, dst + p * 1024);
        } else if constexpr (VAR == 1) {
            dma16_group<4>(src + o, (unsigned)lane * 16u, dst);
            dma16_group<2>(src + o + 4096, (unsigned)lane * 16u, dst + 4096);
        }
        o = (o + 6144u * 4 * WAVES) & (window - 1u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 24; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[u & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (VAR != 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    float sink = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) sink += acc[i][e];
    if (lane == 0) { rec[((size_t)blockIdx.x * 8 + wid) * 2] = t1 - t0; rec[((size_t)blockIdx.x * 8 + wid) * 2 + 1] = (long long)sink; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int VAR, int WAVES> static void run(const char *name, const char *gsrc, long long *rec, unsigned window) {
    const int nblk = 4096;
    auto kern = blk_kernel<VAR, WAVES>;
    const size_t LDS = (size_t)WAVES * 4 * 12 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 256), LDS, 0, gsrc, rec, nblk, window);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 256), LDS, 0, gsrc, rec, nblk, window);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(256 * 8 * 2);
    CK(hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES * 4; ++w) { cyc += (double)h[(b * 8 + w) * 2]; ++n; }
    printf("%-64s %d wave(s)/SIMD: %7.1f cycles per block per wave (768 = MFMA only), %.3f ms\n", name, WAVES, cyc / n / nblk, ms);
}

int main() {
    char *gsrc; long long *rec; unsigned *out;
    CK(hipMalloc(&gsrc, (size_t)258 << 20)); CK(hipMalloc(&rec, 256 * 8 * 2 * 8)); CK(hipMalloc(&out, 8192));
    std::vector<unsigned> hs(8192 / 4);
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = (unsigned)i;
    CK(hipMemset(gsrc, 0, (size_t)258 << 20));
    CK(hipMemcpy(gsrc, hs.data(), 8192, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 8192, 0, gsrc, out);
    std::vector<unsigned> ho(8192 / 4);
    CK(hipMemcpy(ho.data(), out, 8192, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 8192 / 4; ++i) {
        const bool inside = i >= 2048 / 4 && i < (2048 + 4096) / 4;
        const unsigned want = inside ? (unsigned)(i - 2048 / 4) : 0xdeadbeefu;
        if (ho[i] != want) { if (bad < 4) printf("  LDS word %d: got %08x want %08x\n", i, ho[i], want); ++bad; }
    }
    printf("part 1: one M0 write, four global_load_lds_dwordx4 with offset 0/1024/2048/3072: %s\n", bad ? "WRONG placement" : "pieces land at LDS base + offset, source + offset (OK)");
    for (int rep = 0; rep < 2; ++rep) {
        for (unsigned window : {1u << 16, 1u << 20}) {
            printf("source window per CU %u KB\n", window >> 10);
            run<2, 1>("no DMA", gsrc, rec, window);
            run<0, 1>("six dma16 (M0 saved / written / restored per DMA)", gsrc, rec, window);
            run<1, 1>("4 + 2 DMAs sharing two M0 writes", gsrc, rec, window);
            run<2, 2>("no DMA", gsrc, rec, window);
            run<0, 2>("six dma16 (M0 saved / written / restored per DMA)", gsrc, rec, window);
            run<1, 2>("4 + 2 DMAs sharing two M0 writes", gsrc, rec, window);
        }
    }
    return 0;
}
