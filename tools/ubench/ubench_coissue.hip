// What does one co-resident instruction cost a stream of v_mfma_f32_32x32x2_f32 on the same SIMD (gfx950)?
// Block = 512 threads: waves 0-3 (one per SIMD) run NM MFMAs; waves 4-7 run NF filler instructions of one kind.
// price = (t_both - t_mfma_alone) / NF   in SIMD cycles per filler wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_coissue.hip -o tools/ubench/ubench_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifndef MFMA_F16
#define MFMA_F16 0      // 1: the MFMA stream is v_mfma_f32_32x32x16_f16 (32 cycles each) instead of v_mfma_f32_32x32x2_f32 (64)
#endif

template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, const float *gin, int nm, int nf, int with_mfma) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wid < 4) {
        if (!with_mfma) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        float a = lane * 1e-3f, b = 1.0f + blockIdx.x;
        for (int it = 0; it < nm / 16; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#if MFMA_F16
                    f16x8 ha, hb; for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)a; hb[e] = (_Float16)b; }
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[i], 0, 0, 0);
#else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#endif
                }
        }
        float s = 0; for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        float x0 = lane * 0.001f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
        f32x4 v = {x0, x1, x2, x3};
        const float *gp = gin + (size_t)(blockIdx.x * 256 + (threadIdx.x - 256)) * 4;
        float *lp = lds + (threadIdx.x - 256) * 4;
        for (int it = 0; it < nf / 16; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if constexpr (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2)); }
                else if constexpr (KIND == 1) { asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); }
                else if constexpr (KIND == 2) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)lp)); }
                else if constexpr (KIND == 3) { asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)lp), "v"(v)); }
                else if constexpr (KIND == 4) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(gp)); }
                else if constexpr (KIND == 5) { asm volatile("v_mov_b32 %0, %1" : "=v"(x0) : "v"(x1)); }
                else if constexpr (KIND == 6) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1)); }
                else if constexpr (KIND == 7) { asm volatile("s_nop 0"); }
                else if constexpr (KIND == 8) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&v) : "v"(*((double*)&v + 1)), "v"(*((double*)&v + 1))); }
                else if constexpr (KIND == 9) { asm volatile("v_rcp_f32 %0, %0" : "+v"(x0)); }
                else if constexpr (KIND == 10) {   // LDS-DMA: global -> LDS without a VGPR destination
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gp,
                                                     (void __attribute__((address_space(3))) *)(lds + (wid - 4) * 256 * (u & 3)), 16, 0, 0);
                }
                else if constexpr (KIND == 11) { asm volatile("global_load_dword %0, %1, off" : "=v"(x3) : "v"(gp)); }
                else if constexpr (KIND == 12) {   // 4 independent VALU chains
                    if ((u & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1));
                    else if ((u & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(x1));
                    else if ((u & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(x1));
                    else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[3]) : "v"(x1));
                }
                else if constexpr (KIND == 13) {   // one 16-byte load per 16 slots (low VMEM rate)
                    if (u == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(gp)); else asm volatile("s_nop 7");
                }
                else if constexpr (KIND == 15) { asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(gp), "v"(v) : "memory"); }
                else if constexpr (KIND == 17) { asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(gp), "v"(v) : "memory"); }
                else if constexpr (KIND == 18) { asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(gp), "v"(*(double*)&v) : "memory"); }
                else if constexpr (KIND == 19) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(gp), "v"(v) : "memory"); }
                else if constexpr (KIND == 20) { asm volatile("global_store_dword %0, %1, off" :: "v"(gp), "v"(x1) : "memory"); }
                else if constexpr (KIND == 16) { asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x0)); }
                else if constexpr (KIND == 14) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(*(double*)&v) : "v"(gp)); }
            }
            if constexpr (KIND == 2 || KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (KIND == 4 || KIND == 10 || KIND == 11 || KIND == 13 || KIND == 14 || KIND == 15 || KIND == 17 || KIND == 18 || KIND == 19 || KIND == 20) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        out[blockIdx.x * 512 + threadIdx.x] = x0 + v[0] + v[1];
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int KIND> int run(const char *name, float *out, float *gin) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nm = 1 << 16;   // 65536 MFMAs = 4.19M cycles per SIMD
    auto t = [&](int nf, int with) -> float {
        for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, gin, nm, nf, with);
        hipEventRecord(e0);
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, gin, nm, nf, with);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
    };
    const float alone = t(0, 1);
    for (int nf : {1 << 14, 1 << 16}) {
        const float both = t(nf, 1), fill = t(nf, 0);
        printf("%-22s nf=%7d  mfma alone %.3f ms | filler alone %.3f ms | both %.3f ms | price %.2f cyc/filler (at 2.4 GHz) | filler alone %.1f cyc each\n", name, nf, alone,
               fill, both, (both - alone) * 1e-3 * 2.4e9 / nf, fill * 1e-3 * 2.4e9 / nf);
    }
    return 0;
}
int main() {
    printf("MFMA stream: %s\n", MFMA_F16 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_32x32x2_f32");
    float *out, *gin; CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&gin, 1 << 22)); CK(hipMemset(gin, 0, 1 << 22));
    run<12>("v_fma_f32 x4 indep", out, gin);
#if MFMA_F16
    run<0>("v_fma_f32 dependent", out, gin); run<1>("v_exp_f32", out, gin); run<9>("v_rcp_f32", out, gin); run<8>("v_pk_fma_f32", out, gin);
    run<16>("v_cvt_f16_f32", out, gin); run<2>("ds_read_b128", out, gin); run<3>("ds_write_b128", out, gin); run<15>("global_store_dwordx4", out, gin);
    run<17>("global_store_dwordx4 nt", out, gin); run<19>("global_store_dwordx4 sc0 sc1", out, gin); run<18>("global_store_dwordx2", out, gin); run<20>("global_store_dword", out, gin);
#endif
    run<4>("global_load_dwordx4", out, gin); run<14>("global_load_dwordx2", out, gin); run<11>("global_load_dword", out, gin);
    run<10>("global_load_lds_dwordx4", out, gin); run<13>("1 dwordx4 per 16 slots", out, gin);
    return 0;
}
