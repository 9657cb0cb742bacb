// Compile-time evidence for the closing note of the split-fp16 hidden layer (profiles/f16x3_closure_r06.txt).
// The design asked about: ONE compute wave per SIMD holding TWO accumulator sets of its 64 x 128 wave tile (2 x 8 x 16 = 256 registers) so
// that the GroupNorm / SiLU / split epilogue of tile t can issue in the MFMA shadows of tile t + 1 in the same wave, with the LDS-DMAs
// owned by small loader waves.  This kernel is the register skeleton of that wave - two live accumulator sets, the double-buffered
// fragments of the product's k loop (zedo_gemm16.hip: fa[2][4][2] + fb[2][2][2] f16x8 = 96 registers), an epilogue slice working on the
// other set - and nothing else.  Build and read what the compiler allocates:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/ubench/ubench_agpr_regs.hip -o /tmp/agpr.o -save-temps=obj ; grep -E "NumVgprs|NumAgprs|TotalNumVgprs|Occupancy" /tmp/*gfx950.s
// Result (ROCm 7.2): TotalNumVgprs far above 256 -> Occupancy 1.  VGPRs are allocated per KERNEL (one granulated count in the kernel
// descriptor, the same for every wave of the dispatch): with a 512-register file per SIMD lane there is no room for a second wave on the
// SIMD - not even a 24-register loader wave, which would be allocated the same count.  The compute wave would have to issue its own six
// LDS-DMAs per k block (~650 cycles beside 768 of MFMA, profiles/vmem_issue_r05.txt) with nobody to cover them: slower than the
// product's two waves per SIMD.  Not built further.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 1) void agpr_skeleton(const f16x8 *__restrict__ frag, float *__restrict__ out, int tiles, int kblocks) {
    constexpr int TI = 4, TJ = 2;
    f32x16 acc[2][TI][TJ];
    for (int s = 0; s < 2; ++s) for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int e = 0; e < 16; ++e) acc[s][i][j][e] = 0.f;
    f16x8 fa[2][TI][2], fb[2][TJ][2];
    const f16x8 *p = frag + threadIdx.x;
    float epi = 0.f;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {       // tile t accumulates into set `half`, the epilogue slices read set 1 - half
            for (int kb0 = 0; kb0 < kblocks; kb0 += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {        // unrolled: fragment sets and the epilogue slice's accumulator are compile-time indices
                    const int kb = kb0 + u, set = u & 1;
#pragma unroll
                    for (int i = 0; i < TI; ++i) for (int pl = 0; pl < 2; ++pl) fa[set ^ 1][i][pl] = p[(kb * 12 + i * 2 + pl) * 256];
#pragma unroll
                    for (int j = 0; j < TJ; ++j) for (int pl = 0; pl < 2; ++pl) fb[set ^ 1][j][pl] = p[(kb * 12 + 8 + j * 2 + pl) * 256];
#pragma unroll
                    for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                        for (int i = 0; i < TI; ++i)
#pragma unroll
                            for (int j = 0; j < TJ; ++j)
                                acc[half][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i][prod == 0], fb[set][j][prod == 1], acc[half][i][j], 0, 0, 0);
                    // one epilogue slice of the OTHER set per k block (stand-in arithmetic: every element of that set stays live)
#pragma unroll
                    for (int e = 0; e < 16; ++e) epi = __builtin_fmaf(acc[1 - half][u & 3][u >> 2][e], 1.0001f, epi);
                }
            }
        }
    }
    float s_ = epi;
    for (int s = 0; s < 2; ++s) for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int e = 0; e < 16; ++e) s_ += acc[s][i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s_;
}
int main() { return 0; }
