// Does the operand LAYOUT limit the LDS-DMA stream of the f16x3 hidden layer?  (round 5)
// Four loader waves per CU replay exactly the DMA stream of the 128 x 256 tiles of zedo_gemm16.hip (6 tiles per CU, 64 k blocks of
// 24 one-KB instructions: 16 W + 8 X row groups; the four column tiles of a row tile on one XCD) and nothing else, from
//   layout 0  [row][K/16][64 bytes]   - the product's "planes" rows: 4 KB between the rows of a k block (each instruction = 16 rows x 64 B)
//   layout 1  [K/16][row][64 bytes]   - k-block-major: the rows of a k block are contiguous (each instruction = 1 KB contiguous)
// -> microseconds for the whole stream = the floor the memory system sets for the layer, whatever the MFMA side does.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ubench_dma_layout.hip -o tools/ubench/ubench_dma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void dma16(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
}
template <int LAYOUT, int INFLIGHT>
__global__ __launch_bounds__(256) void k(const char *W, const char *X, int Mp, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const int G = gridDim.x, bid = blockIdx.x;
    const int q8 = G >> 3, r8 = G & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int N = 1024;
    for (int t = lid; t < ntiles; t += G) {
        const int m0 = (t / 4) * 128, n0 = (t % 4) * 256;
        for (int kb = 0; kb < 64; ++kb) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int p = w * 6 + i;                       // 0..15 W row groups, 16..23 X row groups
                const bool isw = p < 16;
                const int row = (isw ? n0 + p * 16 : m0 + (p - 16) * 16) + (lane >> 2);
                const char *base = isw ? W : X;
                const size_t rows_total = isw ? (size_t)N : (size_t)Mp;
                const size_t off = LAYOUT == 0 ? (size_t)row * 4096 + (size_t)kb * 64 + (lane & 3) * 16
                                               : ((size_t)kb * rows_total + row) * 64 + (lane & 3) * 16;
                const unsigned long long u = (unsigned long long)base;
                dma16(reinterpret_cast<const char *>(u), (unsigned)off, __builtin_amdgcn_readfirstlane(lds0 + ((kb & 3) * 24 + p) * 1024));
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int LAYOUT, int INFLIGHT> void run(const char *W, const char *X, int Mp, const char *name) {
    auto kern = k<LAYOUT, INFLIGHT>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    const int ntiles = (Mp / 128) * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 98304, 0, W, X, Mp, ntiles);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 98304, 0, W, X, Mp, ntiles);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    const double bytes = (double)ntiles * 64 * 24 * 1024;
    printf("%-64s %7.1f us   %.2f TB/s through LDS-DMA\n", name, ms * 1e3, bytes / ms / 1e9);
}
int main(int argc, char **argv) {
    const int Mp = argc > 1 ? atoi(argv[1]) : 49152;
    char *W, *X;
    CK(hipMalloc(&W, (size_t)1024 * 4096)); CK(hipMalloc(&X, (size_t)Mp * 4096));   // 4-byte X is < 4 GB offsets: Mp <= 1M rows
    CK(hipMemset(W, 0, (size_t)1024 * 4096)); CK(hipMemset(X, 0, (size_t)Mp * 4096));
    printf("DMA stream of one 1024 x 1024 hidden layer over %d rows (128 x 256 tiles, four loader waves per CU, nothing else running)\n", Mp);
    run<0, 12>(W, X, Mp, "layout [row][k/16][64 B] (product), 12 in flight per wave");
    run<0, 24>(W, X, Mp, "layout [row][k/16][64 B] (product), 24 in flight per wave");
    run<0, 48>(W, X, Mp, "layout [row][k/16][64 B] (product), 48 in flight per wave");
    run<1, 12>(W, X, Mp, "layout [k/16][row][64 B] (k-block-major), 12 in flight per wave");
    run<1, 24>(W, X, Mp, "layout [k/16][row][64 B] (k-block-major), 24 in flight per wave");
    run<1, 48>(W, X, Mp, "layout [k/16][row][64 B] (k-block-major), 48 in flight per wave");
    return 0;
}
