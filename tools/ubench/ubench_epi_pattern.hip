// What is the lane -> address map of the split-fp16 epilogue worth?  (round 5, after the planes became k-block-major)
// The hidden layers of zedo_gemm16.hip write their output (and read the residual) as fp16 planes [N/16][rows][64 B]: a 128 x 256 tile
// is 16 channel groups (3.2 MB apart at 50 944 rows) x 128 rows x 64 B.  Per phase a 256-thread workgroup moves 64 rows x 16 groups
// x 64 B = 4096 16-byte chunks = 16 wave instructions per wave.  This harness replays ONLY that traffic (no MFMA, no LDS) with three
// lane maps:
//   A  product (round 5):  lane -> (group = l % 16, row = l / 16); instruction q moves bytes 16 q .. 16 q + 15 of the lane's 64-byte row piece:
//                          every instruction touches 64 separate 16-byte pieces, four instructions fill 16 pieces of 256 bytes
//   B  quad-dense:         lane -> (q = l & 3, group = (l >> 2) & 3, row = l >> 4): every instruction fills 16 whole 64-byte row pieces
//   C  dense:              lane -> (q = l & 3, row = l >> 2), one group per instruction: every instruction fills 1 KB of contiguous memory
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ubench_epi_pattern.hip -o tools/ubench/ubench_epi_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int MODE>   // MODE 1: stores, 2: loads, 3: loads + stores (residual layer)
__global__ __launch_bounds__(256, 2) void epi_kernel(char *out, const char *res, int ld, int ntile_n, float *sink) {
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int mt = blockIdx.x / ntile_n, nt = blockIdx.x % ntile_n;
    const size_t grp = (size_t)ld * 64;
    const size_t base = ((size_t)(nt * 16) * ld + (size_t)mt * 128) * 64;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4 val = {1.f * tid, 2.f, 3.f, 4.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x4 r[16];
        size_t off[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            int sr, cg, q;
            if constexpr (PAT == 0) { const int qi = (n >> 2) * 256 + tid; sr = qi / 16; cg = qi % 16; q = n & 3; }
            else if constexpr (PAT == 1) { q = l & 3; cg = ((l >> 2) & 3) + 4 * (n & 3); sr = (l >> 4) + 4 * w + 16 * (n >> 2); }
            else { q = l & 3; cg = n; sr = (l >> 2) + 16 * w; }
            const int grow = (sr >> 5) * 64 + j * 32 + (sr & 31);
            off[n] = base + (size_t)cg * grp + (size_t)grow * 64 + 16 * q;
        }
        if constexpr (MODE & 2) {
#pragma unroll
            for (int n = 0; n < 16; ++n) r[n] = *reinterpret_cast<const f32x4 *>(res + off[n]);
#pragma unroll
            for (int n = 0; n < 16; ++n) acc += r[n];
        }
        if constexpr (MODE & 1) {
#pragma unroll
            for (int n = 0; n < 16; ++n) *reinterpret_cast<f32x4 *>(out + off[n]) = (MODE & 2) ? r[n] + val : val;
        }
    }
    if (MODE == 2 && acc[0] == 12345.678f) sink[0] = acc[1];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int PAT, int MODE> static int run(const char *name, char *out, char *res, int rows, float *sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = (rows / 128) * 4;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((epi_kernel<PAT, MODE>), dim3(blocks), dim3(256), 0, 0, out, res, rows, 4, sink);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((epi_kernel<PAT, MODE>), dim3(blocks), dim3(256), 0, 0, out, res, rows, 4, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
    const double bytes = (double)rows * 1024 * 4 * ((MODE & 1) + ((MODE >> 1) & 1));
    printf("%-58s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9);
    return 0;
}

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 50944;     // multiple of 128
    char *out, *res; float *sink;
    const size_t bytes = (size_t)rows * 1024 * 4;
    CK(hipMalloc(&out, bytes)); CK(hipMalloc(&res, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(out, 0, bytes)); CK(hipMemset(res, 0, bytes));
    printf("epilogue traffic of one split-fp16 hidden layer alone, %d rows x 1024 channels as planes (%.0f MB per direction)\n", rows, bytes / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 1>("A product map, stores", out, res, rows, sink);
        run<1, 1>("B quad-dense (64-byte row pieces), stores", out, res, rows, sink);
        run<2, 1>("C dense (1 KB per instruction), stores", out, res, rows, sink);
        run<0, 2>("A product map, loads", out, res, rows, sink);
        run<1, 2>("B quad-dense, loads", out, res, rows, sink);
        run<2, 2>("C dense, loads", out, res, rows, sink);
        run<0, 3>("A product map, residual load + store", out, res, rows, sink);
        run<1, 3>("B quad-dense, residual load + store", out, res, rows, sink);
        run<2, 3>("C dense, residual load + store", out, res, rows, sink);
    }
    return 0;
}
