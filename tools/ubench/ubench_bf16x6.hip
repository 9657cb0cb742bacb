// Feasibility micro-benchmark: fp32-accurate dense layer on the bf16 matrix pipe ("bf16x6").
//
// Every fp32 operand is split exactly into three bf16 pieces a = ah + am + al (+ O(2^-27 |a|)); the product block
// keeps the six piece products of order >= 2^-18 (hh, hm, mh, hl, lh, mm); the dropped ones are <= 2^-26
// relative - below the 2^-24 rounding of an fp32 product.  Accumulation is fp32 inside v_mfma_f32_32x32x16_bf16.
// Cost: 6 bf16 MFMAs (32 cycles each) per 32x32x16 block = 192 cycles, against 8 fp32 MFMAs (64 cycles each)
// = 512 cycles: 2.67x less matrix-pipe time, for 1.5x the operand bytes.
//
// Operand layout in HBM and LDS: P[row][k/16][3 planes][16] bf16 (96 contiguous bytes per row per 16-k block).
// This file only answers: is it accurate (vs fp64), and how fast is the main loop?  GN+SiLU epilogue, fp32 output.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ubench_bf16x6.hip -o tools/ubench/ubench_bf16x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Args {
    const uint16_t *X3;   // [M][K/16][3][16] bf16
    const uint16_t *W3;   // [N][K/16][3][16] bf16
    const float *bias, *gamma, *beta;
    float *out;           // [M][N] fp32
    int M, N, K;
};

__device__ __forceinline__ void dma16(const char *sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(sbase), "v"(voff), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ float silu_fast(float y) {
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.44269504088896340736f));
}

// BM x BN tile, WM x WN waves, NBUF-deep ring of 16-k blocks.  i = channel (W rows), j = batch row (X rows).
// SWZ = 1: chunk (plane, k half) of tile row r is stored at k half ^ ((r >> 3) & 1): rows r and r + 8 are 768 B = 3 bank rows apart, so without it
// every 16-lane group of a ds_read_b128 hits each 16-byte slot twice (2-way conflict, LDS reads at half rate)
template <int BM, int BN, int WM, int WN, int NBUF, int NODMA = 0, int SWZ = 0, int WPE = 1>
__global__ __launch_bounds__(WM *WN * 64, WPE) void bf_kernel(Args a) {
    constexpr int NW = WM * WN, TM = BM / WM, TN = BN / WN, TJ = TM / 32, TI = TN / 32;
    constexpr int RB = 96;                                   // bytes per row per k-block
    constexpr int IA = BN * 6 / 64 / NW, IB = BM * 6 / 64 / NW;   // DMA instructions per wave per k-block
    static_assert((BN * 6) % (64 * NW) == 0 && (BM * 6) % (64 * NW) == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ring slot = [BN rows x 96 B][BM rows x 96 B]
    constexpr int SLOT = (BN + BM) * RB;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int ncol = a.N / BN;
    const int m0 = (lid / ncol) * BM, n0 = (lid % ncol) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN, li = lane & 31, kh = lane >> 5;
    const size_t rstride = (size_t)(a.K / 16) * RB;          // bytes between rows in HBM
    const char *Wbase = reinterpret_cast<const char *>(a.W3) + (size_t)n0 * rstride;
    const char *Xbase = reinterpret_cast<const char *>(a.X3) + (size_t)m0 * rstride;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    // per-lane source offsets: DMA instruction p of this wave moves 16-byte chunks g = (wid*I + p)*64 + lane of the tile
    unsigned woff[IA], xoff[IB];
#pragma unroll
    for (int p = 0; p < IA; ++p) { const int g = (wid * IA + p) * 64 + lane; const int r_ = g / 6, c_ = (g % 6) ^ (SWZ ? ((r_ >> 3) & 1) : 0); woff[p] = (unsigned)(r_ * rstride + c_ * 16); }
#pragma unroll
    for (int p = 0; p < IB; ++p) { const int g = (wid * IB + p) * 64 + lane; const int r_ = g / 6, c_ = (g % 6) ^ (SWZ ? ((r_ >> 3) & 1) : 0); xoff[p] = (unsigned)(r_ * rstride + c_ * 16); }
    auto dma = [&](int kb, int slot) {
        const char *wk = Wbase + (size_t)kb * RB, *xk = Xbase + (size_t)kb * RB;
#pragma unroll
        for (int p = 0; p < IA; ++p) dma16(wk, woff[p], lds0 + slot * SLOT + (wid * IA + p) * 1024);
#pragma unroll
        for (int p = 0; p < IB; ++p) dma16(xk, xoff[p], lds0 + slot * SLOT + BN * RB + (wid * IB + p) * 1024);
    };
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int KB = a.K / 16;
    constexpr int IPW = IA + IB;
    // fragment groups of one 16-k block: G1 = {A.h, A.m, B.h, B.m} (feeds mm, hm, mh, hh), G2 = {A.l, B.l} (feeds hl, lh)
    bf16x8 a1[2][TI][2], b1[2][TJ][2], a2[TI], b2[TJ];
    const int khs = SWZ ? (kh ^ ((li >> 3) & 1)) : kh;    // tile bases are multiples of 32 rows: (row >> 3) & 1 == (li >> 3) & 1
    auto readG1 = [&](int set, int slot) {
        const char *As = smem + slot * SLOT + (wn * TN + li) * RB + khs * 16;
        const char *Bs = smem + slot * SLOT + BN * RB + (wm * TM + li) * RB + khs * 16;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a1[set][i][pl] = *reinterpret_cast<const bf16x8 *>(As + i * 32 * RB + pl * 32);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) b1[set][j][pl] = *reinterpret_cast<const bf16x8 *>(Bs + j * 32 * RB + pl * 32);
    };
    auto readG2 = [&](int slot) {
        const char *As = smem + slot * SLOT + (wn * TN + li) * RB + khs * 16 + 64;
        const char *Bs = smem + slot * SLOT + BN * RB + (wm * TM + li) * RB + khs * 16 + 64;
#pragma unroll
        for (int i = 0; i < TI; ++i) a2[i] = *reinterpret_cast<const bf16x8 *>(As + i * 32 * RB);
#pragma unroll
        for (int j = 0; j < TJ; ++j) b2[j] = *reinterpret_cast<const bf16x8 *>(Bs + j * 32 * RB);
    };
    // product-major order: consecutive MFMAs hit different accumulators (no back-to-back dependent issue)
#define FOR_TILES(stmt) _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) { stmt; }
    auto mmaM1 = [&](int set) {
        FOR_TILES(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[set][i][1], b1[set][j][1], acc[i][j], 0, 0, 0))   // mm
        FOR_TILES(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[set][i][0], b1[set][j][1], acc[i][j], 0, 0, 0))   // hm
        FOR_TILES(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[set][i][1], b1[set][j][0], acc[i][j], 0, 0, 0))   // mh
        FOR_TILES(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[set][i][0], b1[set][j][0], acc[i][j], 0, 0, 0))   // hh
    };
    auto mmaM2 = [&](int set) {
        FOR_TILES(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[set][i][0], b2[j], acc[i][j], 0, 0, 0))   // hl
        FOR_TILES(acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b1[set][j][0], acc[i][j], 0, 0, 0))   // lh
    };
    // block kb in ring slot kb % NBUF.  iteration: [G1(kb) in set kb&1]
    //    read G2(kb); M1(kb); vmcnt; barrier (block kb fully read by all, block kb+1 landed);
    //    DMA(block kb+NBUF -> slot of kb); read G1(kb+1) into the other set; M2(kb)
    static_assert(NBUF % 2 == 0, "ring depth even (fragment set = kb & 1 must be static)");
#pragma unroll
    for (int t = 0; t < NBUF; ++t) dma(min(t, KB - 1), t);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * IPW) : "memory");
    __syncthreads();
    readG1(0, 0);
    for (int kb0 = 0; kb0 < KB; kb0 += NBUF) {
#pragma unroll
        for (int slot = 0; slot < NBUF; ++slot) {
            const int kb = kb0 + slot;
            constexpr int dummy = 0; (void)dummy;
            const int set = slot & 1, nxt = (slot + 1) % NBUF;
            readG2(slot);
            mmaM1(set);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * IPW) : "memory");
            __syncthreads();
            if (!NODMA) dma(min(kb + NBUF, KB - 1), slot);
            readG1(set ^ 1, nxt);
            mmaM2(set);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // epilogue (direct stores; only the main loop is under test here)
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int cbase = n0 + wn * TN + i * 32 + 4 * kh;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int m = m0 + wm * TM + j * 32 + li;
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * g + e] = acc[i][j][4 * g + e] + a.bias[cbase + 8 * g + e];
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += v[e];
            s += __shfl_xor(s, 32);
            const float mean = s * (1.f / 32.f);
            float qs = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { v[e] -= mean; qs += v[e] * v[e]; }
            qs += __shfl_xor(qs, 32);
            const float rstd = __builtin_amdgcn_rsqf(qs * (1.f / 32.f) + 1e-5f);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = silu_fast(v[4 * g + e] * (rstd * a.gamma[cbase + 8 * g + e]) + a.beta[cbase + 8 * g + e]);
                *reinterpret_cast<f32x4 *>(a.out + (size_t)m * a.N + cbase + 8 * g) = o;
            }
        }
    }
}

static inline uint16_t bf16_rn(float f) {   // round to nearest even
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(const std::vector<float> &src, int rows, int K, std::vector<uint16_t> &dst) {
    dst.resize((size_t)rows * K * 3);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) {
            const float a = src[(size_t)r * K + k];
            const uint16_t h = bf16_rn(a); const float r1 = a - bf16_to_f(h);
            const uint16_t m = bf16_rn(r1); const float r2 = r1 - bf16_to_f(m);
            const uint16_t l = bf16_rn(r2);
            uint16_t *d = &dst[((size_t)r * (K / 16) + k / 16) * 48 + (k % 16)];
            d[0] = h; d[16] = m; d[32] = l;
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int BM, int BN, int WM, int WN, int NBUF, int NODMA = 0, int SWZ = 0, int WPE = 1>
int run(const char *name, Args a, const std::vector<int> &rows, const std::vector<double> &cref) {
    constexpr size_t lds = (size_t)NBUF * (BM + BN) * 96;
    auto kern = bf_kernel<BM, BN, WM, WN, NBUF, NODMA, SWZ, WPE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nwg = (a.M / BM) * (a.N / BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(a.out, 0xff, (size_t)a.M * a.N * 4));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), lds, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> y((size_t)a.M * a.N);
    CK(hipMemcpy(y.data(), a.out, y.size() * 4, hipMemcpyDeviceToHost));
    double maxd = 0; int nbad = 0;
    for (size_t ri = 0; ri < rows.size(); ++ri) {
        double d = 0;
        for (int n = 0; n < a.N; ++n) { double dd = fabs((double)y[(size_t)rows[ri] * a.N + n] - cref[ri * a.N + n]); if (!(dd <= d)) d = dd; }
        if (!(d <= 1e-4)) ++nbad;
        if (!(d <= maxd)) maxd = d;
    }
    for (int r = 0; r < 300; ++r) hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), lds, 0, a);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(kern, dim3(nwg), dim3(WM * WN * 64), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 100;
    printf("%-34s: %7.1f us  %6.1f TF(fp32-equivalent)  max|y-ref64| %.1e  bad rows %d  [lds %zu KB]\n", name, ms * 1e3,
           2.0 * a.M * a.N * a.K / ms / 1e9, maxd, nbad, lds / 1024);
    return 0;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 49152, N = 1024, K = 1024;
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N), hg(N), hbe(N);
    std::mt19937 rng(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (auto &v : hx) v = u(rng);
    for (auto &v : hw) v = u(rng) * 0.03f;
    for (auto &v : hb) v = u(rng);
    for (auto &v : hg) v = 1.0f + 0.5f * u(rng);
    for (auto &v : hbe) v = 0.2f * u(rng);
    std::vector<uint16_t> x3, w3; split3(hx, M, K, x3); split3(hw, N, K, w3);
    std::vector<int> rows; for (int r = 0; r < M; r += 997) rows.push_back(r);
    std::vector<double> cref(rows.size() * (size_t)N);
    for (size_t ri = 0; ri < rows.size(); ++ri) {
        const float *xr = &hx[(size_t)rows[ri] * K];
        std::vector<double> v(N);
        for (int n = 0; n < N; ++n) { double s = hb[n]; const float *wr = &hw[(size_t)n * K]; for (int k = 0; k < K; ++k) s += (double)xr[k] * wr[k]; v[n] = s; }
        for (int g = 0; g < N / 32; ++g) {
            double mu = 0, var = 0;
            for (int c = 0; c < 32; ++c) mu += v[g * 32 + c];
            mu /= 32;
            for (int c = 0; c < 32; ++c) var += (v[g * 32 + c] - mu) * (v[g * 32 + c] - mu);
            var /= 32;
            for (int c = 0; c < 32; ++c) { double y = (v[g * 32 + c] - mu) / sqrt(var + 1e-5) * hg[g * 32 + c] + hbe[g * 32 + c]; cref[ri * N + g * 32 + c] = y / (1 + exp(-y)); }
        }
    }
    Args a{};
    uint16_t *dx, *dw; float *db, *dg, *dbe, *dy;
    CK(hipMalloc(&dx, x3.size() * 2)); CK(hipMalloc(&dw, w3.size() * 2)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dg, N * 4));
    CK(hipMalloc(&dbe, N * 4)); CK(hipMalloc(&dy, (size_t)M * N * 4));
    CK(hipMemcpy(dx, x3.data(), x3.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w3.data(), w3.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, hg.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbe, hbe.data(), N * 4, hipMemcpyHostToDevice));
    a.X3 = dx; a.W3 = dw; a.bias = db; a.gamma = dg; a.beta = dbe; a.out = dy; a.M = M; a.N = N; a.K = K;
    run<128, 128, 2, 2, 2>("128x128 4 waves ring2", a, rows, cref);
    run<128, 128, 2, 2, 2, 0, 1>("128x128 4 waves ring2 SWZ", a, rows, cref);
    run<128, 128, 2, 2, 2, 0, 1, 3>("128x128 4 waves ring2 SWZ lb3", a, rows, cref);
    run<128, 128, 2, 2, 2, 1, 1>("128x128 ring2 SWZ NO in-loop DMA", a, rows, cref);
    run<256, 256, 4, 2, 2, 0, 1>("256x256 8 waves ring2 SWZ", a, rows, cref);
    run<256, 256, 4, 2, 2, 1, 1>("256x256 ring2 SWZ NO in-loop DMA", a, rows, cref);
    run<256, 128, 2, 2, 2, 0, 1>("256x128 4 waves ring2 SWZ", a, rows, cref);
    run<128, 128, 2, 2, 4>("128x128 4 waves ring4", a, rows, cref);
    run<256, 256, 4, 2, 2>("256x256 8 waves ring2", a, rows, cref);
    run<128, 128, 2, 2, 2, 1>("128x128 ring2 NO in-loop DMA", a, rows, cref);
    run<256, 256, 4, 2, 2, 1>("256x256 ring2 NO in-loop DMA", a, rows, cref);
    run<256, 128, 2, 2, 2>("256x128 4 waves ring2", a, rows, cref);
    run<256, 128, 2, 2, 4>("256x128 4 waves ring4", a, rows, cref);
    return 0;
}
