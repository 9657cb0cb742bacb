// Standalone micro-benchmark (no torch): sustained fp32-MFMA peak / shader clock on this box, and the
// dense-layer kernel variants of zedo_gemm.hip at the BASELINE shape.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZEDO_UBENCH tools/ubench/ubench_gemm.hip -o tools/ubench/ubench_gemm
#include "../../zedo-release_amd/csrc/zedo_gemm.hip"
#include <cstdio>
#include <chrono>
#include <vector>
#include <random>
#include <string>
#include <cstring>

using namespace zedo;

// the split-K reduce of post_dense lives in zedo_geom.hip; this harness never reaches it (LayerArgs::scratch stays null)
namespace zedo {
hipError_t launch_post_reduce(float *, const float *, const float *, float, float, int, float *, const float *, float *, int, int, int, int, long long, hipStream_t) {
    return hipErrorNotSupported;
}
}

__global__ __launch_bounds__(256) void mfma_peak_kernel(float *out, int iters, long long *clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.0f;
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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    int M = argc > 1 ? atoi(argv[1]) : 50944;
    const int N = 1024, K = 1024;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *dout; long long *dclk; CK(hipMalloc(&dout, 1 << 22)); CK(hipMalloc(&dclk, 16));
    for (int iters : {2000, 20000, 200000}) {
        const int blocks = 256 * 2;   // 2 blocks of 4 waves per CU -> 2 waves per SIMD
        hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, dout, 100, dclk);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, dout, iters, dclk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long clk[2]; CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
        double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
        printf("mfma peak: iters %7d  %8.3f ms  %7.1f TF   shader clock %.3f GHz (clock64/wall_clock64@100MHz)\n", iters, ms,
               flop / ms / 1e9, (double)clk[0] / ((double)clk[1] / 100e6) / 1e9);
    }
    // ---- GEMM variants
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N), hg(N, 1.f), hbe(N, 0.f);
    std::mt19937 rng(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (auto &v : hx) v = u(rng);
    for (auto &v : hw) v = u(rng) * 0.03f;
    for (auto &v : hb) v = u(rng);
    float *dx, *dw, *db, *dg, *dbe, *dy;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dg, N * 4));
    CK(hipMalloc(&dbe, N * 4)); CK(hipMalloc(&dy, hx.size() * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, hg.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbe, hbe.data(), N * 4, hipMemcpyHostToDevice));
    LayerArgs a{}; a.X = dx; a.ldx = K; a.W = dw; a.ldw = K; a.bias = db; a.gamma = dg; a.beta = dbe; a.out = dy; a.ldo = N; a.K = K; a.N = N; a.Mp = M;
    if (argc > 2 && atoi(argv[2]) >= 0) {   // timeline dump: ubench_gemm M variant out.bin
        const int var = atoi(argv[2]);
        long long *dtl; const size_t NWG = (size_t)(M / 64) * 8; CK(hipMalloc(&dtl, NWG * 64)); CK(hipMemset(dtl, 0, NWG * 64));
        for (int r = 0; r < 300; ++r) launch_variant(a, var, 0);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &dtl, sizeof(dtl)));
        launch_variant(a, var, 0);
        CK(hipDeviceSynchronize());
        std::vector<long long> tl(NWG * 8);
        CK(hipMemcpy(tl.data(), dtl, tl.size() * 8, hipMemcpyDeviceToHost));
        FILE *f = fopen(argc > 3 ? argv[3] : "gpurun_out/wg_timeline.bin", "wb");
        fwrite(tl.data(), 8, tl.size(), f); fclose(f);
        printf("timeline of variant %d written\n", var);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == -1) {   // overlap test: hidden layer on stream 1, pre_dense-like layer on stream 2
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        float *dx2, *dy2; CK(hipMalloc(&dx2, (size_t)M * 64 * 4)); CK(hipMalloc(&dy2, (size_t)M * N * 4));
        CK(hipMemcpy(dx2, hx.data(), (size_t)M * 64 * 4, hipMemcpyHostToDevice));
        LayerArgs b = a; b.X = dx2; b.ldx = 64; b.ldw = K; b.K = 64; b.out = dy2;
        auto timeit = [&](int mode) -> float {
            for (int r = 0; r < 200; ++r) { if (mode & 1) launch_layer(a, EPI_GN_SILU, s1); if (mode & 2) launch_layer(b, EPI_GN_SILU, s2); }
            hipDeviceSynchronize();
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int r = 0; r < 100; ++r) { if (mode & 1) launch_layer(a, EPI_GN_SILU, s1); if (mode & 2) launch_layer(b, EPI_GN_SILU, s2); }
            hipDeviceSynchronize();
            return std::chrono::duration<float, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 100;
        };
        const float th = timeit(1), tl = timeit(2), tb = timeit(3);
        printf("rows %d: hidden alone %.1f us, light alone %.1f us, both on two streams %.1f us (sum %.1f)\n", M, th, tl, tb, th + tl);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == -2) {   // two independent half-batch chains on two streams vs one full-batch chain
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        const int half = (M / 2 / 256) * 256;
        LayerArgs lo = a, hi = a; lo.Mp = half; hi.Mp = M - half; hi.X = a.X + (size_t)half * K; hi.out = a.out + (size_t)half * N;
        auto run2 = [&](int mode) -> float {
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) hipDeviceSynchronize();
                auto t0 = std::chrono::high_resolution_clock::now();
                for (int r = 0; r < 200; ++r) {
                    const int epi = (r & 1) ? EPI_GN_SILU_RES : EPI_GN_SILU;
                    if (mode == 0) launch_layer(a, epi, s1);
                    else { launch_layer(lo, epi, s1); launch_layer(hi, epi, s2); }
                }
                hipDeviceSynchronize();
                if (rep == 1) return std::chrono::duration<float, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 200;
            }
            return 0.f;
        };
        const float t1 = run2(0), t2 = run2(1), t1b = run2(0), t2b = run2(1);
        printf("rows %d: one chain %.1f / %.1f us per layer; two half-batch chains on two streams %.1f / %.1f us\n", M, t1, t1b, t2, t2b);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == -3) {   // thin layers: where does the time of pre_dense / post_dense go?  (ablations)
        float *dxp, *dyp, *dwpre, *dwpost;
        CK(hipMalloc(&dxp, (size_t)M * 64 * 4)); CK(hipMalloc(&dyp, (size_t)M * 64 * 4));
        CK(hipMalloc(&dwpre, (size_t)N * 64 * 4)); CK(hipMalloc(&dwpost, (size_t)64 * K * 4));
        std::vector<float> xp((size_t)M * 64, 0.f), wpre((size_t)N * 64, 0.f), wpost((size_t)64 * K, 0.f);
        for (int r = 0; r < M; ++r) for (int c = 0; c < 51; ++c) xp[(size_t)r * 64 + c] = u(rng);
        for (int n = 0; n < N; ++n) for (int c = 0; c < 51; ++c) wpre[(size_t)n * 64 + c] = u(rng) * 0.1f;
        for (int n = 0; n < 51; ++n) for (int k = 0; k < K; ++k) wpost[(size_t)n * K + k] = u(rng) * 0.03f;
        CK(hipMemcpy(dxp, xp.data(), xp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dyp, xp.data(), xp.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dwpre, wpre.data(), wpre.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwpost, wpost.data(), wpost.size() * 4, hipMemcpyHostToDevice));
        LayerArgs pre{}; pre.X = dxp; pre.ldx = 64; pre.W = dwpre; pre.ldw = 64; pre.bias = db; pre.gamma = dg; pre.beta = dbe; pre.out = dy; pre.ldo = N;
        pre.K = 64; pre.N = N; pre.Mp = M; pre.kzero8 = 1;
        LayerArgs post{}; post.X = dx; post.ldx = K; post.W = dwpost; post.ldw = K; post.bias = db; post.out = dyp; post.ldo = 64; post.K = K; post.N = 64;
        post.Mp = M; post.sde_a = 1.0001f; post.sde_c = -0.01f;
        auto timeit = [&](const char *name, auto fn, double flop, double bytes) {
            for (int r = 0; r < 300; ++r) fn();
            CK(hipEventRecord(e0));
            for (int r = 0; r < 200; ++r) fn();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 200;
            printf("%-74s %7.1f us   %6.1f TF issued   %5.2f TB/s algorithmic\n", name, ms * 1e3, flop / ms / 1e9, bytes / ms / 1e9);
            return 0;
        };
        const double fpre56 = 2.0 * M * 56 * N, fpre64 = 2.0 * M * 64 * N, fpost = 2.0 * M * K * 64;
        const double bpre = (double)M * (N + 64) * 4, bpost = (double)M * (K + 128) * 4;
        timeit("pre_dense 64x128 K=56 (product: skips the zero k group) + GN + SiLU", [&] { launch_cfg<64, 128, 2, 4, EPI_GN_SILU, 2, 32, SCHED_THIN, 1, 1>(pre, 0); }, fpre56, bpre);
        timeit("pre_dense 64x128 K=64 + GN + SiLU", [&] { launch_cfg<64, 128, 2, 4, EPI_GN_SILU, 2, 32, SCHED_THIN>(pre, 0); }, fpre64, bpre);
        // (the no-epilogue / no-tile-load ablations of round 2 needed template parameters the product tile no longer carries: their
        //  numbers are in profiles/ubench_thin_r02.txt)
        timeit("pre_dense K=56, bias-only epilogue (stores, no GroupNorm / SiLU) [ablation]", [&] { launch_cfg<64, 128, 2, 4, EPI_BIAS, 2, 32, SCHED_THIN, 1, 1>(pre, 0); }, fpre56, bpre);
        timeit("post_dense 64x64 ring 4 + SDE update (product shape, no reprojection)", [&] { launch_cfg<64, 64, 2, 2, EPI_SDE, 4, 32, SCHED_THIN>(post, 0); }, fpost, bpost);
        return 0;
    }
    // CPU reference (double) for GN+SiLU on a sample of rows spread over the whole batch
    std::vector<int> rows;
    for (int r = 0; r < M; r += 997) rows.push_back(r);
    for (int r = M - 300; r < M; r += 7) rows.push_back(r);
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
    std::vector<float> y((size_t)M * N);
    for (int var = 0; var < UBENCH_NVAR; ++var) {
        if (getenv("UBENCH_ONLY") && !strstr(getenv("UBENCH_ONLY"), ("," + std::to_string(var) + ",").c_str())) continue;
        CK(hipMemset(dy, 0xff, (size_t)M * N * 4));
        hipError_t e = launch_variant(a, var, 0);
        if (e != hipSuccess) { printf("variant %d: launch error %s\n", var, hipGetErrorString(e)); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
        double maxd = 0; int badrow = -1, nbad = 0;
        for (size_t ri = 0; ri < rows.size(); ++ri) {
            double d = 0;
            for (int n = 0; n < N; ++n) { double dd = fabs((double)y[(size_t)rows[ri] * N + n] - cref[ri * N + n]); if (!(dd <= d)) d = dd; }
            if (!(d <= 1e-4)) { if (badrow < 0) badrow = rows[ri]; ++nbad; }
            if (!(d <= maxd)) maxd = d;
        }
        const int reps = getenv("UBENCH_REPS") ? atoi(getenv("UBENCH_REPS")) : 100;
        for (int r = 0; r < 250; ++r) launch_variant(a, var, 0);   // clocks ramp up over ~100 ms: warm up first
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch_variant(a, var, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("variant %2d (%-50s): %7.1f us %6.1f TF  max|y-ref64| %.1e  bad rows %d (first %d)\n", var, variant_name(var), ms * 1e3,
               2.0 * M * N * K / ms / 1e9, maxd, nbad, badrow);
    }
    return 0;
}
