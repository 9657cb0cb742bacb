// Stand-alone harness for the PRODUCT split-fp16 hidden layer (zedo_gemm16.hip: launch_layer16, all its tile shapes and its
// real epilogue), with per-workgroup timelines.  Build / run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DZEDO_UBENCH tools/ubench/ubench_gemm16.hip -o tools/ubench/ubench_gemm16
//   tools/ubench/ubench_gemm16 50752 [timeline.bin]      then      python tools/ubench/timeline_stats.py timeline.bin
#include "../../zedo-release_amd/csrc/zedo_gemm16.hip"
#include <cstdio>
#include <random>
#include <vector>

using namespace zedo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 50752, N = 1024, K = 1024;
    const char *dump = argc > 2 ? argv[2] : nullptr;
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N), hg(N), hbe(N);
    std::mt19937 rng(1); std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (auto &v : hx) v = u(rng);
    for (auto &v : hw) v = u(rng) * 0.03125f;
    for (auto &v : hb) v = u(rng);
    for (auto &v : hg) v = 1.0f + 0.5f * u(rng);
    for (auto &v : hbe) v = 0.2f * u(rng);
    float *dx, *dw, *db, *dg, *dbe; uint16_t *px, *pw, *pres, *pout;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dg, N * 4)); CK(hipMalloc(&dbe, N * 4));
    CK(hipMalloc(&px, hx.size() * 4)); CK(hipMalloc(&pw, hw.size() * 4)); CK(hipMalloc(&pres, hx.size() * 4)); CK(hipMalloc(&pout, hx.size() * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, hg.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbe, hbe.data(), N * 4, hipMemcpyHostToDevice));
    CK(launch_split_planes(dx, M, K, K, 1.0f, px, 0)); CK(launch_split_planes(dx, M, K, K, 1.0f, pres, 0));
    CK(launch_split_planes(dw, N, K, K, 16384.0f * 32.0f, pw, 0));
    long long *dtl = nullptr;
    const int maxwg = (M / 64 + 2) * 16;
    CK(hipMalloc(&dtl, (size_t)maxwg * 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int epi : {EPI_GN_SILU, EPI_GN_SILU_RES}) {
        Layer16Args a{};
        a.X = px; a.W = pw; a.bias = db; a.gamma = dg; a.beta = dbe; a.unscale = 1.0f / (16384.0f * 32.0f);
        a.res = epi == EPI_GN_SILU_RES ? pres : nullptr; a.out = epi == EPI_GN_SILU_RES ? (void *)pres : (void *)pout; a.out_f32 = 0;
        a.K = K; a.N = N; a.Mp = M;
        long long *nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline16), &nul, sizeof(nul)));
        for (int r = 0; r < 100; ++r) CK(launch_layer16(a, epi, 0));
        CK(hipEventRecord(e0));
        for (int r = 0; r < 100; ++r) CK(launch_layer16(a, epi, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 100;
        printf("launch_layer16 %-16s M = %d: %7.1f us  (%.1f fp32-equivalent TFLOP/s)\n", epi == EPI_GN_SILU ? "GN_SILU" : "GN_SILU_RES", M, ms * 1e3,
               2.0 * M * N * K / ms / 1e9);
        if (dump) {
            CK(hipMemset(dtl, 0, (size_t)maxwg * 64));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline16), &dtl, sizeof(dtl)));
            CK(launch_layer16(a, epi, 0));
            CK(hipDeviceSynchronize());
            std::vector<long long> h((size_t)maxwg * 8);
            CK(hipMemcpy(h.data(), dtl, h.size() * 8, hipMemcpyDeviceToHost));
            std::string path = std::string(dump) + (epi == EPI_GN_SILU ? ".plain" : ".res");
            FILE *f = fopen(path.c_str(), "wb"); fwrite(h.data(), 8, h.size(), f); fclose(f);
            printf("   timeline -> %s\n", path.c_str());
        }
    }
    return 0;
}
