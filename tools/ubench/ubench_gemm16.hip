// Stand-alone harness for the PRODUCT split-fp16 hidden layer (zedo_gemm16.hip: launch_layer16, all its tile shapes and its
// real epilogue), with per-workgroup timelines.  Build / run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DZEDO_UBENCH tools/ubench/ubench_gemm16.hip -o tools/ubench/ubench_gemm16
//   tools/ubench/ubench_gemm16 50752 [timeline.bin]      then      python tools/ubench/timeline_stats.py timeline.bin
// Round 5: also runs the persistent tile ping-pong kernel (launch_layer16_tp) on the same operands, times it and compares its
// output with the pair kernel's BIT FOR BIT (both epilogues; the residual one in place, as the product calls it).
#include "../../zedo-release_amd/csrc/zedo_gemm16.hip"
#include <cstdio>
#include <random>
#include <string>
#include <vector>

using namespace zedo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 50752, N = 1024, K = 1024;
    const char *dump = argc > 2 ? argv[2] : nullptr;
    const int Ma = (M + 127) / 128 * 128;     // the ping-pong kernel READS whole 128-row tiles (the product pads its workspace)
    std::vector<float> hx((size_t)Ma * K), hw((size_t)N * K), hb(N), hg(N), hbe(N);
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
    uint16_t *pres0, *outA, *outB;
    CK(hipMalloc(&pres0, hx.size() * 4)); CK(hipMalloc(&outA, hx.size() * 4)); CK(hipMalloc(&outB, hx.size() * 4));
    CK(launch_split_planes(dx, Ma, K, K, 1.0f, px, Ma, 0)); CK(launch_split_planes(dx, Ma, K, K, 0.5f, pres0, Ma, 0));
    CK(hipMemcpy(pres, pres0, hx.size() * 4, hipMemcpyDeviceToDevice));
    CK(launch_split_planes(dw, N, K, K, 16384.0f * 32.0f, pw, N, 0));
    long long *dtl = nullptr;
    const int maxwg = (M / 64 + 2) * 16;
    CK(hipMalloc(&dtl, (size_t)maxwg * 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int epi : {EPI_GN_SILU, EPI_GN_SILU_RES}) {
        Layer16Args a{};
        a.X = px; a.W = pw; a.bias = db; a.gamma = dg; a.beta = dbe; a.unscale = 1.0f / (16384.0f * 32.0f);
        a.res = epi == EPI_GN_SILU_RES ? pres : nullptr; a.out = epi == EPI_GN_SILU_RES ? (void *)pres : (void *)pout; a.out_f32 = 0;
        a.K = K; a.N = N; a.Mp = M; a.ldx = a.ldo = Ma;
        long long *nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &nul, sizeof(nul)));
        for (int r = 0; r < 100; ++r) CK(launch_layer16(a, epi, 0));
        CK(hipEventRecord(e0));
        for (int r = 0; r < 100; ++r) CK(launch_layer16(a, epi, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 100;
        printf("launch_layer16 %-16s M = %d: %7.1f us  (%.1f fp32-equivalent TFLOP/s)\n", epi == EPI_GN_SILU ? "GN_SILU" : "GN_SILU_RES", M, ms * 1e3,
               2.0 * M * N * K / ms / 1e9);
        if (dump) {
            CK(hipMemset(dtl, 0, (size_t)maxwg * 64));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &dtl, sizeof(dtl)));
            CK(launch_layer16(a, epi, 0));
            CK(hipDeviceSynchronize());
            std::vector<long long> h((size_t)maxwg * 8);
            CK(hipMemcpy(h.data(), dtl, h.size() * 8, hipMemcpyDeviceToHost));
            std::string path = std::string(dump) + (epi == EPI_GN_SILU ? ".plain" : ".res");
            FILE *f = fopen(path.c_str(), "wb"); fwrite(h.data(), 8, h.size(), f); fclose(f);
            printf("   timeline -> %s\n", path.c_str());
        }
        // ---- the tile ping-pong kernel on the same operands: bitwise equal?  how fast?
        {
            const bool res = epi == EPI_GN_SILU_RES;
            Layer16Args b = a;
            std::vector<uint16_t> hA((size_t)Ma * N * 2), hB((size_t)Ma * N * 2);
            auto run_once = [&](bool tp, uint16_t *dst, std::vector<uint16_t> &host) -> int {
                if (res) { CK(hipMemcpy(dst, pres0, (size_t)Ma * N * 4, hipMemcpyDeviceToDevice)); b.res = dst; }
                else CK(hipMemset(dst, 0xee, (size_t)Ma * N * 4));
                b.out = dst;
                CK(tp ? launch_layer16_tp(b, epi, 0) : launch_layer16(b, epi, 0));
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(host.data(), dst, host.size() * 2, hipMemcpyDeviceToHost));
                return 0;
            };
            if (run_once(false, outA, hA) || run_once(true, outB, hB)) return 1;
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < hA.size(); ++i) { const size_t row = (i / 32) % Ma; if (row < (size_t)M && hA[i] != hB[i]) { if (!bad) first = i; ++bad; } }
            size_t tail_bad = 0;      // (planes are k-block-major: the rows beyond M of every block belong to the workspace padding and may be written)
            printf("   tile ping-pong vs pair kernel: %zu of %zu halfwords differ%s; rows beyond M written (allowed: workspace padding): %zu\n", bad, hA.size(),
                   bad ? (std::string(" (first at block ") + std::to_string(first / 32 / Ma) + " row " + std::to_string((first / 32) % Ma) + ")").c_str() : " (bitwise equal)", tail_bad);
            b.res = res ? outB : nullptr; b.out = outB;
            for (int r = 0; r < 50; ++r) CK(launch_layer16_tp(b, epi, 0));
            CK(hipEventRecord(e0));
            for (int r = 0; r < 100; ++r) CK(launch_layer16_tp(b, epi, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms2; CK(hipEventElapsedTime(&ms2, e0, e1)); ms2 /= 100;
            long long *dclk; CK(hipMalloc(&dclk, 16)); CK(hipMemset(dclk, 0, 16));
            b.clk = dclk; CK(launch_layer16_tp(b, epi, 0)); CK(hipDeviceSynchronize()); b.clk = nullptr;
            long long ck[2]; CK(hipMemcpy(ck, dclk, 16, hipMemcpyDeviceToHost));
            const double ghz = ck[1] > 0 ? (double)ck[0] / ((double)ck[1] / 100e6) / 1e9 : 0.0;
            {   // barrier waits per ROLE (compute / DMA pair / epilogue pair), summed over all waves and workgroups
                CK(hipMemset(dtl, 0, (size_t)maxwg * 64));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &dtl, sizeof(dtl)));
                CK(launch_layer16_tp(b, epi, 0)); CK(hipDeviceSynchronize());
                long long *nul2 = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &nul2, sizeof(nul2)));
                std::vector<long long> h(256 * 8 * 8);
                CK(hipMemcpy(h.data(), dtl, h.size() * 8, hipMemcpyDeviceToHost));
                double wt[4] = {0, 0, 0, 0}, tt[4] = {0, 0, 0, 0};
                for (size_t w = 0; w < 256 * 8; ++w) for (int r = 0; r < 4; ++r) { wt[r] += (double)h[w * 8 + 2 * r]; tt[r] += (double)h[w * 8 + 2 * r + 1]; }
                const double per = 256.0 * 8 * 3 * 65;      // (wave, iteration) pairs per role: 8 waves x 6 tiles / 2 roles x 65 barriers
                printf("   cycles per iteration - compute wave: %.0f (of which at the barrier %.0f) | support wave: vmcnt wait + barrier %.0f (barrier %.0f), DMA issue %.0f, epilogue piece %.0f\n",
                       tt[0] / per, wt[0] / per, tt[1] / per, wt[1] / per, tt[2] / per, tt[3] / per);
            }
            printf("   launch_layer16_tp %-12s M = %d: %7.1f us  (%.1f fp32-equivalent TFLOP/s; pair kernel %.1f us)  clock %.3f GHz, MFMA-only %.0f us\n",
                   res ? "GN_SILU_RES" : "GN_SILU", M, ms2 * 1e3, 2.0 * M * N * K / ms2 / 1e9, ms * 1e3, ghz,
                   3.0 * 32.0 * ((double)M / 32) * (N / 32) * (K / 16) / 1024.0 / (ghz * 1e3));
        }
    }
    return 0;
}
