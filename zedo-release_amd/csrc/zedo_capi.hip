// C ABI of libzedo_hip.so (declared in include/zedo_hip.h): handle management, table building and the
// launch sequences of the ZeDO hot path on gfx950.
// Process-wide state, all of it outside the data path: the sampled-timing diagnostic (g_prof: one profiling session at a
// time; its bookkeeping is behind a mutex, the launch paths read one atomic flag), the ZEDO_CHUNK_ROWS value read once, and per-device launch attributes
// cached in zedo_gemm.hip.  Everything a call computes with lives in the caller's buffers or the opaque handles.
#include "../../include/zedo_hip.h"
#include "zedo_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

using namespace zedo;

struct zedo_weights {
    int J3, hidden, embed, n_blocks;
    float *d_all;           // one allocation holding everything below
    const float *W_pre;     // [H][XLD]   pre_dense.weight, K zero-padded 51 -> 64
    const float *W_hid[4];  // [H][H]     b{1,2}_dense{1,2}.weight
    const float *W_post;    // [XLD][H]   post_dense.weight, rows zero-padded 51 -> 64
    const float *b_post;    // [XLD]
    const float *gamma[NLAYER], *beta[NLAYER];
    const float *W_s, *b_s;         // shared_time_embed.0
    const float *W_t[NLAYER];       // [H][E]  *_t.weight
    const float *b_sum[NLAYER];     // [H]     *_t.bias + dense.bias (both row-invariant)
    // scratch of zedo_schedule_create for schedules of up to ROW_PAD timestamps (the per-step surface builds
    // one-entry schedules): posemb [ROW_PAD][E] | temb [ROW_PAD][E] | t [ROW_PAD]; guarded by scratch_mu
    float *d_scratch;
    std::mutex scratch_mu;
    // ZEDO_MATH_F16X3 (zedo_weights_set_math): the four hidden weights as split-fp16 planes of W * 2^wshift
    int math;
    uint16_t *d_W16;            // hidden [4][H][H/16][2][16] | pre_dense [H][4][2][16] | post_dense [XLD][H/16][2][16]
    float wmax_hid[6];          // max |w| of the four hidden layers, pre_dense, post_dense (from the host copy at create time)
    float act_bound;            // upper bound of every activation |h| the network can produce (from gamma / beta, see zedo_weights_create)
    float unscale[6];           // 2^-wshift, same order
    bool finite16;              // no NaN / inf among the six weight matrices and the GroupNorm parameters (fmax cannot see a NaN)
    float row_ratio16;          // min over the six matrices and their non-zero rows of max|w_row| / max|w_matrix| (1 = every row as large as the largest)
};

struct zedo_schedule {
    int S, Sp, nl, hidden;
    float *d_tbias;  // [Sp][NLAYER][H]
    std::vector<float> a, c;
};

#define HIPCHK(x)                      \
    do {                               \
        hipError_t e_ = (x);           \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static int check_device();
static int check_device_fwd() { return check_device(); }

// ---- optional sampled kernel timing (diagnostics for bench.py's roofline object) ------------------
// Every `every`-th launch of a kernel class is bracketed by two hipEvents recorded on the launch stream
// itself, so the elapsed time is that kernel's duration as the stream saw it.  Off by default.
namespace {
struct Prof {
    std::atomic<bool> on{false};     // the only member the launch paths read without the lock
    std::mutex mu;                   // every other member: sessions are serialised, sampling from several host threads is safe
    int every = 1;
    struct Pair { hipEvent_t a, b; int cls; };
    std::vector<Pair> pool;
    long long *d_clk = nullptr;   // [pool.size()][2]: {shader cycles, 100 MHz ticks} of workgroup 0 of a sampled hidden launch
    size_t clk_cap = 0;
    double ghz = 0.0;             // result of the last session
    double bracket_ms = 0.0;      // result of the last session: what an EMPTY event pair on the launch stream measures
    hipStream_t last_stream = nullptr;
    size_t used = 0;
    long long seen[ZEDO_PROF_CLASSES] = {};
} g_prof;

struct ProfScope {
    Prof::Pair *pr = nullptr;
    hipStream_t st;
    long long *clk = nullptr;     // where the sampled launch may drop its clock pair
    ProfScope(int cls, hipStream_t s) : st(s) {
        if (!g_prof.on.load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (!g_prof.on.load(std::memory_order_relaxed)) return;
        if ((g_prof.seen[cls]++ % g_prof.every) != 0 || g_prof.used >= g_prof.pool.size()) return;
        if (g_prof.d_clk && g_prof.used < g_prof.clk_cap) clk = g_prof.d_clk + 2 * g_prof.used;
        pr = &g_prof.pool[g_prof.used++];
        pr->cls = cls;
        g_prof.last_stream = st;
        (void)hipEventRecord(pr->a, st);
    }
    ~ProfScope() { if (pr) (void)hipEventRecord(pr->b, st); }
};
}  // namespace

extern "C" int zedo_probe_mfma_peak(int iters, double *h_tflops, double *h_shader_ghz, void *stream) {
    if (iters < 1000 || iters > 10000000) return ZEDO_E_BADARG;
    if (int rc = check_device_fwd()) return rc;
    HIPCHK(probe_mfma_peak(iters, h_tflops, h_shader_ghz, (hipStream_t)stream));
    return ZEDO_OK;
}

extern "C" int zedo_probe_mfma_peak_f16(int iters, double *h_tflops, double *h_shader_ghz, void *stream) {
    if (iters < 1000 || iters > 10000000) return ZEDO_E_BADARG;
    if (int rc = check_device_fwd()) return rc;
    HIPCHK(probe_mfma_peak(iters, h_tflops, h_shader_ghz, (hipStream_t)stream, true));
    return ZEDO_OK;
}

extern "C" int zedo_profile_start(int sample_every, int max_samples) {
    if (sample_every < 1 || max_samples < 1) return ZEDO_E_BADARG;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.pool.size() < (size_t)max_samples) {
        size_t old = g_prof.pool.size();
        g_prof.pool.resize(max_samples);
        for (size_t i = old; i < g_prof.pool.size(); ++i) {
            HIPCHK(hipEventCreate(&g_prof.pool[i].a));
            HIPCHK(hipEventCreate(&g_prof.pool[i].b));
        }
    }
    if (g_prof.clk_cap < g_prof.pool.size()) {
        (void)hipFree(g_prof.d_clk);
        g_prof.d_clk = nullptr; g_prof.clk_cap = 0;
        if (hipMalloc(&g_prof.d_clk, sizeof(long long) * 2 * g_prof.pool.size()) == hipSuccess) g_prof.clk_cap = g_prof.pool.size();
    }
    if (g_prof.d_clk) HIPCHK(hipMemset(g_prof.d_clk, 0, sizeof(long long) * 2 * g_prof.clk_cap));
    g_prof.used = 0;
    for (auto &v : g_prof.seen) v = 0;
    g_prof.every = sample_every;
    g_prof.on.store(true, std::memory_order_release);
    return ZEDO_OK;
}

extern "C" double zedo_profile_shader_ghz(void) { return g_prof.ghz; }
extern "C" double zedo_profile_bracket_ms(void) { return g_prof.bracket_ms; }

extern "C" int zedo_profile_stop(double *h_total_ms, long long *h_samples, long long *h_launches) {
    g_prof.on.store(false, std::memory_order_release);
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double tot[ZEDO_PROF_CLASSES] = {};
    long long cnt[ZEDO_PROF_CLASSES] = {};
    for (size_t i = 0; i < g_prof.used; ++i) {
        HIPCHK(hipEventSynchronize(g_prof.pool[i].b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, g_prof.pool[i].a, g_prof.pool[i].b));
        tot[g_prof.pool[i].cls] += ms;
        cnt[g_prof.pool[i].cls] += 1;
    }
    // What the bracket itself costs: the median of 15 EMPTY event pairs on the stream the samples were taken on (the
    // samples above are synchronised, the stream is idle).  A sampled duration contains it once; zedo_profile_stop
    // reports the raw sums, callers subtract samples x zedo_profile_bracket_ms() (bench.py does).
    g_prof.bracket_ms = 0.0;
    if (g_prof.used) {
        float em[15];
        int got = 0;
        for (int i = 0; i < 15; ++i) {
            if (hipEventRecord(g_prof.pool[0].a, g_prof.last_stream) != hipSuccess) break;
            if (hipEventRecord(g_prof.pool[0].b, g_prof.last_stream) != hipSuccess) break;
            if (hipEventSynchronize(g_prof.pool[0].b) != hipSuccess) break;
            if (hipEventElapsedTime(&em[got], g_prof.pool[0].a, g_prof.pool[0].b) != hipSuccess) break;
            ++got;
        }
        if (got) {
            std::sort(em, em + got);
            g_prof.bracket_ms = em[got / 2];
        }
    }
    g_prof.ghz = 0.0;
    if (g_prof.d_clk && g_prof.used) {
        std::vector<long long> ck(2 * g_prof.used);
        HIPCHK(hipMemcpy(ck.data(), g_prof.d_clk, sizeof(long long) * ck.size(), hipMemcpyDeviceToHost));
        double cyc = 0, ticks = 0;
        for (size_t i = 0; i < g_prof.used; ++i) if (ck[2 * i + 1] > 0) { cyc += (double)ck[2 * i]; ticks += (double)ck[2 * i + 1]; }
        if (ticks > 0) g_prof.ghz = cyc / (ticks / 100e6) / 1e9;
    }
    for (int c = 0; c < ZEDO_PROF_CLASSES; ++c) {
        if (h_total_ms) h_total_ms[c] = tot[c];
        if (h_samples) h_samples[c] = cnt[c];
        if (h_launches) h_launches[c] = g_prof.seen[c];
    }
    g_prof.used = 0;
    return ZEDO_OK;
}

static size_t chunk_rows_cap() {
    // read once; a function-local static is initialised exactly once even when several host threads make their first call together
    static const size_t cap = [] {
        const char *e = getenv("ZEDO_CHUNK_ROWS");
        const long v = e ? atol(e) : 0;
        return v > 0 ? (size_t)round_up((int)v, ROW_PAD) : (size_t)1 << 20;
    }();
    return cap;
}

extern "C" int zedo_abi_version(void) { return ZEDO_ABI_VERSION; }

extern "C" const char *zedo_error_string(int code) {
    switch (code) {
        case ZEDO_OK: return "ok";
        case ZEDO_E_BADARG: return "zedo: bad argument (null pointer, size, or unsupported dimension)";
        case ZEDO_E_NOGPU: return "zedo: no gfx950 device";
        case ZEDO_E_WORKSPACE: return "zedo: workspace too small";
    }
    return code > 0 ? hipGetErrorString((hipError_t)code) : "zedo: unknown error";
}

static int check_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ZEDO_E_NOGPU;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return ZEDO_E_NOGPU;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) return ZEDO_E_NOGPU;
    return ZEDO_OK;
}

extern "C" int zedo_weights_create(const float *h_params, size_t n_floats, int n_joints, int joint_dim, int hidden,
                                   int embed, int n_blocks, void *stream, zedo_weights_t **out) {
    if (!h_params || !out) return ZEDO_E_BADARG;
    const int J3 = n_joints * joint_dim;
    if (hidden != HID || embed != EMB || n_blocks != 2 || J3 < 1 || J3 > XLD) return ZEDO_E_BADARG;
    const size_t H = hidden, E = embed;
    const size_t expect = (H * J3 + H) + (H * E + H) + 2 * H + (E * E + E) + 4 * ((H * H + H) + (H * E + H) + 2 * H) +
                          ((size_t)J3 * H + J3);
    if (n_floats != expect) return ZEDO_E_BADARG;
    if (int rc = check_device()) return rc;
    hipStream_t st = (hipStream_t)stream;

    // device layout (floats)
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
    const size_t o_Wpre = take(H * XLD), o_Whid0 = take(4 * H * H), o_Wpost = take((size_t)XLD * H), o_bpost = take(XLD);
    const size_t o_gamma = take(NLAYER * H), o_beta = take(NLAYER * H), o_Ws = take(E * E), o_bs = take(E);
    const size_t o_Wt = take(NLAYER * H * E), o_bsum = take(NLAYER * H);
    std::vector<float> img(off, 0.0f);

    const float *p = h_params;
    auto next = [&](size_t n) { const float *q = p; p += n; return q; };
    float wmax_hid_tmp[6] = {0, 0, 0, 0, 0, 0};
    // |SiLU(GroupNorm(.))| <= max|gamma| sqrt(31) + max|beta| per layer (a group of 32 normalised values has |v| <= sqrt(31));
    // h = pre, then h += h2 twice: |h| <= bound[0] + bound[2] + bound[4]
    float gn_bound[NLAYER] = {0, 0, 0, 0, 0};
    // std::fmax drops a NaN operand, so the maxima above / below cannot see one: finiteness of the six weight matrices and
    // of every GroupNorm parameter is tracked on its own (zedo_weights_set_math refuses f16x3 when it is false)
    bool finite = true;
    auto all_finite = [&](const float *v, size_t n) {
        for (size_t q = 0; q < n; ++q) finite &= (bool)std::isfinite(v[q]);
    };
    // The split-fp16 copy of a matrix carries ONE power-of-two scale (max |w| -> [2^13, 2^14)): a row whose largest weight is 2^-s of
    // the matrix maximum keeps 22 - max(0, s + t - 16) significant bits in an element 2^-t below that row maximum (its low piece
    // goes fp16-denormal).  Rows within 2^-8 of the matrix maximum (every initialisation and every GroupNorm-ed trained layer seen)
    // keep fp32-level accuracy; zedo_weights_set_math refuses the mode below that (ZEDO_F16X3_MIN_ROW_RATIO).
    float row_ratio = 1.0f;
    auto row_ratio_of = [&](const float *wm, size_t rows, size_t cols, float mat_max) {
        if (!(mat_max > 0.f)) return;
        for (size_t r = 0; r < rows; ++r) {
            float rm = 0.f;
            for (size_t c = 0; c < cols; ++c) rm = std::fmax(rm, std::fabs(wm[r * cols + c]));
            if (rm > 0.f) row_ratio = std::fmin(row_ratio, rm / mat_max);       // all-zero rows carry no information to lose
        }
    };
    auto gn_bound_of = [&](const float *g, const float *be, size_t n) {
        float gm = 0.f, bm = 0.f;
        for (size_t q = 0; q < n; ++q) { gm = std::fmax(gm, std::fabs(g[q])); bm = std::fmax(bm, std::fabs(be[q])); }
        all_finite(g, n); all_finite(be, n);
        return gm * 5.5677643f + bm;     // sqrt(31)
    };
    // pre_dense
    const float *w_pre = next(H * J3), *b_pre = next(H), *w_pre_t = next(H * E), *b_pre_t = next(H);
    const float *g_pre = next(H), *be_pre = next(H);
    gn_bound[0] = gn_bound_of(g_pre, be_pre, H);
    const float *w_s = next(E * E), *b_s = next(E);
    for (size_t n = 0; n < H; ++n) memcpy(&img[o_Wpre + n * XLD], w_pre + n * J3, sizeof(float) * J3);
    for (size_t q = 0; q < H * J3; ++q) wmax_hid_tmp[4] = std::fmax(wmax_hid_tmp[4], std::fabs(w_pre[q]));
    all_finite(w_pre, H * J3);
    row_ratio_of(w_pre, H, J3, wmax_hid_tmp[4]);
    memcpy(&img[o_gamma], g_pre, sizeof(float) * H);
    memcpy(&img[o_beta], be_pre, sizeof(float) * H);
    memcpy(&img[o_Ws], w_s, sizeof(float) * E * E);
    memcpy(&img[o_bs], b_s, sizeof(float) * E);
    memcpy(&img[o_Wt], w_pre_t, sizeof(float) * H * E);
    for (size_t n = 0; n < H; ++n) img[o_bsum + n] = b_pre_t[n] + b_pre[n];
    for (int l = 1; l < NLAYER; ++l) {
        const float *w = next(H * H), *b = next(H), *wt = next(H * E), *bt = next(H), *g = next(H), *be = next(H);
        memcpy(&img[o_Whid0 + (size_t)(l - 1) * H * H], w, sizeof(float) * H * H);
        gn_bound[l] = gn_bound_of(g, be, H);
        float wm = 0.f;
        for (size_t q = 0; q < H * H; ++q) wm = std::fmax(wm, std::fabs(w[q]));
        all_finite(w, H * H);
        wmax_hid_tmp[l - 1] = wm;
        row_ratio_of(w, H, H, wm);
        memcpy(&img[o_Wt + (size_t)l * H * E], wt, sizeof(float) * H * E);
        memcpy(&img[o_gamma + (size_t)l * H], g, sizeof(float) * H);
        memcpy(&img[o_beta + (size_t)l * H], be, sizeof(float) * H);
        // h += dense(x); h += dense_t(temb): (acc + b) + (acc_t + b_t).  Summing the two biases first
        // changes only the rounding order of a row-invariant constant.
        for (size_t n = 0; n < H; ++n) img[o_bsum + (size_t)l * H + n] = bt[n] + b[n];
    }
    const float *w_post = next((size_t)J3 * H), *b_post = next(J3);
    memcpy(&img[o_Wpost], w_post, sizeof(float) * J3 * H);
    for (size_t q = 0; q < (size_t)J3 * H; ++q) wmax_hid_tmp[5] = std::fmax(wmax_hid_tmp[5], std::fabs(w_post[q]));
    all_finite(w_post, (size_t)J3 * H);
    row_ratio_of(w_post, J3, H, wmax_hid_tmp[5]);
    memcpy(&img[o_bpost], b_post, sizeof(float) * J3);

    zedo_weights *w = new (std::nothrow) zedo_weights();
    if (!w) return (int)hipErrorOutOfMemory;
    w->J3 = J3; w->hidden = hidden; w->embed = embed; w->n_blocks = n_blocks;
    w->d_scratch = nullptr;
    w->math = ZEDO_MATH_F32; w->d_W16 = nullptr;
    for (int l = 0; l < 6; ++l) { w->wmax_hid[l] = wmax_hid_tmp[l]; w->unscale[l] = 1.0f; }
    w->act_bound = std::fmax(std::fmax(gn_bound[1], gn_bound[3]), gn_bound[0] + gn_bound[2] + gn_bound[4]);
    w->finite16 = finite;
    w->row_ratio16 = row_ratio;
    hipError_t e = hipMalloc(&w->d_all, off * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&w->d_scratch, sizeof(float) * ((size_t)2 * ROW_PAD * EMB + ROW_PAD));
    if (e != hipSuccess) { (void)hipFree(w->d_all); delete w; return (int)e; }
    e = hipMemcpyAsync(w->d_all, img.data(), off * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(w->d_all); (void)hipFree(w->d_scratch); delete w; return (int)e; }
    const float *d = w->d_all;
    w->W_pre = d + o_Wpre;
    for (int l = 0; l < 4; ++l) w->W_hid[l] = d + o_Whid0 + (size_t)l * H * H;
    w->W_post = d + o_Wpost; w->b_post = d + o_bpost;
    for (int l = 0; l < NLAYER; ++l) {
        w->gamma[l] = d + o_gamma + (size_t)l * H; w->beta[l] = d + o_beta + (size_t)l * H;
        w->W_t[l] = d + o_Wt + (size_t)l * H * E;  w->b_sum[l] = d + o_bsum + (size_t)l * H;
    }
    w->W_s = d + o_Ws; w->b_s = d + o_bs;
    *out = w;
    return ZEDO_OK;
}

extern "C" void zedo_weights_destroy(zedo_weights_t *w) {
    if (!w) return;
    (void)hipFree(w->d_all);
    (void)hipFree(w->d_scratch);
    (void)hipFree(w->d_W16);
    delete w;
}

extern "C" int zedo_weights_get_math(const zedo_weights_t *w) { return w ? w->math : ZEDO_E_BADARG; }

extern "C" int zedo_weights_set_math(zedo_weights_t *w, int mode, void *stream) {
    if (!w || (mode != ZEDO_MATH_F32 && mode != ZEDO_MATH_F16X3)) return ZEDO_E_BADARG;
    if (mode == ZEDO_MATH_F16X3 && !w->d_W16) {
        hipStream_t st = (hipStream_t)stream;
        if (!w->finite16) return ZEDO_E_BADARG;      // a NaN / inf weight or GroupNorm parameter has no fp16 image
        for (int l = 0; l < 6; ++l)
            if (!std::isfinite(w->wmax_hid[l])) return ZEDO_E_BADARG;
        // activations are stored as UNSCALED fp16 pieces: refuse the mode for a network whose GroupNorm parameters allow an
        // activation near the fp16 range (65504) instead of overflowing silently (trained checkpoints: O(10))
        if (!(w->act_bound < 32768.0f)) return ZEDO_E_BADARG;
        // one scale per matrix: rows far below the matrix maximum would lose bits (see zedo_weights_create)
        if (!(w->row_ratio16 >= 1.0f / 256.0f)) return ZEDO_E_BADARG;
        const size_t per = (size_t)HID * HID * 2;                                 // uint16 per hidden layer
        HIPCHK(hipMalloc(&w->d_W16, sizeof(uint16_t) * (4 * per + (size_t)HID * XLD * 2 + (size_t)XLD * HID * 2)));
        hipError_t e = hipSuccess;
        for (int l = 0; l < 6 && e == hipSuccess; ++l) {
            // power-of-two scale that puts max |w| into [2^13, 2^14): the low pieces of all but vanishing weights stay normal
            int ex = 0;
            const float wm = w->wmax_hid[l];
            if (wm > 0.f) (void)std::frexp(wm, &ex);                             // wm = f * 2^ex, f in [0.5, 1)
            const int shift = wm > 0.f ? 14 - ex : 0;
            w->unscale[l] = std::ldexp(1.0f, -shift);
            const float sc = std::ldexp(1.0f, shift);
            if (l < 4) e = launch_split_planes(w->W_hid[l], HID, HID, HID, sc, w->d_W16 + (size_t)l * per, HID, st);
            else if (l == 4) e = launch_split_planes(w->W_pre, HID, XLD, XLD, sc, w->d_W16 + 4 * per, HID, st);
            else e = launch_split_planes(w->W_post, XLD, HID, HID, sc, w->d_W16 + 4 * per + (size_t)HID * XLD * 2, XLD, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)hipFree(w->d_W16); w->d_W16 = nullptr; return (int)e; }
    }
    w->math = mode;
    return ZEDO_OK;
}

extern "C" int zedo_schedule_create(const zedo_weights_t *w, const float *h_t, int S, float label_scale, float beta_min,
                                    float beta_max, int n_sde, void *stream, zedo_schedule_t **out) {
    if (!w || !h_t || !out || S < 1 || n_sde < 1 || !(label_scale > 0.f)) return ZEDO_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int Sp = round_up(S, ROW_PAD);
    zedo_schedule *s = new (std::nothrow) zedo_schedule();
    if (!s) return (int)hipErrorOutOfMemory;
    s->S = S; s->Sp = Sp; s->nl = NLAYER; s->hidden = HID;
    s->a.resize(S); s->c.resize(S);
    // x' = x + drift*dt, drift = -beta/2 x - beta*disc*score, score = -eps/std, dt = -1/n_sde
    //   => a = 1 + beta/(2 n_sde),  c = -beta*disc/(n_sde*std)      (sde_lib.py:187-198, sampling.py:185-190)
    for (int i = 0; i < S; ++i) {
        const double t = (double)h_t[i] * ((double)label_scale / 999.0), b0 = (double)beta_min, b1 = (double)beta_max;
        const double beta = b0 + t * (b1 - b0);
        const double disc = 1.0 - std::exp(-2.0 * b0 * t - (b1 - b0) * t * t);
        const double sd = 1.0 - std::exp(2.0 * (-0.25 * t * t * (b1 - b0) - 0.5 * t * b0));
        s->a[i] = (float)(1.0 + 0.5 * beta / n_sde);
        s->c[i] = (float)(-(beta * disc) / (n_sde * sd));
#ifdef ZEDO_MUT_SDE_C       // tools/mutation_check.py only
        s->c[i] = (float)(-(beta * disc) / (n_sde * sd) * (1.0 + 1e-4));
#endif
    }
    // short schedules (the per-step surface asks for one timestamp at a time) borrow the scratch owned by the
    // weights handle: no hipMalloc / hipFree of temporaries per call
    const bool borrow = Sp == ROW_PAD;
    std::unique_lock<std::mutex> lk(const_cast<zedo_weights_t *>(w)->scratch_mu, std::defer_lock);
    float *d_t = nullptr, *d_pe = nullptr, *d_temb = nullptr;
    s->d_tbias = nullptr;
    hipError_t e = hipSuccess;
    if (borrow) {
        lk.lock();
        d_pe = w->d_scratch; d_temb = d_pe + (size_t)ROW_PAD * EMB; d_t = d_temb + (size_t)ROW_PAD * EMB;
    } else {
        e = hipMalloc(&d_t, sizeof(float) * S);
        if (e == hipSuccess) e = hipMalloc(&d_pe, sizeof(float) * (size_t)Sp * EMB);
        if (e == hipSuccess) e = hipMalloc(&d_temb, sizeof(float) * (size_t)Sp * EMB);
    }
    if (e == hipSuccess) e = hipMalloc(&s->d_tbias, sizeof(float) * (size_t)Sp * NLAYER * HID);
    if (e == hipSuccess) e = hipMemcpyAsync(d_t, h_t, sizeof(float) * S, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_posemb(d_t, S, Sp, label_scale, d_pe, st);
    if (e == hipSuccess) {
        // temb = SiLU(W_s pe + b_s)   (model.py:128-131,259)
        LayerArgs a{};
        a.X = d_pe; a.ldx = EMB; a.W = w->W_s; a.ldw = EMB; a.bias = w->b_s; a.out = d_temb; a.ldo = EMB;
        a.K = EMB; a.N = EMB; a.Mp = Sp;
        e = launch_layer(a, EPI_BIAS_SILU, st);
    }
    for (int l = 0; l < NLAYER && e == hipSuccess; ++l) {
        LayerArgs a{};
        a.X = d_temb; a.ldx = EMB; a.W = w->W_t[l]; a.ldw = EMB; a.bias = w->b_sum[l];
        a.out = s->d_tbias + (size_t)l * HID; a.ldo = NLAYER * HID; a.K = EMB; a.N = HID; a.Mp = Sp;
        e = launch_layer(a, EPI_BIAS, st);
    }
    // h_t may be freed by the caller and the borrowed scratch is released with the lock: nothing enqueued above may still be
    // running then - on the error path as well (an earlier launch or the copy may be in flight when a later step failed)
    {
        const hipError_t es = hipStreamSynchronize(st);
        if (e == hipSuccess) e = es;
    }
    if (!borrow) { (void)hipFree(d_t); (void)hipFree(d_pe); (void)hipFree(d_temb); }
    if (e != hipSuccess) { (void)hipFree(s->d_tbias); delete s; return (int)e; }
    *out = s;
    return ZEDO_OK;
}

extern "C" void zedo_schedule_destroy(zedo_schedule_t *s) {
    if (!s) return;
    (void)hipFree(s->d_tbias);
    delete s;
}

extern "C" int zedo_schedule_read(const zedo_schedule_t *s, float *h_tbias, float *h_a, float *h_c) {
    if (!s) return ZEDO_E_BADARG;
    if (h_tbias) HIPCHK(hipMemcpy(h_tbias, s->d_tbias, sizeof(float) * (size_t)s->S * NLAYER * HID, hipMemcpyDeviceToHost));
    if (h_a) memcpy(h_a, s->a.data(), sizeof(float) * s->S);
    if (h_c) memcpy(h_c, s->c.data(), sizeof(float) * s->S);
    return ZEDO_OK;
}

// rows of the workspace buffers: the chunk's rows rounded up to BATCH_PAD (64), the smallest row tile every layer accepts - no launch
// touches a row behind the batch's own padded rows (the 128-row tiles of either math mode cover floor(rows / 128) * 128 rows, the
// remainder runs on 64-row tiles)
static inline size_t ws_rows(int B) { return (size_t)round_up((int)std::min((size_t)B, chunk_rows_cap()), BATCH_PAD); }

extern "C" size_t zedo_workspace_bytes(int B) {
    if (B < 1) return 0;
    return ws_rows(B) * (size_t)(XLD + 2 * HID) * sizeof(float);
}

extern "C" int zedo_reproj_prepare(const float *d_uv, const float *d_K, const float *d_conf, int N, int J, float *d_geom,
                                   float *d_conf_clamped, void *stream) {
    if (!d_uv || !d_K || !d_geom || N < 1 || J < 1) return ZEDO_E_BADARG;
    HIPCHK(launch_reproj_prepare(d_uv, d_K, d_conf, N, J, d_geom, d_conf_clamped, (hipStream_t)stream));
    return ZEDO_OK;
}

extern "C" int zedo_reproj_degenerate(const float *d_geom, int N, int J, int *h_count, void *stream) {
    if (!d_geom || !h_count || N < 1 || J < 1) return ZEDO_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int *d_count = nullptr;
    HIPCHK(hipMalloc(&d_count, sizeof(int)));
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(int), st);
    if (e == hipSuccess) e = launch_reproj_degenerate(d_geom, N, J, d_count, st);
    if (e == hipSuccess) e = hipMemcpyAsync(h_count, d_count, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_count);
    return (int)e;
}

extern "C" int zedo_reproj_grad(const float *d_x, const float *d_geom, float *d_T, int solve_T, float *d_g, int B, int N,
                                int J, long long row_offset, void *stream) {
    if (!d_x || !d_geom || !d_T || !d_g || B < 1 || N < 1 || J != 17 || row_offset < 0) return ZEDO_E_BADARG;
    HIPCHK(launch_reproj_grad(d_x, d_geom, d_T, solve_T, d_g, B, N, J, row_offset, (hipStream_t)stream));
    return ZEDO_OK;
}

// The six dense layers of one score-network evaluation on Bp padded rows (model.py:264-291), ending either in
// eps -> xpad (EPI_BIAS, for zedo_score_eps) or in the SDE update of xpad (EPI_SDE).
struct NextReproj {          // reprojection of the next loop iteration, fused into the SDE epilogue (may be all-null)
    const float *geom = nullptr;
    float *T = nullptr;
    int solve = 0, B = 0, N = 0;
    long long row0 = 0;
};

// The six dense layers of one score-network evaluation (model.py:264-291) as three pieces - pre_dense, the four hidden layers,
// post_dense (round 6 measured post_dense of iteration i and pre_dense of iteration i + 1 in ONE launch: bit-identical, slower in both
// math modes, not adopted - profiles/seam_r06.txt).  h / h1: activations [rows][1024] fp32 (exact-fp32 mode) or split-fp16 planes
// (ZEDO_MATH_F16X3: the same 4 bytes per element, k-block-major with `ld` rows per k block); the pose state xpad, the time-bias rows
// and the SDE / reprojection epilogue are fp32 in both modes.
struct Net {
    const zedo_weights *w;
    float *xpad, *h, *h1;
    int Bp, ld;
    hipStream_t st;
    bool f16() const { return w->math == ZEDO_MATH_F16X3; }
    const uint16_t *W_pre16() const { return w->d_W16 + 4 * (size_t)HID * HID * 2; }
    const uint16_t *W_post16() const { return W_pre16() + (size_t)HID * XLD * 2; }

    LayerArgs pre_args(const float *tb) const {
        LayerArgs a{};
        a.Mp = Bp; a.X = xpad; a.ldx = XLD; a.W = w->W_pre; a.ldw = XLD; a.K = XLD; a.N = HID;
        a.kzero8 = w->J3 <= XLD - 8;   // 51 real inputs: k = 56..63 are zero in xpad and in the padded weight
        a.bias = tb; a.gamma = w->gamma[0]; a.beta = w->beta[0]; a.out = h; a.ldo = HID;
        return a;
    }
    Layer16Args pre_args16(const float *tb) const {     // X = the fp32 pose rows, split by the kernel
        Layer16Args b{};
        b.K = XLD; b.N = HID; b.Mp = Bp; b.ldo = ld; b.Xf32 = xpad; b.W = W_pre16(); b.unscale = w->unscale[4];
        b.bias = tb; b.gamma = w->gamma[0]; b.beta = w->beta[0]; b.out = h; b.out_f32 = 0;
        return b;
    }
    LayerArgs post_args(bool sde, float sa, float sc, float *eps_out, const NextReproj &nr) const {
        LayerArgs a{};
        a.Mp = Bp; a.X = h; a.ldx = HID; a.W = w->W_post; a.ldw = HID; a.K = HID; a.N = XLD; a.bias = w->b_post; a.ldo = XLD;
        a.scratch = h1 + (size_t)Bp * XLD;      // h1 is free here ([Bp][XLD] of it may hold eps_out): K-quarter sums of small batches
        if (sde) {
            a.out = xpad; a.sde_a = sa; a.sde_c = sc;
            a.rp_geom = nr.geom; a.rp_T = nr.T; a.rp_solve = nr.solve; a.rp_B = nr.B; a.rp_N = nr.N; a.rp_row0 = nr.row0;
        } else {
            a.out = eps_out;
        }
        return a;
    }
    Layer16Args post_args16(bool sde, float sa, float sc, float *eps_out, const NextReproj &nr) const {
        Layer16Args b{};
        b.K = HID; b.N = XLD; b.Mp = Bp; b.ldx = ld; b.X = reinterpret_cast<const uint16_t *>(h);
        b.W = W_post16(); b.unscale = w->unscale[5]; b.bias = w->b_post;
        if (sde) {
            b.xio = xpad; b.sde_a = sa; b.sde_c = sc;
            b.rp_geom = nr.geom; b.rp_T = nr.T; b.rp_solve = nr.solve; b.rp_B = nr.B; b.rp_N = nr.N; b.rp_row0 = nr.row0;
        } else {
            b.out = eps_out;
        }
        return b;
    }

    // pre_dense + pre_gnorm + SiLU
    hipError_t pre(const float *tb) const {
        ProfScope ps(ZEDO_PROF_PRE, st);
        if (f16()) return launch_layer16(pre_args16(tb), EPI_GN_SILU, st);
        return launch_layer(pre_args(tb), EPI_GN_SILU, st);
    }
    // the two residual blocks: h1 = f(h); h = h + f(h1), twice
    hipError_t hidden(const float *tb) const {
        hipError_t e = hipSuccess;
        for (int blk = 0; blk < 2 && e == hipSuccess; ++blk) {
            const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
            if (f16()) {
                const size_t per = (size_t)HID * HID * 2;                 // uint16 per hidden weight matrix
                Layer16Args b{};
                b.K = HID; b.N = HID; b.Mp = Bp; b.ldx = b.ldo = ld; b.out_f32 = 0;
                b.X = reinterpret_cast<const uint16_t *>(h); b.W = w->d_W16 + (size_t)(l1 - 1) * per; b.unscale = w->unscale[l1 - 1];
                b.bias = tb + (size_t)l1 * HID; b.gamma = w->gamma[l1]; b.beta = w->beta[l1]; b.out = h1;
                { ProfScope ps(ZEDO_PROF_HIDDEN, st); b.clk = ps.clk; e = launch_layer16(b, EPI_GN_SILU, st); b.clk = nullptr; }
                if (e != hipSuccess) break;
                // h = h + h2: the residual comes from h's planes and the sum goes back into them, in place
                b.X = reinterpret_cast<const uint16_t *>(h1); b.W = w->d_W16 + (size_t)(l2 - 1) * per; b.unscale = w->unscale[l2 - 1];
                b.bias = tb + (size_t)l2 * HID; b.gamma = w->gamma[l2]; b.beta = w->beta[l2];
                b.res = reinterpret_cast<const uint16_t *>(h); b.out = h;
                { ProfScope ps(ZEDO_PROF_HIDDEN, st); b.clk = ps.clk; e = launch_layer16(b, EPI_GN_SILU_RES, st); b.clk = nullptr; }
            } else {
                LayerArgs a{};
                a.Mp = Bp; a.X = h; a.ldx = HID; a.W = w->W_hid[l1 - 1]; a.ldw = HID; a.K = HID; a.N = HID;
                a.bias = tb + (size_t)l1 * HID; a.gamma = w->gamma[l1]; a.beta = w->beta[l1]; a.out = h1; a.ldo = HID;
                { ProfScope ps(ZEDO_PROF_HIDDEN, st); a.clk = ps.clk; e = launch_layer(a, EPI_GN_SILU, st); a.clk = nullptr; }
                if (e != hipSuccess) break;
                a.X = h1; a.W = w->W_hid[l2 - 1];
                a.bias = tb + (size_t)l2 * HID; a.gamma = w->gamma[l2]; a.beta = w->beta[l2]; a.out = h;  // h = h + h2, in place
                { ProfScope ps(ZEDO_PROF_HIDDEN, st); a.clk = ps.clk; e = launch_layer(a, EPI_GN_SILU_RES, st); a.clk = nullptr; }
            }
        }
        return e;
    }
    // post_dense, ending either in eps -> eps_out (EPI_BIAS, zedo_score_eps) or in the SDE update of xpad (EPI_SDE) [+ nr]
    hipError_t post(bool sde, float sa, float sc, float *eps_out, const NextReproj &nr) const {
        ProfScope ps(ZEDO_PROF_POST, st);
        if (f16()) return launch_layer16(post_args16(sde, sa, sc, eps_out, nr), sde ? EPI_SDE : EPI_BIAS, st);
        return launch_layer(post_args(sde, sa, sc, eps_out, nr), sde ? EPI_SDE : EPI_BIAS, st);
    }
    hipError_t all(const float *tb, bool sde, float sa, float sc, float *eps_out, const NextReproj &nr = NextReproj()) const {
        hipError_t e = pre(tb);
        if (e == hipSuccess) e = hidden(tb);
        if (e == hipSuccess) e = post(sde, sa, sc, eps_out, nr);
        return e;
    }
};

struct Ws { float *xpad, *h, *h1; size_t rows; };
static Ws carve(void *ws, int B) {
    Ws r; r.rows = ws_rows(B);
    r.xpad = (float *)ws; r.h = r.xpad + r.rows * XLD; r.h1 = r.h + r.rows * HID;
    return r;
}

static int step_common(const zedo_weights_t *w, const zedo_schedule_t *s, int step, const float *d_x_in, float *d_out,
                       bool sde, int B, void *ws, size_t ws_bytes, hipStream_t st) {
    if (!w || !s || !d_x_in || !d_out || !ws || B < 1 || step < 0 || step >= s->S) return ZEDO_E_BADARG;
    if (ws_bytes < zedo_workspace_bytes(B)) return ZEDO_E_WORKSPACE;
    const size_t cap = chunk_rows_cap();
    const float *tb = s->d_tbias + (size_t)step * NLAYER * HID;
    for (size_t r0 = 0; r0 < (size_t)B; r0 += cap) {
        const int Bc = (int)std::min(cap, (size_t)B - r0), Bp = round_up(Bc, BATCH_PAD);
        Ws k = carve(ws, B);
        HIPCHK(launch_pack_rows(d_x_in + r0 * w->J3, k.xpad, Bc, Bp, w->J3, st));
        const Net net{w, k.xpad, k.h, k.h1, Bp, (int)k.rows, st};
        if (sde) {
            HIPCHK(net.all(tb, true, s->a[step], s->c[step], nullptr));
            HIPCHK(launch_unpack_rows(k.xpad, d_out + r0 * w->J3, Bc, w->J3, st));
        } else {
            // eps lands in h1's first XLD columns region: reuse h1 as [Bp][XLD]
            HIPCHK(net.all(tb, false, 0.f, 0.f, k.h1));
            HIPCHK(launch_unpack_rows(k.h1, d_out + r0 * w->J3, Bc, w->J3, st));
        }
    }
    return ZEDO_OK;
}

extern "C" int zedo_score_eps(const zedo_weights_t *w, const zedo_schedule_t *s, int step, const float *d_x, float *d_eps,
                              int B, void *d_workspace, size_t workspace_bytes, void *stream) {
    return step_common(w, s, step, d_x, d_eps, false, B, d_workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int zedo_sde_step(const zedo_weights_t *w, const zedo_schedule_t *s, int step, float *d_x, int B,
                             void *d_workspace, size_t workspace_bytes, void *stream) {
    return step_common(w, s, step, d_x, d_x, true, B, d_workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int zedo_oil_run(const zedo_weights_t *w, const zedo_schedule_t *s, float *d_x, const float *d_geom, float *d_T,
                            int step_begin, int step_end, int switch_step, int B, int N, long long row_offset,
                            void *d_workspace, size_t workspace_bytes, void *stream) {
    if (!w || !s || !d_x || !d_geom || !d_T || !d_workspace || B < 1 || N < 1 || row_offset < 0) return ZEDO_E_BADARG;
    if (step_begin < 0 || step_end > s->S || step_begin > step_end || w->J3 != 51) return ZEDO_E_BADARG;
    if (workspace_bytes < zedo_workspace_bytes(B)) return ZEDO_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
#ifdef ZEDO_MUT_SWITCH      // tools/mutation_check.py only: the least-squares T starts one iteration late
    switch_step += 1;
#endif
    const size_t cap = chunk_rows_cap();
    for (size_t r0 = 0; r0 < (size_t)B; r0 += cap) {
        const int Bc = (int)std::min(cap, (size_t)B - r0), Bp = round_up(Bc, BATCH_PAD);
        Ws k = carve(d_workspace, B);
        HIPCHK(launch_pack_rows(d_x + r0 * w->J3, k.xpad, Bc, Bp, w->J3, st));
        // gradient_field_gen + "denoise_x += joint_gradient" (run/opt_main.py:203-208) of the first iteration; the
        // correction of every later iteration i+1 rides in the epilogue of iteration i's post_dense launch
        // (ZEDO_UNFUSED_REPROJ=1: one launch per iteration, the A/B and parity reference)
        static const bool unfused = getenv("ZEDO_UNFUSED_REPROJ") != nullptr;
        const Net net{w, k.xpad, k.h, k.h1, Bp, (int)k.rows, st};
        for (int i = step_begin; i < step_end; ++i) {
            if (i == step_begin || unfused) {
                ProfScope ps(ZEDO_PROF_REPROJ, st);
                HIPCHK(launch_reproj_step_padded(k.xpad, d_geom, d_T + r0 * 3, i >= switch_step, Bc, N,
                                                 row_offset + (long long)r0, st));
            }
            NextReproj nr;
            if (i + 1 < step_end && !unfused) {
                nr.geom = d_geom; nr.T = d_T + r0 * 3; nr.solve = (i + 1) >= switch_step; nr.B = Bc; nr.N = N;
                nr.row0 = row_offset + (long long)r0;
            }
            // sampling_fn(...) (run/opt_main.py:210-218) -> x = a_i x + c_i eps(x, t_i)  [+ the next correction]
            HIPCHK(net.all(s->d_tbias + (size_t)i * NLAYER * HID, true, s->a[i], s->c[i], nullptr, nr));
        }
        HIPCHK(launch_unpack_rows(k.xpad, d_x + r0 * w->J3, Bc, w->J3, st));
    }
    return ZEDO_OK;
}

extern "C" int zedo_ipo_fit_resume(const float *d_x0, const float *d_uv, const float *d_K, const int *h_keylist, int k,
                                   int axes_mask, float ipo_T, float min_scale, float max_scale, int iters,
                                   double normaliser, float *d_R, float *d_T, float *d_q, float *d_scale,
                                   float *d_state, int it_begin, int B, int H, int N, int J, long long row_offset,
                                   void *stream) {
    if (!d_x0 || !d_uv || !d_K || !h_keylist || !d_R || !d_T || B < 1 || H < 1 || N < 1 || J < 1 || k < 1 || k > 17 ||
        iters < 0 || it_begin < 0 || (it_begin > 0 && !d_state) || !(normaliser > 0) || row_offset < 0)
        return ZEDO_E_BADARG;
    if (row_offset + (long long)B > (long long)H * N) return ZEDO_E_BADARG;   // x0[h] would be read out of bounds
    for (int i = 0; i < k; ++i)
        if (h_keylist[i] < 0 || h_keylist[i] >= J) return ZEDO_E_BADARG;
    HIPCHK(launch_ipo_fit(d_x0, d_uv, d_K, h_keylist, k, axes_mask, ipo_T, min_scale, max_scale, iters, normaliser, d_R, d_T,
                          d_q, d_scale, d_state, it_begin, B, N, J, row_offset, (hipStream_t)stream));
    return ZEDO_OK;
}

extern "C" int zedo_ipo_fit(const float *d_x0, const float *d_uv, const float *d_K, const int *h_keylist, int k,
                            int axes_mask, float ipo_T, float min_scale, float max_scale, int iters, double normaliser,
                            float *d_R, float *d_T, float *d_q, float *d_scale, int B, int H, int N, int J,
                            long long row_offset, void *stream) {
    return zedo_ipo_fit_resume(d_x0, d_uv, d_K, h_keylist, k, axes_mask, ipo_T, min_scale, max_scale, iters, normaliser,
                               d_R, d_T, d_q, d_scale, nullptr, 0, B, H, N, J, row_offset, stream);
}

extern "C" int zedo_rotate_init(const float *d_x0, const float *d_R, float *d_x, int B, int H, int N, int J,
                                long long row_offset, void *stream) {
    if (!d_x0 || !d_R || !d_x || B < 1 || H < 1 || N < 1 || J < 1 || row_offset < 0) return ZEDO_E_BADARG;
    if (row_offset + (long long)B > (long long)H * N) return ZEDO_E_BADARG;
    HIPCHK(launch_rotate_init(d_x0, d_R, d_x, B, N, J, row_offset, (hipStream_t)stream));
    return ZEDO_OK;
}

extern "C" int zedo_pose_min(const double *d_err, int B, int N, long long row_offset, double *d_best, int *d_best_h,
                             void *stream) {
    if (!d_err || !d_best || !d_best_h || B < 1 || N < 1 || row_offset < 0) return ZEDO_E_BADARG;
    HIPCHK(launch_pose_min(d_err, B, N, row_offset, d_best, d_best_h, (hipStream_t)stream));
    return ZEDO_OK;
}

extern "C" int zedo_min_mpjpe(const float *d_pred, const double *d_gt, int B, int N, int J, long long row_offset,
                              int procrustes, double *d_err, double *d_best, int *d_best_h, void *stream) {
    if (!d_pred || !d_gt || !d_err || !d_best || !d_best_h || B < 1 || N < 1 || J < 1 || row_offset < 0)
        return ZEDO_E_BADARG;
    HIPCHK(launch_min_mpjpe(d_pred, d_gt, B, N, J, row_offset, procrustes, d_err, d_best, d_best_h, (hipStream_t)stream));
    return ZEDO_OK;
}
