// Developer tool: A/B of the large-batch recurrence kernels through the C ABI, without Python.
//   hipcc -O2 -o gpurun_out/gru_ab profiles/gru_ab.cpp -Iinclude -Lmtad-gat-pytorch_amd -lmtadgat -Wl,-rpath,$PWD/mtad-gat-pytorch_amd
//   gru_ab [windows] [iters]
// For a list of model shapes: random parameters (reference initialiser ranges), uniform [0,1) windows, the full forward
// with gru_kernel = 1 (tile-major k_gru) and = 2 (chunk-major k_gru_cm); prints the largest output differences and the
// per-family kernel times of both.  The tile-major kernel is the trusted side (it passes the parity suite).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "mtadgat.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(2); } } while (0)
#define MK(x) do { int r_ = (x); if (r_ != 0) { std::printf("mtadgat error %d (%s) at %s:%d\n", r_, mtadgat_last_error(), __FILE__, __LINE__); return 1; } } while (0)

struct Shape { const char* name; int F, W, out, k, H, fcn, fch, Hr, gl, rl; };

static std::vector<float> rnd(std::mt19937& g, size_t n, float a) {
    std::uniform_real_distribution<float> d(-a, a);
    std::vector<float> v(n);
    for (auto& x : v) x = d(g);
    return v;
}

static float g_sih = 1.f, g_shh = 1.f, g_sb = 1.f;
static int run_shape(const Shape& sh, long n, int iters, int kernel_b) {
    mtadgat_config c{};
    c.n_features = sh.F; c.window_size = sh.W; c.out_dim = sh.out; c.kernel_size = sh.k; c.use_gatv2 = 1;
    c.feat_embed = 2 * sh.W; c.time_embed = 2 * sh.F; c.gru_n_layers = sh.gl; c.gru_hid_dim = sh.H;
    c.forecast_n_linear = sh.fcn; c.forecast_hid_dim = sh.fch; c.recon_n_layers = sh.rl; c.recon_hid_dim = sh.Hr; c.alpha = 0.2f;
    std::mt19937 g(1234);
    const int F = sh.F, W = sh.W, H = sh.H, Hr = sh.Hr;
    std::vector<std::vector<float>> keep;
    auto mk = [&](size_t cnt, float a) { keep.push_back(rnd(g, cnt, a)); return keep.back().data(); };
    mtadgat_params p{};
    p.conv_weight = mk((size_t)F * F * sh.k, 1.f / std::sqrt((float)F * sh.k)); p.conv_bias = mk(F, 1.f / std::sqrt((float)F * sh.k));
    p.feat_lin_weight = mk((size_t)2 * W * 2 * W, 1.f / std::sqrt(2.f * W)); p.feat_lin_bias = mk(2 * W, 1.f / std::sqrt(2.f * W));
    p.feat_a = mk(2 * W, 0.2f); p.feat_bias = mk((size_t)F * F, 0.5f);
    p.temp_lin_weight = mk((size_t)2 * F * 2 * F, 1.f / std::sqrt(2.f * F)); p.temp_lin_bias = mk(2 * F, 1.f / std::sqrt(2.f * F));
    p.temp_a = mk(2 * F, 0.2f); p.temp_bias = mk((size_t)W * W, 0.5f);
    for (int l = 0; l < sh.gl; ++l) {
        const int in = l == 0 ? 3 * F : H;
        const float a = 1.f / std::sqrt((float)H);
        p.gru_w_ih[l] = mk((size_t)3 * H * in, a * g_sih); p.gru_w_hh[l] = mk((size_t)3 * H * H, a * g_shh);
        p.gru_b_ih[l] = mk(3 * H, a * g_sb); p.gru_b_hh[l] = mk(3 * H, a * g_sb);
    }
    for (int i = 0; i < sh.fcn; ++i) {
        const int in = i == 0 ? H : sh.fch, out = i == sh.fcn - 1 ? sh.out : sh.fch;
        p.fc_weight[i] = mk((size_t)out * in, 1.f / std::sqrt((float)in)); p.fc_bias[i] = mk(out, 1.f / std::sqrt((float)in));
    }
    for (int l = 0; l < sh.rl; ++l) {
        const int in = l == 0 ? H : Hr;
        const float a = 1.f / std::sqrt((float)Hr);
        p.rec_w_ih[l] = mk((size_t)3 * Hr * in, a); p.rec_w_hh[l] = mk((size_t)3 * Hr * Hr, a);
        p.rec_b_ih[l] = mk(3 * Hr, a); p.rec_b_hh[l] = mk(3 * Hr, a);
    }
    p.rec_fc_weight = mk((size_t)sh.out * Hr, 1.f / std::sqrt((float)Hr)); p.rec_fc_bias = mk(sh.out, 1.f / std::sqrt((float)Hr));

    mtadgat_handle h = nullptr;
    MK(mtadgat_create(&c, &h));
    MK(mtadgat_set_precision(h, 2));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    MK(mtadgat_load_weights(h, &p, s));

    std::vector<float> xh((size_t)n * W * F);
    {
        std::uniform_real_distribution<float> d(0.f, 1.f);
        for (auto& v : xh) v = d(g);
    }
    float *x, *preds[2], *recons[2], *hend[2];
    void* ws;
    CK(hipMalloc(&x, xh.size() * 4));
    CK(hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    const size_t wsb = mtadgat_workspace_bytes(h, n);
    CK(hipMalloc(&ws, wsb));
    for (int k = 0; k < 2; ++k) {
        CK(hipMalloc(&preds[k], (size_t)n * sh.out * 4));
        CK(hipMalloc(&recons[k], (size_t)n * W * sh.out * 4));
        CK(hipMalloc(&hend[k], (size_t)n * H * 4));
        CK(hipMemset(preds[k], 0xff, (size_t)n * sh.out * 4));
        CK(hipMemset(recons[k], 0xff, (size_t)n * W * sh.out * 4));
        CK(hipMemset(hend[k], 0xff, (size_t)n * H * 4));
    }
    const int kernels[2] = {1, kernel_b};
    double ms[2][MTADGAT_PROFILE_SLOTS];
    float wall[2];
    for (int k = 0; k < 2; ++k) {
        MK(mtadgat_set_option(h, "gru_kernel", kernels[k]));
        MK(mtadgat_profile_enable(h, 0));
        MK(mtadgat_forward(h, x, n, preds[k], recons[k], hend[k], ws, wsb, s));      // warm-up (and the compared result)
        CK(hipStreamSynchronize(s));
        MK(mtadgat_profile_enable(h, 1));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int it = 0; it < iters; ++it) MK(mtadgat_forward(h, x, n, preds[k], recons[k], hend[k], ws, wsb, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&wall[k], e0, e1));
        wall[k] /= iters;
        int64_t cnt[MTADGAT_PROFILE_SLOTS];
        MK(mtadgat_profile_read(h, ms[k], cnt));
        for (int j = 0; j < MTADGAT_PROFILE_SLOTS; ++j) ms[k][j] /= iters;
    }
    auto cmp = [&](const float* a, const float* b, size_t cnt, const char* what) {
        std::vector<float> ha(cnt), hb(cnt);
        CK(hipMemcpy(ha.data(), a, cnt * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), b, cnt * 4, hipMemcpyDeviceToHost));
        double md = 0, ma = 0;
        size_t bad = 0, at = 0;
        for (size_t i = 0; i < cnt; ++i) {
            const double d = std::fabs((double)ha[i] - (double)hb[i]);
            if (!(d <= 1e30)) { ++bad; continue; }
            if (d > md) { md = d; at = i; }
            ma = std::fmax(ma, std::fabs((double)ha[i]));
        }
        std::printf("    %-7s max|a-b| = %.3e at %zu (a=%.6f b=%.6f)  max|a| = %.3f  non-finite/nan diffs = %zu of %zu\n", what, md, at,
                    cnt ? ha[at] : 0.f, cnt ? hb[at] : 0.f, ma, bad, cnt);
    };
    std::printf("shape %s: F=%d W=%d out=%d H=%d Hr=%d layers %d/%d, %ld windows, kernel A=1 (tile-major) B=%d\n", sh.name, F, W, sh.out, H, Hr, sh.gl,
                sh.rl, n, kernel_b);
    cmp(hend[0], hend[1], (size_t)n * H, "h_end");
    if (std::getenv("AB_DUMP")) {
        std::vector<float> ha(2 * H), hb(2 * H);
        CK(hipMemcpy(ha.data(), hend[0], 2 * H * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), hend[1], 2 * H * 4, hipMemcpyDeviceToHost));
        for (int w = 0; w < 2; ++w)
            for (int u = 0; u < H; ++u) std::printf("      win %d unit %3d  a % .6f  b % .6f\n", w, u, ha[w * H + u], hb[w * H + u]);
    }
    cmp(preds[0], preds[1], (size_t)n * sh.out, "preds");
    cmp(recons[0], recons[1], (size_t)n * W * sh.out, "recons");
    for (int k = 0; k < 2; ++k) {
        std::printf("    kernel %d: forward %.3f ms  |", kernels[k], wall[k]);
        for (int j = 0; j < MTADGAT_PROFILE_SLOTS; ++j) std::printf(" %s %.3f", mtadgat_profile_name(j), ms[k][j]);
        std::printf("\n");
    }
    std::fflush(stdout);
    for (int k = 0; k < 2; ++k) { CK(hipFree(preds[k])); CK(hipFree(recons[k])); CK(hipFree(hend[k])); }
    CK(hipFree(x)); CK(hipFree(ws));
    mtadgat_destroy(h);
    CK(hipStreamDestroy(s));
    return 0;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? std::atol(argv[1]) : 65536;
    const int iters = argc > 2 ? std::atoi(argv[2]) : 3;
    const int kb = argc > 3 ? std::atoi(argv[3]) : 2;
    const char* only = argc > 4 && std::strcmp(argv[4], "all") != 0 ? argv[4] : nullptr;
    if (argc > 5) g_sih = (float)std::atof(argv[5]);
    if (argc > 6) g_shh = (float)std::atof(argv[6]);
    if (argc > 7) g_sb = (float)std::atof(argv[7]);
    std::printf("GRU weight scales: w_ih x %g, w_hh x %g, biases x %g\n", g_sih, g_shh, g_sb);
    const Shape shapes[] = {
        {"msl", 55, 100, 1, 7, 150, 4, 150, 150, 1, 1},
        {"smd", 38, 100, 38, 7, 150, 4, 150, 150, 1, 1},
        {"smap", 25, 100, 1, 7, 150, 4, 150, 150, 1, 1},
        {"h100", 20, 40, 2, 5, 100, 2, 64, 100, 1, 1},       // NCG 4: two groups of two tiles
        {"h96", 20, 40, 1, 5, 96, 2, 64, 70, 1, 1},          // NCG 3 (GRU) and 3 (decoder, H = 70)
        {"h40x2", 13, 30, 3, 3, 40, 2, 36, 44, 2, 1},        // NCG 2, stacked GRU layers (second layer: row input without range guard)
    };
    int rc = 0;
    for (const Shape& sh : shapes) {
        if (only && std::strcmp(only, sh.name) != 0) continue;
        rc |= run_shape(sh, n, iters, kb);
    }
    return rc;
}
