// Host-side planning and weight packing: reference-format parameters -> the tile
// streams the gfx950 kernels consume (mtadgat_kernels.hip).  Pure host C++.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#include "mtadgat_host.h"

namespace mtadgat {

namespace {

size_t align64(size_t v) { return (v + 63) / 64 * 64; }

// tile format: [NT][Q][64 lanes][4]; lane (j = lane&31, g = lane>>5), element s holds
// M[32n + j][8q + 4g + s] (zero outside the matrix)
template <class Get>
void pack_tiles(float* out, int NT, int Q, Get get) {
    for (int n = 0; n < NT; ++n)
        for (int q = 0; q < Q; ++q) {
            float* o = out + ((size_t)n * Q + q) * 256;
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) o[lane * 4 + s] = get(32 * n + (lane & 31), 8 * q + 4 * (lane >> 5) + s);
        }
}

// GRU stream format: [NCG][Q][3 gates][64 lanes][4]
template <class Get>
void pack_gru_tiles(float* out, int NCG, int Q, Get get /*(gate,row,k)*/) {
    for (int c = 0; c < NCG; ++c)
        for (int q = 0; q < Q; ++q)
            for (int st = 0; st < 3; ++st) {
                float* o = out + (((size_t)c * Q + q) * 3 + st) * 256;
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) o[lane * 4 + s] = get(st, 32 * c + (lane & 31), 8 * q + 4 * (lane >> 5) + s);
            }
}

inline uint16_t f2bf(float f) {                      // round to nearest even, as v_cvt_pk_bf16_f32
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// feature index of element e of lane-half g inside a 16-feature bf16 chunk (mtadgat_device.h)
inline int bf_k(int g, int e) { return e < 4 ? 4 * g + e : 8 + 4 * g + (e - 4); }

// bf16 tile format: [NT][Q16][64 lanes][8 bf16]; lane (j, g) element e holds M[32n + j][16q + bf_k(g, e)]
template <class Get>
void pack_tiles_bf16(float* out, int NT, int Q, Get get) {
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
    for (int n = 0; n < NT; ++n)
        for (int q = 0; q < Q; ++q) {
            uint16_t* o = o16 + ((size_t)n * Q + q) * 512;
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) o[lane * 8 + e] = f2bf(get(32 * n + (lane & 31), 16 * q + bf_k(lane >> 5, e)));
        }
}

// bf16 GRU stream: [NCG][Q16][3 gates][64 lanes][8 bf16]
template <class Get>
void pack_gru_tiles_bf16(float* out, int NCG, int Q, Get get /*(gate,row,k)*/) {
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
    for (int c = 0; c < NCG; ++c)
        for (int q = 0; q < Q; ++q)
            for (int st = 0; st < 3; ++st) {
                uint16_t* o = o16 + (((size_t)c * Q + q) * 3 + st) * 512;
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) o[lane * 8 + e] = f2bf(get(st, 32 * c + (lane & 31), 16 * q + bf_k(lane >> 5, e)));
            }
}

}  // namespace

std::string validate_and_plan(Model& m) {
    const mtadgat_config& c = m.cfg;
    if (c.n_features < 1 || c.window_size < 1 || c.out_dim < 1) return "n_features, window_size and out_dim must be >= 1";
    if (c.kernel_size < 1 || (c.kernel_size & 1) == 0) return "kernel_size must be odd (the reference's ConvLayer shortens the window otherwise)";
    if (c.feat_embed < 1 || c.time_embed < 1) return "embedding dims must be >= 1";
    if (c.n_features > 512 || c.window_size > 512) return "n_features and window_size up to 512 nodes are supported";
    if (c.gru_n_layers < 1 || c.gru_n_layers > MTADGAT_MAX_LAYERS) return "gru_n_layers out of range";
    if (c.recon_n_layers < 1 || c.recon_n_layers > MTADGAT_MAX_LAYERS) return "recon_n_layers out of range";
    if (c.forecast_n_linear < 1 || c.forecast_n_linear > MTADGAT_MAX_LAYERS) return "forecast_n_linear out of range";
    if (c.gru_hid_dim < 1 || c.gru_hid_dim > 256 || c.recon_hid_dim < 1 || c.recon_hid_dim > 256)
        return "GRU hidden sizes 1..256 are supported";
    if (c.forecast_hid_dim < 1) return "forecast_hid_dim must be >= 1";

    m.F = c.n_features;
    m.W = c.window_size;
    m.Fp = round_up(m.F, 8);
    m.Wp = round_up(m.W, 8);
    m.Dp = round_up(3 * m.F, 8);
    m.taps = c.kernel_size;
    m.pad = (c.kernel_size - 1) / 2;

    size_t off = 0;
    auto take = [&](size_t n) {
        size_t o = off;
        off = align64(off + n);
        return o;
    };
    // conv
    m.convNT = (m.F + 31) / 32;
    {
        const int Q = m.taps * m.Fp / 8;
        m.conv_w_off = take((size_t)m.convNT * Q * 256);
        m.conv_b_off = take((size_t)m.convNT * 32);
        m.conv_wraw_off = take((size_t)m.F * m.F * m.taps);
        m.Fp16 = round_up(m.F, 16);
        m.conv_w16_off = take((size_t)m.convNT * (m.taps * m.Fp16 / 16) * 256);
        m.conv_wf16_off = take((size_t)m.convNT * (m.taps * m.Fp16 / 8) * 256);
        m.conv_w3_off = take((size_t)m.convNT * (m.taps * m.Fp16 / 16) * 3 * 256);
        m.conv_w2h_off = take((size_t)m.convNT * (m.taps * m.Fp16 / 16) * 2 * 256);
        m.conv_scale_off = take(4);
    }
    // GAT layers
    auto plan_gat = [&](GatPlan& g, int K, int D, int E) {
        g.K = K; g.D = D; g.E = E;
        const int ptcap = c.use_gatv2 ? round_up(E, 8) + 8 : 0;
        g.ldl = round_up(ptcap + 1, 32);
        g.rt_rows = g.ldl;
        g.Kp = round_up(K, 4);
        g.NT_L = g.ldl / 32;
        g.NT = 2 * g.NT_L;
        g.PT = g.P8 = 0;
        attend_plan(K, &g.rows_per_blk, &g.nblk, &g.IB);
        // fused plan: one workgroup (f_nw waves) per window, tiles of the layer resident in LDS.
        // A wave owns 4*IBL query rows, a lane JPL key nodes (16*JPL >= K); see k_gat.
        g.fused = false;
        g.Q = (D + 7) / 8;
        if (K <= 128 && D <= 128) {
            const int ibl = 4, ibw = 4 * ibl;
            const int nwa = (K + ibw - 1) / ibw;                 // waves that own query rows
            const int ntask = 2 * ((K + 31) / 32);               // projection tiles per part
            int nw = nwa;
            if (ntask > nwa) nw = std::min(8, std::max(nwa, std::min(ntask, round_up(nwa, 4))));   // extra waves only project
            const int rows = nwa * ibw;
            const int qf = (D + 8) / 8;                          // chunks incl. the bias row D (times the ones column of Vs)
            const int vld = 8 * qf + 4;
            const size_t lr = round_up((int)std::max((size_t)(rows + K) * 34, (size_t)rows * 68), 4);
            const size_t bytes = ((size_t)round_up(K, 16) * vld + lr) * sizeof(float);
            if (bytes <= 80 * 1024) {
                g.fused = true; g.f_nw = nw; g.f_IBL = ibl; g.f_JPL = (K + 15) / 16; g.f_RJ = 16; g.f_vld = vld; g.f_lr = (int)lr;
                // 8 lanes along the key axis when that pads the keys less (K = 55: 56 instead of 64)
                if (K <= 56 && ((K + 7) / 8) % 2 == 1) { g.f_RJ = 8; g.f_IBL = 2; g.f_JPL = (K + 7) / 8; }
                g.f_lds_bytes = bytes;
                g.Q = qf;
            }
        }
        g.w_off = take((size_t)g.NT * g.Q * 256);
        g.b_off = take((size_t)g.NT * 32);
        g.bias_off = take((size_t)K * K);
        g.ord_off = take(16);
        g.PTcap = ptcap;
        g.Q16 = g.fused ? (D + 1 + 15) / 16 : 0;
        if (g.fused) {      // fp16-piece build: pitch 4 x odd halfs (conflict-free 8-byte operand reads of 32 consecutive nodes)
            g.fh_IBL = g.f_IBL; g.fh_JPL = g.f_JPL; g.fh_RJ = g.f_RJ;
            // (8 lanes along the keys also where that pads them less than 16 -- K = 100: 104 instead of 112 keys -- costs as much in
            // LDS operand reads as it saves in pair instructions: 10.15 vs 10.05 ms for the two layers; measurement hook)
            if (const char* e_ = getenv("MTADGAT_GATH_RJ8"))
                if (atoi(e_) && g.f_RJ == 16 && round_up(K, 8) < round_up(K, 16) && (K + 7) / 8 <= 15) { g.fh_RJ = 8; g.fh_IBL = 2; g.fh_JPL = (K + 7) / 8; }
            {   // row blocks of the owning waves: as few padded rows as 16-row and short (12- / 8-row) blocks allow
                const int shortrows = 16 - 64 / g.fh_RJ;
                int best_rows = 1 << 30, bf = 0, bs = 0;
                for (int nf = 0; nf <= g.f_nw; ++nf)
                    for (int ns = 0; nf + ns <= g.f_nw; ++ns) {
                        const int rows = 16 * nf + shortrows * ns;
                        if (rows >= K && (rows < best_rows || (rows == best_rows && nf + ns < bf + bs))) { best_rows = rows; bf = nf; bs = ns; }
                    }
                g.fh_full = bf; g.fh_short = bs;
            }
            // piece pitch: the smallest 4 x odd halfs that holds the D + 1 features (conflict-free 8-byte operand reads of 32
            // consecutive nodes; the last chunk of a row may read on into the next row: finite values against zero weights)
            g.fh_vld = (D + 1 + 3) & ~3;
            if ((g.fh_vld >> 2) % 2 == 0) g.fh_vld += 4;
            if (g.fh_vld > 16 * g.Q16 + 4) g.fh_vld = 16 * g.Q16 + 4;
            const int orows = (g.fh_full + g.fh_short) * 16;     // rows of L' the owning waves address
            g.fh_lr = (int)round_up((int)std::max((size_t)(orows + K) * 34, (size_t)orows * 36), 4);
            g.fh_lds_bytes = (size_t)g.fh_lr * sizeof(float) + (size_t)2 * ((K + 1) * g.fh_vld + 16) * 2;
            {   // run-ahead projection (k_gath): a second L' / R' buffer when the workgroup has a wave that owns no query rows, the
                // part's weight words fit the projector's registers (Q <= 4) and the LDS does not cost a resident workgroup
                // (160 KB per CU; 16 waves per CU at the kernel's 128 registers).  Two buffers only fit with K rows of L' each.
                g.fh_lr_buf = 0;
                const size_t buf = (size_t)round_up((K + K) * 34, 4);
                const size_t pieces = (size_t)2 * ((K + 1) * g.fh_vld + 16) * 2;
                const size_t total = 2 * buf * sizeof(float) + pieces;
                auto resident = [&](size_t bytes) { return std::min<size_t>((size_t)160 * 1024 / bytes, (size_t)16 / g.f_nw); };
                if (g.f_nw == 8 && g.fh_full + g.fh_short <= 7 && g.Q16 <= 4 && ptcap >= 32 && (size_t)orows * 36 <= 2 * buf &&
                    total <= 160 * 1024 && resident(total) == resident(g.fh_lds_bytes) && !getenv("MTADGAT_GATH_NOAHEAD")) {
                    g.fh_lr_buf = (int)buf; g.fh_lr = (int)(2 * buf); g.fh_lds_bytes = total;
                }
            }
            // measurement hook (profiles/gath_timeline.py): more LDS per workgroup = fewer resident workgroups per CU
            if (const char* e_ = getenv("MTADGAT_GATH_PADLDS")) g.fh_lds_bytes = std::max<size_t>(g.fh_lds_bytes, (size_t)atoi(e_) * 1024);
        }
        g.w16_off = take((size_t)g.NT * g.Q16 * 256);
        g.w3_off = take((size_t)g.NT * g.Q16 * 3 * 256);
        g.w2h_off = take((size_t)g.NT * g.Q16 * 2 * 256);
        g.gscale_off = take(4);
        g.uQ16 = g.fused ? 0 : (g.Q + 1) / 2;
        g.uw3_off = take((size_t)g.NT * g.uQ16 * 3 * 256);
    };
    plan_gat(m.feat, m.F, m.W, c.feat_embed);
    plan_gat(m.temp, m.W, m.F, c.time_embed);
    auto plan16 = [&](GruPlan& g) {
        const int Hp16 = round_up(g.H, 16);
        if (Hp16 > 160 || (g.xmode == 1 && g.Qx != 1)) return;       // weights must fit the wave's registers; 8 folded columns
        g.has16 = true; g.KS16 = (g.H + 3) / 4; g.NT16 = Hp16 / 16;
        g.g16_off = take((size_t)g.NT16 * 3 * g.KS16 * 64);
        g.g16T_off = take((size_t)g.NT16 * 3 * g.KS16 * 64);
        if (g.xmode == 1) g.fold_off = take((size_t)m.W * 3 * g.Hp * 8);
        g.g1_off = take((size_t)g1_waves(g.H, g.Hp, false) * 16 * g1_ksm(g.H) * 64);
        g.g1T_off = take((size_t)g1_waves(g.H, g.Hp, true) * 16 * g1_ksm(g.H) * 64);
    };
    // GRU stack
    m.gru.assign(c.gru_n_layers, GruPlan());
    for (int l = 0; l < c.gru_n_layers; ++l) {
        GruPlan& g = m.gru[l];
        g.in_dim = (l == 0) ? 3 * m.F : c.gru_hid_dim;
        g.H = c.gru_hid_dim;
        g.Hp = round_up(g.H, 32);
        g.NCG = g.Hp / 32;
        g.Qx = (g.in_dim + 7) / 8;
        g.Qxp = round_up(g.Qx, 3);
        g.xmode = 0;
        g.wx_off = take((size_t)g.NCG * g.Qxp * 3 * 256);
        g.wh_off = take((size_t)g.NCG * (4 * g.NCG + 2) * 3 * 256);   // 2 zero chunks per tile: k_gru_split's ring padding
        g.b_off = take((size_t)4 * g.Hp);
        g.Qxp16 = round_up((g.Qx + 1) / 2, 6);       // whole turns of the bf16 weight ring (6 stages; k_gru_split: 3)
        g.wx16_off = take((size_t)g.NCG * g.Qxp16 * 3 * 256);
        g.wh16_off = take((size_t)g.NCG * (2 * g.NCG + 2) * 3 * 256);
        g.wx3_off = take((size_t)g.NCG * g.Qxp16 * 9 * 256);
        g.wh3_off = take((size_t)g.NCG * (2 * g.NCG + 2) * 6 * 256 + 3 * 256);
        g.scale_off = take(4);
        // layer 0 reads h_cat = [conv output (relu: unbounded) | h_feat | h_temp (sigmoids)]: the chunks that touch the
        // first F features keep three bf16 pieces; later layers read a previous layer's state, |h| <= 1
        g.qb3 = l == 0 ? std::min(g.Qxp16, round_up((m.F + 15) / 16, 2)) : 0;
        if (l == 0) g.wx2_off = take((size_t)g.NCG * g.Qxp16 * 6 * 256 + 3 * 256);
        g.wxq_off = take((size_t)g.NCG * g.Qxp16 * 6 * 256 + 3 * 256);
        if (l == 0) {
            g.has_xproj = true;
            g.xproj.in_dim = g.in_dim; g.xproj.out_dim = 3 * g.Hp;
            g.xproj.NT = 3 * g.Hp / 32; g.xproj.Q = (g.in_dim + 7) / 8;
            g.xproj.w_off = take((size_t)g.xproj.NT * g.xproj.Q * 256);
            g.xproj.b_off = take((size_t)3 * g.Hp);
            g.xproj.Q16 = (g.in_dim + 15) / 16;
            g.xproj.w3_off = take((size_t)g.xproj.NT * g.xproj.Q16 * 3 * 256);
            plan16(g);
        }
    }
    // forecasting head
    m.fc.assign(c.forecast_n_linear, LinPlan());
    for (int i = 0; i < c.forecast_n_linear; ++i) {
        LinPlan& p = m.fc[i];
        p.in_dim = (i == 0) ? c.gru_hid_dim : c.forecast_hid_dim;
        p.out_dim = (i == c.forecast_n_linear - 1) ? c.out_dim : c.forecast_hid_dim;
        p.NT = (p.out_dim + 31) / 32;
        p.Q = (p.in_dim + 7) / 8;
        p.w_off = take((size_t)p.NT * p.Q * 256);
        p.b_off = take((size_t)p.NT * 32);
    }
    // reconstruction decoder
    m.rec.assign(c.recon_n_layers, GruPlan());
    for (int l = 0; l < c.recon_n_layers; ++l) {
        GruPlan& g = m.rec[l];
        g.H = c.recon_hid_dim;
        g.Hp = round_up(g.H, 32);
        g.NCG = g.Hp / 32;
        if (l == 0) {
            // decoder input (t, j) = h_end[(t*H + j) / W]  (reference modules.py:279)
            const int Hin = c.gru_hid_dim, T = m.W;
            int nm = 1;
            for (int t = 0; t < T; ++t) {
                const int lo = (int)(((long)t * Hin) / T), hi = (int)(((long)t * Hin + Hin - 1) / T);
                if (hi - lo + 1 > nm) nm = hi - lo + 1;
            }
            g.in_dim = Hin;
            g.xmode = 1;
            g.Qx = (nm + 7) / 8;
            g.Qxp = g.Qx == 1 ? 1 : round_up(g.Qx, 3);
            g.wx_off = take((size_t)T * g.NCG * g.Qxp * 3 * 256);
            g.m0_off = take((size_t)T);
        } else {
            g.in_dim = c.recon_hid_dim;
            g.xmode = 0;
            g.Qx = (g.in_dim + 7) / 8;
            g.Qxp = round_up(g.Qx, 3);
            g.wx_off = take((size_t)g.NCG * g.Qxp * 3 * 256);
        }
        g.wh_off = take((size_t)g.NCG * (4 * g.NCG + 2) * 3 * 256);   // 2 zero chunks per tile: k_gru_split's ring padding
        g.b_off = take((size_t)4 * g.Hp);
        g.Qxp16 = g.xmode == 1 ? (g.Qx == 1 ? 1 : round_up((g.Qx + 1) / 2, 6)) : round_up((g.Qx + 1) / 2, 6);
        g.wx16_off = take((size_t)(g.xmode == 1 ? m.W : 1) * g.NCG * g.Qxp16 * 3 * 256);
        g.wh16_off = take((size_t)g.NCG * (2 * g.NCG + 2) * 3 * 256);
        g.wx3_off = take((size_t)(g.xmode == 1 ? m.W : 1) * g.NCG * g.Qxp16 * 9 * 256);
        g.wh3_off = take((size_t)g.NCG * (2 * g.NCG + 2) * 6 * 256 + 3 * 256);
        g.scale_off = take(4);
        if (g.xmode == 0) g.wxq_off = take((size_t)g.NCG * g.Qxp16 * 6 * 256 + 3 * 256);
        if (l == 0) plan16(g);
    }
    {
        LinPlan& p = m.rec_fc;
        p.in_dim = c.recon_hid_dim;
        p.out_dim = c.out_dim;
        p.NT = (p.out_dim + 31) / 32;
        p.Q = m.rec.back().Hp / 8;
        p.w_off = take((size_t)p.NT * p.Q * 256);
        p.b_off = take((size_t)p.NT * 32);
        p.Q16 = (p.Q + 1) / 2;
        p.w3_off = take((size_t)p.NT * p.Q16 * 3 * 256);
    }

    // ---- backward (training) plans: transposed packs, un-scaled attention projections, gradient index maps
    {
        BwdPlan& b = m.bw;
        b = BwdPlan();
        auto lint = [&](LinTPlan& p, int kdim, int outdim) {
            p.in_dim = kdim; p.out_dim = outdim;
            p.NT = (outdim + 31) / 32; p.Q = (kdim + 7) / 8;
            p.w_off = take((size_t)p.NT * p.Q * 256);
            p.Q16 = (kdim + 15) / 16;
            p.w3_off = take((size_t)p.NT * p.Q16 * 3 * 256);
        };
        auto wg = [&](WgradPlan& p, int M, int N, bool bias) {
            p.M = M; p.N = N; p.has_bias = bias;
            p.Mp = round_up(M, 32); p.Np = round_up(N + 1, 32);
            p.rowW_off = take((size_t)p.Mp);
            p.col_off = take((size_t)p.Np);
            p.rowB_off = take((size_t)p.Mp);
        };
        // flat gradient layout: the reference's parameters in the order of mtadgat_params
        GradLayout& gl = b.gl;
        int64_t go = 0;
        auto gtake = [&](int64_t n) { int64_t o = go; go += n; return o; };
        gl.conv_w = gtake((int64_t)m.F * m.F * m.taps); gl.conv_b = gtake(m.F);
        for (int which = 0; which < 2; ++which) {       // 0 feature, 1 temporal  (mtadgat_params order)
            const GatPlan& g = which == 0 ? m.feat : m.temp;
            const int lin_in = c.use_gatv2 ? 2 * g.D : g.D;
            gl.lin_w[which] = gtake((int64_t)g.E * lin_in); gl.lin_b[which] = gtake(g.E);
            gl.a[which] = gtake(c.use_gatv2 ? g.E : 2 * g.E); gl.bias[which] = gtake((int64_t)g.K * g.K);
        }
        gl.gru_wih.clear(); gl.gru_whh.clear(); gl.gru_bih.clear(); gl.gru_bhh.clear();
        for (int l = 0; l < c.gru_n_layers; ++l) {
            const int in = m.gru[l].in_dim, H = m.gru[l].H;
            gl.gru_wih.push_back(gtake((int64_t)3 * H * in)); gl.gru_whh.push_back(gtake((int64_t)3 * H * H));
            gl.gru_bih.push_back(gtake(3 * H)); gl.gru_bhh.push_back(gtake(3 * H));
        }
        gl.fc_w.clear(); gl.fc_b.clear();
        for (const LinPlan& p : m.fc) { gl.fc_w.push_back(gtake((int64_t)p.out_dim * p.in_dim)); gl.fc_b.push_back(gtake(p.out_dim)); }
        gl.rec_wih.clear(); gl.rec_whh.clear(); gl.rec_bih.clear(); gl.rec_bhh.clear();
        for (int l = 0; l < c.recon_n_layers; ++l) {
            const int in = m.rec[l].in_dim, H = m.rec[l].H;
            gl.rec_wih.push_back(gtake((int64_t)3 * H * in)); gl.rec_whh.push_back(gtake((int64_t)3 * H * H));
            gl.rec_bih.push_back(gtake(3 * H)); gl.rec_bhh.push_back(gtake(3 * H));
        }
        gl.rec_fc_w = gtake((int64_t)c.out_dim * c.recon_hid_dim); gl.rec_fc_b = gtake(c.out_dim);
        gl.total = go;

        b.supported = true;
        for (const GatPlan* g : {&m.feat, &m.temp})
            if (!g->fused) {
                // wide layers go through the generic kernels of mtadgat_bwdw.hip: GATv2 and (round 6) GAT v1, up to 512 nodes / node dimensions
                if (g->K > 512 || g->D > 512) { b.supported = false; b.why = "graph-attention layers with more than 512 nodes / features"; }
            }
        if (b.supported) {
            for (int which = 0; which < 2; ++which) {
                const GatPlan& g = which == 0 ? m.feat : m.temp;
                GatBwdPlan& gb = b.gat[which];
                gb.Ep = round_up(g.E, 32); gb.NTu = gb.Ep / 32;
                gb.wide = !g.fused;
                if (gb.wide && !c.use_gatv2) {       // GAT (v1), wide: plain parameter copies for k_gat_v1_prep / k_bw_v1 / k_gat_v1_finish
                    gb.w1_off = take((size_t)g.E * g.D);
                    gb.b1_off = take((size_t)g.E);
                    gb.a_off = take((size_t)2 * g.E);
                    continue;
                }
                if (gb.wide) {
                    gb.wu_off = take((size_t)2 * gb.NTu * g.Q * 256);
                    gb.wu3_off = take((size_t)2 * gb.NTu * ((g.Q + 1) / 2) * 3 * 256);
                    gb.bu_off = take((size_t)2 * gb.Ep);
                    gb.a_off = take((size_t)gb.Ep);
                    lint(gb.lrT, 2 * gb.Ep, g.D);
                    wg(gb.wg, 2 * gb.Ep, g.D, true);
                    continue;
                }
                gb.att_lds = gat_bwd_att_lds(g.K, g.D, g.f_vld, (g.K + 15) / 16);
                if (!c.use_gatv2) {
                    // GAT (v1): the score backward is linear in the node vectors (mtadgat_bwd.hip): plain parameter copies
                    gb.w1_off = take((size_t)g.E * g.D);
                    gb.b1_off = take((size_t)g.E);
                    gb.a_off = take((size_t)2 * g.E);
                    gb.pair_lds = gat_bwd_v1_lds(g.K, g.D);
                    if (gb.att_lds > 160 * 1024 || gb.pair_lds > 160 * 1024) { b.supported = false; b.why = "attention backward tiles exceed the LDS"; }
                    continue;
                }
                gb.wu_off = take((size_t)2 * gb.NTu * g.Q * 256);
                gb.wu3_off = take((size_t)2 * gb.NTu * ((g.Q + 1) / 2) * 3 * 256);
                gb.bu_off = take((size_t)2 * gb.Ep);         // (round 6: the score backward's projections are a row GEMM here too)
                gb.a_off = take((size_t)gb.Ep);
                lint(gb.lrT, 2 * gb.Ep, g.D);
                wg(gb.wg, 2 * gb.Ep, g.D, true);
                if (gb.att_lds > 160 * 1024) { b.supported = false; b.why = "attention backward tiles exceed the LDS"; }
            }
            auto gru_b = [&](GruBwdPlan& gb, const GruPlan& g) {
                gb.whT_off = take((size_t)g.NCG * 12 * g.NCG * 256);
                gb.whT3_off = take((size_t)g.NCG * 6 * g.NCG * 3 * 256);
                lint(gb.wihT, 3 * g.Hp, g.in_dim);
                wg(gb.wg_ih, 3 * g.Hp, g.in_dim, true);
                wg(gb.wg_hh, 3 * g.Hp, g.H, true);
            };
            b.gru.assign(m.gru.size(), GruBwdPlan());
            b.rec.assign(m.rec.size(), GruBwdPlan());
            for (size_t l = 0; l < m.gru.size(); ++l) gru_b(b.gru[l], m.gru[l]);
            for (size_t l = 0; l < m.rec.size(); ++l) gru_b(b.rec[l], m.rec[l]);
            b.fcT.assign(m.fc.size(), LinTPlan());
            b.fc_wg.assign(m.fc.size(), WgradPlan());
            for (size_t i = 0; i < m.fc.size(); ++i) {
                lint(b.fcT[i], m.fc[i].out_dim, m.fc[i].in_dim);
                wg(b.fc_wg[i], m.fc[i].out_dim, m.fc[i].in_dim, true);
            }
            lint(b.recfcT, c.out_dim, m.rec.back().Hp);
            wg(b.recfc_wg, c.out_dim, c.recon_hid_dim, true);
            wg(b.conv_wg, m.F, m.taps * m.F, true);
            b.zero_off = take(1024);
        }
    }
    m.packed_floats = off;
    return "";
}

void plan_workspace(const Model& m, int64_t n, Workspace& ws) {
    size_t off = 0;
    auto take = [&](size_t cnt) {
        size_t o = off;
        off = align64(off + cnt);
        return o;
    };
    const size_t N = (size_t)n;
    // un-fused path intermediates (also used by the stage entry point mtadgat_gat, which copies its input
    // into xc / xcT): the projected L'/R'^T only exist when a layer's tiles do not fit in LDS
    const bool both_fused = m.temp.fused && m.feat.fused;
    const size_t xc_n = N * m.W * m.Fp, xct_n = N * m.F * m.Wp;
    if (!both_fused) {
        ws.xc = take(xc_n);
        ws.xct = take(xct_n);
    }
    ws.lct = take(m.temp.fused ? 0 : N * m.W * m.temp.ldl);
    ws.rtt = take(m.temp.fused ? 0 : N * m.temp.rt_rows * m.temp.Kp);
    ws.lcf = take(m.feat.fused ? 0 : N * m.F * m.feat.ldl);
    ws.rtf = take(m.feat.fused ? 0 : N * m.feat.rt_rows * m.feat.Kp);
    if (both_fused) {
        // forward() never materialises xc / xc^T then; the stage entry point mtadgat_gat stages its input
        // there, and does not use h_cat: the two share the space
        const size_t hc = N * m.W * m.Dp;
        ws.hcat = take(std::max(hc, align64(xc_n) + xct_n));
        ws.xc = ws.hcat;
        ws.xct = ws.hcat + align64(xc_n);
    } else {
        ws.hcat = take(N * m.W * m.Dp);
    }
    ws.hend = take(N * m.gru.back().Hp);
    const bool gseq = m.gru.size() > 1;
    ws.seq0 = take(gseq ? N * m.W * m.gru[0].Hp : 0);
    ws.seq1 = take(m.gru.size() > 2 ? N * m.W * m.gru[0].Hp : 0);
    size_t fcw = 0;
    for (const LinPlan& p : m.fc) fcw = std::max(fcw, (size_t)p.NT * 32);
    ws.fc0 = take(N * fcw);
    ws.fc1 = take(N * fcw);
    // decoder state sequences: between stacked decoder layers, and -- when the per-step Linear has more than a
    // few outputs -- of the last layer, so that recon_model.fc runs as one row GEMM after the recurrence
    const bool rec16 = m.rec.size() == 1 && m.rec[0].has16 && n <= G16_MAX_WINDOWS;
    ws.rec16 = rec16;
    const bool rseq = m.rec.size() > 1 || m.cfg.out_dim > 4 || rec16;
    ws.rseq0 = take(rseq ? N * m.W * m.rec[0].Hp : 0);
    ws.rseq1 = take((m.rec.size() > 2 || (m.rec.size() > 1 && m.cfg.out_dim > 4)) ? N * m.W * m.rec[0].Hp : 0);
    ws.has_xp = m.gru[0].has_xproj && n <= 16384;          // 64 windows per CU x 256 CUs: above that k_gru streams x itself
    ws.xp = take((ws.has_xp || rec16) ? N * m.W * 3 * std::max(m.gru[0].Hp, m.rec[0].Hp) : 0);
    ws.vmax = take(64);
    ws.winflag = take(both_fused ? (N + 3) / 4 + 16 : 0);
    // series path with stride-1 windows: the convolution of the segment (n + W - 1 rows) and of the windows' edge rows
    ws.cf = take(both_fused ? (N + m.W) * m.Fp : 0);
    ws.el = take(both_fused ? N * 2 * m.pad * m.Fp : 0);
    ws.er = take(both_fused ? N * 2 * m.pad * m.Fp : 0);
    ws.total = off;
}


int wgrad_slabs(long R, int Mp, int Np) {
    // 128 x 128 output blocks (one workgroup of 4 waves each); enough row slabs for ~4 workgroups per CU
    const long tiles = (long)((Mp + 127) / 128) * ((Np + 127) / 128);
    long s = 1024 / (tiles > 0 ? tiles : 1);
    // ... of at least MTADGAT_WGRAD_ROWS rows.  Measured (SMD shape, ms per training step at 256 / 1 024 / 8 192 windows): 128 rows
    // 2.84 / 7.79 / 33.5, 384 rows 2.80 / 6.91 / 33.6, 768 rows 3.16 / 7.12 / 34.0, 1 536 rows 3.99 / 8.49 / 34.2 -- a workgroup walks
    // its rows serially (a latency chain of staged 16-row steps), so short slabs win until the partial blocks (Mp x Np floats per
    // slab, written here and read back by k_wgrad_reduce) cost more than the parallelism buys
    static const long min_rows = getenv("MTADGAT_WGRAD_ROWS") ? atol(getenv("MTADGAT_WGRAD_ROWS")) : 384;
    const long rmax = (R + min_rows - 1) / min_rows;
    if (s > rmax) s = rmax;
    if (s > 512) s = 512;
    if (s >= 8) s = s / 8 * 8;                 // whole groups of 8 slabs: the kernel's XCD-aware block map (k_wgrad_lds)
    return (int)(s < 1 ? 1 : s);
}

void plan_tape(const Model& m, int64_t n, Tape& t) {
    size_t off = 0;
    auto take = [&](size_t cnt) {
        size_t o = off;
        off = align64(off + cnt);
        return o;
    };
    const size_t N = (size_t)n;
    const GruPlan& g = m.gru[0];
    const GruPlan& r = m.rec[0];
    t.hcat = take(N * m.W * m.Dp);
    t.xct = take(N * m.F * m.Wp);
    t.att_f = take(N * m.F * m.F);
    t.att_t = take(N * m.W * m.W);
    t.hend = take(N * g.Hp);
    t.gates_g = take(N * m.W * 4 * g.Hp);
    t.seq_g = take(N * m.W * g.Hp);
    t.gates_d = take(N * m.W * 4 * r.Hp);
    t.seq_d = take(N * m.W * r.Hp);
    t.xdec = take(N * m.W * g.Hp);
    t.xp = take(N * m.W * 3 * std::max(g.Hp, r.Hp));
    t.fc_act.clear();
    for (size_t i = 0; i + 1 < m.fc.size(); ++i) t.fc_act.push_back(take(N * (size_t)m.fc[i].NT * 32));
    t.vmax = take(64);
    t.lct = take(m.temp.fused ? 0 : N * m.W * m.temp.ldl);
    t.rtt = take(m.temp.fused ? 0 : N * m.temp.rt_rows * m.temp.Kp);
    t.lcf = take(m.feat.fused ? 0 : N * m.F * m.feat.ldl);
    t.rtf = take(m.feat.fused ? 0 : N * m.feat.rt_rows * m.feat.Kp);
    t.gates_gu.clear(); t.seq_gu.clear(); t.drop_g.clear(); t.gates_du.clear(); t.seq_du.clear(); t.drop_d.clear();
    for (size_t l = 1; l < m.gru.size(); ++l) {
        t.drop_g.push_back(take(N * m.W * m.gru[l - 1].Hp));
        t.gates_gu.push_back(take(N * m.W * 4 * m.gru[l].Hp));
        t.seq_gu.push_back(take(N * m.W * m.gru[l].Hp));
    }
    for (size_t l = 1; l < m.rec.size(); ++l) {
        t.drop_d.push_back(take(N * m.W * m.rec[l - 1].Hp));
        t.gates_du.push_back(take(N * m.W * 4 * m.rec[l].Hp));
        t.seq_du.push_back(take(N * m.W * m.rec[l].Hp));
    }
    t.total = off;
}

void plan_bwd_workspace(const Model& m, int64_t n, BwdWorkspace& w) {
    size_t off = 0;
    auto take = [&](size_t cnt) {
        size_t o = off;
        off = align64(off + cnt);
        return o;
    };
    const size_t N = (size_t)n;
    const GruPlan& g = m.gru[0];
    const GruPlan& r = m.rec[0];
    const BwdPlan& b = m.bw;
    size_t hpmax = std::max(g.Hp, r.Hp);
    for (const GruPlan& q : m.gru) hpmax = std::max(hpmax, (size_t)q.Hp);
    for (const GruPlan& q : m.rec) hpmax = std::max(hpmax, (size_t)q.Hp);
    w.da = take(N * m.W * 4 * hpmax);
    w.dhcat = take(N * m.W * m.Dp);
    w.dhdec = take(N * m.W * hpmax);
    w.dhend = take(N * g.Hp);
    size_t fcw = 32;
    for (const LinPlan& p : m.fc) fcw = std::max(fcw, (size_t)std::max(p.NT * 32, round_up(p.in_dim, 32)));
    w.dz0 = take(N * fcw);
    w.dz1 = take(N * fcw);
    w.de_f = take(N * m.F * m.F);
    w.de_t = take(N * m.W * m.W);
    w.dv_f = take(N * m.F * m.Wp);
    w.dv_t = take(N * m.W * m.Fp);
    // (GAT v1 reuses these regions for its per-window partials [sum dc_i v_i | sum dd_j v_j | sc sd]: 2 D + 2 floats per window)
    w.dlr_f = take(std::max(N * m.F * 2 * b.gat[0].Ep, m.cfg.use_gatv2 ? (size_t)0 : N * (2 * (size_t)m.W + 2)));
    w.dlr_t = take(std::max(N * m.W * 2 * b.gat[1].Ep, m.cfg.use_gatv2 ? (size_t)0 : N * (2 * (size_t)m.F + 2)));
    w.dap_f = take(N * b.gat[0].Ep);
    w.dap_t = take(N * b.gat[1].Ep);
    w.dpre = take(N * m.W * m.Fp);
    {   // wide attention layers (mtadgat_bwdw.hip), one layer at a time
        size_t ds = 0, lr = 0;
        for (int which = 0; which < 2; ++which) {
            const GatPlan& gp = which == 0 ? m.feat : m.temp;
            // (round 6: the un-scaled projections [L | R] go through memory for EVERY GATv2 layer -- the one-pass score backward
            // k_bw_pair serves the fused layers too)
            if (m.cfg.use_gatv2 || b.gat[which].wide) lr = std::max(lr, N * gp.K * 2 * (size_t)b.gat[which].Ep);
            if (!b.gat[which].wide) continue;
            ds = std::max(ds, N * gp.K * (size_t)round_up(gp.D, 4));
        }
        w.wds = take(ds); w.wlr = take(lr);          // (no transposed copy of d e since round 6: the score backward is one pass)
    }
    // partial sums of the weight-gradient GEMMs: one region each (round 6), so that their reductions can wait for ONE batched launch
    // at the end of the backward (the GEMMs used to share a region, each followed by its own reduction launch: twelve per step)
    size_t wp = 0;
    auto need = [&](const WgradPlan& p, long R) { wp += align64((size_t)wgrad_slabs(R, p.Mp, p.Np) * p.Mp * p.Np); };
    const long RW = (long)n * m.W;
    need(b.conv_wg, RW);
    if (m.cfg.use_gatv2) {
        need(b.gat[0].wg, (long)n * m.F);
        need(b.gat[1].wg, RW);
    }
    for (const GruBwdPlan& gp : b.gru) { need(gp.wg_ih, RW); need(gp.wg_hh, RW); }
    for (const GruBwdPlan& gp : b.rec) { need(gp.wg_ih, RW); need(gp.wg_hh, RW); }
    need(b.recfc_wg, RW);
    for (const WgradPlan& p : b.fc_wg) need(p, (long)n);
    w.wpart_floats = wp;
    w.wpart = take(wp);
    const size_t pv = 2 * (size_t)std::max(m.F, m.W) + 2;             // GAT (v1): per-window partials [p1 | p2 | sc sd]
    w.v1s = take(4 * pv);
    const size_t smax = std::max(std::max((size_t)m.W * m.W, pv), std::max((size_t)m.F * m.F, (size_t)std::max(b.gat[0].Ep, b.gat[1].Ep)));
    w.sums = take(sum_rows_scratch(n, (int)smax));
    for (int which = 0; which < 2; ++which) {
        const size_t K = which == 0 ? (size_t)m.F : (size_t)m.W;
        w.sums_q[2 * which] = m.cfg.use_gatv2 ? take(sum_rows_scratch(n, (int)(K * K))) : 0;
        w.sums_q[2 * which + 1] = m.cfg.use_gatv2 ? take(sum_rows_scratch(n, b.gat[which].Ep)) : 0;
    }
    w.total = off;
}

// embedding columns with a'_k = (1 - alpha)/2 a_k >= 0 first (padded to a multiple of 8), then the negative ones
void gat_column_order(const float* a, int E, double alpha, std::vector<int>& colk, int& P8, int& PT, int* npos) {
    std::vector<int> pos, neg;
    for (int k = 0; k < E; ++k) (((1.0 - alpha) * 0.5 * (double)a[k]) >= 0.0 ? pos : neg).push_back(k);
    if (npos) *npos = (int)pos.size();
    P8 = round_up((int)pos.size(), 8);
    const int N8 = round_up((int)neg.size(), 8);
    PT = P8 + N8;
    colk.assign(PT, -1);
    for (size_t n = 0; n < pos.size(); ++n) colk[n] = pos[n];
    for (size_t n = 0; n < neg.size(); ++n) colk[P8 + n] = neg[n];
}

static void pack_gat(Model& m, GatPlan& g, const float* lin_w, const float* lin_b, const float* a, const float* bias,
                     std::vector<float>& out) {
    const int E = g.E, D = g.D;
    const double alpha = m.cfg.alpha;
    // projected columns: query side [0, ldl) = [L'(PT) | c | 0..], key side [ldl, 2 ldl) = [R'(PT) | d | 0..]
    const int NC = 2 * g.ldl, KS = g.ldl;
    std::vector<double> rows((size_t)NC * D, 0.0), bvec(NC, 0.0);
    if (m.cfg.use_gatv2) {
        // e_ij = a . LeakyReLU(W_l v_i + W_r v_j + b)          (reference modules.py:74-77, :174-177)
        //      = c_i + d_j + sum_k a'_k |L_ik + R_jk|,  LeakyReLU(u) = (1+alpha)/2 u + (1-alpha)/2 |u|
        const int lin_in = 2 * D;
        std::vector<int> colk;
        gat_column_order(a, E, alpha, colk, g.P8, g.PT, &g.npos);
        for (int n = 0; n < g.PT; ++n) {
            const int k = colk[n];
            if (k < 0) continue;
            const double s = std::fabs((1.0 - alpha) * 0.5 * (double)a[k]);
            for (int cidx = 0; cidx < D; ++cidx) {
                rows[(size_t)n * D + cidx] = s * (double)lin_w[(size_t)k * lin_in + cidx];
                rows[(size_t)(KS + n) * D + cidx] = s * (double)lin_w[(size_t)k * lin_in + D + cidx];
            }
            bvec[n] = s * (double)lin_b[k];
        }
        const double hl = (1.0 + alpha) * 0.5;
        for (int k = 0; k < E; ++k) {
            for (int cidx = 0; cidx < D; ++cidx) {
                rows[(size_t)(g.PT) * D + cidx] += hl * (double)a[k] * (double)lin_w[(size_t)k * lin_in + cidx];
                rows[(size_t)(KS + g.PT) * D + cidx] += hl * (double)a[k] * (double)lin_w[(size_t)k * lin_in + D + cidx];
            }
            bvec[g.PT] += hl * (double)a[k] * (double)lin_b[k];
        }
    } else {
        // e_ij = LeakyReLU(a1 . (W v_i + b) + a2 . (W v_j + b))   (reference modules.py:80-83, :180-183)
        g.PT = g.P8 = 0;
        for (int k = 0; k < E; ++k) {
            for (int cidx = 0; cidx < D; ++cidx) {
                rows[(size_t)0 * D + cidx] += (double)a[k] * (double)lin_w[(size_t)k * D + cidx];
                rows[(size_t)KS * D + cidx] += (double)a[E + k] * (double)lin_w[(size_t)k * D + cidx];
            }
            bvec[0] += (double)a[k] * (double)lin_b[k];
            bvec[KS] += (double)a[E + k] * (double)lin_b[k];
        }
    }
    pack_tiles(out.data() + g.w_off, g.NT, g.Q, [&](int n, int k) -> float {
        if (n >= NC) return 0.f;
        if (k < D) return (float)rows[(size_t)n * D + k];
        return (g.fused && k == D) ? (float)bvec[n] : 0.f;       // fused kernel: bias = weight row D
    });
    if (g.fused && m.precision == 1)
        pack_tiles_bf16(out.data() + g.w16_off, g.NT, g.Q16, [&](int n, int k) -> float {
            if (n >= NC) return 0.f;
            if (k < D) return (float)rows[(size_t)n * D + k];
            return k == D ? (float)bvec[n] : 0.f;
        });
    for (int n = 0; n < NC; ++n) out[g.b_off + n] = (float)bvec[n];
    std::memcpy(out.data() + g.bias_off, bias, sizeof(float) * (size_t)g.K * g.K);
    {   // the sign-group boundaries the kernels read (ints in the float image)
        int* ord = reinterpret_cast<int*>(out.data() + g.ord_off);
        ord[0] = g.P8; ord[1] = g.PT; ord[2] = g.npos; ord[3] = 0;
    }
}

static void pack_gru_layer(const GruPlan& g, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                           std::vector<float>& out, bool bf16) {
    const int H = g.H, in = g.in_dim;
    if (g.xmode == 0) {
        pack_gru_tiles(out.data() + g.wx_off, g.NCG, g.Qxp, [&](int st, int r, int k) -> float {
            return (r < H && k < in) ? w_ih[((size_t)st * H + r) * in + k] : 0.f;
        });
    }
    pack_gru_tiles(out.data() + g.wh_off, g.NCG, 4 * g.NCG + 2, [&](int st, int r, int k) -> float {
        return (r < H && k < H) ? w_hh[((size_t)st * H + r) * H + k] : 0.f;
    });
    if (g.xmode == 0 && bf16) {
        pack_gru_tiles_bf16(out.data() + g.wx16_off, g.NCG, g.Qxp16, [&](int st, int r, int k) -> float {
            return (r < H && k < in) ? w_ih[((size_t)st * H + r) * in + k] : 0.f;
        });
    }
    if (bf16)
        pack_gru_tiles_bf16(out.data() + g.wh16_off, g.NCG, 2 * g.NCG + 2, [&](int st, int r, int k) -> float {
            return (r < H && k < H) ? w_hh[((size_t)st * H + r) * H + k] : 0.f;
        });
    float* b = out.data() + g.b_off;
    for (int j = 0; j < g.Hp; ++j) {
        b[0 * g.Hp + j] = j < H ? b_ih[j] + b_hh[j] : 0.f;
        b[1 * g.Hp + j] = j < H ? b_ih[H + j] + b_hh[H + j] : 0.f;
        b[2 * g.Hp + j] = j < H ? b_ih[2 * H + j] : 0.f;
        b[3 * g.Hp + j] = j < H ? b_hh[2 * H + j] : 0.f;
    }
    if (g.has16) {
        float* w = out.data() + g.g16_off;
        float* wT = out.data() + g.g16T_off;
        for (int tile = 0; tile < g.NT16; ++tile)
            for (int gate = 0; gate < 3; ++gate)
                for (int s = 0; s < g.KS16; ++s)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int unit = 16 * tile + (lane & 15), k = 4 * s + (lane >> 4);
                        const bool ok = unit < H && k < H;
                        const size_t o = (((size_t)tile * 3 + gate) * g.KS16 + s) * 64 + lane;
                        w[o] = ok ? w_hh[((size_t)gate * H + unit) * H + k] : 0.f;
                        wT[o] = ok ? w_hh[((size_t)gate * H + k) * H + unit] : 0.f;
                    }
    }
    if (g.has16) {
        const int HK = 16 * g1_ksm(H), nwf = g1_waves(H, g.Hp, false), nwb = g1_waves(H, g.Hp, true), jbs = (H + 15) / 16;
        float* w = out.data() + g.g1_off;
        for (int wv = 0; wv < nwf; ++wv)
            for (int k = 0; k < HK; ++k)
                for (int lane = 0; lane < 64; ++lane) {
                    const int R = wv * 64 + lane;
                    w[((size_t)wv * HK + k) * 64 + lane] = (R < 3 * H && k < H) ? w_hh[(size_t)R * H + k] : 0.f;
                }
        float* wT = out.data() + g.g1T_off;
        for (int wv = 0; wv < nwb; ++wv)
            for (int k = 0; k < HK; ++k)
                for (int lane = 0; lane < 64; ++lane) {
                    const int rr = (wv * 64 + lane) >> 4, p = rr % 3, jb = rr / 3, j = 16 * jb + (lane & 15);
                    wT[((size_t)wv * HK + k) * 64 + lane] = (jb < jbs && j < H && k < H) ? w_hh[((size_t)p * H + k) * H + j] : 0.f;
                }
    }
    if (g.has_xproj) {
        const int Hp = g.Hp;
        pack_tiles(out.data() + g.xproj.w_off, g.xproj.NT, g.xproj.Q, [&](int n, int k) -> float {
            const int blk = n / Hp, u = n % Hp;
            return (blk < 3 && u < H && k < in) ? w_ih[((size_t)blk * H + u) * in + k] : 0.f;
        });
        std::memcpy(out.data() + g.xproj.b_off, b, sizeof(float) * 3 * Hp);
    }
}

std::string pack_weights(Model& m, const mtadgat_params& p, std::vector<float>& out) {
    const mtadgat_config& c = m.cfg;
    out.assign(m.packed_floats, 0.f);
    if (!p.conv_weight || !p.conv_bias || !p.feat_lin_weight || !p.feat_lin_bias || !p.feat_a || !p.feat_bias ||
        !p.temp_lin_weight || !p.temp_lin_bias || !p.temp_a || !p.temp_bias || !p.rec_fc_weight || !p.rec_fc_bias)
        return "null parameter pointer";
    for (int l = 0; l < c.gru_n_layers; ++l)
        if (!p.gru_w_ih[l] || !p.gru_w_hh[l] || !p.gru_b_ih[l] || !p.gru_b_hh[l]) return "null GRU parameter pointer";
    for (int i = 0; i < c.forecast_n_linear; ++i)
        if (!p.fc_weight[i] || !p.fc_bias[i]) return "null forecasting parameter pointer";
    for (int l = 0; l < c.recon_n_layers; ++l)
        if (!p.rec_w_ih[l] || !p.rec_w_hh[l] || !p.rec_b_ih[l] || !p.rec_b_hh[l]) return "null decoder parameter pointer";
    // The regions of the packed image are independent: five host threads fill them concurrently (the image is
    // re-packed after every optimizer step, so this sits on the training step's critical path).
    // conv: K index = tap*Fp + channel   (reference weight (F_out, F_in, k), modules.py:15)
    auto job_conv_feat = [&]() {
        const int F = m.F, Fp = m.Fp, taps = m.taps;
        pack_tiles(out.data() + m.conv_w_off, m.convNT, taps * Fp / 8, [&](int n, int k) -> float {
            const int tap = k / Fp, ch = k % Fp;
            return (n < F && ch < F && tap < taps) ? p.conv_weight[((size_t)n * F + ch) * taps + tap] : 0.f;
        });
        for (int n = 0; n < F; ++n) out[m.conv_b_off + n] = p.conv_bias[n];
        for (size_t k = 0; k < (size_t)F * F * taps; ++k) out[m.conv_wraw_off + k] = p.conv_weight[k];
        const int Fp16 = m.Fp16;
        pack_tiles(out.data() + m.conv_wf16_off, m.convNT, taps * Fp16 / 8, [&](int n, int k) -> float {
            const int tap = k / Fp16, ch = k % Fp16;
            return (n < F && ch < F && tap < taps) ? p.conv_weight[((size_t)n * F + ch) * taps + tap] : 0.f;
        });
        if (m.precision == 1) pack_tiles_bf16(out.data() + m.conv_w16_off, m.convNT, taps * Fp16 / 16, [&](int n, int k) -> float {
            const int tap = k / Fp16, ch = k % Fp16;
            return (n < F && ch < F && tap < taps) ? p.conv_weight[((size_t)n * F + ch) * taps + tap] : 0.f;
        });
        pack_gat(m, m.feat, p.feat_lin_weight, p.feat_lin_bias, p.feat_a, p.feat_bias, out);
    };
    auto job_temp_gru = [&]() {
        pack_gat(m, m.temp, p.temp_lin_weight, p.temp_lin_bias, p.temp_a, p.temp_bias, out);
        for (int l = 0; l < c.gru_n_layers; ++l) pack_gru_layer(m.gru[l], p.gru_w_ih[l], p.gru_w_hh[l], p.gru_b_ih[l], p.gru_b_hh[l], out, m.precision == 1);
    };
    auto job_heads = [&]() {
    for (int i = 0; i < c.forecast_n_linear; ++i) {
        const LinPlan& lp = m.fc[i];
        const float* w = p.fc_weight[i];
        pack_tiles(out.data() + lp.w_off, lp.NT, lp.Q, [&](int n, int k) -> float {
            return (n < lp.out_dim && k < lp.in_dim) ? w[(size_t)n * lp.in_dim + k] : 0.f;
        });
        for (int n = 0; n < lp.out_dim; ++n) out[lp.b_off + n] = p.fc_bias[i][n];
    }
    for (int l = 0; l < c.recon_n_layers; ++l) {
        const GruPlan& g = m.rec[l];
        if (g.xmode == 1) {
            // x_t[j] = h_end[(t*Hin + j) / T]: fold W_ih over the j that share an h_end entry.  The j of one entry
            // are consecutive, so every folded weight is a difference of two prefix sums over j (double)
            const int Hin = g.in_dim, T = m.W, H = g.H;
            const int NMp = 8 * g.Qx;
            std::vector<double> pre((size_t)3 * H * (Hin + 1));
            for (int r = 0; r < 3 * H; ++r) {
                double acc = 0.0;
                pre[(size_t)r * (Hin + 1)] = 0.0;
                for (int j = 0; j < Hin; ++j) {
                    acc += (double)p.rec_w_ih[l][(size_t)r * Hin + j];
                    pre[(size_t)r * (Hin + 1) + j + 1] = acc;
                }
            }
            int* m0 = reinterpret_cast<int*>(out.data() + g.m0_off);
            std::vector<float> fw((size_t)3 * H * NMp);          // folded weights of one step: [gate*H + r][k]
            auto steps = [&](int t_lo, int t_hi, std::vector<float>& f) {
                for (int t = t_lo; t < t_hi; ++t) {
                    const int lo = (int)(((long)t * Hin) / T);
                    m0[t] = lo;
                    for (int k = 0; k < NMp; ++k) {
                        // j with (t*Hin + j) / T == lo + k:  j in [(lo+k)*T - t*Hin, (lo+k+1)*T - t*Hin)
                        long j0 = (long)(lo + k) * T - (long)t * Hin, j1 = j0 + T;
                        j0 = j0 < 0 ? 0 : j0;
                        j1 = j1 > Hin ? Hin : j1;
                        for (int r = 0; r < 3 * H; ++r)
                            f[(size_t)r * NMp + k] = j1 > j0 ? (float)(pre[(size_t)r * (Hin + 1) + j1] - pre[(size_t)r * (Hin + 1) + j0]) : 0.f;
                    }
                    auto fold = [&](int st, int r, int k) -> float { return (r < H && k < NMp) ? f[((size_t)st * H + r) * NMp + k] : 0.f; };
                    pack_gru_tiles(out.data() + g.wx_off + (size_t)t * g.NCG * g.Qxp * 3 * 256, g.NCG, g.Qxp, fold);
                    if (m.precision == 1) pack_gru_tiles_bf16(out.data() + g.wx16_off + (size_t)t * g.NCG * g.Qxp16 * 3 * 256, g.NCG, g.Qxp16, fold);
                    if (g.has16) {
                        float* fo = out.data() + g.fold_off + (size_t)t * 3 * g.Hp * 8;
                        for (int st = 0; st < 3; ++st)
                            for (int r = 0; r < H; ++r)
                                for (int k = 0; k < 8; ++k) fo[((size_t)st * g.Hp + r) * 8 + k] = f[((size_t)st * H + r) * NMp + k];
                    }
                }
            };
            std::vector<float> fw2(fw.size());
            std::thread th(steps, T / 2, T, std::ref(fw2));
            steps(0, T / 2, fw);
            th.join();
        }
        pack_gru_layer(g, p.rec_w_ih[l], p.rec_w_hh[l], p.rec_b_ih[l], p.rec_b_hh[l], out, m.precision == 1);
    }
    {
        const LinPlan& lp = m.rec_fc;
        pack_tiles(out.data() + lp.w_off, lp.NT, lp.Q, [&](int n, int k) -> float {
            return (n < lp.out_dim && k < lp.in_dim) ? p.rec_fc_weight[(size_t)n * lp.in_dim + k] : 0.f;
        });
        for (int n = 0; n < lp.out_dim; ++n) out[lp.b_off + n] = p.rec_fc_bias[n];
    }
    };

    // ---- backward packs (transposed weights, un-scaled attention projections, gradient index maps)
    auto job_bwd = [&]() {
    if (m.bw.supported) {
        const BwdPlan& b = m.bw;
        auto maps = [&](const WgradPlan& wp, auto rowW, auto col, auto rowB) {
            int* rw = reinterpret_cast<int*>(out.data() + wp.rowW_off);
            int* cm = reinterpret_cast<int*>(out.data() + wp.col_off);
            int* rb = reinterpret_cast<int*>(out.data() + wp.rowB_off);
            for (int i = 0; i < wp.Mp; ++i) { rw[i] = i < wp.M ? rowW(i) : -1; rb[i] = i < wp.M ? rowB(i) : -1; }
            for (int i = 0; i < wp.Np; ++i) cm[i] = i < wp.N ? col(i) : -1;
        };
        auto ident = [](int n) { return n; };
        for (int which = 0; which < 2; ++which) {
            const GatPlan& g = which == 0 ? m.feat : m.temp;
            const GatBwdPlan& gb = b.gat[which];
            const float* lw = which == 0 ? p.feat_lin_weight : p.temp_lin_weight;
            const float* lb = which == 0 ? p.feat_lin_bias : p.temp_lin_bias;
            const float* av = which == 0 ? p.feat_a : p.temp_a;
            const int E = g.E, D = g.D, Ep = gb.Ep;
            if (!c.use_gatv2) {
                for (size_t i = 0; i < (size_t)E * D; ++i) out[gb.w1_off + i] = lw[i];
                for (int e = 0; e < E; ++e) out[gb.b1_off + e] = lb[e];
                for (int e = 0; e < 2 * E; ++e) out[gb.a_off + e] = av[e];
                continue;
            }
            pack_tiles(out.data() + gb.wu_off, 2 * gb.NTu, g.Q, [&](int n, int k) -> float {
                const int side = n / Ep, e = n % Ep;
                if (e >= E) return 0.f;
                if (k < D) return lw[(size_t)e * 2 * D + (size_t)side * D + k];
                return (k == D && side == 0) ? lb[e] : 0.f;
            });
            for (int e = 0; e < E; ++e) out[gb.a_off + e] = av[e];
            for (int e = 0; e < E; ++e) out[gb.bu_off + e] = lb[e];      // (the row GEMM takes the projection bias as a vector)
            pack_tiles(out.data() + gb.lrT.w_off, gb.lrT.NT, gb.lrT.Q, [&](int n, int k) -> float {
                const int side = k / Ep, e = k % Ep;
                return (n < D && e < E && side < 2) ? lw[(size_t)e * 2 * D + (size_t)side * D + n] : 0.f;
            });
            maps(gb.wg, [&](int r) { const int side = r / Ep, e = r % Ep; return e < E ? e * 2 * D + side * D : -1; }, ident,
                 [&](int r) { const int side = r / Ep, e = r % Ep; return (side == 0 && e < E) ? e : -1; });
        }
        auto gru_pack = [&](const GruBwdPlan& gb, const GruPlan& g, const float* w_ih, const float* w_hh) {
            const int H = g.H, Hp = g.Hp, in = g.in_dim;
            pack_tiles(out.data() + gb.whT_off, g.NCG, 12 * g.NCG, [&](int n, int k) -> float {      // blocks r | z | nh
                const int blk = k / Hp, u = k % Hp;
                return (n < H && u < H && blk < 3) ? w_hh[((size_t)blk * H + u) * H + n] : 0.f;
            });
            auto gate_ih = [](int blk) { return blk == 0 ? 2 : blk - 1; };                            // blocks n | r | z
            pack_tiles(out.data() + gb.wihT.w_off, gb.wihT.NT, gb.wihT.Q, [&](int n, int k) -> float {
                const int blk = k / Hp, u = k % Hp;
                return (n < in && u < H && blk < 3) ? w_ih[((size_t)gate_ih(blk) * H + u) * in + n] : 0.f;
            });
            maps(gb.wg_ih, [&](int r) { const int blk = r / Hp, u = r % Hp; return u < H ? (gate_ih(blk) * H + u) * in : -1; }, ident,
                 [&](int r) { const int blk = r / Hp, u = r % Hp; return u < H ? gate_ih(blk) * H + u : -1; });
            maps(gb.wg_hh, [&](int r) { const int blk = r / Hp, u = r % Hp; return u < H ? (blk * H + u) * H : -1; }, ident,
                 [&](int r) { const int blk = r / Hp, u = r % Hp; return u < H ? blk * H + u : -1; });
        };
        for (size_t l = 0; l < m.gru.size(); ++l) gru_pack(b.gru[l], m.gru[l], p.gru_w_ih[l], p.gru_w_hh[l]);
        for (size_t l = 0; l < m.rec.size(); ++l) gru_pack(b.rec[l], m.rec[l], p.rec_w_ih[l], p.rec_w_hh[l]);
        for (size_t i = 0; i < m.fc.size(); ++i) {
            const LinPlan& lp = m.fc[i];
            const float* w = p.fc_weight[i];
            pack_tiles(out.data() + b.fcT[i].w_off, b.fcT[i].NT, b.fcT[i].Q, [&](int n, int k) -> float {
                return (n < lp.in_dim && k < lp.out_dim) ? w[(size_t)k * lp.in_dim + n] : 0.f;
            });
            maps(b.fc_wg[i], [&](int r) { return r * lp.in_dim; }, ident, ident);
        }
        {
            const int Hr = c.recon_hid_dim, od = c.out_dim;
            pack_tiles(out.data() + b.recfcT.w_off, b.recfcT.NT, b.recfcT.Q, [&](int n, int k) -> float {
                return (n < Hr && k < od) ? p.rec_fc_weight[(size_t)k * Hr + n] : 0.f;
            });
            maps(b.recfc_wg, [&](int r) { return r * Hr; }, ident, ident);
        }
        {
            const int F = m.F, taps = m.taps;
            maps(b.conv_wg, [&](int r) { return r * F * taps; }, [&](int n) { const int tap = n / F, ch = n % F; return ch * taps + tap; }, ident);
        }
    }
    };
    std::thread t1(job_conv_feat), t2(job_temp_gru), t3(job_heads);
    job_bwd();
    t1.join(); t2.join(); t3.join();
    m.bf16_packed = (m.precision == 1);
    return "";
}


// ---- device-side re-packing: host tables ---------------------------------------------------------------
FlatOffsets flat_offsets(const Model& m) {
    const mtadgat_config& c = m.cfg;
    FlatOffsets f;
    int64_t go = 0;
    auto take = [&](int64_t n) { int64_t o = go; go += n; return o; };
    f.conv_w = take((int64_t)m.F * m.F * m.taps); f.conv_b = take(m.F);
    for (int which = 0; which < 2; ++which) {
        const GatPlan& g = which == 0 ? m.feat : m.temp;
        const int lin_in = c.use_gatv2 ? 2 * g.D : g.D;
        f.lin_w[which] = take((int64_t)g.E * lin_in); f.lin_b[which] = take(g.E);
        f.a[which] = take(c.use_gatv2 ? g.E : 2 * g.E); f.bias[which] = take((int64_t)g.K * g.K);
    }
    for (int l = 0; l < c.gru_n_layers; ++l) {
        const int in = m.gru[l].in_dim, H = m.gru[l].H;
        f.gru_wih.push_back(take((int64_t)3 * H * in)); f.gru_whh.push_back(take((int64_t)3 * H * H));
        f.gru_bih.push_back(take(3 * H)); f.gru_bhh.push_back(take(3 * H));
    }
    for (const LinPlan& p : m.fc) { f.fc_w.push_back(take((int64_t)p.out_dim * p.in_dim)); f.fc_b.push_back(take(p.out_dim)); }
    for (int l = 0; l < c.recon_n_layers; ++l) {
        const int in = m.rec[l].in_dim, H = m.rec[l].H;
        f.rec_wih.push_back(take((int64_t)3 * H * in)); f.rec_whh.push_back(take((int64_t)3 * H * H));
        f.rec_bih.push_back(take(3 * H)); f.rec_bhh.push_back(take(3 * H));
    }
    f.rec_fc_w = take((int64_t)c.out_dim * c.recon_hid_dim); f.rec_fc_b = take(c.out_dim);
    f.total = go;
    return f;
}

void params_from_flat(const Model& m, const FlatOffsets& f, const float* flat, mtadgat_params& p) {
    std::memset(&p, 0, sizeof(p));
    p.conv_weight = flat + f.conv_w; p.conv_bias = flat + f.conv_b;
    p.feat_lin_weight = flat + f.lin_w[0]; p.feat_lin_bias = flat + f.lin_b[0]; p.feat_a = flat + f.a[0]; p.feat_bias = flat + f.bias[0];
    p.temp_lin_weight = flat + f.lin_w[1]; p.temp_lin_bias = flat + f.lin_b[1]; p.temp_a = flat + f.a[1]; p.temp_bias = flat + f.bias[1];
    for (size_t l = 0; l < m.gru.size(); ++l) {
        p.gru_w_ih[l] = flat + f.gru_wih[l]; p.gru_w_hh[l] = flat + f.gru_whh[l];
        p.gru_b_ih[l] = flat + f.gru_bih[l]; p.gru_b_hh[l] = flat + f.gru_bhh[l];
    }
    for (size_t i = 0; i < m.fc.size(); ++i) { p.fc_weight[i] = flat + f.fc_w[i]; p.fc_bias[i] = flat + f.fc_b[i]; }
    for (size_t l = 0; l < m.rec.size(); ++l) {
        p.rec_w_ih[l] = flat + f.rec_wih[l]; p.rec_w_hh[l] = flat + f.rec_whh[l];
        p.rec_b_ih[l] = flat + f.rec_bih[l]; p.rec_b_hh[l] = flat + f.rec_bhh[l];
    }
    p.rec_fc_weight = flat + f.rec_fc_w; p.rec_fc_bias = flat + f.rec_fc_b;
}

// The gather table is obtained from the host packer itself: it is run over parameters whose value is their flat index
// + 1 (exact in fp32 below 2^24), so every position of the image that is a plain copy of a parameter names its
// source, whatever tile format it sits in.  Regions whose content is computed (scaled / sorted / summed / folded) are
// excluded -- the kernels of mtadgat_packdev.hip fill them -- and everything else (zero padding, index maps) is left
// as the first host-side load wrote it.
std::string build_device_tables(Model& m) {
    DevTables& t = m.dt;
    if (m.precision == 1) return "device-side re-packing covers the fp32 image only";
    t.fo = flat_offsets(m);
    if (t.fo.total != m.bw.gl.total) return "internal: flat parameter layout mismatch";
    if (t.fo.total >= (1 << 24)) return "model too large for the index-encoded gather table";
    std::vector<float> synth((size_t)t.fo.total);
    for (int64_t i = 0; i < t.fo.total; ++i) synth[(size_t)i] = (float)(i + 1);
    mtadgat_params p;
    params_from_flat(m, t.fo, synth.data(), p);
    const int keep[4] = {m.feat.PT, m.feat.P8, m.temp.PT, m.temp.P8};
    const int keep_np[2] = {m.feat.npos, m.temp.npos};
    const bool keep_bf = m.bf16_packed;
    std::vector<float> img;
    std::string err = pack_weights(m, p, img);
    m.feat.PT = keep[0]; m.feat.P8 = keep[1]; m.temp.PT = keep[2]; m.temp.P8 = keep[3];
    m.feat.npos = keep_np[0]; m.temp.npos = keep_np[1];
    m.bf16_packed = keep_bf;
    if (!err.empty()) return err;
    t.gidx.assign(m.packed_floats, -1);
    for (size_t i = 0; i < m.packed_floats; ++i) {
        const float v = img[i];
        if (v >= 1.f && v <= (float)t.fo.total && v == std::floor(v)) t.gidx[i] = (int)v - 1;
    }
    auto exclude = [&](size_t off, size_t n) { std::fill(t.gidx.begin() + off, t.gidx.begin() + off + n, -1); };
    for (int which = 0; which < 2; ++which) {
        const GatPlan& g = which == 0 ? m.feat : m.temp;
        exclude(g.w_off, (size_t)g.NT * g.Q * 256);
        exclude(g.b_off, (size_t)g.NT * 32);
        const int D = g.D, NC = 2 * g.ldl;
        std::vector<float> code((size_t)g.NT * g.Q * 256, 0.f);
        pack_tiles(code.data(), g.NT, g.Q, [&](int n, int k) -> float { return (n < NC && k <= D) ? (float)((size_t)n * (D + 1) + k + 1) : 0.f; });
        t.gatcode[which].assign(code.size(), 0);
        for (size_t i = 0; i < code.size(); ++i) t.gatcode[which][i] = (int)code[i];
    }
    auto gru_excl = [&](const GruPlan& g) {
        exclude(g.b_off, (size_t)4 * g.Hp);
        if (g.has_xproj) exclude(g.xproj.b_off, (size_t)3 * g.Hp);
    };
    for (const GruPlan& g : m.gru) gru_excl(g);
    for (const GruPlan& g : m.rec) gru_excl(g);
    t.foldcode.clear();
    {
        const GruPlan& g = m.rec[0];
        if (g.xmode == 1) {
            exclude(g.wx_off, (size_t)m.W * g.NCG * g.Qxp * 3 * 256);
            if (g.has16) exclude(g.fold_off, (size_t)m.W * 3 * g.Hp * 8);
            const int NMp = 8 * g.Qx, H = g.H;
            std::vector<float> code((size_t)g.NCG * g.Qxp * 3 * 256, 0.f);
            pack_gru_tiles(code.data(), g.NCG, g.Qxp, [&](int st, int r, int k) -> float {
                return (r < H && k < NMp) ? (float)(((size_t)st * H + r) * NMp + k + 1) : 0.f;
            });
            t.foldcode.assign(code.size(), 0);
            for (size_t i = 0; i < code.size(); ++i) t.foldcode[i] = (int)code[i];
        }
    }
    return "";
}

}  // namespace mtadgat
