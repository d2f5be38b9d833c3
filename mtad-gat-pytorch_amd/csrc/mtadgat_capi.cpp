// C ABI (include/mtadgat.h) over the gfx950 kernels: handle lifetime, weight upload,
// the forward() launch sequence and the per-stage entry points.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "mtadgat_host.h"

using namespace mtadgat;

struct mtadgat_handle_s {
    Model m;
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return MTADGAT_ERR_HIP;
}
#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);  \
    } while (0)
#define K_TRY(expr, what)                                                                          \
    do {                                                                                           \
        int rc__ = (expr);                                                                         \
        if (rc__ == -2) return fail(MTADGAT_ERR_UNSUPPORTED, std::string(what) + ": unsupported shape"); \
        if (rc__ != 0) return hip_fail((hipError_t)rc__, what);                                    \
    } while (0)

enum Slot { S_CONV = 0, S_PROJ = 1, S_ATTEND = 2, S_GRU = 3, S_FC = 4, S_RECON = 5 };
const char* kSlotNames[MTADGAT_PROFILE_SLOTS] = {"conv", "proj", "attend", "gru", "fc", "recon"};

struct Scope {  // brackets a kernel family with events when profiling is on
    Model& m;
    int slot;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    Scope(Model& m_, int slot_, hipStream_t s_) : m(m_), slot(slot_), s(s_) {
        if (m.profile) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
            (void)hipEventRecord(a, s);
        }
    }
    ~Scope() {
        if (a && b) {
            (void)hipEventRecord(b, s);
            m.ev[slot].emplace_back(a, b);
        }
    }
};

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- stage launch helpers; all pointers device, n windows ------------------------------------
// where the windows come from: a materialised (n, W, F) tensor, or views of a device-resident series
struct XSource {
    int x_bf16 = 0;                  // x points at bfloat16 elements
    const float* x = nullptr;        // windows (n, W, F) -- or the series when gather != 0
    int gather = 0;
    const int64_t* starts = nullptr;
    int64_t start0 = 0, stride = 1;
};

// the batch sizes / options at which the fused front end's convolution is the window-per-workgroup kernel on fp16 pieces
// (precision mode 1 -- bf16 operands -- takes the same front end from 4096 windows: its fp16-piece kernels are faster than the bf16
// builds of k_conv_lds / k_gat, 10.5 against 12.3 ms per 65 536 windows, and closer to the fp32 results; the recurrences stay bf16)
static bool split_front(const Model& m, int64_t n) { return m.precision == 1 && n >= 4096 && m.conv_kernel == 0 && m.gat_kernel == 0; }
static bool conv_win_selected(const Model& m, int64_t n) {
    return (m.precision == 2 || split_front(m, n)) && m.conv_kernel != 1 && (n >= 4096 || m.conv_kernel == 2);
}

// Round 6: the split packs are derived LAZILY, by the first launch that reads them after an upload.  Rounds 2-5 re-derived all of
// them behind every upload -- ~30 launches of a few microseconds each (ranges, scales, splits) after every optimizer step, although a
// training step of the reference's batch size (256 windows) reads none of them: 0.15-0.2 ms of a 2.3 ms step.  An upload now only
// bumps weights_version; ensure_*_split() derive on the stream of the consumer.  Calls that fork onto the second lane derive
// everything on the caller's stream first (ensure_all_split in forward_impl), so the lanes never race for a derivation.
int ensure_gru_split(Model& m, const GruPlan& g, hipStream_t s) {
    if (g.split_ver == m.weights_version) return 0;
    // the layer's power-of-two weight scale (fp16 range of the recurrent pieces), from the largest weight, on the device
    const long outer_x = (long)(g.xmode == 1 ? m.W : 1) * g.NCG;
    float* sc = m.packed_dev + g.scale_off;
    HIP_TRY(hipMemsetAsync(sc, 0, 4 * sizeof(float), s));
    K_TRY(launch_absmax(m.packed_dev + g.wx_off, outer_x * g.Qxp * 3 * 256, sc, s), "weight range");
    K_TRY(launch_absmax(m.packed_dev + g.wh_off, (long)g.NCG * (4 * g.NCG + 2) * 3 * 256, sc, s), "weight range");
    K_TRY(launch_scale_from_max(sc, s), "weight scale");
    K_TRY(launch_split_x(m.packed_dev + g.wx_off, m.packed_dev + g.wx3_off, outer_x, g.Qxp, g.Qxp16, g.qb3, sc + 1, s), "split input weights");
    if (g.wx2_off && g.qb3 > 0)
        K_TRY(launch_split_x(m.packed_dev + g.wx_off, m.packed_dev + g.wx2_off, outer_x, g.Qxp, g.Qxp16, 0, sc + 1, s), "split input weights (fp16)");
    K_TRY(launch_split2h(m.packed_dev + g.wh_off, m.packed_dev + g.wh3_off, g.NCG, 4 * g.NCG + 2, 2 * g.NCG + 2, 3, sc + 1, s),
          "split-fp16 recurrent weights");
    if (g.wxq_off)       // chunk-major copy of the two-piece input pack (k_gru_cm)
        K_TRY(launch_reorder_xq(m.packed_dev + ((g.wx2_off && g.qb3 > 0) ? g.wx2_off : g.wx3_off), m.packed_dev + g.wxq_off, g.NCG, g.Qxp16, s),
              "chunk-major input weights");
    g.split_ver = m.weights_version;
    return 0;
}
int ensure_conv_split(Model& m, hipStream_t s) {
    if (m.split_ver_conv == m.weights_version) return 0;
    // the window convolution's pack: two fp16 pieces of S * W in the 16-channel geometry; three bf16 pieces for the wide models' k_conv_x3
    float* sc = m.packed_dev + m.conv_scale_off;
    const int q8 = m.taps * m.Fp16 / 8;
    HIP_TRY(hipMemsetAsync(sc, 0, 4 * sizeof(float), s));
    K_TRY(launch_absmax(m.packed_dev + m.conv_wf16_off, (long)m.convNT * q8 * 256, sc, s), "convolution weight range");
    K_TRY(launch_scale_from_max(sc, s), "convolution weight scale");
    K_TRY(launch_split2h(m.packed_dev + m.conv_wf16_off, m.packed_dev + m.conv_w2h_off, m.convNT, q8, q8 / 2, 1, sc + 1, s), "split-fp16 convolution weights");
    K_TRY(launch_split3(m.packed_dev + m.conv_wf16_off, m.packed_dev + m.conv_w3_off, m.convNT, m.taps * m.Fp16 / 8, m.taps * m.Fp16 / 16, 1, nullptr, s),
          "split-bf16 convolution weights");
    m.split_ver_conv = m.weights_version;
    return 0;
}
int ensure_gat_split(Model& m, const GatPlan& g, hipStream_t s) {
    const int which = &g == &m.feat ? 0 : 1;
    if (m.split_ver_gat[which] == m.weights_version) return 0;
    if (!g.fused && g.uQ16 > 0)
        K_TRY(launch_split3(m.packed_dev + g.w_off, m.packed_dev + g.uw3_off, g.NT, g.Q, g.uQ16, 1, nullptr, s), "split-bf16 projection weights (row GEMM)");
    if (g.fused) {
        K_TRY(launch_split3(m.packed_dev + g.w_off, m.packed_dev + g.w3_off, g.NT, g.Q, g.Q16, 1, nullptr, s), "split-bf16 projection weights");
        float* sc = m.packed_dev + g.gscale_off;
        HIP_TRY(hipMemsetAsync(sc, 0, 4 * sizeof(float), s));
        K_TRY(launch_absmax(m.packed_dev + g.w_off, (long)g.NT * g.Q * 256, sc, s), "projection weight range");
        K_TRY(launch_scale_from_max(sc, s), "projection weight scale");
        if (m.cfg.use_gatv2)     // k_gath's pack: the compact column order (non-negative group padded to 2, not 8)
            K_TRY(launch_split2h_gath(m.packed_dev + g.w_off, m.packed_dev + g.w2h_off, g.NT_L, g.Q, g.Q16,
                                      reinterpret_cast<const int*>(m.packed_dev + g.ord_off), g.E, sc + 1, s), "split-fp16 projection weights (compact)");
        else
            K_TRY(launch_split2h(m.packed_dev + g.w_off, m.packed_dev + g.w2h_off, g.NT, g.Q, g.Q16, 1, sc + 1, s), "split-fp16 projection weights");
    }
    m.split_ver_gat[which] = m.weights_version;
    return 0;
}
int ensure_lin_split(Model& m, const LinPlan& p, hipStream_t s) {
    if (!p.w3_off || p.Q16 <= 0 || p.w3_version == m.weights_version) return 0;
    K_TRY(launch_split3(m.packed_dev + p.w_off, m.packed_dev + p.w3_off, p.NT, p.Q, p.Q16, 1, nullptr, s), "split-bf16 Linear weights");
    p.w3_version = m.weights_version;
    return 0;
}
// everything an inference call may read, on ONE stream (no-ops while the weights are unchanged)
int ensure_all_split(Model& m, hipStream_t s) {
    int rc;
    for (const GruPlan& g : m.gru) {
        if ((rc = ensure_gru_split(m, g, s))) return rc;
        if (g.has_xproj && (rc = ensure_lin_split(m, g.xproj, s))) return rc;
    }
    for (const GruPlan& g : m.rec)
        if ((rc = ensure_gru_split(m, g, s))) return rc;
    if ((rc = ensure_lin_split(m, m.rec_fc, s))) return rc;
    if ((rc = ensure_conv_split(m, s))) return rc;
    if ((rc = ensure_gat_split(m, m.feat, s)) || (rc = ensure_gat_split(m, m.temp, s))) return rc;
    return 0;
}
// geo (optional): the rows are cut into `geo_W`-row windows instead of the model's W-row ones (run_conv_shared)
int run_conv(Model& m, const XSource& src, int64_t c0, int64_t n, float* xc, float* xct, float* hcat, float* y, hipStream_t s,
             unsigned* vmax = nullptr, int64_t geo_W = 0, bool keep_vmax = false) {
    Scope sc(m, S_CONV, s);
    ConvArgs a{};
    const int64_t Wk = geo_W ? geo_W : m.W;
    if (src.gather) {
        a.X = src.x; a.gather = 1;
        a.starts = src.starts ? reinterpret_cast<const long*>(src.starts + c0) : nullptr;
        a.start0 = src.start0 + c0 * src.stride; a.stride = src.stride;
    } else if (src.x_bf16) {
        a.X = reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(src.x) + c0 * (int64_t)m.W * m.F);
    } else {
        a.X = src.x + c0 * (int64_t)m.W * m.F;
    }
    a.x_bf16 = src.x_bf16;
    a.B = n; a.W = (int)Wk; a.F = m.F; a.Fp = m.Fp; a.taps = m.taps; a.pad = m.pad;
    a.Fq = m.Fp;
    a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + m.conv_w_off);
    // bf16 operand build: inference forward only (hcat / y outputs), when the LDS-staged kernel applies
    if (m.precision == 1 && !xc && !xct && (size_t)(32 + m.taps - 1) * (m.Fp16 + 4) * sizeof(float) <= 20 * 1024 &&
        !(conv_win_selected(m, n) && !geo_W && hcat && !y)) {
        a.bf16 = 1; a.Fq = m.Fp16;
        a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + m.conv_w16_off);
    }
    // (split-bf16 operands were tried for the convolution as well: 3.36 vs 3.37 ms at the flagship shape -- it is bound by its
    // staging and stores, not by the matrix pipe; the fp32 MFMA build stays)
    a.bias = m.packed_dev + m.conv_b_off;
    a.NT = m.convNT;
    a.XC = xc; a.XCT = xct; a.Wpad = m.Wp; a.HCAT = hcat; a.Dp = m.Dp; a.Y = y;
    if (vmax) {
        if (!keep_vmax) HIP_TRY(hipMemsetAsync(vmax, 0, sizeof(unsigned), s));
        a.vmax = vmax;
    }
    // mode 2, the fused front end's call (h_cat only) on whole windows: the window-per-workgroup kernel on fp16 pieces
    if (conv_win_selected(m, n) && !geo_W && hcat && !xc && !xct && !y) {
        ConvArgs b = a;
        b.Fq = m.Fp16;
        if (int rc_ = ensure_conv_split(m, s)) return rc_;
        b.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + m.conv_w2h_off);
        b.wscale = m.packed_dev + m.conv_scale_off + 1;
        if (conv_win_applies(b)) {
            K_TRY(launch_conv_win(b, s), "conv (window per workgroup)");
            return 0;
        }
    }
    // wide models (rows too long for the LDS-staged kernels): the straight-from-memory kernel on three bf16 pieces per operand
    if (m.precision == 2 && !a.bf16 && !src.x_bf16 && m.conv_kernel != 1 && (n * Wk >= 65536 || m.conv_kernel == 2) &&
        (size_t)(32 + m.taps - 1) * (m.Fp + 4) * sizeof(float) > 20 * 1024) {
        if (int rc_ = ensure_conv_split(m, s)) return rc_;
        a.Wp3 = reinterpret_cast<const f32x4*>(m.packed_dev + m.conv_w3_off);
        a.Fq = m.Fp16;
    }
    K_TRY(launch_conv(a, s), "conv");
    return 0;
}

// The convolution of n stride-1 windows of a series without computing a shared row more than once (SURVEY section 8f row 3:
// interior rows are shared by up to W windows; the per-window zero padding, modules.py:14,20, makes the first / last `pad`
// rows of every window its own).  Three launches of the SAME kernel -- so every value equals the per-window launch's, bit
// for bit: the segment as one (n + W - 1)-row window, the windows' first and last 2 pad rows as 2 pad-row windows -- and a
// copy that places the rows into h_cat.  Applies when the LDS-staged kernel does (it records the output range).
bool conv_shared_applies(const Model& m, const XSource& src, int64_t n) {
    if (!src.gather || src.starts || src.stride != 1 || n < 1024 || m.precision == 1 || src.x_bf16) return false;   // (the bf16 build writes h_cat only)
    // the window-per-workgroup kernel reads its window out of the series itself (the rows it re-reads are L2 hits) and is the
    // faster of the two where it runs; taking it here also keeps forward_series == forward on the stacked windows bit for bit
    if (conv_win_selected(m, n) && m.conv_shared != 1) {
        ConvArgs b{};
        b.F = m.F; b.Fp = m.Fp; b.Fq = m.Fp16; b.taps = m.taps; b.pad = m.pad; b.W = m.W; b.Dp = m.Dp; b.NT = m.convNT;
        b.HCAT = m.packed_dev; b.wscale = m.packed_dev;      // (only looked at for presence)
        if (conv_win_applies(b)) return false;
    }
    if (m.taps != 2 * m.pad + 1 || m.pad < 1 || m.W < 4 * m.pad) return false;
    return (size_t)(32 + m.taps - 1) * (m.Fp + 4) * sizeof(float) <= 20 * 1024;
}
int run_conv_shared(Model& m, const XSource& src, int64_t c0, int64_t n, float* hcat, float* cf, float* el, float* er, hipStream_t s,
                    unsigned* vmax) {
    int rc;
    XSource seg = src;
    seg.start0 = src.start0 + c0; seg.stride = 1;
    const int64_t L = n + m.W - 1, EW = 2 * m.pad;
    if ((rc = run_conv(m, seg, 0, 1, cf, nullptr, nullptr, nullptr, s, vmax, L))) return rc;
    if ((rc = run_conv(m, seg, 0, n, el, nullptr, nullptr, nullptr, s, vmax, EW, true))) return rc;
    seg.start0 += m.W - EW;
    if ((rc = run_conv(m, seg, 0, n, er, nullptr, nullptr, nullptr, s, vmax, EW, true))) return rc;
    Scope sc(m, S_CONV, s);
    K_TRY(launch_conv_scatter(cf, el, er, hcat, n, m.W, m.F, m.Fp, m.Dp, m.pad, s), "conv row placement");
    return 0;
}

// The sign-group boundaries handed to the attention kernels by VALUE: GATv2 layers keep the authoritative [P8, PT] in the packed image
// (the device-side re-pack moves them without telling the host), every kernel reads them through its `ord` pointer, and the by-value
// copies are poisoned so that a path that forgot `ord` fails its parity tests instead of using a stale order; GAT (v1) has no
// sign groups (0, 0).
static int pt_by_value(const Model& m, const GatPlan& g) { return m.cfg.use_gatv2 ? -1 : g.PT; }
static int p8_by_value(const Model& m, const GatPlan& g) { return m.cfg.use_gatv2 ? -1 : g.P8; }

int run_proj(Model& m, const GatPlan& g, const float* rows, long ld, int64_t nrows, float* lc, float* rt, hipStream_t s) {
    Scope sc(m, S_PROJ, s);
    RowGemmArgs a{};
    a.X = rows; a.ldx = ld; a.Kvalid = g.D; a.Q = g.Q;
    a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + g.w_off);
    a.bias = m.packed_dev + g.b_off;
    a.Y = lc; a.ldy = g.ldl; a.Nvalid = g.ldl; a.vec_store = 1;
    a.R = nrows; a.NT = g.NT; a.relu = 0;
    a.NT_rm = g.NT_L; a.YT = rt; a.group = g.K; a.YT_rows = g.rt_rows; a.YT_ld = g.Kp;
    if (m.precision == 2 && g.uw3_off && g.uQ16 > 0 && m.rowgemm_kernel != 1 && (nrows >= 65536 || m.rowgemm_kernel == 2)) {       // wide layers: three bf16 pieces per operand
        if (int rc_ = ensure_gat_split(m, g, s)) return rc_;
        a.x3 = 1; a.Q16 = g.uQ16;
        a.Wp3 = reinterpret_cast<const f32x4*>(m.packed_dev + g.uw3_off);
    }
    K_TRY(launch_rowgemm(a, s), "gat projection");
    return 0;
}

int run_attend(Model& m, const GatPlan& g, const float* lc, const float* rt, const float* v, int ldv, int64_t n, float* out,
               long so_w, long so_i, long so_d, hipStream_t s, float* att = nullptr, const DropArgs* drop = nullptr, unsigned drop_stream = 0) {
    Scope sc(m, S_ATTEND, s);
    if (g.K <= 512 && g.D <= 512) {
        // LDS-tiled pair grid of the fused kernel over the HBM-resident projections (BASELINE config 4 shapes)
        K_TRY(launch_gat_wide(lc, rt, g.ldl, g.rt_rows, g.Kp, pt_by_value(m, g), p8_by_value(m, g), m.packed_dev + g.bias_off, v, ldv, g.D, g.K, out, so_w,
                              so_i, so_d, n, m.cfg.use_gatv2 ? 0 : 1, m.cfg.alpha, s, att, drop, drop_stream,
                              m.cfg.use_gatv2 ? reinterpret_cast<const int*>(m.packed_dev + g.ord_off) : nullptr),
              "wide gat attention");
        return 0;
    }
    if (att || drop) return fail(MTADGAT_ERR_UNSUPPORTED, "training forward of an attention layer with more than 512 nodes / features");
    AttendArgs a{};
    a.LC = lc; a.RT = rt; a.ldl = g.ldl; a.rt_rows = g.rt_rows; a.Kp = g.Kp; a.PT = pt_by_value(m, g); a.P8 = p8_by_value(m, g);
    a.ord = m.cfg.use_gatv2 ? reinterpret_cast<const int*>(m.packed_dev + g.ord_off) : nullptr;
    a.bias = m.packed_dev + g.bias_off;
    a.V = v; a.ldv = ldv; a.D = g.D;
    a.out = out; a.so_w = so_w; a.so_i = so_i; a.so_d = so_d;
    a.K = g.K; a.rows_per_blk = g.rows_per_blk; a.nblk = g.nblk;
    a.nwin = n;
    a.xcd_map = 1;
    a.total_blocks = ((n + 7) / 8 * 8) * g.nblk;
    a.v1 = m.cfg.use_gatv2 ? 0 : 1;
    a.alpha = m.cfg.alpha;
    a.ATT = nullptr;
    K_TRY(launch_attend(a, g.IB, s), "gat attention");
    return 0;
}

bool use_fused(const GatPlan& g) { return g.fused; }

// fused layer: V rows (n*K, ldv) -> out, nothing but V read from / out written to HBM
// cv (temporal layer, inference): the window convolution runs inside k_gath's workgroup (fused_conv_args below said it can)
int run_gat_fused(Model& m, const GatPlan& g, const float* v, int ldv, int vt, int64_t n, float* out, long so_w, long so_i,
                  long so_d, hipStream_t s, float* att = nullptr, const DropArgs* drop = nullptr, unsigned drop_stream = 0,
                  const unsigned* vmax = nullptr, const GatConvIn* cv = nullptr) {
    Scope sc(m, S_ATTEND, s);
    GatArgs a{};
    a.V = v; a.ldv = ldv; a.vt = vt; a.D = g.D; a.K = g.K; a.vld = g.f_vld; a.lr_floats = g.f_lr;
    a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + g.w_off);
    a.pbias = m.packed_dev + g.b_off;
    a.NT_L = g.NT_L; a.Q = g.Q; a.PT = pt_by_value(m, g); a.P8 = p8_by_value(m, g);
    a.ord = m.cfg.use_gatv2 ? reinterpret_cast<const int*>(m.packed_dev + g.ord_off) : nullptr;
    if (m.precision == 1 && !att && !(split_front(m, n) && vmax)) {       // bf16 operand build of the projection (inference)
        a.bf16 = 1; a.Q = g.Q16;
        a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + g.w16_off);
    } else if ((m.precision == 2 || (split_front(m, n) && !att)) && (n >= 4096 || (m.gat_kernel == 3 && !att))) {
        // large batches: split-bf16 operands for the projection -- fp32-class L' / R' on the bf16 matrix pipe, which runs
        // beside the pair grid of the other waves (the fp32 MFMA does not: profiles/r02_mfma_valu_overlap.txt).  Measured at
        // (W=100, F=55): feature layer 5.90 -> 5.09 ms, temporal layer 7.40 -> 6.94 ms (two weight chunks in registers; with
        // four the temporal layer's larger pair-grid block spilled and lost)
        if (int rc_ = ensure_gat_split(m, g, s)) return rc_;
        a.bf16 = 2; a.Q = g.Q16;
        a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + g.w3_off);
        // two fp16 pieces instead when the convolution that produced the node values recorded a maximum below 2^15
        a.vmax = vmax;
        a.Wp2 = reinterpret_cast<const f32x4*>(m.packed_dev + g.w2h_off);
        a.scale2 = m.packed_dev + g.gscale_off + 1;
    }
    a.bias = m.packed_dev + g.bias_off;
    a.out = out; a.so_w = so_w; a.so_i = so_i; a.so_d = so_d;
    a.nwin = n;
    a.v1 = m.cfg.use_gatv2 ? 0 : 1;
    a.alpha = m.cfg.alpha;
    a.ATT = att;
    if (drop) a.drop = *drop;
    a.drop_stream = drop_stream;
    // the fp16-piece build of the row-split kernel (node vectors split once per window) serves the two-fp16-piece arithmetic
    // (inference, node values below 2^15 -- decided on the device from the convolution's recorded maximum: of the two launches
    // exactly one does the work)
    const int scols = vt ? g.K : g.D;
    // (training forward -- att set -- from 4096 windows as well: k_gath keeps the softmax rows and applies the dropout)
    if (a.bf16 == 2 && vmax && (!att || n >= 4096) && (m.gat_kernel == 0 || m.gat_kernel == 3) && g.fh_lds_bytes <= 160 * 1024 && aligned16(v) &&
        (ldv & 3) == 0 && ((scols + 3) & ~3) <= ldv) {
        GatArgs b = a;
        b.vld = g.fh_vld; b.lr_floats = g.fh_lr; b.n_full = g.fh_full; b.n_short = g.fh_short;
        b.lr_buf = g.fh_lr_buf;
        b.E = m.cfg.use_gatv2 ? g.E : 0;
        b.dbg = m.gath_dbg;              // (measurement hook: mtadgat_set_option "gath_dbg", profiles/gath_knockout.py)
        if (cv) b.cv = *cv;
        K_TRY(launch_gath(b, g.fh_IBL, g.fh_JPL, g.fh_RJ, g.f_nw, g.fh_lds_bytes, cv != nullptr, s), cv ? "fused convolution + gat (fp16 pieces)" : "fused gat (fp16 pieces)");
        a.skip_h = 1;
        if (cv) a.winflag = cv->flag;    // per-window range guard: k_gat serves exactly the windows k_gath flagged
    } else if (cv) {
        return fail(MTADGAT_ERR_INVALID, "internal: fused convolution without k_gath");
    }
    K_TRY(launch_gat(a, g.f_IBL, g.f_JPL, g.f_RJ, g.f_nw, g.f_lds_bytes, s), "fused gat");
    return 0;
}

// Can the temporal layer's k_gath workgroup compute the convolution of its window itself (mtadgat_gath.hip, CONV build)?  The
// conditions of k_conv_win and of k_gath's launch in run_gat_fused, and the staged input must fit the L' / R' region.  Fills `cv`.
bool fused_conv_args(const Model& m, const XSource& src, int64_t c0, int64_t n, float* hcat, unsigned* vmax, unsigned char* flag, GatConvIn& cv) {
    const GatPlan& g = m.temp;
    if (m.conv_fused == 1 || !(m.precision == 2 || split_front(m, n)) || !conv_win_selected(m, n) || !g.fused || !m.feat.fused) return false;
    if (!(n >= 4096 || m.gat_kernel == 3) || !(m.gat_kernel == 0 || m.gat_kernel == 3) || g.fh_lds_bytes > 160 * 1024) return false;
    if (g.K != m.W || g.D != m.F || !aligned16(hcat) || (m.Dp & 3) != 0 || ((g.D + 3) & ~3) > m.Dp) return false;
    cv = GatConvIn{};
    if (src.gather) {
        cv.X = src.x; cv.gather = 1;
        cv.starts = src.starts ? reinterpret_cast<const long*>(src.starts + c0) : nullptr;
        cv.start0 = src.start0 + c0 * src.stride; cv.stride = src.stride;
    } else if (src.x_bf16) {
        cv.X = reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(src.x) + c0 * (int64_t)m.W * m.F);
    } else {
        cv.X = src.x + c0 * (int64_t)m.W * m.F;
    }
    cv.x_bf16 = src.x_bf16;
    cv.taps = m.taps; cv.pad = m.pad; cv.Fq = m.Fp16; cv.NT = m.convNT; cv.Dp = m.Dp;
    cv.pvx = conv_win_pitch(m.F, m.Fp16);
    cv.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + m.conv_w2h_off);
    cv.bias = m.packed_dev + m.conv_b_off;
    cv.wscale = m.packed_dev + m.conv_scale_off + 1;
    cv.HCAT = hcat; cv.vmax = vmax; cv.flag = flag;
    GatArgs probe{};
    probe.vt = 0; probe.K = g.K; probe.D = g.D; probe.lr_floats = g.fh_lr; probe.cv = cv;
    return gath_conv_applies(probe, g.f_nw, m.F, m.W);
}

// one graph-attention layer from its node rows, fused when the plan allows
int run_gat_layer(Model& m, const GatPlan& g, const float* v, int ldv, int64_t n, float* lc, float* rt, float* out, long so_w,
                  long so_i, long so_d, hipStream_t s) {
    if (use_fused(g)) return run_gat_fused(m, g, v, ldv, 0, n, out, so_w, so_i, so_d, s);
    int rc = run_proj(m, g, v, ldv, n * g.K, lc, rt, s);
    if (rc) return rc;
    return run_attend(m, g, lc, rt, v, ldv, n, out, so_w, so_i, so_d, s);
}

// the small-batch fp32 recurrence kernels apply to a single-layer stack whose weights fit a wave's registers
// (inference in the bf16 mode uses k_gru1 up to 1024 windows -- faster there than the bf16 build of the throughput
// kernels, and exact; the training step keeps the bf16 recurrences it was asked for)
// the part of run_gru_layer's `cm_fit` that does not depend on the call: can the split-operand kernels (k_gru_cm / the X3H
// build of k_gru_split) serve this layer at all?  (hidden sizes above 160, wide windows, input packs that need the range guard
// without a fused front end cannot: such layers keep k_gru16 up to G16_MAX_WINDOWS)
// A Linear over (window, step) rows in the default fp32 arithmetic: from 65 536 rows on the products come from three bf16 pieces per
// operand on the 16-bit matrix pipe (k_rowgemm_x3 / _x3s: 2.7 x less matrix time than the fp32 MFMA, results <= 2e-7 apart), with the
// split pack derived on first use after an upload -- as the wide attention layers' projections and the backward's data gradients
// do.  Round 6: the GRU's hoisted input projection and recon_model.fc ran on the fp32 MFMA whatever the size (config 4: two 2.3 ms
// launches per 896-window chunk, 42 of its 268 ms per 8 192 windows).
int lin_split_operands(Model& m, const LinPlan& p, long rows, RowGemmArgs& a, hipStream_t s) {
    if (!p.w3_off || p.Q16 <= 0 || m.precision != 2 || m.rowgemm_kernel == 1 || !(rows >= 65536 || m.rowgemm_kernel == 2)) return 0;
    if (int rc_ = ensure_lin_split(m, p, s)) return rc_;
    a.x3 = 1; a.Q16 = p.Q16;
    a.Wp3 = reinterpret_cast<const f32x4*>(m.packed_dev + p.w3_off);
    return 0;
}

bool split_kernels_fit(const Model& m, const GruPlan& g) {
    if (m.W > 512 || !(g.Qxp16 == 1 || g.Qxp16 % 2 == 0)) return false;
    if (g.xmode == 1) return g.Qxp16 == 1 && gru_cm_supported(g.NCG, 1, false, 0);
    const bool guard = use_fused(m.temp) && use_fused(m.feat);           // the convolution records its output range only there
    return gru_cm_supported(g.NCG, 0, false, 0) && g.wxq_off != 0 && g.Qx >= 3 && ((guard && g.wx2_off && g.qb3 > 0) || g.qb3 == 0);
}

bool use_g16(const Model& m, const std::vector<GruPlan>& stack, int64_t n, bool training = false) {
    if (stack.size() != 1 || !stack[0].has16) return false;
    if (m.precision == 1) return !training && n <= 1024;
    // inference in the split-operand arithmetic: from SPLIT3_MIN_WINDOWS on the hidden-tile-split kernel's split-operand build
    // is the faster one (run_gru_layer) -- where it applies; the measurement hook can force it at any size
    if (m.precision == 2 && !training && stack[0].NCG >= 2 && split_kernels_fit(m, stack[0]) &&
        (m.gru_kernel == 3 || (m.gru_kernel == 0 && n >= SPLIT3_MIN_WINDOWS))) return false;
    return n <= G16_MAX_WINDOWS;
}

// ... and below G1_MAX_WINDOWS a workgroup takes one window at a time
bool use_g1(int64_t n) {
    return n <= G1_MAX_WINDOWS;
}

// one GRU layer.  x: rows (n*T, ldx) for xmode 0, hin (n, ldx) for xmode 1
int run_gru_layer(Model& m, int slot, const GruPlan& g, const float* x, long ldx, int kx, int64_t n, float* hend,
                  long ldhe, float* seq, const LinPlan* fc, float* yfc, float* ylast, hipStream_t s, float* gates = nullptr,
                  float* xp = nullptr, bool g16 = false, const unsigned* vmax = nullptr) {
    Scope sc(m, slot, s);
    int xmode = g.xmode;
    if (g16) {
        // small batch, fp32: input products of all steps ahead of the recurrence, 16-window groups with register-resident
        // weights (mtadgat_gru16.hip); a per-step Linear is the caller's row GEMM over `seq`
        if (!xp || !g.has16 || fc) return fail(MTADGAT_ERR_INVALID, "internal: k_gru16 without its buffers");
        if (g.xmode == 0) {
            RowGemmArgs r{};
            r.X = x; r.ldx = ldx; r.Kvalid = g.in_dim; r.Q = g.xproj.Q;
            r.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + g.xproj.w_off);
            r.bias = m.packed_dev + g.xproj.b_off;
            r.Y = xp; r.ldy = 3L * g.Hp; r.Nvalid = 3 * g.Hp; r.vec_store = 1;
            r.R = n * m.W; r.NT = g.xproj.NT; r.NT_rm = g.xproj.NT; r.group = 1; r.relu = 0;
            if (int rc_ = lin_split_operands(m, g.xproj, r.R, r, s)) return rc_;
            K_TRY(launch_rowgemm(r, s), "gru input projection");
        } else {
            K_TRY(launch_xproj_dec(x, ldx, kx, m.packed_dev + g.fold_off, reinterpret_cast<const int*>(m.packed_dev + g.m0_off),
                                   m.packed_dev + g.b_off, g.Hp, m.W, n, xp, s), "decoder input projection");
        }
        Gru16Args a{};
        a.XP = xp; a.W16 = m.packed_dev + g.g16_off; a.bias = m.packed_dev + g.b_off;
        a.Hp = g.Hp; a.KS = g.KS16; a.NT16 = g.NT16; a.T = m.W; a.B = n;
        a.Hend = hend; a.ldhe = ldhe; a.ncol = (int)std::min<long>(ldhe, g.Hp); a.Seq = seq; a.Gates = gates; a.H = g.H;
        if (use_g1(n)) {
            a.W16 = m.packed_dev + g.g1_off;
            K_TRY(launch_gru1(a, s), "gru (window per workgroup)");
            return 0;
        }
        K_TRY(launch_gru16(a, s), "gru (16-window groups)");
        return 0;
    }
    // chunk-major recurrence (k_gru_cm): 32 windows per wave, the weight stream shared by a workgroup's waves through LDS.
    // It needs two-piece (fp16) input operands: the decoder's and stacked layers' inputs always are; layer 0's when the
    // convolution's recorded range allows -- decided on the device: both kernels are launched, each returns at once when
    // the launch is the other's.  Measured against the kernels it replaces (MSL shape, GRU layer + decoder): 12 288 windows
    // 5.9 -> 4.0 ms, 32 768: 7.3 -> 4.4, 65 536: 12.0 -> 9.1; from CM_MIN_WINDOWS on.
    const bool cm_fit = m.precision == 2 && !gates && gru_cm_supported(g.NCG, g.xmode, fc != nullptr, fc ? fc->out_dim : 0) &&
                        (g.Qxp16 == 1 || g.Qxp16 % 2 == 0) &&
                        (g.xmode == 1 ? g.Qxp16 == 1 : (g.wxq_off != 0 && g.Qx >= 3 && ((vmax && g.wx2_off && g.qb3 > 0) || g.qb3 == 0))) &&
                        (hend == nullptr || ldhe >= g.Hp) && m.W <= 512;
    // ... and between CM_MIN_WINDOWS and SPLIT3_MAX_WINDOWS the split-operand build of the hidden-tile-split kernel (same
    // packs, same range guard): there a 32-window wave per SIMD is a latency chain, five waves per 32 windows are not
    const bool use_sp = cm_fit && g.NCG >= 2 && (m.gru_kernel == 3 || (m.gru_kernel == 0 && n >= SPLIT3_MIN_WINDOWS && n <= SPLIT3_MAX_WINDOWS));
    const bool use_cm = cm_fit && !use_sp && m.gru_kernel != 1 && m.gru_kernel != 3 && (m.gru_kernel == 2 || n >= CM_MIN_WINDOWS);
    // large batch, split-bf16 operands: fp32-class results from the bf16 matrix pipe (k_gru X3 build), from 1.25 32-window
    // groups per CU on (measured: 12 320 windows 10.5 ms against 13.1 ms for the hidden-tile-split kernel at 12 288; 8 192 windows 7.4 ms there)
    // training forward (gates kept for the backward): the same split-operand build of the hidden-tile-split kernel from
    // SPLIT3_MIN_WINDOWS on -- the input part in the kernel (no pre-projection GEMM on the fp32 pipe), any per-step Linear
    const bool sp_train = gates != nullptr && m.precision == 2 && g.NCG >= 2 && m.gru_kernel != 1 && n >= SPLIT3_MIN_WINDOWS &&
                          (g.Qxp16 == 1 || g.Qxp16 % 6 == 0) &&
                          (g.xmode == 1 ? g.Qxp16 == 1 : ((vmax && g.wx2_off && g.qb3 > 0) || g.qb3 == 0)) && (hend == nullptr || ldhe >= g.Hp) &&
                          ((size_t)g.NCG * 1024 + (fc ? (size_t)g.NCG * fc->out_dim * 32 : 0)) * sizeof(float) <= 64 * 1024;
    const bool x3 = m.precision == 2 && !gates && ((n + 31) / 32 > 5L * cu_count() / 4 || use_cm || use_sp) &&
                    (g.Qxp16 == 1 || g.Qxp16 % 2 == 0);
    if (sp_train) xp = nullptr;
    if (!x3 && xp && g.has_xproj && g.xmode == 0) {
        // small batch: all steps' input products as one throughput GEMM, the recurrence keeps only its h part
        RowGemmArgs r{};
        r.X = x; r.ldx = ldx; r.Kvalid = g.in_dim; r.Q = g.xproj.Q;
        r.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + g.xproj.w_off);
        r.bias = m.packed_dev + g.xproj.b_off;
        r.Y = xp; r.ldy = 3L * g.Hp; r.Nvalid = 3 * g.Hp; r.vec_store = 1;
        r.R = n * m.W; r.NT = g.xproj.NT; r.NT_rm = g.xproj.NT; r.group = 1; r.relu = 0;
        if (int rc_ = lin_split_operands(m, g.xproj, r.R, r, s)) return rc_;
        K_TRY(launch_rowgemm(r, s), "gru input projection");
        x = xp; ldx = 3L * g.Hp; xmode = 3;
    }
    GruArgs a{};
    a.X = x; a.ldx = ldx; a.Kx = kx; a.Qx = g.Qx; a.Qxp = g.Qxp;
    a.m0 = xmode == 1 ? reinterpret_cast<const int*>(m.packed_dev + g.m0_off) : nullptr;
    a.Wx = reinterpret_cast<const f32x4*>(m.packed_dev + g.wx_off);
    a.Wh = reinterpret_cast<const f32x4*>(m.packed_dev + g.wh_off);
    a.bias = m.packed_dev + g.b_off;
    a.whs = 4 * g.NCG + 2;
    if (m.precision == 1) {       // bf16 operand build: the same streams in 16-feature chunks (inference, and the bf16 training step)
        a.Wx = reinterpret_cast<const f32x4*>(m.packed_dev + g.wx16_off);
        a.Wh = reinterpret_cast<const f32x4*>(m.packed_dev + g.wh16_off);
        a.whs = 2 * g.NCG + 2;
        a.Qxp = g.Qxp16;
        a.bf16 = 1;
    }
    if (x3 || sp_train) {
        if (int rc_ = ensure_gru_split(m, g, s)) return rc_;
    }
    if (x3) {
        a.Wx = reinterpret_cast<const f32x4*>(m.packed_dev + g.wx3_off);
        a.Wh = reinterpret_cast<const f32x4*>(m.packed_dev + g.wh3_off);
        a.whs = 2 * g.NCG + 2;
        a.Qxp = g.Qxp16;
        a.bf16 = 1;
        a.x3 = 1;
        a.scale = m.packed_dev + g.scale_off + 1;
        a.qb3 = g.qb3;
        if (vmax && g.wx2_off && g.qb3 > 0) {
            a.vmax = vmax;
            a.Wx2 = reinterpret_cast<const f32x4*>(m.packed_dev + g.wx2_off);
        }
    }
    a.Hp = g.Hp; a.H = g.H; a.T = m.W; a.B = n;
    a.Hend = hend; a.ldhe = ldhe;
    a.Seq = seq; a.ldseq = g.Hp;
    if (fc) {
        a.Wfc = reinterpret_cast<const f32x4*>(m.packed_dev + fc->w_off);
        a.bfc = m.packed_dev + fc->b_off;
        a.NTfc = fc->NT;
        a.Yfc = yfc;
        a.Ylast = ylast;
        a.out_dim = fc->out_dim;
    }
    if (gates) {     // training forward: keep the gate activations of every step
        a.Gates = gates;
        if (sp_train) {
            GruArgs a3 = a;
            a3.Wh = reinterpret_cast<const f32x4*>(m.packed_dev + g.wh3_off);
            a3.whs = 2 * g.NCG + 2; a3.Qxp = g.Qxp16; a3.bf16 = 1; a3.x3 = 1;
            a3.scale = m.packed_dev + g.scale_off + 1;
            const bool guard = xmode == 0 && g.qb3 > 0;              // layer 0: the convolution's channels need the recorded range
            a3.vmax = guard ? vmax : nullptr;
            a3.Wxq = reinterpret_cast<const f32x4*>(m.packed_dev + (guard ? g.wx2_off : g.wx3_off));
            K_TRY(launch_gru_split_x3(a3, g.NCG, xmode, fc != nullptr, s), "gru (training, split operands)");
            if (!guard) return 0;
            a.vmax = vmax; a.skip_xh = 1;                            // the fp32 kernel serves the launch when the range is too large
        }
        K_TRY(launch_gru_train(a, g.NCG, xmode, fc != nullptr, s), "gru (training)");
        return 0;
    }
    if (use_sp) {
        a.Wxq = (xmode == 0 && a.Wx2) ? a.Wx2 : a.Wx;         // the two-piece input pack in [tile][chunk] order
        const size_t lds = ((size_t)g.NCG * 1024 + (fc ? (size_t)g.NCG * fc->out_dim * 32 : 0)) * sizeof(float);
        if (lds <= 64 * 1024) {
            K_TRY(launch_gru_split_x3(a, g.NCG, xmode, fc != nullptr, s), "gru (hidden-tile split, split operands)");
            if (a.vmax == nullptr) return 0;
            a.skip_xh = 1;
        }
    } else if (use_cm) {
        a.Wxq = xmode == 1 ? a.Wx : reinterpret_cast<const f32x4*>(m.packed_dev + g.wxq_off);
        K_TRY(launch_gru_cm(a, g.NCG, xmode, fc != nullptr, s), "gru (chunk-major)");
        if (a.vmax == nullptr) return 0;
        a.skip_xh = 1;
    }
    K_TRY(launch_gru(a, g.NCG, xmode, fc != nullptr, s), "gru");
    return 0;
}

// hcat is the internal (n*W, Dp) buffer: 16-byte aligned rows whose pad columns are zero (the kernels read them unguarded)
int run_gru_stack(Model& m, const float* hcat, long ldx, int64_t n, float* hend, long ldhe, float* ws,
                  const Workspace& o, hipStream_t s, const unsigned* vmax = nullptr) {
    const int L = (int)m.gru.size();
    const float* x = hcat;
    long ld = ldx;
    int kx = 3 * m.F;
    for (int l = 0; l < L; ++l) {
        const bool last = (l == L - 1);
        float* seq = last ? nullptr : ws + ((l & 1) ? o.seq1 : o.seq0);
        const bool g16 = use_g16(m, m.gru, n) && o.has_xp;
        float* xp = (l == 0 && o.has_xp && (g16 || n <= gru_split_max_windows())) ? ws + o.xp : nullptr;
        int rc = run_gru_layer(m, S_GRU, m.gru[l], x, ld, kx, n, last ? hend : nullptr, ldhe, seq, nullptr, nullptr, nullptr, s, nullptr, xp, g16,
                               l == 0 ? vmax : nullptr);
        if (rc) return rc;
        x = seq; ld = m.gru[l].Hp; kx = m.gru[l].H;      // sequence buffers hold all Hp columns, padding lanes are exact zeros
    }
    return 0;
}

// hend is the internal (n, Hp) buffer with zero pad columns
int run_heads(Model& m, const float* hend, long ldh, int64_t n, float* preds, float* recons, float* recons_last,
              float* ws, const Workspace& o, hipStream_t s) {
    if (preds) {
        Scope sc(m, S_FC, s);
        const float* x = hend;
        long ld = ldh;
        const int nfc = (int)m.fc.size();
        for (int i = 0; i < nfc; ++i) {
            const LinPlan& p = m.fc[i];
            const bool last = (i == nfc - 1);
            RowGemmArgs a{};
            a.X = x; a.ldx = ld; a.Kvalid = p.in_dim; a.Q = p.Q;
            a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + p.w_off);
            a.bias = m.packed_dev + p.b_off;
            a.R = n; a.NT = p.NT; a.NT_rm = p.NT; a.group = 1;
            if (last) {
                a.Y = preds; a.ldy = p.out_dim; a.Nvalid = p.out_dim;
                a.vec_store = (p.out_dim % 4 == 0 && aligned16(preds)) ? 1 : 0;
                a.relu = 0;
            } else {
                a.Y = ws + ((i & 1) ? o.fc1 : o.fc0); a.ldy = p.NT * 32; a.Nvalid = p.NT * 32; a.vec_store = 1;
                a.relu = 1;   // eval mode: dropout is the identity (reference modules.py:309-310)
            }
            K_TRY(launch_rowgemm(a, s), "forecasting head");
            x = a.Y; ld = a.ldy;
        }
    }
    if (recons || recons_last) {
        const int L = (int)m.rec.size();
        const float* x = hend;
        long ld = ldh;
        int kx = m.cfg.gru_hid_dim;
        // decoder layer 0 reads hend[m0(t) .. m0(t) + 8*Qx), clamped to the (zero padded) row inside the kernel
        // recon_model.fc (modules.py:282): with few outputs (target dims of MSL / SMAP) it rides inside the recurrence;
        // otherwise the last layer's states go to memory and the Linear is one throughput GEMM over the b*W rows --
        // inside the step loop it would sit on the latency chain with 4 * Qh matrix instructions per 32 outputs
        if (use_g16(m, m.rec, n) && o.rec16) {
            // small batch: k_gru16 keeps the states (or, when only the last step is wanted, the last state) and
            // recon_model.fc is a row GEMM over them
            Scope sc(m, S_RECON, s);
            const GruPlan& g = m.rec[0];
            const LinPlan& p = m.rec_fc;
            float* seq = ws + o.rseq0;
            int rc = run_gru_layer(m, S_RECON, g, x, ld, kx, n, recons ? nullptr : seq, g.Hp, recons ? seq : nullptr, nullptr, nullptr, nullptr, s,
                                   nullptr, ws + o.xp, true);
            if (rc) return rc;
            RowGemmArgs a{};
            a.X = seq; a.ldx = g.Hp; a.Kvalid = g.H; a.Q = p.Q;
            a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + p.w_off);
            a.bias = m.packed_dev + p.b_off;
            float* y = recons ? recons : recons_last;
            a.Y = y; a.ldy = p.out_dim; a.Nvalid = p.out_dim;
            a.vec_store = (p.out_dim % 4 == 0 && aligned16(y)) ? 1 : 0;
            a.R = recons ? n * (int64_t)m.W : n; a.NT = p.NT; a.NT_rm = p.NT; a.group = 1; a.relu = 0;
            // (the arithmetic is chosen by the size of the whole reconstruction, also when only its last step is wanted: score_series
            // equals forward() on the same windows bit for bit)
            if (int rc_ = lin_split_operands(m, p, n * (int64_t)m.W, a, s)) return rc_;
            K_TRY(launch_rowgemm(a, s), "reconstruction Linear");
            if (recons && recons_last)
                K_TRY(launch_copy2d(recons + (int64_t)(m.W - 1) * p.out_dim, (long)m.W * p.out_dim, recons_last, p.out_dim, n, p.out_dim, s),
                      "last reconstruction step");
            return 0;
        }
        const bool hoist_fc = m.cfg.out_dim > 4 && recons != nullptr;
        for (int l = 0; l < L; ++l) {
            const bool last = (l == L - 1);
            const bool fc_in = last && !hoist_fc;
            float* seq = (last && !hoist_fc) ? nullptr : ws + ((l & 1) ? o.rseq1 : o.rseq0);
            int rc = run_gru_layer(m, S_RECON, m.rec[l], x, ld, kx, n, nullptr, 0, seq, fc_in ? &m.rec_fc : nullptr,
                                   fc_in ? recons : nullptr, fc_in ? recons_last : nullptr, s);
            if (rc) return rc;
            x = seq; ld = m.rec[l].Hp; kx = m.rec[l].H;
        }
        if (hoist_fc) {
            Scope sc(m, S_RECON, s);
            const LinPlan& p = m.rec_fc;
            RowGemmArgs a{};
            a.X = x; a.ldx = ld; a.Kvalid = kx; a.Q = p.Q;
            a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + p.w_off);
            a.bias = m.packed_dev + p.b_off;
            a.Y = recons; a.ldy = p.out_dim; a.Nvalid = p.out_dim;
            a.vec_store = (p.out_dim % 4 == 0 && aligned16(recons)) ? 1 : 0;
            a.R = n * (int64_t)m.W; a.NT = p.NT; a.NT_rm = p.NT; a.group = 1; a.relu = 0;
            if (int rc_ = lin_split_operands(m, p, a.R, a, s)) return rc_;
            K_TRY(launch_rowgemm(a, s), "reconstruction Linear");
            if (recons_last)
                K_TRY(launch_copy2d(recons + (int64_t)(m.W - 1) * p.out_dim, (long)m.W * p.out_dim, recons_last, p.out_dim, n, p.out_dim, s),
                      "last reconstruction step");
        }
    }
    return 0;
}

int check_common(mtadgat_handle h, int64_t batch, void* ws, size_t ws_bytes, bool need_ws) {
    if (!h) return fail(MTADGAT_ERR_INVALID, "null handle");
    if (batch < 0) return fail(MTADGAT_ERR_INVALID, "negative batch");
    if (!h->m.have_weights) return fail(MTADGAT_ERR_NOWEIGHTS, "mtadgat_load_weights has not been called");
    if (h->m.precision == 1 && !h->m.bf16_packed)
        return fail(MTADGAT_ERR_NOWEIGHTS, "bf16 precision selected after the weights were loaded: call mtadgat_load_weights again");
    if (need_ws && batch > 0) {
        if (!ws) return fail(MTADGAT_ERR_WORKSPACE, "workspace is NULL");
        if (!aligned16(ws)) return fail(MTADGAT_ERR_WORKSPACE, "workspace must be 16-byte aligned");
        if (ws_bytes < mtadgat_workspace_bytes(h, batch)) return fail(MTADGAT_ERR_WORKSPACE, "workspace too small");
    }
    return 0;
}

}  // namespace

extern "C" {

int mtadgat_abi_version(void) { return MTADGAT_ABI_VERSION; }
const char* mtadgat_last_error(void) { return g_err.c_str(); }

int mtadgat_create(const mtadgat_config* cfg, mtadgat_handle* out) {
    if (!cfg || !out) return fail(MTADGAT_ERR_INVALID, "null argument");
    mtadgat_handle h = new (std::nothrow) mtadgat_handle_s();
    if (!h) return fail(MTADGAT_ERR_INVALID, "out of host memory");
    h->m.cfg = *cfg;
    std::string err = validate_and_plan(h->m);
    if (!err.empty()) {
        delete h;
        return fail(MTADGAT_ERR_UNSUPPORTED, err);
    }
    // default chunk: up to 65 536 windows, fewer when a window needs a lot of scratch (wide models on the
    // un-fused attention path: ~8 MB per window at F=512, W=256), so that the workspace stays within a
    // quarter of the device memory (16 GB when no device can be queried)
    {
        Workspace o;
        plan_workspace(h->m, 65536, o);       // large batches: without the small-batch extras (pre-projected inputs, decoder states)
        const double per_window = (double)o.total * sizeof(float) / 65536.0;
        size_t free_b = 0, total_b = 0;
        double budget = (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) ? (double)total_b / 4 : 16.0 * 1024 * 1024 * 1024;
        const bool unfused = !(h->m.temp.fused && h->m.feat.fused);
        if (unfused) budget = std::min(budget, 6.0 * 1024 * 1024 * 1024);   // projections through HBM: a few hundred windows fill the machine
        int64_t c = (int64_t)(budget / per_window);
        const int64_t gran = unfused ? 128 : 2048;
        c = c / gran * gran;
        h->m.chunk = c < gran ? gran : (c > 65536 ? 65536 : c);
    }
    *out = h;
    return 0;
}

static void free_device_tables(Model& m) {
    DevTables& t = m.dt;
    int cur = 0;
    const bool sw = t.device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != t.device;
    if (sw) (void)hipSetDevice(t.device);
    if (t.gidx_dev) (void)hipFree(t.gidx_dev);
    if (t.foldcode_dev) (void)hipFree(t.foldcode_dev);
    if (t.foldsum_dev) (void)hipFree(t.foldsum_dev);
    for (int i = 0; i < 2; ++i) {
        if (t.gatcode_dev[i]) (void)hipFree(t.gatcode_dev[i]);
        if (t.colk_dev[i]) (void)hipFree(t.colk_dev[i]);
        t.gatcode_dev[i] = t.colk_dev[i] = nullptr;
    }
    if (t.pin) (void)hipHostFree(t.pin);
    if (sw) (void)hipSetDevice(cur);
    t.gidx_dev = t.foldcode_dev = nullptr;
    t.foldsum_dev = nullptr;
    t.pin = nullptr;
    t.ready = false;
    t.device = -1;
}

// the second lane's stream and events belong to the device they were created on (a handle may move: mtadgat_load_weights)
static void free_lane(Model& m) {
    if (!m.lane_stream) return;
    int cur = 0;
    const bool sw = m.lane_device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != m.lane_device;
    if (sw) (void)hipSetDevice(m.lane_device);
    (void)hipStreamSynchronize(m.lane_stream);
    (void)hipStreamDestroy(m.lane_stream);
    (void)hipEventDestroy(m.lane_begin);
    (void)hipEventDestroy(m.lane_end);
    for (hipEvent_t& e : m.fork_ev)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (sw) (void)hipSetDevice(cur);
    m.lane_stream = nullptr;
    m.lane_begin = m.lane_end = nullptr;
    m.lane_device = -1;
}

int mtadgat_destroy(mtadgat_handle h) {
    if (!h) return 0;
    free_device_tables(h->m);
    if (h->m.upload_ev) {
        (void)hipEventSynchronize(h->m.upload_ev);
        (void)hipEventDestroy(h->m.upload_ev);
    }
    free_lane(h->m);
    if (h->m.staging_pinned) (void)hipHostFree(h->m.staging_pinned);
    if (h->m.packed_dev) (void)hipFree(h->m.packed_dev);
    for (auto& v : h->m.ev)
        for (auto& p : v) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    delete h;
    return 0;
}

// split-bf16 packs of the large-batch recurrences, derived on the device from the fp32 packs of the image
// an upload happened: every derived pack is stale from here on
static int run_split3(Model& m, hipStream_t) {
    ++m.weights_version;
    return 0;
}

int mtadgat_load_weights(mtadgat_handle h, const mtadgat_params* p, void* stream) {
    if (!h || !p) return fail(MTADGAT_ERR_INVALID, "null argument");
    Model& m = h->m;
    std::vector<float> host;
    std::string err = pack_weights(m, *p, host);
    if (!err.empty()) return fail(MTADGAT_ERR_INVALID, err);
    // the packed weights live on the device that is current now (the caller's); a handle that moves to
    // another GPU gets a fresh allocation there
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (m.packed_dev && m.packed_device != dev) {
        int cur = dev;
        (void)hipSetDevice(m.packed_device);
        (void)hipFree(m.packed_dev);
        (void)hipSetDevice(cur);
        m.packed_dev = nullptr;
        m.have_weights = false;
    }
    if (!m.packed_dev) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m.packed_dev), m.packed_floats * sizeof(float)));
        m.packed_device = dev;
    }
    hipStream_t s = (hipStream_t)stream;
    // Stream-ordered upload without a device synchronisation: the packed image goes through a pinned staging
    // buffer owned by the handle (an async copy from pageable memory is not reliably ordered with the kernels
    // that follow on the stream); the buffer is only rewritten after the previous upload's event has fired.
    if (m.upload_ev) HIP_TRY(hipEventSynchronize(m.upload_ev));
    else HIP_TRY(hipEventCreateWithFlags(&m.upload_ev, hipEventDisableTiming));
    if (m.staging_pinned && m.staging_floats < m.packed_floats) {
        (void)hipHostFree(m.staging_pinned);
        m.staging_pinned = nullptr;
    }
    if (!m.staging_pinned) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m.staging_pinned), m.packed_floats * sizeof(float), hipHostMallocDefault));
        m.staging_floats = m.packed_floats;
    }
    std::memcpy(m.staging_pinned, host.data(), m.packed_floats * sizeof(float));
    HIP_TRY(hipMemcpyAsync(m.packed_dev, m.staging_pinned, m.packed_floats * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(m.upload_ev, s));
    { int rc = run_split3(m, s); if (rc) return rc; }
    m.have_weights = true;
    return 0;
}

/* Re-packs the weight image from a flat device buffer of the parameters (mtadgat_params field order = the order of
 * the flat gradient buffer, see mtadgat_grad_offsets) without a round trip through the host: the training loop's
 * optimizer.step() -> forward.  Needs one previous mtadgat_load_weights on this device (it lays down the padding
 * and the index maps) and the fp32 image (precision 0 or 2).  No host involvement: the column order of the folded GATv2
 * projection (the sign pattern of the two attention vectors `a`) is derived by a kernel as well. */
int mtadgat_update_weights_device(mtadgat_handle h, const float* flat_dev, int64_t n_floats, void* stream) {
    if (!h || !flat_dev) return fail(MTADGAT_ERR_INVALID, "null argument");
    Model& m = h->m;
    if (!m.have_weights || !m.packed_dev) return fail(MTADGAT_ERR_NOWEIGHTS, "update_weights_device needs a previous load_weights");
    if (m.precision == 1) return fail(MTADGAT_ERR_UNSUPPORTED, "device-side re-packing covers the fp32 image only");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != m.packed_device) return fail(MTADGAT_ERR_INVALID, "the weights live on another device");
    DevTables& t = m.dt;
    hipStream_t s = (hipStream_t)stream;
    auto upload = [&](int*& dst, const std::vector<int>& v) -> int {
        if (v.empty()) return 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dst), v.size() * sizeof(int)));
        HIP_TRY(hipMemcpy(dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
        return 0;
    };
    if (!t.ready || t.device != dev) {
        free_device_tables(m);
        std::string err = build_device_tables(m);
        if (!err.empty()) return fail(MTADGAT_ERR_UNSUPPORTED, err);
        int rc;
        if ((rc = upload(t.gidx_dev, t.gidx)) || (rc = upload(t.gatcode_dev[0], t.gatcode[0])) || (rc = upload(t.gatcode_dev[1], t.gatcode[1])) ||
            (rc = upload(t.foldcode_dev, t.foldcode))) return rc;
        for (int which = 0; which < 2; ++which) {
            const GatPlan& g = which == 0 ? m.feat : m.temp;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.colk_dev[which]), (size_t)(g.ldl + 16) * sizeof(int)));
            t.colk[which].clear();
        }
        t.device = dev;
        t.ready = true;
    }
    if (n_floats != t.fo.total) return fail(MTADGAT_ERR_INVALID, "flat parameter buffer: wrong number of floats");
    // column order of the GATv2 projections: from the signs of `a`, on the device (round 5: rounds 3-4 read `a` back through pinned
    // memory and synchronised the stream here, once per optimizer step).  The kernels take [P8, PT] from the image from now on;
    // the host's copies keep the values of the last host-side load (they only steer heuristics).
    if (m.cfg.use_gatv2)
        for (int which = 0; which < 2; ++which) {
            const GatPlan& g = which == 0 ? m.feat : m.temp;
            K_TRY(launch_gat_colorder(flat_dev + t.fo.a[which], g.E, (double)m.cfg.alpha, t.colk_dev[which], g.ldl + 16,
                                      reinterpret_cast<int*>(m.packed_dev + g.ord_off), s), "projection column order");
        }
    K_TRY(launch_pack_gather(flat_dev, t.gidx_dev, m.packed_dev, (long)m.packed_floats, s), "weight gather");
    for (int which = 0; which < 2; ++which) {
        const GatPlan& g = which == 0 ? m.feat : m.temp;
        PackGatArgs a{};
        a.flat = flat_dev; a.lin_w = t.fo.lin_w[which]; a.lin_b = t.fo.lin_b[which]; a.a = t.fo.a[which];
        a.E = g.E; a.D = g.D; a.KS = g.ldl; a.ord = reinterpret_cast<const int*>(m.packed_dev + g.ord_off);
        a.v2 = m.cfg.use_gatv2 ? 1 : 0; a.fused = g.fused ? 1 : 0;
        a.alpha = m.cfg.alpha; a.colk = t.colk_dev[which]; a.code = t.gatcode_dev[which];
        a.n_code = g.NT * g.Q * 256; a.n_bias = g.NT * 32;
        a.w_out = m.packed_dev + g.w_off; a.b_out = m.packed_dev + g.b_off;
        K_TRY(launch_pack_gat(a, s), "graph-attention projection pack");
    }
    for (size_t l = 0; l < m.gru.size(); ++l) {
        const GruPlan& g = m.gru[l];
        K_TRY(launch_pack_gru_bias(flat_dev, t.fo.gru_bih[l], t.fo.gru_bhh[l], g.H, g.Hp, m.packed_dev + g.b_off,
                                   g.has_xproj ? m.packed_dev + g.xproj.b_off : nullptr, s), "gru bias pack");
    }
    for (size_t l = 0; l < m.rec.size(); ++l) {
        const GruPlan& g = m.rec[l];
        K_TRY(launch_pack_gru_bias(flat_dev, t.fo.rec_bih[l], t.fo.rec_bhh[l], g.H, g.Hp, m.packed_dev + g.b_off,
                                   g.has_xproj ? m.packed_dev + g.xproj.b_off : nullptr, s), "decoder bias pack");
    }
    if (m.rec[0].xmode == 1) {
        const GruPlan& g = m.rec[0];
        PackFoldArgs a{};
        a.flat = flat_dev; a.wih = t.fo.rec_wih[0]; a.Hin = g.in_dim; a.T = m.W; a.H = g.H; a.Hp = g.Hp; a.NMp = 8 * g.Qx;
        a.code = t.foldcode_dev; a.tile_floats = (long)g.NCG * g.Qxp * 3 * 256;
        a.tiles_out = m.packed_dev + g.wx_off; a.fold_out = g.has16 ? m.packed_dev + g.fold_off : nullptr;
        if (!t.foldsum_dev) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.foldsum_dev), (size_t)3 * g.H * (g.in_dim + 1) * sizeof(double)));
        a.prefix = t.foldsum_dev;
        K_TRY(launch_pack_fold(a, s), "decoder input fold");
    }
    { int rc = run_split3(m, s); if (rc) return rc; }      // the split packs follow the fp32 packs they are derived from
    m.bf16_packed = false;
    return 0;
}

int mtadgat_params_fingerprint(const void* const* tensors_dev, const int64_t* n_elements, int n_tensors, uint64_t* out_dev, void* stream) {
    if (!tensors_dev || !n_elements || !out_dev || n_tensors < 0) return fail(MTADGAT_ERR_INVALID, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(uint64_t), s));
    long base = 0;
    for (int t0 = 0; t0 < n_tensors; t0 += FINGERPRINT_MAX_TENSORS) {
        FingerprintArgs a{};
        const int nt = std::min(FINGERPRINT_MAX_TENSORS, n_tensors - t0);
        for (int t = 0; t < nt; ++t) {
            a.ptr[t] = tensors_dev[t0 + t];
            a.count[t] = (long)n_elements[t0 + t];
            a.base[t] = base;
            base += a.count[t];
        }
        K_TRY(launch_fingerprint(a, nt, reinterpret_cast<unsigned long long*>(out_dev), s), "parameter fingerprint");
    }
    return 0;
}

/* (offset, length) pairs, in floats, of the regions of the packed image that are derived on the device from other
 * regions (the split-bf16 packs); returns the number of pairs */
int mtadgat_derived_regions(mtadgat_handle h, int64_t* out, int max_pairs) {
    if (!h || !out) return 0;
    const Model& m = h->m;
    int n = 0;
    auto add = [&](size_t off, size_t len) {
        if (n < max_pairs) { out[2 * n] = (int64_t)off; out[2 * n + 1] = (int64_t)len; }
        ++n;
    };
    auto one = [&](const GruPlan& g) {
        add(g.wx3_off, (size_t)(g.xmode == 1 ? m.W : 1) * g.NCG * g.Qxp16 * 9 * 256);
        add(g.wh3_off, (size_t)g.NCG * (2 * g.NCG + 2) * 6 * 256 + 3 * 256);
        add(g.scale_off, 4);
        if (g.wx2_off) add(g.wx2_off, (size_t)g.NCG * g.Qxp16 * 6 * 256 + 3 * 256);
        if (g.wxq_off) add(g.wxq_off, (size_t)g.NCG * g.Qxp16 * 6 * 256 + 3 * 256);
    };
    for (const GruPlan& g : m.gru) one(g);
    for (const GruPlan& g : m.rec) one(g);
    add(m.feat.w3_off, (size_t)m.feat.NT * m.feat.Q16 * 3 * 256);
    add(m.temp.w3_off, (size_t)m.temp.NT * m.temp.Q16 * 3 * 256);
    if (m.bw.supported) {
        for (const GruBwdPlan& gb : m.bw.gru) add(gb.wihT.w3_off, (size_t)gb.wihT.NT * gb.wihT.Q16 * 3 * 256);
        for (const GruBwdPlan& gb : m.bw.rec) add(gb.wihT.w3_off, (size_t)gb.wihT.NT * gb.wihT.Q16 * 3 * 256);
        for (size_t l = 0; l < m.bw.gru.size(); ++l)
            if (m.bw.gru[l].whT3_off) add(m.bw.gru[l].whT3_off, (size_t)m.gru[l].NCG * 6 * m.gru[l].NCG * 3 * 256);
        for (size_t l = 0; l < m.bw.rec.size(); ++l)
            if (m.bw.rec[l].whT3_off) add(m.bw.rec[l].whT3_off, (size_t)m.rec[l].NCG * 6 * m.rec[l].NCG * 3 * 256);
        for (const LinTPlan& p : m.bw.fcT) add(p.w3_off, (size_t)p.NT * p.Q16 * 3 * 256);
        add(m.bw.recfcT.w3_off, (size_t)m.bw.recfcT.NT * m.bw.recfcT.Q16 * 3 * 256);
        for (int k = 0; k < 2; ++k) add(m.bw.gat[k].lrT.w3_off, (size_t)m.bw.gat[k].lrT.NT * m.bw.gat[k].lrT.Q16 * 3 * 256);
        for (int k = 0; k < 2; ++k)
            if (m.bw.gat[k].wu3_off) add(m.bw.gat[k].wu3_off, (size_t)2 * m.bw.gat[k].NTu * (((k == 0 ? m.feat : m.temp).Q + 1) / 2) * 3 * 256);
    }
    for (const GruPlan& g : m.gru)
        if (g.has_xproj && g.xproj.w3_off) add(g.xproj.w3_off, (size_t)g.xproj.NT * g.xproj.Q16 * 3 * 256);
    if (m.rec_fc.w3_off) add(m.rec_fc.w3_off, (size_t)m.rec_fc.NT * m.rec_fc.Q16 * 3 * 256);
    add(m.conv_w3_off, (size_t)m.convNT * (m.taps * m.Fp16 / 16) * 3 * 256);
    for (const GatPlan* g : {&m.feat, &m.temp}) add(g->uw3_off, (size_t)g->NT * g->uQ16 * 3 * 256);
    add(m.conv_w2h_off, (size_t)m.convNT * (m.taps * m.Fp16 / 16) * 2 * 256);
    add(m.conv_scale_off, 4);
    for (const GatPlan* g : {&m.feat, &m.temp}) {
        add(g->w2h_off, (size_t)g->NT * g->Q16 * 2 * 256);
        add(g->gscale_off, 4);
    }
    return n;
}

/* Host-only self check of the device-side re-pack's gather table (no GPU needed): packs `p` with the host packer, builds
 * the table and counts the image positions whose table entry does not reproduce the packed value from the flat
 * parameter buffer.  *n_gathered receives the number of positions the table covers.  Returns the mismatch count (0 =
 * consistent), or a negative error code. */
int64_t mtadgat_selfcheck_gather_table(mtadgat_handle h, const mtadgat_params* p, int64_t* n_gathered) {
    if (!h || !p) return fail(MTADGAT_ERR_INVALID, "null argument");
    Model& m = h->m;
    const int keep_prec = m.precision;
    m.precision = 0;
    std::vector<float> img;
    std::string err = pack_weights(m, *p, img);
    if (err.empty()) err = build_device_tables(m);
    m.precision = keep_prec;
    if (!err.empty()) return fail(MTADGAT_ERR_UNSUPPORTED, err);
    const DevTables& t = m.dt;
    // the flat parameter buffer in the field order of mtadgat_params
    std::vector<float> flat((size_t)t.fo.total);
    auto put = [&](int64_t off, const float* src, int64_t n) { std::memcpy(flat.data() + off, src, (size_t)n * sizeof(float)); };
    const mtadgat_config& c = m.cfg;
    put(t.fo.conv_w, p->conv_weight, (int64_t)m.F * m.F * m.taps); put(t.fo.conv_b, p->conv_bias, m.F);
    for (int which = 0; which < 2; ++which) {
        const GatPlan& g = which == 0 ? m.feat : m.temp;
        const int lin_in = c.use_gatv2 ? 2 * g.D : g.D;
        put(t.fo.lin_w[which], which == 0 ? p->feat_lin_weight : p->temp_lin_weight, (int64_t)g.E * lin_in);
        put(t.fo.lin_b[which], which == 0 ? p->feat_lin_bias : p->temp_lin_bias, g.E);
        put(t.fo.a[which], which == 0 ? p->feat_a : p->temp_a, c.use_gatv2 ? g.E : 2 * g.E);
        put(t.fo.bias[which], which == 0 ? p->feat_bias : p->temp_bias, (int64_t)g.K * g.K);
    }
    for (size_t l = 0; l < m.gru.size(); ++l) {
        const int in = m.gru[l].in_dim, H = m.gru[l].H;
        put(t.fo.gru_wih[l], p->gru_w_ih[l], (int64_t)3 * H * in); put(t.fo.gru_whh[l], p->gru_w_hh[l], (int64_t)3 * H * H);
        put(t.fo.gru_bih[l], p->gru_b_ih[l], 3 * H); put(t.fo.gru_bhh[l], p->gru_b_hh[l], 3 * H);
    }
    for (size_t i = 0; i < m.fc.size(); ++i) {
        put(t.fo.fc_w[i], p->fc_weight[i], (int64_t)m.fc[i].out_dim * m.fc[i].in_dim); put(t.fo.fc_b[i], p->fc_bias[i], m.fc[i].out_dim);
    }
    for (size_t l = 0; l < m.rec.size(); ++l) {
        const int in = m.rec[l].in_dim, H = m.rec[l].H;
        put(t.fo.rec_wih[l], p->rec_w_ih[l], (int64_t)3 * H * in); put(t.fo.rec_whh[l], p->rec_w_hh[l], (int64_t)3 * H * H);
        put(t.fo.rec_bih[l], p->rec_b_ih[l], 3 * H); put(t.fo.rec_bhh[l], p->rec_b_hh[l], 3 * H);
    }
    put(t.fo.rec_fc_w, p->rec_fc_weight, (int64_t)c.out_dim * c.recon_hid_dim); put(t.fo.rec_fc_b, p->rec_fc_bias, c.out_dim);
    int64_t bad = 0, cov = 0;
    for (size_t i = 0; i < m.packed_floats; ++i) {
        const int g = t.gidx[i];
        if (g < 0) continue;
        ++cov;
        if (std::memcmp(&img[i], &flat[(size_t)g], sizeof(float)) != 0) ++bad;
    }
    if (n_gathered) *n_gathered = cov;
    return bad;
}

int64_t mtadgat_packed_floats(mtadgat_handle h) { return h ? (int64_t)h->m.packed_floats : 0; }
int mtadgat_read_packed(mtadgat_handle h, float* dst_host, int64_t n_floats, void* stream) {
    if (!h || !dst_host) return fail(MTADGAT_ERR_INVALID, "null argument");
    Model& m = h->m;
    if (!m.have_weights || !m.packed_dev) return fail(MTADGAT_ERR_NOWEIGHTS, "no weights loaded");
    if (n_floats != (int64_t)m.packed_floats) return fail(MTADGAT_ERR_INVALID, "wrong size");
    if (int rc_ = ensure_all_split(m, (hipStream_t)stream)) return rc_;      // the image as the kernels would see it
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(dst_host, m.packed_dev, m.packed_floats * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int mtadgat_set_precision(mtadgat_handle h, int mode) {
    if (!h || mode < 0 || mode > 2)
        return fail(MTADGAT_ERR_INVALID, "precision mode must be 0 (fp32 MFMA), 1 (bf16 operands) or 2 (fp32 through split-bf16 operands)");
    h->m.precision = mode;
    return 0;
}

/* 1 when the bf16 weight streams are present in the packed image (they are packed by mtadgat_load_weights only
 * while precision 1 is selected: call set_precision before load_weights, or load again after switching) */
int mtadgat_bf16_ready(mtadgat_handle h) { return (h && h->m.have_weights && h->m.bf16_packed) ? 1 : 0; }

/* Testing / measurement hook: "gru_kernel" = 0 automatic choice of the large-batch recurrence kernel, 1 tile-major (k_gru),
 * 2 chunk-major (k_gru_cm) wherever it applies, 3 hidden-tile split on split operands (k_gru_split X3H) wherever it applies */
int mtadgat_set_option(mtadgat_handle h, const char* name, int value) {
    if (!h || !name) return fail(MTADGAT_ERR_INVALID, "null argument");
    if (std::strcmp(name, "gru_kernel") == 0 && value >= 0 && value <= 3) { h->m.gru_kernel = value; return 0; }
    if (std::strcmp(name, "gat_kernel") == 0 && (value == 0 || value == 1 || value == 3)) { h->m.gat_kernel = value; return 0; }
    if (std::strcmp(name, "wgrad_kernel") == 0 && value >= 0 && value <= 2) { h->m.wgrad_kernel = value; return 0; }
    if (std::strcmp(name, "conv_kernel") == 0 && value >= 0 && value <= 2) { h->m.conv_kernel = value; return 0; }
    if (std::strcmp(name, "rowgemm_kernel") == 0 && value >= 0 && value <= 2) { h->m.rowgemm_kernel = value; return 0; }
    if (std::strcmp(name, "gemm_lds") == 0 && value >= 0 && value <= 1) { set_gemm_lds_off(value); return 0; }      // process-wide; 1 = off
    if (std::strcmp(name, "conv_shared") == 0 && value >= 0 && value <= 1) { h->m.conv_shared = value; return 0; }
    if (std::strcmp(name, "conv_fused") == 0 && value >= 0 && value <= 1) { h->m.conv_fused = value; return 0; }
    if (std::strcmp(name, "lanes") == 0 && value >= 0 && value <= 1) { h->m.lanes = value; return 0; }
    if (std::strcmp(name, "gath_dbg") == 0 && value >= 0 && value <= 63) { h->m.gath_dbg = value; return 0; }
    return fail(MTADGAT_ERR_INVALID, "unknown option or value");
}

// How forward() walks a call of `batch` windows: chunks of at most m.chunk windows one after the other (each planned for its own
// size: a short last chunk gets the small-batch kernels and their buffers), and -- fused front end, split-operand arithmetic --
// a chunk in PIECES that alternate between two lanes (the caller's stream and the handle's own), each lane with its own
// workspace.  Why: the large-batch recurrence (k_gru_cm) is a 100-step latency chain on one 4-wave workgroup per 128 windows
// and per CU, so its time is the same for any chunk up to 32 768 windows and doubles at 32 769; and while it runs it owns the
// register file but not the vector ALU.  Two lanes let one piece's convolution / attention fill the other's recurrence:
// measured (MSL shape, profiles/r04_overlap_experiment.txt) 10 240 windows 6.0 -> 5.4 ms, 12 288: 6.3 -> 5.7, 36 864:
// 15.9 -> 12.4, 40 960: 16.7 -> 14.1, 49 152: 18.0 -> 16.4; no gain at 16 384 .. 32 768 (one piece) and at whole multiples of
// 32 768.  The pieces' results are those of forward() called on each piece (the kernels are chosen by the piece's size).
struct Piece { int64_t c0, n; int lane; };
static void forward_schedule(const Model& m, int64_t batch, std::vector<Piece>& out) {
    out.clear();
    const bool two = m.lanes == 0 && m.precision == 2 && m.temp.fused && m.feat.fused;
    // Round 6: models on the un-fused (wide) attention path walk a call of several chunks with the chunks ALTERNATING between the
    // lanes.  Their chunks are a few hundred windows (projections through HBM: ~7 MB of scratch per window at F = 512, W = 256), so the
    // recurrences of a chunk are small-batch latency chains on a quarter of the CUs (896 windows: ~56 workgroups; GRU layer + decoder
    // 68.7 of config 4's 273 ms in round 5, strictly behind the chunk's front end) -- on the other lane the next chunk's convolution,
    // projections and attention fill the rest of the machine meanwhile.  Each lane has its own workspace (lane_floats).
    const bool alt = m.lanes == 0 && !(m.temp.fused && m.feat.fused) && batch > m.chunk && !getenv("MTADGAT_PIECES");
    if (alt) {
        // the first piece is HALF a chunk: with equal pieces the lanes run in lock step (both front ends side by side, then both
        // recurrences on 2 x 56 CUs: 258 vs 268 ms for one lane); offset by half a chunk, one lane's recurrences sit under the other's
        // front end
        int64_t c0 = 0, ci = 0;
        const int64_t half = std::max<int64_t>(32, m.chunk / 2 / 32 * 32);
        while (c0 < batch) {
            const int64_t n = std::min<int64_t>(ci == 0 ? half : m.chunk, batch - c0);
            out.push_back({c0, n, (int)(ci & 1)});
            c0 += n; ++ci;
        }
        return;
    }
    for (int64_t c0 = 0; c0 < batch; c0 += m.chunk) {
        const int64_t n = std::min<int64_t>(m.chunk, batch - c0);
        int64_t k = 1, base = n;
        if (two && n > SPLIT3_MAX_WINDOWS && n <= 2 * SPLIT3_MAX_WINDOWS) {                // two halves for the hidden-tile-split kernel
            k = 2;
            base = n / 2 / 32 * 32;                                                        // (whole 32-window groups, the last piece takes the rest)
            // a short tail rides beside a full piece instead (9 216 windows: 5.15 -> 4.77 ms; from 10 240 on the halves are as fast)
            if (n - SPLIT3_MAX_WINDOWS <= 1536) base = SPLIT3_MAX_WINDOWS;
        } else if (two && n > 32768 && n % 32768 != 0) {                                   // full k_gru_cm rounds (256 workgroups of 128 windows), then the rest
            // (whole multiples of 32 768 gain nothing from the second lane -- every round is full -- and stay one piece on the caller's stream)
            k = (n + 32767) / 32768;
            base = 32768;
        }
        if (const char* e_ = getenv("MTADGAT_PIECES")) {                                   // measurement hook: "a,b,c": piece sizes, alternating lanes
            int64_t at = 0;
            int lane = 0;
            for (const char* q = e_; *q && at < n;) {
                int64_t len = std::min<int64_t>(n - at, strtoll(q, const_cast<char**>(&q), 10));
                if (*q == ',') ++q;
                if (len <= 0) break;
                out.push_back({c0 + at, len, lane});
                lane ^= 1; at += len;
            }
            if (at < n) out.push_back({c0 + at, n - at, lane});
            continue;
        }
        for (int64_t i = 0, at = 0; i < k; ++i) {
            const int64_t len = i + 1 < k ? base : n - at;
            out.push_back({c0 + at, len, (int)(i & 1)});
            at += len;
        }
    }
}
// workspace: per lane the largest plan of the pieces it runs; lane 1 starts behind lane 0
static void lane_floats(const Model& m, int64_t batch, size_t (&need)[2]) {
    std::vector<Piece> sched;
    forward_schedule(m, batch, sched);
    need[0] = need[1] = 0;
    int64_t seen[2][2] = {{0, 0}, {0, 0}};                    // (plans repeat: sizes seen last per lane)
    for (const Piece& pc : sched) {
        if (pc.n == seen[pc.lane][0] || pc.n == seen[pc.lane][1]) continue;
        seen[pc.lane][1] = seen[pc.lane][0]; seen[pc.lane][0] = pc.n;
        Workspace o;
        plan_workspace(m, pc.n, o);
        need[pc.lane] = std::max(need[pc.lane], o.total);
    }
}
/* Largest value the convolution wrote during the last forward() on this workspace (of its last chunk), read back after
 * the stream has drained: below 2^15 the large-batch kernels formed their products from two fp16 pieces per operand, above
 * from three bf16 pieces (the device-side range guard).  0 when the forward did not record it (un-fused attention path). */
int mtadgat_last_conv_max(mtadgat_handle h, const void* ws, int64_t batch, float* out_host, void* stream) {
    if (!h || !ws || !out_host || batch <= 0) return fail(MTADGAT_ERR_INVALID, "null argument");
    Workspace o;
    std::vector<Piece> sched;                                 // the layout of the call's last piece (each piece has its own plan, each lane its own workspace)
    forward_schedule(h->m, batch, sched);
    size_t need[2];
    lane_floats(h->m, batch, need);
    plan_workspace(h->m, sched.back().n, o);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(out_host, static_cast<const float*>(ws) + (sched.back().lane ? need[0] : 0) + o.vmax, sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int64_t mtadgat_chunk_windows(mtadgat_handle h) { return h ? h->m.chunk : 0; }
int mtadgat_set_chunk_windows(mtadgat_handle h, int64_t w) {
    if (!h || w < 1) return fail(MTADGAT_ERR_INVALID, "chunk must be >= 1");
    h->m.chunk = w;
    return 0;
}

static size_t workspace_floats(const Model& m, int64_t batch) {
    size_t need[2];
    lane_floats(m, batch, need);
    Workspace o;                                              // (the stage entry points plan one chunk on one lane)
    plan_workspace(m, std::min<int64_t>(batch, m.chunk), o);
    return std::max(need[0] + need[1], o.total);
}
size_t mtadgat_workspace_bytes(mtadgat_handle h, int64_t batch) {
    if (!h || batch <= 0) return 0;
    return workspace_floats(h->m, batch) * sizeof(float);
}

static int forward_impl(mtadgat_handle h, const XSource& src, int64_t batch, float* preds, float* recons, float* recons_last,
                        float* hend_out, void* ws_, size_t ws_bytes, void* stream) {
    int rc = check_common(h, batch, ws_, ws_bytes, true);
    if (rc) return rc;
    if (batch == 0) return 0;
    if (!src.x) return fail(MTADGAT_ERR_INVALID, "input is NULL");
    Model& m = h->m;
    hipStream_t s0 = (hipStream_t)stream;
    const int F = m.F, W = m.W;
    if ((rc = ensure_all_split(m, s0))) return rc;        // (no-ops while the weights are unchanged; before any lane forks off)
    std::vector<Piece> sched;
    forward_schedule(m, batch, sched);
    size_t lane_need[2];
    lane_floats(m, batch, lane_need);
    bool second = false;
    for (const Piece& pc : sched) second = second || pc.lane == 1;
    // Small calls (the reference Predictor's 256 windows, prediction.py:31): no launch fills the machine, the forward is a chain of
    // latencies -- so the stages that do not depend on each other run side by side: the feature layer next to the temporal one
    // (both read the convolution, mtad_gat.py:68-69), the forecasting head next to the reconstruction decoder (both read h_end,
    // mtad_gat.py:76-77).  Same kernels, same results; the second stream joins before the call returns.
    // (large calls gain nothing from it: with the convolution as its own launch and both attention layers side by side, 65 536
    // windows take 21.0-21.2 ms against 21.1-21.3 one after the other -- the layers compete for the same vector ALUs)
    const bool fork = !second && sched.size() == 1 && m.lanes == 0 && sched[0].n <= FORK_MAX_WINDOWS && use_fused(m.temp) && use_fused(m.feat);
    if (second || fork) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (m.lane_stream && m.lane_device != dev) free_lane(m);        // the handle moved to another GPU since the lane was made
        if (!m.lane_stream) {
            m.lane_device = dev;
            HIP_TRY(hipStreamCreateWithFlags(&m.lane_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&m.lane_begin, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&m.lane_end, hipEventDisableTiming));
            for (hipEvent_t& e : m.fork_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    if (second) {
        HIP_TRY(hipEventRecord(m.lane_begin, s0));            // the second lane starts after whatever the caller queued before this call
        HIP_TRY(hipStreamWaitEvent(m.lane_stream, m.lane_begin, 0));
    }
    struct Joiner {                                           // the caller's stream waits for the second lane on every way out
        Model& m; hipStream_t s0; bool on;
        ~Joiner() {
            if (on && hipEventRecord(m.lane_end, m.lane_stream) == hipSuccess) (void)hipStreamWaitEvent(s0, m.lane_end, 0);
        }
    } joiner{m, s0, second || fork};
    for (const Piece& pc : sched) {
        const int64_t c0 = pc.c0, n = pc.n;
        hipStream_t s = pc.lane ? m.lane_stream : s0;
        float* ws = static_cast<float*>(ws_) + (pc.lane ? lane_need[0] : 0);
        Workspace o;
        plan_workspace(m, n, o);
        float* xc = ws + o.xc;
        float* xct = ws + o.xct;
        float* hcat = ws + o.hcat;
        if (use_fused(m.temp) && use_fused(m.feat)) {
            // fused front: conv writes only h_cat[:, :F]; each layer's workgroup stages its window from
            // there (the feature layer transposes on the way into LDS) -- no xc / xc^T / L' / R' in HBM
            unsigned* vmax = reinterpret_cast<unsigned*>(ws + o.vmax);
            GatConvIn cv{};
            bool conv_in_gat = false;
            if (conv_shared_applies(m, src, n)) {
                if ((rc = run_conv_shared(m, src, c0, n, hcat, ws + o.cf, ws + o.el, ws + o.er, s, vmax))) return rc;
            } else if (fused_conv_args(m, src, c0, n, hcat, vmax, reinterpret_cast<unsigned char*>(ws + o.winflag), cv)) {
                // the temporal layer's workgroups compute the convolution of their windows themselves: no convolution launch,
                // h_cat[:, :F] is written once and not read back by that layer
                conv_in_gat = true;
                HIP_TRY(hipMemsetAsync(vmax, 0, sizeof(unsigned), s));
            } else if ((rc = run_conv(m, src, c0, n, nullptr, nullptr, hcat, nullptr, s, vmax))) return rc;
            if (fork && !conv_in_gat) {                       // the feature layer on the second lane, beside the temporal one
                HIP_TRY(hipEventRecord(m.fork_ev[0], s));
                HIP_TRY(hipStreamWaitEvent(m.lane_stream, m.fork_ev[0], 0));
                if ((rc = run_gat_fused(m, m.feat, hcat, m.Dp, 1, n, hcat + F, (long)W * m.Dp, 1, m.Dp, m.lane_stream, nullptr, nullptr, 0, vmax))) return rc;
                HIP_TRY(hipEventRecord(m.fork_ev[1], m.lane_stream));
            }
            if ((rc = run_gat_fused(m, m.temp, hcat, m.Dp, 0, n, hcat + 2 * F, (long)W * m.Dp, m.Dp, 1, s, nullptr, nullptr, 0, vmax,
                                           conv_in_gat ? &cv : nullptr))) return rc;
            if (fork && !conv_in_gat) HIP_TRY(hipStreamWaitEvent(s, m.fork_ev[1], 0));
            else if ((rc = run_gat_fused(m, m.feat, hcat, m.Dp, 1, n, hcat + F, (long)W * m.Dp, 1, m.Dp, s, nullptr, nullptr, 0, vmax))) return rc;
        } else {
            if ((rc = run_conv(m, src, c0, n, xc, xct, hcat, nullptr, s))) return rc;
            // temporal layer: nodes = time steps, rows of xc
            if ((rc = run_gat_layer(m, m.temp, xc, m.Fp, n, ws + o.lct, ws + o.rtt, hcat + 2 * F, (long)W * m.Dp, m.Dp, 1, s))) return rc;
            // feature layer: nodes = features, rows of xc^T
            if ((rc = run_gat_layer(m, m.feat, xct, m.Wp, n, ws + o.lcf, ws + o.rtf, hcat + F, (long)W * m.Dp, 1, m.Dp, s))) return rc;
        }
        float* hend = ws + o.hend;
        const long ldh = m.gru.back().Hp;
        if ((rc = run_gru_stack(m, hcat, m.Dp, n, hend, ldh, ws, o, s,
                                (use_fused(m.temp) && use_fused(m.feat)) ? reinterpret_cast<const unsigned*>(ws + o.vmax) : nullptr))) return rc;
        if (hend_out)
            K_TRY(launch_copy2d(hend, ldh, hend_out + c0 * m.cfg.gru_hid_dim, m.cfg.gru_hid_dim, n, m.cfg.gru_hid_dim, s),
                  "h_end copy");
        if (fork && preds && (recons || recons_last)) {
            // the forecasting head on the second lane, beside the reconstruction decoder (their scratch regions are disjoint)
            HIP_TRY(hipEventRecord(m.fork_ev[2], s));
            HIP_TRY(hipStreamWaitEvent(m.lane_stream, m.fork_ev[2], 0));
            if ((rc = run_heads(m, hend, ldh, n, preds + c0 * m.cfg.out_dim, nullptr, nullptr, ws, o, m.lane_stream))) return rc;
            if ((rc = run_heads(m, hend, ldh, n, nullptr, recons ? recons + c0 * (int64_t)W * m.cfg.out_dim : nullptr,
                                recons_last ? recons_last + c0 * m.cfg.out_dim : nullptr, ws, o, s)))
                return rc;
        } else if (preds || recons || recons_last) {
            if ((rc = run_heads(m, hend, ldh, n, preds ? preds + c0 * m.cfg.out_dim : nullptr,
                                recons ? recons + c0 * (int64_t)W * m.cfg.out_dim : nullptr,
                                recons_last ? recons_last + c0 * m.cfg.out_dim : nullptr, ws, o, s)))
                return rc;
        }
    }
    return 0;
}

int mtadgat_forward(mtadgat_handle h, const float* x, int64_t batch, float* preds, float* recons, float* hend_out,
                    void* ws_, size_t ws_bytes, void* stream) {
    XSource src;
    src.x = x;
    return forward_impl(h, src, batch, preds, recons, nullptr, hend_out, ws_, ws_bytes, stream);
}

int mtadgat_forward_xbf16(mtadgat_handle h, const void* x_bf16, int64_t batch, float* preds, float* recons, float* hend_out,
                          void* ws_, size_t ws_bytes, void* stream) {
    if (h && (size_t)(32 + h->m.taps - 1) * (std::max(h->m.Fp, h->m.Fp16) + 4) * sizeof(float) > 20 * 1024)
        return fail(MTADGAT_ERR_UNSUPPORTED, "bfloat16 input is read by the LDS-staged convolution only (n_features too large): pass float32");
    XSource src;
    src.x = static_cast<const float*>(x_bf16);
    src.x_bf16 = 1;
    return forward_impl(h, src, batch, preds, recons, nullptr, hend_out, ws_, ws_bytes, stream);
}

int mtadgat_forward_series(mtadgat_handle h, const float* series, int64_t n_rows, const int64_t* starts, int64_t start0,
                           int64_t stride, int64_t batch, float* preds, float* recons, float* recons_last, void* ws_,
                           size_t ws_bytes, void* stream) {
    if (!h) return fail(MTADGAT_ERR_INVALID, "null handle");
    if (n_rows < h->m.W) return fail(MTADGAT_ERR_INVALID, "series shorter than one window");
    if (!starts) {   // starts_dev cannot be checked on the host; the arithmetic progression can
        if (stride < 0 || start0 < 0 || (batch > 0 && start0 + (batch - 1) * stride + h->m.W > n_rows))
            return fail(MTADGAT_ERR_INVALID, "windows start0 + w*stride .. +W do not lie inside the series");
    }
    XSource src;
    src.x = series; src.gather = 1; src.starts = starts; src.start0 = start0; src.stride = stride;
    return forward_impl(h, src, batch, preds, recons, recons_last, nullptr, ws_, ws_bytes, stream);
}

int mtadgat_conv(mtadgat_handle h, const float* x, int64_t batch, float* y, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(h, batch, ws, ws_bytes, false);
    if (rc) return rc;
    if (batch == 0) return 0;
    if (!x || !y) return fail(MTADGAT_ERR_INVALID, "null tensor");
    XSource src;
    src.x = x;
    return run_conv(h->m, src, 0, batch, nullptr, nullptr, nullptr, y, (hipStream_t)stream);
}

int mtadgat_gat(mtadgat_handle h, int which, const float* xc_in, int64_t batch, float* out, void* ws_, size_t ws_bytes,
                void* stream) {
    int rc = check_common(h, batch, ws_, ws_bytes, true);
    if (rc) return rc;
    if (batch == 0) return 0;
    if (!xc_in || !out) return fail(MTADGAT_ERR_INVALID, "null tensor");
    if (which != 0 && which != 1) return fail(MTADGAT_ERR_INVALID, "which must be 0 (feature) or 1 (temporal)");
    Model& m = h->m;
    hipStream_t s = (hipStream_t)stream;
    float* ws = static_cast<float*>(ws_);
    const int F = m.F, W = m.W;
    for (int64_t c0 = 0; c0 < batch; c0 += m.chunk) {
        const int64_t n = std::min<int64_t>(m.chunk, batch - c0);
        Workspace o;
        plan_workspace(m, std::min<int64_t>(batch, m.chunk), o);
        const float* xin = xc_in + c0 * (int64_t)W * F;
        float* o_c = out + c0 * (int64_t)W * F;
        if (which == 1) {
            K_TRY(launch_copy2d(xin, F, ws + o.xc, m.Fp, n * W, F, s), "pad copy");
            if ((rc = run_gat_layer(m, m.temp, ws + o.xc, m.Fp, n, ws + o.lct, ws + o.rtt, o_c, (long)W * F, F, 1, s))) return rc;
        } else {
            K_TRY(launch_transpose_win(xin, F, ws + o.xct, m.Wp, n, W, F, s), "transpose");
            if ((rc = run_gat_layer(m, m.feat, ws + o.xct, m.Wp, n, ws + o.lcf, ws + o.rtf, o_c, (long)W * F, 1, F, s))) return rc;
        }
    }
    return 0;
}

int mtadgat_gru(mtadgat_handle h, const float* hcat, int64_t batch, float* hend, void* ws_, size_t ws_bytes, void* stream) {
    int rc = check_common(h, batch, ws_, ws_bytes, true);
    if (rc) return rc;
    if (batch == 0) return 0;
    if (!hcat || !hend) return fail(MTADGAT_ERR_INVALID, "null tensor");
    Model& m = h->m;
    float* ws = static_cast<float*>(ws_);
    const int D = 3 * m.F, H = m.cfg.gru_hid_dim;
    for (int64_t c0 = 0; c0 < batch; c0 += m.chunk) {
        const int64_t n = std::min<int64_t>(m.chunk, batch - c0);
        Workspace o;
        plan_workspace(m, std::min<int64_t>(batch, m.chunk), o);
        // the kernel wants 16-byte aligned, zero padded rows: stage the caller's (b, W, 3F) tensor into the padded buffer
        hipStream_t s = (hipStream_t)stream;
        HIP_TRY(hipMemsetAsync(ws + o.hcat, 0, (size_t)n * m.W * m.Dp * sizeof(float), s));
        K_TRY(launch_copy2d(hcat + c0 * (int64_t)m.W * D, D, ws + o.hcat, m.Dp, n * m.W, D, s), "h_cat pad copy");
        if ((rc = run_gru_stack(m, ws + o.hcat, m.Dp, n, hend + c0 * H, H, ws, o, s))) return rc;
    }
    return 0;
}

int mtadgat_heads(mtadgat_handle h, const float* hend, int64_t batch, float* preds, float* recons, void* ws_, size_t ws_bytes,
                  void* stream) {
    int rc = check_common(h, batch, ws_, ws_bytes, true);
    if (rc) return rc;
    if (batch == 0) return 0;
    if (!hend) return fail(MTADGAT_ERR_INVALID, "null tensor");
    Model& m = h->m;
    float* ws = static_cast<float*>(ws_);
    const int H = m.cfg.gru_hid_dim;
    for (int64_t c0 = 0; c0 < batch; c0 += m.chunk) {
        const int64_t n = std::min<int64_t>(m.chunk, batch - c0);
        Workspace o;
        plan_workspace(m, std::min<int64_t>(batch, m.chunk), o);
        hipStream_t s = (hipStream_t)stream;
        const long ldh = m.gru.back().Hp;
        HIP_TRY(hipMemsetAsync(ws + o.hend, 0, (size_t)n * ldh * sizeof(float), s));
        K_TRY(launch_copy2d(hend + c0 * H, H, ws + o.hend, ldh, n, H, s), "h_end pad copy");
        if ((rc = run_heads(m, ws + o.hend, ldh, n, preds ? preds + c0 * m.cfg.out_dim : nullptr,
                            recons ? recons + c0 * (int64_t)m.W * m.cfg.out_dim : nullptr, nullptr, ws, o, s)))
            return rc;
    }
    return 0;
}

int mtadgat_profile_enable(mtadgat_handle h, int on) {
    if (!h) return fail(MTADGAT_ERR_INVALID, "null handle");
    h->m.profile = on != 0;
    return 0;
}

int mtadgat_profile_read(mtadgat_handle h, double ms[MTADGAT_PROFILE_SLOTS], int64_t launches[MTADGAT_PROFILE_SLOTS]) {
    if (!h || !ms || !launches) return fail(MTADGAT_ERR_INVALID, "null argument");
    for (int i = 0; i < MTADGAT_PROFILE_SLOTS; ++i) {
        ms[i] = 0.0;
        launches[i] = 0;
        for (auto& p : h->m.ev[i]) {
            HIP_TRY(hipEventSynchronize(p.second));
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, p.first, p.second));
            ms[i] += t;
            launches[i] += 1;
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        h->m.ev[i].clear();
    }
    return 0;
}

const char* mtadgat_profile_name(int slot) {
    return (slot >= 0 && slot < MTADGAT_PROFILE_SLOTS) ? kSlotNames[slot] : "";
}

}  // extern "C"

// =====================================================================================================
// training step: forward that keeps the activations ("tape") + backward  (reference training.py:106-127)
// =====================================================================================================
namespace {

DropArgs make_drop(float p, uint64_t seed, int64_t win0) {
    DropArgs d{};
    if (p > 0.f) {
        double t = (double)p * 4294967296.0;
        d.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
        if (d.thresh == 0) d.thresh = 1;
        d.keep_scale = 1.0f / (1.0f - p);
    } else {
        d.thresh = 0;
        d.keep_scale = 1.0f;
    }
    d.seed_lo = (unsigned)(seed & 0xffffffffu);
    d.seed_hi = (unsigned)(seed >> 32);
    d.win0 = win0;
    return d;
}

int check_train(mtadgat_handle h, int64_t batch, float p) {
    if (!h) return fail(MTADGAT_ERR_INVALID, "null handle");
    if (batch < 0) return fail(MTADGAT_ERR_INVALID, "negative batch");
    if (!h->m.have_weights) return fail(MTADGAT_ERR_NOWEIGHTS, "mtadgat_load_weights has not been called");
    if (!h->m.bw.supported) return fail(MTADGAT_ERR_UNSUPPORTED, "no HIP backward for this configuration: " + h->m.bw.why);
    if (!(p >= 0.f && p < 1.f)) return fail(MTADGAT_ERR_INVALID, "dropout probability must be in [0, 1)");
    if (h->m.precision == 1)
        return fail(MTADGAT_ERR_UNSUPPORTED, "the training step computes in fp32 (mtadgat_set_precision 0 or 2): the bf16-operand recurrences of "
                                             "rounds 2-5 were slower than it at every batch size and are gone");
    return 0;
}

// d X = d Y W through k_rowgemm with the transposed pack
int run_rowgemm_T(Model& m, const LinTPlan& p, const float* X, long ldx, long R, float* Y, long ldy, int nvalid, bool accumulate,
                  const float* gate, long ldg, float gate_scale, hipStream_t s) {
    RowGemmArgs a{};
    a.X = X; a.ldx = ldx; a.Kvalid = p.in_dim; a.Q = p.Q;
    a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + p.w_off);
    a.bias = m.packed_dev + m.bw.zero_off;
    a.Y = Y; a.ldy = ldy; a.Nvalid = nvalid;
    a.vec_store = ((ldy & 3) == 0 && aligned16(Y)) ? 1 : 0;
    a.R = R; a.NT = p.NT; a.NT_rm = p.NT; a.group = 1; a.relu = 0;
    a.accumulate = accumulate ? 1 : 0;
    a.gate = gate; a.ldg = ldg; a.gate_scale = gate_scale;
    // split-bf16 operands from 64 Ki rows on (batches of >= 656 windows at W = 100: below that the product is a few tens of
    // microseconds either way and the pack would have to be re-split after every optimizer step for nothing)
    if (p.w3_off && ((m.precision == 2 && m.rowgemm_kernel != 1 && R >= 65536) || m.rowgemm_kernel == 2)) {
        if (p.w3_version != m.weights_version) {
            K_TRY(launch_split3(m.packed_dev + p.w_off, m.packed_dev + p.w3_off, p.NT, p.Q, p.Q16, 1, nullptr, s), "split-bf16 transposed weights");
            p.w3_version = m.weights_version;
        }
        a.x3 = 1; a.Q16 = p.Q16;
        a.Wp3 = reinterpret_cast<const f32x4*>(m.packed_dev + p.w3_off);
    }
    K_TRY(launch_rowgemm(a, s), "data-gradient rowgemm");
    return 0;
}

// the un-scaled projections [L | R] = V [W_l ; W_r]^T + [b | 0] of the node rows (GATv2 score backward): one row GEMM over the backward
// plan's pack, on three bf16 pieces per operand from 65 536 rows (as the data-gradient GEMMs; the pack is split on first use after an upload)
int run_bwd_projection(Model& m, const GatPlan& gp, const GatBwdPlan& gb, const float* Vn, long ldv, long rows, float* LR, hipStream_t s) {
    RowGemmArgs r{};
    r.X = Vn; r.ldx = ldv; r.Kvalid = gp.D; r.Q = gp.Q;
    r.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + gb.wu_off);
    r.bias = m.packed_dev + gb.bu_off;
    r.Y = LR; r.ldy = 2L * gb.Ep; r.Nvalid = 2 * gb.Ep; r.vec_store = 1;
    r.R = rows; r.NT = 2 * gb.NTu; r.NT_rm = 2 * gb.NTu; r.group = 1; r.relu = 0;
    if (gb.wu3_off && ((m.precision == 2 && m.rowgemm_kernel != 1 && rows >= 65536) || m.rowgemm_kernel == 2)) {
        const int Q16 = (gp.Q + 1) / 2;
        if (gb.wu3_version != m.weights_version) {
            K_TRY(launch_split3(m.packed_dev + gb.wu_off, m.packed_dev + gb.wu3_off, 2 * gb.NTu, gp.Q, Q16, 1, nullptr, s), "split-bf16 projection weights (backward)");
            gb.wu3_version = m.weights_version;
        }
        r.x3 = 1; r.Q16 = Q16;
        r.Wp3 = reinterpret_cast<const f32x4*>(m.packed_dev + gb.wu3_off);
    }
    K_TRY(launch_rowgemm(r, s), "attention backward (projection)");
    return 0;
}

struct WgradIn {
    const float* A = nullptr; long lda = 0;
    const float* B = nullptr; long ldb = 0; int bmode = 0; int bshift = 0;
    long R = 0; int T = 1;
};
// The weight-gradient GEMMs of a backward leave their slab partials in consecutive regions of `base`; the reductions into the flat
// gradient buffer are queued and run as ONE launch (flush) -- at the end of the backward, or earlier when the region is full.
struct WgradQueue {
    float* base = nullptr;
    size_t cap = 0, used = 0;
    WgradReduceArgs pend[WGRAD_BATCH_MAX];
    int n = 0;
    hipStream_t s = nullptr;
    int flush() {
        if (n == 0) return 0;
        K_TRY(launch_wgrad_reduce_batch(pend, n, s), "weight-gradient reductions");
        n = 0; used = 0;
        return 0;
    }
};
int run_wgrad(Model& m, const WgradPlan& p, const WgradIn& in, WgradQueue& wq, float* outW, float* outB, hipStream_t s) {
    const int nslab_ = wgrad_slabs(in.R, p.Mp, p.Np);
    const size_t need_ = ((size_t)nslab_ * p.Mp * p.Np + 63) & ~(size_t)63;
    if (need_ > wq.cap) return fail(MTADGAT_ERR_WORKSPACE, "internal: weight-gradient partial region too small");
    if (wq.used + need_ > wq.cap || wq.n == WGRAD_BATCH_MAX)
        if (int rc_ = wq.flush()) return rc_;
    float* wpart = wq.base + wq.used;
    wq.used += need_;
    WgradArgs a{};
    a.A = in.A; a.lda = in.lda; a.M = p.M;
    a.B = in.B; a.ldb = in.ldb; a.N = p.N; a.bmode = in.bmode; a.bshift = in.bshift;
    a.T = in.T; a.F = m.F; a.taps = m.taps; a.pad = m.pad;
    a.R = in.R;
    a.nslab = wgrad_slabs(in.R, p.Mp, p.Np);
    long rps = (in.R + a.nslab - 1) / a.nslab;
    rps = (rps + 15) / 16 * 16;
    a.rows_per_slab = rps;
    a.P = wpart; a.Mp = p.Mp; a.Np = p.Np;
    // default fp32 arithmetic: the products from three bf16 pieces per operand (the fp32 MFMA is 2.7x the matrix time)
    a.x3 = (m.precision == 2 && m.wgrad_kernel != 1) || m.wgrad_kernel == 2;
    K_TRY(launch_wgrad(a, s), "weight-gradient GEMM");
    WgradReduceArgs r{};
    r.P = wpart; r.nslab = a.nslab; r.Mp = p.Mp; r.Np = p.Np; r.M = p.M; r.N = p.N;
    r.rowmapW = reinterpret_cast<const int*>(m.packed_dev + p.rowW_off);
    r.colmap = reinterpret_cast<const int*>(m.packed_dev + p.col_off);
    r.rowmapB = p.has_bias && outB ? reinterpret_cast<const int*>(m.packed_dev + p.rowB_off) : nullptr;
    r.outW = outW; r.outB = outB;
    wq.pend[wq.n++] = r;
    return 0;
}
// column sums over the windows (d bias / d a of a GATv2 layer): the first stage now, the reduction in the step's batched launch
int run_sum_rows_queued(const float* src, long ld, long R, int N, float* scratch, float* dst, WgradQueue& wq, hipStream_t s) {
    if (wq.n == WGRAD_BATCH_MAX)
        if (int rc_ = wq.flush()) return rc_;
    int nslab = 0;
    K_TRY(launch_sum_rows_part(src, ld, R, N, scratch, &nslab, s), "column sums");
    if (nslab == 0) return 0;
    WgradReduceArgs r{};
    r.P = scratch; r.nslab = nslab; r.Mp = 1; r.Np = N; r.M = 1; r.N = N;
    r.outW = dst;
    wq.pend[wq.n++] = r;
    return 0;
}

}  // namespace

extern "C" {

int mtadgat_backward_supported(mtadgat_handle h) {
    if (!h) return 0;
    if (!h->m.bw.supported) g_err = "no HIP backward for this configuration: " + h->m.bw.why;
    return h->m.bw.supported ? 1 : 0;
}

size_t mtadgat_tape_bytes(mtadgat_handle h, int64_t batch) {
    if (!h || batch <= 0 || !h->m.bw.supported) return 0;
    Tape t;
    plan_tape(h->m, batch, t);
    return t.total * sizeof(float);
}

size_t mtadgat_backward_workspace_bytes(mtadgat_handle h, int64_t batch) {
    if (!h || batch <= 0 || !h->m.bw.supported) return 0;
    BwdWorkspace w;
    plan_bwd_workspace(h->m, batch, w);
    return w.total * sizeof(float);
}

int64_t mtadgat_grad_floats(mtadgat_handle h) { return h ? h->m.bw.gl.total : 0; }

int mtadgat_grad_offsets(mtadgat_handle h, int64_t* out, int max_n) {
    if (!h || !out) return fail(MTADGAT_ERR_INVALID, "null argument");
    const GradLayout& g = h->m.bw.gl;
    std::vector<int64_t> v = {g.conv_w, g.conv_b, g.lin_w[0], g.lin_b[0], g.a[0], g.bias[0], g.lin_w[1], g.lin_b[1], g.a[1], g.bias[1]};
    for (size_t l = 0; l < g.gru_wih.size(); ++l) v.insert(v.end(), {g.gru_wih[l], g.gru_whh[l], g.gru_bih[l], g.gru_bhh[l]});
    for (size_t i = 0; i < g.fc_w.size(); ++i) { v.push_back(g.fc_w[i]); v.push_back(g.fc_b[i]); }
    for (size_t l = 0; l < g.rec_wih.size(); ++l) v.insert(v.end(), {g.rec_wih[l], g.rec_whh[l], g.rec_bih[l], g.rec_bhh[l]});
    v.insert(v.end(), {g.rec_fc_w, g.rec_fc_b});
    if ((int)v.size() > max_n) return fail(MTADGAT_ERR_INVALID, "offset array too small");
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

int mtadgat_train_layout(mtadgat_handle h, int64_t batch, int64_t* out, int max_n) {
    if (!h || !out || batch <= 0 || !h->m.bw.supported) return fail(MTADGAT_ERR_INVALID, "bad argument");
    Tape t;
    plan_tape(h->m, batch, t);
    BwdWorkspace w;
    plan_bwd_workspace(h->m, batch, w);
    std::vector<int64_t> v = {(int64_t)t.hcat, (int64_t)t.xct, (int64_t)t.att_f, (int64_t)t.att_t, (int64_t)t.hend, (int64_t)t.gates_g,
                              (int64_t)t.seq_g, (int64_t)t.gates_d, (int64_t)t.seq_d, (int64_t)t.xdec,
                              (int64_t)w.da, (int64_t)w.dhcat, (int64_t)w.dhdec, (int64_t)w.dhend, (int64_t)w.dz0, (int64_t)w.dz1,
                              (int64_t)w.de_f, (int64_t)w.de_t, (int64_t)w.dv_f, (int64_t)w.dv_t, (int64_t)w.dlr_f, (int64_t)w.dlr_t,
                              (int64_t)w.dap_f, (int64_t)w.dap_t, (int64_t)w.dpre};
    if ((int)v.size() > max_n) return fail(MTADGAT_ERR_INVALID, "layout array too small");
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

int mtadgat_forward_train(mtadgat_handle h, const float* x, int64_t batch, int64_t window0, float dropout_p, uint64_t seed,
                          float* preds, float* recons, void* tape_, size_t tape_bytes, void* stream) {
    int rc = check_train(h, batch, dropout_p);
    if (rc) return rc;
    if (batch == 0) return 0;
    if (!x || !preds || !recons || !tape_) return fail(MTADGAT_ERR_INVALID, "null tensor");
    Model& m = h->m;
    Tape t;
    plan_tape(m, batch, t);
    if (!aligned16(tape_) || tape_bytes < t.total * sizeof(float)) return fail(MTADGAT_ERR_WORKSPACE, "tape too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    float* T = static_cast<float*>(tape_);
    const int F = m.F, W = m.W;
    const int64_t n = batch;
    const DropArgs drop = make_drop(dropout_p, seed, window0);
    float* hcat = T + t.hcat;
    XSource src;
    src.x = x;
    unsigned* vmax = reinterpret_cast<unsigned*>(T + t.vmax);
    if ((rc = run_conv(m, src, 0, n, nullptr, T + t.xct, hcat, nullptr, s, vmax))) return rc;
    // the two attention layers: fused per-window kernels, or -- wide layers -- projection through memory + k_gat_wide; either way
    // the softmax rows are kept and the dropout of modules.py:90 / :189 is applied inside the kernel
    if (use_fused(m.temp)) {
        if ((rc = run_gat_fused(m, m.temp, hcat, m.Dp, 0, n, hcat + 2 * F, (long)W * m.Dp, m.Dp, 1, s, T + t.att_t, &drop, DROP_TEMP, vmax))) return rc;
    } else {
        if ((rc = run_proj(m, m.temp, hcat, m.Dp, n * W, T + t.lct, T + t.rtt, s))) return rc;
        if ((rc = run_attend(m, m.temp, T + t.lct, T + t.rtt, hcat, m.Dp, n, hcat + 2 * F, (long)W * m.Dp, m.Dp, 1, s, T + t.att_t, &drop, DROP_TEMP))) return rc;
    }
    if (use_fused(m.feat)) {
        if ((rc = run_gat_fused(m, m.feat, hcat, m.Dp, 1, n, hcat + F, (long)W * m.Dp, 1, m.Dp, s, T + t.att_f, &drop, DROP_FEAT, vmax))) return rc;
    } else {
        const float* xct = T + t.xct;
        if ((rc = run_proj(m, m.feat, xct, m.Wp, n * F, T + t.lcf, T + t.rtf, s))) return rc;
        if ((rc = run_attend(m, m.feat, T + t.lcf, T + t.rtf, xct, m.Wp, n, hcat + F, (long)W * m.Dp, 1, m.Dp, s, T + t.att_f, &drop, DROP_FEAT))) return rc;
    }
    // GRU stack (modules.py:235-238): every layer keeps its gates and states; between stacked layers nn.GRU's dropout
    // (training only; reference modules.py:232-233): the dropped sequence is what the next layer reads and is kept as well
    const int Lg = (int)m.gru.size(), Ld = (int)m.rec.size();
    const GruPlan& g = m.gru.back();                 // the layer that produces h_end
    float* hend = T + t.hend;
    {
        const float* xin = hcat;
        long ldin = m.Dp;
        int kx = 3 * F;
        for (int l = 0; l < Lg; ++l) {
            const GruPlan& gl_ = m.gru[l];
            const bool last = l == Lg - 1;
            float* seq = T + (l == 0 ? t.seq_g : t.seq_gu[l - 1]);
            float* gates = T + (l == 0 ? t.gates_g : t.gates_gu[l - 1]);
            if ((rc = run_gru_layer(m, S_GRU, gl_, xin, ldin, kx, n, last ? hend : nullptr, gl_.Hp, seq, nullptr, nullptr, nullptr, s, gates,
                                    l == 0 ? T + t.xp : nullptr, l == 0 && use_g16(m, m.gru, n, true), l == 0 ? vmax : nullptr))) return rc;
            if (!last) {
                float* dr = T + t.drop_g[l];
                K_TRY(launch_seq_dropout(seq, dr, n, W, gl_.H, gl_.Hp, drop, DROP_GRU0 + (unsigned)l, s), "gru inter-layer dropout");
                xin = dr; ldin = gl_.Hp; kx = gl_.H;
            }
        }
    }
    // forecasting head: ReLU + dropout on the hidden layers (modules.py:307-311), activations kept
    {
        Scope sc(m, S_FC, s);
        const float* xin = hend;
        long ld = g.Hp;
        const int nfc = (int)m.fc.size();
        for (int i = 0; i < nfc; ++i) {
            const LinPlan& p = m.fc[i];
            const bool last = (i == nfc - 1);
            RowGemmArgs a{};
            a.X = xin; a.ldx = ld; a.Kvalid = p.in_dim; a.Q = p.Q;
            a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + p.w_off);
            a.bias = m.packed_dev + p.b_off;
            a.R = n; a.NT = p.NT; a.NT_rm = p.NT; a.group = 1;
            if (last) {
                a.Y = preds; a.ldy = p.out_dim; a.Nvalid = p.out_dim;
                a.vec_store = (p.out_dim % 4 == 0 && aligned16(preds)) ? 1 : 0;
                a.relu = 0;
            } else {
                a.Y = T + t.fc_act[i]; a.ldy = p.NT * 32; a.Nvalid = p.NT * 32; a.vec_store = 1;
                a.relu = 1;
                a.drop_thresh = drop.thresh; a.seed_lo = drop.seed_lo; a.seed_hi = drop.seed_hi; a.keep_scale = drop.keep_scale;
                a.drop_stream = DROP_FC0 + (unsigned)i; a.row0 = window0;
            }
            K_TRY(launch_rowgemm(a, s), "forecasting head (training)");
            xin = a.Y; ld = a.ldy;
        }
    }
    // reconstruction decoder, all steps kept
    const GruPlan& r = m.rec[0];
    if (Ld == 1 && use_g16(m, m.rec, n, true)) {
        if ((rc = run_gru_layer(m, S_RECON, r, hend, g.Hp, m.cfg.gru_hid_dim, n, nullptr, 0, T + t.seq_d, nullptr, nullptr, nullptr, s, T + t.gates_d,
                                T + t.xp, true))) return rc;
        Scope sc(m, S_RECON, s);
        const LinPlan& p = m.rec_fc;
        RowGemmArgs a{};
        a.X = T + t.seq_d; a.ldx = r.Hp; a.Kvalid = r.H; a.Q = p.Q;
        a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + p.w_off);
        a.bias = m.packed_dev + p.b_off;
        a.Y = recons; a.ldy = p.out_dim; a.Nvalid = p.out_dim;
        a.vec_store = (p.out_dim % 4 == 0 && aligned16(recons)) ? 1 : 0;
        a.R = n * (int64_t)W; a.NT = p.NT; a.NT_rm = p.NT; a.group = 1; a.relu = 0;
        K_TRY(launch_rowgemm(a, s), "reconstruction Linear (training)");
    } else {
        // layer 0 reads the repeated h_end (modules.py:279), the layers above the (dropped-out) states of the one below; the
        // per-step Linear (modules.py:282) rides in the last layer while its weights fit beside the state in 64 KB of LDS
        // (launch_gru_split), otherwise it is a row GEMM over the kept states -- as in the small-batch branch above
        const float* xin = hend;
        long ldin = g.Hp;
        int kx = m.cfg.gru_hid_dim;
        const GruPlan& rlast = m.rec[Ld - 1];
        const bool fc_rides = ((size_t)rlast.NCG * 1024 + (size_t)rlast.NCG * m.rec_fc.out_dim * 32) * sizeof(float) <= 64 * 1024;
        for (int l = 0; l < Ld; ++l) {
            const GruPlan& rl = m.rec[l];
            const bool last = l == Ld - 1, fused = last && fc_rides;
            float* seq = T + (l == 0 ? t.seq_d : t.seq_du[l - 1]);
            float* gates = T + (l == 0 ? t.gates_d : t.gates_du[l - 1]);
            if ((rc = run_gru_layer(m, S_RECON, rl, xin, ldin, kx, n, nullptr, 0, seq, fused ? &m.rec_fc : nullptr, fused ? recons : nullptr, nullptr, s,
                                    gates))) return rc;
            if (!last) {
                float* dr = T + t.drop_d[l];
                K_TRY(launch_seq_dropout(seq, dr, n, W, rl.H, rl.Hp, drop, DROP_REC0 + (unsigned)l, s), "decoder inter-layer dropout");
                xin = dr; ldin = rl.Hp; kx = rl.H;
            } else if (!fused) {
                Scope sc(m, S_RECON, s);
                const LinPlan& p = m.rec_fc;
                RowGemmArgs a{};
                a.X = seq; a.ldx = rl.Hp; a.Kvalid = rl.H; a.Q = p.Q;
                a.Wp = reinterpret_cast<const f32x4*>(m.packed_dev + p.w_off);
                a.bias = m.packed_dev + p.b_off;
                a.Y = recons; a.ldy = p.out_dim; a.Nvalid = p.out_dim;
                a.vec_store = (p.out_dim % 4 == 0 && aligned16(recons)) ? 1 : 0;
                a.R = n * (int64_t)W; a.NT = p.NT; a.NT_rm = p.NT; a.group = 1; a.relu = 0;
                K_TRY(launch_rowgemm(a, s), "reconstruction Linear (training)");
            }
        }
    }
    K_TRY(launch_xdec(hend, g.Hp, m.cfg.gru_hid_dim, W, n, T + t.xdec, g.Hp, s), "decoder input");
    return 0;
}

int mtadgat_backward(mtadgat_handle h, const float* x, int64_t batch, int64_t window0, float dropout_p, uint64_t seed,
                     const float* d_preds, const float* d_recons, const void* tape_, size_t tape_bytes, float* grads,
                     void* ws_, size_t ws_bytes, void* stream) {
    int rc = check_train(h, batch, dropout_p);
    if (rc) return rc;
    if (batch == 0) return 0;
    if (!x || !d_preds || !d_recons || !tape_ || !grads || !ws_) return fail(MTADGAT_ERR_INVALID, "null tensor");
    Model& m = h->m;
    const BwdPlan& b = m.bw;
    const GradLayout& gl = b.gl;
    Tape t;
    plan_tape(m, batch, t);
    BwdWorkspace w;
    plan_bwd_workspace(m, batch, w);
    if (tape_bytes < t.total * sizeof(float)) return fail(MTADGAT_ERR_WORKSPACE, "tape too small");
    if (!aligned16(ws_) || ws_bytes < w.total * sizeof(float)) return fail(MTADGAT_ERR_WORKSPACE, "backward workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    const float* T = static_cast<const float*>(tape_);
    float* ws = static_cast<float*>(ws_);
    const int F = m.F, W = m.W, od = m.cfg.out_dim;
    const int64_t n = batch;
    const long RW = (long)n * W;
    const DropArgs drop = make_drop(dropout_p, seed, window0);
    const int Lg = (int)m.gru.size(), Ld = (int)m.rec.size();
    const GruPlan& g = m.gru.back();                 // the layer that produced h_end (all GRU layers share the hidden size)
    const GruPlan& r = m.rec.back();                 // the decoder layer under the per-step Linear
    const float* hcat = T + t.hcat;
    const float* hend = T + t.hend;
    WgradQueue wpart;
    wpart.base = ws + w.wpart; wpart.cap = w.wpart_floats; wpart.s = s;
    float* dhend = ws + w.dhend;

    // ---- 1. forecasting head (modules.py:307-311)
    {
        const int nfc = (int)m.fc.size();
        const float* dy = d_preds;
        long lddy = od;
        for (int i = nfc - 1; i >= 0; --i) {
            const LinPlan& p = m.fc[i];
            const float* act = i > 0 ? T + t.fc_act[i - 1] : hend;
            const long lda = i > 0 ? (long)m.fc[i - 1].NT * 32 : g.Hp;
            WgradIn in;
            in.A = dy; in.lda = lddy; in.B = act; in.ldb = lda; in.R = n; in.T = 1;
            if ((rc = run_wgrad(m, b.fc_wg[i], in, wpart, grads + gl.fc_w[i], grads + gl.fc_b[i], s))) return rc;
            float* y = i > 0 ? ws + ((i & 1) ? w.dz1 : w.dz0) : dhend;
            const long ldy = i > 0 ? (long)b.fcT[i].NT * 32 : g.Hp;
            if ((rc = run_rowgemm_T(m, b.fcT[i], dy, lddy, n, y, ldy, (int)ldy, false, i > 0 ? act : nullptr, lda, drop.keep_scale, s))) return rc;
            dy = y; lddy = ldy;
        }
    }
    // One recurrence layer backward: BPTT (gate gradients of every step into `da`) and the layer's weight gradients.
    // dhseq: gradient of every state (nullptr: none); dhend_: of the last state only (nullptr: none); xin: the rows the layer read.
    float* da = ws + w.da;
    auto layer_bwd = [&](const GruPlan& q, const GruBwdPlan& qb, bool small, const float* gates, const float* seq, const float* dhseq,
                         const float* dhend_, const float* xin, long ldxin, int64_t o_wih, int64_t o_whh, int64_t o_bih, int64_t o_bhh,
                         const char* what) -> int {
        if (small) {
            Gru16BwdArgs ga{};
            ga.Gates = gates; ga.Seq = seq; ga.DHseq = dhseq; ga.lddh = q.Hp; ga.DHend = dhend_; ga.ldde = q.Hp;
            ga.W16T = m.packed_dev + q.g16T_off; ga.DA = da; ga.Hp = q.Hp; ga.KS = q.KS16; ga.NT16 = q.NT16; ga.T = W; ga.B = n; ga.H = q.H;
            if (use_g1(n)) {
                ga.W16T = m.packed_dev + q.g1T_off;
                K_TRY(launch_gru1_bwd(ga, s), what);
            } else
                K_TRY(launch_gru16_bwd(ga, s), what);
        } else {
            GruBwdArgs ga{};
            ga.Gates = gates; ga.Seq = seq; ga.DHseq = dhseq; ga.lddh = q.Hp; ga.DHend = dhend_; ga.ldde = q.Hp;
            ga.WhT = reinterpret_cast<const f32x4*>(m.packed_dev + qb.whT_off);
            if (m.precision == 2 && qb.whT3_off) {       // default arithmetic: three bf16 pieces per operand (split on first use after an upload)
                if (qb.whT3_version != m.weights_version) {
                    K_TRY(launch_split3(m.packed_dev + qb.whT_off, m.packed_dev + qb.whT3_off, q.NCG, 12 * q.NCG, 6 * q.NCG, 1, nullptr, s),
                          "split-bf16 transposed recurrent weights");
                    qb.whT3_version = m.weights_version;
                }
                ga.WhT = reinterpret_cast<const f32x4*>(m.packed_dev + qb.whT3_off);
                ga.x3 = 1;
            }
            ga.DA = da; ga.Hp = q.Hp; ga.H = q.H; ga.T = W; ga.NCG = q.NCG; ga.B = n;
            K_TRY(launch_gru_bwd(ga, s), what);
        }
        int rc2;
        WgradIn hh;
        hh.A = da + q.Hp; hh.lda = 4L * q.Hp; hh.bshift = 1; hh.B = seq; hh.ldb = q.Hp; hh.R = RW; hh.T = W;
        if ((rc2 = run_wgrad(m, qb.wg_hh, hh, wpart, grads + o_whh, grads + o_bhh, s))) return rc2;
        WgradIn ih;
        ih.A = da; ih.lda = 4L * q.Hp; ih.B = xin; ih.ldb = ldxin; ih.R = RW; ih.T = W;
        return run_wgrad(m, qb.wg_ih, ih, wpart, grads + o_wih, grads + o_bih, s);
    };
    // ---- 2. reconstruction model (modules.py:276-283): per-step Linear, decoder layers from the top, decoder input
    float* dhdec = ws + w.dhdec;
    {
        const float* seq_top = T + (Ld == 1 ? t.seq_d : t.seq_du[Ld - 2]);
        if ((rc = run_rowgemm_T(m, b.recfcT, d_recons, od, RW, dhdec, r.Hp, r.Hp, false, nullptr, 0, 1.f, s))) return rc;
        WgradIn in;
        in.A = d_recons; in.lda = od; in.B = seq_top; in.ldb = r.Hp; in.R = RW; in.T = W;
        if ((rc = run_wgrad(m, b.recfc_wg, in, wpart, grads + gl.rec_fc_w, grads + gl.rec_fc_b, s))) return rc;
        for (int l = Ld - 1; l >= 0; --l) {
            const GruPlan& q = m.rec[l];
            const float* gates = T + (l == 0 ? t.gates_d : t.gates_du[l - 1]);
            const float* seq = T + (l == 0 ? t.seq_d : t.seq_du[l - 1]);
            const float* xin = l == 0 ? T + t.xdec : T + t.drop_d[l - 1];
            const long ldxin = l == 0 ? m.gru.back().Hp : m.rec[l - 1].Hp;
            if ((rc = layer_bwd(q, b.rec[l], Ld == 1 && use_g16(m, m.rec, n, true), gates, seq, dhdec, nullptr, xin, ldxin, gl.rec_wih[l],
                                gl.rec_whh[l], gl.rec_bih[l], gl.rec_bhh[l], "decoder backward"))) return rc;
            if (l > 0) {
                // d (dropped states of the layer below), then through the dropout: the gradient of that layer's states
                const GruPlan& lo = m.rec[l - 1];
                if ((rc = run_rowgemm_T(m, b.rec[l].wihT, da, 4L * q.Hp, RW, dhdec, lo.Hp, lo.Hp, false, nullptr, 0, 1.f, s))) return rc;
                K_TRY(launch_seq_dropout(dhdec, dhdec, n, W, lo.H, lo.Hp, drop, DROP_REC0 + (unsigned)(l - 1), s), "decoder inter-layer dropout (adjoint)");
            } else {
                // d (decoder input) -> d h_end through the repeat_interleave / view of modules.py:279
                if ((rc = run_rowgemm_T(m, b.rec[0].wihT, da, 4L * q.Hp, RW, dhdec, g.Hp, g.Hp, false, nullptr, 0, 1.f, s))) return rc;
                K_TRY(launch_xdec_bwd(dhdec, g.Hp, m.cfg.gru_hid_dim, W, n, dhend, g.Hp, s), "decoder input adjoint");
            }
        }
    }
    // ---- 3. GRU layers from the top (modules.py:235-238): the last one receives d h_end, the others the gradient of their states
    float* dhcat = ws + w.dhcat;
    for (int l = Lg - 1; l >= 0; --l) {
        const GruPlan& q = m.gru[l];
        const float* gates = T + (l == 0 ? t.gates_g : t.gates_gu[l - 1]);
        const float* seq = T + (l == 0 ? t.seq_g : t.seq_gu[l - 1]);
        const float* xin = l == 0 ? hcat : T + t.drop_g[l - 1];
        const long ldxin = l == 0 ? m.Dp : m.gru[l - 1].Hp;
        const bool top = l == Lg - 1;
        if ((rc = layer_bwd(q, b.gru[l], Lg == 1 && use_g16(m, m.gru, n, true), gates, seq, top ? nullptr : dhdec, top ? dhend : nullptr, xin, ldxin,
                            gl.gru_wih[l], gl.gru_whh[l], gl.gru_bih[l], gl.gru_bhh[l], "gru backward"))) return rc;
        if (l > 0) {
            const GruPlan& lo = m.gru[l - 1];
            if ((rc = run_rowgemm_T(m, b.gru[l].wihT, da, 4L * q.Hp, RW, dhdec, lo.Hp, lo.Hp, false, nullptr, 0, 1.f, s))) return rc;
            K_TRY(launch_seq_dropout(dhdec, dhdec, n, W, lo.H, lo.Hp, drop, DROP_GRU0 + (unsigned)(l - 1), s), "gru inter-layer dropout (adjoint)");
        } else if ((rc = run_rowgemm_T(m, b.gru[0].wihT, da, 4L * q.Hp, RW, dhcat, m.Dp, m.Dp, false, nullptr, 0, 1.f, s)))
            return rc;
    }
    // ---- 4. the two graph-attention layers (modules.py:65-95, :166-193)
    for (int which = 1; which >= 0; --which) {
        const GatPlan& gp = which == 0 ? m.feat : m.temp;
        const GatBwdPlan& gb = b.gat[which];
        const int K = gp.K;
        float* de = ws + (which == 0 ? w.de_f : w.de_t);
        float* dv = ws + (which == 0 ? w.dv_f : w.dv_t);
        float* dlr = ws + (which == 0 ? w.dlr_f : w.dlr_t);
        float* dap = ws + (which == 0 ? w.dap_f : w.dap_t);
        const int lddv = which == 0 ? m.Wp : m.Fp;
        const int colofs = which == 0 ? F : 2 * F;
        if (gb.wide) {
            // wide layer (mtadgat_bwdw.hip): every matrix through memory, generic in K and D
            const long so_w = (long)W * m.Dp, so_i = which == 0 ? 1 : m.Dp, so_d = which == 0 ? m.Dp : 1;
            const float* att = T + (which == 0 ? t.att_f : t.att_t);
            const unsigned dstream = which == 0 ? DROP_FEAT : DROP_TEMP;
            const float* Vn = which == 0 ? T + t.xct : hcat;          // node rows: xc^T rows (feature layer) / h_cat rows (temporal layer)
            const long ldv = which == 0 ? m.Wp : m.Dp;
            const int D = gp.D, ldS = round_up(D, 4), Ep = gb.Ep;
            float* dS = ws + w.wds;
            float* LR = ws + w.wlr;
            K_TRY(launch_bw_ds(hcat + colofs, dhcat + colofs, so_w, so_i, so_d, n, K, D, dS, ldS, s), "attention backward (d S)");
            // d V (aggregation path) = att'^T d S, att' = dropout(att)
            K_TRY(launch_bgemm(att, (long)K * K, 1, K, dS, (long)K * ldS, ldS, 1, dv, (long)K * lddv, lddv, K, D, K, n, &drop, dstream, K, s),
                  "attention backward (aggregation d V)");
            // d att' = d S V^T, then the softmax backward in place -> d e
            K_TRY(launch_bgemm(dS, (long)K * ldS, ldS, 1, Vn, (long)K * ldv, 1, ldv, de, (long)K * K, K, K, K, D, n, nullptr, 0, 0, s),
                  "attention backward (d att)");
            K_TRY(launch_bw_softmax(att, de, n, K, drop, dstream, s), "attention backward (softmax)");
            if (!m.cfg.use_gatv2) {
                // GAT (v1), round 6: the score backward is linear in the node vectors below d s -- k_gat_bwd_v1's algebra with the
                // node rows read from memory (k_bw_v1); prep and finish are the fused path's
                const int E = gp.E, PV = 2 * D + 2;
                float* u = ws + w.v1s + (size_t)which * 2 * (2 * std::max(m.F, m.W) + 2);
                float* P = u + (2 * std::max(m.F, m.W) + 2);
                const float* Wm = m.packed_dev + gb.w1_off;
                const float* bv = m.packed_dev + gb.b1_off;
                const float* av = m.packed_dev + gb.a_off;
                K_TRY(launch_gat_v1_prep(Wm, bv, av, E, D, u, s), "attention backward (v1 vectors)");
                K_TRY(launch_bw_v1(Vn, ldv, D, K, u, de, m.cfg.alpha, dv, lddv, dlr, n, s), "attention backward (v1 scores, wide)");
                HIP_TRY(hipMemsetAsync(P, 0, (size_t)PV * sizeof(float), s));
                K_TRY(launch_sum_rows(dlr, PV, n, PV, ws + w.sums, P, s), "attention backward (v1 sums)");
                K_TRY(launch_gat_v1_finish(P, Wm, bv, av, E, D, grads + gl.lin_w[which], grads + gl.lin_b[which], grads + gl.a[which], s),
                      "attention parameter gradients (v1)");
                K_TRY(launch_sum_rows(de, (long)K * K, n, K * K, ws + w.sums, grads + gl.bias[which], s), "attention bias gradient");
                continue;
            }
            // un-scaled projections [L | R] of the node rows, then the score backward
            if ((rc = run_bwd_projection(m, gp, gb, Vn, ldv, (long)n * K, LR, s))) return rc;
            K_TRY(launch_bw_pair(LR, 2 * Ep, Ep, m.packed_dev + gb.a_off, de, K, m.cfg.alpha, dlr, dap, n, s), "attention backward (pairs)");
            const long RK = (long)n * K;
            if ((rc = run_rowgemm_T(m, gb.lrT, dlr, 2L * Ep, RK, dv, lddv, gp.D, true, nullptr, 0, 1.f, s))) return rc;
            WgradIn in;
            in.A = dlr; in.lda = 2L * Ep; in.R = RK; in.T = 1; in.B = Vn; in.ldb = ldv;
            if ((rc = run_wgrad(m, gb.wg, in, wpart, grads + gl.lin_w[which], grads + gl.lin_b[which], s))) return rc;
            if ((rc = run_sum_rows_queued(de, (long)K * K, n, K * K, ws + w.sums_q[2 * which], grads + gl.bias[which], wpart, s))) return rc;
            if ((rc = run_sum_rows_queued(dap, Ep, n, gp.E, ws + w.sums_q[2 * which + 1], grads + gl.a[which], wpart, s))) return rc;
            continue;
        }
        GatBwdAttArgs aa{};
        aa.V = hcat; aa.ldv = m.Dp; aa.D = gp.D; aa.K = K; aa.vt = which == 0 ? 1 : 0; aa.vld = gp.f_vld;
        aa.H = hcat + colofs; aa.dH = dhcat + colofs;
        aa.so_w = (long)W * m.Dp; aa.so_i = which == 0 ? 1 : m.Dp; aa.so_d = which == 0 ? m.Dp : 1;
        aa.ATT = T + (which == 0 ? t.att_f : t.att_t);
        aa.DE = de; aa.DV = dv; aa.lddv = lddv; aa.nwin = n;
        aa.drop = drop; aa.drop_stream = which == 0 ? DROP_FEAT : DROP_TEMP;
        K_TRY(launch_gat_bwd_att(aa, gp.f_IBL, gp.f_JPL, gp.f_RJ, gp.f_nw, gb.att_lds, s), "attention backward (scores)");
        if (!m.cfg.use_gatv2) {
            // GAT (v1) scores: linear in the node vectors below d s (mtadgat_bwd.hip)
            const int E = gp.E, D = gp.D, PV = 2 * D + 2;
            float* u = ws + w.v1s + (size_t)which * 2 * (2 * std::max(m.F, m.W) + 2);
            float* P = u + (2 * std::max(m.F, m.W) + 2);
            const float* Wm = m.packed_dev + gb.w1_off;
            const float* bv = m.packed_dev + gb.b1_off;
            const float* av = m.packed_dev + gb.a_off;
            K_TRY(launch_gat_v1_prep(Wm, bv, av, E, D, u, s), "attention backward (v1 vectors)");
            K_TRY(launch_gat_bwd_v1(hcat, m.Dp, D, K, aa.vt, u, de, m.cfg.alpha, dv, lddv, dlr, n, s), "attention backward (v1 scores)");
            HIP_TRY(hipMemsetAsync(P, 0, (size_t)PV * sizeof(float), s));
            K_TRY(launch_sum_rows(dlr, PV, n, PV, ws + w.sums, P, s), "attention backward (v1 sums)");
            K_TRY(launch_gat_v1_finish(P, Wm, bv, av, E, D, grads + gl.lin_w[which], grads + gl.lin_b[which], grads + gl.a[which], s),
                  "attention parameter gradients (v1)");
            K_TRY(launch_sum_rows(de, (long)K * K, n, K * K, ws + w.sums, grads + gl.bias[which], s), "attention bias gradient");
            continue;
        }
        // round 6: the un-scaled projections [L | R] of the node rows as one row GEMM, then the one-pass score backward (k_bw_pair,
        // mtadgat_bwdw.hip) -- for fused layers too: the per-window kernel that re-projected L, R inside the workgroup is gone
        {
            float* LR = ws + w.wlr;
            if ((rc = run_bwd_projection(m, gp, gb, which == 0 ? T + t.xct : hcat, which == 0 ? m.Wp : m.Dp, (long)n * K, LR, s))) return rc;
            K_TRY(launch_bw_pair(LR, 2 * gb.Ep, gb.Ep, m.packed_dev + gb.a_off, de, K, m.cfg.alpha, dlr, dap, n, s), "attention backward (pairs)");
        }
        const long RK = (long)n * K;
        if ((rc = run_rowgemm_T(m, gb.lrT, dlr, 2L * gb.Ep, RK, dv, lddv, gp.D, true, nullptr, 0, 1.f, s))) return rc;
        WgradIn in;
        in.A = dlr; in.lda = 2L * gb.Ep; in.R = RK; in.T = 1;
        if (which == 0) { in.B = T + t.xct; in.ldb = m.Wp; } else { in.B = hcat; in.ldb = m.Dp; }
        if ((rc = run_wgrad(m, gb.wg, in, wpart, grads + gl.lin_w[which], grads + gl.lin_b[which], s))) return rc;
        if ((rc = run_sum_rows_queued(de, (long)K * K, n, K * K, ws + w.sums_q[2 * which], grads + gl.bias[which], wpart, s))) return rc;
        if ((rc = run_sum_rows_queued(dap, gb.Ep, n, gp.E, ws + w.sums_q[2 * which + 1], grads + gl.a[which], wpart, s))) return rc;
    }
    // ---- 5. convolution (modules.py:18-22)
    {
        float* dpre = ws + w.dpre;
        K_TRY(launch_dxc(hcat, dhcat, m.Dp, ws + w.dv_t, m.Fp, ws + w.dv_f, m.Wp, n, W, F, dpre, m.Fp, s), "conv pre-activation gradient");
        WgradIn in;
        in.A = dpre; in.lda = m.Fp; in.B = x; in.bmode = 1; in.R = RW; in.T = W;
        if ((rc = run_wgrad(m, b.conv_wg, in, wpart, grads + gl.conv_w, grads + gl.conv_b, s))) return rc;
    }
    return wpart.flush();            // every weight-gradient reduction of the step, one launch
}

/* The keep-masks of nn.GRU's dropout between stacked layers (reference modules.py:233 / :253): mask_gru (gru_n_layers - 1, batch, W, H),
 * mask_rec (recon_n_layers - 1, batch, W, recon_hid_dim); either may be NULL */
int mtadgat_backward_input(mtadgat_handle h, int64_t batch, const void* ws_, size_t ws_bytes, float* dx, void* stream) {
    int rc = check_train(h, batch, 0.f);
    if (rc) return rc;
    if (batch == 0) return 0;
    if (!ws_ || !dx) return fail(MTADGAT_ERR_INVALID, "null tensor");
    Model& m = h->m;
    BwdWorkspace w;
    plan_bwd_workspace(m, batch, w);
    if (!aligned16(ws_) || ws_bytes < w.total * sizeof(float)) return fail(MTADGAT_ERR_WORKSPACE, "backward workspace too small or misaligned");
    if ((size_t)m.W * m.F * sizeof(float) > 64 * 1024) return fail(MTADGAT_ERR_UNSUPPORTED, "input gradient: window too large for one workgroup's LDS");
    const float* ws = static_cast<const float*>(ws_);
    K_TRY(launch_conv_dx(ws + w.dpre, m.Fp, m.packed_dev + m.conv_wraw_off, batch, m.W, m.F, m.taps, m.pad, dx, (hipStream_t)stream), "input gradient");
    return 0;
}

int mtadgat_dropout_masks_rnn(mtadgat_handle h, int64_t batch, int64_t window0, float dropout_p, uint64_t seed, float* mask_gru,
                              float* mask_rec, void* stream) {
    if (!h) return fail(MTADGAT_ERR_INVALID, "null handle");
    if (batch <= 0) return 0;
    Model& m = h->m;
    hipStream_t s = (hipStream_t)stream;
    const DropArgs drop = make_drop(dropout_p, seed, window0);
    if (mask_gru)
        for (size_t l = 0; l + 1 < m.gru.size(); ++l)
            K_TRY(launch_dropmask(drop, DROP_GRU0 + (unsigned)l, batch, (long)m.W * m.gru[l].H, mask_gru + l * (size_t)batch * m.W * m.gru[l].H, s), "dropout mask");
    if (mask_rec)
        for (size_t l = 0; l + 1 < m.rec.size(); ++l)
            K_TRY(launch_dropmask(drop, DROP_REC0 + (unsigned)l, batch, (long)m.W * m.rec[l].H, mask_rec + l * (size_t)batch * m.W * m.rec[l].H, s), "dropout mask");
    return 0;
}

int mtadgat_dropout_masks(mtadgat_handle h, int64_t batch, int64_t window0, float dropout_p, uint64_t seed, float* mask_feat,
                          float* mask_temp, float* mask_fc, void* stream) {
    if (!h) return fail(MTADGAT_ERR_INVALID, "null handle");
    if (batch <= 0) return 0;
    Model& m = h->m;
    hipStream_t s = (hipStream_t)stream;
    const DropArgs drop = make_drop(dropout_p, seed, window0);
    if (mask_feat) K_TRY(launch_dropmask(drop, DROP_FEAT, batch, (long)m.F * m.F, mask_feat, s), "dropout mask");
    if (mask_temp) K_TRY(launch_dropmask(drop, DROP_TEMP, batch, (long)m.W * m.W, mask_temp, s), "dropout mask");
    if (mask_fc) {
        const int hid = m.cfg.forecast_hid_dim;
        for (size_t i = 0; i + 1 < m.fc.size(); ++i)
            K_TRY(launch_dropmask(drop, DROP_FC0 + (unsigned)i, batch, hid, mask_fc + i * (size_t)batch * hid, s), "dropout mask");
    }
    return 0;
}

}  // extern "C"
