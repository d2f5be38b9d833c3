// Host-side model state: validated configuration, derived dimensions, packed weights.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mtadgat.h"
#include "mtadgat_kernels.h"

namespace mtadgat {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Packed projection of one GAT layer (see mtadgat_pack.cpp / DESIGN.md section 3)
struct GatPlan {
    int K = 0;      // nodes            (feature layer: F, temporal layer: W)
    int D = 0;      // node dimension   (feature layer: W, temporal layer: F)
    int E = 0;      // rows of lin.weight
    int PT = 0;     // pairwise columns (pos + neg groups, each padded to 8)
    int P8 = 0;     // columns of the positive group
    int ldl = 0;    // row stride of the query-side rows [L'(PT) | c | pad], multiple of 32
    int rt_rows = 0;// rows per window of the key-side block R'^T: [R'(PT) ; d ; pad], multiple of 32
    int Kp = 0;     // row stride of R'^T (K padded to 4)
    int NT_L = 0;   // tiles of the query side; the key-side tiles follow (stored transposed)
    int NT = 0, Q = 0;
    size_t w_off = 0, b_off = 0, bias_off = 0;   // offsets (floats) into the packed buffer
    size_t ord_off = 0;     // [P8, PT, npos, 0] as ints in the packed buffer: what the kernels read (the device-side re-pack rewrites it)
    int PTcap = 0;          // upper bound of PT: E rounded up to 8, + 8
    int rows_per_blk = 0, nblk = 0, IB = 0;      // attend launch plan (un-fused path)
    // fused per-window kernel (k_gat) plan; fused == false -> k_rowgemm + k_attend through HBM
    bool fused = false;
    int f_nw = 0, f_IBL = 0, f_JPL = 0, f_RJ = 16, f_vld = 0, f_lr = 0;
    size_t f_lds_bytes = 0;
    int fh_full = 0, fh_short = 0;   // k_gath: row-owning waves with 16 rows / with 16 - 64 / RJ rows
    int fh_JPL = 0, fh_RJ = 16, fh_IBL = 4;   // k_gath's pair-grid blocking (8 lanes along the keys whenever that pads them less)
    int fh_lr = 0;          // k_gath: LDS floats of the L' / R' (and attention-row) region
    int fh_lr_buf = 0;      // k_gath run-ahead projection: floats per L' / R' buffer (two of them in fh_lr), 0 = one buffer
    int fh_vld = 0;         // k_gath (fp16-piece build of the fused kernel): piece pitch in halfs, LDS bytes
    size_t fh_lds_bytes = 0;
    int Q16 = 0;            // bf16 build of the fused projection: 16-feature chunks incl. the bias row
    size_t w16_off = 0;
    size_t w3_off = 0;      // split-bf16 pack [tile][Q16][piece][64] of the fused projection, derived on the device
    size_t w2h_off = 0;     // two fp16 pieces of S * W, [tile][Q16][2][64]; gscale_off: [bits of max |W|, S, 1 / S, 0]
    size_t gscale_off = 0;
    int npos = 0;           // embedding columns with a' >= 0 (they come first in the pack)
    // un-fused path (projections through HBM, wide layers): split-bf16 pack of the row GEMM's tiles [tile][uQ16][3][64] (k_rowgemm_x3)
    int uQ16 = 0;
    size_t uw3_off = 0;
};

struct LinPlan {
    int in_dim = 0, out_dim = 0, NT = 0, Q = 0;
    size_t w_off = 0, b_off = 0;
    // split-bf16 pack [tile][Q16][3 pieces][64] for launches of many rows (k_rowgemm_x3 / _x3s), derived on the device on first use
    // after an upload; only planned for the Linears that run over (window, step) rows: the GRU's hoisted input projection and
    // recon_model.fc
    int Q16 = 0;
    size_t w3_off = 0;
    mutable uint64_t w3_version = 0;
};

struct GruPlan {
    int in_dim = 0, H = 0, Hp = 0, NCG = 0, Qx = 0, Qxp = 0;   // Qxp: packed x chunks (1 or a multiple of 3)
    int xmode = 0;          // 0 rows, 1 reference decoder input (modules.py:279)
    size_t wx_off = 0, wh_off = 0, b_off = 0, m0_off = 0;
    // input projection of all steps as one row GEMM ahead of the recurrence (small batches, k_gru_split XMODE 3):
    // rows (b*T, in_dim) -> (b*T, 3*Hp) [W_ir x + b_ir + b_hr | W_iz x + b_iz + b_hz | W_in x + b_in]
    bool has_xproj = false;
    LinPlan xproj;
    mutable uint64_t split_ver = 0;   // Model::weights_version the layer's split packs (scale, wx3, wx2, wh3, wxq) were derived from
    // bf16 operand packs (16-feature chunks, element order of mtadgat_device.h): same streams, half the bytes per feature
    int Qxp16 = 0;          // packed input chunks (1, or a multiple of 3)
    size_t wx16_off = 0, wh16_off = 0;
    // small-batch recurrence (k_gru16 / k_gru16_bwd): W_hh and W_hh^T as 16x16x4 A operands [NT16][3][KS16][64]; the
    // decoder's folded input weights plain, [T][3][Hp][8]
    bool has16 = false;
    int KS16 = 0, NT16 = 0;
    size_t g16_off = 0, g16T_off = 0, fold_off = 0;
    // split-bf16 packs of the large-batch kernel (three bf16 pieces per weight, derived on the device from the fp32 packs)
    size_t wx3_off = 0, wh3_off = 0;   // wx3: three bf16 pieces of S * W_ih; wh3: TWO fp16 pieces of S * W_hh, [gate][piece] words (+ 3 words of slack)
    size_t wx2_off = 0;                // the input pack with all chunks on two fp16 pieces (layer 0, when the conv's range allows)
    size_t wxq_off = 0;                // row-input layers: the two-piece input pack in [chunk][tile][gate][piece] order (k_gru_cm)
    int qb3 = 0;                       // leading input chunks (of 16) kept on three bf16 pieces: the convolution's channels, rounded to 2
    size_t scale_off = 0;              // [bits of max |W| of the layer, S, 1 / S, 0]: S = the power of two that puts it into [2^13, 2^14)
    // one-window-per-workgroup recurrence (k_gru1 / k_gru1_bwd): [waves][16 * g1_ksm(H)][64], a lane's weights for all hidden indices
    size_t g1_off = 0, g1T_off = 0;
};
constexpr int64_t G16_MAX_WINDOWS = 4096;      // 16 windows per workgroup x 256 CUs: beyond that the throughput kernels take over
constexpr int64_t CM_MIN_WINDOWS = 4097;
// ... of which the hidden-tile-split kernel's split-operand build takes the lower band: five waves per 32 windows instead of
// one, a round of 8 192 windows (one workgroup per CU) in ~1.2 ms (GRU + decoder) against the 4 ms of a k_gru_cm round
constexpr int64_t SPLIT3_MIN_WINDOWS = 2561, SPLIT3_MAX_WINDOWS = 8192;      // k_gru_cm (chunk-major recurrence, 128 windows per workgroup) from here on
constexpr int64_t FORK_MAX_WINDOWS = 1024;   // up to here no launch of the forward fills the machine: independent stages run on two streams
constexpr int64_t G1_MAX_WINDOWS = 1792;       // up to 7 windows per CU one after the other; beyond that 16-window groups pay

// ---- backward (training) plans -------------------------------------------------------------------------
// transposed Linear for the data gradient d X = d Y W through k_rowgemm: rows of the pack = input features
struct LinTPlan {
    int in_dim = 0;      // features of d Y the product runs over (the pack's K)
    int out_dim = 0;     // features of d X
    int NT = 0, Q = 0;
    size_t w_off = 0;
    int Q16 = 0;         // 16-feature chunks of the split-bf16 pack [tile][Q16][3 pieces][64], derived on the device (k_rowgemm_x3)
    size_t w3_off = 0;
    mutable uint64_t w3_version = 0;   // the weight upload the split pack was derived from (derived on first use after an upload)
};
// one weight-gradient GEMM: shapes, index maps into the flat gradient buffer (ints stored in the packed buffer)
struct WgradPlan {
    int M = 0, N = 0, Mp = 0, Np = 0;
    size_t rowW_off = 0, col_off = 0, rowB_off = 0;     // offsets (floats) of the int maps in the packed buffer
    bool has_bias = false;
};
struct GatBwdPlan {
    int Ep = 0, NTu = 0;            // E rounded up to 32, tiles per side
    size_t wu_off = 0;              // un-scaled projection tiles [2*NTu][Q][64]
    size_t wu3_off = 0;             // ... as three bf16 pieces [2*NTu][(Q + 1) / 2][3][64], derived on first use after an upload (k_rowgemm_x3, GATv2)
    mutable uint64_t wu3_version = 0;
    size_t a_off = 0;               // a (Ep floats, zero padded)
    LinTPlan lrT;                   // d V += [dL | dR] [W_l ; W_r]
    WgradPlan wg;                   // lin.weight / lin.bias
    size_t att_lds = 0, pair_lds = 0;
    // wide layer (not fused: more than 128 nodes or node dimensions): the generic backward of mtadgat_bwdw.hip -- the un-scaled
    // projection [L | R] = V [W_l ; W_r]^T + [b | 0] is a row GEMM over the pack at wu_off with the bias vector at bu_off (2 Ep)
    bool wide = false;
    size_t bu_off = 0;
    // GAT (v1): plain copies of lin.weight (E x D), lin.bias (E), a (2E) for the score backward
    size_t w1_off = 0, b1_off = 0;
};
struct GruBwdPlan {
    size_t whT_off = 0;             // W_hh^T tiles for k_gru_bwd
    size_t whT3_off = 0;            // ... as three bf16 pieces [NCG][6*NCG][3][64], derived on first use after an upload (k_gru_bwd<true>)
    mutable uint64_t whT3_version = 0;
    LinTPlan wihT;                  // d x = d a W_ih
    WgradPlan wg_ih, wg_hh;
};
struct GradLayout {                 // offsets (floats) into the flat gradient buffer, reference parameter shapes
    int64_t conv_w = 0, conv_b = 0;
    int64_t lin_w[2] = {0, 0}, lin_b[2] = {0, 0}, a[2] = {0, 0}, bias[2] = {0, 0};   // [0] feature, [1] temporal
    std::vector<int64_t> gru_wih, gru_whh, gru_bih, gru_bhh;      // per GRU layer
    std::vector<int64_t> fc_w, fc_b;
    std::vector<int64_t> rec_wih, rec_whh, rec_bih, rec_bhh;      // per decoder layer
    int64_t rec_fc_w = 0, rec_fc_b = 0;
    int64_t total = 0;
};
struct BwdPlan {
    bool supported = false;
    std::string why;                // reason when not supported
    GatBwdPlan gat[2];              // [0] feature, [1] temporal
    std::vector<GruBwdPlan> gru, rec;   // per layer
    std::vector<LinTPlan> fcT;
    std::vector<WgradPlan> fc_wg;
    LinTPlan recfcT;                // d h_t (decoder) = d recons_t W_fc
    WgradPlan recfc_wg, conv_wg;
    size_t zero_off = 0;            // >= 1024 zero floats (bias of the transposed rowgemms)
    GradLayout gl;
};

// offsets (floats) of every reference parameter in the flat parameter buffer of mtadgat_update_weights_device: the
// fields of mtadgat_params in declaration order (the order of the flat gradient buffer)
struct FlatOffsets {
    int64_t conv_w = 0, conv_b = 0;
    int64_t lin_w[2] = {0, 0}, lin_b[2] = {0, 0}, a[2] = {0, 0}, bias[2] = {0, 0};   // [0] feature, [1] temporal
    std::vector<int64_t> gru_wih, gru_whh, gru_bih, gru_bhh, fc_w, fc_b, rec_wih, rec_whh, rec_bih, rec_bhh;
    int64_t rec_fc_w = 0, rec_fc_b = 0, total = 0;
};
// device-side re-packing (mtadgat_packdev.hip): index tables built once on the host by running the host packer over
// parameters whose values are their own flat indices
struct DevTables {
    bool ready = false;
    FlatOffsets fo;
    std::vector<int> gidx;            // [packed_floats]: flat parameter index copied to this position, or -1 (left alone)
    std::vector<int> gatcode[2];      // [NT*Q*256] of a layer's projection tiles: row * (D + 1) + k + 1, or 0
    std::vector<int> foldcode;        // [NCG*Qxp*3*256] of one step's folded decoder input tiles: (gate*H + r) * NMp + k + 1, or 0
    std::vector<int> colk[2];         // current embedding-column order of the two layers (sorted by the sign of a)
    int* gidx_dev = nullptr;
    int* gatcode_dev[2] = {nullptr, nullptr};
    int* foldcode_dev = nullptr;
    double* foldsum_dev = nullptr;    // [3 H][Hin + 1] running sums of the decoder's W_ih rows (k_fold_prefix -> k_pack_fold), rebuilt per re-pack
    int* colk_dev[2] = {nullptr, nullptr};
    float* pin = nullptr;             // pinned staging: the two a vectors coming down, the column orders going up
    size_t pin_floats = 0;
    int device = -1;
};

struct Model {
    mtadgat_config cfg{};
    int F = 0, W = 0, Fp = 0, Wp = 0, Dp = 0, taps = 0, pad = 0;
    // conv
    int convNT = 0;
    size_t conv_w_off = 0, conv_b_off = 0;
    int Fp16 = 0;                    // bf16 conv build: channels padded to 16
    size_t conv_w16_off = 0;
    size_t conv_wf16_off = 0;        // fp32 tiles in the 16-channel geometry (source of the split-bf16 pack)
    size_t conv_wraw_off = 0;        // conv.weight as the reference stores it, (F, F, taps): the input gradient of mtadgat_backward_input
    size_t conv_w3_off = 0;          // three bf16 pieces of those tiles [tile][taps Fp16 / 16][3][64], derived on the device (k_conv_x3: wide models)
    size_t conv_w2h_off = 0, conv_scale_off = 0;   // k_conv_win: two fp16 pieces of S * W [tile][taps * Fp16 / 16][2][64], [bits of max |W|, S, 1 / S, 0]
    GatPlan feat, temp;
    std::vector<GruPlan> gru, rec;
    std::vector<LinPlan> fc;
    LinPlan rec_fc;          // per-step Linear on the decoder state (tile format over Hp_r)
    BwdPlan bw;
    size_t packed_floats = 0;
    float* packed_dev = nullptr;
    int packed_device = -1;          // device ordinal packed_dev was allocated on
    float* staging_pinned = nullptr; // pinned host image of the last upload (source of the async copy)
    size_t staging_floats = 0;
    hipEvent_t upload_ev = nullptr;  // recorded after the last upload
    // second lane of forward(): pieces of a call alternate between the caller's stream and this one (forward_schedule, mtadgat_capi.cpp)
    hipStream_t lane_stream = nullptr;
    hipEvent_t lane_begin = nullptr, lane_end = nullptr;
    hipEvent_t fork_ev[3] = {nullptr, nullptr, nullptr};   // small calls: stage hand-offs between the caller's stream and the second lane
    int lane_device = -1;            // the device the second lane's stream and events were created on
    int lanes = 0;                   // 0 automatic, 1 everything on the caller's stream
    bool have_weights = false;
    bool bf16_packed = false;        // the bf16 streams of the packed image are current (packed only when precision == 1 at load time)
    int precision = 0;               // 0: fp32 operands (default, <= 1e-5 parity); 1: bf16 MFMA operands, fp32 accumulate / state
    int64_t chunk = 65536;
    int wgrad_kernel = 0;            // weight-gradient GEMMs of the training step (testing hook): 0 automatic (split-bf16 operands in mode 2), 1 fp32 MFMA, 2 split-bf16 always
    int conv_shared = 0;             // series scoring (testing hook): 1 keeps the shared-row convolution where the window-per-workgroup kernel would run
    uint64_t weights_version = 0;    // counts weight uploads / device-side re-packs
    // the split packs (two-fp16-piece / three-bf16-piece copies of the fp32 packs, power-of-two scales) are derived on the device
    // on first use after an upload: the upload each of them was last derived from (ensure_*_split, mtadgat_capi.cpp)
    uint64_t split_ver_conv = 0, split_ver_gat[2] = {0, 0};
    int rowgemm_kernel = 0;          // data-gradient row GEMMs of mtadgat_backward in mode 2 (testing hook): 0 automatic (split-bf16 operands from 4096 rows), 1 fp32 MFMA, 2 split-bf16 always
    int conv_kernel = 0;             // convolution of the fused front end in mode 2 (testing hook): 0 automatic (k_conv_win from 4096 windows), 1 k_conv_lds, 2 k_conv_win at any batch size
    int gath_dbg = 0;                // measurement hook: GatArgs::dbg of k_gath (knock-out bits, profiles/gath_knockout.py)
    int conv_fused = 0;              // the convolution inside the temporal layer's k_gath workgroup: 0 automatic (wherever both kernels apply), 1 off
    int gat_kernel = 0;              // fused attention layers in the split-operand arithmetic (testing hook): 0 automatic (k_gath, the fp16-piece build, from 4096 windows), 1 k_gat only, 3 k_gath at any batch size
    int gru_kernel = 0;              // large-batch recurrence: 0 automatic, 1 tile-major k_gru, 2 chunk-major k_gru_cm (testing hook: mtadgat_set_option)
    DevTables dt;
    // profiling
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[MTADGAT_PROFILE_SLOTS];
};

struct Workspace {
    // offsets in floats for a chunk of `n` windows
    size_t xc, xct, lct, rtt, lcf, rtf, hcat, hend, seq0, seq1, fc0, fc1, rseq0, rseq1, xp, total;
    bool has_xp;         // room for the pre-projected GRU input (batches the hidden-tile-split kernel serves)
    bool rec16;          // room for the decoder's pre-projected input and state sequence (k_gru16)
    size_t vmax;         // one word: bits of the largest convolution output of the chunk (range guard of the fp16 operand pieces)
    size_t winflag;      // one byte per window: the fused convolution's per-window range flag (k_gath CONV build -> k_gat)
    size_t cf, el, er;   // convolution rows shared by stride-1 windows of a series (run_conv_shared): segment rows, edge rows
};

// activations kept between the training forward and the backward (caller-owned "tape"), offsets in floats
struct Tape {
    size_t hcat, xct, att_f, att_t, hend, gates_g, seq_g, gates_d, seq_d, xdec, xp, total;
    size_t vmax;         // one word: bits of the largest convolution output (range guard of the split-operand recurrences)
    size_t lct, rtt, lcf, rtf;   // wide attention layers: the projections L' / R'^T of the training forward (zero-sized for fused layers)
    // stacked recurrences: gates / state sequences of the layers above the first, and the (dropped-out) state sequences
    // that feed them (nn.GRU's inter-layer dropout, modules.py:233 / :253)
    std::vector<size_t> gates_gu, seq_gu, drop_g, gates_du, seq_du, drop_d;
    std::vector<size_t> fc_act;     // outputs of the hidden forecasting layers (after ReLU + dropout)
};
// scratch of the backward
struct BwdWorkspace {
    size_t da, dhcat, dhdec, dhend, dz0, dz1, de_f, de_t, dv_f, dv_t, dlr_f, dlr_t, dap_f, dap_t, dpre, wpart, sums, total;
    size_t sums_q[4];    // GATv2: first-stage column sums of d e / d a' of the two layers, reduced in the step's batched reduction launch
    size_t v1s;          // GAT (v1): [u1 | u2 | k1 k2] and the batch sums [P1 | P2 | SC SD] of the two layers
    size_t wds, wlr;         // wide attention layers (one layer at a time): d S (N K ldS), [L | R] (N K 2 Ep)
    size_t wpart_floats;
};

std::string validate_and_plan(Model& m);                       // "" on success
FlatOffsets flat_offsets(const Model& m);
void params_from_flat(const Model& m, const FlatOffsets& fo, const float* flat, mtadgat_params& p);
std::string build_device_tables(Model& m);                     // host side of DevTables ("" on success)
void gat_column_order(const float* a, int E, double alpha, std::vector<int>& colk, int& P8, int& PT, int* npos = nullptr);
void plan_workspace(const Model& m, int64_t n, Workspace& ws); // sizes for n windows
void plan_tape(const Model& m, int64_t n, Tape& t);
void plan_bwd_workspace(const Model& m, int64_t n, BwdWorkspace& w);
int wgrad_slabs(long R, int Mp, int Np);                       // row slabs of one weight-gradient GEMM
// packs params into host buffer `out` (size m.packed_floats); returns "" on success
std::string pack_weights(Model& m, const mtadgat_params& p, std::vector<float>& out);

}  // namespace mtadgat
