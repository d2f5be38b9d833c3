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
    int rows_per_blk = 0, nblk = 0, IB = 0;      // attend launch plan (un-fused path)
    // fused per-window kernel (k_gat) plan; fused == false -> k_rowgemm + k_attend through HBM
    bool fused = false;
    int f_nw = 0, f_IBL = 0, f_JPL = 0, f_RJ = 16, f_vld = 0, f_lr = 0;
    size_t f_lds_bytes = 0;
};

struct GruPlan {
    int in_dim = 0, H = 0, Hp = 0, NCG = 0, Qx = 0, Qxp = 0;   // Qxp: packed x chunks (1 or a multiple of 3)
    int xmode = 0;          // 0 rows, 1 reference decoder input (modules.py:279)
    size_t wx_off = 0, wh_off = 0, b_off = 0, m0_off = 0;
};

struct LinPlan {
    int in_dim = 0, out_dim = 0, NT = 0, Q = 0;
    size_t w_off = 0, b_off = 0;
};

struct Model {
    mtadgat_config cfg{};
    int F = 0, W = 0, Fp = 0, Wp = 0, Dp = 0, taps = 0, pad = 0;
    // conv
    int convNT = 0;
    size_t conv_w_off = 0, conv_b_off = 0;
    GatPlan feat, temp;
    std::vector<GruPlan> gru, rec;
    std::vector<LinPlan> fc;
    LinPlan rec_fc;          // per-step Linear on the decoder state (tile format over Hp_r)
    size_t packed_floats = 0;
    float* packed_dev = nullptr;
    int packed_device = -1;          // device ordinal packed_dev was allocated on
    std::vector<float> staging;      // host image of the last upload (source of the async copy)
    hipEvent_t upload_ev = nullptr;  // recorded after the last upload
    bool have_weights = false;
    int64_t chunk = 65536;
    // profiling
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[MTADGAT_PROFILE_SLOTS];
};

struct Workspace {
    // offsets in floats for a chunk of `n` windows
    size_t xc, xct, lct, rtt, lcf, rtf, hcat, hend, seq0, seq1, fc0, fc1, rseq0, rseq1, total;
};

std::string validate_and_plan(Model& m);                       // "" on success
void plan_workspace(const Model& m, int64_t n, Workspace& ws); // sizes for n windows
// packs params into host buffer `out` (size m.packed_floats); returns "" on success
std::string pack_weights(Model& m, const mtadgat_params& p, std::vector<float>& out);

}  // namespace mtadgat
