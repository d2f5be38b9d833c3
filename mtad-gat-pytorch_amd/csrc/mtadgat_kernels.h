// Kernel argument blocks and launchers (internal; the public ABI is include/mtadgat.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mtadgat {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Y[r,:] = act(W X[r,:] + bias); W packed as tiles [NT][Q][64 lanes] of float4 (mtadgat_pack.cpp)
struct RowGemmArgs {
    const float* X;      // (R, ldx) data rows
    long ldx;
    int Kvalid;          // valid input features per row
    int Q;               // ceil(Kvalid / 8)
    const f32x4* Wp;     // packed weight tiles
    const float* bias;   // NT*32 floats (zero padded)
    float* Y;            // (R, ldy)
    long ldy;
    int Nvalid;          // columns stored: [0, Nvalid)
    int vec_store;       // 1: ldy % 4 == 0 and Y 16-byte aligned
    long R;
    int NT;              // output tiles of 32 columns
    int relu;
    // tiles n >= NT_rm are stored transposed per group of `group` consecutive rows:
    //   YT[(row / group) * YT_rows + (col - 32*NT_rm)) * YT_ld + row % group]
    int NT_rm;           // number of leading row-major tiles (== NT when nothing is transposed)
    float* YT;
    int group, YT_rows, YT_ld;
    // ---- training-path epilogue options (all off when zero-initialised)
    int accumulate;      // 1: Y += result (row-major tiles only)
    const float* gate;   // backward through ReLU (+ dropout): result = gate[row*ldg + col] > 0 ? result * gate_scale : 0
    long ldg;
    float gate_scale;
    // forward dropout after the ReLU (Forecasting_Model, reference modules.py:309-310): element (row0 + row, col)
    // of dropout stream `drop_stream` is zeroed with probability drop_thresh / 2^32, kept ones scaled by keep_scale
    unsigned drop_thresh, seed_lo, seed_hi, drop_stream;
    float keep_scale;
    long row0;
    // split-bf16 build (k_rowgemm_x3): three bf16 pieces per operand on the 16-bit matrix pipe, fp32-class products
    int x3;              // 1: use Wp3 / Q16 instead of Wp / Q
    const f32x4* Wp3;    // [tile][Q16][3 pieces][64]
    int Q16;
};

// dropout streams (one per dropout site of the reference: modules.py:90, :189, :310)
enum { DROP_FEAT = 1, DROP_TEMP = 2, DROP_FC0 = 16, DROP_GRU0 = 64, DROP_REC0 = 96 };    // GRU0 / REC0 + l: between stacked layers l and l + 1

struct DropArgs {
    unsigned thresh, seed_lo, seed_hi;   // P(drop) = thresh / 2^32; thresh == 0: no dropout
    float keep_scale;                    // 1 / (1 - p)
    long win0;                           // global index of window 0 of this launch (masks do not depend on chunking)
};

struct ConvArgs {
    const float* X;      // (B, W, F) windows, or -- gather mode -- the series (n_rows, F)
    // gather mode (starts != null or stride > 0): window w = series rows [s_w, s_w + W), s_w = starts[w] or
    // start0 + w*stride  (reference: SlidingWindowDataset.__getitem__, utils.py:114-117)
    const long* starts;
    long start0, stride;
    int gather;
    long B;
    int W, F, Fp, taps, pad;
    int Fq;              // channel count the MFMA chunks run over: Fp (fp32 build), F rounded up to 16 (bf16 build)
    int bf16;            // 1: Wp is the bf16 pack (k_conv_lds only)
    unsigned* vmax;      // k_conv_lds: bits of the largest output written so far (atomic max), or null
    int x_bf16;          // 1: X holds bfloat16 (read directly by k_conv_lds; no fp32 copy of the input exists)
    int pvh;             // k_conv_win: LDS pitch of the staged pieces (halfs), set by its launcher
    int dbg;             // k_conv_win measurement hook (bit 0: no MFMA loop, 1: no epilogue, 2: loads + halo only)
    const float* wscale; // k_conv_win: [S, 1 / S] of its two-fp16-piece weight pack (Wp then points at that pack, Fq = F rounded up to 16)
    const f32x4* Wp;     // packed (F x taps*Fq), NT tiles, Q = taps*Fq/8 (bf16: /16)
    const float* bias;   // NT*32
    int NT;
    float* XC;           // (B*W, Fp)   or null
    float* XCT;          // (B*F, Wp)   or null
    int Wpad;            // row stride of XCT
    float* HCAT;         // (B*W, Dp) columns [0,F)   or null
    int Dp;
    float* Y;            // (B*W, F) plain output   or null
    const f32x4* Wp3;    // split-bf16 pack [tile][taps Fq / 16][3][64] (k_conv_x3; Fq = F rounded up to 16)
};

struct AttendArgs {
    const float* LC;     // (B*K, ldl) per query node: [L'(PT) | c | pad]
    const float* RT;     // (B, rt_rows, Kp) per window, key-node-minor: rows [0,PT) = R', row PT = d
    int ldl, rt_rows, Kp, PT, P8;
    const int* ord;      // device-side [P8, PT] (overrides PT / P8 when set)
    const float* bias;   // (K, K) or null
    const float* V;      // (B*K, ldv) node feature rows
    int ldv, D;
    float* out;          // out[win*so_w + i*so_i + d*so_d]
    long so_w, so_i, so_d;
    int K, rows_per_blk, nblk;
    long total_blocks;   // round_up(B, 8) * nblk
    long nwin;           // B
    int xcd_map;         // 1: XCD-aware block -> (window, row block) map
    int v1;
    float alpha;
    float* ATT;          // optional (B, K, K) dump of the attention matrix
};

// the window convolution computed inside the temporal layer's k_gath workgroup (mtadgat_gath.hip, CONV build): what k_conv_win
// takes (ConvArgs), for one window per workgroup
struct GatConvIn {
    const float* X;      // (B, W, F) windows, or -- gather -- the series (n_rows, F); bfloat16 elements when x_bf16
    const long* starts;
    long start0, stride;
    int gather, x_bf16;
    int taps, pad, Fq, NT;
    int pvx;             // LDS pitch (halfs) of the staged input pieces: conv_win_pitch(F, Fq)
    int Dp;              // row stride of h_cat
    const f32x4* Wp;     // two fp16 pieces of S * W, [tile][taps Fq / 16][2][64]
    const float* bias;   // NT * 32
    const float* wscale; // [S, 1 / S]
    float* HCAT;         // (B*W, Dp): columns [0, F) and the zero padding [3F, Dp) are written
    unsigned* vmax;      // bits of the largest output written so far (atomic max; zeroed by the caller), or null
    unsigned char* flag; // (B): 1 = the window's outputs reach 2^15 -- k_gat's bf16-piece build has to serve it; or null
};

// fused per-window graph-attention layer (projection + scores + softmax + aggregation in one workgroup)
struct GatArgs {
    const float* V;      // vt == 0: (B*K, ldv) node rows; vt == 1: (B*D, ldv) rows whose columns are the nodes
    int ldv, D, K;
    int vt;
    int vld;             // LDS row stride of the staged V rows
    int lr_floats;       // LDS floats reserved for L' / R' (and the aliased softmax rows) ahead of the V rows
    const f32x4* Wp;     // packed projection tiles [2*NT_L][Q][64]: query-side tiles then key-side tiles
    int bf16;            // 1: Wp is the bf16 pack, Q counts 16-feature chunks; 2: split pack (three bf16 pieces per weight)
    // bf16 == 2 and the node values are known to fit fp16 (vmax: bits of the largest |value| the producing convolution
    // wrote, < 2^15): two fp16 pieces instead -- Wp2 = [tile][Q][2 pieces][64] of S * W, scale2 = [S, 1 / S]
    const unsigned* vmax;
    const f32x4* Wp2;
    const float* scale2;
    const float* pbias;  // projection bias, 2*NT_L*32
    int NT_L, Q, PT, P8;
    // the sign-group boundaries [P8, PT] as kept in device memory by the weight packers (the device-side re-pack derives the column
    // order of the folded projection from the signs of `a` without telling the host): when set it overrides PT / P8 above
    const int* ord;
    const float* bias;   // (K, K) attention bias or null
    float* out;          // out[win*so_w + i*so_i + d*so_d]
    long so_w, so_i, so_d;
    long nwin;
    int v1;
    float alpha;
    // training mode: the softmax rows (before dropout) are kept for the backward, dropout (modules.py:90 / :189)
    // is applied to the attention matrix inside the kernel
    float* ATT;          // (B, K, K) or null
    DropArgs drop;
    unsigned drop_stream;
    int n_full, n_short; // k_gath: waves owning 16 query rows / 16 - 64 / RJ query rows (the rest of the workgroup only projects)
    int dbg;             // k_gath measurement hooks: knock-outs (bit 0: no pair grid, 1: no projection, 2: return before the softmax; results invalid) and
                         // sensitivity probes (bit 3: ~5 k idle cycles ahead of the convolution, 4: 1 000 extra VALU instructions per wave ahead of
                         // the first pair grid, 5: ~5 k idle cycles there; results unchanged) -- profiles/r06_gath_experiments.txt
    int E;               // k_gath: > 0 = embedding columns of a GATv2 layer whose Wp2 pack is in the compact column order (sign per 2-column step)
    int lr_buf;          // k_gath: > 0 = floats per L' / R' buffer of the run-ahead projection (two buffers inside lr_floats, K rows of L' each); 0 = one buffer
    int skip_h;          // 1: return at once when *vmax < 2^15 -- k_gath (launched ahead of this kernel) serves that case
    const unsigned char* winflag;   // k_gat behind a CONV launch of k_gath: serve exactly the windows whose flag is set
    GatConvIn cv;        // k_gath, CONV build
};

// backward of one graph-attention layer, part 1 (per window): d e_ij (the gradient of the attention scores
// = of the layer's bias before the sum over windows) and the aggregation path's d V
struct GatBwdAttArgs {
    const float* V;      // node rows as in GatArgs (vt: transposed source)
    int ldv, D, K, vt, vld;
    const float* H;      // the layer's forward output, H[win*so_w + i*so_i + d*so_d]
    const float* dH;     // its gradient, same indexing
    long so_w, so_i, so_d;
    const float* ATT;    // (B, K, K) softmax rows kept by the forward
    float* DE;           // (B, K, K) out
    float* DV;           // (B*K, lddv) out: sum_i att'_ij dS_i (node-major rows)
    int lddv;
    long nwin;
    DropArgs drop;
    unsigned drop_stream;
};

struct GruBwdArgs {
    const float* Gates;  // (B*T, 4*Hp): r | z | n | q (= W_hn h + b_hn) kept by the forward
    const float* Seq;    // (B*T, Hp): h_t
    const float* DHseq;  // (B*T, lddh) external gradient of every h_t, or null
    long lddh;
    const float* DHend;  // (B, ldde) external gradient of h_T, or null
    long ldde;
    const f32x4* WhT;    // [NCG][12*NCG][64]: W_hh^T tiles over the [dr | dz | dnh] features
    float* DA;           // (B*T, 4*Hp) out: dn_x | dr | dz | dn_h  (pre-activation gradients)
    int Hp, H, T, NCG;
    long B;
    int bf16;            // (unused: the bf16 training recurrences are gone; a launch with bf16 != 0 is refused)
    int x3;              // 1: WhT is the split-bf16 pack [NCG][6*NCG][3 pieces][64] of 16-feature chunks (k_gru_bwd<true>)
};

// dW[m][n] = sum_rows A[row][m] * B[row][n]   (+ a virtual all-ones column n == N: bias gradients)
struct WgradArgs {
    const float* A;      // (R, lda), M valid columns
    long lda;
    int M;
    int bshift;          // bmode 0: B rows are taken one step earlier, B[row - 1], zero at the first step of a window
                         // (the all-ones column is not shifted: bias gradients sum over every step)
    const float* B;      // bmode 0: (R, ldb), N valid columns
    long ldb;
    int N;
    int bmode;           // 0 plain rows; 1 conv im2col: B = x (b, T, F), column n = tap*F + ch -> x[b, t + tap - pad, ch]
    int T;               // steps per window (ashift / bmode 1: row = win*T + t)
    int F, taps, pad;
    long R;
    long rows_per_slab;
    int nslab;
    float* P;            // partial sums [nslab][Mp][Np],  Mp = 32*ceil(M/32), Np = 32*ceil((N+1)/32)
    int Mp, Np;
    int x3;              // 1: products from three bf16 pieces per operand on the 16-bit matrix pipe (fp32-class sums)
    int plain_map;       // measurement hook: 1 = no XCD-aware tile / slab remap
    int noskip;          // measurement hook: 1 = the tiles that lie wholly in the padding of d W are issued as well
};

// out[rowmap[m] + colmap[n]] += sum_slab P[slab][m][n];  column N (ones) -> outB[rowmapB[m]]
struct WgradReduceArgs {
    const float* P;
    int nslab, Mp, Np, M, N;
    const int* rowmapW;  // element offset of row m in outW, or -1.  null: a plain vector sum (M = 1): outW[n] += sum_s P[s*Np + n]
    const int* colmap;   // element offset of column n, or -1
    const int* rowmapB;  // element offset in outB, or -1  (null: no bias output)
    float* outW;
    float* outB;
};

// several reductions in ONE launch (round 6: a training step had twelve of them, 5 - 7 us each, every one behind its GEMM)
constexpr int WGRAD_BATCH_MAX = 16;
struct WgradReduceBatch {
    WgradReduceArgs e[WGRAD_BATCH_MAX];
    unsigned first_block[WGRAD_BATCH_MAX + 1];     // workgroups [first_block[i], first_block[i + 1]) serve entry i
    int n;
};

struct GruArgs {
    const float* X;      // XMODE 0: (B*T, ldx) input rows; XMODE 1: (B, ldx) hin rows
    long ldx;
    int Kx;              // valid input features (XMODE 1: valid hin entries)
    int Qx;              // real input chunks per step
    int Qxp;             // packed input chunks per step (1, or Qx rounded up to a multiple of 3 with zero chunks)
    const int* m0;       // XMODE 1: first hin index used at step t
    const f32x4* Wx;     // [c][Qxp][3][64]  (decoder input: [t][c][Qxp][3][64])
    const f32x4* Wh;     // [c][whs][3][64]
    const float* bias;   // [4][Hp]: b_ir+b_hr | b_iz+b_hz | b_in | b_hn
    int whs;             // chunks per tile in Wh: 4*NCG + 2 (the last two all zero: ring padding of k_gru_split)
    int Hp, H, T;
    long B;
    float* Hend;         // (B, ldhe) or null
    long ldhe;
    float* Seq;          // (B*T, ldseq) or null
    long ldseq;
    const f32x4* Wfc;    // [NTfc][4*NCG][64]
    const float* bfc;    // NTfc*32
    int NTfc;
    float* Yfc;          // (B*T, out_dim) or null (then only the last step is produced)
    float* Ylast;        // (B, out_dim): the per-step Linear at the last step only, or null
    int out_dim;
    float* Gates;        // training: (B*T, 4*Hp) r | z | n | q kept for the backward (k_gru_split only), or null
    int bf16;            // 1: Wx / Wh are bf16 packs (16-feature chunks): bf16 MFMA operands, fp32 accumulation and state
    int x3;              // 1 (with bf16 = 1): split packs; k_gru (large batches) only.  Wx: three bf16 pieces per weight,
                         // [gate][piece] words per chunk; Wh: two fp16 pieces, [gate][piece]; all weights scaled by S
    const float* scale;  // x3: device pointer to [S, 1 / S] (a power of two chosen per layer at load time)
    int qb3;             // x3: leading input chunks on three bf16 pieces (9 words); the others on two fp16 pieces (6 words)
    const unsigned* vmax; // x3, layer 0: bits of the largest value the producing convolution wrote, or null (unknown)
    const f32x4* Wx2;    // x3: the input pack with ALL chunks on two fp16 pieces (used when *vmax < 2^15)
    // chunk-major recurrence (k_gru_cm, mtadgat_gru_cm.hip): the two-piece input pack re-ordered [chunk][tile][gate][piece]
    // (the decoder's per-step pack already is one contiguous block per step)
    const f32x4* Wxq;
    int skip_xh;         // k_gru: return at once when *vmax < 2^15 (k_gru_cm serves that case)
};

// small-batch recurrences (mtadgat_gru16.hip): 16 windows per workgroup, one wave per 16-unit hidden tile, weights in registers
struct Gru16Args {
    const float* XP;     // (B*T, 3*Hp): input products of every step incl. biases [r | z | n]
    const float* W16;    // [NT16][3][KS][64]: W_hh as v_mfma_f32_16x16x4_f32 A operands
    const float* bias;   // [4][Hp]; row 3 = b_hn
    int Hp, KS, NT16, T; // KS = ceil(H / 4) k-steps, NT16 = ceil(H / 16) tiles (one wave each, at most 10)
    int H;               // k_gru1: hidden size (W16 then is the k_gru1 pack [waves][16 * g1_ksm(H)][64])
    long B;
    float* Hend;         // (B, ldhe) or null; columns below ncol are written
    long ldhe;
    int ncol;
    float* Seq;          // (B*T, Hp) or null
    float* Gates;        // training: (B*T, 4*Hp) r | z | n | q, or null
};
struct Gru16BwdArgs {
    const float* Gates;  // (B*T, 4*Hp)
    const float* Seq;    // (B*T, Hp)
    const float* DHseq;  // (B*T, lddh) or null
    long lddh;
    const float* DHend;  // (B, ldde) or null
    long ldde;
    const float* W16T;   // [NT16][3][KS][64]: W_hh^T per gate block [dr | dz | dnh]
    float* DA;           // (B*T, 4*Hp) out: dn_x | dr | dz | dn_h
    int Hp, KS, NT16, T;
    int H;               // k_gru1_bwd (W16T then is its pack)
    long B;
};
// k_gru1 / k_gru1_bwd geometry, shared with the packer: groups of 16 hidden indices held in registers, waves per workgroup
inline int g1_ksm(int H) {
    const int ks = (H + 15) / 16;
    return ks <= 3 ? 3 : ks <= 6 ? 6 : ks <= 8 ? 8 : 10;
}
inline int g1_waves(int H, int Hp, bool bwd) {
    const int rows = bwd ? (3 * ((H + 15) / 16) + 3) / 4 : (3 * H + 63) / 64;      // bwd: 16-lane rows = (gate block, 16 outputs)
    const int gate = (Hp + 63) / 64;                                                   // lanes of the gate phase
    return rows > gate ? rows : gate;
}

// device-side re-packing (mtadgat_packdev.hip); offsets are floats into the flat parameter buffer
struct PackGatArgs {
    const float* flat;
    long lin_w, lin_b, a;
    int E, D, KS;        // KS = projected columns per side (ldl)
    const int* ord;      // [P8, PT, npos] of the layer (k_gat_colorder); PT: see GatArgs
    int v2, fused;
    double alpha;
    const int* colk;     // [PT] embedding column of sorted column n, or -1 (padding)
    const int* code;     // [n_code] tile position -> row * (D + 1) + k + 1, or 0
    int n_code, n_bias;
    float* w_out;        // projection tiles
    float* b_out;        // [n_bias] bias vector of the un-fused path
};
struct PackFoldArgs {
    const float* flat;
    long wih;
    int Hin, T, H, Hp, NMp;
    const int* code;     // [tile_floats] position in one step's tiles -> (gate*H + r) * NMp + k + 1, or 0
    long tile_floats;
    float* tiles_out;    // [T][tile_floats]
    float* fold_out;     // [T][3][Hp][8] or null
    double* prefix;      // [3 H][Hin + 1] scratch: prefix[R][j] = sum_{j' < j} W_ih[R][j'] (filled by the launcher's first kernel)
};

constexpr int FINGERPRINT_MAX_TENSORS = 48;
struct FingerprintArgs {
    const void* ptr[FINGERPRINT_MAX_TENSORS];
    long count[FINGERPRINT_MAX_TENSORS];    // 32-bit elements
    long base[FINGERPRINT_MAX_TENSORS];     // global index of the tensor's first element
};

// compute units of the current device (cached per device ordinal)
inline int cu_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cache[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev] = n;
    }
    return cache[dev];
}

int launch_rowgemm(const RowGemmArgs& a, hipStream_t s);
// measurement / test hook (process-wide): 1 = the many-row launches of the split-bf16 row GEMM and wide convolution use the one-wave
// kernels (k_rowgemm_x3 / k_conv_x3) instead of the workgroup kernels that share the weight words through LDS
void set_gemm_lds_off(int off);
int launch_conv(const ConvArgs& a, hipStream_t s);
bool conv_win_applies(const ConvArgs& a);
int launch_conv_win(const ConvArgs& a, hipStream_t s);
void attend_plan(int K, int* rows_per_blk, int* nblk, int* IB);
int launch_attend(const AttendArgs& a, int IB, hipStream_t s);
int launch_gat(const GatArgs& a, int IBL, int JPL, int rj, int nw, size_t lds_bytes, hipStream_t s);
int launch_gath(const GatArgs& a, int IBL, int JPL, int rj, int nw, size_t lds_bytes, bool conv, hipStream_t s);
bool gath_conv_applies(const GatArgs& a, int nw, int F, int W);
int conv_win_pitch(int F, int Fq);
size_t conv_win_lds(int W, int F, int Fq, int taps);
int launch_gat_wide(const float* LC, const float* RT, int ldl, int rt_rows, int Kp, int PT, int P8, const float* bias,
                    const float* V, int ldv, int D, int K, float* out, long so_w, long so_i, long so_d, long nwin, int v1,
                    float alpha, hipStream_t s, float* att = nullptr, const DropArgs* drop = nullptr, unsigned drop_stream = 0,
                    const int* ord = nullptr);
int launch_gat_colorder(const float* a_dev, int E, double alpha, int* colk_dev, int ncolk, int* ord_dev, hipStream_t s);
// backward of wide graph-attention layers (mtadgat_bwdw.hip)
int launch_bw_ds(const float* H, const float* dH, long so_w, long so_i, long so_d, long nwin, int K, int D, float* dS, int ldS, hipStream_t s);
int launch_bgemm(const float* A, long sAb, long sAm, long sAk, const float* B, long sBb, long sBk, long sBn, float* C, long sCb, long ldc,
                 int M, int N, int Kc, long nb, const DropArgs* drop, unsigned drop_stream, int dropK, hipStream_t s);
int launch_bw_softmax(const float* ATT, float* DE, long nwin, int K, const DropArgs& drop, unsigned drop_stream, hipStream_t s);
// GAT (v1) score backward of a wide layer: k_gat_bwd_v1's outputs with the node rows read from memory (K <= 512)
int launch_bw_v1(const float* Vn, long ldv, int D, int K, const float* u, const float* DE, float alpha, float* DV, int lddv, float* part,
                 long nwin, hipStream_t s);
int launch_bw_pair(const float* LR, int ldlr, int Ep, const float* avec, const float* DE, int K, float alpha, float* DLR, float* DAp,
                   long nwin, hipStream_t s);
int launch_gru(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s);
int launch_gru_train(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s);     // always the hidden-tile-split kernel
bool gru_cm_supported(int ncg, int xmode, bool fc, int out_dim);
int launch_gru_cm(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s);
int launch_gru_split_x3(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s);
int launch_reorder_xq(const float* src, float* dst, int ncg, int Qd, hipStream_t s);
long gru_split_max_windows();
int launch_gru_bwd(const GruBwdArgs& a, hipStream_t s);
int launch_fingerprint(const FingerprintArgs& a, int n_tensors, unsigned long long* out, hipStream_t s);
int launch_split3(const float* src, float* dst, long n_outer, int Qs, int Qd, int G, const float* scale, hipStream_t s);
int launch_split_x(const float* src, float* dst, long n_outer, int Qs, int Qd, int qb, const float* scale, hipStream_t s);
int launch_split2h(const float* src, float* dst, long n_outer, int Qs, int Qd, int G, const float* scale, hipStream_t s);
// k_gath's pack of a GATv2 projection: two fp16 pieces in the compact column order (mtadgat_packdev.hip)
int launch_split2h_gath(const float* src, float* dst, int NT_L, int Qs, int Qd, const int* ord, int E, const float* scale, hipStream_t s);
int launch_absmax(const float* src, long n, float* sc, hipStream_t s);
int launch_scale_from_max(float* sc, hipStream_t s);
int launch_pack_gather(const float* flat, const int* gidx, float* img, long n, hipStream_t s);
int launch_pack_gat(const PackGatArgs& a, hipStream_t s);
int launch_pack_gru_bias(const float* flat, long bih, long bhh, int H, int Hp, float* b, float* bx, hipStream_t s);
int launch_pack_fold(const PackFoldArgs& a, hipStream_t s);
int launch_gru16(const Gru16Args& a, hipStream_t s);
int launch_gru1(const Gru16Args& a, hipStream_t s);
int launch_gru1_bwd(const Gru16BwdArgs& a, hipStream_t s);
int launch_gru16_bwd(const Gru16BwdArgs& a, hipStream_t s);
int launch_xproj_dec(const float* hend, long ldh, int Hin, const float* fold, const int* m0, const float* bias, int Hp, int T, long B,
                     float* XP, hipStream_t s);
int launch_gat_bwd_att(const GatBwdAttArgs& a, int IBL, int JPL, int rj, int nw, size_t lds_bytes, hipStream_t s);
size_t gat_bwd_att_lds(int K, int D, int vld, int nwa);
int launch_wgrad(const WgradArgs& a, hipStream_t s);
int launch_wgrad_reduce(const WgradReduceArgs& a, hipStream_t s);
int launch_wgrad_reduce_batch(const WgradReduceArgs* e, int n, hipStream_t s);
// dst[n] += sum_r src[r*ld + n]  (n < N), two-stage through `scratch` (>= sum_rows_scratch(R, N) floats)
size_t sum_rows_scratch(long R, int N);
int launch_sum_rows(const float* src, long ld, long R, int N, float* scratch, float* dst, hipStream_t s);
// the first stage alone: partial sums [*nslab][N] into `scratch`; the caller queues their reduction (WgradReduceArgs with null maps)
int launch_sum_rows_part(const float* src, long ld, long R, int N, float* scratch, int* nslab, hipStream_t s);
// GAT (v1) score backward (mtadgat_bwd.hip)
size_t gat_bwd_v1_lds(int K, int D);
int launch_gat_v1_prep(const float* Wm, const float* bv, const float* av, int E, int D, float* u, hipStream_t s);
int launch_gat_bwd_v1(const float* V, int ldv, int D, int K, int vt, const float* u, const float* DE, float alpha, float* DV, int lddv,
                      float* part, long nwin, hipStream_t s);
int launch_gat_v1_finish(const float* P, const float* Wm, const float* bv, const float* av, int E, int D, float* gW, float* gb, float* ga,
                         hipStream_t s);
// decoder input of the reference (modules.py:279) materialised: X[(b*T + t)*ldx + j] = hend[b*ldh + (t*H + j) / T]
int launch_xdec(const float* hend, long ldh, int H, int T, long B, float* X, long ldx, hipStream_t s);
// and its adjoint: dhend[b*ldh + m] += sum over the flat positions f = t*H + j with f / T == m of dX[(b*T + t)*ldx + j]
int launch_xdec_bwd(const float* dX, long ldx, int H, int T, long B, float* dhend, long ldh, hipStream_t s);
// dpre[(b*T + t)*ldp + f] = xc > 0 ? dhcat[.., f] + dvt[(b*T + t)*ldt + f] + dvf[(b*F + f)*ldf + t] : 0   (xc = hcat[.., f])
// d x of the convolution (the input gradient of mtadgat_backward_input): dx[b, t, i] = sum_{o, j} w[o, i, j] dpre[b, t - j + pad, o]
int launch_conv_dx(const float* dpre, long ldp, const float* w, long B, int T, int F, int taps, int pad, float* dx, hipStream_t s);
int launch_dxc(const float* hcat, const float* dhcat, long ldh, const float* dvt, long ldt, const float* dvf, long ldf,
               long B, int T, int F, float* dpre, long ldp, hipStream_t s);
// keep-mask (1 / 0) of a dropout stream: mask[w*n + idx] for windows win0 + w (test hook)
int launch_dropmask(const DropArgs& d, unsigned stream, long nwin, long n_per_win, float* mask, hipStream_t s);
// nn.GRU's dropout between stacked layers on a (nwin * T, ld) state sequence (columns >= H are written as zero); also its adjoint
int launch_seq_dropout(const float* src, float* dst, long nwin, int T, int H, int ld, const DropArgs& d, unsigned stream, hipStream_t s);
int launch_copy2d(const float* src, long lds, float* dst, long ldd, long R, int ncols, hipStream_t s);
int launch_conv_scatter(const float* cf, const float* el, const float* er, float* hcat, long n, int W, int F, int Fp, int Dp, int pad,
                        hipStream_t s);
int launch_transpose_win(const float* src, long lds, float* dst, long ldd, long B, int R, int C, hipStream_t s);

}  // namespace mtadgat
