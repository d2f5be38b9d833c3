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
    const f32x4* Wp;     // packed (F x taps*Fp), NT tiles, Q = taps*Fp/8
    const float* bias;   // NT*32
    int NT;
    float* XC;           // (B*W, Fp)   or null
    float* XCT;          // (B*F, Wp)   or null
    int Wpad;            // row stride of XCT
    float* HCAT;         // (B*W, Dp) columns [0,F)   or null
    int Dp;
    float* Y;            // (B*W, F) plain output   or null
};

struct AttendArgs {
    const float* LC;     // (B*K, ldl) per query node: [L'(PT) | c | pad]
    const float* RT;     // (B, rt_rows, Kp) per window, key-node-minor: rows [0,PT) = R', row PT = d
    int ldl, rt_rows, Kp, PT, P8;
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

// fused per-window graph-attention layer (projection + scores + softmax + aggregation in one workgroup)
struct GatArgs {
    const float* V;      // vt == 0: (B*K, ldv) node rows; vt == 1: (B*D, ldv) rows whose columns are the nodes
    int ldv, D, K;
    int vt;
    int vld;             // LDS row stride of the staged V rows
    int lr_floats;       // LDS floats reserved for L' / R' (and the aliased softmax rows) ahead of the V rows
    const f32x4* Wp;     // packed projection tiles [2*NT_L][Q][64]: query-side tiles then key-side tiles
    const float* pbias;  // projection bias, 2*NT_L*32
    int NT_L, Q, PT, P8;
    const float* bias;   // (K, K) attention bias or null
    float* out;          // out[win*so_w + i*so_i + d*so_d]
    long so_w, so_i, so_d;
    long nwin;
    int v1;
    float alpha;
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
int launch_conv(const ConvArgs& a, hipStream_t s);
void attend_plan(int K, int* rows_per_blk, int* nblk, int* IB);
int launch_attend(const AttendArgs& a, int IB, hipStream_t s);
int launch_gat(const GatArgs& a, int IBL, int JPL, int rj, int nw, size_t lds_bytes, hipStream_t s);
int launch_gru(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s);
int launch_copy2d(const float* src, long lds, float* dst, long ldd, long R, int ncols, hipStream_t s);
int launch_transpose_win(const float* src, long lds, float* dst, long ldd, long B, int R, int C, hipStream_t s);

}  // namespace mtadgat
