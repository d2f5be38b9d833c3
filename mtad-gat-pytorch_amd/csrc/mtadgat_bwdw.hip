// Backward of a WIDE graph-attention layer (more than 128 nodes or node dimensions: the fused per-window kernels of
// mtadgat_bwd.hip keep a whole window in LDS and stop there).  Reference: FeatureAttentionLayer.forward modules.py:65-95,
// TemporalAttentionLayer.forward modules.py:166-193 under loss.backward(), training.py:126 -- the reference trains any shape.
//
// Same outputs as the fused layers' k_gat_bwd_att (+ the score backward, which since round 6 is this file's k_bw_pair for every
// GATv2 layer), so everything downstream is shared (mtadgat_capi.cpp: the data-gradient row
// GEMM d V += [dL | dR] [W_l ; W_r], the weight-gradient GEMM, the batch sums of d e and d a):
//     DE  (B, K, K)        d e_ij, the gradient of the attention scores (= of the layer's bias before the sum over windows)
//     DV  (B*K, lddv)      the aggregation path's d V
//     DLR (B*K, 2 Ep)      [d L | d R], the gradients of the un-scaled projections L = V W_l^T + b, R = V W_r^T
//     DAp (B, Ep)          per-window partial sums of d a
// but every matrix goes through HBM and the kernels are generic in K and D (<= 512), not tuned per shape:
//     k_bw_ds        d S = d H . H (1 - H)                                      (h = sigmoid(S), modules.py:93 / :191)
//     k_bgemm        d V = att'^T d S   (att' = dropout(att), applied while the operand is staged)        per window
//     k_bgemm        d att' = d S V^T                                                                     per window
//     k_bw_softmax   d e = att . (d att - sum_j att d att), d att = mask . d att'                         in place, a wave per row
//     k_rowgemm      [L | R] = V [W_l ; W_r]^T + [b | 0]      (mtadgat_kernels.hip, the un-scaled pack of the backward plan)
//     k_bw_pair      GATv2 scores e_ij = sum_k a_k LeakyReLU(L_ik + R_jk) (modules.py:74-77 / :174-177):
//                        d z_ijk = d e_ij a_k [u > 0 ? 1 : alpha],  d L_ik = sum_j d z,  d R_jk = sum_i d z,
//                        d a_k = sum_ij d e_ij LeakyReLU(u)
//                    in ONE register-blocked pass over a 32-column block of the embedding (round 6), no atomics.
// GAT (v1) layers of this size (round 6): the score backward e_ij = LeakyReLU(c_i + d_j) is linear in the node vectors below d s
// (mtadgat_bwd.hip: k_gat_v1_prep / k_gat_bwd_v1 / k_gat_v1_finish); k_bw_v1 is k_gat_bwd_v1 with the node rows read from memory
// instead of an LDS copy of the window -- same outputs (d V +=, per-window partials [p1 | p2 | sc sd]), so prep and finish are shared.
#include "mtadgat_device.h"

namespace mtadgat {

namespace {

// ---- d S[(win K + i) ldS + d] = d H . H . (1 - H),  H / d H at [win so_w + i so_i + d so_d]
__global__ __launch_bounds__(256) void k_bw_ds(const float* __restrict__ H, const float* __restrict__ dH, long so_w, long so_i, long so_d,
                                               long nwin, int K, int D, float* __restrict__ dS, int ldS) {
    const long n = nwin * K * (long)D;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const long win = idx / ((long)K * D);
        const int rem = (int)(idx - win * (long)K * D);
        int i, d;
        if (so_d == 1) { i = rem / D; d = rem - i * D; }      // the fastest thread index follows the unit stride of H
        else { d = rem / K; i = rem - d * K; }
        const long o = win * so_w + i * so_i + d * so_d;
        const float h = H[o];
        dS[(win * K + i) * (long)ldS + d] = dH[o] * h * (1.f - h);
    }
}

// ---- batched GEMM on the fp32 matrix unit: C_b (M x N) = A_b (M x Kc) B_b (Kc x N), arbitrary element strides for A and B
//   A_b(m, k) = A[b sAb + m sAm + k sAk]   (optionally times the dropout keep factor of attention element (k, m): A = att^T)
//   B_b(k, n) = B[b sBb + k sBk + n sBn]
//   C_b(m, n) = C[b sCb + m ldc + n]
// One workgroup = a 64 x 64 block of C for one b; wave (wm, wn) owns a 32 x 32 quadrant on v_mfma_f32_32x32x2_f32; the
// operands go through LDS 16 columns of the contraction at a time (clamped unconditional loads, zero where out of range).
struct BGemmArgs {
    const float* A; long sAb, sAm, sAk;
    const float* B; long sBb, sBk, sBn;
    float* C; long sCb, ldc;
    int M, N, Kc;
    long nb;
    int adrop;                 // 1: A_b(m, k) *= keep(k * dropK + m) ? keep_scale : 0
    int dropK;
    DropArgs drop;
    unsigned drop_stream;
};

__global__ __launch_bounds__(256) void k_bgemm(const BGemmArgs a) {
    constexpr int KS = 16, AP = KS + 1, BP = 64 + 1;
    __shared__ float As[64 * AP];
    __shared__ float Bs[KS * BP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, g = lane >> 5;
    const int mt = (a.M + 63) >> 6, nt = (a.N + 63) >> 6;
    const long blk = blockIdx.x;
    const long b = blk / (mt * nt);
    const int t2 = (int)(blk - b * (mt * nt));
    const int m0 = (t2 / nt) * 64, n0 = (t2 % nt) * 64;
    const float* __restrict__ Ab = a.A + b * a.sAb;
    const float* __restrict__ Bb = a.B + b * a.sBb;
    unsigned key = 0;
    if (a.adrop) key = drop_window_key(a.drop, a.drop_stream, b);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // staging maps: the thread index runs along whichever operand index has the smaller stride (coalesced either way)
    const bool a_k_fast = a.sAk <= a.sAm, b_n_fast = a.sBn <= a.sBk;
    for (int k0 = 0; k0 < a.Kc; k0 += KS) {
        float av[4], bv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int u = tid + n * 256;                   // 1024 elements of each tile
            const int am = a_k_fast ? u / KS : u % 64, ak = a_k_fast ? u % KS : u / 64;
            const int gm = m0 + am, gk = k0 + ak;
            const int cm = gm < a.M ? gm : a.M - 1, ck = gk < a.Kc ? gk : a.Kc - 1;
            float v = Ab[cm * a.sAm + ck * a.sAk];
            if (a.adrop && a.drop.thresh) v *= drop_keep(key, (unsigned)(ck * a.dropK + cm), a.drop.thresh) ? a.drop.keep_scale : 0.f;
            av[n] = (gm < a.M && gk < a.Kc) ? v : 0.f;
            const int bk = b_n_fast ? u / 64 : u % KS, bn = b_n_fast ? u % 64 : u / KS;
            const int hk = k0 + bk, hn = n0 + bn;
            const int dk = hk < a.Kc ? hk : a.Kc - 1, dn = hn < a.N ? hn : a.N - 1;
            const float w = Bb[dk * a.sBk + dn * a.sBn];
            bv[n] = (hk < a.Kc && hn < a.N) ? w : 0.f;
        }
        __syncthreads();                                   // the previous tiles are consumed
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int u = tid + n * 256;
            const int am = a_k_fast ? u / KS : u % 64, ak = a_k_fast ? u % KS : u / 64;
            As[am * AP + ak] = av[n];
            const int bk = b_n_fast ? u / 64 : u % KS, bn = b_n_fast ? u % 64 : u / KS;
            Bs[bk * BP + bn] = bv[n];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KS; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(wm * 32 + i) * AP + kk + g], Bs[(kk + g) * BP + wn * 32 + i], acc, 0, 0, 0);
    }
    // accumulator register 4 q + s of lane (i, g) = C[row 8 q + 4 g + s][column i] of the quadrant
    float* __restrict__ Cb = a.C + b * a.sCb;
    const int col = n0 + wn * 32 + i;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int row = m0 + wm * 32 + 8 * q + 4 * g + s4;
            if (row < a.M && col < a.N) Cb[row * a.ldc + col] = acc[4 * q + s4];
        }
}

// ---- softmax backward, in place: DE holds d att' (the gradient of the dropped attention matrix) on entry, d e on exit.
// One wave per row (K <= 512: eight elements per lane).
__global__ __launch_bounds__(256) void k_bw_softmax(const float* __restrict__ ATT, float* __restrict__ DE, long nwin, int K, DropArgs drop,
                                                    unsigned drop_stream) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nwin * K) return;
    const long win = row / K;
    const int i = (int)(row - win * K);
    const unsigned key = drop_window_key(drop, drop_stream, win);
    const float* __restrict__ ap = ATT + row * K;
    float* __restrict__ dp = DE + row * K;
    float av[8], dv[8];
    float csum = 0.f;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int j = lane + 64 * n;
        const int jc = j < K ? j : K - 1;
        float sc = 1.f;
        if (drop.thresh) sc = drop_keep(key, (unsigned)(i * K + jc), drop.thresh) ? drop.keep_scale : 0.f;
        av[n] = j < K ? ap[jc] : 0.f;
        dv[n] = j < K ? dp[jc] * sc : 0.f;
        csum += av[n] * dv[n];
    }
    csum = wave_sum(csum);
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int j = lane + 64 * n;
        if (j < K) dp[j] = av[n] * (dv[n] - csum);
    }
}

// ---- GATv2 score backward for one window and one 32-column block of the embedding (round 6: one pass, register-blocked).
// u_ijk = L_ik + R_jk,  s_ijk = u > 0 ? 1 : alpha,  d L_ik = a_k sum_j d e_ij s_ijk,  d R_jk = a_k sum_i d e_ij s_ijk, and -- since
// LeakyReLU(u) = s u is linear in (L, R) once s is fixed -- d a_k = sum_i L_ik dl_ik + sum_j R_jk dr_jk with the un-scaled sums dl, dr:
// the pair loop is FIVE vector instructions per (i, j, k) for all three outputs (add, compare, select, two fma).
// 32 G threads (G = 16 key groups up to 256 keys, 32 above): lane & 31 = column k, the half-waves are the key groups: group q keeps
// R_jk and dr_jk of its JB = ceil(K / G) <= 16 keys in registers across all rows (no transposed copy of d e, no second pass; <= 125
// registers: four waves per SIMD).  Rows go in batches of eight: d e rows of the batch in LDS
// (read as 16-byte broadcasts), the groups' partial dl through LDS, summed by 256 of the threads, which also take the L dl term of d a.
// (The first version -- thread = (column, an eighth of the rows), one row at a time, three LDS reads and six instructions per pair and
// pass, two passes -- took 12.2 ms per launch at config 4's shapes: 48 % of that shape's training step.)
struct BwPairArgs {
    const float* LR;     // (B*K, ldlr): [L (Ep) | R (Ep)]
    int ldlr, Ep;
    const float* avec;   // (Ep) a, zero padded
    const float* DE;     // (B, K, K)
    int K;
    float alpha;
    float* DLR;          // (B*K, ldlr): [d L | d R]
    float* DAp;          // (B, Ep)
    long nwin;
};

template <int JB, int G>
__global__ __launch_bounds__(32 * G) void k_bw_pair(const BwPairArgs a) {
    constexpr int QN = (JB + 3) / 4, JBP = 4 * QN;                    // keys per group; slots of a group in a staged d e row (whole 16-byte words)
    constexpr int KP = G * JBP, IB = 8, NT = 32 * G;
    __shared__ __attribute__((aligned(16))) float des[IB * KP];      // d e rows of the batch, zero beyond K
    __shared__ float red[G * IB * 32];                                // partial dl of the groups
    __shared__ float Lb[IB * 32];                                     // L rows of the batch (this block's 32 columns)
    const int K = a.K, nkb = a.Ep >> 5;
    const long win = blockIdx.x / nkb;
    const int kb = (int)(blockIdx.x - win * nkb);
    const int tid = threadIdx.x, k = tid & 31, q = tid >> 5;
    const int j0 = q * JB;                                            // this group's keys: [j0, j0 + JB)
    const float* __restrict__ lr = a.LR + (win * K) * (long)a.ldlr + 32 * kb + k;
    const float* __restrict__ de = a.DE + win * (long)K * K;
    float* __restrict__ dlr = a.DLR + (win * K) * (long)a.ldlr + 32 * kb + k;
    const float alpha = a.alpha;
    float R[JB], dr[JB];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
        const int j = j0 + jj;
        const float v = lr[(long)(j < K ? j : K - 1) * a.ldlr + a.Ep];
        R[jj] = j < K ? v : 0.f;
        dr[jj] = 0.f;
    }
    float da = 0.f;
    for (int i0 = 0; i0 < K; i0 += IB) {
        __syncthreads();                                   // the previous batch's readers of des / red are done
        // the batch's rows of d e (coalesced; rows and keys past K as zeros) and this thread's L values
        {
            float v[IB * KP / NT];
#pragma unroll
            for (int n = 0; n < IB * KP / NT; ++n) {
                const int u = tid + n * NT;
                const int r = u / KP, sl = u - r * KP;
                const int j = (sl / JBP) * JB + (sl % JBP);            // slot -> key (positions JB .. JBP - 1 of a group: padding)
                const int ic = i0 + r < K ? i0 + r : K - 1, jc = j < K ? j : K - 1;
                v[n] = de[(long)ic * K + jc];
            }
#pragma unroll
            for (int n = 0; n < IB * KP / NT; ++n) {
                const int u = tid + n * NT;
                const int r = u / KP, sl = u - r * KP;
                const int j = (sl / JBP) * JB + (sl % JBP);
                des[u] = (i0 + r < K && sl % JBP < JB && j < K) ? v[n] : 0.f;
            }
        }
        if (tid < IB * 32) Lb[tid] = lr[(long)(i0 + (tid >> 5) < K ? i0 + (tid >> 5) : K - 1) * a.ldlr];      // (lr carries this thread's column k)
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < IB; ++r) {                     // (not unrolled: eight rows' d e quads at once spill the key registers)
            const float lv = Lb[r * 32 + k];
            float dl = 0.f;
#pragma unroll
            for (int qd = 0; qd < QN; ++qd) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(&des[r * KP + q * JBP + 4 * qd]);   // (one address per half-wave: broadcast)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * qd + e < JB) {
                        const float u = lv + R[4 * qd + e];
                        const float sl = u > 0.f ? 1.f : alpha;
                        dl = __builtin_fmaf(g4[e], sl, dl);
                        dr[4 * qd + e] = __builtin_fmaf(g4[e], sl, dr[4 * qd + e]);
                    }
            }
            red[(q * IB + r) * 32 + k] = dl;
        }
        __syncthreads();
        if (tid < IB * 32) {                               // thread (r = tid >> 5, k): the row's dl over all key groups
            const int r = tid >> 5;
            float dl = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) dl += red[(g2 * IB + r) * 32 + k];
            if (i0 + r < K) {
                da = __builtin_fmaf(Lb[tid], dl, da);
                dlr[(long)(i0 + r) * a.ldlr] = dl * a.avec[32 * kb + k];
            }
        }
    }
    const float ak = a.avec[32 * kb + k];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
        const int j = j0 + jj;
        if (j < K) {
            da = __builtin_fmaf(R[jj], dr[jj], da);
            dlr[(long)j * a.ldlr + a.Ep] = dr[jj] * ak;
        }
    }
    __syncthreads();
    red[q * 32 + k] = da;
    __syncthreads();
    if (tid < 32) {
        float sacc = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < G; ++g2) sacc += red[g2 * 32 + tid];
        a.DAp[win * a.Ep + 32 * kb + tid] = sacc;
    }
}

// ---- GAT (v1) score backward of a wide layer, one workgroup (256 threads) per window.  Vn: node rows (win K + node) ldv, D columns.
//   c_i = u1 . v_i + ub1, d_j = u2 . v_j + ub2;  d s_ij = d e_ij [c_i + d_j > 0 ? 1 : alpha];  dc_i = sum_j d s_ij, dd_j = sum_i d s_ij
//   d V[node] += dc u1 + dd u2;  part = [sum_i dc_i v_i | sum_j dd_j v_j | sum dc, sum dd]
// LDS: cq[K] | dk[K] | dc[K] | dd[K] (K <= 512)
__global__ __launch_bounds__(256) void k_bw_v1(const float* __restrict__ Vn, long ldv, int D, int K, const float* __restrict__ u,
                                               const float* __restrict__ DE, float alpha, float* __restrict__ DV, int lddv,
                                               float* __restrict__ part) {
    __shared__ float cq[512], dk[512], dc[512], dd[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long win = blockIdx.x;
    const float* __restrict__ vw = Vn + win * K * ldv;
    for (int x = tid; x < K; x += 256) { dc[x] = 0.f; dd[x] = 0.f; }
    // c / d of every node: a wave per node, lanes over the columns (coalesced row reads)
    for (int i = wave; i < K; i += 4) {
        float s1 = 0.f, s2 = 0.f;
        for (int col = lane; col < D; col += 64) { const float v = vw[(long)i * ldv + col]; s1 += u[col] * v; s2 += u[D + col] * v; }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { cq[i] = s1 + u[2 * D]; dk[i] = s2 + u[2 * D + 1]; }
    }
    __syncthreads();
    // d s: a wave per query row, lanes over the keys; row sums by a wave reduction, column sums per lane (keys lane + 64 h)
    const float* __restrict__ de = DE + win * (long)K * K;
    float colacc[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) colacc[h] = 0.f;
    for (int i = wave; i < K; i += 4) {
        const float ci = cq[i];
        float rs = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int j = lane + 64 * h;
            if (j < K) {
                const float ds = de[(long)i * K + j] * (ci + dk[j] > 0.f ? 1.f : alpha);
                rs += ds; colacc[h] += ds;
            }
        }
        rs = wave_sum(rs);
        if (lane == 0) dc[i] = rs;
    }
    // the four waves add their column sums one after the other (fixed order, no float atomics)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int h = 0; h < 8; ++h)
                if (lane + 64 * h < K) dd[lane + 64 * h] += colacc[h];
        }
        __syncthreads();
    }
    for (long x = tid; x < (long)K * D; x += 256) {
        const int node = (int)(x / D), col = (int)(x - (long)node * D);
        DV[(win * K + node) * (long)lddv + col] += dc[node] * u[col] + dd[node] * u[D + col];
    }
    float* __restrict__ po = part + win * (long)(2 * D + 2);
    for (int col = tid; col < D; col += 256) {
        float p1 = 0.f, p2 = 0.f;
        for (int i = 0; i < K; ++i) { const float v = vw[(long)i * ldv + col]; p1 += dc[i] * v; p2 += dd[i] * v; }
        po[col] = p1; po[D + col] = p2;
    }
    if (tid == 255) {
        float sc = 0.f, sd = 0.f;
        for (int i = 0; i < K; ++i) { sc += dc[i]; sd += dd[i]; }
        po[2 * D] = sc; po[2 * D + 1] = sd;
    }
}

}  // namespace

int launch_bw_ds(const float* H, const float* dH, long so_w, long so_i, long so_d, long nwin, int K, int D, float* dS, int ldS, hipStream_t s) {
    if (nwin <= 0) return 0;
    const long n = nwin * K * (long)D;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_bw_ds, dim3((unsigned)(blocks > 65535 * 8 ? 65535 * 8 : blocks)), dim3(256), 0, s, H, dH, so_w, so_i, so_d, nwin, K, D, dS, ldS);
    LAUNCH_CHECK();
    return 0;
}

// C_b = A_b B_b for b < nb (strides in elements; see BGemmArgs); adrop: the A operand is an attention matrix read transposed,
// A_b(m, k) = att_b[k][m], and gets the dropout keep factor of element (k, m) of window b
int launch_bgemm(const float* A, long sAb, long sAm, long sAk, const float* B, long sBb, long sBk, long sBn, float* C, long sCb, long ldc,
                 int M, int N, int Kc, long nb, const DropArgs* drop, unsigned drop_stream, int dropK, hipStream_t s) {
    if (nb <= 0 || M <= 0 || N <= 0) return 0;
    BGemmArgs a{};
    a.A = A; a.sAb = sAb; a.sAm = sAm; a.sAk = sAk; a.B = B; a.sBb = sBb; a.sBk = sBk; a.sBn = sBn; a.C = C; a.sCb = sCb; a.ldc = ldc;
    a.M = M; a.N = N; a.Kc = Kc; a.nb = nb;
    if (drop) { a.adrop = 1; a.drop = *drop; a.drop_stream = drop_stream; a.dropK = dropK; }
    const long blocks = nb * ((M + 63) / 64) * ((N + 63) / 64);
    if (blocks > 0x7fffffffL) return -2;
    hipLaunchKernelGGL(k_bgemm, dim3((unsigned)blocks), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

int launch_bw_softmax(const float* ATT, float* DE, long nwin, int K, const DropArgs& drop, unsigned drop_stream, hipStream_t s) {
    if (nwin <= 0) return 0;
    if (K > 512) return -2;
    const long rows = nwin * K;
    hipLaunchKernelGGL(k_bw_softmax, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ATT, DE, nwin, K, drop, drop_stream);
    LAUNCH_CHECK();
    return 0;
}

int launch_bw_pair(const float* LR, int ldlr, int Ep, const float* avec, const float* DE, int K, float alpha, float* DLR, float* DAp,
                   long nwin, hipStream_t s) {
    if (nwin <= 0) return 0;
    if ((Ep & 31) != 0 || K > 512 || K < 1) return -2;
    BwPairArgs a{};
    a.LR = LR; a.ldlr = ldlr; a.Ep = Ep; a.avec = avec; a.DE = DE; a.K = K; a.alpha = alpha; a.DLR = DLR; a.DAp = DAp; a.nwin = nwin;
    const long blocks = nwin * (Ep / 32);
    if (blocks > 0x7fffffffL) return -2;
    if (K <= 256) {                                        // 16 key groups x jb keys >= K
        switch ((K + 15) / 16) {
#define BWP_CASE(N) case N: hipLaunchKernelGGL((k_bw_pair<N, 16>), dim3((unsigned)blocks), dim3(512), 0, s, a); break;
            BWP_CASE(1) BWP_CASE(2) BWP_CASE(3) BWP_CASE(4) BWP_CASE(5) BWP_CASE(6) BWP_CASE(7) BWP_CASE(8)
            BWP_CASE(9) BWP_CASE(10) BWP_CASE(11) BWP_CASE(12) BWP_CASE(13) BWP_CASE(14) BWP_CASE(15) BWP_CASE(16)
#undef BWP_CASE
            default: return -2;
        }
    } else {                                               // 32 key groups
        switch ((K + 31) / 32) {
#define BWP_CASE(N) case N: hipLaunchKernelGGL((k_bw_pair<N, 32>), dim3((unsigned)blocks), dim3(1024), 0, s, a); break;
            BWP_CASE(9) BWP_CASE(10) BWP_CASE(11) BWP_CASE(12) BWP_CASE(13) BWP_CASE(14) BWP_CASE(15) BWP_CASE(16)
#undef BWP_CASE
            default: return -2;
        }
    }
    LAUNCH_CHECK();
    return 0;
}

int launch_bw_v1(const float* Vn, long ldv, int D, int K, const float* u, const float* DE, float alpha, float* DV, int lddv, float* part,
                 long nwin, hipStream_t s) {
    if (nwin <= 0) return 0;
    if (K > 512 || D < 1) return -2;
    hipLaunchKernelGGL(k_bw_v1, dim3((unsigned)nwin), dim3(256), 0, s, Vn, ldv, D, K, u, DE, alpha, DV, lddv, part);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
