// Re-packing of the weight image on the device (mtadgat_update_weights_device): after an optimizer step the
// parameters are already in HBM, and the 1.9 MB tile image is rebuilt from them by a handful of small kernels instead
// of a device -> host copy, a host-side pack and an upload (reference training.py:127: optimizer.step() between two
// forwards).  Same arithmetic as the host packer (mtadgat_pack.cpp), in the same order, in double where it is.
#include "mtadgat_device.h"

namespace mtadgat {

// every position of the image that is a plain copy of a parameter
// (four entries per thread: one 16-byte index load, then the four parameter loads together)
__global__ void k_pack_gather(const float* __restrict__ flat, const int* __restrict__ gidx, float* __restrict__ img, long n) {
    const long i = 4 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    if (i + 4 <= n) {
        const int4 g = *reinterpret_cast<const int4*>(gidx + i);
        const float v0 = flat[g.x < 0 ? 0 : g.x], v1 = flat[g.y < 0 ? 0 : g.y], v2 = flat[g.z < 0 ? 0 : g.z], v3 = flat[g.w < 0 ? 0 : g.w];
        if (g.x >= 0) img[i] = v0;
        if (g.y >= 0) img[i + 1] = v1;
        if (g.z >= 0) img[i + 2] = v2;
        if (g.w >= 0) img[i + 3] = v3;
        return;
    }
    for (long j = i; j < n; ++j) {
        const int g = gidx[j];
        if (g >= 0) img[j] = flat[g];
    }
}

// sum_e [hl *] x[e xs] * y[e ys] in double, added in the order of e (the host packer's sum): the loads of eight terms are issued
// together -- one load pair per dependent add made a 200-term row 19 us of load latency
template <bool HL>
__device__ __forceinline__ double dot_in_order(const float* __restrict__ x, long xs, const float* __restrict__ y, long ys, int E, double hl) {
#pragma clang fp contract(off)
    double acc = 0.0;
    constexpr int NB = 8;
    for (int e0 = 0; e0 < E; e0 += NB) {
        float xv[NB], yv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = min(e0 + u, E - 1);
            xv[u] = x[(long)e * xs];
            yv[u] = y[(long)e * ys];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (e0 + u < E) {
                if (HL) acc += hl * (double)xv[u] * (double)yv[u];
                else acc += (double)xv[u] * (double)yv[u];
            }
    }
    return acc;
}
// row n (of the 2 * KS projected columns), input feature k (k == D: the bias entry) of a graph-attention layer's
// folded projection (pack_gat): GATv2 columns scaled by |a'_k| in sign-sorted order, column PT = the linear part
// summed over the embedding; GAT v1: the two rank-1 columns
__device__ float gat_row_value(const PackGatArgs& a, int n, int k) {
#pragma clang fp contract(off)          // the host packer's products and sums, not fused ones
    const int side = n >= a.KS ? 1 : 0, nn = side ? n - a.KS : n;
    const int D = a.D, E = a.E;
    const float* __restrict__ lw = a.flat + a.lin_w;
    const float* __restrict__ lb = a.flat + a.lin_b;
    const float* __restrict__ av = a.flat + a.a;
    if (a.v2) {
        const int lin_in = 2 * D;
        const int PT = a.ord[1];
        if (nn < PT) {
            const int kk = a.colk[nn];
            if (kk < 0) return 0.f;
            const double s = fabs((1.0 - a.alpha) * 0.5 * (double)av[kk]);
            if (k < D) return (float)(s * (double)lw[(long)kk * lin_in + side * D + k]);
            return side == 0 ? (float)(s * (double)lb[kk]) : 0.f;
        }
        if (nn == PT) {
            const double hl = (1.0 + a.alpha) * 0.5;
            double acc = 0.0;
            if (k < D) acc = dot_in_order<true>(av, 1, lw + side * D + k, lin_in, E, hl);
            else if (side == 0) acc = dot_in_order<true>(av, 1, lb, 1, E, hl);
            return (float)acc;
        }
        return 0.f;
    }
    if (nn != 0) return 0.f;
    if (k < D) return (float)dot_in_order<false>(av + side * E, 1, lw + k, D, E, 0.0);
    return (float)dot_in_order<false>(av + side * E, 1, lb, 1, E, 0.0);
}

// Column order of a GATv2 layer's folded projection (gat_column_order, mtadgat_pack.cpp, on the device): embedding columns with
// a'_k = (1 - alpha) / 2 a_k >= 0 first, in their own order, padded to a multiple of 8; then the negative ones, likewise.
// colk[n] = embedding column of sorted column n or -1; ord = [P8, PT, number of non-negative columns].  One workgroup of 256
// threads: a counting pass, then a placement pass whose ranks come from wave ballots + the per-wave counts in LDS (both groups keep
// ascending embedding order, as the host packer writes them).
__global__ void __launch_bounds__(256) k_gat_colorder(const float* __restrict__ av, int E, double alpha, int* __restrict__ colk, int ncolk, int* __restrict__ ord) {
    __shared__ int s_cnt[4];
    __shared__ int s_np;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int n = tid; n < ncolk; n += 256) colk[n] = -1;
    int np_local = 0;
    for (int k = tid; k < E; k += 256) np_local += ((1.0 - alpha) * 0.5 * (double)av[k]) >= 0.0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) np_local += __shfl_xor(np_local, o);
    if (lane == 0) s_cnt[wv] = np_local;
    __syncthreads();
    if (tid == 0) s_np = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
    const int np = s_np, P8 = (np + 7) / 8 * 8, N8 = (E - np + 7) / 8 * 8;
    int basep = 0, basen = 0;                           // columns of each group placed by the chunks before this one
    for (int k0 = 0; k0 < E; k0 += 256) {
        const int k = k0 + tid;
        const bool in = k < E;
        const bool pos = in && ((1.0 - alpha) * 0.5 * (double)av[k]) >= 0.0;
        const unsigned long long bp = __ballot(pos), bi = __ballot(in);
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        __syncthreads();                                // the previous chunk's counts have been read
        if (lane == 0) s_cnt[wv] = __popcll(bp);
        __syncthreads();
        int wp = 0, tp = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = s_cnt[w]; if (w < wv) wp += c; tp += c; }
        const int nin = min(256, E - k0);
        const int rp = wp + __popcll(bp & below);                                   // non-negative columns of this chunk before k
        const int rn = (64 * wv - wp) + __popcll((bi & ~bp) & below);               // negative ones (every wave before this one is full)
        if (in) {
            if (pos) colk[basep + rp] = k;
            else colk[P8 + basen + rn] = k;
        }
        basep += tp; basen += nin - tp;
    }
    if (tid == 0) { ord[0] = P8; ord[1] = P8 + N8; ord[2] = np; ord[3] = 0; }
}

__global__ void k_pack_gat(const PackGatArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n_code) {
        const int c = a.code[i];
        float v = 0.f;
        if (c > 0) {
            const int n = (c - 1) / (a.D + 1), k = (c - 1) - n * (a.D + 1);
            if (k < a.D || a.fused) v = gat_row_value(a, n, k);      // the bias rides as weight row D in the fused kernel only
        }
        a.w_out[i] = v;
    }
    if (i < a.n_bias) a.b_out[i] = i < 2 * a.KS ? gat_row_value(a, i, a.D) : 0.f;
}

// [b_ir + b_hr | b_iz + b_hz | b_in | b_hn] rows of Hp (and their copy as the bias of the hoisted input projection)
__global__ void k_pack_gru_bias(const float* __restrict__ flat, long bih, long bhh, int H, int Hp, float* __restrict__ b, float* __restrict__ bx) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Hp) return;
    const bool ok = j < H;
    const float* __restrict__ bi = flat + bih;
    const float* __restrict__ bh = flat + bhh;
    const float v0 = ok ? bi[j] + bh[j] : 0.f, v1 = ok ? bi[H + j] + bh[H + j] : 0.f, v2 = ok ? bi[2 * H + j] : 0.f, v3 = ok ? bh[2 * H + j] : 0.f;
    b[j] = v0; b[Hp + j] = v1; b[2 * Hp + j] = v2; b[3 * Hp + j] = v3;
    if (bx) { bx[j] = v0; bx[Hp + j] = v1; bx[2 * Hp + j] = v2; }
}

// decoder input x_t[j] = h_end[(t Hin + j) / T] (reference modules.py:279): the columns of W_ih that share an h_end
// entry summed per step (double), into the per-step tiles of k_gru's folded input and the plain array of k_gru16
// Round 6: the column sums come from per-row running sums in double (k_fold_prefix, one pass over W_ih per re-pack) -- every one of
// the T x (tile slots) outputs used to walk its own up-to-T columns (60 us per optimizer step at the reference's shapes, 3 % of a
// 256-window training step); a difference of two double prefix sums rounds to the same float as the direct double sum except on
// rounding ties (the device image was never bit-identical to the host packer's in this one pack: other summation order).
__device__ float fold_value(const double* __restrict__ prefix, int Hin, int T, int R, int t, int k) {
    const int lo = (int)(((long)t * Hin) / T);
    long j0 = (long)(lo + k) * T - (long)t * Hin, j1 = j0 + T;
    j0 = j0 < 0 ? 0 : j0;
    j1 = j1 > Hin ? Hin : j1;
    if (j1 <= j0) return 0.f;
    const double* __restrict__ pr = prefix + (long)R * (Hin + 1);
    return (float)(pr[j1] - pr[j0]);
}
// (a thread per row; staging 64 rows through LDS for coalesced loads measured 45 us against 17: the staging loop serialises)
__global__ void k_fold_prefix(const float* __restrict__ wih, int rows, int Hin, double* __restrict__ prefix) {
#pragma clang fp contract(off)
    const int R = blockIdx.x * blockDim.x + threadIdx.x;
    if (R >= rows) return;
    double acc = 0.0;
    double* __restrict__ pr = prefix + (long)R * (Hin + 1);
    pr[0] = 0.0;
    const float* __restrict__ wr = wih + (long)R * Hin;
    constexpr int NB = 16;
    for (int j0 = 0; j0 < Hin; j0 += NB) {              // NB loads in flight, added in order
        float v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = wr[min(j0 + u, Hin - 1)];
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (j0 + u < Hin) { acc += (double)v[u]; pr[j0 + u + 1] = acc; }
    }
}

__global__ void k_pack_fold(const PackFoldArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n_tiles = (long)a.T * a.tile_floats;
    if (i < n_tiles) {
        const int t = (int)(i / a.tile_floats);
        const int c = a.code[i - (long)t * a.tile_floats];
        float v = 0.f;
        if (c > 0) {
            const int R = (c - 1) / a.NMp, k = (c - 1) - R * a.NMp;
            v = fold_value(a.prefix, a.Hin, a.T, R, t, k);
        }
        a.tiles_out[i] = v;
        return;
    }
    const long j = i - n_tiles;
    if (a.fold_out && j < (long)a.T * 3 * a.Hp * 8) {
        const int k = (int)(j & 7);
        const long q = j >> 3;
        const int u = (int)(q % a.Hp);
        const long q2 = q / a.Hp;
        const int st = (int)(q2 % 3), t = (int)(q2 / 3);
        a.fold_out[j] = u < a.H ? fold_value(a.prefix, a.Hin, a.T, st * a.H + u, t, k) : 0.f;
    }
}

// fp32 tile packs -> split-bf16 packs (mtadgat_device.h, "x3"): bf16 chunk qd of a tile pairs the fp32 chunks 2 qd and
// 2 qd + 1 of the same lane (that IS the element order of the bf16 operand), each value becomes three bf16 pieces.
// src [outer][Qs][G][64] f32x4, dst [outer][Qd][G][3 pieces][64] 16-byte words
// largest |value| of a region, as float bits (non-negative floats order like unsigned integers)
__global__ void k_absmax(const float* __restrict__ src, long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
// sc = [bits of max | S | 1 / S | 0]: S = the power of two that puts the largest weight into [2^13, 2^14) (fp16: 2^16 - 32 max)
__global__ void k_scale_from_max(float* __restrict__ sc) {
    const float m = __uint_as_float(reinterpret_cast<const unsigned*>(sc)[0]);
    int e = 0;
    if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e);        // m = f 2^e, f in [0.5, 1)
    int sh = 14 - e;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    sc[1] = ldexpf(1.f, sh);
    sc[2] = ldexpf(1.f, -sh);
    sc[3] = 0.f;
}

// ... into two fp16 pieces of scale * value (mtadgat_device.h): dst [outer][Qd][G][2 pieces][64] words
__global__ void k_split2h(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n_outer, int Qs, int Qd, int G, const float* __restrict__ scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = n_outer * Qd * G * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int g = (int)(r % G);
    const long r2 = r / G;
    const int qd = (int)(r2 % Qd);
    const long o = r2 / Qd;
    const float S = scale[0];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a = 2 * qd < Qs ? src[((o * Qs + 2 * qd) * G + g) * 64 + lane] : z;
    f32x4 b = 2 * qd + 1 < Qs ? src[((o * Qs + 2 * qd + 1) * G + g) * 64 + lane] : z;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] *= S; b[e] *= S; }
    f32x4 hi, lo;
    split2h(a, b, hi, lo);
    f32x4* __restrict__ d = dst + (((o * Qd + qd) * G + g) * 2) * 64 + lane;
    d[0] = hi; d[64] = lo;
}

// k_gath's two-fp16-piece pack of a GATv2 projection in the COMPACT column order (round 6).  The fp32 pack keeps the sign-sorted
// columns with each group padded to 8 -- [non-negative a' (npos) | pad to P8 | negative | pad to PT | c / d at PT] -- because the tile
// loops of k_gat / k_gat_wide / k_attend walk whole 8-column tiles of one sign.  k_gath's pair grid takes the sign per 2-column STEP,
// so its pack pads the non-negative group to 2 only: [npos | pad to P2 | negative | pad to PT2 | c / d at PT2], P2 = round_up(npos, 2),
// PT2 = round_up(P2 + nneg, 8) -- one 8-column tile less whenever the two paddings add up to 8 or more (E = 110: 120 -> 112 columns,
// 15 -> 14 tiles of the temporal pair grid at the flagship shape).  ord = [P8, PT, npos] as k_gat_colorder / the host packer wrote it.
// src: fp32 tiles [2 NT_L][Qs][64] (query-side tiles then key-side tiles), dst: [2 NT_L][Qd][2 pieces][64].
__device__ __forceinline__ int gath_compact_src(int c, int npos, int nneg, int P8, int PT) {
    const int P2 = (npos + 1) & ~1, PT2 = (P2 + nneg + 7) & ~7;
    if (c < npos) return c;
    if (c < P2) return -1;
    if (c < P2 + nneg) return P8 + (c - P2);
    if (c == PT2) return PT;
    return -1;
}
__global__ void k_split2h_gath(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int NT_L, int Qs, int Qd, const int* __restrict__ ord, int E,
                               const float* __restrict__ scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)2 * NT_L * Qd * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int qd = (int)(r % Qd);
    const int o = (int)(r / Qd);                        // destination tile
    const int side = o >= NT_L ? 1 : 0, n = o - side * NT_L;
    const int i = lane & 31;
    const int P8 = ord[0], PT = ord[1], npos = ord[2];
    const int cs = gath_compact_src(32 * n + i, npos, E - npos, P8, PT);
    const float S = scale[0];
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (cs >= 0) {
        const long so = (long)side * NT_L + (cs >> 5);
        const int sl = (lane & 32) | (cs & 31);
        if (2 * qd < Qs) a = src[(so * Qs + 2 * qd) * 64 + sl];
        if (2 * qd + 1 < Qs) b = src[(so * Qs + 2 * qd + 1) * 64 + sl];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] *= S; b[e] *= S; }
    f32x4 hi, lo;
    split2h(a, b, hi, lo);
    f32x4* __restrict__ d = dst + (((long)o * Qd + qd) * 2) * 64 + lane;
    d[0] = hi; d[64] = lo;
}

// input-part weights of a recurrent layer, three gates per chunk: chunks below qb as three bf16 pieces (9 words), the
// others as two fp16 pieces (6 words), all of S * W
__global__ void k_split_x(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n_outer, int Qs, int Qd, int qb, const float* __restrict__ scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = n_outer * Qd * 3 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int g = (int)(r % 3);
    const long r2 = r / 3;
    const int qd = (int)(r2 % Qd);
    const long o = r2 / Qd;
    const float S = scale[0];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a = 2 * qd < Qs ? src[((o * Qs + 2 * qd) * 3 + g) * 64 + lane] : z;
    f32x4 b = 2 * qd + 1 < Qs ? src[((o * Qs + 2 * qd + 1) * 3 + g) * 64 + lane] : z;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] *= S; b[e] *= S; }
    const long wtot = (long)qb * 9 + (long)(Qd - qb) * 6;               // words per outer block
    if (qd < qb) {
        f32x4 hi, mid, lo;
        split3(a, b, hi, mid, lo);
        f32x4* __restrict__ d = dst + (o * wtot + (long)qd * 9 + g * 3) * 64 + lane;
        d[0] = hi; d[64] = mid; d[128] = lo;
    } else {
        f32x4 hi, lo;
        split2h(a, b, hi, lo);
        f32x4* __restrict__ d = dst + (o * wtot + (long)qb * 9 + (long)(qd - qb) * 6 + g * 2) * 64 + lane;
        d[0] = hi; d[64] = lo;
    }
}
int launch_split_x(const float* src, float* dst, long n_outer, int Qs, int Qd, int qb, const float* scale, hipStream_t s) {
    const long total = n_outer * Qd * 3 * 64;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_split_x, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), n_outer, Qs, Qd, qb, scale);
    LAUNCH_CHECK();
    return 0;
}

__global__ void k_split3(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n_outer, int Qs, int Qd, int G, const float* __restrict__ scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = n_outer * Qd * G * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int g = (int)(r % G);
    const long r2 = r / G;
    const int qd = (int)(r2 % Qd);
    const long o = r2 / Qd;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 a = 2 * qd < Qs ? src[((o * Qs + 2 * qd) * G + g) * 64 + lane] : z;
    f32x4 b = 2 * qd + 1 < Qs ? src[((o * Qs + 2 * qd + 1) * G + g) * 64 + lane] : z;
    if (scale) {                               // a power of two: the pieces of S x are S times the pieces of x
        const float S = scale[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] *= S; b[e] *= S; }
    }
    f32x4 hi, mid, lo;
    split3(a, b, hi, mid, lo);
    f32x4* __restrict__ d = dst + (((o * Qd + qd) * G + g) * 3) * 64 + lane;
    d[0] = hi; d[64] = mid; d[128] = lo;
}
int launch_split3(const float* src, float* dst, long n_outer, int Qs, int Qd, int G, const float* scale, hipStream_t s) {
    const long total = n_outer * Qd * G * 64;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_split3, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), n_outer, Qs, Qd, G, scale);
    LAUNCH_CHECK();
    return 0;
}
int launch_split2h(const float* src, float* dst, long n_outer, int Qs, int Qd, int G, const float* scale, hipStream_t s) {
    const long total = n_outer * Qd * G * 64;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_split2h, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), n_outer, Qs, Qd, G, scale);
    LAUNCH_CHECK();
    return 0;
}
int launch_split2h_gath(const float* src, float* dst, int NT_L, int Qs, int Qd, const int* ord, int E, const float* scale, hipStream_t s) {
    const long total = (long)2 * NT_L * Qd * 64;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_split2h_gath, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), NT_L, Qs, Qd, ord, E, scale);
    LAUNCH_CHECK();
    return 0;
}
// sc[0] (as bits) = max |value| over the regions handed to launch_absmax since it was cleared; then sc[1] = S, sc[2] = 1 / S
int launch_absmax(const float* src, long n, float* sc, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_absmax, dim3(64), dim3(256), 0, s, src, n, reinterpret_cast<unsigned*>(sc));
    LAUNCH_CHECK();
    return 0;
}
int launch_scale_from_max(float* sc, hipStream_t s) {
    hipLaunchKernelGGL(k_scale_from_max, dim3(1), dim3(1), 0, s, sc);
    LAUNCH_CHECK();
    return 0;
}

// bit-exact checksum of a list of parameter tensors: sum of (32-bit pattern x an odd multiplier derived from the
// global element index) mod 2^64 -- any change of any single element changes it
__global__ void k_fingerprint(const FingerprintArgs a, unsigned long long* __restrict__ out) {
    const int t = blockIdx.y;
    const unsigned* __restrict__ p = reinterpret_cast<const unsigned*>(a.ptr[t]);
    const long n = a.count[t], base = a.base[t];
    unsigned long long acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long g = (unsigned long long)(base + i);
        const unsigned long long mult = (((unsigned long long)mix32((unsigned)g * 2654435761u ^ 0x5EEDu) << 20) ^ (g * 0x9E3779B97F4A7C15ull)) | 1ull;
        acc += (unsigned long long)p[i] * mult;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

int launch_fingerprint(const FingerprintArgs& a, int n_tensors, unsigned long long* out, hipStream_t s) {
    if (n_tensors <= 0) return 0;
    hipLaunchKernelGGL(k_fingerprint, dim3(16, (unsigned)n_tensors), dim3(256), 0, s, a, out);
    LAUNCH_CHECK();
    return 0;
}

int launch_pack_gather(const float* flat, const int* gidx, float* img, long n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_pack_gather, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, flat, gidx, img, n);
    LAUNCH_CHECK();
    return 0;
}
int launch_gat_colorder(const float* a_dev, int E, double alpha, int* colk_dev, int ncolk, int* ord_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_gat_colorder, dim3(1), dim3(256), 0, s, a_dev, E, alpha, colk_dev, ncolk, ord_dev);
    LAUNCH_CHECK();
    return 0;
}
int launch_pack_gat(const PackGatArgs& a, hipStream_t s) {
    const int n = a.n_code > a.n_bias ? a.n_code : a.n_bias;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_pack_gat, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}
int launch_pack_gru_bias(const float* flat, long bih, long bhh, int H, int Hp, float* b, float* bx, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_gru_bias, dim3((unsigned)((Hp + 63) / 64)), dim3(64), 0, s, flat, bih, bhh, H, Hp, b, bx);
    LAUNCH_CHECK();
    return 0;
}
int launch_pack_fold(const PackFoldArgs& a, hipStream_t s) {
    const long n = (long)a.T * a.tile_floats + (a.fold_out ? (long)a.T * 3 * a.Hp * 8 : 0);
    if (n <= 0) return 0;
    if (!a.prefix) return -2;
    hipLaunchKernelGGL(k_fold_prefix, dim3((unsigned)((3 * a.H + 63) / 64)), dim3(64), 0, s, a.flat + a.wih, 3 * a.H, a.Hin, a.prefix);
    hipLaunchKernelGGL(k_pack_fold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
