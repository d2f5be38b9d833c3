// Backward kernels of the MTAD-GAT hot path (training step, reference training.py:106-127: loss.backward()
// through MTAD_GAT.forward) + launchers.  gfx950, wave64, fp32.
//
//   k_wgrad_lds / k_wgrad_reduce   every weight / bias gradient: dW = A^T B over all (window, step) rows, split over
//                              row slabs, deterministic two-stage sum, scattered into the reference's parameter layout
//   k_gru_bwd                  back-propagation through time of a GRU layer (GRULayer / RNNDecoder, modules.py:235-257)
//   k_gat_bwd_att / _pair      backward of a graph-attention layer (modules.py:65-95, :166-193), per window
//   small kernels              decoder-input adjoint (modules.py:279), conv pre-activation gradient, dropout masks
//
// The data gradients of the Linear layers (d X = d Y W) reuse k_rowgemm with transposed weight packs.
#include <cstdlib>
#include "mtadgat_device.h"

namespace mtadgat {

// ---------------------------------------------------------------------------
// wgrad: P[slab][m][n] = sum_{rows of the slab} A[row][m] * B[row][n], LDS-tiled: a workgroup of 4 waves owns a 128 x 128 block of d W (wave (wm, wn): 64 x 64 =
// 2 x 2 MFMA tiles) and walks its row slab 16 rows at a time.  The 16 x 128 pieces of A and B are staged in LDS
// by all 256 threads (coalesced dword loads, unconditional from clamped addresses, masked afterwards; the loader
// modes -- im2col for the convolution, previous-step rows, the all-ones bias column -- are applied here), double
// buffered so the loads of piece p + 1 fly during the MFMAs of piece p; every element of A and B is read from
// memory once per block row / column instead of once per wave.  MFMA operands: lane (c, kk) reads
// As[row 8 h + 4 kk + s][64 wm + 32 tm + c] -- 32 consecutive floats per half-wave, conflict free.
// ---------------------------------------------------------------------------
constexpr int WG_ROWS = 16;
// VEC: A and B rows are 16-byte aligned with row strides that are multiples of 4 floats (all internal buffers):
// the pieces are staged with one float4 load per 4 columns (few, wide requests) instead of dword loads.
// X3: the products from three bf16 pieces per operand on the 16-bit matrix pipe (six v_mfma_f32_32x32x16_bf16 per 16 data rows and
// tile instead of eight v_mfma_f32_32x32x2_f32: 2.7x less matrix time, fp32-class sums -- mtadgat_device.h); the 16 rows of a
// staging step are exactly one chunk, a lane gathers its eight rows of a column from the fp32 tile and splits them
template <int BMODE, bool VEC, bool X3 = false>
__global__ __launch_bounds__(256, (VEC || BMODE == 1) ? 3 : 2) void k_wgrad_lds(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][WG_ROWS][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][WG_ROWS][128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int c = lane & 31, kk = lane >> 5;
    const int Nb = (a.Np + 127) >> 7;
    // XCD-aware map: workgroups go to the 8 XCDs round-robin in linear id order; the output tiles of one row slab read the same
    // A / B rows, so they are given ids that land on ONE XCD (ids k, k + 8, ...): the slab then comes from HBM once and serves the
    // other tiles from that XCD's L2 (with the plain (tile, slab) grid and 8 tiles, tile x of every slab ran on XCD x: A was
    // fetched once per column block of B and B once per column block of A)
    unsigned tile_id = blockIdx.x, slab_id = blockIdx.y;
    if ((gridDim.y & 7u) == 0u && !a.plain_map) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned k = lin >> 3;
        slab_id = (k / gridDim.x) * 8u + (lin & 7u);
        tile_id = k % gridDim.x;
    }
    const int mb = (int)tile_id / Nb, nb = (int)tile_id - mb * Nb;
    const int slab = (int)slab_id;
    const long rbeg = (long)slab * a.rows_per_slab;
    const long rend = rbeg + a.rows_per_slab < a.R ? rbeg + a.rows_per_slab : a.R;
    const int T = a.T > 0 ? a.T : 1;

    // staging role: thread -> column (tid & 127) of rows (tid >> 7) + 2 e, e = 0..7
    const int scol = tid & 127, srow0 = tid >> 7;
    const int mcol = 128 * mb + scol, ncol = 128 * nb + scol;
    const int mc = mcol < a.M ? mcol : a.M - 1;
    const int nc = ncol < a.N ? ncol : a.N - 1;
    int tapn = 0, chn = 0;
    if (BMODE == 1) { tapn = nc / a.F; chn = nc - tapn * a.F; }
    const bool need_t = (BMODE == 1) || a.bshift;

    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

    // tiles of this wave that lie wholly in the padding of d W (M, N rounded up to 32, the block to 128) issue nothing: with
    // N = 166 + 1 the second 128-column block holds two real 32-column tiles of four, with M = 480 the fourth block row three of
    // four -- at the GRU layer's shapes 30 % of the 2 x 2 x (blocks) tiles (wave-uniform predicates: no divergence)
    const bool mval[2] = {128 * mb + 64 * wm < a.Mp, 128 * mb + 64 * wm + 32 < a.Mp};
    const bool nval[2] = {128 * nb + 64 * wn < a.Np, 128 * nb + 64 * wn + 32 < a.Np};
    // The loads of a piece are issued raw (clamped addresses, nothing looks at the values) and masked only when they are
    // written to LDS, after the MFMAs of the piece before: a select right behind the load made the compiler fold the load
    // into the select's control flow and wait for each one in turn (four serial memory round trips per staging step).
    float av[8], bv[8];
    // vector staging role: thread -> float4 (tid & 31) of rows (tid >> 5) + 8 e, e = 0..1
    const int vc4 = (tid & 31) * 4, vrow0 = tid >> 5;
    const int colA = 128 * mb + vc4, colB = 128 * nb + vc4;
    const int ccA = colA + 3 < (int)a.lda ? colA : (int)a.lda - 4;
    const int ccB = colB + 3 < (int)a.ldb ? colB : (int)a.ldb - 4;
    auto gload = [&](long r0) {
        if (VEC) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long row = r0 + vrow0 + 8 * e;
                const long rowc = row < rend ? row : a.R - 1;
                long br = rowc - (a.bshift ? 1 : 0);
                br = br < 0 ? 0 : br;
                const f32x4 va = *reinterpret_cast<const f32x4*>(a.A + rowc * a.lda + ccA);
                const f32x4 vb = *reinterpret_cast<const f32x4*>(a.B + br * a.ldb + ccB);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { av[4 * e + s4] = va[s4]; bv[4 * e + s4] = vb[s4]; }
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long row = r0 + srow0 + 2 * e;
            const long rowc = row < rend ? row : a.R - 1;
            av[e] = a.A[rowc * a.lda + mc];
            if (BMODE == 1) {
                long xr = rowc + tapn - a.pad;
                xr = xr < 0 ? 0 : (xr < a.R ? xr : a.R - 1);
                bv[e] = a.B[xr * a.F + chn];
            } else {
                long br = rowc - (a.bshift ? 1 : 0);
                br = br < 0 ? 0 : br;
                bv[e] = a.B[br * a.ldb + nc];
            }
        }
    };
    // r0: the first row of the piece held in av / bv
    auto sstore = [&](int buf, long r0) {
        if (VEC) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long row = r0 + vrow0 + 8 * e;
                const bool rok = row < rend;
                const long rowc = rok ? row : a.R - 1;
                const bool tok = !(a.bshift && ((unsigned)rowc % (unsigned)T) == 0);
                const bool aok = rok && ccA == colA, bok = rok && tok && ccB == colB;
                f32x4 va, vb;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    va[s4] = (aok && colA + s4 < a.M) ? av[4 * e + s4] : 0.f;
                    const float x = (bok && colB + s4 < a.N) ? bv[4 * e + s4] : 0.f;
                    vb[s4] = (rok && colB + s4 == a.N) ? 1.f : x;               // the all-ones column
                }
                *reinterpret_cast<f32x4*>(&As[buf][vrow0 + 8 * e][vc4]) = va;
                *reinterpret_cast<f32x4*>(&Bs[buf][vrow0 + 8 * e][vc4]) = vb;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long row = r0 + srow0 + 2 * e;
            const bool rok = row < rend;
            const long rowc = rok ? row : a.R - 1;
            bool ok = rok && ncol < a.N;
            if (BMODE == 1) {
                const int t = (int)((unsigned)rowc % (unsigned)T);      // R = windows * steps fits 32 bits
                const int tt = t + tapn - a.pad;
                ok = ok && tt >= 0 && tt < T;
            } else if (need_t)
                ok = ok && !(a.bshift && ((unsigned)rowc % (unsigned)T) == 0);
            As[buf][srow0 + 2 * e][scol] = (rok && mcol < a.M) ? av[e] : 0.f;
            const float vb = ok ? bv[e] : 0.f;
            Bs[buf][srow0 + 2 * e][scol] = (rok && ncol == a.N) ? 1.f : vb;        // the all-ones column: bias gradients
        }
    };

    // the main loop for NX x NY real tiles of this wave (the valid tiles are the leading ones), chosen once outside the loop: a
    // predicate inside it costs the software pipeline its counted waits (21 % slower at 256 windows when tried that way)
    auto main_loop = [&](auto nx_tag, auto ny_tag) {
        constexpr int NX = decltype(nx_tag)::value, NY = decltype(ny_tag)::value;
        gload(rbeg);
        sstore(0, rbeg);
        __syncthreads();
        int buf = 0;
        for (long r = rbeg; r < rend; r += WG_ROWS) {
            const bool more = r + WG_ROWS < rend;
            if (more) gload(r + WG_ROWS);                    // in flight during the MFMAs below
            if constexpr (X3) {
                static_assert(WG_ROWS == 16, "one 16-row chunk per staging step");
                f32x4 ap[2][3], bp[2][3];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 lo, hi;
                    if constexpr (NY > 0)
                        if (u < NX) {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) { lo[s4] = As[buf][4 * kk + s4][64 * wm + 32 * u + c]; hi[s4] = As[buf][8 + 4 * kk + s4][64 * wm + 32 * u + c]; }
                            split3(lo, hi, ap[u][0], ap[u][1], ap[u][2]);
                        }
                    if constexpr (NX > 0)
                        if (u < NY) {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) { lo[s4] = Bs[buf][4 * kk + s4][64 * wn + 32 * u + c]; hi[s4] = Bs[buf][8 + 4 * kk + s4][64 * wn + 32 * u + c]; }
                            split3(lo, hi, bp[u][0], bp[u][1], bp[u][2]);
                        }
                }
#pragma unroll
                for (int x = 0; x < NX; ++x)
#pragma unroll
                    for (int y = 0; y < NY; ++y) acc[x][y] = mfma_s3(ap[x], bp[y], acc[x][y]);
            } else
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int lr = 8 * h + 4 * kk + s4;
                    float af[2], bf[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        af[u] = As[buf][lr][64 * wm + 32 * u + c];
                        bf[u] = Bs[buf][lr][64 * wn + 32 * u + c];
                    }
#pragma unroll
                    for (int x = 0; x < NX; ++x)
#pragma unroll
                        for (int y = 0; y < NY; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[x], bf[y], acc[x][y], 0, 0, 0);
                }
            if (more) sstore(buf ^ 1, r + WG_ROWS);
            __syncthreads();
            buf ^= 1;
        }
    };
    if (rbeg < rend) {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        const int nx = a.noskip ? 2 : (mval[0] ? 1 : 0) + (mval[1] ? 1 : 0), ny = a.noskip ? 2 : (nval[0] ? 1 : 0) + (nval[1] ? 1 : 0);
        if (nx == 2 && ny == 2) main_loop(I2{}, I2{});
        else if (nx == 0 || ny == 0) main_loop(I0{}, I0{});
        else if (nx == 2) main_loop(I2{}, I1{});
        else if (ny == 2) main_loop(I1{}, I2{});
        else main_loop(I1{}, I1{});
    }
    float* __restrict__ P = a.P + (long)slab * a.Mp * a.Np;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int m0 = 128 * mb + 64 * wm + 32 * x, n = 128 * nb + 64 * wn + 32 * y + c;
            if (m0 < a.Mp && n < a.Np) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = m0 + (q & 3) + 8 * (q >> 2) + 4 * kk;
                    P[(long)m * a.Np + n] = acc[x][y][q];
                }
            }
        }
}

// 64 outputs per workgroup, four threads per output (slabs g, g + 4, ...; the four partial sums are added in a fixed order through
// LDS): with one thread per output the 64 .. 128 slab reads of an element were one chain on a launch of ~1 workgroup per CU
// (14 us for 20 MB at 256 windows, twelve launches per training step)
__device__ __forceinline__ void wgrad_reduce_block(const WgradReduceArgs& a, const unsigned block) {
    __shared__ float part[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long idx = (long)block * 64 + e;
    const int NN = a.N + 1;
    const bool ok = idx < (long)a.M * NN;
    int m = 0, n = 0;
    float v = 0.f;
    if (ok) {
        m = (int)(idx / NN);
        n = (int)(idx - (long)m * NN);
        if (a.rowmapW || n < a.N)
            for (int s = g; s < a.nslab; s += 4) v += a.P[((long)s * a.Mp + m) * a.Np + n];
    }
    part[g][e] = v;
    __syncthreads();
    if (g != 0 || !ok) return;
    v = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    if (!a.rowmapW) {
        if (n < a.N) a.outW[n] += v;                    // a queued column sum (launch_sum_rows_part)
    } else if (n < a.N) {
        const int ro = a.rowmapW[m], co = a.colmap[n];
        if (ro >= 0 && co >= 0) a.outW[ro + co] += v;
    } else if (a.rowmapB) {
        const int ro = a.rowmapB[m];
        if (ro >= 0) a.outB[ro] += v;
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const WgradReduceArgs a) { wgrad_reduce_block(a, blockIdx.x); }
// every reduction of a training step in one launch (round 6): workgroups [first_block[i], first_block[i + 1]) serve entry i
__global__ __launch_bounds__(256) void k_wgrad_reduce_batch(const WgradReduceBatch b) {
    int i = 0;
#pragma unroll 1
    while (i + 1 < b.n && blockIdx.x >= b.first_block[i + 1]) ++i;                // (uniform, <= 15 steps)
    wgrad_reduce_block(b.e[i], blockIdx.x - b.first_block[i]);
}

int launch_wgrad(const WgradArgs& a, hipStream_t s) {
    if (a.R <= 0 || a.M <= 0 || a.N <= 0) return 0;
    {
        static const int plain = getenv("MTADGAT_WGRAD_PLAIN") ? 1 : 0;       // measurement hook: the (tile, slab) grid as launched
        const_cast<WgradArgs&>(a).plain_map = plain;
        static const int noskip = getenv("MTADGAT_WGRAD_NOSKIP") ? 1 : 0;    // measurement hook: issue the all-padding tiles as well
        const_cast<WgradArgs&>(a).noskip = noskip;
        const dim3 grid((unsigned)(((a.Mp + 127) / 128) * ((a.Np + 127) / 128)), (unsigned)a.nslab);
        const bool vec = a.bmode == 0 && (a.lda & 3) == 0 && (a.ldb & 3) == 0 && a.lda >= 4 && a.ldb >= 4 &&
                         (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 15) == 0;
        if (a.x3) {
            if (a.bmode == 1)
                hipLaunchKernelGGL((k_wgrad_lds<1, false, true>), grid, dim3(256), 0, s, a);
            else if (vec)
                hipLaunchKernelGGL((k_wgrad_lds<0, true, true>), grid, dim3(256), 0, s, a);
            else
                hipLaunchKernelGGL((k_wgrad_lds<0, false, true>), grid, dim3(256), 0, s, a);
        } else if (a.bmode == 1)
            hipLaunchKernelGGL((k_wgrad_lds<1, false>), grid, dim3(256), 0, s, a);
        else if (vec)
            hipLaunchKernelGGL((k_wgrad_lds<0, true>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((k_wgrad_lds<0, false>), grid, dim3(256), 0, s, a);
    }
    LAUNCH_CHECK();
    return 0;
}

int launch_wgrad_reduce_batch(const WgradReduceArgs* e, int n, hipStream_t s) {
    if (n <= 0) return 0;
    if (n > WGRAD_BATCH_MAX) return -2;
    WgradReduceBatch b{};
    unsigned blocks = 0;
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const long total = (long)e[i].M * (e[i].N + 1);
        if (total <= 0) continue;
        b.e[k] = e[i];
        b.first_block[k] = blocks;
        blocks += (unsigned)((total + 63) / 64);
        ++k;
    }
    b.first_block[k] = blocks;
    b.n = k;
    if (k == 0) return 0;
    hipLaunchKernelGGL(k_wgrad_reduce_batch, dim3(blocks), dim3(256), 0, s, b);
    LAUNCH_CHECK();
    return 0;
}

int launch_wgrad_reduce(const WgradReduceArgs& a, hipStream_t s) {
    const long total = (long)a.M * (a.N + 1);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// column sums over many rows (d bias of the attention layers = sum over windows of d e; d a partials)
// ---------------------------------------------------------------------------
static inline int sum_rows_slabs(long R) { return (int)(R < 64 ? (R < 1 ? 1 : R) : 64); }
size_t sum_rows_scratch(long R, int N) { return (size_t)sum_rows_slabs(R) * (size_t)N; }

__global__ void k_sum_rows1(const float* __restrict__ src, long ld, long R, int N, long rows_per_slab, float* __restrict__ part) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const long r0 = (long)blockIdx.y * rows_per_slab;
    const long r1 = r0 + rows_per_slab < R ? r0 + rows_per_slab : R;
    float v = 0.f;
    for (long r = r0; r < r1; ++r) v += src[r * ld + n];
    part[(long)blockIdx.y * N + n] = v;
}
// (four threads per output, as k_wgrad_reduce: the slab reads of an element are four short chains instead of one)
__global__ __launch_bounds__(256) void k_sum_rows2(const float* __restrict__ part, int nslab, int N, float* __restrict__ dst) {
    __shared__ float ps[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + e;
    float v = 0.f;
    if (n < N)
        for (int s = g; s < nslab; s += 4) v += part[(long)s * N + n];
    ps[g][e] = v;
    __syncthreads();
    if (g == 0 && n < N) dst[n] += (ps[0][e] + ps[1][e]) + (ps[2][e] + ps[3][e]);
}
int launch_sum_rows_part(const float* src, long ld, long R, int N, float* scratch, int* nslab, hipStream_t s) {
    *nslab = 0;
    if (R <= 0 || N <= 0) return 0;
    const int S = sum_rows_slabs(R);
    const long rps = (R + S - 1) / S;
    hipLaunchKernelGGL(k_sum_rows1, dim3((unsigned)((N + 255) / 256), (unsigned)S), dim3(256), 0, s, src, ld, R, N, rps, scratch);
    LAUNCH_CHECK();
    *nslab = S;
    return 0;
}
int launch_sum_rows(const float* src, long ld, long R, int N, float* scratch, float* dst, hipStream_t s) {
    if (R <= 0 || N <= 0) return 0;
    const int S = sum_rows_slabs(R);
    const long rps = (R + S - 1) / S;
    hipLaunchKernelGGL(k_sum_rows1, dim3((unsigned)((N + 255) / 256), (unsigned)S), dim3(256), 0, s, src, ld, R, N, rps, scratch);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_rows2, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, scratch, S, N, dst);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// GRU backward through time.  Workgroup = 32 windows, wave c = hidden tile c (as k_gru_split).  Per step,
// from the last to the first:
//     d n = d h (1 - z),  d a_n = d n (1 - n^2),  d a_z = d h (h_{t-1} - n) z (1 - z),
//     d a_r = d a_n q r (1 - r),  d a_nh = d a_n r          (q = W_hn h_{t-1} + b_hn)
//     d h_{t-1} = d h z + W_hr^T d a_r + W_hz^T d a_z + W_hn^T d a_nh   (+ the external gradient of h_{t-1})
// The three gradient blocks are published in LDS in F-layout and every wave runs its 32-unit tile of the
// transposed recurrent product on the MFMA (same chunk / lane mapping as the forward).  d a_* of every step
// are written out for the weight-gradient GEMMs and the input-gradient rowgemm.
// ---------------------------------------------------------------------------
// (Rounds 2-5 carried a bf16-operand build of the transposed recurrent product as an opt-in "bf16 training step"; it lost to this
// fp32 step at every batch size -- 7.98 vs 2.32 ms at 256 windows, 33.9 vs 28.5 at 8 192 -- and is gone: a bf16 request trains here.)
// X3 (round 6, the default arithmetic): the transposed recurrent product on three bf16 pieces per operand (mfma_s3: six
// v_mfma_f32_32x32x16_bf16 per 16 features instead of eight v_mfma_f32_32x32x2_f32 -- 2.7x less matrix time, fp32-class results;
// the gradient blocks are split where they are published, the weights once per upload: whT3).
template <bool X3>
__global__ __launch_bounds__(512) void k_gru_bwd(const GruBwdArgs a) {
    constexpr bool BF = false;
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    f32x4* __restrict__ das = reinterpret_cast<f32x4*>(gsm);          // fp32: [3][NCG][4][64] float4; X3: [3][NCG][2][3 pieces][64] containers
    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NCG = a.NCG;
    const int i = lane & 31, g = lane >> 5;
    const long win = (long)blockIdx.x * 32 + i;
    const long winc = win < a.B ? win : a.B - 1;
    const int T = a.T, Hp = a.Hp;
    constexpr int CPT = (BF || X3) ? 2 : 4;                           // chunks per 32-unit tile
    constexpr int RING = X3 ? 6 : (BF ? 3 : 4);                       // chunks in flight (a divisor of NQ = 3 CPT NCG)
    constexpr int WPC = X3 ? 3 : 1;                                   // 16-byte words per weight chunk and lane
    const int NQ = 3 * CPT * NCG;
    const f32x4* __restrict__ W = a.WhT + (long)c * NQ * (64 * WPC) + lane;
    const int col0 = 32 * c + 4 * g;

    f32x16 dh;
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] = 0.f;
    if (a.DHend) {
        const float* p = a.DHend + winc * a.ldde + col0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 8 * m);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) dh[4 * m + s4] = v[s4];
        }
    }
    f32x4 wr[RING][WPC];
#pragma unroll
    for (int u = 0; u < RING; ++u)
#pragma unroll
        for (int pc = 0; pc < WPC; ++pc) wr[u][pc] = W[(u * WPC + pc) * 64];

    for (int t = T - 1; t >= 0; --t) {
        const long row = winc * T + t;
        const float* gp = a.Gates + row * (4L * Hp) + col0;
        const float* hp = a.Seq + (row - (t > 0 ? 1 : 0)) * (long)Hp + col0;
        f32x4 r4[4], z4[4], n4[4], q4[4], h4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            r4[m] = *reinterpret_cast<const f32x4*>(gp + 8 * m);
            z4[m] = *reinterpret_cast<const f32x4*>(gp + Hp + 8 * m);
            n4[m] = *reinterpret_cast<const f32x4*>(gp + 2 * Hp + 8 * m);
            q4[m] = *reinterpret_cast<const f32x4*>(gp + 3 * Hp + 8 * m);
            h4[m] = *reinterpret_cast<const f32x4*>(hp + 8 * m);
        }
        if (a.DHseq) {
            const float* dp = a.DHseq + row * a.lddh + col0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(dp + 8 * m);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) dh[4 * m + s4] += v[s4];
            }
        }
        const float hmask = t > 0 ? 1.f : 0.f;         // h_{-1} = 0
        f32x16 dhz;
        f32x4 dan4[4], dar4[4], daz4[4], dnh4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float dv = dh[4 * m + s4];
                const float r = r4[m][s4], z = z4[m][s4], n = n4[m][s4], q = q4[m][s4], hprev = h4[m][s4] * hmask;
                const float dan = dv * (1.f - z) * (1.f - n * n);
                dan4[m][s4] = dan;
                daz4[m][s4] = dv * (hprev - n) * z * (1.f - z);
                dar4[m][s4] = dan * q * r * (1.f - r);
                dnh4[m][s4] = dan * r;
                dhz[4 * m + s4] = dv * z;
            }
        if (win < a.B) {
            float* op = a.DA + row * (4L * Hp) + col0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                *reinterpret_cast<f32x4*>(op + 8 * m) = dan4[m];
                *reinterpret_cast<f32x4*>(op + Hp + 8 * m) = dar4[m];
                *reinterpret_cast<f32x4*>(op + 2 * Hp + 8 * m) = daz4[m];
                *reinterpret_cast<f32x4*>(op + 3 * Hp + 8 * m) = dnh4[m];
            }
        }
        if (X3) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                f32x4 p3[3][3];
                split3(dar4[2 * mm], dar4[2 * mm + 1], p3[0][0], p3[0][1], p3[0][2]);
                split3(daz4[2 * mm], daz4[2 * mm + 1], p3[1][0], p3[1][1], p3[1][2]);
                split3(dnh4[2 * mm], dnh4[2 * mm + 1], p3[2][0], p3[2][1], p3[2][2]);
#pragma unroll
                for (int gt = 0; gt < 3; ++gt)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) das[(((gt * NCG + c) * 2 + mm) * 3 + pc) * 64 + lane] = p3[gt][pc];
            }
        } else if (BF) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                das[((0 * NCG + c) * 2 + mm) * 64 + lane] = cvt8(dar4[2 * mm], dar4[2 * mm + 1]);
                das[((1 * NCG + c) * 2 + mm) * 64 + lane] = cvt8(daz4[2 * mm], daz4[2 * mm + 1]);
                das[((2 * NCG + c) * 2 + mm) * 64 + lane] = cvt8(dnh4[2 * mm], dnh4[2 * mm + 1]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                das[((0 * NCG + c) * 4 + m) * 64 + lane] = dar4[m];
                das[((1 * NCG + c) * 4 + m) * 64 + lane] = daz4[m];
                das[((2 * NCG + c) * 4 + m) * 64 + lane] = dnh4[m];
            }
        }
        __syncthreads();
        f32x16 acc = dhz;
        for (int q0 = 0; q0 < NQ; q0 += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                if constexpr (X3) {
                    f32x4 x3[3];
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) x3[pc] = das[((q0 + u) * 3 + pc) * 64 + lane];
                    acc = mfma_s3(wr[u], x3, acc);
                } else {
                    const f32x4 xv = das[(q0 + u) * 64 + lane];
                    acc = BF ? mfma_bf(wr[u][0], xv, acc) : mfma4(wr[u][0], xv, acc);
                }
                int qn = q0 + u + RING;
                qn = qn >= NQ ? qn - NQ : qn;
#pragma unroll
                for (int pc = 0; pc < WPC; ++pc) wr[u][pc] = W[(qn * WPC + pc) * 64];
            }
        }
        dh = acc;
        __syncthreads();
    }
}

int launch_gru_bwd(const GruBwdArgs& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    const size_t lds = (size_t)3 * a.NCG * (a.x3 ? 6 : 4) * 64 * sizeof(f32x4);
    if (a.NCG < 1 || a.NCG > 8 || lds > 160 * 1024) return -2;
    if (a.bf16) return -2;
    const void* fn = a.x3 ? reinterpret_cast<const void*>(&k_gru_bwd<true>) : reinterpret_cast<const void*>(&k_gru_bwd<false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if (a.x3) hipLaunchKernelGGL(k_gru_bwd<true>, dim3((unsigned)((a.B + 31) / 32)), dim3(64 * a.NCG), lds, s, a);
    else hipLaunchKernelGGL(k_gru_bwd<false>, dim3((unsigned)((a.B + 31) / 32)), dim3(64 * a.NCG), lds, s, a);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// graph attention backward, part 1 (one workgroup per window, thread <-> (query row, key) pair grid of the
// forward k_gat):  h = sigmoid(S), S = att' V, att' = dropout(att), att = softmax_j(e)
//     d S = d h * h (1 - h);  d att' = d S V^T;  d att = d att' * mask / (1 - p)
//     d e_ij = att_ij (d att_ij - sum_j att_ij d att_ij)              -> DE   (also d bias before the batch sum)
//     d V_j (aggregation path) = sum_i att'_ij d S_i                     -> DV   (MFMA 16x16x4 + LDS float atomics)
// ---------------------------------------------------------------------------

// LDS: d S [Kp16][vld] | { V [Kp16][vld] (d att phase)  /  att'^T [Kp16 keys][Kp16 + 4 rows] (d V phase) }
size_t gat_bwd_att_lds(int K, int D, int vld, int nwa) {
    (void)D; (void)nwa;
    const int Kp16 = (K + 15) & ~15;
    const size_t v = (size_t)Kp16 * vld, t = (size_t)Kp16 * (Kp16 + 4);
    return (v + (v > t ? v : t)) * sizeof(float);
}

template <int RJ>
__device__ __forceinline__ float bw_row_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    if (RJ == 16) v += dpp_move<0x140>(v);
    return v;
}

template <int IBL, int JPL, int RJ>
__global__ __launch_bounds__(512) void k_gat_bwd_att(const GatBwdAttArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RI = 64 / RJ;
    constexpr int IBW = RI * IBL;
    static_assert(IBW == 16, "a wave owns 16 query rows");
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long win = blockIdx.x;
    const int K = a.K, D = a.D, vld = a.vld;
    const int Kp16 = (K + 15) & ~15;
    const int NWA = Kp16 >> 4;
    float* __restrict__ dSs = smem;
    float* __restrict__ Vs = dSs + Kp16 * vld;           // d att phase
    float* __restrict__ attT = Vs;                        // d V phase (aliases V: a barrier separates the phases)
    const int AP = Kp16 + 4;                              // pitch of an att'^T row (one key, the query rows)
    const int lj = lane % RJ, li = lane / RJ;

    // ---- stage V, d S (zero padded).  Unconditional loads from clamped addresses in batches of eight,
    // every load of a batch issued before its first LDS store (round 6).  The first version guarded each load (`in range ? src[..] : 0`):
    // a branch with s_waitcnt vmcnt(0) at the join per element, i.e. 13 + 13 serial memory round trips per window at the MSL shape.
    {
        const float* __restrict__ vsrc = a.V + win * (long)(a.vt ? D : K) * a.ldv;
        const float* __restrict__ hsrc = a.H + win * a.so_w;
        const float* __restrict__ dsrc = a.dH + win * a.so_w;
        const int total = Kp16 * vld;
        const bool vfast = a.vt != 0, sfast = a.so_d != 1;       // the element order that walks the unit stride of each source
        constexpr int MAXU = 8;
        for (int base = 0; base < total; base += MAXU * nthr) {
            float vv[MAXU], hv[MAXU], dv[MAXU];
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = base + tid + n * nthr;
                const int uc = u < total ? u : total - 1;
                int node, col;
                if (vfast) { col = uc / Kp16; node = uc - col * Kp16; } else { node = uc / vld; col = uc - node * vld; }
                const int nc = node < K ? node : K - 1, cc = col < D ? col : D - 1;
                vv[n] = vsrc[vfast ? (long)cc * a.ldv + nc : (long)nc * a.ldv + cc];       // (ONE load from a selected address: two loads behind a select are a branch)
                if (sfast) { col = uc / Kp16; node = uc - col * Kp16; } else { node = uc / vld; col = uc - node * vld; }
                const int n2 = node < K ? node : K - 1, c2 = col < D ? col : D - 1;
                const long o = (long)n2 * a.so_i + (long)c2 * a.so_d;
                hv[n] = hsrc[o];
                dv[n] = dsrc[o];
            }
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = base + tid + n * nthr;
                if (u < total) {
                    int node, col;
                    if (vfast) { col = u / Kp16; node = u - col * Kp16; } else { node = u / vld; col = u - node * vld; }
                    Vs[node * vld + col] = (node < K && col < D) ? vv[n] : 0.f;
                    if (sfast) { col = u / Kp16; node = u - col * Kp16; } else { node = u / vld; col = u - node * vld; }
                    dSs[node * vld + col] = (node < K && col < D) ? dv[n] * hv[n] * (1.f - hv[n]) : 0.f;
                }
            }
        }
    }
    __syncthreads();

    const bool rows_owner = wave < NWA;
    const int i0 = (rows_owner ? wave : 0) * IBW;
    float acc[IBL][JPL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ii][jj] = 0.f;

    if (rows_owner) {
        // ---- d att'_ij = d S_i . V_j over the D features (columns >= D are zero in dSs)
        const float* lrow[IBL];
        const float* rrow[JPL];
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) lrow[ii] = dSs + (i0 + li + RI * ii) * vld;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            const int j = lj + RJ * jj;
            rrow[jj] = Vs + (j < Kp16 ? j : Kp16 - 1) * vld;
        }
        const int Dr = (D + 1) & ~1;
        for (int d = 0; d < Dr; d += 2) {
            f32x2 l[IBL], r[JPL];
#pragma unroll
            for (int ii = 0; ii < IBL; ++ii) l[ii] = *reinterpret_cast<const f32x2*>(lrow[ii] + d);
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) r[jj] = *reinterpret_cast<const f32x2*>(rrow[jj] + d);
#pragma unroll
            for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
                for (int jj = 0; jj < JPL; ++jj) acc[ii][jj] += l[ii][0] * r[jj][0] + l[ii][1] * r[jj][1];
        }
        // ---- softmax backward (rows live in RJ adjacent lanes x JPL registers); acc becomes att'
        const unsigned key = drop_window_key(a.drop, a.drop_stream, win);
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) {
            const int irow = i0 + li + RI * ii;
            const int irc = irow < K ? irow : K - 1;
            const float* __restrict__ ap = a.ATT + (win * K + irc) * (long)K;
            float attv[JPL], dat[JPL];
            float csum = 0.f;
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = lj + RJ * jj;
                const int jc = j < K ? j : K - 1;
                const bool valid = irow < K && j < K;
                const float avl = ap[jc];                  // (unconditional from the clamped address: the row's loads go out together)
                const float av = valid ? avl : 0.f;
                float sc = 1.f;
                if (a.drop.thresh) sc = drop_keep(key, (unsigned)(irc * K + jc), a.drop.thresh) ? a.drop.keep_scale : 0.f;
                const float datt = acc[ii][jj] * sc;
                csum += av * datt;
                attv[jj] = av;
                dat[jj] = datt;
                acc[ii][jj] = av * sc;
            }
            csum = bw_row_sum<RJ>(csum);
            float* __restrict__ dep = a.DE + (win * K + irc) * (long)K;
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = lj + RJ * jj;
                if (irow < K && j < K) dep[j] = attv[jj] * (dat[jj] - csum);
            }
        }
    }
    // ---- d V (aggregation path): d V[key][feature] = sum over ALL query rows att'[row][key] d S[row][feature]  (round 6).
    // The waves' att' rows go through LDS transposed (att'^T over the V tile, which is dead now); then a wave owns 16 KEYS and
    // contracts over every row on v_mfma_f32_16x16x4_f32 -- A[m = key][k], B[k][n = feature], the contraction index of lane
    // group kb running over rows kb Kp16/4 + s so that a lane's A words are contiguous (16-byte reads, kept across the feature
    // tiles) -- and writes its block of d V straight to memory.  Before: a wave contracted over ITS 16 rows only and the waves
    // added their partial [K][D] blocks into an LDS accumulator one after the other (2 x 7 barrier-separated steps per window),
    // with 111 KB of LDS per workgroup (one per CU; now 79 KB at the MSL temporal layer, 53 KB at the feature layer: two / three).
    __syncthreads();                                       // every wave is done with V
    if (rows_owner) {
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = lj + RJ * jj;
                if (j < Kp16) attT[j * AP + i0 + li + RI * ii] = acc[ii][jj];        // (rows / keys past K carry zeros)
            }
    }
    __syncthreads();
    {
        const int nr = lane & 15, kb = lane >> 4;
        const int NW = nthr >> 6, NKT = Kp16 >> 4, S4 = Kp16 >> 4;       // S4: 16-byte words of a lane's share of an att'^T row (Kp16 / 4 rows)
        const int DT = (D + 15) >> 4;
        const int rbase = kb * (Kp16 >> 2);                // this lane group's rows: rbase + s, s < Kp16 / 4
        for (int kt = wave; kt < NKT; kt += NW) {
            constexpr int S4M = 8;                         // K <= 128
            f32x4 aw[S4M];
            const f32x4* __restrict__ ap4 = reinterpret_cast<const f32x4*>(attT + (16 * kt + nr) * AP + rbase);
#pragma unroll
            for (int w4 = 0; w4 < S4M; ++w4)
                if (w4 < S4) aw[w4] = ap4[w4];
            for (int dt = 0; dt < DT; ++dt) {
                const int dcol = 16 * dt + nr;
                const float* __restrict__ bp = dSs + rbase * vld + (dcol < vld ? dcol : vld - 1);
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w4 = 0; w4 < S4M; ++w4)
                    if (w4 < S4) {
                        float bv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = bp[(4 * w4 + e) * vld];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[w4][e], bv[e], o, 0, 0, 0);
                    }
                // D register q of lane (nr, kb) = out[key 4 kb + q][feature nr] of the block
                if (dcol < D) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int key = 16 * kt + 4 * kb + q;
                        if (key < K) a.DV[(win * K + key) * (long)a.lddv + dcol] = o[q];
                    }
                }
            }
        }
    }
}

#define GBA_CASE(I, J, RJ_)                                                                                      \
    if (IBL == I && JPL == J && rj == RJ_) {                                                                     \
        if (lds_bytes > 64 * 1024) {                                                                             \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gat_bwd_att<I, J, RJ_>),        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);     \
            if (e_ != hipSuccess) return (int)e_;                                                                \
        }                                                                                                        \
        hipLaunchKernelGGL((k_gat_bwd_att<I, J, RJ_>), dim3(grid), dim3(64 * nw), lds_bytes, s, a);              \
        launched = true;                                                                                         \
    }

int launch_gat_bwd_att(const GatBwdAttArgs& a, int IBL, int JPL, int rj, int nw, size_t lds_bytes, hipStream_t s) {
    if (a.nwin <= 0) return 0;
    if (rj * JPL < a.K || nw * 16 < a.K || nw > 8 || lds_bytes > 160 * 1024) return -2;
    const unsigned grid = (unsigned)a.nwin;
    bool launched = false;
    GBA_CASE(4, 1, 16) GBA_CASE(4, 2, 16) GBA_CASE(4, 3, 16) GBA_CASE(4, 4, 16) GBA_CASE(4, 5, 16) GBA_CASE(4, 6, 16) GBA_CASE(4, 7, 16) GBA_CASE(4, 8, 16)
    GBA_CASE(2, 1, 8) GBA_CASE(2, 3, 8) GBA_CASE(2, 5, 8) GBA_CASE(2, 7, 8)
    if (!launched) return -2;
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// graph attention backward, part 2 for GAT (v1) scores  e_ij = LeakyReLU(s_ij),  s_ij = c_i + d_j,
//     c_i = a1 . (W v_i + b),  d_j = a2 . (W v_j + b)          (reference modules.py:80-83 / :180-183)
// Everything below d s_ij = d e_ij [s_ij > 0 ? 1 : alpha] is linear in the node vectors, so no projected embedding is ever
// formed: with u1 = W^T a1, u2 = W^T a2 (k_gat_v1_prep, once per call), d c_i = sum_j d s_ij and d d_j = sum_i d s_ij,
//     d v_i += d c_i u1 + d d_i u2
//     per window:  p1 = sum_i d c_i v_i,  p2 = sum_j d d_j v_j,  sc = sum d c,  sd = sum d d      (k_gat_bwd_v1)
//     over the batch (k_sum_rows) P1, P2, SC, SD, then (k_gat_v1_finish)
//     d W = a1 (x) P1 + a2 (x) P2,   d b = a1 SC + a2 SD,   d a1 = W P1 + b SC,   d a2 = W P2 + b SD.
// ---------------------------------------------------------------------------
__global__ void k_gat_v1_prep(const float* __restrict__ Wm, const float* __restrict__ bv, const float* __restrict__ av, int E, int D,
                              float* __restrict__ u) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < D) {
        float s1 = 0.f, s2 = 0.f;
        for (int e = 0; e < E; ++e) {
            const float w = Wm[(long)e * D + d];
            s1 += av[e] * w;
            s2 += av[E + e] * w;
        }
        u[d] = s1; u[D + d] = s2;
    } else if (d == D || d == D + 1) {
        float s = 0.f;
        for (int e = 0; e < E; ++e) s += av[(d - D) * E + e] * bv[e];
        u[2 * D + (d - D)] = s;
    }
}

// one workgroup (256 threads) per window.  LDS: Vs [K][D + 1] | cq[K] | dk[K] | dc[K] | dd[K]
__global__ __launch_bounds__(256) void k_gat_bwd_v1(const float* __restrict__ V, int ldv, int D, int K, int vt, const float* __restrict__ u,
                                                    const float* __restrict__ DE, float alpha, float* __restrict__ DV, int lddv,
                                                    float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long win = blockIdx.x;
    const int ld = D + 1;
    float* __restrict__ Vs = sm;
    float* __restrict__ cq = Vs + K * ld;
    float* __restrict__ dk = cq + K;
    float* __restrict__ dc = dk + K;
    float* __restrict__ dd = dc + K;
    {
        const float* __restrict__ vsrc = V + win * (long)(vt ? D : K) * ldv;
        if (!vt) {
            for (int x = tid; x < K * D; x += 256) { const int node = x / D, col = x - node * D; Vs[node * ld + col] = vsrc[(long)node * ldv + col]; }
        } else {
            for (int x = tid; x < K * D; x += 256) { const int col = x / K, node = x - col * K; Vs[node * ld + col] = vsrc[(long)col * ldv + node]; }
        }
        for (int x = tid; x < 2 * K; x += 256) dc[x] = 0.f;      // dc and dd
    }
    __syncthreads();
    for (int i = tid; i < K; i += 256) {
        float s1 = 0.f, s2 = 0.f;
        for (int col = 0; col < D; ++col) { const float v = Vs[i * ld + col]; s1 += u[col] * v; s2 += u[D + col] * v; }
        cq[i] = s1 + u[2 * D]; dk[i] = s2 + u[2 * D + 1];
    }
    __syncthreads();
    // d s: a wave per query row (lanes over the keys): row sums by a wave reduction, column sums kept per lane
    const float* __restrict__ de = DE + win * (long)K * K;
    float colacc[2] = {0.f, 0.f};                  // keys lane, lane + 64  (K <= 128)
    for (int i = wave; i < K; i += 4) {
        const float ci = cq[i];
        float rs = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = lane + 64 * h;
            if (j < K) {
                const float sij = ci + dk[j];
                const float ds = de[(long)i * K + j] * (sij > 0.f ? 1.f : alpha);
                rs += ds; colacc[h] += ds;
            }
        }
        rs = wave_sum(rs);
        if (lane == 0) dc[i] = rs;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (lane + 64 * h < K) atomicAdd(&dd[lane + 64 * h], colacc[h]);
    __syncthreads();
    for (int x = tid; x < K * D; x += 256) {
        const int node = x / D, col = x - node * D;
        DV[(win * K + node) * (long)lddv + col] += dc[node] * u[col] + dd[node] * u[D + col];
    }
    float* __restrict__ po = part + win * (long)(2 * D + 2);
    for (int col = tid; col < D; col += 256) {
        float p1 = 0.f, p2 = 0.f;
        for (int i = 0; i < K; ++i) { const float v = Vs[i * ld + col]; p1 += dc[i] * v; p2 += dd[i] * v; }
        po[col] = p1; po[D + col] = p2;
    }
    if (tid == 255) {
        float sc = 0.f, sd = 0.f;
        for (int i = 0; i < K; ++i) { sc += dc[i]; sd += dd[i]; }
        po[2 * D] = sc; po[2 * D + 1] = sd;
    }
}

// P = [P1 (D) | P2 (D) | SC | SD] summed over the batch; gradients accumulate into the flat buffer
__global__ void k_gat_v1_finish(const float* __restrict__ P, const float* __restrict__ Wm, const float* __restrict__ bv,
                                const float* __restrict__ av, int E, int D, float* __restrict__ gW, float* __restrict__ gb,
                                float* __restrict__ ga) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)E * D) {
        const int e = (int)(idx / D), d = (int)(idx - (long)e * D);
        gW[idx] += av[e] * P[d] + av[E + e] * P[D + d];
    }
    if (idx < E) {
        const int e = (int)idx;
        const float sc = P[2 * D], sd = P[2 * D + 1];
        gb[e] += av[e] * sc + av[E + e] * sd;
        float s1 = 0.f, s2 = 0.f;
        for (int d = 0; d < D; ++d) { const float w = Wm[(long)e * D + d]; s1 += w * P[d]; s2 += w * P[D + d]; }
        ga[e] += s1 + bv[e] * sc;
        ga[E + e] += s2 + bv[e] * sd;
    }
}

size_t gat_bwd_v1_lds(int K, int D) { return ((size_t)K * (D + 1) + 4 * (size_t)K) * sizeof(float); }

int launch_gat_v1_prep(const float* Wm, const float* bv, const float* av, int E, int D, float* u, hipStream_t s) {
    hipLaunchKernelGGL(k_gat_v1_prep, dim3((unsigned)((D + 2 + 255) / 256)), dim3(256), 0, s, Wm, bv, av, E, D, u);
    LAUNCH_CHECK();
    return 0;
}
int launch_gat_bwd_v1(const float* V, int ldv, int D, int K, int vt, const float* u, const float* DE, float alpha, float* DV, int lddv,
                      float* part, long nwin, hipStream_t s) {
    if (nwin <= 0) return 0;
    if (K > 128) return -2;
    const size_t lds = gat_bwd_v1_lds(K, D);
    if (lds > 64 * 1024) {
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gat_bwd_v1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return (int)e_;
    }
    hipLaunchKernelGGL(k_gat_bwd_v1, dim3((unsigned)nwin), dim3(256), lds, s, V, ldv, D, K, vt, u, DE, alpha, DV, lddv, part);
    LAUNCH_CHECK();
    return 0;
}
int launch_gat_v1_finish(const float* P, const float* Wm, const float* bv, const float* av, int E, int D, float* gW, float* gb, float* ga,
                         hipStream_t s) {
    const long total = (long)E * D;
    hipLaunchKernelGGL(k_gat_v1_finish, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P, Wm, bv, av, E, D, gW, gb, ga);
    LAUNCH_CHECK();
    return 0;
}

// (graph attention backward, part 2 -- the GATv2 score backward -- is k_bw_pair in mtadgat_bwdw.hip since round 6: the un-scaled
// projections come from a row GEMM and one register-blocked pass serves fused and wide layers alike.  The per-window kernel that
// re-projected L, R on the fp32 MFMA inside the workgroup, k_gat_bwd_pair, took 4.07 + 2.21 ms per 8 192 MSL windows against
// 1.7 + 1.1 + 2 x 0.25 for projection GEMM + k_bw_pair, and is gone.)

// ---------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------
__global__ void k_xdec(const float* __restrict__ hend, long ldh, int H, int T, long B, float* __restrict__ X, long ldx) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = B * T * ldx;
    if (idx >= total) return;
    const long row = idx / ldx;
    const int j = (int)(idx - row * ldx);
    const long b = row / T;
    const int t = (int)(row - b * T);
    X[idx] = j < H ? hend[b * ldh + ((long)t * H + j) / T] : 0.f;
}
int launch_xdec(const float* hend, long ldh, int H, int T, long B, float* X, long ldx, hipStream_t s) {
    const long total = B * T * ldx;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_xdec, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hend, ldh, H, T, B, X, ldx);
    LAUNCH_CHECK();
    return 0;
}

// adjoint of the decoder input x_t[j] = h_end[(t H + j) / T] (modules.py:279): d h_end[m] += sum of the T consecutive entries
// m T .. m T + T - 1 of the window's flattened (T, H) block.  One workgroup per window: coalesced reads of the block into LDS,
// then one thread per entry sums its T values (blocks beyond 60 KB: sums through LDS float atomics instead); H <= 256.
template <bool STAGED>
__global__ __launch_bounds__(256) void k_xdec_bwd(const float* __restrict__ dX, long ldx, int H, int T, long B, float* __restrict__ dhend, long ldh) {
    extern __shared__ float xs[];                   // STAGED: the window's T x H block, flattened without the row padding
    __shared__ float acc[256];
    const long b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* __restrict__ src = dX + b * T * ldx;
    const int total = T * H;
    if constexpr (STAGED) {
        for (int f = tid; f < total; f += 256) { const int t = f / H, j = f - t * H; xs[f] = src[(long)t * ldx + j]; }
        __syncthreads();
        if (tid < H) {
            float v = 0.f;
            for (int k = 0; k < T; ++k) v += xs[tid * T + k];
            dhend[b * ldh + tid] += v;
        }
    } else {
        acc[tid] = 0.f;
        __syncthreads();
        for (int f0 = 0; f0 < total; f0 += 256) {
            const int f = f0 + tid;
            if (f < total) {
                const int t = f / H, j = f - t * H;
                atomicAdd(&acc[f / T], src[(long)t * ldx + j]);
            }
        }
        __syncthreads();
        if (tid < H) dhend[b * ldh + tid] += acc[tid];
    }
}
int launch_xdec_bwd(const float* dX, long ldx, int H, int T, long B, float* dhend, long ldh, hipStream_t s) {
    if (B <= 0) return 0;
    if (H > 256) return -2;
    const size_t lds = (size_t)T * H * sizeof(float);
    if (lds <= 60 * 1024)       // (with the 1 KiB of acc below the 64 KiB that need no opt-in)
        hipLaunchKernelGGL(k_xdec_bwd<true>, dim3((unsigned)B), dim3(256), lds, s, dX, ldx, H, T, B, dhend, ldh);
    else
        hipLaunchKernelGGL(k_xdec_bwd<false>, dim3((unsigned)B), dim3(256), 0, s, dX, ldx, H, T, B, dhend, ldh);
    LAUNCH_CHECK();
    return 0;
}

__global__ void k_dxc(const float* __restrict__ hcat, const float* __restrict__ dhcat, long ldh, const float* __restrict__ dvt,
                      long ldt, const float* __restrict__ dvf, long ldf, long B, int T, int F, float* __restrict__ dpre, long ldp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = B * T * ldp;
    if (idx >= total) return;
    const long row = idx / ldp;
    const int f = (int)(idx - row * ldp);
    float v = 0.f;
    if (f < F) {
        const long b = row / T;
        const int t = (int)(row - b * T);
        if (hcat[row * ldh + f] > 0.f) v = dhcat[row * ldh + f] + dvt[row * ldt + f] + dvf[(b * F + f) * ldf + t];
    }
    dpre[idx] = v;
}
int launch_dxc(const float* hcat, const float* dhcat, long ldh, const float* dvt, long ldt, const float* dvf, long ldf, long B,
               int T, int F, float* dpre, long ldp, hipStream_t s) {
    const long total = B * T * ldp;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_dxc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hcat, dhcat, ldh, dvt, ldt, dvf, ldf, B, T, F, dpre, ldp);
    LAUNCH_CHECK();
    return 0;
}

// Input gradient of the convolution (reference: autograd through ConvLayer.forward, modules.py:18-22; no caller of the reference
// asks for it, so this is a plain kernel, not a tuned one): one workgroup per window, the window's pre-activation gradients
// in LDS, a thread per (t, i) output, the (F, F, taps) weights from L2.
__global__ __launch_bounds__(256) void k_conv_dx(const float* __restrict__ dpre, long ldp, const float* __restrict__ w, long B, int T, int F,
                                                 int taps, int pad, float* __restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* __restrict__ g = dpre + b * T * ldp;
    for (int u = tid; u < T * F; u += 256) {
        const int t = u / F, o = u - t * F;
        smem[t * F + o] = g[(long)t * ldp + o];
    }
    __syncthreads();
    for (int u = tid; u < T * F; u += 256) {
        const int t = u / F, i = u - t * F;
        float acc = 0.f;
        for (int j = 0; j < taps; ++j) {
            const int tt = t - j + pad;
            if (tt < 0 || tt >= T) continue;
            const float* __restrict__ gr = smem + tt * F;
            const float* __restrict__ wp = w + (long)i * taps + j;
            for (int o = 0; o < F; ++o) acc = __builtin_fmaf(wp[(long)o * F * taps], gr[o], acc);
        }
        dx[(b * T + t) * (long)F + i] = acc;
    }
}
int launch_conv_dx(const float* dpre, long ldp, const float* w, long B, int T, int F, int taps, int pad, float* dx, hipStream_t s) {
    if (B <= 0) return 0;
    const size_t lds = (size_t)T * F * sizeof(float);
    if (lds > 64 * 1024) return -2;
    hipLaunchKernelGGL(k_conv_dx, dim3((unsigned)B), dim3(256), lds, s, dpre, ldp, w, B, T, F, taps, pad, dx);
    LAUNCH_CHECK();
    return 0;
}

__global__ void k_dropmask(const DropArgs d, unsigned stream, long nwin, long n, float* __restrict__ mask) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nwin * n) return;
    const long w = idx / n;
    const unsigned e = (unsigned)(idx - w * n);
    const unsigned key = drop_window_key(d, stream, w);
    mask[idx] = (d.thresh == 0 || drop_keep(key, e, d.thresh)) ? 1.f : 0.f;
}
// y[row][j] = x[row][j] * keep(window, t * H + j) / (1 - p): the dropout nn.GRU applies to the outputs of every stacked layer
// but the last (reference modules.py:233 / :253, training only); the same map is its own adjoint (src = dst allowed)
__global__ void k_seq_dropout(const float* __restrict__ src, float* __restrict__ dst, long nwin, int T, int H, int ld, const DropArgs d,
                              unsigned stream) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nwin * T * ld) return;
    const long row = idx / ld;
    const int j = (int)(idx - row * ld);
    const long w = row / T;
    const int t = (int)(row - w * T);
    float v = 0.f;
    if (j < H) {
        v = src[idx];
        if (d.thresh) v = drop_keep(drop_window_key(d, stream, w), (unsigned)(t * H + j), d.thresh) ? v * d.keep_scale : 0.f;
    }
    dst[idx] = v;
}
int launch_seq_dropout(const float* src, float* dst, long nwin, int T, int H, int ld, const DropArgs& d, unsigned stream, hipStream_t s) {
    const long total = nwin * T * ld;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_seq_dropout, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, nwin, T, H, ld, d, stream);
    LAUNCH_CHECK();
    return 0;
}
int launch_dropmask(const DropArgs& d, unsigned stream, long nwin, long n_per_win, float* mask, hipStream_t s) {
    const long total = nwin * n_per_win;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_dropmask, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d, stream, nwin, n_per_win, mask);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
