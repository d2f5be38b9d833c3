// k_rowgemm (Linear layers), k_conv / k_conv_lds (ConvLayer), small copy kernels + launchers
#include <cstdlib>
#include "mtadgat_device.h"

namespace mtadgat {

// ---------------------------------------------------------------------------
// rowgemm: Y[r, :] = act(W * X[r, :] + bias) for R data rows, 32 rows per wave.
//   reference: every nn.Linear on the path -- the GAT `lin` projections
//   (modules.py:76-77, :81, :176-177, :181; re-associated as DESIGN.md section 3
//   describes) and Forecasting_Model (modules.py:307-311).
// ---------------------------------------------------------------------------
template <int NTB>
__device__ __forceinline__ void rowgemm_epilogue(const RowGemmArgs& a, const f32x16 (&acc)[NTB], int n0, long row, long rowc, int g) {
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb) {
        if (n0 + nb >= a.NT) break;
        const bool transposed = (n0 + nb) >= a.NT_rm;
        const long grp = row / a.group;
        const int member = (int)(row - grp * a.group);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int col = 32 * (n0 + nb) + 8 * m + 4 * g;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
            f32x4 v;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float t = acc[nb][4 * m + s] + bv[s];
                v[s] = a.relu ? fmaxf(t, 0.f) : t;
            }
            if (a.drop_thresh) {      // training forward: dropout after the ReLU (modules.py:309-310)
                const DropArgs dd{a.drop_thresh, a.seed_lo, a.seed_hi, a.keep_scale, a.row0};
                const unsigned key = drop_window_key(dd, a.drop_stream, rowc);
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = drop_keep(key, (unsigned)(col + s), a.drop_thresh) ? v[s] * a.keep_scale : 0.f;
            }
            // (the loads below are unconditional from clamped addresses and masked afterwards: guarded element loads were a
            // branch and a full memory round trip each, up to 32 in series per lane)
            if (a.gate) {             // backward through ReLU (+ dropout): the kept activation tells which units were live
                const float* gp = a.gate + rowc * a.ldg;
                float gv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) gv[s] = gp[col + s < a.Nvalid ? col + s : a.Nvalid - 1];
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = (col + s < a.Nvalid && gv[s] > 0.f) ? v[s] * a.gate_scale : 0.f;
            }
            if (a.accumulate && !transposed) {
                const float* yo = a.Y + rowc * a.ldy;
                float yv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) yv[s] = yo[col + s < a.Nvalid ? col + s : a.Nvalid - 1];
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] += (col + s < a.Nvalid) ? yv[s] : 0.f;
            }
            if (row < a.R) {
                if (transposed) {   // lanes i <-> consecutive group members: coalesced 4-byte stores
                    float* tp = a.YT + (grp * a.YT_rows + (col - 32 * a.NT_rm)) * (long)a.YT_ld + member;
#pragma unroll
                    for (int s = 0; s < 4; ++s) tp[(long)s * a.YT_ld] = v[s];
                } else {
                    float* yp = a.Y + row * a.ldy + col;
                    if (a.vec_store && col + 3 < a.Nvalid) {
                        *reinterpret_cast<f32x4*>(yp) = v;
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            if (col + s < a.Nvalid) yp[s] = v[s];
                    }
                }
            }
        }
    }
}

// XV: the rows are read with 16-byte loads (row stride a multiple of 4 floats) -- a template parameter, not a run-time flag: with
// both loaders in one body the compiler serialises the chunk's loads on the registers the two paths share
template <int NTB, bool XV>
__global__ __launch_bounds__(64) void k_rowgemm(const RowGemmArgs a) {
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    const long row = (long)blockIdx.x * 32 + i;
    const long rowc = row < a.R ? row : a.R - 1;
    const float* __restrict__ xrow = a.X + rowc * a.ldx;
    const f32x4* __restrict__ Wp = a.Wp;
    const int Q = a.Q;

    // few rows (small batches): the output tiles are spread over gridDim.y waves so that the launch fills the machine
    for (int n0 = blockIdx.y * NTB; n0 < a.NT; n0 += NTB * gridDim.y) {
        f32x16 acc[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

        if constexpr (NTB == 1) {
            // one tile per wave: the launches that use this build are latency chains (a head Linear on 256 rows: a few waves, nothing
            // else on their SIMD), so the chunks go four at a time -- every load of the four issued before the first MFMA, one
            // memory round trip per four chunks instead of one per chunk (19 chunks of a 150-wide layer: 20 -> 8 us per launch)
            constexpr int U = 4;
            const int n = n0 < a.NT ? n0 : a.NT - 1;
            for (int q = 0; q < Q; q += U) {
                f32x4 xr[U], w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int qq = q + u < Q ? q + u : Q - 1;
                    xr[u] = feat4_raw<XV>(xrow, 8 * qq + 4 * g, a.Kvalid);
                    w[u] = Wp[((long)n * Q + qq) * 64 + lane];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (q + u < Q) acc[0] = mfma4(w[u], feat4_fix<XV>(xr[u], 8 * (q + u) + 4 * g, a.Kvalid), acc[0]);
            }
        } else {
        f32x4 xr = feat4_raw<XV>(xrow, 4 * g, a.Kvalid);
        f32x4 w[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
            w[nb] = Wp[((long)n * Q) * 64 + lane];
        }
        for (int q = 0; q < Q; ++q) {
            const int qn = (q + 1 < Q) ? q + 1 : q;
            const f32x4 xn = feat4_raw<XV>(xrow, 8 * qn + 4 * g, a.Kvalid);        // raw: looked at one chunk later
            const f32x4 xv = feat4_fix<XV>(xr, 8 * q + 4 * g, a.Kvalid);
            f32x4 wn[NTB];
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
                wn[nb] = Wp[((long)n * Q + qn) * 64 + lane];
            }
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) acc[nb] = mfma4(w[nb], xv, acc[nb]);
            xr = xn;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) w[nb] = wn[nb];
        }
        }
        rowgemm_epilogue<NTB>(a, acc, n0, row, rowc, g);
    }
}

// split-bf16 build: the same rows / tiles, the K loop in 16-feature chunks -- the eight values a lane holds of a chunk are split
// into three bf16 pieces (split3), the weights come pre-split ([tile][Q16][piece][64], derived on the device), six
// v_mfma_f32_32x32x16_bf16 per chunk and tile (mfma_s3): 2.7 x less matrix time than the four fp32 MFMAs of two 8-feature
// chunks, products of 24-bit significands.  Used by the data-gradient products of mtadgat_backward (d X = d Y W over all
// (window, step) rows), which ran at the fp32-MFMA rate.
template <int NTB, bool XV>
__global__ __launch_bounds__(64) void k_rowgemm_x3(const RowGemmArgs a) {
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    const long row = (long)blockIdx.x * 32 + i;
    const long rowc = row < a.R ? row : a.R - 1;
    const float* __restrict__ xrow = a.X + rowc * a.ldx;
    const f32x4* __restrict__ Wp = a.Wp3;
    const int Q = a.Q16;
    for (int n0 = blockIdx.y * NTB; n0 < a.NT; n0 += NTB * gridDim.y) {
        f32x16 acc[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        f32x4 ra = feat4_raw<XV>(xrow, 4 * g, a.Kvalid), rb = feat4_raw<XV>(xrow, 8 + 4 * g, a.Kvalid);
        f32x4 w[NTB][3];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) w[nb][pc] = Wp[(((long)n * Q) * 3 + pc) * 64 + lane];
        }
        for (int q = 0; q < Q; ++q) {
            const int qn = (q + 1 < Q) ? q + 1 : q;
            const f32x4 ran = feat4_raw<XV>(xrow, 16 * qn + 4 * g, a.Kvalid), rbn = feat4_raw<XV>(xrow, 16 * qn + 8 + 4 * g, a.Kvalid);    // raw
            const f32x4 xa = feat4_fix<XV>(ra, 16 * q + 4 * g, a.Kvalid), xb = feat4_fix<XV>(rb, 16 * q + 8 + 4 * g, a.Kvalid);
            f32x4 xp[3];
            split3(xa, xb, xp[0], xp[1], xp[2]);
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                acc[nb] = mfma_s3(w[nb], xp, acc[nb]);
                const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) w[nb][pc] = Wp[(((long)n * Q + qn) * 3 + pc) * 64 + lane];       // the next chunk's words, behind this tile's MFMAs
            }
            ra = ran; rb = rbn;
        }
        rowgemm_epilogue<NTB>(a, acc, n0, row, rowc, g);
    }
}

// workgroup barrier for an LDS hand-off: this wave's LDS traffic is done, vector-memory requests stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS-DMA of three consecutive KiB (global_load_lds_dwordx4: 16 bytes per lane from uniform base + per-lane byte offset to LDS byte
// address M0 + 16 lane; the immediate offset moves source and destination alike -- as in k_gru_cm).  M0 is written and restored here.
__device__ __forceinline__ void glds3(const void* sbase, unsigned voff, unsigned ldsdst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsdst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Many rows: a workgroup of four waves owns 256 rows (two row tiles per wave) x four output tiles and shares the weight words of a
// chunk through LDS.  In k_rowgemm_x3 every wave pulls every weight word through the CU's vector L1 itself: 12 KB per 24 MFMAs,
// measured (TCP_TOTAL_CACHE_ACCESSES / TCP_TCC_READ_REQ on the data-gradient GEMMs of an 8 192-window training step) 427 M line
// accesses per launch, three quarters of them weight words -- 0.7 ms of the 1.3 ms launch at the L1's 64 bytes per clock.  Here
// the chunk's 12 KB are loaded once per workgroup (coalesced, double buffered in LDS, one barrier per chunk) and each word a wave
// reads from LDS feeds two MFMAs: 1.32 -> 0.95 ms per launch.  (Taking the rows through a wave-private LDS tile as well -- 32
// features = 128 contiguous bytes per row, every line read once -- was built and is no faster: 0.99 ms with 27 spilled registers;
// what is left is the one-chunk prefetch distance of both operands.)  Rows with a stride of whole 16-byte words only.  Same
// chunk order and terms per output element: results are bit-identical to k_rowgemm_x3.
__global__ __launch_bounds__(256, 2) void k_rowgemm_x3s(const RowGemmArgs a) {
    constexpr bool XV = true;
    constexpr int NTB = 4, WWORDS = NTB * 3 * 64;         // 16-byte words of a chunk: [tile][piece][lane]
    __shared__ __attribute__((aligned(16))) f32x4 wsh[3][WWORDS];      // ring of three chunks, filled by LDS-DMA
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    long row[2], rowc[2];
    const float* __restrict__ xrow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        row[t] = (long)blockIdx.x * 256 + 64 * wave + 32 * t + i;
        rowc[t] = row[t] < a.R ? row[t] : a.R - 1;
        xrow[t] = a.X + rowc[t] * a.ldx;
    }
    const f32x4* __restrict__ Wp = a.Wp3;
    const int Q = a.Q16;
    for (int n0 = blockIdx.y * NTB; n0 < a.NT; n0 += NTB * gridDim.y) {
        // Round 6 (DESIGN section 8.4 of round 5): both operands two chunks ahead.  The weight words of chunk q + 2 travel by LDS-DMA
        // into a ring of three slots -- wave w brings tile w's three KiB, no staging registers, nothing for the compiler's vmcnt
        // accounting to wait on -- the rows' raw words of chunk q + 2 into registers; the chunk barrier is s_waitcnt lgkmcnt(0) +
        // s_barrier (__syncthreads() also drains vmcnt: it waited for the requests issued at the top of the same iteration, so
        // one chunk of MFMAs, ~1.5 k cycles, had to cover a loaded HBM round trip).  A wave's own requests retire in order:
        // vmcnt(7) at the end of iteration q leaves only the 3 + 4 requests of chunk q + 2 outstanding, i.e. chunk q + 1 has landed.
        const f32x4* __restrict__ wtile = Wp + ((long)((n0 + wave < a.NT) ? n0 + wave : a.NT - 1) * Q) * 192;      // this wave's tile: 192 words per chunk
        const unsigned wlds = (unsigned)(uintptr_t)&wsh[0][0] + (unsigned)wave * 3072u;
        const unsigned voff = (unsigned)lane << 4;
        f32x16 acc[2][NTB];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][nb][r] = 0.f;
        f32x4 ra[2], rb[2], ra1[2], rb1[2];
        const int q1 = Q > 1 ? 1 : 0;
        lds_barrier();                                   // (a previous pass's readers are done with the ring)
        glds3(wtile, voff, wlds);
#pragma unroll
        for (int t = 0; t < 2; ++t) { ra[t] = feat4_raw<XV>(xrow[t], 4 * g, a.Kvalid); rb[t] = feat4_raw<XV>(xrow[t], 8 + 4 * g, a.Kvalid); }
        glds3(wtile + (long)q1 * 192, voff, wlds + (unsigned)(WWORDS * 16));
#pragma unroll
        for (int t = 0; t < 2; ++t) { ra1[t] = feat4_raw<XV>(xrow[t], 16 * q1 + 4 * g, a.Kvalid); rb1[t] = feat4_raw<XV>(xrow[t], 16 * q1 + 8 + 4 * g, a.Kvalid); }
        wait_vmcnt<7>();
        lds_barrier();
        int slot = 0;
        for (int q = 0; q < Q; ++q) {
            const int qnn = (q + 2 < Q) ? q + 2 : Q - 1;
            const int slot2 = slot == 0 ? 2 : slot - 1;                                   // (q + 2) % 3
            glds3(wtile + (long)qnn * 192, voff, wlds + (unsigned)slot2 * (unsigned)(WWORDS * 16));
            f32x4 ran[2], rbn[2], xp[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ran[t] = feat4_raw<XV>(xrow[t], 16 * qnn + 4 * g, a.Kvalid);
                rbn[t] = feat4_raw<XV>(xrow[t], 16 * qnn + 8 + 4 * g, a.Kvalid);
                split3(feat4_fix<XV>(ra[t], 16 * q + 4 * g, a.Kvalid), feat4_fix<XV>(rb[t], 16 * q + 8 + 4 * g, a.Kvalid), xp[t][0], xp[t][1], xp[t][2]);
            }
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                f32x4 w[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) w[pc] = wsh[slot][(nb * 3 + pc) * 64 + lane];
                acc[0][nb] = mfma_s3(w, xp[0], acc[0][nb]);
                acc[1][nb] = mfma_s3(w, xp[1], acc[1][nb]);
            }
            wait_vmcnt<7>();
            lds_barrier();
            slot = slot == 2 ? 0 : slot + 1;
#pragma unroll
            for (int t = 0; t < 2; ++t) { ra[t] = ra1[t]; rb[t] = rb1[t]; ra1[t] = ran[t]; rb1[t] = rbn[t]; }
        }
        wait_vmcnt<0>();
        rowgemm_epilogue<NTB>(a, acc[0], n0, row[0], rowc[0], g);
        rowgemm_epilogue<NTB>(a, acc[1], n0, row[1], rowc[1], g);
    }
}

// ---------------------------------------------------------------------------
// conv: xc[b,t,o] = ReLU(bias[o] + sum_{i,j} w[o,i,j] * x[b, t+j-pad, i]), zero
// outside the window.  reference ConvLayer.forward, modules.py:18-22.
// Implicit GEMM: data rows = (window, t), K = taps x Fp, loaded straight from x.
// Writes xc (b*W, Fp), its transpose xcT (b*F, Wp) and columns [0,F) of h_cat.
// ---------------------------------------------------------------------------
template <int NTB>
__global__ __launch_bounds__(64) void k_conv(const ConvArgs a) {
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    const long row = (long)blockIdx.x * 32 + i;
    const long R = a.B * a.W;
    const long rowc = row < R ? row : R - 1;
    const long win = rowc / a.W;
    const int t = (int)(rowc - win * a.W);
    // gather mode: the window is a view of the device-resident series (SlidingWindowDataset.__getitem__, utils.py:114-117)
    const float* __restrict__ xwin = a.gather ? a.X + (a.starts ? a.starts[win] : a.start0 + win * a.stride) * (long)a.F
                                              : a.X + win * (long)a.W * a.F;
    const int QF = a.Fp >> 3;
    const int Q = a.taps * QF;
    const f32x4* __restrict__ Wp = a.Wp;

    if (a.HCAT && row < R && g == 0)      // zero the alignment padding of the h_cat row (the GRU reads it unguarded)
        for (int c = 3 * a.F; c < a.Dp; ++c) a.HCAT[row * a.Dp + c] = 0.f;
    // unconditional loads from clamped addresses, masked afterwards (a guarded load is a branch with a full memory
    // round trip per chunk); one 16-byte load when the rows allow it
    const bool xvec4 = (a.F & 3) == 0;
    auto loadx = [&](int q) -> f32x4 {
        const int tap = q / QF;
        const int cb = q - tap * QF;
        const int tt = t + tap - a.pad;
        const int c0 = 8 * cb + 4 * g;
        const bool tok = tt >= 0 && tt < a.W;
        const int ttc = tt < 0 ? 0 : (tt < a.W ? tt : a.W - 1);
        const float* p = xwin + (long)ttc * a.F;
        f32x4 v;
        if (xvec4) {
            const int cc = c0 + 3 < a.F ? c0 : a.F - 4;
            v = *reinterpret_cast<const f32x4*>(p + cc);
        } else {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) v[s4] = p[c0 + s4 < a.F ? c0 + s4 : a.F - 1];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) v[s4] = (tok && c0 + s4 < a.F) ? v[s4] : 0.f;
        return v;
    };

    for (int n0 = 0; n0 < a.NT; n0 += NTB) {
        f32x16 acc[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

        f32x4 xv = loadx(0);
        f32x4 w[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
            w[nb] = Wp[((long)n * Q) * 64 + lane];
        }
        for (int q = 0; q < Q; ++q) {
            const int qn = (q + 1 < Q) ? q + 1 : q;
            const f32x4 xn = loadx(qn);
            f32x4 wn[NTB];
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
                wn[nb] = Wp[((long)n * Q + qn) * 64 + lane];
            }
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) acc[nb] = mfma4(w[nb], xv, acc[nb]);
            xv = xn;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) w[nb] = wn[nb];
        }
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            if (n0 + nb >= a.NT) break;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * (n0 + nb) + 8 * m + 4 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int o = col + s;
                    const float v = fmaxf(acc[nb][4 * m + s] + bv[s], 0.f);
                    if (row < R && o < a.F) {
                        if (a.XC) a.XC[row * a.Fp + o] = v;
                        if (a.XCT) a.XCT[(win * a.F + o) * (long)a.Wpad + t] = v;
                        if (a.HCAT) a.HCAT[row * a.Dp + o] = v;
                        if (a.Y) a.Y[row * a.F + o] = v;
                    }
                }
            }
        }
    }
}

// conv straight from memory on split-bf16 operands (wide models, whose rows do not fit the LDS-staged kernels): the K loop
// in 16-channel chunks of one tap, the eight input values a lane holds of a chunk split into three bf16 pieces (split3), the
// weights pre-split on the device ([tile][taps Fq / 16][3][64], Fq = F rounded up to 16), six v_mfma_f32_32x32x16_bf16 per
// chunk and tile (mfma_s3) instead of eight v_mfma_f32_32x32x2_f32: 2.7 x less matrix time, products of 24-bit significands.
template <int NTB>
__global__ __launch_bounds__(64) void k_conv_x3(const ConvArgs a) {
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    const long row = (long)blockIdx.x * 32 + i;
    const long R = a.B * a.W;
    const long rowc = row < R ? row : R - 1;
    const long win = rowc / a.W;
    const int t = (int)(rowc - win * a.W);
    const float* __restrict__ xwin = a.gather ? a.X + (a.starts ? a.starts[win] : a.start0 + win * a.stride) * (long)a.F
                                              : a.X + win * (long)a.W * a.F;
    const int QF = a.Fq >> 4;
    const int Q = a.taps * QF;
    const f32x4* __restrict__ Wp = a.Wp3;
    if (a.HCAT && row < R && g == 0)
        for (int c = 3 * a.F; c < a.Dp; ++c) a.HCAT[row * a.Dp + c] = 0.f;
    const bool xvec4 = (a.F & 3) == 0;
    auto loadx = [&](int q, int half) -> f32x4 {      // channels 16 cb + 8 half + 4 g .. + 3 of input row t + tap - pad
        const int tap = q / QF;
        const int cb = q - tap * QF;
        const int tt = t + tap - a.pad;
        const int c0 = 16 * cb + 8 * half + 4 * g;
        const bool tok = tt >= 0 && tt < a.W;
        const int ttc = tt < 0 ? 0 : (tt < a.W ? tt : a.W - 1);
        const float* p = xwin + (long)ttc * a.F;
        f32x4 v;
        if (xvec4) {
            const int cc = c0 + 3 < a.F ? c0 : a.F - 4;
            v = *reinterpret_cast<const f32x4*>(p + cc);
        } else {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) v[s4] = p[c0 + s4 < a.F ? c0 + s4 : a.F - 1];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) v[s4] = (tok && c0 + s4 < a.F) ? v[s4] : 0.f;
        return v;
    };
    for (int n0 = 0; n0 < a.NT; n0 += NTB) {
        f32x16 acc[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        f32x4 xa = loadx(0, 0), xb = loadx(0, 1);
        f32x4 w[NTB][3];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) w[nb][pc] = Wp[(((long)n * Q) * 3 + pc) * 64 + lane];
        }
        for (int q = 0; q < Q; ++q) {
            const int qn = (q + 1 < Q) ? q + 1 : q;
            const f32x4 xan = loadx(qn, 0), xbn = loadx(qn, 1);
            f32x4 xp[3];
            split3(xa, xb, xp[0], xp[1], xp[2]);
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                acc[nb] = mfma_s3(w[nb], xp, acc[nb]);
                const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) w[nb][pc] = Wp[(((long)n * Q + qn) * 3 + pc) * 64 + lane];
            }
            xa = xan; xb = xbn;
        }
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            if (n0 + nb >= a.NT) break;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * (n0 + nb) + 8 * m + 4 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int o = col + s;
                    const float v = fmaxf(acc[nb][4 * m + s] + bv[s], 0.f);
                    if (row < R && o < a.F) {
                        if (a.XC) a.XC[row * a.Fp + o] = v;
                        if (a.XCT) a.XCT[(win * a.F + o) * (long)a.Wpad + t] = v;
                        if (a.HCAT) a.HCAT[row * a.Dp + o] = v;
                        if (a.Y) a.Y[row * a.F + o] = v;
                    }
                }
            }
        }
    }
}

// k_conv_x3 for many rows, as k_rowgemm_x3s: a workgroup of four waves owns 256 output rows (two row tiles per wave) x four
// output tiles, the weight words of a chunk go through LDS once per workgroup and feed two MFMAs per LDS read.  F a multiple of 4.
// Same chunk order and terms per output element: results are bit-identical to k_conv_x3.
__global__ __launch_bounds__(256, 2) void k_conv_x3s(const ConvArgs a) {
    constexpr int NTB = 4, WWORDS = NTB * 3 * 64;
    __shared__ __attribute__((aligned(16))) f32x4 wsh[3][WWORDS];      // ring of three chunks, filled by LDS-DMA (k_rowgemm_x3s)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const long R = a.B * a.W;
    long row[2], win[2];
    int t[2];
    const float* __restrict__ xwin[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        row[rt] = (long)blockIdx.x * 256 + 64 * wave + 32 * rt + i;
        const long rowc = row[rt] < R ? row[rt] : R - 1;
        win[rt] = rowc / a.W;
        t[rt] = (int)(rowc - win[rt] * a.W);
        xwin[rt] = a.gather ? a.X + (a.starts ? a.starts[win[rt]] : a.start0 + win[rt] * a.stride) * (long)a.F
                            : a.X + win[rt] * (long)a.W * a.F;
        if (a.HCAT && row[rt] < R && g == 0)
            for (int c = 3 * a.F; c < a.Dp; ++c) a.HCAT[row[rt] * a.Dp + c] = 0.f;
    }
    const int QF = a.Fq >> 4;
    const int Q = a.taps * QF;
    const f32x4* __restrict__ Wp = a.Wp3;
    // channels 16 cb + 8 half + 4 g .. + 3 of input row t + tap - pad: the raw 16 bytes (clamped address), masked where they are used
    auto rawx = [&](int rt, int tap, int cb, int half) -> f32x4 {
        const int tt = t[rt] + tap - a.pad;
        const int c0 = 16 * cb + 8 * half + 4 * g;
        const int ttc = tt < 0 ? 0 : (tt < a.W ? tt : a.W - 1);
        return *reinterpret_cast<const f32x4*>(xwin[rt] + (long)ttc * a.F + (c0 + 3 < a.F ? c0 : a.F - 4));
    };
    auto fixx = [&](f32x4 v, int rt, int tap, int cb, int half) -> f32x4 {
        const int tt = t[rt] + tap - a.pad;
        const int c0 = 16 * cb + 8 * half + 4 * g;
        const bool ok = tt >= 0 && tt < a.W && c0 + 3 < a.F;       // F % 4 == 0: a group is whole or beyond the row
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) v[s4] = ok ? v[s4] : 0.f;
        return v;
    };
    for (int n0 = blockIdx.y * NTB; n0 < a.NT; n0 += NTB * gridDim.y) {
        // (round 6, as k_rowgemm_x3s: weight words two chunks ahead by LDS-DMA into a ring of three slots, input words two chunks
        // ahead in registers, chunk barrier without the vector-memory drain)
        const f32x4* __restrict__ wtile = Wp + ((long)((n0 + wave < a.NT) ? n0 + wave : a.NT - 1) * Q) * 192;
        const unsigned wlds = (unsigned)(uintptr_t)&wsh[0][0] + (unsigned)wave * 3072u;
        const unsigned voff = (unsigned)lane << 4;
        f32x16 acc[2][NTB];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][nb][r] = 0.f;
        f32x4 ra[2], rb[2], ra1[2], rb1[2];
        int tap = 0, cb = 0;                              // chunk q = tap * QF + cb
        int tap1 = 0, cb1 = Q > 1 ? 1 : 0;                // chunk q + 1
        if (cb1 == QF) { cb1 = 0; tap1 = 1; }
        lds_barrier();                                    // (a previous pass's readers are done with the ring)
        glds3(wtile, voff, wlds);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) { ra[rt] = rawx(rt, 0, 0, 0); rb[rt] = rawx(rt, 0, 0, 1); }
        glds3(wtile + (long)(Q > 1 ? 1 : 0) * 192, voff, wlds + (unsigned)(WWORDS * 16));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) { ra1[rt] = rawx(rt, tap1, cb1, 0); rb1[rt] = rawx(rt, tap1, cb1, 1); }
        wait_vmcnt<7>();
        lds_barrier();
        int slot = 0;
        for (int q = 0; q < Q; ++q) {
            int tap2 = tap1, cb2 = cb1 + 1;               // chunk q + 2 (clamped to the last one)
            if (cb2 == QF) { cb2 = 0; ++tap2; }
            if (q + 2 >= Q) { tap2 = tap1; cb2 = cb1; }
            const int qnn = (q + 2 < Q) ? q + 2 : Q - 1;
            const int slot2 = slot == 0 ? 2 : slot - 1;
            glds3(wtile + (long)qnn * 192, voff, wlds + (unsigned)slot2 * (unsigned)(WWORDS * 16));
            f32x4 ran[2], rbn[2], xp[2][3];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                ran[rt] = rawx(rt, tap2, cb2, 0);
                rbn[rt] = rawx(rt, tap2, cb2, 1);
                split3(fixx(ra[rt], rt, tap, cb, 0), fixx(rb[rt], rt, tap, cb, 1), xp[rt][0], xp[rt][1], xp[rt][2]);
            }
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                f32x4 w[3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) w[pc] = wsh[slot][(nb * 3 + pc) * 64 + lane];
                acc[0][nb] = mfma_s3(w, xp[0], acc[0][nb]);
                acc[1][nb] = mfma_s3(w, xp[1], acc[1][nb]);
            }
            wait_vmcnt<7>();
            lds_barrier();
            slot = slot == 2 ? 0 : slot + 1;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { ra[rt] = ra1[rt]; rb[rt] = rb1[rt]; ra1[rt] = ran[rt]; rb1[rt] = rbn[rt]; }
            tap = tap1; cb = cb1; tap1 = tap2; cb1 = cb2;
        }
        wait_vmcnt<0>();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                if (n0 + nb >= a.NT) break;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int col = 32 * (n0 + nb) + 8 * m + 4 * g;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int o = col + s4;
                        const float v = fmaxf(acc[rt][nb][4 * m + s4] + bv[s4], 0.f);
                        if (row[rt] < R && o < a.F) {
                            if (a.XC) a.XC[row[rt] * a.Fp + o] = v;
                            if (a.XCT) a.XCT[(win[rt] * a.F + o) * (long)a.Wpad + t[rt]] = v;
                            if (a.HCAT) a.HCAT[row[rt] * a.Dp + o] = v;
                            if (a.Y) a.Y[row[rt] * a.F + o] = v;
                        }
                    }
                }
            }
    }
}

// conv, LDS-staged variant (used when 32+taps-1 input rows of Fp floats fit a wave's LDS budget):
// the wave copies the input rows its 32 output rows touch into LDS once with coalesced loads and
// takes every MFMA B operand from there -- the straight-from-global version above re-reads each
// input row `taps` times with 4-byte gathers (PMC: 5-9x FETCH amplification, TA-bound).
// BF: bf16 operand build -- 16 input channels per chunk (K index = tap * Fq + channel, Fq = F rounded up to 16),
// the staged fp32 rows are converted on the way into the matrix unit, fp32 accumulation.
// X3 (with the bf16 chunk geometry): split-bf16 operands -- three weight pieces per chunk, the staged fp32 input split
// where it is consumed, six bf16 MFMAs per chunk and tile (mtadgat_device.h)
template <int NTB, bool BF = false, bool X3 = false>
__global__ __launch_bounds__(256) void k_conv_lds(const ConvArgs a) {
    // blockDim.x / 64 independent waves per workgroup, each with its own 32 output rows and LDS slice:
    // one-wave workgroups are launched too slowly to keep the matrix pipes fed at ~50 us per wave
    extern __shared__ __attribute__((aligned(16))) float xs_all[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    float* __restrict__ xs = xs_all + wv * ((32 + a.taps - 1) * (a.Fq + 4));
    const int i = lane & 31, g = lane >> 5;
    const long R = a.B * a.W;
    const long r0 = ((long)blockIdx.x * nwv + wv) * 32;
    const long row = r0 + i;
    const long rowc = row < R ? row : R - 1;
    const long win = rowc / a.W;
    const int t = (int)(rowc - win * a.W);
    const int Fld = a.Fq + 4;
    const int nrow = 32 + a.taps - 1;
    // source row of LDS row rr: flat row r0 - pad + rr of the (B*W, F) batch; in gather mode the windows
    // are views of the device-resident series (no (b, W, F) copy exists) and (window, t) advance together
    long gw = 0; int gt = 0;                          // window / position of the current flat row (gather mode)
    if (a.gather) {
        const long f0 = r0 - a.pad < 0 ? 0 : r0 - a.pad;
        gw = f0 / a.W;
        gt = (int)(f0 - gw * a.W);
    }
    // element offset of the source row (the input is fp32, or -- x_bf16 -- bfloat16 read directly: no cast pass over x)
    auto src_row = [&](int rr, bool& ok) -> long {
        const long flat = r0 - a.pad + rr;
        ok = flat >= 0 && flat < R;
        long srow = ok ? flat : 0;
        if (a.gather) {
            const long wc = gw < a.B ? gw : a.B - 1;
            const long s0 = a.starts ? a.starts[wc] : a.start0 + wc * a.stride;
            srow = s0 + gt;
            if (flat >= 0 && ++gt == a.W) { gt = 0; ++gw; }      // rows are visited in increasing order, once each
        }
        return srow * a.F;
    };
    const unsigned short* __restrict__ Xh = reinterpret_cast<const unsigned short*>(a.X);
    auto xload = [&](long off) -> float {
        if (a.x_bf16) return __builtin_bit_cast(float, (unsigned)Xh[off] << 16);
        return a.X[off];
    };
    if (a.F <= 64 && Fld <= 128) {
        // one element per lane and row; every load is unconditional (clamped column, a valid row for rows
        // outside the batch) and all of them are issued before the first LDS store: one memory round trip
        // per wave instead of one per row (a guarded load compiles to a branch + s_waitcnt vmcnt(0))
        // (round 6: the rows' source offsets first -- wave-uniform, with the gather mode's `starts` loads and position updates --, then
        // nothing but the loads, in one loop per input type: with the offset arithmetic, the gather branch and the bf16 / fp32
        // choice inside the load loop every row was a control-flow region of its own with s_waitcnt vmcnt(0) at its join, 38 serial
        // memory round trips per wave -- the whole 38 - 51 us of this kernel at 256 windows)
        constexpr int MAXR = 40;
        float v[MAXR];
        long srcs[MAXR];
        bool oks[MAXR];
        const int colc = lane < a.F ? lane : a.F - 1;
#pragma unroll
        for (int rr = 0; rr < MAXR; ++rr) {
            srcs[rr] = 0; oks[rr] = false;
            if (rr < nrow) srcs[rr] = src_row(rr, oks[rr]);       // wave-uniform
        }
        if (a.x_bf16) {
#pragma unroll
            for (int rr = 0; rr < MAXR; ++rr) v[rr] = __builtin_bit_cast(float, (unsigned)Xh[srcs[rr] + colc] << 16);
        } else {
#pragma unroll
            for (int rr = 0; rr < MAXR; ++rr) v[rr] = a.X[srcs[rr] + colc];
        }
#pragma unroll
        for (int rr = 0; rr < MAXR; ++rr) v[rr] = (oks[rr] && lane < a.F) ? v[rr] : 0.f;
#pragma unroll
        for (int rr = 0; rr < MAXR; ++rr)
            if (rr < nrow) {
                if (lane < Fld) xs[rr * Fld + lane] = v[rr];
                if (lane + 64 < Fld) xs[rr * Fld + lane + 64] = 0.f;       // channel padding beyond 64 columns (bf16 build)
            }
        for (int rr = MAXR; rr < nrow; ++rr) {
            bool ok;
            const long src = src_row(rr, ok);
            if (lane < Fld) xs[rr * Fld + lane] = (ok && lane < a.F) ? xload(src + lane) : 0.f;
            if (lane + 64 < Fld) xs[rr * Fld + lane + 64] = 0.f;
        }
    } else {
        for (int rr = 0; rr < nrow; ++rr) {
            bool ok;
            const long src = src_row(rr, ok);
            for (int col = lane; col < Fld; col += 64) xs[rr * Fld + col] = (ok && col < a.F) ? xload(src + col) : 0.f;
        }
    }
    __syncthreads();
    const int QF = BF ? a.Fq >> 4 : a.Fq >> 3;
    const int Q = a.taps * QF;
    const f32x4* __restrict__ Wp = a.Wp;
    if (a.HCAT && row < R && g == 0)      // zero the alignment padding of the h_cat row (the GRU reads it unguarded)
        for (int c = 3 * a.F; c < a.Dp; ++c) a.HCAT[row * a.Dp + c] = 0.f;
    // B operand of chunk (tap, cb): input row i + tap of the staged block, columns 8 cb + 4 g .. +3 (bf16 build: the
    // lane's eight columns of the 16-channel chunk, converted); rows of the neighbouring window count as zero
    // padding (per window, modules.py:14,20)
    const float* __restrict__ xrow = static_cast<const float*>(__builtin_assume_aligned(xs, 16)) + i * Fld + 4 * g;
    constexpr int NP = X3 ? 3 : 1;
    auto loadx3 = [&](int tap, int cb, f32x4 (&xs)[3]) {
        const int tt = t + tap - a.pad;
        const bool ok = tt >= 0 && tt < a.W;
        f32x4 lo = *reinterpret_cast<const f32x4*>(xrow + tap * Fld + 16 * cb);
        f32x4 hi = *reinterpret_cast<const f32x4*>(xrow + tap * Fld + 16 * cb + 8);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { lo[s4] = ok ? lo[s4] : 0.f; hi[s4] = ok ? hi[s4] : 0.f; }
        split3(lo, hi, xs[0], xs[1], xs[2]);
    };
    auto loadx = [&](int tap, int cb) -> f32x4 {
        const int tt = t + tap - a.pad;
        const bool ok = tt >= 0 && tt < a.W;
        if (BF) {
            f32x4 lo = *reinterpret_cast<const f32x4*>(xrow + tap * Fld + 16 * cb);
            f32x4 hi = *reinterpret_cast<const f32x4*>(xrow + tap * Fld + 16 * cb + 8);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { lo[s4] = ok ? lo[s4] : 0.f; hi[s4] = ok ? hi[s4] : 0.f; }
            return cvt8(lo, hi);
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(xrow + tap * Fld + 8 * cb);
        v[0] = ok ? v[0] : 0.f; v[1] = ok ? v[1] : 0.f; v[2] = ok ? v[2] : 0.f; v[3] = ok ? v[3] : 0.f;
        return v;
    };
    auto mm = [&](const f32x4 w, const f32x4 x, f32x16 acc) -> f32x16 { return BF ? mfma_bf(w, x, acc) : mfma4(w, x, acc); };
    float vmx = 0.f;                      // largest output of this lane (outputs are >= 0: ReLU)
    for (int n0 = 0; n0 < a.NT; n0 += NTB) {
        f32x16 acc[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const f32x4* wq[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
            wq[nb] = Wp + ((long)n * Q) * (64 * NP) + lane;
        }
        // two register sets in turn (no copies): the weights of chunk q + 1 are in flight during the MFMAs
        // of chunk q
        int tap = 0, cb = 0;
        if constexpr (X3) {
            // chunk = [piece][64 lanes] words; the pieces of chunk q + 1 are requested before the MFMAs of chunk q
            f32x4 wa[NTB][3], wb[NTB][3];
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) wa[nb][pc] = wq[nb][pc * 64];
            for (int q = 0; q < Q; ++q) {
                const int q1 = q + 1 < Q ? q + 1 : q;
#pragma unroll
                for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) wb[nb][pc] = wq[nb][((long)q1 * 3 + pc) * 64];
                f32x4 xs[3];
                loadx3(tap, cb, xs);
                if (++cb == QF) { cb = 0; ++tap; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NTB; ++nb) acc[nb] = mfma_s3(wa[nb], xs, acc[nb]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) wa[nb][pc] = wb[nb][pc];
            }
        }
        f32x4 w0[NTB], w1[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) w0[nb] = wq[nb][0];
        for (int q = 0; !X3 && q < Q; q += 2) {
            const int q1 = q + 1 < Q ? q + 1 : q;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) w1[nb] = wq[nb][(long)q1 * 64];
            const f32x4 xv0 = loadx(tap, cb);
            if (++cb == QF) { cb = 0; ++tap; }
            __builtin_amdgcn_sched_barrier(0);      // keep the next chunk's weight loads ahead of these MFMAs
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) acc[nb] = mm(w0[nb], xv0, acc[nb]);
            __builtin_amdgcn_sched_barrier(0);
            const int q2 = q + 2 < Q ? q + 2 : Q - 1;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) w0[nb] = wq[nb][(long)q2 * 64];
            if (q + 1 < Q) {
                const f32x4 xv1 = loadx(tap, cb);
                if (++cb == QF) { cb = 0; ++tap; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NTB; ++nb) acc[nb] = mm(w1[nb], xv1, acc[nb]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            if (n0 + nb >= a.NT) break;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * (n0 + nb) + 8 * m + 4 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
                f32x4 v;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) v[s4] = fmaxf(acc[nb][4 * m + s4] + bv[s4], 0.f);
                if (row < R) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) vmx = (col + s4 < a.F) ? fmaxf(vmx, v[s4]) : vmx;
                    if (a.HCAT) {   // rows of h_cat are 16-byte aligned: one store per 4 channels
                        float* hp = a.HCAT + row * a.Dp + col;
                        if (col + 3 < a.F) {
                            *reinterpret_cast<f32x4*>(hp) = v;
                        } else {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                if (col + s4 < a.F) hp[s4] = v[s4];
                        }
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int o = col + s4;
                        if (o < a.F) {
                            if (a.XC) a.XC[row * a.Fp + o] = v[s4];
                            if (a.XCT) a.XCT[(win * a.F + o) * (long)a.Wpad + t] = v[s4];
                            if (a.Y) a.Y[row * a.F + o] = v[s4];
                        }
                    }
                }
            }
        }
    }
    if (a.vmax) {
        // range of the node values for the attention layers' fp16 operand pieces: one atomic per wave at most, and only
        // while the maximum still grows
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmx = fmaxf(vmx, __shfl_xor(vmx, off));
        if (lane == 0 && !(vmx <= __uint_as_float(*a.vmax))) atomicMax(a.vmax, __float_as_uint(vmx));     // (NaN: recorded, disables the fp16 path)
    }
}

// small helper: copy a (R, ncols) row-major matrix into a padded (R, ld) one, or back.
__global__ void k_copy2d(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long R, int ncols) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = R * ncols;
    if (idx < total) {
        const long r = idx / ncols;
        const int c = (int)(idx - r * ncols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}
// Stride-1 windows of one series share their convolution rows (reference quirk: the zero padding is
// per WINDOW, modules.py:14,20): row t of window w equals the row s_w + t of the convolution over the whole segment
// unless a tap leaves the window, i.e. for the first / last `pad` rows.  CF: convolution of the segment as one long
// window; EL / ER: of the windows' first / last EW = 2 pad rows as EW-row windows (their first / last pad rows are the
// edge rows).  This kernel places the rows into h_cat[:, :F] and zeroes the row's alignment padding.
__global__ void k_conv_scatter(const float* __restrict__ CF, const float* __restrict__ EL, const float* __restrict__ ER, float* __restrict__ HCAT,
                               long n, int W, int F, int Fp, int Dp, int pad, int EW) {
    const int G4 = (F + 3) >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * W * G4) return;
    const long row = idx / G4;
    const int c = (int)(idx - row * G4) * 4;
    const long w = row / W;
    const int t = (int)(row - w * W);
    const float* __restrict__ src = t < pad ? EL + (w * EW + t) * (long)Fp
                                  : (t >= W - pad ? ER + (w * EW + (t - (W - EW))) * (long)Fp : CF + (w + t) * (long)Fp);
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);          // rows are Fp floats apart (F rounded up to 8): aligned, in range
    float* __restrict__ dst = HCAT + row * Dp + c;
    if (c + 3 < F) {
        *reinterpret_cast<f32x4*>(dst) = v;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < F) dst[e] = v[e];
    }
    if (c == 0)
        for (int k = 3 * F; k < Dp; ++k) HCAT[row * Dp + k] = 0.f;
}
// transpose per window: src (B, R, C) with row stride lds -> dst (B, C, R) with row stride ldd
__global__ void k_transpose_win(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long B, int R,
                                int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = B * R * C;
    if (idx < total) {
        const long b = idx / ((long)R * C);
        const long rem = idx - b * (long)R * C;
        const int r = (int)(rem / C);
        const int c = (int)(rem - (long)r * C);
        dst[(b * C + c) * ldd + r] = src[(b * R + r) * lds + c];
    }
}
static int g_gemm_lds_off = getenv("MTADGAT_ROWGEMM_NOLDS") ? 1 : 0;
void set_gemm_lds_off(int off) { g_gemm_lds_off = off; }

template <bool XV>
static void launch_rowgemm_xv(const RowGemmArgs& a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.R + 31) / 32);
    const long groups4 = (a.NT + 3) / 4;
    if (a.x3) {
        const int s_off = g_gemm_lds_off;
        if (XV && a.NT >= 4 && a.R >= 64 * 2048 && !s_off) {
            // plenty of rows of whole 16-byte words: 256 per workgroup, the chunk's weight words through LDS (k_rowgemm_x3s)
            const unsigned gridw = (unsigned)((a.R + 255) / 256);
            const long want = (2048 + gridw - 1) / gridw;
            const unsigned split = (unsigned)(want < 1 ? 1 : (want > groups4 ? groups4 : want));
            hipLaunchKernelGGL(k_rowgemm_x3s, dim3(gridw, split), dim3(256), 0, s, a);
        } else if (a.NT >= 4) {
            const long want = (4096 + grid - 1) / grid;
            const unsigned split = (unsigned)(want < 1 ? 1 : (want > groups4 ? groups4 : want));
            hipLaunchKernelGGL((k_rowgemm_x3<4, XV>), dim3(grid, split), dim3(64), 0, s, a);
        } else if (a.NT >= 2)
            hipLaunchKernelGGL((k_rowgemm_x3<2, XV>), dim3(grid), dim3(64), 0, s, a);
        else
            hipLaunchKernelGGL((k_rowgemm_x3<1, XV>), dim3(grid), dim3(64), 0, s, a);
        return;
    }
    if (a.NT >= 2 && (long)grid * groups4 < 1024) {
        // a latency chain on a few waves (a head Linear on 256 rows): one output tile per wave
        hipLaunchKernelGGL((k_rowgemm<1, XV>), dim3(grid, (unsigned)a.NT), dim3(64), 0, s, a);
    } else if (a.NT >= 4 && (long)grid * groups4 < 8192) {
        // mid-size launches (the GRU input projection of a 256-window batch: 800 row blocks x 15 tiles): two tiles per wave (108
        // registers, four waves per SIMD, instead of 172 and two) spread over gridDim.y -- 256-window forward 0.456 -> 0.429 ms
        const long groups2 = (a.NT + 1) / 2;
        hipLaunchKernelGGL((k_rowgemm<2, XV>), dim3(grid, (unsigned)groups2), dim3(64), 0, s, a);
    } else if (a.NT >= 4) {
        const long want = (4096 + grid - 1) / grid;                 // ~4 waves per SIMD
        const unsigned split = (unsigned)(want < 1 ? 1 : (want > groups4 ? groups4 : want));
        hipLaunchKernelGGL((k_rowgemm<4, XV>), dim3(grid, split), dim3(64), 0, s, a);
    } else if (a.NT >= 2)
        hipLaunchKernelGGL((k_rowgemm<2, XV>), dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((k_rowgemm<1, XV>), dim3(grid), dim3(64), 0, s, a);
}
int launch_rowgemm(const RowGemmArgs& a, hipStream_t s) {
    if (a.R <= 0) return 0;
    if ((a.ldx & 3) == 0 && a.Kvalid >= 4)
        launch_rowgemm_xv<true>(a, s);
    else
        launch_rowgemm_xv<false>(a, s);
    LAUNCH_CHECK();
    return 0;
}

int launch_conv(const ConvArgs& a, hipStream_t s) {
    const long R = a.B * a.W;
    if (R <= 0) return 0;
    const unsigned grid = (unsigned)((R + 31) / 32);
    const size_t lds = (size_t)(32 + a.taps - 1) * (a.Fq + 4) * sizeof(float);
    if ((a.bf16 || a.x_bf16) && lds > 20 * 1024) return -2;
    constexpr size_t lds_max = 20 * 1024;
    if (lds <= lds_max || a.bf16 || a.x_bf16) {       // >= 8 waves per CU keep their tile in LDS
        if (lds > 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_lds<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_lds<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        const unsigned wpb = (grid >= 4096 && 4 * lds <= 64 * 1024) ? 4 : 1;     // waves per workgroup
        const unsigned g4 = (grid + wpb - 1) / wpb;
        if (a.bf16) {
            if (a.NT >= 2)
                hipLaunchKernelGGL((k_conv_lds<2, true>), dim3(g4), dim3(64 * wpb), wpb * lds, s, a);
            else
                hipLaunchKernelGGL((k_conv_lds<1, true>), dim3(g4), dim3(64 * wpb), wpb * lds, s, a);
        } else if (a.NT >= 2)
            hipLaunchKernelGGL((k_conv_lds<2, false>), dim3(g4), dim3(64 * wpb), wpb * lds, s, a);
        else
            hipLaunchKernelGGL((k_conv_lds<1, false>), dim3(g4), dim3(64 * wpb), wpb * lds, s, a);
    } else {
        // the straight-from-memory kernel does not record the output range: mark it unknown (a NaN pattern)
        if (a.vmax && hipMemsetAsync(a.vmax, 0xFF, sizeof(unsigned), s) != hipSuccess) return -3;
        if (a.Wp3) {                 // split-bf16 operands (precision mode 2, large launches: run_conv)
            const int s_off = g_gemm_lds_off;
            if (a.NT >= 4 && R >= 64 * 2048 && (a.F & 3) == 0 && a.F >= 4 && !s_off)
                {
                    // groups of four output tiles spread over gridDim.y until the launch has >= 2 048 workgroups: a chunk of config 4
                    // (229 376 rows) was 896 workgroups of four passes each on 512 slots -- 1.75 rounds of very long workgroups
                    const unsigned gridw = (unsigned)((R + 255) / 256);
                    const long groups4 = (a.NT + 3) / 4, want = (2048 + gridw - 1) / gridw;
                    const unsigned split = (unsigned)(want < 1 ? 1 : (want > groups4 ? groups4 : want));
                    hipLaunchKernelGGL(k_conv_x3s, dim3(gridw, split), dim3(256), 0, s, a);
                }
            else if (a.NT >= 4)
                hipLaunchKernelGGL(k_conv_x3<4>, dim3(grid), dim3(64), 0, s, a);
            else if (a.NT >= 2)
                hipLaunchKernelGGL(k_conv_x3<2>, dim3(grid), dim3(64), 0, s, a);
            else
                hipLaunchKernelGGL(k_conv_x3<1>, dim3(grid), dim3(64), 0, s, a);
        } else if (a.NT >= 2)
            hipLaunchKernelGGL(k_conv<2>, dim3(grid), dim3(64), 0, s, a);
        else
            hipLaunchKernelGGL(k_conv<1>, dim3(grid), dim3(64), 0, s, a);
    }
    LAUNCH_CHECK();
    return 0;
}

int launch_copy2d(const float* src, long lds, float* dst, long ldd, long R, int ncols, hipStream_t s) {
    const long total = R * ncols;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, lds, dst, ldd, R, ncols);
    LAUNCH_CHECK();
    return 0;
}
int launch_conv_scatter(const float* cf, const float* el, const float* er, float* hcat, long n, int W, int F, int Fp, int Dp, int pad,
                        hipStream_t s) {
    const long total = n * W * ((F + 3) >> 2);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_conv_scatter, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cf, el, er, hcat, n, W, F, Fp, Dp, pad, 2 * pad);
    LAUNCH_CHECK();
    return 0;
}
int launch_transpose_win(const float* src, long lds, float* dst, long ldd, long B, int R, int C, hipStream_t s) {
    const long total = B * R * C;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_transpose_win, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, lds, dst, ldd, B, R,
                       C);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
