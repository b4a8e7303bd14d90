// K2 — aggregation + channel expansion + BN + residual + ReLU of unit_agcn
// (model/unit_agcn.py:87-93), eval mode with folded BatchNorm:
//
//   u_s[k,t,w] = sum_v x[k,t,v] * P_s[v,w]                                   (:87-88, per frame)
//   y[o,t,w]   = relu( bn_scale[o] * ( sum_s ( Wd_s[o,:] . u_s[:,t,w] + bd_s[o] ) ) + bn_shift[o]
//                      + res[o,t,w] )                                         (:88-93)
//   res        = down_scale[o] * ( Wdown[o,:] . x[:,t,w] + bdown[o] ) + down_shift[o]   (Cin != Cout)
//              = x[o,t,w]                                                      (Cin == Cout, :57-58)
//
// The kernel is HBM-store-bound: 4*Cout*T*V bytes written per clip against 4*Cin*T*V read.
//  * agcn_expand_small_kernel<CIN,S>: the stem shape (Cin=3, 3 subsets).  Everything linear is
//    folded into one (Cout x F) matrix, F = (S+1)*CIN: per pixel the F features [u_0,u_1,u_2,x]
//    live in registers and every output channel is F FMAs + bias + ReLU, stored coalesced along
//    the pixel axis (each wave writes 256 contiguous bytes per channel).
//  * agcn_expand_generic_kernel: any Cin/Cout (used by the deeper TCN_GCN_unit layers).
#include <type_traits>

#include "common.h"

namespace stgcn {

namespace {

template <int CIN, int S>
__global__ __launch_bounds__(256) void agcn_expand_small_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ Wd,
    const float *__restrict__ bd, const float *__restrict__ Wdown, const float *__restrict__ bdown,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift,
    const float *__restrict__ down_scale, const float *__restrict__ down_shift,
    float *__restrict__ y, int Cout, int T, int V, int TF, int mode) {
    constexpr int F = (S + 1) * CIN;
    constexpr int FP = (F + 1 + 3) / 4 * 4;  // row of folded weights: F weights, bias, pad to x4
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TF;
    const int tf = min(TF, T - t0);
    const int px = tf * V;
    const int PXM = TF * V;
    float *Ws = smem;                          // [Cout][FP]
    float *Ps = Ws + (size_t)Cout * FP;        // [S][V][V]
    float *Xs = Ps + S * V * V;                // [CIN][PXM]

    // fold: W12[o][s*CIN+k] = bn_scale[o]*Wd[s][o][k]; W12[o][S*CIN+k] = down_scale[o]*Wdown[o][k]
    for (int e = tid; e < Cout * FP; e += 256) {
        const int o = e / FP, f = e - o * FP;
        float val = 0.f;
        if (f < S * CIN) {
            const int s = f / CIN, k = f - s * CIN;
            val = bn_scale[o] * Wd[((size_t)s * Cout + o) * CIN + k];
        } else if (f < F) {
            val = down_scale[o] * Wdown[o * CIN + (f - S * CIN)];
        } else if (f == F) {
            float b = 0.f;
            for (int s = 0; s < S; ++s) b += bd[s * Cout + o];
            val = fmaf(bn_scale[o], b, bn_shift[o]) + fmaf(down_scale[o], bdown[o], down_shift[o]);
        }
        Ws[e] = val;
    }
    const float *Pn = P + (size_t)n * S * V * V;
    for (int e = tid; e < S * V * V; e += 256) Ps[e] = Pn[e];
    const float *xn = x + (size_t)n * CIN * T * V;
    for (int e = tid; e < CIN * px; e += 256) {
        const int k = e / px, p = e - k * px;
        Xs[k * PXM + p] = xn[((size_t)k * T + t0) * V + p];
    }
    __syncthreads();

    for (int p = tid; p < px; p += 256) {
        const int tt = p / V, w = p - tt * V;
        float feat[F];
#pragma unroll
        for (int f = 0; f < S * CIN; ++f) feat[f] = 0.f;
        for (int v = 0; v < V; ++v) {
            float xv[CIN];
#pragma unroll
            for (int k = 0; k < CIN; ++k) xv[k] = Xs[k * PXM + tt * V + v];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float pv = Ps[(s * V + v) * V + w];
#pragma unroll
                for (int k = 0; k < CIN; ++k) feat[s * CIN + k] = fmaf(xv[k], pv, feat[s * CIN + k]);
            }
        }
#pragma unroll
        for (int k = 0; k < CIN; ++k) feat[S * CIN + k] = Xs[k * PXM + p];

        float *yo = y + ((size_t)n * Cout * T + t0) * V + p;
        for (int o = 0; o < Cout; ++o) {
            const float *wr = Ws + o * FP;
            float acc = wr[F];
#pragma unroll
            for (int f = 0; f < F; ++f) acc = fmaf(wr[f], feat[f], acc);
            yo[(size_t)o * T * V] = (mode & 1) ? acc : fmaxf(acc, 0.f);  // mode bit 0: raw (pre-activation) output
        }
    }
}

// The same with FOUR consecutive pixels per thread and 16-byte stores.  One dword per lane and channel (above) is bound by
// store issue, not HBM: ~7 B/clk/CU, 3.8 TB/s measured on 256 clips; 16 B per lane doubles what a CU can issue, which
// leaves HBM as the bound.  Needs T*V % 4 == 0 and frame chunks of a multiple of 4 pixels (16-byte aligned rows).
template <int CIN, int S>
__global__ __launch_bounds__(256) void agcn_expand_small4_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ Wd,
    const float *__restrict__ bd, const float *__restrict__ Wdown, const float *__restrict__ bdown,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift,
    const float *__restrict__ down_scale, const float *__restrict__ down_shift,
    float *__restrict__ y, int Cout, int T, int V, int TF, int mode) {
    constexpr int F = (S + 1) * CIN;
    constexpr int FP = (F + 1 + 3) / 4 * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TF;
    const int tf = min(TF, T - t0);
    const int px = tf * V;                     // multiple of 4 (host side)
    const int PXM = TF * V;
    float *Ws = smem;                          // [Cout][FP]
    float *Ps = Ws + (size_t)Cout * FP;        // [S][V][V]
    float *Xs = Ps + S * V * V;                // [CIN][PXM]
    for (int e = tid; e < Cout * FP; e += 256) {
        const int o = e / FP, f = e - o * FP;
        float val = 0.f;
        if (f < S * CIN) {
            const int s = f / CIN, k = f - s * CIN;
            val = bn_scale[o] * Wd[((size_t)s * Cout + o) * CIN + k];
        } else if (f < F) {
            val = down_scale[o] * Wdown[o * CIN + (f - S * CIN)];
        } else if (f == F) {
            float b = 0.f;
            for (int s = 0; s < S; ++s) b += bd[s * Cout + o];
            val = fmaf(bn_scale[o], b, bn_shift[o]) + fmaf(down_scale[o], bdown[o], down_shift[o]);
        }
        Ws[e] = val;
    }
    const float *Pn = P + (size_t)n * S * V * V;
    for (int e = tid; e < S * V * V; e += 256) Ps[e] = Pn[e];
    const float *xn = x + (size_t)n * CIN * T * V;
    for (int e = tid; e < CIN * px; e += 256) {
        const int k = e / px, p = e - k * px;
        Xs[k * PXM + p] = xn[((size_t)k * T + t0) * V + p];
    }
    __syncthreads();
    for (int p4 = tid * 4; p4 < px; p4 += 1024) {
        float feat[4][F];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p4 + q, tt = p / V, w = p - tt * V;
#pragma unroll
            for (int f = 0; f < S * CIN; ++f) feat[q][f] = 0.f;
            for (int v = 0; v < V; ++v) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float pv = Ps[(s * V + v) * V + w];
#pragma unroll
                    for (int k = 0; k < CIN; ++k) feat[q][s * CIN + k] = fmaf(Xs[k * PXM + tt * V + v], pv, feat[q][s * CIN + k]);
                }
            }
#pragma unroll
            for (int k = 0; k < CIN; ++k) feat[q][S * CIN + k] = Xs[k * PXM + p];
        }
        float *yo = y + ((size_t)n * Cout * T + t0) * V + p4;
        const float lo = (mode & 1) ? -__builtin_huge_valf() : 0.f;   // mode bit 0: raw (pre-activation) output
        for (int o = 0; o < Cout; ++o) {
            const float4 *wr = reinterpret_cast<const float4 *>(Ws + o * FP);
            const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
            const float wv[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
            float a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = wv[F];
#pragma unroll
                for (int f = 0; f < F; ++f) a[q] = fmaf(wv[f], feat[q][f], a[q]);
            }
            *reinterpret_cast<float4 *>(yo + (size_t)o * T * V) =
                make_float4(fmaxf(a[0], lo), fmaxf(a[1], lo), fmaxf(a[2], lo), fmaxf(a[3], lo));
        }
    }
}

constexpr int OB = 32;  // output channels per register block in the generic kernel

__global__ __launch_bounds__(256) void agcn_expand_generic_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ Wd,
    const float *__restrict__ bd, const float *__restrict__ Wdown, const float *__restrict__ bdown,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift,
    const float *__restrict__ down_scale, const float *__restrict__ down_shift,
    float *__restrict__ y, int Cin, int Cout, int T, int V, int S, int TF, int mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TF;
    const int tf = min(TF, T - t0);
    const int px = tf * V;
    const int PXM = TF * V;
    float *Ps = smem;              // [S][V][V]
    float *Xs = Ps + S * V * V;    // [Cin][PXM]
    const float *Pn = P + (size_t)n * S * V * V;
    for (int e = tid; e < S * V * V; e += 256) Ps[e] = Pn[e];
    const float *xn = x + (size_t)n * Cin * T * V;
    for (int e = tid; e < Cin * px; e += 256) {
        const int k = e / px, p = e - k * px;
        Xs[k * PXM + p] = xn[((size_t)k * T + t0) * V + p];
    }
    __syncthreads();
    const bool identity = (Wdown == nullptr);
    for (int p = tid; p < px; p += 256) {
        const int tt = p / V, w = p - tt * V;
        float *yo = y + ((size_t)n * Cout * T + t0) * V + p;
        for (int o0 = 0; o0 < Cout; o0 += OB) {
            float acc[OB], res[OB];
#pragma unroll
            for (int j = 0; j < OB; ++j) { acc[j] = 0.f; res[j] = 0.f; }
            for (int k = 0; k < Cin; ++k) {
                const float *xr = Xs + k * PXM + tt * V;
                for (int s = 0; s < S; ++s) {
                    float u = 0.f;
                    for (int v = 0; v < V; ++v) u = fmaf(xr[v], Ps[(s * V + v) * V + w], u);
                    const float *wd = Wd + ((size_t)s * Cout + o0) * Cin + k;
#pragma unroll
                    for (int j = 0; j < OB; ++j)
                        if (o0 + j < Cout) acc[j] = fmaf(wd[(size_t)j * Cin], u, acc[j]);
                }
                if (!identity) {
                    const float xv = xr[w];
                    const float *wn = Wdown + (size_t)o0 * Cin + k;
#pragma unroll
                    for (int j = 0; j < OB; ++j)
                        if (o0 + j < Cout) res[j] = fmaf(wn[(size_t)j * Cin], xv, res[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < OB; ++j) {
                const int o = o0 + j;
                if (o < Cout) {
                    float b = 0.f;
                    for (int s = 0; s < S; ++s) b += bd[s * Cout + o];
                    float val = fmaf(bn_scale[o], acc[j] + b, bn_shift[o]);
                    if (identity) { if (!(mode & 2)) val += Xs[o * PXM + p]; }  // mode bit 1: leave the residual out
                    else val += fmaf(down_scale[o], res[j] + bdown[o], down_shift[o]);
                    yo[(size_t)o * T * V] = (mode & 1) ? val : fmaxf(val, 0.f);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Generic C_in / C_out on the fp32 matrix cores (C_in, C_out multiples of 16, 3 subsets: every deeper TCN_GCN_unit
// layer, model/ST_TR/ST_TR_new.py:355).  The old generic kernel recomputed the aggregation per block of 32 output channels
// on plain FMAs: 8.2 ms at 64 -> 64 and 82 ms at 128 -> 128 channels (256 clips, T = 180).
//
// Workgroup (8 waves) = one clip x one frame chunk of <= 256 pixels x ALL output channels; the input channels go by in
// chunks of 16.  Per chunk:
//   x rows -> LDS;  u_s[c][t,w] = sum_v x[c][t,v] P_s[v,w] as 16 x 16 x (V) blocks (block = (subset, frame, w block): the
//   16 rows of a block are the chunk's 16 channels, so no row table is needed) -> LDS rows (s, c);
//   out[o][p] += sum_kk A[o][kk] B[kk][p],  kk = (s, c) for the three subsets then (x, c) for the conv residual:
//   A = Wd_s * bn_scale / Wdown * down_scale (the BatchNorm scales folded in at fragment load: ONE accumulator serves both
//   branches), fragments in registers per chunk, B = the LDS rows (pitch = 16 mod 64 floats: conflict-free).
// Epilogue: + folded constants (+ x for the identity residual), ReLU unless raw, 64-byte segments per channel row.
// v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation (same 1e-4 gate as every fp32 path).
template <int NOW /* o-blocks per wave */, int NPB /* pixel blocks per wave */>
__global__ __launch_bounds__(512, (NOW == 1 ? 4 : 2)) void agcn_expand_mfma_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ Wd,
    const float *__restrict__ bd, const float *__restrict__ Wdown, const float *__restrict__ bdown,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift,
    const float *__restrict__ down_scale, const float *__restrict__ down_shift,
    float *__restrict__ y, int Cin, int Cout, int T, int V, int TF, int PXP, int mode, unsigned long long *dbg) {
#ifdef STGCN_ABLATION   // per-wave phase clocks (diagnostic builds; tools/stamps_k2g.py): 0 x rows -> LDS, 1 barrier waits,
                        // 2 aggregation, 3 weight fragments, 4 expansion MFMAs, 5 epilogue, 6 whole kernel
#define KE_STAMP(var) unsigned long long var = 0; if (dbg) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); }
#define KE_ACC(slot, a, b) if (dbg) { tsum[slot] += (b) - (a); }
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
#define KE_STAMP(var)
#define KE_ACC(slot, a, b)
#endif
    KE_STAMP(t_begin)
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int n = blockIdx.y, t0 = blockIdx.x * TF;
    const int tf = min(TF, T - t0), px = tf * V;
    const int VV = V * V, nvb = (V + 15) / 16;
    float *Ps = smem;                       // [3][V][V]
    float *Ub = Ps + 3 * VV;                // [64][PXP]: rows (s, c) for s = 0..2, then the x rows (c)
    float *Xs = Ub + 48 * PXP;              // [16][PXX]: the aggregation reads these rows as the MFMA's A operand — lanes along
                                            // the ROWS (one channel per lane), k along the joints: pitch = 4 mod 64 floats puts the
                                            // 16 rows x 4 k on 64 distinct banks (with the B tiles' pitch, 16 mod 64, the reads
                                            // were 4-way bank-conflicted)
    constexpr int PXX = 256 + 4;
    const bool identity = (Wdown == nullptr);
    const int KK = identity ? 48 : 64;      // contraction rows per chunk
    const int nob = Cout / 16;
    // wave -> its NOW o-blocks (stride 8 / wpo) and its NPB pixel blocks
    const int wpo = nob >= 8 ? 1 : 8 / nob;                 // waves per o-block
    const int ob0 = wave / wpo, pb0 = wave % wpo;           // first o-block (step 8 / wpo = nob when wpo > 1), first pixel block
    const int ostep = 8 / wpo;
    const size_t plane = (size_t)T * V;
    const float *Pn = P + (size_t)n * 3 * VV;
    for (int e = tid; e < 3 * VV; e += 512) Ps[e] = Pn[e];
    const float *xn = x + (size_t)n * Cin * plane + (size_t)t0 * V;

    f32x4 acc[NOW][NPB];
#pragma unroll
    for (int a = 0; a < NOW; ++a)
#pragma unroll
        for (int b = 0; b < NPB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // x rows of a chunk: a wave per two channel rows, lanes along the pixels; fetched one chunk ahead into registers
    float xr_[2][4];
    auto xfetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = wave + 8 * j;                     // LDS row r of the chunk holds channel c0 + rho(r): see the weight fragments
            const float *xr = xn + (size_t)(c0 + (((r & 3) << 2) | (r >> 2))) * plane;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int p = lane + 64 * q; xr_[j][q] = p < px ? xr[p] : 0.f; }
        }
    };
    xfetch(0);
    for (int c0 = 0; c0 < Cin; c0 += 16) {
        KE_STAMP(t0_)
        __syncthreads();                                    // previous chunk's rows fully consumed (and Ps loaded)
        KE_STAMP(t1_)
        KE_ACC(1, t0_, t1_)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) Xs[(wave + 8 * j) * PXX + lane + 64 * q] = xr_[j][q];
        if (c0 + 16 < Cin) xfetch(c0 + 16);
        KE_STAMP(t2_)
        KE_ACC(0, t1_, t2_)
        __syncthreads();
        KE_STAMP(t3_)
        KE_ACC(1, t2_, t3_)
        // ---- aggregation u_s = x P_s: 16 x 16 blocks (rows = the chunk's 16 channels, columns = a block of joints w) per frame.
        // A unit = (subset, w block, frame quarter): its P fragments are read ONCE and serve every frame of the quarter and
        // every k-step count is a compile-time constant of the joint count class — no index division per block, no branch
        // between a read and its MFMA.  (As one block per trip
        // with two divisions, per-step tests and a single-step remainder loop this phase cost 390 cycles per MFMA: 24 - 43 %
        // of the kernel, tools/stamps_k2g.py.)
        auto aggregate = [&](auto nks_c) __attribute__((always_inline)) {
            constexpr int NKS = decltype(nks_c)::value;
            const int npair = 3 * nvb;
            for (int unit = wave; unit < 4 * npair; unit += 8) {
                const int fq = unit / npair, pair = unit - fq * npair, s = pair / nvb, wb = pair - s * nvb;
                const int w = wb * 16 + l16;
                const bool wok = w < V;
                float pf[NKS];
#pragma unroll
                for (int k = 0; k < NKS; ++k) {
                    const int v = 4 * k + lq;
                    const float pv = Ps[s * VV + min(v, V - 1) * V + (wok ? w : 0)];
                    pf[k] = (v < V && wok) ? pv : 0.f;
                }
                const float *xrow = Xs + l16 * PXX;
                for (int t = fq; t < tf; t += 4) {
                    float af[NKS];
#pragma unroll
                    for (int k = 0; k < NKS; ++k) {
                        const int v = 4 * k + lq;
                        const float xv = xrow[t * V + min(v, V - 1)];
                        af[k] = v < V ? xv : 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < NKS; ++k) a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[k], pf[k], a4, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (wok) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) Ub[(s * 16 + 4 * lq + i) * PXP + t * V + w] = a4[i];
                    }
                }
            }
        };
        if (V <= 24) {
            aggregate(std::integral_constant<int, 6>{});
        } else {     // wider frames: one block per trip, four k-steps at a time (more register-resident operands spill here:
                     // two workgroups per CU cap this kernel at 128 registers)
            for (int u = wave; u < 3 * tf * nvb; u += 8) {
                const int s = u / (tf * nvb), rem = u - s * tf * nvb, t = rem / nvb, wb = rem - t * nvb;
                const int w = wb * 16 + l16;
                f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
                const float *xr = Xs + l16 * PXX + t * V;
                const float *pr = Ps + s * VV + (w < V ? w : 0);
                const int nks = (V + 3) / 4;
                for (int ks = 0; ks < nks; ks += 4) {
                    float av[4], bv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int v = 4 * (ks + q) + lq;
                        const float xv = xr[min(v, V - 1)], pv = pr[min(v, V - 1) * V];
                        av[q] = v < V ? xv : 0.f;
                        bv[q] = (v < V && w < V) ? pv : 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], a4, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (w < V) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) Ub[(s * 16 + 4 * lq + i) * PXP + t * V + w] = a4[i];
                }
            }
        }
        KE_STAMP(t4_)
        KE_ACC(2, t3_, t4_)
        // ---- this wave's weight fragments of the chunk (BatchNorm scales folded in)
        float wf[NOW][16];
#pragma unroll
        for (int a = 0; a < NOW; ++a) {
            const int o = (ob0 + a * ostep) * 16 + l16;
            const float sm_ = bn_scale[o], sd_ = identity ? 0.f : down_scale[o];
            // k-step ks = 4 s + j of lane group lq multiplies LDS row 16 s + 4 j + lq, which holds channel c0 + 4 lq + j (the rows
            // of a chunk are stored with their two 2-bit index fields swapped): a lane's four k-steps of a subset are ONE
            // 16-byte load of its weight row — as dwords (16 rows x 16 bytes per instruction) the fragments were 9 - 14 % of
            // the kernel (tools/stamps_k2g.py)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x4 w4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (s < 3) w4 = *reinterpret_cast<const f32x4 *>(Wd + ((size_t)s * Cout + o) * Cin + c0 + 4 * lq);
                else if (!identity) w4 = *reinterpret_cast<const f32x4 *>(Wdown + (size_t)o * Cin + c0 + 4 * lq);
                const float sc = s < 3 ? sm_ : sd_;
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[a][4 * s + j] = w4[j] * sc;
            }
        }
#ifdef STGCN_ABLATION
        if (dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        KE_STAMP(t5_)
        KE_ACC(3, t4_, t5_)
        __syncthreads();
        KE_STAMP(t6_)
        KE_ACC(1, t5_, t6_)
        // ---- out += A B
#pragma unroll
        for (int b = 0; b < NPB; ++b) {
            const int pb = pb0 * NPB + b;                   // a wave's pixel blocks are CONTIGUOUS (row segments of the epilogue)
            if (pb * 16 < px) {
                const float *br = Ub + lq * PXP + pb * 16 + l16;
                // Four k-steps' operands at a time, ONE uniform branch per group (the identity residual has 12 k-steps, the conv
                // residual 16) and a scheduling fence between the reads and the MFMAs: with a test around every step each step
                // was its own basic block — read, wait out the LDS round trip, multiply (tools/stamps_k1g.py found the same
                // pattern costing the attention kernel 68 % of its time).
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    if (16 * kg < KK) {
                        float bf[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)          // (kg = 3: the conv residual's x rows, on their own pitch)
                            bf[q] = kg < 3 ? br[(size_t)4 * (4 * kg + q) * PXP] : Xs[(4 * q + lq) * PXX + pb * 16 + l16];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int a = 0; a < NOW; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[a][4 * kg + q], bf[q], acc[a][b], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        KE_STAMP(t7_)
        KE_ACC(4, t6_, t7_)
    }
    KE_STAMP(t_epi)
    // ---- epilogue: + folded constants (+ x for the identity residual), ReLU unless raw.
    // Even V (every row segment then starts on an 8-byte boundary): each 16-channel x 128-pixel piece of a wave goes through
    // the wave's own 8 KiB slice of the (now free) operand tiles and leaves as ONE 512-byte row segment per store instruction
    // (8 bytes per lane) — straight from the accumulators a store covered 4 rows x 64 bytes, 19 - 22 % of the kernel
    // (tools/stamps_k2g.py).  Odd V keeps the element-wise stores.
    float *yn = y + (size_t)n * Cout * plane + (size_t)t0 * V;
    auto row_const = [&](int o) {
        float cst = bn_shift[o], bsum = 0.f;
        for (int s = 0; s < 3; ++s) bsum += bd[s * Cout + o];
        cst = fmaf(bn_scale[o], bsum, cst);
        if (!identity) cst += fmaf(down_scale[o], bdown[o], down_shift[o]);
        return cst;
    };
    if ((V & 1) == 0) {
        constexpr int SP = 132;                             // slice pitch (floats): 4 rows apart = 16 banks apart
        __syncthreads();                                    // every wave is past its last read of Ub / Xs
        float *sl = Ub + wave * 16 * SP;
        const bool addx = identity && !(mode & 2);
        const float lo = (mode & 1) ? -__builtin_huge_valf() : 0.f;
#pragma unroll
        for (int a = 0; a < NOW; ++a) {
            const int o0 = (ob0 + a * ostep) * 16;
            float cst[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) cst[i] = row_const(o0 + 4 * lq + i);
#pragma unroll
            for (int h = 0; h < NPB / 8; ++h) {             // 8 pixel blocks = 128 pixels per pass
                const int pbase = (pb0 * NPB + h * 8) * 16;
                if (pbase < px) {
#pragma unroll
                    for (int b8 = 0; b8 < 8; ++b8)
#pragma unroll
                        for (int i = 0; i < 4; ++i) sl[(4 * lq + i) * SP + b8 * 16 + l16] = acc[a][h * 8 + b8][i] + cst[i];
                    // (same wave writes and reads the slice: program order is enough, no barrier)
                    const int pp = pbase + 2 * lane;        // this lane's two pixels of every row
                    if (pp < px) {                          // (px is even)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float2 v = *reinterpret_cast<const float2 *>(sl + r * SP + 2 * lane);
                            const size_t off = (size_t)(o0 + r) * plane + pp;
                            if (addx) {
                                const float2 xv = *reinterpret_cast<const float2 *>(xn + off);
                                v.x += xv.x; v.y += xv.y;
                            }
                            *reinterpret_cast<float2 *>(yn + off) = make_float2(fmaxf(v.x, lo), fmaxf(v.y, lo));
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < NOW; ++a) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = (ob0 + a * ostep) * 16 + 4 * lq + i;
                const float cst = row_const(o);
#pragma unroll
                for (int b = 0; b < NPB; ++b) {
                    const int p = (pb0 * NPB + b) * 16 + l16;
                    if (p < px) {
                        float val = acc[a][b][i] + cst;
                        if (identity && !(mode & 2)) val += xn[(size_t)o * plane + p];   // mode bit 1: leave the residual out
                        yn[(size_t)o * plane + p] = (mode & 1) ? val : fmaxf(val, 0.f);
                    }
                }
            }
        }
    }
#ifdef STGCN_ABLATION
    if (dbg) {
        KE_STAMP(t_end)
        KE_ACC(5, t_epi, t_end)
        KE_ACC(6, t_begin, t_end)
        if (lane == 0 && blockIdx.x == 0 && blockIdx.y < 8)
            for (int i = 0; i < 8; ++i) dbg[(blockIdx.y * 8 + wave) * 8 + i] = tsum[i];
    }
#endif
}

}  // namespace

int launch_agcn_expand(const float *x, const float *P, const float *Wd, const float *bd,
                       const float *Wdown, const float *bdown, const float *bn_scale,
                       const float *bn_shift, const float *down_scale, const float *down_shift,
                       float *y, int N, int Cin, int Cout, int T, int V, int S, int mode, hipStream_t st) {
    if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "agcn: N=%d > 65535 clips per call", N);
    if (Cin == 3 && S == 3 && Wdown != nullptr && V <= 256) {
        constexpr int F = 12, FP = 16;
        (void)F;
        int TF = 256 / V;
        if (TF < 1) TF = 1;
        if (TF > T) TF = T;
        if ((T * V) % 4 == 0) {                // 16-byte stores: four pixels per thread, chunks of ~1024 pixels
            const int gran = V % 4 == 0 ? 1 : (V % 2 == 0 ? 2 : 4);      // frames per chunk such that chunk*V % 4 == 0
            int TF4 = 1024 / V / gran * gran;
            if (TF4 < gran) TF4 = gran;
            if (TF4 > T) TF4 = T;               // (T*V % 4 == 0: the single chunk is aligned as well)
            const size_t lds4 = ((size_t)Cout * FP + (size_t)S * V * V + (size_t)3 * TF4 * V) * 4;
            if (lds4 <= (size_t)kLdsBytes) {
                STGCN_HIP_CHECK(allow_lds(agcn_expand_small4_kernel<3, 3>, lds4));
                hipLaunchKernelGGL((agcn_expand_small4_kernel<3, 3>), dim3(ceil_div(T, TF4), N), dim3(256), lds4, st, x, P, Wd, bd,
                                   Wdown, bdown, bn_scale, bn_shift, down_scale, down_shift, y, Cout, T, V, TF4, mode);
                STGCN_LAUNCH_CHECK("agcn_expand_small4_kernel");
                return STGCN_OK;
            }
        }
        const size_t lds = ((size_t)Cout * FP + (size_t)S * V * V + (size_t)3 * TF * V) * 4;
        if (lds <= (size_t)kLdsBytes) {
            STGCN_HIP_CHECK(allow_lds(agcn_expand_small_kernel<3, 3>, lds));
            hipLaunchKernelGGL((agcn_expand_small_kernel<3, 3>), dim3(ceil_div(T, TF), N), dim3(256), lds,
                               st, x, P, Wd, bd, Wdown, bdown, bn_scale, bn_shift, down_scale, down_shift,
                               y, Cout, T, V, TF, mode);
            STGCN_LAUNCH_CHECK("agcn_expand_small_kernel");
            return STGCN_OK;
        }
    }
    // generic shapes on the matrix cores
    if (S == 3 && Cin % 16 == 0 && Cout % 16 == 0 && V <= 64 && (Cout == 64 || Cout == 128 || Cout == 256) && !(ablate_mask() & 4096)) {
        int TF = 256 / V;
        if (TF < 1) TF = 1;
        if (TF > T) TF = T;
        const int PXP = 256 + 16;
        const size_t lds = ((size_t)3 * V * V + (size_t)64 * PXP) * 4;
        const dim3 grid(ceil_div(T, TF), N);
#define LAUNCH_EXP(NOW_, NPB_)                                                                                         \
    do {                                                                                                               \
        STGCN_HIP_CHECK(allow_lds((agcn_expand_mfma_kernel<NOW_, NPB_>), lds));                                        \
        hipLaunchKernelGGL((agcn_expand_mfma_kernel<NOW_, NPB_>), grid, dim3(512), lds, st, x, P, Wd, bd, Wdown, bdown, bn_scale, \
                           bn_shift, down_scale, down_shift, y, Cin, Cout, T, V, TF, PXP, mode, debug_buffer());      \
    } while (0)
        if (Cout == 64) LAUNCH_EXP(1, 8);          // 4 o-blocks x 2 waves each: 8 of the 16 pixel blocks per wave
        else if (Cout == 128) LAUNCH_EXP(1, 16);   // 8 o-blocks, one wave each
        else LAUNCH_EXP(2, 16);                    // 16 o-blocks, two per wave
#undef LAUNCH_EXP
        STGCN_LAUNCH_CHECK("agcn_expand_mfma_kernel");
        return STGCN_OK;
    }
    // generic, any shape (plain FMAs)
    const size_t budget = (size_t)96 * 1024 / 4;
    const size_t pfl = (size_t)S * V * V;
    if (pfl + (size_t)Cin * V > (size_t)kLdsBytes / 4)
        return fail(STGCN_ERR_UNSUPPORTED, "agcn: Cin=%d V=%d does not fit LDS", Cin, V);
    int TF = 1;
    if (budget > pfl) TF = (int)((budget - pfl) / ((size_t)Cin * V));
    if (TF > 256 / V) TF = 256 / V;
    if (TF < 1) TF = 1;
    if (TF > T) TF = T;
    const size_t lds = (pfl + (size_t)Cin * TF * V) * 4;
    STGCN_HIP_CHECK(allow_lds(agcn_expand_generic_kernel, lds));
    hipLaunchKernelGGL(agcn_expand_generic_kernel, dim3(ceil_div(T, TF), N), dim3(256), lds, st, x, P, Wd,
                       bd, Wdown, bdown, bn_scale, bn_shift, down_scale, down_shift, y, Cin, Cout, T, V, S,
                       TF, mode);
    STGCN_LAUNCH_CHECK("agcn_expand_generic_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
