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
    // generic
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
