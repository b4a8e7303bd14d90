// Backward of unit_agcn (model/unit_agcn.py:73-93) in TRAINING mode for the stem's shape class (C_in <= 4 with a
// "down" branch), i.e. what autograd derives for
//   P_s  = softmax_v( Gram(Wa_s x + ba_s, Wb_s x + bb_s) / (inter_c*T) ) + A_s + PA_s
//   u_s  = x P_s ;  zm = sum_s (Wd_s u_s + bd_s) ;  zd = Wdown x + bdown ;  y = relu( BN_m(zm) + BN_d(zd) )
// from dy.  x is data (no dx).  After the two elementwise BatchNorm statistics passes (tcn_backward.hip) ONE kernel
// does the rest, one workgroup per clip at a time:
//   * per frame chunk and 32-channel block it rebuilds g = dy*[y>0] and the two pre-BatchNorm gradients
//       dzm = gm*invm*(g - mean(g) - xhat_m*mean(g*xhat_m)),  dzd likewise            (never written to HBM)
//   * dWd_s[o,k] += dzm[o,p]*u_s[k,p],  dbd_s[o] += dzm[o,p],  dWdown[o,k] += dzd[o,p]*x[k,p],  dbdown[o] += dzd[o,p]
//   * du_s[k,p]  = sum_o Wd_s[o,k]*dzm[o,p] ;  dP_s[v,w] += sum_{k,t} x[k,t,v]*du_s[k,t,w]
//   * at the end of the clip: dPA += dP ;  soft-max backward  dS = Q*(dP - colsum(Q*dP))/(inter_c*T), Q = P - A_eff ;
//     dM_s[k,l] += sum_{t,v,w} x~[k,t,v]*dS_s[v,w]*x~[l,t,w]   (x~ = [x;1]: the 4x4 bilinear form the forward folds the
//     two embeddings into, M_s = Wa~_s^T Wb~_s)
// and a last tiny kernel sums the per-workgroup partials in a fixed order and maps dM to dWa, dba, dWb, dbb.
#include "common.h"

namespace stgcn {

namespace {

constexpr int PXMAX = 256;   // pixels per frame chunk
constexpr int DPITCH = PXMAX + 1;

struct BnRef {
    const float *z, *scale, *shift, *mean, *invstd, *coef;   // coef: [gamma*invstd | mean(g) | mean(g*xhat)] x C
};

template <int CIN, int S, int NOB>
__global__ __launch_bounds__(256) void agcn_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ A_eff, BnRef m, BnRef d,
    const float *__restrict__ dy, const float *__restrict__ Wd, float *__restrict__ part_w /* [grid][Cout][WCOLS] */,
    float *__restrict__ part_pa /* [grid][S][V][V] */, float *__restrict__ part_m /* [grid][S][C1][C1] */, int N, int Cout,
    int T, int V, int inter_c, int TF) {
    constexpr int SC = S * CIN, C1 = CIN + 1;
    constexpr int WCOLS = SC + 1 + CIN + 1;          // per output channel: dWd (SC), dbd, dWdown (CIN), dbdown
    constexpr int MAINF = SC + 1;                    // main features + constant 1
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int VV = V * V;
    float *Ps = sm;                                  // [S][V][V]  P of the clip
    float *dPs = Ps + S * VV;                        // [S][V][V]  dP, later dS
    float *Xs = dPs + S * VV;                        // [CIN][PXMAX]
    float *Fs = Xs + CIN * PXMAX;                    // [SC][PXMAX] u_s
    float *DUs = Fs + SC * PXMAX;                    // [SC][PXMAX] du_s
    float *Dm = DUs + SC * PXMAX;                    // [32][DPITCH] dzm of the channel block
    float *Dd = Dm + 32 * DPITCH;                    // [32][DPITCH] dzd
    float *Wl = Dd + 32 * DPITCH;                    // [Cout][SC]   Wd re-ordered: Wl[o][s*CIN+k]
    float *red = Wl + Cout * SC;                     // [4][S*C1*C1] block reduction of dM
    const size_t plane = (size_t)T * V;

    for (int e = tid; e < Cout * SC; e += 256) {
        const int o = e / SC, f = e - o * SC, s = f / CIN, k = f - s * CIN;
        Wl[e] = Wd[((size_t)s * Cout + o) * CIN + k];
    }
    float accw[NOB][3];                              // this thread's slice of part_w: (channel tid>>3 of block ob) x 3 columns
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) accw[ob][0] = accw[ob][1] = accw[ob][2] = 0.f;
    float accm[S][C1][C1];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < C1; ++k)
#pragma unroll
            for (int l = 0; l < C1; ++l) accm[s][k][l] = 0.f;
    const int ol = tid >> 3, fg = tid & 7;           // phase (b): channel within the block, feature group
    float *my_pa = part_pa + (size_t)blockIdx.x * S * VV;
    for (int e = tid; e < S * VV; e += 256) my_pa[e] = 0.f;

    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        __syncthreads();
        const float *Pn = P + (size_t)n * S * VV;
        for (int e = tid; e < S * VV; e += 256) { Ps[e] = Pn[e]; dPs[e] = 0.f; }
        const float *xn = x + (size_t)n * CIN * plane;
        for (int t0 = 0; t0 < T; t0 += TF) {
            const int px = min(TF, T - t0) * V;
            __syncthreads();
            for (int e = tid; e < CIN * px; e += 256) {
                const int k = e / px, p = e - k * px;
                Xs[k * PXMAX + p] = xn[(size_t)k * plane + (size_t)t0 * V + p];
            }
            __syncthreads();
            float du[SC];
#pragma unroll
            for (int f = 0; f < SC; ++f) du[f] = 0.f;
            if (tid < px) {                          // u_s[k] of this thread's pixel (model/unit_agcn.py:87-88)
                const int tt = tid / V, w = tid - tt * V;
                float u[SC];
#pragma unroll
                for (int f = 0; f < SC; ++f) u[f] = 0.f;
                for (int v = 0; v < V; ++v) {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float pw = Ps[(s * V + v) * V + w];
#pragma unroll
                        for (int k = 0; k < CIN; ++k) u[s * CIN + k] = fmaf(Xs[k * PXMAX + tt * V + v], pw, u[s * CIN + k]);
                    }
                }
#pragma unroll
                for (int f = 0; f < SC; ++f) Fs[f * PXMAX + tid] = u[f];
            }
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                __syncthreads();                     // Fs complete / previous block consumed
                // rebuild dzm, dzd of channels ob*32 .. +31 for the chunk's pixels
                for (int e = tid; e < 32 * px; e += 256) {
                    const int r = e / px, p = e - r * px, c = ob * 32 + r;
                    const size_t g = ((size_t)n * Cout + c) * plane + (size_t)t0 * V + p;
                    const float zm = m.z[g], zd = d.z[g];
                    const float pre = fmaf(zm, m.scale[c], m.shift[c]) + fmaf(zd, d.scale[c], d.shift[c]);
                    const float gg = pre > 0.f ? dy[g] : 0.f;
                    Dm[r * DPITCH + p] = m.coef[c] * (gg - m.coef[Cout + c] - (zm - m.mean[c]) * m.invstd[c] * m.coef[2 * Cout + c]);
                    Dd[r * DPITCH + p] = d.coef[c] * (gg - d.coef[Cout + c] - (zd - d.mean[c]) * d.invstd[c] * d.coef[2 * Cout + c]);
                }
                __syncthreads();
                if (tid < px) {                      // (a) du_s[k] += Wd_s[o][k] * dzm[o]
                    for (int r = 0; r < 32; ++r) {
                        const float dv = Dm[r * DPITCH + tid];
                        const float *wr = Wl + (ob * 32 + r) * SC;
#pragma unroll
                        for (int f = 0; f < SC; ++f) du[f] = fmaf(wr[f], dv, du[f]);
                    }
                }
                {                                    // (b) weight / bias gradients of channel ob*32 + ol
                    const float *dmr = Dm + ol * DPITCH, *ddr = Dd + ol * DPITCH;
                    const int f0 = fg, f1 = fg + 8;  // main columns (feature SC is the constant 1 = bias)
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                    for (int p = 0; p < px; ++p) {
                        const float dm = dmr[p], dd = ddr[p];
                        a0 = fmaf(dm, f0 < SC ? Fs[f0 * PXMAX + p] : 1.f, a0);
                        if (f1 < MAINF) a1 = fmaf(dm, f1 < SC ? Fs[f1 * PXMAX + p] : 1.f, a1);
                        if (fg <= CIN) a2 = fmaf(dd, fg < CIN ? Xs[fg * PXMAX + p] : 1.f, a2);
                    }
                    accw[ob][0] += a0;
                    accw[ob][1] += a1;
                    accw[ob][2] += a2;
                }
            }
            __syncthreads();
            if (tid < px) {
#pragma unroll
                for (int f = 0; f < SC; ++f) DUs[f * PXMAX + tid] = du[f];
            }
            __syncthreads();
            const int tf = px / V;
            for (int e = tid; e < S * VV; e += 256) {  // dP_s[v][w] += sum_{k,t} x[k,t,v] * du_s[k,t,w]
                const int s = e / VV, vw = e - s * VV, v = vw / V, w = vw - v * V;
                float a = 0.f;
                for (int tt = 0; tt < tf; ++tt)
#pragma unroll
                    for (int k = 0; k < CIN; ++k)
                        a = fmaf(Xs[k * PXMAX + tt * V + v], DUs[(s * CIN + k) * PXMAX + tt * V + w], a);
                dPs[e] += a;
            }
        }
        __syncthreads();
        for (int e = tid; e < S * VV; e += 256) my_pa[e] += dPs[e];   // PA enters P additively (unit_agcn.py:76,85)
        __syncthreads();
        // soft-max backward over v (dim -2), one thread per column (s, w)
        const float denom = (float)(inter_c * T);
        for (int e = tid; e < S * V; e += 256) {
            const int s = e / V, w = e - s * V;
            float dot = 0.f;
            for (int v = 0; v < V; ++v) {
                const int i = (s * V + v) * V + w;
                dot = fmaf(Ps[i] - A_eff[i], dPs[i], dot);
            }
            for (int v = 0; v < V; ++v) {
                const int i = (s * V + v) * V + w;
                dPs[i] = (Ps[i] - A_eff[i]) * (dPs[i] - dot) / denom;
            }
        }
        // dM_s[k][l] += sum_{t,v} x~[k,t,v] * ( sum_w dS_s[v,w] * x~[l,t,w] )
        for (int t0 = 0; t0 < T; t0 += TF) {
            const int px = min(TF, T - t0) * V;
            __syncthreads();
            for (int e = tid; e < CIN * px; e += 256) {
                const int k = e / px, p = e - k * px;
                Xs[k * PXMAX + p] = xn[(size_t)k * plane + (size_t)t0 * V + p];
            }
            __syncthreads();
            for (int it = tid; it < S * px; it += 256) {
                const int s = it / px, p = it - s * px, tt = p / V, v = p - tt * V;
                float r[C1];
#pragma unroll
                for (int l = 0; l < C1; ++l) r[l] = 0.f;
                const float *ds = dPs + (s * V + v) * V;
                for (int w = 0; w < V; ++w) {
                    const float dsv = ds[w];
#pragma unroll
                    for (int l = 0; l < CIN; ++l) r[l] = fmaf(dsv, Xs[l * PXMAX + tt * V + w], r[l]);
                    r[CIN] += dsv;
                }
                float xt[C1];
#pragma unroll
                for (int k = 0; k < CIN; ++k) xt[k] = Xs[k * PXMAX + p];
                xt[CIN] = 1.f;
#pragma unroll
                for (int s2 = 0; s2 < S; ++s2)
                    if (s2 == s) {
#pragma unroll
                        for (int k = 0; k < C1; ++k)
#pragma unroll
                            for (int l = 0; l < C1; ++l) accm[s2][k][l] = fmaf(xt[k], r[l], accm[s2][k][l]);
                    }
            }
        }
    }

    // ---- partials of this workgroup ------------------------------------------------------------
    float *my_w = part_w + (size_t)blockIdx.x * Cout * WCOLS;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        float *row = my_w + (size_t)(ob * 32 + ol) * WCOLS;
        row[fg] = accw[ob][0];                                   // main column fg (fg < 8 <= SC)
        if (fg + 8 < MAINF) row[fg + 8] = accw[ob][1];           // main columns 8 .. SC (SC = bias)
        if (fg <= CIN) row[MAINF + fg] = accw[ob][2];            // down columns
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < C1; ++k)
#pragma unroll
            for (int l = 0; l < C1; ++l) {
                float v = accm[s][k][l];
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                if (lane == 0) red[wave * S * C1 * C1 + (s * C1 + k) * C1 + l] = v;
            }
    __syncthreads();
    if (tid < S * C1 * C1)
        part_m[(size_t)blockIdx.x * S * C1 * C1 + tid] =
            red[tid] + red[S * C1 * C1 + tid] + red[2 * S * C1 * C1 + tid] + red[3 * S * C1 * C1 + tid];
}

// Sums the partials in a fixed order and maps them to the parameter gradients.  One workgroup.
template <int CIN, int S>
__global__ __launch_bounds__(256) void agcn_bwd_final_kernel(
    const float *__restrict__ part_w, const float *__restrict__ part_pa, const float *__restrict__ part_m, int parts,
    const float *__restrict__ Wa, const float *__restrict__ ba, const float *__restrict__ Wb, const float *__restrict__ bb,
    float *__restrict__ dWa, float *__restrict__ dba, float *__restrict__ dWb, float *__restrict__ dbb,
    float *__restrict__ dWd, float *__restrict__ dbd, float *__restrict__ dWdown, float *__restrict__ dbdown,
    float *__restrict__ dPA, int Cout, int V, int inter_c) {
    constexpr int SC = S * CIN, C1 = CIN + 1, WCOLS = SC + 1 + CIN + 1, MAINF = SC + 1;
    __shared__ float dM[S * C1 * C1];
    const int tid = threadIdx.x;
    for (int e = tid; e < Cout * WCOLS; e += 256) {
        float s = 0.f;
        for (int p = 0; p < parts; ++p) s += part_w[(size_t)p * Cout * WCOLS + e];
        const int o = e / WCOLS, c = e - o * WCOLS;
        if (c < SC) dWd[((size_t)(c / CIN) * Cout + o) * CIN + (c % CIN)] = s;
        else if (c == SC) { for (int q = 0; q < S; ++q) dbd[q * Cout + o] = s; }   // every bd_s adds straight into zm
        else if (c < MAINF + CIN) dWdown[o * CIN + (c - MAINF)] = s;
        else dbdown[o] = s;
    }
    for (int e = tid; e < S * V * V; e += 256) {
        float s = 0.f;
        for (int p = 0; p < parts; ++p) s += part_pa[(size_t)p * S * V * V + e];
        dPA[e] = s;
    }
    if (tid < S * C1 * C1) {
        float s = 0.f;
        for (int p = 0; p < parts; ++p) s += part_m[(size_t)p * S * C1 * C1 + tid];
        dM[tid] = s;
    }
    __syncthreads();
    // M_s[k][l] = sum_c Wa~_s[c][k] * Wb~_s[c][l]   =>   dWa~[c][k] = sum_l dM[k][l] Wb~[c][l],  dWb~[c][l] = sum_k dM[k][l] Wa~[c][k]
    for (int e = tid; e < S * inter_c * C1; e += 256) {
        const int s = e / (inter_c * C1), rc = e - s * inter_c * C1, c = rc / C1, j = rc - c * C1;
        const int row = s * inter_c + c;
        float ga = 0.f, gb = 0.f;
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const float wb = i < CIN ? Wb[row * CIN + i] : bb[row];
            const float wa = i < CIN ? Wa[row * CIN + i] : ba[row];
            ga = fmaf(dM[(s * C1 + j) * C1 + i], wb, ga);      // j = k
            gb = fmaf(dM[(s * C1 + i) * C1 + j], wa, gb);      // j = l
        }
        if (j < CIN) { dWa[row * CIN + j] = ga; dWb[row * CIN + j] = gb; }
        else { dba[row] = ga; dbb[row] = gb; }
    }
}

struct AgcnBwdPlan {
    bool ok = false;
    int TF = 0, grid = 0;
    size_t lds = 0;
};

inline AgcnBwdPlan plan_agcn_bwd(int N, int Cin, int Cout, int T, int V, int S) {
    AgcnBwdPlan pl;
    if (Cin != 3 || S != 3 || (Cout != 64 && Cout != 128 && Cout != 256) || V > PXMAX) return pl;
    const int SC = S * Cin, C1 = Cin + 1;
    int TF = PXMAX / V;
    if (TF > T) TF = T;
    const size_t fl = (size_t)2 * S * V * V + (size_t)(Cin + 2 * SC) * PXMAX + (size_t)2 * 32 * DPITCH + (size_t)Cout * SC +
                      (size_t)4 * S * C1 * C1;
    pl.lds = fl * 4;
    if (pl.lds > (size_t)kLdsBytes) return pl;
    pl.TF = TF;
    pl.grid = N < 256 ? N : 256;
    pl.ok = true;
    return pl;
}

}  // namespace

bool agcn_bwd_supported(int N, int Cin, int Cout, int T, int V, int S) { return plan_agcn_bwd(N, Cin, Cout, T, V, S).ok; }

// partials: [grid][Cout][WCOLS] + [grid][S][V][V] + [grid][S][C1][C1] floats
size_t agcn_bwd_part_bytes(int N, int Cin, int Cout, int T, int V, int S) {
    const AgcnBwdPlan pl = plan_agcn_bwd(N, Cin, Cout, T, V, S);
    if (!pl.ok) return 0;
    const int C1 = Cin + 1, WCOLS = S * Cin + 1 + Cin + 1;
    return (size_t)pl.grid * ((size_t)Cout * WCOLS + (size_t)S * V * V + (size_t)S * C1 * C1) * sizeof(float);
}

// m_* / d_*: z, scale, shift, mean, invstd, coef of the main / down BatchNorm (coef from launch_bn_bwd_finalize)
int launch_agcn_bwd(const float *x, const float *P, const float *A_eff, const float *const m_[6], const float *const d_[6],
                    const float *dy, const float *Wa, const float *ba, const float *Wb, const float *bb, const float *Wd,
                    float *part, float *dWa, float *dba, float *dWb, float *dbb, float *dWd, float *dbd, float *dWdown,
                    float *dbdown, float *dPA, int N, int Cin, int Cout, int T, int V, int inter_c, int S, hipStream_t st) {
    const AgcnBwdPlan pl = plan_agcn_bwd(N, Cin, Cout, T, V, S);
    if (!pl.ok)
        return fail(STGCN_ERR_UNSUPPORTED,
                    "agcn backward covers Cin=3, 3 subsets, Cout in {64,128,256} with a down branch (got Cin=%d S=%d Cout=%d V=%d)",
                    Cin, S, Cout, V);
    const int C1 = Cin + 1, WCOLS = S * Cin + 1 + Cin + 1;
    float *part_w = part, *part_pa = part_w + (size_t)pl.grid * Cout * WCOLS, *part_m = part_pa + (size_t)pl.grid * S * V * V;
    const BnRef m{m_[0], m_[1], m_[2], m_[3], m_[4], m_[5]}, d{d_[0], d_[1], d_[2], d_[3], d_[4], d_[5]};
#define LAUNCH_BWD(NOB)                                                                                          \
    do {                                                                                                         \
        STGCN_HIP_CHECK(allow_lds((agcn_bwd_kernel<3, 3, NOB>), pl.lds));                                        \
        hipLaunchKernelGGL((agcn_bwd_kernel<3, 3, NOB>), dim3(pl.grid), dim3(256), pl.lds, st, x, P, A_eff, m, d, dy, \
                           Wd, part_w, part_pa, part_m, N, Cout, T, V, inter_c, pl.TF);                          \
    } while (0)
    if (Cout == 64) LAUNCH_BWD(2);
    else if (Cout == 128) LAUNCH_BWD(4);
    else LAUNCH_BWD(8);
#undef LAUNCH_BWD
    STGCN_LAUNCH_CHECK("agcn_bwd_kernel");
    hipLaunchKernelGGL((agcn_bwd_final_kernel<3, 3>), dim3(1), dim3(256), 0, st, part_w, part_pa, part_m, pl.grid, Wa, ba,
                       Wb, bb, dWa, dba, dWb, dbb, dWd, dbd, dWdown, dbdown, dPA, Cout, V, inter_c);
    STGCN_LAUNCH_CHECK("agcn_bwd_final_kernel");
    (void)C1;
    return STGCN_OK;
}

}  // namespace stgcn
