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
// and two tiny kernels sum the per-workgroup partials in a fixed order and map dM to dWa, dba, dWb, dbb.
#include "common.h"

namespace stgcn {

namespace {

constexpr int PXMAX = 256;   // pixels per frame chunk
constexpr int FP = 260;      // row pitch of the x / feature / du tiles: rows 4 banks apart (a pitch of 256 floats put all
                             // 16 feature rows of phase (b) on the same banks: 2/3 of the LDS cycles were conflicts)
constexpr int DP = 260;      // row pitch of the dz tiles: 16-byte aligned rows, 8 consecutive rows on distinct banks
constexpr int NTB = 512;     // threads per workgroup
constexpr int NCST = 12;     // per-channel constants (10 used)

struct BnRef {
    const float *z, *scale, *shift, *mean, *invstd, *coef;   // coef: [gamma*invstd | mean(g) | mean(g*xhat)] x C
};

template <int CIN, int S, int NOB>
__global__ __launch_bounds__(NTB) void agcn_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ A_eff, BnRef m, BnRef d,
    const float *__restrict__ dy, const float *__restrict__ Wd, float *__restrict__ part_w /* [grid][Cout][WCOLS] */,
    float *__restrict__ part_pa /* [grid][S][V][V] */, float *__restrict__ part_m /* [grid][S][C1][C1] */, int N, int Cout,
    int T, int V, int inter_c, int TF) {
    constexpr int SC = S * CIN, C1 = CIN + 1;
    constexpr int WCOLS = SC + 1 + CIN + 1;          // per output channel: dWd (SC), dbd, dWdown (CIN), dbdown
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int VV = V * V;
    float *Dm = sm;                                  // [32][DP] dzm of the channel block   (16-byte aligned rows)
    float *Dd = Dm + 32 * DP;                        // [32][DP] dzd
    float *Xs = Dd + 32 * DP;                        // [CIN][FP]
    float *Fs = Xs + CIN * FP;                       // [SC][FP] u_s
    float *DUs = Fs + SC * FP;                       // [SC][FP] du_s
    float *cst = DUs + SC * FP;                   // [Cout][NCST] per-channel constants of the two BatchNorm backward maps
    float *red = cst + Cout * NCST;                  // [8][S*C1*C1] block reduction of dM
    float *Ps = red + 8 * S * C1 * C1;               // [S][V][V]  P of the clip
    float *dPs = Ps + S * VV;                        // [S][V][V]  dP, later dS
    const size_t plane = (size_t)T * V;

    // pre = (sm*zm + tm) + (sd*zd + td) ;  dzm = am*g + bm*zm + cm ;  dzd = ad*g + bd*zd + cd     (g = dy where pre > 0)
    for (int c = tid; c < Cout; c += NTB) {
        float *q = cst + c * NCST;
        const float km = m.coef[c], c1 = m.coef[Cout + c], c2m = m.coef[2 * Cout + c], im = m.invstd[c];
        const float kd = d.coef[c], c2d = d.coef[2 * Cout + c], id = d.invstd[c];
        q[0] = m.scale[c]; q[1] = d.scale[c]; q[2] = m.shift[c]; q[9] = d.shift[c];
        q[3] = km; q[4] = -km * im * c2m; q[5] = km * (im * c2m * m.mean[c] - c1);
        q[6] = kd; q[7] = -kd * id * c2d; q[8] = kd * (id * c2d * d.mean[c] - c1);
    }
    float accw[NOB];                                 // column (tid & 15) of channel (tid >> 4) of block ob in part_w
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) accw[ob] = 0.f;
    float accm[S][C1][C1];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < C1; ++k)
#pragma unroll
            for (int l = 0; l < C1; ++l) accm[s][k][l] = 0.f;
    const int ol = tid >> 4, col = tid & 15;         // phase (b): channel within the block, column of part_w
    const int pa = tid & 255, half = tid >> 8;       // phase (a): pixel, half of the block's 32 channels
    float *my_pa = part_pa + (size_t)blockIdx.x * S * VV;
    for (int e = tid; e < S * VV; e += NTB) my_pa[e] = 0.f;

    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        __syncthreads();
        const float *Pn = P + (size_t)n * S * VV;
        for (int e = tid; e < S * VV; e += NTB) { Ps[e] = Pn[e]; dPs[e] = 0.f; }
        const float *xn = x + (size_t)n * CIN * plane;
        for (int t0 = 0; t0 < T; t0 += TF) {
            const int px = min(TF, T - t0) * V;
            __syncthreads();
            for (int e = tid; e < CIN * px; e += NTB) {
                const int k = e / px, p = e - k * px;
                Xs[k * FP + p] = xn[(size_t)k * plane + (size_t)t0 * V + p];
            }
            for (int e = tid; e < SC * FP; e += NTB) DUs[e] = 0.f;
            __syncthreads();
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
                        for (int k = 0; k < CIN; ++k) u[s * CIN + k] = fmaf(Xs[k * FP + tt * V + v], pw, u[s * CIN + k]);
                    }
                }
#pragma unroll
                for (int f = 0; f < SC; ++f) Fs[f * FP + tid] = u[f];
            }
            float du[SC];
#pragma unroll
            for (int f = 0; f < SC; ++f) du[f] = 0.f;
            // the three full-size operands of a channel block travel HBM -> registers one block ahead of their use
            // (12 independent loads per lane in flight; fetched in place, the tile build was a chain of load round trips)
            float rzm[4][PXMAX / 64], rzd[4][PXMAX / 64], rdy[4][PXMAX / 64];
            auto prefetch = [&](int ob) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const size_t g0 = ((size_t)n * Cout + ob * 32 + wave * 4 + rr) * plane + (size_t)t0 * V;
#pragma unroll
                    for (int i = 0; i < PXMAX / 64; ++i) {
                        const size_t g = g0 + min(lane + 64 * i, px - 1);
                        rzm[rr][i] = m.z[g];
                        rzd[rr][i] = d.z[g];
                        rdy[rr][i] = dy[g];
                    }
                }
            };
            prefetch(0);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                __syncthreads();                     // Fs complete / previous block consumed
                // rebuild dzm, dzd of channels ob*32 .. +31: wave w owns rows 4w .. 4w+3, lanes stride the pixels
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = wave * 4 + rr, c = ob * 32 + r;
                    const float *q = cst + c * NCST;
                    const float s_m = q[0], s_d = q[1], t_m = q[2], t_d = q[9], am = q[3], bm = q[4], cm = q[5], ad = q[6], bd = q[7],
                                cd = q[8];
#pragma unroll
                    for (int i = 0; i < PXMAX / 64; ++i) {
                        const int p = lane + 64 * i;
                        const float zm = rzm[rr][i], zd = rzd[rr][i];
                        const float gg = fmaf(zm, s_m, t_m) + fmaf(zd, s_d, t_d) > 0.f ? rdy[rr][i] : 0.f;   // the forward's own expression
                        if (p < px) {
                            Dm[r * DP + p] = fmaf(am, gg, fmaf(bm, zm, cm));
                            Dd[r * DP + p] = fmaf(ad, gg, fmaf(bd, zd, cd));
                        }
                    }
                }
                if (ob + 1 < NOB) prefetch(ob + 1);
                __syncthreads();
                if (pa < px) {                       // (a) du_s[k] += Wd_s[o][k] * dzm[o], this half's 16 channels
                    for (int r = half * 16; r < half * 16 + 16; ++r) {
                        const float dv = Dm[r * DP + pa];
                        const int o = ob * 32 + r;   // (wave-uniform: the weights come through scalar loads)
#pragma unroll
                        for (int s = 0; s < S; ++s)
#pragma unroll
                            for (int k = 0; k < CIN; ++k)
                                du[s * CIN + k] = fmaf(Wd[((size_t)s * Cout + o) * CIN + k], dv, du[s * CIN + k]);
                    }
                }
                if (col < WCOLS) {                   // (b) column `col` of channel ob*32 + ol
                    const bool main_side = col <= SC;
                    const float *dr = (main_side ? Dm : Dd) + ol * DP;
                    const bool is_bias = col == SC || col == WCOLS - 1;
                    const float *fr = col < SC ? Fs + col * FP : Xs + (is_bias ? 0 : col - SC - 1) * FP;
                    float a = 0.f;
                    const int px4 = px & ~3;
                    if (is_bias) {
                        for (int p = 0; p < px4; p += 4) {
                            const float4 dq = *reinterpret_cast<const float4 *>(dr + p);
                            a += (dq.x + dq.y) + (dq.z + dq.w);
                        }
                        for (int p = px4; p < px; ++p) a += dr[p];
                    } else {
                        for (int p = 0; p < px4; p += 4) {
                            const float4 dq = *reinterpret_cast<const float4 *>(dr + p);
                            const float4 fq = *reinterpret_cast<const float4 *>(fr + p);
                            a = fmaf(dq.x, fq.x, fmaf(dq.y, fq.y, fmaf(dq.z, fq.z, fmaf(dq.w, fq.w, a))));
                        }
                        for (int p = px4; p < px; ++p) a = fmaf(dr[p], fr[p], a);
                    }
                    accw[ob] += a;
                }
            }
            if (pa < px) {                           // the two halves' du (a + b: order-independent)
#pragma unroll
                for (int f = 0; f < SC; ++f) atomicAdd(&DUs[f * FP + pa], du[f]);
            }
            __syncthreads();
            const int tf = px / V;
            for (int e = tid; e < S * VV; e += NTB) {  // dP_s[v][w] += sum_{k,t} x[k,t,v] * du_s[k,t,w]
                const int s = e / VV, vw = e - s * VV, v = vw / V, w = vw - v * V;
                float a = 0.f;
                for (int tt = 0; tt < tf; ++tt)
#pragma unroll
                    for (int k = 0; k < CIN; ++k)
                        a = fmaf(Xs[k * FP + tt * V + v], DUs[(s * CIN + k) * FP + tt * V + w], a);
                dPs[e] += a;
            }
        }
        __syncthreads();
        for (int e = tid; e < S * VV; e += NTB) my_pa[e] += dPs[e];   // PA enters P additively (unit_agcn.py:76,85)
        __syncthreads();
        // soft-max backward over v (dim -2), one thread per column (s, w)
        const float denom = (float)(inter_c * T);
        for (int e = tid; e < S * V; e += NTB) {
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
            for (int e = tid; e < CIN * px; e += NTB) {
                const int k = e / px, p = e - k * px;
                Xs[k * FP + p] = xn[(size_t)k * plane + (size_t)t0 * V + p];
            }
            __syncthreads();
            for (int it = tid; it < S * px; it += NTB) {
                const int s = it / px, p = it - s * px, tt = p / V, v = p - tt * V;
                float r[C1];
#pragma unroll
                for (int l = 0; l < C1; ++l) r[l] = 0.f;
                const float *ds = dPs + (s * V + v) * V;
                for (int w = 0; w < V; ++w) {
                    const float dsv = ds[w];
#pragma unroll
                    for (int l = 0; l < CIN; ++l) r[l] = fmaf(dsv, Xs[l * FP + tt * V + w], r[l]);
                    r[CIN] += dsv;
                }
                float xt[C1];
#pragma unroll
                for (int k = 0; k < CIN; ++k) xt[k] = Xs[k * FP + p];
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
    if (col < WCOLS) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) my_w[(size_t)(ob * 32 + ol) * WCOLS + col] = accw[ob];
    }
    __syncthreads();
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
    if (tid < S * C1 * C1) {
        float v = 0.f;
        for (int w8 = 0; w8 < 8; ++w8) v += red[w8 * S * C1 * C1 + tid];
        part_m[(size_t)blockIdx.x * S * C1 * C1 + tid] = v;
    }
}

// Sums the per-workgroup partials in a fixed order.  Workgroup = 32 output elements x 8 groups of partials: thread
// (e = tid & 31, g = tid >> 5) adds partials g, g+8, ...; the 8 sub-sums of an element are then added in order g = 0..7.
//   out_w [Cout][WCOLS] -> dWd (S,Cout,CIN), dbd (S,Cout), dWdown (Cout,CIN), dbdown (Cout);  dPA;  dM -> dm_out
template <int CIN, int S>
__global__ __launch_bounds__(256) void agcn_bwd_reduce_kernel(
    const float *__restrict__ part_w, const float *__restrict__ part_pa, const float *__restrict__ part_m, int parts,
    float *__restrict__ dWd, float *__restrict__ dbd, float *__restrict__ dWdown, float *__restrict__ dbdown,
    float *__restrict__ dPA, float *__restrict__ dm_out, int Cout, int V) {
    constexpr int SC = S * CIN, C1 = CIN + 1, WCOLS = SC + 1 + CIN + 1, MAINF = SC + 1;
    const int nw = Cout * WCOLS, npa = S * V * V, nm = S * C1 * C1;
    __shared__ float sub[8][32];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const float *src = nullptr;
    size_t stride = 0;
    int i = 0;
    if (e < nw) { src = part_w; stride = nw; i = e; }
    else if (e < nw + npa) { src = part_pa; stride = npa; i = e - nw; }
    else if (e < nw + npa + nm) { src = part_m; stride = nm; i = e - nw - npa; }
    float a = 0.f;
    if (src != nullptr)
        for (int p = grp; p < parts; p += 8) a += src[(size_t)p * stride + i];
    sub[grp][el] = a;
    __syncthreads();
    if (grp != 0 || src == nullptr) return;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += sub[g][el];
    if (e < nw) {
        const int o = e / WCOLS, c = e - o * WCOLS;
        if (c < SC) dWd[((size_t)(c / CIN) * Cout + o) * CIN + (c % CIN)] = s;
        else if (c == SC) { for (int q = 0; q < S; ++q) dbd[q * Cout + o] = s; }   // every bd_s adds straight into zm
        else if (c < MAINF + CIN) dWdown[o * CIN + (c - MAINF)] = s;
        else dbdown[o] = s;
    } else if (e < nw + npa) {
        dPA[i] = s;
    } else {
        dm_out[i] = s;
    }
}

// M_s[k][l] = sum_c Wa~_s[c][k] * Wb~_s[c][l]  =>  dWa~[c][k] = sum_l dM[k][l] Wb~[c][l],  dWb~[c][l] = sum_k dM[k][l] Wa~[c][k]
template <int CIN, int S>
__global__ __launch_bounds__(256) void agcn_bwd_embed_kernel(
    const float *__restrict__ dM, const float *__restrict__ Wa, const float *__restrict__ ba, const float *__restrict__ Wb,
    const float *__restrict__ bb, float *__restrict__ dWa, float *__restrict__ dba, float *__restrict__ dWb,
    float *__restrict__ dbb, int inter_c) {
    constexpr int C1 = CIN + 1;
    for (int e = threadIdx.x; e < S * inter_c * C1; e += 256) {
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
    const size_t fl = (size_t)2 * 32 * DP + (size_t)(Cin + 2 * SC) * FP + (size_t)Cout * NCST + (size_t)8 * S * C1 * C1 +
                      (size_t)2 * S * V * V;
    pl.lds = fl * 4;
    if (pl.lds > (size_t)kLdsBytes) return pl;
    pl.TF = TF;
    pl.grid = N < 256 ? N : 256;
    pl.ok = true;
    return pl;
}

}  // namespace

bool agcn_bwd_supported(int N, int Cin, int Cout, int T, int V, int S) { return plan_agcn_bwd(N, Cin, Cout, T, V, S).ok; }

// partials: [grid][Cout][WCOLS] + [grid][S][V][V] + [grid][S][C1][C1] floats, then the summed dM (S*C1*C1)
size_t agcn_bwd_part_bytes(int N, int Cin, int Cout, int T, int V, int S) {
    const AgcnBwdPlan pl = plan_agcn_bwd(N, Cin, Cout, T, V, S);
    if (!pl.ok) return 0;
    const int C1 = Cin + 1, WCOLS = S * Cin + 1 + Cin + 1;
    return ((size_t)pl.grid * ((size_t)Cout * WCOLS + (size_t)S * V * V + (size_t)S * C1 * C1) + (size_t)S * C1 * C1) *
           sizeof(float);
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
    float *dm_sum = part_m + (size_t)pl.grid * S * C1 * C1;
    const BnRef m{m_[0], m_[1], m_[2], m_[3], m_[4], m_[5]}, d{d_[0], d_[1], d_[2], d_[3], d_[4], d_[5]};
#define LAUNCH_BWD(NOB)                                                                                          \
    do {                                                                                                         \
        STGCN_HIP_CHECK(allow_lds((agcn_bwd_kernel<3, 3, NOB>), pl.lds));                                        \
        hipLaunchKernelGGL((agcn_bwd_kernel<3, 3, NOB>), dim3(pl.grid), dim3(NTB), pl.lds, st, x, P, A_eff, m, d, dy, \
                           Wd, part_w, part_pa, part_m, N, Cout, T, V, inter_c, pl.TF);                          \
    } while (0)
    if (Cout == 64) LAUNCH_BWD(2);
    else if (Cout == 128) LAUNCH_BWD(4);
    else LAUNCH_BWD(8);
#undef LAUNCH_BWD
    STGCN_LAUNCH_CHECK("agcn_bwd_kernel");
    const int total = Cout * WCOLS + S * V * V + S * C1 * C1;
    hipLaunchKernelGGL((agcn_bwd_reduce_kernel<3, 3>), dim3(ceil_div(total, 32)), dim3(256), 0, st, part_w, part_pa, part_m,
                       pl.grid, dWd, dbd, dWdown, dbdown, dPA, dm_sum, Cout, V);
    STGCN_LAUNCH_CHECK("agcn_bwd_reduce_kernel");
    hipLaunchKernelGGL((agcn_bwd_embed_kernel<3, 3>), dim3(1), dim3(256), 0, st, dm_sum, Wa, ba, Wb, bb, dWa, dba, dWb, dbb,
                       inter_c);
    STGCN_LAUNCH_CHECK("agcn_bwd_embed_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
