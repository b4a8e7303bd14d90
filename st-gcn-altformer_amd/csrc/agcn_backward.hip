// Backward of unit_agcn (model/unit_agcn.py:73-93) in TRAINING mode for the stem's shape class (C_in = 3, 3 subsets,
// a "down" branch), i.e. what autograd derives for
//   P_s  = softmax_v( Gram(Wa_s x + ba_s, Wb_s x + bb_s) / (inter_c*T) ) + A_s + PA_s
//   u_s  = x P_s ;  zm = sum_s (Wd_s u_s + bd_s) ;  zd = Wdown x + bdown ;  y = relu( BN_m(zm) + BN_d(zd) )
// from dy.  x is data (no dx).
//
// MOMENT FORM.  Both pre-BatchNorm branches are LINEAR in 12 per-pixel features (u: the 9 aggregated values, x: the 3
// inputs), and a training-mode BatchNorm's input gradient is affine in (g, z):  dz = a*g + b*z + c  per channel, with
// g = dy*[y > 0] and (a, b, c) from the batch sums of g and g*z.  Substituting z = W.feature + bias, EVERYTHING the
// backward needs from the two full-size tensors (dy and the ReLU mask) collapses into
//   G[o][j]   = sum_pixels g[o,p] * F_j[p]          F = (u_0..u_8, 1, x_0..x_2)          (C_out x 13 numbers)
//   h_f[p]    = sum_o  Wd[f][o]*gamma_m[o]*invstd_m[o] * g[o,p]                            (9 values per pixel)
// and the second moments of the features the forward already took (agcn_train.hip):
//   sum g        = G[.,9]            sum g*zm = Wm.G[.,0:9] + bm*G[.,9]       sum g*zd = Wdown.G[.,10:13] + bdown*G[.,9]
//   dWd[o,f]     = a_m*G[o,f] + b_m*sum(zm*u_f) + c_m*sum(u_f),   sum(zm*u_f) = n*(Wm[o].E[u u_f] + bm[o] E[u_f])   (dWdown alike)
//   du_f[p]      = h_f[p] + sum_f' R[f][f'] u_f'[p] + r0[f],      R = Wm^T diag(b_m) Wm,  r0 = Wm^T (b_m*bm + c_m)
// so neither branch is rebuilt, no separate statistics pass over dy runs, and the heavy kernel reads 8 bytes per output
// element (dy, y) instead of 12 + a rebuilt 8:
//   1. agcn_bwd_gather_kernel   one pass over dy / y: G partials per workgroup (fp32 MFMA 16x16x4 from an LDS tile) and
//                               h (VALU, in the registers the loads land in) -> HBM (36 B per pixel)
//   2. reduce + finalize        G summed in fp64 in a fixed order; BatchNorm coefficients, dgamma/dbeta of both
//                               BatchNorms, dWd, dbd, dWdown, dbdown, R, r0 (all in fp64 on a handful of numbers)
//   3. agcn_bwd_attn_kernel     per clip: du = h + R u + r0, dP_s[v,w] = sum_{k,t} x[k,t,v] du_s[k,t,w], dPA += dP,
//                               soft-max backward  dS = Q*(dP - colsum(Q*dP))/(inter_c*T), Q = P - A_eff, and
//                               dM_s[k,l] += sum_{t,v,w} x~[k,t,v]*dS_s[v,w]*x~[l,t,w]   (x~ = [x;1]: the 4x4 bilinear form
//                               the forward folds the two embeddings into, M_s = Wa~_s^T Wb~_s)
//   4. two tiny kernels sum the per-workgroup partials in a fixed order and map dM to dWa, dba, dWb, dbb.
// Measured (256 clips, T=180, V=22): see DESIGN.md section 6.
#include "common.h"

namespace stgcn {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int PXMAX = 256;   // pixels per frame chunk
constexpr int FP = 260;      // row pitch of the LDS tiles (floats): 16-byte aligned rows, 16 consecutive rows on distinct
                             // banks for the 16-byte MFMA fragment reads
constexpr int NTB = 512;     // threads per workgroup
constexpr int CIN = 3, S = 3, SC = S * CIN, C1 = CIN + 1;
constexpr int NG = 16;       // columns of G: 0..8 g*u_f, 9 g, 10..12 g*x_k, 13..15 zero
constexpr int NMOM = 63;     // E[u] 9, E[u u^T] 45 (upper triangle), E[x] 3, E[x x^T] 6   (agcn_train.hip)
constexpr int NRR = SC * SC + SC;   // R (9x9) then r0 (9)

// One pass over dy and y.  Workgroup = one clip at a time, frame chunks of <= 256 pixels, channel blocks of 32:
//   wave w owns rows 4w..4w+3 of a block, lane l the pixels 4l..4l+3 of the chunk (one 16-byte load per row and tensor,
//   issued one block ahead: 8 loads of 16 B per lane = 64 KiB per CU in flight);
//   g -> LDS tile (double-buffered: one barrier per block) and, in the same registers, h += (Wd*gamma*invstd)[f][o] * g;
//   then wave (half, qg) multiplies rows half*16..+15 of the tile with the 16 feature rows over its 64 pixels
//   (v_mfma_f32_16x16x4_f32: exact fp32 products, A = g[o][p], B = F[j][p], 4 pixels per instruction).
template <int NOB, bool VEC>
__global__ __launch_bounds__(NTB) void agcn_bwd_gather_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ y, const float *__restrict__ dy,
    const float *__restrict__ wda /* [Cout][12]: Wd[f][o]*gamma[o]*invstd[o] (agcn_bwd_prep_kernel) */,
    float *__restrict__ part_g /* [grid][Cout][NG] */, float *__restrict__ hbuf /* [N][SC][T*V] */, int N, int T, int V, int TF,
    int abl) {
#ifdef STGCN_ABLATION   // diagnostic builds: phases can be switched off to price them (results are then wrong)
#define GATHER_ON(bit) (!(abl & (bit)))
#else
#define GATHER_ON(bit) true
#endif
    constexpr int Cout = NOB * 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: row addresses stay in SGPRs (as per-lane 64-bit
                                                                 // values hipcc hoisted all of them out of the loops and spilled)
    const int VV = V * V;
    float *Gt = sm;                        // [2][32][FP]  g of the channel block
    float *Fr = Gt + 2 * 32 * FP;          // [16][FP]     feature rows: u (9), ones, x (3), zeros (3)
    float *Ps = Fr + (NG + SC) * FP;       // [S][V][V]    (the SC rows in between: with Gt and Fr, room for the 8 waves'
                                           //              h partials at a chunk's end)
    static_assert(8 * SC * PXMAX <= (2 * 32 + NG + SC) * FP, "h exchange must fit in front of Ps");
    float *Xs = Fr + 10 * FP;
    const size_t plane = (size_t)T * V;

    f32x4 acc[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int half = wave & 1, qg = wave >> 1;
    const int nchunks = (T + TF - 1) / TF;
    const int nsteps = nchunks * NOB;

    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        const float *yn = y + (size_t)n * Cout * plane, *dyn = dy + (size_t)n * Cout * plane;
        // rows wave*4 .. +3 of block (step % NOB) of chunk (step / NOB): 16 bytes per row and tensor
        auto issue = [&](f32x4 (&ry)[4], f32x4 (&rd)[4], int step) __attribute__((always_inline)) {
            if (step >= nsteps || !GATHER_ON(1)) return;
            const int ch = step / NOB, ob = step - ch * NOB;
            const int px = min(TF, T - ch * TF) * V;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const size_t g0 = (size_t)(ob * 32 + wave * 4 + rr) * plane + (size_t)ch * TF * V;
                const float *yr = yn + g0, *dr = dyn + g0;      // wave-uniform
                if (VEC) {
                    const unsigned i4 = (unsigned)min(lane, px / 4 - 1) * 4u;
                    ry[rr] = *reinterpret_cast<const f32x4 *>(yr + i4);
                    rd[rr] = *reinterpret_cast<const f32x4 *>(dr + i4);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned p = (unsigned)min(4 * lane + j, px - 1);
                        ry[rr][j] = yr[p];
                        rd[rr][j] = dr[p];
                    }
                }
            }
        };
        // the loads of block step s are issued right after step s-2 has been consumed out of the same registers: two
        // blocks (128 KiB per CU) in flight, ~1.5 block periods ahead of their use
        f32x4 ya[4], da[4], yb[4], db[4];
        issue(ya, da, 0);
        issue(yb, db, 1);
        const float *xn = x + (size_t)n * CIN * plane;
        float xr[2];                                         // x of the next chunk: HBM -> registers a chunk ahead
        auto xfetch = [&](int ch) __attribute__((always_inline)) {
            const int px = min(TF, T - ch * TF) * V;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * NTB, k = e >> 8, p = e & 255;
                xr[i] = (k < CIN && p < px) ? xn[(size_t)k * plane + (size_t)ch * TF * V + p] : 0.f;
            }
        };
        xfetch(0);
        __syncthreads();
        const float *Pn = P + (size_t)n * S * VV;
        for (int e = tid; e < S * VV; e += NTB) Ps[e] = Pn[e];

        for (int ch = 0; ch < nchunks; ++ch) {
            const int t0 = ch * TF, px = min(TF, T - t0) * V;
            __syncthreads();                                 // previous chunk's h exchange fully consumed
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * NTB, k = e >> 8, p = e & 255;
                if (k < CIN) Xs[k * FP + p] = xr[i];
            }
            if (ch + 1 < nchunks) xfetch(ch + 1);
            for (int e = tid; e < PXMAX; e += NTB) Fr[9 * FP + e] = e < px ? 1.f : 0.f;
            for (int e = tid; e < 3 * FP; e += NTB) Fr[13 * FP + e] = 0.f;
            __syncthreads();
            if (tid < PXMAX) {                               // u_s[k] of this thread's pixel (model/unit_agcn.py:87-88)
                float u[SC];
#pragma unroll
                for (int f = 0; f < SC; ++f) u[f] = 0.f;
                if (tid < px && GATHER_ON(16)) {
                    const int tt = tid / V, w = tid - tt * V;
                    for (int v = 0; v < V; ++v) {
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            const float pw = Ps[(s * V + v) * V + w];
#pragma unroll
                            for (int k = 0; k < CIN; ++k) u[s * CIN + k] = fmaf(Xs[k * FP + tt * V + v], pw, u[s * CIN + k]);
                        }
                    }
                }
#pragma unroll
                for (int f = 0; f < SC; ++f) Fr[f * FP + tid] = u[f];
            }
            __syncthreads();
            f32x4 bq[4];                                     // this wave's feature fragments: B[k = pixel][n = feature row]
#pragma unroll
            for (int qi = 0; qi < 4; ++qi)
                bq[qi] = *reinterpret_cast<const f32x4 *>(Fr + (lane & 15) * FP + (qg * 4 + qi) * 16 + 4 * (lane >> 4));
            float h[SC][4];
#pragma unroll
            for (int f = 0; f < SC; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j) h[f][j] = 0.f;

            auto consume = [&](const f32x4 (&ry)[4], const f32x4 (&rd)[4], int ob, int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = wave * 4 + rr;
                    f32x4 g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = (ry[rr][j] > 0.f && 4 * lane + j < px) ? rd[rr][j] : 0.f;
                    *reinterpret_cast<f32x4 *>(Gt + (buf * 32 + r) * FP + 4 * lane) = g;
                    const float *wq = wda + (ob * 32 + r) * 12;      // wave-uniform: scalar loads, SGPR operands
#pragma unroll
                    for (int f = 0; f < (GATHER_ON(2) ? SC : 0); ++f) {
                        const float wf = wq[f];
#pragma unroll
                        for (int j = 0; j < 4; ++j) h[f][j] = fmaf(wf, g[j], h[f][j]);
                    }
                }
            };
            auto gram = [&](int ob, int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    const int q = qg * 4 + qi;
                    if (q * 16 < px && GATHER_ON(4)) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(Gt + (buf * 32 + half * 16 + (lane & 15)) * FP + q * 16 + 4 * (lane >> 4));
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bq[qi][j], acc[ob], 0, 0, 0);
                    }
                }
            };
#pragma unroll
            for (int ob = 0; ob < NOB; ob += 2) {
                consume(ya, da, ob, 0);
                __builtin_amdgcn_sched_barrier(0);
                issue(ya, da, ch * NOB + ob + 2);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                gram(ob, 0);
                __builtin_amdgcn_sched_barrier(0);
                consume(yb, db, ob + 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                issue(yb, db, ch * NOB + ob + 3);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                gram(ob + 1, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // h of the chunk: the 8 waves' partial sums meet in LDS (the tile, feature and Hs regions are free now and
            // contiguous: 8 x 9 x 256 floats), each (f, pixel) is summed in wave order and stored.  (LDS float atomics
            // took 19 us per chunk here — 344 us of the kernel's 655.)
            __syncthreads();                                 // every wave is past its last tile / fragment read
            if (GATHER_ON(8)) {
                float *hp = sm + (size_t)wave * SC * PXMAX;
#pragma unroll
                for (int f = 0; f < SC; ++f)
                    *reinterpret_cast<f32x4 *>(hp + f * PXMAX + 4 * lane) = f32x4{h[f][0], h[f][1], h[f][2], h[f][3]};
            }
            __syncthreads();
            float *hn = hbuf + (size_t)n * SC * plane + (size_t)t0 * V;
            for (int e = tid; e < SC * PXMAX; e += NTB) {
                const int f = e >> 8, p = e & 255;
                float a = sm[e];
#pragma unroll
                for (int w8 = 1; w8 < 8; ++w8) a += sm[(size_t)w8 * SC * PXMAX + e];
                if (p < px) hn[(size_t)f * plane + p] = a;
            }
        }
    }

    // ---- partials of this workgroup: the four pixel groups of a half, summed in a fixed order -------------------
    __syncthreads();
    float *red = Gt;                       // [4][Cout][NG]
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            red[((size_t)qg * Cout + ob * 32 + half * 16 + 4 * (lane >> 4) + i) * NG + (lane & 15)] = acc[ob][i];
    __syncthreads();
    float *my_g = part_g + (size_t)blockIdx.x * Cout * NG;
    for (int e = tid; e < Cout * NG; e += NTB)
        my_g[e] = ((red[e] + red[Cout * NG + e]) + red[2 * Cout * NG + e]) + red[3 * Cout * NG + e];
}

__global__ __launch_bounds__(256) void agcn_bwd_prep_kernel(const float *__restrict__ Wd, const float *__restrict__ bn_w,
                                                            const float *__restrict__ inv_m, float *__restrict__ wda, int Cout) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Cout * 12) return;
    const int o = e / 12, f = e - o * 12;
    wda[e] = f < SC ? Wd[((size_t)(f / CIN) * Cout + o) * CIN + (f % CIN)] * bn_w[o] * inv_m[o] : 0.f;
}

// G partials -> fp64 sums, fixed order: thread (e = tid & 31, grp = tid >> 5) adds partials grp, grp+32, ...; the 32 sub-sums of
// an element are then added in order.
__global__ __launch_bounds__(1024) void agcn_bwd_gsum_kernel(const float *__restrict__ part_g, int parts, int total,
                                                             double *__restrict__ G) {
    __shared__ double sub[32][32];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    double a = 0.0;
    if (e < total)
        for (int p = grp; p < parts; p += 32) a += (double)part_g[(size_t)p * total + e];
    sub[grp][el] = a;
    __syncthreads();
    if (grp != 0 || e >= total) return;
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 32; ++g) s += sub[g][el];
    G[e] = s;
}

__device__ inline double muu_at(const double *muu, int i, int j) {      // upper triangle, row-major (agcn_train.hip)
    if (i > j) { const int t = i; i = j; j = t; }
    return muu[i * SC - i * (i - 1) / 2 + (j - i)];
}
__device__ inline double mxx_at(const double *mxx, int i, int j) {
    if (i > j) { const int t = i; i = j; j = t; }
    return mxx[i * CIN - i * (i - 1) / 2 + (j - i)];
}

// Everything that is a function of G and the feature moments (one workgroup, fp64): see the file header.
__global__ __launch_bounds__(256) void agcn_bwd_finalize_kernel(
    const double *__restrict__ G, const double *__restrict__ mom, double count, const float *__restrict__ Wd,
    const float *__restrict__ bd, const float *__restrict__ Wdown, const float *__restrict__ bdown,
    const float *__restrict__ bn_w, const float *__restrict__ dbn_w, const float *__restrict__ stats,
    float *__restrict__ dWd, float *__restrict__ dbd, float *__restrict__ dWdown, float *__restrict__ dbdown,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ ddgamma, float *__restrict__ ddbeta,
    float *__restrict__ rr /* NRR */, int Cout) {
    __shared__ double bS[256], cS[256];
    __shared__ float Wm[256 * SC];          // Wm[o][f] = Wd[f / 3][o][f % 3]: the R sums below read it 2 * Cout * 81 times
    const double *mu = mom, *muu = mom + SC, *mx = mom + SC + 45, *mxx = mom + SC + 45 + CIN;
    // the moments are only there after a moments-path forward (agcn_train.hip marks them); after a materialising forward this
    // block is uninitialised memory: every output of this kernel — and through rr everything downstream — becomes NaN
    // instead of a plausible wrong gradient
    const bool ok = reinterpret_cast<const unsigned *>(stats)[STGCN_MOMENTS_MARK_SLOT(Cout)] == STGCN_MOMENTS_MAGIC;
    const double count_ = count;
    count = ok ? count_ : __builtin_nan("");
    for (int e = threadIdx.x; e < Cout * SC; e += 256) {
        const int o = e / SC, f = e - o * SC;
        Wm[e] = Wd[((size_t)(f / CIN) * Cout + o) * CIN + (f % CIN)];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < Cout; o += 256) {
        double w[SC], bsum = 0.0, wd[CIN];
        for (int s = 0; s < S; ++s) bsum += (double)bd[s * Cout + o];
        for (int f = 0; f < SC; ++f) w[f] = (double)Wm[o * SC + f];
        for (int k = 0; k < CIN; ++k) wd[k] = (double)Wdown[o * CIN + k];
        const double bdn = (double)bdown[o];
        const double *Gr = G + (size_t)o * NG;
        const double Sg = Gr[9], c1 = Sg / count;
        const double mean_m = stats[o], inv_m = stats[Cout + o], mean_d = stats[2 * Cout + o], inv_d = stats[3 * Cout + o];
        {   // main BatchNorm and the three conv_d
            double gz = bsum * Sg, zsum = bsum;
            for (int f = 0; f < SC; ++f) { gz += w[f] * Gr[f]; zsum += w[f] * mu[f]; }
            const double dg = inv_m * (gz - mean_m * Sg), c2 = dg / count, km = (double)bn_w[o] * inv_m;
            const double am = km, bm = -km * inv_m * c2, cm = km * (inv_m * c2 * mean_m - c1);
            for (int f = 0; f < SC; ++f) {
                double zu = bsum * mu[f];
                for (int f2 = 0; f2 < SC; ++f2) zu += w[f2] * muu_at(muu, f2, f);
                dWd[((size_t)(f / CIN) * Cout + o) * CIN + (f % CIN)] = (float)(am * Gr[f] + count * (bm * zu + cm * mu[f]));
            }
            const float db = (float)(am * Sg + count * (bm * zsum + cm));
            for (int s = 0; s < S; ++s) dbd[s * Cout + o] = db;          // every bd_s adds straight into zm
            dgamma[o] = (float)(dg + 0.0 * count);          // (0 * count: NaN without the moments mark)
            dbeta[o] = (float)(Sg + 0.0 * count);
            bS[o] = bm;
            cS[o] = bm * bsum + cm;
        }
        {   // residual BatchNorm and the down conv
            double gz = bdn * Sg, zsum = bdn;
            for (int k = 0; k < CIN; ++k) { gz += wd[k] * Gr[10 + k]; zsum += wd[k] * mx[k]; }
            const double dg = inv_d * (gz - mean_d * Sg), c2 = dg / count, kd = (double)dbn_w[o] * inv_d;
            const double ad = kd, bdd = -kd * inv_d * c2, cd = kd * (inv_d * c2 * mean_d - c1);
            for (int k = 0; k < CIN; ++k) {
                double zx = bdn * mx[k];
                for (int k2 = 0; k2 < CIN; ++k2) zx += wd[k2] * mxx_at(mxx, k2, k);
                dWdown[o * CIN + k] = (float)(ad * Gr[10 + k] + count * (bdd * zx + cd * mx[k]));
            }
            dbdown[o] = (float)(ad * Sg + count * (bdd * zsum + cd));
            ddgamma[o] = (float)(dg + 0.0 * count);
            ddbeta[o] = (float)(Sg + 0.0 * count);
        }
    }
    __syncthreads();
    // R[f][f2] = sum_o Wm[o][f] b_m[o] Wm[o][f2],  r0[f] = sum_o Wm[o][f] (b_m[o] bm[o] + c_m[o]):  2 threads per entry
    {
        const int t = threadIdx.x >> 1, part = threadIdx.x & 1;
        double a = 0.0;
        if (t < NRR) {
            if (t < SC * SC) {
                const int f = t / SC, f2 = t - f * SC;
                for (int o = part; o < Cout; o += 2) a += (double)Wm[o * SC + f] * bS[o] * (double)Wm[o * SC + f2];
            } else {
                const int f = t - SC * SC;
                for (int o = part; o < Cout; o += 2) a += (double)Wm[o * SC + f] * cS[o];
            }
        }
        a += __shfl_xor(a, 1, 64);
        if (t < NRR && part == 0) rr[t] = (float)(a + 0.0 * count);
    }
}

// Per clip, on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), ONE pass over x in frame chunks:
//   u_s[(k,t),w]    = sum_v x[(k,t),v] P_s[v,w]                       -> LDS            (rows (k,t), 16 x 16 blocks)
//   du              = h + R u + r0                                     VALU, <= 45 FMAs per thread
//   dP_s[v,w]      += sum_{(k,t)} x[(k,t),v] du_s[(k,t),w]             accumulators stay in registers across the chunks
//   X2[(k,v),(l,w)] += sum_t x~[k,t,v] x~[l,t,w]                       x~ = [x;1]; upper-triangle blocks, registers
// then, per clip: dPA partial, soft-max backward dS = Q*(dP - colsum(Q*dP))/(inter_c*T), Q = P - A_eff, and
//   dM_s[k][l] = sum_{v,w} dS_s[v,w] X2[(k,v),(l,w)]                   (the 4x4 bilinear form of the two embeddings)
// Work units are dealt round-robin to the 8 waves: u blocks (subset, row block, column block), dP blocks (subset, v block,
// w block, K part), X2 blocks (I <= J).  MAXU / MAXG: static bounds of the latter two per wave.
// K loop of one 16 x 16 block: operands of four k-steps are fetched before their four MFMAs (hipcc does not pipeline the
// plain loop: every step then waits out its own LDS round trips — two dependent ones where a row table is involved).
template <class FA, class FB>
__device__ __forceinline__ void mfma_ksteps(f32x4 &acc, int k0, int k1, FA fa, FB fb) {
    int ks = k0;
    for (; ks + 4 <= k1; ks += 4) {
        float a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = fa(ks + q); b[q] = fb(ks + q); }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], acc, 0, 0, 0);
    }
    for (; ks < k1; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa(ks), fb(ks), acc, 0, 0, 0);
}

constexpr int ROWTAB = 784;   // (k,t) row -> LDS offset k*FP + t*V of the chunk, -1 = no such row

template <int MAXU, int MAXG>
__global__ __launch_bounds__(NTB) void agcn_bwd_attn_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ A_eff, const float *__restrict__ hbuf,
    const float *__restrict__ rr, float *__restrict__ part_pa /* [grid][S][V][V] */, float *__restrict__ part_m /* [grid][S][C1][C1] */,
    int N, int T, int V, int inter_c, int TF, int KS) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int VV = V * V, V3 = 3 * V;
    const int nvb = (V + 15) / 16, NGB = (V3 + 1 + 15) / 16, XP = NGB * 16 + 4;
    float *Xs = sm;                                  // [CIN][FP]
    float *Us = Xs + CIN * FP;                       // [SC][FP]  u_s
    float *DUs = Us + SC * FP;                       // [SC][FP]  du_s
    float *Rs = DUs + SC * FP;                       // [96]      R, r0
    float *dMs = Rs + 96;                            // [48]      dM of this workgroup's clips
    int *rowtab = reinterpret_cast<int *>(dMs + 48); // [ROWTAB]
    float *Ps = reinterpret_cast<float *>(rowtab + ROWTAB);   // [S][V][V]  P of the clip
    float *dPs = Ps + S * VV;                        // [S][V][V]  dP, later dS
    float *X2s = dPs + S * VV;                       // [NGB*16][XP]
    const size_t plane = (size_t)T * V;
    for (int e = tid; e < NRR; e += NTB) Rs[e] = rr[e];
    if (tid < 48) dMs[tid] = 0.f;
    float *my_pa = part_pa + (size_t)blockIdx.x * S * VV;
    for (int e = tid; e < S * VV; e += NTB) my_pa[e] = 0.f;

    // ---- this wave's X2 blocks (I <= J): per-lane source of the A / B value: >= 0 LDS offset k*FP + v, -1 ones row, -2 zero
    const int nblk = NGB * (NGB + 1) / 2;
    int gI[MAXG], gJ[MAXG], goA[MAXG], goB[MAXG];
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
        int g = wave + 8 * i, I = 0;
        if (g < nblk) {
            while (g >= NGB - I) { g -= NGB - I; ++I; }
            gI[i] = I; gJ[i] = I + g;
        } else { gI[i] = -1; gJ[i] = -1; }
        const int ia = gI[i] * 16 + l16, ib = gJ[i] * 16 + l16;
        goA[i] = ia < V3 ? (ia / V) * FP + (ia % V) : (ia == V3 ? -1 : -2);
        goB[i] = ib < V3 ? (ib / V) * FP + (ib % V) : (ib == V3 ? -1 : -2);
    }
    // ---- this wave's dP units
    const int ncombo = S * nvb * nvb, nunits = ncombo * KS;
    const int ksteps_d = (3 * TF + 3) / 4, kpp = (ksteps_d + KS - 1) / KS;
    int uS[MAXU], uV[MAXU], uW[MAXU], uK0[MAXU], uK1[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int unit = wave + 8 * i;
        const int kq = unit % KS, combo = unit / KS;
        uS[i] = unit < nunits ? combo / (nvb * nvb) : -1;
        uV[i] = (combo / nvb) % nvb;
        uW[i] = combo % nvb;
        uK0[i] = kq * kpp;
        uK1[i] = min(uK0[i] + kpp, ksteps_d);
    }
    const int nrb = (3 * TF + 15) / 16, n_uunits = S * nrb * nvb, ksteps_u = (V + 3) / 4;

    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        f32x4 accP[MAXU], accG[MAXG];
#pragma unroll
        for (int i = 0; i < MAXU; ++i) accP[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXG; ++i) accG[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        const float *Pn = P + (size_t)n * S * VV;
        for (int e = tid; e < S * VV; e += NTB) { Ps[e] = Pn[e]; dPs[e] = 0.f; }
        const float *xn = x + (size_t)n * CIN * plane;
        const float *hn = hbuf + (size_t)n * SC * plane;
        // x and h of a chunk travel HBM -> registers one chunk ahead (their latency was exposed twice per chunk)
        float xr[2], hr[5];
        auto fetch = [&](int t0) __attribute__((always_inline)) {
            const int px = min(TF, T - t0) * V;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * NTB, k = e >> 8, p = e & 255;
                xr[i] = (k < CIN && p < px) ? xn[(size_t)k * plane + (size_t)t0 * V + p] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int e = tid + i * NTB, f = e >> 8, p = e & 255;
                hr[i] = (f < SC && p < px) ? hn[(size_t)f * plane + (size_t)t0 * V + p] : 0.f;
            }
        };
        fetch(0);
        for (int t0 = 0; t0 < T; t0 += TF) {
            const int tf = min(TF, T - t0), px = tf * V;
            __syncthreads();                                 // previous chunk's tiles fully consumed
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * NTB, k = e >> 8, p = e & 255;
                if (k < CIN) Xs[k * FP + p] = xr[i];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int e = tid + i * NTB, f = e >> 8, p = e & 255;
                if (f < SC) DUs[f * FP + p] = hr[i];
            }
            if (t0 + TF < T) fetch(t0 + TF);
            for (int r = tid; r < 4 * ksteps_d + 16 && r < ROWTAB; r += NTB) {
                const int k = r / TF, t = r - k * TF;
                rowtab[r] = (k < CIN && t < tf) ? k * FP + t * V : -1;
            }
            __syncthreads();
            // ---- u blocks: A = x[(k,t)][v], B = P_s[v][w]
            for (int uu = wave; uu < n_uunits; uu += 8) {
                const int s = uu / (nrb * nvb), rb = (uu / nvb) % nrb, wb = uu % nvb;
                const int off = rowtab[rb * 16 + l16], w = wb * 16 + l16;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                mfma_ksteps(acc, 0, ksteps_u,
                            [&](int ks) { const int v = 4 * ks + lq; return (off >= 0 && v < V) ? Xs[off + v] : 0.f; },
                            [&](int ks) { const int v = 4 * ks + lq; return (v < V && w < V) ? Ps[(s * V + v) * V + w] : 0.f; });
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o2 = rowtab[rb * 16 + 4 * lq + i];
                    if (o2 >= 0 && w < V) Us[s * CIN * FP + o2 + w] = acc[i];
                }
            }
            // ---- X2 blocks: A = x~[(k,v)][t], B = x~[(l,w)][t]
            {
                const int ksteps_g = (tf + 3) / 4;
#pragma unroll
                for (int i = 0; i < MAXG; ++i) {
                    if (gI[i] >= 0) {
                        const int oa = goA[i], ob_ = goB[i];
                        mfma_ksteps(accG[i], 0, ksteps_g,
                                    [&](int ks) { const int t = 4 * ks + lq; return t >= tf ? 0.f : (oa >= 0 ? Xs[oa + t * V] : (oa == -1 ? 1.f : 0.f)); },
                                    [&](int ks) { const int t = 4 * ks + lq; return t >= tf ? 0.f : (ob_ >= 0 ? Xs[ob_ + t * V] : (ob_ == -1 ? 1.f : 0.f)); });
                    }
                }
            }
            __syncthreads();
            // ---- du = h + R u + r0: thread = (pixel, half of the 9 rows)
            {
                const int p = tid & 255, f0 = (tid >> 8) ? 5 : 0, f1 = (tid >> 8) ? SC : 5;
                if (p < px) {
                    float u[SC];
#pragma unroll
                    for (int f = 0; f < SC; ++f) u[f] = Us[f * FP + p];
                    for (int f = f0; f < f1; ++f) {
                        float d = DUs[f * FP + p] + Rs[SC * SC + f];          // (h of this pixel, staged above)
#pragma unroll
                        for (int f2 = 0; f2 < SC; ++f2) d = fmaf(Rs[f * SC + f2], u[f2], d);
                        DUs[f * FP + p] = d;
                    }
                }
            }
            __syncthreads();
            // ---- dP blocks: A = x[(k,t)][v] (transposed use), B = du_s[(k,t)][w]
#pragma unroll
            for (int i = 0; i < MAXU; ++i) {
                if (uS[i] >= 0) {
                    const int v = uV[i] * 16 + l16, w = uW[i] * 16 + l16;
                    const float *du = DUs + uS[i] * CIN * FP;
                    mfma_ksteps(accP[i], uK0[i], uK1[i],
                                [&](int ks) { const int off = rowtab[4 * ks + lq]; return (off >= 0 && v < V) ? Xs[off + v] : 0.f; },
                                [&](int ks) { const int off = rowtab[4 * ks + lq]; return (off >= 0 && w < V) ? du[off + w] : 0.f; });
                }
            }
        }
        // ---- the clip's dP (K parts meet in LDS) and X2
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (uS[i] >= 0) {
                const int w = uW[i] * 16 + l16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = uV[i] * 16 + 4 * lq + j;
                    if (v < V && w < V) atomicAdd(&dPs[(uS[i] * V + v) * V + w], accP[i][j]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            if (gI[i] >= 0) {
                const int b = gJ[i] * 16 + l16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int a = gI[i] * 16 + 4 * lq + j;
                    X2s[a * XP + b] = accG[i][j];
                    if (gI[i] != gJ[i]) X2s[b * XP + a] = accG[i][j];
                }
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
        __syncthreads();
        // dM_s[k][l] += sum_{v,w} dS_s[v,w] * X2[(k,v)][(l,w)]   ((3,.) = the ones row)
        for (int o = wave; o < S * C1 * C1; o += 8) {
            const int s = o / (C1 * C1), k = (o / C1) % C1, l = o % C1;
            float a = 0.f;
            for (int e = lane; e < VV; e += 64) {
                const int v = e / V, w = e - v * V;
                const int ia = k < CIN ? k * V + v : V3, ib = l < CIN ? l * V + w : V3;
                a = fmaf(dPs[s * VV + e], X2s[ia * XP + ib], a);
            }
            for (int sh = 32; sh > 0; sh >>= 1) a += __shfl_down(a, sh, 64);
            if (lane == 0) dMs[o] += a;
        }
    }
    __syncthreads();
    if (tid < S * C1 * C1) part_m[(size_t)blockIdx.x * S * C1 * C1 + tid] = dMs[tid];
}

// Sums the per-workgroup partials of the attention kernel in a fixed order.  Workgroup = 32 output elements x 8 groups of
// partials: thread (e = tid & 31, g = tid >> 5) adds partials g, g+8, ...; the 8 sub-sums are then added in order g = 0..7.
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
    bool ok = false, vec = false;
    int TF = 0, grid = 0, KS = 1, cls = 0;
    size_t lds_gather = 0, lds_attn = 0;
};

inline AgcnBwdPlan plan_agcn_bwd(int N, int Cin, int Cout, int T, int V, int S_) {
    AgcnBwdPlan pl;
    if (Cin != CIN || S_ != S || (Cout != 64 && Cout != 128 && Cout != 256) || V > 64) return pl;
    const size_t plane = (size_t)T * V;
    int TF = PXMAX / V;
    if (TF > T) TF = T;
    pl.vec = plane % 4 == 0;
    if (pl.vec)                                          // 16-byte loads: every chunk starts on a multiple of 4 pixels
        while (TF > 1 && (TF * V) % 4 != 0) --TF;
    if (pl.vec && (TF * V) % 4 != 0) pl.vec = false;
    pl.lds_gather = ((size_t)2 * 32 * FP + (size_t)NG * FP + (size_t)SC * FP + (size_t)S * V * V) * 4;
    {   // attention kernel: instantiation class by V, K split of its dP units, LDS
        const int nvb = (V + 15) / 16, NGB = (3 * V + 1 + 15) / 16, XP = NGB * 16 + 4;
        pl.KS = nvb == 1 ? 8 : (nvb == 2 ? 2 : 1);
        const int upw = (S * nvb * nvb * pl.KS + 7) / 8, gpw = (NGB * (NGB + 1) / 2 + 7) / 8;
        pl.cls = (upw <= 3 && gpw <= 2) ? 0 : ((upw <= 4 && gpw <= 7) ? 1 : -1);
        if (pl.cls < 0 || 3 * TF + 19 > ROWTAB) return pl;       // wider graphs: the generic GEMM chain serves them
        pl.lds_attn = ((size_t)(CIN + 2 * SC) * FP + 96 + 48 + ROWTAB + (size_t)2 * S * V * V + (size_t)NGB * 16 * XP) * 4;
    }
    if (pl.lds_gather > (size_t)kLdsBytes || pl.lds_attn > (size_t)kLdsBytes) return pl;
    if ((size_t)4 * Cout * NG > (size_t)2 * 32 * FP) return pl;   // the final reduction reuses the g tile
    pl.TF = TF;
    pl.grid = N < 256 ? N : 256;
    pl.ok = true;
    return pl;
}

// workspace of the fused path (bytes, each block 256-aligned):
//   [G: Cout*NG doubles][rr: 96 floats][dM sum: S*C1*C1 floats][wda: Cout*12 floats] | part_g [grid][Cout][NG] | hbuf [N][SC][T*V]
//   | part_pa [grid][S][V][V] | part_m [grid][S][C1][C1]
struct AgcnBwdWs {
    size_t g = 0, rr = 0, dm = 0, wda = 0, part_g = 0, hbuf = 0, part_pa = 0, part_m = 0, total = 0;
};
inline AgcnBwdWs ws_layout(const AgcnBwdPlan &pl, int N, int Cout, int T, int V) {
    AgcnBwdWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
    w.g = take((size_t)Cout * NG * sizeof(double));
    w.rr = take(96 * sizeof(float));
    w.dm = take((size_t)S * C1 * C1 * sizeof(float));
    w.wda = take((size_t)Cout * 12 * sizeof(float));
    w.part_g = take((size_t)pl.grid * Cout * NG * sizeof(float));
    w.hbuf = take((size_t)N * SC * T * V * sizeof(float));
    w.part_pa = take((size_t)pl.grid * S * V * V * sizeof(float));
    w.part_m = take((size_t)pl.grid * S * C1 * C1 * sizeof(float));
    w.total = off;
    return w;
}

}  // namespace

bool agcn_bwd_supported(int N, int Cin, int Cout, int T, int V, int S_) { return plan_agcn_bwd(N, Cin, Cout, T, V, S_).ok; }

size_t agcn_bwd_ws_bytes(int N, int Cin, int Cout, int T, int V, int S_) {
    const AgcnBwdPlan pl = plan_agcn_bwd(N, Cin, Cout, T, V, S_);
    return pl.ok ? ws_layout(pl, N, Cout, T, V).total : 0;
}

// y: the forward's output (the ReLU mask); stats: the forward's save_stats (4*Cout floats: batch mean / invstd of both
// BatchNorms, then the 63 feature moments as doubles — agcn_train.hip)
int launch_agcn_bwd(const float *x, const float *P, const float *A_eff, const float *y, const float *dy, const float *Wa,
                    const float *ba, const float *Wb, const float *bb, const float *Wd, const float *bd, const float *Wdown,
                    const float *bdown, const float *bn_w, const float *dbn_w, const float *stats, void *ws, float *dWa,
                    float *dba, float *dWb, float *dbb, float *dWd, float *dbd, float *dWdown, float *dbdown, float *dgamma,
                    float *dbeta, float *ddgamma, float *ddbeta, float *dPA, int N, int Cin, int Cout, int T, int V, int inter_c,
                    int S_, hipStream_t st) {
    const AgcnBwdPlan pl = plan_agcn_bwd(N, Cin, Cout, T, V, S_);
    if (!pl.ok)
        return fail(STGCN_ERR_UNSUPPORTED,
                    "agcn backward (moment form) covers Cin=3, 3 subsets, Cout in {64,128,256} with a down branch (got Cin=%d S=%d Cout=%d V=%d)",
                    Cin, S_, Cout, V);
    const AgcnBwdWs w = ws_layout(pl, N, Cout, T, V);
    char *base = (char *)ws;
    double *G = (double *)(base + w.g);
    float *rr = (float *)(base + w.rr), *dm_sum = (float *)(base + w.dm), *part_g = (float *)(base + w.part_g);
    float *wda = (float *)(base + w.wda);
    float *hbuf = (float *)(base + w.hbuf), *part_pa = (float *)(base + w.part_pa), *part_m = (float *)(base + w.part_m);
    const double *mom = (const double *)(stats + 4 * Cout);
    hipLaunchKernelGGL(agcn_bwd_prep_kernel, dim3(ceil_div(Cout * 12, 256)), dim3(256), 0, st, Wd, bn_w, stats + Cout, wda, Cout);
    STGCN_LAUNCH_CHECK("agcn_bwd_prep_kernel");
#define LAUNCH_GATHER(NOB, VEC)                                                                                        \
    do {                                                                                                               \
        STGCN_HIP_CHECK(allow_lds((agcn_bwd_gather_kernel<NOB, VEC>), pl.lds_gather));                                 \
        hipLaunchKernelGGL((agcn_bwd_gather_kernel<NOB, VEC>), dim3(pl.grid), dim3(NTB), pl.lds_gather, st, x, P, y, dy, wda, \
                           part_g, hbuf, N, T, V, pl.TF, ablate_mask());                                         \
    } while (0)
    if (pl.vec) {
        if (Cout == 64) LAUNCH_GATHER(2, true);
        else if (Cout == 128) LAUNCH_GATHER(4, true);
        else LAUNCH_GATHER(8, true);
    } else {
        if (Cout == 64) LAUNCH_GATHER(2, false);
        else if (Cout == 128) LAUNCH_GATHER(4, false);
        else LAUNCH_GATHER(8, false);
    }
#undef LAUNCH_GATHER
    STGCN_LAUNCH_CHECK("agcn_bwd_gather_kernel");
    hipLaunchKernelGGL(agcn_bwd_gsum_kernel, dim3(ceil_div(Cout * NG, 32)), dim3(1024), 0, st, part_g, pl.grid, Cout * NG, G);
    STGCN_LAUNCH_CHECK("agcn_bwd_gsum_kernel");
    hipLaunchKernelGGL(agcn_bwd_finalize_kernel, dim3(1), dim3(256), 0, st, G, mom, (double)N * T * V, Wd, bd, Wdown, bdown, bn_w,
                       dbn_w, stats, dWd, dbd, dWdown, dbdown, dgamma, dbeta, ddgamma, ddbeta, rr, Cout);
    STGCN_LAUNCH_CHECK("agcn_bwd_finalize_kernel");
    if (pl.cls == 0) {
        STGCN_HIP_CHECK(allow_lds((agcn_bwd_attn_kernel<3, 2>), pl.lds_attn));
        hipLaunchKernelGGL((agcn_bwd_attn_kernel<3, 2>), dim3(pl.grid), dim3(NTB), pl.lds_attn, st, x, P, A_eff, hbuf, rr, part_pa,
                           part_m, N, T, V, inter_c, pl.TF, pl.KS);
    } else {
        STGCN_HIP_CHECK(allow_lds((agcn_bwd_attn_kernel<4, 7>), pl.lds_attn));
        hipLaunchKernelGGL((agcn_bwd_attn_kernel<4, 7>), dim3(pl.grid), dim3(NTB), pl.lds_attn, st, x, P, A_eff, hbuf, rr, part_pa,
                           part_m, N, T, V, inter_c, pl.TF, pl.KS);
    }
    STGCN_LAUNCH_CHECK("agcn_bwd_attn_kernel");
    const int total = S * V * V + S * C1 * C1;
    hipLaunchKernelGGL((agcn_bwd_reduce_kernel<3, 3>), dim3(ceil_div(total, 32)), dim3(256), 0, st, (const float *)nullptr, part_pa,
                       part_m, pl.grid, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, dPA, dm_sum, 0, V);
    STGCN_LAUNCH_CHECK("agcn_bwd_reduce_kernel");
    hipLaunchKernelGGL((agcn_bwd_embed_kernel<3, 3>), dim3(1), dim3(256), 0, st, dm_sum, Wa, ba, Wb, bb, dWa, dba, dWb, dbb,
                       inter_c);
    STGCN_LAUNCH_CHECK("agcn_bwd_embed_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
