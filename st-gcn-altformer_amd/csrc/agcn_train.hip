// Training-mode forward of unit_agcn WITHOUT materialising its two pre-BatchNorm branches (stem shape class:
// C_in = 3, 3 subsets, down branch).
//
// Both branches are linear in a handful of per-pixel features,
//     zm[o] = sum_f Wm[o][f] * u[f] + bm[o]     (u = the 9 aggregated values x P_s,  Wm = [Wd_0|Wd_1|Wd_2],  bm = sum_s bd_s)
//     zd[o] = sum_k Wdown[o][k] * x[k] + bdown[o]
// so the batch statistics nn.BatchNorm2d takes over (N,T,V) (model/unit_agcn.py:91-92, 54-55 with self.training)
// follow from the first and second moments of u (9 + 45 numbers) and x (3 + 6):
//     E[zm[o]] = Wm[o].E[u] + bm[o],    E[zm[o]^2] = Wm[o]^T E[u u^T] Wm[o] + 2 bm[o] Wm[o].E[u] + bm[o]^2 .
// One pass over x and P accumulates the 63 moments (fp32 within a clip, fp64 across clips), a one-workgroup kernel turns
// them into every channel's (scale, shift), saved mean / invstd and running-buffer update, and the eval-mode expansion
// kernel then writes the module output directly.  Against the materialising path this drops two full-size tensors
// (2 x 4*C*T*V bytes per clip written, read twice) and three elementwise passes.
#include "common.h"

namespace stgcn {

namespace {

constexpr int NMOM = 9 + 45 + 3 + 6;   // u, uu^T (upper triangle), x, xx^T (upper triangle)
constexpr int MOM_NT = 1024;           // threads of the moments kernel: four waves per SIMD hide the LDS latency of the u loop
                                       // (with 256 threads — one wave per SIMD — the same code took 73 us for 256 clips)

template <int CIN, int S>
__global__ __launch_bounds__(MOM_NT) void agcn_moments_kernel(const float *__restrict__ x, const float *__restrict__ P,
                                                           double *__restrict__ part /* [grid][NMOM] */, int N, int T,
                                                           int V, int TF) {
    constexpr int SC = S * CIN;
    static_assert(SC == 9 && CIN == 3, "moment layout is written for 3 channels x 3 subsets");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Ps = sm;                    // [S][V][V]
    float *Xs = Ps + S * V * V;        // [CIN][TF*V]
    __shared__ float red[MOM_NT / 64][NMOM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PXM = TF * V;
    double total = 0.0;                // thread tid < NMOM: moment `tid` summed over this workgroup's clips
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        __syncthreads();
        const float *Pn = P + (size_t)n * S * V * V;
        for (int e = tid; e < S * V * V; e += MOM_NT) Ps[e] = Pn[e];
        const float *xn = x + (size_t)n * CIN * T * V;
        float m[NMOM];
#pragma unroll
        for (int i = 0; i < NMOM; ++i) m[i] = 0.f;
        for (int t0 = 0; t0 < T; t0 += TF) {
            const int px = min(TF, T - t0) * V;
            __syncthreads();
            for (int e = tid; e < CIN * px; e += MOM_NT) {
                const int k = e / px, p = e - k * px;
                Xs[k * PXM + p] = xn[((size_t)k * T + t0) * V + p];
            }
            __syncthreads();
            for (int p = tid; p < px; p += MOM_NT) {
                const int tt = p / V, w = p - tt * V;
                float u[SC];
#pragma unroll
                for (int f = 0; f < SC; ++f) u[f] = 0.f;
                for (int v = 0; v < V; ++v) {   // same order of operations as agcn_expand_small_kernel
                    float xv[CIN];
#pragma unroll
                    for (int k = 0; k < CIN; ++k) xv[k] = Xs[k * PXM + tt * V + v];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float pv = Ps[(s * V + v) * V + w];
#pragma unroll
                        for (int k = 0; k < CIN; ++k) u[s * CIN + k] = fmaf(xv[k], pv, u[s * CIN + k]);
                    }
                }
#pragma unroll
                for (int i = 0; i < SC; ++i) m[i] += u[i];
#pragma unroll
                for (int i = 0; i < SC; ++i)
#pragma unroll
                    for (int j = i; j < SC; ++j) {
                        const int q = SC + i * SC - i * (i - 1) / 2 + (j - i);      // upper triangle, row-major
                        m[q] = fmaf(u[i], u[j], m[q]);
                    }
                float xp[CIN];
#pragma unroll
                for (int k = 0; k < CIN; ++k) xp[k] = Xs[k * PXM + p];
#pragma unroll
                for (int i = 0; i < CIN; ++i) m[SC + 45 + i] += xp[i];
#pragma unroll
                for (int i = 0; i < CIN; ++i)
#pragma unroll
                    for (int j = i; j < CIN; ++j) {
                        const int q = SC + 45 + CIN + i * CIN - i * (i - 1) / 2 + (j - i);
                        m[q] = fmaf(xp[i], xp[j], m[q]);
                    }
            }
        }
        // block reduction of the clip's 63 moments
#pragma unroll
        for (int i = 0; i < NMOM; ++i) {
            float v = m[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) red[wave][i] = v;
        }
        __syncthreads();
        if (tid < NMOM) {                  // fixed order
            double c = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < MOM_NT / 64; ++w8) c += (double)red[w8][tid];
            total += c;
        }
    }
    if (tid < NMOM) part[(size_t)blockIdx.x * NMOM + tid] = total;
}

// moments -> per-channel batch statistics of both BatchNorms -> (scale, shift), saved mean/invstd, running buffers
template <int CIN, int S>
__global__ __launch_bounds__(1024) void agcn_moments_finalize_kernel(
    const double *__restrict__ part, int parts, double count, const float *__restrict__ Wd, const float *__restrict__ bd,
    const float *__restrict__ Wdown, const float *__restrict__ bdown, const float *__restrict__ bn_w,
    const float *__restrict__ bn_b, float *__restrict__ bn_rm, float *__restrict__ bn_rv, const float *__restrict__ dbn_w,
    const float *__restrict__ dbn_b, float *__restrict__ dbn_rm, float *__restrict__ dbn_rv, float momentum, float eps,
    float *__restrict__ s_m, float *__restrict__ t_m, float *__restrict__ s_d, float *__restrict__ t_d,
    float *__restrict__ save_stats /* 4*Cout + 128 floats, or NULL */, int Cout) {
    constexpr int SC = S * CIN;
    __shared__ double mom[NMOM];
    __shared__ double sub[16][64];
    const int tid = threadIdx.x;
    {   // 16 groups of partials per moment (16 waves: the loads of a group are a chain of round trips), then a
        // fixed-order sum of the 16 sub-sums
        const int i = tid & 63, grp = tid >> 6;
        double s = 0.0;
        if (i < NMOM)
            for (int p = grp; p < parts; p += 16) s += part[(size_t)p * NMOM + i];
        sub[grp][i] = s;
    }
    __syncthreads();
    if (tid < NMOM) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += sub[g][tid];
        mom[tid] = s / count;
        // the backward's moment form (agcn_backward.hip) reads the feature moments back: 63 doubles behind the 4*Cout floats
        if (save_stats) reinterpret_cast<double *>(save_stats + 4 * Cout)[tid] = mom[tid];
        if (save_stats && tid == 0)   // validity mark for the moment-form backward (it used to trust the caller)
            reinterpret_cast<unsigned *>(save_stats)[STGCN_MOMENTS_MARK_SLOT(Cout)] = STGCN_MOMENTS_MAGIC;
    }
    __syncthreads();
    const double *mu = mom, *muu = mom + SC, *mx = mom + SC + 45, *mxx = mom + SC + 45 + CIN;
    for (int o = tid; o < Cout; o += 1024) {
        double w[SC], b = 0.0;
        for (int s = 0; s < S; ++s) {
            b += (double)bd[s * Cout + o];
            for (int k = 0; k < CIN; ++k) w[s * CIN + k] = (double)Wd[((size_t)s * Cout + o) * CIN + k];
        }
        double lin = 0.0, quad = 0.0;
        int q = 0;
        for (int i = 0; i < SC; ++i) lin += w[i] * mu[i];
        for (int i = 0; i < SC; ++i)
            for (int j = i; j < SC; ++j, ++q) quad += (i == j ? 1.0 : 2.0) * w[i] * w[j] * muu[q];
        const double mean_m = lin + b;
        double var_m = quad + 2.0 * b * lin + b * b - mean_m * mean_m;
        double wd[CIN];
        for (int k = 0; k < CIN; ++k) wd[k] = (double)Wdown[o * CIN + k];
        const double bdn = (double)bdown[o];
        double lind = 0.0, quadd = 0.0;
        q = 0;
        for (int i = 0; i < CIN; ++i) lind += wd[i] * mx[i];
        for (int i = 0; i < CIN; ++i)
            for (int j = i; j < CIN; ++j, ++q) quadd += (i == j ? 1.0 : 2.0) * wd[i] * wd[j] * mxx[q];
        const double mean_d = lind + bdn;
        double var_d = quadd + 2.0 * bdn * lind + bdn * bdn - mean_d * mean_d;
        if (var_m < 0.0) var_m = 0.0;
        if (var_d < 0.0) var_d = 0.0;
        const double unb = count > 1.0 ? count / (count - 1.0) : 1.0;
        {
            const float inv = 1.f / sqrtf((float)var_m + eps), s = bn_w[o] * inv;
            s_m[o] = s;
            t_m[o] = bn_b[o] - (float)mean_m * s;
            if (save_stats) { save_stats[o] = (float)mean_m; save_stats[Cout + o] = inv; }
            bn_rm[o] = (1.f - momentum) * bn_rm[o] + momentum * (float)mean_m;
            bn_rv[o] = (1.f - momentum) * bn_rv[o] + momentum * (float)(var_m * unb);
        }
        {
            const float inv = 1.f / sqrtf((float)var_d + eps), s = dbn_w[o] * inv;
            s_d[o] = s;
            t_d[o] = dbn_b[o] - (float)mean_d * s;
            if (save_stats) { save_stats[2 * Cout + o] = (float)mean_d; save_stats[3 * Cout + o] = inv; }
            dbn_rm[o] = (1.f - momentum) * dbn_rm[o] + momentum * (float)mean_d;
            dbn_rv[o] = (1.f - momentum) * dbn_rv[o] + momentum * (float)(var_d * unb);
        }
    }
}

inline int moments_grid(int N) { return N < 256 ? N : 256; }

}  // namespace

bool agcn_moments_supported(int Cin, int V, int S) {
    return Cin == 3 && S == 3 && ((size_t)S * V * V + (size_t)Cin * (MOM_NT / V > 0 ? MOM_NT / V : 1) * V) * 4 <= (size_t)kLdsBytes && V <= MOM_NT;
}

size_t agcn_moments_ws_bytes(int N) { return (size_t)moments_grid(N) * NMOM * sizeof(double); }

// part: agcn_moments_ws_bytes(N) bytes of scratch.  Writes s_m, t_m, s_d, t_d (Cout each) and updates the running buffers.
int launch_agcn_moments(const float *x, const float *P, double *part, const float *Wd, const float *bd, const float *Wdown,
                        const float *bdown, const float *bn_w, const float *bn_b, float *bn_rm, float *bn_rv,
                        const float *dbn_w, const float *dbn_b, float *dbn_rm, float *dbn_rv, float momentum, float eps,
                        float *s_m, float *t_m, float *s_d, float *t_d, float *save_stats, int N, int Cin, int Cout, int T,
                        int V, int S, hipStream_t st) {
    if (!agcn_moments_supported(Cin, V, S))
        return fail(STGCN_ERR_UNSUPPORTED, "agcn moments: covers Cin=3, 3 subsets (got %d, %d, V=%d)", Cin, S, V);
    int TF = MOM_NT / V;
    if (TF < 1) TF = 1;
    if (TF > T) TF = T;
    const size_t lds = ((size_t)S * V * V + (size_t)Cin * TF * V) * 4;
    const int grid = moments_grid(N);
    STGCN_HIP_CHECK(allow_lds((agcn_moments_kernel<3, 3>), lds));
    hipLaunchKernelGGL((agcn_moments_kernel<3, 3>), dim3(grid), dim3(MOM_NT), lds, st, x, P, part, N, T, V, TF);
    STGCN_LAUNCH_CHECK("agcn_moments_kernel");
    hipLaunchKernelGGL((agcn_moments_finalize_kernel<3, 3>), dim3(1), dim3(1024), 0, st, part, grid, (double)N * T * V, Wd, bd,
                       Wdown, bdown, bn_w, bn_b, bn_rm, bn_rv, dbn_w, dbn_b, dbn_rm, dbn_rv, momentum, eps, s_m, t_m, s_d, t_d,
                       save_stats, Cout);
    STGCN_LAUNCH_CHECK("agcn_moments_finalize_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
