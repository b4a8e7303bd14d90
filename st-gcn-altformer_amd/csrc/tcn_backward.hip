// Backward of the temporal conv block Unit2D (model/net.py:47-57) in TRAINING mode, i.e. what autograd derives for
//   y = relu( BN_batch( conv_t(x) + b ) )                      (train_sttran.py:185-191)
// from dy:   g      = dy * [y > 0]
//            dgamma = sum g*xhat,  dbeta = sum g,   xhat = (z - mean) * invstd,  z = conv_t(x) + b (saved by the forward)
//            dz     = gamma*invstd * ( g - mean(g) - xhat*mean(g*xhat) )
//            db     = sum dz                     (analytically 0 behind a batch-statistics BatchNorm)
//            dW[o,c,k] = sum_{n,t,v} dz[n,o,t,v] * x[n,c,t*s+k-pad,v]                       ("wgrad")
//            dx[n,c,t,v] = sum_{o,k} W[o,c,k] * dz[n,o,(t+pad-k)/s,v]                        ("dgrad")
//
// Kernels here: the two elementwise BatchNorm+ReLU backward passes (also used by the graph-conv backward, which has
// two BatchNorms under one ReLU), the weight flip that turns dgrad (stride 1) into a forward temporal conv run by the
// existing matrix-core kernels, the matrix-core wgrad (bf16x3, same arithmetic contract as the forward), and plain
// VALU dgrad / wgrad kernels for every other shape (and as the fp32 cross-check).
#include <algorithm>

#include "bf16_common.h"

namespace stgcn {

namespace {

using namespace bf16k;

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm(batch statistics) + ReLU backward.  Pre-activation of element e of channel c:
//     a = za*sa[c] + ta[c]  +  ( zb*sb[c] + tb[c]   |  zb (identity residual, sb == NULL)  |  0 (zb == NULL) )
// g = dy where a > 0 else 0.   sums[c] += g,  sums[C+c] += g*xhat_a,  sums[2C+c] += g*xhat_b (second BatchNorm).
// ------------------------------------------------------------------------------------------------------------------
struct BnSide {
    const float *z, *scale, *shift, *mean, *invstd;
};

__device__ __forceinline__ void block_sum3(double &s0, double &s1, double &s2, double (&red)[3][4]) {
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_down(s0, o, 64);
        s1 += __shfl_down(s1, o, 64);
        s2 += __shfl_down(s2, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s0; red[1][w] = s1; red[2][w] = s2; }
    __syncthreads();
    s0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    s2 = red[2][0] + red[2][1] + red[2][2] + red[2][3];
}

// Per-channel constants of one workgroup: pre-activation, ReLU mask and xhat of an element from its raw values.
struct ChanCoef {
    float sa, ta, ma, ia, sb, tb, mb, ib;
    int mode;                                     // side b: 0 none, 1 identity residual, 2 second BatchNorm
    __device__ ChanCoef(const BnSide &a, const BnSide &b, int c) {
        sa = a.scale[c]; ta = a.shift[c]; ma = a.mean[c]; ia = a.invstd[c];
        mode = b.z == nullptr ? 0 : (b.scale != nullptr ? 2 : 1);
        sb = mode == 2 ? b.scale[c] : 1.f; tb = mode == 2 ? b.shift[c] : 0.f;
        mb = mode == 2 ? b.mean[c] : 0.f; ib = mode == 2 ? b.invstd[c] : 0.f;
    }
    __device__ __forceinline__ float masked(float za, float zb, float dy) const {   // the forward's own expression
        float v = fmaf(za, sa, ta);
        if (mode) v += fmaf(zb, sb, tb);
        return v > 0.f ? dy : 0.f;
    }
};

// grid = (chunks, C)
__global__ __launch_bounds__(256) void bn_relu_bwd_stats_kernel(BnSide a, BnSide b, const float *__restrict__ dy,
                                                                 double *__restrict__ sums, int N, int C, size_t plane) {
    const int c = blockIdx.y;
    const ChannelRows it(N, plane);
    const ChanCoef k(a, b, c);
    const bool two = k.mode == 2;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int n = it.n_lo; n < it.n_hi; ++n) {
        const size_t base = ((size_t)n * C + c) * plane;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;        // fp32 within one strip, fp64 across
        auto one = [&](float za, float zb, float d) {
            const float g = k.masked(za, zb, d);
            a0 += g;
            a1 = fmaf(g, (za - k.ma) * k.ia, a1);
            if (two) a2 = fmaf(g, (zb - k.mb) * k.ib, a2);
        };
        if (it.vec) {
            for (size_t p = threadIdx.x * 4; p < plane; p += 1024) {
                const float4 va = *reinterpret_cast<const float4 *>(a.z + base + p);
                const float4 vb = k.mode ? *reinterpret_cast<const float4 *>(b.z + base + p) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 vd = *reinterpret_cast<const float4 *>(dy + base + p);
                one(va.x, vb.x, vd.x); one(va.y, vb.y, vd.y); one(va.z, vb.z, vd.z); one(va.w, vb.w, vd.w);
            }
        } else {
            for (size_t p = threadIdx.x; p < plane; p += 256) one(a.z[base + p], k.mode ? b.z[base + p] : 0.f, dy[base + p]);
        }
        s0 += (double)a0;
        s1 += (double)a1;
        s2 += (double)a2;
    }
    __shared__ double red[3][4];
    block_sum3(s0, s1, s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], s0);
        atomicAdd(&sums[C + c], s1);
        if (two) atomicAdd(&sums[2 * C + c], s2);
    }
}

// dgamma / dbeta and the per-channel coefficients of pass 2:  coef[c] = gamma*invstd,  coef[C+c] = mean(g),
// coef[2C+c] = mean(g*xhat)
__global__ void bn_bwd_finalize_kernel(const double *__restrict__ sums, int which /* 1: side a, 2: side b */, double count,
                                       const float *__restrict__ gamma, const float *__restrict__ invstd,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ coef,
                                       int C, int frozen /* statistics are constants: no mean terms in pass 2 */) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double sg = sums[c], sgx = sums[which * C + c];
    dgamma[c] = (float)sgx;
    dbeta[c] = (float)sg;
    coef[c] = gamma[c] * invstd[c];
    coef[C + c] = frozen ? 0.f : (float)(sg / count);
    coef[2 * C + c] = frozen ? 0.f : (float)(sgx / count);
}

// pass 2: dza = coefa[c] * (g - mean(g) - xhat_a*mean(g*xhat_a)), same for side b; bsum[c] += sum dza (conv bias grad),
// bsum[C+c] += sum dzb.   grid = (chunks, C)
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(BnSide a, BnSide b, const float *__restrict__ dy,
                                                                 const float *__restrict__ coefa,
                                                                 const float *__restrict__ coefb, float *__restrict__ dza,
                                                                 float *__restrict__ dzb, double *__restrict__ bsum, int N,
                                                                 int C, size_t plane, float *__restrict__ gout) {
    const int c = blockIdx.y;
    const ChannelRows it(N, plane);
    const ChanCoef k(a, b, c);
    const bool two = dzb != nullptr;
    const float ka = coefa[c], c1 = coefa[C + c], c2a = coefa[2 * C + c];
    const float kb = two ? coefb[c] : 0.f, c2b = two ? coefb[2 * C + c] : 0.f;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int n = it.n_lo; n < it.n_hi; ++n) {
        const size_t base = ((size_t)n * C + c) * plane;
        float a0 = 0.f, a1 = 0.f;
        auto one = [&](float za, float zb, float d, float &oa, float &ob, float &og) {
            const float g = k.masked(za, zb, d);
            og = g;                               // gout: the masked cotangent itself = dL/dx of an identity residual ("+ x")
            oa = ka * (g - c1 - (za - k.ma) * k.ia * c2a);
            a0 += oa;
            if (two) {
                ob = kb * (g - c1 - (zb - k.mb) * k.ib * c2b);
                a1 += ob;
            }
        };
        if (it.vec) {
            for (size_t p = threadIdx.x * 4; p < plane; p += 1024) {
                const float4 va = *reinterpret_cast<const float4 *>(a.z + base + p);
                const float4 vb = k.mode ? *reinterpret_cast<const float4 *>(b.z + base + p) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 vd = *reinterpret_cast<const float4 *>(dy + base + p);
                float4 oa, ob = make_float4(0.f, 0.f, 0.f, 0.f), og;
                one(va.x, vb.x, vd.x, oa.x, ob.x, og.x); one(va.y, vb.y, vd.y, oa.y, ob.y, og.y);
                one(va.z, vb.z, vd.z, oa.z, ob.z, og.z); one(va.w, vb.w, vd.w, oa.w, ob.w, og.w);
                *reinterpret_cast<float4 *>(dza + base + p) = oa;
                if (two) *reinterpret_cast<float4 *>(dzb + base + p) = ob;
                if (gout) *reinterpret_cast<float4 *>(gout + base + p) = og;
            }
        } else {
            for (size_t p = threadIdx.x; p < plane; p += 256) {
                float oa, ob = 0.f, og;
                one(a.z[base + p], k.mode ? b.z[base + p] : 0.f, dy[base + p], oa, ob, og);
                dza[base + p] = oa;
                if (two) dzb[base + p] = ob;
                if (gout) gout[base + p] = og;
            }
        }
        s0 += (double)a0;
        s1 += (double)a1;
    }
    if (bsum == nullptr) return;
    __shared__ double red[3][4];
    block_sum3(s0, s1, s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&bsum[c], s0);
        if (two) atomicAdd(&bsum[C + c], s1);
    }
}

__global__ void doubles_to_floats_kernel(const double *__restrict__ src, float *__restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

// ------------------------------------------------------------------------------------------------------------------
// dgrad, stride 1: dx = conv_t(dz, Wf) with Wf[c][o][k] = W[o][c][K-1-k]  -> the forward kernels do the contraction
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample2_kernel(const float *__restrict__ dz, float *__restrict__ dzu, size_t total,
                                                         int Tout, int T, int V) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const size_t row = e / ((size_t)T * V);
    const int r = (int)(e - row * (size_t)T * V), t = r / V, v = r - t * V;
    dzu[e] = ((t & 1) == 0 && (t >> 1) < Tout) ? dz[(row * Tout + (t >> 1)) * V + v] : 0.f;
}

__global__ void weight_flip_kernel(const float *__restrict__ W, float *__restrict__ Wf, int Cout, int Cin, int K) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Cout * Cin * K) return;
    const int k = e % K, o = (e / K) % Cout, c = e / (K * Cout);
    Wf[e] = W[((size_t)o * Cin + c) * K + (K - 1 - k)];
}

// dgrad, any stride (plain VALU): one workgroup = (256 input pixels of a clip, one input channel c)
__global__ __launch_bounds__(256) void tcn_dgrad_valu_kernel(const float *__restrict__ dz, const float *__restrict__ W,
                                                              float *__restrict__ dx, int Cin, int Cout, int T, int V,
                                                              int K, int stride, int Tout) {
    extern __shared__ float wl[];  // [Cout][K] taps of input channel c
    const int c = blockIdx.y, n = blockIdx.z;
    for (int e = threadIdx.x; e < Cout * K; e += 256) wl[e] = W[((size_t)(e / K) * Cin + c) * K + (e % K)];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= T * V) return;
    const int t = p / V, v = p - t * V, pad = (K - 1) / 2;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int num = t + pad - k;
        if (num < 0 || num % stride != 0) continue;
        const int to = num / stride;
        if (to >= Tout) continue;
        const float *dzp = dz + ((size_t)n * Cout * Tout + to) * V + v;
        for (int o = 0; o < Cout; ++o) acc = fmaf(wl[o * K + k], dzp[(size_t)o * Tout * V], acc);
    }
    dx[((size_t)n * Cin + c) * T * V + p] = acc;
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad, plain VALU (any shape; the fp32 cross-check).  Workgroup = 16 output channels x 16 input channels, all K taps
// (K <= KMAXV), looping over (clip, frame chunk) units blockIdx.z, +gridDim.z, ...; partial sums -> atomicAdd.
// ------------------------------------------------------------------------------------------------------------------
constexpr int KMAXV = 9;
constexpr int TFV = 4;  // output frames per unit

__global__ __launch_bounds__(256) void tcn_wgrad_valu_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                                                              float *__restrict__ dW, int N, int Cin, int Cout, int T,
                                                              int V, int K, int stride, int Tout) {
    extern __shared__ float sm[];
    const int pad = (K - 1) / 2;
    const int FR = (TFV - 1) * stride + K;       // input frames a unit touches
    const int pd = TFV * V + 1, px = FR * V + 1;  // row pitches (odd-ish: spreads the 16 rows over banks)
    float *dzs = sm;                              // [16][pd]
    float *xs = sm + 16 * pd;                     // [16][px]
    const int ol = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const int o0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
    const int chunks = (Tout + TFV - 1) / TFV;
    float acc[KMAXV];
#pragma unroll
    for (int k = 0; k < KMAXV; ++k) acc[k] = 0.f;
    for (int u = blockIdx.z; u < N * chunks; u += gridDim.z) {
        const int n = u / chunks, to0 = (u - n * chunks) * TFV;
        const int f0 = to0 * stride - pad;
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * TFV * V; e += 256) {
            const int r = e / (TFV * V), q = e - r * TFV * V, tt = q / V;
            const bool ok = o0 + r < Cout && to0 + tt < Tout;
            dzs[r * pd + q] = ok ? dz[((size_t)n * Cout + o0 + r) * Tout * V + (size_t)to0 * V + q] : 0.f;
        }
        for (int e = threadIdx.x; e < 16 * FR * V; e += 256) {
            const int r = e / (FR * V), q = e - r * FR * V, f = f0 + q / V;
            const bool ok = c0 + r < Cin && f >= 0 && f < T;
            xs[r * px + q] = ok ? x[((size_t)n * Cin + c0 + r) * T * V + (size_t)f0 * V + q] : 0.f;
        }
        __syncthreads();
        for (int tt = 0; tt < TFV; ++tt)
            for (int v = 0; v < V; ++v) {
                const float d = dzs[ol * pd + tt * V + v];
                const float *xr = xs + cl * px + tt * stride * V + v;
#pragma unroll
                for (int k = 0; k < KMAXV; ++k)
                    if (k < K) acc[k] = fmaf(d, xr[k * V], acc[k]);
            }
    }
    if (o0 + ol < Cout && c0 + cl < Cin)
        for (int k = 0; k < K; ++k) atomicAdd(&dW[((size_t)(o0 + ol) * Cin + c0 + cl) * K + k], acc[k]);
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad on the bf16 matrix cores (stride 1, K <= 9, Cout % 128 == 0 or Cout == 64, Cin % 32 == 0).
//
// GEMM view:  dW[o][(c,k)] = sum_p A[o][p] * B[p][(c,k)],  A = dz,  B[p][(c,k)] = x[c][p + (k-pad)*V]  — the
// contraction runs over PIXELS, which are contiguous in memory for a fixed channel in both operands, i.e. both are
// already "k-major" as the MFMA wants them.  The only obstacle is the tap shift of k*V pixels (44 B for V = 22): LDS
// fragment reads must be 16-byte aligned.  The tiles are therefore stored with the frame pitch padded to Vp = a
// multiple of 8 pixels (22 -> 24, zeros in between): a tap shift is then k*Vp pixels = a multiple of 16 B.
//
// Workgroup (512 threads) = 128 output channels x 32 input channels x all taps; wave w: 32-channel block ob = w & 3 of
// dz, tap group tg = w >> 2 (taps [0,KH) / [KH,K), KH = ceil(K/2)) — the two waves of a SIMD hold the two groups.
// It loops over (clip, TFM output frames) units u = blockIdx.z, +gridDim.z, ...: both tiles are loaded (fp32 -> bf16
// hi/lo, 8-pixel units, 16-byte LDS stores), then TFM*Vp/16 k-steps of KH x 3 MFMAs.  The unit after is prefetched
// into registers during the k-steps.  Partial sums of the workgroup go to part[blockIdx.z] and are summed afterwards
// in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------------------------
constexpr int TFM_MAX = 4;   // output frames per unit (the plan picks 4, 2 or 1)
constexpr int KHMAX = 5;     // taps per wave
constexpr int WG_THREADS = 512;

template <int TERMS, int UB /* B units per thread: 3 or 5 */>
__global__ __launch_bounds__(WG_THREADS) void tcn_wgrad_mfma_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                                                                     float *__restrict__ part, int N, int Cin, int Cout,
                                                                     int T, int V, int K, int Vp, int pitchA, int pitchB,
                                                                     int TFM /* output frames per unit */) {
    extern __shared__ __attribute__((aligned(16))) char smw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ob = wave & 3, tg = wave >> 2;
    const int KH = (K + 1) / 2;
    const int k_lo = tg * KH;                     // this wave's taps k_lo .. k_lo+KHMAX-1; those >= K are computed on real
    const int pad = (K - 1) / 2;                  // (finite) frames and dropped at the end: the tap loop has no branches
    // XCD-aware block mapping: hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2).
    // The gridDim.x channel groups of one split read the SAME dz tile; with the natural numbering they sat on different
    // XCDs and every one of them missed.  Re-deal: id L -> (split zi, channel group cg) such that all channel groups of a
    // split share L % 8.  (needs gridDim.z % 8 == 0; otherwise the natural numbering)
    int cg = blockIdx.x, zi = blockIdx.z;
    if ((gridDim.z & 7) == 0 && !(TFM & 0x100)) {
        const int L = blockIdx.x + gridDim.x * blockIdx.z;
        zi = (L & 7) + 8 * (L / (8 * gridDim.x));
        cg = (L >> 3) % gridDim.x;
    }
    TFM &= 0xff;                                  // (bit 8: diagnostic builds, STGCN_ABLATE=1 — natural block numbering)
    const int c0 = cg * 32, o0 = blockIdx.y * 128;
    const int upf = Vp / 8;                       // 8-pixel units per frame
    const int FRB = TFM + 2 * KHMAX - 1;          // input frames per unit (taps 0 .. 2*KHMAX-1)
    const int unitsA = 128 * TFM * upf, unitsB = 32 * FRB * upf;
    // LDS: A hi | A lo | B hi | B lo    (row pitches in bytes, 16 B x odd)
    char *Ahi = smw, *Alo = Ahi + 128 * pitchA, *Bhi = Alo + 128 * pitchA, *Blo = Bhi + 32 * pitchB;
    const int chunks = (T + TFM - 1) / TFM;       // stride 1: Tout == T
    const int nunits = N * chunks;

    f32x16 acc[KHMAX];
#pragma unroll
    for (int k = 0; k < KHMAX; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // staging: thread handles the 8-pixel units e = tid, tid+512, ... of A and of B.  All loads are unconditional on
    // clamped addresses (then zeroed by a select), so the prefetch is a straight run of loads without branches.
    constexpr int UA = 3;                         // A units per thread (the plan checks unitsA <= UA*512, unitsB <= UB*512)
    float pa[UA][8], pb[UB][8];
    int a_d[UA], b_d[UB];                         // unit descriptors: row << 16 | frame << 8 | first joint
#pragma unroll
    for (int i = 0; i < UA; ++i) {
        const int e = min(tid + i * WG_THREADS, unitsA - 1);
        const int row = e / (TFM * upf), q = e - row * TFM * upf, tt = q / upf;
        a_d[i] = row << 16 | tt << 8 | (q - tt * upf) * 8;
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
        const int e = min(tid + i * WG_THREADS, unitsB - 1);
        const int row = e / (FRB * upf), q = e - row * FRB * upf, ff = q / upf;
        b_d[i] = row << 16 | ff << 8 | (q - ff * upf) * 8;
    }
#define D_ROW(d) ((d) >> 16)
#define D_FR(d) (((d) >> 8) & 0xff)
#define D_V0(d) ((d) & 0xff)
    // Loads go through buffer resources (one per operand and clip): a lane needs ONE offset register per 8-pixel unit,
    // the 8 loads differ in the instruction's immediate offset, and a unit outside the clip (frame < 0 or >= T) is given
    // an offset past num_records, which the hardware answers with zeros — no branches, no per-load address registers.
    const unsigned clipA = (unsigned)((size_t)Cout * T * V * 4), clipB = (unsigned)((size_t)Cin * T * V * 4);
    constexpr unsigned OOB = 0x7ffffff0u;
    auto fetch = [&](int u) {
        const int n = __builtin_amdgcn_readfirstlane(u / chunks);
        const int t0 = (u - n * chunks) * TFM;
        const __amdgpu_buffer_rsrc_t ra =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dz + (size_t)n * Cout * T * V), 0, clipA, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)n * Cin * T * V), 0, clipB, 0x00020000);
#pragma unroll
        for (int i = 0; i < UA; ++i) {
            const int t = t0 + D_FR(a_d[i]);
            const unsigned off = t < T ? (unsigned)((((o0 + D_ROW(a_d[i])) * T + t) * V + D_V0(a_d[i])) * 4) : OOB;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float val = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, off + 4 * j, 0, 0));
                pa[i][j] = (D_V0(a_d[i]) + j < V) ? val : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int f = t0 - pad + D_FR(b_d[i]);
            const unsigned off = (f >= 0 && f < T) ? (unsigned)((((c0 + D_ROW(b_d[i])) * T + f) * V + D_V0(b_d[i])) * 4) : OOB;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float val = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, off + 4 * j, 0, 0));
                pb[i][j] = (D_V0(b_d[i]) + j < V) ? val : 0.f;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < UA; ++i) {
            const int e = tid + i * WG_THREADS;
            if (e < unitsA) {
                const int q = D_FR(a_d[i]) * upf + (D_V0(a_d[i]) >> 3);
                uint4 hi, lo;
                split8(pa[i], hi, lo);
                *reinterpret_cast<uint4 *>(Ahi + D_ROW(a_d[i]) * pitchA + q * 16) = hi;
                if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(Alo + D_ROW(a_d[i]) * pitchA + q * 16) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int e = tid + i * WG_THREADS;
            if (e < unitsB) {
                const int q = D_FR(b_d[i]) * upf + (D_V0(b_d[i]) >> 3);
                uint4 hi, lo;
                split8(pb[i], hi, lo);
                *reinterpret_cast<uint4 *>(Bhi + D_ROW(b_d[i]) * pitchB + q * 16) = hi;
                if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(Blo + D_ROW(b_d[i]) * pitchB + q * 16) = lo;
            }
        }
    };

    const int ksteps = TFM * Vp / 16;             // the plan guarantees TFM*Vp % 16 == 0
    const int h = lane >> 5;
    const char *arow = Ahi + (ob * 32 + (lane & 31)) * pitchA + h * 16;
    const char *brow = Bhi + (lane & 31) * pitchB + h * 16 + k_lo * Vp * 2;
    const int aoff_lo = 128 * pitchA, boff_lo = 32 * pitchB;
    const int tapb = Vp * 2;                      // bytes between consecutive taps of a B fragment
    const int ntap2 = (tg == 0 ? KH : K - KH) - 3;    // taps of this wave's group beyond the first three (K = 9: 2 and 1)

    int u = zi;
    if (u < nunits) fetch(u);
    for (; u < nunits; u += gridDim.z) {
        __syncthreads();                          // previous unit fully consumed
        stash();
        __syncthreads();
        if (u + (int)gridDim.z < nunits) fetch(u + gridDim.z);   // in flight during the MFMAs below
        for (int ks = 0; ks < ksteps; ++ks) {
            // fragment reads of a group of taps first, then its MFMAs (which wait with counted lgkmcnt, in issue order);
            // two groups (3 + 2 taps) keep the live fragments at 32 registers
            uint4 ah, al;
            ah = *reinterpret_cast<const uint4 *>(arow + ks * 32);
            al = ah;
            if constexpr (TERMS == 3) al = *reinterpret_cast<const uint4 *>(arow + aoff_lo + ks * 32);
#define WGRAD_GROUP(K0, KN)                                                                                          \
    {                                                                                                                \
        uint4 bh[KN], bl[KN];                                                                                        \
        _Pragma("unroll") for (int kk = 0; kk < KN; ++kk) {                                                          \
            bh[kk] = *reinterpret_cast<const uint4 *>(brow + ks * 32 + (K0 + kk) * tapb);                            \
            bl[kk] = bh[kk];                                                                                         \
            if constexpr (TERMS == 3) bl[kk] = *reinterpret_cast<const uint4 *>(brow + boff_lo + ks * 32 + (K0 + kk) * tapb); \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int kk = 0; kk < KN; ++kk) {                                                          \
            if constexpr (TERMS == 3) {                                                                              \
                acc[K0 + kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah),                \
                                                                       __builtin_bit_cast(bf16x8, bl[kk]), acc[K0 + kk], 0, 0, 0); \
                acc[K0 + kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al),                \
                                                                       __builtin_bit_cast(bf16x8, bh[kk]), acc[K0 + kk], 0, 0, 0); \
            }                                                                                                        \
            acc[K0 + kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah),                    \
                                                                   __builtin_bit_cast(bf16x8, bh[kk]), acc[K0 + kk], 0, 0, 0); \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
            WGRAD_GROUP(0, 3)
            // (the second tap group of K = 9 has four taps, not five: its fifth was computed for nothing — 10 % of the MFMAs)
            if (ntap2 >= 2) WGRAD_GROUP(3, 2)
            else if (ntap2 == 1) WGRAD_GROUP(3, 1)
#undef WGRAD_GROUP
        }
    }
#undef D_ROW
#undef D_FR
#undef D_V0
    // D[row = o][col = c]: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float *dst = part + (size_t)zi * Cout * Cin * K;
#pragma unroll
    for (int kk = 0; kk < KHMAX; ++kk) {
        const int k = k_lo + kk;
        if (kk < KH && k < K) {                   // (taps beyond this wave's group or beyond K were computed for nothing)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (o < Cout) dst[((size_t)o * Cin + c0 + (lane & 31)) * K + k] = acc[kk][r];   // (rows >= Cout: dz read as zeros)
            }
        }
    }
}

// Fixed-order sum of the per-split partials: thread (q = tid & 63, grp = tid >> 6) adds the 16-byte element q of splits
// grp, grp+4, ...; the four sub-sums are then added in order.  (One thread per float walking all splits in sequence took
// 19 us of a 4.5 ms training step.)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float *__restrict__ part, float *__restrict__ out, int parts, size_t n) {
    __shared__ float4 sub[4][64];
    const int q = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const size_t i = ((size_t)blockIdx.x * 64 + q) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i + 3 < n) {
        for (int p = grp; p < parts; p += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(part + (size_t)p * n + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    } else if (i < n) {
        float *sv = &s.x;
        for (int p = grp; p < parts; p += 4)
            for (int e = 0; e < 4 && i + e < n; ++e) sv[e] += part[(size_t)p * n + i + e];
    }
    sub[grp][q] = s;
    __syncthreads();
    if (grp != 0 || i >= n) return;
    float4 r = sub[0][q];
    for (int g = 1; g < 4; ++g) { r.x += sub[g][q].x; r.y += sub[g][q].y; r.z += sub[g][q].z; r.w += sub[g][q].w; }
    if (i + 3 < n) *reinterpret_cast<float4 *>(out + i) = r;
    else { const float *rv = &r.x; for (int e = 0; e < 4 && i + e < n; ++e) out[i + e] = rv[e]; }
}

struct WgradPlan {
    bool ok = false;
    int Vp = 0, tfm = 0, ub = 0, pitchA = 0, pitchB = 0, splits = 0;
    size_t lds = 0;
};

inline WgradPlan plan_wgrad(int N, int Cin, int Cout, int T, int V, int K, int stride) {
    WgradPlan pl;
    if (stride != 1 || K > 2 * KHMAX - 1 || K < 1 || (Cout % 128 != 0 && Cout != 64) || Cin % 32 != 0) return pl;   // (64: half-empty tile)
    if ((K & 1) == 0) return pl;   // the matrix-core kernel indexes dz and x with one frame count (Tout == T: odd K only)
    if ((size_t)(Cin > Cout ? Cin : Cout) * T * V * 4 >= ((size_t)1 << 31)) return pl;   // per-clip buffer resources
    const int Vp = (V + 7) / 8 * 8;
    const int upf = Vp / 8;
    int TFM = 0;
    for (int t = TFM_MAX; t >= 1 && !TFM; t >>= 1)
        if ((t * Vp) % 16 == 0 && 128 * t * upf <= 3 * WG_THREADS && 32 * (t + 2 * KHMAX - 1) * upf <= 5 * WG_THREADS) TFM = t;
    if (!TFM) return pl;
    const int FRB = TFM + 2 * KHMAX - 1;          // the kernel stages taps 0 .. 2*KHMAX-1 whatever K is
    auto odd16 = [](int bytes) { int u = (bytes + 15) / 16; return (u | 1) * 16; };   // 16 B x odd: conflict-free rows
    pl.Vp = Vp;
    pl.tfm = TFM;
    pl.ub = 32 * FRB * upf <= 3 * WG_THREADS ? 3 : 5;
    pl.pitchA = odd16(TFM * Vp * 2);
    pl.pitchB = odd16(FRB * Vp * 2);
    pl.lds = (size_t)2 * 128 * pl.pitchA + (size_t)2 * 32 * pl.pitchB;
    if (pl.lds > (size_t)kLdsBytes) return pl;
    const int chunks = (T + TFM - 1) / TFM;
    const int wgs = (Cin / 32) * ceil_div(Cout, 128);
    int splits = 256 / wgs;                       // about one workgroup per CU
    if (splits < 1) splits = 1;
    if (splits > N * chunks) splits = N * chunks;
    pl.splits = splits;
    pl.ok = true;
    return pl;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
// sums: 3*C doubles (zeroed here).  Side b: z == NULL (none), scale == NULL (identity residual), else second BatchNorm.
int launch_bn_relu_bwd_stats(const float *za, const float *sa, const float *ta, const float *ma, const float *ia,
                             const float *zb, const float *sb, const float *tb, const float *mb, const float *ib,
                             const float *dy, double *sums, int N, int C, size_t plane, hipStream_t st) {
    STGCN_HIP_CHECK(hipMemsetAsync(sums, 0, sizeof(double) * 3 * C, st));
    const BnSide a{za, sa, ta, ma, ia}, b{zb, sb, tb, mb, ib};
    hipLaunchKernelGGL(bn_relu_bwd_stats_kernel, dim3(bn_chunks(N, plane, C), C), dim3(256), 0, st, a, b, dy, sums, N,
                       C, plane);
    STGCN_LAUNCH_CHECK("bn_relu_bwd_stats_kernel");
    return STGCN_OK;
}

int launch_bn_bwd_finalize(const double *sums, int which, double count, const float *gamma, const float *invstd,
                           float *dgamma, float *dbeta, float *coef, int C, hipStream_t st, bool frozen) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, sums, which, count, gamma, invstd,
                       dgamma, dbeta, coef, C, frozen ? 1 : 0);
    STGCN_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    return STGCN_OK;
}

// bsum: 2*C doubles (zeroed here) or NULL
int launch_bn_relu_bwd_apply(const float *za, const float *sa, const float *ta, const float *ma, const float *ia,
                             const float *zb, const float *sb, const float *tb, const float *mb, const float *ib,
                             const float *dy, const float *coefa, const float *coefb, float *dza, float *dzb, double *bsum,
                             int N, int C, size_t plane, hipStream_t st, float *gout) {
    if (bsum) STGCN_HIP_CHECK(hipMemsetAsync(bsum, 0, sizeof(double) * 2 * C, st));
    const BnSide a{za, sa, ta, ma, ia}, b{zb, sb, tb, mb, ib};
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(bn_chunks(N, plane, C), C), dim3(256), 0, st, a, b, dy, coefa,
                       coefb, dza, dzb, bsum, N, C, plane, gout);
    STGCN_LAUNCH_CHECK("bn_relu_bwd_apply_kernel");
    return STGCN_OK;
}

int launch_doubles_to_floats(const double *src, float *dst, int n, hipStream_t st) {
    hipLaunchKernelGGL(doubles_to_floats_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, src, dst, n);
    STGCN_LAUNCH_CHECK("doubles_to_floats_kernel");
    return STGCN_OK;
}

// dzu[r][t][v] = dz[r][t/2][v] for even t < 2*Tout, 0 elsewhere (r = (clip, channel)): the stride-2 block's backward as the
// stride-1 block's — dx = conv_t(dzu, flipped W), dW[k] = sum_t dzu[t] x[t+k-pad] — so that both run on the matrix cores
int launch_upsample2(const float *dz, float *dzu, size_t rows, int Tout, int T, int V, hipStream_t st) {
    const size_t total = rows * (size_t)T * V;
    hipLaunchKernelGGL(upsample2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dz, dzu, total, Tout, T, V);
    STGCN_LAUNCH_CHECK("upsample2_kernel");
    return STGCN_OK;
}

int launch_weight_flip(const float *W, float *Wf, int Cout, int Cin, int K, hipStream_t st) {
    hipLaunchKernelGGL(weight_flip_kernel, dim3(ceil_div(Cout * Cin * K, 256)), dim3(256), 0, st, W, Wf, Cout, Cin, K);
    STGCN_LAUNCH_CHECK("weight_flip_kernel");
    return STGCN_OK;
}

int launch_tcn_dgrad_valu(const float *dz, const float *W, float *dx, int N, int Cin, int Cout, int T, int V, int K,
                          int stride, int Tout, hipStream_t st) {
    if (N > 65535 || Cin > 65535) return fail(STGCN_ERR_UNSUPPORTED, "tcn dgrad: N=%d Cin=%d exceed the grid", N, Cin);
    const size_t lds = (size_t)Cout * K * 4;
    if (lds > (size_t)kLdsBytes) return fail(STGCN_ERR_UNSUPPORTED, "tcn dgrad: Cout*K=%d does not fit LDS", Cout * K);
    STGCN_HIP_CHECK(allow_lds(tcn_dgrad_valu_kernel, lds));
    hipLaunchKernelGGL(tcn_dgrad_valu_kernel, dim3(ceil_div(T * V, 256), Cin, N), dim3(256), lds, st, dz, W, dx, Cin, Cout, T,
                       V, K, stride, Tout);
    STGCN_LAUNCH_CHECK("tcn_dgrad_valu_kernel");
    return STGCN_OK;
}

bool tcn_wgrad_mfma_supported(int N, int Cin, int Cout, int T, int V, int K, int stride) {
    return plan_wgrad(N, Cin, Cout, T, V, K, stride).ok;
}

size_t tcn_wgrad_ws_bytes(int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) {
        const WgradPlan pl = plan_wgrad(N, Cin, Cout, T, V, K, stride);
        size_t splits = pl.ok ? (size_t)pl.splits : 0;
        if (tcn_wgrad_v6_supported(N, Cin, Cout, T, V, K, stride)) splits = std::max(splits, (size_t)tcn_wgrad_v6_splits(N, Cin, Cout, T));
        if (splits) return splits * Cout * Cin * K * sizeof(float);
    }
    return 0;
}

// dW (Cout,Cin,K) = sum over pixels; `part` = tcn_wgrad_ws_bytes scratch (matrix-core path)
int launch_tcn_wgrad(const float *dz, const float *x, float *dW, float *part, int N, int Cin, int Cout, int T, int V, int K,
                     int stride, int Tout, unsigned flags, hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const WgradPlan pl = plan_wgrad(N, Cin, Cout, T, V, K, stride);
    // one wave per SIMD, ring input tile (tcn_wgrad_v6.hip) where it covers the shape  (diagnostic builds: STGCN_ABLATE=2 off)
    if ((math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) && part != nullptr && tcn_wgrad_v6_supported(N, Cin, Cout, T, V, K, stride) &&
        !(ablate_mask() & 2)) {
        const int rc = launch_tcn_wgrad_v6(dz, x, part, N, Cin, Cout, T, V, K, flags, st);
        if (rc != STGCN_OK) return rc;
        const size_t n = (size_t)Cout * Cin * K;
        hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, dW,
                           tcn_wgrad_v6_splits(N, Cin, Cout, T), n);
        STGCN_LAUNCH_CHECK("sum_partials_kernel");
        return STGCN_OK;
    }
    if ((math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) && pl.ok && part != nullptr) {
        const dim3 grid(Cin / 32, ceil_div(Cout, 128), pl.splits);
#define LAUNCH_WGRAD(TERMS, UBN)                                                                                       \
    do {                                                                                                               \
        STGCN_HIP_CHECK(allow_lds((tcn_wgrad_mfma_kernel<TERMS, UBN>), pl.lds));                                       \
        hipLaunchKernelGGL((tcn_wgrad_mfma_kernel<TERMS, UBN>), grid, dim3(WG_THREADS), pl.lds, st, dz, x, part, N, Cin, Cout, \
                           T, V, K, pl.Vp, pl.pitchA, pl.pitchB, pl.tfm | ((ablate_mask() & 1) ? 0x100 : 0));                                                       \
    } while (0)
        if (math == STGCN_MATH_BF16X3) { if (pl.ub == 3) LAUNCH_WGRAD(3, 3); else LAUNCH_WGRAD(3, 5); }
        else { if (pl.ub == 3) LAUNCH_WGRAD(1, 3); else LAUNCH_WGRAD(1, 5); }
#undef LAUNCH_WGRAD
        STGCN_LAUNCH_CHECK("tcn_wgrad_mfma_kernel");
        const size_t n = (size_t)Cout * Cin * K;
        hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, dW, pl.splits, n);   // 256 floats per workgroup
        STGCN_LAUNCH_CHECK("sum_partials_kernel");
        return STGCN_OK;
    }
    if (K > KMAXV) return fail(STGCN_ERR_UNSUPPORTED, "tcn wgrad: K=%d > %d", K, KMAXV);
    const int FR = (TFV - 1) * stride + K;
    const size_t lds = ((size_t)16 * (TFV * V + 1) + (size_t)16 * (FR * V + 1)) * 4;
    if (lds > (size_t)kLdsBytes) return fail(STGCN_ERR_UNSUPPORTED, "tcn wgrad: V=%d stride=%d does not fit LDS", V, stride);
    STGCN_HIP_CHECK(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)Cout * Cin * K, st));
    const int chunks = ceil_div(Tout, TFV);
    int splits = 2048 / (ceil_div(Cin, 16) * ceil_div(Cout, 16));
    if (splits < 1) splits = 1;
    if (splits > N * chunks) splits = N * chunks;
    STGCN_HIP_CHECK(allow_lds(tcn_wgrad_valu_kernel, lds));
    hipLaunchKernelGGL(tcn_wgrad_valu_kernel, dim3(ceil_div(Cin, 16), ceil_div(Cout, 16), splits), dim3(256), lds, st, dz, x,
                       dW, N, Cin, Cout, T, V, K, stride, Tout);
    STGCN_LAUNCH_CHECK("tcn_wgrad_valu_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
