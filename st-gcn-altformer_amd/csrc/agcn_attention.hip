// K1 — adjacency attention of the adaptive graph conv (model/unit_agcn.py:81-85):
//
//   P[n,s,v,w] = softmax_v( sum_{c<inter_c,t<T} a_s[c,t,v] * b_s[c,t,w] / (inter_c*T) ) + A_eff[s,v,w]
//   a_s = Wa_s x + ba_s,  b_s = Wb_s x + bb_s   (1x1 convs)
//
// Two kernels:
//  * attention_folded_kernel (small Cin, the stem has Cin = 3).  With x~ = [x; 1] the Gram is a
//    bilinear form  S_s[v,w] = sum_{k,l} M_s[k,l] * G[(k,v),(l,w)],  M_s = Wa~_s^T Wb~_s ((Cin+1)^2),
//    G = X~^T X~ with X~ the T x (Cin*V+1) matrix of one clip (last column = 1).  G is shared by
//    all subsets, so one workgroup per clip streams x once (coalesced), keeps a register tile of G
//    per thread, and the 32-channel embeddings are never materialised.
//  * attention_generic_kernel (any Cin): embeddings built chunk-by-chunk in LDS, one workgroup per
//    (clip, subset).
//
// Both are HBM-read-bound in principle (x is read once: 4*Cin*T*V bytes per clip) and tiny next to
// the temporal conv; P (N,S,V,V) is written once.
#include "bf16_common.h"

namespace stgcn {

namespace {

constexpr int MS_FLOATS = 128;  // LDS reserved for the S*(Cin+1)^2 bilinear matrices
constexpr int GB = 16;          // Gram block edge: one v_mfma_f32_16x16x4_f32 accumulator

using f32x4 = __attribute__((ext_vector_type(4))) float;

// Column soft-max of Sm[s][v][w] over v, then P = soft + A_eff, left in Sm (the caller copies it to global memory
// coalesced and, in the folded kernel, feeds the feature pass from it).  SIXTEEN lanes per (s,w) column, each holding
// up to 4 of its V <= 64 entries, max / sum through width-16 shuffles: with one thread per column (the first form) 66 of
// 1,024 threads walked 22-long dependent chains of LDS reads and expf — 12 % of the kernel.
__device__ __forceinline__ void softmax_columns(float *Sm, const float *__restrict__ A_eff, int S, int V, int s0,
                                                int tid, int nthreads) {
    const int sub = tid & 15;
    for (int c = tid >> 4; c < S * V; c += nthreads >> 4) {
        const int s = c / V, w = c - s * V;
        float *col = Sm + (size_t)s * V * V + w;
        const float *Ae = A_eff + (size_t)(s0 + s) * V * V + w;
        float val[4], adj[4];
        float m = -__builtin_huge_valf();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = sub + 16 * i;
            adj[i] = v < V ? Ae[v * V] : 0.f;            // issued first: the only global latency of this phase
            val[i] = v < V ? col[v * V] : -__builtin_huge_valf();
            m = fmaxf(m, val[i]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            val[i] = (sub + 16 * i < V) ? expf(val[i] - m) : 0.f;
            sum += val[i];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = sub + 16 * i;
            if (v < V) col[v * V] = val[i] / sum + adj[i];
        }
    }
}

// NW waves per workgroup (16 or 8).  The Gram G = X~^T X~ runs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact
// fp32 fma chains, the VALU rate without the VALU's operand traffic): G is cut into 16x16 blocks, only the upper triangle
// is computed, wave w owns blocks w, w+NW, ... (at most MAXB) and walks the clip 4 frames per MFMA.  Both operands of a
// block are the same kind of read — lane l takes X~[t0 + (l>>4)][16*I + (l&15)] — and with a row pitch of 16 (mod 32)
// floats the four frame rows of a fragment fall on disjoint banks: no conflicts (the register-tiled VALU form spent
// 36 % of its LDS cycles in bank conflicts and 23 % of the kernel in this phase).
template <int MAXB, int NW>
__global__ __launch_bounds__(64 * NW) void attention_folded_kernel(
    const float *__restrict__ x, const float *__restrict__ A_eff, const float *__restrict__ Wa,
    const float *__restrict__ ba, const float *__restrict__ Wb, const float *__restrict__ bb,
    float *__restrict__ P, float *__restrict__ feat, uint4 *__restrict__ pfrag, int Cin, int T, int V, int inter_c,
    int S, int TC, int Rp, int feat_slice_off, int sq_behind, int xsc, int xsp, float *__restrict__ xcopy,
    unsigned long long *dbg, int pf_v0, float *__restrict__ ybound) {
    // x element (channel k, pixel p) of a clip sits at k*xsc + p*xsp: (T*V, 1) for (N,Cin,T,V), (1, Cin) for (N,T,V,Cin).
    // xcopy (optional): channel-major copy of x for kernels downstream that read it in that layout.
#ifdef STGCN_ABLATION  // in-kernel cycle stamps (diagnostic builds only)
#define K1_STAMP(i) if (dbg && threadIdx.x == 0 && blockIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); dbg[1024 + (i)] = t_; }
#else
#define K1_STAMP(i)
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTH = 64 * NW;
    K1_STAMP(0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    const int C1 = Cin + 1;
    const int R = Cin * V + 1;  // columns of X~ (last one is the constant 1)
    float *Ms = smem;
    float *U = smem + MS_FLOATS;  // Xs[TC][Rp] while accumulating, then Gs[R][R] + Sm[S][V][V]

    // M_s[k][l] = sum_c Wa~_s[c][k] * Wb~_s[c][l]   (bias folded in as column Cin).  The embedding weights are first
    // copied to LDS (coalesced) so the 48 dot products do not each chain 64 dependent global loads.
    {
        const int rows = S * inter_c;
        float *Wl = U;  // [2][rows][C1]  (U is free until the first clip chunk is staged)
        for (int e = tid; e < 2 * rows * C1; e += NTH) {
            const int ab = e / (rows * C1), rc = e - ab * rows * C1;
            const int row = rc / C1, k = rc - row * C1;
            const float *W = ab ? Wb : Wa;
            const float *bv = ab ? bb : ba;
            Wl[e] = (k < Cin) ? W[row * Cin + k] : bv[row];
        }
        __syncthreads();
        for (int e = tid; e < S * C1 * C1; e += NTH) {
            const int s = e / (C1 * C1), kl = e - s * C1 * C1;
            const int k = kl / C1, l = kl - k * C1;
            float acc = 0.f;
            for (int c = 0; c < inter_c; ++c) {
                const int row = s * inter_c + c;
                acc = fmaf(Wl[row * C1 + k], Wl[(rows + row) * C1 + l], acc);
            }
            Ms[e] = acc;
        }
        // (the first __syncthreads() of the chunk loop below orders these reads before U is overwritten)
    }

    // this wave's Gram blocks: linear index b = wave + i*NW over the upper triangle, rows first
    const int nb = (R + GB - 1) / GB;
    const int nblk = nb * (nb + 1) / 2;
    int bI[MAXB], bJ[MAXB];
    f32x4 acc[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        int b = wave + i * NW, I = 0;
        if (b >= nblk) b = 0;                    // (idle slot: computes block 0 again, never stored)
        while (b >= nb - I) { b -= nb - I; ++I; }
        bI[i] = I;
        bJ[i] = I + b;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const float *xn = x + (size_t)n * Cin * T * V;
    const int CV = Cin * V;
    float xmax = 0.f;                      // max |x| of the clip (ybound)
    K1_STAMP(1)
    for (int t0 = 0; t0 < T; t0 += TC) {
        const int tc = min(TC, T - t0);
        __syncthreads();  // previous chunk fully consumed
        // padding columns (zeros) and the constant-1 column: TC * (Rp - CV) entries
        {
            const int padw = Rp - CV;
            for (int e = tid; e < TC * padw; e += NTH) {
                const int tt = e / padw, r = CV + (e - tt * padw);
                U[tt * Rp + r] = (r == CV && tt < tc) ? 1.f : 0.f;
            }
        }
        // data columns: channel k, frame tt, joint v  <-  x[k][t0+tt][v]; (tt,v) advanced incrementally (no division).
        // The loads of a group of 4 strides x all channels are issued together and stored afterwards: as a plain
        // load -> store loop the fill was a chain of ~12 HBM round trips (the first fill measured 21k cycles of 120k).
        {
            const int stepT = NTH / V, stepV = NTH - stepT * V;
            int tt = tid / V, v = tid - tt * V;
            constexpr int GRP = 4, KMAX = 4;         // (Cin <= 4 on this path)
            for (int e0 = tid; e0 < TC * V; e0 += GRP * NTH) {
                float xv[KMAX][GRP];
                int tts[GRP], vs[GRP];
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    tts[i] = tt;
                    vs[i] = v;
                    tt += stepT;
                    v += stepV;
                    if (v >= V) { v -= V; ++tt; }
                }
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
#pragma unroll
                    for (int i = 0; i < GRP; ++i) {
                        const int e = e0 + i * NTH;
                        const bool ok = k < Cin && e < TC * V && tts[i] < tc;
                        const size_t idx = (size_t)min(k, Cin - 1) * xsc + ((size_t)t0 * V + min(e, tc * V - 1)) * xsp;
                        const float val = xn[idx];
                        xv[k][i] = ok ? val : 0.f;
                        xmax = fmaxf(xmax, fabsf(xv[k][i]));
                    }
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
#pragma unroll
                    for (int i = 0; i < GRP; ++i) {
                        const int e = e0 + i * NTH;
                        if (k < Cin && e < TC * V) {
                            U[tts[i] * Rp + k * V + vs[i]] = xv[k][i];
                            if (xcopy && tts[i] < tc) xcopy[((size_t)n * Cin + k) * T * V + (size_t)t0 * V + e] = xv[k][i];
                        }
                    }
            }
        }
        __syncthreads();
        K1_STAMP(2)
        {
            const float *frag = U + (lane >> 4) * Rp + (lane & 15);
            for (int tt0 = 0; tt0 < tc; tt0 += 4) {     // (rows tc .. TC-1 of the chunk are zero: TC is a multiple of 4)
                const float *row = frag + tt0 * Rp;
#pragma unroll
                for (int i = 0; i < MAXB; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[bI[i] * GB], row[bJ[i] * GB], acc[i], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    K1_STAMP(3)
    float *Gs = U;           // [R][R]
    float *Sm = U + R * R;   // [S][V][V]
    // accumulators -> Gs, both triangles: lane holds G[16I + 4*(l>>4) + r][16J + (l&15)], r = 0..3
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        if (wave + i * NW < nblk) {
            const int gj = bJ[i] * GB + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = bI[i] * GB + 4 * (lane >> 4) + r;
                if (gi < R && gj < R) {
                    Gs[gi * R + gj] = acc[i][r];
                    Gs[gj * R + gi] = acc[i][r];
                }
            }
        }
    }
    __syncthreads();
    K1_STAMP(4)
    const float denom = (float)(inter_c * T);  // A1.size(-1) at unit_agcn.py:84
    for (int e = tid; e < S * V * V; e += NTH) {
        const int s = e / (V * V), vw = e - s * V * V;
        const int v = vw / V, w = vw - v * V;
        const float *M = Ms + s * C1 * C1;
        float sacc = 0.f;
        for (int k = 0; k < C1; ++k) {
            const int gr = (k < Cin) ? k * V + v : CV;
            for (int l = 0; l < C1; ++l) {
                const int gc = (l < Cin) ? l * V + w : CV;
                sacc = fmaf(M[k * C1 + l], Gs[gr * R + gc], sacc);
            }
        }
        Sm[e] = sacc / denom;
    }
    __syncthreads();
    K1_STAMP(5)
    softmax_columns(Sm, A_eff, S, V, 0, tid, NTH);
    __syncthreads();
    {   // P (N,S,V,V): coalesced copy of the finished matrices
        float *Pn = P + (size_t)n * S * V * V;
        for (int e = tid; e < S * V * V; e += NTH) Pn[e] = Sm[e];
    }
    K1_STAMP(6)
    // Optional: what bounds every value the fused kernel's producer can emit for this clip (stem_f16mx.hip scales its
    // e4m3 operands by it): max|x| and, per subset, max|x| * max_w sum_v |P_s[v][w]|  (|u_s| = |sum_v x P_s[v][w]| <= that).
    if (ybound != nullptr) {
        float *xm_s = Ms + 64, *cs_s = Ms + 96;     // (the bilinear matrices at the head of Ms are no longer needed)
        float m = xmax;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
        if (lane == 0) xm_s[wave] = m;
        if (tid < 4) cs_s[tid] = 0.f;
        __syncthreads();
        if (tid < S * V) {                 // one thread per (subset, column): its abs-sum, then the subset's maximum
            const int s = tid / V, w = tid - s * V;
            float a = 0.f;
            for (int v = 0; v < V; ++v) a += fabsf(Sm[(s * V + v) * V + w]);
            atomicMax(reinterpret_cast<unsigned *>(&cs_s[s]), __float_as_uint(a));     // (non-negative floats order like their bits)
        }
        __syncthreads();
        if (tid == 0) {
            float xm = 0.f;
            for (int i = 0; i < NW; ++i) xm = fmaxf(xm, xm_s[i]);
            ybound[n * 4 + 0] = xm;
            for (int s = 0; s < 3; ++s) ybound[n * 4 + 1 + s] = s < S ? xm * cs_s[s] : 0.f;
        }
    }

    // Optional feature pass for the fused stem (Cin = 3, S = 3 only): per pixel (t,w) the 12 graph-conv
    // features [u_0, u_1, u_2, x] with u_s[k] = sum_v x[k,t,v] P_s[v,w]  (model/unit_agcn.py:87-88), then a
    // constant 1 that multiplies the folded bias and 3 zeros; the 16 values are split into bf16 hi + lo residual
    // (the operand form of the stem kernel's matrix-core producer) -> feat[n][t*V+w] = 64 B per pixel, coalesced:
    // [hi f0-7][hi f8-15][lo f0-7][lo f8-15].
    // Optional (instead of the feature pass): the attention matrices as matrix-core B fragments for the stem kernel that
    // computes the features itself (stem_bf16_v4.hip, FK form): u_s[k][t][w] = sum_v x[k][t][v] P_s[v][w] is a
    // (4 frames x 4 channels) x 32 x 16 product per v_mfma_f32_16x16x32_bf16, B[v][w'] = P_s[v][16h + w'].  Fragment
    // f = (s*2 + h)*2 + (0: bf16 hi, 1: lo residual), lane l holds v = 8*(l>>4) .. +7 of column w = 16h + (l&15); zeros
    // past V.  12 KiB per clip instead of 64 B per pixel (253 KiB at T=180, V=22).
    if (pfrag != nullptr && pf_v0 == 0) {      // (host side guarantees S == 3, V <= 32)
        uint4 *pf = pfrag + (size_t)n * 12 * 64;
        for (int e = tid; e < 6 * 64; e += NTH) {
            const int sh = e >> 6, l = e & 63, s = sh >> 1, h = sh & 1;
            const int w = 16 * h + (l & 15), v0 = 8 * (l >> 4);
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = (w < V && v0 + j < V) ? Sm[(s * V + v0 + j) * V + w] : 0.f;
            uint4 hi, lo;
            bf16k::split8(pv, hi, lo);
            pf[(sh * 2 + 0) * 64 + l] = hi;
            pf[(sh * 2 + 1) * 64 + l] = lo;
        }
    }
    // Wide frames (32 < V <= 64, the two-hand graph): the fused kernel splits the JOINT axis in two column halves —
    // joints [0, V0) and [V0, V) — that it treats as two narrow clips (the temporal conv never mixes joints), while the
    // aggregation still sums over all V joints: two k-steps of 32.  Fragment ((((half*3 + s)*2 + h)*2 + ks)*2 + img):
    // lane l holds v = 32*ks + 8*(l>>4) .. +7 of column w = j0(half) + 16h + (l&15); zeros past V / past the half.
    // 48 KiB per clip.
    if (pfrag != nullptr && pf_v0 > 0) {
        uint4 *pf = pfrag + (size_t)n * 48 * 64;
        for (int e = tid; e < 24 * 64; e += NTH) {
            const int f = e >> 6, l = e & 63;                      // f = ((half*3 + s)*2 + h)*2 + ks
            const int ks = f & 1, h = (f >> 1) & 1, hs = f >> 2, s = hs % 3, half = hs / 3;
            const int j0 = half ? pf_v0 : 0, vh = half ? V - pf_v0 : pf_v0;
            const int wl = 16 * h + (l & 15), w = j0 + wl, v0 = 32 * ks + 8 * (l >> 4);
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = (wl < vh && v0 + j < V) ? Sm[(s * V + v0 + j) * V + w] : 0.f;
            uint4 hi, lo;
            bf16k::split8(pv, hi, lo);
            pf[(f * 2 + 0) * 64 + l] = hi;
            pf[(f * 2 + 1) * 64 + l] = lo;
        }
    }
    if (feat == nullptr) return;
    // LDS operands of the feature loop, interleaved so that one ds_read_b128 brings what three ds_read_b32 did:
    //   Sq[v][w] = (P_0, P_1, P_2, -)[v][w]  and  Xq[pixel] = (x_0, x_1, x_2, -).  Xq lives in the dead Gram region;
    //   Sq behind Gs and Sm when the LDS budget allows (sq_behind), else at the head of the Gram region as well
    //   (wide frames: the 139 x 139 Gram of V = 46 leaves no room behind it).
    const int sq_off = sq_behind ? ((R * R + S * V * V + 3) & ~3) : 0;
    float4 *Sq = reinterpret_cast<float4 *>(U + sq_off);
    float4 *Xq = reinterpret_cast<float4 *>(U) + (sq_behind ? 0 : V * V);
    const int TCF = (R * R - (sq_behind ? 0 : 4 * V * V)) / (4 * V);   // (R*R = (3V+1)^2 > 9 V^2: at least one frame)
    for (int e = tid; e < V * V; e += NTH) Sq[e] = make_float4(Sm[e], Sm[V * V + e], Sm[2 * V * V + e], 0.f);
    // Each lane builds one 64-byte feature row; storing it directly is 16 B per lane at a 64-B stride (store-issue
    // bound, measured ~4 B/clk/CU).  Instead the wave parks its 64 rows (4 KiB) in a private LDS slice and writes
    // them back out lane-linear: four fully coalesced 1-KiB stores.
    uint4 *slice = reinterpret_cast<uint4 *>(smem + feat_slice_off) + (tid >> 6) * 256;
    uint4 *fo = reinterpret_cast<uint4 *>(feat) + (size_t)n * T * V * 4;
    for (int t0 = 0; t0 < T; t0 += TCF) {
        const int tcf = min(TCF, T - t0);
        const int px = tcf * V;
        __syncthreads();  // Sq complete / previous chunk consumed
        for (int e = tid; e < px; e += NTH) {
            const size_t g = ((size_t)t0 * V + e) * xsp;
            Xq[e] = make_float4(xn[g], xn[(size_t)xsc + g], xn[2 * (size_t)xsc + g], 0.f);
        }
        __syncthreads();
        for (int p0 = (tid & ~63); p0 < px; p0 += NTH) {   // 64 consecutive pixels per wave
            const int p = p0 + lane;
            if (p < px) {
                const int tt = p / V, w = p - tt * V;
                float u[9];
#pragma unroll
                for (int f = 0; f < 9; ++f) u[f] = 0.f;
                const float4 *xr = Xq + tt * V;
                const float4 *pr = Sq + w;
#pragma unroll 2
                for (int v = 0; v < V; ++v) {
                    const float4 xv = xr[v];
                    const float4 pv = pr[v * V];
                    u[0] = fmaf(xv.x, pv.x, u[0]); u[1] = fmaf(xv.y, pv.x, u[1]); u[2] = fmaf(xv.z, pv.x, u[2]);
                    u[3] = fmaf(xv.x, pv.y, u[3]); u[4] = fmaf(xv.y, pv.y, u[4]); u[5] = fmaf(xv.z, pv.y, u[5]);
                    u[6] = fmaf(xv.x, pv.z, u[6]); u[7] = fmaf(xv.y, pv.z, u[7]); u[8] = fmaf(xv.z, pv.z, u[8]);
                }
                const float4 xp = Xq[p];
                const float fa[8] = {u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]};
                const float fb[8] = {u[8], xp.x, xp.y, xp.z, 1.f, 0.f, 0.f, 0.f};
                uint4 ha, la, hb, lb;  // bf16 hi / lo residual of features 0-7 and 8-15
                bf16k::split8(fa, ha, la);
                bf16k::split8(fb, hb, lb);
                slice[lane * 4 + 0] = ha;
                slice[lane * 4 + 1] = hb;
                slice[lane * 4 + 2] = la;
                slice[lane * 4 + 3] = lb;
            }
            // wave-private: same wave reads what it wrote (the compiler's lgkmcnt wait orders it)
            const int nrow = min(64, px - p0);
            uint4 *dst = fo + ((size_t)t0 * V + p0) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = j * 64 + lane;
                if (idx < nrow * 4) dst[idx] = slice[idx];
            }
        }
    }
    K1_STAMP(7)
}

// Generic Cin: one workgroup per (subset, clip).
template <int MAXIT>
__global__ __launch_bounds__(256) void attention_generic_kernel(
    const float *__restrict__ x, const float *__restrict__ A_eff, const float *__restrict__ Wa,
    const float *__restrict__ ba, const float *__restrict__ Wb, const float *__restrict__ bb,
    float *__restrict__ P, int Cin, int T, int V, int inter_c, int S, int TC, int xsc, int xsp,
    float *__restrict__ xcopy) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int s = blockIdx.x, n = blockIdx.y;
    const int PXC = TC * V;
    float *Xs = smem;                    // [Cin][PXC]
    float *As = Xs + (size_t)Cin * PXC;  // [inter_c][PXC]
    float *Bs = As + (size_t)inter_c * PXC;
    const float *xn = x + (size_t)n * Cin * T * V;
    const float *wa = Wa + (size_t)s * inter_c * Cin;
    const float *wb = Wb + (size_t)s * inter_c * Cin;

    float acc[MAXIT];
    int pv[MAXIT], pw[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int e = tid + it * 256;
        const int ee = (e < V * V) ? e : 0;
        pv[it] = ee / V;
        pw[it] = ee % V;
        acc[it] = 0.f;
    }
    for (int t0 = 0; t0 < T; t0 += TC) {
        const int tc = min(TC, T - t0);
        const int px = tc * V;
        __syncthreads();
        for (int e = tid; e < Cin * px; e += 256) {
            const int k = e / px, p = e - k * px;
            const float xv = xn[(size_t)k * xsc + ((size_t)t0 * V + p) * xsp];
            Xs[k * PXC + p] = xv;
            if (xcopy && s == 0) xcopy[((size_t)n * Cin + k) * T * V + (size_t)t0 * V + p] = xv;
        }
        __syncthreads();
        for (int e = tid; e < inter_c * px; e += 256) {
            const int c = e / px, p = e - c * px;
            float a = ba[s * inter_c + c], b = bb[s * inter_c + c];
            for (int k = 0; k < Cin; ++k) {
                const float xv = Xs[k * PXC + p];
                a = fmaf(wa[c * Cin + k], xv, a);
                b = fmaf(wb[c * Cin + k], xv, b);
            }
            As[c * PXC + p] = a;
            Bs[c * PXC + p] = b;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            float a = acc[it];
            for (int c = 0; c < inter_c; ++c)
                for (int tt = 0; tt < tc; ++tt)
                    a = fmaf(As[c * PXC + tt * V + pv[it]], Bs[c * PXC + tt * V + pw[it]], a);
            acc[it] = a;
        }
    }
    __syncthreads();
    float *Sm = smem;  // [V][V]
    const float denom = (float)(inter_c * T);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int e = tid + it * 256;
        if (e < V * V) Sm[e] = acc[it] / denom;
    }
    __syncthreads();
    softmax_columns(Sm, A_eff, 1, V, s, tid, 256);
    __syncthreads();
    float *Pn = P + ((size_t)n * S + s) * V * V;
    for (int e = tid; e < V * V; e += 256) Pn[e] = Sm[e];
}

// Generic Cin on the fp32 matrix cores (Cin % 4 == 0, inter_c = 8, 16, 32 or 64: every deeper TCN_GCN_unit layer of the
// ST-TR family, model/ST_TR/ST_TR_new.py:355).  One workgroup (8 waves) per CLIP, x streamed once in frame chunks through
// LDS and shared by the three subsets; per chunk and subset
//   E = [Wa_s; Wb_s] x + bias      16 x 16 blocks (rows, pixels), K = Cin: a wave owns one row block, keeps its weight fragments
//                                  in registers (loaded once per chunk and subset, KS = Cin/4 of them) and walks pixel blocks
//   S_s[v][w] += sum_{(c,t)} a[c][t,v] b[c][t,w]   16 x 16 blocks (v, w) x K parts dealt to the waves, K = inter_c * frames,
//                                  accumulators in registers across the chunks
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation), then the column soft-max of the VALU kernel.
// Row pitch of the LDS tiles = 16 mod 64 floats: the four k rows of a B fragment read land on disjoint banks.
template <int KS, int MAXB>
__global__ __launch_bounds__(512) void attention_generic_mfma_kernel(
    const float *__restrict__ x, const float *__restrict__ A_eff, const float *__restrict__ Wa,
    const float *__restrict__ ba, const float *__restrict__ Wb, const float *__restrict__ bb,
    float *__restrict__ P, int Cin, int T, int V, int inter_c, int S, int TC, int PXC, int xsc, int xsp,
    float *__restrict__ xcopy, unsigned long long *dbg) {
#ifdef STGCN_ABLATION   // per-wave phase clocks (diagnostic builds; tools/stamps_k1g.py): 0 staging, 1 barrier waits, 2 embeddings,
                        // 3 Gram, 4 tail (K-part sums, soft-max, store), 5 whole kernel
#define KG_STAMP(var) unsigned long long var = 0; if (dbg) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); }
#define KG_ACC(slot, a, b) if (dbg) { tsum[slot] += (b) - (a); }
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0};
#else
#define KG_STAMP(var)
#define KG_ACC(slot, a, b)
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, lq = lane >> 4;
    const int n = blockIdx.x;
    KG_STAMP(t_begin)
    // gridDim.y == S: one workgroup per (clip, subset) — batches with fewer clips than CUs (the deeper layers' 64-clip steps
    // ran on 64 of 256 CUs); x is then staged once per subset (from L2 for the second and third).  gridDim.y == 1: all subsets.
    const int only = gridDim.y > 1 ? (int)blockIdx.y : -1;
    if (only > 0) xcopy = nullptr;
    float *Xs = smem;                                   // [Cin][PXC]
    float *Es = Xs + (size_t)Cin * PXC;                 // [2*inter_c][PXC]: a rows, then b rows (one subset at a time)
    const float *xn = x + (size_t)n * Cin * T * V;
    const int nvb = (V + 15) / 16, nblk = nvb * nvb, R2 = 2 * inter_c, nrb = R2 / 16;
    const int R8 = nrb < 8 ? nrb : 8, pstep = 8 / R8;   // waves per row block = pstep
    const int KSP = nblk >= 8 ? 1 : 8 / nblk;           // K parts of a Gram block (nblk = 1, 4: 8, 2)
    const int nunits = nblk * KSP;
    const int ks_n = Cin / 4;
    const int icm = inter_c - 1, icl = 31 - __builtin_clz(inter_c);

    f32x4 acc[3][MAXB];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < MAXB; ++i) acc[s][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight fragments and biases of this wave's row block for one subset
    // Channel c sits in LDS row rho(c) = c with its two low 2-bit fields swapped (C_in % 16 == 0): k-step ks = 4j + i of lane
    // group lq then multiplies channel 16j + 4lq + i, so a lane's fragments of four k-steps are ONE 16-byte load of its weight
    // row (16 instead of 64 load instructions per lane at 256 channels; as dwords — 16 rows x 16 bytes per instruction — their
    // issue alone was a third of the kernel), and the B reads keep the conflict-free row 4 ks + lq.
    const bool perm = (Cin & 15) == 0;
    auto rho = [&](int c) { return perm ? ((c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3)) : c; };
    float wf[KS], bias[4];
    auto load_w = [&](int s) __attribute__((always_inline)) {
        const int rb = wave % R8, row = rb * 16 + l16;
        const float *wr = row < inter_c ? Wa + ((size_t)s * inter_c + row) * Cin : Wb + ((size_t)s * inter_c + row - inter_c) * Cin;
        if (perm) {
#pragma unroll
            for (int j = 0; j < KS / 4; ++j) {
                const f32x4 w4 = 16 * j < Cin ? *reinterpret_cast<const f32x4 *>(wr + 16 * j + 4 * lq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[4 * j + i] = w4[i];
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[ks] = ks < ks_n ? wr[4 * ks + lq] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rb * 16 + 4 * lq + i;
            bias[i] = r < inter_c ? ba[s * inter_c + r] : bb[s * inter_c + r - inter_c];
        }
    };
    load_w(only >= 0 ? only : 0);
    // x of the NEXT chunk travels HBM -> registers while this chunk is multiplied (the staging was 23 - 40 % of the kernel once
    // the embeddings were fixed: tools/stamps_k1g.py): row wave + 8 r, pixels lane + 64 q.  The register tile is sized for the
    // chunks the LDS budget gives at C_in = 4 KS (many rows <-> few pixels); other shapes stage in place as before.
    // (KS = 64 — 256 channels, 44-pixel chunks — stays with in-place staging: its 32 extra registers spilled.)
    constexpr int RW = KS == 64 ? 1 : KS / 2, QM = KS == 16 ? 6 : (KS == 32 ? 3 : 1);
    const bool pre = KS != 64 && Cin <= 8 * RW && TC * V <= 64 * QM;
    float xr[RW][QM];
    auto xfetch = [&](int t0f) __attribute__((always_inline)) {
        const int pxf = min(TC, T - t0f) * V;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int k = wave + 8 * r;
            const float *xrow = xn + (size_t)k * xsc + (size_t)t0f * V * xsp;
#pragma unroll
            for (int q = 0; q < QM; ++q) {
                const int p = lane + 64 * q;
                xr[r][q] = (k < Cin && p < pxf) ? xrow[(size_t)p * xsp] : 0.f;
            }
        }
    };
    if (pre) xfetch(0);
    for (int t0 = 0; t0 < T; t0 += TC) {
        const int tc = min(TC, T - t0), px = tc * V, npb = (px + 15) / 16;
        KG_STAMP(t_b0)
        __syncthreads();
        KG_STAMP(t_s0)
        KG_ACC(1, t_b0, t_s0)
        if (pre) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int k = wave + 8 * r;
#pragma unroll
                for (int q = 0; q < QM; ++q) {
                    const int p = lane + 64 * q;
                    if (k < Cin && p < px) {
                        Xs[rho(k) * PXC + p] = xr[r][q];
                        if (xcopy) xcopy[((size_t)n * Cin + k) * T * V + (size_t)t0 * V + p] = xr[r][q];
                    }
                }
            }
            if (t0 + TC < T) xfetch(t0 + TC);
        } else {
        // a wave per channel row, lanes along the pixels: no index division.  RB rows x QP 64-pixel pieces per trip, all RB*QP
        // loads issued before the first LDS store: one load per trip left the whole staging a chain of exposed memory round
        // trips — 32 per chunk at 256 channels, where a chunk is two frames (44 pixels: one piece per row) and the chunks are
        // many: 36 of that kernel's 40 us per chunk.  Few pixels per chunk -> many rows per trip.
        auto stage = [&](auto rb_c, auto qp_c) __attribute__((always_inline)) {
            constexpr int RB = decltype(rb_c)::value, QP = decltype(qp_c)::value;
            for (int k = wave; k < Cin; k += 8 * RB) {
                for (int p0 = lane; p0 < px; p0 += 64 * QP) {
                    float v[RB][QP];
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const float *xr = xn + (size_t)(k + 8 * r) * xsc + (size_t)t0 * V * xsp;
#pragma unroll
                        for (int q = 0; q < QP; ++q) {
                            const int p = p0 + 64 * q;
                            v[r][q] = (k + 8 * r < Cin && p < px) ? xr[(size_t)p * xsp] : 0.f;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int q = 0; q < QP; ++q) {
                            const int p = p0 + 64 * q, kk = k + 8 * r;
                            if (kk < Cin && p < px) {
                                Xs[rho(kk) * PXC + p] = v[r][q];
                                if (xcopy) xcopy[((size_t)n * Cin + kk) * T * V + (size_t)t0 * V + p] = v[r][q];
                            }
                        }
                }
            }
        };
        if (px <= 64) stage(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
        else if (px <= 128) stage(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
        else stage(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
        }
        KG_STAMP(t_s1)
        KG_ACC(0, t_s0, t_s1)
        // contraction index of the Gram: idx = t * inter_c + c (inter_c is a power of two: shifts, no table)
        const int nk = inter_c * tc;
        const int kpp = ((nk / 4 + KSP - 1) / KSP);     // k-steps per K part  (inter_c % 8 == 0: nk % 4 == 0)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s < S && (only < 0 || s == only)) {
                KG_STAMP(t_b1)
                __syncthreads();                        // Xs ready; the previous subset's Gram is done with Es
                KG_STAMP(t_e00)
                KG_ACC(1, t_b1, t_e00)
#ifdef STGCN_ABLATION
                if (dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // price the wait for the prefetched fragments apart
#endif
                KG_STAMP(t_e0)
                KG_ACC(4, t_e00, t_e0)
                // ---- embeddings (this wave's row block: nrb <= 8, so exactly one; its fragments were fetched a phase ahead)
                {
                    const int rb = wave % R8;
                    for (int pb = wave / R8; pb < npb; pb += pstep) {
                        const int p = pb * 16 + l16;
                        const float *xb = Xs + lq * PXC + (p < px ? p : 0);     // (columns >= px: computed, never read)
                        f32x4 e4 = f32x4{0.f, 0.f, 0.f, 0.f};
                        // eight k-steps per trip: their eight LDS reads in flight, then the MFMAs, and NO branch in between — with
                        // `if (ks < ks_n)` around every step each step was its own basic block: read, wait out the LDS round
                        // trip, multiply (285 cycles per step, 68 % of the kernel: tools/stamps_k1g.py).  Steps beyond Cin / 4
                        // (Cin < 4 KS) read the last valid rows against zero weight fragments.  (Two pixel blocks per trip —
                        // two accumulator chains, sixteen reads in flight — measured 15 - 20 % slower.)
                        const int ks_last = ks_n - 1;
#pragma unroll
                        for (int k8 = 0; k8 < KS; k8 += 8) {
                            float b8[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) b8[q] = xb[(size_t)4 * min(k8 + q, ks_last) * PXC];
                            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every read to its MFMA, one register)
#pragma unroll
                            for (int q = 0; q < 8; ++q) e4 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[k8 + q], b8[q], e4, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) Es[(rb * 16 + 4 * lq + i) * PXC + p] = e4[i] + bias[i];
                    }
                }
                // the fragments of the NEXT embedding phase (next subset, or the first one of the next chunk): in flight during
                // this subset's Gram — fetched at the head of a phase they were 16 - 64 exposed global loads per chunk and subset
                KG_STAMP(t_e05)
                // (one subset per workgroup: the fragments loaded before the first chunk serve every chunk.  Their 64 uncoalesced
                //  loads per lane — 16 rows x 16 bytes per instruction — took a third of the kernel at 256 channels, as ISSUE time)
                if (only < 0) load_w(s + 1 < S ? s + 1 : 0);
                KG_STAMP(t_e1)
                KG_ACC(2, t_e0, t_e05)
                KG_ACC(4, t_e05, t_e1)
                __syncthreads();
                KG_STAMP(t_g0)
                KG_ACC(1, t_e1, t_g0)
                // ---- Gram: unit = (block, K part)
                const float *As = Es, *Bs = Es + (size_t)inter_c * PXC;
#pragma unroll
                for (int i = 0; i < MAXB; ++i) {
                    const int unit = wave + 8 * i;
                    if (unit < nunits) {
                        const int blk = unit / KSP, kq = unit - blk * KSP;
                        const int v = (blk / nvb) * 16 + l16, w = (blk % nvb) * 16 + l16;
                        const int k1 = min((kq + 1) * kpp, nk / 4);
                        const int va = v < V ? v : 0, wa_ = w < V ? w : 0;
                        const float vm = v < V ? 1.f : 0.f, wm = w < V ? 1.f : 0.f;   // (rows / columns beyond V: zeroed operands)
                        int ks = kq * kpp;
                        for (; ks + 8 <= k1; ks += 8) {      // eight k-steps per trip: sixteen LDS reads in flight, then the MFMAs
                            float a8[8], b8[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int idx = 4 * (ks + q) + lq;
                                const int off = (idx & icm) * PXC + (idx >> icl) * V;
                                a8[q] = As[off + va] * vm;
                                b8[q] = Bs[off + wa_] * wm;
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc[s][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a8[q], b8[q], acc[s][i], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        for (; ks + 4 <= k1; ks += 4) {
                            float a4[4], b4[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int idx = 4 * (ks + q) + lq;
                                const int off = (idx & icm) * PXC + (idx >> icl) * V;
                                a4[q] = As[off + va] * vm;
                                b4[q] = Bs[off + wa_] * wm;
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[s][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q], b4[q], acc[s][i], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        for (; ks < k1; ++ks) {
                            const int idx = 4 * ks + lq;
                            const int off = (idx & icm) * PXC + (idx >> icl) * V;
                            acc[s][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[off + va] * vm, Bs[off + wa_] * wm, acc[s][i], 0, 0, 0);
                        }
                    }
                }
                KG_STAMP(t_g1)
                KG_ACC(3, t_g0, t_g1)
            }
        }
    }
    // ---- the K parts of a block meet in LDS (fixed order), soft-max, store
    KG_STAMP(t_t0)
    __syncthreads();
    const int VV = V * V;
    float *parts = smem;                                // [S][KSP][V][V]
    float *Sm = parts + (size_t)S * KSP * VV;           // [S][V][V]
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < S && (only < 0 || s == only)) {
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int unit = wave + 8 * i;
                if (unit < nunits) {
                    const int blk = unit / KSP, kq = unit - blk * KSP;
                    const int w = (blk % nvb) * 16 + l16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int v = (blk / nvb) * 16 + 4 * lq + j;
                        if (v < V && w < V) parts[((size_t)s * KSP + kq) * VV + v * V + w] = acc[s][i][j];
                    }
                }
            }
        }
    }
    __syncthreads();
    const float denom = (float)(inter_c * T);
    const int e_lo = only < 0 ? 0 : only * VV, e_hi = only < 0 ? S * VV : e_lo + VV;
    for (int e = e_lo + tid; e < e_hi; e += 512) {
        const int s = e / VV, vw = e - s * VV;
        float a = 0.f;
        for (int kq = 0; kq < KSP; ++kq) a += parts[((size_t)s * KSP + kq) * VV + vw];
        Sm[e] = a / denom;
    }
    __syncthreads();
    if (only < 0) softmax_columns(Sm, A_eff, S, V, 0, tid, 512);
    else softmax_columns(Sm + e_lo, A_eff, 1, V, only, tid, 512);
    __syncthreads();
    float *Pn = P + (size_t)n * S * VV;
    for (int e = e_lo + tid; e < e_hi; e += 512) Pn[e] = Sm[e];
#ifdef STGCN_ABLATION
    if (dbg) {
        KG_STAMP(t_end)
        KG_ACC(4, t_t0, t_end)
        KG_ACC(5, t_begin, t_end)
        if (lane == 0 && blockIdx.x < 8 && blockIdx.y == 0)
            for (int i = 0; i < 6; ++i) dbg[(blockIdx.x * 8 + wave) * 8 + i] = tsum[i];
    }
#endif
}

}  // namespace

// true when launch_attention can also emit the (N, T*V, 16) feature tensor (folded kernel, Cin = 3, S = 3)
// launch geometry of the folded kernel (shared by the capability query and the launcher)
struct FoldedPlan {
    bool ok = false;
    int maxb = 0, nw = 0, TC = 0, Rp = 0, slice_off = 0, sq_behind = 0;
    size_t lds = 0;
};

// nw = waves per workgroup (16 or 8)
static FoldedPlan plan_folded_nw(int Cin, int T, int V, int inter_c, int S, bool with_features, int nw, bool sq_behind = false) {
    FoldedPlan pl;
    const int C1 = Cin + 1, R = Cin * V + 1;
    const int nb = ceil_div(R, GB), nblk = nb * (nb + 1) / 2;
    const size_t gs_floats = (size_t)R * R + (size_t)S * V * V + (sq_behind ? (size_t)4 * V * V + 4 : 0);
    pl.sq_behind = sq_behind ? 1 : 0;
    if (Cin > 4 || S * C1 * C1 > MS_FLOATS || V > 64) return pl;
    if (with_features && (Cin != 3 || S != 3)) return pl;
    pl.maxb = ceil_div(nblk, nw);                            // Gram blocks (accumulators) per wave
    if (pl.maxb > 12) return pl;
    pl.nw = nw;
    pl.Rp = (nb | 1) * GB;                                   // >= nb*16 and = 16 (mod 32) floats: conflict-free fragment reads
    const size_t slices = with_features ? (size_t)nw * 4096 : 0;   // 4 KiB per wave for the coalesced feature rows
    // chunk of frames held in LDS: the whole clip when it fits in ~96 KiB, else as many frames as do (multiple of 4:
    // one MFMA step contracts 4 frames)
    size_t budget = (size_t)96 * 1024 / 4;
    if (gs_floats > budget) budget = gs_floats;
    if ((MS_FLOATS + budget) * 4 + slices > (size_t)kLdsBytes) budget = ((size_t)kLdsBytes - slices) / 4 - MS_FLOATS;
    if (budget < gs_floats) return pl;                       // Gram + attention matrices must fit
    int TC = (int)(budget / pl.Rp) & ~3;
    if (TC > (T + 3) / 4 * 4) TC = (T + 3) / 4 * 4;
    if (TC < 4) return pl;
    size_t u_floats = (size_t)TC * pl.Rp;
    if (u_floats < gs_floats) u_floats = gs_floats;
    const size_t wl_floats = (size_t)2 * S * inter_c * C1;   // LDS copy of the embedding weights
    if (u_floats < wl_floats) u_floats = wl_floats;
    u_floats = (u_floats + 3) / 4 * 4;                       // keep the feature slices 16-B aligned
    pl.TC = TC;
    pl.slice_off = (int)(MS_FLOATS + u_floats);              // in floats
    pl.lds = (MS_FLOATS + u_floats) * 4 + slices;
    pl.ok = pl.lds <= (size_t)kLdsBytes;
    return pl;
}

// 1024 threads where LDS allows, else 512 (wide frames: V = 46 with the feature slices)
static FoldedPlan plan_folded(int Cin, int T, int V, int inter_c, int S, bool with_features) {
    if (with_features) {   // the interleaved P behind the Gram matrix: longer frame chunks in the feature pass
        const FoldedPlan p16b = plan_folded_nw(Cin, T, V, inter_c, S, true, 16, true);
        if (p16b.ok && p16b.maxb <= 4) return p16b;
    }
    const FoldedPlan p16 = plan_folded_nw(Cin, T, V, inter_c, S, with_features, 16);
    if (p16.ok && p16.maxb <= 4) return p16;
    return plan_folded_nw(Cin, T, V, inter_c, S, with_features, 8);
}

// true when launch_attention can also emit the (N, T*V, 16) feature tensor (folded kernel, Cin = 3, S = 3)
bool attention_emits_features(int Cin, int V, int S) {
    return plan_folded(Cin, 1 << 20, V, 32, S, true).ok;
}

int launch_attention(const float *x, const float *A_eff, const float *Wa, const float *ba,
                     const float *Wb, const float *bb, float *P, float *feat, int N, int Cin, int T, int V,
                     int inter_c, int S, hipStream_t st, bool x_ntvc, float *xcopy, void *pfrag, int pf_v0, float *ybound) {
    const int xsc = x_ntvc ? 1 : T * V, xsp = x_ntvc ? Cin : 1;
    if (pfrag != nullptr && (Cin != 3 || S != 3 || (pf_v0 == 0 && V > 32) || feat != nullptr))
        return fail(STGCN_ERR_UNSUPPORTED, "attention: fragment output covers Cin=3, 3 subsets, V<=32 (got %d, %d, %d)", Cin, S, V);
    if (pfrag != nullptr && pf_v0 != 0 && (pf_v0 < 1 || pf_v0 > 32 || V - pf_v0 < 1 || V - pf_v0 > 32))
        return fail(STGCN_ERR_UNSUPPORTED, "attention: joint split %d | %d outside 1..32 per half", pf_v0, V - pf_v0);
    if (feat != nullptr && (Cin != 3 || S != 3))
        return fail(STGCN_ERR_UNSUPPORTED, "attention: the feature pass covers Cin=3, 3 subsets (got %d, %d)", Cin, S);
    const FoldedPlan pl = plan_folded(Cin, T, V, inter_c, S, feat != nullptr);
    if (feat != nullptr && !pl.ok)
        return fail(STGCN_ERR_UNSUPPORTED, "attention: V=%d too large for the feature pass", V);
    if (pfrag != nullptr && !pl.ok) return fail(STGCN_ERR_UNSUPPORTED, "attention: V=%d outside the folded kernel", V);
    if (ybound != nullptr && (!pl.ok || S * V > 512)) return fail(STGCN_ERR_UNSUPPORTED, "attention: no bound output for V=%d", V);
    if (pl.ok) {
        const int TC = pl.TC, Rp = pl.Rp, slice_off = pl.slice_off;
        const size_t lds = pl.lds;
#define LAUNCH_FOLDED(MI, TSL)                                                                          \
    do {                                                                                               \
        STGCN_HIP_CHECK(allow_lds((attention_folded_kernel<MI, TSL>), lds));                           \
        hipLaunchKernelGGL((attention_folded_kernel<MI, TSL>), dim3(N), dim3(64 * TSL), lds, st, x, A_eff, \
                           Wa, ba, Wb, bb, P, feat, (uint4 *)pfrag, Cin, T, V, inter_c, S, TC, Rp, slice_off, pl.sq_behind, xsc, xsp, xcopy, debug_buffer(), pf_v0, ybound); \
    } while (0)
        if (pl.nw == 16) {
            if (pl.maxb <= 1) LAUNCH_FOLDED(1, 16);
            else if (pl.maxb <= 2) LAUNCH_FOLDED(2, 16);
            else LAUNCH_FOLDED(4, 16);
        } else {
            if (pl.maxb <= 2) LAUNCH_FOLDED(2, 8);
            else if (pl.maxb <= 6) LAUNCH_FOLDED(6, 8);
            else LAUNCH_FOLDED(12, 8);
        }
#undef LAUNCH_FOLDED
        STGCN_LAUNCH_CHECK("attention_folded_kernel");
        return STGCN_OK;
    }
    // generic path: on the matrix cores when the shapes tile (every TCN_GCN_unit layer of the reference does)
    const bool ic_ok = inter_c == 8 || inter_c == 16 || inter_c == 32 || inter_c == 64;
    if (Cin % 4 == 0 && Cin <= 256 && ic_ok && S <= 3 && V <= 64 && !(ablate_mask() & 2048)) {   // (diagnostic builds: 2048 = the VALU kernel)
        const size_t budget = (size_t)128 * 1024 / 4;
        int TC = (int)(budget / ((size_t)V * (Cin + 2 * inter_c)));
        if (TC > T) TC = T;
        if (TC < 1) TC = 1;
        auto pitch = [&](int tc) { return ((tc * V + 63) & ~63) + 16; };
        while (TC > 1 && (size_t)pitch(TC) * (Cin + 2 * inter_c) > budget) --TC;
        const int PXC = pitch(TC);
        const int nvb = (V + 15) / 16, nblk = nvb * nvb, KSP = nblk >= 8 ? 1 : 8 / nblk;
        size_t fl = (size_t)PXC * (Cin + 2 * inter_c) + (size_t)inter_c * TC + 4;
        const size_t tail = (size_t)S * (KSP + 1) * V * V;
        if (fl < tail) fl = tail;
        const size_t lds = fl * 4;
        if (lds <= (size_t)kLdsBytes) {
            const int per_wave = ceil_div(nblk * KSP, 8), ks = Cin / 4;
#define LAUNCH_GMFMA(KSN, MB)                                                                                \
    do {                                                                                                     \
        STGCN_HIP_CHECK(allow_lds((attention_generic_mfma_kernel<KSN, MB>), lds));                           \
        hipLaunchKernelGGL((attention_generic_mfma_kernel<KSN, MB>), dim3(N, N * 2 <= 256 ? S : 1), dim3(512), lds, st, x, A_eff, Wa, ba, Wb, bb, P, \
                           Cin, T, V, inter_c, S, TC, PXC, xsc, xsp, xcopy, debug_buffer());                 \
    } while (0)
            if (per_wave <= 1) {
                if (ks <= 16) LAUNCH_GMFMA(16, 1); else if (ks <= 32) LAUNCH_GMFMA(32, 1); else LAUNCH_GMFMA(64, 1);
            } else {
                if (ks <= 16) LAUNCH_GMFMA(16, 2); else if (ks <= 32) LAUNCH_GMFMA(32, 2); else LAUNCH_GMFMA(64, 2);
            }
#undef LAUNCH_GMFMA
            STGCN_LAUNCH_CHECK("attention_generic_mfma_kernel");
            return STGCN_OK;
        }
    }
    const int maxit = ceil_div(V * V, 256);
    if (maxit > 16) return fail(STGCN_ERR_UNSUPPORTED, "attention: V=%d too large (max 64)", V);
    const size_t budget = (size_t)96 * 1024 / 4;
    int TC = (int)(budget / ((size_t)V * (Cin + 2 * inter_c)));
    if (TC > T) TC = T;
    if (TC < 1) TC = 1;
    size_t fl = (size_t)TC * V * (Cin + 2 * inter_c);
    if (fl < (size_t)V * V) fl = (size_t)V * V;
    const size_t lds = fl * 4;
    if (lds > (size_t)kLdsBytes)
        return fail(STGCN_ERR_UNSUPPORTED, "attention: Cin=%d inter_c=%d V=%d needs %zu B of LDS", Cin,
                    inter_c, V, lds);
#define LAUNCH_GENERIC(MI)                                                                       \
    do {                                                                                         \
        STGCN_HIP_CHECK(allow_lds(attention_generic_kernel<MI>, lds));                           \
        hipLaunchKernelGGL(attention_generic_kernel<MI>, dim3(S, N), dim3(256), lds, st, x, A_eff, \
                           Wa, ba, Wb, bb, P, Cin, T, V, inter_c, S, TC, xsc, xsp, xcopy);     \
    } while (0)
    if (maxit <= 2) LAUNCH_GENERIC(2);
    else if (maxit <= 4) LAUNCH_GENERIC(4);
    else if (maxit <= 9) LAUNCH_GENERIC(9);
    else LAUNCH_GENERIC(16);
#undef LAUNCH_GENERIC
    STGCN_LAUNCH_CHECK("attention_generic_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
