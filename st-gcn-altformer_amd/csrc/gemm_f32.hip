// Strided, batched fp32 GEMM on the fp32 matrix cores — the building block of the GENERIC unit_agcn backward
// (agcn_backward_generic.hip), where every step is a small per-clip matrix product with its own operand orientation:
//
//     C[b][m][n] (+)= alpha * sum_k A[b][m][k] * B[b][k][n]  (+ bias[m])
//
// with an element stride per index (a_sm, a_sk, a_sb, ...): a 1x1 convolution and its input gradient (weights read
// transposed through the strides), the per-frame joint mixing x.P / du.P^T (model/unit_agcn.py:87-88), the joint Gram
// matrices dP = x^T du, and the per-clip slices of the weight gradients (summed over clips afterwards, fixed order).
// Arithmetic: v_mfma_f32_32x32x2_f32, i.e. exact fp32 fma chains — the backward's 1e-4 contract needs no hi/lo splitting.
//
// Workgroup = 256 threads = 4 waves, each on 32 x 32 blocks of C; K walked in chunks through LDS.  Three arrangements of the
// four waves (template WM x WN x WK), picked per product by its shape:
//   2 x 2 x 1   64 x 64 tile, K chunk 16 — the square-ish products (1x1 convolutions, their input / weight gradients)
//   4 x 1 x 1   128 x 32 tile           — products whose N is one joint row (V = 22 / 25 / 46 <= 32: x.P, du.P^T, da / db):
//                                         a 64-wide tile left half the waves on columns that do not exist
//   1 x 1 x 4   32 x 32 tile, K chunk 64, the four waves on four quarters of every chunk, summed through LDS in the order
//               0..3 — the joint Gram matrices dP = x^T du (V x V from K = C_in*T): on the 64 x 64 tile 8/9 of both operand
//               loads were padding
// Tiles are loaded along whichever index has unit stride (coalesced when there is one); everything is bounds-checked, so any
// M, N, K works.  Optional two-level indices: the batch (clip, subset) and the contraction index (subset, joint), so that the
// per-subset products of the chain run as ONE launch (agcn_backward_generic.hip).
#include "common.h"

namespace stgcn {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int WM, int WN, int WK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    static_assert(WM * WN * WK == 4, "four waves");
    constexpr int TM = 32 * WM, TN = 32 * WN, KC = 16 * WK;
    constexpr int APAD = KC + 1;      // As[m][k] pitch: lanes run along m -> odd pitch, conflict-free fragment reads
    constexpr int BPAD = TN + 1;
    constexpr int SMEM = TM * APAD + KC * BPAD;
    static_assert(WK == 1 || SMEM >= WK * 1024, "the K-quarter sums reuse the operand tiles");
    __shared__ float smem[SMEM];
    float *As = smem, *Bs = smem + TM * APAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = (wave / WM) % WN, wk = wave / (WM * WN);
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN, b = blockIdx.z / ks, part = blockIdx.z - b * ks;
    const int bq = g.b_inner ? b / g.b_inner : b, br = g.b_inner ? b - bq * g.b_inner : 0;
    const float *A = g.A + (size_t)bq * g.a_sb + (size_t)br * g.a_sb2;
    const float *B = g.B + (size_t)bq * g.b_sb + (size_t)br * g.b_sb2;
    float *C = g.C + (size_t)bq * g.c_sb + (size_t)br * g.c_sb2 + (size_t)part * g.c_ss;
    const int kc = ks > 1 ? ((g.K + ks - 1) / ks + KC - 1) / KC * KC : g.K;      // this part's share of K
    const int k_lo = part * kc, k_hi = min(g.K, k_lo + kc);                       // (an empty part writes zeros)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // consecutive threads along the unit-stride index of each operand
    const bool a_k_fast = (g.k_inner ? g.a_sk2 : g.a_sk) == 1;
    const bool b_n_fast = g.b_sn == 1 || (g.k_inner ? g.b_sk2 : g.b_sk) != 1;
    // The tile elements of this thread: element i of A is (row m_i, local k kl_i) with either the k (k-fast: 256 % KC == 0)
    // or the row (m-fast: 256 % TM == 0) shared by all i — so the row offsets (and their two-level split) are computed ONCE,
    // and a chunk costs one multiply per operand plus adds (the offsets inside one batch entry fit 32 bits: checked by the
    // launcher).  The first version recomputed four 64-bit products and two divisions per element and chunk: ~300 vector
    // instructions per chunk against eight MFMAs.
    constexpr int EA = TM * KC / 256, EB = KC * TN / 256;
    int a_ro[EA], b_co[EB];                    // row / column offset, or -1 outside the matrix
    const int a_kl0 = a_k_fast ? tid % KC : tid / TM, a_dk = a_k_fast ? 0 : 256 / TM;   // local k of element i = a_kl0 + i*a_dk
    const int b_kl0 = b_n_fast ? tid / TN : tid % KC, b_dk = b_n_fast ? 256 / TN : 0;
#pragma unroll
    for (int i = 0; i < EA; ++i) {
        const int gm = m0 + (a_k_fast ? tid / KC + i * (256 / KC) : tid % TM);
        const int q = g.m_inner ? gm / g.m_inner : gm, r = g.m_inner ? gm - q * g.m_inner : 0;
        a_ro[i] = gm < g.M ? q * (int)g.a_sm + r * (int)g.a_sm2 : -1;
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
        const int gn = n0 + (b_n_fast ? tid % TN : tid / KC + i * (256 / KC));
        b_co[i] = gn < g.N ? gn * (int)g.b_sn : -1;
    }
    const int a_sk = (int)g.a_sk, a_sk2 = (int)g.a_sk2, b_sk = (int)g.b_sk, b_sk2 = (int)g.b_sk2;
    // fetched one K chunk AHEAD into registers: the loads of chunk k0+KC are in flight while chunk k0 is multiplied
    // (these products are small — the latency was all exposed)
    float ra[EA], rb[EB];
    auto fetch = [&](int k0) {
        if (g.k_inner == 0) {
            const int ka = k0 + a_kl0, kb = k0 + b_kl0, oa = ka * a_sk, ob = kb * b_sk, da = a_dk * a_sk, db = b_dk * b_sk;
#pragma unroll
            for (int i = 0; i < EA; ++i)
                ra[i] = (a_ro[i] >= 0 && ka + i * a_dk < k_hi) ? A[a_ro[i] + oa + i * da] : 0.f;
#pragma unroll
            for (int i = 0; i < EB; ++i)
                rb[i] = (b_co[i] >= 0 && kb + i * b_dk < k_hi) ? B[b_co[i] + ob + i * db] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < EA; ++i) {
                const int gk = k0 + a_kl0 + i * a_dk, kq = gk / g.k_inner, kr = gk - kq * g.k_inner;
                ra[i] = (a_ro[i] >= 0 && gk < k_hi) ? A[a_ro[i] + kq * a_sk + kr * a_sk2] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < EB; ++i) {
                const int gk = k0 + b_kl0 + i * b_dk, kq = gk / g.k_inner, kr = gk - kq * g.k_inner;
                rb[i] = (b_co[i] >= 0 && gk < k_hi) ? B[b_co[i] + kq * b_sk + kr * b_sk2] : 0.f;
            }
        }
    };
    // LDS slots of the same elements
    const int a_ls = a_k_fast ? (tid / KC) * APAD + tid % KC : (tid % TM) * APAD + tid / TM;
    const int a_ld = a_k_fast ? (256 / KC) * APAD : 256 / TM;
    const int b_ls = b_n_fast ? (tid / TN) * BPAD + tid % TN : (tid % KC) * BPAD + tid / KC;
    const int b_ld = b_n_fast ? (256 / TN) * BPAD : 256 / KC;
    if (k_lo < k_hi) fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EA; ++i) As[a_ls + i * a_ld] = ra[i];
#pragma unroll
        for (int i = 0; i < EB; ++i) Bs[b_ls + i * b_ld] = rb[i];
        __syncthreads();
        if (k0 + KC < k_hi) fetch(k0 + KC);
        const float *ap = As + (wm * 32 + (lane & 31)) * APAD + wk * 16 + (lane >> 5);
        const float *bp = Bs + (wk * 16 + (lane >> 5)) * BPAD + wn * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk], bp[2 * kk * BPAD], acc, 0, 0, 0);
    }
    if constexpr (WK > 1) {      // the four K quarters of the same 32 x 32 block: summed in the order 0, 1, 2, 3
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[wk * 1024 + r * 64 + lane] = acc[r];
        __syncthreads();
        if (wk != 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = smem[r * 64 + lane];
#pragma unroll
            for (int j = 1; j < WK; ++j) t += smem[j * 1024 + r * 64 + lane];
            acc[r] = t;
        }
    }
    // D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
    const int gn = n0 + wn * 32 + (lane & 31);
    if (gn >= g.N) return;
    const float nb = g.nbias ? g.nbias[gn] : 0.f;
    const int c_col = gn * (int)g.c_sn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm < g.M) {
            const int cq = g.m_inner ? gm / g.m_inner : gm, cr = g.m_inner ? gm - cq * g.m_inner : 0;
            float v = g.alpha * acc[r] + nb;
            if (g.bias) v += g.bias[gm];
            if (g.cbias) v += g.cbias[(size_t)cq * g.cb_sq + (size_t)cr * g.cb_sr + (size_t)gn * g.cb_sn];
            float *c = C + (cq * (int)g.c_sm + cr * (int)g.c_sm2 + c_col);
            *c = g.accumulate ? *c + v : v;
        }
    }
}

// out[rep*rep_stride + e] = sum_{p < parts} part[p*n + e], deterministic: a workgroup owns 32 consecutive e; its ROWS thread
// rows each sum one ROWS-th of the parts in the order p = lo, lo+1, ... (loads issued four at a time), and the sub-sums are
// added in the order 0..ROWS-1.  (One thread per e walking all parts serially took 86 us for the 1024 parts of a split
// weight-gradient product; eight rows still 35 us for 960 parts of 4 K floats — 128 workgroups, 120 dependent loads each.)
template <int ROWS>
__global__ __launch_bounds__(32 * ROWS) void sum_parts_kernel(const float *__restrict__ part, float *__restrict__ out, int parts,
                                                              size_t n, int reps, size_t rep_stride) {
    __shared__ float sub[ROWS][32];
    const int le = threadIdx.x & 31, row = threadIdx.x >> 5;
    const size_t e = (size_t)blockIdx.x * 32 + le;
    const int per = (parts + ROWS - 1) / ROWS, lo = row * per, hi = min(parts, lo + per);
    float a = 0.f;
    if (e < n) {
        int p = lo;
        for (; p + 3 < hi; p += 4) {
            const float v0 = part[(size_t)p * n + e], v1 = part[(size_t)(p + 1) * n + e], v2 = part[(size_t)(p + 2) * n + e],
                        v3 = part[(size_t)(p + 3) * n + e];
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; p < hi; ++p) a += part[(size_t)p * n + e];
    }
    sub[row][le] = a;
    __syncthreads();
    if (row == 0 && e < n) {
        float t = sub[0][le];
#pragma unroll
        for (int r = 1; r < ROWS; ++r) t += sub[r][le];
        for (int r = 0; r < reps; ++r) out[(size_t)r * rep_stride + e] = t;
    }
}

// Rows times a small square matrix, the per-frame joint mixing of unit_agcn (model/unit_agcn.py:87-88) and its transposes in
// the backward:   out[b][r][w] (+)= sum_{i < nsum} sum_v in[b][i][r][v] * M[b][i][v][w]      (V <= 64 joints)
//   u_s = x P_s;  dx += sum_s du_s P_s^T (nsum = 3, M read transposed);  da = b dS^T;  db = a dS.
// These are streaming operations — a row of V floats in, a row of V floats out, V^2 FMAs per row — and ran on the generic
// MFMA GEMM at 40 vector instructions per MFMA (index arithmetic and bounds of a 128 x 32 tile for N = K = 22): 64 us for
// 128 MB.  Here: 256 rows per workgroup staged through LDS with coalesced 16-byte accesses, one row per thread, the matrices in
// LDS read as 16-byte broadcasts (every lane the same address), exact fp32 FMAs in the order v = 0, 1, ...
struct RowMixArgs {
    const float *in, *M;
    float *out;
    int R, V, nsum, accumulate;
    long long in_sb, in_sb2, in_ss;      // batch (outer, inner) and summand strides of `in` (floats); rows are V contiguous floats
    long long m_sb, m_sb2, m_ss;         // the same for M
    int m_sv, m_sw;                      // element strides of M's (v, w): (V, 1) plain, (1, V) transposed
    long long out_sb, out_sb2;
    int b_inner;                         // batch b = bq * b_inner + br (0: single level)
};

template <int VP /* >= V: output columns per thread, 24 / 32 / 48 / 64 */>
__global__ __launch_bounds__(256) void rowmix_kernel(RowMixArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rm_lds[];
    float *Ms = rm_lds;                               // [nsum][V][VP]  (row v: the VP outputs' coefficients, zero padded)
    float *tile = Ms + a.nsum * a.V * VP;             // [256 rows][V] as in memory (+ 4 floats of slack)
    const int tid = threadIdx.x, V = a.V;
    const int b = blockIdx.y, bq = a.b_inner ? b / a.b_inner : b, br = a.b_inner ? b - bq * a.b_inner : 0;
    const int r0 = blockIdx.x * 256, rows = min(256, a.R - r0);
    const float *Mb = a.M + (size_t)bq * a.m_sb + (size_t)br * a.m_sb2;
    for (int e = tid; e < a.nsum * V * VP; e += 256) {
        const int i = e / (V * VP), rem = e - i * V * VP, v = rem / VP, w = rem - v * VP;
        Ms[e] = w < V ? Mb[(size_t)i * a.m_ss + (size_t)v * a.m_sv + (size_t)w * a.m_sw] : 0.f;
    }
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f32x2 acc2[VP / 2];                               // pairs: v_pk_fma_f32 does two FMAs per instruction
#pragma unroll
    for (int w = 0; w < VP / 2; ++w) acc2[w] = f32x2{0.f, 0.f};
    const int n = rows * V;                           // floats of the row block (contiguous in memory)
    for (int i = 0; i < a.nsum; ++i) {
        const float *src = a.in + (size_t)bq * a.in_sb + (size_t)br * a.in_sb2 + (size_t)i * a.in_ss + (size_t)r0 * V;
        __syncthreads();                              // Ms ready / previous summand's tile consumed
        // (prefetching the next summand's rows into registers — 16 x 16 bytes per thread — measured 1.7x SLOWER: occupancy)
        if ((((size_t)src) & 15) == 0) {
            for (int e = tid * 4; e < n; e += 1024) {
                if (e + 3 < n) *reinterpret_cast<float4 *>(tile + e) = *reinterpret_cast<const float4 *>(src + e);
                else for (int j = e; j < n; ++j) tile[j] = src[j];
            }
        } else {
            for (int e = tid; e < n; e += 256) tile[e] = src[e];
        }
        __syncthreads();
        if (tid < rows) {
            const float *row = tile + tid * V;
            const float *Mi = Ms + i * V * VP;
            for (int v = 0; v < V; ++v) {
                const float xv = row[v];
                const f32x2 x2 = f32x2{xv, xv};
#pragma unroll
                for (int w4 = 0; w4 < VP / 4; ++w4) {
                    const float4 m4 = *reinterpret_cast<const float4 *>(Mi + v * VP + 4 * w4);     // broadcast read
                    acc2[2 * w4] = __builtin_elementwise_fma(x2, f32x2{m4.x, m4.y}, acc2[2 * w4]);
                    acc2[2 * w4 + 1] = __builtin_elementwise_fma(x2, f32x2{m4.z, m4.w}, acc2[2 * w4 + 1]);
                }
            }
        }
    }
    __syncthreads();                                  // the last tile is consumed: reuse it for the output rows
    if (tid < rows) {
        float *row = tile + tid * V;
#pragma unroll
        for (int w = 0; w < VP; ++w)
            if (w < V) row[w] = acc2[w >> 1][w & 1];
    }
    __syncthreads();
    float *dst = a.out + (size_t)bq * a.out_sb + (size_t)br * a.out_sb2 + (size_t)r0 * V;
    if ((((size_t)dst) & 15) == 0) {
        for (int e = tid * 4; e < n; e += 1024) {
            if (e + 3 < n) {
                float4 v = *reinterpret_cast<const float4 *>(tile + e);
                if (a.accumulate) {
                    const float4 o = *reinterpret_cast<const float4 *>(dst + e);
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *reinterpret_cast<float4 *>(dst + e) = v;
            } else {
                for (int j = e; j < n; ++j) dst[j] = a.accumulate ? dst[j] + tile[j] : tile[j];
            }
        }
    } else {
        for (int e = tid; e < n; e += 256) dst[e] = a.accumulate ? dst[e] + tile[e] : tile[e];
    }
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float *__restrict__ dst, const float *__restrict__ src, size_t n) {
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e + 3 < n) {
        float4 a = *reinterpret_cast<float4 *>(dst + e);
        const float4 s = *reinterpret_cast<const float4 *>(src + e);
        a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
        *reinterpret_cast<float4 *>(dst + e) = a;
    } else {
        for (size_t i = e; i < n; ++i) dst[i] += src[i];
    }
}

// out[r] = sum_c in[r*cols + c]: one wave per row (bias gradients: rows = (clip, channel), cols = T*V)
__global__ __launch_bounds__(256) void row_sum_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *p = in + (size_t)row * cols;
    float a = 0.f;
    for (int c = lane; c < cols; c += 64) a += p[c];
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if (lane == 0) out[row] = a;
}

// Soft-max backward over v (dim -2) of every (clip, subset) matrix: Q = P - A_eff, dS = Q * (dP - colsum(Q * dP)) * alpha.
// P, dP, dS: [N][S][V][V]; one workgroup per matrix.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float *__restrict__ P, const float *__restrict__ A_eff,
                                                         const float *__restrict__ dP, float *__restrict__ dS, int V,
                                                         int S, float alpha) {
    extern __shared__ float dot[];   // [V]
    const int ns = blockIdx.x, s = ns % S;
    const float *Pn = P + (size_t)ns * V * V, *Ae = A_eff + (size_t)s * V * V, *dPn = dP + (size_t)ns * V * V;
    float *dSn = dS + (size_t)ns * V * V;
    for (int w = threadIdx.x; w < V; w += 256) {
        float d = 0.f;
        for (int v = 0; v < V; ++v) d = fmaf(Pn[v * V + w] - Ae[v * V + w], dPn[v * V + w], d);
        dot[w] = d;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < V * V; e += 256) {
        const int w = e % V;
        dSn[e] = (Pn[e] - Ae[e]) * (dPn[e] - dot[w]) * alpha;
    }
}

}  // namespace

int launch_gemm_f32(const GemmArgs &g, int batch, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || batch <= 0) return fail(STGCN_ERR_ARG, "gemm: empty problem");
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    if (ks > 1 && (g.accumulate || g.bias || g.nbias || g.cbias))
        return fail(STGCN_ERR_ARG, "gemm: a split contraction writes plain partial sums (no bias, no accumulate)");
    {   // offsets inside one batch entry are 32-bit in the kernel
        auto span = [](long long n, long long inner, long long s1, long long s2) {
            if (s1 < 0 || s2 < 0) return (long long)1 << 40;
            return inner > 0 ? ((n - 1) / inner) * s1 + (inner - 1) * s2 : (n - 1) * s1;
        };
        const long long lim = ((long long)1 << 31) - 1;
        const long long sa = span(g.M, g.m_inner, g.a_sm, g.a_sm2) + span(g.K, g.k_inner, g.a_sk, g.a_sk2);
        const long long sb = span(g.K, g.k_inner, g.b_sk, g.b_sk2) + span(g.N, 0, g.b_sn, 0);
        const long long sc = span(g.M, g.m_inner, g.c_sm, g.c_sm2) + span(g.N, 0, g.c_sn, 0);
        if (sa > lim || sb > lim || sc > lim)
            return fail(STGCN_ERR_UNSUPPORTED, "gemm: an operand of one batch entry spans more than 2^31 elements (or a negative stride)");
    }
    if (g.b_inner > 0 && batch % g.b_inner != 0) return fail(STGCN_ERR_ARG, "gemm: batch %d is not a multiple of its inner count %d", batch, g.b_inner);
    // wave arrangement by shape (see the kernel's header)
    const int form = (g.M <= 32 && g.N <= 32 && g.K >= 256) ? 2 : (g.N <= 32 ? 1 : 0);
    const int TM = form == 0 ? 64 : (form == 1 ? 128 : 32), TN = form == 0 ? 64 : 32;
    if ((long long)batch * ks > 65535 || ceil_div(g.N, TN) > 65535) return fail(STGCN_ERR_UNSUPPORTED, "gemm: grid too large");
    const dim3 grid(ceil_div(g.M, TM), ceil_div(g.N, TN), batch * ks);
    if (form == 0) hipLaunchKernelGGL((gemm_f32_kernel<2, 2, 1>), grid, dim3(256), 0, st, g);
    else if (form == 1) hipLaunchKernelGGL((gemm_f32_kernel<4, 1, 1>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<1, 1, 4>), grid, dim3(256), 0, st, g);
    STGCN_LAUNCH_CHECK("gemm_f32_kernel");
    return STGCN_OK;
}

// out[(n,t)][v][e] (model_ST.py:152-155) or out[(n,v)][t][e] (STGCN_EMBED_TS, model_TS.py:161-163)
//   = sum_c z[n][c][t][v] W[e][c] + b[e] (+ pos[v][e] / pos[t][e]):  per clip  C[row = (t,v)][col = e] = z[n]^T . W^T
// (pixels on the rows so that the lanes of a store run along e, the contiguous index of the output)
int launch_patch_embed(const float *z, const float *W, const float *b, const float *pos, float *out, int N, int C, int E,
                       int T, int V, unsigned flags, hipStream_t st) {
    const long long P = (long long)T * V;
    const bool ntvc = (flags & STGCN_IN_NTVC) != 0, ts = (flags & STGCN_EMBED_TS) != 0;
    GemmArgs g{};
    g.m_inner = V;                                   // row (t,v): q = t, r = v
    g.A = z; g.a_sb = (long long)C * P;
    if (ntvc) { g.a_sm = (long long)V * C; g.a_sm2 = C; g.a_sk = 1; }
    else { g.a_sm = V; g.a_sm2 = 1; g.a_sk = P; }
    g.B = W; g.b_sk = 1; g.b_sn = C; g.b_sb = 0;    // B[k = c][n = e] = W[e][c]
    g.C = out; g.c_sn = 1; g.c_sb = P * E;
    if (ts) { g.c_sm = E; g.c_sm2 = (long long)T * E; }         // output row (v, t)
    else { g.c_sm = (long long)V * E; g.c_sm2 = E; }            // output row (t, v)
    g.bias = nullptr;
    g.nbias = b;
    g.cbias = pos; g.cb_sn = 1;
    if (ts) g.cb_sq = E; else g.cb_sr = E;
    g.M = (int)P; g.N = E; g.K = C;
    g.alpha = 1.f; g.accumulate = 0;
    return launch_gemm_f32(g, N, st);
}

int launch_sum_parts(const float *part, float *out, int parts, size_t n, hipStream_t st, int reps, size_t rep_stride) {
    const dim3 grid((unsigned)((n + 31) / 32));
    if (parts >= 64)
        hipLaunchKernelGGL((sum_parts_kernel<32>), grid, dim3(1024), 0, st, part, out, parts, n, reps, rep_stride);
    else
        hipLaunchKernelGGL((sum_parts_kernel<8>), grid, dim3(256), 0, st, part, out, parts, n, reps, rep_stride);
    STGCN_LAUNCH_CHECK("sum_parts_kernel");
    return STGCN_OK;
}

int launch_add_inplace(float *dst, const float *src, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, dst, src, n);
    STGCN_LAUNCH_CHECK("add_inplace_kernel");
    return STGCN_OK;
}

int launch_row_sum(const float *in, float *out, int rows, int cols, hipStream_t st) {
    hipLaunchKernelGGL(row_sum_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, in, out, rows, cols);
    STGCN_LAUNCH_CHECK("row_sum_kernel");
    return STGCN_OK;
}

int launch_rowmix(const float *in, const float *M, float *out, int R, int V, int nsum, int accumulate, long long in_sb,
                  long long in_sb2, long long in_ss, long long m_sb, long long m_sb2, long long m_ss, bool m_transposed,
                  long long out_sb, long long out_sb2, int batch, int b_inner, hipStream_t st) {
    if (V < 1 || V > 64 || R < 1 || nsum < 1 || batch < 1) return fail(STGCN_ERR_ARG, "rowmix: V=%d R=%d nsum=%d", V, R, nsum);
    if (batch > 65535) return fail(STGCN_ERR_UNSUPPORTED, "rowmix: batch %d > 65535", batch);
    RowMixArgs a{in, M, out, R, V, nsum, accumulate, in_sb, in_sb2, in_ss, m_sb, m_sb2, m_ss, m_transposed ? 1 : V,
                 m_transposed ? V : 1, out_sb, out_sb2, b_inner};
    const int VP = V <= 24 ? 24 : (V <= 32 ? 32 : (V <= 48 ? 48 : 64));      // output columns a thread carries (zero padded)
    const size_t lds = ((size_t)nsum * V * VP + (size_t)256 * V + 4) * sizeof(float);
    const dim3 grid(ceil_div(R, 256), batch);
#define LAUNCH_RM(VPC)                                                                     \
    do {                                                                                   \
        STGCN_HIP_CHECK(allow_lds(rowmix_kernel<VPC>, lds));                               \
        hipLaunchKernelGGL(rowmix_kernel<VPC>, grid, dim3(256), lds, st, a);               \
    } while (0)
    if (VP == 24) LAUNCH_RM(24); else if (VP == 32) LAUNCH_RM(32); else if (VP == 48) LAUNCH_RM(48); else LAUNCH_RM(64);
#undef LAUNCH_RM
    STGCN_LAUNCH_CHECK("rowmix_kernel");
    return STGCN_OK;
}

int launch_softmax_bwd(const float *P, const float *A_eff, const float *dP, float *dS, int N, int V, int S, float alpha,
                       hipStream_t st) {
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(N * S), dim3(256), V * sizeof(float), st, P, A_eff, dP, dS, V, S, alpha);
    STGCN_LAUNCH_CHECK("softmax_bwd_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
