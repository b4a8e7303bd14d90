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
// Workgroup = 64 x 64 tile of C, 256 threads = 4 waves (2 x 2 quadrants of 32 x 32), K walked in chunks of 16 through
// LDS.  Tiles are loaded along whichever index has unit stride (coalesced when there is one); everything is bounds-checked,
// so any M, N, K works (the joint matrices are 22 or 46 wide).
#include "common.h"

namespace stgcn {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int GT = 64;    // tile edge (M and N)
constexpr int GK = 16;    // K chunk
constexpr int APAD = GK + 1;   // As[m][k] pitch: lanes run along m -> odd pitch, conflict-free fragment reads
constexpr int BPAD = GT + 1;

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float As[GT * APAD];
    __shared__ float Bs[GK * BPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.x * GT, n0 = blockIdx.y * GT, b = blockIdx.z;
    const float *A = g.A + (size_t)b * g.a_sb;
    const float *B = g.B + (size_t)b * g.b_sb;
    float *C = g.C + (size_t)b * g.c_sb;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool a_k_fast = g.a_sk == 1;     // consecutive threads along the unit-stride index of each operand
    const bool b_n_fast = g.b_sn == 1 || g.b_sk != 1;
    for (int k0 = 0; k0 < g.K; k0 += GK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < GT * GK / 256; ++i) {
            const int e = tid + i * 256;
            const int m = a_k_fast ? e / GK : e % GT, k = a_k_fast ? e % GK : e / GT;
            const int gm = m0 + m, gk = k0 + k;
            As[m * APAD + k] = (gm < g.M && gk < g.K) ? A[(size_t)gm * g.a_sm + (size_t)gk * g.a_sk] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < GT * GK / 256; ++i) {
            const int e = tid + i * 256;
            const int k = b_n_fast ? e / GT : e % GK, n = b_n_fast ? e % GT : e / GK;
            const int gk = k0 + k, gn = n0 + n;
            Bs[k * BPAD + n] = (gk < g.K && gn < g.N) ? B[(size_t)gk * g.b_sk + (size_t)gn * g.b_sn] : 0.f;
        }
        __syncthreads();
        const float *ap = As + (wm * 32 + (lane & 31)) * APAD + (lane >> 5);
        const float *bp = Bs + (lane >> 5) * BPAD + wn * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * kk], bp[2 * kk * BPAD], acc, 0, 0, 0);
    }
    // D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
    const int gn = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm < g.M && gn < g.N) {
            float v = g.alpha * acc[r];
            if (g.bias) v += g.bias[gm];
            float *c = C + (size_t)gm * g.c_sm + (size_t)gn * g.c_sn;
            *c = g.accumulate ? *c + v : v;
        }
    }
}

// out[e] = sum_{p < parts} part[p*n + e]  in the order p = 0, 1, ... (deterministic)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float *__restrict__ part, float *__restrict__ out, int parts,
                                                       size_t n) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float a = 0.f;
    for (int p = 0; p < parts; ++p) a += part[(size_t)p * n + e];
    out[e] = a;
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

// Soft-max backward over v (dim -2) of one (clip, subset) matrix: Q = P - A_eff, dS = Q * (dP - colsum(Q * dP)) * alpha.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float *__restrict__ P, const float *__restrict__ A_eff,
                                                         const float *__restrict__ dP, float *__restrict__ dS, int V,
                                                         int S, int s, float alpha) {
    extern __shared__ float dot[];   // [V]
    const int n = blockIdx.x;
    const float *Pn = P + ((size_t)n * S + s) * V * V, *Ae = A_eff + (size_t)s * V * V, *dPn = dP + (size_t)n * V * V;
    float *dSn = dS + (size_t)n * V * V;
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
    if (batch > 65535 || ceil_div(g.N, GT) > 65535) return fail(STGCN_ERR_UNSUPPORTED, "gemm: grid too large");
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(ceil_div(g.M, GT), ceil_div(g.N, GT), batch), dim3(256), 0, st, g);
    STGCN_LAUNCH_CHECK("gemm_f32_kernel");
    return STGCN_OK;
}

int launch_sum_parts(const float *part, float *out, int parts, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, out, parts, n);
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

int launch_softmax_bwd(const float *P, const float *A_eff, const float *dP, float *dS, int N, int V, int S, int s, float alpha,
                       hipStream_t st) {
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(N), dim3(256), V * sizeof(float), st, P, A_eff, dP, dS, V, S, s, alpha);
    STGCN_LAUNCH_CHECK("softmax_bwd_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
