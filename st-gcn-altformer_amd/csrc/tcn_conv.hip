// K3 — temporal conv block of Unit2D (model/net.py:47-57), eval mode with folded BatchNorm:
//
//   out[o,t,v] = relu( shift[o] + sum_{c<Cin,k<K} (scale[o]*W[o,c,k]) * in[c, t*stride + k - pad, v] )
//
// and KF — the fused stem (tcn0(gcn0(x)), ST_GCN_AltFormer.py:70-72) where `in` is produced on the
// fly from the 12 graph-conv features per pixel and never touches HBM.
//
// The contraction is an implicit GEMM  Out[Cout x pixels] = Wp[Cout x (Cin*K)] * B[(Cin*K) x pixels]
// on the matrix cores.  With the (C, T*V) layout a temporal tap is a flat shift by V pixels, so one
// LDS tile of (channels x input-pixel-span) serves all K taps: B[(c,k), q] = tile[c][off(q) + k*V].
//
// f32 path (STGCN_MATH_F32): v_mfma_f32_32x32x2_f32 — bit-exact fp32 fma chains, 64 cycles per
// instruction per SIMD, so the kernel is bound by the fp32 MFMA roof (157 TFLOP/s); one ds_read_b32
// and a quarter of a 16-byte weight load per MFMA keep LDS and L2 far from their limits.
//   workgroup  = 256 threads = 4 waves, tile = 128 output channels x NP=128 output pixels of one clip
//   wave w     = output-channel block w (32 channels) x 4 pixel blocks of 32   (4 x f32x16 accumulators)
//   LDS        = 2 x [CC=16 channels][ROW] fp32, double-buffered over the Cin/16 channel chunks
//   weights    = pre-packed in fragment order (stgcn_tcn_pack), streamed from L2 with one
//                global_load_dwordx4 per lane per 4 k-steps, prefetched one tap ahead
#include "common.h"

namespace stgcn {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Diagnostic builds (-DSTGCN_ABLATION, never shipped) can switch phases off through env STGCN_ABLATE to
// price them: 1 = producer in the main loop, 2 = MFMAs, 4 = epilogue stores.  Outputs are then wrong.
#ifdef STGCN_ABLATION
#define STGCN_ABL(bit) ((abl & (bit)) != 0)
#else
#define STGCN_ABL(bit) false
#endif

constexpr int NP = 128;  // output pixels per workgroup (4 MFMA column blocks)
constexpr int CC = 16;   // input channels per LDS chunk
constexpr int W12P = 16; // padded row of the folded graph-conv matrix: 12 weights, bias, pad

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <bool BF16OUT>
__device__ __forceinline__ void store_out(void *y, size_t idx, float v) {
    if constexpr (BF16OUT) reinterpret_cast<unsigned short *>(y)[idx] = f32_to_bf16_rne(v);
    else reinterpret_cast<float *>(y)[idx] = v;
}

// ---------------------------------------------------------------------------------------
// plain VALU kernel: any shape.  Wp = scale[o]*W[o][c][k] in the original (Cout,Cin,K) order.
// One thread per output pixel, OBV output channels per block in registers.
// ---------------------------------------------------------------------------------------
constexpr int OBV = 16;

template <bool BF16OUT>
__global__ __launch_bounds__(256) void tcn_valu_kernel(const float *__restrict__ x,
                                                       const float *__restrict__ Wp,
                                                       const float *__restrict__ shift, void *y, int Cin,
                                                       int Cout, int T, int V, int K, int stride, int Tout,
                                                       float lo /* 0: ReLU, -inf: raw pre-activation */) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int o0 = blockIdx.y * OBV;
    const int n = blockIdx.z;
    if (q >= Tout * V) return;
    const int pad = (K - 1) / 2;
    const int t = q / V, v = q - t * V;
    float acc[OBV];
#pragma unroll
    for (int j = 0; j < OBV; ++j) acc[j] = 0.f;
    const float *xn = x + (size_t)n * Cin * T * V;
    for (int c = 0; c < Cin; ++c) {
        for (int k = 0; k < K; ++k) {
            const int ti = t * stride + k - pad;
            if (ti < 0 || ti >= T) continue;
            const float xv = xn[((size_t)c * T + ti) * V + v];
#pragma unroll
            for (int j = 0; j < OBV; ++j)
                if (o0 + j < Cout) acc[j] = fmaf(Wp[((size_t)(o0 + j) * Cin + c) * K + k], xv, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < OBV; ++j)
        if (o0 + j < Cout)
            store_out<BF16OUT>(y, ((size_t)n * Cout + o0 + j) * Tout * V + q, fmaxf(acc[j] + shift[o0 + j], lo));
}

// The same along the JOINT axis (Unit2D(dim=3), model/net.py:28-36: Conv2d kernel (1,K), padding (0,pad), stride (1,stride)):
// y[n][o][t][w] = relu( sum_{c,k} Wp[o][c][k] * x[n][c][t][w*stride + k - pad] + shift[o] ),  w < Vout.  Read in place — the
// module used to transpose into a contiguous copy, run the frame-axis kernel and transpose back.
template <bool BF16OUT>
__global__ __launch_bounds__(256) void tcn_valu_joint_axis_kernel(const float *__restrict__ x, const float *__restrict__ Wp,
                                                                  const float *__restrict__ shift, void *y, int Cin, int Cout,
                                                                  int T, int V, int K, int stride, int Vout, float lo) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int o0 = blockIdx.y * OBV;
    const int n = blockIdx.z;
    if (q >= T * Vout) return;
    const int pad = (K - 1) / 2;
    const int t = q / Vout, w = q - t * Vout;
    float acc[OBV];
#pragma unroll
    for (int j = 0; j < OBV; ++j) acc[j] = 0.f;
    const float *xn = x + (size_t)n * Cin * T * V + (size_t)t * V;
    for (int c = 0; c < Cin; ++c) {
        for (int k = 0; k < K; ++k) {
            const int vi = w * stride + k - pad;
            if (vi < 0 || vi >= V) continue;
            const float xv = xn[(size_t)c * T * V + vi];
#pragma unroll
            for (int j = 0; j < OBV; ++j)
                if (o0 + j < Cout) acc[j] = fmaf(Wp[((size_t)(o0 + j) * Cin + c) * K + k], xv, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < OBV; ++j)
        if (o0 + j < Cout)
            store_out<BF16OUT>(y, ((size_t)n * Cout + o0 + j) * T * Vout + q, fmaxf(acc[j] + shift[o0 + j], lo));
}

// ---------------------------------------------------------------------------------------
// weight packing
//   VALU : Wp[o][c][k]                                        = scale[o]*W[o][c][k]
//   F32  : Wp[mb][ch][k][half][lane][u]  (float, u<4)         = scale[o]*W[o][c][k]
//          o = mb*32 + (lane&31),  c = ch*CC + 2*(half*4+u) + (lane>>5)
//          -> the A fragment of v_mfma_f32_32x32x2_f32 for k-step (ch,k,half*4+u) is element u of the
//             float4 a lane loads at [mb][ch][k][half][lane]
// ---------------------------------------------------------------------------------------
__global__ void tcn_pack_valu_kernel(const float *__restrict__ W, const float *__restrict__ scale,
                                     float *__restrict__ Wp, int Cin, int Cout, int K) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)Cout * Cin * K) return;
    const int o = (int)(e / ((size_t)Cin * K));
    Wp[e] = scale[o] * W[e];
}

__global__ void tcn_pack_f32_kernel(const float *__restrict__ W, const float *__restrict__ scale,
                                    float *__restrict__ Wp, int Cin, int Cout, int K) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)Cout * Cin * K) return;
    const int u = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    size_t r = e >> 8;  // [mb][ch][k][half]
    const int half = (int)(r & 1);
    r >>= 1;
    const int k = (int)(r % K);
    r /= K;
    const int nch = Cin / CC;
    const int ch = (int)(r % nch);
    const int mb = (int)(r / nch);
    const int o = mb * 32 + (lane & 31);
    const int c = ch * CC + 2 * (half * 4 + u) + (lane >> 5);
    Wp[e] = scale[o] * W[((size_t)o * Cin + c) * K + k];
}

// ---------------------------------------------------------------------------------------
// f32 MFMA main loop pieces
// ---------------------------------------------------------------------------------------
struct TileGeom {
    int q0, q_last;   // first / last valid output pixel (flat t*V+v) of this tile
    int t_first;      // frame of q0
    int span;         // input pixels per channel row held in LDS
    int origin;       // flat input pixel of LDS column 0 (may be negative: zero padding)
};

__device__ __forceinline__ TileGeom tile_geom(int tile, int V, int K, int stride, int Tout) {
    TileGeom g;
    g.q0 = tile * NP;
    g.q_last = min(g.q0 + NP, Tout * V) - 1;
    g.t_first = g.q0 / V;
    const int t_last = g.q_last / V;
    g.span = ((t_last - g.t_first) * stride + K) * V;
    g.origin = (g.t_first * stride - (K - 1) / 2) * V;
    return g;
}

// per-lane LDS offsets (in floats) of the B operand for the 4 pixel blocks, tap 0, channel (lane>>5)
__device__ __forceinline__ void lane_b_offsets(const TileGeom &g, int lane, int V, int stride, int ROW,
                                               int (&off)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int q = g.q0 + j * 32 + (lane & 31);
        q = min(q, g.q_last);  // ragged last tile: clamp so every read stays inside the tile
        const int t = q / V, v = q - t * V;
        off[j] = (lane >> 5) * ROW + (t - g.t_first) * stride * V + v;
    }
}

// B operands of one k-step: one float per lane for each of the 4 pixel blocks
struct B4 {
    float v0, v1, v2, v3;
};

__device__ __forceinline__ B4 load_b4(const float *__restrict__ b, int o0, int o1, int o2, int o3) {
    return B4{b[o0], b[o1], b[o2], b[o3]};
}

// One k-step, software-pipelined: issue the LDS reads of the NEXT k-step, then the 4 MFMAs of this one
// (channel pair c2 of the chunk against the 4 pixel blocks).  hipcc otherwise groups 8 ds_reads in front
// of 8 MFMAs and the matrix pipe idles for one LDS latency per group.
#define STGCN_KSTEP(A, NEXT_PTR)                                                     \
    do {                                                                             \
        const B4 bn = load_b4((NEXT_PTR), o0, o1, o2, o3);                           \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((A), bq.v0, acc[0], 0, 0, 0);  \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((A), bq.v1, acc[1], 0, 0, 0);  \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32((A), bq.v2, acc[2], 0, 0, 0);  \
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32((A), bq.v3, acc[3], 0, 0, 0);  \
        bq = bn;                                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                           \
    } while (0)

// 32 MFMAs of one tap of one chunk: 8 k-steps x 4 pixel blocks.  On entry bq holds the B operands of
// k-step 0 of this tap; on exit those of k-step 0 at `nextbuf` (the next tap, or any valid address when
// there is none).
__device__ __forceinline__ void mfma_tap(f32x16 (&acc)[4], float4 a0, float4 a1,
                                         const float *__restrict__ buf, const float *__restrict__ nextbuf,
                                         int o0, int o1, int o2, int o3, int ROW, B4 &bq) {
    STGCN_KSTEP(a0.x, buf + 2 * ROW);
    STGCN_KSTEP(a0.y, buf + 4 * ROW);
    STGCN_KSTEP(a0.z, buf + 6 * ROW);
    STGCN_KSTEP(a0.w, buf + 8 * ROW);
    STGCN_KSTEP(a1.x, buf + 10 * ROW);
    STGCN_KSTEP(a1.y, buf + 12 * ROW);
    STGCN_KSTEP(a1.z, buf + 14 * ROW);
    STGCN_KSTEP(a1.w, nextbuf);
}

template <bool BF16OUT>
__device__ __forceinline__ void epilogue_store(const f32x16 (&acc)[4], const TileGeom &g,
                                               const float *__restrict__ shift, void *y, int n, int Cout,
                                               int mb, int lane, int pixels_per_clip, float lo = 0.f,
                                               bool ntvc = false /* (N,T,V,C) output */) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float sh = shift[o];
        const size_t base = ((size_t)n * Cout + o) * pixels_per_clip;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = g.q0 + j * 32 + (lane & 31);
            const size_t idx = ntvc ? ((size_t)n * pixels_per_clip + q) * Cout + o : base + q;
            if (q <= g.q_last) store_out<BF16OUT>(y, idx, fmaxf(acc[j][r] + sh, lo));
        }
    }
}

// ---------------------------------------------------------------------------------------
// K3: temporal conv from a (N,Cin,T,V) tensor in HBM/L2
// ---------------------------------------------------------------------------------------
template <int JPR, bool BF16OUT>
__global__ __launch_bounds__(256) void tcn_mfma_f32_kernel(const float *__restrict__ x,
                                                           const float4 *__restrict__ Wp,
                                                           const float *__restrict__ shift, void *y,
                                                           int Cin, int Cout, int T, int V, int K,
                                                           int stride, int Tout, int ROW, float lo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n = blockIdx.z;
    const int mb = blockIdx.y * 4 + wave;
    const TileGeom g = tile_geom(blockIdx.x, V, K, stride, Tout);
    const int TV = T * V;
    const int nch = Cin / CC;
    float *buf0 = smem, *buf1 = smem + CC * ROW;

    int off[4];
    lane_b_offsets(g, lane, V, stride, ROW, off);

    // column j of the LDS row <-> flat input pixel origin + j; outside [0,T*V) is the conv's zero pad
    int jcol[JPR];
    bool jok[JPR], jwr[JPR];
#pragma unroll
    for (int jj = 0; jj < JPR; ++jj) {
        jcol[jj] = tid + jj * 256;
        const int gi = g.origin + jcol[jj];
        jwr[jj] = jcol[jj] < g.span;
        jok[jj] = jwr[jj] && gi >= 0 && gi < TV;
    }
    const float *xn = x + (size_t)n * Cin * TV + g.origin;

    float pre[CC][JPR];
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int jj = 0; jj < JPR; ++jj)
                pre[c][jj] = jok[jj] ? xn[(size_t)(ch * CC + c) * TV + jcol[jj]] : 0.f;
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int jj = 0; jj < JPR; ++jj)
                if (jwr[jj]) buf[c * ROW + jcol[jj]] = pre[c][jj];
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    prefetch(0);
    commit(buf0);
    __syncthreads();

    const float4 *wp = Wp + (size_t)mb * nch * K * 2 * 64 + lane;  // + (kidx*2 + half)*64
    const int nk = nch * K;
    float4 a0 = wp[0], a1 = wp[64];
    int kidx = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const float *cur = (ch & 1) ? buf1 : buf0;
        float *nxt = (ch & 1) ? buf0 : buf1;
        const bool more = ch + 1 < nch;
        B4 bq = load_b4(cur, off[0], off[1], off[2], off[3]);
        if (more) prefetch(ch + 1);
        for (int k = 0; k < K; ++k, ++kidx) {
            float4 n0 = a0, n1 = a1;
            if (kidx + 1 < nk) {
                n0 = wp[(size_t)(kidx + 1) * 128];
                n1 = wp[(size_t)(kidx + 1) * 128 + 64];
            }
            mfma_tap(acc, a0, a1, cur + k * V, cur + (k + 1 < K ? k + 1 : k) * V, off[0], off[1], off[2], off[3],
                     ROW, bq);
            a0 = n0;
            a1 = n1;
        }
        if (more) commit(nxt);
        __syncthreads();
    }
    epilogue_store<BF16OUT>(acc, g, shift, y, n, Cout, mb, lane, Tout * V, lo);
}

// ---------------------------------------------------------------------------------------
// KF: fused stem.  The Cin0(=3)-channel skeleton tile, the clip's attention matrices P and the
// folded (C x 12) graph-conv matrix are staged in LDS once; each thread keeps the 12 graph-conv
// features of its JPR tile columns in registers and produces channel chunk ch+1 of relu(W12.feat+b)
// into the second LDS buffer while the matrix cores consume chunk ch.
// ---------------------------------------------------------------------------------------
template <int JPR, bool BF16OUT>
__global__ __launch_bounds__(256) void stem_mfma_f32_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ W12,
    const float4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int C, int T, int V, int K,
    int ROW, int abl) {
    constexpr int CIN0 = 3, S = 3, F = 12;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n = blockIdx.z;
    const int mb = blockIdx.y * 4 + wave;
    const TileGeom g = tile_geom(blockIdx.x, V, K, 1, T);
    const int TV = T * V;
    const int nch = C / CC;
    float *W12s = smem;                      // [C][W12P]
    float *buf0 = W12s + (size_t)C * W12P;   // [CC][ROW]
    float *buf1 = buf0 + CC * ROW;           // [CC][ROW]; until chunk 1 is produced it holds Ps and Xs
    float *Ps = buf1;                        // [S][V][V]
    float *Xs = Ps + S * V * V;              // [CIN0][span]

    for (int e = tid; e < C * W12P; e += 256) W12s[e] = W12[e];
    const float *Pn = P + (size_t)n * S * V * V;
    for (int e = tid; e < S * V * V; e += 256) Ps[e] = Pn[e];
    const float *xn = x + (size_t)n * CIN0 * TV;
    for (int e = tid; e < CIN0 * g.span; e += 256) {
        const int k = e / g.span, j = e - k * g.span;
        const int gi = g.origin + j;
        Xs[e] = (gi >= 0 && gi < TV) ? xn[(size_t)k * TV + gi] : 0.f;
    }
    __syncthreads();

    // graph-conv features of this thread's tile columns (origin is frame aligned)
    int jcol[JPR];
    bool jok[JPR], jwr[JPR];
    float feat[JPR][F];
#pragma unroll
    for (int jj = 0; jj < JPR; ++jj) {
        const int j = tid + jj * 256;
        jcol[jj] = j;
        const int gi = g.origin + j;
        jwr[jj] = j < g.span;
        jok[jj] = jwr[jj] && gi >= 0 && gi < TV;
#pragma unroll
        for (int f = 0; f < F; ++f) feat[jj][f] = 0.f;
        if (jok[jj]) {
            const int fr = j / V, w = j - fr * V;
            for (int v = 0; v < V; ++v) {
                float xv[CIN0];
#pragma unroll
                for (int k = 0; k < CIN0; ++k) xv[k] = Xs[k * g.span + fr * V + v];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float pv = Ps[(s * V + v) * V + w];
#pragma unroll
                    for (int k = 0; k < CIN0; ++k)
                        feat[jj][s * CIN0 + k] = fmaf(xv[k], pv, feat[jj][s * CIN0 + k]);
                }
            }
#pragma unroll
            for (int k = 0; k < CIN0; ++k) feat[jj][S * CIN0 + k] = Xs[k * g.span + j];
        }
    }

    // one activation row (channel o) of the tile -> LDS
    auto produce_row = [&](float *buf, int c, int o) {
        const float4 *wr = reinterpret_cast<const float4 *>(W12s + o * W12P);
        const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
        const float wv[F] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
        for (int jj = 0; jj < JPR; ++jj) {
            float a = w3.x;
#pragma unroll
            for (int f = 0; f < F; ++f) a = fmaf(wv[f], feat[jj][f], a);
            a = jok[jj] ? fmaxf(a, 0.f) : 0.f;  // outside the clip the temporal conv sees zero padding
            if (jwr[jj]) buf[c * ROW + jcol[jj]] = a;
        }
    };

    int off[4];
    lane_b_offsets(g, lane, V, 1, ROW, off);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    for (int c = 0; c < CC; ++c) produce_row(buf0, c, c);
    __syncthreads();  // chunk 0 visible; every wave is done with Ps/Xs, buf1 may now be overwritten

    const float4 *wp = Wp + (size_t)mb * nch * K * 2 * 64 + lane;
    const int nk = nch * K;
    const int rpt = (CC + K - 1) / K;  // rows of the next chunk produced per tap iteration
    float4 a0 = wp[0], a1 = wp[64];
    int kidx = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const float *cur = (ch & 1) ? buf1 : buf0;
        float *nxt = (ch & 1) ? buf0 : buf1;
        const bool more = ch + 1 < nch;
        B4 bq = load_b4(cur, off[0], off[1], off[2], off[3]);
        for (int k = 0; k < K; ++k, ++kidx) {
            float4 n0 = a0, n1 = a1;
            if (kidx + 1 < nk) {
                n0 = wp[(size_t)(kidx + 1) * 128];
                n1 = wp[(size_t)(kidx + 1) * 128 + 64];
            }
            if (!STGCN_ABL(2))
                mfma_tap(acc, a0, a1, cur + k * V, cur + (k + 1 < K ? k + 1 : k) * V, off[0], off[1], off[2], off[3],
                         ROW, bq);
            if (more && !STGCN_ABL(1)) {
                const int c_end = min(CC, (k + 1) * rpt);
                for (int c = k * rpt; c < c_end; ++c) produce_row(nxt, c, (ch + 1) * CC + c);
            }
            a0 = n0;
            a1 = n1;
        }
        __syncthreads();
    }
    if (!STGCN_ABL(4)) epilogue_store<BF16OUT>(acc, g, shift, y, n, C, mb, lane, TV, 0.f, (abl & OPT_OUT_NTVC) != 0);
}

// fold the graph-conv linear stages into W12[C][W12P] (see agcn_expand.hip for the algebra)
__global__ void stem_fold_kernel(const float *__restrict__ Wd, const float *__restrict__ bd,
                                 const float *__restrict__ Wdown, const float *__restrict__ bdown,
                                 const float *__restrict__ bn_scale, const float *__restrict__ bn_shift,
                                 const float *__restrict__ down_scale, const float *__restrict__ down_shift,
                                 float *__restrict__ W12, int Cin, int C, int S) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= C * W12P) return;
    const int F = (S + 1) * Cin;
    const int o = e / W12P, f = e - o * W12P;
    float val = 0.f;
    if (f < S * Cin) {
        const int s = f / Cin, k = f - s * Cin;
        val = bn_scale[o] * Wd[((size_t)s * C + o) * Cin + k];
    } else if (f < F) {
        val = down_scale[o] * Wdown[o * Cin + (f - S * Cin)];
    } else if (f == F) {
        float b = 0.f;
        for (int s = 0; s < S; ++s) b += bd[s * C + o];
        val = fmaf(bn_scale[o], b, bn_shift[o]) + fmaf(down_scale[o], bdown[o], down_shift[o]);
    }
    W12[e] = val;
}

// LDS row stride (floats) that holds the widest tile of a launch
inline int row_stride(int V, int K, int stride, int Tout) {
    int dt = ceil_div(NP - 1, V);
    if (dt > Tout - 1) dt = Tout - 1;
    const int span = (dt * stride + K) * V;
    return span + 1;
}

inline bool mfma_f32_shape_ok(int Cin, int Cout, int V, int K, int stride, int Tout) {
    if (Cin % CC != 0 || Cout % 128 != 0) return false;
    const int ROW = row_stride(V, K, stride, Tout);
    if (ROW - 1 > 3 * 256) return false;
    return (size_t)2 * CC * ROW * 4 <= (size_t)kLdsBytes;
}

}  // namespace

// =========================================================================================
size_t tcn_packed_single_bytes(int Cin, int Cout, int K, unsigned flags) {
    // f32 and VALU layouts are one float per weight, the bf16 layout two bf16 images; the latter pads 64 output channels to 128
    const int CoutP = bf16_packs(Cin, Cout, flags & STGCN_MATH_MASK) ? (Cout + 127) / 128 * 128 : Cout;
    return align_up((size_t)Cin * CoutP * K * sizeof(float), 256);
}

// bf16 modes, K = 9: a second copy in pair order (K3v6, tcn_bf16_v6.hip) follows the first
size_t tcn_packed_bytes(int Cin, int Cout, int K, unsigned flags) {
    const size_t one = tcn_packed_single_bytes(Cin, Cout, K, flags);
    return tcn_v6_packs(Cin, Cout, K, flags & STGCN_MATH_MASK) ? 2 * one : one;
}

// true when launch_tcn_pack lays the weights out in MFMA fragment order for this shape
static bool packs_as_mfma(int Cin, int Cout, unsigned math) {
    return math == STGCN_MATH_F32 && Cin % CC == 0 && Cout % 128 == 0;
}

int launch_tcn_pack(const float *W, const float *scale, void *Wp, int Cin, int Cout, int K, unsigned flags,
                    hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const size_t total = (size_t)Cin * Cout * K;
    const int blocks = (int)((total + 255) / 256);
    if (bf16_packs(Cin, Cout, math)) {
        const int rc = launch_tcn_pack_bf16(W, scale, Wp, Cin, Cout, K, st);
        if (rc != STGCN_OK || !tcn_v6_packs(Cin, Cout, K, math)) return rc;
        return launch_tcn_pack_pairs_padded(W, scale, (char *)Wp + tcn_packed_single_bytes(Cin, Cout, K, flags), Cin, Cout, st);
    }
    if (packs_as_mfma(Cin, Cout, math)) {
        hipLaunchKernelGGL(tcn_pack_f32_kernel, dim3(blocks), dim3(256), 0, st, W, scale, (float *)Wp, Cin,
                           Cout, K);
    } else if (math <= STGCN_MATH_F32_VALU) {  // shapes the matrix-core kernels do not cover: VALU layout
        hipLaunchKernelGGL(tcn_pack_valu_kernel, dim3(blocks), dim3(256), 0, st, W, scale, (float *)Wp, Cin,
                           Cout, K);
    } else {
        return fail(STGCN_ERR_UNSUPPORTED, "tcn_pack: math mode %u not built", math);
    }
    STGCN_LAUNCH_CHECK("tcn_pack");
    return STGCN_OK;
}

int launch_tcn(const float *x, const void *Wp, const float *shift, void *y, int N, int Cin, int Cout,
               int T, int V, int K, int stride, unsigned flags, hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    const float lo = (flags & STGCN_RAW) ? -__builtin_huge_valf() : 0.f;  // raw = pre-activation (training-mode BN)
    const int pad = (K - 1) / 2;
    if (flags & STGCN_CONV_ALONG_V) {      // Unit2D(dim=3): the joint axis
        if (math != STGCN_MATH_F32_VALU) return fail(STGCN_ERR_UNSUPPORTED, "tcn: STGCN_CONV_ALONG_V goes with STGCN_MATH_F32_VALU");
        const int Vout = (V + 2 * pad - K) / stride + 1;
        if (Vout < 1) return fail(STGCN_ERR_ARG, "tcn: V=%d K=%d stride=%d gives no output joint", V, K, stride);
        if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "tcn: N=%d > 65535 clips per call", N);
        const dim3 gridv(ceil_div(T * Vout, 256), ceil_div(Cout, OBV), N);
        if (bf16out)
            hipLaunchKernelGGL((tcn_valu_joint_axis_kernel<true>), gridv, dim3(256), 0, st, x, (const float *)Wp, shift, y, Cin, Cout,
                               T, V, K, stride, Vout, lo);
        else
            hipLaunchKernelGGL((tcn_valu_joint_axis_kernel<false>), gridv, dim3(256), 0, st, x, (const float *)Wp, shift, y, Cin, Cout,
                               T, V, K, stride, Vout, lo);
        STGCN_LAUNCH_CHECK("tcn_valu_joint_axis_kernel");
        return STGCN_OK;
    }
    const int Tout = (T + 2 * pad - K) / stride + 1;
    if (Tout < 1) return fail(STGCN_ERR_ARG, "tcn: T=%d K=%d stride=%d gives no output frame", T, K, stride);
    if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "tcn: N=%d > 65535 clips per call", N);
    if (math > STGCN_MATH_F32_VALU) return fail(STGCN_ERR_ARG, "tcn: unknown math mode %u", math);
    if (bf16_packs(Cin, Cout, math))
        return launch_tcn_bf16(x, nullptr, nullptr, Wp, shift, y, N, Cin, Cout, T, V, K, stride, flags, false, st);

    if (packs_as_mfma(Cin, Cout, math)) {
        if (!mfma_f32_shape_ok(Cin, Cout, V, K, stride, Tout))
            return fail(STGCN_ERR_UNSUPPORTED,
                        "tcn: f32 MFMA kernel needs a tile row <= 768 floats (V=%d K=%d stride=%d); "
                        "use STGCN_MATH_F32_VALU",
                        V, K, stride);
        const int ROW = row_stride(V, K, stride, Tout);
        const int jpr = ceil_div(ROW - 1, 256);
        const size_t lds = (size_t)2 * CC * ROW * 4;
        const dim3 grid(ceil_div(Tout * V, NP), Cout / 128, N);
#define LAUNCH_TCN(J, B)                                                                            \
    do {                                                                                            \
        STGCN_HIP_CHECK(allow_lds(tcn_mfma_f32_kernel<J, B>, lds));                                 \
        hipLaunchKernelGGL((tcn_mfma_f32_kernel<J, B>), grid, dim3(256), lds, st, x, (const float4 *)Wp, \
                           shift, y, Cin, Cout, T, V, K, stride, Tout, ROW, lo);                    \
    } while (0)
        if (jpr == 1) { if (bf16out) LAUNCH_TCN(1, true); else LAUNCH_TCN(1, false); }
        else if (jpr == 2) { if (bf16out) LAUNCH_TCN(2, true); else LAUNCH_TCN(2, false); }
        else { if (bf16out) LAUNCH_TCN(3, true); else LAUNCH_TCN(3, false); }
#undef LAUNCH_TCN
        STGCN_LAUNCH_CHECK("tcn_mfma_f32_kernel");
        return STGCN_OK;
    }
    const dim3 grid(ceil_div(Tout * V, 256), ceil_div(Cout, OBV), N);
    if (bf16out)
        hipLaunchKernelGGL((tcn_valu_kernel<true>), grid, dim3(256), 0, st, x, (const float *)Wp, shift, y, Cin,
                           Cout, T, V, K, stride, Tout, lo);
    else
        hipLaunchKernelGGL((tcn_valu_kernel<false>), grid, dim3(256), 0, st, x, (const float *)Wp, shift, y,
                           Cin, Cout, T, V, K, stride, Tout, lo);
    STGCN_LAUNCH_CHECK("tcn_valu_kernel");
    return STGCN_OK;
}

bool tcn_mfma_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16)
        return bf16_supported(Cin, Cout, T, V, K, stride, flags, false);
    const int Tout = (T + 2 * ((K - 1) / 2) - K) / stride + 1;
    return Tout >= 1 && packs_as_mfma(Cin, Cout, math) && mfma_f32_shape_ok(Cin, Cout, V, K, stride, Tout);
}

// ---- fused stem -------------------------------------------------------------------------
// prep blob: [ W12 : C*W12P floats, 256-B aligned ][ packed temporal weights ]
static size_t stem_w12_bytes(int C) { return align_up((size_t)C * W12P * sizeof(float), 256); }

// bf16 modes, K = 9: a second copy of the temporal weights in KF6's pair order follows the first
static bool stem_prep_has_pairs(int C, int K, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    return (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) && K == 9 && C % 128 == 0;
}

// STGCN_STEM_F16MX: a third packing (KF7, stem_f16mx.hip) behind the pair-order copy
static bool stem_prep_has_f16mx(int C, int K, unsigned flags) {
    return (flags & STGCN_STEM_F16MX) && (flags & STGCN_MATH_MASK) == STGCN_MATH_BF16X3 && stem_prep_has_pairs(C, K, flags);
}
static size_t stem_f16mx_offset(int C, int K, unsigned flags) { return stem_w12_bytes(C) + 2 * tcn_packed_single_bytes(C, C, K, flags); }

size_t stem_prep_bytes(int Cin, int C, int K, int S, unsigned flags) {
    (void)Cin; (void)S;
    return stem_w12_bytes(C) + tcn_packed_single_bytes(C, C, K, flags) * (stem_prep_has_pairs(C, K, flags) ? 2 : 1) +
           (stem_prep_has_f16mx(C, K, flags) ? align_up(stem_f16mx_prep_bytes(C, K), 256) : 0);
}

static bool stem_shape_ok(int Cin, int C, int V, int K, int S, int T) {
    if (Cin != 3 || S != 3) return false;
    if (!mfma_f32_shape_ok(C, C, V, K, 1, T)) return false;
    const int ROW = row_stride(V, K, 1, T);
    if ((size_t)S * V * V + (size_t)Cin * (ROW - 1) > (size_t)CC * ROW) return false;  // Ps+Xs alias buf1
    return ((size_t)C * W12P + (size_t)2 * CC * ROW) * 4 <= (size_t)kLdsBytes;
}

bool stem_fused_supported(int Cin, int C, int T, int V, int K, int S, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (Cin != 3 || S != 3 || T < 1) return false;
    if (stem_v4_supported(Cin, C, T, V, K, S, flags)) return true;
    if (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) return bf16_supported(C, C, T, V, K, 1, flags, true);
    return math == STGCN_MATH_F32 && stem_shape_ok(Cin, C, V, K, S, T);
}

int launch_stem_prepare(const float *Wd, const float *bd, const float *Wdown, const float *bdown,
                        const float *bn_scale, const float *bn_shift, const float *down_scale,
                        const float *down_shift, const float *Wt, const float *t_scale, void *prep, int Cin,
                        int C, int K, int S, unsigned flags, hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math != STGCN_MATH_F32 && math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16)
        return fail(STGCN_ERR_UNSUPPORTED, "stem: no fused kernel for math mode %u", math);
    if (Cin != 3 || S != 3 || C % 128 != 0)
        return fail(STGCN_ERR_UNSUPPORTED, "stem: fused kernel covers Cin=3, 3 subsets, C%%128==0 (got Cin=%d S=%d C=%d)",
                    Cin, S, C);
    hipLaunchKernelGGL(stem_fold_kernel, dim3(ceil_div(C * W12P, 256)), dim3(256), 0, st, Wd, bd, Wdown, bdown,
                       bn_scale, bn_shift, down_scale, down_shift, (float *)prep, Cin, C, S);
    STGCN_LAUNCH_CHECK("stem_fold_kernel");
    int rc = launch_tcn_pack(Wt, t_scale, (char *)prep + stem_w12_bytes(C), C, C, K, flags, st);
    if (rc != STGCN_OK || !stem_prep_has_pairs(C, K, flags)) return rc;
    rc = launch_tcn_pack_bf16_pairs(Wt, t_scale, (char *)prep + stem_w12_bytes(C) + tcn_packed_single_bytes(C, C, K, flags), C, C, st);
    if (rc != STGCN_OK || !stem_prep_has_f16mx(C, K, flags)) return rc;
    return launch_stem_f16mx_prepare((const float *)prep, Wt, t_scale, (char *)prep + stem_f16mx_offset(C, K, flags), C, st);
}

// workspace of the fused stem: [ P : N*S*V*V floats, 256-B aligned ] then ONE of
//   [ features : N*T*V x 64 B (16 features as bf16 hi + lo) ]   when the large-tile kernel serves the shape and reads
//                                                                features (wide frames), or
//   [ fragments: N x 12 KiB (attention matrices as bf16 hi/lo MFMA B fragments; 48 KiB for wide frames) ]  when it computes
//                                                                them itself, or
//   [ x copy   : N*Cin*T*V floats, channel-major ]              with STGCN_IN_NTVC on the kernels that read x themselves
static size_t stem_ws_p_bytes(int N, int V, int S) { return align_up((size_t)N * S * V * V * sizeof(float), 256); }

static size_t stem_ws_feat_bytes(int N, int Cin, int C, int T, int V, int K, int S, unsigned flags) {
    if (!stem_v4_supported(Cin, C, T, V, K, S, flags)) return 0;
    return stem_v4_features_in_kernel(C, T, V, K, flags) ? (size_t)N * (V > 32 ? 48 : 12) * 1024   // (wide frames: both joint halves)
                                                         : (size_t)N * T * V * 16 * sizeof(float);
}

size_t stem_ws_bytes(int N, int Cin, int C, int T, int V, int K, int S, unsigned flags) {
    size_t b = stem_ws_p_bytes(N, V, S);
    if (stem_v4_supported(Cin, C, T, V, K, S, flags)) {
        b += stem_ws_feat_bytes(N, Cin, C, T, V, K, S, flags);
        if (stem_f16mx_supported(C, T, V, K, flags)) b += align_up((size_t)N * 4 * sizeof(float), 256);   // per-clip bounds (KF7)
    } else if (flags & STGCN_IN_NTVC) b += (size_t)N * Cin * T * V * sizeof(float);
    return b;
}

// (N,4) floats behind the fragments when KF7 serves the shape, else NULL
float *stem_ws_bounds(void *ws, int N, int Cin, int C, int T, int V, int K, int S, unsigned flags) {
    if (!stem_v4_supported(Cin, C, T, V, K, S, flags) || !stem_f16mx_supported(C, T, V, K, flags)) return nullptr;
    return reinterpret_cast<float *>(static_cast<char *>(ws) + stem_ws_p_bytes(N, V, S) + stem_ws_feat_bytes(N, Cin, C, T, V, K, S, flags));
}

float *stem_ws_features(void *ws, int N, int Cin, int C, int T, int V, int K, int S, unsigned flags) {
    if (!stem_v4_supported(Cin, C, T, V, K, S, flags)) return nullptr;
    return reinterpret_cast<float *>(static_cast<char *>(ws) + stem_ws_p_bytes(N, V, S));
}

float *stem_ws_xcopy(void *ws, int N, int Cin, int C, int T, int V, int K, int S, unsigned flags) {
    if (!(flags & STGCN_IN_NTVC) || stem_v4_supported(Cin, C, T, V, K, S, flags)) return nullptr;
    return reinterpret_cast<float *>(static_cast<char *>(ws) + stem_ws_p_bytes(N, V, S));
}

int launch_stem(const float *x, const float *P, const float *feat, const void *prep, const float *t_shift,
                void *out, int N, int Cin, int C, int T, int V, int S, int K, unsigned flags, hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "stem: N=%d > 65535 clips per call", N);
    if (feat != nullptr && stem_v4_supported(Cin, C, T, V, K, S, flags)) {
        if (stem_f16mx_supported(C, T, V, K, flags) && stem_v4_features_in_kernel(C, T, V, K, flags) && !(ablate_mask() & 512))
            return launch_stem_f16mx(x, (flags & STGCN_IN_NTVC) != 0, feat,
                                     (const char *)feat + stem_ws_feat_bytes(N, Cin, C, T, V, K, S, flags), prep,
                                     (const char *)prep + stem_f16mx_offset(C, K, flags), t_shift, out, N, C, T, V, K, flags, st);
        return launch_stem_v4(x, (flags & STGCN_IN_NTVC) != 0, feat, prep, (const char *)prep + stem_w12_bytes(C), t_shift,
                              out, N, C, T, V, K, flags, st);
    }
    if (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) {
        if (Cin != 3 || S != 3)
            return fail(STGCN_ERR_UNSUPPORTED, "stem: fused kernel covers Cin=3, 3 subsets (got %d, %d)", Cin, S);
        return launch_tcn_bf16(x, P, (const float *)prep, (const char *)prep + stem_w12_bytes(C), t_shift, out, N, C,
                               C, T, V, K, 1, flags, true, st);
    }
    if (math != STGCN_MATH_F32)
        return fail(STGCN_ERR_UNSUPPORTED, "stem: no fused kernel for math mode %u", math);
    if (!stem_shape_ok(Cin, C, V, K, S, T))
        return fail(STGCN_ERR_UNSUPPORTED,
                    "stem: fused kernel does not cover Cin=%d S=%d C=%d V=%d K=%d T=%d; call the two-stage path",
                    Cin, S, C, V, K, T);
    const int ROW = row_stride(V, K, 1, T);
    const int jpr = ceil_div(ROW - 1, 256);
    const size_t lds = ((size_t)C * W12P + (size_t)2 * CC * ROW) * 4;
    const float *W12 = (const float *)prep;
    const float4 *Wp = (const float4 *)((const char *)prep + stem_w12_bytes(C));
    const dim3 grid(ceil_div(T * V, NP), C / 128, N);
#define LAUNCH_STEM(J, B)                                                                               \
    do {                                                                                                \
        STGCN_HIP_CHECK(allow_lds(stem_mfma_f32_kernel<J, B>, lds));                                    \
        hipLaunchKernelGGL((stem_mfma_f32_kernel<J, B>), grid, dim3(256), lds, st, x, P, W12, Wp, t_shift, \
                           out, C, T, V, K, ROW, ablate_mask() | ((flags & STGCN_OUT_NTVC) ? OPT_OUT_NTVC : 0)); \
    } while (0)
    if (jpr == 1) { if (bf16out) LAUNCH_STEM(1, true); else LAUNCH_STEM(1, false); }
    else if (jpr == 2) { if (bf16out) LAUNCH_STEM(2, true); else LAUNCH_STEM(2, false); }
    else { if (bf16out) LAUNCH_STEM(3, true); else LAUNCH_STEM(3, false); }
#undef LAUNCH_STEM
    STGCN_LAUNCH_CHECK("stem_mfma_f32_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
