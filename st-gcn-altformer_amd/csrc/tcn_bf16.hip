// K3/KF on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate), two arithmetic modes:
//
//   STGCN_MATH_BF16X3 : every fp32 operand is split x = hi + lo (both bf16, lo = bf16(x - hi)) and the
//                       product is hi*hi + hi*lo + lo*hi — three MFMAs per k-step, error ~2^-17 per
//                       product (the dropped lo*lo term and the residual of the split), which keeps
//                       the stem inside the 1e-4 fp32 parity gate at 3/16 of the fp32-MFMA issue cost.
//   STGCN_MATH_BF16   : operands rounded to bf16 (hi only), one MFMA per k-step.
//
// Same implicit GEMM as tcn_conv.hip (out[Cout x pixels] = Wp[Cout x Cin*K] * B[Cin*K x pixels], a
// temporal tap = a flat shift by V pixels), different operand staging because the bf16 MFMA wants 8
// consecutive k (= 8 consecutive channels of one tap) per lane:
//   LDS image   : [pixel][32 channels] bf16, 64 B per pixel, one image for hi and one for lo, per
//                 channel chunk of 32 (two k-steps of 16), double buffered.  The 16-byte channel
//                 group q of pixel p sits at p*64 + ((q ^ ((p>>2)&3)) << 4): a ds_read_b128 of 16
//                 consecutive pixels then touches all 16 slots of the 256-B bank row (conflict-free).
//   weights     : packed [mb][chunk][tap][k-step][hi|lo][lane][8 bf16] with the BN scale folded in
//                 before the split; a lane streams 16 B per operand straight from L2, one tap ahead.
//   workgroup   : 2*NPB threads; wave = (output-channel block of 32) x (half of the NPB pixels = 4
//                 MFMA column blocks); the two waves of a SIMD share their weight fragments through L1.
//   fused stem  : each thread owns one or two tile pixels, keeps their 12 graph-conv features in
//                 registers and writes relu(W12.feat+b), split into hi/lo, for 8 channels at a time as
//                 one ds_write_b128 per image, interleaved with the MFMAs of the current chunk; the
//                 (C x 13) folded matrix is read through the scalar cache (wave-uniform).
#include "common.h"

namespace stgcn {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

#ifdef STGCN_ABLATION  // diagnostic builds only: 1 = producer, 2 = MFMAs, 4 = epilogue, 8 = B reads, 16 = A loads
#define STGCN_ABL(bit) ((abl & (bit)) != 0)
#else
#define STGCN_ABL(bit) false
#endif

constexpr int CCB = 32;   // input channels per LDS chunk (two 16-deep k-steps per tap)
constexpr int PXB = 64;   // bytes per pixel row of one image
constexpr int W12P = 16;  // row of the folded graph-conv matrix: 12 weights, bias, pad

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {  // RNE; a in the low half
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo_to_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// 8 fp32 -> 8 bf16 "hi" and 8 bf16 "lo" residuals
__device__ __forceinline__ void split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        l[i] = pack_bf16x2(v[2 * i] - bf16_lo_to_f32(h[i]), v[2 * i + 1] - bf16_hi_to_f32(h[i]));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

__device__ __forceinline__ int lds_off(int p, int q) { return p * PXB + ((q ^ ((p >> 2) & 3)) << 4); }

template <bool BF16OUT>
__device__ __forceinline__ void store_out(void *y, size_t idx, float v) {
    if constexpr (BF16OUT) reinterpret_cast<unsigned short *>(y)[idx] = (unsigned short)(pack_bf16x2(v, 0.f) & 0xffffu);
    else reinterpret_cast<float *>(y)[idx] = v;
}

// weight packing: Wp (bf16) index ((((mb*nch+ch)*K+tap)*2+kb)*2+img)*64+lane)*8+j
//   o = mb*32 + (lane&31), c = ch*32 + kb*16 + 8*(lane>>5) + j, value = split(scale[o]*W[o][c][tap])[img]
__global__ void tcn_pack_bf16_kernel(const float *__restrict__ W, const float *__restrict__ scale,
                                     unsigned short *__restrict__ Wp, int Cin, int Cout, int K) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (weight, img)
    if (e >= (size_t)Cout * Cin * K * 2) return;
    const int j = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    size_t r = e >> 9;
    const int img = (int)(r & 1);
    r >>= 1;
    const int kb = (int)(r & 1);
    r >>= 1;
    const int tap = (int)(r % K);
    r /= K;
    const int nch = Cin / CCB;
    const int ch = (int)(r % nch);
    const int mb = (int)(r / nch);
    const int o = mb * 32 + (lane & 31);
    const int c = ch * CCB + kb * 16 + 8 * (lane >> 5) + j;
    const float w = scale[o] * W[((size_t)o * Cin + c) * K + tap];
    const unsigned h = pack_bf16x2(w, 0.f) & 0xffffu;
    const unsigned l = pack_bf16x2(w - bf16_lo_to_f32(h), 0.f) & 0xffffu;
    Wp[e] = (unsigned short)(img ? l : h);
}

struct TileGeomB {
    int q0, q_last, t_first, span, origin;
};

template <int NPB>
__device__ __forceinline__ TileGeomB tile_geom_b(int tile, int V, int K, int stride, int Tout) {
    TileGeomB g;
    g.q0 = tile * NPB;
    g.q_last = min(g.q0 + NPB, Tout * V) - 1;
    g.t_first = g.q0 / V;
    const int t_last = g.q_last / V;
    g.span = ((t_last - g.t_first) * stride + K) * V;
    g.origin = (g.t_first * stride - (K - 1) / 2) * V;
    return g;
}

// B operands of one k-step for the wave's 4 pixel blocks
template <int TERMS>
struct BFrag {
    uint4 hi[4];
    uint4 lo[TERMS == 3 ? 4 : 1];
};

template <int TERMS>
__device__ __forceinline__ void load_bfrag(BFrag<TERMS> &b, const char *__restrict__ img_hi, int img_bytes,
                                           const int (&addr)[4], int xr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b.hi[j] = *reinterpret_cast<const uint4 *>(img_hi + (addr[j] ^ xr));
        if constexpr (TERMS == 3) b.lo[j] = *reinterpret_cast<const uint4 *>(img_hi + img_bytes + (addr[j] ^ xr));
    }
}

template <int TERMS>
__device__ __forceinline__ void mfma_kstep_bf16(f32x16 (&acc)[4], const uint4 &a_hi, const uint4 &a_lo,
                                                const BFrag<TERMS> &b) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a_hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi[j]);
        if constexpr (TERMS == 3) {
            const bf16x8 al = __builtin_bit_cast(bf16x8, a_lo);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, b.lo[j]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
    }
}

// -----------------------------------------------------------------------------------------------
// FUSED = false : x is the (N,Cin,T,V) fp32 input of the temporal conv
// FUSED = true  : x is the (N,3,T,V) skeleton, P the attention matrices, W12 the folded graph conv
// -----------------------------------------------------------------------------------------------
template <int NPB, int JPR, int TERMS, bool BF16OUT, bool FUSED>
__global__ __launch_bounds__(2 * NPB) void tcn_mfma_bf16_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ W12,
    const uint4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int Cin, int Cout, int T, int V,
    int K, int stride, int Tout, int ROWS /* pixel rows per image */, int abl) {
    constexpr int NT = 2 * NPB;
    constexpr int CIN0 = 3, S = 3, F = 12;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int mb = blockIdx.y * 4 + (wave & 3);
    const int nh = wave >> 2;
    const int n = blockIdx.z;
    const TileGeomB g = tile_geom_b<NPB>(blockIdx.x, V, K, stride, Tout);
    const int TV = T * V;
    const int nch = Cin / CCB;
    const int img_bytes = ROWS * PXB;
    const int buf_bytes = img_bytes * (TERMS == 3 ? 2 : 1);
    char *buf0 = smem_b;
    char *buf1 = smem_b + buf_bytes;

    // tile columns owned by this thread: column j <-> flat input pixel origin + j
    int jcol[JPR];
    bool jok[JPR], jwr[JPR];
#pragma unroll
    for (int jj = 0; jj < JPR; ++jj) {
        jcol[jj] = tid + jj * NT;
        const int gi = g.origin + jcol[jj];
        jwr[jj] = jcol[jj] < g.span;
        jok[jj] = jwr[jj] && gi >= 0 && gi < TV;
    }

    // ---- producer state ------------------------------------------------------------------
    float feat[FUSED ? JPR : 1][F];
    float pre8[FUSED ? 1 : JPR][8];  // !FUSED: one 8-channel unit of this thread's pixels, in flight from HBM/L2
    const float *xn = x + (size_t)n * (FUSED ? CIN0 : Cin) * TV;
    if constexpr (FUSED) {
        float *Ps = reinterpret_cast<float *>(buf1);  // [S][V][V]   (buf1 is free until chunk 1 is produced)
        float *Xs = Ps + S * V * V;                   // [CIN0][span]
        const float *Pn = P + (size_t)n * S * V * V;
        for (int e = tid; e < S * V * V; e += NT) Ps[e] = Pn[e];
        for (int e = tid; e < CIN0 * g.span; e += NT) {
            const int k = e / g.span, j = e - k * g.span;
            const int gi = g.origin + j;
            Xs[e] = (gi >= 0 && gi < TV) ? xn[(size_t)k * TV + gi] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < JPR; ++jj) {
#pragma unroll
            for (int f = 0; f < F; ++f) feat[jj][f] = 0.f;
            if (jok[jj]) {
                const int j = jcol[jj];
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
    }

    // one 8-channel group (unit u of chunk ch) of this thread's pixels -> LDS images of `buf`
    auto produce_unit = [&](char *buf, int ch, int u) {
#pragma unroll
        for (int jj = 0; jj < JPR; ++jj) {
            float v[8];
            if constexpr (FUSED) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float *wr = W12 + (size_t)(ch * CCB + u * 8 + i) * W12P;  // wave-uniform: scalar loads
                    float a = wr[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) a = fmaf(wr[f], feat[jj][f], a);
                    v[i] = jok[jj] ? fmaxf(a, 0.f) : 0.f;  // outside the clip the conv sees zero padding
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = pre8[jj][i];
            }
            if (jwr[jj]) {
                uint4 hi, lo;
                split8(v, hi, lo);
                const int off = lds_off(jcol[jj], u);
                *reinterpret_cast<uint4 *>(buf + off) = hi;
                if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(buf + img_bytes + off) = lo;
            }
        }
    };
    auto load_unit = [&](int ch, int u) {  // !FUSED: 8 channels of this thread's pixels, coalesced along pixels
        if constexpr (!FUSED) {
#pragma unroll
            for (int jj = 0; jj < JPR; ++jj)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    pre8[jj][i] = jok[jj] ? xn[(size_t)(ch * CCB + u * 8 + i) * TV + g.origin + jcol[jj]] : 0.f;
        }
    };

    // ---- consumer state ------------------------------------------------------------------
    // pixel row (tile-local) of each of the wave's 4 column blocks at tap 0
    int prow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int q = g.q0 + nh * 128 + j * 32 + (lane & 31);
        q = min(q, g.q_last);
        const int t = q / V, v = q - t * V;
        prow[j] = (t - g.t_first) * stride * V + v;
    }
    const int h = lane >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    for (int u = 0; u < 4; ++u) {
        load_unit(0, u);
        produce_unit(buf0, 0, u);
    }
    __syncthreads();  // chunk 0 visible; (fused) every wave is done with Ps/Xs in buf1

    // A fragments: [kb*2 + img], one tap ahead
    const uint4 *wp = Wp + (size_t)mb * nch * K * 4 * 64 + lane;  // + (kidx*4 + kb*2 + img)*64
    const int nk = nch * K;
    uint4 aq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) aq[i] = (TERMS == 3 || !(i & 1)) ? wp[i * 64] : make_uint4(0, 0, 0, 0);
    const int upt = (4 + K - 1) / K;
    int kidx = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const char *cur = (ch & 1) ? buf1 : buf0;
        char *nxt = (ch & 1) ? buf0 : buf1;
        const bool more = ch + 1 < nch;
        int addr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) addr[j] = lds_off(prow[j], h);
        BFrag<TERMS> bq;
        load_bfrag<TERMS>(bq, cur, img_bytes, addr, 0);
        for (int k = 0; k < K; ++k, ++kidx) {
            uint4 an[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) an[i] = aq[i];
            if (kidx + 1 < nk && !STGCN_ABL(16)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (TERMS == 3 || !(i & 1)) an[i] = wp[((size_t)(kidx + 1) * 4 + i) * 64];
            }
            // k-step 0 of this tap: fetch k-step 1 (same pixels, channel groups q^2 -> address ^ 32)
            BFrag<TERMS> bn = bq;
            if (!STGCN_ABL(8)) load_bfrag<TERMS>(bn, cur, img_bytes, addr, 32);
            if (!STGCN_ABL(2)) mfma_kstep_bf16<TERMS>(acc, aq[0], aq[1], bq);
            __builtin_amdgcn_sched_group_barrier(0x100, TERMS == 3 ? 8 : 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TERMS == 3 ? 12 : 4, 0);
            // k-step 1: fetch k-step 0 of the next tap (V pixel rows further; same tap again at the end)
            const int kn = (k + 1 < K) ? k + 1 : k;
#pragma unroll
            for (int j = 0; j < 4; ++j) addr[j] = lds_off(prow[j] + kn * V, h);
            if (!STGCN_ABL(8)) load_bfrag<TERMS>(bq, cur, img_bytes, addr, 0);
            if (!STGCN_ABL(2)) mfma_kstep_bf16<TERMS>(acc, aq[2], aq[3], bn);
            __builtin_amdgcn_sched_group_barrier(0x100, TERMS == 3 ? 8 : 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TERMS == 3 ? 12 : 4, 0);
            if (more && !STGCN_ABL(1)) {
                if (K >= 8) {  // unit u of the next chunk: loads issued at tap 2u, split + LDS store at tap 2u+1
                    if ((k >> 1) < 4) {
                        if (k & 1) produce_unit(nxt, ch + 1, k >> 1);
                        else load_unit(ch + 1, k >> 1);
                    }
                } else {
                    const int u_end = min(4, (k + 1) * upt);
                    for (int u = k * upt; u < u_end; ++u) {
                        load_unit(ch + 1, u);
                        produce_unit(nxt, ch + 1, u);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) aq[i] = an[i];
        }
        __syncthreads();
    }

    // epilogue: D[row = channel][col = pixel], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int ppc = Tout * V;
    if (STGCN_ABL(4)) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float sh = shift[o];
        const size_t base = ((size_t)n * Cout + o) * ppc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = g.q0 + nh * 128 + j * 32 + (lane & 31);
            if (q <= g.q_last) store_out<BF16OUT>(y, base + q, fmaxf(acc[j][r] + sh, 0.f));
        }
    }
}

// pixel rows per LDS image for the widest tile of a launch
inline int rows_needed(int NPB, int V, int K, int stride, int Tout) {
    int dt = ceil_div(NPB - 1, V);
    if (dt > Tout - 1) dt = Tout - 1;
    return (dt * stride + K) * V;
}

struct Bf16Plan {
    int npb = 0, jpr = 0, rows = 0;
    size_t lds = 0;
};

inline bool plan_bf16(int Cin, int Cout, int V, int K, int stride, int Tout, int terms, bool fused, Bf16Plan &pl) {
    if (Cin % CCB != 0 || Cout % 128 != 0) return false;
    const int cands[2] = {256, 128};
    for (int ci = 0; ci < 2; ++ci) {
        const int npb = cands[ci];
        const int rows = rows_needed(npb, V, K, stride, Tout);
        const int jpr = ceil_div(rows, 2 * npb);
        if (jpr > (npb == 128 ? 3 : 2)) continue;
        const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
        const size_t lds = 2 * buf;
        if (lds > (size_t)kLdsBytes) continue;
        if (fused && ((size_t)3 * V * V + (size_t)3 * rows) * 4 > buf) continue;  // Ps + Xs alias buf1
        // prefer one big tile per CU only when it still leaves the smaller geometry no better fit
        pl.npb = npb;
        pl.jpr = jpr;
        pl.rows = rows;
        pl.lds = lds;
        return true;
    }
    return false;
}

template <int NPB, int JPR, int TERMS, bool FUSED>
int launch_variant(const float *x, const float *P, const float *W12, const uint4 *Wp, const float *shift, void *y,
                   int N, int Cin, int Cout, int T, int V, int K, int stride, int Tout, const Bf16Plan &pl,
                   bool bf16out, hipStream_t st) {
    const dim3 grid(ceil_div(Tout * V, NPB), Cout / 128, N);
    if (bf16out) {
        auto kern = tcn_mfma_bf16_kernel<NPB, JPR, TERMS, true, FUSED>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(2 * NPB), pl.lds, st, x, P, W12, Wp, shift, y, Cin, Cout, T, V, K, stride,
                           Tout, pl.rows, ablate_mask());
    } else {
        auto kern = tcn_mfma_bf16_kernel<NPB, JPR, TERMS, false, FUSED>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(2 * NPB), pl.lds, st, x, P, W12, Wp, shift, y, Cin, Cout, T, V, K, stride,
                           Tout, pl.rows, ablate_mask());
    }
    STGCN_LAUNCH_CHECK("tcn_mfma_bf16_kernel");
    return STGCN_OK;
}

template <int TERMS, bool FUSED>
int dispatch(const float *x, const float *P, const float *W12, const uint4 *Wp, const float *shift, void *y, int N,
             int Cin, int Cout, int T, int V, int K, int stride, int Tout, const Bf16Plan &pl, bool bf16out,
             hipStream_t st) {
#define GO(NPB, JPR)                                                                                         \
    return launch_variant<NPB, JPR, TERMS, FUSED>(x, P, W12, Wp, shift, y, N, Cin, Cout, T, V, K, stride, Tout, \
                                                  pl, bf16out, st)
    if (pl.npb == 256 && pl.jpr == 1) GO(256, 1);
    if (pl.npb == 256 && pl.jpr == 2) GO(256, 2);
    if (pl.npb == 128 && pl.jpr == 1) GO(128, 1);
    if (pl.npb == 128 && pl.jpr == 2) GO(128, 2);
    GO(128, 3);
#undef GO
}

}  // namespace

bool bf16_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags, bool fused) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16) return false;
    const int Tout = (T + 2 * ((K - 1) / 2) - K) / stride + 1;
    if (Tout < 1) return false;
    Bf16Plan pl;
    return plan_bf16(Cin, Cout, V, K, stride, Tout, math == STGCN_MATH_BF16X3 ? 3 : 1, fused, pl);
}

bool bf16_packs(int Cin, int Cout, unsigned math) {
    return (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) && Cin % CCB == 0 && Cout % 128 == 0;
}

int launch_tcn_pack_bf16(const float *W, const float *scale, void *Wp, int Cin, int Cout, int K, hipStream_t st) {
    const size_t total = (size_t)Cin * Cout * K * 2;
    hipLaunchKernelGGL(tcn_pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, scale,
                       (unsigned short *)Wp, Cin, Cout, K);
    STGCN_LAUNCH_CHECK("tcn_pack_bf16_kernel");
    return STGCN_OK;
}

int launch_tcn_bf16(const float *x, const float *P, const float *W12, const void *Wp, const float *shift, void *y,
                    int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags, bool fused,
                    hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    const int terms = math == STGCN_MATH_BF16X3 ? 3 : 1;
    const int Tout = (T + 2 * ((K - 1) / 2) - K) / stride + 1;
    Bf16Plan pl;
    if (Tout < 1 || !plan_bf16(Cin, Cout, V, K, stride, Tout, terms, fused, pl))
        return fail(STGCN_ERR_UNSUPPORTED,
                    "bf16 MFMA kernel does not cover Cin=%d Cout=%d V=%d K=%d stride=%d T=%d (needs Cin%%32==0, "
                    "Cout%%128==0, tile rows that fit LDS)", Cin, Cout, V, K, stride, T);
    const uint4 *wp = (const uint4 *)Wp;
    if (fused) {
        if (terms == 3) return dispatch<3, true>(x, P, W12, wp, shift, y, N, Cin, Cout, T, V, K, stride, Tout, pl, bf16out, st);
        return dispatch<1, true>(x, P, W12, wp, shift, y, N, Cin, Cout, T, V, K, stride, Tout, pl, bf16out, st);
    }
    if (terms == 3) return dispatch<3, false>(x, P, W12, wp, shift, y, N, Cin, Cout, T, V, K, stride, Tout, pl, bf16out, st);
    return dispatch<1, false>(x, P, W12, wp, shift, y, N, Cin, Cout, T, V, K, stride, Tout, pl, bf16out, st);
}

}  // namespace stgcn
