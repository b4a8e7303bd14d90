// K3/KF on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate), two arithmetic modes:
//
//   STGCN_MATH_BF16X3 : every fp32 operand is split x = hi + lo (both bf16, lo = bf16(x - hi)) and the
//                       product is hi*hi + hi*lo + lo*hi — three MFMAs per k-step, error ~2^-17 per
//                       product (the dropped lo*lo term and the residual of the split), which keeps
//                       the stem inside the 1e-4 fp32 parity gate at 3/16 of the fp32-MFMA issue cost.
//   STGCN_MATH_BF16   : operands rounded to bf16 (hi only), one MFMA per k-step.
//
// Same implicit GEMM as tcn_conv.hip (out[Cout x pixels] = Wp[Cout x Cin*K] * B[Cin*K x pixels], a
// temporal tap = a flat shift by V pixels); the operand staging differs because the bf16 MFMA wants 8
// consecutive k (= 8 consecutive channels of one tap) per lane:
//   workgroup   : 256 threads = 4 waves as 2 (channel halves) x 2 (pixel halves); tile = 128 output
//                 channels x 128 output pixels; a wave owns 2x2 MFMA blocks (64 channels x 64 pixels, 64
//                 accumulator registers).  ~42 KB (V=22) of LDS keeps 2 workgroups per CU, so one
//                 workgroup's prologue / epilogue / barriers hide under the other's MFMAs.
//   LDS image   : [pixel][16 channels] bf16 = 32 B per pixel, one image for hi and one for lo, per
//                 channel chunk of 16 (one k-step per tap), double buffered.  The 16-byte channel group h
//                 of pixel p sits at p*32 + ((h ^ ((p>>3)&1)) << 4): a ds_read_b128 of 16 consecutive
//                 pixels touches all 16 slots of the 256-B bank row (conflict-free).
//   weights     : packed [mb][chunk][tap][hi|lo][lane][8 bf16] with the BN scale folded in before the
//                 split; a lane streams 16 B per fragment straight from L2, two taps ahead (the two
//                 pixel-half waves of a channel half read the same fragments -> L1 hits).
//   producer    : the next chunk is staged while the current one is consumed, in the SAME basic block as the
//                 MFMAs (tap loop fully unrolled for K = 9).  Stand-alone temporal conv (tcn_mfma_bf16_kernel):
//                 fp32 activations loaded from HBM/L2, coalesced along pixels, split and stored by the VALU.
//                 Fused stem at 128-pixel tiles (stem_mfma_bf16_kernel): relu(W12 . features) on the matrix cores
//                 from an LDS feature tile; the 256-pixel persistent forms (fused stem KF4, and K3v4 for the stand-alone
//                 K = 9 / stride 1 block, which this file's kernel serves for every other shape) live in stem_bf16_v4.hip.
#include "bf16_common.h"

namespace stgcn {

namespace {

using namespace bf16k;

constexpr int NPB = 128;  // output pixels per workgroup
constexpr int NT = 256;   // threads per workgroup

// weight packing: Wp (bf16) index (((((mb*nch+ch)*K+tap)*2+img)*64+lane)*8+j
//   o = mb*32 + (lane&31), c = ch*16 + 8*(lane>>5) + j, value = split(scale[o]*W[o][c][tap])[img]
// (Cout = 64 is packed as 128 rows, the upper 64 zero: the 128-channel tiles then serve it, half of their MFMAs idle —
//  still an order of magnitude ahead of the plain-FMA kernel; the epilogues skip rows >= Cout)
__global__ void tcn_pack_bf16_kernel(const float *__restrict__ W, const float *__restrict__ scale,
                                     unsigned short *__restrict__ Wp, int Cin, int Cout, int CoutP, int K) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (weight, img)
    if (e >= (size_t)CoutP * Cin * K * 2) return;
    const int j = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    size_t r = e >> 9;
    const int img = (int)(r & 1);
    r >>= 1;
    const int tap = (int)(r % K);
    r /= K;
    const int nch = Cin / CCB;
    const int ch = (int)(r % nch);
    const int mb = (int)(r / nch);
    const int o = mb * 32 + (lane & 31);
    const int c = ch * CCB + 8 * (lane >> 5) + j;
    const float w = o < Cout ? scale[o] * W[((size_t)o * Cin + c) * K + tap] : 0.f;
    const unsigned h = pack_bf16x2(w, 0.f) & 0xffffu;
    const unsigned l = pack_bf16x2(w - bf16_lo_to_f32(h), 0.f) & 0xffffu;
    Wp[e] = (unsigned short)(img ? l : h);
}

// -----------------------------------------------------------------------------------------------
// Stand-alone temporal conv block (Unit2D) on the bf16 matrix cores: x is the (N,Cin,T,V) fp32 input.
// Each thread owns JPR tile pixels; it loads 8 channels of them (coalesced along pixels), splits them into bf16
// hi/lo and stores one 16-byte group per image.  KT = 9: tap loop fully unrolled with that staging interleaved
// (loads at taps 0 and 3, stores at taps 3 and 7); KT = 0: any K, runtime tap loop, staging between chunks.
// act_lo = 0 for ReLU, -inf for the raw pre-activation output (training-mode BatchNorm).
// -----------------------------------------------------------------------------------------------
template <int JPR, int TERMS, bool BF16OUT, int KT>
__global__ __launch_bounds__(NT) void tcn_mfma_bf16_kernel(
    const float *__restrict__ x, const uint4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int Cin,
    int Cout, int T, int V, int Krt, int stride, int Tout, int ROWS /* pixel rows per image */, int abl,
    float act_lo) {
    const int K = KT ? KT : Krt;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int mb0 = blockIdx.y * 4 + wm * 2;
    const int n = blockIdx.z;
    const TileGeomB g = tile_geom_b(blockIdx.x, V, K, stride, Tout);
    const int TV = T * V;
    const int nch = Cin / CCB;
    const int img_bytes = ROWS * PXB;
    const int buf_bytes = img_bytes * (TERMS == 3 ? 2 : 1);
    char *buf0 = smem_b;
    char *buf1 = smem_b + buf_bytes;

    // tile columns owned by this thread: column j <-> flat input pixel origin + j.  Columns past the tile's
    // span are redirected to the spare row ROWS-1 (a dump row that is never read) so stores need no branch.
    int jcol[JPR], jdst[JPR];
    bool jok[JPR];
#pragma unroll
    for (int jj = 0; jj < JPR; ++jj) {
        jcol[jj] = tid + jj * NT;
        const int gi = g.origin + jcol[jj];
        const bool wr = jcol[jj] < g.span;
        jok[jj] = wr && gi >= 0 && gi < TV;
        jdst[jj] = wr ? jcol[jj] : ROWS - 1;
    }

    // ---- staging of the activation chunk -------------------------------------------------------
    float pv[JPR][8];  // 8 channels (one unit) of this thread's pixels, in flight from HBM/L2
    const float *xn = x + (size_t)n * Cin * TV;
    auto load_unit = [&](int ch, int u) {
        const int c0 = min(ch, nch - 1) * CCB + u * 8;   // (past the last chunk: re-read it; discarded)
#pragma unroll
        for (int jj = 0; jj < JPR; ++jj)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                pv[jj][i] = jok[jj] ? xn[(size_t)(c0 + i) * TV + g.origin + jcol[jj]] : 0.f;
    };
    auto store_unit = [&](char *buf, int u) {
#pragma unroll
        for (int jj = 0; jj < JPR; ++jj) {
            uint4 hi, lo;
            split8(pv[jj], hi, lo);
            const int off = lds_off(jdst[jj], u);
            *reinterpret_cast<uint4 *>(buf + off) = hi;
            if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(buf + img_bytes + off) = lo;
        }
    };

    // ---- consumer state ------------------------------------------------------------------
    int prow[2];  // tile-local pixel row of the wave's 2 column blocks at tap 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int q = g.q0 + (wn * 2 + j) * 32 + (lane & 31);
        q = min(q, g.q_last);
        const int t = q / V, v = q - t * V;
        prow[j] = (t - g.t_first) * stride * V + v;
    }
    const int h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    load_unit(0, 0);
    store_unit(buf0, 0);
    load_unit(0, 1);
    store_unit(buf0, 1);
    __syncthreads();  // chunk 0 visible

    // weight fragments: [m].hi/.lo of flat k index kidx = ch*K + tap at wpm[m][(kidx*2 + img)*64]
    const int nk = nch * K;
    const uint4 *wpm0 = Wp + (size_t)(mb0 + 0) * nk * 2 * 64 + lane;
    const uint4 *wpm1 = Wp + (size_t)(mb0 + 1) * nk * 2 * 64 + lane;
    auto load_a = [&](Frag2<TERMS> &a, int kidx) {
        const int kc = min(kidx, nk - 1);  // past the end: re-read the last fragment (discarded)
        if (STGCN_ABL(16)) return;
        a.hi[0] = wpm0[(size_t)kc * 128];
        a.hi[1] = wpm1[(size_t)kc * 128];
        if constexpr (TERMS == 3) {
            a.lo[0] = wpm0[(size_t)kc * 128 + 64];
            a.lo[1] = wpm1[(size_t)kc * 128 + 64];
        }
    };
    auto load_b = [&](Frag2<TERMS> &b, const char *buf, int tap) {
        if (STGCN_ABL(8)) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = lds_off(prow[j] + tap * V, h);
            b.hi[j] = *reinterpret_cast<const uint4 *>(buf + off);
            if constexpr (TERMS == 3) b.lo[j] = *reinterpret_cast<const uint4 *>(buf + img_bytes + off);
        }
    };

    Frag2<TERMS> a0 = {}, a1 = {}, a2 = {};  // taps kidx, kidx+1, kidx+2
    load_a(a0, 0);
    load_a(a1, 1);
    int kidx = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const char *cur = (ch & 1) ? buf1 : buf0;
        char *nxt = (ch & 1) ? buf0 : buf1;
        Frag2<TERMS> b0 = {}, b1 = {};
        load_b(b0, cur, 0);
        if constexpr (KT == 9) {
            // ---- fully unrolled: one basic block per chunk, staging slices between the MFMAs ----
#define STGCN_TAP(TAP)                                                                                   \
    do {                                                                                                 \
        load_a(a2, kidx + 2);                                                                            \
        load_b(b1, cur, (TAP) + 1 < 9 ? (TAP) + 1 : (TAP));                                              \
        if (!STGCN_ABL(2)) mfma_kstep_bf16<TERMS>(acc, a0, b0);                                          \
        if (!STGCN_ABL(1)) {                                                                             \
            if ((TAP) == 0) load_unit(ch + 1, 0);                                                        \
            if ((TAP) == 3) {                                                                            \
                store_unit(nxt, 0);                                                                      \
                load_unit(ch + 1, 1);                                                                    \
            }                                                                                            \
            if ((TAP) == 7) store_unit(nxt, 1);                                                          \
        }                                                                                                \
        a0 = a1;                                                                                         \
        a1 = a2;                                                                                         \
        b0 = b1;                                                                                         \
        ++kidx;                                                                                          \
    } while (0)
            STGCN_TAP(0);
            STGCN_TAP(1);
            STGCN_TAP(2);
            STGCN_TAP(3);
            STGCN_TAP(4);
            STGCN_TAP(5);
            STGCN_TAP(6);
            STGCN_TAP(7);
            STGCN_TAP(8);
#undef STGCN_TAP
        } else {
            for (int k = 0; k < K; ++k, ++kidx) {
                load_a(a2, kidx + 2);
                load_b(b1, cur, k + 1 < K ? k + 1 : k);
                if (!STGCN_ABL(2)) mfma_kstep_bf16<TERMS>(acc, a0, b0);
                a0 = a1;
                a1 = a2;
                b0 = b1;
            }
            if (ch + 1 < nch) {
                load_unit(ch + 1, 0);
                store_unit(nxt, 0);
                load_unit(ch + 1, 1);
                store_unit(nxt, 1);
            }
        }
        __syncthreads();
    }

    // epilogue: D[row = channel][col = pixel], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (STGCN_ABL(4)) return;
    const int ppc = Tout * V;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = (mb0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (o >= Cout) continue;                      // padded rows of a 64-channel layer
            const float sh = shift[o];
            const size_t base = ((size_t)n * Cout + o) * ppc;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = g.q0 + (wn * 2 + j) * 32 + (lane & 31);
                if (q <= g.q_last) store_out<BF16OUT>(y, base + q, fmaxf(acc[m][j][r] + sh, act_lo));
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------
// Fused stem on the bf16 matrix cores.  Same consumer as above; the producer is itself an MFMA:
//   Fs   : LDS tile [tile pixel][16] fp32 = the 12 graph-conv features of the pixel, a validity flag
//          (1 inside the clip, else the whole row is 0 -> the conv sees its zero padding), 3 zeros.
//   y    = relu(W12[16 ch x 16] . Fs^T[16 x 16 px])  by 4 x v_mfma_f32_16x16x4_f32 (exact fp32) per
//          16-pixel block; the D fragment (lane = pixel, 4 consecutive channels per lane group) is
//          split into bf16 hi/lo and stored as 8 bytes per image.  One block per wave per tap, in the
//          same basic block as the 12 consumer MFMAs: ~25 VALU per block instead of ~60 FMAs per tap.
// PB = producer blocks per wave per chunk (rows/16/4 rounded up), KT as above.
// -----------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int PB, int TERMS, bool BF16OUT, int KT>
__global__ __launch_bounds__(NT) void stem_mfma_bf16_kernel(
    const float *__restrict__ x, const float *__restrict__ P, const float *__restrict__ W12,
    const uint4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int C, int T, int V, int Krt,
    int ROWS /* pixel rows per image, multiple of 16 */, int abl) {
    constexpr int CIN0 = 3, S = 3, F = 12;
    const int K = KT ? KT : Krt;
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int mb0 = blockIdx.y * 4 + wm * 2;
    const int n = blockIdx.z;
    const TileGeomB g = tile_geom_b(blockIdx.x, V, K, 1, T);
    const int TV = T * V;
    const int nch = C / CCB;
    const int img_bytes = ROWS * PXB;
    const int buf_bytes = img_bytes * (TERMS == 3 ? 2 : 1);
    char *buf0 = smem_b;
    char *buf1 = smem_b + buf_bytes;
    float *Fs = reinterpret_cast<float *>(smem_b + 2 * buf_bytes);  // 4 planes of [ROWS] float4
    const int nblk = (g.span + 15) >> 4;                            // 16-pixel producer blocks of this tile
#ifdef STGCN_ABLATION
    if (STGCN_ABL(32)) {  // experiment: de-phase the second workgroup slot of every CU by ~half a tile
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lin >= 256 && lin < 512)
            for (int i = 0; i < (abl >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif

    // ---- prologue: stage P and the skeleton tile, build the feature tile ---------------------
    {
        float *Ps = reinterpret_cast<float *>(smem_b);  // [S][V][V]  (buf0/buf1 are free until chunk 0 is produced)
        float *Xs = Ps + S * V * V;                     // [CIN0][span]
        const float *Pn = P + (size_t)n * S * V * V;
        const float *xn = x + (size_t)n * CIN0 * TV;
        for (int e = tid; e < S * V * V; e += NT) Ps[e] = Pn[e];
        for (int e = tid; e < CIN0 * g.span; e += NT) {
            const int k = e / g.span, j = e - k * g.span;
            const int gi = g.origin + j;
            Xs[e] = (gi >= 0 && gi < TV) ? xn[(size_t)k * TV + gi] : 0.f;
        }
        __syncthreads();
        for (int j = tid; j < ROWS; j += NT) {
            float feat[16];
#pragma unroll
            for (int f = 0; f < 16; ++f) feat[f] = 0.f;
            const int gi = g.origin + j;
            if (j < g.span && gi >= 0 && gi < TV) {
                const int fr = j / V, w = j - fr * V;
                const float *xr = Xs + fr * V;
                const float *pc = Ps + w;
#pragma unroll 2
                for (int v = 0; v < V; ++v) {
                    float xv[CIN0];
#pragma unroll
                    for (int k = 0; k < CIN0; ++k) xv[k] = xr[k * g.span + v];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float pw = pc[(s * V + v) * V];
#pragma unroll
                        for (int k = 0; k < CIN0; ++k) feat[s * CIN0 + k] = fmaf(xv[k], pw, feat[s * CIN0 + k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < CIN0; ++k) feat[S * CIN0 + k] = Xs[k * g.span + j];
                feat[F] = 1.f;  // multiplies the folded bias; 0 outside the clip
            }
            float4 *dst = reinterpret_cast<float4 *>(Fs) + j;  // plane q holds features 4q..4q+3 of every pixel
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[(size_t)q * ROWS] = make_float4(feat[4 * q], feat[4 * q + 1], feat[4 * q + 2], feat[4 * q + 3]);
        }
        __syncthreads();  // Fs complete; Ps/Xs dead -> buf0/buf1 may be overwritten
    }

    // ---- producer: one 16-pixel block of chunk `ch` -> hi/lo images of `buf` -------------------
    const int pl = lane & 15, pg = lane >> 4;  // pixel within block / channel group (4 channels) = k quarter
    auto load_w12 = [&](int ch) {              // A operand of the producer: W12[ch*16 + pl][4*pg .. 4*pg+3]
        const int o = min(ch, nch - 1) * CCB + pl;
        return *reinterpret_cast<const float4 *>(W12 + (size_t)o * W12P + 4 * pg);
    };
    auto produce_block = [&](char *buf, const float4 &wa, int bi) {
        const int p = bi * 16 + pl;
        const float4 fb = reinterpret_cast<const float4 *>(Fs)[(size_t)pg * ROWS + p];  // conflict-free: slot = p mod 16
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, fb.x, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, fb.y, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, fb.z, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, fb.w, d, 0, 0, 0);
        // lane holds channels 4*pg .. 4*pg+3 of pixel p
        const float v0 = fmaxf(d[0], 0.f), v1 = fmaxf(d[1], 0.f), v2 = fmaxf(d[2], 0.f), v3 = fmaxf(d[3], 0.f);
        const unsigned h0 = pack_bf16x2(v0, v1), h1 = pack_bf16x2(v2, v3);
        const int off = lds_off(p, pg >> 1) + (pg & 1) * 8;
        *reinterpret_cast<uint2 *>(buf + off) = make_uint2(h0, h1);
        if constexpr (TERMS == 3) {
            const unsigned l0 = pack_bf16x2(v0 - bf16_lo_to_f32(h0), v1 - bf16_hi_to_f32(h0));
            const unsigned l1 = pack_bf16x2(v2 - bf16_lo_to_f32(h1), v3 - bf16_hi_to_f32(h1));
            *reinterpret_cast<uint2 *>(buf + img_bytes + off) = make_uint2(l0, l1);
        }
    };

    // ---- consumer state ------------------------------------------------------------------
    int prow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int q = g.q0 + (wn * 2 + j) * 32 + (lane & 31);
        q = min(q, g.q_last);
        const int t = q / V, v = q - t * V;
        prow[j] = (t - g.t_first) * V + v;
    }
    const int h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    float4 wcur = load_w12(0);
    for (int b = wave; b < nblk; b += 4) produce_block(buf0, wcur, b);
    float4 wnext = load_w12(1);
    __syncthreads();  // chunk 0 visible

    const int nk = nch * K;
    const uint4 *wpm0 = Wp + (size_t)(mb0 + 0) * nk * 2 * 64 + lane;
    const uint4 *wpm1 = Wp + (size_t)(mb0 + 1) * nk * 2 * 64 + lane;
    auto load_a = [&](Frag2<TERMS> &a, int kidx) {
        const int kc = min(kidx, nk - 1);
        if (STGCN_ABL(16)) return;
        a.hi[0] = wpm0[(size_t)kc * 128];
        a.hi[1] = wpm1[(size_t)kc * 128];
        if constexpr (TERMS == 3) {
            a.lo[0] = wpm0[(size_t)kc * 128 + 64];
            a.lo[1] = wpm1[(size_t)kc * 128 + 64];
        }
    };
    auto load_b = [&](Frag2<TERMS> &b, const char *buf, int tap) {
        if (STGCN_ABL(8)) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = lds_off(prow[j] + tap * V, h);
            b.hi[j] = *reinterpret_cast<const uint4 *>(buf + off);
            if constexpr (TERMS == 3) b.lo[j] = *reinterpret_cast<const uint4 *>(buf + img_bytes + off);
        }
    };

    Frag2<TERMS> a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {};  // weight fragments of taps kidx .. kidx+4
    load_a(a0, 0);
    load_a(a1, 1);
    load_a(a2, 2);
    load_a(a3, 3);
    int kidx = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const char *cur = (ch & 1) ? buf1 : buf0;
        char *nxt = (ch & 1) ? buf0 : buf1;
        wcur = wnext;                 // W12 rows of chunk ch+1
        wnext = load_w12(ch + 2);
        Frag2<TERMS> b0 = {}, b1 = {};
        load_b(b0, cur, 0);
        if constexpr (KT == 9) {
#define STGCN_TAP(TAP)                                                                                  \
    do {                                                                                                \
        load_a(a4, kidx + 4);                                                                           \
        load_b(b1, cur, (TAP) + 1 < 9 ? (TAP) + 1 : (TAP));                                             \
        if (!STGCN_ABL(2)) mfma_kstep_bf16<TERMS>(acc, a0, b0);                                         \
        if ((TAP) < PB && !STGCN_ABL(1)) produce_block(nxt, wcur, min(wave + 4 * (TAP), nblk - 1));     \
        a0 = a1;                                                                                        \
        a1 = a2;                                                                                        \
        a2 = a3;                                                                                        \
        a3 = a4;                                                                                        \
        b0 = b1;                                                                                        \
        ++kidx;                                                                                         \
    } while (0)
            STGCN_TAP(0);
            STGCN_TAP(1);
            STGCN_TAP(2);
            STGCN_TAP(3);
            STGCN_TAP(4);
            STGCN_TAP(5);
            STGCN_TAP(6);
            STGCN_TAP(7);
            STGCN_TAP(8);
#undef STGCN_TAP
        } else {
            for (int k = 0; k < K; ++k, ++kidx) {
                load_a(a4, kidx + 4);
                load_b(b1, cur, k + 1 < K ? k + 1 : k);
                if (!STGCN_ABL(2)) mfma_kstep_bf16<TERMS>(acc, a0, b0);
                a0 = a1;
                a1 = a2;
                a2 = a3;
                a3 = a4;
                b0 = b1;
            }
            if (ch + 1 < nch)
                for (int b = wave; b < nblk; b += 4) produce_block(nxt, wcur, b);
        }
        __syncthreads();
    }

    if (STGCN_ABL(4)) return;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = (mb0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float sh = shift[o];
            const size_t base = ((size_t)n * C + o) * TV;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = g.q0 + (wn * 2 + j) * 32 + (lane & 31);
                const size_t idx = (abl & OPT_OUT_NTVC) ? ((size_t)n * TV + q) * C + o : base + q;  // (N,T,V,C) | (N,C,T,V)
                if (q <= g.q_last) store_out<BF16OUT>(y, idx, fmaxf(acc[m][j][r] + sh, 0.f));
            }
        }
    }
}

struct StemPlan {
    int pb = 0, rows = 0;
    size_t lds = 0;
};

inline bool plan_stem_bf16(int C, int V, int K, int T, int terms, StemPlan &pl) {
    if (C % CCB != 0 || C % 128 != 0) return false;
    int dt = ceil_div(NPB - 1, V);
    if (dt > T - 1) dt = T - 1;
    const int span = (dt + K) * V;
    const int rows = (span + 15) / 16 * 16;
    const int pb = ceil_div(rows / 16, 4);
    if (K == 9 ? pb > 9 : false) return false;  // unrolled path: at most one producer block per tap
    const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
    const size_t fs = (size_t)rows * 64;
    const size_t px = ((size_t)3 * V * V + (size_t)3 * span) * 4;  // Ps + Xs alias the two image buffers
    if (px > 2 * buf) return false;
    const size_t lds = 2 * buf + fs;
    if (lds > (size_t)kLdsBytes) return false;
    pl.pb = pb;
    pl.rows = rows;
    pl.lds = lds;
    return true;
}

template <int PB, int TERMS, int KT>
int launch_stem_variant(const float *x, const float *P, const float *W12, const uint4 *Wp, const float *shift, void *y,
                        int N, int C, int T, int V, int K, const StemPlan &pl, bool bf16out, int opt, hipStream_t st) {
    const dim3 grid(ceil_div(T * V, NPB), C / 128, N);
    if (bf16out) {
        auto kern = stem_mfma_bf16_kernel<PB, TERMS, true, KT>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT), pl.lds, st, x, P, W12, Wp, shift, y, C, T, V, K, pl.rows, ablate_mask() | opt);
    } else {
        auto kern = stem_mfma_bf16_kernel<PB, TERMS, false, KT>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT), pl.lds, st, x, P, W12, Wp, shift, y, C, T, V, K, pl.rows, ablate_mask() | opt);
    }
    STGCN_LAUNCH_CHECK("stem_mfma_bf16_kernel");
    return STGCN_OK;
}

template <int TERMS>
int dispatch_stem(const float *x, const float *P, const float *W12, const uint4 *Wp, const float *shift, void *y, int N,
                  int C, int T, int V, int K, const StemPlan &pl, bool bf16out, int opt, hipStream_t st) {
#define GO(PB, KT) return launch_stem_variant<PB, TERMS, KT>(x, P, W12, Wp, shift, y, N, C, T, V, K, pl, bf16out, opt, st)
    if (K == 9) {
        if (pl.pb <= 3) GO(3, 9);
        if (pl.pb <= 6) GO(6, 9);
        GO(9, 9);
    }
    GO(9, 0);
#undef GO
}

// pixel rows per LDS image for the widest tile of a launch, plus one spare "dump" row
inline int rows_needed(int V, int K, int stride, int Tout) {
    int dt = ceil_div(NPB - 1, V);
    if (dt > Tout - 1) dt = Tout - 1;
    return (dt * stride + K) * V + 1;
}

struct Bf16Plan {
    int jpr = 0, rows = 0;
    size_t lds = 0;
};

inline bool plan_bf16(int Cin, int Cout, int V, int K, int stride, int Tout, int terms, Bf16Plan &pl) {
    if (Cin % CCB != 0 || (Cout % 128 != 0 && Cout != 64)) return false;
    const int rows = rows_needed(V, K, stride, Tout);
    const int jpr = ceil_div(rows - 1, NT);
    if (jpr > 3) return false;
    const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
    const size_t lds = 2 * buf;
    if (lds > (size_t)kLdsBytes) return false;
    pl.jpr = jpr;
    pl.rows = rows;
    pl.lds = lds;
    return true;
}

template <int JPR, int TERMS, int KT>
int launch_variant(const float *x, const uint4 *Wp, const float *shift, void *y, int N, int Cin, int Cout, int T, int V,
                   int K, int stride, int Tout, const Bf16Plan &pl, bool bf16out, float act_lo, hipStream_t st) {
    const dim3 grid(ceil_div(Tout * V, NPB), ceil_div(Cout, 128), N);
    if (bf16out) {
        auto kern = tcn_mfma_bf16_kernel<JPR, TERMS, true, KT>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT), pl.lds, st, x, Wp, shift, y, Cin, Cout, T, V, K, stride, Tout, pl.rows,
                           ablate_mask(), act_lo);
    } else {
        auto kern = tcn_mfma_bf16_kernel<JPR, TERMS, false, KT>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT), pl.lds, st, x, Wp, shift, y, Cin, Cout, T, V, K, stride, Tout, pl.rows,
                           ablate_mask(), act_lo);
    }
    STGCN_LAUNCH_CHECK("tcn_mfma_bf16_kernel");
    return STGCN_OK;
}

template <int TERMS>
int dispatch_tcn(const float *x, const uint4 *Wp, const float *shift, void *y, int N, int Cin, int Cout, int T, int V,
                 int K, int stride, int Tout, const Bf16Plan &pl, bool bf16out, float act_lo, hipStream_t st) {
#define GO(JPR, KT) \
    return launch_variant<JPR, TERMS, KT>(x, Wp, shift, y, N, Cin, Cout, T, V, K, stride, Tout, pl, bf16out, act_lo, st)
    if (K == 9) {
        if (pl.jpr == 1) GO(1, 9);
        if (pl.jpr == 2) GO(2, 9);
        GO(3, 9);
    }
    if (pl.jpr == 1) GO(1, 0);
    if (pl.jpr == 2) GO(2, 0);
    GO(3, 0);
#undef GO
}

}  // namespace

bool bf16_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags, bool fused) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16) return false;
    const int Tout = (T + 2 * ((K - 1) / 2) - K) / stride + 1;
    if (Tout < 1) return false;
    const int terms = math == STGCN_MATH_BF16X3 ? 3 : 1;
    if (fused) {
        StemPlan sp;
        return Cin == Cout && stride == 1 && plan_stem_bf16(Cin, V, K, T, terms, sp);
    }
    if (tcn_v6_supported(Cin, Cout, T, V, K, stride, flags) || tcn_v4_supported(Cin, Cout, T, V, K, stride, flags)) return true;
    Bf16Plan pl;
    return plan_bf16(Cin, Cout, V, K, stride, Tout, terms, pl);
}

bool bf16_packs(int Cin, int Cout, unsigned math) {
    return (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) && Cin % CCB == 0 && (Cout % 128 == 0 || Cout == 64);
}

int launch_tcn_pack_bf16(const float *W, const float *scale, void *Wp, int Cin, int Cout, int K, hipStream_t st) {
    const int CoutP = (Cout + 127) / 128 * 128;
    const size_t total = (size_t)Cin * CoutP * K * 2;
    hipLaunchKernelGGL(tcn_pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, scale,
                       (unsigned short *)Wp, Cin, Cout, CoutP, K);
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
    if (fused) {
        StemPlan sp;
        if (Cin != Cout || stride != 1 || !plan_stem_bf16(Cin, V, K, T, terms, sp))
            return fail(STGCN_ERR_UNSUPPORTED, "fused bf16 stem kernel does not cover C=%d V=%d K=%d T=%d", Cin, V, K, T);
        const int opt = (flags & STGCN_OUT_NTVC) ? OPT_OUT_NTVC : 0;
        if (terms == 3) return dispatch_stem<3>(x, P, W12, (const uint4 *)Wp, shift, y, N, Cin, T, V, K, sp, bf16out, opt, st);
        return dispatch_stem<1>(x, P, W12, (const uint4 *)Wp, shift, y, N, Cin, T, V, K, sp, bf16out, opt, st);
    }
    // K = 9, stride 1: large-tile persistent kernels — one wave per SIMD on 16x16x32 where that form covers the shape
    // (diagnostic builds: mask 8192 keeps the eight-wave kernel for A/B runs in one process)
    if (tcn_v6_supported(Cin, Cout, T, V, K, stride, flags) && !(ablate_mask() & 8192))
        return launch_tcn_v6(x, (const char *)Wp + tcn_packed_single_bytes(Cin, Cout, K, flags), shift, y, N, Cin, Cout, T, V, K,
                             stride, flags, st);
    if (tcn_v4_supported(Cin, Cout, T, V, K, stride, flags))
        return launch_tcn_v4(x, Wp, shift, y, N, Cin, Cout, T, V, K, stride, flags, st);
    Bf16Plan pl;
    if (Tout < 1 || !plan_bf16(Cin, Cout, V, K, stride, Tout, terms, pl))
        return fail(STGCN_ERR_UNSUPPORTED,
                    "bf16 MFMA kernel does not cover Cin=%d Cout=%d V=%d K=%d stride=%d T=%d (needs Cin%%16==0, "
                    "Cout%%128==0, tile rows that fit LDS)", Cin, Cout, V, K, stride, T);
    const uint4 *wp = (const uint4 *)Wp;
    const float act_lo = (flags & STGCN_RAW) ? -__builtin_huge_valf() : 0.f;  // raw = pre-activation (training-mode BN)
    if (terms == 3) return dispatch_tcn<3>(x, wp, shift, y, N, Cin, Cout, T, V, K, stride, Tout, pl, bf16out, act_lo, st);
    return dispatch_tcn<1>(x, wp, shift, y, N, Cin, Cout, T, V, K, stride, Tout, pl, bf16out, act_lo, st);
}

}  // namespace stgcn
