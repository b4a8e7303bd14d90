// Helpers shared by the bf16 matrix-core kernels (tcn_bf16.hip, stem_bf16_v4.hip).
#pragma once

#include "common.h"

namespace stgcn {
namespace bf16k {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

#ifdef STGCN_ABLATION  // diagnostic builds only: 1 = producer, 2 = MFMAs, 4 = epilogue, 8 = B reads, 16 = A loads
#define STGCN_ABL(bit) ((abl & (bit)) != 0)
#else
#define STGCN_ABL(bit) false
#endif

constexpr int CCB = 16;   // input channels per LDS chunk = one 16-deep k-step per tap
constexpr int PXB = 32;   // bytes per pixel row of one image
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

__device__ __forceinline__ int lds_off(int p, int h) { return p * PXB + ((h ^ ((p >> 3) & 1)) << 4); }

template <bool BF16OUT>
__device__ __forceinline__ void store_out(void *y, size_t idx, float v) {
    if constexpr (BF16OUT) reinterpret_cast<unsigned short *>(y)[idx] = (unsigned short)(pack_bf16x2(v, 0.f) & 0xffffu);
    else reinterpret_cast<float *>(y)[idx] = v;
}

struct TileGeomB {
    int q0, q_last, t_first, span, origin;
};

__device__ __forceinline__ TileGeomB tile_geom_b(int tile, int V, int K, int stride, int Tout, int np = 128) {
    TileGeomB g;
    g.q0 = tile * np;
    g.q_last = min(g.q0 + np, Tout * V) - 1;
    g.t_first = g.q0 / V;
    const int t_last = g.q_last / V;
    g.span = ((t_last - g.t_first) * stride + K) * V;
    g.origin = (g.t_first * stride - (K - 1) / 2) * V;
    return g;
}

template <int TERMS>
struct Frag2 {  // operands of one k-step for 2 MFMA blocks: [block] hi (+ lo)
    uint4 hi[2];
    uint4 lo[TERMS == 3 ? 2 : 1];
};

// 12 (or 4) MFMAs of one k-step: 2 channel blocks x 2 pixel blocks
template <int TERMS>
__device__ __forceinline__ void mfma_kstep_bf16(f32x16 (&acc)[2][2], const Frag2<TERMS> &a, const Frag2<TERMS> &b) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi[m]);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi[n]);
            if constexpr (TERMS == 3) {
                const bf16x8 al = __builtin_bit_cast(bf16x8, a.lo[m]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, b.lo[n]);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m][n], 0, 0, 0);
            }
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m][n], 0, 0, 0);
        }
    }
}


// the 6 (or 2) MFMAs of channel block m of one k-step (one half of mfma_kstep_bf16)
template <int TERMS>
__device__ __forceinline__ void mfma_half_bf16(f32x16 (&acc)[2][2], const Frag2<TERMS> &a, const Frag2<TERMS> &b, int m) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi[m]);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi[n]);
        if constexpr (TERMS == 3) {
            const bf16x8 al = __builtin_bit_cast(bf16x8, a.lo[m]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, b.lo[n]);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m][n], 0, 0, 0);
        }
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m][n], 0, 0, 0);
    }
}

// B-side operands of one k-step for NJ pixel blocks, and the half-step on a 2 x NJ accumulator tile
template <int TERMS, int NJ>
struct FragB {
    uint4 hi[NJ];
    uint4 lo[TERMS == 3 ? NJ : 1];
};

template <int TERMS, int NJ>
__device__ __forceinline__ void mfma_half_bf16(f32x16 (&acc)[2][NJ], const Frag2<TERMS> &a, const FragB<TERMS, NJ> &b, int m) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi[m]);
#pragma unroll
    for (int n = 0; n < NJ; ++n) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi[n]);
        if constexpr (TERMS == 3) {
            const bf16x8 al = __builtin_bit_cast(bf16x8, a.lo[m]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, b.lo[n]);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m][n], 0, 0, 0);
        }
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m][n], 0, 0, 0);
    }
}

}  // namespace bf16k
}  // namespace stgcn
