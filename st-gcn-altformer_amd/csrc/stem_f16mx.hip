// KF7 — the fused stem with the temporal conv as  fp16 x fp16  +  two block-scaled e4m3 residual products
// (STGCN_STEM_F16MX on top of STGCN_MATH_BF16X3): same tile, LDS images, feature phase, producer MFMAs, weight ring and
// epilogue as KF6 (stem_bf16_v6.hip: read that file first), with the three bf16 terms of a product replaced by
//
//     W y  =  Wh yh            v_mfma_f32_16x16x32_f16      K = 32 per instruction (a pair of k-steps), 16 cycles
//           + Wl y8 + W8 r8    v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3), K = 128 per instruction, 32 cycles
//
// where Wh = fp16(W), Wl = W - Wh, yh = fp16 of the activation, r = y - yh; y8, r8, W8, Wl8 are e4m3 roundings
// after power-of-two pre-scales that the instruction's E8M0 scale operands undo.  A residual product is ~2^-11 of the
// leading one, so three mantissa bits on its operands leave ~2^-15 relative: tools/math_error_2term.py measures
// max|err|/max|ref| = 1.9e-5 against the 1e-4 gate (NOT inside the stricter mixed criterion allclose(rtol 1e-4,
// atol 1e-5 max|ref|) that the three-bf16 arithmetic also meets: an opt-in mode, not the default), and
// tools/micro/mfma_mx.hip 1.42x the rate of the three bf16 terms in a bare loop on random data.  Per period of two chunks
// (18 k-steps): 288 fp16 MFMAs + 128 + 64 scaled MFMAs = 10,752 matrix-core cycles instead of 864 x 16 = 13,824.
//
// K of a scaled MFMA (layout probed on gfx950, tools/micro/mfma_mx_probe.hip: lane l holds row/column l & 15; its bytes
// 0-15 are K = 16g .. 16g+15, bytes 16-31 are K = 64 + 16g .. (g = l >> 4); the E8M0 byte of lane group b scales K block
// 32b .. 32b+31 — constant scales here, so only the pairing of A and B bytes matters):
//   * regular product of chunk c (taps 0-7): lane group g carries taps 2g (bytes 0-15 = the chunk's 16 channels) and 2g+1;
//     the activation bytes are two 16-byte reads of the chunk's e4m3 image rows (pixel row + tap*V).  A chunk's four pairs
//     hold its 2 terms x 8 channel blocks x 4 pixel blocks = 64 products, 16 per pair (one 8 KiB weight piece per pair:
//     the `lo` half of KF6's ring slot).
//   * tap 8 of both chunks of a period: one product per (term, channel block, pixel block) with lane group 0 = (even chunk
//     tap 8 | odd chunk tap 8) and zero weights in lane groups 1-3 — a quarter-full instruction, 64 per period in pair 4
//     (the pair whose two k-steps straddle the chunks; both chunks are resident there).  Its weights are the 8 KiB piece
//     of pair 4: sixteen operands of 16 rows x 32 B.
// Scales: K1 leaves max|x| and max|x| * (largest column abs-sum of each attention matrix) per clip, stgcn_stem_prepare the
// largest row abs-sums of the folded graph-conv matrix per feature group: their product bounds every |y| the producer can
// emit, so the pre-scaled values stay below e4m3's 448 by construction (no saturation path), whatever the input's units.
#include <type_traits>

#include "bf16_common.h"


namespace stgcn {

namespace {

// max(x, 0) as ONE v_max_f32 (fmaxf canonicalises its operand first: a second v_max per element in the producer's slots)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

using namespace bf16k;

constexpr int NP6 = 256;   // output pixels per tile
constexpr int NT6 = 256;   // threads per workgroup: one wave per SIMD
constexpr int KT6 = 9;     // temporal taps
constexpr int FRAG6 = 1024;
constexpr int PAIR6 = 16 * FRAG6;   // weights of one pair: 8 blocks of 16 channels x (hi, lo)
constexpr int RING6 = 3 * PAIR6;
constexpr int EPI6 = 4096; // epilogue staging per wave: 16 channels x 64 pixels fp32

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x8 = __attribute__((ext_vector_type(8))) int;

// matrix-core slots of a pair.  Regular pair (48): 4 fp16 MFMAs, then 14 x (scaled, fp16, fp16), then 2 scaled;
// the pair with the tap-8 products (96): 4 fp16, then 28 x (scaled, scaled, fp16), then 8 scaled.
constexpr bool mx_slot_reg(int s) { return s >= 4 && (s >= 46 || (s - 4) % 3 == 0); }
constexpr int mx_index_reg(int s) { return s >= 46 ? 14 + (s - 46) : (s - 4) / 3; }
constexpr int hi_index_reg(int s) { return s < 4 ? s : 4 + 2 * ((s - 4) / 3) + ((s - 4) % 3 - 1); }
constexpr bool mx_slot_t8(int s) { return s >= 4 && (s >= 88 || (s - 4) % 3 != 2); }
constexpr int mx_index_t8(int s) { return s >= 88 ? 56 + (s - 88) : 2 * ((s - 4) / 3) + (s - 4) % 3; }
constexpr int hi_index_t8(int s) { return s < 4 ? s : 4 + (s - 4) / 3; }
typedef __attribute__((address_space(3))) void *lptr6_t;

__device__ __forceinline__ void dma16v6(const void *g, unsigned lds_addr) {
    // M0 = LDS destination (wave-uniform).  M0 is declared clobbered instead of saved and restored around every transfer:
    // nothing else in these kernels lives in M0, and the three extra scalar instructions per transfer are not free when a
    // single wave owns the SIMD (they sit in the MFMA stream).
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait6() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void dma_wait6_keep4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }

struct FragB6 { uint4 hi[4], lo[4]; };     // activations of one pair: 4 pixel blocks of 16

// compile-time loop: f(std::integral_constant<int, I>{}) for I = I0 .. N-1
template <int I, int N, class F>
__device__ __forceinline__ void static_for6(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for6<I + 1, N>(f);
    }
}


struct TileInfo6 {
    int n, Vh, j0, half;
    TileGeomB g;
};

// WIDE: V0 = joints of the first half, tpc1 = tiles of a clip's second half (tiles_per_clip counts both halves)
template <bool BF16OUT>
__global__ __launch_bounds__(NT6) void stem_f16mx_kernel(
    const uint4 *__restrict__ pfrag, const float *__restrict__ x, int xsc, int xsp, const float *__restrict__ W12,
    const uint4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int C, int T, int V, int ROWS,
    int tiles_per_clip, int ntiles, int abl, unsigned long long *dbg, const float *__restrict__ meta,
    const float *__restrict__ bounds) {
    constexpr int TERMS = 3;                 // (LDS budget of the three-image form: fp16 image + two fp8 images = 64 B per row)
    constexpr bool WIDE = false;
    const int V0 = 0, tpc1 = 0;
#ifdef STGCN_ABLATION  // in-kernel cycle stamps (diagnostic builds only; dbg == NULL otherwise)
#define V6_STAMP(var) unsigned long long var = 0; if (dbg) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); }
#define V6_ACC(slot, a, b) if (dbg) { tsum[slot] += (b) - (a); }
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
#define V6_STAMP(var)
#define V6_ACC(slot, a, b)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem6[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = pixel quarter of the tile
    const int TV = T * V;
    const int nch = C / CCB;                 // channel chunks (C = 128 -> 8)
    const int npairs = nch * KT6 / 2;        // K = 32 steps per tile (nch even: host side)
    const int img_bytes = ROWS * PXB;
    const int buf_bytes = img_bytes * (TERMS == 3 ? 2 : 1);
    // LDS carve: W12 (bf16 hi/lo) | weight ring (3 pairs) | images buf0, buf1 (= epilogue staging, 4 x 4 KiB) | Fs | Pf
    uint4 *W12q = reinterpret_cast<uint4 *>(smem6);
    char *ring = smem6 + C * W12P * 4;
    char *buf0 = ring + RING6;
    char *buf1 = buf0 + buf_bytes;
    uint4 *Fs = reinterpret_cast<uint4 *>(buf0 + max(2 * buf_bytes, 4 * EPI6));
    // WIDE, three-term arithmetic: the half's 24 fragments (24 KiB) sit in the second image buffer, which is idle from the
    // end of a tile's main loop to the next tile's first period (the budget has no 24 KiB of its own)
    const uint4 *Pf = (WIDE && TERMS == 3) ? reinterpret_cast<const uint4 *>(buf1) : Fs + 4 * ROWS;
    const unsigned lds0 = (unsigned)(size_t)(lptr6_t)smem6;
    const unsigned ring_lds = lds0 + (unsigned)(ring - smem6);
    const unsigned pf_lds = lds0 + (unsigned)(reinterpret_cast<const char *>(Pf) - smem6);
    // a chunk buffer: [fp16 image: ROWS x 32 B, swizzled as lds_off][e4m3 image of y: ROWS x 16 B][e4m3 image of the residual]
    const int img8h = img_bytes, img8l = img_bytes + ROWS * 16;

    const int cg = blockIdx.y;               // 128-channel group of the output
    // the four weight fragments this wave DMAs per pair: 16-channel blocks 2*wave, 2*wave+1, images hi and lo
    const uint4 *wsrc = Wp + ((size_t)(cg * 8 + 2 * wave) * npairs * 2) * 64 + lane;
    // fragment d = (block-in-wave, image) of weight pair `qsrc` (index within a tile's pairs) -> ring slot `slot`
    auto dma_frag = [&](int qsrc, int slot, int d) {
        const int bw = d >> 1, img = d & 1;
        dma16v6(wsrc + ((size_t)(bw * npairs + qsrc) * 2 + img) * 64, ring_lds + slot * PAIR6 + ((2 * wave + bw) * 2 + img) * FRAG6);
    };
    // tile -> clip, joint half and geometry.  WIDE: a clip's tiles alternate between the halves (half-0 tile i, half-1 tile i,
    // ...; the first half may own one more), so that the two column halves of a frame range are written close in time
    auto tile_info = [&](int tile) {
        TileInfo6 ti;
        ti.n = tile / tiles_per_clip;
        const int r = tile - ti.n * tiles_per_clip;
        int idx;
        if (r < 2 * tpc1) { ti.half = r & 1; idx = r >> 1; }
        else { ti.half = 0; idx = r - tpc1; }
        ti.Vh = ti.half ? V - V0 : V0;
        ti.j0 = ti.half ? V0 : 0;
        ti.g = tile_geom_b(idx, ti.Vh, KT6, 1, T, NP6);
        return ti;
    };
    auto dma_pfrag = [&](int tile) {         // 12 KiB: the clip's attention fragments -> Pf  (WIDE: the half's 24 KiB)
        if constexpr (WIDE) {
            const TileInfo6 ti = tile_info(tile);
            const uint4 *src = pfrag + ((size_t)ti.n * 48 + ti.half * 24) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 6; ++i) dma16v6(src + (wave + 4 * i) * 64, pf_lds + (wave + 4 * i) * FRAG6);
        } else {
            const int n = tile / tiles_per_clip;
            const uint4 *src = pfrag + (size_t)n * 12 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 3; ++i) dma16v6(src + (wave + 4 * i) * 64, pf_lds + (wave + 4 * i) * FRAG6);
        }
    };

    // ---- features of a tile from x and the clip's attention fragments (see stem_bf16_v4.hip, FK form) -------------
    struct XRegs { float xa[WIDE ? 16 : 8]; float xp[3]; };
    auto load_x = [&](XRegs &xr, int tile, int u) {
        int ln = tid & 63;                   // opaque per call: keeps lane-only address terms from being hoisted and spilled
        asm volatile("" : "+v"(ln));
        const int mb = u >> 1, hh = u & 1;
        TileInfo6 ti;
        if constexpr (WIDE) ti = tile_info(tile);
        else {
            ti.n = tile / tiles_per_clip;
            ti.g = tile_geom_b(tile - ti.n * tiles_per_clip, V, KT6, 1, T, NP6);
        }
        const int n = ti.n;
        const TileGeomB g = ti.g;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (size_t)n * 3 * TV), 0, (unsigned)(3 * TV * 4), 0x00020000);
        const int tf = g.t_first - (KT6 - 1) / 2 + 4 * mb;
        if constexpr (WIDE) {
            // (every offset is computed unconditionally and made opaque before the select: with the product inside the
            //  conditional hipcc turns each of the 19 selects into a branch around its load)
            const int k = ln & 3, t = tf + ((ln & 15) >> 2), v0 = 8 * (ln >> 4);
            const bool okr = (k < 3) & (t >= 0) & (t < T);
            unsigned base = (unsigned)((k * xsc + (t * V + v0) * xsp) * 4);
            asm volatile("" : "+v"(base));
#pragma unroll
            for (int j = 0; j < 16; ++j) {     // joints 0-31 and 32-63: the two k-steps of the aggregation
                const int dv = (j & 7) + 32 * (j >> 3);
                const unsigned off = (okr & (v0 + dv < V)) ? base + (unsigned)(dv * xsp * 4) : 0x7ffffff0u;   // (&: no short-circuit branch)
                xr.xa[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            }
            const int t2 = tf + (ln >> 4), w = 16 * hh + (ln & 15);        // w: column within the half
            const bool ok = (t2 >= 0) & (t2 < T) & (w < ti.Vh);
            unsigned base2 = (unsigned)(((t2 * V + ti.j0 + w) * xsp) * 4);
            asm volatile("" : "+v"(base2));
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2) {
                const unsigned off = ok ? base2 + (unsigned)(k2 * xsc * 4) : 0x7ffffff0u;
                xr.xp[k2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            }
        } else {
            {
                const int k = ln & 3, t = tf + ((ln & 15) >> 2), v0 = 8 * (ln >> 4);
                const bool okr = k < 3 && t >= 0 && t < T;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned off = (okr && v0 + j < V) ? (unsigned)((k * xsc + (t * V + v0 + j) * xsp) * 4) : 0x7ffffff0u;
                    xr.xa[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
                }
            }
            {
                const int t = tf + (ln >> 4), w = 16 * hh + (ln & 15);
                const bool ok = t >= 0 && t < T && w < V;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const unsigned off = ok ? (unsigned)((k * xsc + (t * V + w) * xsp) * 4) : 0x7ffffff0u;
                    xr.xp[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
                }
            }
        }
    };
    auto feature_unit = [&](const TileInfo6 &ti, int u, const XRegs &xr) {
        const TileGeomB &g = ti.g;
        int ln = tid & 63;
        asm volatile("" : "+v"(ln));
        const int mb = u >> 1, hh = u & 1;
        f32x4 d[3];
        if constexpr (WIDE) {
            float xk[2][8];
#pragma unroll
            for (int j = 0; j < 16; ++j) xk[j >> 3][j & 7] = xr.xa[j];
#pragma unroll
            for (int s = 0; s < 3; ++s) d[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 xh, xl;
                split8(xk[ks], xh, xl);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, xh), al = __builtin_bit_cast(bf16x8, xl);
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int f = ((s * 2 + hh) * 2 + ks) * 2;
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, Pf[(f + 0) * 64 + ln]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, Pf[(f + 1) * 64 + ln]);
                    d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, d[s], 0, 0, 0);
                    d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d[s], 0, 0, 0);
                    d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d[s], 0, 0, 0);
                    d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d[s], 0, 0, 0);
                }
            }
        } else {
            uint4 xh, xl;
            split8(xr.xa, xh, xl);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, xh), al = __builtin_bit_cast(bf16x8, xl);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, Pf[((s * 2 + hh) * 2 + 0) * 64 + ln]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, Pf[((s * 2 + hh) * 2 + 1) * 64 + ln]);
                d[s] = f32x4{0.f, 0.f, 0.f, 0.f};
                d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, d[s], 0, 0, 0);
                d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d[s], 0, 0, 0);
                d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d[s], 0, 0, 0);
                d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d[s], 0, 0, 0);
            }
        }
        const int Vh = ti.Vh;                                 // joints of this tile's pixel space (= V unless WIDE)
        const int w = 16 * hh + (ln & 15);
        const int p = (4 * mb + (ln >> 4)) * Vh + w;         // pixel row of the tile
        const int gi = g.origin + p;
        const bool valid = p < g.span && gi >= 0 && gi < T * Vh; // else: the temporal conv's zero padding
        const float one = valid ? 1.f : 0.f;
        const float fa[8] = {d[0][0] * one, d[0][1] * one, d[0][2] * one, d[1][0] * one,
                             d[1][1] * one, d[1][2] * one, d[2][0] * one, d[2][1] * one};
        const float fb[8] = {d[2][2] * one, xr.xp[0] * one, xr.xp[1] * one, xr.xp[2] * one, one, 0.f, 0.f, 0.f};
        uint4 ha, la, hb, lb;
        split8(fa, ha, la);
        split8(fb, hb, lb);
        if (w < Vh && p < ROWS) {
            Fs[p] = ha;
            Fs[(size_t)ROWS + p] = hb;
            Fs[(size_t)2 * ROWS + p] = la;
            Fs[(size_t)3 * ROWS + p] = lb;
        }
    };
    // units wave, wave+4, wave+8 arrive prefetched; any further ones (narrow frames only) are loaded here
    auto feature_phase = [&](int tile, const XRegs &x0, const XRegs &x1, const XRegs &x2) {
        TileInfo6 ti;
        if constexpr (WIDE) ti = tile_info(tile);
        else {
            ti.n = tile / tiles_per_clip;
            ti.Vh = V;
            ti.j0 = ti.half = 0;
            ti.g = tile_geom_b(tile - ti.n * tiles_per_clip, V, KT6, 1, T, NP6);
        }
        const TileGeomB &g = ti.g;
        const int need = min(ROWS, ((g.span + 15) >> 4) << 4);       // rows the producer will read
        const int nun = (((need + ti.Vh - 1) / ti.Vh + 3) >> 2) * 2; // M-blocks x 2 joint halves
        const bool two = ti.Vh > 16;
        for (int u = wave; u < nun; u += 4) {
            if (!two && (u & 1)) continue;
            if (u == wave) feature_unit(ti, u, x0);
            else if (u == wave + 4) feature_unit(ti, u, x1);
            else if (u == wave + 8) feature_unit(ti, u, x2);
            else {
                XRegs xr;
                load_x(xr, tile, u);
                feature_unit(ti, u, xr);
            }
        }
    };

    // ---- producer: one 16-pixel block of chunk `ch` -> hi/lo images of `buf` (see stem_bf16_v4.hip) -----------------
    const int pl = lane & 15, pg = lane >> 4;
    struct Prod { uint4 wh, wl, fb; f32x4 d; int p; };
    auto prod_load = [&](Prod &pr, int ch, int bi) {
        pr.p = bi * 16 + pl;
        pr.wh = W12q[(size_t)(pg & 1) * C + ch * CCB + pl];
        pr.wl = W12q[(size_t)(2 + (pg & 1)) * C + ch * CCB + pl];
        pr.fb = Fs[(size_t)pg * ROWS + pr.p];
    };
    auto prod_mfma = [&](Prod &pr) {
        const bf16x8 f = __builtin_bit_cast(bf16x8, pr.fb);
        pr.d = f32x4{0.f, 0.f, 0.f, 0.f};
        pr.d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pr.wh), f, pr.d, 0, 0, 0);
        pr.d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pr.wl), f, pr.d, 0, 0, 0);
    };
    // The same inside the slot-structured loop, with a VGPR destination: the result feeds VALU work, and through the
    // builtin hipcc computed it in AGPRs and copied it out (4 v_accvgpr_read + an s_nop 6 per block).  As inline asm the
    // hazard recogniser does not see the matrix-core write: the consumer sits four slots (>= 4 main MFMAs, 64+ cycles)
    // further down, far beyond the 7 wait states a 4-pass MFMA result needs; the second MFMA accumulates onto the first
    // with identical vDst / SrcC (back-to-back forwarding).
    auto prod_mfma_slots = [&](Prod &pr) {
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        const u32x4 wh = __builtin_bit_cast(u32x4, pr.wh), wl = __builtin_bit_cast(u32x4, pr.wl), fb = __builtin_bit_cast(u32x4, pr.fb);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(pr.d) : "v"(wh), "v"(fb));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(pr.d) : "v"(wl), "v"(fb));
    };
    // split of one produced value: h = fp16 (round to nearest), residual v - h exact in fp32 and <= 2^-11 |v|; both v and the residual also as e4m3 after the tile's power-of-two pre-scales (sH, sL: no value can
    // exceed 256 of e4m3's 448 by the bound K1 and stgcn_stem_prepare supply)
    using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
    auto prod_finish = [&](char *buf, const Prod &pr, float sY) {
        constexpr float sH = 0.015625f, sL = 32.f;
        const float v0 = relu1(pr.d[0]) * sY, v1 = relu1(pr.d[1]) * sY, v2 = relu1(pr.d[2]) * sY, v3 = relu1(pr.d[3]) * sY;
        const f16x2 h01 = f16x2{(_Float16)v0, (_Float16)v1}, h23 = f16x2{(_Float16)v2, (_Float16)v3};
        const int off = lds_off(pr.p, pg >> 1) + (pg & 1) * 8;
        *reinterpret_cast<uint2 *>(buf + off) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
        int y8 = __builtin_amdgcn_cvt_pk_fp8_f32(v0 * sH, v1 * sH, 0, false);
        y8 = __builtin_amdgcn_cvt_pk_fp8_f32(v2 * sH, v3 * sH, y8, true);
        int l8 = __builtin_amdgcn_cvt_pk_fp8_f32((v0 - (float)h01[0]) * sL, (v1 - (float)h01[1]) * sL, 0, false);
        l8 = __builtin_amdgcn_cvt_pk_fp8_f32((v2 - (float)h23[0]) * sL, (v3 - (float)h23[1]) * sL, l8, true);
        *reinterpret_cast<int *>(buf + img8h + pr.p * 16 + pg * 4) = y8;
        *reinterpret_cast<int *>(buf + img8l + pr.p * 16 + pg * 4) = l8;
    };

    // ---- one-time setup ----------------------------------------------------------------------
    for (int e = tid; e < C * 2; e += NT6) {   // W12 -> bf16 hi/lo planes [hi k0-7][hi k8-15][lo k0-7][lo k8-15] of [C] x 16 B
        const int c = e >> 1, kh = e & 1;
        float w8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w8[i] = W12[c * W12P + kh * 8 + i];
        uint4 hi, lo;
        split8(w8, hi, lo);
        W12q[(size_t)kh * C + c] = hi;
        W12q[(size_t)(2 + kh) * C + c] = lo;
    }
    int tile = blockIdx.x;
    {
        XRegs x0 = {}, x1 = {}, x2 = {};
        if (tile < ntiles) {
            dma_pfrag(tile);
            load_x(x0, tile, wave);
            load_x(x1, tile, wave + 4);
            load_x(x2, tile, wave + 8);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) { dma_frag(0, 0, d); dma_frag(1, 1, d); }
        dma_wait6();
        __syncthreads();                      // W12q, Pf(tile), weight pairs 0 and 1 landed
        if (tile < ntiles) feature_phase(tile, x0, x1, x2);
        __syncthreads();
    }

    // ring bookkeeping without divisions: slot of the current pair, and (slot, source index) of the pair two ahead
    int gq = 0, slot0 = 0, slot2 = 2, q2 = 2 % npairs;
    const int sel = lane >> 5, chh = (lane >> 4) & 1;   // B fragment lane groups: step of the pair, channel half
    for (; tile < ntiles; tile += gridDim.x) {
        TileInfo6 ti;
        if constexpr (WIDE) ti = tile_info(tile);
        else {
            ti.n = tile / tiles_per_clip;
            ti.Vh = V;
            ti.j0 = ti.half = 0;
            ti.g = tile_geom_b(tile - ti.n * tiles_per_clip, V, KT6, 1, T, NP6);
        }
        const int n = ti.n;
        const TileGeomB g = ti.g;
        const int Vh = ti.Vh;
        const int nblk = (g.span + 15) >> 4;
        const int next_tile = tile + gridDim.x;

        V6_STAMP(t_0)
        // ---- this tile's scale: |y| <= By for every value the producer can emit (bounds[n]: max|x| of the clip and
        // max|x| * max column abs-sum of each attention matrix, from K1; meta[0..4]: max row abs-sums of the folded
        // graph-conv matrix per feature group, from stgcn_stem_prepare).  The producer multiplies its values by sY = 2^eY
        // with By * sY < 2^14: no fp16 overflow whatever the input's units, and the e4m3 pre-scales become constants
        // (y' * 2^-6 < 2^8 < 448; the residual of the fp16 rounding, <= 2^-11 y', times 2^5 likewise).  The epilogue undoes
        // sY together with the weights' 2^eW (meta[5]) in the multiply-add that applies the shift.
        float sY, osc;
        {
            const float By = 1.01f * (meta[0] * bounds[n * 4 + 1] + meta[1] * bounds[n * 4 + 2] + meta[2] * bounds[n * 4 + 3] +
                                      meta[3] * bounds[n * 4 + 0] + meta[4]);
            int e = (int)((__float_as_uint(By) >> 23) & 0xffu) - 126;       // By < 2^e
            e = max(-60, min(60, e));
            const int eY = __builtin_amdgcn_readfirstlane(14 - e);
            sY = __uint_as_float((unsigned)(eY + 127) << 23);
            osc = __uint_as_float((unsigned)max(1, min(254, 127 - eY - (int)meta[5])) << 23);   // (clamped: a valid power of two in any case)
        }
        constexpr float sH = 0.015625f, sL = 32.f;                          // 2^-6, 2^5
        constexpr int scYh = 127 + 6, scYl = 127 - 5;   // E8M0 bytes (byte 0 = opsel 0 of the scale operand) that undo them in the scaled MFMA
        // chunk 0 of this tile
        for (int b = wave; b < nblk; b += 4) {
            Prod pr;
            prod_load(pr, 0, b);
            prod_mfma(pr);
            prod_finish(buf0, pr, sY);
        }
        // LDS offsets of this lane's activation rows per tap, for the wave's FIRST 16-pixel block (see stem_bf16_v6.hip)
        unsigned boff[KT6];
        // ... and of its e4m3 rows: lane group g = lane >> 4 of a scaled MFMA carries taps 2g and 2g + 1 (16 channels = 16
        // bytes each); the tap-8 product of a period carries tap 8 of both chunks in lane group 0
        unsigned o8a, o8b, o8t;
        {
            const int q = g.q0 + wave * 64 + (lane & 15);
            const int prow = q - g.t_first * Vh;
#pragma unroll
            for (int tap = 0; tap < KT6; ++tap) boff[tap] = (unsigned)lds_off(prow + tap * Vh, chh);
            const int g4 = lane >> 4;
            o8a = (unsigned)((prow + 2 * g4 * Vh) * 16);
            o8b = (unsigned)((prow + (2 * g4 + 1) * Vh) * 16);
            o8t = (unsigned)((prow + 8 * Vh) * 16);
        }
        f32x4 acc[8][4];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                      // chunk 0 visible
        V6_STAMP(t_1)
        V6_ACC(0, t_0, t_1)

        auto rd = [&](const char *p) { return *reinterpret_cast<const uint4 *>(p); };
        auto cat8 = [](const uint4 &a, const uint4 &b) {
            return i32x8{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, (int)b.z, (int)b.w};
        };
        // fp16 activation fragments of the pair with local steps l0 = 2*pi, l1 = l0 + 1 (as stem_bf16_v6.hip)
        auto load_b = [&](FragB6 &b, auto l0_c, auto nb_c) {
            constexpr int l0 = decltype(l0_c)::value, l1 = l0 + 1, nb = decltype(nb_c)::value;
            const char *b0 = (l0 >= KT6 ? buf1 : buf0), *b1 = (l1 >= KT6 ? buf1 : buf0);
            const unsigned o0 = boff[l0 % KT6], o1 = boff[l1 % KT6];
            b.hi[nb] = rd((sel ? b1 : b0) + (sel ? o1 : o0) + (nb * 16 * PXB));
        };
        // e4m3 fragment (32 B per lane) of pixel block nb: taps 2g, 2g+1 of one chunk's image `img` (0: y, 1: residual) ...
        auto load_b8 = [&](const char *buf, int img, int nb) {
            const char *p = buf + img8h + img * (ROWS * 16) + nb * 256;
            return cat8(rd(p + o8a), rd(p + o8b));
        };
        // ... and tap 8 of the period's two chunks (even chunk in buf0, odd chunk in buf1)
        auto load_b8t = [&](int img, int nb) {
            const int o = img8h + img * (ROWS * 16) + nb * 256;
            return cat8(rd(buf0 + o + o8t), rd(buf1 + o + o8t));
        };
        using IC0 = std::integral_constant<int, 0>;
        FragB6 b_cur = {}, b_nxt = {};
        i32x8 b8A[4], b8B[4];                 // two sets of e4m3 activation fragments (4 pixel blocks each), see the table below
        uint4 ah0n = rd(ring + slot0 * PAIR6 + lane * 16);
        i32x8 a8n = cat8(rd(ring + slot0 * PAIR6 + lane * 16 + 1 * FRAG6), rd(ring + slot0 * PAIR6 + lane * 16 + 3 * FRAG6));
        static_for6<0, 4>([&](auto nb_c) {      // pair 0 of the tile (chunk 0 is complete)
            constexpr int nb = decltype(nb_c)::value;
            load_b(b_cur, IC0{}, nb_c);
            b8A[nb] = load_b8(buf0, 0, nb);
            b8B[nb] = b8A[nb];
        });
        const int nper = nch / 2;
        for (int per = 0; per < nper; ++per) {
            if (per + 1 == nper && next_tile < ntiles) dma_pfrag(next_tile);   // Pf is idle after the tile's feature phase
            static_for6<0, 9>([&](auto pi_c) {
                constexpr int pi = decltype(pi_c)::value;
                constexpr int l0 = 2 * pi;
                constexpr bool T8 = pi == 4;            // the pair that also carries the period's tap-8 residual products
                constexpr int NS = T8 ? 96 : 48;        // matrix-core slots of the pair
                constexpr int FPS = 96 / NS;            // filler positions per slot
                // residual products of this pair: chunk (even: buf0 / odd: buf1), index within the chunk's four pairs
                constexpr int ci = pi < 4 ? pi : pi - 5;              // 0..3 (unused when T8)
                constexpr int term = ci >> 1, mbase = 4 * (ci & 1);   // term 0: W_lo8 x y8, term 1: W_hi8 x r8
                // production: pairs 0-2 -> chunk 2per+1 into buf1; pairs 5-7 -> chunk 2per+2 into buf0 (3, 3, 2 blocks)
                constexpr int win = pi <= 2 ? 0 : (pi >= 5 && pi <= 7 ? 1 : -1);
                constexpr int wpi = win == 0 ? pi : pi - 5;
                constexpr int npb = win < 0 ? 0 : (wpi < 2 ? 3 : 2);
                char *pbuf = win == 0 ? buf1 : buf0;
                const int pch = min(2 * per + 1 + (win == 1 ? 1 : 0), nch - 1);
                const int slot1 = slot0 == 2 ? 0 : slot0 + 1;
                const char *aslot = ring + slot0 * PAIR6 + lane * 16;
                const char *anext = ring + slot1 * PAIR6 + lane * 16;
                // tap-8 operand `op` = term*8 + mb of THIS pair's piece / of the NEXT pair's: 16 rows x 32 B, two per fragment
                const char *a8t_cur = ring + slot0 * PAIR6 + FRAG6 + (lane & 15) * 32;
                const char *a8t_nxt = ring + slot1 * PAIR6 + FRAG6 + (lane & 15) * 32;
                const int a8t_live = lane < 16 ? -1 : 0;   // lane groups 1-3 of a tap-8 operand are zero (their K range is unused)
                auto rd_a8t = [&](const char *base, int op) {
                    const char *p = base + (op >> 1) * 2 * FRAG6 + (op & 1) * 512;
                    const uint4 r0 = rd(p), r1 = rd(p + 16);
                    return i32x8{(int)r0.x & a8t_live, (int)r0.y & a8t_live, (int)r0.z & a8t_live, (int)r0.w & a8t_live,
                                 (int)r1.x & a8t_live, (int)r1.y & a8t_live, (int)r1.z & a8t_live, (int)r1.w & a8t_live};
                };
                V6_STAMP(t_p0)
                uint4 ah[2];
                ah[0] = ah0n;
                i32x8 a8[2];
                a8[0] = a8n;
                Prod pr = {};
                f16x2 ph01 = {}, ph23 = {};
                float pv0 = 0.f, pv1 = 0.f, pv2 = 0.f, pv3 = 0.f;
                int poff = 0, p8off = 0, py8 = 0, pl8 = 0;
                // legacy filler v (0 .. 95): what stem_bf16_v6.hip places between its MFMAs, minus the lo-image work
                auto filler = [&](auto v_c) {
                    constexpr int v = decltype(v_c)::value;
                    if constexpr (v % 12 == 2 && v / 12 < 7) ah[(v / 12 + 1) & 1] = rd(aslot + ((v / 12 + 1) * 2) * FRAG6);
                    if constexpr (v >= 40 && v < 48 && v % 2 == 0) {
                        constexpr int nb = (v - 40) / 2;
                        constexpr int ln = (l0 + 2) % 18;     // (pair 8 -> pair 0 of the next period / tile: chunk in buf0)
                        load_b(b_nxt, std::integral_constant<int, ln>{}, std::integral_constant<int, nb>{});
                    }
                    if constexpr (npb > 0 && v >= 8 && (v - 8) / 28 < npb) {
                        constexpr int b = (v - 8) / 28, w = (v - 8) % 28;
                        if constexpr (w == 0) {
                            pr.p = min(wave + 4 * (3 * wpi + b), nblk - 1) * 16 + pl;
                            pr.wh = W12q[(size_t)(pg & 1) * C + pch * CCB + pl];
                        }
                        if constexpr (w == 1) pr.wl = W12q[(size_t)(2 + (pg & 1)) * C + pch * CCB + pl];
                        if constexpr (w == 2) pr.fb = Fs[(size_t)pg * ROWS + pr.p];
                        if constexpr (w == 8) prod_mfma(pr);
                        if constexpr (w == 13) { pv0 = relu1(pr.d[0]); pv1 = relu1(pr.d[1]); pv2 = relu1(pr.d[2]); pv3 = relu1(pr.d[3]); }
                        if constexpr (w == 14) { pv0 *= sY; pv1 *= sY; pv2 *= sY; pv3 *= sY; }
                        if constexpr (w == 15) { ph01 = f16x2{(_Float16)pv0, (_Float16)pv1}; ph23 = f16x2{(_Float16)pv2, (_Float16)pv3}; }
                        if constexpr (w == 16) { poff = lds_off(pr.p, pg >> 1) + (pg & 1) * 8; p8off = img8h + pr.p * 16 + pg * 4; }
                        if constexpr (w == 17)
                            *reinterpret_cast<uint2 *>(pbuf + poff) = make_uint2(__builtin_bit_cast(unsigned, ph01), __builtin_bit_cast(unsigned, ph23));
                        if constexpr (w == 18) {
                            py8 = __builtin_amdgcn_cvt_pk_fp8_f32(pv0 * sH, pv1 * sH, 0, false);
                            py8 = __builtin_amdgcn_cvt_pk_fp8_f32(pv2 * sH, pv3 * sH, py8, true);
                        }
                        if constexpr (w == 19) *reinterpret_cast<int *>(pbuf + p8off) = py8;
                        if constexpr (w == 20) { pv0 -= (float)ph01[0]; pv1 -= (float)ph01[1]; }
                        if constexpr (w == 21) { pv2 -= (float)ph23[0]; pv3 -= (float)ph23[1]; }
                        if constexpr (w == 22) {
                            pl8 = __builtin_amdgcn_cvt_pk_fp8_f32(pv0 * sL, pv1 * sL, 0, false);
                            pl8 = __builtin_amdgcn_cvt_pk_fp8_f32(pv2 * sL, pv3 * sL, pl8, true);
                        }
                        if constexpr (w == 23) *reinterpret_cast<int *>(pbuf + p8off + ROWS * 16) = pl8;
                    }
                    if constexpr (v >= 4 && v < 8) dma_frag(q2, slot2, v - 4);
                    if constexpr (v == 88) ah0n = rd(anext);
                };
                // fillers of the residual products (same position -> slot mapping as the legacy ones).
                //   e4m3 activation sets:  pairs 0,1 use A (chunk 2per, y) | 2,3 use B (chunk 2per, r) | 4: A (tap 8, y) then
                //   B (tap 8, r) | 5,6 use A (chunk 2per+1, y) | 7,8 use B (chunk 2per+1, r); each set is loaded one pair ahead.
                auto mxfill = [&](auto u_c) {
                    constexpr int u = decltype(u_c)::value;
                    if constexpr (!T8) {
                        // weight operand mbl = 1..3 of this pair, one operand ahead (two 16-byte reads each)
                        if constexpr (u == 10 || u == 34 || u == 58) {
                            constexpr int mbl = (u - 10) / 24 + 1;
                            a8[mbl & 1] = cat8(rd(aslot + ((mbl * 2) * 2 + 1) * FRAG6), rd(aslot + ((mbl * 2 + 1) * 2 + 1) * FRAG6));
                        }
                        // operand 0 of the next pair
                        if constexpr (u == 90) {
                            if constexpr (pi == 3) a8n = rd_a8t(a8t_nxt, 0);
                            else a8n = cat8(rd(anext + 1 * FRAG6), rd(anext + 3 * FRAG6));
                        }
                        if constexpr (u >= 64 && u < 72 && u % 2 == 0) {
                            constexpr int nb = (u - 64) / 2;
                            if constexpr (pi == 1) b8B[nb] = load_b8(buf0, 1, nb);
                            if constexpr (pi == 3) b8A[nb] = load_b8t(0, nb);
                            if constexpr (pi == 6) b8B[nb] = load_b8(buf1, 1, nb);
                            if constexpr (pi == 8) b8A[nb] = load_b8(buf0, 0, nb);   // (next period's / tile's even chunk)
                        }
                    } else {
                        // operand k = 1..15 of the tap-8 products, read behind the last product of operand k - 2
                        if constexpr (u >= 3 && (u + 3) % 6 == 0 && (u + 3) / 6 <= 15) {
                            constexpr int kk = (u + 3) / 6;
                            a8[kk & 1] = rd_a8t(a8t_cur, kk);
                        }
                        if constexpr (u == 94) a8n = cat8(rd(anext + 1 * FRAG6), rd(anext + 3 * FRAG6));
                        if constexpr (u >= 8 && u < 16 && u % 2 == 0) b8B[(u - 8) / 2] = load_b8t(1, (u - 8) / 2);
                        if constexpr (u >= 52 && u < 60 && u % 2 == 0) b8A[(u - 52) / 2] = load_b8(buf1, 0, (u - 52) / 2);
                    }
                };
                static_for6<0, NS>([&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
                    constexpr bool mx = T8 ? mx_slot_t8(s) : mx_slot_reg(s);
                    if constexpr (!mx) {
                        constexpr int h = T8 ? hi_index_t8(s) : hi_index_reg(s);
                        constexpr int mb = h >> 2, nb = h & 3;
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[mb & 1]),
                                                                              __builtin_bit_cast(f16x8, b_cur.hi[nb]), acc[mb][nb], 0, 0, 0);
                        asm volatile("" : "+a"(acc[mb][nb]));
                    } else if constexpr (!T8) {
                        constexpr int m = mx_index_reg(s), mbl = m >> 2, nb = m & 3, mb = mbase + mbl;
                        constexpr bool setB = (pi == 2 || pi == 3 || pi == 7 || pi == 8);
                        const i32x8 bb = setB ? b8B[nb] : b8A[nb];
                        acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[mbl & 1], bb, acc[mb][nb], 0, 0, 0,
                                                                                       term ? scYh : scYl, 0, term ? scYl : scYh);
                        asm volatile("" : "+a"(acc[mb][nb]));
                    } else {
                        constexpr int m = mx_index_t8(s), t8 = m >> 5, kk = m >> 2, mb = kk & 7, nb = m & 3;
                        const i32x8 bb = t8 ? b8B[nb] : b8A[nb];
                        acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[kk & 1], bb, acc[mb][nb], 0, 0, 0,
                                                                                       t8 ? scYh : scYl, 0, t8 ? scYl : scYh);
                        asm volatile("" : "+a"(acc[mb][nb]));
                    }
                    static_for6<s * FPS, (s + 1) * FPS>(filler);
                    static_for6<s * FPS, (s + 1) * FPS>(mxfill);
                    __builtin_amdgcn_sched_barrier(0);
                });
                b_cur = b_nxt;
                V6_STAMP(t_s1)
                V6_ACC((pi == 4 ? 6 : (pi == 0 ? 7 : (pi == 3 ? 5 : 4))), t_p0, t_s1)
                dma_wait6();                  // pair gq+2's weights (issued early in this pair) have landed
                __syncthreads();              // ... and are visible; produced image rows are visible; slot gq%3 is free
                V6_STAMP(t_s2)
                V6_ACC(2, t_s1, t_s2)
                ++gq;
                slot0 = slot1;
                slot2 = slot2 == 2 ? 0 : slot2 + 1;
                q2 = q2 + 1 == npairs ? 0 : q2 + 1;
            });
        }
        dma_wait6();                          // (the next tile's attention fragments)
        V6_STAMP(t_2)
        V6_ACC(1, t_1, t_2)

        // ---- epilogue: each 16-channel x 64-pixel block through this wave's 4 KiB staging slice, 16 B per lane ----------
        // D[row = channel 4*(lane>>4) + r][col = pixel lane&15] per 16x16 block.  Store addresses = scalar base + one
        // per-lane term; the last tile of a clip keeps per-lane bounds checks.
        // WIDE: the second image buffer is idle from here on (every wave is past the last pair's barrier): the next tile's
        // attention fragments go there now, land during the stores and are waited for in front of the feature phase
        if constexpr (WIDE)
            if (next_tile < ntiles) dma_pfrag(next_tile);
        XRegs xn0, xn1, xn2;                  // next tile's x: in flight while this tile's results are stored
        load_x(xn0, min(next_tile, ntiles - 1), wave);
        load_x(xn1, min(next_tile, ntiles - 1), wave + 4);
        load_x(xn2, min(next_tile, ntiles - 1), wave + 8);
        __builtin_amdgcn_sched_barrier(0);
        float *stg = reinterpret_cast<float *>(buf0 + wave * EPI6);
        const int qw = g.q0 + wave * 64;
        const bool full = g.q0 + NP6 - 1 <= g.q_last;            // (scalar) every pixel of the tile lies inside the clip
        if constexpr (WIDE) {
            // half-space pixel q = t*Vh + v'  ->  pixel t*V + j0 + v' of the clip
            auto clip_pixel = [&](int q) { const int t = q / Vh; return t * V + ti.j0 + (q - t * Vh); };
            if (abl & OPT_OUT_NTVC) {
                // (N,T,V,C): as the narrow form, with the four pixels a lane stores mapped one by one
                unsigned pt[4];
                bool pok[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int q = qw + it * 16 + (lane >> 2);
                    pok[it] = q <= g.q_last;
                    pt[it] = (unsigned)(clip_pixel(min(q, g.q_last)) * C + 4 * (lane & 3));
                }
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) {
                    const int ob = cg * 128 + mb * 16;
                    const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 4 * (lane >> 4));
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const int px = nb * 16 + (lane & 15);
                        const float4 v = make_float4(fmaxf(fmaf(acc[mb][nb][0], osc, sh4.x), 0.f), fmaxf(fmaf(acc[mb][nb][1], osc, sh4.y), 0.f),
                                                     fmaxf(fmaf(acc[mb][nb][2], osc, sh4.z), 0.f), fmaxf(fmaf(acc[mb][nb][3], osc, sh4.w), 0.f));
                        *reinterpret_cast<float4 *>(stg + px * 16 + (((lane >> 4) ^ (px & 3)) << 2)) = v;
                    }
                    const size_t tbase = (size_t)n * TV * C + ob;            // scalar
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int idx = it * 64 + lane, px = idx >> 2, sl = idx & 3;
                        const float4 v = *reinterpret_cast<const float4 *>(stg + px * 16 + ((sl ^ (px & 3)) << 2));
                        if (pok[it]) {
                            if constexpr (BF16OUT)
                                *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(y) + tbase + pt[it]) =
                                    make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                            else
                                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(y) + tbase + pt[it]) = v;
                        }
                    }
                }
            } else {
                // (N,C,T,V): a lane owns ONE pair of pixels of the wave's 64 (2*(lane&31), +1: V, V0 and Vh are even, so a
                // pair never straddles a frame or the halves and sits 8-byte aligned in the clip) and walks the 16 channel
                // rows of a block two at a time: eight 8-byte stores per block
                const int qp = qw + 2 * (lane & 31);
                const bool pok = qp <= g.q_last;
                const unsigned lterm = (unsigned)((lane >> 5) * TV + clip_pixel(min(qp, g.q_last)));
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) {
                    const int ob = cg * 128 + mb * 16;
                    const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 4 * (lane >> 4));
                    const float shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            stg[(4 * (lane >> 4) + r) * 64 + nb * 16 + (lane & 15)] = fmaxf(fmaf(acc[mb][nb][r], osc, shv[r]), 0.f);
                    const size_t tbase = ((size_t)n * C + ob) * TV;           // scalar
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const float2 v = *reinterpret_cast<const float2 *>(stg + (it * 2 + (lane >> 5)) * 64 + 2 * (lane & 31));
                        const size_t sbase = tbase + (size_t)(it * 2) * TV;    // scalar
                        if (pok) {
                            if constexpr (BF16OUT)
                                *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned short *>(y) + sbase + lterm) = pack_bf16x2(v.x, v.y);
                            else
                                *reinterpret_cast<float2 *>(reinterpret_cast<float *>(y) + sbase + lterm) = v;
                        }
                    }
                }
            }
        } else if (abl & OPT_OUT_NTVC) {
            // (N,T,V,C): staged pixel-major [64 px][16 ch]: a lane's four channels of a pixel are one 16-byte slot
            // (slot XOR-swizzled by the pixel: conflict-free b128 accesses); a store then writes 16 pixels x 64 B
            const unsigned lterm = (unsigned)((lane >> 2) * C + 4 * (lane & 3));
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int ob = cg * 128 + mb * 16;
                const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 4 * (lane >> 4));
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int px = nb * 16 + (lane & 15);
                    const float4 v = make_float4(fmaxf(fmaf(acc[mb][nb][0], osc, sh4.x), 0.f), fmaxf(fmaf(acc[mb][nb][1], osc, sh4.y), 0.f),
                                                 fmaxf(fmaf(acc[mb][nb][2], osc, sh4.z), 0.f), fmaxf(fmaf(acc[mb][nb][3], osc, sh4.w), 0.f));
                    *reinterpret_cast<float4 *>(stg + px * 16 + (((lane >> 4) ^ (px & 3)) << 2)) = v;
                }
                const size_t tbase = ((size_t)n * TV + qw) * C + ob;      // scalar
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = it * 64 + lane, px = idx >> 2, sl = idx & 3;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + px * 16 + ((sl ^ (px & 3)) << 2));
                    if (full || qw + px <= g.q_last) {
                        if constexpr (BF16OUT) {
                            unsigned short *yb = reinterpret_cast<unsigned short *>(y) + tbase + (size_t)(it * 16) * C;
                            *reinterpret_cast<uint2 *>(yb + lterm) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                        } else {
                            float *yb = reinterpret_cast<float *>(y) + tbase + (size_t)(it * 16) * C;
                            *reinterpret_cast<float4 *>(yb + lterm) = v;
                        }
                    }
                }
            }
        } else {
            // element offset of (row = idx>>4, 4-pixel group c4 = 4*(idx&15)) for idx = it*64 + lane
            const unsigned lterm = (unsigned)((lane >> 4) * TV + 4 * (lane & 15));
            const int c4l = 4 * (lane & 15);
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int ob = cg * 128 + mb * 16;
                const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 4 * (lane >> 4));
                const float shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stg[(4 * (lane >> 4) + r) * 64 + nb * 16 + (lane & 15)] = fmaxf(fmaf(acc[mb][nb][r], osc, shv[r]), 0.f);
                const size_t tbase = ((size_t)n * C + ob) * TV + qw;      // scalar
                const bool al16 = ((tbase & 3) == 0) && (TV % 4 == 0);    // 16-byte (8-byte for bf16) aligned rows
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const float4 v = *reinterpret_cast<const float4 *>(stg + (it * 4 + (lane >> 4)) * 64 + c4l);
                    const size_t sbase = tbase + (size_t)(it * 4) * TV;    // scalar
                    if (full && al16) {
                        if constexpr (BF16OUT)
                            *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(y) + sbase + lterm) =
                                make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                        else
                            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(y) + sbase + lterm) = v;
                    } else {                                     // last tile of a clip / unaligned rows: element by element
                        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (qw + c4l + e <= g.q_last) store_out<BF16OUT>(y, sbase + lterm + e, e4[e]);
                    }
                }
            }
        }
        V6_STAMP(t_3)
        V6_ACC(3, t_2, t_3)
        if (next_tile < ntiles) {             // its fragments landed at the last stage barrier, its x during the stores;
            if constexpr (WIDE) {             // (WIDE: fragments issued at the head of this epilogue — landed, then visible)
                dma_wait6();
                __syncthreads();
            }
            feature_phase(next_tile, xn0, xn1, xn2);   // Fs lies behind the staging area: no barrier needed in front
            V6_STAMP(t_4)
            __syncthreads();                  // Fs complete, every wave's staging reads done (chunk 0 overwrites buf0)
        }
        V6_STAMP(t_5)
    }
#ifdef STGCN_ABLATION
    if (dbg && lane == 0 && blockIdx.x < 8 && blockIdx.y == 0)
        for (int i = 0; i < 8; ++i) dbg[(blockIdx.x * 8 + wave) * 8 + i] = tsum[i];
#endif
}

struct V7Plan {
    int rows = 0, tiles_per_clip = 0;
    size_t lds = 0;
};

inline bool plan_v7(int C, int T, int V, int K, V7Plan &pl) {
    if (K != KT6 || C % 128 != 0 || V > 32) return false;
    int dt = ceil_div(NP6 - 1, V);
    if (dt > T - 1) dt = T - 1;
    const int span = (dt + K) * V;
    if (ceil_div(ceil_div(span, 16), 4) > 8) return false;   // producer: 3 + 3 + 2 blocks per wave and chunk
    const int rows = (span + 15) / 16 * 16;
    const size_t buf = (size_t)rows * PXB * 2;               // fp16 image + two e4m3 images = 64 B per row
    const size_t img = 2 * buf > (size_t)4 * EPI6 ? 2 * buf : (size_t)4 * EPI6;
    pl.lds = (size_t)C * W12P * 4 + RING6 + img + (size_t)rows * 64 + 12 * FRAG6;
    if (pl.lds > (size_t)kLdsBytes) return false;
    pl.rows = rows;
    pl.tiles_per_clip = ceil_div(T * V, NP6);
    return true;
}

// ---- weight packing ---------------------------------------------------------------------------------------------------
// meta (8 floats, device): [0..2] max_o sum_k |W12[o][3s+k]| (s = 0..2), [3] max_o sum_k |W12[o][9+k]|, [4] max_o |W12[o][12]|,
// [5] eW: the packed weights are W' * 2^eW with max|W'| * 2^eW < 2^14 (fp16 range used well whatever the weights' scale; the
// kernel's epilogue undoes it), [6], [7] unused
__global__ __launch_bounds__(256) void f16mx_meta_kernel(const float *__restrict__ W12, const float *__restrict__ W,
                                                         const float *__restrict__ scale, float *__restrict__ meta, int C, int Cin) {
    __shared__ float red[6][4];
    float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int o = threadIdx.x; o < C; o += 256) {
        const float *r = W12 + (size_t)o * W12P;
#pragma unroll
        for (int s = 0; s < 3; ++s) m[s] = fmaxf(m[s], fabsf(r[3 * s]) + fabsf(r[3 * s + 1]) + fabsf(r[3 * s + 2]));
        m[3] = fmaxf(m[3], fabsf(r[9]) + fabsf(r[10]) + fabsf(r[11]));
        m[4] = fmaxf(m[4], fabsf(r[12]));
    }
    for (size_t e = threadIdx.x; e < (size_t)C * Cin * KT6; e += 256) m[5] = fmaxf(m[5], fabsf(scale[e / ((size_t)Cin * KT6)] * W[e]));
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        for (int o = 32; o > 0; o >>= 1) m[i] = fmaxf(m[i], __shfl_down(m[i], o, 64));
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = m[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r[6];
        for (int i = 0; i < 6; ++i) r[i] = fmaxf(fmaxf(red[i][0], red[i][1]), fmaxf(red[i][2], red[i][3]));
        for (int i = 0; i < 5; ++i) meta[i] = r[i];
        int e = (int)((__float_as_uint(r[5]) >> 23) & 0xffu) - 126;   // max|W'| < 2^e
        e = max(-90, min(90, e));
        e = max(-60, min(60, e));
        meta[5] = (float)(14 - e);
        meta[6] = meta[7] = 0.f;
    }
}

__device__ __forceinline__ unsigned e4m3_of(float v) { return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xffu; }

// Wq: [ob 0..7 per 128-channel group][pair q][img][lane] x 16 B — img 0: fp16 fragment of (ob, pair) in KF6's pair order;
// img 1: the pair's 8 KiB piece of e4m3 weights, fragment ob of it (see the file header)
__global__ void f16mx_pack_kernel(const float *__restrict__ W, const float *__restrict__ scale, const float *__restrict__ meta,
                                  uint4 *__restrict__ Wq, int Cin, int Cout) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;   // one thread per 16-byte unit
    const int npairs = Cin / CCB * KT6 / 2;
    if (e >= (size_t)(Cout / 16) * npairs * 2 * 64) return;
    const int l = (int)(e & 63), img = (int)((e >> 6) & 1);
    const int q = (int)((e >> 7) % npairs), obg = (int)((e >> 7) / npairs);    // obg: 16-channel block over ALL output channels
    const int ob = obg & 7, cg = obg >> 3;
    const float sW = exp2f(meta[5]);
    constexpr float sWH = 0.015625f, sWL = 32.f;       // 2^-6, 2^5: the e4m3 pre-scales (undone by constant E8M0 bytes in the kernel)
    auto wv = [&](int o, int c, int tap) { return scale[o] * W[((size_t)o * Cin + c) * KT6 + tap] * sW; };
    unsigned out[4] = {0, 0, 0, 0};
    if (img == 0) {
        const int o = obg * 16 + (l & 15), f = 2 * q + (l >> 5);
        const int c0 = (f / KT6) * CCB + 8 * ((l >> 4) & 1), tap = f % KT6;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16 a = (_Float16)wv(o, c0 + 2 * j, tap), b = (_Float16)wv(o, c0 + 2 * j + 1, tap);
            out[j] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
        }
    } else {
        const int pi = q % 9, per = q / 9;
        int o, chunk, tap, term;
        if (pi != 4) {
            const int ci = pi < 4 ? pi : pi - 5;
            term = ci >> 1;
            const int mb = 4 * (ci & 1) + (ob >> 1), half = ob & 1;
            chunk = 2 * per + (pi > 4 ? 1 : 0);
            o = cg * 128 + mb * 16 + (l & 15);
            tap = 2 * (l >> 4) + half;
        } else {
            const int op = 2 * ob + (l >> 5), u16 = l & 31;       // operand (term*8 + mb): 16 rows x 32 B
            term = op >> 3;
            o = cg * 128 + (op & 7) * 16 + (u16 >> 1);
            chunk = 2 * per + (u16 & 1);
            tap = 8;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float w = wv(o, chunk * CCB + k, tap);
            const float wh = (float)(_Float16)w;
            const unsigned b = term == 0 ? e4m3_of((w - wh) * sWL) : e4m3_of(wh * sWH);
            out[k >> 2] |= b << (8 * (k & 3));
        }
    }
    Wq[e] = make_uint4(out[0], out[1], out[2], out[3]);
}

int launch_v7(const uint4 *pf, const float *x, int xsc, int xsp, const float *W12, const uint4 *Wq, const float *shift, void *y,
              int N, int C, int T, int V, const V7Plan &pl, bool bf16out, int opt, int num_cu, const float *meta,
              const float *bounds, hipStream_t st) {
    const int ntiles = N * pl.tiles_per_clip;
    const dim3 grid(ntiles < num_cu ? ntiles : num_cu, C / 128, 1);
    if (bf16out) {
        auto kern = stem_f16mx_kernel<true>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT6), pl.lds, st, pf, x, xsc, xsp, W12, Wq, shift, y, C, T, V, pl.rows,
                           pl.tiles_per_clip, ntiles, opt, debug_buffer(), meta, bounds);
    } else {
        auto kern = stem_f16mx_kernel<false>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT6), pl.lds, st, pf, x, xsc, xsp, W12, Wq, shift, y, C, T, V, pl.rows,
                           pl.tiles_per_clip, ntiles, opt, debug_buffer(), meta, bounds);
    }
    STGCN_LAUNCH_CHECK("stem_f16mx_kernel");
    return STGCN_OK;
}

}  // namespace

bool stem_f16mx_supported(int C, int T, int V, int K, unsigned flags) {
    if ((flags & STGCN_MATH_MASK) != STGCN_MATH_BF16X3 || !(flags & STGCN_STEM_F16MX)) return false;
    V7Plan pl;
    return T >= 1 && plan_v7(C, T, V, K, pl);
}

// bytes of KF7's weights behind the other packings of the prep blob: 256-byte header (meta) + the pair-order blob
size_t stem_f16mx_prep_bytes(int C, int K) { return 256 + (size_t)C * C * K * 4; }

// W12: the folded graph-conv matrix already in the prep blob; dst: 256-byte header + weights
int launch_stem_f16mx_prepare(const float *W12, const float *Wt, const float *t_scale, void *dst, int C, hipStream_t st) {
    float *meta = (float *)dst;
    hipLaunchKernelGGL(f16mx_meta_kernel, dim3(1), dim3(256), 0, st, W12, Wt, t_scale, meta, C, C);
    STGCN_LAUNCH_CHECK("f16mx_meta_kernel");
    const size_t units = (size_t)(C / 16) * (C / CCB * KT6 / 2) * 2 * 64;
    hipLaunchKernelGGL(f16mx_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, Wt, t_scale, meta,
                       (uint4 *)((char *)dst + 256), C, C);
    STGCN_LAUNCH_CHECK("f16mx_pack_kernel");
    return STGCN_OK;
}

int launch_stem_f16mx(const float *x, bool x_ntvc, const void *pfrag, const void *bounds, const void *prep_w12, const void *mx_blob,
                      const float *shift, void *out, int N, int C, int T, int V, int K, unsigned flags, hipStream_t st) {
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    const int opt = (flags & STGCN_OUT_NTVC) ? OPT_OUT_NTVC : 0;
    V7Plan pl;
    if (!plan_v7(C, T, V, K, pl))
        return fail(STGCN_ERR_UNSUPPORTED, "stem f16mx kernel does not cover C=%d T=%d V=%d K=%d", C, T, V, K);
    if ((size_t)3 * T * V * 4 >= ((size_t)1 << 31))
        return fail(STGCN_ERR_UNSUPPORTED, "stem f16mx: clip of T=%d V=%d exceeds a buffer resource", T, V);
    int dev = 0, num_cu = 256;
    STGCN_HIP_CHECK(hipGetDevice(&dev));
    STGCN_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int xsc = x_ntvc ? 1 : T * V, xsp = x_ntvc ? 3 : 1;
    return launch_v7((const uint4 *)pfrag, x, xsc, xsp, (const float *)prep_w12, (const uint4 *)((const char *)mx_blob + 256), shift,
                     out, N, C, T, V, pl, bf16out, opt, num_cu, (const float *)mx_blob, (const float *)bounds, st);
}

}  // namespace stgcn
