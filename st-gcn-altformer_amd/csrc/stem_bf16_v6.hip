// KF6 — the fused stem on the bf16 matrix cores with ONE WAVE PER SIMD (256 threads, 4 waves, up to 512 registers each) on
// v_mfma_f32_16x16x32_bf16.  Headline kernel for V <= ~25 (the 256-pixel tile); KF4 (stem_bf16_v4.hip) keeps the rest.
//
// Same tile, LDS images, in-kernel feature computation and producer as KF4's FK form — what changes:
//   * ownership: a wave owns ALL 128 output channels of its 64-pixel quarter (8 x 4 accumulator blocks of 16 x 16 = 128
//     registers).  Every weight fragment is read once by each of 4 waves instead of twice by 8, an activation fragment feeds
//     eight channel blocks: LDS read traffic per MFMA is halved; and there is no second wave on the SIMD to lose matrix-pipe
//     arbitration to (KF4's older waves ran ahead and then waited 23 % of their time at the stage barriers).
//   * with nothing else on the SIMD, clumps of other instructions between MFMAs are no longer hidden, so the loop is written
//     as SLOTS: one MFMA followed by at most a couple of fillers (compile-time loop over the slots, sched_barrier(0) after
//     each; the `filler` lambda says which filler sits in which slot), no branches, as little scalar work as possible
//     (division-free ring counters, M0 clobbered rather than saved around an LDS-DMA).  The first version of this kernel
//     with KF4's fenced phases was 2.7 % SLOWER than KF4 (hipcc gathers 30-40 instructions between groups of MFMAs); a
//     uniform branch per producer piece cost 5 %.
//   * the MFMA shape: the kernel runs at the rate the chip sustains for issued bf16 MFMA on random operands (power-limited,
//     DESIGN.md section 3), so what is left is energy per FLOP: the 16x16x32 form delivers ~1.12-1.15x the FLOP/s of
//     32x32x16 under that limit (MI355X_MICROARCH.md "DVFS give-back" (7); this pool: 1,836 vs 1,644 TFLOP/s with this
//     kernel's LDS operand traffic, tools/micro/mfma_shape.hip) — measured here as an 11 % higher clock at equal wall time
//     before the scalar work was trimmed, 2.8 % faster than the same kernel on 32x32x16 after.
//
// K = 32 of one MFMA = 16 channels x TWO consecutive k-steps of the flat (16-channel chunk, tap) sequence.  A tile has
// nch * 9 steps (72: even), so steps pair up without padding; every second chunk boundary falls inside a pair
// (tap 8 of chunk c with tap 0 of chunk c+1: the two lane halves of a B fragment then read different image buffers).
// The loop is written per PERIOD of 9 pairs = 2 chunks (static taps, static buffers), periods in a dynamic loop:
//   pairs 0-2 produce chunk 2p+1 into buf1, pair 4 straddles, pairs 5-7 produce chunk 2p+2 into buf0 — one pair before
//   the chunk's first reader, so that the reader's activation fragments can be prefetched across the barrier.
// Weights: repacked per pair ([16-channel block][pair][hi|lo][lane] x 16 B, stgcn_stem_prepare), ring of 3 pair slots
// (16 KiB each) filled by LDS-DMA two pairs ahead — issued early in a pair, waited for (vmcnt(0)) at its end, so that at a
// pair's start BOTH the current and the next pair are resident and published: the next pair's first fragments are read
// before the barrier, no pair opens with an exposed LDS read.  A wave keeps only the current and the next 16-channel
// block's weight fragments in registers (read one block ahead of use).
// Per-tile tail: epilogue staged 16 channels x 64 pixels at a time through LDS (16-byte stores, scalar base + one per-lane
// term), next tile's x loads in flight during the stores, feature phase, chunk-0 production.
//
// WIDE (stem_bf16_v6w.hip compiles this file with STGCN_V6_WIDE): frames of 32 < V <= 64 joints (the two-hand graph, V = 46).
// The temporal conv never mixes joints, so the joint axis is cut into two halves [0, V0) and [V0, V) (V0 a multiple of 4,
// both halves <= 32 joints) and each (clip, half) is walked like a narrow clip of Vh joints: same tile, images, producer
// and main loop.  What differs sits in the per-tile tail only: the aggregation u_s = x P_s sums over ALL V joints (two
// k-steps of 32 per feature MFMA; 24 attention fragments per half, which live in the idle second image buffer between
// the main loops instead of a region of their own), and the epilogue maps a half-space pixel (t, v') to the clip's
// (t, j0 + v'): pairs of pixels stay 8-byte aligned (V, V0 even), so rows go out as 8-byte stores.
// The two instantiations live in separate translation units so that neither perturbs the other's code generation.
#include <type_traits>

#include "bf16_common.h"

#ifdef STGCN_V6_WIDE
#define V6W true
#else
#define V6W false
#endif

namespace stgcn {

namespace {

// max(x, 0) as ONE v_max_f32 (fmaxf canonicalises its operand first: a second v_max per element in the producer's slots)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

using namespace bf16k;

// Result stores of the NARROW kernel's epilogue: NON-TEMPORAL (round 3).  The 0.5 GB of output per launch pass through the same
// 4 MiB L2s that serve the weight ring (2.4 GB per launch, re-read by every tile); streamed as `nt` whole 128-byte lines displace
// less of it: 0.9-1.7 % off the kernel in same-box A/Bs of two libraries (tools/ab_libs.sh; write-through `sc1` stores
// instead: 11 % slower).  The WIDE form keeps plain stores: its 8-byte pair stores fill a line from two workgroups a tile
// apart, and pushed out early as `nt` halves they measured 3 % slower.  -DV6_PLAIN_STORES builds the plain form for that A/B.
template <bool NT>
__device__ __forceinline__ void st_out4(float *p, const float4 &v) {
#ifndef V6_PLAIN_STORES
    if constexpr (NT) {
        using f32x4v = __attribute__((ext_vector_type(4))) float;
        __builtin_nontemporal_store(f32x4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4v *>(p));
        return;
    }
#endif
    *reinterpret_cast<float4 *>(p) = v;
}

constexpr int NP6 = 256;   // output pixels per tile
constexpr int NT6 = 256;   // threads per workgroup: one wave per SIMD
constexpr int KT6 = 9;     // temporal taps
constexpr int FRAG6 = 1024;
constexpr int PAIR6 = 16 * FRAG6;   // weights of one pair: 8 blocks of 16 channels x (hi, lo)
constexpr int RING6 = 3 * PAIR6;
constexpr int EPI6 = 4096; // epilogue staging per wave: 16 channels x 64 pixels fp32

using f32x4 = __attribute__((ext_vector_type(4))) float;
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

#ifndef STGCN_V6_WIDE
// weight packing for KF6: Wq (bf16) index ((((ob*npairs + q)*2 + img)*64 + lane)*8 + j
//   o = ob*16 + (lane&15); step f = 2q + (lane>>5); chunk f/9, tap f%9; c = chunk*16 + 8*((lane>>4)&1) + j
__global__ void tcn_pack_bf16_pairs_kernel(const float *__restrict__ W, const float *__restrict__ scale,
                                           unsigned short *__restrict__ Wq, int Cin, int Cout) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (weight, img)
    if (e >= (size_t)Cout * Cin * KT6 * 2) return;
    const int j = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    size_t r = e >> 9;
    const int img = (int)(r & 1);
    r >>= 1;
    const int npairs = Cin / CCB * KT6 / 2;
    const int q = (int)(r % npairs);
    const int ob = (int)(r / npairs);
    const int o = ob * 16 + (lane & 15);
    const int f = 2 * q + (lane >> 5);
    const int c = (f / KT6) * CCB + 8 * ((lane >> 4) & 1) + j, tap = f % KT6;
    const float w = scale[o] * W[((size_t)o * Cin + c) * KT6 + tap];
    const unsigned h = pack_bf16x2(w, 0.f) & 0xffffu;
    const unsigned l = pack_bf16x2(w - bf16_lo_to_f32(h), 0.f) & 0xffffu;
    Wq[e] = (unsigned short)(img ? l : h);
}
#endif

struct TileInfo6 {
    int n, Vh, j0, half;
    TileGeomB g;
};

// WIDE: V0 = joints of the first half, tpc1 = tiles of a clip's second half (tiles_per_clip counts both halves)
template <int TERMS, bool BF16OUT, bool WIDE>
__global__ __launch_bounds__(NT6) void stem_bf16_v6_kernel(
    const uint4 *__restrict__ pfrag, const float *__restrict__ x, int xsc, int xsp, const float *__restrict__ W12,
    const uint4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int C, int T, int V, int ROWS,
    int tiles_per_clip, int ntiles, int abl, unsigned long long *dbg, int V0, int tpc1) {
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
    auto prod_finish = [&](char *buf, const Prod &pr) {
        const float v0 = relu1(pr.d[0]), v1 = relu1(pr.d[1]), v2 = relu1(pr.d[2]), v3 = relu1(pr.d[3]);
        const unsigned h0 = pack_bf16x2(v0, v1), h1 = pack_bf16x2(v2, v3);
        const int off = lds_off(pr.p, pg >> 1) + (pg & 1) * 8;
        *reinterpret_cast<uint2 *>(buf + off) = make_uint2(h0, h1);
        if constexpr (TERMS == 3) {
            const unsigned l0 = pack_bf16x2(v0 - bf16_lo_to_f32(h0), v1 - bf16_hi_to_f32(h0));
            const unsigned l1 = pack_bf16x2(v2 - bf16_lo_to_f32(h1), v3 - bf16_hi_to_f32(h1));
            *reinterpret_cast<uint2 *>(buf + img_bytes + off) = make_uint2(l0, l1);
        }
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
        // chunk 0 of this tile
        for (int b = wave; b < nblk; b += 4) {
            Prod pr;
            prod_load(pr, 0, b);
            prod_mfma(pr);
            prod_finish(buf0, pr);
        }
        // LDS offsets of this lane's activation rows per tap, for the wave's FIRST 16-pixel block: block nb sits exactly
        // nb * 16 * PXB bytes further (16 more pixels leave the swizzle bit (row >> 3) & 1 alone), which rides in the
        // ds_read immediate — 9 offset registers instead of 36 (the kernel is at 256 VGPRs + copies through AGPRs).
        // Pixels past the clip's last one (last tile only) read rows of the image that exist but hold stale data: their
        // results are never stored.
        unsigned boff[KT6];
        {
            const int q = g.q0 + wave * 64 + (lane & 15);
            const int prow = q - g.t_first * Vh;
#pragma unroll
            for (int tap = 0; tap < KT6; ++tap) boff[tap] = (unsigned)lds_off(prow + tap * Vh, chh);
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
        // activation fragments of the pair with local steps l0 = 2*pi, l1 = l0 + 1 (0 .. 17 within a period: chunk l/9 in
        // buffer l/9, tap l%9); lanes 0-31 carry step l0, lanes 32-63 step l1
        auto load_b = [&](FragB6 &b, auto l0_c, auto nb_c, auto lo_c) {
            constexpr int l0 = decltype(l0_c)::value, l1 = l0 + 1, nb = decltype(nb_c)::value;
            constexpr bool lo_img = decltype(lo_c)::value;
            const char *b0 = (l0 >= KT6 ? buf1 : buf0), *b1 = (l1 >= KT6 ? buf1 : buf0);
            const unsigned o0 = boff[l0 % KT6], o1 = boff[l1 % KT6];
            const char *p = (sel ? b1 : b0) + (sel ? o1 : o0) + (nb * 16 * PXB) + (lo_img ? img_bytes : 0);
            if constexpr (lo_img) b.lo[nb] = rd(p); else b.hi[nb] = rd(p);
        };
        using IC0 = std::integral_constant<int, 0>;
        FragB6 b_cur = {}, b_nxt = {};
        uint4 ah0n = rd(ring + slot0 * PAIR6 + lane * 16), al0n = rd(ring + slot0 * PAIR6 + lane * 16 + FRAG6);
        static_for6<0, 4>([&](auto nb_c) {      // pair 0 of the tile (chunk 0 is complete)
            load_b(b_cur, IC0{}, nb_c, std::false_type{});
            if constexpr (TERMS == 3) load_b(b_cur, IC0{}, nb_c, std::true_type{});
        });
        const int nper = nch / 2;
        for (int per = 0; per < nper; ++per) {
            if constexpr (!WIDE)
                if (per + 1 == nper && next_tile < ntiles) dma_pfrag(next_tile);   // Pf is idle after the tile's feature phase
            static_for6<0, 9>([&](auto pi_c) {
                constexpr int pi = decltype(pi_c)::value;
                constexpr int l0 = 2 * pi;
                // production: pairs 0-2 -> chunk 2per+1 into buf1; pairs 5-7 -> chunk 2per+2 into buf0 (3, 3, 2 blocks)
                constexpr int win = pi <= 2 ? 0 : (pi >= 5 && pi <= 7 ? 1 : -1);
                constexpr int wpi = win == 0 ? pi : pi - 5;
                constexpr int npb = win < 0 ? 0 : (wpi < 2 ? 3 : 2);
                char *pbuf = win == 0 ? buf1 : buf0;
                const int pch = min(2 * per + 1 + (win == 1 ? 1 : 0), nch - 1);
                const int slot1 = slot0 == 2 ? 0 : slot0 + 1;
                const char *aslot = ring + slot0 * PAIR6 + lane * 16;
                const char *anext = ring + slot1 * PAIR6 + lane * 16;
                // weight fragments: current / next 16-channel block.  Block 0 of THIS pair was read during the previous one
                // (ah0n / al0n): pairs gq and gq+1 are both resident and published at a pair's start (their DMAs are issued
                // early in a pair and waited for at its end), so no pair opens with an exposed LDS read.
                V6_STAMP(t_p0)                // (diagnostic builds: time per kind of pair, slots 4-7 of the stamp buffer)
                uint4 ah[2], al[2];
                ah[0] = ah0n;
                if constexpr (TERMS == 3) al[0] = al0n;
                Prod pr = {};
                unsigned ph0 = 0, ph1 = 0;
                float pv0 = 0.f, pv1 = 0.f, pv2 = 0.f, pv3 = 0.f;
                int poff = 0;
                constexpr int NM = 32 * TERMS;                // MFMAs of the pair
                // filler v (0 .. 95; with TERMS == 1 three share a slot)
                auto filler = [&](auto v_c) {
                    constexpr int v = decltype(v_c)::value;
                    // next block's weight fragments, one block ahead: block mb+1 at fillers 12*mb + 2, + 3
                    if constexpr (v % 12 == 2 && v / 12 < 7) ah[(v / 12 + 1) & 1] = rd(aslot + ((v / 12 + 1) * 2) * FRAG6);
                    if constexpr (TERMS == 3 && v % 12 == 3 && v / 12 < 7) al[(v / 12 + 1) & 1] = rd(aslot + ((v / 12 + 1) * 2 + 1) * FRAG6);
                    // next pair's activation fragments (its chunk was published one pair ago at the latest)
                    if constexpr (v >= 40 && v < 48) {
                        constexpr int nb = (v - 40) / 2;
                        constexpr int ln = (l0 + 2) % 18;     // (pair 8 -> pair 0 of the next period / tile: chunk in buf0)
                        using LN = std::integral_constant<int, ln>;
                        using NB = std::integral_constant<int, nb>;
                        if constexpr (v % 2 == 0) load_b(b_nxt, LN{}, NB{}, std::false_type{});
                        else if constexpr (TERMS == 3) load_b(b_nxt, LN{}, NB{}, std::true_type{});
                    }
                    // producer blocks: block b of this pair occupies fillers 8 + 28*b ...
                    if constexpr (npb > 0 && v >= 8 && (v - 8) / 28 < npb) {
                        constexpr int b = (v - 8) / 28, w = (v - 8) % 28;
                        auto piece = [&]() {
                            if constexpr (w == 0) {
                                pr.p = min(wave + 4 * (3 * wpi + b), nblk - 1) * 16 + pl;
                                pr.wh = W12q[(size_t)(pg & 1) * C + pch * CCB + pl];
                            }
                            if constexpr (w == 1) pr.wl = W12q[(size_t)(2 + (pg & 1)) * C + pch * CCB + pl];
                            if constexpr (w == 2) pr.fb = Fs[(size_t)pg * ROWS + pr.p];
                            if constexpr (w == 10) { if constexpr (TERMS == 3) prod_mfma_slots(pr); else prod_mfma(pr); }   // (TERMS == 1 packs three fillers per slot: too close to the consumer for the unchecked form)
                            if constexpr (w == 14) {
                                pv0 = relu1(pr.d[0]); pv1 = relu1(pr.d[1]); pv2 = relu1(pr.d[2]); pv3 = relu1(pr.d[3]);
                            }
                            if constexpr (w == 15) { ph0 = pack_bf16x2(pv0, pv1); ph1 = pack_bf16x2(pv2, pv3); }
                            if constexpr (w == 16) poff = lds_off(pr.p, pg >> 1) + (pg & 1) * 8;
                            if constexpr (w == 17) *reinterpret_cast<uint2 *>(pbuf + poff) = make_uint2(ph0, ph1);
                            if constexpr (TERMS == 3 && w == 18) { pv0 -= bf16_lo_to_f32(ph0); pv1 -= bf16_hi_to_f32(ph0); }
                            if constexpr (TERMS == 3 && w == 19) { pv2 -= bf16_lo_to_f32(ph1); pv3 -= bf16_hi_to_f32(ph1); }
                            if constexpr (TERMS == 3 && w == 20) { ph0 = pack_bf16x2(pv0, pv1); ph1 = pack_bf16x2(pv2, pv3); }
                            if constexpr (TERMS == 3 && w == 21) *reinterpret_cast<uint2 *>(pbuf + img_bytes + poff) = make_uint2(ph0, ph1);
                        };
                        constexpr bool has_work = w <= 2 || w == 10 || (w >= 14 && w <= 17) || (TERMS == 3 && w >= 18 && w <= 21);
                        // (no branch around it: in the tile's last period the second window re-produces the last chunk
                        //  into the idle buffer — a uniform branch per piece cost more than the redundant work)
                        if constexpr (has_work) piece();
                    }
                    // weights of pair gq + 2 -> the slot pair gq - 1 occupied (its readers passed the last barrier); issued
                    // early so that they have landed by the end of the pair
                    if constexpr (v >= 4 && v < 8) dma_frag(q2, slot2, v - 4);
                    // block 0 of the next pair
                    if constexpr (v == 88) ah0n = rd(anext);
                    if constexpr (TERMS == 3 && v == 89) al0n = rd(anext + FRAG6);
                };
                static_for6<0, NM>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value;
                    // (the three terms of a block back to back on one accumulator: interleaving the pixel blocks instead
                    //  measured 2 % slower)
                    constexpr int mb = i / (4 * TERMS), nb = (i / TERMS) % 4, term = i % TERMS;
                    const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah[mb & 1]), b_h = __builtin_bit_cast(bf16x8, b_cur.hi[nb]);
                    if constexpr (TERMS == 3) {
                        const bf16x8 a_l = __builtin_bit_cast(bf16x8, al[mb & 1]), b_l = __builtin_bit_cast(bf16x8, b_cur.lo[nb]);
                        if constexpr (term == 0) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l, acc[mb][nb], 0, 0, 0);
                        else if constexpr (term == 1) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h, acc[mb][nb], 0, 0, 0);
                        else acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h, acc[mb][nb], 0, 0, 0);
                    } else {
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h, acc[mb][nb], 0, 0, 0);
                    }
                    // (the accumulators' home is the AGPR file: without this hint hipcc kept 16 of the 32 blocks in VGPRs and
                    //  copied each through an AGPR quad around its MFMAs — 64 v_accvgpr_write per period)
                    if constexpr (term == TERMS - 1) asm volatile("" : "+a"(acc[mb][nb]));
#ifndef V6_NOFILL   // (diagnostic variant: the MFMA stream alone; results are wrong)
                    static_for6<i * (96 / NM), (i + 1) * (96 / NM)>(filler);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                });
                b_cur = b_nxt;
                V6_STAMP(t_s1)
                V6_ACC((pi == 4 ? 6 : (pi == 0 ? 7 : (pi == 3 ? 5 : 4))), t_p0, t_s1)
                dma_wait6();                  // pair gq+2's weights (issued ~1,500 cycles ago) have landed
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
                        const float4 v = make_float4(fmaxf(acc[mb][nb][0] + sh4.x, 0.f), fmaxf(acc[mb][nb][1] + sh4.y, 0.f),
                                                     fmaxf(acc[mb][nb][2] + sh4.z, 0.f), fmaxf(acc[mb][nb][3] + sh4.w, 0.f));
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
                                st_out4<false>(reinterpret_cast<float *>(y) + tbase + pt[it], v);
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
                            stg[(4 * (lane >> 4) + r) * 64 + nb * 16 + (lane & 15)] = fmaxf(acc[mb][nb][r] + shv[r], 0.f);
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
                    const float4 v = make_float4(fmaxf(acc[mb][nb][0] + sh4.x, 0.f), fmaxf(acc[mb][nb][1] + sh4.y, 0.f),
                                                 fmaxf(acc[mb][nb][2] + sh4.z, 0.f), fmaxf(acc[mb][nb][3] + sh4.w, 0.f));
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
                            st_out4<true>(yb + lterm, v);
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
                        stg[(4 * (lane >> 4) + r) * 64 + nb * 16 + (lane & 15)] = fmaxf(acc[mb][nb][r] + shv[r], 0.f);
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
                            st_out4<true>(reinterpret_cast<float *>(y) + sbase + lterm, v);
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

struct V6Plan {
    int rows = 0, tiles_per_clip = 0;
    size_t lds = 0;
    int v0 = 0, tpc1 = 0;                     // WIDE: joints of the first half, tiles of a clip's second half
};

// rows of the image one (pixel space of Vh joints) tile needs; 0 when the producer cannot cover it
inline int v6_rows(int T, int Vh, int K) {
    int dt = ceil_div(NP6 - 1, Vh);
    if (dt > T - 1) dt = T - 1;
    const int span = (dt + K) * Vh;
    if (ceil_div(ceil_div(span, 16), 4) > 8) return 0;       // producer: 3 + 3 + 2 blocks per wave and chunk
    return (span + 15) / 16 * 16;
}

#ifndef STGCN_V6_WIDE
inline bool plan_v6(int C, int T, int V, int K, int terms, V6Plan &pl) {
    if (K != KT6 || C % 128 != 0 || V > 32) return false;    // (C % 32 == 0: an even number of 16-channel chunks)
    const int rows = v6_rows(T, V, K);
    if (rows == 0) return false;
    const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
    const size_t img = 2 * buf > (size_t)4 * EPI6 ? 2 * buf : (size_t)4 * EPI6;
    pl.lds = (size_t)C * W12P * 4 + RING6 + img + (size_t)rows * 64 + 12 * FRAG6;
    if (pl.lds > (size_t)kLdsBytes) return false;
    pl.rows = rows;
    pl.tiles_per_clip = ceil_div(T * V, NP6);
    return true;
}
#else
inline bool plan_v6(int C, int T, int V, int K, int terms, V6Plan &pl) {
    const int v0 = stem_wide_split(V);
    if (K != KT6 || C % 128 != 0 || v0 == 0) return false;
    const int r0 = v6_rows(T, v0, K), r1 = v6_rows(T, V - v0, K);
    if (r0 == 0 || r1 == 0) return false;
    const int rows = r0 > r1 ? r0 : r1;
    const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
    const size_t img = 2 * buf > (size_t)4 * EPI6 ? 2 * buf : (size_t)4 * EPI6;
    // the half's 24 fragments: inside the second image buffer with three terms (it must hold them), else behind Fs
    if (terms == 3 && buf < (size_t)24 * FRAG6) return false;
    pl.lds = (size_t)C * W12P * 4 + RING6 + img + (size_t)rows * 64 + (terms == 3 ? 0 : 24 * FRAG6);
    if (pl.lds > (size_t)kLdsBytes) return false;
    pl.rows = rows;
    pl.v0 = v0;
    pl.tpc1 = ceil_div(T * (V - v0), NP6);
    pl.tiles_per_clip = ceil_div(T * v0, NP6) + pl.tpc1;
    return pl.tiles_per_clip - pl.tpc1 >= pl.tpc1;           // (the interleaved tile order assumes it: V0 >= V - V0)
}
#endif

template <int TERMS>
int launch_v6(const uint4 *pf, const float *x, int xsc, int xsp, const float *W12, const uint4 *Wq, const float *shift, void *y,
              int N, int C, int T, int V, const V6Plan &pl, bool bf16out, int opt, int num_cu, hipStream_t st) {
    const int ntiles = N * pl.tiles_per_clip;
    const dim3 grid(ntiles < num_cu ? ntiles : num_cu, C / 128, 1);
    if (bf16out) {
        auto kern = stem_bf16_v6_kernel<TERMS, true, V6W>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT6), pl.lds, st, pf, x, xsc, xsp, W12, Wq, shift, y, C, T, V, pl.rows,
                           pl.tiles_per_clip, ntiles, opt, debug_buffer(), pl.v0, pl.tpc1);
    } else {
        auto kern = stem_bf16_v6_kernel<TERMS, false, V6W>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT6), pl.lds, st, pf, x, xsc, xsp, W12, Wq, shift, y, C, T, V, pl.rows,
                           pl.tiles_per_clip, ntiles, opt, debug_buffer(), pl.v0, pl.tpc1);
    }
    STGCN_LAUNCH_CHECK("stem_bf16_v6_kernel");
    return STGCN_OK;
}

}  // namespace

#ifndef STGCN_V6_WIDE
#define V6_SUPPORTED stem_v6_supported
#define V6_LAUNCH launch_stem_v6
#else
#define V6_SUPPORTED stem_v6w_supported
#define V6_LAUNCH launch_stem_v6w
#endif

bool V6_SUPPORTED(int C, int T, int V, int K, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16) return false;
    V6Plan pl;
    return T >= 1 && plan_v6(C, T, V, K, math == STGCN_MATH_BF16X3 ? 3 : 1, pl);
}

#ifndef STGCN_V6_WIDE
// temporal weights (Cout,Cin,9) * scale -> KF6's pair order (same size as the 32x32x16 packing)
int launch_tcn_pack_bf16_pairs(const float *W, const float *scale, void *Wq, int Cin, int Cout, hipStream_t st) {
    const size_t total = (size_t)Cin * Cout * KT6 * 2;
    hipLaunchKernelGGL(tcn_pack_bf16_pairs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, scale,
                       (unsigned short *)Wq, Cin, Cout);
    STGCN_LAUNCH_CHECK("tcn_pack_bf16_pairs_kernel");
    return STGCN_OK;
}
#endif

int V6_LAUNCH(const float *x, bool x_ntvc, const void *pfrag, const void *prep_w12, const void *Wq, const float *shift,
              void *out, int N, int C, int T, int V, int K, unsigned flags, hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const int terms = math == STGCN_MATH_BF16X3 ? 3 : 1;
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    const int opt = (flags & STGCN_OUT_NTVC) ? OPT_OUT_NTVC : 0;
    V6Plan pl;
    if (!plan_v6(C, T, V, K, terms, pl))
        return fail(STGCN_ERR_UNSUPPORTED, "stem v6 kernel does not cover C=%d T=%d V=%d K=%d", C, T, V, K);
    if ((size_t)3 * T * V * 4 >= ((size_t)1 << 31) || (size_t)T * V * C >= ((size_t)1 << 31))
        return fail(STGCN_ERR_UNSUPPORTED, "stem v6: clip of T=%d V=%d exceeds a buffer resource", T, V);
    int dev = 0, num_cu = 256;
    STGCN_HIP_CHECK(hipGetDevice(&dev));
    STGCN_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int xsc = x_ntvc ? 1 : T * V, xsp = x_ntvc ? 3 : 1;
    const uint4 *pf = (const uint4 *)pfrag;
    const float *W12 = (const float *)prep_w12;
    const uint4 *wq = (const uint4 *)Wq;
    return terms == 3 ? launch_v6<3>(pf, x, xsc, xsp, W12, wq, shift, out, N, C, T, V, pl, bf16out, opt, num_cu, st)
                      : launch_v6<1>(pf, x, xsc, xsp, W12, wq, shift, out, N, C, T, V, pl, bf16out, opt, num_cu, st);
}

}  // namespace stgcn
