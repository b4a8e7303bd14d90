// KF5 — the fused stem on the bf16 matrix cores with ONE WAVE PER SIMD (256 threads, 4 waves, up to 512 VGPRs each).
//
// Same tile, same LDS images, same weight ring, same in-kernel feature computation and producer as KF4's FK form
// (stem_bf16_v4.hip) — what changes is who owns what:
//   * a wave owns ALL 128 output channels of its 64-pixel quarter: 4 x 2 accumulator blocks (128 VGPRs).  A tap is then
//     24 consecutive MFMAs per wave for 12 ds_read_b128 (the four waves read every weight fragment once each instead of
//     eight waves reading half of them twice): LDS read traffic per MFMA is halved, and an activation fragment is reused
//     by four channel blocks instead of two.
//   * there is no second wave on the SIMD: the two co-resident waves of KF4 contended for the matrix pipe by age (the
//     older one ran ahead and then waited ~23 % of its time at the stage barriers), and every stage opened with both of
//     them waiting for their first fragment reads.  One wave per SIMD issues its MFMAs back to back; the other pipes'
//     work (fragment reads, producer, DMA issue) sits in the gaps of its own stream.
// The per-tile tail (epilogue, feature phase, chunk-0 production) is the same code with four waves.
#include <type_traits>

#include "bf16_common.h"

namespace stgcn {

namespace {

using namespace bf16k;

constexpr int NP5 = 256;   // output pixels per tile
constexpr int NT5 = 256;   // threads per workgroup: one wave per SIMD
constexpr int KT5 = 9;     // temporal taps
constexpr int STG5 = 3;    // taps per weight stage
constexpr int FRAG5 = 1024;
constexpr int STAGE5 = STG5 * 8 * FRAG5;
constexpr int EPI5 = 8192; // epilogue staging per wave: 32 channels x 64 pixels fp32

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void *lptr5_t;

// LDS-DMA through inline asm (see stem_bf16_v4.hip: the builtin makes hipcc wait for the DMA in front of the next ds_read)
__device__ __forceinline__ void dma16v5(const void *g, unsigned lds_addr) {
    // M0 = LDS destination (wave-uniform).  M0 is declared clobbered instead of saved and restored around every transfer:
    // nothing else in these kernels lives in M0, and the three extra scalar instructions per transfer are not free when a
    // single wave owns the SIMD (they sit in the MFMA stream).
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait5() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct FragA5 { uint4 hi[4], lo[4]; };     // weights of one tap: 4 channel blocks, hi (+ lo)
struct FragB5 { uint4 hi[2], lo[2]; };     // activations of one tap: 2 pixel blocks

// MFMA number I of a tap (I = 0 .. 8*TERMS-1): channel block I / (2*TERMS), pixel block (I / TERMS) % 2, term I % TERMS
// (terms of a block are consecutive: one accumulation chain issues back to back at full rate on gfx950)
template <int TERMS, int I>
__device__ __forceinline__ void mfma_one5(f32x16 (&acc)[4][2], const FragA5 &a, const FragB5 &b) {
    constexpr int m = I / (2 * TERMS), j = (I / TERMS) % 2, term = I % TERMS;
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi[m]), bh = __builtin_bit_cast(bf16x8, b.hi[j]);
    if constexpr (TERMS == 3) {
        const bf16x8 al = __builtin_bit_cast(bf16x8, a.lo[m]), bl = __builtin_bit_cast(bf16x8, b.lo[j]);
        if constexpr (term == 0) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m][j], 0, 0, 0);
        else if constexpr (term == 1) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m][j], 0, 0, 0);
        else acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m][j], 0, 0, 0);
    } else {
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m][j], 0, 0, 0);
    }
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <int I, int N, class F>
__device__ __forceinline__ void static_for5(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for5<I + 1, N>(f);
    }
}

template <int PB, int TERMS, bool BF16OUT>
__global__ __launch_bounds__(NT5) void stem_bf16_v5_kernel(
    const uint4 *__restrict__ pfrag, const float *__restrict__ x, int xsc, int xsp, const float *__restrict__ W12,
    const uint4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int C, int T, int V, int ROWS,
    int tiles_per_clip, int ntiles, int abl, unsigned long long *dbg) {
#ifdef STGCN_ABLATION  // in-kernel cycle stamps (diagnostic builds only; dbg == NULL otherwise)
#define V5_STAMP(var) unsigned long long var = 0; if (dbg) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); }
#define V5_ACC(slot, a, b) if (dbg) { tsum[slot] += (b) - (a); }
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
#define V5_STAMP(var)
#define V5_ACC(slot, a, b)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem5[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = pixel quarter of the tile
    const int TV = T * V;
    const int nch = C / CCB;                 // channel chunks (C = 128 -> 8)
    const int nstage = nch * (KT5 / STG5);   // weight stages per tile
    const int img_bytes = ROWS * PXB;
    const int buf_bytes = img_bytes * (TERMS == 3 ? 2 : 1);
    // LDS carve: W12 (bf16 hi/lo) | weight ring | images buf0, buf1 (= epilogue staging, 4 x 8 KiB) | Fs | Pf
    uint4 *W12q = reinterpret_cast<uint4 *>(smem5);
    char *ring = smem5 + C * W12P * 4;
    char *buf0 = ring + 2 * STAGE5;
    char *buf1 = buf0 + buf_bytes;
    uint4 *Fs = reinterpret_cast<uint4 *>(buf0 + max(2 * buf_bytes, 4 * EPI5));
    const uint4 *Pf = Fs + 4 * ROWS;
    const unsigned lds0 = (unsigned)(size_t)(lptr5_t)smem5;
    const unsigned ring_lds = lds0 + (unsigned)(ring - smem5);
    const unsigned pf_lds = lds0 + (unsigned)(reinterpret_cast<const char *>(Pf) - smem5);

    const int cg = blockIdx.y;               // 128-channel group of the output
    // the two weight fragments this wave DMAs each tap: channel block `wave`, images hi and lo
    const uint4 *wsrc = Wp + ((size_t)(cg * 4 + wave) * nch * KT5 * 2) * 64 + lane;
    auto dma_stage = [&](int gs) {           // stage gs (3 taps) -> ring slot gs & 1
        const unsigned dst = ring_lds + (gs & 1) * STAGE5 + wave * 2 * FRAG5;
        const int gsm = gs % nstage;         // weights repeat for every tile
#pragma unroll
        for (int t = 0; t < STG5; ++t) {
            dma16v5(wsrc + (size_t)(gsm * STG5 + t) * 128, dst + t * 8 * FRAG5);
            dma16v5(wsrc + (size_t)(gsm * STG5 + t) * 128 + 64, dst + t * 8 * FRAG5 + FRAG5);
        }
    };
    auto dma_pfrag = [&](int tile) {         // 12 KiB: the clip's attention fragments -> Pf
        const int n = tile / tiles_per_clip;
        const uint4 *src = pfrag + (size_t)n * 12 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) dma16v5(src + (wave + 4 * i) * 64, pf_lds + (wave + 4 * i) * FRAG5);
    };

    // ---- features of a tile from x and the clip's attention fragments (see stem_bf16_v4.hip, FK form) -------------
    struct XRegs { float xa[8]; float xp[3]; };
    auto load_x = [&](XRegs &xr, int tile, int u) {
        int ln = tid & 63;                   // opaque per call: keeps lane-only address terms from being hoisted and spilled
        asm volatile("" : "+v"(ln));
        const int mb = u >> 1, hh = u & 1;
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT5, 1, T, NP5);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (size_t)n * 3 * TV), 0, (unsigned)(3 * TV * 4), 0x00020000);
        const int tf = g.t_first - (KT5 - 1) / 2 + 4 * mb;
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
    };
    auto feature_unit = [&](const TileGeomB &g, int u, const XRegs &xr) {
        int ln = tid & 63;
        asm volatile("" : "+v"(ln));
        const int mb = u >> 1, hh = u & 1;
        uint4 xh, xl;
        split8(xr.xa, xh, xl);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, xh), al = __builtin_bit_cast(bf16x8, xl);
        f32x4 d[3];
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
        const int w = 16 * hh + (ln & 15);
        const int p = (4 * mb + (ln >> 4)) * V + w;          // pixel row of the tile
        const int gi = g.origin + p;
        const bool valid = p < g.span && gi >= 0 && gi < TV; // else: the temporal conv's zero padding
        const float one = valid ? 1.f : 0.f;
        const float fa[8] = {d[0][0] * one, d[0][1] * one, d[0][2] * one, d[1][0] * one,
                             d[1][1] * one, d[1][2] * one, d[2][0] * one, d[2][1] * one};
        const float fb[8] = {d[2][2] * one, xr.xp[0] * one, xr.xp[1] * one, xr.xp[2] * one, one, 0.f, 0.f, 0.f};
        uint4 ha, la, hb, lb;
        split8(fa, ha, la);
        split8(fb, hb, lb);
        if (w < V && p < ROWS) {
            Fs[p] = ha;
            Fs[(size_t)ROWS + p] = hb;
            Fs[(size_t)2 * ROWS + p] = la;
            Fs[(size_t)3 * ROWS + p] = lb;
        }
    };
    // units wave, wave+4, wave+8 arrive prefetched; any further ones (narrow frames only) are loaded here
    auto feature_phase = [&](int tile, const XRegs &x0, const XRegs &x1, const XRegs &x2) {
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT5, 1, T, NP5);
        const int need = min(ROWS, ((g.span + 15) >> 4) << 4);       // rows the producer will read
        const int nun = (((need + V - 1) / V + 3) >> 2) * 2;         // M-blocks x 2 joint halves
        const bool two = V > 16;
        for (int u = wave; u < nun; u += 4) {
            if (!two && (u & 1)) continue;
            if (u == wave) feature_unit(g, u, x0);
            else if (u == wave + 4) feature_unit(g, u, x1);
            else if (u == wave + 8) feature_unit(g, u, x2);
            else {
                XRegs xr;
                load_x(xr, tile, u);
                feature_unit(g, u, xr);
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
    auto prod_finish = [&](char *buf, const Prod &pr) {
        const float v0 = fmaxf(pr.d[0], 0.f), v1 = fmaxf(pr.d[1], 0.f), v2 = fmaxf(pr.d[2], 0.f), v3 = fmaxf(pr.d[3], 0.f);
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
    for (int e = tid; e < C * 2; e += NT5) {   // W12 -> bf16 hi/lo planes [hi k0-7][hi k8-15][lo k0-7][lo k8-15] of [C] x 16 B
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
        dma_stage(0);
        dma_wait5();
        __syncthreads();                      // W12q, Pf(tile), weight stage 0 landed
        if (tile < ntiles) feature_phase(tile, x0, x1, x2);
        __syncthreads();
    }

    int gs = 0;                               // running weight-stage counter (ring slot = gs & 1)
    int gnext = 1 % nstage;                   // index of stage gs + 1 within a tile's stages (kept without a division)
    const int h = lane >> 5;
    for (; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT5, 1, T, NP5);
        const int nblk = (g.span + 15) >> 4;
        const int next_tile = tile + gridDim.x;

        V5_STAMP(t_0)
        // chunk 0 of this tile
        for (int b = wave; b < nblk; b += 4) {
            Prod pr;
            prod_load(pr, 0, b);
            prod_mfma(pr);
            prod_finish(buf0, pr);
        }
        int prow[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int q = g.q0 + (wave * 2 + j) * 32 + (lane & 31);
            q = min(q, g.q_last);
            const int t = q / V, v = q - t * V;
            prow[j] = (t - g.t_first) * V + v;
        }
        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
        __syncthreads();                      // chunk 0 visible
        V5_STAMP(t_1)
        V5_ACC(0, t_0, t_1)

        // One tap of one wave = 8*TERMS MFMAs.  With a single wave on the SIMD nothing else covers a clump of LDS reads or
        // producer arithmetic between two MFMAs — the matrix pipe just drains — so the tap is written as SLOTS: one MFMA
        // followed by at most a couple of other instructions, fenced by sched_barrier(0) (hipcc otherwise gathers the
        // fillers into clumps of 30-40 instructions between MFMA groups; measured 67 % matrix-pipe occupancy in this loop
        // against 75 % for the eight-wave kernel, whose second wave hides such clumps).  Filler placement per tap:
        //   slots 0-2   producer operands (3 reads); at a stage start also this tap's own weight fragments, two per slot,
        //               each channel block one block ahead of its use (the first two are read in front of slot 0)
        //   slots 6-13  next tap's weight fragments, slots 14-17 next tap's activation fragments
        //   slot 11     the producer's two small MFMAs, slots 14-21 its ReLU / split arithmetic, slots 22-23 its stores
        //   slots 18-23 first tap of a stage: the six loads of the next weight stage; second tap: their LDS stores
        constexpr int NM = 8 * TERMS;
        unsigned boff[KT5][2];                // LDS offsets of this wave's activation fragments (tap, pixel block)
#pragma unroll
        for (int tap = 0; tap < KT5; ++tap)
#pragma unroll
            for (int j = 0; j < 2; ++j) boff[tap][j] = (unsigned)lds_off(prow[j] + tap * V, h);
        FragA5 a_cur = {}, a_nxt = {};
        FragB5 b_cur = {}, b_nxt = {};
        auto rd = [&](const char *p) { return *reinterpret_cast<const uint4 *>(p); };
        auto chunk = [&](auto last_c, int ch) {
            constexpr bool LAST = decltype(last_c)::value;
            const char *cur = (ch & 1) ? buf1 : buf0;
            char *nxt = (ch & 1) ? buf0 : buf1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {       // activation fragments of tap 0 (the chunk image is complete: last barrier)
                b_cur.hi[j] = rd(cur + boff[0][j]);
                if constexpr (TERMS == 3) b_cur.lo[j] = rd(cur + img_bytes + boff[0][j]);
            }
            static_for5<0, KT5 / STG5>([&](auto st_c) {
                constexpr int st = decltype(st_c)::value;
                const char *aslot = ring + (gs & 1) * STAGE5 + lane * 16;
                a_cur.hi[0] = rd(aslot);         // this stage's first weight fragments: published by the barrier just passed
                if constexpr (TERMS == 3) a_cur.lo[0] = rd(aslot + FRAG5);
                // Next weight stage -> the other ring slot THROUGH REGISTERS: six 16-byte loads per lane in the first tap
                // (from L2: each fragment is fetched once per workgroup), six ds_write_b128 in the second.  LDS-DMA costs
                // the issuing wave 100-185 cycles per 1-KiB piece inside a busy phase (MI355X_MICROARCH.md) — six pieces per
                // stage are 20 % of a tile when no second wave on the SIMD covers them; with 512 VGPRs per wave the 24
                // staging registers are free.
                uint4 wr0, wr1, wr2, wr3, wr4, wr5;   // (named scalars: an array captured by the nested lambdas went to scratch)
                const uint4 *wnext = wsrc + (size_t)(gnext * STG5) * 128;
                char *wdst = ring + ((gs + 1) & 1) * STAGE5 + wave * 2 * FRAG5 + lane * 16;
                static_for5<0, STG5>([&](auto tt_c) {
                    constexpr int tt = decltype(tt_c)::value;
                    constexpr int tap = st * STG5 + tt;
                    constexpr bool prod = tap < PB && !LAST;
                    Prod pr = {};
                    unsigned ph0 = 0, ph1 = 0;
                    float pv0 = 0.f, pv1 = 0.f, pv2 = 0.f, pv3 = 0.f;
                    int poff = 0;
                    // filler number v of the tap (24 of them; with TERMS == 1 three share a slot)
                    auto filler = [&](auto v_c) {
                        constexpr int v = decltype(v_c)::value;
                        if constexpr (prod && v == 0) {
                            pr.p = min(wave + 4 * tap, nblk - 1) * 16 + pl;
                            pr.wh = W12q[(size_t)(pg & 1) * C + (ch + 1) * CCB + pl];
                        }
                        if constexpr (prod && v == 1) pr.wl = W12q[(size_t)(2 + (pg & 1)) * C + (ch + 1) * CCB + pl];
                        if constexpr (prod && v == 2) pr.fb = Fs[(size_t)pg * ROWS + pr.p];
                        if constexpr (tt == 0 && v < 6) {              // own weight fragments of channel blocks 1-3
                            constexpr int m = 1 + v / 2;
                            if constexpr (v % 2 == 0) a_cur.hi[m] = rd(aslot + (m * 2) * FRAG5);
                            else if constexpr (TERMS == 3) a_cur.lo[m] = rd(aslot + (m * 2 + 1) * FRAG5);
                        }
                        if constexpr (tt + 1 < STG5 && v >= 6 && v < 14) {   // next tap's weight fragments
                            constexpr int m = (v - 6) / 2;
                            if constexpr (v % 2 == 0) a_nxt.hi[m] = rd(aslot + ((tt + 1) * 8 + m * 2) * FRAG5);
                            else if constexpr (TERMS == 3) a_nxt.lo[m] = rd(aslot + ((tt + 1) * 8 + m * 2 + 1) * FRAG5);
                        }
                        if constexpr (tap + 1 < KT5 && v >= 14 && v < 18) {  // next tap's activation fragments
                            constexpr int j = (v - 14) / 2;
                            if constexpr (v % 2 == 0) b_nxt.hi[j] = rd(cur + boff[tap + 1][j]);
                            else if constexpr (TERMS == 3) b_nxt.lo[j] = rd(cur + img_bytes + boff[tap + 1][j]);
                        }
                        if constexpr (prod && v == 11) prod_mfma(pr);
                        if constexpr (prod && v == 14) {
                            pv0 = fmaxf(pr.d[0], 0.f); pv1 = fmaxf(pr.d[1], 0.f); pv2 = fmaxf(pr.d[2], 0.f); pv3 = fmaxf(pr.d[3], 0.f);
                        }
                        if constexpr (prod && v == 15) { ph0 = pack_bf16x2(pv0, pv1); ph1 = pack_bf16x2(pv2, pv3); }
                        if constexpr (prod && v == 16) poff = lds_off(pr.p, pg >> 1) + (pg & 1) * 8;
                        if constexpr (prod && v == 17) *reinterpret_cast<uint2 *>(nxt + poff) = make_uint2(ph0, ph1);
                        if constexpr (prod && TERMS == 3 && v == 18) { pv0 -= bf16_lo_to_f32(ph0); pv1 -= bf16_hi_to_f32(ph0); }
                        if constexpr (prod && TERMS == 3 && v == 19) { pv2 -= bf16_lo_to_f32(ph1); pv3 -= bf16_hi_to_f32(ph1); }
                        if constexpr (prod && TERMS == 3 && v == 20) { ph0 = pack_bf16x2(pv0, pv1); ph1 = pack_bf16x2(pv2, pv3); }
                        if constexpr (prod && TERMS == 3 && v == 21) *reinterpret_cast<uint2 *>(nxt + img_bytes + poff) = make_uint2(ph0, ph1);
                        if constexpr (tt == 0 && v >= 18) {             // next weight stage: loads (tap t, image f)
                            constexpr int d = v - 18, t = d / 2, f = d % 2;
                            const uint4 wv = wnext[t * 128 + f * 64];
                            if constexpr (d == 0) wr0 = wv; else if constexpr (d == 1) wr1 = wv; else if constexpr (d == 2) wr2 = wv;
                            else if constexpr (d == 3) wr3 = wv; else if constexpr (d == 4) wr4 = wv; else wr5 = wv;
                        }
                        if constexpr (tt == 1 && v >= 18) {             // ... and their LDS stores, one tap later
                            constexpr int d = v - 18, t = d / 2, f = d % 2;
                            const uint4 wv = d == 0 ? wr0 : d == 1 ? wr1 : d == 2 ? wr2 : d == 3 ? wr3 : d == 4 ? wr4 : wr5;
                            *reinterpret_cast<uint4 *>(wdst + f * FRAG5 + t * 8 * FRAG5) = wv;
                        }
                    };
                    static_for5<0, NM>([&](auto i_c) {
                        constexpr int i = decltype(i_c)::value;
                        mfma_one5<TERMS, i>(acc, a_cur, b_cur);
                        static_for5<i * (24 / NM), (i + 1) * (24 / NM)>(filler);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    a_cur = a_nxt;
                    b_cur = b_nxt;
                });
                V5_STAMP(t_s1)
                if constexpr (LAST) dma_wait5();   // (the next tile's attention fragments travel by LDS-DMA)
                __syncthreads();              // stage done: next weights stored and visible; chunk boundary at st == 2
                V5_STAMP(t_s2)
                V5_ACC(2, t_s1, t_s2)
                ++gs;
                gnext = gnext + 1 == nstage ? 0 : gnext + 1;
            });
        };
        for (int ch = 0; ch + 1 < nch; ++ch) chunk(std::false_type{}, ch);
        if (next_tile < ntiles) dma_pfrag(next_tile);                  // Pf is idle after the tile's feature phase
        chunk(std::true_type{}, nch - 1);
        V5_STAMP(t_2)
        V5_ACC(1, t_1, t_2)

        // ---- epilogue: each 32-channel x 64-pixel block through this wave's 8 KiB staging slice, 16 B per lane ------
        // Store addresses = a per-(tile, block, store) SCALAR base + one per-lane term that never changes: a full tile
        // (15 of 16) takes the branch-free path where a store costs no vector arithmetic at all; the last tile of a clip
        // keeps the per-lane bounds checks.  (With 64-bit per-lane index arithmetic and a predicate per store the tail of
        // a tile executed ~3,800 vector instructions per wave — ~15 % of the tile with no second wave to hide it.)
        XRegs xn0, xn1, xn2;                  // next tile's x: in flight while this tile's results are stored
        load_x(xn0, min(next_tile, ntiles - 1), wave);
        load_x(xn1, min(next_tile, ntiles - 1), wave + 4);
        load_x(xn2, min(next_tile, ntiles - 1), wave + 8);
        __builtin_amdgcn_sched_barrier(0);
        float *stg = reinterpret_cast<float *>(buf0 + wave * EPI5);
        const int qw = g.q0 + wave * 64;
        const bool full = g.q0 + NP5 - 1 <= g.q_last;            // (scalar) every pixel of the tile lies inside the clip
        if (abl & OPT_OUT_NTVC) {
            const int hh = lane >> 5;
            // element offset of (pixel px = idx>>3, 4-channel slot sl = idx&7) for idx = it*64 + lane: (it*8 + lane>>3)*C + 4*(lane&7)
            const unsigned lterm = (unsigned)((lane >> 3) * C + 4 * (lane & 7));
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ob = cg * 128 + m * 32;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 8 * gq + 4 * hh);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int px = j * 32 + (lane & 31);
                        const float4 v = make_float4(fmaxf(acc[m][j][4 * gq + 0] + sh4.x, 0.f), fmaxf(acc[m][j][4 * gq + 1] + sh4.y, 0.f),
                                                     fmaxf(acc[m][j][4 * gq + 2] + sh4.z, 0.f), fmaxf(acc[m][j][4 * gq + 3] + sh4.w, 0.f));
                        *reinterpret_cast<float4 *>(stg + px * 32 + (((2 * gq + hh) ^ (px & 7)) << 2)) = v;
                    }
                }
                const size_t tbase = ((size_t)n * TV + qw) * C + ob;      // scalar
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int idx = it * 64 + lane, px = idx >> 3, sl = idx & 7;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + px * 32 + ((sl ^ (px & 7)) << 2));
                    if (full || qw + px <= g.q_last) {
                        if constexpr (BF16OUT) {
                            unsigned short *yb = reinterpret_cast<unsigned short *>(y) + tbase + (size_t)(it * 8) * C;
                            *reinterpret_cast<uint2 *>(yb + lterm) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                        } else {
                            float *yb = reinterpret_cast<float *>(y) + tbase + (size_t)(it * 8) * C;
                            *reinterpret_cast<float4 *>(yb + lterm) = v;
                        }
                    }
                }
            }
        } else {
            // element offset of (row = idx>>4, 4-pixel group c4 = 4*(idx&15)) for idx = it*64 + lane: (it*4 + lane>>4)*TV + 4*(lane&15)
            const unsigned lterm = (unsigned)((lane >> 4) * TV + 4 * (lane & 15));
            const int c4l = 4 * (lane & 15);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ob = cg * 128 + m * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float sh = shift[ob + cr];
#pragma unroll
                    for (int j = 0; j < 2; ++j) stg[cr * 64 + j * 32 + (lane & 31)] = fmaxf(acc[m][j][r] + sh, 0.f);
                }
                const size_t tbase = ((size_t)n * C + ob) * TV + qw;      // scalar
                const bool al16 = ((tbase & 3) == 0) && (TV % 4 == 0);    // 16-byte (8-byte for bf16) aligned rows
#pragma unroll
                for (int it = 0; it < 8; ++it) {
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
        V5_STAMP(t_3)
        V5_ACC(3, t_2, t_3)
        if (next_tile < ntiles) {             // its fragments landed at the last stage barrier, its x during the stores;
            feature_phase(next_tile, xn0, xn1, xn2);   // Fs lies behind the staging area: no barrier needed in front
            V5_STAMP(t_4)
            V5_ACC(5, t_3, t_4)
            __syncthreads();                  // Fs complete, every wave's staging reads done (chunk 0 overwrites buf0)
        }
        V5_STAMP(t_5)
        V5_ACC(4, t_3, t_5)
    }
#ifdef STGCN_ABLATION
    if (dbg && lane == 0 && blockIdx.x < 8 && blockIdx.y == 0)
        for (int i = 0; i < 8; ++i) dbg[(blockIdx.x * 8 + wave) * 8 + i] = tsum[i];
#endif
}

struct V5Plan {
    int pb = 0, rows = 0, tiles_per_clip = 0;
    size_t lds = 0;
};

inline bool plan_v5(int C, int T, int V, int K, int terms, V5Plan &pl) {
    if (K != KT5 || C % 128 != 0 || V > 32) return false;
    int dt = ceil_div(NP5 - 1, V);
    if (dt > T - 1) dt = T - 1;
    const int span = (dt + K) * V;
    const int rows = (span + 15) / 16 * 16;
    const int pb = ceil_div(ceil_div(span, 16), 4);  // producer blocks per wave per chunk (4 producing waves)
    if (pb > KT5) return false;
    const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
    const size_t img = 2 * buf > (size_t)4 * EPI5 ? 2 * buf : (size_t)4 * EPI5;
    pl.lds = (size_t)C * W12P * 4 + 2 * STAGE5 + img + (size_t)rows * 64 + 12 * FRAG5;
    if (pl.lds > (size_t)kLdsBytes) return false;
    pl.pb = pb;
    pl.rows = rows;
    pl.tiles_per_clip = ceil_div(T * V, NP5);
    return true;
}

template <int PB, int TERMS>
int launch_v5(const uint4 *pf, const float *x, int xsc, int xsp, const float *W12, const uint4 *Wp, const float *shift, void *y,
              int N, int C, int T, int V, const V5Plan &pl, bool bf16out, int opt, int num_cu, hipStream_t st) {
    const int ntiles = N * pl.tiles_per_clip;
    const dim3 grid(ntiles < num_cu ? ntiles : num_cu, C / 128, 1);
    if (bf16out) {
        auto kern = stem_bf16_v5_kernel<PB, TERMS, true>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT5), pl.lds, st, pf, x, xsc, xsp, W12, Wp, shift, y, C, T, V, pl.rows,
                           pl.tiles_per_clip, ntiles, opt, debug_buffer());
    } else {
        auto kern = stem_bf16_v5_kernel<PB, TERMS, false>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT5), pl.lds, st, pf, x, xsc, xsp, W12, Wp, shift, y, C, T, V, pl.rows,
                           pl.tiles_per_clip, ntiles, opt, debug_buffer());
    }
    STGCN_LAUNCH_CHECK("stem_bf16_v5_kernel");
    return STGCN_OK;
}

}  // namespace

bool stem_v5_supported(int C, int T, int V, int K, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16) return false;
    V5Plan pl;
    return T >= 1 && plan_v5(C, T, V, K, math == STGCN_MATH_BF16X3 ? 3 : 1, pl);
}

int launch_stem_v5(const float *x, bool x_ntvc, const void *pfrag, const void *prep_w12, const void *Wp, const float *shift,
                   void *out, int N, int C, int T, int V, int K, unsigned flags, hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const int terms = math == STGCN_MATH_BF16X3 ? 3 : 1;
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    const int opt = (flags & STGCN_OUT_NTVC) ? OPT_OUT_NTVC : 0;
    V5Plan pl;
    if (!plan_v5(C, T, V, K, terms, pl))
        return fail(STGCN_ERR_UNSUPPORTED, "stem v5 kernel does not cover C=%d T=%d V=%d K=%d", C, T, V, K);
    if ((size_t)3 * T * V * 4 >= ((size_t)1 << 31))
        return fail(STGCN_ERR_UNSUPPORTED, "stem v5: clip of T=%d V=%d exceeds a buffer resource", T, V);
    int dev = 0, num_cu = 256;
    STGCN_HIP_CHECK(hipGetDevice(&dev));
    STGCN_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int xsc = x_ntvc ? 1 : T * V, xsp = x_ntvc ? 3 : 1;
    const uint4 *pf = (const uint4 *)pfrag;
    const float *W12 = (const float *)prep_w12;
    const uint4 *wp = (const uint4 *)Wp;
#define GO5(PB)                                                                                                              \
    return terms == 3 ? launch_v5<PB, 3>(pf, x, xsc, xsp, W12, wp, shift, out, N, C, T, V, pl, bf16out, opt, num_cu, st)     \
                      : launch_v5<PB, 1>(pf, x, xsc, xsp, W12, wp, shift, out, N, C, T, V, pl, bf16out, opt, num_cu, st)
    if (pl.pb <= 8) GO5(8);
    GO5(9);
#undef GO5
}

}  // namespace stgcn
