// KF4 — fused stem on the bf16 matrix cores, large-tile persistent form (STGCN_MATH_BF16X3 / _BF16, K = 9).
//
// Differences from the 128-pixel kernels in tcn_bf16.hip, each aimed at a measured loss of those:
//   * tile = 128 channels x 256 pixels, 512 threads = 8 waves as 2 (channel halves) x 4 (pixel quarters), one
//     workgroup per CU: the halo (8V pixels) costs 1.8x instead of 2.6x producer work and the weight
//     fragments are shared by 4 waves.
//   * weights reach the waves through an LDS ring filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
//     wave-instruction = exactly one packed MFMA fragment, lane-linear): 8 fragments per tap, one per
//     wave, staged 3 taps (one "stage") ahead into the other half of a 2-stage ring; a stage ends with
//     vmcnt(0) + barrier.  No weight traffic through VGPRs/L1 (it saturated the vector L1 before).
//   * the 12 graph-conv features per pixel are NOT recomputed per tile: the attention kernel emits them
//     once per clip, already split into bf16 hi/lo (feat[n][pixel] = 64 B: [hi f0-7][hi f8-15][lo f0-7]
//     [lo f8-15]) and the tile's rows are DMA'd into the feature tile Fs; rows outside the clip are zeroed
//     (the temporal conv's zero padding).
//   * persistent: grid = one workgroup per CU looping over tiles; the next tile's feature rows are DMA'd
//     during the last channel chunk of the current tile (Fs is idle then), so a tile's serial part is
//     only its epilogue stores + the production of channel chunk 0.
//
//   * the producer relu(W12 . Fs^T) runs on the bf16 matrix cores as well (two v_mfma_f32_16x16x32_bf16 per
//     16x16 block, all four hi/lo products).
//
//   * FK form (NJ = 2, the 256-pixel tile): the feature tile is not read from HBM at all.  The attention kernel leaves the
//     clip's three attention matrices as bf16 hi/lo MFMA B fragments (12 KiB per clip); at the start of a tile waves
//     compute u_s[k][t][w] = sum_v x[k][t][v] P_s[v][w] for the tile's frames as (4 frames x 4 channels) x 32 x 16
//     products (v_mfma_f32_16x16x32_bf16, all four hi/lo terms): with rows ordered (frame, channel) and one
//     accumulator per subset, ONE lane ends up holding all nine u values of its pixel (t,w) — no shuffles — adds the
//     three x values, splits to bf16 hi/lo and writes the pixel's 64-byte row into Fs.  x is read straight from the
//     caller's tensor (either layout) by buffer loads issued at the start of the previous tile's epilogue; the
//     fragments arrive by LDS-DMA during its last channel chunk.  The 64-B-per-pixel feature tensor (253 KiB per clip
//     written by the attention kernel and read 1.8x here) is gone; wide frames (NJ = 1) keep it.
//
// Everything else (LDS image layout, k-step = 16 channels of one tap, 2x2 MFMA blocks per wave, hi/lo split of the
// produced activations) is as described in tcn_bf16.hip.
#include "bf16_common.h"

namespace stgcn {

namespace {

using namespace bf16k;

constexpr int NP4 = 256;   // output pixels per tile (NJ = 2; the narrow form of the fused kernel uses 128)
constexpr int NT4 = 512;   // threads per workgroup
constexpr int KT4 = 9;     // temporal taps (the only kernel size this kernel is built for)
constexpr int STG = 3;     // taps per weight stage
constexpr int FRAG = 1024; // bytes of one packed MFMA weight fragment (64 lanes x 16 B)
constexpr int STAGE_BYTES = STG * 8 * FRAG;
constexpr int EPI_BYTES = 8192;  // epilogue staging per wave: 32 channels x 64 pixels fp32

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// LDS-DMA: 64 lanes x 16 B -> LDS at l (wave-uniform) + lane*16.  Issued through inline asm on purpose: with the
// builtin hipcc (ROCm 7.2) treats the DMA as a pending LDS write that any later ds_read may alias and puts
// `s_waitcnt vmcnt(0)` in front of the very next ds_read — the whole DMA latency exposed every stage (measured
// ~270 cycles per DMA instruction).  An asm DMA is invisible to that bookkeeping; dma_wait() before the stage
// barrier is then OUR job (vmcnt is in order, so the compiler's own counted waits only get stricter).
__device__ __forceinline__ void dma16(const void *g, unsigned lds_addr) {  // lds_addr: LDS byte address, wave-uniform
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// NJ = 32-pixel blocks per wave: 2 -> 256-pixel tile (the form described above); 1 -> 128-pixel tile for wide frames
// (V = 46: the 256-pixel tile's rows do not fit LDS).  With NJ = 1 the folded graph-conv matrix is not copied to LDS
// (the producer reads its rows through L1 and splits them in registers), which is what makes the tile fit.
template <int PB, int TERMS, bool BF16OUT, int NJ>
__global__ __launch_bounds__(NT4) void stem_bf16_v4_kernel(
    const float4 *__restrict__ feat, const float *__restrict__ x, int xsc, int xsp, const float *__restrict__ W12,
    const uint4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int C, int T, int V, int ROWS,
    int tiles_per_clip, int ntiles, int abl, unsigned long long *dbg) {
#ifdef STGCN_ABLATION  // in-kernel cycle stamps (diagnostic builds only; dbg == NULL otherwise)
#define STGCN_STAMP(var) unsigned long long var = 0; if (dbg) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); }
#define STGCN_ACC(slot, a, b) if (dbg) { tsum[slot] += (b) - (a); }
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#else
#define STGCN_STAMP(var)
#define STGCN_ACC(slot, a, b)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    constexpr int NPX = 128 * NJ;            // output pixels per tile
    constexpr bool W12LDS = NJ == 2;
    constexpr bool FK = NJ == 2;             // features computed in the kernel (`feat` = attention fragments)
    const int TV = T * V;
    const int nch = C / CCB;                 // channel chunks (C = 128 -> 8)
    const int nstage = nch * (KT4 / STG);    // weight stages per tile
    const int img_bytes = ROWS * PXB;
    const int buf_bytes = img_bytes * (TERMS == 3 ? 2 : 1);
    // LDS carve
    uint4 *W12q = reinterpret_cast<uint4 *>(smem4);                 // folded graph conv as bf16 hi/lo: 4 planes of [C] x 16 B
    char *ring = smem4 + (W12LDS ? C * W12P * 4 : 0);               // 2 stages x 3 taps x 8 fragments
    char *buf0 = ring + 2 * STAGE_BYTES;
    char *buf1 = buf0 + buf_bytes;
    // (the image buffers double as the epilogue's staging area, 8 KiB per wave; the feature tile is prefetched for
    //  the next tile while that epilogue runs, so it starts behind BOTH)
    // FK: Fs is rebuilt AFTER the epilogue (behind a barrier), so the staging area may run into it; the fragments Pf,
    // prefetched during the last chunk, sit behind everything the epilogue touches.
    uint4 *Fs = reinterpret_cast<uint4 *>(buf0 + (FK ? 2 * buf_bytes : max(2 * buf_bytes, 8 * EPI_BYTES * NJ / 2)));  // features as bf16 hi/lo: 4 planes of [ROWS] x 16 B
    const uint4 *Pf = reinterpret_cast<const uint4 *>(buf0 + max(2 * buf_bytes + 4 * ROWS * 16, 8 * EPI_BYTES * NJ / 2));   // FK: 12 fragments
    // LDS byte addresses for the DMA destinations (M0), derived from the array base by plain arithmetic
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem4;
    const unsigned ring_lds = lds0 + (unsigned)(ring - smem4);
    const unsigned fs_lds = lds0 + (unsigned)(reinterpret_cast<char *>(Fs) - smem4);
    const unsigned pf_lds = lds0 + (unsigned)(reinterpret_cast<const char *>(Pf) - smem4);

    const int cg = blockIdx.y;               // 128-channel group of the output
    // weight fragment this wave DMAs each tap: f = wave -> m-block (f>>1), image (f&1)
    const uint4 *wsrc = Wp + ((size_t)(cg * 4 + (wave >> 1)) * nch * KT4 * 2 + (wave & 1)) * 64 + lane;
    auto dma_stage = [&](int gs) {           // stage gs (taps 3gs..3gs+2 of the flat k index) -> ring slot gs&1
        const unsigned dst = ring_lds + (gs & 1) * STAGE_BYTES + wave * FRAG;
        const int gsm = gs % nstage;         // weights repeat for every tile
        if (STGCN_ABL(64)) return;
#pragma unroll
        for (int t = 0; t < STG; ++t) dma16(wsrc + (size_t)(gsm * STG + t) * 128, dst + t * 8 * FRAG);
    };
    // feature rows of tile `tile` -> Fs (rows whose pixel lies outside the clip are fixed up afterwards)
    auto dma_features = [&](int tile) {
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT4, 1, T, NPX);
        const float4 *src = feat + (size_t)n * TV * 4;
        for (int c = wave * 64; c < 4 * ROWS; c += 8 * 64) {       // chunk c = plane q, rows j0 .. j0+63
            const int q = c / ROWS, j = c - q * ROWS + lane;       // ROWS % 64 == 0
            int gi = g.origin + j;
            gi = max(0, min(gi, TV - 1));
            dma16(src + (size_t)gi * 4 + q, fs_lds + (unsigned)(q * ROWS + (c - q * ROWS)) * 16u);
        }
    };
    auto zero_invalid_rows = [&](int tile) {
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT4, 1, T, NPX);
        for (int j = tid; j < ROWS; j += NT4) {
            const int gi = g.origin + j;
            if (j >= g.span || gi < 0 || gi >= TV) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Fs[(size_t)q * ROWS + j] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };

    // ---- FK: features of a tile from x and the clip's attention fragments ------------------------------------
    // Work unit u = (M-block mb = u>>1, joint half hh = u&1): M-block = tile frames 4mb .. 4mb+3 (frame 0 = the first
    // halo frame), half = joint columns 16hh .. 16hh+15.  As A operand lane l holds row (frame (l&15)>>2, channel l&3)
    // and joints 8*(l>>4) .. +7; as accumulator it holds frame l>>4, joint column l&15.  Wave w owns units w, w+8, ...:
    // with 21 frames x 22 joints that is 12 units, two for waves 0-3 and one for waves 4-7 — three per SIMD.
    struct XRegs { float xa[8]; float xp[3]; };
    auto dma_pfrag = [&](int tile) {         // 12 KiB: the clip's fragments -> Pf
        const int n = tile / tiles_per_clip;
        const uint4 *src = reinterpret_cast<const uint4 *>(feat) + (size_t)n * 12 * 64 + lane;
        for (int f = wave; f < 12; f += 8) dma16(src + f * 64, pf_lds + f * FRAG);
    };
    auto load_x = [&](XRegs &xr, int tile, int u) {
        int lane = tid & 63;                 // opaque per call: keeps the lane-only address terms out of scratch (see epilogue)
        asm volatile("" : "+v"(lane));
        const int mb = u >> 1, hh = u & 1;
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT4, 1, T, NPX);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (size_t)n * 3 * TV), 0, (unsigned)(3 * TV * 4), 0x00020000);
        const int tf = g.t_first - (KT4 - 1) / 2 + 4 * mb;
        {
            const int k = lane & 3, t = tf + ((lane & 15) >> 2), v0 = 8 * (lane >> 4);
            const bool okr = k < 3 && t >= 0 && t < T;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned off = (okr && v0 + j < V) ? (unsigned)((k * xsc + (t * V + v0 + j) * xsp) * 4) : 0x7ffffff0u;
                xr.xa[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            }
        }
        {
            const int t = tf + (lane >> 4), w = 16 * hh + (lane & 15);
            const bool ok = t >= 0 && t < T && w < V;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned off = ok ? (unsigned)((k * xsc + (t * V + w) * xsp) * 4) : 0x7ffffff0u;
                xr.xp[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            }
        }
    };
    auto feature_unit = [&](const TileGeomB &g, int u, const XRegs &xr) {
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int mb = u >> 1, hh = u & 1;
        uint4 xh, xl;
        split8(xr.xa, xh, xl);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, xh), al = __builtin_bit_cast(bf16x8, xl);
        f32x4 d[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, Pf[((s * 2 + hh) * 2 + 0) * 64 + lane]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, Pf[((s * 2 + hh) * 2 + 1) * 64 + lane]);
            d[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, d[s], 0, 0, 0);
            d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d[s], 0, 0, 0);
            d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d[s], 0, 0, 0);
            d[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d[s], 0, 0, 0);
        }
        const int w = 16 * hh + (lane & 15);
        const int p = (4 * mb + (lane >> 4)) * V + w;        // pixel row of the tile
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
    // units w and w+8 arrive prefetched (xa / xb); any further ones (narrow frames only) are loaded here
    auto feature_phase = [&](int tile, const XRegs &xa, const XRegs &xb) {
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT4, 1, T, NPX);
        const int need = min(ROWS, ((g.span + 15) >> 4) << 4);       // rows the producer will read
        const int nun = (((need + V - 1) / V + 3) >> 2) * 2;         // M-blocks x 2 halves
        const bool two = V > 16;
        for (int u = wave; u < nun; u += 8) {
            if (!two && (u & 1)) continue;
            if (u == wave) feature_unit(g, u, xa);
            else if (u == wave + 8) feature_unit(g, u, xb);
            else {
                XRegs xr;
                load_x(xr, tile, u);
                feature_unit(g, u, xr);
            }
        }
    };

    // ---- producer: one 16-pixel block of chunk `ch` -> hi/lo images of `buf` -------------------
    // relu(W12 . f) for 16 channels x 16 pixels as TWO v_mfma_f32_16x16x32_bf16: both operands arrive split into bf16
    // hi/lo (the features by the attention kernel, W12 at setup) and the 32-deep k axis carries two 16-feature terms,
    //   MFMA 1:  A = [w_hi | w_hi],  B = [f_hi | f_lo]      MFMA 2:  A = [w_lo | w_lo],  same B
    // = all four hi/lo products (error ~2^-17 per product, the same order as the hi/lo split of the result that
    // follows).  32 matrix-pipe cycles per block against 128 for the four v_mfma_f32_16x16x4_f32 used before.
    // Then ReLU, hi/lo split and two 8-byte LDS stores.  (Spreading the MFMAs between the consumer MFMAs and
    // finishing a tap later was measured 3 % slower: more live registers, same pipe time.)
    const int pl = lane & 15, pg = lane >> 4;
    struct Prod { uint4 wh, wl, fb; f32x4 d; int p; };
    auto prod_load = [&](Prod &pr, int ch, int bi) {
        pr.p = bi * 16 + pl;
        if constexpr (W12LDS) {
            pr.wh = W12q[(size_t)(pg & 1) * C + ch * CCB + pl];
            pr.wl = W12q[(size_t)(2 + (pg & 1)) * C + ch * CCB + pl];
        } else {
            const float4 *wr = reinterpret_cast<const float4 *>(W12 + (size_t)(ch * CCB + pl) * W12P + (pg & 1) * 8);
            const float4 w0 = wr[0], w1 = wr[1];
            const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            split8(w8, pr.wh, pr.wl);
        }
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
    auto produce_block = [&](char *buf, int ch, int bi) {  // un-pipelined form (chunk 0 of a tile)
        Prod pr;
        prod_load(pr, ch, bi);
        prod_mfma(pr);
        prod_finish(buf, pr);
    };
    // ---- one-time setup ----------------------------------------------------------------------
    for (int e = tid; e < (W12LDS ? C * 2 : 0); e += NT4) {  // W12 -> bf16 hi/lo, planes [hi k0-7][hi k8-15][lo k0-7][lo k8-15] of [C] x 16 B
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
        XRegs x0 = {}, x1 = {};               // (block scope: nothing of it stays live into the tile loop)
        if constexpr (FK) {
            if (tile < ntiles) { dma_pfrag(tile); load_x(x0, tile, wave); load_x(x1, tile, wave + 8); }
        } else {
            if (tile < ntiles) dma_features(tile);
        }
        dma_stage(0);
        dma_wait();
        __syncthreads();                      // W12q, Fs(tile) / Pf(tile), weight stage 0 landed
        if (tile < ntiles) {
            if constexpr (FK) feature_phase(tile, x0, x1);
            else zero_invalid_rows(tile);
        }
        __syncthreads();
    }

#ifdef STGCN_ABLATION
    if (STGCN_ABL(32))  // experiment: de-phase the CUs so their epilogue store bursts do not coincide
        for (int i = 0; i < (int)(blockIdx.x & 7) * (abl >> 8); ++i) __builtin_amdgcn_s_sleep(16);
#endif
#ifdef STGCN_ABLATION
    if (STGCN_ABL(128) && wave >= 4) __builtin_amdgcn_s_setprio(1);  // experiment: static priority for the younger half
#endif
    int gs = 0;                               // running weight-stage counter (ring slot = gs & 1)
    const int h = lane >> 5;
    for (; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT4, 1, T, NPX);
        const int nblk = (g.span + 15) >> 4;
        const int next_tile = tile + gridDim.x;

        STGCN_STAMP(t_tile0)
        // chunk 0 of this tile
        for (int b = wave; b < nblk; b += 8) produce_block(buf0, 0, b);

        int prow[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int q = g.q0 + (wn * NJ + j) * 32 + (lane & 31);
            q = min(q, g.q_last);
            const int t = q / V, v = q - t * V;
            prow[j] = (t - g.t_first) * V + v;
        }
        f32x16 acc[2][NJ];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
        __syncthreads();                      // chunk 0 visible
        STGCN_STAMP(t_tile1)
        STGCN_ACC(0, t_tile0, t_tile1)

        // operand fetch helpers (LDS -> registers)
        auto load_a = [&](Frag2<TERMS> &a, const char *aslot, int tt) {
            if (STGCN_ABL(16)) return;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                a.hi[m] = *reinterpret_cast<const uint4 *>(aslot + (tt * 8 + m * 2) * FRAG);
                if constexpr (TERMS == 3) a.lo[m] = *reinterpret_cast<const uint4 *>(aslot + (tt * 8 + m * 2 + 1) * FRAG);
            }
        };
        auto load_b = [&](FragB<TERMS, NJ> &b, const char *img, int tap) {
            if (STGCN_ABL(8)) return;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int off = lds_off(prow[j] + tap * V, h);
                b.hi[j] = *reinterpret_cast<const uint4 *>(img + off);
                if constexpr (TERMS == 3) b.lo[j] = *reinterpret_cast<const uint4 *>(img + img_bytes + off);
            }
        };

        // Software pipeline: the fragments of tap t+1 are fetched before the MFMAs of tap t are issued.  The activation
        // fragments also cross the stage barriers (the chunk image is stable for the whole chunk); the weight
        // fragments of a new stage can only be read after the barrier that publishes its DMA.
        // (Measured neutral-to-negative, kept out: running waves 4-7 one tap out of phase with their SIMD partners
        //  0-3 — stage barrier in front of the last tap's MFMAs — 2 % slower; the same for all 8 waves 2 % slower;
        //  other producer/DMA taps for waves 4-7 and a static s_setprio for that half: neutral.  The kernel is
        //  clock-limited under load — tools/power_probe.py: 17 % faster on zero operands, same instruction stream —
        //  so removed stalls come back partly as a lower clock.)
        Frag2<TERMS> a_cur = {}, a_nxt = {};
        FragB<TERMS, NJ> b_cur = {}, b_nxt = {};
        for (int ch = 0; ch < nch; ++ch) {
            const char *cur = (ch & 1) ? buf1 : buf0;
            char *nxt = (ch & 1) ? buf0 : buf1;
            const bool last = ch + 1 == nch;
            if (last && next_tile < ntiles) {                          // Fs / Pf are idle during the last chunk
                if constexpr (FK) dma_pfrag(next_tile);
                else dma_features(next_tile);
            }
            load_b(b_cur, cur, 0);
#pragma unroll
            for (int st = 0; st < KT4 / STG; ++st, ++gs) {
                STGCN_STAMP(t_s0)
                const char *aslot = ring + (gs & 1) * STAGE_BYTES + (wm * 4) * FRAG + lane * 16;
                load_a(a_cur, aslot, 0);
#pragma unroll
                for (int tt = 0; tt < STG; ++tt) {
                    const int tap = st * STG + tt;
                    // Hard scheduling fences (sched_barrier(0)): left to itself hipcc sinks the fragment reads of
                    // tap t+1 below the MFMAs of tap t and waits for them right in front of their first use, and
                    // does the same with the producer's operands.  Phases of a tap:
                    //   reads (producer operands first, then the next tap's fragments) | 6 MFMAs | producer MFMAs +
                    //   weight DMA | 6 MFMAs | producer ReLU / split / stores
                    const bool prod = tap < PB && !last && !STGCN_ABL(1);
                    Prod pr = {};
                    if (prod) prod_load(pr, ch + 1, min(wave + 8 * tap, nblk - 1));
                    if (tt + 1 < STG) load_a(a_nxt, aslot, tt + 1);
                    if (tap + 1 < KT4) load_b(b_nxt, cur, tap + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!STGCN_ABL(2)) mfma_half_bf16<TERMS, NJ>(acc, a_cur, b_cur, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (prod) prod_mfma(pr);
                    // next weight stage -> other ring slot (its readers passed the last barrier); issued behind the
                    // first MFMAs so the DMA's issue cost does not delay the start of the stage
                    if (tt == 0) dma_stage(gs + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!STGCN_ABL(2)) mfma_half_bf16<TERMS, NJ>(acc, a_cur, b_cur, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (prod) prod_finish(nxt, pr);
                    a_cur = a_nxt;
                    b_cur = b_nxt;
                }
                STGCN_STAMP(t_s1)
                dma_wait();
                __syncthreads();              // stage done: next weights landed (vmcnt 0) and visible; chunk boundary at st==2
                STGCN_STAMP(t_s2)
                STGCN_ACC(1, t_s0, t_s1)
                STGCN_ACC(2, t_s1, t_s2)
            }
        }

        // epilogue: D[row = channel][col = pixel], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
        // Per-lane dword stores are store-issue-bound (measured ~7 B/clk/CU): transpose each 32-channel x 64-pixel
        // block through this wave's 8 KiB slice of the (now idle) image buffers and store 16 B per lane, so one
        // wave-instruction writes four 256-B channel rows.
        STGCN_STAMP(t_e0)
        // Opaque copies of the tile's scalars: everything the epilogue derives from them (store addresses: 64-bit
        // multiplies per lane) is then computed here.  Left visible, hipcc hoists that arithmetic above the channel loop
        // and spills it across the loop; a scratch reload in front of the next tile's feature phase then waits
        // (vmcnt(0)) for every store of this epilogue.
        int n_e = n, q0_e = g.q0, qlast_e = g.q_last, lane_e = lane;   // (lane too: lane-only terms are kernel invariants)
        asm volatile("" : "+s"(n_e), "+s"(q0_e), "+s"(qlast_e), "+v"(lane_e));
        XRegs xnext, xnext2;                  // (declared per tile and always fully written: dead across the loop back-edge)
        if constexpr (FK) {                   // next tile's x: in flight while this tile's results are stored
            load_x(xnext, min(next_tile, ntiles - 1), wave);
            load_x(xnext2, min(next_tile, ntiles - 1), wave + 8);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!STGCN_ABL(4) && (abl & OPT_OUT_NTVC)) {
            // (N,T,V,C) output: the block is staged pixel-major ([32*NJ pixels][32 channels], 16-byte slots XOR-swizzled by
            // the pixel so the b128 accesses are conflict-free); a wave-instruction then writes 8 pixels x 128 B.
            float *stg = reinterpret_cast<float *>(buf0 + wave * (EPI_BYTES * NJ / 2));   // 32 ch x 32*NJ px per wave
            const int qw = q0_e + wn * 32 * NJ;
            const int hh = lane_e >> 5;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int ob = cg * 128 + (wm * 2 + m) * 32;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 8 * gq + 4 * hh);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int px = j * 32 + (lane_e & 31);
                        const float4 v = make_float4(fmaxf(acc[m][j][4 * gq + 0] + sh4.x, 0.f), fmaxf(acc[m][j][4 * gq + 1] + sh4.y, 0.f),
                                                     fmaxf(acc[m][j][4 * gq + 2] + sh4.z, 0.f), fmaxf(acc[m][j][4 * gq + 3] + sh4.w, 0.f));
                        *reinterpret_cast<float4 *>(stg + px * 32 + (((2 * gq + hh) ^ (px & 7)) << 2)) = v;
                    }
                }
#pragma unroll
                for (int it = 0; it < 4 * NJ; ++it) {
                    const int idx = it * 64 + lane_e, px = idx >> 3, sl = idx & 7;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + px * 32 + ((sl ^ (px & 7)) << 2));
                    const int q = qw + px;
                    const size_t gidx = ((size_t)n_e * TV + q) * C + ob + 4 * sl;
                    if (q <= qlast_e) {
                        if constexpr (BF16OUT) {
                            *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(y) + gidx) =
                                make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                        } else {
                            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(y) + gidx) = v;
                        }
                    }
                }
            }
        } else if (!STGCN_ABL(4)) {
            float *stg = reinterpret_cast<float *>(buf0 + wave * (EPI_BYTES * NJ / 2));   // 32 ch x 32*NJ px per wave
            constexpr int PW = 32 * NJ;                          // pixel columns of this wave
            const int qw = q0_e + wn * PW;                       // first of them
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int ob = cg * 128 + (wm * 2 + m) * 32;     // first output channel of the block
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cr = (r & 3) + 8 * (r >> 2) + 4 * (lane_e >> 5);
                    const float sh = shift[ob + cr];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) stg[cr * PW + j * 32 + (lane_e & 31)] = fmaxf(acc[m][j][r] + sh, 0.f);
                }
#pragma unroll
                for (int it = 0; it < 4 * NJ; ++it) {
                    const int idx = it * 64 + lane_e, row = idx / (PW / 4), c4 = (idx % (PW / 4)) * 4;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + row * PW + c4);
                    const int q = qw + c4;
                    const size_t gidx = ((size_t)n_e * C + ob + row) * TV + q;
                    if (q + 3 <= qlast_e && (!BF16OUT || (gidx & 1) == 0)) {  // (bf16: keep the 8-B store dword-aligned)
                        if constexpr (BF16OUT) {
                            *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(y) + gidx) =
                                make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                        } else {
                            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(y) + gidx) = v;
                        }
                    } else {                                     // ragged end of the clip
                        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (q + e <= qlast_e) store_out<BF16OUT>(y, gidx + e, e4[e]);
                    }
                }
            }
        }
        STGCN_STAMP(t_e1)
        STGCN_ACC(3, t_e0, t_e1)
        if (next_tile < ntiles) {
            if constexpr (FK) {               // its fragments landed at the last stage barrier, its x during the stores
                __syncthreads();              // every wave's staging reads are done (the staging runs into Fs)
                STGCN_STAMP(t_f0)
                feature_phase(next_tile, xnext, xnext2);
                STGCN_STAMP(t_f1)
                STGCN_ACC(5, t_e1, t_f0)
                STGCN_ACC(6, t_f0, t_f1)
            } else {                          // its feature rows landed at the last stage barrier
                zero_invalid_rows(next_tile);
            }
            __syncthreads();
        }
        STGCN_STAMP(t_e2)
        STGCN_ACC(4, t_e1, t_e2)
    }
#ifdef STGCN_ABLATION
    if (dbg && lane == 0 && blockIdx.x < 8 && blockIdx.y == 0)
        for (int i = 0; i < 8; ++i) dbg[(blockIdx.x * 8 + wave) * 8 + i] = tsum[i];
#endif
}

// =====================================================================================================
// K3v4 — the stand-alone temporal conv block (Unit2D: K = 9, stride 1) in the same large-tile persistent form:
// same tile, weight ring, fenced tap body and epilogue as KF4; the matrix-core producer is replaced by staging of the
// real fp32 input.  A thread owns two (pixel, 8-channel) units of the chunk image: 8 coalesced dword loads each
// (lanes run along the pixels of one channel), issued at tap 0 for the NEXT chunk, split into bf16 hi/lo and stored
// as one 16-byte LDS store per image at taps 6 and 7.  Chunk 0 of the next tile is loaded during the last chunk and
// kept in registers across the epilogue (the image buffers are the epilogue's staging area).
// Serves the training path (raw forward conv, and the input gradient = this kernel on flipped weights) and Unit2D.eval.
// =====================================================================================================
template <int TERMS, bool BF16OUT>
__global__ __launch_bounds__(NT4) void tcn_bf16_v4_kernel(
    const float *__restrict__ x, const uint4 *__restrict__ Wp, const float *__restrict__ shift, void *y, int Cin, int C,
    int T, int V, int ROWS, int tiles_per_clip, int ntiles, float act_lo, int abl) {
    extern __shared__ __attribute__((aligned(16))) char smem4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int TV = T * V;
    const int nch = Cin / CCB;
    const int nstage = nch * (KT4 / STG);
    const int img_bytes = ROWS * PXB;
    const int buf_bytes = img_bytes * (TERMS == 3 ? 2 : 1);
    char *ring = smem4;                                             // 2 stages x 3 taps x 8 fragments
    char *buf0 = ring + 2 * STAGE_BYTES;                            // images; also the epilogue staging (8 x EPI_BYTES)
    char *buf1 = buf0 + buf_bytes;
    const unsigned ring_lds = (unsigned)(size_t)(lptr_t)smem4;
    const int cg = blockIdx.y;
    const uint4 *wsrc = Wp + ((size_t)(cg * 4 + (wave >> 1)) * nch * KT4 * 2 + (wave & 1)) * 64 + lane;
    auto dma_stage = [&](int gs) {
        const unsigned dst = ring_lds + (gs & 1) * STAGE_BYTES + wave * FRAG;
        const int gsm = gs % nstage;
#pragma unroll
        for (int t = 0; t < STG; ++t) dma16(wsrc + (size_t)(gsm * STG + t) * 128, dst + t * 8 * FRAG);
    };

    // ---- input staging: units u = tid, tid + 512 of the chunk image; unit = (pixel row u % ROWS, channel half u / ROWS)
    // (buffer-resource loads: one offset register per unit, the 8 channel strides ride in the scalar offset, and a
    //  pixel outside the clip gets an offset past num_records, which reads as zero)
    // (buffer-resource loads: one offset register per unit, the 8 channel strides ride in the scalar offset, and a
    //  pixel outside the clip gets an offset past num_records, which reads as zero)
    float pv[2][8];
    auto load_units = [&](int n, const TileGeomB &g, int ch) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + ((size_t)n * Cin + ch * CCB) * TV), 0, (unsigned)(CCB * TV * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = tid + i * NT4, hh = u >= ROWS ? 1 : 0, p = u - hh * ROWS;
            const int gi = g.origin + p;
            const bool ok = u < 2 * ROWS && p < g.span && gi >= 0 && gi < TV;
            const unsigned off = ok ? (unsigned)((hh * 8 * TV + gi) * 4) : 0x7ffffff0u;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                pv[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, c * TV * 4, 0));
        }
    };
    auto store_unit = [&](char *buf, int i) {
        const int u = tid + i * NT4, hh = u >= ROWS ? 1 : 0, p = u - hh * ROWS;
        if (u < 2 * ROWS) {
            uint4 hi, lo;
            split8(pv[i], hi, lo);
            const int off = lds_off(p, hh);
            *reinterpret_cast<uint4 *>(buf + off) = hi;
            if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(buf + img_bytes + off) = lo;
        }
    };

    int tile = blockIdx.x;
    dma_stage(0);
    if (tile < ntiles) {
        const int n = tile / tiles_per_clip;
        load_units(n, tile_geom_b(tile - n * tiles_per_clip, V, KT4, 1, T, NP4), 0);
    }
    dma_wait();                                   // (vmcnt(0): weight stage 0 and the first chunk's loads)
    int gs = 0;
    const int h = lane >> 5;
    for (; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT4, 1, T, NP4);
        const int next_tile = tile + gridDim.x;
        TileGeomB g2 = g;
        int n2 = n;
        if (next_tile < ntiles) {
            n2 = next_tile / tiles_per_clip;
            g2 = tile_geom_b(next_tile - n2 * tiles_per_clip, V, KT4, 1, T, NP4);
        }
        store_unit(buf0, 0);                      // chunk 0 (loaded before the loop / during the previous tile's last chunk)
        store_unit(buf0, 1);
        int prow[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int q = g.q0 + (wn * 2 + j) * 32 + (lane & 31);
            q = min(q, g.q_last);
            const int t = q / V, v = q - t * V;
            prow[j] = (t - g.t_first) * V + v;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
        __syncthreads();                          // chunk 0 and the current weight stage visible
        auto load_a = [&](Frag2<TERMS> &a, const char *aslot, int tt) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                a.hi[m] = *reinterpret_cast<const uint4 *>(aslot + (tt * 8 + m * 2) * FRAG);
                if constexpr (TERMS == 3) a.lo[m] = *reinterpret_cast<const uint4 *>(aslot + (tt * 8 + m * 2 + 1) * FRAG);
            }
        };
        auto load_b = [&](Frag2<TERMS> &b, const char *img, int tap) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = lds_off(prow[j] + tap * V, h);
                b.hi[j] = *reinterpret_cast<const uint4 *>(img + off);
                if constexpr (TERMS == 3) b.lo[j] = *reinterpret_cast<const uint4 *>(img + img_bytes + off);
            }
        };
        Frag2<TERMS> a_cur = {}, b_cur = {}, a_nxt = {}, b_nxt = {};
        for (int ch = 0; ch < nch; ++ch) {
            const char *cur = (ch & 1) ? buf1 : buf0;
            char *nxt = (ch & 1) ? buf0 : buf1;
            const bool last = ch + 1 == nch;
            load_b(b_cur, cur, 0);
#pragma unroll
            for (int st = 0; st < KT4 / STG; ++st, ++gs) {
                const char *aslot = ring + (gs & 1) * STAGE_BYTES + (wm * 4) * FRAG + lane * 16;
                load_a(a_cur, aslot, 0);
#pragma unroll
                for (int tt = 0; tt < STG; ++tt) {
                    const int tap = st * STG + tt;
                    if (tt + 1 < STG) load_a(a_nxt, aslot, tt + 1);
                    if (tap + 1 < KT4) load_b(b_nxt, cur, tap + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_half_bf16<TERMS>(acc, a_cur, b_cur, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tt == 0) dma_stage(gs + 1);
                    if (tap == 0) {               // the next chunk's (or, in the last chunk, the next tile's first) input
                        if (!last) load_units(n, g, ch + 1);
                        else if (next_tile < ntiles) load_units(n2, g2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_half_bf16<TERMS>(acc, a_cur, b_cur, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!last && tap == 6) store_unit(nxt, 0);
                    if (!last && tap == 7) store_unit(nxt, 1);
                    a_cur = a_nxt;
                    b_cur = b_nxt;
                }
                dma_wait();
                __syncthreads();                  // stage done: next weights landed and visible; chunk boundary at st == 2
            }
        }
        // epilogue (see KF4): each 32-channel x 64-pixel block through this wave's 8 KiB staging slice, 16 B per lane
        {
            int n_e = n, q0_e = g.q0, qlast_e = g.q_last, lane_e = lane;   // opaque copies: see KF4's epilogue
            asm volatile("" : "+s"(n_e), "+s"(q0_e), "+s"(qlast_e), "+v"(lane_e));
            float *stg = reinterpret_cast<float *>(buf0 + wave * EPI_BYTES);
            const int qw = q0_e + wn * 64;
            if (abl & OPT_OUT_NTVC) {
                const int hh = lane_e >> 5;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int ob = cg * 128 + (wm * 2 + m) * 32;
                    if (ob >= C) continue;                // padded rows of a 64-channel layer (wave-uniform)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 8 * gq + 4 * hh);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int px = j * 32 + (lane_e & 31);
                            const float4 v = make_float4(fmaxf(acc[m][j][4 * gq + 0] + sh4.x, act_lo), fmaxf(acc[m][j][4 * gq + 1] + sh4.y, act_lo),
                                                         fmaxf(acc[m][j][4 * gq + 2] + sh4.z, act_lo), fmaxf(acc[m][j][4 * gq + 3] + sh4.w, act_lo));
                            *reinterpret_cast<float4 *>(stg + px * 32 + (((2 * gq + hh) ^ (px & 7)) << 2)) = v;
                        }
                    }
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int idx = it * 64 + lane_e, px = idx >> 3, sl = idx & 7;
                        const float4 v = *reinterpret_cast<const float4 *>(stg + px * 32 + ((sl ^ (px & 7)) << 2));
                        const int q = qw + px;
                        const size_t gidx = ((size_t)n_e * TV + q) * C + ob + 4 * sl;
                        if (q <= qlast_e) {
                            if constexpr (BF16OUT)
                                *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(y) + gidx) =
                                    make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                            else
                                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(y) + gidx) = v;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int ob = cg * 128 + (wm * 2 + m) * 32;
                    if (ob >= C) continue;                // padded rows of a 64-channel layer (wave-uniform)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cr = (r & 3) + 8 * (r >> 2) + 4 * (lane_e >> 5);
                        const float sh = shift[ob + cr];
#pragma unroll
                        for (int j = 0; j < 2; ++j) stg[cr * 64 + j * 32 + (lane_e & 31)] = fmaxf(acc[m][j][r] + sh, act_lo);
                    }
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int idx = it * 64 + lane_e, row = idx >> 4, c4 = (idx & 15) * 4;
                        const float4 v = *reinterpret_cast<const float4 *>(stg + row * 64 + c4);
                        const int q = qw + c4;
                        const size_t gidx = ((size_t)n_e * C + ob + row) * TV + q;
                        if (q + 3 <= qlast_e && (gidx & 3) == 0) {       // 16-byte (8-byte for bf16) aligned store
                            if constexpr (BF16OUT)
                                *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(y) + gidx) =
                                    make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                            else
                                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(y) + gidx) = v;
                        } else {
                            const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (q + e <= qlast_e) store_out<BF16OUT>(y, gidx + e, e4[e]);
                        }
                    }
                }
            }
        }
        __syncthreads();                          // staging consumed before the next tile's chunk 0 is stored
    }
}

struct V4Plan {
    int pb = 0, rows = 0, tiles_per_clip = 0, nj = 0;
    size_t lds = 0;
};

inline bool plan_v4_nj(int C, int T, int V, int K, int terms, int nj, V4Plan &pl) {
    if (K != KT4 || C % 128 != 0) return false;
    const int np = 128 * nj;
    int dt = ceil_div(np - 1, V);
    if (dt > T - 1) dt = T - 1;
    const int span = (dt + K) * V;
    const bool fk = nj == 2;                         // features computed in the kernel
    if (fk && V > 32) return false;                  // (one 32-deep k-step over the joints, two 16-column halves)
    // feature DMA moves 64 rows per wave-instruction; the in-kernel form only needs whole 16-pixel producer blocks
    const int rows = fk ? (span + 15) / 16 * 16 : (span + 63) / 64 * 64;
    const int pb = ceil_div(ceil_div(span, 16), 8);  // producer blocks per wave per chunk (8 producing waves)
    if (pb > KT4) return false;
    const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
    const size_t stage = (size_t)8 * EPI_BYTES * nj / 2;                       // epilogue staging: 8 waves x 32 ch x 32*nj px
    size_t lds = (nj == 2 ? (size_t)C * W12P * 4 : 0) + 2 * STAGE_BYTES;
    if (fk) {   // images | Fs (the staging may run into it) | attention fragments (behind the staging)
        const size_t body = 2 * buf + (size_t)rows * 64;
        lds += (body > stage ? body : stage) + 12 * FRAG;
    } else {
        lds += (2 * buf > stage ? 2 * buf : stage) + (size_t)rows * 64;       // images / epilogue staging, then Fs
    }
    if (lds > (size_t)kLdsBytes) return false;
    pl.pb = pb;
    pl.rows = rows;
    pl.tiles_per_clip = ceil_div(T * V, np);
    pl.nj = nj;
    pl.lds = lds;
    return true;
}

// the 256-pixel tile where its rows fit LDS (V <= ~25), else the 128-pixel tile (V = 46)
inline bool plan_v4(int C, int T, int V, int K, int terms, V4Plan &pl) {
    return plan_v4_nj(C, T, V, K, terms, 2, pl) || plan_v4_nj(C, T, V, K, terms, 1, pl);
}

template <int PB, int TERMS, int NJ>
int launch_v4(const float4 *feat, const float *x, int xsc, int xsp, const float *W12, const uint4 *Wp, const float *shift,
              void *y, int N, int C, int T, int V, const V4Plan &pl, bool bf16out, int opt, int num_cu, hipStream_t st) {
    const int ntiles = N * pl.tiles_per_clip;
    const int gx = ntiles < num_cu ? ntiles : num_cu;
    const dim3 grid(gx, C / 128, 1);
    if (bf16out) {
        auto kern = stem_bf16_v4_kernel<PB, TERMS, true, NJ>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT4), pl.lds, st, feat, x, xsc, xsp, W12, Wp, shift, y, C, T, V, pl.rows,
                           pl.tiles_per_clip, ntiles, ablate_mask() | opt, debug_buffer());
    } else {
        auto kern = stem_bf16_v4_kernel<PB, TERMS, false, NJ>;
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));
        hipLaunchKernelGGL(kern, grid, dim3(NT4), pl.lds, st, feat, x, xsc, xsp, W12, Wp, shift, y, C, T, V, pl.rows,
                           pl.tiles_per_clip, ntiles, ablate_mask() | opt, debug_buffer());
    }
    STGCN_LAUNCH_CHECK("stem_bf16_v4_kernel");
    return STGCN_OK;
}

struct T4Plan {
    int rows = 0, tiles_per_clip = 0;
    size_t lds = 0;
};

inline bool plan_t4(int Cin, int Cout, int T, int V, int K, int stride, int terms, T4Plan &pl) {
    if (K != KT4 || stride != 1 || (Cout % 128 != 0 && Cout != 64) || Cin % CCB != 0 || T < 1) return false;
    int dt = ceil_div(NP4 - 1, V);
    if (dt > T - 1) dt = T - 1;
    const int span = (dt + K) * V;
    const int rows = (span + 7) / 8 * 8;
    if (rows > NT4) return false;                 // two (pixel, 8-channel) units per thread
    const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
    const size_t img = 2 * buf > (size_t)8 * EPI_BYTES ? 2 * buf : (size_t)8 * EPI_BYTES;
    pl.lds = 2 * STAGE_BYTES + img;
    if (pl.lds > (size_t)kLdsBytes) return false;
    pl.rows = rows;
    pl.tiles_per_clip = ceil_div(T * V, NP4);
    return true;
}

}  // namespace

bool tcn_v4_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16) return false;
    T4Plan pl;
    return plan_t4(Cin, Cout, T, V, K, stride, math == STGCN_MATH_BF16X3 ? 3 : 1, pl);
}

int launch_tcn_v4(const float *x, const void *Wp, const float *shift, void *y, int N, int Cin, int Cout, int T, int V, int K,
                  int stride, unsigned flags, hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const int terms = math == STGCN_MATH_BF16X3 ? 3 : 1;
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    const float act_lo = (flags & STGCN_RAW) ? -__builtin_huge_valf() : 0.f;
    const int opt = (flags & STGCN_OUT_NTVC) ? OPT_OUT_NTVC : 0;
    T4Plan pl;
    if (!plan_t4(Cin, Cout, T, V, K, stride, terms, pl))
        return fail(STGCN_ERR_UNSUPPORTED, "tcn v4 kernel does not cover Cin=%d Cout=%d T=%d V=%d K=%d stride=%d", Cin, Cout, T,
                    V, K, stride);
    int dev = 0, num_cu = 256;
    STGCN_HIP_CHECK(hipGetDevice(&dev));
    STGCN_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int ntiles = N * pl.tiles_per_clip;
    const dim3 grid(ntiles < num_cu ? ntiles : num_cu, ceil_div(Cout, 128), 1);
#define LAUNCH_T4(TERMS, B)                                                                                       \
    do {                                                                                                          \
        auto kern = tcn_bf16_v4_kernel<TERMS, B>;                                                                 \
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(NT4), pl.lds, st, x, (const uint4 *)Wp, shift, y, Cin, Cout, T, V, pl.rows, \
                           pl.tiles_per_clip, ntiles, act_lo, opt);                                               \
    } while (0)
    if (terms == 3) { if (bf16out) LAUNCH_T4(3, true); else LAUNCH_T4(3, false); }
    else { if (bf16out) LAUNCH_T4(1, true); else LAUNCH_T4(1, false); }
#undef LAUNCH_T4
    STGCN_LAUNCH_CHECK("tcn_bf16_v4_kernel");
    return STGCN_OK;
}

bool stem_v4_supported(int Cin, int C, int T, int V, int K, int S, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16) return false;
    if (Cin != 3 || S != 3 || T < 1) return false;
    V4Plan pl;
    return plan_v4(C, T, V, K, math == STGCN_MATH_BF16X3 ? 3 : 1, pl) && attention_emits_features(Cin, V, S);
}

bool stem_v4_features_in_kernel(int C, int T, int V, int K, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    V4Plan pl;
    if (!((math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) && plan_v4(C, T, V, K, math == STGCN_MATH_BF16X3 ? 3 : 1, pl)))
        return false;
    // the 256-pixel tile, or — wide frames — KF6 over the two joint halves (diagnostic builds: mask bit 256 keeps KF4)
    return pl.nj == 2 || (stem_v6w_supported(C, T, V, K, flags) && !(ablate_mask() & 256));
}

int launch_stem_v4(const float *x, bool x_ntvc, const float *feat, const void *prep_w12, const void *Wp, const float *shift,
                   void *out, int N, int C, int T, int V, int K, unsigned flags, hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const int terms = math == STGCN_MATH_BF16X3 ? 3 : 1;
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    const int opt = (flags & STGCN_OUT_NTVC) ? OPT_OUT_NTVC : 0;
    V4Plan pl;
    if (!plan_v4(C, T, V, K, terms, pl))
        return fail(STGCN_ERR_UNSUPPORTED, "stem v4 kernel does not cover C=%d T=%d V=%d K=%d", C, T, V, K);
    int dev = 0, num_cu = 256;
    STGCN_HIP_CHECK(hipGetDevice(&dev));
    STGCN_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    // the 256-pixel tile runs with one wave per SIMD (KF6, stem_bf16_v6.hip) where that form covers the shape
    // (diagnostic builds: mask bit 256 keeps the eight-wave form for A/B runs in one process)
    if (pl.nj == 2 && stem_v6_supported(C, T, V, K, flags) && !(ablate_mask() & 256))
        return launch_stem_v6(x, x_ntvc, feat, prep_w12, (const char *)Wp + tcn_packed_single_bytes(C, C, K, flags), shift, out, N, C, T,
                              V, K, flags, st);
    // wide frames (the two-hand graph): the same one-wave-per-SIMD kernel over the two joint halves (stem_bf16_v6w.hip)
    if (pl.nj == 1 && stem_v6w_supported(C, T, V, K, flags) && !(ablate_mask() & 256))
        return launch_stem_v6w(x, x_ntvc, feat, prep_w12, (const char *)Wp + tcn_packed_single_bytes(C, C, K, flags), shift, out, N, C,
                               T, V, K, flags, st);
    const float4 *f4 = (const float4 *)feat;
    const float *W12 = (const float *)prep_w12;
    const uint4 *wp = (const uint4 *)Wp;
    const int xsc = x_ntvc ? 1 : T * V, xsp = x_ntvc ? 3 : 1;   // element (channel k, pixel p) of a clip at k*xsc + p*xsp
    if (pl.nj == 2 && (size_t)3 * T * V * 4 >= ((size_t)1 << 31))
        return fail(STGCN_ERR_UNSUPPORTED, "stem v4: clip of T=%d V=%d exceeds a buffer resource", T, V);
#define GO(PB, NJ)                                                                                              \
    return terms == 3 ? launch_v4<PB, 3, NJ>(f4, x, xsc, xsp, W12, wp, shift, out, N, C, T, V, pl, bf16out, opt, num_cu, st)     \
                      : launch_v4<PB, 1, NJ>(f4, x, xsc, xsp, W12, wp, shift, out, N, C, T, V, pl, bf16out, opt, num_cu, st)
    if (pl.nj == 1) {
        if (pl.pb <= 6) GO(6, 1);
        GO(9, 1);
    }
    if (pl.pb <= 4) GO(4, 2);
    if (pl.pb <= 6) GO(6, 2);
    GO(9, 2);
#undef GO
}

}  // namespace stgcn
