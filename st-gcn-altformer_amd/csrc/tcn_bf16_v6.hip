// K3v6 — the stand-alone temporal conv block (Unit2D: K = 9, stride 1) in KF6's form (stem_bf16_v6.hip): ONE WAVE PER SIMD
// (256 threads; a wave owns all 128 output channels of 64 pixels) on v_mfma_f32_16x16x32_bf16, slot-structured loop, weights
// through a ring of 3 pair slots filled by LDS-DMA two pairs ahead, accumulators pinned to the AGPR file.  Serves Unit2D.eval,
// the training forward (raw mode) and the input gradient (this kernel on flipped weights).
//
// What replaces KF6's matrix-core producer is staging of the real fp32 input: the chunk image (16 channels x the tile's
// pixel rows incl. the temporal halo) is 2 * ROWS (pixel, 8-channel) units, four per lane; a unit is 8 coalesced dword
// loads through a buffer resource (pixels outside the clip read as zeros), split into bf16 hi / lo and stored as one
// 16-byte LDS store per image.  The units travel as a SEQUENCE of (tile, chunk) elements that runs across this workgroup's
// tiles: while element e is converted and stored (window A of a period: pairs 0-2 -> buf1, window B: pairs 5-7 -> buf0),
// each unit's registers are re-loaded with element e+1 right behind its store — two or more pairs ahead of its own window.
// In a tile's last period window B therefore stores chunk 0 of the NEXT tile into buf0 (idle since pair 4): a tile starts
// with its first chunk in place, there is no chunk-0 phase, and the epilogue has its own 16 KiB of staging.
// The input loads share vmcnt with the weight DMAs.  They are issued BEHIND the pair's four DMAs, and the pair ends with
// s_waitcnt vmcnt(n) for its n input loads: the DMAs (older) have landed, the loads stay in flight across the barrier.
#include <type_traits>

#include "bf16_common.h"

namespace stgcn {

namespace {

using namespace bf16k;

constexpr int NP6 = 256;   // output pixels per tile
constexpr int NT6 = 256;   // threads per workgroup: one wave per SIMD
constexpr int KT6 = 9;     // temporal taps
constexpr int FRAG6 = 1024;
constexpr int PAIR6 = 16 * FRAG6;   // weights of one pair: 8 blocks of 16 channels x (hi, lo)
constexpr int RING6 = 3 * PAIR6;
constexpr int EPI6 = 4096; // epilogue staging per wave: 16 channels x 64 pixels fp32

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void *lptr6_t;

__device__ __forceinline__ void dma16t6(const void *g, unsigned lds_addr) {
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr);   // (M0 clobbered, not saved: see stem_bf16_v6.hip)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void vm_wait_keep() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct FragB6 { uint4 hi[4], lo[4]; };     // activations of one pair: 4 pixel blocks of 16

template <int I, int N, class F>
__device__ __forceinline__ void static_for6(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for6<I + 1, N>(f);
    }
}

// weight packing in pair order (stem_bf16_v6.hip), with the output channels padded to a multiple of 128 (64 -> 128: rows
// beyond Cout are zero and never stored)
__global__ void tcn_pack_pairs_padded_kernel(const float *__restrict__ W, const float *__restrict__ scale,
                                             unsigned short *__restrict__ Wq, int Cin, int Cout, int CoutP) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (weight, img)
    if (e >= (size_t)CoutP * Cin * KT6 * 2) return;
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
    const float w = o < Cout ? scale[o] * W[((size_t)o * Cin + c) * KT6 + tap] : 0.f;
    const unsigned h = pack_bf16x2(w, 0.f) & 0xffffu;
    const unsigned l = pack_bf16x2(w - bf16_lo_to_f32(h), 0.f) & 0xffffu;
    Wq[e] = (unsigned short)(img ? l : h);
}

// STATS (the training forward, model/net.py:40 in .train()): the epilogue also sums every output channel's values and
// squares over the stored pixels — fp32 over the 64 pixels of a staged row, fp64 from there on (2 KiB of LDS per workgroup,
// one pair of global fp64 atomics per workgroup and channel at the end) — into stats[c], stats[C + c]: the batch statistics
// of BatchNorm without the separate pass over z (bn_batch_stats_kernel: 105 us of the stem's training step).
// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}

template <int TERMS, bool BF16OUT, bool STATS = false>
__global__ __launch_bounds__(NT6) void tcn_bf16_v6_kernel(const float *__restrict__ x, const uint4 *__restrict__ Wp,
                                                          const float *__restrict__ shift, void *y, int Cin, int C, int T, int V,
                                                          int ROWS, int tiles_per_clip, int ntiles, float act_lo, int abl,
                                                          double *__restrict__ stats = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem6[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = pixel quarter of the tile
    const int TV = T * V;
    const int nch = Cin / CCB;               // channel chunks (even: host side)
    const int npairs = nch * KT6 / 2;        // K = 32 steps per tile
    const int img_bytes = ROWS * PXB;
    const int buf_bytes = img_bytes * (TERMS == 3 ? 2 : 1);
    // LDS carve: weight ring (3 pairs) | images buf0, buf1 | epilogue staging (4 x 4 KiB)
    char *ring = smem6;
    char *buf0 = ring + RING6;
    char *buf1 = buf0 + buf_bytes;
    char *stage = buf1 + buf_bytes;
    const unsigned ring_lds = (unsigned)(size_t)(lptr6_t)smem6;
    double *sstat = reinterpret_cast<double *>(stage + 4 * EPI6);     // STATS: [2][128] (sum, sum of squares)
    if constexpr (STATS) sstat[tid] = 0.0;                            // (the main loop's barriers order this before the first add)

    const int cg = blockIdx.y;               // 128-channel group of the output
    // the four weight fragments this wave DMAs per pair: 16-channel blocks 2*wave, 2*wave+1, images hi and lo
    const uint4 *wsrc = Wp + ((size_t)(cg * 8 + 2 * wave) * npairs * 2) * 64 + lane;
    auto dma_frag = [&](int qsrc, int slot, int d) {
        const int bw = d >> 1, img = d & 1;
        dma16t6(wsrc + ((size_t)(bw * npairs + qsrc) * 2 + img) * 64, ring_lds + slot * PAIR6 + ((2 * wave + bw) * 2 + img) * FRAG6);
    };

    // ---- input staging: this lane's four (pixel row, channel half) units of a chunk image -----------------------------
    // (a lane whose unit index runs past the image repeats the last unit: same address, same data, no branch)
    int up[4], uoff[4];                      // pixel row, LDS offset
    unsigned ugo[4];                         // byte offset of (channel half, pixel row) in the chunk's 16 channel planes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = min(tid + i * NT6, 2 * ROWS - 1);
        const int hh = u >= ROWS ? 1 : 0;
        up[i] = u - hh * ROWS;
        uoff[i] = lds_off(up[i], hh);
        ugo[i] = (unsigned)((hh * 8 * TV + up[i]) * 4);
    }
    int so[8];                               // scalar offsets of a unit's 8 channel planes
#pragma unroll
    for (int c = 0; c < 8; ++c) so[c] = __builtin_amdgcn_readfirstlane(c * TV * 4);
    float pv[4][8];
    // The element (tile, chunk) the registers hold / are being filled with belongs to the current tile or (from the last
    // period's window A on) to the workgroup's next one.  Per tile, all scalar: the byte offset of the tile's first pixel row,
    // the range [lo, lo + w) of pixel rows that lie inside the clip, and the address of the clip's channel plane 0.
    struct EGeom { int o4, lo; unsigned w; const float *base; };
    auto geom_of = [&](int t) {
        const int tl = min(t, ntiles - 1);
        const int en = tl / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tl - en * tiles_per_clip, V, KT6, 1, T, NP6);
        const int lo = max(0, -g.origin), hi = min(t < ntiles ? g.span : 0, TV - g.origin);   // past the last tile: empty
        return EGeom{g.origin * 4, lo, (unsigned)max(hi - lo, 0), x + (size_t)en * Cin * TV};
    };
    EGeom gcur = geom_of(blockIdx.x), gnxt = geom_of(blockIdx.x + gridDim.x);
    int e_ch = 0, e_o4 = gcur.o4, e_lo = gcur.lo;
    unsigned e_w = gcur.w;
    bool e_next = false;
    __amdgpu_buffer_rsrc_t e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gcur.base), 0, (unsigned)(CCB * TV * 4), 0x00020000);
    auto e_advance = [&]() {                 // next element of the sequence (branch-free: it sits in an MFMA slot)
        const bool wrap = e_ch + 1 == nch;
        e_ch = wrap ? 0 : e_ch + 1;
        e_next = e_next || wrap;
        e_o4 = e_next ? gnxt.o4 : gcur.o4;
        e_lo = e_next ? gnxt.lo : gcur.lo;
        e_w = e_next ? gnxt.w : gcur.w;
        const float *b = (e_next ? gnxt.base : gcur.base) + (size_t)e_ch * CCB * TV;
        e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b), 0, (unsigned)(CCB * TV * 4), 0x00020000);
    };
    // 8 dword loads of a unit of the current element (lanes run along the pixels of one channel; the 8 channel strides ride
    // in the scalar offset; a pixel outside the clip gets an offset past num_records, which reads as zero)
    auto unit_off = [&](int i) -> unsigned {
        return (unsigned)(up[i] - e_lo) < e_w ? ugo[i] + (unsigned)e_o4 : 0x7ffffff0u;
    };
    auto load_unit_one = [&](int i, int c, unsigned off) {
        pv[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(e_rs, off, so[c], 0));
    };
    auto load_unit = [&](int i) {
        const unsigned off = unit_off(i);
#pragma unroll
        for (int c = 0; c < 8; ++c) load_unit_one(i, c, off);
    };
    auto store_unit = [&](char *buf, int i) {
        uint4 hi, lo;
        split8(pv[i], hi, lo);
        *reinterpret_cast<uint4 *>(buf + uoff[i]) = hi;
        if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(buf + img_bytes + uoff[i]) = lo;
    };

    // ---- one-time setup: element 0 -> buf0, element 1 -> registers, weight pairs 0 and 1 ---------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) load_unit(i);
#pragma unroll
    for (int d = 0; d < 4; ++d) { dma_frag(0, 0, d); dma_frag(1, 1, d); }
#pragma unroll
    for (int i = 0; i < 4; ++i) store_unit(buf0, i);
    e_advance();
#pragma unroll
    for (int i = 0; i < 4; ++i) load_unit(i);
    vm_wait_keep<32>();                       // the eight weight fragments have landed (the 32 input loads stay in flight)
    __syncthreads();

    // ring bookkeeping without divisions: slot of the current pair, and (slot, source index) of the pair two ahead
    int gq = 0, slot0 = 0, slot2 = 2, q2 = 2 % npairs;
    const int sel = lane >> 5, chh = (lane >> 4) & 1;   // B fragment lane groups: step of the pair, channel half
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_clip;
        const TileGeomB g = tile_geom_b(tile - n * tiles_per_clip, V, KT6, 1, T, NP6);
        // LDS offsets of this lane's activation rows per tap, for the wave's first 16-pixel block (block nb sits nb*16*PXB
        // bytes further: ds_read immediates; see stem_bf16_v6.hip)
        unsigned boff[KT6];
        {
            const int q = g.q0 + wave * 64 + (lane & 15);
            const int prow = q - g.t_first * V;
#pragma unroll
            for (int tap = 0; tap < KT6; ++tap) boff[tap] = (unsigned)lds_off(prow + tap * V, chh);
        }
        f32x4 acc[8][4];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto rd = [&](const char *p) { return *reinterpret_cast<const uint4 *>(p); };
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
        static_for6<0, 4>([&](auto nb_c) {      // pair 0 of the tile (its chunk 0 was stored during the previous tile)
            load_b(b_cur, IC0{}, nb_c, std::false_type{});
            if constexpr (TERMS == 3) load_b(b_cur, IC0{}, nb_c, std::true_type{});
        });
        const int nper = nch / 2;
        for (int per = 0; per < nper; ++per) {
            static_for6<0, 9>([&](auto pi_c) {
                constexpr int pi = decltype(pi_c)::value;
                constexpr int l0 = 2 * pi;
                // staging windows: pairs 0-2 store the held element into buf1, pairs 5-7 the next one into buf0; units
                // (0, 1), (2), (3) of the lane per pair, each re-loaded with the element after right behind its store
                constexpr int win = pi <= 2 ? 0 : (pi >= 5 && pi <= 7 ? 1 : -1);
                constexpr int wpi = win == 0 ? pi : pi - 5;
                constexpr int nun = win < 0 ? 0 : (wpi == 0 ? 2 : 1);      // units of this pair
                constexpr int u0 = wpi == 0 ? 0 : wpi + 1;                 // first unit of this pair
                char *pbuf = win == 0 ? buf1 : buf0;
                const int slot1 = slot0 == 2 ? 0 : slot0 + 1;
                const char *aslot = ring + slot0 * PAIR6 + lane * 16;
                const char *anext = ring + slot1 * PAIR6 + lane * 16;
                uint4 ah[2], al[2];
                ah[0] = ah0n;
                if constexpr (TERMS == 3) al[0] = al0n;
                // staging state of the unit in flight through this pair's fillers
                unsigned sh0 = 0, sh1 = 0, sh2 = 0, sh3 = 0, sl0 = 0, sl1 = 0, sl2 = 0, sl3 = 0, goff = 0;
                constexpr int NM = 32 * TERMS;                // MFMAs of the pair
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
                    // staging: unit b of this pair occupies fillers 8 + 28*b ... (behind the weight DMAs at 4-7)
                    if constexpr (nun > 0 && v >= 8 && (v - 8) / 28 < nun) {
                        constexpr int b = (v - 8) / 28, w = (v - 8) % 28, ui = u0 + b;
                        if constexpr (w == 0 && b == 0 && wpi == 0) e_advance();     // the element the re-loads fetch
                        if constexpr (w == 1) { sh0 = pack_bf16x2(pv[ui][0], pv[ui][1]); sh1 = pack_bf16x2(pv[ui][2], pv[ui][3]); }
                        if constexpr (w == 2) { sh2 = pack_bf16x2(pv[ui][4], pv[ui][5]); sh3 = pack_bf16x2(pv[ui][6], pv[ui][7]); }
                        if constexpr (w == 3) *reinterpret_cast<uint4 *>(pbuf + uoff[ui]) = make_uint4(sh0, sh1, sh2, sh3);
                        if constexpr (TERMS == 3 && w == 4) sl0 = pack_bf16x2(pv[ui][0] - bf16_lo_to_f32(sh0), pv[ui][1] - bf16_hi_to_f32(sh0));
                        if constexpr (TERMS == 3 && w == 5) sl1 = pack_bf16x2(pv[ui][2] - bf16_lo_to_f32(sh1), pv[ui][3] - bf16_hi_to_f32(sh1));
                        if constexpr (TERMS == 3 && w == 6) sl2 = pack_bf16x2(pv[ui][4] - bf16_lo_to_f32(sh2), pv[ui][5] - bf16_hi_to_f32(sh2));
                        if constexpr (TERMS == 3 && w == 7) sl3 = pack_bf16x2(pv[ui][6] - bf16_lo_to_f32(sh3), pv[ui][7] - bf16_hi_to_f32(sh3));
                        if constexpr (TERMS == 3 && w == 8) *reinterpret_cast<uint4 *>(pbuf + img_bytes + uoff[ui]) = make_uint4(sl0, sl1, sl2, sl3);
                        if constexpr (w == 9) goff = unit_off(ui);
                        if constexpr (w >= 10 && w < 18) load_unit_one(ui, w - 10, goff);
                    }
                    // weights of pair gq + 2 -> the slot pair gq - 1 occupied (its readers passed the last barrier)
                    if constexpr (v >= 4 && v < 8) dma_frag(q2, slot2, v - 4);
                    // block 0 of the next pair
                    if constexpr (v == 88) ah0n = rd(anext);
                    if constexpr (TERMS == 3 && v == 89) al0n = rd(anext + FRAG6);
                };
                static_for6<0, NM>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value;
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
                    if constexpr (term == TERMS - 1) asm volatile("" : "+a"(acc[mb][nb]));   // accumulators live in AGPRs
                    static_for6<i * (96 / NM), (i + 1) * (96 / NM)>(filler);
                    __builtin_amdgcn_sched_barrier(0);
                });
                b_cur = b_nxt;
                // pair gq+2's weights have landed; this pair's input loads (issued behind them) stay in flight
                vm_wait_keep<nun * 8>();
                __syncthreads();              // ... weights and stored image rows are visible; slot gq%3 is free
                ++gq;
                slot0 = slot1;
                slot2 = slot2 == 2 ? 0 : slot2 + 1;
                q2 = q2 + 1 == npairs ? 0 : q2 + 1;
            });
        }

        // ---- epilogue: each 16-channel x 64-pixel block through this wave's 4 KiB staging slice, 16 B per lane ----------
        // (Round 3 tried the transposed product instead — pixels as the A operand, so that a lane holds four consecutive pixels
        //  of one channel and stores 16 bytes straight from the accumulators, no staging: parity-green and 3 % SLOWER per
        //  launch in a same-box A/B.  A store instruction then covers 16 channel rows x 64 bytes instead of 4 rows x 256: the
        //  epilogue is bound by the cache lines a CU's store path touches (~9 B/clk/CU either way), not by the LDS round trip.)
        float *stg = reinterpret_cast<float *>(stage + wave * EPI6);
        const int qw = g.q0 + wave * 64;
        const bool full = g.q0 + NP6 - 1 <= g.q_last;            // (scalar) every pixel of the tile lies inside the clip
        if (abl & OPT_OUT_NTVC) {
            const unsigned lterm = (unsigned)((lane >> 2) * C + 4 * (lane & 3));
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int ob = cg * 128 + mb * 16;
                if (ob >= C) continue;                           // padded rows of a 64-channel layer
                const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 4 * (lane >> 4));
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int px = nb * 16 + (lane & 15);
                    const float4 v = make_float4(fmaxf(acc[mb][nb][0] + sh4.x, act_lo), fmaxf(acc[mb][nb][1] + sh4.y, act_lo),
                                                 fmaxf(acc[mb][nb][2] + sh4.z, act_lo), fmaxf(acc[mb][nb][3] + sh4.w, act_lo));
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
            const unsigned lterm = (unsigned)((lane >> 4) * TV + 4 * (lane & 15));
            const int c4l = 4 * (lane & 15);
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int ob = cg * 128 + mb * 16;
                if (ob >= C) continue;                           // padded rows of a 64-channel layer
                const float4 sh4 = *reinterpret_cast<const float4 *>(shift + ob + 4 * (lane >> 4));
                const float shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stg[(4 * (lane >> 4) + r) * 64 + nb * 16 + (lane & 15)] = fmaxf(acc[mb][nb][r] + shv[r], act_lo);
                const size_t tbase = ((size_t)n * C + ob) * TV + qw;      // scalar
                const bool al16 = ((tbase & 3) == 0) && (TV % 4 == 0);    // 16-byte (8-byte for bf16) aligned rows
                [[maybe_unused]] float k1 = 0.f, k2 = 0.f;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const float4 v = *reinterpret_cast<const float4 *>(stg + (it * 4 + (lane >> 4)) * 64 + c4l);
                    const size_t sbase = tbase + (size_t)(it * 4) * TV;    // scalar
                    if constexpr (STATS) {   // row (it*4 + lane>>4) of the block: 16 lanes x 4 pixels
                        const float e4[4] = {v.x, v.y, v.z, v.w};
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = (full || qw + c4l + e <= g.q_last) ? e4[e] : 0.f;
                            s1 += a;
                            s2 = fmaf(a, a, s2);
                        }
                        // the row's 16 lanes: quad butterflies, then the mirrored half-row and row (DPP: no LDS round trip —
                        // with ds_bpermute shuffles this cost more than the separate statistics pass it replaces)
                        s1 = row16_sum(s1);
                        s2 = row16_sum(s2);
                        if ((lane & 15) == it) { k1 = s1; k2 = s2; }   // lane (lane & 15) = it keeps row it*4 + (lane >> 4)
                    }
                    if (full && al16) {
                        if constexpr (BF16OUT)
                            *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(y) + sbase + lterm) =
                                make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                        else
                            {   // non-temporal (round 3: whole lines streamed past the L2 that serves the weight ring: -0.7 % on the
                                // training step, -1 % on the two-stage eval path in same-box A/Bs; see stem_bf16_v6.hip)
                                using f32x4v = __attribute__((ext_vector_type(4))) float;
                                __builtin_nontemporal_store(f32x4v{v.x, v.y, v.z, v.w},
                                                            reinterpret_cast<f32x4v *>(reinterpret_cast<float *>(y) + sbase + lterm));
                            }
                    } else {                                     // last tile of a clip / unaligned rows: element by element
                        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (qw + c4l + e <= g.q_last) store_out<BF16OUT>(y, sbase + lterm + e, e4[e]);
                    }
                }
                if constexpr (STATS) {
                    if ((lane & 15) < 4) {
                        const int ch = mb * 16 + (lane & 15) * 4 + (lane >> 4);
                        atomicAdd(&sstat[ch], (double)k1);
                        atomicAdd(&sstat[128 + ch], (double)k2);
                    }
                }
            }
        }
        // the held element (next tile, chunk 1) becomes "current tile"
        gcur = gnxt;
        gnxt = geom_of(tile + 2 * gridDim.x);
        e_next = false;
    }
    vm_wait_keep<0>();                        // nothing of this workgroup stays in flight behind its end
    if constexpr (STATS) {
        __syncthreads();
        const int ch = cg * 128 + (tid & 127);
        if (ch < C) atomicAdd(&stats[(tid >> 7) * C + ch], sstat[tid]);
    }
}

struct T6Plan {
    int rows = 0, tiles_per_clip = 0;
    size_t lds = 0;
};

inline bool plan_t6(int Cin, int Cout, int T, int V, int K, int stride, int terms, T6Plan &pl) {
    if (K != KT6 || stride != 1 || T < 1 || V > 32) return false;
    if ((Cout % 128 != 0 && Cout != 64) || Cin % 32 != 0) return false;   // (an even number of 16-channel chunks)
    int dt = ceil_div(NP6 - 1, V);
    if (dt > T - 1) dt = T - 1;
    const int span = (dt + K) * V;
    const int rows = (span + 15) / 16 * 16;
    if (2 * rows > 4 * NT6) return false;                        // four (pixel, 8-channel) units per lane
    const size_t buf = (size_t)rows * PXB * (terms == 3 ? 2 : 1);
    pl.lds = RING6 + 2 * buf + (size_t)4 * EPI6;
    if (pl.lds > (size_t)kLdsBytes) return false;
    if ((size_t)(Cin > Cout ? Cin : Cout) * T * V * 4 >= ((size_t)1 << 31)) return false;   // per-clip buffer resources
    pl.rows = rows;
    pl.tiles_per_clip = ceil_div(T * V, NP6);
    return true;
}

}  // namespace

bool tcn_v6_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if (math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16) return false;
    T6Plan pl;
    return plan_t6(Cin, Cout, T, V, K, stride, math == STGCN_MATH_BF16X3 ? 3 : 1, pl);
}

// the STATS form needs 2 KiB of LDS on top (its per-workgroup fp64 sums)
bool tcn_v6_stats_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags) {
    const unsigned math = flags & STGCN_MATH_MASK;
    if ((math != STGCN_MATH_BF16X3 && math != STGCN_MATH_BF16) || (flags & (STGCN_OUT_BF16 | STGCN_OUT_NTVC))) return false;
    T6Plan pl;
    return plan_t6(Cin, Cout, T, V, K, stride, math == STGCN_MATH_BF16X3 ? 3 : 1, pl) &&
           pl.lds + 2 * 128 * sizeof(double) <= (size_t)kLdsBytes;
}

// true when launch_tcn_pack appends the pair-order copy of the weights for (Cin, Cout, K, math) — shape-independent part
// of tcn_v6_supported
bool tcn_v6_packs(int Cin, int Cout, int K, unsigned math) {
    return (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) && K == KT6 && Cin % 32 == 0 && (Cout % 128 == 0 || Cout == 64);
}

int launch_tcn_pack_pairs_padded(const float *W, const float *scale, void *Wq, int Cin, int Cout, hipStream_t st) {
    const int CoutP = (Cout + 127) / 128 * 128;
    const size_t total = (size_t)Cin * CoutP * KT6 * 2;
    hipLaunchKernelGGL(tcn_pack_pairs_padded_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, scale,
                       (unsigned short *)Wq, Cin, Cout, CoutP);
    STGCN_LAUNCH_CHECK("tcn_pack_pairs_padded_kernel");
    return STGCN_OK;
}

// stats (optional, 2*Cout doubles ZEROED by the caller): per-channel sum and sum of squares of the stored output
// (fp32, (N,C,T,V) layout only) — the training forward's batch statistics
int launch_tcn_v6(const float *x, const void *Wq, const float *shift, void *y, int N, int Cin, int Cout, int T, int V, int K,
                  int stride, unsigned flags, hipStream_t st, double *stats) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const int terms = math == STGCN_MATH_BF16X3 ? 3 : 1;
    const bool bf16out = (flags & STGCN_OUT_BF16) != 0;
    const float act_lo = (flags & STGCN_RAW) ? -__builtin_huge_valf() : 0.f;
    const int opt = (flags & STGCN_OUT_NTVC) ? OPT_OUT_NTVC : 0;
    T6Plan pl;
    if (!plan_t6(Cin, Cout, T, V, K, stride, terms, pl))
        return fail(STGCN_ERR_UNSUPPORTED, "tcn v6 kernel does not cover Cin=%d Cout=%d T=%d V=%d K=%d stride=%d", Cin, Cout, T,
                    V, K, stride);
    int dev = 0, num_cu = 256;
    STGCN_HIP_CHECK(hipGetDevice(&dev));
    STGCN_HIP_CHECK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int ntiles = N * pl.tiles_per_clip;
    const dim3 grid(ntiles < num_cu ? ntiles : num_cu, ceil_div(Cout, 128), 1);
    if (stats != nullptr) {
        if (bf16out || opt) return fail(STGCN_ERR_UNSUPPORTED, "tcn v6 kernel: channel statistics go with the fp32 (N,C,T,V) output");
        const size_t lds = pl.lds + 2 * 128 * sizeof(double);
        if (lds > (size_t)kLdsBytes) return fail(STGCN_ERR_UNSUPPORTED, "tcn v6 kernel: no LDS left for the channel statistics");
#define LAUNCH_T6S(TERMS)                                                                                         \
    do {                                                                                                          \
        auto kern = tcn_bf16_v6_kernel<TERMS, false, true>;                                                       \
        STGCN_HIP_CHECK(allow_lds(kern, lds));                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(NT6), lds, st, x, (const uint4 *)Wq, shift, y, Cin, Cout, T, V, pl.rows, \
                           pl.tiles_per_clip, ntiles, act_lo, opt, stats);                                        \
    } while (0)
        if (terms == 3) LAUNCH_T6S(3); else LAUNCH_T6S(1);
#undef LAUNCH_T6S
        STGCN_LAUNCH_CHECK("tcn_bf16_v6_kernel");
        return STGCN_OK;
    }
#define LAUNCH_T6(TERMS, B)                                                                                       \
    do {                                                                                                          \
        auto kern = tcn_bf16_v6_kernel<TERMS, B>;                                                                 \
        STGCN_HIP_CHECK(allow_lds(kern, pl.lds));                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(NT6), pl.lds, st, x, (const uint4 *)Wq, shift, y, Cin, Cout, T, V, pl.rows, \
                           pl.tiles_per_clip, ntiles, act_lo, opt, nullptr);                                      \
    } while (0)
    if (terms == 3) { if (bf16out) LAUNCH_T6(3, true); else LAUNCH_T6(3, false); }
    else { if (bf16out) LAUNCH_T6(1, true); else LAUNCH_T6(1, false); }
#undef LAUNCH_T6
    STGCN_LAUNCH_CHECK("tcn_bf16_v6_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
