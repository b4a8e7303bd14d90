// Weight gradient of the temporal conv in the one-wave-per-SIMD form (cf. tcn_bf16_v6.hip), for 17 <= V <= 24, stride 1,
// odd K <= 9, Cin % 32 == 0, Cout % 128 == 0 or 64.
//
//   dW[o][c][k] = sum_p dz[o][p] * x[c][p + (k - pad) * V]        (GEMM over PIXELS, see tcn_backward.hip)
//
// The eight-wave kernel (tcn_wgrad_mfma_kernel) stages a unit with all waves, then multiplies with all waves; priced in a
// diagnostic build, of 1.3 ms only 0.26 were MFMAs — the rest vector-memory issue, conversion, barriers and skeleton
// (DESIGN.md section 7).  Here:
//   * 256 threads, one wave per SIMD; wave w owns dz channel block w (32 channels) x 32 input channels x ALL taps:
//     9 accumulators of 32x32 (144 registers, pinned to the AGPR file); no tap groups, no wasted tenth tap;
//   * a workgroup walks whole CLIPS in units of TWO output frames (3 k-steps of 16 padded pixels, 81 MFMAs per wave);
//   * the input tile is a RING of 16 frame slots per channel row: a unit adds its two new frames (the other eight of its
//     ten-frame window are already there); four MFMA-free lead-in units per clip fill the ring;
//   * two dz tiles; staging of unit g+1 (convert + LDS stores) and the loads of unit g+2 are cut into pieces of a few
//     instructions that sit in the slots BEHIND the MFMAs of unit g — there is no staging phase; one barrier per unit;
//   * fragment reads of the next tap group are issued before the current group's MFMAs.
// Partial sums per workgroup, summed in a fixed order by sum_partials_kernel as before.
#include <type_traits>

#include "bf16_common.h"

namespace stgcn {

namespace {

using namespace bf16k;

constexpr int WQ_THREADS = 256;
constexpr int WQ_TFM = 2;        // output frames per unit
constexpr int WQ_UPF = 3;        // 8-pixel pieces per (padded) frame: Vp = 24
constexpr int WQ_VP = 24;
constexpr int WQ_RING = 16;      // frame slots of the input ring
constexpr int WQ_WIN = WQ_TFM + 8;   // window frames of a unit (taps 0 .. 8)
constexpr int WQ_LEAD = 4;       // lead-in units per clip: (WQ_WIN - WQ_TFM) / WQ_TFM
constexpr int WQ_PITCH_A = 112;  // bytes per dz row of a unit: 2 frames x 48 B, 16 B x odd
constexpr int WQ_PITCH_B = 784;  // bytes per input row: 16 slots x 48 B, 16 B x odd
constexpr int WQ_FRB = 48;       // bytes per frame slot

template <int I, int N, class F>
__device__ __forceinline__ void static_forq(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_forq<I + 1, N>(f);
    }
}

// NOB: 32-channel dz blocks per wave.  NOB = 1: wave w = block w, 32 input channels per workgroup.  NOB = 2: wave w = blocks
// 2 (w & 1), +1 and input-channel block w >> 1 of a 64-channel group — every input fragment read then feeds two blocks: with
// NOB = 1 the LDS moved 80 KiB per k-step for 864 cycles of MFMAs (640 cycles of its bandwidth before any conflict) and
// bound the kernel.
template <int TERMS, int NOB>
__global__ __launch_bounds__(WQ_THREADS) void tcn_wgrad_v6_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                                                                   float *__restrict__ part, int N, int Cin, int Cout, int T,
                                                                   int V, int K, int ipw /* items per workgroup */,
                                                                   int spc /* frame segments per clip */, int Tseg /* frames per
                                                                   segment, even */, unsigned long long *dbg, int dbg_mode) {
    // (no run-time ablation switches in here: inside the unrolled MFMA groups they tripled the kernel's time; diagnostic
    //  builds can stamp the clock at group boundaries: tools/stamps_wgrad.py)
#ifdef STGCN_ABLATION
#define WQ_STAMP(var) unsigned long long var = 0; if (dbg) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); }
#define WQ_ACC(slot, a, b) if (dbg) { tsum[slot] += (b) - (a); }
    unsigned long long tsum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int stamp_groups = dbg ? dbg_mode : 0;   // 1: also stamp every MFMA group (drains the LDS queue: perturbs)
#else
#define WQ_STAMP(var)
#define WQ_ACC(slot, a, b)
#endif
    extern __shared__ __attribute__((aligned(16))) char smq[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pad = (K - 1) / 2;
    int cg = blockIdx.x, zi = blockIdx.z;          // XCD-aware numbering: the channel groups of a split share an L2
    if ((gridDim.z & 7) == 0) {
        const int L = blockIdx.x + gridDim.x * blockIdx.z;
        zi = (L & 7) + 8 * (L / (8 * gridDim.x));
        cg = (L >> 3) % gridDim.x;
    }
    constexpr int CB = NOB;                        // 32-channel input blocks per workgroup
    const int c0 = cg * 32 * CB, o0 = blockIdx.y * 128;
    const int ob0 = NOB == 1 ? wave : (wave & 1) * 2, cb = NOB == 1 ? 0 : wave >> 1;
    // LDS: dz tile 0 (hi | lo) | dz tile 1 (hi | lo) | input ring (hi | lo)
    constexpr int AIMG = 128 * WQ_PITCH_A, ATILE = 2 * AIMG, BIMG = 32 * CB * WQ_PITCH_B;
    char *Bring = smq + 2 * ATILE;
    // A work ITEM is a frame segment of a clip (spc segments of Tseg frames; spc = 1: the whole clip).  Small batches — the
    // deeper layers' 64-clip steps put 128 workgroups on 256 CUs — are cut into segments; a segment's lead-in units load the
    // four frames in front of it (real frames inside the clip, zeros before its first) against a zero dz tile, as a clip's do.
    const int upc = Tseg / WQ_TFM + WQ_LEAD;       // units per item incl. the lead-in
    const int items = N * spc;
    const int clip0 = zi * ipw, clip1 = min(clip0 + ipw, items);        // (item range; the names date from whole-clip items)
    const int nun = (clip1 > clip0 ? clip1 - clip0 : 0) * upc;         // units of this workgroup

    f32x16 acc[NOB][9], acc8[NOB];                // (acc8: tap 8 of the two-block form, kept in VGPRs — see the MFMA loop)
#pragma unroll
    for (int b = 0; b < NOB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc8[b][r] = 0.f;
#pragma unroll
    for (int b = 0; b < NOB; ++b)
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][k][r] = 0.f;

    // ---- staging units of this lane: three dz pieces (row, frame, piece) and — lanes < 192 — one input piece ----------
    int a_row[3], a_tt[3], a_uq[3], a_lds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = tid + i * WQ_THREADS;        // < 768 = 128 rows x 6 pieces
        a_row[i] = e / 6;
        const int q = e - a_row[i] * 6;
        a_tt[i] = q / 3;
        a_uq[i] = q - a_tt[i] * 3;
        a_lds[i] = a_row[i] * WQ_PITCH_A + q * 16;
    }
    int b_row[CB], b_ff[CB], b_uq[CB], b_nv[CB];
    bool b_live[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int e = tid + j * WQ_THREADS;        // < 192 * CB = 32 * CB rows x 6 pieces
        b_live[j] = e < 192 * CB;
        const int be = min(e, 192 * CB - 1);
        b_row[j] = be / 6;
        b_ff[j] = (be - b_row[j] * 6) / 3;
        b_uq[j] = be - b_row[j] * 6 - b_ff[j] * 3;
        b_nv[j] = V - b_uq[j] * 8;              // valid columns of the piece
    }
    float pa[3][8], pb[CB][8];
    const unsigned clipA = (unsigned)((size_t)Cout * T * V * 4), clipB = (unsigned)((size_t)Cin * T * V * 4);
    constexpr unsigned OOB = 0x7ffffff0u;
    // unit g of this workgroup -> (clip, first output frame t0; t0 < 0: lead-in)
    // (readfirstlane: the values are uniform, but unless hipcc KNOWS it every buffer load below becomes a waterfall loop over
    //  the lanes' resource descriptors — 129 v_readfirstlane and 32 loops per unit in the first build)
    auto unit_clip = [&](int g) { return __builtin_amdgcn_readfirstlane((clip0 + g / upc) / spc); };
    auto unit_ts = [&](int g) { return __builtin_amdgcn_readfirstlane(((clip0 + g / upc) % spc) * Tseg); };   // segment's first frame
    auto unit_t0 = [&](int g) { return __builtin_amdgcn_readfirstlane(unit_ts(g) + (g % upc - WQ_LEAD) * WQ_TFM); };
    auto unit_dz = [&](int g) { return g % upc >= WQ_LEAD; };           // a lead-in unit multiplies a zero dz tile
    // dz pieces are NOT masked beyond column V: the padded columns of a frame only ever meet the same columns of an input frame,
    // which are zeroed below (what dz holds there is the next row's first pixels or, past the clip, the resource's zeros).
    auto a_off = [&](int i, bool live, int t0, int te) -> unsigned {    // live: the unit exists and is not a lead-in; te: segment end
        const int t = t0 + a_tt[i];
        return (live && t < te) ? (unsigned)((((o0 + a_row[i]) * T + t) * V + a_uq[i] * 8) * 4) : OOB;
    };
    auto a_rsrc = [&](int n) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dz + (size_t)n * Cout * T * V), 0, clipA, 0x00020000);
    };
    auto b_off = [&](int q, bool live, int t0) -> unsigned {
        const int f = t0 - pad + (WQ_WIN - WQ_TFM) + b_ff[q];         // the unit's new frames: window frames 8, 9
        return (live && f >= 0 && f < T) ? (unsigned)((((c0 + b_row[q]) * T + f) * V + b_uq[q] * 8) * 4) : OOB;
    };
    auto b_rsrc = [&](int n) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)n * Cin * T * V), 0, clipB, 0x00020000);
    };
    auto load4 = [&](float *dst, __amdgpu_buffer_rsrc_t r, unsigned off) {      // (hipcc merges the four into one dwordx4)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off + 4 * j, 0, 0));
    };
    auto load_a = [&](int i, bool live, int n, int t0, int te) {
        const unsigned off = a_off(i, live, t0, te);
        load4(&pa[i][0], a_rsrc(n), off);
        load4(&pa[i][4], a_rsrc(n), off + 16);
    };
    auto load_b = [&](bool live, int n, int t0) {
#pragma unroll
        for (int q = 0; q < CB; ++q) {
            const unsigned off = b_off(q, live, t0);
            load4(&pb[q][0], b_rsrc(n), off);
            load4(&pb[q][4], b_rsrc(n), off + 16);
        }
    };
    auto mask_b = [&](int q) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[q][j] = (j < b_nv[q]) ? pb[q][j] : 0.f;
    };
    auto store_a = [&](int i, char *atile) {
        uint4 hi, lo;
        split8(pa[i], hi, lo);
        *reinterpret_cast<uint4 *>(atile + a_lds[i]) = hi;
        if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(atile + AIMG + a_lds[i]) = lo;
    };
    // Ring position of a unit = 2 * (its index in this workgroup): it runs on across clips, so the slots a unit's new frames
    // go to (positions 2g+8, 2g+9) never lie in the window of the unit being multiplied (2(g-1) .. 2(g-1)+9).
    auto store_b = [&](int g) {                  // into the ring slots of unit g's new frames
#pragma unroll
        for (int q = 0; q < CB; ++q) {
            const int slot = (2 * g + (WQ_WIN - WQ_TFM) + b_ff[q]) & (WQ_RING - 1);
            uint4 hi, lo;
            mask_b(q);
            split8(pb[q], hi, lo);
            if (b_live[q]) {
                char *p = Bring + b_row[q] * WQ_PITCH_B + slot * WQ_FRB + b_uq[q] * 16;
                *reinterpret_cast<uint4 *>(p) = hi;
                if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(p + BIMG) = lo;
            }
        }
    };

    // ---- fragment addressing ---------------------------------------------------------------------------------------
    // k-step ks, lane half h: piece q = 2 ks + h of the unit's 6 -> frame tt = q / 3, piece in frame uq = q % 3
    const int h = lane >> 5;
    const int a_lane = (ob0 * 32 + (lane & 31)) * WQ_PITCH_A + h * 16;           // + ks * 32  (+ 32 rows for the second block)
    int b_lane[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) b_lane[ks] = (cb * 32 + (lane & 31)) * WQ_PITCH_B + ((2 * ks + h) % 3) * 16;

    // ---- prologue: ring zeroed (lead-in units multiply it by a zero dz tile: it must hold finite numbers), unit 0 staged,
    //      unit 1 in the registers
    for (int e = tid; e < 2 * BIMG / 16; e += WQ_THREADS) reinterpret_cast<uint4 *>(Bring)[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (nun > 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) load_a(i, unit_dz(0), unit_clip(0), unit_t0(0), min(T, unit_ts(0) + Tseg));
        load_b(true, unit_clip(0), unit_t0(0));
#pragma unroll
        for (int i = 0; i < 3; ++i) store_a(i, smq);
        store_b(0);
        const bool l1 = 1 < nun;
        const int n1 = __builtin_amdgcn_readfirstlane(l1 ? unit_clip(1) : 0), t1 = __builtin_amdgcn_readfirstlane(l1 ? unit_t0(1) : -1);
#pragma unroll
        for (int i = 0; i < 3; ++i) load_a(i, l1 && unit_dz(1), n1, t1, min(T, unit_ts(1) + Tseg));
        load_b(l1, n1, t1);
    }
    __syncthreads();

    // the first group's input fragments of a unit are read during the last group of the unit before (frames 0 .. 2 of a window
    // were staged three units earlier: no barrier between them and these reads is needed)
    uint4 b0h[3], b0l[3];
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
        b0h[kk] = *reinterpret_cast<const uint4 *>(Bring + b_lane[0] + kk * WQ_FRB);
        b0l[kk] = b0h[kk];
        if constexpr (TERMS == 3) b0l[kk] = *reinterpret_cast<const uint4 *>(Bring + BIMG + b_lane[0] + kk * WQ_FRB);
    }
    int u2 = 2, c2 = clip0;                       // unit g+2: index within its item (upc >= 5) and item — carried, not divided
    for (int g = 0; g < nun; ++g) {
        WQ_STAMP(t_u0)
        const bool l2 = g + 2 < nun;              // the unit whose loads are issued during this one (all scalar)
        // (readfirstlane: hipcc does not see that the carried counters are uniform — without it every load is a waterfall loop)
        const int ts2 = __builtin_amdgcn_readfirstlane((c2 % spc) * Tseg), te2 = min(T, ts2 + Tseg);
        const bool la2 = l2 && u2 >= WQ_LEAD;     // its dz tile is real (not a lead-in)
        const int n2 = __builtin_amdgcn_readfirstlane(l2 ? c2 / spc : 0), t2 = __builtin_amdgcn_readfirstlane(l2 ? ts2 + (u2 - WQ_LEAD) * WQ_TFM : -1);
        char *acur = smq + (g & 1) * ATILE, *anxt = smq + ((g + 1) & 1) * ATILE;
        // Staging of unit g+1 (convert + LDS stores) and the loads of unit g+2, cut into PIECES of a few instructions that go
        // into the slots after the MFMAs (an MFMA occupies the pipe for 32 cycles; instructions placed right behind it issue
        // meanwhile — clumped between the groups they left the pipe idle for a quarter of the unit).  Piece list, in order:
        //   per dz register set i = 0..2 (13 pieces): pack pair 0..3 (hi, lo) | write hi | write lo | offset | load 0..3 | load 4..7
        //   per input set q (10 pieces):              pack pair 0..3 (hi with columns >= V zeroed, lo) | write hi | write lo
        //   per input set q (3 pieces):               offset | load 0..3 | load 4..7
        unsigned sh[4] = {0, 0, 0, 0}, sl[4] = {0, 0, 0, 0}, goff = OOB;
        const __amdgpu_buffer_rsrc_t ra2 = a_rsrc(n2), rb2 = b_rsrc(n2);
        constexpr int PA_N = 13, PB_N = 10;       // pieces per dz set / per input set (without its 3 load pieces)
        constexpr int NP = 3 * PA_N + (PB_N + 3) * CB;
        // a pack is two pieces (an MFMA covers 32 cycles = eight VALU issues; a whole pair was ten instructions)
        auto pack_hi = [&](float *v, int pr, int nv) __attribute__((always_inline)) {
            v[2 * pr] = (2 * pr < nv) ? v[2 * pr] : 0.f;
            v[2 * pr + 1] = (2 * pr + 1 < nv) ? v[2 * pr + 1] : 0.f;
            sh[pr] = pack_bf16x2(v[2 * pr], v[2 * pr + 1]);
        };
        auto pack_lo = [&](const float *v, int pr) __attribute__((always_inline)) {
            if constexpr (TERMS == 3) sl[pr] = pack_bf16x2(v[2 * pr] - bf16_lo_to_f32(sh[pr]), v[2 * pr + 1] - bf16_hi_to_f32(sh[pr]));
        };
        auto piece = [&](auto p_c) __attribute__((always_inline)) {
            constexpr int P = decltype(p_c)::value;
#if defined(STGCN_ABLATION) && (defined(WQ_NOPACK) || defined(WQ_NOWRITE) || defined(WQ_NOLOADS))
            {   // diagnostic variants (results wrong): price one kind of piece
                constexpr bool isA = P < 3 * PA_N, isB = !isA && P < 3 * PA_N + PB_N * CB;
                constexpr int rr = isA ? P % PA_N : (isB ? (P - 3 * PA_N) % PB_N : 11);
#ifdef WQ_NOPACK
                if constexpr (rr < 8) return;
#endif
#ifdef WQ_NOWRITE
                if constexpr (rr == 8 || rr == 9) return;
#endif
#ifdef WQ_NOLOADS
                if constexpr (rr >= 10) return;
#endif
            }
#endif
            if constexpr (P < 3 * PA_N) {
                constexpr int i = P / PA_N, r = P % PA_N;
                if constexpr (r < 8) { if constexpr (r % 2 == 0) pack_hi(pa[i], r / 2, 8); else pack_lo(pa[i], r / 2); }
                else if constexpr (r == 8) *reinterpret_cast<uint4 *>(anxt + a_lds[i]) = make_uint4(sh[0], sh[1], sh[2], sh[3]);
                else if constexpr (r == 9) { if constexpr (TERMS == 3) *reinterpret_cast<uint4 *>(anxt + AIMG + a_lds[i]) = make_uint4(sl[0], sl[1], sl[2], sl[3]); }
                else if constexpr (r == 10) goff = a_off(i, la2, t2, te2);
                else if constexpr (r == 11) load4(&pa[i][0], ra2, goff);
                else load4(&pa[i][4], ra2, goff + 16);
            } else if constexpr (P < 3 * PA_N + PB_N * CB) {
                constexpr int q = (P - 3 * PA_N) / PB_N, r = (P - 3 * PA_N) % PB_N;
                if constexpr (r < 8) { if constexpr (r % 2 == 0) pack_hi(pb[q], r / 2, b_nv[q]); else pack_lo(pb[q], r / 2); }
                else {
                    // ring slots of unit g+1's new frames (past the last unit they take zeros nobody reads)
                    const int slot = (2 * (g + 1) + (WQ_WIN - WQ_TFM) + b_ff[q]) & (WQ_RING - 1);
                    char *dp = Bring + b_row[q] * WQ_PITCH_B + slot * WQ_FRB + b_uq[q] * 16;
                    if constexpr (r == 8) { if (b_live[q]) *reinterpret_cast<uint4 *>(dp) = make_uint4(sh[0], sh[1], sh[2], sh[3]); }
                    else if constexpr (TERMS == 3) { if (b_live[q]) *reinterpret_cast<uint4 *>(dp + BIMG) = make_uint4(sl[0], sl[1], sl[2], sl[3]); }
                }
            } else {
                constexpr int q = (P - 3 * PA_N - PB_N * CB) / 3, r = (P - 3 * PA_N - PB_N * CB) % 3;
                if constexpr (r == 0) goff = b_off(q, l2, t2);
                else if constexpr (r == 1) load4(&pb[q][0], rb2, goff);
                else load4(&pb[q][4], rb2, goff + 16);
            }
        };
        {   // (lead-in units run the MFMAs as well, on an all-zero dz tile: a branch around them made hipcc shuffle the
            //  144 accumulator registers at the join — 700 copies per unit)
            // ring offsets of the window frames m = 0 .. 9 of this unit (scalar)
            int so[10];
#pragma unroll
            for (int m = 0; m < 10; ++m) so[m] = ((2 * g + m) & (WQ_RING - 1)) * WQ_FRB;
            auto b_addr = [&](auto ks_c, auto k_c) -> const char * {
                constexpr int ks = decltype(ks_c)::value, k = decltype(k_c)::value;
                // frame of the piece: ks = 0 -> 0, ks = 2 -> 1, ks = 1 -> the lane half
                const int s_lo = so[k], s_hi = so[k + 1];
                const int sel = ks == 0 ? s_lo : (ks == 2 ? s_hi : (h ? s_hi : s_lo));
                return Bring + b_lane[ks] + sel;
            };
            uint4 bh[2][3], bl[2][3];
            uint4 ahs[2][NOB], als[2][NOB];       // dz fragments of a k-step, read one k-step ahead
#pragma unroll
            for (int b = 0; b < NOB; ++b) {
                ahs[0][b] = *reinterpret_cast<const uint4 *>(acur + a_lane + b * 32 * WQ_PITCH_A);
                als[0][b] = ahs[0][b];
                if constexpr (TERMS == 3) als[0][b] = *reinterpret_cast<const uint4 *>(acur + AIMG + a_lane + b * 32 * WQ_PITCH_A);
            }
            static_forq<0, 9>([&](auto s_c) {                   // 9 MFMA groups: (k-step, tap group)
                constexpr int s = decltype(s_c)::value, ks = s / 3, grp = s % 3, set = s & 1;
#ifdef STGCN_ABLATION
                unsigned long long t_g0 = 0;
                if (stamp_groups) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_g0) :: "memory");
#endif
                // fillers of this group: the next group's input fragments (6 reads), the next k-step's dz fragments (2 per
                // block, at the k-step's first group), then this group's share of the staging pieces
                constexpr int NRB = 3 * (TERMS == 3 ? 2 : 1);       // (group 8: the NEXT unit's group 0)
                constexpr int NRA = (grp == 0 && ks + 1 < 3) ? NOB * (TERMS == 3 ? 2 : 1) : 0;
                constexpr int P0 = s * NP / 9, P1 = (s + 1) * NP / 9;
                constexpr int NF = NRB + NRA + (P1 - P0);
                constexpr int NM = (TERMS == 3 ? 3 : 1) * 3 * NOB;
                auto filler = [&](auto f_c) __attribute__((always_inline)) {
                    constexpr int f = decltype(f_c)::value;
#if defined(STGCN_ABLATION) && defined(WQ_NOREADS)       // diagnostic variants (results wrong): price the fillers
                    if constexpr (f < NRB + NRA) return;
#endif
#if defined(STGCN_ABLATION) && defined(WQ_NOPIECES)
                    if constexpr (f >= NRB + NRA) return;
#endif
                    if constexpr (f < NRB) {
                        constexpr int kk = f % 3, lo = f / 3, s1 = s + 1;
                        if constexpr (s1 == 9) {              // next unit: its window frame kk is this unit's frame kk + 2
                            const char *p = Bring + b_lane[0] + so[kk + 2];
                            if constexpr (lo == 0) {
                                b0h[kk] = *reinterpret_cast<const uint4 *>(p);
                                if constexpr (TERMS != 3) b0l[kk] = b0h[kk];
                            } else b0l[kk] = *reinterpret_cast<const uint4 *>(p + BIMG);
                        } else {
                            const char *p = b_addr(std::integral_constant<int, s1 / 3>{}, std::integral_constant<int, (s1 % 3) * 3 + kk>{});
                            if constexpr (lo == 0) bh[s1 & 1][kk] = *reinterpret_cast<const uint4 *>(p);
                            else bl[s1 & 1][kk] = *reinterpret_cast<const uint4 *>(p + BIMG);
                        }
                    } else if constexpr (f < NRB + NRA) {
                        constexpr int b = (f - NRB) % NOB, lo = (f - NRB) / NOB, k1 = ks + 1;
                        const char *p = acur + a_lane + b * 32 * WQ_PITCH_A + k1 * 32;
                        if constexpr (lo == 0) {
                            ahs[k1 & 1][b] = *reinterpret_cast<const uint4 *>(p);
                            if constexpr (TERMS != 3) als[k1 & 1][b] = ahs[k1 & 1][b];
                        } else als[k1 & 1][b] = *reinterpret_cast<const uint4 *>(p + AIMG);
                    } else {
                        piece(std::integral_constant<int, P0 + f - NRB - NRA>{});
                    }
                };
                // one MFMA of term `term` (0: hi x lo, 1: lo x hi, 2: hi x hi) of (block b, tap grp*3 + kk)
                auto mfma1 = [&](auto kk_c, auto b_c, auto term_c) __attribute__((always_inline)) {
                    constexpr int kk = decltype(kk_c)::value, b = decltype(b_c)::value, term = decltype(term_c)::value, tap = grp * 3 + kk;
                    const uint4 av = term == 1 ? als[ks & 1][b] : ahs[ks & 1][b];
                    const uint4 bv = s == 0 ? (term == 0 ? b0l[kk] : b0h[kk]) : (term == 0 ? bl[set][kk] : bh[set][kk]);
                    if constexpr (NOB == 2 && tap == 8) {
                        // 2 x 9 accumulators are 288 registers, 32 more than the AGPR file: through the builtin hipcc rotated
                        // blocks between the two files (850 copies per unit).  The last tap's accumulators live in VGPRs,
                        // multiplied by VGPR-form MFMAs (inline asm; the only other reader is the epilogue).
                        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
                        const u32x4 a4 = __builtin_bit_cast(u32x4, av), b4 = __builtin_bit_cast(u32x4, bv);
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc8[b]) : "v"(a4), "v"(b4));
                    } else {
                        acc[b][tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[b][tap], 0, 0, 0);
                    }
                };
                // slot m: MFMA m — order (term, tap, block): consecutive MFMAs go to different accumulators (the three terms of
                // one block are a dependent chain) — then fillers [m NF / NM, (m+1) NF / NM)
                static_forq<0, NM>([&](auto m_c) {
                    constexpr int m = decltype(m_c)::value;
                    constexpr int term = TERMS == 3 ? m / (3 * NOB) : 2, kk = (m / NOB) % 3, b = m % NOB;
                    mfma1(std::integral_constant<int, kk>{}, std::integral_constant<int, b>{}, std::integral_constant<int, term>{});
                    static_forq<m * NF / NM, (m + 1) * NF / NM>([&](auto f_c) { filler(f_c); });
                    __builtin_amdgcn_sched_barrier(0);
                });
#ifdef STGCN_ABLATION
                if (stamp_groups) {
                    unsigned long long t_g1;
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_g1) :: "memory");
                    tsum[s] += t_g1 - t_g0;
                }
#endif
            });
        }
        WQ_STAMP(t_u1)
        __syncthreads();                          // unit g+1 is staged; tile g & 1 and the ring slots behind the window are free
        if (++u2 == upc) { u2 = 0; ++c2; }
        WQ_STAMP(t_u2)
        WQ_ACC(9, t_u0, t_u1)                     // the unit's work
        WQ_ACC(10, t_u1, t_u2)                    // barrier wait
    }
#ifdef STGCN_ABLATION
    if (dbg && lane == 0 && zi < 8 && cg == 0 && blockIdx.y == 0)
        for (int i = 0; i < 12; ++i) dbg[(zi * 4 + wave) * 12 + i] = tsum[i];
#endif

    // D[row = o][col = c]: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float *dst = part + (size_t)zi * Cout * Cin * K;
#pragma unroll
    for (int b = 0; b < NOB; ++b) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (k < K) {                          // (taps beyond K were computed on real (finite) frames and are dropped)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = o0 + (ob0 + b) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float val = (NOB == 2 && k == 8) ? acc8[b][r] : acc[b][k][r];
                    if (o < Cout) dst[((size_t)o * Cin + c0 + cb * 32 + (lane & 31)) * K + k] = val;   // (rows >= Cout: dz read as zeros)
                }
            }
        }
    }
}

}  // namespace

bool tcn_wgrad_v6_supported(int N, int Cin, int Cout, int T, int V, int K, int stride) {
    if (stride != 1 || K < 1 || K > 9 || (K & 1) == 0) return false;
    if ((Cout % 128 != 0 && Cout != 64) || Cin % 32 != 0) return false;
    if (V < 17 || V > WQ_VP || N < 1 || T < 1) return false;
    if ((size_t)(Cin > Cout ? Cin : Cout) * T * V * 4 >= ((size_t)1 << 31)) return false;   // per-clip buffer resources
    return true;
}

// two dz blocks per wave (64 input channels per workgroup) where the shape allows  (diagnostic builds: STGCN_ABLATE=4 off)
static bool wgrad_v6_two_blocks(int Cin, int Cout) { return Cin % 64 == 0 && Cout % 128 == 0 && !(ablate_mask() & 4); }

// work items = (clip, frame segment): whole clips when there are enough of them for one workgroup per CU, else segments of at
// least 16 frames (a segment pays four lead-in units)
struct WqItems { int splits, spc, Tseg, ipw; };
static WqItems wgrad_v6_items(int N, int Cin, int Cout, int T) {
    const int wgs = (Cin / (wgrad_v6_two_blocks(Cin, Cout) ? 64 : 32)) * ceil_div(Cout, 128);
    int want = 256 / wgs;                         // about one workgroup per CU
    if (want < 1) want = 1;
    int spc = 1;
    // (segments only in the one-block form: with two 32-channel blocks per wave the segmented kernel measured 7 % SLOWER on all
    //  256 CUs than whole clips on 128 — 221 vs 206 us at 128 channels, 64 clips — while the one-block form gained 30 %)
    if (N < want && !wgrad_v6_two_blocks(Cin, Cout)) {
        spc = ceil_div(want, N);
        const int cap = T / 16 > 1 ? T / 16 : 1;
        if (spc > cap) spc = cap;
    }
    WqItems it;
    it.Tseg = ceil_div(ceil_div(T, spc), WQ_TFM) * WQ_TFM;
    it.spc = ceil_div(T, it.Tseg);
    const int items = N * it.spc;
    it.splits = want < items ? want : items;
    it.ipw = ceil_div(items, it.splits);
    it.splits = ceil_div(items, it.ipw);          // (no empty workgroups: every partial slice is written)
    return it;
}

int tcn_wgrad_v6_splits(int N, int Cin, int Cout, int T) { return wgrad_v6_items(N, Cin, Cout, T).splits; }

// partial sums: part[splits][Cout][Cin][K] (the caller sums them in a fixed order)
int launch_tcn_wgrad_v6(const float *dz, const float *x, float *part, int N, int Cin, int Cout, int T, int V, int K, unsigned flags,
                        hipStream_t st) {
    const unsigned math = flags & STGCN_MATH_MASK;
    const WqItems it = wgrad_v6_items(N, Cin, Cout, T);
    const bool two = wgrad_v6_two_blocks(Cin, Cout);
    const dim3 grid(Cin / (two ? 64 : 32), ceil_div(Cout, 128), it.splits);
    const size_t lds = (size_t)2 * 2 * 128 * WQ_PITCH_A + (size_t)2 * 32 * (two ? 2 : 1) * WQ_PITCH_B;
#define LAUNCH_WQ(TERMS, NOB)                                                                                         \
    do {                                                                                                              \
        STGCN_HIP_CHECK(allow_lds((tcn_wgrad_v6_kernel<TERMS, NOB>), lds));                                           \
        hipLaunchKernelGGL((tcn_wgrad_v6_kernel<TERMS, NOB>), grid, dim3(WQ_THREADS), lds, st, dz, x, part, N, Cin, Cout, T, V, K, \
                           it.ipw, it.spc, it.Tseg, debug_buffer(), (ablate_mask() & 64) ? 1 : 0);                                                                                                    \
    } while (0)
    if (math == STGCN_MATH_BF16X3) { if (two) LAUNCH_WQ(3, 2); else LAUNCH_WQ(3, 1); }
    else { if (two) LAUNCH_WQ(1, 2); else LAUNCH_WQ(1, 1); }
#undef LAUNCH_WQ
    STGCN_LAUNCH_CHECK("tcn_wgrad_v6_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
