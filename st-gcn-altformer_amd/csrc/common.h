// Shared host/device helpers for libstgcn_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/stgcn_hip.h"

// save_stats[4*Cout + 126] (one of the two spare floats behind the 63 moment doubles): written by the moments-path forward,
// cleared by the materialising path, checked by the moment-form backward (which poisons its outputs with NaN without it).
#define STGCN_MOMENTS_MAGIC 0x4D4F4D31u   /* "MOM1" */
#define STGCN_MOMENTS_MARK_SLOT(Cout) (4 * (Cout) + 126)

namespace stgcn {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kLdsBytes = 160 * 1024;  // LDS per CU (and max per workgroup) on gfx950

// thread-local last-error text (stgcn_last_error)
void set_error(const char *fmt, ...);
int fail(stgcn_status st, const char *fmt, ...);

#define STGCN_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return stgcn::fail(STGCN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define STGCN_LAUNCH_CHECK(name)                                                           \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess)                                                              \
            return stgcn::fail(STGCN_ERR_HIP, "launch of %s failed: %s", name,             \
                               hipGetErrorString(_e));                                     \
    } while (0)

// phases switched off by a diagnostic build (see tcn_conv.hip); always 0 in the shipped library
// kernel option bits that travel in the high half of the kernels' `abl` argument (the low half is the ablation mask)
constexpr int OPT_OUT_NTVC = 1 << 16;  // store the output as (N,T,V,C) instead of (N,C,T,V)

static inline int ablate_mask() {
#ifdef STGCN_ABLATION
    const char *e = getenv("STGCN_ABLATE");
    return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

// diagnostic builds: device buffer for in-kernel cycle stamps (address in env STGCN_DBG_PTR), else NULL
static inline unsigned long long *debug_buffer() {
#ifdef STGCN_ABLATION
    const char *e = getenv("STGCN_DBG_PTR");
    return e ? reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0)) : nullptr;
#else
    return nullptr;
#endif
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Opt a kernel in to more than 64 KiB of dynamic LDS.
template <typename K>
static inline hipError_t allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Iteration shared by the elementwise kernels: workgroup (blockIdx.x, c = blockIdx.y) owns clips
// [n_lo, n_hi) of channel c; a clip's plane of the channel is contiguous, so there is no division in the loop and,
// when the plane is a multiple of 4 floats, every access is a 16-byte one.
struct ChannelRows {
    int n_lo, n_hi;
    bool vec;
    __device__ ChannelRows(int N, size_t plane) {
        const int per = (N + (int)gridDim.x - 1) / (int)gridDim.x;
        n_lo = blockIdx.x * per;
        n_hi = min(N, n_lo + per);
        vec = (plane & 3) == 0;
    }
};

// grid.x of those kernels: ~64 K elements per workgroup and channel, at most one workgroup per clip; tensors too small to
// give a thousand workgroups that way (the deeper layers' 64-clip steps: 64 - 128 workgroups of 256 threads for the whole
// chip, 0.5 TB/s) are cut finer, down to 8 K elements per workgroup and channel.
static inline int bn_chunks(int N, size_t plane, int C = 0) {
    int chunks = (int)(((size_t)N * plane + 65535) / 65536);
    if (C > 0 && chunks * C < 1024) {
        const int fine = (int)(((size_t)N * plane + 8191) / 8192);
        chunks = (1024 + C - 1) / C;
        if (chunks > fine) chunks = fine;
    }
    if (chunks > N) chunks = N;
    if (chunks > 64) chunks = 64;
    return chunks < 1 ? 1 : chunks;
}

// ---------------------------------------------------------------------------------------
// launchers implemented in the kernel translation units (all enqueue on `st`, return status)
// ---------------------------------------------------------------------------------------
int launch_bn_fold(const float *w, const float *b, const float *rm, const float *rv,
                   const float *cb, float eps, float *scale, float *shift, int C, hipStream_t st);

int launch_attention(const float *x, const float *A_eff, const float *Wa, const float *ba,
                     const float *Wb, const float *bb, float *P, float *feat, int N, int Cin, int T,
                     int V, int inter_c, int S, hipStream_t st, bool x_ntvc = false, float *xcopy = nullptr,
                     void *pfrag = nullptr,    // pfrag: (N,12,64) x 16 B attention B-fragments instead of features
                     int pf_v0 = 0,            // > 0: wide frames, (N,48,64) x 16 B fragments for the joint split V0 | V - V0
                     float *ybound = nullptr); // (N,4): max|x| and max|x| * largest column abs-sum of P_s per clip (KF7's scales)

int launch_agcn_expand(const float *x, const float *P, const float *Wd, const float *bd,
                       const float *Wdown, const float *bdown, const float *bn_scale,
                       const float *bn_shift, const float *down_scale, const float *down_shift,
                       float *y, int N, int Cin, int Cout, int T, int V, int S, int mode, hipStream_t st);

// temporal conv
size_t tcn_packed_bytes(int Cin, int Cout, int K, unsigned flags);          // everything launch_tcn_pack writes
size_t tcn_packed_single_bytes(int Cin, int Cout, int K, unsigned flags);   // the first layout alone (a pair-order copy may follow)
int launch_tcn_pack(const float *W, const float *scale, void *Wp, int Cin, int Cout, int K,
                    unsigned flags, hipStream_t st);
int launch_tcn(const float *x, const void *Wp, const float *shift, void *y, int N, int Cin,
               int Cout, int T, int V, int K, int stride, unsigned flags, hipStream_t st);

// bf16 matrix-core variants (tcn_bf16.hip)
bool bf16_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags, bool fused);
bool bf16_packs(int Cin, int Cout, unsigned math);
int launch_tcn_pack_bf16(const float *W, const float *scale, void *Wp, int Cin, int Cout, int K, hipStream_t st);
int launch_tcn_bf16(const float *x, const float *P, const float *W12, const void *Wp, const float *shift, void *y,
                    int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags, bool fused,
                    hipStream_t st);

bool tcn_mfma_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags);
bool stem_fused_supported(int Cin, int C, int T, int V, int K, int S, unsigned flags);

// large-tile persistent bf16 stem (stem_bf16_v4.hip); consumes the feature tensor the attention kernel emits
bool attention_emits_features(int Cin, int V, int S);
bool stem_v4_supported(int Cin, int C, int T, int V, int K, int S, unsigned flags);
// true: the kernel computes the graph-conv features itself from x and the attention fragments (`feat` then points at the
// (N,12,64) x 16 B fragments the attention kernel wrote); false: `feat` is the (N,T*V) x 64 B feature tensor
bool stem_v4_features_in_kernel(int C, int T, int V, int K, unsigned flags);
int launch_stem_v4(const float *x, bool x_ntvc, const float *feat, const void *prep_w12, const void *Wp, const float *shift,
                   void *out, int N, int C, int T, int V, int K, unsigned flags, hipStream_t st);  // honours STGCN_OUT_NTVC

// the same tile with ONE WAVE PER SIMD on v_mfma_f32_16x16x32_bf16 (stem_bf16_v6.hip: 256 threads, a wave owns all 128
// channels of 64 pixels); reads the temporal weights in its own pair order, which stgcn_stem_prepare appends to the prep
// blob behind the 32x32x16 packing
bool stem_v6_supported(int C, int T, int V, int K, unsigned flags);
// ... and for wide frames (32 < V <= 64, V even: the two-hand graph) the same kernel over the two joint halves [0, V0) and
// [V0, V), each handled like a narrow clip (stem_bf16_v6w.hip); the attention kernel then emits 48 fragments per clip
static inline int stem_wide_split(int V) {     // V0 (a multiple of 4: 16-byte aligned half rows), or 0 when V does not split
    if (V <= 32 || V > 64 || (V & 1)) return 0;
    const int v0 = (V / 2 + 3) / 4 * 4;
    return (v0 <= 32 && V - v0 >= 1) ? v0 : 0;
}
bool stem_v6w_supported(int C, int T, int V, int K, unsigned flags);
// ... and KF7 (stem_f16mx.hip, STGCN_STEM_F16MX): KF6 with fp16 x fp16 + two scaled-e4m3 residual products per k-step group
bool stem_f16mx_supported(int C, int T, int V, int K, unsigned flags);
size_t stem_f16mx_prep_bytes(int C, int K);
int launch_stem_f16mx_prepare(const float *W12, const float *Wt, const float *t_scale, void *dst, int C, hipStream_t st);
int launch_stem_f16mx(const float *x, bool x_ntvc, const void *pfrag, const void *bounds, const void *prep_w12, const void *mx_blob,
                      const float *shift, void *out, int N, int C, int T, int V, int K, unsigned flags, hipStream_t st);
int launch_stem_v6w(const float *x, bool x_ntvc, const void *pfrag, const void *prep_w12, const void *Wq, const float *shift,
                    void *out, int N, int C, int T, int V, int K, unsigned flags, hipStream_t st);
int launch_tcn_pack_bf16_pairs(const float *W, const float *scale, void *Wq, int Cin, int Cout, hipStream_t st);
int launch_stem_v6(const float *x, bool x_ntvc, const void *pfrag, const void *prep_w12, const void *Wq, const float *shift,
                   void *out, int N, int C, int T, int V, int K, unsigned flags, hipStream_t st);

// stand-alone temporal conv in the large-tile persistent form (stem_bf16_v4.hip): K = 9, stride 1, Cout % 128 == 0
bool tcn_v4_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags);
// ... and in KF6's form (tcn_bf16_v6.hip): one wave per SIMD, pair-order weights (appended to the packed blob by launch_tcn_pack)
bool tcn_v6_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags);
bool tcn_v6_stats_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags);   // launch_tcn_v6(..., stats)
bool tcn_v6_packs(int Cin, int Cout, int K, unsigned math);
int launch_tcn_pack_pairs_padded(const float *W, const float *scale, void *Wq, int Cin, int Cout, hipStream_t st);
int launch_tcn_v6(const float *x, const void *Wq, const float *shift, void *y, int N, int Cin, int Cout, int T, int V, int K,
                  int stride, unsigned flags, hipStream_t st, double *stats = nullptr);   // stats: see tcn_bf16_v6.hip
int launch_tcn_v4(const float *x, const void *Wp, const float *shift, void *y, int N, int Cin, int Cout, int T, int V, int K,
                  int stride, unsigned flags, hipStream_t st);

// training-mode BatchNorm helpers (train_bn.hip)
int launch_bn_batch_stats(const float *z, double *sums, int N, int C, size_t plane, hipStream_t st);
int launch_bn_train_finalize(const double *sums, double count, const float *weight, const float *bias,
                             float *running_mean, float *running_var, float momentum, float eps, float *scale,
                             float *shift, int C, hipStream_t st, float *save_mean = nullptr,
                             float *save_invstd = nullptr);
int launch_bn_frozen_finalize(const float *weight, const float *bias, const float *running_mean, const float *running_var,
                              float eps, float *scale, float *shift, int C, hipStream_t st, float *save_mean = nullptr,
                              float *save_invstd = nullptr);
int launch_bn_scale_shift(const float *weight, const float *bias, const float *mean, const float *invstd, float *scale,
                          float *shift, int C, hipStream_t st);
int launch_bn_apply(const float *za, const float *sa, const float *ta, const float *zb, const float *sb,
                    const float *tb, float *y, size_t total, int C, size_t plane, hipStream_t st);

// backward of the training-mode blocks (tcn_backward.hip)
int launch_bn_relu_bwd_stats(const float *za, const float *sa, const float *ta, const float *ma, const float *ia,
                             const float *zb, const float *sb, const float *tb, const float *mb, const float *ib,
                             const float *dy, double *sums, int N, int C, size_t plane, hipStream_t st);
int launch_bn_bwd_finalize(const double *sums, int which, double count, const float *gamma, const float *invstd,
                           float *dgamma, float *dbeta, float *coef, int C, hipStream_t st, bool frozen = false);
int launch_upsample2(const float *dz, float *dzu, size_t rows, int Tout, int T, int V, hipStream_t st);
int launch_bn_relu_bwd_apply(const float *za, const float *sa, const float *ta, const float *ma, const float *ia,
                             const float *zb, const float *sb, const float *tb, const float *mb, const float *ib,
                             const float *dy, const float *coefa, const float *coefb, float *dza, float *dzb, double *bsum,
                             int N, int C, size_t plane, hipStream_t st,
                             float *gout = nullptr);   // optional: the masked cotangent g itself (identity residual: dL/dx of "+ x")
int launch_doubles_to_floats(const double *src, float *dst, int n, hipStream_t st);
int launch_weight_flip(const float *W, float *Wf, int Cout, int Cin, int K, hipStream_t st);
int launch_tcn_dgrad_valu(const float *dz, const float *W, float *dx, int N, int Cin, int Cout, int T, int V, int K,
                          int stride, int Tout, hipStream_t st);
bool tcn_wgrad_mfma_supported(int N, int Cin, int Cout, int T, int V, int K, int stride);
size_t tcn_wgrad_ws_bytes(int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags);
// one wave per SIMD, input tile as a ring (tcn_wgrad_v6.hip): 17 <= V <= 24
bool tcn_wgrad_v6_supported(int N, int Cin, int Cout, int T, int V, int K, int stride);
int tcn_wgrad_v6_splits(int N, int Cin, int Cout, int T);
int launch_tcn_wgrad_v6(const float *dz, const float *x, float *part, int N, int Cin, int Cout, int T, int V, int K, unsigned flags,
                        hipStream_t st);
int launch_tcn_wgrad(const float *dz, const float *x, float *dW, float *part, int N, int Cin, int Cout, int T, int V, int K,
                     int stride, int Tout, unsigned flags, hipStream_t st);

// training-mode graph conv without materialised branches (agcn_train.hip)
bool agcn_moments_supported(int Cin, int V, int S);
size_t agcn_moments_ws_bytes(int N);
int launch_agcn_moments(const float *x, const float *P, double *part, const float *Wd, const float *bd, const float *Wdown,
                        const float *bdown, const float *bn_w, const float *bn_b, float *bn_rm, float *bn_rv,
                        const float *dbn_w, const float *dbn_b, float *dbn_rm, float *dbn_rv, float momentum, float eps,
                        float *s_m, float *t_m, float *s_d, float *t_d, float *save_stats, int N, int Cin, int Cout, int T,
                        int V, int S, hipStream_t st);

// backward of the training-mode graph conv (agcn_backward.hip)
bool agcn_bwd_supported(int N, int Cin, int Cout, int T, int V, int S);
size_t agcn_bwd_ws_bytes(int N, int Cin, int Cout, int T, int V, int S);
int launch_agcn_bwd(const float *x, const float *P, const float *A_eff, const float *y, const float *dy, const float *Wa,
                    const float *ba, const float *Wb, const float *bb, const float *Wd, const float *bd, const float *Wdown,
                    const float *bdown, const float *bn_w, const float *dbn_w, const float *stats, void *ws, float *dWa,
                    float *dba, float *dWb, float *dbb, float *dWd, float *dbd, float *dWdown, float *dbdown, float *dgamma,
                    float *dbeta, float *ddgamma, float *ddbeta, float *dPA, int N, int Cin, int Cout, int T, int V, int inter_c,
                    int S, hipStream_t st);

// strided batched fp32 GEMM + small helpers (gemm_f32.hip): C[b][m][n] (+)= alpha * sum_k A[b][m][k] B[b][k][n] (+ bias[m])
struct GemmArgs {
    const float *A, *B;
    float *C;
    const float *bias;                 // per output row m, or NULL
    int M, N, K;
    long long a_sm, a_sk, a_sb;        // element strides: row, contraction index, batch
    long long b_sk, b_sn, b_sb;
    long long c_sm, c_sn, c_sb;
    float alpha;
    int accumulate;                    // 0: C = ..., 1: C += ...
    // optional two-level ROW index: row i = q*m_inner + r addresses A at q*a_sm + r*a_sm2 and C at q*c_sm + r*c_sm2
    // (m_inner = 0: single level, i*a_sm / i*c_sm).  nbias (optional): per column n.  cbias (optional) adds
    // cbias[q*cb_sq + r*cb_sr + n*cb_sn].
    int m_inner = 0;
    long long a_sm2 = 0, c_sm2 = 0;
    const float *nbias = nullptr;
    const float *cbias = nullptr;
    long long cb_sq = 0, cb_sr = 0, cb_sn = 0;
    // optional split of the contraction index: part j of ksplit covers k in [j*kc, (j+1)*kc) (kc a multiple of the K chunk)
    // and writes C + j*c_ss — the caller sums the parts in a fixed order (launch_sum_parts).  For products with a long K
    // and a small C (the per-clip weight-gradient slices, the joint Gram matrices) whose one tile per clip left most CUs idle.
    int ksplit = 1;
    long long c_ss = 0;
    // optional two-level BATCH index (the chain's (clip, subset) products): b = bq*b_inner + br addresses A at
    // bq*a_sb + br*a_sb2, B and C alike (b_inner = 0: single level)
    int b_inner = 0;
    long long a_sb2 = 0, b_sb2 = 0, c_sb2 = 0;
    // optional two-level CONTRACTION index ((subset, joint) in dx += sum_s du_s P_s^T): k = kq*k_inner + kr addresses A at
    // kq*a_sk + kr*a_sk2 and B at kq*b_sk + kr*b_sk2 (k_inner = 0: single level)
    int k_inner = 0;
    long long a_sk2 = 0, b_sk2 = 0;
};
int launch_gemm_f32(const GemmArgs &g, int batch, hipStream_t st);
// out[rep*rep_stride + e] = sum_p part[p*n + e] in a fixed order, written `reps` times (one bias gradient shared by the subsets)
int launch_sum_parts(const float *part, float *out, int parts, size_t n, hipStream_t st, int reps = 1, size_t rep_stride = 0);
int launch_add_inplace(float *dst, const float *src, size_t n, hipStream_t st);
int launch_row_sum(const float *in, float *out, int rows, int cols, hipStream_t st);
// out[b][r][:] (+)= sum_{i < nsum} in[b][i][r][:] . M[b][i]  (rows of V floats times V x V matrices; gemm_f32.hip)
int launch_rowmix(const float *in, const float *M, float *out, int R, int V, int nsum, int accumulate, long long in_sb,
                  long long in_sb2, long long in_ss, long long m_sb, long long m_sb2, long long m_ss, bool m_transposed,
                  long long out_sb, long long out_sb2, int batch, int b_inner, hipStream_t st);
int launch_softmax_bwd(const float *P, const float *A_eff, const float *dP, float *dS, int N, int V, int S, float alpha,
                       hipStream_t st);

// first patch embedding of the transformer heads on the stem output (gemm_f32.hip: one strided GEMM per clip)
int launch_patch_embed(const float *z, const float *W, const float *b, const float *pos, float *out, int N, int C, int E,
                       int T, int V, unsigned flags, hipStream_t st);

// generic backward of the training-mode graph conv: any Cin / Cout / subsets, identity or conv residual, optional dx
// (agcn_backward_generic.hip)
size_t agcn_bwd_generic_ws_floats(int N, int Cin, int Cout, int T, int V, int inter_c, int S);
int launch_agcn_bwd_generic(const float *x, const float *P, const float *A_eff, const float *dzm, const float *dzd,
                            const float *Wa, const float *ba, const float *Wb, const float *bb, const float *Wd,
                            const float *Wdown, float *ws, float *dWa, float *dba, float *dWb, float *dbb, float *dWd,
                            float *dbd, float *dWdown, float *dbdown, float *dPA, float *dx, int dx_initialised, int N,
                            int Cin, int Cout, int T, int V, int inter_c, int S, hipStream_t st);

// fused stem
size_t stem_prep_bytes(int Cin, int C, int K, int S, unsigned flags);
int launch_stem_prepare(const float *Wd, const float *bd, const float *Wdown, const float *bdown,
                        const float *bn_scale, const float *bn_shift, const float *down_scale,
                        const float *down_shift, const float *Wt, const float *t_scale, void *prep,
                        int Cin, int C, int K, int S, unsigned flags, hipStream_t st);
size_t stem_ws_bytes(int N, int Cin, int C, int T, int V, int K, int S, unsigned flags);
float *stem_ws_features(void *ws, int N, int Cin, int C, int T, int V, int K, int S, unsigned flags);  // NULL if unused
float *stem_ws_bounds(void *ws, int N, int Cin, int C, int T, int V, int K, int S, unsigned flags);    // NULL unless KF7 serves the shape
float *stem_ws_xcopy(void *ws, int N, int Cin, int C, int T, int V, int K, int S, unsigned flags);     // NULL if unused
int launch_stem(const float *x, const float *P, const float *feat, const void *prep, const float *t_shift,
                void *out, int N, int Cin, int C, int T, int V, int S, int K, unsigned flags, hipStream_t st);

}  // namespace stgcn
