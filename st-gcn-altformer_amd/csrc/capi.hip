// extern "C" surface of libstgcn_hip.so (declared in include/stgcn_hip.h): argument validation,
// thread-local error text, and dispatch to the kernel launchers.  No allocation, no sync.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace stgcn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(stgcn_status st, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return (int)st;
}

namespace {

__global__ void bn_fold_kernel(const float *__restrict__ w, const float *__restrict__ b,
                               const float *__restrict__ rm, const float *__restrict__ rv,
                               const float *__restrict__ cb, float eps, float *__restrict__ scale,
                               float *__restrict__ shift, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    // same operation order as ATen's CPU batch-norm transform: w / sqrt(var + eps)
    const float s = w[c] / sqrtf(rv[c] + eps);
    const float centre = (cb ? cb[c] : 0.f) - rm[c];
    scale[c] = s;
    shift[c] = fmaf(centre, s, b[c]);
}

// per-rank reductions for the data-parallel harness: stats = [clips, sum(probe), sum(probe^2), correct] with
// probe[n][c] = out[n][c][0][0]; one workgroup, one launch (replaces ~6 tiny torch kernels per step).
// correct = #{n : argmax_c logits[n][c] == labels[n]} — get_acc of SHREC/ST_TS/train_sttran.py:105-109 (np.argmax on the
// host there).  np.argmax semantics: the LOWEST index among equal maxima, and a NaN counts as the maximum (first NaN wins).
__device__ inline int argmax_row(const float *__restrict__ row, int classes) {
    float best = row[0];
    int arg = 0;
    if (best != best) return 0;
    for (int c = 1; c < classes; ++c) {
        const float v = row[c];
        if (v != v) return c;          // NaN: np.argmax / torch.argmax return its index
        if (v > best) { best = v; arg = c; }
    }
    return arg;
}

template <bool BF16>
__global__ __launch_bounds__(256) void step_stats_kernel(const void *__restrict__ out, float *__restrict__ stats,
                                                          int N, int C, size_t clip_stride, size_t chan_stride,
                                                          float n_local,
                                                          const float *__restrict__ logits,
                                                          const long long *__restrict__ labels,
                                                          long long *__restrict__ pred, int n_logits, int classes) {
    __shared__ float red[3][4];
    float s1 = 0.f, s2 = 0.f, hit = 0.f;
    if (out != nullptr)
        for (int e = threadIdx.x; e < N * C; e += 256) {
            const size_t at = (size_t)(e / C) * clip_stride + (size_t)(e % C) * chan_stride;   // element (n, c, 0, 0)
            float v;
            if constexpr (BF16) v = __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(out)[at] << 16);
            else v = reinterpret_cast<const float *>(out)[at];
            s1 += v;
            s2 = fmaf(v, v, s2);
        }
    if (logits != nullptr)
        for (int n = threadIdx.x; n < n_logits; n += 256) {
            const int a = argmax_row(logits + (size_t)n * classes, classes);
            if (pred != nullptr) pred[n] = a;
            if (labels != nullptr && labels[n] == (long long)a) hit += 1.f;
        }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o, 64);
        s2 += __shfl_down(s2, o, 64);
        hit += __shfl_down(hit, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s1; red[1][w] = s2; red[2][w] = hit; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[0] = n_local;
        stats[1] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        stats[2] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        stats[3] = red[2][0] + red[2][1] + red[2][2] + red[2][3];   // exact: a count below 2^24
    }
}

}  // namespace

int launch_bn_fold(const float *w, const float *b, const float *rm, const float *rv, const float *cb,
                   float eps, float *scale, float *shift, int C, hipStream_t st) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, w, b, rm, rv, cb, eps, scale,
                       shift, C);
    STGCN_LAUNCH_CHECK("bn_fold_kernel");
    return STGCN_OK;
}

}  // namespace stgcn

using namespace stgcn;

#define REQUIRE_PTR(p)                                                              \
    do {                                                                            \
        if ((p) == nullptr) return fail(STGCN_ERR_ARG, "%s: %s is NULL", __func__, #p); \
    } while (0)
#define REQUIRE_POS(v)                                                                        \
    do {                                                                                      \
        if ((v) <= 0) return fail(STGCN_ERR_ARG, "%s: %s = %d must be positive", __func__, #v, (int)(v)); \
    } while (0)

extern "C" {

int stgcn_version(void) { return STGCN_ABI_VERSION; }

const char *stgcn_last_error(void) { return g_err; }

int stgcn_bn_fold(const float *weight, const float *bias, const float *running_mean,
                  const float *running_var, const float *conv_bias, float eps, float *scale, float *shift,
                  int C, void *stream) {
    REQUIRE_PTR(weight); REQUIRE_PTR(bias); REQUIRE_PTR(running_mean); REQUIRE_PTR(running_var);
    REQUIRE_PTR(scale); REQUIRE_PTR(shift); REQUIRE_POS(C);
    return launch_bn_fold(weight, bias, running_mean, running_var, conv_bias, eps, scale, shift, C,
                          (hipStream_t)stream);
}

int stgcn_agcn_attention(const float *x, const float *A_eff, const float *Wa, const float *ba,
                         const float *Wb, const float *bb, float *P, int N, int Cin, int T, int V,
                         int inter_c, int subsets, void *stream) {
    REQUIRE_PTR(x); REQUIRE_PTR(A_eff); REQUIRE_PTR(Wa); REQUIRE_PTR(ba); REQUIRE_PTR(Wb); REQUIRE_PTR(bb);
    REQUIRE_PTR(P);
    REQUIRE_POS(N); REQUIRE_POS(Cin); REQUIRE_POS(T); REQUIRE_POS(V); REQUIRE_POS(inter_c); REQUIRE_POS(subsets);
    if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "attention: N=%d > 65535 clips per call", N);
    return launch_attention(x, A_eff, Wa, ba, Wb, bb, P, nullptr, N, Cin, T, V, inter_c, subsets,
                            (hipStream_t)stream);
}

int stgcn_agcn_forward(const float *x, const float *A_eff, const float *Wa, const float *ba,
                       const float *Wb, const float *bb, const float *Wd, const float *bd,
                       const float *Wdown, const float *bdown, const float *bn_scale,
                       const float *bn_shift, const float *down_scale, const float *down_shift, float *P_ws,
                       float *y, int N, int Cin, int Cout, int T, int V, int inter_c, int subsets,
                       void *stream) {
    REQUIRE_PTR(Wd); REQUIRE_PTR(bd); REQUIRE_PTR(bn_scale); REQUIRE_PTR(bn_shift); REQUIRE_PTR(y);
    REQUIRE_POS(Cout);
    if ((Wdown == nullptr) != (bdown == nullptr) || (Wdown == nullptr) != (down_scale == nullptr) ||
        (Wdown == nullptr) != (down_shift == nullptr))
        return fail(STGCN_ERR_ARG, "agcn_forward: Wdown/bdown/down_scale/down_shift must be all set or all NULL");
    if (Wdown == nullptr && Cin != Cout)
        return fail(STGCN_ERR_ARG, "agcn_forward: identity residual needs Cin == Cout (got %d, %d)", Cin, Cout);
    int rc = stgcn_agcn_attention(x, A_eff, Wa, ba, Wb, bb, P_ws, N, Cin, T, V, inter_c, subsets, stream);
    if (rc != STGCN_OK) return rc;
    return launch_agcn_expand(x, P_ws, Wd, bd, Wdown, bdown, bn_scale, bn_shift, down_scale, down_shift, y, N,
                              Cin, Cout, T, V, subsets, 0, (hipStream_t)stream);
}

size_t stgcn_tcn_packed_bytes(int Cin, int Cout, int K, unsigned flags) {
    if (Cin <= 0 || Cout <= 0 || K <= 0) return 0;
    return tcn_packed_bytes(Cin, Cout, K, flags);
}

int stgcn_tcn_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags) {
    if (Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || K <= 0 || stride <= 0) return 0;
    return tcn_mfma_supported(Cin, Cout, T, V, K, stride, flags) ? 1 : 0;
}

int stgcn_stem_supported(int Cin, int C, int T, int V, int K, int subsets, unsigned flags) {
    if (Cin <= 0 || C <= 0 || T <= 0 || V <= 0 || K <= 0 || subsets <= 0) return 0;
    return stem_fused_supported(Cin, C, T, V, K, subsets, flags) ? 1 : 0;
}

int stgcn_tcn_pack(const float *W, const float *scale, void *Wp, int Cin, int Cout, int K, unsigned flags,
                   void *stream) {
    REQUIRE_PTR(W); REQUIRE_PTR(scale); REQUIRE_PTR(Wp);
    REQUIRE_POS(Cin); REQUIRE_POS(Cout); REQUIRE_POS(K);
    return launch_tcn_pack(W, scale, Wp, Cin, Cout, K, flags, (hipStream_t)stream);
}

int stgcn_tcn_forward_packed(const float *x, const void *Wp, const float *shift, void *y, int N, int Cin,
                             int Cout, int T, int V, int K, int stride, unsigned flags, void *stream) {
    REQUIRE_PTR(x); REQUIRE_PTR(Wp); REQUIRE_PTR(shift); REQUIRE_PTR(y);
    REQUIRE_POS(N); REQUIRE_POS(Cin); REQUIRE_POS(Cout); REQUIRE_POS(T); REQUIRE_POS(V); REQUIRE_POS(K);
    REQUIRE_POS(stride);
    return launch_tcn(x, Wp, shift, y, N, Cin, Cout, T, V, K, stride, flags, (hipStream_t)stream);
}

int stgcn_tcn_forward(const float *x, const float *W, const float *scale, const float *shift, void *y, int N,
                      int Cin, int Cout, int T, int V, int K, int stride, void *ws, size_t ws_bytes,
                      unsigned flags, void *stream) {
    REQUIRE_PTR(ws);
    REQUIRE_POS(Cin); REQUIRE_POS(Cout); REQUIRE_POS(K);
    const size_t need = tcn_packed_bytes(Cin, Cout, K, flags);
    if (ws_bytes < need)
        return fail(STGCN_ERR_WORKSPACE, "tcn_forward: workspace %zu B < %zu B", ws_bytes, need);
    int rc = stgcn_tcn_pack(W, scale, ws, Cin, Cout, K, flags, stream);
    if (rc != STGCN_OK) return rc;
    return stgcn_tcn_forward_packed(x, ws, shift, y, N, Cin, Cout, T, V, K, stride, flags, stream);
}

size_t stgcn_stem_prep_bytes(int Cin, int C, int K, int subsets, unsigned flags) {
    if (Cin <= 0 || C <= 0 || K <= 0 || subsets <= 0) return 0;
    return stem_prep_bytes(Cin, C, K, subsets, flags);
}

int stgcn_stem_prepare(const float *Wd, const float *bd, const float *Wdown, const float *bdown,
                       const float *bn_scale, const float *bn_shift, const float *down_scale,
                       const float *down_shift, const float *Wt, const float *t_scale, void *prep, int Cin,
                       int C, int K, int subsets, unsigned flags, void *stream) {
    REQUIRE_PTR(Wd); REQUIRE_PTR(bd); REQUIRE_PTR(Wdown); REQUIRE_PTR(bdown); REQUIRE_PTR(bn_scale);
    REQUIRE_PTR(bn_shift); REQUIRE_PTR(down_scale); REQUIRE_PTR(down_shift); REQUIRE_PTR(Wt);
    REQUIRE_PTR(t_scale); REQUIRE_PTR(prep);
    REQUIRE_POS(Cin); REQUIRE_POS(C); REQUIRE_POS(K); REQUIRE_POS(subsets);
    return launch_stem_prepare(Wd, bd, Wdown, bdown, bn_scale, bn_shift, down_scale, down_shift, Wt, t_scale,
                               prep, Cin, C, K, subsets, flags, (hipStream_t)stream);
}

size_t stgcn_stem_ws_bytes(int N, int Cin, int C, int T, int V, int K, int subsets, unsigned flags) {
    if (N <= 0 || Cin <= 0 || C <= 0 || T <= 0 || V <= 0 || K <= 0 || subsets <= 0) return 0;
    return stem_ws_bytes(N, Cin, C, T, V, K, subsets, flags);
}

const char *stgcn_stem_kernel_name(int Cin, int C, int T, int V, int K, int subsets, unsigned flags) {
    if (Cin <= 0 || C <= 0 || T <= 0 || V <= 0 || K <= 0 || subsets <= 0) return "";
    if (!stem_fused_supported(Cin, C, T, V, K, subsets, flags)) return "";
    if (stem_v4_supported(Cin, C, T, V, K, subsets, flags)) {
        if (stem_v4_features_in_kernel(C, T, V, K, flags) &&
            (stem_v6_supported(C, T, V, K, flags) || stem_v6w_supported(C, T, V, K, flags)))
            return stem_f16mx_supported(C, T, V, K, flags) ? "stem_f16mx_kernel" : "stem_bf16_v6_kernel";
        return "stem_bf16_v4_kernel";
    }
    const unsigned math = flags & STGCN_MATH_MASK;
    return (math == STGCN_MATH_BF16X3 || math == STGCN_MATH_BF16) ? "stem_mfma_bf16_kernel" : "stem_mfma_f32_kernel";
}

int stgcn_stem_features_used(int Cin, int C, int T, int V, int K, int subsets, unsigned flags) {
    if (Cin <= 0 || C <= 0 || T <= 0 || V <= 0 || K <= 0 || subsets <= 0) return 0;
    return stem_v4_supported(Cin, C, T, V, K, subsets, flags) ? 1 : 0;
}

int stgcn_stem_attention(const float *x, const float *A_eff, const float *Wa, const float *ba, const float *Wb,
                         const float *bb, void *ws, size_t ws_bytes, int N, int Cin, int C, int T, int V,
                         int inter_c, int subsets, int K, unsigned flags, void *stream) {
    REQUIRE_PTR(x); REQUIRE_PTR(A_eff); REQUIRE_PTR(Wa); REQUIRE_PTR(ba); REQUIRE_PTR(Wb); REQUIRE_PTR(bb);
    REQUIRE_PTR(ws);
    REQUIRE_POS(N); REQUIRE_POS(Cin); REQUIRE_POS(C); REQUIRE_POS(T); REQUIRE_POS(V); REQUIRE_POS(inter_c);
    REQUIRE_POS(subsets); REQUIRE_POS(K);
    if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "stem: N=%d > 65535 clips per call", N);
    const size_t need = stem_ws_bytes(N, Cin, C, T, V, K, subsets, flags);
    if (ws_bytes < need) return fail(STGCN_ERR_WORKSPACE, "stem: workspace %zu B < %zu B", ws_bytes, need);
    float *fpart = stem_ws_features(ws, N, Cin, C, T, V, K, subsets, flags);   // feature rows, or attention fragments
    const bool frags = fpart != nullptr && stem_v4_features_in_kernel(C, T, V, K, flags);
    return launch_attention(x, A_eff, Wa, ba, Wb, bb, (float *)ws, frags ? nullptr : fpart, N, Cin, T, V, inter_c, subsets,
                            (hipStream_t)stream, (flags & STGCN_IN_NTVC) != 0,
                            stem_ws_xcopy(ws, N, Cin, C, T, V, K, subsets, flags), frags ? fpart : nullptr,
                            frags ? stem_wide_split(V) : 0,    // wide frames: fragments for the two joint halves
                            frags ? stem_ws_bounds(ws, N, Cin, C, T, V, K, subsets, flags) : nullptr);
}

int stgcn_stem_tail_prepared(const float *x, const void *ws, size_t ws_bytes, const void *prep, const float *t_shift,
                             void *out, int N, int Cin, int C, int T, int V, int subsets, int K, unsigned flags,
                             void *stream) {
    REQUIRE_PTR(x); REQUIRE_PTR(ws); REQUIRE_PTR(prep); REQUIRE_PTR(t_shift); REQUIRE_PTR(out);
    REQUIRE_POS(N); REQUIRE_POS(Cin); REQUIRE_POS(C); REQUIRE_POS(T); REQUIRE_POS(V); REQUIRE_POS(subsets);
    REQUIRE_POS(K);
    const size_t need = stem_ws_bytes(N, Cin, C, T, V, K, subsets, flags);
    if (ws_bytes < need) return fail(STGCN_ERR_WORKSPACE, "stem: workspace %zu B < %zu B", ws_bytes, need);
    if (const float *xc = stem_ws_xcopy(const_cast<void *>(ws), N, Cin, C, T, V, K, subsets, flags)) x = xc;  // (N,T,V,Cin) input
    return launch_stem(x, (const float *)ws, stem_ws_features(const_cast<void *>(ws), N, Cin, C, T, V, K, subsets, flags),
                       prep, t_shift, out, N, Cin, C, T, V, subsets, K, flags, (hipStream_t)stream);
}

int stgcn_stem_forward_prepared(const float *x, const float *A_eff, const float *Wa, const float *ba,
                                const float *Wb, const float *bb, const void *prep, const float *t_shift, void *ws,
                                size_t ws_bytes, void *out, int N, int Cin, int C, int T, int V, int inter_c,
                                int subsets, int K, unsigned flags, void *stream) {
    int rc = stgcn_stem_attention(x, A_eff, Wa, ba, Wb, bb, ws, ws_bytes, N, Cin, C, T, V, inter_c, subsets, K, flags,
                                  stream);
    if (rc != STGCN_OK) return rc;
    return stgcn_stem_tail_prepared(x, ws, ws_bytes, prep, t_shift, out, N, Cin, C, T, V, subsets, K, flags, stream);
}

// ---- training-mode forward ------------------------------------------------------------------------------------
// workspace layout (agcn): [ones C][zeros C][scale/shift 4*C floats][sums 2 x 2C doubles][z_main N*C*T*V][z_down N*C*T*V]
static size_t train_small_bytes(int C) { return align_up((size_t)C * 6 * sizeof(float) + (size_t)C * 4 * sizeof(double), 256); }

// materialise != 0: room for the two pre-BatchNorm branches (needed when they are to be saved, or for shapes the
// moments path does not cover); 0: the moments path's scratch only
size_t stgcn_agcn_train_ws_bytes(int N, int Cin, int Cout, int T, int V, int subsets, int materialise) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || subsets <= 0) return 0;
    if (!materialise && agcn_moments_supported(Cin, V, subsets)) return train_small_bytes(Cout) + agcn_moments_ws_bytes(N);
    return train_small_bytes(Cout) + (size_t)2 * N * Cout * T * V * sizeof(float);
}

__global__ static void fill_ones_zeros_kernel(float *ones, float *zeros, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) { ones[c] = 1.f; zeros[c] = 0.f; }
}

int stgcn_agcn_forward_train(const float *x, const float *A_eff, const float *Wa, const float *ba, const float *Wb,
                             const float *bb, const float *Wd, const float *bd, const float *Wdown,
                             const float *bdown, const float *bn_weight, const float *bn_bias, float *bn_running_mean,
                             float *bn_running_var, const float *dbn_weight, const float *dbn_bias,
                             float *dbn_running_mean, float *dbn_running_var, float momentum, float eps, float *P_ws,
                             void *ws, size_t ws_bytes, float *y, float *save_zm, float *save_zd, float *save_stats,
                             int N, int Cin, int Cout, int T, int V, int inter_c, int subsets, unsigned flags, void *stream) {
    REQUIRE_PTR(Wd); REQUIRE_PTR(bd); REQUIRE_PTR(bn_weight); REQUIRE_PTR(bn_bias); REQUIRE_PTR(bn_running_mean);
    REQUIRE_PTR(bn_running_var); REQUIRE_PTR(ws); REQUIRE_PTR(y); REQUIRE_POS(Cout);
    const bool has_down = Wdown != nullptr;
    if (has_down && (!bdown || !dbn_weight || !dbn_bias || !dbn_running_mean || !dbn_running_var))
        return fail(STGCN_ERR_ARG, "agcn_forward_train: down branch given without its bias / BatchNorm tensors");
    if (!has_down && Cin != Cout)
        return fail(STGCN_ERR_ARG, "agcn_forward_train: identity residual needs Cin == Cout (got %d, %d)", Cin, Cout);
    int rc = stgcn_agcn_attention(x, A_eff, Wa, ba, Wb, bb, P_ws, N, Cin, T, V, inter_c, subsets, stream);
    if (rc != STGCN_OK) return rc;
    const bool frozen = (flags & STGCN_BN_FROZEN) != 0;     // running statistics: the branches are materialised
    const bool moments = !frozen && has_down && !save_zm && !save_zd && agcn_moments_supported(Cin, V, subsets);
    if (ws_bytes < stgcn_agcn_train_ws_bytes(N, Cin, Cout, T, V, subsets, moments ? 0 : 1))
        return fail(STGCN_ERR_WORKSPACE, "agcn_forward_train: workspace %zu B too small", ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    float *ones = (float *)ws, *zeros = ones + Cout, *s1 = zeros + Cout, *t1 = s1 + Cout, *s2 = t1 + Cout, *t2 = s2 + Cout;
    if (moments) {   // batch statistics from the moments of the 12 per-pixel features; the branches are never written
        double *part = (double *)((char *)ws + train_small_bytes(Cout));
        rc = launch_agcn_moments(x, P_ws, part, Wd, bd, Wdown, bdown, bn_weight, bn_bias, bn_running_mean, bn_running_var,
                                 dbn_weight, dbn_bias, dbn_running_mean, dbn_running_var, momentum, eps, s1, t1, s2, t2,
                                 save_stats, N, Cin, Cout, T, V, subsets, st);
        if (rc != STGCN_OK) return rc;
        return launch_agcn_expand(x, P_ws, Wd, bd, Wdown, bdown, s1, t1, s2, t2, y, N, Cin, Cout, T, V, subsets, 0, st);
    }
    double *sums1 = (double *)(t2 + Cout), *sums2 = sums1 + 2 * Cout;
    float *zm = save_zm ? save_zm : (float *)((char *)ws + train_small_bytes(Cout));
    float *zd = save_zd ? save_zd : (float *)((char *)ws + train_small_bytes(Cout)) + (size_t)N * Cout * T * V;
    // save_stats (4*Cout): batch mean, invstd of the main BatchNorm, then of the down BatchNorm
    float *sv_mm = save_stats, *sv_im = save_stats ? save_stats + Cout : nullptr;
    float *sv_md = save_stats ? save_stats + 2 * Cout : nullptr, *sv_id = save_stats ? save_stats + 3 * Cout : nullptr;
    const size_t plane = (size_t)T * V, total = (size_t)N * Cout * plane;
    if (save_stats)   // no feature moments on this path: clear their block incl. the validity mark the moment-form backward checks
        STGCN_HIP_CHECK(hipMemsetAsync(save_stats + 4 * Cout, 0, 128 * sizeof(float), st));
    hipLaunchKernelGGL(fill_ones_zeros_kernel, dim3(ceil_div(Cout, 256)), dim3(256), 0, st, ones, zeros, Cout);
    STGCN_LAUNCH_CHECK("fill_ones_zeros_kernel");
    // main branch, pre-BN: sum_s conv_d_s(x P_s)   (unit scale on the main path, zero on the residual path, no ReLU).
    // Outside the stem class the residual rows are left out of the contraction altogether (no Wdown: "identity" with the
    // residual term off, mode bit 1) and conv_down runs as one plain product below — the expansion kernel run a second time
    // with the main scales at zero did the whole work of both branches again (2 x 208 us at 64 -> 128 channels, 64 clips).
    const bool down_as_gemm = has_down && Cin != 3;
    rc = down_as_gemm ? launch_agcn_expand(x, P_ws, Wd, bd, nullptr, nullptr, ones, zeros, nullptr, nullptr, zm, N, Cin, Cout, T, V,
                                           subsets, 1 | 2, st)
                      : launch_agcn_expand(x, P_ws, Wd, bd, Wdown, bdown, ones, zeros, has_down ? zeros : nullptr,
                                           has_down ? zeros : nullptr, zm, N, Cin, Cout, T, V, subsets, 1 | 2, st);
    if (rc != STGCN_OK) return rc;
    if (frozen) {
        rc = launch_bn_frozen_finalize(bn_weight, bn_bias, bn_running_mean, bn_running_var, eps, s1, t1, Cout, st, sv_mm, sv_im);
    } else {
        rc = launch_bn_batch_stats(zm, sums1, N, Cout, plane, st);
        if (rc != STGCN_OK) return rc;
        rc = launch_bn_train_finalize(sums1, (double)N * plane, bn_weight, bn_bias, bn_running_mean, bn_running_var, momentum,
                                      eps, s1, t1, Cout, st, sv_mm, sv_im);
    }
    if (rc != STGCN_OK) return rc;
    if (has_down) {  // residual branch, pre-BN: conv_down(x)
        if (down_as_gemm) {     // zd[n] = Wdown x[n] + bdown
            const long long Pl = (long long)plane;
            GemmArgs g{Wdown, x, zd, bdown, Cout, (int)plane, Cin, Cin, 1, 0, Pl, 1, (long long)Cin * Pl, Pl, 1, (long long)Cout * Pl, 1.f, 0};
            rc = launch_gemm_f32(g, N, st);
        } else {
            rc = launch_agcn_expand(x, P_ws, Wd, bd, Wdown, bdown, zeros, zeros, ones, zeros, zd, N, Cin, Cout, T, V, subsets,
                                    1, st);
        }
        if (rc != STGCN_OK) return rc;
        if (frozen) {
            rc = launch_bn_frozen_finalize(dbn_weight, dbn_bias, dbn_running_mean, dbn_running_var, eps, s2, t2, Cout, st, sv_md,
                                           sv_id);
        } else {
            rc = launch_bn_batch_stats(zd, sums2, N, Cout, plane, st);
            if (rc != STGCN_OK) return rc;
            rc = launch_bn_train_finalize(sums2, (double)N * plane, dbn_weight, dbn_bias, dbn_running_mean, dbn_running_var,
                                          momentum, eps, s2, t2, Cout, st, sv_md, sv_id);
        }
        if (rc != STGCN_OK) return rc;
        return launch_bn_apply(zm, s1, t1, zd, s2, t2, y, total, Cout, plane, st);
    }
    return launch_bn_apply(zm, s1, t1, x, nullptr, nullptr, y, total, Cout, plane, st);  // identity residual: + x
}

// ---- backward of the training-mode graph conv (stem shape class) --------------------------------------------------
// workspace: [sums 3C dbl][coef_m 3C][coef_d 3C][scale_m, shift_m, scale_d, shift_d][ones, zeros] | per-workgroup partials
//            | (recompute) the two pre-BatchNorm branches
static size_t agcn_bwd_small_bytes(int Cout) {
    return align_up((size_t)Cout * 3 * sizeof(double) + (size_t)Cout * 12 * sizeof(float), 256);
}

// The moment-form backward (agcn_backward.hip) serves the stem's shape class after a moments-path forward when no input
// gradient is wanted; everything else (any Cin / Cout / subsets, identity residual, dx, saved branches) takes the generic
// GEMM chain.
static bool agcn_bwd_use_fused(int N, int Cin, int Cout, int T, int V, int S, bool has_down, bool want_dx, bool moments) {
    return has_down && !want_dx && moments && agcn_bwd_supported(N, Cin, Cout, T, V, S);
}
static size_t agcn_bwd_generic_bytes(int N, int Cin, int Cout, int T, int V, int S) {
    const int inter_c = Cout / 4 > 0 ? Cout / 4 : 1;   // upper bound used for sizing: unit_agcn's coff_embedding = 4
    return align_up(((size_t)2 * N * Cout * T * V + agcn_bwd_generic_ws_floats(N, Cin, Cout, T, V, inter_c, S)) * sizeof(float), 256);
}

// recompute: bit 0 = the two pre-BatchNorm branches are not supplied (rebuilt in the workspace); bit 1 = size for the
// generic path (input gradient wanted / identity residual / a shape outside the stem class).  Never 0 for valid sizes.
size_t stgcn_agcn_backward_ws_bytes(int N, int Cin, int Cout, int T, int V, int subsets, int recompute) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || subsets <= 0 || V > 64) return 0;
    const size_t branches = (recompute & 1) ? (size_t)2 * N * Cout * T * V * sizeof(float) : 0;
    const size_t fused = agcn_bwd_ws_bytes(N, Cin, Cout, T, V, subsets);
    if (fused && !(recompute & 2)) return fused;                      // the moment form: needs neither branch
    const size_t generic = agcn_bwd_small_bytes(Cout) + agcn_bwd_generic_bytes(N, Cin, Cout, T, V, subsets) + branches;
    return generic > fused ? generic : fused;
}

int stgcn_agcn_backward_train(const float *x, const float *A_eff, const float *Wa, const float *ba, const float *Wb,
                              const float *bb, const float *Wd, const float *bd, const float *Wdown, const float *bdown,
                              const float *P, const float *zm, const float *zd, const float *bn_weight,
                              const float *bn_bias, const float *dbn_weight, const float *dbn_bias,
                              const float *save_stats, const float *y, const float *dy, float *dWa, float *dba, float *dWb,
                              float *dbb,
                              float *dWd, float *dbd, float *dWdown, float *dbdown, float *dgamma, float *dbeta,
                              float *ddgamma, float *ddbeta, float *dPA, float *dx, void *ws, size_t ws_bytes, int N, int Cin,
                              int Cout, int T, int V, int inter_c, int subsets, unsigned flags, void *stream) {
    REQUIRE_PTR(x); REQUIRE_PTR(A_eff); REQUIRE_PTR(Wa); REQUIRE_PTR(ba); REQUIRE_PTR(Wb); REQUIRE_PTR(bb); REQUIRE_PTR(Wd);
    REQUIRE_PTR(bd); REQUIRE_PTR(P); REQUIRE_PTR(bn_weight); REQUIRE_PTR(bn_bias);
    REQUIRE_PTR(save_stats); REQUIRE_PTR(dy); REQUIRE_PTR(dWa); REQUIRE_PTR(dba);
    REQUIRE_PTR(dWb); REQUIRE_PTR(dbb); REQUIRE_PTR(dWd); REQUIRE_PTR(dbd);
    REQUIRE_PTR(dgamma); REQUIRE_PTR(dbeta); REQUIRE_PTR(dPA); REQUIRE_PTR(ws);
    REQUIRE_POS(N); REQUIRE_POS(Cin); REQUIRE_POS(Cout); REQUIRE_POS(T); REQUIRE_POS(V); REQUIRE_POS(inter_c); REQUIRE_POS(subsets);
    const bool has_down = Wdown != nullptr;
    if (has_down) {
        REQUIRE_PTR(bdown); REQUIRE_PTR(dbn_weight); REQUIRE_PTR(dbn_bias); REQUIRE_PTR(dWdown); REQUIRE_PTR(dbdown);
        REQUIRE_PTR(ddgamma); REQUIRE_PTR(ddbeta);
    } else if (Cin != Cout) {
        return fail(STGCN_ERR_ARG, "agcn_backward: identity residual needs Cin == Cout (got %d, %d)", Cin, Cout);
    }
    if (V > 64) return fail(STGCN_ERR_UNSUPPORTED, "agcn_backward: V=%d > 64", V);
    if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "agcn_backward: N=%d > 65535 clips per call", N);
    // the stem-class moment form wants what a moments-path forward leaves: no saved branches, the output y, the moments
    // the stem-class moment form is the closed form of the BATCH-statistics BatchNorm: frozen statistics take the GEMM chain
    const bool frozen = (flags & STGCN_BN_FROZEN) != 0;
    const bool fused = !frozen &&
                       agcn_bwd_use_fused(N, Cin, Cout, T, V, subsets, has_down, dx != nullptr, zm == nullptr && y != nullptr);
    if (inter_c > (Cout / 4 > 0 ? Cout / 4 : 1) && !fused)
        return fail(STGCN_ERR_UNSUPPORTED, "agcn_backward: inter_c=%d > Cout/4 (workspace is sized for coff_embedding >= 4)", inter_c);
    if (has_down ? ((zm == nullptr) != (zd == nullptr)) : (zd != nullptr))
        return fail(STGCN_ERR_ARG, "agcn_backward: give both saved branches or neither (zd only with a down branch)");
    const bool recompute = zm == nullptr;
    const size_t need = stgcn_agcn_backward_ws_bytes(N, Cin, Cout, T, V, subsets, (recompute ? 1 : 0) | (fused ? 0 : 2));
    if (ws_bytes < need) return fail(STGCN_ERR_WORKSPACE, "agcn_backward: workspace %zu B < %zu B", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    if (fused)
        return launch_agcn_bwd(x, P, A_eff, y, dy, Wa, ba, Wb, bb, Wd, bd, Wdown, bdown, bn_weight, dbn_weight, save_stats, ws,
                               dWa, dba, dWb, dbb, dWd, dbd, dWdown, dbdown, dgamma, dbeta, ddgamma, ddbeta, dPA, N, Cin, Cout,
                               T, V, inter_c, subsets, st);
    const size_t plane = (size_t)T * V, total = (size_t)N * Cout * plane;
    double *sums = (double *)ws;
    float *coefm = (float *)(sums + 3 * Cout), *coefd = coefm + 3 * Cout, *sm_ = coefd + 3 * Cout, *tm_ = sm_ + Cout,
          *sd_ = tm_ + Cout, *td_ = sd_ + Cout, *ones = td_ + Cout, *zeros = ones + Cout;
    char *body = (char *)ws + agcn_bwd_small_bytes(Cout);
    const size_t body_bytes = agcn_bwd_generic_bytes(N, Cin, Cout, T, V, subsets);
    int rc;
    if (recompute) {   // the forward kept no branches (moments path): rebuild them with the raw-mode expansion kernel
        float *zmw = (float *)(body + body_bytes);
        float *zdw = zmw + total;
        hipLaunchKernelGGL(fill_ones_zeros_kernel, dim3(ceil_div(Cout, 256)), dim3(256), 0, st, ones, zeros, Cout);
        STGCN_LAUNCH_CHECK("fill_ones_zeros_kernel");
        rc = launch_agcn_expand(x, P, Wd, bd, Wdown, bdown, ones, zeros, has_down ? zeros : nullptr, has_down ? zeros : nullptr,
                                zmw, N, Cin, Cout, T, V, subsets, 1 | 2, st);
        if (rc != STGCN_OK) return rc;
        zm = zmw;
        if (has_down) {
            rc = launch_agcn_expand(x, P, Wd, bd, Wdown, bdown, zeros, zeros, ones, zeros, zdw, N, Cin, Cout, T, V, subsets, 1, st);
            if (rc != STGCN_OK) return rc;
            zd = zdw;
        }
    }
    const float *mean_m = save_stats, *inv_m = save_stats + Cout, *mean_d = save_stats + 2 * Cout, *inv_d = save_stats + 3 * Cout;
    rc = launch_bn_scale_shift(bn_weight, bn_bias, mean_m, inv_m, sm_, tm_, Cout, st);
    if (rc != STGCN_OK) return rc;
    if (has_down) {
        rc = launch_bn_scale_shift(dbn_weight, dbn_bias, mean_d, inv_d, sd_, td_, Cout, st);
        if (rc != STGCN_OK) return rc;
    }
    // side b of the ReLU's argument: the second BatchNorm, or (scale == NULL) the identity residual x itself
    const float *zb = has_down ? zd : x, *sb = has_down ? sd_ : nullptr, *tb = has_down ? td_ : nullptr;
    const float *mb = has_down ? mean_d : nullptr, *ib = has_down ? inv_d : nullptr;
    rc = launch_bn_relu_bwd_stats(zm, sm_, tm_, mean_m, inv_m, zb, sb, tb, mb, ib, dy, sums, N, Cout, plane, st);
    if (rc != STGCN_OK) return rc;
    rc = launch_bn_bwd_finalize(sums, 1, (double)N * plane, bn_weight, inv_m, dgamma, dbeta, coefm, Cout, st, frozen);
    if (rc != STGCN_OK) return rc;
    if (has_down) {
        rc = launch_bn_bwd_finalize(sums, 2, (double)N * plane, dbn_weight, inv_d, ddgamma, ddbeta, coefd, Cout, st, frozen);
        if (rc != STGCN_OK) return rc;
    }
    // generic path: materialise both pre-BatchNorm gradients, then the GEMM chain
    float *dzm = (float *)body, *dzd = dzm + total, *gws = dzd + total;
    // (identity residual, unit_agcn.py:57-58,92: dL/dx of the "+ x" term is the masked cotangent itself — written by the same
    //  pass as dx's first term; it was a kernel of its own re-reading zm, x and dy)
    const bool dx_from_g = dx != nullptr && !has_down;
    rc = launch_bn_relu_bwd_apply(zm, sm_, tm_, mean_m, inv_m, zb, sb, tb, mb, ib, dy, coefm, has_down ? coefd : nullptr, dzm,
                                  has_down ? dzd : nullptr, nullptr, N, Cout, plane, st, dx_from_g ? dx : nullptr);
    if (rc != STGCN_OK) return rc;
    const int dx_init = dx_from_g ? 1 : 0;
    return launch_agcn_bwd_generic(x, P, A_eff, dzm, has_down ? dzd : nullptr, Wa, ba, Wb, bb, Wd, Wdown, gws, dWa, dba, dWb, dbb,
                                   dWd, dbd, dWdown, dbdown, dPA, dx, dx_init, N, Cin, Cout, T, V, inter_c, subsets, st);
}

// workspace layout (tcn): [ones C][zeros C][scale, shift][sums 2C doubles][packed weights][z N*Cout*Tout*V]
size_t stgcn_tcn_train_ws_bytes(int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags) {
    flags &= ~STGCN_BN_FROZEN;
    if (N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || K <= 0 || stride <= 0) return 0;
    const int Tout = (T + 2 * ((K - 1) / 2) - K) / stride + 1;
    if (Tout < 1) return 0;
    return train_small_bytes(Cout) + tcn_packed_bytes(Cin, Cout, K, flags) + (size_t)N * Cout * Tout * V * sizeof(float);
}

int stgcn_tcn_forward_train(const float *x, const float *W, const float *conv_bias, const float *bn_weight,
                            const float *bn_bias, float *bn_running_mean, float *bn_running_var, float momentum,
                            float eps, void *ws, size_t ws_bytes, float *y, float *save_z, float *save_mean,
                            float *save_invstd, int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags,
                            void *stream) {
    const bool frozen = (flags & STGCN_BN_FROZEN) != 0;
    flags &= ~STGCN_BN_FROZEN;
    REQUIRE_PTR(x); REQUIRE_PTR(W); REQUIRE_PTR(bn_weight); REQUIRE_PTR(bn_bias); REQUIRE_PTR(bn_running_mean);
    REQUIRE_PTR(bn_running_var); REQUIRE_PTR(ws); REQUIRE_PTR(y);
    REQUIRE_POS(N); REQUIRE_POS(Cin); REQUIRE_POS(Cout); REQUIRE_POS(T); REQUIRE_POS(V); REQUIRE_POS(K); REQUIRE_POS(stride);
    const size_t need = stgcn_tcn_train_ws_bytes(N, Cin, Cout, T, V, K, stride, flags);
    if (need == 0) return fail(STGCN_ERR_ARG, "tcn_forward_train: T=%d K=%d stride=%d gives no output frame", T, K, stride);
    if (ws_bytes < need) return fail(STGCN_ERR_WORKSPACE, "tcn_forward_train: workspace %zu B < %zu B", ws_bytes, need);
    if (flags & STGCN_OUT_BF16) return fail(STGCN_ERR_UNSUPPORTED, "tcn_forward_train: fp32 output only");
    hipStream_t st = (hipStream_t)stream;
    const int Tout = (T + 2 * ((K - 1) / 2) - K) / stride + 1;
    float *ones = (float *)ws, *zeros = ones + Cout, *s1 = zeros + Cout, *t1 = s1 + Cout;
    double *sums = (double *)(t1 + 3 * Cout);
    char *packed = (char *)ws + train_small_bytes(Cout);
    float *z = save_z ? save_z : (float *)(packed + tcn_packed_bytes(Cin, Cout, K, flags));   // conv_t(x) + b
    const size_t plane = (size_t)Tout * V, total = (size_t)N * Cout * plane;
    hipLaunchKernelGGL(fill_ones_zeros_kernel, dim3(ceil_div(Cout, 256)), dim3(256), 0, st, ones, zeros, Cout);
    STGCN_LAUNCH_CHECK("fill_ones_zeros_kernel");
    int rc = launch_tcn_pack(W, ones, packed, Cin, Cout, K, flags, st);   // unit scale: the raw convolution
    if (rc != STGCN_OK) return rc;
    // the one-wave kernel sums the batch statistics in its epilogue (no separate pass over z)
    const unsigned cflags = (flags & STGCN_MATH_MASK) | STGCN_RAW;
#ifdef STGCN_NO_CONV_STATS    /* A/B builds: the separate statistics pass */
    const bool stats_in_conv = false;
#else
    const bool stats_in_conv = !frozen && tcn_v6_stats_supported(Cin, Cout, T, V, K, stride, cflags) && !(ablate_mask() & 8192);
#endif
    if (stats_in_conv) {
        STGCN_HIP_CHECK(hipMemsetAsync(sums, 0, sizeof(double) * 2 * Cout, st));
        rc = launch_tcn_v6(x, packed + tcn_packed_single_bytes(Cin, Cout, K, cflags), conv_bias ? conv_bias : zeros, z, N, Cin, Cout,
                           T, V, K, stride, cflags, st, sums);
    } else {
        rc = launch_tcn(x, packed, conv_bias ? conv_bias : zeros, z, N, Cin, Cout, T, V, K, stride, cflags, st);
    }
    if (rc != STGCN_OK) return rc;
    if (frozen) {
        rc = launch_bn_frozen_finalize(bn_weight, bn_bias, bn_running_mean, bn_running_var, eps, s1, t1, Cout, st, save_mean,
                                       save_invstd);
    } else {
        rc = stats_in_conv ? STGCN_OK : launch_bn_batch_stats(z, sums, N, Cout, plane, st);
        if (rc != STGCN_OK) return rc;
        rc = launch_bn_train_finalize(sums, (double)N * plane, bn_weight, bn_bias, bn_running_mean, bn_running_var, momentum,
                                      eps, s1, t1, Cout, st, save_mean, save_invstd);
    }
    if (rc != STGCN_OK) return rc;
    return launch_bn_apply(z, s1, t1, nullptr, nullptr, nullptr, y, total, Cout, plane, st);
}

// ---- backward of the training-mode temporal conv block ---------------------------------------------------------
// workspace: [sums 3C dbl][bsum 2C dbl][coef 3C][scale C][shift C][ones Cin][zeros Cin] | dz | flipped W | packed | wgrad partials
static size_t tcn_bwd_small_bytes(int Cin, int Cout) {
    return align_up((size_t)Cout * 5 * sizeof(double) + ((size_t)Cout * 5 + (size_t)Cin * 2) * sizeof(float), 256);
}
static unsigned tcn_dgrad_flags(int Cin, int Cout, int Tout, int V, int K, unsigned flags) {
    unsigned math = flags & STGCN_MATH_MASK;   // dgrad = forward conv with Cout input and Cin output channels
    if (math != STGCN_MATH_F32_VALU && !tcn_mfma_supported(Cout, Cin, Tout, V, K, 1, math)) math = STGCN_MATH_F32_VALU;
    return math;
}

// The input gradient of a stride-1 block is the forward kernel on the flipped weights ONLY for odd K: the transposed
// conv pads K-1-pad frames, which equals the forward's pad = (K-1)/2 when K is odd.  With an even K the forward drops a
// frame (Tout = T-1) and that shortcut would write T-2 misaligned frames: even K runs the general VALU dgrad instead.
static bool tcn_dgrad_by_forward(int K, int stride) { return stride == 1 && (K & 1) == 1; }
// A stride-2 block with an odd K (TCN_GCN_unit's downsampling layers, model/ST_TR/ST_TR_new.py:362-372) runs its backward as the
// stride-1 block's on dz upsampled with zero frames (launch_upsample2): matrix-core wgrad and dgrad instead of plain FMAs.
static bool tcn_bwd_upsampled(int K, int stride) { return stride == 2 && (K & 1) == 1; }

size_t stgcn_tcn_backward_ws_bytes(int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags) {
    flags &= ~STGCN_BN_FROZEN;
    if (N <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || K <= 0 || stride <= 0) return 0;
    const int Tout = (T + 2 * ((K - 1) / 2) - K) / stride + 1;
    if (Tout < 1) return 0;
    size_t b = tcn_bwd_small_bytes(Cin, Cout) + align_up((size_t)N * Cout * Tout * V * sizeof(float), 256);
    const bool up = tcn_bwd_upsampled(K, stride);
    if (up) b += align_up((size_t)N * Cout * T * V * sizeof(float), 256);
    const int es = up ? 1 : stride, eTout = up ? T : Tout;
    if (tcn_dgrad_by_forward(K, es))
        b += align_up((size_t)Cout * Cin * K * sizeof(float), 256) +
             align_up(tcn_packed_bytes(Cout, Cin, K, tcn_dgrad_flags(Cin, Cout, eTout, V, K, flags)), 256);
    return b + tcn_wgrad_ws_bytes(N, Cin, Cout, T, V, K, es, flags);
}

int stgcn_tcn_backward_train(const float *x, const float *W, const float *z, const float *bn_weight,
                             const float *bn_bias, const float *save_mean, const float *save_invstd, const float *dy,
                             float *dx, float *dW, float *dbias, float *dgamma, float *dbeta, void *ws, size_t ws_bytes,
                             int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags, void *stream) {
    const bool frozen = (flags & STGCN_BN_FROZEN) != 0;
    flags &= ~STGCN_BN_FROZEN;
    REQUIRE_PTR(x); REQUIRE_PTR(W); REQUIRE_PTR(z); REQUIRE_PTR(bn_weight); REQUIRE_PTR(bn_bias); REQUIRE_PTR(save_mean);
    REQUIRE_PTR(save_invstd); REQUIRE_PTR(dy); REQUIRE_PTR(dW); REQUIRE_PTR(dgamma); REQUIRE_PTR(dbeta); REQUIRE_PTR(ws);
    REQUIRE_POS(N); REQUIRE_POS(Cin); REQUIRE_POS(Cout); REQUIRE_POS(T); REQUIRE_POS(V); REQUIRE_POS(K); REQUIRE_POS(stride);
    const size_t need = stgcn_tcn_backward_ws_bytes(N, Cin, Cout, T, V, K, stride, flags);
    if (need == 0) return fail(STGCN_ERR_ARG, "tcn_backward: T=%d K=%d stride=%d gives no output frame", T, K, stride);
    if (ws_bytes < need) return fail(STGCN_ERR_WORKSPACE, "tcn_backward: workspace %zu B < %zu B", ws_bytes, need);
    if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "tcn_backward: N=%d > 65535 clips per call", N);
    hipStream_t st = (hipStream_t)stream;
    const int Tout = (T + 2 * ((K - 1) / 2) - K) / stride + 1;
    const size_t plane = (size_t)Tout * V;
    double *sums = (double *)ws, *bsum = sums + 3 * Cout;
    float *coef = (float *)(bsum + 2 * Cout), *scale = coef + 3 * Cout, *shift = scale + Cout, *ones = shift + Cout,
          *zeros = ones + Cin;
    char *p = (char *)ws + tcn_bwd_small_bytes(Cin, Cout);
    float *dz = (float *)p;
    p += align_up((size_t)N * Cout * plane * sizeof(float), 256);
    int rc = launch_bn_scale_shift(bn_weight, bn_bias, save_mean, save_invstd, scale, shift, Cout, st);
    if (rc != STGCN_OK) return rc;
    rc = launch_bn_relu_bwd_stats(z, scale, shift, save_mean, save_invstd, nullptr, nullptr, nullptr, nullptr, nullptr, dy,
                                  sums, N, Cout, plane, st);
    if (rc != STGCN_OK) return rc;
    rc = launch_bn_bwd_finalize(sums, 1, (double)N * plane, bn_weight, save_invstd, dgamma, dbeta, coef, Cout, st, frozen);
    if (rc != STGCN_OK) return rc;
    rc = launch_bn_relu_bwd_apply(z, scale, shift, save_mean, save_invstd, nullptr, nullptr, nullptr, nullptr, nullptr, dy,
                                  coef, nullptr, dz, nullptr, dbias ? bsum : nullptr, N, Cout, plane, st);
    if (rc != STGCN_OK) return rc;
    if (dbias) {
        rc = launch_doubles_to_floats(bsum, dbias, Cout, st);
        if (rc != STGCN_OK) return rc;
    }
    int Tz = Tout;                               // frames of the gradient tensor the two conv gradients read
    if (tcn_bwd_upsampled(K, stride)) {
        float *dzu = (float *)p;
        p += align_up((size_t)N * Cout * T * V * sizeof(float), 256);
        rc = launch_upsample2(dz, dzu, (size_t)N * Cout, Tout, T, V, st);
        if (rc != STGCN_OK) return rc;
        dz = dzu;
        stride = 1;
        Tz = T;
    }
    if (dx != nullptr) {
        if (tcn_dgrad_by_forward(K, stride)) {   // dx = conv_t(dz, flipped W): the forward kernels, raw output
            float *Wf = (float *)p;
            p += align_up((size_t)Cout * Cin * K * sizeof(float), 256);
            const unsigned dfl = tcn_dgrad_flags(Cin, Cout, Tz, V, K, flags);
            void *packed = p;
            p += align_up(tcn_packed_bytes(Cout, Cin, K, dfl), 256);
            hipLaunchKernelGGL(fill_ones_zeros_kernel, dim3(ceil_div(Cin, 256)), dim3(256), 0, st, ones, zeros, Cin);
            STGCN_LAUNCH_CHECK("fill_ones_zeros_kernel");
            rc = launch_weight_flip(W, Wf, Cout, Cin, K, st);
            if (rc != STGCN_OK) return rc;
            rc = launch_tcn_pack(Wf, ones, packed, Cout, Cin, K, dfl, st);
            if (rc != STGCN_OK) return rc;
            rc = launch_tcn(dz, packed, zeros, dx, N, Cout, Cin, Tz, V, K, 1, dfl | STGCN_RAW, st);
            if (rc != STGCN_OK) return rc;
        } else {
            rc = launch_tcn_dgrad_valu(dz, W, dx, N, Cin, Cout, T, V, K, stride, Tz, st);
            if (rc != STGCN_OK) return rc;
        }
    } else if (tcn_dgrad_by_forward(K, stride)) {
        p += align_up((size_t)Cout * Cin * K * sizeof(float), 256) +
             align_up(tcn_packed_bytes(Cout, Cin, K, tcn_dgrad_flags(Cin, Cout, Tz, V, K, flags)), 256);
    }
    float *part = tcn_wgrad_ws_bytes(N, Cin, Cout, T, V, K, stride, flags) ? (float *)p : nullptr;
    return launch_tcn_wgrad(dz, x, dW, part, N, Cin, Cout, T, V, K, stride, Tz, flags, st);
}

int stgcn_patch_embed(const float *z, const float *W, const float *b, const float *pos, float *out, int N, int C, int E,
                      int T, int V, unsigned flags, void *stream) {
    REQUIRE_PTR(z); REQUIRE_PTR(W); REQUIRE_PTR(b); REQUIRE_PTR(out);
    REQUIRE_POS(N); REQUIRE_POS(C); REQUIRE_POS(E); REQUIRE_POS(T); REQUIRE_POS(V);
    if (flags & ~(STGCN_IN_NTVC | STGCN_EMBED_TS)) return fail(STGCN_ERR_ARG, "patch_embed: unknown flag bits 0x%x", flags);
    if (N > 65535) return fail(STGCN_ERR_UNSUPPORTED, "patch_embed: N=%d > 65535 clips per call", N);
    return launch_patch_embed(z, W, b, pos, out, N, C, E, T, V, flags, (hipStream_t)stream);
}

int stgcn_step_stats(const void *out, int out_is_bf16, float *stats, int N, int C, long clip_stride, long chan_stride,
                     float n_local,
                     const float *logits, const long long *labels, long long *pred, int n_logits, int classes,
                     void *stream) {
    REQUIRE_PTR(stats);
    if (out != nullptr) { REQUIRE_POS(N); REQUIRE_POS(C); REQUIRE_POS(clip_stride); REQUIRE_POS(chan_stride); }
    if (logits != nullptr) {
        REQUIRE_POS(n_logits); REQUIRE_POS(classes);
        if (n_logits >= (1 << 24)) return fail(STGCN_ERR_UNSUPPORTED, "step_stats: %d rows of logits per call", n_logits);
    } else if (labels != nullptr || pred != nullptr) {
        return fail(STGCN_ERR_ARG, "step_stats: labels / pred given without logits");
    }
    if (out == nullptr && logits == nullptr) return fail(STGCN_ERR_ARG, "step_stats: neither out nor logits given");
    if (out_is_bf16)
        hipLaunchKernelGGL(step_stats_kernel<true>, dim3(1), dim3(256), 0, (hipStream_t)stream, out, stats, N, C,
                           (size_t)clip_stride, (size_t)chan_stride, n_local, logits, labels, pred, n_logits, classes);
    else
        hipLaunchKernelGGL(step_stats_kernel<false>, dim3(1), dim3(256), 0, (hipStream_t)stream, out, stats, N, C,
                           (size_t)clip_stride, (size_t)chan_stride, n_local, logits, labels, pred, n_logits, classes);
    STGCN_LAUNCH_CHECK("step_stats_kernel");
    return STGCN_OK;
}

}  // extern "C"
