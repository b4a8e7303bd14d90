/*
 * stgcn_hip.h — C ABI of libstgcn_hip.so: the MI355X (gfx950) ST-GCN stem.
 *
 * The reference (zjtggssg/ST-GCN-AltFormer) is pure Python/PyTorch and has no FFI of
 * its own; the boundary it offers for this path is two nn.Modules.  Each entry point
 * below replaces the torch ops executed inside one of their forward() bodies, so that
 * a re-implementation of those modules (st-gcn-altformer_amd/model/{unit_agcn,net}.py,
 * bound with ctypes — see INTEGRATION.md) is a drop-in:
 *
 *   stgcn_agcn_attention / stgcn_agcn_forward   <- model/unit_agcn.py:73-93  (unit_agcn.forward)
 *   stgcn_tcn_*                                 <- model/net.py:47-57        (Unit2D.forward, dim=2)
 *   stgcn_stem_*                                <- model/AltFormer/ST_GCN_AltFormer.py:70-72
 *                                                  (tcn0(gcn0(x)) fused, intermediate kept on chip)
 *   stgcn_bn_fold                               <- eval-mode nn.BatchNorm2d at unit_agcn.py:54,60, net.py:40
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (e.g. the PyTorch caching
 *     allocator).  The library allocates nothing, frees nothing, keeps no global mutable
 *     state and never synchronises: all work is enqueued on `stream` (a hipStream_t
 *     passed as void*; NULL = the default stream).  Safe to call from several host
 *     threads / devices concurrently (the caller selects the device, as PyTorch does).
 *   - Tensors are dense row-major fp32 unless stated: x is (N, C, T, V) exactly as
 *     unit_agcn.forward receives it.
 *   - Return value: 0 on success, a negative stgcn_status otherwise; the message for the
 *     calling thread is available from stgcn_last_error().  Nothing throws or exits.
 *   - `flags`: low 4 bits select the arithmetic of the temporal-conv contraction
 *     (stgcn_math); STGCN_OUT_BF16 stores the final activation as bf16.
 */
#ifndef STGCN_HIP_H
#define STGCN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STGCN_ABI_VERSION 8

typedef enum {
    STGCN_OK = 0,
    STGCN_ERR_ARG = -1,          /* null pointer / non-positive dimension */
    STGCN_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels cover  */
    STGCN_ERR_WORKSPACE = -3,    /* caller workspace too small            */
    STGCN_ERR_HIP = -4           /* a HIP runtime call failed             */
} stgcn_status;

typedef enum {
    STGCN_MATH_F32 = 0,      /* v_mfma_f32_32x32x2_f32: exact fp32 fma chain (default)      */
    STGCN_MATH_BF16X3 = 1,   /* fp32 split into bf16 hi+lo, 3 bf16 MFMAs, fp32 accumulate   */
    STGCN_MATH_BF16 = 2,     /* operands rounded to bf16, fp32 accumulate                   */
    STGCN_MATH_F32_VALU = 3  /* plain fp32 VALU kernels (any shape; on-device cross-check)  */
} stgcn_math;

#define STGCN_MATH_MASK 0xFu
#define STGCN_OUT_BF16 0x10u
#define STGCN_RAW 0x20u /* stgcn_tcn_*: store scale-folded conv + shift WITHOUT the ReLU (pre-activation) */
/* Layout fusion with the callers either side of the stem (stgcn_stem_* entry points only):
 *   STGCN_IN_NTVC : x is (N,T,V,Cin), the layout the data loader delivers — replaces the permute + contiguous copy
 *                   at model/AltFormer/ST_GCN_AltFormer.py:62-68;
 *   STGCN_OUT_NTVC: out is (N,T,V,C), channels last — `rearrange 'b c f p -> (b f) p c'` (model_ST.py:152) and
 *                   `'b c f p -> (b p) f c'` (model_TS.py:161) then are views, not copies. */
#define STGCN_IN_NTVC 0x40u
#define STGCN_OUT_NTVC 0x80u
/* stgcn_patch_embed: rows ordered (clip, joint, frame) — model_TS.py:161 — instead of (clip, frame, joint) */
#define STGCN_EMBED_TS 0x100u
/* stgcn_*_forward_train / stgcn_*_backward_train: BatchNorm on its RUNNING statistics (module.eval() with autograd — frozen-
 * BatchNorm fine-tuning, saliency; the reference's nn.BatchNorm2d stays differentiable in eval mode, model/net.py:52,
 * model/unit_agcn.py:54,91).  Forward: normalises with running_mean / running_var, leaves both untouched, and saves them as
 * (mean, invstd).  Backward: mean and invstd are constants — dz = gamma*invstd*g, dgamma = sum g*xhat, dbeta = sum g. */
#define STGCN_BN_FROZEN 0x200u
/* stgcn_stem_* entry points, with STGCN_MATH_BF16X3: where the kernel covers the shape (V <= ~25 joints, C % 128 == 0, K = 9)
 * the fused stem's temporal conv runs as fp16 x fp16 plus two block-scaled e4m3 residual products
 * (v_mfma_scale_f32_16x16x128_f8f6f4) instead of three bf16 products: the same 1e-4 contract (measured 2-4e-5 of max|ref|,
 * tools/math_error_2term.py) at two thirds of the matrix-core cycles.  Everything else about the call is unchanged; shapes
 * outside the kernel run the three-bf16 kernels.  The flag must be the same in stgcn_stem_prep_bytes / _prepare /
 * _ws_bytes / _attention / _tail (the prep blob carries a third weight packing, the workspace a per-clip bound). */
#define STGCN_STEM_F16MX 0x400u
/* stgcn_tcn_forward[_packed] with STGCN_MATH_F32_VALU: the 1-D convolution runs along the JOINT axis (Unit2D(dim=3),
 * model/net.py:28-36: kernel (1,K), padding (0,pad), stride (1,stride)); y is (N,Cout,T,V_out), V_out = (V+2*pad-K)/stride+1.
 * Inference only; no reference model builds this variant (plain-FMA kernel). */
#define STGCN_CONV_ALONG_V 0x800u

int stgcn_version(void);
const char *stgcn_last_error(void);

/* ---- eval-mode BatchNorm folding -------------------------------------------------------
 * scale[c] = weight[c] / sqrt(running_var[c] + eps)
 * shift[c] = bias[c] + (conv_bias[c] - running_mean[c]) * scale[c]      (conv_bias may be NULL)
 */
int stgcn_bn_fold(const float *weight, const float *bias, const float *running_mean,
                  const float *running_var, const float *conv_bias, float eps, float *scale,
                  float *shift, int C, void *stream);

/* ---- adaptive graph convolution (unit_agcn) -------------------------------------------
 * A_eff (S,V,V) = self.A + self.PA (unit_agcn.py:75-76), any dense values.
 * Wa,Wb (S,inter_c,Cin), ba,bb (S,inter_c)  : conv_a / conv_b 1x1 weights and biases
 * Wd (S,Cout,Cin), bd (S,Cout)              : conv_d
 * Wdown (Cout,Cin), bdown (Cout)            : down[0]; pass NULL for both when Cin == Cout
 *                                             (the reference then adds x itself, :57-58)
 * bn_scale/bn_shift, down_scale/down_shift  : folded eval BatchNorm of self.bn / down[1]
 *
 * stgcn_agcn_attention: P[n,s,v,w] = softmax_v( sum_{c,t} a[c,t,v] b[c,t,w] / (inter_c*T) ) + A_eff[s,v,w]
 *                       (unit_agcn.py:81-85); P is (N,S,V,V).
 * stgcn_agcn_forward  : the whole forward; P_ws (N,S,V,V) is caller workspace and holds P on return;
 *                       y is (N,Cout,T,V).
 */
int stgcn_agcn_attention(const float *x, const float *A_eff, const float *Wa, const float *ba,
                         const float *Wb, const float *bb, float *P, int N, int Cin, int T, int V,
                         int inter_c, int subsets, void *stream);

int stgcn_agcn_forward(const float *x, const float *A_eff, const float *Wa, const float *ba,
                       const float *Wb, const float *bb, const float *Wd, const float *bd,
                       const float *Wdown, const float *bdown, const float *bn_scale,
                       const float *bn_shift, const float *down_scale, const float *down_shift,
                       float *P_ws, float *y, int N, int Cin, int Cout, int T, int V, int inter_c,
                       int subsets, void *stream);

/* ---- temporal conv block (Unit2D, dim=2) -----------------------------------------------
 * W (Cout,Cin,K) = conv.weight with the trailing 1 squeezed; pad = (K-1)/2; T_out = (T+2*pad-K)/stride+1.
 * scale/shift = folded BN (shift includes the conv bias).  y is (N,Cout,T_out,V), fp32 or bf16
 * (STGCN_OUT_BF16).
 *
 * The contraction reads W re-ordered into MFMA fragment order with `scale` multiplied in
 * ("packed").  Pack once per weight update with stgcn_tcn_pack into a buffer of
 * stgcn_tcn_packed_bytes() bytes, then call stgcn_tcn_forward_packed per batch; or call
 * stgcn_tcn_forward, which packs into `ws` (>= stgcn_tcn_packed_bytes) on every call.
 */
size_t stgcn_tcn_packed_bytes(int Cin, int Cout, int K, unsigned flags);
/* 1 when the matrix-core kernel of `flags` covers the shape, 0 when only STGCN_MATH_F32_VALU does */
int stgcn_tcn_supported(int Cin, int Cout, int T, int V, int K, int stride, unsigned flags);
int stgcn_tcn_pack(const float *W, const float *scale, void *Wp, int Cin, int Cout, int K,
                   unsigned flags, void *stream);
int stgcn_tcn_forward_packed(const float *x, const void *Wp, const float *shift, void *y, int N,
                             int Cin, int Cout, int T, int V, int K, int stride, unsigned flags,
                             void *stream);
int stgcn_tcn_forward(const float *x, const float *W, const float *scale, const float *shift,
                      void *y, int N, int Cin, int Cout, int T, int V, int K, int stride, void *ws,
                      size_t ws_bytes, unsigned flags, void *stream);

/* ---- fused stem: tcn0(gcn0(x)) ----------------------------------------------------------
 * Same operands as the two calls above (the temporal conv has Cin = Cout = C, stride 1).
 * The (N,C,T,V) activation between the two modules is produced tile by tile in LDS and never
 * written to HBM.  `prep` holds the folded graph-conv weights and the packed temporal weights:
 * fill it with stgcn_stem_prepare (stgcn_stem_prep_bytes bytes) once per weight update.
 */
size_t stgcn_stem_prep_bytes(int Cin, int C, int K, int subsets, unsigned flags);
/* 1 when the fused kernel covers the shape (else run stgcn_agcn_forward + stgcn_tcn_forward) */
int stgcn_stem_supported(int Cin, int C, int T, int V, int K, int subsets, unsigned flags);
int stgcn_stem_prepare(const float *Wd, const float *bd, const float *Wdown, const float *bdown,
                       const float *bn_scale, const float *bn_shift, const float *down_scale,
                       const float *down_shift, const float *Wt, const float *t_scale, void *prep,
                       int Cin, int C, int K, int subsets, unsigned flags, void *stream);
/* Workspace of the fused stem (caller-provided, stgcn_stem_ws_bytes bytes): the attention matrices
 * P (N,S,V,V) at offset 0 (valid on return) followed by per-pixel graph-conv features for the kernels
 * that consume them, or (STGCN_IN_NTVC on the other kernels) a channel-major copy of x. */
size_t stgcn_stem_ws_bytes(int N, int Cin, int C, int T, int V, int K, int subsets, unsigned flags);
/* Name of the kernel stgcn_stem_tail_prepared launches for the shape ("stem_bf16_v6_kernel", "stem_bf16_v4_kernel",
 * "stem_mfma_bf16_kernel", "stem_mfma_f32_kernel"; "" when no fused kernel covers it) — for profilers and benchmarks
 * that match kernel names in rocprofv3 output. */
const char *stgcn_stem_kernel_name(int Cin, int C, int T, int V, int K, int subsets, unsigned flags);
/* 1 when the large-tile persistent kernel (which reads the feature part of the workspace) serves the shape */
int stgcn_stem_features_used(int Cin, int C, int T, int V, int K, int subsets, unsigned flags);
/* The two halves of stgcn_stem_forward_prepared, separately launchable (e.g. to time them):
 * stgcn_stem_attention fills the workspace; stgcn_stem_tail_prepared launches only the fused
 * graph-conv + temporal-conv kernel on a filled workspace. */
int stgcn_stem_attention(const float *x, const float *A_eff, const float *Wa, const float *ba,
                         const float *Wb, const float *bb, void *ws, size_t ws_bytes, int N, int Cin,
                         int C, int T, int V, int inter_c, int subsets, int K, unsigned flags,
                         void *stream);
int stgcn_stem_tail_prepared(const float *x, const void *ws, size_t ws_bytes, const void *prep,
                             const float *t_shift, void *out, int N, int Cin, int C, int T, int V,
                             int subsets, int K, unsigned flags, void *stream);
int stgcn_stem_forward_prepared(const float *x, const float *A_eff, const float *Wa,
                                const float *ba, const float *Wb, const float *bb,
                                const void *prep, const float *t_shift, void *ws, size_t ws_bytes,
                                void *out, int N, int Cin, int C, int T, int V, int inter_c,
                                int subsets, int K, unsigned flags, void *stream);

/* ---- training-mode forward (batch-statistics BatchNorm) ------------------------------------------
 * Same math as the eval entry points, but every BatchNorm2d normalises with the statistics of the
 * batch over (N,T,V) and updates its running buffers in place exactly like torch (momentum, unbiased
 * variance); num_batches_tracked is the caller's to increment.  bn_* / dbn_* are the raw BatchNorm
 * tensors (weight, bias, running_mean, running_var), not folded scale/shift.  `ws` is caller workspace
 * of stgcn_*_train_ws_bytes bytes.
 *   stgcn_agcn_forward_train <- model/unit_agcn.py:73-93 with self.training (P_ws, y as in stgcn_agcn_forward)
 *   stgcn_tcn_forward_train  <- model/net.py:47-57 with self.training (dropout p = 0)
 */
/* materialise = 1: room for the two pre-BatchNorm branches (when save_zm/save_zd are wanted, or the shape is outside
 * the stem class); 0: the stem class (C_in=3, 3 subsets, down branch) derives the batch statistics from the moments of its
 * 12 per-pixel features and never writes the branches — scratch of a few hundred KB. */
size_t stgcn_agcn_train_ws_bytes(int N, int Cin, int Cout, int T, int V, int subsets, int materialise);
int stgcn_agcn_forward_train(const float *x, const float *A_eff, const float *Wa, const float *ba,
                             const float *Wb, const float *bb, const float *Wd, const float *bd,
                             const float *Wdown, const float *bdown, const float *bn_weight,
                             const float *bn_bias, float *bn_running_mean, float *bn_running_var,
                             const float *dbn_weight, const float *dbn_bias, float *dbn_running_mean,
                             float *dbn_running_var, float momentum, float eps, float *P_ws, void *ws,
                             size_t ws_bytes, float *y, float *save_zm, float *save_zd, float *save_stats,
                             int N, int Cin, int Cout, int T, int V, int inter_c, int subsets,
                             unsigned flags /* 0 or STGCN_BN_FROZEN (then size ws with materialise = 1) */, void *stream);
/* save_zm / save_zd (N,Cout,T,V): the two pre-BatchNorm branches (sum_s conv_d_s(x P_s), conv_down(x)) — asking for
 * them selects the materialising path; save_stats (STGCN_AGCN_SAVE_STATS_FLOATS(Cout) floats, 8-byte aligned): batch
 * mean, invstd of `bn`, then of the down BatchNorm (4*Cout), followed — on the moments path only — by the 63 feature
 * moments as doubles, which the stem-class backward reads back, and a validity mark in one of the two spare floats
 * behind them (ABI 8): the materialising path clears that block, and the moment-form backward answers NaN in every
 * gradient when it is handed statistics without the mark (a forward that saved the branches, followed by a backward
 * call with zm == NULL) instead of working from uninitialised moments.  All optional (NULL). */
#define STGCN_AGCN_SAVE_STATS_FLOATS(Cout) (4 * (Cout) + 128)
/* recompute: bit 0 = zm / zd are not supplied (the forward ran the moments path; the generic path rebuilds them in the
 * workspace, the stem-class path never needs them); bit 1 = size for the generic path — needed when an input gradient
 * is wanted (dx != NULL), with the identity residual, or for a shape outside the stem class (the call below picks the
 * path from its arguments; size for what you will ask).  0 for an invalid size. */
size_t stgcn_agcn_backward_ws_bytes(int N, int Cin, int Cout, int T, int V, int subsets, int recompute);
/* Gradients of every parameter of unit_agcn (model/unit_agcn.py:35-62) from dy (N,Cout,T,V) in training mode:
 * dWa/dWb (S,inter_c,Cin), dba/dbb (S,inter_c), dWd (S,Cout,Cin), dbd (S,Cout), dWdown (Cout,Cin), dbdown, the
 * two BatchNorms' dgamma/dbeta (main, then "dd" = down), dPA (S,V,V), and — optionally — dx (N,Cin,T,V), the gradient
 * of the input that the deeper TCN_GCN_unit layers need (model/ST_TR/ST_TR_new.py:355-372); NULL when x is data.
 * Identity residual (Cin == Cout, unit_agcn.py:57-58): pass NULL for Wdown, bdown, the down BatchNorm tensors, zd and
 * the four down-gradient outputs.  zm / zd: the branches the forward saved, or NULL.  y: the forward's output.
 * Two implementations:
 *   - the stem's shape class (Cin = 3, 3 subsets, Cout in {64,128,256}, down branch, no dx) with zm == zd == NULL and
 *     y != NULL — i.e. after a moments-path forward whose save_stats carries the feature moments: ONE pass over dy / y
 *     (fp32 MFMA), everything else from moments (csrc/agcn_backward.hip; no branch is rebuilt, no statistics pass);
 *   - a chain of strided batched fp32-MFMA GEMMs for everything else (V <= 64, inter_c <= Cout/4); y is not read. */
int stgcn_agcn_backward_train(const float *x, const float *A_eff, const float *Wa, const float *ba,
                              const float *Wb, const float *bb, const float *Wd, const float *bd,
                              const float *Wdown, const float *bdown, const float *P, const float *zm,
                              const float *zd, const float *bn_weight, const float *bn_bias,
                              const float *dbn_weight, const float *dbn_bias, const float *save_stats,
                              const float *y, const float *dy, float *dWa, float *dba, float *dWb, float *dbb, float *dWd,
                              float *dbd, float *dWdown, float *dbdown, float *dgamma, float *dbeta,
                              float *ddgamma, float *ddbeta, float *dPA, float *dx, void *ws, size_t ws_bytes,
                              int N, int Cin, int Cout, int T, int V, int inter_c, int subsets,
                              unsigned flags /* 0 or STGCN_BN_FROZEN (generic path: size ws with recompute bit 1) */, void *stream);
size_t stgcn_tcn_train_ws_bytes(int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags);
/* save_z (N,Cout,T_out,V), save_mean, save_invstd (Cout): optional outputs for the backward — the raw
 * convolution conv_t(x)+b and the batch statistics, torch's save_mean / save_invstd.  NULL: not kept. */
int stgcn_tcn_forward_train(const float *x, const float *W, const float *conv_bias,
                            const float *bn_weight, const float *bn_bias, float *bn_running_mean,
                            float *bn_running_var, float momentum, float eps, void *ws, size_t ws_bytes,
                            float *y, float *save_z, float *save_mean, float *save_invstd, int N, int Cin,
                            int Cout, int T, int V, int K, int stride, unsigned flags, void *stream);

/* ---- backward of the training-mode blocks ------------------------------
 * What autograd derives for y = relu(BatchNorm_batch(conv_t(x) + b)) (model/net.py:47-57 under
 * train_sttran.py:185-191) from dy (N,Cout,T_out,V):
 *   dW (Cout,Cin,K), dbias (Cout, NULL when the conv has no bias), dgamma / dbeta (Cout) of the
 *   BatchNorm, and dx (N,Cin,T,V; NULL when the input needs no gradient).
 * z, save_mean, save_invstd are the tensors stgcn_tcn_forward_train saved.  The weight gradient
 * runs on the bf16 matrix cores with the arithmetic of `flags` (BF16X3: fp32 contract) where the
 * shape allows (stride 1, Cout%128==0, Cin%32==0, K<=9), on plain fp32 FMAs otherwise; the input
 * gradient of a stride-1 block is the forward kernel on the flipped weights. */
size_t stgcn_tcn_backward_ws_bytes(int N, int Cin, int Cout, int T, int V, int K, int stride, unsigned flags);
int stgcn_tcn_backward_train(const float *x, const float *W, const float *z, const float *bn_weight,
                             const float *bn_bias, const float *save_mean, const float *save_invstd,
                             const float *dy, float *dx, float *dW, float *dbias, float *dgamma,
                             float *dbeta, void *ws, size_t ws_bytes, int N, int Cin, int Cout, int T,
                             int V, int K, int stride, unsigned flags, void *stream);

/* ---- first patch embedding of the transformer heads, on the stem's output -----------------------------
 * What the callers do next with z = tcn0(gcn0(x)):
 *   ST head: rearrange 'b c f p -> (b f) p c', Spatial_patch_to_embedding = nn.Linear(C, E), += Spatial_pos_embed
 *            (model/AltFormer/model_ST.py:101-103,152-155)                      -> out ((N*T), V, E)
 *   TS head: rearrange 'b c f p -> (b p) f c', temporal_patch_to_embedding = nn.Linear(C, E), += Temporal_pos_embed
 *            (model/AltFormer/model_TS.py:110-111,161-163; STGCN_EMBED_TS)      -> out ((N*V), T, E)
 * as ONE strided GEMM per clip: the rearrange is the addressing of the product, no copy of z is made.
 * z: (N,C,T,V), or (N,T,V,C) with STGCN_IN_NTVC (what stgcn_stem_* writes with STGCN_OUT_NTVC);
 * W (E,C) = Linear.weight, b (E) = Linear.bias, pos: (V,E) [ST] / (T,E) [TS] or NULL.  fp32 throughout (exact fp32
 * matrix-core arithmetic).  out fp32, dense. */
int stgcn_patch_embed(const float *z, const float *W, const float *b, const float *pos, float *out, int N, int C,
                      int E, int T, int V, unsigned flags, void *stream);

/* ---- data-parallel harness -------------------------------------------------------------------
 * Per-rank reductions that the ranks all-reduce once per step — the data-parallel form of the
 * reference's accuracy reduction get_acc (SHREC/ST_TS/train_sttran.py:105-109: np.argmax of the
 * logits on the host, compared with the labels, summed): stats[0] = n_local, stats[1] = sum probe,
 * stats[2] = sum probe^2, stats[3] = #{n : argmax_c logits[n][c] == labels[n]}, with
 * probe[n][c] = out[n][c][0][0] = element n*clip_stride + c*chan_stride of `out` (ABI 8: element strides instead of a
 * plane size, so the channels-last result of STGCN_OUT_NTVC — clip_stride = T*V*C, chan_stride = 1 — is probed in place;
 * a dense (N,C,T,V) tensor has clip_stride = C*T*V, chan_stride = T*V).
 * logits (n_logits, classes) fp32 and labels (n_logits) int64 are optional (NULL: stats[3] = 0);
 * pred (n_logits) int64, optional, receives the class indices.  argmax follows numpy: the lowest
 * index among equal maxima, a NaN is the maximum.  `out` may be NULL when only the count is wanted.
 * One launch, one workgroup. */
int stgcn_step_stats(const void *out, int out_is_bf16, float *stats, int N, int C, long clip_stride,
                     long chan_stride, float n_local, const float *logits, const long long *labels, long long *pred,
                     int n_logits, int classes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STGCN_HIP_H */
