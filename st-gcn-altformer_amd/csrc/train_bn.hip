// Training-mode BatchNorm2d pieces (nn.BatchNorm2d at model/unit_agcn.py:54,60 and model/net.py:40 with
// module.training == True): batch statistics over (N,T,V) per channel, the running-buffer update torch performs
// (momentum 0.1, unbiased variance), and the elementwise normalise + residual + ReLU.
//
// The pre-activation tensors are produced by the same kernels as in eval mode run in "raw" mode (unit scale,
// zero shift, no ReLU); sums are accumulated in fp64 (one atomicAdd pair per workgroup and channel), so the
// statistics do not depend on the reduction order beyond fp64 rounding.
#include "common.h"

namespace stgcn {

namespace {

// sums[c] += sum z[n][c][:],  sums[C + c] += sum z^2     grid = (chunks, C)
__global__ __launch_bounds__(256) void bn_batch_stats_kernel(const float *__restrict__ z, double *__restrict__ sums,
                                                              int N, int C, size_t plane) {
    const int c = blockIdx.y;
    const ChannelRows it(N, plane);
    double s1 = 0.0, s2 = 0.0;
    for (int n = it.n_lo; n < it.n_hi; ++n) {
        const float *zr = z + ((size_t)n * C + c) * plane;
        float a1 = 0.f, a2 = 0.f;                  // fp32 within one (clip, channel, thread) strip, fp64 across
        if (it.vec) {
            for (size_t p = threadIdx.x * 4; p < plane; p += 1024) {
                const float4 v = *reinterpret_cast<const float4 *>(zr + p);
                a1 += (v.x + v.y) + (v.z + v.w);
                a2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, a2))));
            }
        } else {
            for (size_t p = threadIdx.x; p < plane; p += 256) {
                const float v = zr[p];
                a1 += v;
                a2 = fmaf(v, v, a2);
            }
        }
        s1 += (double)a1;
        s2 += (double)a2;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o, 64);
        s2 += __shfl_down(s2, o, 64);
    }
    __shared__ double red[2][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s1; red[1][w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&sums[C + c], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// batch mean / biased variance -> (scale, shift) of  y = z*scale + shift ; running buffers updated like torch:
//   running = (1-m)*running + m*batch   with the UNBIASED variance (count/(count-1))
__global__ void bn_train_finalize_kernel(const double *__restrict__ sums, double count, const float *__restrict__ weight,
                                         const float *__restrict__ bias, float *__restrict__ running_mean,
                                         float *__restrict__ running_var, float momentum, float eps,
                                         float *__restrict__ scale, float *__restrict__ shift, int C,
                                         float *__restrict__ save_mean, float *__restrict__ save_invstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double mean = sums[c] / count;
    double var = sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float inv = 1.f / sqrtf((float)var + eps);
    const float s = weight[c] * inv;
    scale[c] = s;
    shift[c] = bias[c] - (float)mean * s;
    if (save_mean) save_mean[c] = (float)mean;       // what the backward needs (torch's save_mean / save_invstd)
    if (save_invstd) save_invstd[c] = inv;
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

// STGCN_BN_FROZEN: the running statistics stand in for the batch's; nothing is updated
__global__ void bn_frozen_finalize_kernel(const float *__restrict__ weight, const float *__restrict__ bias,
                                          const float *__restrict__ running_mean, const float *__restrict__ running_var,
                                          float eps, float *__restrict__ scale, float *__restrict__ shift, int C,
                                          float *__restrict__ save_mean, float *__restrict__ save_invstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float mean = running_mean[c], inv = 1.f / sqrtf(running_var[c] + eps);
    const float s = weight[c] * inv;
    scale[c] = s;
    shift[c] = bias[c] - mean * s;
    if (save_mean) save_mean[c] = mean;
    if (save_invstd) save_invstd[c] = inv;
}

// y = relu( za*sa[c] + ta[c] + r ),  r = zb*sb[c] + tb[c]  (sb given), = zb (sb NULL, identity residual), = 0 (zb NULL)
// grid = (chunks, C)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ za, const float *__restrict__ sa,
                                                        const float *__restrict__ ta, const float *__restrict__ zb,
                                                        const float *__restrict__ sb, const float *__restrict__ tb,
                                                        float *__restrict__ y, int N, int C, size_t plane) {
    const int c = blockIdx.y;
    const ChannelRows it(N, plane);
    const float s_a = sa[c], t_a = ta[c];
    const int mode = zb == nullptr ? 0 : (sb != nullptr ? 2 : 1);
    const float s_b = mode == 2 ? sb[c] : 1.f, t_b = mode == 2 ? tb[c] : 0.f;
    auto one = [&](float a, float b) {
        float v = fmaf(a, s_a, t_a);
        if (mode) v += fmaf(b, s_b, t_b);          // (mode 1: 1*b + 0, the same bits as v + b)
        return fmaxf(v, 0.f);
    };
    for (int n = it.n_lo; n < it.n_hi; ++n) {
        const size_t base = ((size_t)n * C + c) * plane;
        if (it.vec) {
            for (size_t p = threadIdx.x * 4; p < plane; p += 1024) {
                const float4 a = *reinterpret_cast<const float4 *>(za + base + p);
                const float4 b = mode ? *reinterpret_cast<const float4 *>(zb + base + p) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(y + base + p) = make_float4(one(a.x, b.x), one(a.y, b.y), one(a.z, b.z), one(a.w, b.w));
            }
        } else {
            for (size_t p = threadIdx.x; p < plane; p += 256) y[base + p] = one(za[base + p], mode ? zb[base + p] : 0.f);
        }
    }
}

// scale/shift of y = z*scale + shift from the saved batch statistics (backward: the ReLU mask needs the forward's y)
__global__ void bn_scale_shift_kernel(const float *__restrict__ weight, const float *__restrict__ bias,
                                      const float *__restrict__ mean, const float *__restrict__ invstd,
                                      float *__restrict__ scale, float *__restrict__ shift, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float s = weight[c] * invstd[c];
    scale[c] = s;
    shift[c] = bias[c] - mean[c] * s;
}

}  // namespace

int launch_bn_scale_shift(const float *weight, const float *bias, const float *mean, const float *invstd, float *scale,
                          float *shift, int C, hipStream_t st) {
    hipLaunchKernelGGL(bn_scale_shift_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, weight, bias, mean, invstd, scale,
                       shift, C);
    STGCN_LAUNCH_CHECK("bn_scale_shift_kernel");
    return STGCN_OK;
}

int launch_bn_batch_stats(const float *z, double *sums, int N, int C, size_t plane, hipStream_t st) {
    STGCN_HIP_CHECK(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st));
    hipLaunchKernelGGL(bn_batch_stats_kernel, dim3(bn_chunks(N, plane, C), C), dim3(256), 0, st, z, sums, N, C, plane);
    STGCN_LAUNCH_CHECK("bn_batch_stats_kernel");
    return STGCN_OK;
}

int launch_bn_train_finalize(const double *sums, double count, const float *weight, const float *bias,
                             float *running_mean, float *running_var, float momentum, float eps, float *scale,
                             float *shift, int C, hipStream_t st, float *save_mean, float *save_invstd) {
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, sums, count, weight, bias,
                       running_mean, running_var, momentum, eps, scale, shift, C, save_mean, save_invstd);
    STGCN_LAUNCH_CHECK("bn_train_finalize_kernel");
    return STGCN_OK;
}

int launch_bn_frozen_finalize(const float *weight, const float *bias, const float *running_mean, const float *running_var,
                              float eps, float *scale, float *shift, int C, hipStream_t st, float *save_mean,
                              float *save_invstd) {
    hipLaunchKernelGGL(bn_frozen_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, weight, bias, running_mean,
                       running_var, eps, scale, shift, C, save_mean, save_invstd);
    STGCN_LAUNCH_CHECK("bn_frozen_finalize_kernel");
    return STGCN_OK;
}

int launch_bn_apply(const float *za, const float *sa, const float *ta, const float *zb, const float *sb,
                    const float *tb, float *y, size_t total, int C, size_t plane, hipStream_t st) {
    const int N = (int)(total / ((size_t)C * plane));
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_chunks(N, plane, C), C), dim3(256), 0, st, za, sa, ta, zb, sb, tb, y, N, C, plane);
    STGCN_LAUNCH_CHECK("bn_apply_kernel");
    return STGCN_OK;
}

}  // namespace stgcn
