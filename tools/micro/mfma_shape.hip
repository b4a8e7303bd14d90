// Micro-benchmark: the consumer loop of the fused stem kernel reduced to its operand traffic — LDS fragment reads +
// bf16 MFMAs on random data, 8 waves per CU, every CU busy — once with v_mfma_f32_32x32x16_bf16 (2x2 blocks per wave,
// 8 ds_read_b128 per 12 MFMAs) and once with v_mfma_f32_16x16x32_bf16 (4x4 blocks, 16 reads per 48 MFMAs): same
// bytes, same FLOPs.  Question: which shape delivers more FLOP/s under the clock the chip holds (DVFS)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o gpurun_out/mfma_shape && gpurun_out/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int SHAPE>
__global__ __launch_bounds__(512) void loop_kernel(const uint4 *__restrict__ src, float *__restrict__ out, int iters) {
    extern __shared__ uint4 lds[];   // 96 KiB of fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 6144; e += 512) lds[e] = src[(blockIdx.x * 6144 + e) % (1 << 20)];
    __syncthreads();
    const uint4 *base = lds + lane;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
            const int o = ((it * 8 + wave) * 512) % (6144 - 512);
            uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = base[o + i * 64]; al[i] = base[o + 128 + i * 64];
                bh[i] = base[o + 256 + i * 64]; bl[i] = base[o + 384 + i * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, bl[n]), acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[m]), __builtin_bit_cast(bf16x8, bh[n]), acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, bh[n]), acc[m][n], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if ((it & 3) == 3) __syncthreads();
        }
        float s = 0.f;
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        f32x4 acc[4][4] = {};
        for (int it = 0; it < iters; it += 2) {   // one K=32 step = two K=16 steps of the other kernel
            const int o = ((it * 8 + wave) * 512) % (6144 - 1024);
            uint4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = base[o + i * 64]; al[i] = base[o + 256 + i * 64];
                bh[i] = base[o + 512 + i * 64]; bl[i] = base[o + 768 + i * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, bl[n]), acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[m]), __builtin_bit_cast(bf16x8, bh[n]), acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, bh[n]), acc[m][n], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if ((it & 3) == 2) __syncthreads();
        }
        float s = 0.f;
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
        out[blockIdx.x * 512 + tid] = s;
    }
}

int main() {
    const size_t n = 1 << 20;
    std::vector<unsigned> h(n * 4);
    srand(1);
    for (auto &v : h) {   // random bf16 pairs with sane exponents
        unsigned a = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15), b = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
        v = a | (b << 16);
    }
    uint4 *src; float *out;
    hipMalloc(&src, n * 16); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    const int iters = 1152 * 8;   // ~ what one CU does for 8 tiles
    hipFuncSetAttribute((const void *)loop_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute((const void *)loop_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16}) {
            for (int w = 0; w < 3; ++w) {
                if (shape == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(256), dim3(512), 98304, 0, src, out, iters);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(256), dim3(512), 98304, 0, src, out, iters);
            }
            hipEventRecord(e0);
            const int L = 20;
            for (int w = 0; w < L; ++w) {
                if (shape == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(256), dim3(512), 98304, 0, src, out, iters);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(256), dim3(512), 98304, 0, src, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= L;
            const double flops = 256.0 * 8 * iters * 12 * 32768.0;   // per launch (both shapes issue the same FLOPs)
            printf("shape %2d: %.3f ms  %.0f TFLOP/s issued\n", shape, ms, flops / ms / 1e9);
        }
    return 0;
}
