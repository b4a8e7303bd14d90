// Issue cost of v_mfma_f32_16x16x4_f32 on gfx950: a chain of dependent accumulations (what a 16 x 16 block of an fp32 product
// is) against 2 / 4 independent chains, one and two waves per SIMD.  s_memtime around 1024 MFMAs per chain.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_chain.hip -o gpurun_out/mfma_f32_chain && gpurun_out/mfma_f32_chain
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CH>
__global__ void k(float *out, unsigned long long *cyc, float a0, float b0) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
#define RUN(CH, THREADS)                                                                                  \
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(THREADS), 0, 0, out, cyc, 1.f, 2.f); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(THREADS), 0, 0, out, cyc, 1.f, 2.f); hipDeviceSynchronize(); \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                          \
    printf("%d chain(s), %d wave(s) per SIMD: %.1f cycles per MFMA of a wave\n", CH, THREADS / 256, (double)h / (1024.0 * CH));
    RUN(1, 256) RUN(2, 256) RUN(4, 256) RUN(1, 512) RUN(2, 512) RUN(4, 512)
    return 0;
}
