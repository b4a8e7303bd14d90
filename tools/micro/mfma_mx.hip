// Micro-benchmark + layout check for the candidate arithmetic of profiles/r03_math_error_2term.txt: the leading term on the
// fp16 matrix cores and the two residual terms as block-scaled fp8 (v_mfma_scale_f32_16x16x128_f8f6f4), against the shipped
// three bf16 terms.  One wave per SIMD, a wave owns 8 x 4 accumulator blocks of 16 x 16 (KF6's geometry), operands re-read
// from LDS with ds_read_b128 on random data, every CU busy.
//   (1) layout: lane l of an fp8 operand holds row/column l&15, K = 32*(l>>4) .. +31 (32 bytes, K ascending), and byte
//       `opsel` of its scale register is the E8M0 scale of that (row, 32-wide K block) — checked against a CPU product;
//   (2) rate: per K = 128 and block, 12 bf16 MFMAs (16x16x32)  vs  4 fp16 MFMAs + 2 scaled fp8 MFMAs (16x16x128).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mx.hip -o gpurun_out/mfma_mx && gpurun_out/mfma_mx
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x8 = __attribute__((ext_vector_type(8))) int;

__global__ void layout_kernel(const i32x8 *a, const i32x8 *b, const int *sa, const int *sb, float *d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) d[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];   // standard C/D map: col = l&15, row = 4*(l>>4)+r
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

// MODE 0: bf16x3 (12 x 16x16x32 per K=128 and block); MODE 1: fp16 + 2 scaled fp8; MODE 2: fp16 only (4 per K=128)
template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(const uint4 *__restrict__ src, float *__restrict__ out, int iters) {
    extern __shared__ uint4 lds[];   // 96 KiB of operand fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 6144; e += 256) lds[e] = src[(blockIdx.x * 6144 + e) % (1 << 20)];
    __syncthreads();
    const uint4 *base = lds + lane;
    f32x4 acc[8][4] = {};
    for (int it = 0; it < iters; ++it) {         // one iteration = K 128 for the wave's 128 x 64 tile
        const int o = ((it * 4 + wave) * 448) % (6144 - 3072);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                uint4 ah[8], al[8], bh[4], bl[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) { ah[i] = base[o + (ks * 24 + i) * 64 % 3072]; al[i] = base[o + (ks * 24 + 8 + i) * 64 % 3072]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { bh[i] = base[o + (ks * 24 + 16 + i) * 64 % 3072]; bl[i] = base[o + (ks * 24 + 20 + i) * 64 % 3072]; }
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, bl[n]), acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al[m]), __builtin_bit_cast(bf16x8, bh[n]), acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[m]), __builtin_bit_cast(bf16x8, bh[n]), acc[m][n], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {     // leading term: fp16 x fp16, K = 32 per MFMA
                uint4 ah[8], bh[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) ah[i] = base[o + (ks * 12 + i) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) bh[i] = base[o + (ks * 12 + 8 + i) * 64];
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah[m]), __builtin_bit_cast(f16x8, bh[n]), acc[m][n], 0, 0, 0);
            }
            if constexpr (MODE == 1) {
                // residual terms: W_lo8 * y_hi8 and W_hi8 * y_lo8, K = 128 per MFMA: 32 B per lane and operand (two b128 reads)
                i32x8 wl[8], wh[8], yh[4], yl[4];
                auto rd2 = [&](int f) {
                    const uint4 p = base[o + (48 + 2 * f) * 64 % 3072], q = base[o + (48 + 2 * f + 1) * 64 % 3072];
                    // keep fp8 bytes finite: clear the top exponent bit of every byte (no NaN encodings, magnitudes < 2)
                    const unsigned msk = 0xbfbfbfbfu;
                    return i32x8{(int)(p.x & msk), (int)(p.y & msk), (int)(p.z & msk), (int)(p.w & msk),
                                 (int)(q.x & msk), (int)(q.y & msk), (int)(q.z & msk), (int)(q.w & msk)};
                };
#pragma unroll
                for (int i = 0; i < 8; ++i) { wl[i] = rd2(i); wh[i] = rd2(8 + i); }
#pragma unroll
                for (int i = 0; i < 4; ++i) { yh[i] = rd2(16 + i); yl[i] = rd2(20 + i); }
                const int sc = 0x7f7f7f7f;       // scales 2^0
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wl[m], yh[n], acc[m][n], 0, 0, 0, sc, 0, sc);
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wh[m], yl[n], acc[m][n], 0, 0, 0, sc, 0, sc);
                    }
            }
        }
        if ((it & 1) == 1) __syncthreads();
    }
    float s = 0.f;
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main() {
    // ---- (1) layout check ------------------------------------------------------------------------------------------
    {
        std::vector<unsigned char> A(16 * 128), B(128 * 16), SA(16 * 4), SB(16 * 4);
        srand(7);
        auto rnd8 = [] { unsigned char v; do v = rand() & 0xff; while ((v & 0x7f) == 0x7f || ((v >> 3) & 15) > 9); return v; };
        for (auto &v : A) v = rnd8();
        for (auto &v : B) v = rnd8();
        for (auto &v : SA) v = 125 + rand() % 5;
        for (auto &v : SB) v = 125 + rand() % 5;
        std::vector<int> ha(64 * 8), hb(64 * 8), hsa(64), hsb(64);
        for (int l = 0; l < 64; ++l) {
            unsigned char *pa = (unsigned char *)&ha[l * 8], *pb = (unsigned char *)&hb[l * 8];
            for (int j = 0; j < 32; ++j) {
                pa[j] = A[(l & 15) * 128 + 32 * (l >> 4) + j];           // A[row][k]
                pb[j] = B[(32 * (l >> 4) + j) * 16 + (l & 15)];           // B[k][col]
            }
            hsa[l] = SA[(l & 15) * 4 + (l >> 4)] | 0x55aa5500;           // byte 0 = the scale (opsel 0); other bytes junk
            hsb[l] = SB[(l & 15) * 4 + (l >> 4)] | 0x33cc3300;
        }
        int *da, *db, *dsa, *dsb; float *dd;
        hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
        hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, (const i32x8 *)da, (const i32x8 *)db, dsa, dsb, dd);
        std::vector<float> D(256);
        hipMemcpy(D.data(), dd, 1024, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int k = 0; k < 128; ++k)
                    ref += (double)e4m3(A[i * 128 + k]) * ldexp(1.0, SA[i * 4 + k / 32] - 127) * e4m3(B[k * 16 + j]) * ldexp(1.0, SB[j * 4 + k / 32] - 127);
                worst = fmax(worst, fabs(ref - D[i * 16 + j]));
                scale = fmax(scale, fabs(ref));
            }
        printf("layout check (e4m3 x e4m3, per-lane scales, opsel 0): max|err| %.3e of max|ref| %.3e  -> %s\n", worst, scale,
               worst <= 1e-5 * scale ? "layout as assumed" : "LAYOUT DIFFERS");
    }
    // ---- (2) rate ----------------------------------------------------------------------------------------------------
    const size_t n = 1 << 20;
    std::vector<unsigned> h(n * 4);
    srand(1);
    for (auto &v : h) {   // random 16-bit pairs with sane exponents (valid as bf16 and as fp16: 0x3c00 +- ...)
        unsigned a = 0x3800 + (rand() & 0x7ff) + ((rand() & 1) << 15), b = 0x3800 + (rand() & 0x7ff) + ((rand() & 1) << 15);
        v = a | (b << 16);
    }
    uint4 *src; float *out;
    hipMalloc(&src, n * 16); hipMalloc(&out, 256 * 256 * 4);
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    const int iters = 288 * 8;       // K = 128 steps: ~ what one CU does for 32 tiles of the stem (9 per tile and chunk pair)
    hipFuncSetAttribute((const void *)loop_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute((const void *)loop_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipFuncSetAttribute((const void *)loop_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](int mode) {
        if (mode == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(256), dim3(256), 98304, 0, src, out, iters);
        else if (mode == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(256), dim3(256), 98304, 0, src, out, iters);
        else hipLaunchKernelGGL(loop_kernel<2>, dim3(256), dim3(256), 98304, 0, src, out, iters);
    };
    const char *names[3] = {"bf16x3 (12 x 16x16x32 bf16 per K=128)", "fp16 + 2 scaled e4m3 (4 x 16x16x32 f16 + 2 x 16x16x128)", "fp16 alone (4 x 16x16x32 f16)"};
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 5; ++w) launch(mode);
            hipEventRecord(e0);
            const int L = 30;
            for (int w = 0; w < L; ++w) launch(mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= L;
            const double alg = 256.0 * 4 * iters * 32 * (2.0 * 16 * 16 * 128);   // algorithmic FLOPs: one product per K element
            printf("%-58s: %.3f ms  %.0f TFLOP/s algorithmic (x3 issued for bf16x3)\n", names[mode], ms, alg / ms / 1e9);
        }
    return 0;
}
