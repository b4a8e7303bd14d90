// Probe of v_mfma_scale_f32_16x16x128_f8f6f4's operand and scale layout (gfx950): see the printout.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x8 = __attribute__((ext_vector_type(8))) int;

template <int OPA, int OPB>
__global__ void k(const i32x8 *a, const i32x8 *b, const int *sa, const int *sb, float *d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, OPA, sa[l], OPB, sb[l]);
    for (int r = 0; r < 4; ++r) d[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}
int *da, *db, *dsa, *dsb; float *dd;
std::vector<float> run(const std::vector<int> &ha, const std::vector<int> &hb, const std::vector<int> &sa, const std::vector<int> &sb, int opa = 0, int opb = 0) {
    hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    if (opa == 0 && opb == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, (const i32x8 *)da, (const i32x8 *)db, dsa, dsb, dd);
    else if (opa == 1) hipLaunchKernelGGL((k<1, 0>), dim3(1), dim3(64), 0, 0, (const i32x8 *)da, (const i32x8 *)db, dsa, dsb, dd);
    else hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, (const i32x8 *)da, (const i32x8 *)db, dsa, dsb, dd);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dd, 1024, hipMemcpyDeviceToHost);
    return D;
}
int main() {
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
    const int ONE = 0x38383838;
    std::vector<int> ones(64 * 8, ONE), zeros(64 * 8, 0), s1(64, 0x7f7f7f7f);
    // (a) all ones, unit scales: every D = 128?
    { auto D = run(ones, ones, s1, s1); printf("(a) ones x ones, unit scales: D[0][0] = %g, D[15][15] = %g\n", D[0], D[255]); }
    // (b) random data, unit scales, assumed layout k = 32*(l>>4) + j for both operands
    {
        std::vector<unsigned char> A(16 * 128), B(128 * 16);
        srand(7);
        auto rnd8 = [] { unsigned char v; do v = rand() & 0xff; while ((v & 0x7f) == 0x7f || ((v >> 3) & 15) > 9); return v; };
        for (auto &v : A) v = rnd8();
        for (auto &v : B) v = rnd8();
        for (int hyp = 0; hyp < 2; ++hyp) {
            std::vector<int> ha(64 * 8), hb(64 * 8);
            for (int l = 0; l < 64; ++l) {
                unsigned char *pa = (unsigned char *)&ha[l * 8], *pb = (unsigned char *)&hb[l * 8];
                for (int j = 0; j < 32; ++j) {
                    const int kk = hyp == 0 ? 32 * (l >> 4) + j : (j < 16 ? 16 * (l >> 4) + j : 64 + 16 * (l >> 4) + (j - 16));
                    pa[j] = A[(l & 15) * 128 + kk];
                    pb[j] = B[kk * 16 + (l & 15)];
                }
            }
            auto D = run(ha, hb, s1, s1);
            double worst = 0, scale = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int kk = 0; kk < 128; ++kk) ref += (double)e4m3(A[i * 128 + kk]) * e4m3(B[kk * 16 + j]);
                worst = fmax(worst, fabs(ref - D[i * 16 + j])); scale = fmax(scale, fabs(ref));
            }
            printf("(b) random e4m3, unit scales, K hypothesis %d: max|err| %.3e of %.3e\n", hyp, worst, scale);
        }
    }
    // (c) which rows / how much does ONE lane's scale byte move?  ones x ones; lane la's A-scale byte `byte` = 128 (x2)
    for (int byte = 0; byte < 4; ++byte)
        for (int la : {0, 1, 16, 17, 32, 48, 63}) {
            std::vector<int> sa = s1;
            sa[la] = (sa[la] & ~(0xff << (8 * byte))) | (128 << (8 * byte));
            auto D = run(ones, ones, sa, s1);
            printf("(c) A-scale lane %2d byte %d = 2.0 (opsel 0): ", la, byte);
            int shown = 0;
            for (int i = 0; i < 16; ++i) if (D[i * 16] != 128.f) { printf("row %d -> %g  ", i, D[i * 16]); ++shown; }
            if (!shown) printf("no change");
            printf("\n");
        }
    // (d) same with opsel 1 on A
    for (int byte = 0; byte < 4; ++byte) {
        std::vector<int> sa = s1;
        sa[17] = (sa[17] & ~(0xff << (8 * byte))) | (128 << (8 * byte));
        auto D = run(ones, ones, sa, s1, 1, 0);
        printf("(d) A-scale lane 17 byte %d = 2.0 (opsel 1): ", byte);
        int shown = 0;
        for (int i = 0; i < 16; ++i) if (D[i * 16] != 128.f) { printf("row %d -> %g  ", i, D[i * 16]); ++shown; }
        if (!shown) printf("no change");
        printf("\n");
    }
    // (e) B side: lane lb's B-scale
    for (int lb : {0, 1, 16, 33}) {
        std::vector<int> sb = s1;
        sb[lb] = (sb[lb] & ~0xff) | 128;
        auto D = run(ones, ones, s1, sb);
        printf("(e) B-scale lane %2d byte 0 = 2.0: ", lb);
        for (int j = 0; j < 16; ++j) if (D[j] != 128.f) printf("col %d -> %g  ", j, D[j]);
        printf("\n");
    }
    // (g) which lane group's scale covers which bytes?  A row 0 = ones only in the 8 bytes [8q, 8q+8) of lane group ga, B = ones
    for (int ga = 0; ga < 4; ++ga)
        for (int q = 0; q < 4; ++q) {
            printf("(g) A bytes %2d-%2d of lane group %d are scaled by the scale of lane group:", 8 * q, 8 * q + 7, ga);
            for (int gs = 0; gs < 4; ++gs) {
                std::vector<int> ha(64 * 8, 0), sa = s1;
                ha[(16 * ga) * 8 + 2 * q] = ONE; ha[(16 * ga) * 8 + 2 * q + 1] = ONE;
                sa[16 * gs] = (sa[16 * gs] & ~0xff) | 128;
                auto D = run(ha, ones, sa, s1);
                if (D[0] == 16.f) printf("  [%d]", gs);
                else if (D[0] != 8.f) printf("  ?%g", D[0]);
            }
            printf("\n");
        }
    // (f) random data with random per-lane scales: clean upper bytes vs junk upper bytes; A only, B only, both
    {
        std::vector<unsigned char> A(16 * 128), B(128 * 16), SA(64), SB(64);
        srand(11);
        auto rnd8 = [] { unsigned char v; do v = rand() & 0xff; while ((v & 0x7f) == 0x7f || ((v >> 3) & 15) > 9); return v; };
        for (auto &v : A) v = rnd8();
        for (auto &v : B) v = rnd8();
        for (auto &v : SA) v = 125 + rand() % 5;
        for (auto &v : SB) v = 125 + rand() % 5;
        std::vector<int> ha(64 * 8), hb(64 * 8);
        for (int l = 0; l < 64; ++l) {
            unsigned char *pa = (unsigned char *)&ha[l * 8], *pb = (unsigned char *)&hb[l * 8];
            for (int j = 0; j < 32; ++j) { pa[j] = A[(l & 15) * 128 + 32 * (l >> 4) + j]; pb[j] = B[(32 * (l >> 4) + j) * 16 + (l & 15)]; }
        }
        for (int variant = 0; variant < 6; ++variant) {
            const bool useA = variant % 3 != 1, useB = variant % 3 != 0, junk = variant >= 3;
            std::vector<int> sa(64), sb(64);
            for (int l = 0; l < 64; ++l) {
                const int va = useA ? SA[(l & 15) * 4 + (l >> 4)] : 127, vb = useB ? SB[(l & 15) * 4 + (l >> 4)] : 127;
                sa[l] = va | (junk ? 0x55aa5500 : 0x7f7f7f00);
                sb[l] = vb | (junk ? 0x33cc3300 : 0x7f7f7f00);
            }
            auto D = run(ha, hb, sa, sb);
            double worst = 0, scale = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int kk = 0; kk < 128; ++kk)
                    ref += (double)e4m3(A[i * 128 + kk]) * ldexp(1.0, (useA ? SA[i * 4 + kk / 32] : 127) - 127) *
                           e4m3(B[kk * 16 + j]) * ldexp(1.0, (useB ? SB[j * 4 + kk / 32] : 127) - 127);
                worst = fmax(worst, fabs(ref - D[i * 16 + j])); scale = fmax(scale, fabs(ref));
            }
            printf("(f) random data, scales on %s%s, %s upper bytes: max|err| %.3e of %.3e\n", useA ? "A" : "", useB ? "B" : "",
                   junk ? "junk" : "0x7f", worst, scale);
        }
    }
    return 0;
}
