// Generic backward of unit_agcn (model/unit_agcn.py:73-93) in TRAINING mode: any C_in / C_out / number of subsets,
// conv + BatchNorm residual ("down") or the identity residual (C_in == C_out, :57-58), and the gradient of the INPUT —
// what the deeper TCN_GCN_unit layers need (model/ST_TR/ST_TR_new.py:355-372: gcn1 = unit_agcn(in, out) sits behind other
// layers, so dx must flow).  The stem's own shape class (C_in = 3, x is data) keeps its single fused kernel
// (agcn_backward.hip); this file is a chain of strided batched fp32 GEMMs (gemm_f32.hip), one clip per batch entry:
//
//   given dzm = dL/d(sum_s conv_d_s(x P_s))  and  dzd = dL/d(conv_down(x))   (the BatchNorm+ReLU backward, done by the caller)
//   residual        dWdown = sum_n dzd x^T ;  dbdown = sum dzd ;  dx  = Wdown^T dzd          (identity: dx = g, by the caller)
//   per subset s    u     = x P_s                                  (per frame: rows (channel, frame), V x V matrix)
//                   dWd_s = sum_n dzm u^T ;  dbd_s = sum dzm       (every bd_s adds straight into zm)
//                   du    = Wd_s^T dzm ;   dx += du P_s^T ;   dP = x^T du ;   dPA_s = sum_n dP
//                   dS    = Q * (dP - colsum(Q * dP)) / (inter_c*T),  Q = P_s - A_eff_s      (soft-max over dim -2)
//                   a = Wa_s x + ba_s ;  b = Wb_s x + bb_s         (the embeddings, recomputed)
//                   da[c,t,v] = sum_w dS[v,w] b[c,t,w] ;  db[c,t,w] = sum_v dS[v,w] a[c,t,v]
//                   dWa_s = sum_n da x^T ;  dba_s = sum da ;  (same for b) ;  dx += Wa_s^T da + Wb_s^T db
// Per-clip slices of the weight gradients are summed over clips in a fixed order (deterministic).
#include "common.h"

namespace stgcn {

namespace {

struct Ws {
    float *U, *DU, *A, *B, *DA, *DB, *DP, *DS, *part, *bpart;
};

// Products with one C tile per clip and a long contraction (dW slices: K = T*V; joint Grams: K = C_in*T) are split along K
// so that about a thousand workgroups run: parts [ksplit][N][m], summed in a fixed order.  The part buffer caps the split.
constexpr size_t PART_CAP_FLOATS = (size_t)16 << 20;     // 64 MiB

size_t part_floats(int N, int Cin, int Cout, int inter_c, int V) {
    size_t m = (size_t)Cout * Cin;
    if ((size_t)inter_c * Cin > m) m = (size_t)inter_c * Cin;
    if ((size_t)V * V > m) m = (size_t)V * V;
    const size_t one = (size_t)N * m;
    return one * 16 <= PART_CAP_FLOATS ? one * 16 : (one > PART_CAP_FLOATS ? one : PART_CAP_FLOATS);
}

// split factor of a product with `tiles` C tiles per clip, contraction length K and m floats of C per clip
int k_split(int N, int tiles, int K, size_t m, size_t part_cap) {
    long long ks = 1024 / ((long long)N * tiles);
    if (ks > 16) ks = 16;
    if (ks > K / 128) ks = K / 128;                      // at least 128 contraction steps per part
    const long long fit = (long long)(part_cap / ((size_t)N * m));
    if (ks > fit) ks = fit;
    if ((long long)N * ks > 65535) ks = 65535 / N;
    return ks < 1 ? 1 : (int)ks;
}

}  // namespace

size_t agcn_bwd_generic_ws_floats(int N, int Cin, int Cout, int T, int V, int inter_c, int S) {
    (void)S;
    const size_t P = (size_t)T * V;
    const size_t big = (size_t)N * Cin * P, emb = (size_t)N * inter_c * P, vv = (size_t)N * V * V;
    const size_t rows = (size_t)N * (Cout > inter_c ? Cout : inter_c);
    return 2 * big + 4 * emb + 2 * vv + part_floats(N, Cin, Cout, inter_c, V) + rows + 64;
}

int launch_agcn_bwd_generic(const float *x, const float *Pm, const float *A_eff, const float *dzm, const float *dzd,
                            const float *Wa, const float *ba, const float *Wb, const float *bb, const float *Wd,
                            const float *Wdown, float *ws, float *dWa, float *dba, float *dWb, float *dbb, float *dWd,
                            float *dbd, float *dWdown, float *dbdown, float *dPA, float *dx, int dx_initialised, int N,
                            int Cin, int Cout, int T, int V, int inter_c, int S, hipStream_t st) {
    const long long P = (long long)T * V, VV = (long long)V * V;
    const int R = Cin * T, RI = inter_c * T;           // rows of the per-frame joint products
    const size_t big = (size_t)N * Cin * P, emb = (size_t)N * inter_c * P, vv = (size_t)N * VV;
    Ws w;
    w.U = ws; w.DU = w.U + big; w.A = w.DU + big; w.B = w.A + emb; w.DA = w.B + emb; w.DB = w.DA + emb;
    w.DP = w.DB + emb; w.DS = w.DP + vv; w.part = w.DS + vv; w.bpart = w.part + part_floats(N, Cin, Cout, inter_c, V);
    int rc;
    auto gemm = [&](const float *A, long long asm_, long long ask, long long asb, const float *B, long long bsk, long long bsn,
                    long long bsb, float *C, long long csm, long long csn, long long csb, int M, int Nn, int K,
                    const float *bias, float alpha, int acc) {
        GemmArgs g{A, B, C, bias, M, Nn, K, asm_, ask, asb, bsk, bsn, bsb, csm, csn, csb, alpha, acc};
        return launch_gemm_f32(g, N, st);
    };
    // sum over clips of dY[n] (rows x P) X[n]^T (cols x P)  ->  out (rows x cols);  bias gradient -> bout (rows)
    const size_t part_cap = part_floats(N, Cin, Cout, inter_c, V);
    auto wgrad = [&](const float *dY, int rows, const float *X, int cols, float *out, float *bout) {
        const size_t m = (size_t)rows * cols;
        GemmArgs g{dY, X, w.part, nullptr, rows, cols, (int)P, P, 1, (long long)rows * P, 1, P, (long long)cols * P, cols, 1,
                   (long long)m, 1.f, 0};
        g.ksplit = k_split(N, ceil_div(rows, 64) * ceil_div(cols, 64), (int)P, m, part_cap);
        g.c_ss = (long long)N * (long long)m;             // parts [ksplit][N][rows*cols]
        int r = launch_gemm_f32(g, N, st);
        if (r != STGCN_OK) return r;
        r = launch_sum_parts(w.part, out, N * g.ksplit, m, st);
        if (r != STGCN_OK || bout == nullptr) return r;
        r = launch_row_sum(dY, w.bpart, N * rows, (int)P, st);
        if (r != STGCN_OK) return r;
        return launch_sum_parts(w.bpart, bout, N, (size_t)rows, st);
    };
#define OK(expr) do { rc = (expr); if (rc != STGCN_OK) return rc; } while (0)
    // ---- residual branch ---------------------------------------------------------------------------------------
    if (Wdown != nullptr) {
        OK(wgrad(dzd, Cout, x, Cin, dWdown, dbdown));
        if (dx != nullptr) {      // dx = Wdown^T dzd : A[m = c][k = o] = Wdown[o][c]
            OK(gemm(Wdown, 1, Cin, 0, dzd, P, 1, (long long)Cout * P, dx, P, 1, (long long)Cin * P, Cin, (int)P, Cout, nullptr,
                    1.f, dx_initialised));
            dx_initialised = 1;
        }
    }
    if (dx != nullptr && !dx_initialised) return fail(STGCN_ERR_ARG, "agcn backward: dx has no initial term");
    for (int s = 0; s < S; ++s) {
        const float *Ps = Pm + (size_t)s * VV;                    // clip stride S*V*V
        const float *Wd_s = Wd + (size_t)s * Cout * Cin;
        const float *Wa_s = Wa + (size_t)s * inter_c * Cin, *Wb_s = Wb + (size_t)s * inter_c * Cin;
        // u = x P_s
        OK(gemm(x, V, 1, (long long)Cin * P, Ps, V, 1, (long long)S * VV, w.U, V, 1, (long long)Cin * P, R, V, V, nullptr, 1.f, 0));
        OK(wgrad(dzm, Cout, w.U, Cin, dWd + (size_t)s * Cout * Cin, dbd + (size_t)s * Cout));
        // du = Wd_s^T dzm
        OK(gemm(Wd_s, 1, Cin, 0, dzm, P, 1, (long long)Cout * P, w.DU, P, 1, (long long)Cin * P, Cin, (int)P, Cout, nullptr, 1.f, 0));
        if (dx != nullptr)        // dx += du P_s^T : B[k = w][n = v] = P_s[v][w]
            OK(gemm(w.DU, V, 1, (long long)Cin * P, Ps, 1, V, (long long)S * VV, dx, V, 1, (long long)Cin * P, R, V, V, nullptr, 1.f, 1));
        // dP = x^T du  (V x V per clip), dPA_s = sum over clips
        {
            GemmArgs g{x, w.DU, w.DP, nullptr, V, V, R, 1, V, (long long)Cin * P, V, 1, (long long)Cin * P, V, 1, VV, 1.f, 0};
            g.ksplit = k_split(N, ceil_div(V, 64) * ceil_div(V, 64), R, (size_t)VV, part_cap);
            if (g.ksplit > 1) {                           // parts [ksplit][N][V*V] in the part buffer, summed per clip into DP
                g.C = w.part;
                g.c_ss = (long long)N * VV;
            }
            OK(launch_gemm_f32(g, N, st));
            if (g.ksplit > 1) OK(launch_sum_parts(w.part, w.DP, g.ksplit, (size_t)N * VV, st));
        }
        OK(launch_sum_parts(w.DP, dPA + (size_t)s * VV, N, (size_t)VV, st));
        OK(launch_softmax_bwd(Pm, A_eff, w.DP, w.DS, N, V, S, s, 1.f / (float)(inter_c * T), st));
        // embeddings a, b (recomputed)
        OK(gemm(Wa_s, Cin, 1, 0, x, P, 1, (long long)Cin * P, w.A, P, 1, (long long)inter_c * P, inter_c, (int)P, Cin,
                ba + (size_t)s * inter_c, 1.f, 0));
        OK(gemm(Wb_s, Cin, 1, 0, x, P, 1, (long long)Cin * P, w.B, P, 1, (long long)inter_c * P, inter_c, (int)P, Cin,
                bb + (size_t)s * inter_c, 1.f, 0));
        // da[r][v] = sum_w b[r][w] dS[v][w] ;  db[r][w] = sum_v a[r][v] dS[v][w]
        OK(gemm(w.B, V, 1, (long long)inter_c * P, w.DS, 1, V, VV, w.DA, V, 1, (long long)inter_c * P, RI, V, V, nullptr, 1.f, 0));
        OK(gemm(w.A, V, 1, (long long)inter_c * P, w.DS, V, 1, VV, w.DB, V, 1, (long long)inter_c * P, RI, V, V, nullptr, 1.f, 0));
        OK(wgrad(w.DA, inter_c, x, Cin, dWa + (size_t)s * inter_c * Cin, dba + (size_t)s * inter_c));
        OK(wgrad(w.DB, inter_c, x, Cin, dWb + (size_t)s * inter_c * Cin, dbb + (size_t)s * inter_c));
        if (dx != nullptr) {      // dx += Wa_s^T da + Wb_s^T db
            OK(gemm(Wa_s, 1, Cin, 0, w.DA, P, 1, (long long)inter_c * P, dx, P, 1, (long long)Cin * P, Cin, (int)P, inter_c, nullptr,
                    1.f, 1));
            OK(gemm(Wb_s, 1, Cin, 0, w.DB, P, 1, (long long)inter_c * P, dx, P, 1, (long long)Cin * P, Cin, (int)P, inter_c, nullptr,
                    1.f, 1));
        }
    }
#undef OK
    return STGCN_OK;
}

}  // namespace stgcn
