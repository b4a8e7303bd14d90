// Generic backward of unit_agcn (model/unit_agcn.py:73-93) in TRAINING mode: any C_in / C_out / number of subsets,
// conv + BatchNorm residual ("down") or the identity residual (C_in == C_out, :57-58), and the gradient of the INPUT —
// what the deeper TCN_GCN_unit layers need (model/ST_TR/ST_TR_new.py:355-372: gcn1 = unit_agcn(in, out) sits behind other
// layers, so dx must flow).  The stem's own shape class (C_in = 3, x is data) keeps its single fused kernel
// (agcn_backward.hip); this file is a chain of strided batched fp32 GEMMs (gemm_f32.hip), one (clip, subset) pair or one
// clip with the subsets stacked along M / K per batch entry — every line below is ONE launch for all subsets:
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
// (Round 3: the chain ran subset by subset — 40 products + 25 part sums per call, x re-read for each embedding and dx
//  re-read and re-written per term; now 15 products: the embeddings stack the subsets along M, dx += sum_s du_s P_s^T
//  contracts over (subset, joint), the rest runs over (clip, subset) batches, and bd's gradient is computed once.)
#include "common.h"

namespace stgcn {

namespace {

// Products with few C tiles per batch entry and a long contraction (dW slices: K = T*V; joint Grams: K = C_in*T) are split
// along K so that about a thousand workgroups run: parts [ksplit][batch][m], summed in a fixed order.  The part buffer caps
// the split.
constexpr size_t PART_CAP_FLOATS = (size_t)16 << 20;     // 64 MiB

size_t part_m(int Cin, int Cout, int inter_c, int V, int S) {     // floats of C per clip of the largest split product
    size_t m = (size_t)S * Cout * Cin;
    if ((size_t)S * V * V > m) m = (size_t)S * V * V;
    (void)inter_c;                                                // S*inter_c*Cin <= S*Cout*Cin
    return m;
}

size_t part_floats(int N, int Cin, int Cout, int inter_c, int V, int S) {
    const size_t one = (size_t)N * part_m(Cin, Cout, inter_c, V, S);
    return one * 16 <= PART_CAP_FLOATS ? one * 16 : (one > PART_CAP_FLOATS ? one : PART_CAP_FLOATS);
}

// split factor of a product with `tiles` C tiles over all its batch entries, `batch` entries, contraction length K and m
// floats of C per entry
int k_split(int batch, int tiles, int K, size_t m, size_t part_cap) {
    long long ks = 1024 / ((long long)batch * tiles);
    if (ks > 16) ks = 16;
    if (ks > K / 128) ks = K / 128;                      // at least 128 contraction steps per part
    const long long fit = (long long)(part_cap / ((size_t)batch * m));
    if (ks > fit) ks = fit;
    if ((long long)batch * ks > 65535) ks = 65535 / batch;
    return ks < 1 ? 1 : (int)ks;
}

}  // namespace

size_t agcn_bwd_generic_ws_floats(int N, int Cin, int Cout, int T, int V, int inter_c, int S) {
    const size_t P = (size_t)T * V;
    const size_t big = (size_t)N * S * Cin * P, emb = (size_t)N * S * inter_c * P, vv = (size_t)N * S * V * V;
    const size_t rows = (size_t)N * (Cout > S * inter_c ? Cout : S * inter_c);
    return 2 * big + 4 * emb + 2 * vv + part_floats(N, Cin, Cout, inter_c, V, S) + rows + 64;
}

int launch_agcn_bwd_generic(const float *x, const float *Pm, const float *A_eff, const float *dzm, const float *dzd,
                            const float *Wa, const float *ba, const float *Wb, const float *bb, const float *Wd,
                            const float *Wdown, float *ws, float *dWa, float *dba, float *dWb, float *dbb, float *dWd,
                            float *dbd, float *dWdown, float *dbdown, float *dPA, float *dx, int dx_initialised, int N,
                            int Cin, int Cout, int T, int V, int inter_c, int S, hipStream_t st) {
    const long long P = (long long)T * V, VV = (long long)V * V;
    const int R = Cin * T, RI = inter_c * T;           // rows of the per-frame joint products
    const int SI = S * inter_c;                         // the embeddings of all subsets, stacked
    const long long xs = (long long)Cin * P, zs = (long long)Cout * P, es = (long long)inter_c * P;   // per-clip / per-subset strides
    const size_t big = (size_t)N * S * Cin * P, emb = (size_t)N * S * inter_c * P, vv = (size_t)N * S * VV;
    const size_t part_cap = part_floats(N, Cin, Cout, inter_c, V, S);
    float *U = ws, *DU = U + big, *EA = DU + big, *EB = EA + emb, *DA = EB + emb, *DB = DA + emb, *DP = DB + emb, *DS = DP + vv,
          *part = DS + vv, *bpart = part + part_cap;
    int rc;
#define OK(expr) do { rc = (expr); if (rc != STGCN_OK) return rc; } while (0)
    // sum over clips (and K parts) of dY[b] (rows x P) X[b]^T (cols x P) -> out; batch entries = N*inner, entry b = (n, i):
    // dY + n*dy_sb, X + n*x_sb + i*x_sb2, the slice of entry (n, i) at part[..][n][i][rows*cols];  bias gradient: the row sums
    // of dY summed over clips -> bout[rows], written `breps` times `rows` apart
    auto wgrad = [&](const float *dY, long long dy_sb, int rows, const float *X, long long x_sb, long long x_sb2, int cols,
                     int inner, float *out, float *bout, int breps) {
        const size_t m = (size_t)rows * cols;
        GemmArgs g{dY, X, part, nullptr, rows, cols, (int)P, P, 1, dy_sb, 1, P, x_sb, cols, 1, (long long)(inner * m), 1.f, 0};
        g.b_inner = inner; g.a_sb2 = 0; g.b_sb2 = x_sb2; g.c_sb2 = (long long)m;
        g.ksplit = k_split(N * inner, ceil_div(rows, 64) * ceil_div(cols, 64), (int)P, m, part_cap);
        g.c_ss = (long long)N * inner * (long long)m;     // parts [ksplit][N][inner][rows*cols]
        int r = launch_gemm_f32(g, N * inner, st);
        if (r != STGCN_OK) return r;
        // out[i][rows*cols] = sum over (part, clip): the (j, n) slices are inner*m apart
        r = launch_sum_parts(part, out, N * g.ksplit, (size_t)inner * m, st);
        if (r != STGCN_OK || bout == nullptr) return r;
        r = launch_row_sum(dY, bpart, N * rows, (int)P, st);
        if (r != STGCN_OK) return r;
        return launch_sum_parts(bpart, bout, N, (size_t)rows, st, breps, (size_t)rows);
    };
    // ---- residual branch ---------------------------------------------------------------------------------------
    if (Wdown != nullptr) {
        OK(wgrad(dzd, zs, Cout, x, xs, 0, Cin, 1, dWdown, dbdown, 1));
        if (dx != nullptr) {      // dx = Wdown^T dzd : A[m = c][k = o] = Wdown[o][c]
            GemmArgs g{Wdown, dzd, dx, nullptr, Cin, (int)P, Cout, 1, Cin, 0, P, 1, zs, P, 1, xs, 1.f, dx_initialised};
            OK(launch_gemm_f32(g, N, st));
            dx_initialised = 1;
        }
    }
    if (dx != nullptr && !dx_initialised) return fail(STGCN_ERR_ARG, "agcn backward: dx has no initial term");
    // ---- u_s = x P_s for every (clip, subset): rows (channel, frame), V x V matrix --------------------------------
    OK(launch_rowmix(x, Pm, U, R, V, 1, 0, xs, 0, 0, (long long)S * VV, VV, 0, false, (long long)S * xs, xs, N * S, S, st));
    // dWd_s = sum_n dzm u_s^T ;  dbd_s = sum dzm (the same vector for every subset)
    OK(wgrad(dzm, zs, Cout, U, (long long)S * xs, xs, Cin, S, dWd, dbd, S));
    // ---- du_s = Wd_s^T dzm : A[m = c][k = o] = Wd[s][o][c] -------------------------------------------------------
    {
        GemmArgs g{Wd, dzm, DU, nullptr, Cin, (int)P, Cout, 1, Cin, 0, P, 1, zs, P, 1, (long long)S * xs, 1.f, 0};
        g.b_inner = S; g.a_sb2 = (long long)Cout * Cin; g.b_sb2 = 0; g.c_sb2 = xs;
        OK(launch_gemm_f32(g, N * S, st));
    }
    if (dx != nullptr) {          // dx += sum_s du_s P_s^T : k = (s, w);  A[m = r][k] = du_s[r][w],  B[k][n = v] = P_s[v][w]
        OK(launch_rowmix(DU, Pm, dx, R, V, S, 1, (long long)S * xs, 0, xs, (long long)S * VV, 0, VV, true, xs, 0, N, 0, st));
    }
    // ---- dP_s = x^T du_s (V x V per clip and subset), dPA_s = sum over clips ------------------------------------------
    {
        GemmArgs g{x, DU, DP, nullptr, V, V, R, 1, V, xs, V, 1, (long long)S * xs, V, 1, (long long)S * VV, 1.f, 0};
        g.b_inner = S; g.a_sb2 = 0; g.b_sb2 = xs; g.c_sb2 = VV;
        g.ksplit = k_split(N * S, ceil_div(V, 64) * ceil_div(V, 64), R, (size_t)VV, part_cap);
        if (g.ksplit > 1) {                           // parts [ksplit][N][S][V*V] in the part buffer, summed into DP
            g.C = part;
            g.c_ss = (long long)N * S * VV;
        }
        OK(launch_gemm_f32(g, N * S, st));
        if (g.ksplit > 1) OK(launch_sum_parts(part, DP, g.ksplit, (size_t)N * S * VV, st));
    }
    OK(launch_sum_parts(DP, dPA, N, (size_t)S * VV, st));
    OK(launch_softmax_bwd(Pm, A_eff, DP, DS, N, V, S, 1.f / (float)(inter_c * T), st));
    // ---- embeddings a, b of all subsets (recomputed): rows (s, c') stacked, x read once per product -----------------
    {
        GemmArgs g{Wa, x, EA, ba, SI, (int)P, Cin, Cin, 1, 0, P, 1, xs, P, 1, (long long)S * es, 1.f, 0};
        OK(launch_gemm_f32(g, N, st));
        g.A = Wb; g.C = EB; g.bias = bb;
        OK(launch_gemm_f32(g, N, st));
    }
    // da_s[r][v] = sum_w b_s[r][w] dS_s[v][w] ;  db_s[r][w] = sum_v a_s[r][v] dS_s[v][w]     (rows r = (c', t))
    OK(launch_rowmix(EB, DS, DA, RI, V, 1, 0, (long long)S * es, es, 0, (long long)S * VV, VV, 0, true, (long long)S * es, es, N * S, S, st));
    OK(launch_rowmix(EA, DS, DB, RI, V, 1, 0, (long long)S * es, es, 0, (long long)S * VV, VV, 0, false, (long long)S * es, es, N * S, S, st));
    // dWa = sum_n da x^T (rows (s, c') stacked: the layout of Wa itself) ;  dba = sum da ;  the same for b
    OK(wgrad(DA, (long long)S * es, SI, x, xs, 0, Cin, 1, dWa, dba, 1));
    OK(wgrad(DB, (long long)S * es, SI, x, xs, 0, Cin, 1, dWb, dbb, 1));
    if (dx != nullptr) {          // dx += Wa^T da + Wb^T db, contracted over (s, c')
        GemmArgs g{Wa, DA, dx, nullptr, Cin, (int)P, SI, 1, Cin, 0, P, 1, (long long)S * es, P, 1, xs, 1.f, 1};
        OK(launch_gemm_f32(g, N, st));
        g.A = Wb; g.B = DB;
        OK(launch_gemm_f32(g, N, st));
    }
#undef OK
    return STGCN_OK;
}

}  // namespace stgcn
