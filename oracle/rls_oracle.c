/*
 * rls_oracle.c -- CPU oracle of the recursive-least-squares readout update (TEST INFRASTRUCTURE ONLY).
 *
 * Restates RLS.update (reference rectipy/edges.py:227-234) as driven by Network._rls
 * (rectipy/network.py:1093-1121):
 *     z = beta^-1 P x;  k = 1/(1 + x.z);  W += outer(k (y - W x), z);  P -= k z z^T;  loss = |y - W x|^2
 * with the summation trees of rectipy_amd/csrc/rls.hip so that the HIP result can be compared bit for bit.
 * Pinned against golden vectors from the reference in tests/test_rls.py (G14).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "rectipy_hip.h"

static float butterfly64(float* s) {
    float t[64];
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; ++l) t[l] = s[l] + s[l ^ off];
        memcpy(s, t, sizeof t);
    }
    return s[0];
}

/* 1024 strided partial sums combined by a halving tree (rls_scalars_kernel) */
static float block_dot(const float* a, const float* b, int n) {
    static float red[1024];
#pragma omp threadprivate(red)
    for (int t = 0; t < 1024; ++t) red[t] = 0.0f;
    for (int j = 0; j < n; ++j) red[j & 1023] = fmaf(a[j], b[j], red[j & 1023]);
    for (int off = 512; off >= 1; off >>= 1)
        for (int t = 0; t < off; ++t) red[t] = red[t] + red[t + off];
    return red[0];
}

/* the same tree in double precision (rls_blk_scalars_kernel: x.z and the c_ij of the blocked sweep) */
static double block_dot64(const float* a, const float* b, int n) {
    static double red[1024];
#pragma omp threadprivate(red)
    for (int t = 0; t < 1024; ++t) red[t] = 0.0;
    for (int j = 0; j < n; ++j) red[j & 1023] = fma((double)a[j], (double)b[j], red[j & 1023]);
    for (int off = 512; off >= 1; off >>= 1)
        for (int t = 0; t < off; ++t) red[t] = red[t] + red[t + off];
    return red[0];
}

static float wave_dot(const float* a, const float* b, int n) { /* rls_predict_kernel */
    float s[64];
    for (int l = 0; l < 64; ++l) s[l] = 0.0f;
    for (int j = 0; j < n; ++j) s[j & 63] = fmaf(a[j], b[j], s[j & 63]);
    return butterfly64(s);
}

/* fused pass: optional rank-1 update of row i, then its dot with x_next */
static float pass_row(float* p, int n, int vec4, int update, float kz, const float* z, const float* x_next) {
    float s[64];
    if (update)
        for (int j = 0; j < n; ++j) p[j] = p[j] - kz * z[j];
    if (!x_next) return 0.0f;
    if (vec4) {
        float a[256];
        for (int q = 0; q < 256; ++q) a[q] = 0.0f;
        for (int j = 0; j < n; ++j) a[j & 255] = fmaf(p[j], x_next[j], a[j & 255]);
        for (int l = 0; l < 64; ++l) s[l] = (a[4 * l] + a[4 * l + 1]) + (a[4 * l + 2] + a[4 * l + 3]);
    } else {
        for (int l = 0; l < 64; ++l) s[l] = 0.0f;
        for (int j = 0; j < n; ++j) s[j & 63] = fmaf(p[j], x_next[j], s[j & 63]);
    }
    return butterfly64(s);
}

/* tiled pass (rp_rls.partials given, 16-byte path): row i is cut into tiles of 1024 columns; inside a tile, wave w owns
   the columns 256 w + 4 l + c: products a[l][c], lane sum, butterfly -> s_w; the tile partial is
   ((s_0 + s_1) + s_2) + s_3; the row's dot = tile partials added ascending (rls_tile_kernel) */
static float pass_row_tiled(float* p, int n, int update, float kz, const float* z, const float* x_next) {
    if (update)
        for (int j = 0; j < n; ++j) p[j] = p[j] - kz * z[j];
    if (!x_next) return 0.0f;
    float total = 0.0f;
    for (int c0 = 0, c = 0; c0 < n; c0 += 1024, ++c) {
        float sw[4];
        for (int w = 0; w < 4; ++w) {
            float a[256], s[64];
            for (int q = 0; q < 256; ++q) {
                const int j = c0 + 256 * w + q;
                a[q] = j < n ? fmaf(p[j], x_next[j], 0.0f) : 0.0f;
            }
            for (int l = 0; l < 64; ++l) s[l] = (a[4 * l] + a[4 * l + 1]) + (a[4 * l + 2] + a[4 * l + 3]);
            sw[w] = butterfly64(s);
        }
        const float part = ((sw[0] + sw[1]) + sw[2]) + sw[3];
        total = c == 0 ? part : total + part;
    }
    return total;
}

/* symmetric pass (rp_rls.colparts given; rls_sym_kernel / rls_sym_colsum_kernel / rls_sym_zn_kernel / rls_mirror_kernel
   of rls.hip): only the 1024 x 1024 blocks on and above the diagonal are visited, as tiles of R = sym_rows rows x 1024
   columns.  Per tile and row: the optional rank-1 update, the row's tile partial (the tiled pass's tree) and, above the
   diagonal, the column partial chain over the tile's rows.  colsum[I][j] = the chunks of block row I, ascending;
   zn_i = beta_inv * (colsum[0][i] + ... + colsum[I_i - 1][i], then the row partials from the diagonal block on). */
static void sym_pass(const rp_rls* r, int update, float k, const float* z, const float* x_next, float* zn) {
    const int R = r->sym_rows > 0 ? r->sym_rows : 64;
    const int n = r->n_in, nb = (n + 1023) / 1024, chunks = (n + R - 1) / R, per_block = 1024 / R;
#pragma omp parallel for schedule(dynamic)
    for (int chunk = 0; chunk < chunks; ++chunk) {
        const int row0 = chunk * R, I = row0 >> 10;
        for (int ct = I; ct < nb; ++ct) {
            const int c0 = ct * 1024;
            float cp[1024];
            for (int q = 0; q < 1024; ++q) cp[q] = 0.0f;
            for (int rr = 0; rr < R && row0 + rr < n; ++rr) {
                const int i = row0 + rr;
                float* p = r->P + (int64_t)i * r->ldp;
                if (update) {
                    const float kz = k * z[i];
                    for (int q = 0; q < 1024 && c0 + q < n; ++q) p[c0 + q] = p[c0 + q] - kz * z[c0 + q];
                }
                if (!x_next) continue;
                float sw[4];
                for (int w = 0; w < 4; ++w) {
                    float a[256], s[64];
                    for (int q = 0; q < 256; ++q) {
                        const int j = c0 + 256 * w + q;
                        a[q] = j < n ? fmaf(p[j], x_next[j], 0.0f) : 0.0f;
                    }
                    for (int l = 0; l < 64; ++l) s[l] = (a[4 * l] + a[4 * l + 1]) + (a[4 * l + 2] + a[4 * l + 3]);
                    sw[w] = butterfly64(s);
                }
                r->partials[(int64_t)i * nb + ct] = ((sw[0] + sw[1]) + sw[2]) + sw[3];
                if (ct > I)
                    for (int q = 0; q < 1024 && c0 + q < n; ++q) cp[q] = fmaf(p[c0 + q], x_next[i], cp[q]);
            }
            if (x_next && ct > I)
                for (int q = 0; q < 1024 && c0 + q < n; ++q) r->colparts[(int64_t)chunk * r->ldc + c0 + q] = cp[q];
        }
    }
    if (!x_next) return;
    /* column sums per block row (the z' kernel keeps them in LDS; here: a scratch of the oracle's own) */
    float* colsum = (float*)malloc(sizeof(float) * (size_t)nb * (size_t)n);
    for (int I = 0; I + 1 < nb; ++I) {
        const int c0 = I * per_block, c1 = c0 + per_block < chunks ? c0 + per_block : chunks;
        for (int j = (I + 1) << 10; j < n; ++j) {
            float s = r->colparts[(int64_t)c0 * r->ldc + j];
            for (int c = c0 + 1; c < c1; ++c) s = s + r->colparts[(int64_t)c * r->ldc + j];
            colsum[(int64_t)I * n + j] = s;
        }
    }
    for (int i = 0; i < n; ++i) {
        const int I = i >> 10;
        float s = 0.0f;
        int first = 1;
        for (int b = 0; b < I; ++b) {
            const float v = colsum[(int64_t)b * n + i];
            s = first ? v : s + v;
            first = 0;
        }
        for (int ct = I; ct < nb; ++ct) {
            const float v = r->partials[(int64_t)i * nb + ct];
            s = first ? v : s + v;
            first = 0;
        }
        zn[i] = r->beta_inv * s;
    }
    free(colsum);
}

static void sym_mirror(const rp_rls* r) {
    const int n = r->n_in;
    for (int i = 0; i < n; ++i)
        for (int j = ((i >> 10) + 1) << 10; j < n; ++j) r->P[(int64_t)j * r->ldp + i] = r->P[(int64_t)i * r->ldp + j];
}

/* blocked symmetric sweep (rp_rls.block > 1): P is visited once per `block` updates.  The pass applies the rank-1 updates
   of the block before (one after the other: element by element the arithmetic of the fused kernel) and forms
   Y_j = beta^-1 P x_j for the block's inputs; the block's z vectors follow from Y and the earlier z of the block. */
static int sweep_blocked(const rp_rls* r, int32_t steps, int32_t stride, const float* X, int64_t ldx, const float* Y,
                         int64_t ldy, float* pred, int64_t ldpred, float* loss) {
    const int n = r->n_in, m = r->n_out, K = r->block;
    float* Yb = r->blk;
    float* Zb = Yb + (int64_t)K * n;
    double* cm = (double*)(Zb + (int64_t)K * n);  /* c_ij (update by update) / G_lj (Gram form), double */
    double* am = cm + K * K;                      /* A_jl of the Gram form */
    float* kk = (float*)(am + K * K);
    float* hm = kk + K;                           /* H_oj = W_o . x_j */
    float* ke = hm + (int64_t)m * K;              /* k_j err_j[o] */
    float* err = r->scratch + 4;
    const int updates = (steps + stride - 1) / stride;
    const int gram = (stride == 1 || !pred) && m <= 64; /* rp_rls.block, r05: the block's scalars from its Gram matrix */
    int pending = 0;
    for (int u0 = 0; u0 < updates; u0 += K) {
        const int cnt = updates - u0 < K ? updates - u0 : K;
        for (int i = 0; i < pending; ++i) sym_pass(r, 1, kk[i], Zb + (int64_t)i * n, 0, 0);
        for (int j = 0; j < cnt; ++j) sym_pass(r, 0, 0.0f, 0, X + (int64_t)(u0 + j) * stride * ldx, Yb + (int64_t)j * n);
        if (gram) {
            double c[8];
            for (int l = 0; l < cnt; ++l)
                for (int j = 0; j < cnt; ++j)
                    cm[l * K + j] = block_dot64(Yb + (int64_t)l * n, X + (int64_t)(u0 + j) * stride * ldx, n);
            for (int o = 0; o < m; ++o)
                for (int j = 0; j < cnt; ++j)
                    hm[o * K + j] = block_dot(r->W + (int64_t)o * r->ldw, X + (int64_t)(u0 + j) * stride * ldx, n);
            for (int j = 0; j < cnt; ++j) {
                const int t = (u0 + j) * stride;
                const float* y = Y + (int64_t)t * ldy;
                double* Aj = am + j * K;
                for (int l = 0; l < cnt; ++l) Aj[l] = l == j ? 1.0 : 0.0;
                for (int i = 0; i < j; ++i) {
                    double cij = 0.0;
                    for (int l = 0; l <= i; ++l) cij = fma(am[i * K + l], cm[l * K + j], cij);
                    c[i] = cij;
                    const double g = (double)r->beta_inv * ((double)kk[i] * cij);
                    for (int l = 0; l <= i; ++l) Aj[l] = fma(-g, am[i * K + l], Aj[l]);
                }
                double d = 0.0;
                for (int l = 0; l <= j; ++l) d = fma(Aj[l], cm[l * K + j], d);
                const float k = (float)(1.0 / (1.0 + d));
                kk[j] = k;
                float l2 = 0.0f;
                for (int o = 0; o < m; ++o) {
                    double yh = (double)hm[o * K + j];
                    for (int i = 0; i < j; ++i) yh = fma((double)ke[i * m + o], c[i], yh);
                    const float yhat = (float)yh;
                    const float e = y[o] - yhat;
                    l2 = fmaf(e, e, l2);
                    ke[j * m + o] = k * e;
                    if (pred) pred[(int64_t)t * ldpred + o] = yhat;
                }
                const int seg = t + stride <= steps ? stride : steps - t;
                if (loss)
                    for (int q = 0; q < seg; ++q) loss[t + q] = l2;
            }
            for (int q = 0; q < n; ++q) {
                for (int j = 0; j < cnt; ++j) {
                    double acc = (double)Yb[(int64_t)j * n + q];
                    for (int l = 0; l < j; ++l) acc = fma(am[j * K + l], (double)Yb[(int64_t)l * n + q], acc);
                    Zb[(int64_t)j * n + q] = (float)acc;
                }
                for (int o = 0; o < m; ++o) {
                    float w = r->W[(int64_t)o * r->ldw + q];
                    for (int j = 0; j < cnt; ++j) w = w + ke[j * m + o] * Zb[(int64_t)j * n + q];
                    r->W[(int64_t)o * r->ldw + q] = w;
                }
            }
            pending = cnt;
            continue;
        }
        for (int j = 0; j < cnt; ++j) {
            const int t = (u0 + j) * stride;
            const float* x = X + (int64_t)t * ldx;
            const float* y = Y + (int64_t)t * ldy;
            float* z = Zb + (int64_t)j * n;
            /* the recurrence in double (z_j is a small difference of large vectors when consecutive inputs are alike) */
            for (int q = 0; q < n; ++q) {
                double acc = (double)Yb[(int64_t)j * n + q];
                for (int i = 0; i < j; ++i) {
                    const double g = (double)r->beta_inv * ((double)kk[i] * cm[i * K + j]);
                    acc = fma(-g, (double)Zb[(int64_t)i * n + q], acc);
                }
                z[q] = (float)acc;
            }
            const float k = (float)(1.0 / (1.0 + block_dot64(z, x, n)));
            float l = 0.0f;
            for (int o = 0; o < m; ++o) {
                const float yhat = block_dot(r->W + (int64_t)o * r->ldw, x, n);
                const float e = y[o] - yhat;
                l = fmaf(e, e, l);
                err[o] = e;
                if (pred) pred[(int64_t)t * ldpred + o] = yhat;
            }
            kk[j] = k;
            for (int jn = j + 1; jn < cnt; ++jn) cm[j * K + jn] = block_dot64(z, X + (int64_t)(u0 + jn) * stride * ldx, n);
            const int seg = t + stride <= steps ? stride : steps - t;
            if (loss)
                for (int q = 0; q < seg; ++q) loss[t + q] = l;
            for (int o = 0; o < m; ++o)
                for (int q = 0; q < n; ++q) {
                    float* w = r->W + (int64_t)o * r->ldw + q;
                    *w = *w + (k * err[o]) * z[q];
                }
            if (pred)
                for (int q = 1; q < seg; ++q)
                    for (int o = 0; o < m; ++o)
                        pred[(int64_t)(t + q) * ldpred + o] = wave_dot(r->W + (int64_t)o * r->ldw, X + (int64_t)(t + q) * ldx, n);
        }
        pending = cnt;
    }
    for (int i = 0; i < pending; ++i) sym_pass(r, 1, kk[i], Zb + (int64_t)i * n, 0, 0);
    if (!r->defer_mirror) sym_mirror(r);
    return RP_OK;
}

int oracle_rls_sweep(const rp_rls* r, int32_t steps, int32_t stride, const float* X, int64_t ldx, const float* Y,
                     int64_t ldy, float* pred, int64_t ldpred, float* loss) {
    const int n = r->n_in, m = r->n_out;
    if (steps == 0) return RP_OK;
    const int vec4 = (n % 4 == 0) && (r->ldp % 4 == 0) && (ldx % 4 == 0);
    const int tiled = vec4 && r->partials != 0;
    const int sym = tiled && r->colparts != 0;
    if (sym && r->block > 1) return sweep_blocked(r, steps, stride, X, ldx, Y, ldy, pred, ldpred, loss);
    float* za = r->z;
    float* zb = r->zn;
    float* err = r->scratch + 4;
    if (sym)
        sym_pass(r, 0, 0.0f, 0, X, za);
    else
        for (int i = 0; i < n; ++i)
            za[i] = r->beta_inv * (tiled ? pass_row_tiled(r->P + (int64_t)i * r->ldp, n, 0, 0.0f, 0, X)
                                         : pass_row(r->P + (int64_t)i * r->ldp, n, vec4, 0, 0.0f, 0, X));
    for (int t = 0; t < steps; t += stride) {
        const float* x = X + (int64_t)t * ldx;
        const float* y = Y + (int64_t)t * ldy;
        const float k = 1.0f / (1.0f + block_dot(x, za, n));
        float l = 0.0f;
        for (int o = 0; o < m; ++o) {
            const float yhat = block_dot(r->W + (int64_t)o * r->ldw, x, n);
            const float e = y[o] - yhat;
            l = fmaf(e, e, l);
            err[o] = e;
            if (pred) pred[(int64_t)t * ldpred + o] = yhat;
        }
        const int seg = t + stride <= steps ? stride : steps - t;
        if (loss)
            for (int q = 0; q < seg; ++q) loss[t + q] = l;
        for (int o = 0; o < m; ++o)
            for (int j = 0; j < n; ++j) {
                float* w = r->W + (int64_t)o * r->ldw + j;
                *w = *w + (k * err[o]) * za[j];
            }
        const int next = t + stride;
        const float* x_next = next < steps ? X + (int64_t)next * ldx : 0;
        if (sym) sym_pass(r, 1, k, za, x_next, zb);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < (sym ? 0 : n); ++i) {
            const float v = tiled ? pass_row_tiled(r->P + (int64_t)i * r->ldp, n, 1, k * za[i], za, x_next)
                                  : pass_row(r->P + (int64_t)i * r->ldp, n, vec4, 1, k * za[i], za, x_next);
            if (x_next) zb[i] = r->beta_inv * v;
        }
        if (pred)
            for (int q = 1; q < seg; ++q)
                for (int o = 0; o < m; ++o)
                    pred[(int64_t)(t + q) * ldpred + o] = wave_dot(r->W + (int64_t)o * r->ldw, X + (int64_t)(t + q) * ldx, n);
        float* tmp = za;
        za = zb;
        zb = tmp;
    }
    if (sym && !r->defer_mirror) sym_mirror(r);
    return RP_OK;
}

/* rp_rls_mirror: the lower blocks from the authoritative upper ones (what a sweep with defer_mirror = 1 left undone) */
int oracle_rls_mirror(const rp_rls* r) {
    sym_mirror(r);
    return RP_OK;
}
