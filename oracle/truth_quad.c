/* truth_quad.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * High-precision "truth" eigenvalues of the banded generalized pencil  H_l c = E S c  that the reference hands to
 * DSYGV (matrices.f90:244-248), computed WITHOUT any of the algorithms under test: bisection on the inertia of
 * H_l - x S, counted as the negative pivots of its banded LDL^T factorisation (Sylvester's law), all in
 * __float128 (113-bit significand, ~34 digits).  The inputs are the bit-exact double-precision bands the oracle
 * assembles (pinned bit-for-bit against the compiled reference, tests/test_oracle_golden.py), so the result is the
 * exact spectrum of the matrices the reference's LAPACK and the GPU path both start from; it measures the error of
 * EACH of them instead of their difference.
 *
 * Band layout as everywhere in this repo: B[d*n + i] = M(i, i+d), d = 0..k-1 (upper band, half width p = k-1).
 *
 * build: gcc -O2 -fopenmp -fPIC -shared -o libtruth.so truth_quad.c   (quad arithmetic comes from libgcc)
 */
#include <stdlib.h>
#include <string.h>

typedef __float128 q_t;

/* number of eigenvalues of (H, S) below x: negative pivots of LDL^T(H - x S), S positive definite */
static int inertia_count(int n, int k, const double *SB, const double *HB, q_t x)
{
    const int p = k - 1, w = p + 1;
    q_t *M = (q_t *)malloc(sizeof(q_t) * w * w);    /* Schur-complement window, rows/cols j .. j+p */
    int cnt = 0;
    /* window at j = 0 */
    for (int r = 0; r < w; ++r)
        for (int c = r; c < w; ++c) {
            const int d = c - r;
            q_t v = 0;
            if (c < n) v = (q_t)HB[d * n + r] - x * (q_t)SB[d * n + r];
            M[r * w + c] = v; M[c * w + r] = v;
        }
    const q_t tiny = 1e-300Q;
    for (int j = 0; j < n; ++j) {
        q_t d = M[0];
        if (d == 0) d = -tiny;
        if (d < 0) ++cnt;
        const q_t inv = 1 / d;
        q_t col0[64];
        for (int r = 1; r < w; ++r) col0[r] = M[r];     /* row 0 = column 0 (symmetric), saved before the shift */
        /* eliminate row/col 0 and shift the window up-left by one (reads touch the upper triangle only) */
        for (int r = 1; r < w; ++r) {
            const q_t lr = col0[r] * inv;
            for (int c = r; c < w; ++c) {
                const q_t v = M[r * w + c] - lr * col0[c];
                M[(r - 1) * w + (c - 1)] = v;
                M[(c - 1) * w + (r - 1)] = v;
            }
        }
        /* new last row/col: global index g = j + 1 + p, entries A(j+1+r, g), r = 0..p */
        const int g = j + 1 + p;
        for (int r = 0; r < w; ++r) {
            const int i = j + 1 + r, dd = g - i;
            q_t v = 0;
            if (g < n && dd < k) v = (q_t)HB[dd * n + i] - x * (q_t)SB[dd * n + i];
            M[r * w + p] = v; M[p * w + r] = v;
        }
    }
    free(M);
    return cnt;
}

int orc_truth_count(int n, int k, const double *SB, const double *HB, double x_hi, double x_lo)
{
    return inertia_count(n, k, SB, HB, (q_t)x_hi + (q_t)x_lo);
}

/* Eigenvalue number idx[t] (0-based, ascending) for every t, bracketed around est[t] +- width[t] (the bracket is
 * widened until the counts confirm it) and bisected until it is narrower than rtol * |x| + atol.  Result as an
 * unevaluated sum of two doubles out_hi + out_lo.  Returns 0, or 1 + t if the bracket of target t never closed. */
int orc_truth_eigs(int n, int k, const double *SB, const double *HB, int nt, const int *idx, const double *est,
                   const double *width, double rtol, double atol, double *out_hi, double *out_lo)
{
    int status = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < nt; ++t) {
        const int m = idx[t];
        q_t wd = (q_t)width[t];
        q_t lo = (q_t)est[t] - wd, hi = (q_t)est[t] + wd;
        int ok = 0;
        for (int tries = 0; tries < 60; ++tries) {
            const int cl = inertia_count(n, k, SB, HB, lo), ch = inertia_count(n, k, SB, HB, hi);
            if (cl <= m && ch >= m + 1) { ok = 1; break; }
            wd *= 4;
            if (cl > m) lo -= wd;
            if (ch < m + 1) hi += wd;
        }
        if (!ok) {
#pragma omp critical
            status = 1 + t;
            continue;
        }
        for (int it = 0; it < 400; ++it) {
            const q_t mid = (lo + hi) / 2;
            const q_t am = mid < 0 ? -mid : mid;
            if (hi - lo <= (q_t)rtol * am + (q_t)atol) break;
            if (inertia_count(n, k, SB, HB, mid) > m) hi = mid; else lo = mid;
        }
        const q_t x = (lo + hi) / 2;
        const double h = (double)x;
        out_hi[t] = h;
        out_lo[t] = (double)(x - (q_t)h);
    }
    return status;
}
