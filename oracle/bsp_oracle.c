/*
 * bsp_oracle.c -- CPU restatement of the BspAtom hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the MI355X build.  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  All indices that the reference keeps 1-based are kept 1-based
 * inside the routines (arrays are passed 0-based and shifted locally), so the loops
 * read like the Fortran they restate.
 *
 * Pinned against: tests/golden/ fixtures generated from the compiled reference
 * (oracle/ref/build_ref.sh -> oracle/_ref/) -- S, T, V and U_l bit-for-bit.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no FMA contraction: the reference
 * oracle binary was built for baseline x86-64, which has no FMA instruction).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_OK 0
#define ORC_ERR_BSPLVB 2  /* 'FATAL ERROR - BSPLVB' STOP, bsplvb.f90:30-34 */

typedef struct {
    /* VARS_BSP (ReadInputs.f90:15, defaults :27-36) */
    int kind_grid, k, ka, nfun, kind_bc1, kind_bc2;
    double ra, rb, rmax;
    /* VARS_TISE (ReadInputs.f90:16-17, defaults :75-84) */
    int n0_ini, l_ini, m_ini, l_fin, lmax, kind_pot;
    double emax_fin, zatom;
    /* derived (ReadInputs.f90:39-69, :87) */
    int nbc1, nbc2, nkp, nointv, nintv_exp, nintv_lin, imax;
    double gsize;
    /* Rogers potential (ReadInputs.f90:95-128) */
    int numn[3], ntot;
    double alphan[3];
    /* Simons-Fues (ReadInputs.f90:130-141); Bl(l)=0 for l>3 */
    double bl[4];
} orc_cfg;

/* ReadInputs.f90:39-69,87,95-141 -- derived sizes and potential parameters. */
int orc_derive(orc_cfg *c)
{
    if (c->ka == 0) c->ka = c->k + 3;                       /* :39 */
    c->nbc1 = c->k; c->nbc2 = c->k;                         /* :42-43 */
    if (c->kind_bc1 == 0) c->nbc1 = c->k - 1;               /* :44 */
    if (c->kind_bc2 == 0) c->nbc2 = c->k - 1;               /* :45 */
    c->nkp = c->nfun + c->k;                                /* :47 */
    c->nointv = c->nkp - c->nbc1 - c->nbc2 + 1;             /* :48 */
    c->gsize = c->rb - c->ra;                               /* :50 */
    c->nintv_exp = 0; c->nintv_lin = 0; c->imax = 0;
    if (c->kind_grid == 2) {                                /* :52-69 */
        double dx = c->gsize / c->nointv;
        double rimax = (c->rmax - c->ra) / dx;
        c->imax = (int)lround(rimax);                       /* NINT: half away from zero */
        c->nintv_exp = 3 * c->imax;
        c->nintv_lin = c->nointv - c->imax;
        c->nointv = c->nintv_exp + c->nintv_lin;
        c->nkp = c->nointv + c->nbc1 + c->nbc2 - 1;
        c->nfun = c->nkp - c->k;
    }
    if (c->l_fin > c->lmax) c->lmax = c->l_fin;             /* :87 */
    c->ntot = 0;
    c->alphan[0] = c->alphan[1] = c->alphan[2] = 0.0;
    c->bl[0] = c->bl[1] = c->bl[2] = c->bl[3] = 0.0;
    if (c->kind_pot == 1) {                                 /* :95-128 */
        static const double aj[3][4] = {
            {0.8855, 0.2549, -0.0901, 0.0},
            {0.3386, 1.1323, -0.4904, 0.0},
            {0.1437, 0.9129, -0.6940, 0.2503}};
        c->numn[0] = 2; c->numn[1] = 8; c->numn[2] = 8;
        for (int i = 0; i < 3; ++i) {
            c->ntot += c->numn[i];                          /* running Ntot, :116 */
            double xn = (double)(c->zatom - c->ntot);
            if (xn == 0.0) xn = 1.0;
            double suman = 0.0;
            for (int j = 0; j <= 3; ++j) {
                /* xn**j with integer j: repeated multiplication */
                double p = 1.0;
                for (int q = 0; q < j; ++q) p *= xn;
                suman = suman + aj[i][j] / p;
            }
            c->alphan[i] = (xn + 1.0) * suman;
        }
    } else if (c->kind_pot == 2) {                          /* :130-141 */
        c->bl[0] = 0.72657; c->bl[1] = 0.47095; c->bl[2] = -0.55508; c->bl[3] = -0.04008;
    }
    return ORC_OK;
}

/* Modules.f90:112-153 -- Numerical-Recipes gauleg, tolerance 10*EPSILON(1d0). */
void orc_gauleg(double x1, double x2, double *x, double *w, int n)
{
    const double PI = acos(-1.0);                           /* Modules.f90:9 */
    const double EPS1 = DBL_EPSILON * 10;                   /* :129 */
    int m = (n + 1) / 2;
    double xm = 0.5 * (x2 + x1);
    double xl = 0.5 * (x2 - x1);
    /* pp lives across iterations of i, as the Fortran local does: for odd n the middle node
     * starts at z = cos(pi/2) ~ 6e-17, |z - z1| <= EPS1 with z1 = 0, the Newton loop never
     * runs and the weight is formed from the PREVIOUS node's pp (reference quirk, kept). */
    double pp = 0.0;
    for (int i = 1; i <= m; ++i) {
        double z = cos(PI * (i - .25) / (n + .5));
        double z1 = 0.0;
        while (fabs(z - z1) > EPS1) {
            double p1 = 1.0, p2 = 0.0, p3;
            for (int j = 1; j <= n; ++j) {
                p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            z1 = z;
            z = z1 - p1 / pp;
        }
        x[i - 1] = xm - xl * z;
        x[n - i] = xm + xl * z;
        w[i - 1] = 2.0 * xl / ((1.0 - z * z) * pp * pp);
        w[n - i] = w[i - 1];
    }
}

/* grid.f90:14-91 -- knot sequence rt(1:nkp) and derivative coefficients Aind(nfun,2)
 * (column-major: aind[i-1] = Aind(i,1), aind[nfun+i-1] = Aind(i,2)). */
void orc_grid(const orc_cfg *c, double *rt0, double *aind)
{
    double *rt = rt0 - 1;                                   /* 1-based view */
    int nkp = c->nkp, nbc1 = c->nbc1, nbc2 = c->nbc2, k = c->k, nfun = c->nfun;
    for (int i = 1; i <= nkp; ++i) rt[i] = 0.0;
    for (int i = 1; i <= nbc1; ++i) rt[i] = c->ra;          /* :16-18 */
    for (int i = nkp - nbc2 + 1; i <= nkp; ++i) rt[i] = c->rb;  /* :19-21 */
    if (c->kind_grid == 0) {                                /* :23-29 */
        for (int i = nbc1 + 1; i <= nkp - nbc2; ++i)
            rt[i] = c->ra + (double)(i - nbc1) * c->gsize / (double)c->nointv;
    } else if (c->kind_grid == 1) {                         /* :31-42 */
        double delta = 0.01;
        double hin = log(c->gsize / delta) / (double)(c->nointv - 1);
        int j = 1;
        rt[nbc1 + 1] = delta;
        for (int i = nbc1 + 2; i <= nkp - nbc2; ++i) {
            rt[i] = rt[nbc1 + 1] * exp(hin * j);
            j = j + 1;
        }
    } else if (c->kind_grid == 2) {                         /* :44-61 */
        double delta = 0.01;
        double hin = log((c->rmax - c->ra) / delta) / (double)(c->nintv_exp - 1);
        int j = 1;
        rt[nbc1 + 1] = delta;
        for (int i = 2; i <= c->nintv_exp; ++i) {
            rt[i + nbc1] = delta * exp(hin * j);
            j = j + 1;
        }
        double dr = (c->rb - c->rmax) / (double)c->nintv_lin;
        for (int i = c->nintv_exp + 1; i <= c->nointv; ++i)
            rt[i + nbc1] = c->rmax + (double)(i - c->nintv_exp) * dr;
    }
    for (int i = 1; i <= nfun; ++i) {                       /* :79-91 */
        double A1 = 0.0, A2 = 0.0;
        double dr = rt[i + k - 1] - rt[i];
        if (dr > 0.0) A1 = 1.0 / dr;
        dr = rt[i + k] - rt[i + 1];
        if (dr > 0.0) A2 = 1.0 / dr;
        aind[i - 1] = A1;
        aind[nfun + i - 1] = A2;
    }
}

/* interv.f90:86-117 -- linear scan from the top; returns 1-based left, sets *mflag. */
int orc_interv(const double *xt0, int lxt, double x, int *mflag)
{
    const double *xt = xt0 - 1;
    int left = 1;
    if (x > xt[lxt]) { *mflag = 1; return 1; }
    else if (x < xt[1]) { *mflag = -1; return 1; }
    else *mflag = 0;
    if (x == xt[lxt]) {
        left = lxt;
        for (;;) {
            if (xt[left] < xt[lxt]) return left;
            left = left - 1;
            if (left < 1) return 1;   /* guard: reference would run off the array */
        }
    } else {
        int ilo = lxt - 1;
        for (;;) {
            if (x < xt[ilo + 1] && x >= xt[ilo]) { left = ilo; break; }
            ilo = ilo - 1;
            if (ilo == 0) break;
        }
    }
    return left;
}

/* bsplvb.f90:10-52 -- de Boor recurrence, index=1 only. biatx has jhigh entries. */
int orc_bsplvb(const double *t0, int jhigh, double x, int left, double *biatx0)
{
    const double *t = t0 - 1;
    double *biatx = biatx0 - 1;
    double deltal[64], deltar[64];
    int j = 1;
    biatx[1] = 1.0;
    if (jhigh <= j) return ORC_OK;
    if (t[left + 1] <= t[left]) return ORC_ERR_BSPLVB;       /* :30-34 */
    for (;;) {
        deltar[j] = t[left + j] - x;
        deltal[j] = x - t[left + 1 - j];
        double saved = 0.0;
        for (int i = 1; i <= j; ++i) {
            double term = biatx[i] / (deltar[i] + deltal[j + 1 - i]);
            biatx[i] = saved + deltar[i] * term;
            saved = deltal[j + 1 - i] * term;
        }
        biatx[j + 1] = saved;
        j = j + 1;
        if (jhigh <= j) break;
    }
    return ORC_OK;
}

/* Modules.f90:71-110 -- all k non-zero B-splines and first derivatives at r.
 * bsp[k], dbsp[k]; returns status, *left_out 1-based. */
int orc_bspall(const orc_cfg *c, const double *rt, const double *aind, double r,
               int *left_out, double *bsp, double *dbsp)
{
    int k = c->k, nfun = c->nfun, mflag;
    double bsp1[64], bspp[66];
    for (int j = 0; j < k; ++j) bsp[j] = 0.0;
    for (int j = 0; j < k - 1; ++j) bsp1[j] = 0.0;
    int left = orc_interv(rt, c->nkp, r, &mflag);
    int st = orc_bsplvb(rt, k, r, left, bsp);
    if (st) return st;
    st = orc_bsplvb(rt, k - 1, r, left, bsp1);
    if (st) return st;
    for (int j = 0; j <= k; ++j) bspp[j] = 0.0;             /* bspp(1:k+1) */
    for (int j = 1; j <= k - 1; ++j) bspp[j] = bsp1[j - 1]; /* bspp(j+1)=bsp1(j) */
    for (int j = 1; j <= k; ++j) {
        int jp = j + (left - k);
        double A1 = 0.0, A2 = 0.0;
        if (jp >= 1 && jp <= nfun) { A1 = aind[jp - 1]; A2 = aind[nfun + jp - 1]; }
        double b1 = bspp[j - 1], b2 = bspp[j];
        dbsp[j - 1] = (double)(k - 1) * (A1 * b1 - A2 * b2);
    }
    *left_out = left;
    return ORC_OK;
}

/* Modules.f90:263-295 -- central potential. */
double orc_selpot(const orc_cfg *c, double r)
{
    double Vr = 0.0;
    if (c->kind_pot == 0) {
        Vr = -c->zatom / r;
    } else if (c->kind_pot == 1) {
        Vr = 0.0;
        for (int i = 0; i < 3; ++i) {
            int ni = c->numn[i];
            Vr = Vr + ni * exp(-c->alphan[i] * r);
        }
        Vr = -1.0 * (c->zatom - c->ntot + Vr) / r;
    } else if (c->kind_pot == 2) {
        Vr = -c->zatom / r;
    }
    return Vr;
}

/* matrices.f90:68-186 -- reference-faithful per-pair assembly (KIND_PI = 0 terms only).
 * Dense column-major outputs S,V,T (nfun x nfun) and U (nfun x nfun x (lmax+1)).
 * Cost is the reference's: nfun*k^2*ka BSPALL calls each with an O(nkp) interv scan. */
int orc_matrix_svt(const orc_cfg *c, const double *rt0, const double *aind,
                   const double *xg, const double *wg,
                   double *S, double *V, double *T, double *U)
{
    const double *rt = rt0 - 1;
    int nfun = c->nfun, k = c->k, ka = c->ka, lmax = c->lmax;
    size_t nn = (size_t)nfun * nfun;
    double bsp[64], dbsp[64];
    double *sumU = (double *)malloc(sizeof(double) * (lmax + 1));
    memset(S, 0, nn * sizeof(double));
    memset(V, 0, nn * sizeof(double));
    memset(T, 0, nn * sizeof(double));
    memset(U, 0, nn * (lmax + 1) * sizeof(double));
    for (int ibra = 1; ibra <= nfun; ++ibra) {
        for (int jket = 1; jket <= nfun; ++jket) {
            int ibetmin = ibra > jket ? ibra : jket;                    /* :71 */
            int ibetmax = (ibra < jket ? ibra : jket) + k - 1;          /* :72 */
            double sumS = 0.0, sumT = 0.0, sumV = 0.0;
            for (int l = 0; l <= lmax; ++l) sumU[l] = 0.0;
            for (int ibet = ibetmin; ibet <= ibetmax; ++ibet) {
                double f1 = (rt[ibet + 1] + rt[ibet]) / 2.0;            /* :91 */
                double f2 = (rt[ibet + 1] - rt[ibet]) / 2.0;            /* :92 */
                for (int igl = 1; igl <= ka; ++igl) {
                    double r = f1 + xg[igl - 1] * f2;                   /* :96 */
                    double dr = f2 * wg[igl - 1];                       /* :97 */
                    int left;
                    int st = orc_bspall(c, rt0, aind, r, &left, bsp, dbsp);
                    if (st) { free(sumU); return st; }
                    if (r == 0.0) r = DBL_EPSILON;                      /* :102 */
                    double Vpot = orc_selpot(c, r);
                    int ifun = ibra - (left - k);
                    int jfun = jket - (left - k);
                    double fbra = bsp[ifun - 1], fket = bsp[jfun - 1];
                    double dfbra = dbsp[ifun - 1], dfket = dbsp[jfun - 1];
                    sumS = sumS + fbra * fket * dr;                     /* :145 */
                    sumV = sumV + fbra * Vpot * fket * dr;              /* :146 */
                    sumT = sumT + dfbra * 0.5 * dfket * dr;             /* :147 */
                    for (int lf = 0; lf <= lmax; ++lf) {                /* :148-153 */
                        double Vcent = (double)(lf * (lf + 1)) / (2.0 * (r * r));
                        double Vl = 0.0;
                        if (c->kind_pot == 2) Vl = (lf <= 3 ? c->bl[lf] : 0.0) / (r * r);
                        sumU[lf] = sumU[lf] + fbra * (Vcent + Vl) * fket * dr;
                    }
                }
            }
            size_t off = (size_t)(jket - 1) * nfun + (ibra - 1);
            S[off] = sumS; V[off] = sumV; T[off] = sumT;
            for (int lf = 0; lf <= lmax; ++lf) U[(size_t)lf * nn + off] = sumU[lf];
        }
    }
    free(sumU);
    return ORC_OK;
}

/* Same sums as orc_matrix_svt but only for the upper band (jket = ibra .. ibra+k-1),
 * with `left` found by bisection instead of the linear scan (same result for
 * non-decreasing knots).  Accumulation order per element is unchanged (intervals
 * ascending, points ascending), so values are bit-identical to the per-pair loop.
 * Band layout: X[d*nfun + (i-1)] = X(i, i+d), d = 0..k-1.  HB[l] = (T + U_l) + V
 * exactly as matrices.f90:244.  Used for sizes where the O(nfun^2) dense form or the
 * reference's O(nkp) scan is too slow for a test. */
int orc_assemble_bands(const orc_cfg *c, const double *rt0, const double *aind,
                       const double *xg, const double *wg, int l0, int nl,
                       double *SB, double *HB)
{
    const double *rt = rt0 - 1;
    int nfun = c->nfun, k = c->k, ka = c->ka, nkp = c->nkp;
    /* point table */
    int npt = (nkp - 1) * ka;
    double *tb = (double *)malloc(sizeof(double) * (size_t)npt * (2 * k + 3));
    int *tl = (int *)malloc(sizeof(int) * npt);
    for (int ibet = 1; ibet <= nkp - 1; ++ibet) {
        double f1 = (rt[ibet + 1] + rt[ibet]) / 2.0;
        double f2 = (rt[ibet + 1] - rt[ibet]) / 2.0;
        for (int igl = 1; igl <= ka; ++igl) {
            int p = (ibet - 1) * ka + (igl - 1);
            double *e = tb + (size_t)p * (2 * k + 3);
            double r = f1 + xg[igl - 1] * f2, dr = f2 * wg[igl - 1];
            int left;
            int st = orc_bspall(c, rt0, aind, r, &left, e, e + k);
            if (st) { free(tb); free(tl); return st; }
            if (r == 0.0) r = DBL_EPSILON;
            e[2 * k] = r; e[2 * k + 1] = dr; e[2 * k + 2] = orc_selpot(c, r);
            tl[p] = left;
        }
    }
    for (int d = 0; d < k; ++d) {
        for (int ibra = 1; ibra <= nfun; ++ibra) {
            int jket = ibra + d;
            size_t off = (size_t)d * nfun + (ibra - 1);
            if (jket > nfun) {
                SB[off] = 0.0;
                for (int l = 0; l < nl; ++l) HB[(size_t)l * k * nfun + off] = 0.0;
                continue;
            }
            double sumS = 0.0, sumT = 0.0, sumV = 0.0;
            double sumU[1024];
            for (int l = 0; l < nl; ++l) sumU[l] = 0.0;
            for (int ibet = jket; ibet <= ibra + k - 1; ++ibet) {
                for (int igl = 1; igl <= ka; ++igl) {
                    int p = (ibet - 1) * ka + (igl - 1);
                    const double *e = tb + (size_t)p * (2 * k + 3);
                    int left = tl[p];
                    int ifun = ibra - (left - k), jfun = jket - (left - k);
                    double fbra = e[ifun - 1], fket = e[jfun - 1];
                    double dfbra = e[k + ifun - 1], dfket = e[k + jfun - 1];
                    double r = e[2 * k], dr = e[2 * k + 1], Vpot = e[2 * k + 2];
                    sumS = sumS + fbra * fket * dr;
                    sumV = sumV + fbra * Vpot * fket * dr;
                    sumT = sumT + dfbra * 0.5 * dfket * dr;
                    for (int l = 0; l < nl; ++l) {
                        int lf = l0 + l;
                        double Vcent = (double)((long long)lf * (lf + 1)) / (2.0 * (r * r));
                        double Vl = 0.0;
                        if (c->kind_pot == 2) Vl = (lf <= 3 ? c->bl[lf] : 0.0) / (r * r);
                        sumU[l] = sumU[l] + fbra * (Vcent + Vl) * fket * dr;
                    }
                }
            }
            SB[off] = sumS;
            for (int l = 0; l < nl; ++l)
                HB[(size_t)l * k * nfun + off] = (sumT + sumU[l]) + sumV;   /* :244 */
        }
    }
    free(tb); free(tl);
    return ORC_OK;
}

/* SURVEY 8(f).2 -- the dipole matrices MATRIX_SVT accumulates in the same quadrature loop and keeps in
 * rij for KIND_PI = 1, 2 (matrices.f90:141-144, 159-163):
 *   RB[0] = sumr = int B_i r B_j dr        -> rij(:,:,1) for KIND_PI = 1 (length gauge)
 *   RB[1] = sumc = int B_i (1/r) B_j dr    -> rij(:,:,1) for KIND_PI = 2 (velocity gauge)
 *   RB[2] = sumd = int B_i B_j' dr         -> rij(:,:,2) for KIND_PI = 2
 * The reference fills both triangles (jket = 1..nfun, matrices.f90:69) and they are not bit-symmetric, so the
 * FULL band is returned: RB[(c*(2k-1) + (d + k-1))*nfun + (i-1)] = X_c(i, i+d), d = -(k-1)..k-1; same
 * interval range (ibetmin..ibetmax, :71-72), point order and expression order as the reference
 * (left to right: ((fbra * (1/r)) * fket) * dr). */
int orc_dipole_bands(const orc_cfg *c, const double *rt0, const double *aind,
                     const double *xg, const double *wg, double *RB)
{
    const double *rt = rt0 - 1;
    int nfun = c->nfun, k = c->k, ka = c->ka, nkp = c->nkp;
    int npt = (nkp - 1) * ka, nd = 2 * k - 1;
    double *tb = (double *)malloc(sizeof(double) * (size_t)npt * (2 * k + 3));
    int *tl = (int *)malloc(sizeof(int) * npt);
    for (int ibet = 1; ibet <= nkp - 1; ++ibet) {
        double f1 = (rt[ibet + 1] + rt[ibet]) / 2.0;
        double f2 = (rt[ibet + 1] - rt[ibet]) / 2.0;
        for (int igl = 1; igl <= ka; ++igl) {
            int p = (ibet - 1) * ka + (igl - 1);
            double *e = tb + (size_t)p * (2 * k + 3);
            double r = f1 + xg[igl - 1] * f2, dr = f2 * wg[igl - 1];
            int left;
            int st = orc_bspall(c, rt0, aind, r, &left, e, e + k);
            if (st) { free(tb); free(tl); return st; }
            if (r == 0.0) r = DBL_EPSILON;
            e[2 * k] = r; e[2 * k + 1] = dr; e[2 * k + 2] = 0.0;
            tl[p] = left;
        }
    }
    for (int d = -(k - 1); d <= k - 1; ++d) {
        for (int ibra = 1; ibra <= nfun; ++ibra) {
            int jket = ibra + d;
            size_t off = (size_t)(d + k - 1) * nfun + (ibra - 1);
            double sumr = 0.0, sumc = 0.0, sumd = 0.0;
            if (jket >= 1 && jket <= nfun) {
                int ibetmin = ibra > jket ? ibra : jket;                     /* :71 */
                int ibetmax = (ibra < jket ? ibra : jket) + k - 1;           /* :72 */
                for (int ibet = ibetmin; ibet <= ibetmax; ++ibet) {
                    for (int igl = 1; igl <= ka; ++igl) {
                        int p = (ibet - 1) * ka + (igl - 1);
                        const double *e = tb + (size_t)p * (2 * k + 3);
                        int left = tl[p];
                        int ifun = ibra - (left - k), jfun = jket - (left - k);
                        double fbra = e[ifun - 1], fket = e[jfun - 1], dfket = e[k + jfun - 1];
                        double r = e[2 * k], dr = e[2 * k + 1];
                        sumc = sumc + fbra * (1.0 / r) * fket * dr;     /* :141 */
                        sumd = sumd + fbra * dfket * dr;                /* :142 */
                        sumr = sumr + fbra * r * fket * dr;             /* :144 */
                    }
                }
            }
            RB[off] = sumr;
            RB[(size_t)nd * nfun + off] = sumc;
            RB[(size_t)2 * nd * nfun + off] = sumd;
        }
    }
    free(tb); free(tl);
    return ORC_OK;
}

/* Bsp_Atom.f90:118-146 -- tabulate u(r) = sum_j c_j B_j(r) on npts+1 points. */
int orc_write_wf(const orc_cfg *c, const double *rt0, const double *ci, int n,
                 int npts, double *r_out, double *u_out)
{
    int k = c->k, mflag;
    double bsp[64];
    double dr = (c->rb - c->ra) / (double)npts;
    for (int i = 0; i <= npts; ++i) {
        double r = c->ra + (double)i * dr;
        for (int j = 0; j < k; ++j) bsp[j] = 0.0;
        int left = orc_interv(rt0, c->nkp, r, &mflag);
        int st = orc_bsplvb(rt0, k, r, left, bsp);
        if (st) return st;
        int jmin = left - k + 1, jmax = jmin + k - 1;
        double sumf = 0.0;
        for (int j = jmin; j <= jmax; ++j) {
            int jfun = j - (left - k);
            double fr = 0.0;
            if (j >= 1 && j <= n) fr = ci[j - 1];
            if (jfun >= 1 && jfun <= k) sumf = sumf + fr * bsp[jfun - 1];
        }
        r_out[i] = r; u_out[i] = sumf;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------
 * Plain dense generalized symmetric eigensolver, the published LAPACK DSYGV(1,'V','U')
 * algorithm chain in its unblocked textbook form (the reference links Intel MKL,
 * version unpinned -- matrices.f90:248, src/Makefile:23):
 *   DPOTRF 'U'  : B = U^T U                       (info = n+i if minor i not PD)
 *   DSYGST 1,'U': C = U^-T A U^-1
 *   DSYTRD/DORGTR: Householder tridiagonalisation, accumulating Q
 *   DSTEQR      : implicit QL with Wilkinson shift     (info = i on non-convergence)
 *   DTRSM       : x = U^-1 y
 * A, B column-major n x n, only the upper triangles are read.  On exit w ascending,
 * A <- eigenvectors (Z^T B Z = I), B <- U.  Small sizes only (O(n^3) scalar code).
 * ---------------------------------------------------------------------------------- */
static double pythag(double a, double b) { return hypot(a, b); }

int orc_dsygv(int n, double *A, double *B, double *w)
{
#define a(i, j) A[(size_t)(j) * n + (i)]
#define b(i, j) B[(size_t)(j) * n + (i)]
    /* Cholesky B = U^T U (upper) */
    for (int j = 0; j < n; ++j) {
        double s = b(j, j);
        for (int p = 0; p < j; ++p) s -= b(p, j) * b(p, j);
        if (s <= 0.0 || s != s) return n + j + 1;
        s = sqrt(s);
        b(j, j) = s;
        for (int i = j + 1; i < n; ++i) {
            double t = b(j, i);
            for (int p = 0; p < j; ++p) t -= b(p, j) * b(p, i);
            b(j, i) = t / s;
        }
        for (int i = j + 1; i < n; ++i) b(i, j) = 0.0;
    }
    /* symmetrise A from its upper triangle */
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) a(i, j) = a(j, i);
    /* C = U^-T A U^-1: first X = U^-T A (forward substitution on columns) */
    for (int col = 0; col < n; ++col)
        for (int i = 0; i < n; ++i) {
            double s = a(i, col);
            for (int p = 0; p < i; ++p) s -= b(p, i) * a(p, col);
            a(i, col) = s / b(i, i);
        }
    /* then C = X U^-1 (rows of X): C(i,:) U = X(i,:) */
    for (int row = 0; row < n; ++row)
        for (int j = 0; j < n; ++j) {
            double s = a(row, j);
            for (int p = 0; p < j; ++p) s -= a(row, p) * b(p, j);
            a(row, j) = s / b(j, j);
        }
    /* enforce symmetry (average) */
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) {
            double m = 0.5 * (a(i, j) + a(j, i));
            a(i, j) = m; a(j, i) = m;
        }
    /* Householder tridiagonalisation with accumulation (tred2) */
    double *e = (double *)calloc(n, sizeof(double));
    double *d = w;
    for (int i = n - 1; i > 0; --i) {
        int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int kk = 0; kk <= l; ++kk) scale += fabs(a(i, kk));
            if (scale == 0.0) e[i] = a(i, l);
            else {
                for (int kk = 0; kk <= l; ++kk) { a(i, kk) /= scale; h += a(i, kk) * a(i, kk); }
                double f = a(i, l);
                double g = (f >= 0.0 ? -sqrt(h) : sqrt(h));
                e[i] = scale * g;
                h -= f * g;
                a(i, l) = f - g;
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    a(j, i) = a(i, j) / h;
                    g = 0.0;
                    for (int kk = 0; kk <= j; ++kk) g += a(j, kk) * a(i, kk);
                    for (int kk = j + 1; kk <= l; ++kk) g += a(kk, j) * a(i, kk);
                    e[j] = g / h;
                    f += e[j] * a(i, j);
                }
                double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = a(i, j);
                    e[j] = g = e[j] - hh * f;
                    for (int kk = 0; kk <= j; ++kk) a(j, kk) -= (f * e[kk] + g * a(i, kk));
                }
            }
        } else e[i] = a(i, l);
        d[i] = h;
    }
    d[0] = 0.0; e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        int l = i - 1;
        if (d[i] != 0.0) {
            for (int j = 0; j <= l; ++j) {
                double g = 0.0;
                for (int kk = 0; kk <= l; ++kk) g += a(i, kk) * a(kk, j);
                for (int kk = 0; kk <= l; ++kk) a(kk, j) -= g * a(kk, i);
            }
        }
        d[i] = a(i, i);
        a(i, i) = 1.0;
        for (int j = 0; j <= l; ++j) a(j, i) = a(i, j) = 0.0;
    }
    /* implicit QL (tqli) */
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) <= DBL_EPSILON * dd) break;
            }
            if (m != l) {
                if (iter++ == 60) { free(e); return l + 1; }
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = pythag(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i], bb = c * e[i];
                    e[i + 1] = (r = pythag(f, g));
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * bb;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - bb;
                    for (int kk = 0; kk < n; ++kk) {
                        f = a(kk, i + 1);
                        a(kk, i + 1) = s * a(kk, i) + c * f;
                        a(kk, i) = c * a(kk, i) - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    free(e);
    /* sort ascending (selection sort with column swaps) */
    for (int i = 0; i < n - 1; ++i) {
        int kmin = i;
        for (int j = i + 1; j < n; ++j) if (d[j] < d[kmin]) kmin = j;
        if (kmin != i) {
            double t = d[i]; d[i] = d[kmin]; d[kmin] = t;
            for (int r = 0; r < n; ++r) { t = a(r, i); a(r, i) = a(r, kmin); a(r, kmin) = t; }
        }
    }
    /* back-transform x = U^-1 y */
    for (int col = 0; col < n; ++col)
        for (int i = n - 1; i >= 0; --i) {
            double s = a(i, col);
            for (int p = i + 1; p < n; ++p) s -= b(i, p) * a(p, col);
            a(i, col) = s / b(i, i);
        }
    return 0;
#undef a
#undef b
}
