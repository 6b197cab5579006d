// host_setup.cpp -- host-side, once-per-problem set-up of libbspatom (plain C++, no HIP):
// derived sizes (READ_INPUTS), knot sequence and Aind (GRID), Gauss-Legendre rule (gauleg) and
// the central-potential table (SELPOT).  These are O(nfun) scalar loops that the reference also
// runs once on the host; they are kept bit-compatible with it (no FMA contraction: this file is
// compiled with -ffp-contract=off) because the assembled matrices are checked bit-for-bit.
#include "host_setup.h"
#include <cfloat>
#include <cmath>

namespace bsp {

// ReadInputs.f90:27-36, :75-84
void input_defaults(bspatom_input *in)
{
    in->kind_grid = 0; in->ra = 0.0; in->rb = 0.0; in->rmax = 0.0;
    in->k = 0; in->ka = 0; in->nfun = 0; in->kind_bc1 = 0; in->kind_bc2 = 0;
    in->kind_pot = 0; in->n0_ini = 1; in->l_ini = 0; in->m_ini = 0; in->l_fin = 0; in->lmax = 0;
    in->emax_fin = -1.0; in->zatom = 1.0;
}

// ReadInputs.f90:39-69 (sizes, exp-linear resize), :87 (lmax), :95-141 (potential parameters)
int derive(const bspatom_input &in, HostSetup *h)
{
    h->in = in;
    int k = in.k, nfun = in.nfun;
    h->ka = (in.ka == 0) ? k + 3 : in.ka;
    h->nbc1 = (in.kind_bc1 == 0) ? k - 1 : k;
    h->nbc2 = (in.kind_bc2 == 0) ? k - 1 : k;
    int nkp = nfun + k;
    int nointv = nkp - h->nbc1 - h->nbc2 + 1;
    h->gsize = in.rb - in.ra;
    h->nintv_exp = 0; h->nintv_lin = 0;
    if (in.kind_grid == 2) {
        const double dx = h->gsize / nointv;
        const double rimax = (in.rmax - in.ra) / dx;
        const int imax = (int)std::lround(rimax);            // NINT
        h->nintv_exp = 3 * imax;
        h->nintv_lin = nointv - imax;
        nointv = h->nintv_exp + h->nintv_lin;
        nkp = nointv + h->nbc1 + h->nbc2 - 1;
        nfun = nkp - k;
    }
    h->nfun = nfun; h->k = k; h->nkp = nkp; h->nointv = nointv;
    h->lmax = (in.l_fin > in.lmax) ? in.l_fin : in.lmax;
    if (k < 2 || nfun < k || h->ka < 1 || nointv < 1) return -2;
    if (k > 16 || h->ka > 32) return -5;                      // BSPATOM_MAX_K / BSPATOM_MAX_KA (include/bspatom.h): device tables
    if (in.kind_grid < 0 || in.kind_grid > 2 || in.kind_pot < 0 || in.kind_pot > 2) return -2;
    if (in.kind_grid == 2 && (h->nintv_exp < 2 || h->nintv_lin < 1)) return -2;
    if (in.kind_grid == 1 && nointv < 2) return -2;
    // A box of no extent or not-a-number parameters: the reference runs on (all knots coincide, BSPLVB STOPs with
    // 'FATAL ERROR' or the matrices fill with NaN); here they are an argument error before anything is launched.
    if (!std::isfinite(in.ra) || !std::isfinite(in.rb) || !std::isfinite(in.rmax) || !std::isfinite(in.zatom) ||
        !std::isfinite(in.emax_fin) || !(in.rb > in.ra)) return -2;
    if (in.kind_grid == 1 && !(h->gsize > 0.01)) return -2;                  // exponential grid: first interior knot at 0.01 (grid.f90:35)
    if (in.n0_ini < 1 || in.l_ini < 0 || in.l_fin < 0 || h->lmax < 0) return -2;
    h->ntot = 0;
    for (int i = 0; i < 3; ++i) { h->alphan[i] = 0.0; h->numn[i] = 0; }
    for (int i = 0; i < 4; ++i) h->bl[i] = 0.0;
    if (in.kind_pot == 1) {                                  // Rogers potential, Ca+ coefficients
        static const double aj[3][4] = {{0.8855, 0.2549, -0.0901, 0.0},
                                        {0.3386, 1.1323, -0.4904, 0.0},
                                        {0.1437, 0.9129, -0.6940, 0.2503}};
        h->numn[0] = 2; h->numn[1] = 8; h->numn[2] = 8;
        for (int i = 0; i < 3; ++i) {
            h->ntot += h->numn[i];                           // the running Ntot enters xn
            double xn = (double)(in.zatom - h->ntot);
            if (xn == 0.0) xn = 1.0;
            double suman = 0.0, xp = 1.0;
            for (int j = 0; j <= 3; ++j) {
                suman = suman + aj[i][j] / xp;
                xp *= xn;
            }
            h->alphan[i] = (xn + 1.0) * suman;
        }
    } else if (in.kind_pot == 2) {                           // Simons-Fues, Rb
        h->bl[0] = 0.72657; h->bl[1] = 0.47095; h->bl[2] = -0.55508; h->bl[3] = -0.04008;
    }
    return 0;
}

// Modules.f90:112-153.  `pp` deliberately outlives the node loop: for odd n the middle node
// starts within tolerance of z1 = 0, the Newton loop is skipped, and the reference forms that
// weight from the previous node's pp.  Reproduced, since it changes every matrix element.
void gauleg(double x1, double x2, double *x, double *w, int n)
{
    const double pi = std::acos(-1.0);
    const double tol = DBL_EPSILON * 10;
    const int m = (n + 1) / 2;
    const double xm = 0.5 * (x2 + x1), xl = 0.5 * (x2 - x1);
    double pp = 0.0;
    for (int i = 1; i <= m; ++i) {
        double z = std::cos(pi * (i - .25) / (n + .5));
        double z1 = 0.0;
        while (std::fabs(z - z1) > tol) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double p3 = p2;
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

// grid.f90:14-91
void build_grid(HostSetup *h)
{
    const int nkp = h->nkp, nbc1 = h->nbc1, nbc2 = h->nbc2, k = h->k, nfun = h->nfun;
    const double ra = h->in.ra, rb = h->in.rb, rmax = h->in.rmax;
    h->rt.assign(nkp, 0.0);
    double *rt = h->rt.data() - 1;                           // 1-based view
    for (int i = 1; i <= nbc1; ++i) rt[i] = ra;
    for (int i = nkp - nbc2 + 1; i <= nkp; ++i) rt[i] = rb;
    if (h->in.kind_grid == 0) {
        for (int i = nbc1 + 1; i <= nkp - nbc2; ++i)
            rt[i] = ra + (double)(i - nbc1) * h->gsize / (double)h->nointv;
    } else if (h->in.kind_grid == 1) {
        const double delta = 0.01;
        const double hin = std::log(h->gsize / delta) / (double)(h->nointv - 1);
        rt[nbc1 + 1] = delta;
        int j = 1;
        for (int i = nbc1 + 2; i <= nkp - nbc2; ++i, ++j) rt[i] = rt[nbc1 + 1] * std::exp(hin * j);
    } else {
        const double delta = 0.01;
        const double hin = std::log((rmax - ra) / delta) / (double)(h->nintv_exp - 1);
        rt[nbc1 + 1] = delta;
        int j = 1;
        for (int i = 2; i <= h->nintv_exp; ++i, ++j) rt[i + nbc1] = delta * std::exp(hin * j);
        const double dr = (rb - rmax) / (double)h->nintv_lin;
        for (int i = h->nintv_exp + 1; i <= h->nointv; ++i)
            rt[i + nbc1] = rmax + (double)(i - h->nintv_exp) * dr;
    }
    h->aind.assign(2 * (size_t)nfun, 0.0);
    for (int i = 1; i <= nfun; ++i) {
        double a1 = 0.0, a2 = 0.0;
        double dr = rt[i + k - 1] - rt[i];
        if (dr > 0.0) a1 = 1.0 / dr;
        dr = rt[i + k] - rt[i + 1];
        if (dr > 0.0) a2 = 1.0 / dr;
        h->aind[i - 1] = a1;
        h->aind[nfun + i - 1] = a2;
    }
    h->xg.assign(h->ka, 0.0);
    h->wg.assign(h->ka, 0.0);
    gauleg(-1.0, 1.0, h->xg.data(), h->wg.data(), h->ka);
}

// Modules.f90:263-295
double selpot(const HostSetup &h, double r)
{
    double vr = 0.0;
    if (h.in.kind_pot == 0) vr = -h.in.zatom / r;
    else if (h.in.kind_pot == 1) {
        for (int i = 0; i < 3; ++i) vr = vr + h.numn[i] * std::exp(-h.alphan[i] * r);
        vr = -1.0 * (h.in.zatom - h.ntot + vr) / r;
    } else vr = -h.in.zatom / r;
    return vr;
}

// SELPOT at every quadrature point of every knot interval (matrices.f90:91-103):
// vpot[(ibet-1)*ka + g], r = f1 + xg*f2 with r = 0 replaced by eps.
void build_vpot(HostSetup *h)
{
    const int nint = h->nkp - 1, ka = h->ka;
    h->vpot.assign((size_t)nint * ka, 0.0);
    const double *rt = h->rt.data() - 1;
    for (int ibet = 1; ibet <= nint; ++ibet) {
        const double f1 = (rt[ibet + 1] + rt[ibet]) / 2.0;
        const double f2 = (rt[ibet + 1] - rt[ibet]) / 2.0;
        for (int g = 0; g < ka; ++g) {
            double r = f1 + h->xg[g] * f2;
            if (r == 0.0) r = DBL_EPSILON;
            h->vpot[(size_t)(ibet - 1) * ka + g] = selpot(*h, r);
        }
    }
}

}  // namespace bsp
