// assemble.hip -- Gauss-Legendre assembly of the banded overlap S and Hamiltonians H(l).
//
// Replaces MATRIX_SVT (reference matrices.f90:68-186) + BSPALL (Modules.f90:71-110) + bsplvb
// (bsplvb.f90:10-52) + interv (interv.f90:86-117) and the H = T + U(l) + V sum of SOLVE_SYSTEM
// (matrices.f90:244).  Results are BIT-IDENTICAL to the reference's dense matrices on their
// upper band: every product and sum is formed in the reference's order, this file is compiled
// with -ffp-contract=off (no FMA), divisions are IEEE, and SELPOT values come from the host
// table (libm exp for KIND_POT=1 is not reproducible on the device).
//
// Kernel 1 (point_table_kernel): ONE WAVEFRONT PER KNOT INTERVAL [rt(ibet), rt(ibet+1)]; lane
// g < ka owns Gauss-Legendre point g, runs interv + the two bsplvb recurrences with its
// deltal/deltar work arrays in LDS, and writes B_j(r), B_j'(r), r, dr, V(r) and `left` to the
// point table in HBM (coalesced rows of 2k+3 doubles).
// Kernel 2 (band_kernel): one workgroup owns TI consecutive rows of the band; it stages the
// point-table rows of the TI+k-1 intervals those rows touch in LDS, then each thread walks
// (row i, diagonal d, channel l) items and accumulates S, V, T, U_l over intervals ascending and
// points ascending -- the reference's summation order -- and stores SB[d][i], HB[l][d][i] with i
// fastest (coalesced).
//
// Band layout (upper): SB[d*nfun + i] = S(i, i+d), HB[(l*k + d)*nfun + i] = H_l(i, i+d),
// 0-based i, d = 0..k-1, zero where i+d >= nfun.
#include "common.h"

namespace bsp {

constexpr int KMAX = 16;   // B-spline order limit of the device tables

// interv.f90:86-117 semantics (largest ilo with xt(ilo) <= x < xt(ilo+1), scanning from the top),
// evaluated as: try the caller's guess first, otherwise the faithful scan.  1-based result.
__device__ static int interv_dev(const double *__restrict__ xt0, int lxt, double x, int guess)
{
    const double *xt = xt0 - 1;
    if (x > xt[lxt]) return 1;
    if (x < xt[1]) return 1;
    if (x == xt[lxt]) {
        int left = lxt;
        while (left >= 1) {
            if (xt[left] < xt[lxt]) return left;
            --left;
        }
        return 1;
    }
    if (guess >= 1 && guess < lxt && x < xt[guess + 1] && x >= xt[guess]) {
        // the reference scans down from lxt-1 and stops at the first hit; with non-decreasing
        // knots the hit is unique, so the guess is that hit.
        return guess;
    }
    int left = 1;
    for (int ilo = lxt - 1; ilo >= 1; --ilo)
        if (x < xt[ilo + 1] && x >= xt[ilo]) { left = ilo; break; }
    return left;
}

// bsplvb.f90:24-50, index = 1.  dl/dr: per-lane work arrays (LDS).  Returns 0 or BSP_ERR_BSPLVB.
__device__ static int bsplvb_dev(const double *__restrict__ t0, int jhigh, double x, int left,
                                 double *biatx0, double *dl0, double *dr0)
{
    const double *t = t0 - 1;
    double *biatx = biatx0 - 1, *deltal = dl0 - 1, *deltar = dr0 - 1;
    int j = 1;
    biatx[1] = 1.0;
    if (jhigh <= j) return 0;
    if (t[left + 1] <= t[left]) return BSP_ERR_BSPLVB;
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
    return 0;
}

// grid: one workgroup of WPB wavefronts handles WPB intervals; wave w -> interval ibet.
constexpr int PT_WPB = 2;
constexpr int KAMAX = 32;  // Gauss-Legendre points per interval limit
__global__ __launch_bounds__(64 * PT_WPB) void point_table_kernel(
    int nkp, int k, int ka, int nfun, const double *__restrict__ rt0, const double *__restrict__ aind,
    const double *__restrict__ xg, const double *__restrict__ wg, const double *__restrict__ vpot,
    double *__restrict__ ptab, int *__restrict__ leftv, int *status)
{
    // per-lane scratch in LDS: bsp[k], bsp1[k], deltal[k], deltar[k]
    __shared__ double lds[PT_WPB][KAMAX][4 * KMAX + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ibet = blockIdx.x * PT_WPB + wave + 1;          // 1-based interval
    if (ibet > nkp - 1 || lane >= ka) return;
    const double *rt = rt0 - 1;
    double *bsp = &lds[wave][lane][0], *bsp1 = bsp + KMAX, *dl = bsp + 2 * KMAX, *dr_ = bsp + 3 * KMAX;
    const double f1 = (rt[ibet + 1] + rt[ibet]) / 2.0;       // matrices.f90:91
    const double f2 = (rt[ibet + 1] - rt[ibet]) / 2.0;       // :92
    double r = f1 + xg[lane] * f2;                           // :96
    const double dr = f2 * wg[lane];                         // :97
    // BSPALL, Modules.f90:84-108
    for (int j = 0; j < k; ++j) { bsp[j] = 0.0; bsp1[j] = 0.0; }
    const int left = interv_dev(rt0, nkp, r, ibet);
    int st = bsplvb_dev(rt0, k, r, left, bsp, dl, dr_);
    if (!st) st = bsplvb_dev(rt0, k - 1, r, left, bsp1, dl, dr_);
    if (st) { atomicExch(status, BSP_ERR_BSPLVB); return; }
    const int p = (ibet - 1) * ka + lane;
    double *e = ptab + (size_t)p * (2 * k + 3);
    for (int j = 1; j <= k; ++j) {
        const int jp = j + (left - k);
        double A1 = 0.0, A2 = 0.0;
        if (jp >= 1 && jp <= nfun) { A1 = aind[jp - 1]; A2 = aind[nfun + jp - 1]; }
        const double b1 = (j >= 2) ? bsp1[j - 2] : 0.0;       // bspp(j)   = bsp1(j-1), bspp(1) = 0
        const double b2 = (j <= k - 1) ? bsp1[j - 1] : 0.0;   // bspp(j+1) = bsp1(j),  bspp(k+1) = 0
        e[j - 1] = bsp[j - 1];
        e[k + j - 1] = (double)(k - 1) * (A1 * b1 - A2 * b2);
    }
    if (r == 0.0) r = 2.220446049250313e-16;                  // :102, eps = EPSILON(1.D0)
    e[2 * k] = r;
    e[2 * k + 1] = dr;
    e[2 * k + 2] = vpot[p];                                   // SELPOT(r), host table
    leftv[p] = left;
}

constexpr int BAND_LCH = 16;  // channels per workgroup (grid.y chunks the l range)
constexpr int BAND_LG = 4;    // channels per item (they share the item's reads of the point table)

// TI = band rows per workgroup (runtime: chosen so that the staged table fits in LDS)
__global__ __launch_bounds__(256) void band_kernel(int TI, int nfun, int k, int ka, int nkp, int kind_pot,
                                                  const double *__restrict__ bl, int l0, int nl,
                                                  const double *__restrict__ ptab,
                                                  const int *__restrict__ leftv,
                                                  double *__restrict__ SB, double *__restrict__ HB)
{
    extern __shared__ double sm[];
    const int W = 2 * k + 3;
    const int i0 = blockIdx.x * TI;                           // first 0-based row of this block
    const int lbeg = blockIdx.y * BAND_LCH;
    const int lcnt = (nl - lbeg < BAND_LCH) ? (nl - lbeg) : BAND_LCH;
    // intervals needed (1-based): ibet = jket .. ibra+k-1 with ibra = i0+1 .. i0+TI  ->
    // [i0+1, i0+TI+k-1], clipped to nkp-1
    const int ib0 = i0 + 1;
    int nint = TI + k - 1;
    if (ib0 + nint - 1 > nkp - 1) nint = nkp - 1 - ib0 + 1;
    const int npts = nint * ka;
    double *tab = sm;                                         // [npts][W]
    int *lf = reinterpret_cast<int *>(sm + (size_t)(TI + k - 1) * ka * W);
    for (int idx = threadIdx.x; idx < npts * W; idx += blockDim.x)
        tab[idx] = ptab[(size_t)(ib0 - 1) * ka * W + idx];
    for (int idx = threadIdx.x; idx < npts; idx += blockDim.x) lf[idx] = leftv[(ib0 - 1) * ka + idx];
    __syncthreads();

    const int nrow = (nfun - i0 < TI) ? (nfun - i0) : TI;
    // One item = one band entry (row ii, diagonal d) for BAND_LG consecutive channels: the basis values, r, dr and the potential of
    // a quadrature point are read from LDS once for the group (the kernel is bound by those reads -- seven per point and entry, at
    // irregular addresses: one entry per channel and item took 1.37 ms for 127 channels of n = 4096, all of it in front of the
    // reduction), the sums S, V, T are the same for every channel, and only the centrifugal sum has a term per channel.  Every sum
    // adds the same terms in the same order as before: the same bits (tests/test_gpu_stages.py: the bands against the golden files and the oracle, bit for bit).
    const int ngrp = (lcnt + BAND_LG - 1) / BAND_LG;
    const int items = TI * k * ngrp;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int ii = it % TI;
        const int d = (it / TI) % k;
        const int lg = it / (TI * k);
        const int lfirst = lbeg + lg * BAND_LG;
        const int ln = (lbeg + lcnt - lfirst < BAND_LG) ? (lbeg + lcnt - lfirst) : BAND_LG;
        if (ii >= nrow) continue;
        const int ibra = i0 + ii + 1, jket = ibra + d;        // 1-based
        const size_t off = (size_t)d * nfun + (ibra - 1);
        if (jket > nfun) {
            if (lfirst == 0) SB[off] = 0.0;
            for (int u = 0; u < ln; ++u) HB[(size_t)(lfirst + u) * k * nfun + off] = 0.0;
            continue;
        }
        double cl[BAND_LG], blv[BAND_LG], sumU[BAND_LG];
#pragma unroll
        for (int u = 0; u < BAND_LG; ++u) {
            const int lfq = l0 + lfirst + u;
            cl[u] = (double)((long long)lfq * (lfq + 1));
            blv[u] = (kind_pot == 2 && lfq <= 3) ? bl[lfq] : 0.0;
            sumU[u] = 0.0;
        }
        double sumS = 0.0, sumV = 0.0, sumT = 0.0;
        for (int ibet = jket; ibet <= ibra + k - 1; ++ibet) {
            const double *eb = tab + (size_t)(ibet - ib0) * ka * W;
            const int *lb = lf + (ibet - ib0) * ka;
            for (int g = 0; g < ka; ++g) {
                const double *e = eb + g * W;
                const int left = lb[g];
                int ifun = ibra - (left - k), jfun = jket - (left - k);
                ifun = ifun < 1 ? 1 : (ifun > k ? k : ifun);   // in range for valid knot sequences
                jfun = jfun < 1 ? 1 : (jfun > k ? k : jfun);
                const double fbra = e[ifun - 1], fket = e[jfun - 1];
                const double dfbra = e[k + ifun - 1], dfket = e[k + jfun - 1];
                const double r = e[2 * k], dr = e[2 * k + 1], Vpot = e[2 * k + 2];
                sumS = sumS + fbra * fket * dr;                       // matrices.f90:145
                sumV = sumV + fbra * Vpot * fket * dr;                // :146
                sumT = sumT + dfbra * 0.5 * dfket * dr;               // :147
#pragma unroll
                for (int u = 0; u < BAND_LG; ++u) {
                    const double Vcent = cl[u] / (2.0 * (r * r));     // :149
                    double Vl = 0.0;
                    if (kind_pot == 2) Vl = blv[u] / (r * r);         // :151
                    sumU[u] = sumU[u] + fbra * (Vcent + Vl) * fket * dr;   // :152
                }
            }
        }
        if (lfirst == 0) SB[off] = sumS;                              // :180
#pragma unroll
        for (int u = 0; u < BAND_LG; ++u)
            if (u < ln) HB[(size_t)(lfirst + u) * k * nfun + off] = (sumT + sumU[u]) + sumV;   // :244  Tij + Uij(l) + Vij
    }
}

// the dipole matrices MATRIX_SVT accumulates in the same quadrature loop and keeps in rij for
// KIND_PI = 1, 2 (matrices.f90:141-144, 159-163): sumr = int B_i r B_j, sumc = int B_i (1/r) B_j,
// sumd = int B_i B_j'.  The reference fills both triangles (jket = 1..nfun, :69) and they are not bit-symmetric,
// so the full band is produced: RB[(c*(2k-1) + (d+k-1))*nfun + i] = X_c(i, i+d), d = -(k-1)..k-1, c = 0 (r),
// 1 (1/r), 2 (d/dr).  Same staging, interval range (:71-72), point order and expression order as the reference.
__global__ __launch_bounds__(256) void dipole_band_kernel(int TI, int nfun, int k, int ka, int nkp,
                                                         const double *__restrict__ ptab,
                                                         const int *__restrict__ leftv, double *__restrict__ RB)
{
    extern __shared__ double sm[];
    const int W = 2 * k + 3, nd = 2 * k - 1;
    const int i0 = blockIdx.x * TI;
    const int ib0 = i0 + 1;
    int nint = TI + k - 1;
    if (ib0 + nint - 1 > nkp - 1) nint = nkp - 1 - ib0 + 1;
    const int npts = nint * ka;
    double *tab = sm;
    int *lf = reinterpret_cast<int *>(sm + (size_t)(TI + k - 1) * ka * W);
    for (int idx = threadIdx.x; idx < npts * W; idx += blockDim.x)
        tab[idx] = ptab[(size_t)(ib0 - 1) * ka * W + idx];
    for (int idx = threadIdx.x; idx < npts; idx += blockDim.x) lf[idx] = leftv[(ib0 - 1) * ka + idx];
    __syncthreads();
    const int nrow = (nfun - i0 < TI) ? (nfun - i0) : TI;
    for (int it = threadIdx.x; it < TI * nd; it += blockDim.x) {
        const int ii = it % TI, dd = it / TI, d = dd - (k - 1);
        if (ii >= nrow) continue;
        const int ibra = i0 + ii + 1, jket = ibra + d;        // 1-based
        const size_t off = (size_t)dd * nfun + (ibra - 1);
        double sumr = 0.0, sumc = 0.0, sumd = 0.0;
        if (jket >= 1 && jket <= nfun) {
            const int ibetmin = ibra > jket ? ibra : jket;                    // matrices.f90:71
            const int ibetmax = (ibra < jket ? ibra : jket) + k - 1;          // :72
            for (int ibet = ibetmin; ibet <= ibetmax; ++ibet) {
                const double *eb = tab + (size_t)(ibet - ib0) * ka * W;
                const int *lb = lf + (ibet - ib0) * ka;
                for (int g = 0; g < ka; ++g) {
                    const double *e = eb + g * W;
                    const int left = lb[g];
                    int ifun = ibra - (left - k), jfun = jket - (left - k);
                    ifun = ifun < 1 ? 1 : (ifun > k ? k : ifun);
                    jfun = jfun < 1 ? 1 : (jfun > k ? k : jfun);
                    const double fbra = e[ifun - 1], fket = e[jfun - 1], dfket = e[k + jfun - 1];
                    const double r = e[2 * k], dr = e[2 * k + 1];
                    sumc = sumc + fbra * (1.0 / r) * fket * dr;               // :141
                    sumd = sumd + fbra * dfket * dr;                          // :142
                    sumr = sumr + fbra * r * fket * dr;                       // :144
                }
            }
        }
        RB[off] = sumr;
        RB[(size_t)nd * nfun + off] = sumc;
        RB[(size_t)2 * nd * nfun + off] = sumd;
    }
}

int launch_dipole_bands(int nfun, int k, int ka, int nkp, const double *d_ptab, const int *d_left, double *d_RB,
                        hipStream_t st)
{
    const int W = 2 * k + 3;
    int TI = 32;
    size_t lds = 0;
    for (; TI >= 4; TI /= 2) {
        lds = (size_t)(TI + k - 1) * ka * W * sizeof(double) + (size_t)(TI + k - 1) * ka * sizeof(int);
        if (lds <= 150 * 1024) break;
    }
    if (TI < 4) return BSP_ERR_UNSUPPORTED;
    static bool attr = false;
    if (!attr) {
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dipole_band_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(dipole_band_kernel, dim3((nfun + TI - 1) / TI), dim3(256), lds, st, TI, nfun, k, ka, nkp, d_ptab, d_left,
                       d_RB);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

int launch_point_table(int nkp, int k, int ka, int nfun, const double *d_rt, const double *d_aind,
                       const double *d_xg, const double *d_wg, const double *d_vpot, double *d_ptab,
                       int *d_left, int *d_status, hipStream_t st)
{
    if (k > KMAX || ka > KAMAX || k < 2) return BSP_ERR_ARG;
    const int nint = nkp - 1;
    hipLaunchKernelGGL(point_table_kernel, dim3((nint + PT_WPB - 1) / PT_WPB), dim3(64 * PT_WPB), 0, st,
                       nkp, k, ka, nfun, d_rt, d_aind, d_xg, d_wg, d_vpot, d_ptab, d_left, d_status);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

int launch_assemble_bands(int nfun, int k, int ka, int nkp, int kind_pot, const double *d_bl, int l0,
                          int nl, const double *d_ptab, const int *d_left, double *d_SB, double *d_HB,
                          hipStream_t st)
{
    const int W = 2 * k + 3;
    int TI = 32;
    size_t lds = 0;
    for (; TI >= 4; TI /= 2) {
        lds = (size_t)(TI + k - 1) * ka * W * sizeof(double) + (size_t)(TI + k - 1) * ka * sizeof(int);
        if (lds <= 150 * 1024) break;
    }
    if (TI < 4) return BSP_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(band_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    // channels are split over grid.y in chunks of BAND_LCH so that one launch fills the chip
    hipLaunchKernelGGL(band_kernel, dim3((nfun + TI - 1) / TI, (nl + BAND_LCH - 1) / BAND_LCH), dim3(256),
                       lds, st, TI, nfun, k, ka, nkp, kind_pot, d_bl, l0, nl, d_ptab, d_left, d_SB, d_HB);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
