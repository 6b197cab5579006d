// eigvec.hip -- selected eigenvectors by inverse iteration on the BANDED pencil (H_l - E S), and
// the wave-function tabulation of WRITE_WF.
//
// The reference asks DSYGV for all eigenvectors ('V', matrices.f90:248) but in KIND_PI=0 mode
// consumes exactly one column: Hij(:, n0_ini) of channel l_ini (matrices.f90:267 -> WRITE_WF,
// Bsp_Atom.f90:101-152).  Given an eigenvalue E from the dense path, the eigenvector of the
// original pencil is recovered in O(n k^2) by inverse iteration with a banded LU (partial
// pivoting, LINPACK dgbfa-style row window), so no back-transformation through the two
// tridiagonalisation stages is needed.  One wavefront per requested vector.
//
// Output normalisation: c^T S c = 1 (DSYGV ITYPE=1), sign chosen so that the first significant
// coefficient is positive (LAPACK's sign is arbitrary; CHKPHS is commented out, matrices.f90:382).
#include "common.h"

namespace bsp {

constexpr int EB_MAX = 15;                 // half-bandwidth limit (k <= 16)
constexpr int EWC = 2 * EB_MAX + 2;        // window column slots

__device__ __forceinline__ double pencil_entry(const double *__restrict__ SB, const double *__restrict__ HB,
                                               int n, int b, double E, int r, int c)
{
    const int d = (r > c) ? (r - c) : (c - r);
    if (d > b || r < 0 || c < 0 || r >= n || c >= n) return 0.0;
    const int lo = (r > c) ? c : r;
    return HB[(size_t)d * n + lo] - E * SB[(size_t)d * n + lo];
}

// work layout per vector: U[n][2b+1], L[n][b], ipiv[n] (as doubles), tmp[n]
__global__ __launch_bounds__(64) void invit_kernel(int n, int k, const double *__restrict__ SB,
                                                  const double *__restrict__ HBall,
                                                  const int *__restrict__ chan, const double *__restrict__ Eall,
                                                  double *workall, double *vecall, int *info)
{
    extern __shared__ double y[];                       // n doubles
    __shared__ double Wd[EB_MAX + 1][EWC];
    __shared__ double xs[64];
    const int lane = threadIdx.x;
    const int b = k - 1, WC = 2 * b + 2, NR = b + 1;
    const size_t iv = blockIdx.x;
    const double E = Eall[iv];
    const double *HB = HBall + (size_t)chan[iv] * k * n;
    double *work = workall + iv * ((size_t)n * (3 * b + 3));
    double *U = work, *Lm = work + (size_t)n * (2 * b + 1), *piv = Lm + (size_t)n * b, *tmp = piv + n;
    double *vec = vecall + iv * (size_t)n;

    // scale for the zero-pivot perturbation: ~ eps * max|M_jj|
    double dmax = 0.0;
    for (int j = lane; j < n; j += 64) dmax = fmax(dmax, fabs(HB[j] - E * SB[j]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off));
    const double pertol = 2.220446049250313e-16 * fmax(dmax, 1e-300);

    // ---- banded LU with partial pivoting ----
    for (int idx = lane; idx < NR * WC; idx += 64) {
        const int r = idx / WC, cs = idx % WC;          // rows 0..b, columns 0..2b+1 at start
        Wd[r % NR][cs] = pencil_entry(SB, HB, n, b, E, r, cs);
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const int nrow = (n - 1 - j < b) ? (n - 1 - j) : b;       // rows below the pivot row
        // pivot search over rows j .. j+nrow in column j
        double av = -1.0; int ar = j;
        if (lane <= nrow) av = fabs(Wd[(j + lane) % NR][j % WC]), ar = j + lane;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(av, off);
            const int orr = __shfl_xor(ar, off);
            if (ov > av || (ov == av && orr < ar)) { av = ov; ar = orr; }
        }
        const int p = ar;
        if (p != j) {
            for (int cs = lane; cs < WC; cs += 64) {
                const double t1 = Wd[j % NR][cs];
                Wd[j % NR][cs] = Wd[p % NR][cs];
                Wd[p % NR][cs] = t1;
            }
        }
        __syncthreads();
        double pv = Wd[j % NR][j % WC];
        if (fabs(pv) < pertol) pv = (pv < 0.0) ? -pertol : pertol;
        double lt = 0.0;
        if (lane >= 1 && lane <= nrow) lt = Wd[(j + lane) % NR][j % WC] / pv;
        __syncthreads();
        if (lane == 0) { Wd[j % NR][j % WC] = pv; piv[j] = (double)p; }
        if (lane >= 1 && lane <= nrow) Wd[(j + lane) % NR][j % WC] = 0.0;
        if (lane >= 1 && lane <= b) Lm[(size_t)j * b + lane - 1] = lt;
        xs[lane] = lt;
        __syncthreads();
        // elimination: rows t = 1..nrow, columns cc = 1..2b
        for (int idx = lane; idx < nrow * 2 * b; idx += 64) {
            const int t = 1 + idx / (2 * b), cc = 1 + idx % (2 * b);
            const int cs = (j + cc) % WC;
            Wd[(j + t) % NR][cs] -= xs[t] * Wd[j % NR][cs];
        }
        // store U row j
        for (int cc = lane; cc <= 2 * b; cc += 64) U[(size_t)j * (2 * b + 1) + cc] = Wd[j % NR][(j + cc) % WC];
        __syncthreads();
        // slide the window: row j leaves, row j+b+1 enters (columns j+1 .. j+2b+1; slot of j cleared)
        const int rn = j + b + 1;
        for (int cs = lane; cs < WC; cs += 64) {
            // absolute column that maps to slot cs within [j+1, j+2b+2)
            int c = (j + 1) + ((cs - (j + 1) % WC) % WC + WC) % WC;
            Wd[j % NR][cs] = (rn < n) ? pencil_entry(SB, HB, n, b, E, rn, c) : 0.0;
        }
        __syncthreads();
    }

    // ---- inverse iteration: 3 solves ----
    for (int j = lane; j < n; j += 64) y[j] = 1.0;
    __syncthreads();
    for (int iter = 0; iter < 3; ++iter) {
        // forward: y <- L^-1 P y
        for (int j = 0; j < n; ++j) {
            const int p = (int)piv[j];
            double yj;
            if (p != j) {
                if (lane == 0) { const double t1 = y[j]; y[j] = y[p]; y[p] = t1; }
                __syncthreads();
            }
            yj = y[j];
            const int nrow = (n - 1 - j < b) ? (n - 1 - j) : b;
            if (lane >= 1 && lane <= nrow) y[j + lane] -= Lm[(size_t)j * b + lane - 1] * yj;
            __syncthreads();
        }
        // backward: x_j = (y_j - sum_{cc=1..2b} U[j][cc] x_{j+cc}) / U[j][0]
        for (int j = n - 1; j >= 0; --j) {
            double s = 0.0;
            if (lane >= 1 && lane <= 2 * b && j + lane < n) s = U[(size_t)j * (2 * b + 1) + lane] * y[j + lane];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) y[j] = (y[j] - s) / U[(size_t)j * (2 * b + 1)];
            __syncthreads();
        }
        // normalise by max-abs, then rhs = S x for the next iteration
        double mx = 0.0;
        for (int j = lane; j < n; j += 64) mx = fmax(mx, fabs(y[j]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        const double sc = (mx > 0.0) ? 1.0 / mx : 1.0;
        for (int j = lane; j < n; j += 64) { y[j] *= sc; tmp[j] = y[j]; }
        __syncthreads();
        if (iter < 2) {
            for (int j = lane; j < n; j += 64) {
                double s = 0.0;
                for (int d = -b; d <= b; ++d) {
                    const int c = j + d;
                    if (c >= 0 && c < n) {
                        const int ad = d < 0 ? -d : d, lo = d < 0 ? c : j;
                        s += SB[(size_t)ad * n + lo] * tmp[c];
                    }
                }
                y[j] = s;
            }
            __syncthreads();
        }
    }
    // ---- S-normalise: c^T S c = 1, sign: first significant coefficient positive ----
    double q = 0.0;
    for (int j = lane; j < n; j += 64) {
        double s = 0.0;
        for (int d = -b; d <= b; ++d) {
            const int c = j + d;
            if (c >= 0 && c < n) {
                const int ad = d < 0 ? -d : d, lo = d < 0 ? c : j;
                s += SB[(size_t)ad * n + lo] * y[c];
            }
        }
        q += y[j] * s;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    int first = n;
    for (int j = lane; j < n; j += 64)
        if (fabs(y[j]) > 1e-8 && j < first) first = j;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(first, off); first = o < first ? o : first; }
    double sgn = 1.0;
    if (first < n && y[first] < 0.0) sgn = -1.0;
    if (!(q > 0.0)) { if (lane == 0) atomicExch(info, (int)iv + 1); q = 1.0; }
    const double nrm = sgn / sqrt(q);
    for (int j = lane; j < n; j += 64) vec[j] = y[j] * nrm;
}

size_t invit_work_doubles(int n, int k) { return (size_t)n * (3 * (k - 1) + 3); }

int launch_inverse_iteration(int n, int k, int nvec, const double *d_SB, const double *d_HB, const int *d_chan,
                             const double *d_E, double *d_work, double *d_vec, int *d_info, hipStream_t st)
{
    if (k - 1 > EB_MAX || k < 2) return BSP_ERR_ARG;
    const size_t lds = (size_t)n * sizeof(double);
    if (lds > 140 * 1024) return BSP_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(invit_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(invit_kernel, dim3(nvec), dim3(64), lds, st, n, k, d_SB, d_HB, d_chan, d_E, d_work, d_vec, d_info);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// ---- WRITE_WF (Bsp_Atom.f90:118-146): one thread per tabulation point ------------------------
__global__ void wf_kernel(int nkp, int k, int n, const double *__restrict__ rt0, const double *__restrict__ c,
                          double ra, double rb, int npts, double *rout, double *uout, int *status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > npts) return;
    const double *t = rt0 - 1;
    const double dr = (rb - ra) / (double)npts;
    const double r = ra + (double)i * dr;
    // interv.f90:86-117
    int left;
    if (r > t[nkp] || r < t[1]) left = 1;
    else if (r == t[nkp]) { left = nkp; while (left > 1 && !(t[left] < t[nkp])) --left; }
    else {
        int lo = 1, hi = nkp;                      // t[lo] <= r < t[hi]; largest ilo with t[ilo] <= r
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t[mid] <= r) lo = mid; else hi = mid; }
        left = lo;
    }
    // bsplvb.f90:24-50 (order k, index 1)
    double biatx[17], dl[17], dR[17];
    for (int j = 0; j <= k; ++j) biatx[j] = 0.0;
    biatx[1] = 1.0;
    if (k > 1) {
        if (t[left + 1] <= t[left]) { atomicExch(status, BSP_ERR_BSPLVB); return; }   // FATAL ERROR - BSPLVB
        for (int j = 1; j < k; ++j) {
            dR[j] = t[left + j] - r;
            dl[j] = r - t[left + 1 - j];
            double saved = 0.0;
            for (int q = 1; q <= j; ++q) {
                const double term = biatx[q] / (dR[q] + dl[j + 1 - q]);
                biatx[q] = saved + dR[q] * term;
                saved = dl[j + 1 - q] * term;
            }
            biatx[j + 1] = saved;
        }
    }
    double sumf = 0.0;
    for (int jf = 1; jf <= k; ++jf) {
        const int j = jf + (left - k);
        double fr = 0.0;
        if (j >= 1 && j <= n) fr = c[j - 1];
        sumf = sumf + fr * biatx[jf];
    }
    rout[i] = r;
    uout[i] = sumf;
}

int launch_wf_tabulate(int nkp, int k, int n, const double *d_rt, const double *d_c, double ra, double rb,
                       int npts, double *d_r, double *d_u, int *d_status, hipStream_t st)
{
    if (k > 16) return BSP_ERR_ARG;
    hipLaunchKernelGGL(wf_kernel, dim3((npts + 1 + 127) / 128), dim3(128), 0, st, nkp, k, n, d_rt, d_c, ra, rb,
                       npts, d_r, d_u, d_status);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
