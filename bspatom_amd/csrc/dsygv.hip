// dsygv.hip -- `bsp_dsygv_`: the LAPACK symbol boundary of the hot path.  Same Fortran-77 ABI and
// semantics as DSYGV as called at reference matrices.f90:248
//     CALL DSYGV(1,'V','U',nfun,Hij,nfun,Bij,nfun,En,WORK,LWORK,INFO)
// libbspatom_lapack.so (lapack_shim.c) exports the plain name `dsygv_` and forwards here, so the reference's own
// objects link against it in place of MKL's DSYGV with no source change (the test
// tests/test_gpu_solve.py::test_reference_binary_on_gpu_dsygv runs the reference's own program linked that way).
// The pencil at that call site is banded (half-width k-1); the band is detected on the host, the pipeline of
// capi.hip does the rest on the GPU.  JOBZ='V' eigenvectors come from batched inverse iteration on the banded pencil
// (eigvec.hip), S-orthonormalised inside clusters of close eigenvalues (below).
// info: 0 ok; -i = argument i illegal (LAPACK numbering; -5 also for a pencil that is not banded with half-width
// <= 15, which the reference never produces); n+i = leading minor i of B not positive definite; n = the GPU path
// failed (no device, out of memory, a kernel reported an error: message on stderr) -- LAPACK's "failed to converge"
// class, which the reference treats as fatal (matrices.f90:250-254).
#include <cmath>
#include <cstring>
#include <vector>
#include "common.h"

using namespace bsp;

static inline int round_up64(int x) { return (x + 63) / 64 * 64; }

template <class T> struct DBuf {
    T *p = nullptr;
    ~DBuf() { hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)); }
};

extern "C" void bsp_dsygv_(const int *itype, const char *jobz, const char *uplo, const int *n_, double *a,
                           const int *lda_, double *b, const int *ldb_, double *w, double *work,
                           const int *lwork, int *info, size_t, size_t)      // hidden CHARACTER lengths: size_t (flang, gfortran >= 8)
{
    const int n = *n_, lda = *lda_, ldb = *ldb_;
    const bool wantz = (*jobz == 'V' || *jobz == 'v');
    const bool upper = (*uplo == 'U' || *uplo == 'u');
    *info = 0;
    if (*itype != 1) *info = -1;                     // only A x = lambda B x (the reference's case)
    else if (!wantz && !(*jobz == 'N' || *jobz == 'n')) *info = -2;
    else if (!upper && !(*uplo == 'L' || *uplo == 'l')) *info = -3;
    else if (n < 0) *info = -4;
    else if (lda < (n > 1 ? n : 1)) *info = -6;
    else if (ldb < (n > 1 ? n : 1)) *info = -8;
    const int lwmin = 3 * n > 1 ? 3 * n - 1 : 1;            // DSYGV: LWORK >= max(1, 3n-1); nothing of WORK is used here
    if (!*info && *lwork != -1 && *lwork < lwmin) *info = -11;
    if (*info) return;
    if (*lwork == -1) { work[0] = (double)lwmin; return; }  // workspace query
    if (n == 0) return;
    work[0] = (double)lwmin;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "bsp_dsygv_: no HIP device (libbspatom has no CPU path)\n");
        *info = n;
        return;
    }
    // one process per GPU: the device the process's problems live on (capi.hip), else the caller's current device, which is
    // then latched like a problem's would be -- the per-process streams and scratch of the pipeline live on it
    {
        int dev = process_device();
        if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
        if (process_device_check(dev) != BSP_OK || hipSetDevice(dev) != hipSuccess) { *info = n; return; }
        process_device_latch(dev);
    }
    auto at = [&](const double *m, int ld, int i, int j) -> double {   // element (i,j), i <= j, of the stored triangle
        return upper ? m[(size_t)j * ld + i] : m[(size_t)i * ld + j];
    };
    int bw = 0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i)
            if (j - i > bw && (at(a, lda, i, j) != 0.0 || at(b, ldb, i, j) != 0.0)) bw = j - i;
    if (bw > 15) {
        fprintf(stderr, "bsp_dsygv_: pencil half-bandwidth %d > 15: dense overlap matrices are outside the "
                        "B-spline hot path this library replaces\n", bw);
        *info = -5;
        return;
    }
    const int k = (bw < 1 ? 1 : bw) + 1, npad = round_up64(n);
    std::vector<double> SB((size_t)k * n, 0.0), HB((size_t)k * n, 0.0);
    for (int d = 0; d < k; ++d)
        for (int i = 0; i + d < n; ++i) {
            SB[(size_t)d * n + i] = at(b, ldb, i, i + d);
            HB[(size_t)d * n + i] = at(a, lda, i, i + d);
        }
    DBuf<double> dSB, dHB, dUB, dr, dY, dC, dAB, dd, de, dE, dvw, dvec;
    DBuf<char> dwork, dctl;
    DBuf<int> dinfo, dchan, dstatus;
    const size_t nn = (size_t)npad * npad;
    bool ok = dSB.alloc((size_t)k * n) == hipSuccess && dHB.alloc((size_t)k * n) == hipSuccess &&
              dUB.alloc((size_t)k * n) == hipSuccess && dr.alloc(n) == hipSuccess && dY.alloc(nn) == hipSuccess &&
              dC.alloc(nn) == hipSuccess && dAB.alloc(ab_stride(npad)) == hipSuccess &&
              dd.alloc(npad) == hipSuccess && de.alloc(npad) == hipSuccess && dE.alloc(n) == hipSuccess &&
              dwork.alloc(sy2sb_work_bytes(npad, 64, 1)) == hipSuccess && dinfo.alloc(1) == hipSuccess &&
              dstatus.alloc(1) == hipSuccess && dctl.alloc(sb2st_ctl_bytes(1)) == hipSuccess;
    auto fail = [&](const char *what) {
        fprintf(stderr, "bsp_dsygv_: %s\n", what);
        *info = n;
    };
    if (!ok) { fail("out of device memory"); return; }
    hipMemcpy(dSB.p, SB.data(), SB.size() * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(dHB.p, HB.data(), HB.size() * sizeof(double), hipMemcpyHostToDevice);
    hipMemset(dinfo.p, 0, sizeof(int));
    hipMemset(dstatus.p, 0, sizeof(int));
    // a status word and a control block of its own: a ring time-out or an exchange-frame violation of the bulge
    // chasing must not end in info = 0 (round-1 advisor finding)
    PipeBufs pb{dUB.p, dr.p, dY.p, dC.p, dAB.p, dd.p, de.p, dwork.p, dinfo.p, dstatus.p, dctl.p};
    int rc = pipeline_enqueue(n, npad, k, 1, dSB.p, dHB.p, pb, dE.p, 0, nullptr);
    if (rc || hipDeviceSynchronize() != hipSuccess) { fail("the solve pipeline failed"); return; }
    int cinfo = 0, cstat = 0;
    hipMemcpy(&cinfo, dinfo.p, sizeof(int), hipMemcpyDeviceToHost);
    hipMemcpy(&cstat, dstatus.p, sizeof(int), hipMemcpyDeviceToHost);
    if (cinfo) { *info = n + cinfo; return; }
    if (cstat) { fail("a kernel of the solve pipeline reported an error"); return; }
    hipMemcpy(w, dE.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (wantz) {
        // all n eigenvectors, in chunks that bound the scratch (invit_work_doubles per vector)
        const int chunk = n < 512 ? n : 512;
        std::vector<int> chan(chunk, 0);
        if (dchan.alloc(chunk) != hipSuccess || dvw.alloc((size_t)chunk * invit_work_doubles(n, k)) != hipSuccess ||
            dvec.alloc((size_t)n * n) != hipSuccess) { fail("out of device memory (eigenvectors)"); return; }
        hipMemcpy(dchan.p, chan.data(), chunk * sizeof(int), hipMemcpyHostToDevice);
        for (int done = 0; done < n && !rc; done += chunk) {
            const int m = (n - done < chunk) ? n - done : chunk;
            rc = launch_inverse_iteration(n, k, m, dSB.p, dHB.p, dchan.p, dE.p + done, dvw.p, dvec.p + (size_t)done * n, dinfo.p, 0);
        }
        if (rc || hipDeviceSynchronize() != hipSuccess) { fail("the inverse iteration failed"); return; }
        hipMemcpy(&cinfo, dinfo.p, sizeof(int), hipMemcpyDeviceToHost);
        if (cinfo) { fail("an inverse iteration broke down"); *info = cinfo <= n ? cinfo : n; return; }
        std::vector<double> V((size_t)n * n);
        hipMemcpy(V.data(), dvec.p, V.size() * sizeof(double), hipMemcpyDeviceToHost);
        // DSYGV's contract is Z^T B Z = I for ALL n vectors.  Independent inverse iterations give that to ~eps/gap;
        // inside a cluster of eigenvalues closer than 1e-5 |lambda|_max the vectors are re-orthogonalised in the
        // B inner product (modified Gram-Schmidt against the earlier members, as LAPACK's DSTEIN does within its
        // clusters) -- host arithmetic on the band, O(n k) per pair.
        double lmax = 0.0;
        for (int i = 0; i < n; ++i) lmax = std::fmax(lmax, std::fabs(w[i]));
        const double ctol = 1e-5 * lmax;
        std::vector<double> Sv(n);
        auto band_mv = [&](const double *x, double *y) {           // y = B x from the upper band
            for (int i = 0; i < n; ++i) y[i] = SB[i] * x[i];
            for (int d = 1; d < k; ++d)
                for (int i = 0; i + d < n; ++i) {
                    const double s = SB[(size_t)d * n + i];
                    y[i] += s * x[i + d]; y[i + d] += s * x[i];
                }
        };
        int c0 = 0;                                                // first member of the current cluster
        for (int j = 1; j < n; ++j) {
            if (w[j] - w[j - 1] > ctol) { c0 = j; continue; }
            double *vj = V.data() + (size_t)j * n;
            for (int pass = 0; pass < 2; ++pass)
                for (int i = c0; i < j; ++i) {
                    const double *vi = V.data() + (size_t)i * n;
                    band_mv(vi, Sv.data());
                    double dot = 0.0;
                    for (int t = 0; t < n; ++t) dot += vj[t] * Sv[t];
                    for (int t = 0; t < n; ++t) vj[t] -= dot * vi[t];
                }
            band_mv(vj, Sv.data());
            double nr = 0.0;
            for (int t = 0; t < n; ++t) nr += vj[t] * Sv[t];
            nr = 1.0 / std::sqrt(nr);
            for (int t = 0; t < n; ++t) vj[t] *= nr;
        }
        for (int j = 0; j < n; ++j) memcpy(a + (size_t)j * lda, V.data() + (size_t)j * n, n * sizeof(double));
    }
    // B <- Cholesky factor in the referenced triangle (U for 'U', L = U^T for 'L')
    std::vector<double> UB((size_t)k * n);
    hipMemcpy(UB.data(), dUB.p, UB.size() * sizeof(double), hipMemcpyDeviceToHost);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i <= j; ++i) {
            const double v = (j - i < k) ? UB[(size_t)(j - i) * n + i] : 0.0;
            if (upper) b[(size_t)j * ldb + i] = v; else b[(size_t)i * ldb + j] = v;
        }
}
