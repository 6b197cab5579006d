// dsygv.hip -- `bsp_dsygv_`: the LAPACK symbol boundary of the hot path.  Same Fortran-77 ABI and
// semantics as DSYGV as called at reference matrices.f90:248
//     CALL DSYGV(1,'V','U',nfun,Hij,nfun,Bij,nfun,En,WORK,LWORK,INFO)
// so the reference can link against libbspatom in place of MKL with a one-token rename (or an
// `-Wl,--defsym,dsygv_=bsp_dsygv_`).  The pencil at that call site is banded (half-width k-1);
// the band is detected on the host, the dense/banded pipeline of capi.hip does the rest on the GPU.
// JOBZ='V' eigenvectors come from batched inverse iteration on the banded pencil (eigvec.hip).
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
                           const int *lwork, int *info, int, int)
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
    if (*info) return;
    if (*lwork == -1) { work[0] = (double)(3 * n > 1 ? 3 * n - 1 : 1); return; }   // workspace query
    if (n == 0) return;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "bsp_dsygv_: no HIP device (libbspatom has no CPU path)\n");
        *info = -99;
        return;
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
    DBuf<char> dwork;
    DBuf<int> dinfo, dchan;
    const size_t nn = (size_t)npad * npad;
    bool ok = dSB.alloc((size_t)k * n) == hipSuccess && dHB.alloc((size_t)k * n) == hipSuccess &&
              dUB.alloc((size_t)k * n) == hipSuccess && dr.alloc(n) == hipSuccess && dY.alloc(nn) == hipSuccess &&
              dC.alloc(nn) == hipSuccess && dAB.alloc(ab_stride(npad)) == hipSuccess &&
              dd.alloc(npad) == hipSuccess && de.alloc(npad) == hipSuccess && dE.alloc(n) == hipSuccess &&
              dwork.alloc(sy2sb_work_bytes(npad, 64, 1)) == hipSuccess && dinfo.alloc(1) == hipSuccess;
    if (!ok) { *info = -98; return; }
    hipMemcpy(dSB.p, SB.data(), SB.size() * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(dHB.p, HB.data(), HB.size() * sizeof(double), hipMemcpyHostToDevice);
    hipMemset(dinfo.p, 0, sizeof(int));
    PipeBufs pb{dUB.p, dr.p, dY.p, dC.p, dAB.p, dd.p, de.p, dwork.p, dinfo.p};
    int rc = pipeline_enqueue(n, npad, k, 1, dSB.p, dHB.p, pb, dE.p, 0, nullptr);
    if (rc || hipDeviceSynchronize() != hipSuccess) { *info = -97; return; }
    int cinfo = 0;
    hipMemcpy(&cinfo, dinfo.p, sizeof(int), hipMemcpyDeviceToHost);
    if (cinfo) { *info = n + cinfo; return; }
    hipMemcpy(w, dE.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (wantz) {
        std::vector<int> chan(n, 0);
        if (dchan.alloc(n) != hipSuccess || dvw.alloc((size_t)n * invit_work_doubles(n, k)) != hipSuccess ||
            dvec.alloc((size_t)n * n) != hipSuccess) { *info = -98; return; }
        hipMemcpy(dchan.p, chan.data(), n * sizeof(int), hipMemcpyHostToDevice);
        rc = launch_inverse_iteration(n, k, n, dSB.p, dHB.p, dchan.p, dE.p, dvw.p, dvec.p, dinfo.p, 0);
        if (rc || hipDeviceSynchronize() != hipSuccess) { *info = -97; return; }
        std::vector<double> V((size_t)n * n);
        hipMemcpy(V.data(), dvec.p, V.size() * sizeof(double), hipMemcpyDeviceToHost);
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
