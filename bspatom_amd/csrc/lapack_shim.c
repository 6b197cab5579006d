/* lapack_shim.c -> libbspatom_lapack.so: the plain LAPACK name of the one routine the reference's hot path calls.
 *
 * The reference links `-mkl` and calls DSYGV once per l-channel (matrices.f90:248, src/Makefile:23).  Putting
 * -lbspatom_lapack in FRONT of the LAPACK library on that link line resolves `dsygv_` here, and from here to
 * bsp_dsygv_ in libbspatom.so (GPU); every other BLAS/LAPACK symbol the program uses still comes from the CPU
 * library.  Kept out of libbspatom.so itself so that loading the GPU library into a process that also holds a
 * real LAPACK (numpy, scipy) never shadows that library's dsygv_.
 * Hidden CHARACTER lengths are size_t (flang, gfortran >= 8, ifort on x86-64 pass them in registers either way). */
#include <stddef.h>

void bsp_dsygv_(const int *itype, const char *jobz, const char *uplo, const int *n, double *a, const int *lda,
                double *b, const int *ldb, double *w, double *work, const int *lwork, int *info, size_t jobz_len,
                size_t uplo_len);

void dsygv_(const int *itype, const char *jobz, const char *uplo, const int *n, double *a, const int *lda, double *b,
            const int *ldb, double *w, double *work, const int *lwork, int *info, size_t jobz_len, size_t uplo_len)
{
    bsp_dsygv_(itype, jobz, uplo, n, a, lda, b, ldb, w, work, lwork, info, jobz_len, uplo_len);
}
