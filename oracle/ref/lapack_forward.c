/*
 * lapack_forward.c -- link-time forwarders (TEST INFRASTRUCTURE, oracle/_ref build only).
 *
 * The reference links Intel MKL (src/Makefile:23, `-mkl`), which this image lacks.  The
 * image does carry a real LAPACK: scipy's bundled OpenBLAS 0.3.28 (netlib LAPACK 3.12.0),
 * whose Fortran symbols are prefixed `scipy_`.  These one-line forwarders give the
 * reference objects the unprefixed names they call; no arithmetic is implemented here.
 * (lld refuses --defsym onto shared-library symbols, hence a C object.)
 */
#include <complex.h>
typedef double _Complex zc;
#ifndef NO_DSYGV   /* Bsp_Atom_gpu.x takes dsygv_ from libbspatom_lapack.so instead (build_ref.sh) */
extern void scipy_dsygv_(int*, char*, char*, int*, double*, int*, double*, int*, double*, double*, int*, int*, long, long);
void dsygv_(int* it, char* jz, char* ul, int* n, double* a, int* lda, double* b, int* ldb, double* w, double* wk, int* lw, int* info, long l1, long l2)
{ scipy_dsygv_(it, jz, ul, n, a, lda, b, ldb, w, wk, lw, info, l1, l2); }
#endif
extern void scipy_dgemv_(char*, int*, int*, double*, double*, int*, double*, int*, double*, double*, int*, long);
void dgemv_(char* t, int* m, int* n, double* al, double* a, int* lda, double* x, int* ix, double* be, double* y, int* iy, long l)
{ scipy_dgemv_(t, m, n, al, a, lda, x, ix, be, y, iy, l); }
extern void scipy_dsymv_(char*, int*, double*, double*, int*, double*, int*, double*, double*, int*, long);
void dsymv_(char* u, int* n, double* al, double* a, int* lda, double* x, int* ix, double* be, double* y, int* iy, long l)
{ scipy_dsymv_(u, n, al, a, lda, x, ix, be, y, iy, l); }
extern double scipy_ddot_(int*, double*, int*, double*, int*);
double ddot_(int* n, double* x, int* ix, double* y, int* iy) { return scipy_ddot_(n, x, ix, y, iy); }
extern void scipy_zhemv_(char*, int*, zc*, zc*, int*, zc*, int*, zc*, zc*, int*, long);
void zhemv_(char* u, int* n, zc* al, zc* a, int* lda, zc* x, int* ix, zc* be, zc* y, int* iy, long l)
{ scipy_zhemv_(u, n, al, a, lda, x, ix, be, y, iy, l); }
extern void scipy_zgemv_(char*, int*, int*, zc*, zc*, int*, zc*, int*, zc*, zc*, int*, long);
void zgemv_(char* t, int* m, int* n, zc* al, zc* a, int* lda, zc* x, int* ix, zc* be, zc* y, int* iy, long l)
{ scipy_zgemv_(t, m, n, al, a, lda, x, ix, be, y, iy, l); }
extern zc scipy_zdotu_(int*, zc*, int*, zc*, int*);
zc zdotu_(int* n, zc* x, int* ix, zc* y, int* iy) { return scipy_zdotu_(n, x, ix, y, iy); }
extern zc scipy_zdotc_(int*, zc*, int*, zc*, int*);
zc zdotc_(int* n, zc* x, int* ix, zc* y, int* iy) { return scipy_zdotc_(n, x, ix, y, iy); }
