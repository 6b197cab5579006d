// bandsect.hip -- ONE eigenvalue of the banded pencil (H_l, S) itself, before anything is reduced.
//
// The reference consumes one eigenvector, Hij(:, n0_ini) of channel l_ini (matrices.f90:267).  The inverse iteration that delivers it
// (eigvec.hip) needs that eigenvalue and nothing else, and took it from the tridiagonal matrix at the END of the pipeline: one
// workgroup's multisection alone on the GPU (1.1 ms), then 14 ms of ONE wave beside the batched bisection -- since the bisection
// shares its points (12.8 ms) the step waited 2.8 ms for the vector.  Here the eigenvalue comes from the pencil as assembled:
//
//   the number of eigenvalues of (H, S) below x  =  the number of negative pivots of  H - x S = L D L^T          (Sylvester; S > 0)
//
// with L D L^T taken WITHOUT pivoting along the band (half-width b <= 8: a window of (b + 1)(b + 2) / 2 entries per shift in
// registers, one new row of the band per step).  256 shifts per round -- one per thread, all walking the same rows, which a
// workgroup stages through LDS 64 rows at a time -- so a bracket shrinks 257-fold per round: a first round on the powers of two of both
// signs, seven more to 2 eps |x|.  8.3 ms of one workgroup, right after the assembly, with 60 ms of reductions ahead of it; the
// inverse iteration follows in the same workgroup (eigvec.hip::early_vector_kernel -- the device code is in bandsect.h for that;
// the kernel of this file serves the stage entry bspatom_stage_band_eigenvalue): the vector is there long before the spectra are.
//
// Without pivoting the factorisation can grow (|L| up to 1e6 on C4's pencil at unlucky shifts) and a count can then be off next to an
// eigenvalue; the search needs no monotonicity (lo = the point before the FIRST point whose count exceeds m), its result is an
// eigenvalue of the pencil to ~1e-13 of the spectrum's width in practice, and the caller does not rest on that: once the spectra are
// there it checks the value against eigenvalue m of the tridiagonal matrix and its neighbours, and otherwise forgets the early
// vector (capi.hip::solve_impl; bspatom_eigvec then computes it from the spectra as for any other vector).
#include "bandsect.h"

namespace bsp {

__global__ __launch_bounds__(BS_T) void band_multisect_kernel(int n, int k, const double *__restrict__ SB, const double *__restrict__ HB,
                                                             int m, double *out)
{
    __shared__ BandSectLds L;
    const double lam = band_multisect(n, k, SB, HB, m, L);
    if (threadIdx.x == 0) *out = lam;
}

// eigenvalue m (0-based, ascending) of the pencil (HB, SB) (upper bands, [d][i] = A(i, i + d), k rows of n) -> *d_out
int launch_band_multisect(int n, int k, const double *d_SB, const double *d_HB, int m, double *d_out, hipStream_t st)
{
    if (k < 2 || k - 1 > BS_B || m < 0 || m >= n) return BSP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(band_multisect_kernel, dim3(1), dim3(BS_T), 0, st, n, k, d_SB, d_HB, m, d_out);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
