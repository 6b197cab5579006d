/*
 * bspatom.h -- C ABI of libbspatom: MI355X (gfx950) drop-in for the hot path of
 * carlosmwh1985/BspAtom: MATRIX_SVT + SOLVE_SYSTEM (reference src/matrices.f90:1-394), i.e. the
 * Gauss-Legendre assembly of the banded S, H(l) and the LAPACK DSYGV generalized eigen-solve.
 *
 * The reference has no plugin API: MATRIX_SVT / SOLVE_SYSTEM take no arguments and talk through
 * Fortran module globals (src/Modules.f90:21-203).  This header is therefore the explicit form
 * of that implicit interface; each entry point names the reference code it replaces.
 * All pointers are plain host pointers unless the name says `_dev`; the caller owns every
 * buffer it passes; the library owns the device memory of a problem between create/destroy.
 * All calls are blocking and must come from one host thread per problem.
 * Return value: 0 = ok, < 0 = BSPATOM_ERR_*, per-channel LAPACK-style codes go to `info[]`.
 */
#ifndef BSPATOM_H
#define BSPATOM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BSPATOM_OK 0
#define BSPATOM_ERR_HIP (-1)         /* HIP runtime error (text on stderr) */
#define BSPATOM_ERR_ARG (-2)         /* invalid argument */
#define BSPATOM_ERR_BSPLVB (-3)      /* 'FATAL ERROR - BSPLVB' STOP of bsplvb.f90:30-34 */
#define BSPATOM_ERR_NOGPU (-4)       /* no gfx950 device: there is no CPU fallback */
#define BSPATOM_ERR_UNSUPPORTED (-5)  /* a size outside this build's limits (below), or a switch combination that has no kernel */

/* Size limits of this build (bspatom_problem_create / bspatom_host_setup return BSPATOM_ERR_UNSUPPORTED beyond them, with the
 * limit on stderr).  The reference allocates everything by nfun / nkp (matrices.f90:20,222-225; bsplvb.f90:22) and has no
 * such limits, except that its Enl.dat record is I4 (matrices.f90:391): nfun <= 9999 is all it can write.
 *   BSPATOM_MAX_NFUN  functions per channel: 10048 (npad; every nfun the reference's output format allows);
 *   BSPATOM_MAX_K     B-spline order k (device tables of the assembly, band half-width of the Cholesky / inverse iteration);
 *   BSPATOM_MAX_KA    Gauss-Legendre points per interval (ka = k + 3 by default: 19 at k = 16). */
#define BSPATOM_MAX_NFUN 10048
#define BSPATOM_MAX_K 16
#define BSPATOM_MAX_KA 32

/* Namelist values of VARS_BSP / VARS_TISE (src/ReadInputs.f90:15-17) with the reference's
 * defaults (:27-36, :75-84) as the zero-initialised-then-`bspatom_input_defaults` state. */
typedef struct bspatom_input {
    int32_t kind_grid, k, ka, nfun, kind_bc1, kind_bc2;
    double ra, rb, rmax;
    int32_t n0_ini, l_ini, m_ini, l_fin, lmax, kind_pot;
    double emax_fin, zatom;
} bspatom_input;

/* Sizes derived exactly as READ_INPUTS does (src/ReadInputs.f90:39-69, :87). */
typedef struct bspatom_sizes {
    int32_t nfun, k, ka, nkp, nointv, nbc1, nbc2, lmax, nintv_exp, nintv_lin, npad;
} bspatom_sizes;

typedef struct bspatom_problem bspatom_problem;   /* opaque */

/* ---- set-up: READ_INPUTS sizes + GRID + gauleg + SELPOT tables (host), upload ------------- */
void bspatom_input_defaults(bspatom_input *in);                       /* ReadInputs.f90:27-36,75-84 */
int bspatom_device_count(void);
/* Host-only part of the set-up (no GPU needed): sizes as READ_INPUTS derives them; rt[nkp],
 * aind[2*nfun], xg[ka], wg[ka] as GRID/gauleg build them (call once with NULL arrays for the sizes). */
int bspatom_host_setup(const bspatom_input *in, bspatom_sizes *s, double *rt, double *aind, double *xg,
                       double *wg);
/* Creates the problem on HIP device `device`: derives sizes (ReadInputs.f90:39-141), builds the knot
 * sequence and Aind (grid.f90:14-91), the Gauss-Legendre rule (Modules.f90:112-153) and the
 * potential table (Modules.f90:263-295) on the host, uploads them.  One process per GPU: every problem of a
 * process must name the device of the first one (BSPATOM_ERR_UNSUPPORTED otherwise). */
int bspatom_problem_create(const bspatom_input *in, int device, bspatom_problem **out);
void bspatom_problem_destroy(bspatom_problem *p);
int bspatom_problem_sizes(const bspatom_problem *p, bspatom_sizes *s);
/* The route bspatom_solve takes for this problem under the current switches (BSP_ROUTE / option "route"): 2 = band route
 * (csrc/crawford.hip: the pencil stays banded; k - 1 <= 8), 1 = dense route (standard form, two-stage tridiagonalisation). */
int bspatom_problem_route(const bspatom_problem *p);
/* Host copies of rt[nkp], aind[2*nfun] (column-major Aind(nfun,2)), xg[ka], wg[ka]; any may be NULL. */
int bspatom_problem_grid(const bspatom_problem *p, double *rt, double *aind, double *xg, double *wg);

/* ---- the hot path --------------------------------------------------------------------------- */
/* MATRIX_SVT (matrices.f90:68-186) + `Hij = Tij + Uij(:,:,l) + Vij` (:244) for channels
 * l0 .. l0+nl-1, upper bands: SB[d*nfun + i] = S(i,i+d); HB[(l*k + d)*nfun + i] = H_l(i,i+d),
 * 0-based, d < k.  Bit-identical to the reference's dense matrices on the band.  SB/HB may be NULL
 * (results stay on the device for bspatom_solve). */
int bspatom_assemble(bspatom_problem *p, int l0, int nl, double *SB, double *HB);

/* SURVEY 8(f).2 -- the dipole matrices that MATRIX_SVT accumulates in the same quadrature loop and keeps in
 * rij for KIND_PI = 1, 2 (matrices.f90:141-144, 159-163): c = 0: int B_i r B_j dr (rij(:,:,1), length gauge),
 * c = 1: int B_i (1/r) B_j dr and c = 2: int B_i B_j' dr (rij(:,:,1), rij(:,:,2), velocity gauge).  The
 * reference fills both triangles and they are not bit-symmetric, so the FULL band is returned:
 * RB[(c*(2k-1) + (d+k-1))*nfun + i] = X_c(i, i+d), 0-based i, d = -(k-1)..k-1.  Bit-identical to the
 * reference's rij on the band (which is all of it).  RB: 3*(2k-1)*nfun doubles. */
int bspatom_dipole_bands(bspatom_problem *p, double *RB);

/* SOLVE_SYSTEM's l-loop (matrices.f90:242-265): assembly + DSYGV eigenvalues for channels
 * l0 .. l0+nl-1.  E[l*nfun + i] ascending per channel (column-major Enl(nfun,0:lmax), :230).
 * info[l]: 0 ok; nfun+i: leading minor i of S not positive definite (DSYGV convention, :250-254). */
int bspatom_solve(bspatom_problem *p, int l0, int nl, double *E, int32_t *info);
/* Same, spectra left in device memory (E_dev: nl*nfun doubles on the problem's device, e.g. a
 * torch tensor's data_ptr) so that ranks can exchange them with RCCL without a host round trip. */
int bspatom_solve_dev(bspatom_problem *p, int l0, int nl, double *E_dev, int32_t *info);

/* Eigenvector column `n0` (1-based, as n0_ini) of channel l -- the only column of DSYGV's 'V'
 * output that KIND_PI=0 consumes (matrices.f90:267): inverse iteration on the banded pencil
 * (H_l - E S), normalised c^T S c = 1; sign arbitrary (CHKPHS is commented out, :382).
 * Requires a previous bspatom_solve covering channel l.  c[nfun]. */
int bspatom_eigvec(bspatom_problem *p, int l, int n0, double *c);
/* The eigenvectors n0 .. n0+count-1 (1-based) of channel l, i.e. columns n0.. of DSYGV's 'V' output
 * Hij(:, n0:n0+count-1) at matrices.f90:248 -- what the KIND_PI >= 3 branch keeps as ctemp(:,1:ntemp,l)
 * (matrices.f90:331) and writes to Eigenvec_All.dat (:366-378, FORMAT 300 `I5,5000G20.10`): batched inverse
 * iteration on the banded pencil, each column normalised c^T S c = 1, sign arbitrary (as LAPACK's).
 * Z[j*nfun + i] = component i of eigenvector n0+j.  Requires a previous bspatom_solve covering l. */
int bspatom_eigvecs(bspatom_problem *p, int l, int n0, int count, double *Z);

/* Dipole matrix elements between eigenvectors of the last solved batch: the DGEMV + DDOT of TRANS_AMP for the
 * plane-wave branches KIND_PI = 1, 2 (reference PhotoIon.f90:95-107):
 *   D[i] = c(l_fin, n0_fin + i)^T (a[0] R_r + a[1] R_{1/r} + a[2] R_{d/dr}) c(l_ini, n0_ini),  i = 0 .. count-1,
 * with R_r = int B_i r B_j, R_{1/r} = int B_i B_j / r, R_{d/dr} = int B_i B_j' (the rij of MATRIX_SVT,
 * matrices.f90:141-144,159-163) and S-normalised eigenvectors whose signs are this library's (first significant
 * coefficient positive).  The reference's T_fi(n) = An c0 D: the angular factors c0, a[] (THREE_J) and the density of
 * states An are host arithmetic (bspatom_amd/host.py::trans_amp).  n0_* are 1-based. */
int bspatom_dipole_elements(bspatom_problem *p, int l_ini, int n0_ini, int l_fin, int n0_fin, int count,
                            const double a[3], double *D);

/* WRITE_WF (Bsp_Atom.f90:118-146): u(r_i) = sum_j c_j B_j(r_i), r_i = ra + i*(rb-ra)/npts,
 * i = 0..npts.  Returns BSPATOM_ERR_BSPLVB where the reference STOPs. r[npts+1], u[npts+1]. */
int bspatom_write_wf(bspatom_problem *p, const double *c, int npts, double *r, double *u);

/* The eigenvector the reference consumes (l_ini, n0_ini; matrices.f90:267) is computed during bspatom_solve when its channel is in
 * the batch.  On the band route its eigenvalue comes from the pencil's inertia right after the assembly (csrc/bandsect.hip), and the
 * solve checks it against the spectra when they are there.  state of the last solve: 0 = no early vector (other route, channel not in
 * the batch, BSP_VEC_EARLY=0), 1 = early vector kept, -1 = check failed, vector dropped (bspatom_eigvec computes it on demand). */
int bspatom_early_vector_state(const bspatom_problem *p, int32_t *state);
/* Per-stage device time of the last bspatom_solve* call, HIP events on the library's stream
 * (milliseconds): [0] point table + bands, [1] Cholesky + standard form, [2] sy2sb,
 * [3] sb2st, [4] bisection, [5] total.  Also the number of launches of the sy2sb GEMM. */
int bspatom_last_timing(const bspatom_problem *p, double ms[6]);

/* Per-kernel launch durations (measurement only; no reference counterpart).  With bspatom_set_option("ktime", 1) every launch
 * of the kernels below is bracketed by two HIP events on its own stream; this call waits for the device, sums the elapsed
 * times and launch counts per slot since the previous call into ms[] / launches[] (cap >= the slot count, which it returns)
 * and forgets them.  Slots: 0 rank-128 update (syr2k), 1 symm, 2 panel QR, 3 the small products of the panel chain,
 * 4 sb2sb_mfma_kernel, 5 sbr_rows_kernel<8> / <16> (sb16st_kernel with BSP_SB16_ROWS=0), 6 batched bisection, 7 Cholesky + standard form,
 * 8 the band route's reduction (crawford.hip); bspatom_kernel_slot_name(i)
 * names them.  Launches on different streams overlap: the sums are sums of launch durations, not wall time. */
int bspatom_kernel_times(double *ms, int32_t *launches, int cap);
const char *bspatom_kernel_slot_name(int slot);

/* ---- the one exchange of the sharded path (csrc/comm.hip) ----------------------------------- */
/* One process per GPU, the l-loop of matrices.f90:242-248 sharded: the ranks exchange nothing while they solve; at the end their
 * result records are gathered.  These calls give a host without Python (bsp_atom_host.x) the RCCL all-gather that
 * bspatom_amd/parallel.py issues through torch.distributed.  No reference counterpart (the reference is one process).
 *   bspatom_run_token   an id shared by the processes of ONE launch and by no other launch: "<pid of the launcher>.<its start
 *                       time>[.<TORCHELASTIC_RUN_ID>]" (buf: >= 128 bytes).  Also names the files of the no-RCCL fallback.
 *   bspatom_comm_create collective over `world` processes (rank 0 .. world-1), each on its own GPU (the device of the process's
 *                       problems).  RCCL is loaded here (dlopen), its unique id travels through `dir`/ncclid.<token>.
 *                       BSPATOM_ERR_UNSUPPORTED (before anything is exchanged, the same on every rank): more ranks than GPUs
 *                       (ranks share a device) or no librccl -- the caller falls back to its file exchange.
 *   bspatom_comm_allgather  recv[r*count .. (r+1)*count) = rank r's send[0 .. count) on every rank (host buffers).
 *   bspatom_comm_collectives  number of all-gathers issued on this communicator. */
typedef struct bspatom_comm bspatom_comm;
int bspatom_run_token(char *buf, int cap);
int bspatom_comm_create(int rank, int world, const char *dir, bspatom_comm **out);
int bspatom_comm_allgather(bspatom_comm *c, const double *send, double *recv, long count);
int bspatom_comm_collectives(const bspatom_comm *c);
void bspatom_comm_destroy(bspatom_comm *c);

/* ---- LAPACK symbol boundary ---------------------------------------------------- */
/* Fortran-77 ABI of DSYGV as called at matrices.f90:248.  ITYPE=1, UPLO='U' or 'L'; A and B must
 * be banded with half-width <= 15 (they are, at the reference's call site); JOBZ='N' returns the
 * eigenvalues, JOBZ='V' additionally returns all n B-orthonormal eigenvectors (inverse iteration,
 * re-orthogonalised inside clusters).  LWORK >= max(1, 3n-1) or -1 (query) as for DSYGV.
 * info = -k for a bad k-th argument (LAPACK numbering), n+i if B is not positive definite, n if the GPU
 * path failed (message on stderr).  Trailing hidden CHARACTER lengths: size_t, as flang / gfortran pass them.
 * libbspatom_lapack.so exports the same routine under the plain name `dsygv_` (csrc/lapack_shim.c). */
void bsp_dsygv_(const int *itype, const char *jobz, const char *uplo, const int *n, double *a,
                const int *lda, double *b, const int *ldb, double *w, double *work, const int *lwork,
                int *info, size_t jobz_len, size_t uplo_len);

/* bsp_dsygv_ keeps its device buffers in a process-wide pool between calls (the reference calls DSYGV once per l in a loop); at
 * most 4 GiB stay parked after a call.  This returns every idle buffer of that pool to the driver. */
void bspatom_release_scratch(void);

/* ---- run-time switches (tests, A/B comparisons) ----------------------------------------------- */
/* The BSP_* environment variables of DESIGN.md 4.4 are read once per process; these two calls read and
 * change the same switches afterwards, by lower-case name without the prefix ("sb2st_ring",
 * "sb2st_version", "sb2st_force_abort", "panel_qr", "bisect" ...).  Unknown name: BSPATOM_ERR_ARG.
 * No reference counterpart (the reference has no switches on this path). */
int bspatom_set_option(const char *name, int value);
int bspatom_get_option(const char *name, int *value);

/* ---- stage-level entry points (parity tests, profiling; host buffers, column-major) ---------- */
/* C[b] = alpha * op(A[b]) op(B[b]) + beta * C[b], element strides as in csrc/common.h GemmDesc. */
int bspatom_stage_gemm(int M, int N, int K, int batch, const double *A, long sAm, long sAk, long bA,
                       long lenA, const double *B, long sBk, long sBn, long bB, long lenB, double *C,
                       long sCm, long sCn, long bC, long lenC, double alpha, double beta);
/* S = U^T U and C_l = U^-T H_l U^-1 from upper bands (n, k as above); C: nl x npad x npad. */
int bspatom_stage_standard_form(int n, int k, int nl, const double *SB, const double *HB, double *UB,
                                double *C, int32_t *info);
/* dense symmetric (npad multiple of 64, full storage) -> lower band AB[d + j*128], d <= 64 */
int bspatom_stage_sy2sb(int npad, int batch, const double *A, double *AB);
/* the panel factorisation of sy2sb alone (tests): panel = A[c0+64 .., c0 .. c0+63] of each dense npad x npad matrix (column-major).
 * On return the panel holds [R; 0]; V[b][c][0..m-1], W[b][c][0..m-1] (m = npad - c0 - 64) with I - V T V^T = I - W V^T orthogonal and
 * (I - W V^T)^T panel = [R; 0].  The panel QR inside LAPACK DSYTRD's blocked reduction (matrices.f90:248). */
int bspatom_stage_panel(int npad, int c0, int batch, double *A, double *V, double *W);
/* band (AB as above, leading n x n) -> tridiagonal d[n], e[n-1] (ld npad) */
int bspatom_stage_sb2st(int n, int npad, int batch, const double *AB, double *d, double *e);
/* first half of the two-step route (sb2st_version 9): band 64 -> band 16 in place, same layout */
int bspatom_stage_sb2sb(int n, int npad, int batch, double *AB);
/* band route (csrc/crawford.hip; BSP_ROUTE): the banded pencil (H_l, S) of nl channels, upper bands as bspatom_assemble returns
 * them, to the banded standard-form matrix orthogonally similar to L^-1 H_l L^-T (S = L L^T), half-width 2 (k - 1) - 1 <= 15,
 * in the layout above: AB[l][d + j*128] = A_l(j + d, j), npad = n rounded up to 64.  k - 1 <= 8.  info: 0, or the order of the
 * minor at which the factorisation of the (index-reversed) overlap broke down.  Replaces DPOTRF + DSYGST + the dense stage of
 * DSYTRD inside DSYGV (matrices.f90:248) in 6 n^2 (k - 1) flop and no dense matrix. */
int bspatom_stage_crawford(int n, int k, int nl, const double *SB, const double *HB, double *AB, int32_t *info);
/* eigenvalue m (0-based, ascending) of ONE banded pencil (H, S), upper bands as bspatom_assemble returns them, k - 1 <= 8, by
 * multisection on the inertia of H - x S (csrc/bandsect.hip): what starts the inverse iteration for the eigenvector the reference
 * consumes (matrices.f90:267) while the reductions of the batch are still running */
int bspatom_stage_band_eigenvalue(int n, int k, const double *SB, const double *HB, int m, double *lambda);
/* eigenvalues of tridiagonal matrices, ascending */
int bspatom_stage_bisect(int n, int batch, const double *d, const double *e, double *w);

#ifdef __cplusplus
}
#endif
#endif
