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
#include "bandsect.h"

namespace bsp {

constexpr int EB_MAX = 15;                 // half-bandwidth limit (k <= 16)

__device__ __forceinline__ double pencil_entry(const double *__restrict__ SB, const double *__restrict__ HB,
                                               int n, int b, double E, int r, int c)
{
    const int d = (r > c) ? (r - c) : (c - r);
    if (d > b || r < 0 || c < 0 || r >= n || c >= n) return 0.0;
    const int lo = (r > c) ? c : r;
    return HB[(size_t)d * n + lo] - E * SB[(size_t)d * n + lo];
}

// work layout per vector: U[n][2b+1], L[n][b], ipiv[n] (as doubles), tmp[n]
// wave reductions on the DPP path (a ds_bpermute butterfly costs an LDS round trip per level, and every one of the
// n dependent steps of this kernel has a reduction in it)
template <int CTRL>
__device__ __forceinline__ double ev_dpp(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ double ev_readlane(double x, int l)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return r.d;
}
__device__ __forceinline__ double ev_wave_sum(double x)          // every lane gets the sum over the 64 lanes
{
    x += ev_dpp<0x111>(x); x += ev_dpp<0x112>(x); x += ev_dpp<0x114>(x); x += ev_dpp<0x118>(x);   // row_shr 1, 2, 4, 8 (zeros shift in)
    return (ev_readlane(x, 15) + ev_readlane(x, 31)) + (ev_readlane(x, 47) + ev_readlane(x, 63));
}
__device__ __forceinline__ double ev_row0_max(double x)          // max over lanes 0..15 (x >= 0 there), uniform
{
    x = fmax(x, ev_dpp<0x111>(x)); x = fmax(x, ev_dpp<0x112>(x)); x = fmax(x, ev_dpp<0x114>(x)); x = fmax(x, ev_dpp<0x118>(x));
    return ev_readlane(x, 15);
}

// The kernel is ONE wavefront walking n dependent steps: a step's synchronisation must not wait for the global stores
// of the step before (U, L rows: fire and forget) -- __syncthreads() does (s_waitcnt vmcnt(0)), which put a memory
// round trip into every column of the factorisation.
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// BT >= b = k - 1: rows of the register window (instances 8 and 15).  One wavefront (lane = its lane); y: n doubles of LDS; work:
// n (3 b + 3) doubles; *info <- iv + 1 if the iterate vanished.
template <int BT>
__device__ __forceinline__ void invit_body(int n, int k, const double *__restrict__ SB, const double *__restrict__ HB, const double E,
                                           double *work, double *vec, int *info, int iv, double *y, const int lane)
{
    const int b = k - 1;
    double *U = work, *Lm = work + (size_t)n * (2 * b + 1), *piv = Lm + (size_t)n * b, *tmp = piv + n;

    // scale for the zero-pivot perturbation: ~ eps * max|M_jj|
    double dmax = 0.0;
    for (int j = lane; j < n; j += 64) dmax = fmax(dmax, fabs(HB[j] - E * SB[j]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off));
    const double pertol = 2.220446049250313e-16 * fmax(dmax, 1e-300);

    // ---- banded LU with partial pivoting (LINPACK dgbfa's row window), the window in REGISTERS (round 4): lane c (0 .. 2 b) holds
    // column j + c of the rows j .. j + b, a[t] = M(j + t, j + c).  The pivot search runs down lane 0's registers, the pivot row
    // and the multipliers go round by readlane, the rank-1 update is b multiply-adds inside every lane, and the window slides
    // by renaming its rows and a one-lane DPP shift of its columns: no LDS, no barrier in the n dependent steps.  (The first
    // form kept the window as a ring in LDS with five barriers and as many dependent LDS round trips per column: ~2.5 us per
    // column, 10 of the kernel's 11 ms.)  Pivot choice, perturbation of a vanishing pivot, multipliers and updates are the first
    // form's, operation for operation. ----
    double a[BT + 1];
#pragma unroll
    for (int t = 0; t <= BT; ++t) a[t] = (t <= b && lane <= 2 * b) ? pencil_entry(SB, HB, n, b, E, t, lane) : 0.0;
    // The row that enters the window at step j (row j + b + 1) does not depend on the elimination: its entries are fetched PF
    // steps at a time (one per lane and step), so that the memory latency is paid once per PF columns
    // ... and a block AHEAD: requested at the start of a block for the next one (with one buffer every block of PF dependent steps
    // began with a full round trip to memory -- 512 of them per pass over the matrix, most of what this kernel took)
    constexpr int PF = 8;
    double pe[PF], pn[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) pn[u] = (lane <= 2 * b) ? pencil_entry(SB, HB, n, b, E, u + b + 1, u + 1 + lane) : 0.0;
    for (int j = 0; j < n; ++j) {
        if ((j & (PF - 1)) == 0) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                pe[u] = pn[u];
                pn[u] = (lane <= 2 * b) ? pencil_entry(SB, HB, n, b, E, j + PF + u + b + 1, j + PF + u + 1 + lane) : 0.0;
            }
        }
        const int nrow = (n - 1 - j < b) ? (n - 1 - j) : b;       // rows below the pivot row
        // pivot search over rows j .. j + nrow in column j (lane 0's registers); ties go to the smallest row
        double mxv = fabs(a[0]);
        int dpl = 0;
#pragma unroll
        for (int t = 1; t <= BT; ++t) {
            const double v = (t <= nrow) ? fabs(a[t]) : -1.0;
            if (v > mxv) { mxv = v; dpl = t; }
        }
        const int dp = __builtin_amdgcn_readfirstlane(dpl);
        if (dp != 0) {                                               // uniform: rows j and j + dp change places, in every column
#pragma unroll
            for (int t = 1; t <= BT; ++t)
                if (t == dp) { const double t1 = a[0]; a[0] = a[t]; a[t] = t1; }
        }
        double pv = ev_readlane(a[0], 0);
        if (fabs(pv) < pertol) pv = (pv < 0.0) ? -pertol : pertol;
        const double rpv = 1.0 / pv;
        double lmine = 0.0;                                          // this lane's multiplier (row j + lane), for the store
#pragma unroll
        for (int t = 1; t <= BT; ++t) {
            const double lt = (t <= nrow) ? ev_readlane(a[t], 0) * rpv : 0.0;
            lmine = (lane == t) ? lt : lmine;
            a[t] = (lane >= 1) ? a[t] - lt * a[0] : 0.0;             // columns j + 1 .. j + 2 b; column j is eliminated
        }
        if (lane == 0) { a[0] = pv; piv[j] = (double)(j + dp); }
        if (lane <= 2 * b) U[(size_t)j * (2 * b + 1) + lane] = a[0];
        if (lane >= 1 && lane <= b) Lm[(size_t)j * b + lane - 1] = lmine;
        // slide the window: row j leaves, row j + b + 1 enters; column j leaves, column j + 2 b + 1 enters (zero above the entering row)
        double pin = pe[0];
#pragma unroll
        for (int u = 1; u < PF; ++u) pin = ((j & (PF - 1)) == u) ? pe[u] : pin;
#pragma unroll
        for (int t = 0; t < BT; ++t) a[t] = ev_dpp<0x130>(a[t + 1]);  // wave_shl:1 -- lane c takes lane c + 1
        a[BT] = 0.0;
#pragma unroll
        for (int t = 0; t <= BT; ++t)
            if (t == b) a[t] = pin;                                  // uniform
    }

    // ---- inverse iteration: 3 solves ----
    for (int j = lane; j < n; j += 64) y[j] = 1.0;
    __syncthreads();                                   // also: U, L, piv are in memory (vmcnt(0))
    for (int iter = 0; iter < 3; ++iter) {
        // forward: y <- L^-1 P y.  The b + 1 entries y[j .. j + b] that step j can touch live in a REGISTER window, lane i holding
        // y[j + i]: the pivot value goes round by readlane, the window moves on by a one-lane DPP shift, the entry that enters
        // it comes from LDS a block of PF steps ahead and the finished y[j] goes back there without anybody waiting for it --
        // no LDS round trip, no barrier inside the n dependent steps.  The operations and their order are the first form's.
        {
            double w = (lane <= b && lane < n) ? y[lane] : 0.0;
            double lvn[PF], pvn[PF];                     // the next block's multipliers and pivot rows, requested a block ahead
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                lvn[u] = (u < n && lane >= 1 && lane <= b) ? Lm[(size_t)u * b + lane - 1] : 0.0;
                pvn[u] = (u < n) ? piv[u] : 0.0;
            }
            for (int j0 = 0; j0 < n; j0 += PF) {
                double lv[PF], pvv[PF], yin[PF];         // multipliers, pivot rows, entering entries of PF steps: independent of the recurrence
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int j = j0 + u, jn = j + PF;
                    lv[u] = lvn[u]; pvv[u] = pvn[u];
                    lvn[u] = (jn < n && lane >= 1 && lane <= b) ? Lm[(size_t)jn * b + lane - 1] : 0.0;
                    pvn[u] = (jn < n) ? piv[jn] : 0.0;
                    yin[u] = (j + b + 1 < n) ? y[j + b + 1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int j = j0 + u;
                    if (j < n) {
                        const int p = (int)pvv[u];
                        if (p != j) {                                    // uniform: rows j and p change places (p - j <= b: inside the window)
                            const double t0 = ev_readlane(w, 0), tq = ev_readlane(w, p - j);
                            w = lane == 0 ? tq : (lane == p - j ? t0 : w);
                        }
                        const double yj = ev_readlane(w, 0);
                        const int nrow = (n - 1 - j < b) ? (n - 1 - j) : b;
                        if (lane >= 1 && lane <= nrow) w -= lv[u] * yj;
                        if (lane == 0) y[j] = yj;
                        w = ev_dpp<0x130>(w);                            // wave_shl:1 -- lane i takes lane i + 1
                        if (lane == b) w = yin[u];
                    }
                }
            }
            lds_sync();
        }
        // backward: x_j = (y_j - sum_{cc=1..2b} U[j][cc] x_{j+cc}) / U[j][0], the 2 b entries x[j + 1 .. j + 2 b] in a register window
        // likewise (lane cc holds x[j + cc]; zero beyond the matrix)
        {
            double w = 0.0;
            double uvn[PF], udn[PF];                     // the next block's rows of U and their pivots, requested a block ahead
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int j = n - 1 - u;
                uvn[u] = (j >= 0 && lane >= 1 && lane <= 2 * b) ? U[(size_t)j * (2 * b + 1) + lane] : 0.0;
                udn[u] = (j >= 0) ? U[(size_t)j * (2 * b + 1)] : 1.0;
            }
            for (int j0 = n - 1; j0 >= 0; j0 -= PF) {
                double uv[PF], ru[PF], yv[PF];           // row j of U: lane cc holds U[j][cc]; 1 / U[j][0] and y[j] for every lane
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int j = j0 - u, jn = j - PF;
                    uv[u] = uvn[u];
                    ru[u] = (j >= 0) ? 1.0 / udn[u] : 0.0;
                    uvn[u] = (jn >= 0 && lane >= 1 && lane <= 2 * b) ? U[(size_t)jn * (2 * b + 1) + lane] : 0.0;
                    udn[u] = (jn >= 0) ? U[(size_t)jn * (2 * b + 1)] : 1.0;
                    yv[u] = (j >= 0) ? y[j] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int j = j0 - u;
                    if (j >= 0) {
                        double s2 = 0.0;
                        if (lane >= 1 && lane <= 2 * b && j + lane < n) s2 = uv[u] * w;
                        s2 = ev_wave_sum(s2);
                        const double xj = (yv[u] - s2) * ru[u];
                        if (lane == 0) y[j] = xj;
                        w = ev_dpp<0x138>(w);                            // wave_shr:1 -- lane i takes lane i - 1
                        if (lane == 1) w = xj;
                    }
                }
            }
            lds_sync();
        }
        // normalise by max-abs, then rhs = S x for the next iteration
        double mx = 0.0;
        for (int j = lane; j < n; j += 64) mx = fmax(mx, fabs(y[j]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        const double sc = (mx > 0.0) ? 1.0 / mx : 1.0;
        // change against the previous iterate (tmp still holds it), up to sign: below 1e-12 of the max-norm after the
        // SECOND solve the third is skipped (it costs a fifth of the kernel, which is a chain of n dependent steps
        // per pass on one wavefront and the last thing the solve waits for)
        double dp = 0.0, dm = 0.0;
        for (int j = lane; j < n; j += 64) {
            const double v = y[j] * sc;
            if (iter == 1) { const double o = tmp[j]; dp = fmax(dp, fabs(v - o)); dm = fmax(dm, fabs(v + o)); }
            y[j] = v; tmp[j] = v;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { dp = fmax(dp, __shfl_xor(dp, off)); dm = fmax(dm, __shfl_xor(dm, off)); }
        __syncthreads();
        if (iter == 1 && fmin(dp, dm) <= 1e-12) break;
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
            lds_sync();
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
    if (!(q > 0.0)) { if (lane == 0) atomicExch(info, iv + 1); q = 1.0; }
    const double nrm = sgn / sqrt(q);
    for (int j = lane; j < n; j += 64) vec[j] = y[j] * nrm;
}

template <int BT>
__global__ __launch_bounds__(64) void invit_kernel(int n, int k, const double *__restrict__ SB, const double *__restrict__ HBall,
                                                  const int *__restrict__ chan, const double *__restrict__ Eall,
                                                  double *workall, double *vecall, int *info)
{
    extern __shared__ double y[];                       // n doubles
    // one wavefront on a chain of dependent steps, usually beside the batched bisection whose waves are pure VALU work:
    // issue priority over them
    __builtin_amdgcn_s_setprio(3);
    const int iv = blockIdx.x, b = k - 1;
    invit_body<BT>(n, k, SB, HBall + (size_t)chan[iv] * k * n, Eall[iv], workall + (size_t)iv * ((size_t)n * (3 * b + 3)),
                   vecall + (size_t)iv * n, info, iv, y, threadIdx.x);
}

// The consumed eigenvector EARLY (capi.hip::solve_impl, band route): eigenvalue m of the pencil of one channel by multisection on its
// inertia (bandsect.h: 256 threads), then the inverse iteration by the first wavefront of the SAME workgroup while the others leave.
// One launch, because the workgroup asks for a CU's whole LDS (own_lds bytes of dynamic LDS it never touches): it runs beside the
// band reduction, whose 3000 launches each wait for their slowest wave -- and a wave that shares its SIMD with one of these runs
// at two thirds of its speed (measured: reduction 28 -> 31 ms; at a lower priority these waves took 30 ms instead of 17 and were
// still holding LDS when the chase needed every slot of the chip).  The CU is claimed while the reduction's first launches are
// still small, and kept: two launches would have to find a drained CU twice.  1 / 256 of the chip for 17 ms.
__global__ __launch_bounds__(BS_T) void early_vector_kernel(int n, int k, const double *__restrict__ SB, const double *__restrict__ HB,
                                                           int m, double *Eout, double *work, double *vec, int *info)
{
    extern __shared__ double y[];                       // n doubles (and the padding up to a CU's LDS)
    __shared__ BandSectLds L;
    const double lam = band_multisect(n, k, SB, HB, m, L);
    if (threadIdx.x == 0) *Eout = lam;
    if (threadIdx.x >= 64) return;                      // (a barrier counts the waves that are left)
    invit_body<BS_B>(n, k, SB, HB, lam, work, vec, info, 0, y, threadIdx.x);
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
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(invit_kernel<8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(invit_kernel<EB_MAX>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        attr_set = true;
    }
    if (k - 1 <= 8) hipLaunchKernelGGL(invit_kernel<8>, dim3(nvec), dim3(64), lds, st, n, k, d_SB, d_HB, d_chan, d_E, d_work, d_vec, d_info);
    else hipLaunchKernelGGL(invit_kernel<EB_MAX>, dim3(nvec), dim3(64), lds, st, n, k, d_SB, d_HB, d_chan, d_E, d_work, d_vec, d_info);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// eigenvalue m (0-based) of the pencil (d_HB: ONE channel's band) -> *d_E, its eigenvector -> d_vec; d_work as above for one vector
int launch_early_vector(int n, int k, const double *d_SB, const double *d_HB, int m, double *d_E, double *d_work, double *d_vec,
                        int *d_info, hipStream_t st, bool own_cu)
{
    if (k < 2 || k - 1 > BS_B || m < 0 || m >= n) return BSP_ERR_UNSUPPORTED;
    size_t lds = (size_t)n * sizeof(double);
    const size_t stat = sizeof(BandSectLds), room = 160 * 1024 - stat - 512;
    if (lds > room) return BSP_ERR_UNSUPPORTED;
    if (own_cu && lds + stat < OWN_CU_LDS) lds = OWN_CU_LDS - stat;
    static bool attr_set = false;
    if (!attr_set) {
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(early_vector_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)room));
        attr_set = true;
    }
    hipLaunchKernelGGL(early_vector_kernel, dim3(1), dim3(BS_T), lds, st, n, k, d_SB, d_HB, m, d_E, d_work, d_vec, d_info);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// ---- dipole matrix elements between eigenvectors (TRANS_AMP, PhotoIon.f90:95-107) ----------------------------------
// v = (a0 R_r + a1 R_{1/r} + a2 R_{d/dr}) x with the full bands RB[c][d + k - 1][i] = R_c(i, i + d) of assemble.hip
// (the reference forms A(:,:) = c1 rij(:,:,1) + c2 rij(:,:,2) and calls DGEMV on the dense matrix; its zeros outside the
// band contribute nothing).  One thread per row, diagonals in ascending column order.
__global__ __launch_bounds__(256) void band_apply_kernel(int n, int k, const double *__restrict__ RB, double a0, double a1,
                                                        double a2, const double *__restrict__ x, double *__restrict__ v)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t cs = (size_t)(2 * k - 1) * n;
    double s = 0.0;
    for (int d = -(k - 1); d <= k - 1; ++d) {
        const int j = i + d;
        if (j < 0 || j >= n) continue;
        const size_t idx = (size_t)(d + k - 1) * n + i;
        const double a = (a0 * RB[idx] + a1 * RB[cs + idx]) + a2 * RB[2 * cs + idx];
        s += a * x[j];
    }
    v[i] = s;
}

// D[j] = Z[j][:] . v  (the DDOT of every final state): one workgroup per vector, strided partial sums and a fixed
// reduction tree, so the result does not depend on timing
__global__ __launch_bounds__(256) void dots_kernel(int n, const double *__restrict__ Z, const double *__restrict__ v,
                                                  double *__restrict__ D)
{
    __shared__ double red[256];
    const double *z = Z + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += z[i] * v[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) D[blockIdx.x] = red[0];
}

int launch_band_apply(int n, int k, const double *d_RB, const double a[3], const double *d_x, double *d_v, hipStream_t st)
{
    hipLaunchKernelGGL(band_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, k, d_RB, a[0], a[1], a[2], d_x, d_v);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

int launch_dots(int n, int m, const double *d_Z, const double *d_v, double *d_D, hipStream_t st)
{
    if (m <= 0) return BSP_OK;
    hipLaunchKernelGGL(dots_kernel, dim3(m), dim3(256), 0, st, n, d_Z, d_v, d_D);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// ---- WRITE_WF (Bsp_Atom.f90:118-146): one thread per tabulation point ------------------------
// K = the order k as a compile-time constant: the three work arrays of BSPLVB are indexed by unrolled loops and live in
// registers (with k a run-time value they were 432 bytes of scratch per thread: round-3 verdict, "no scratch on the path")
template <int K>
__global__ __launch_bounds__(128) void wf_kernel(int nkp, int n, const double *__restrict__ rt0, const double *__restrict__ c,
                          double ra, double rb, int npts, double *rout, double *uout, int *status)
{
    constexpr int k = K;
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
    double biatx[K + 1], dl[K + 1], dR[K + 1];
#pragma unroll
    for (int j = 0; j <= k; ++j) biatx[j] = 0.0;
    biatx[1] = 1.0;
    if (k > 1) {
        if (t[left + 1] <= t[left]) { atomicExch(status, BSP_ERR_BSPLVB); return; }   // FATAL ERROR - BSPLVB
#pragma unroll
        for (int j = 1; j < k; ++j) {
            dR[j] = t[left + j] - r;
            dl[j] = r - t[left + 1 - j];
            double saved = 0.0;
#pragma unroll
            for (int q = 1; q <= j; ++q) {
                const double term = biatx[q] / (dR[q] + dl[j + 1 - q]);
                biatx[q] = saved + dR[q] * term;
                saved = dl[j + 1 - q] * term;
            }
            biatx[j + 1] = saved;
        }
    }
    double sumf = 0.0;
#pragma unroll
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
    if (k > 16 || k < 1) return BSP_ERR_ARG;
    const dim3 grid((npts + 1 + 127) / 128), block(128);
    switch (k) {
#define WF_CASE(K) case K: hipLaunchKernelGGL(wf_kernel<K>, grid, block, 0, st, nkp, n, d_rt, d_c, ra, rb, npts, d_r, d_u, d_status); break;
        WF_CASE(1) WF_CASE(2) WF_CASE(3) WF_CASE(4) WF_CASE(5) WF_CASE(6) WF_CASE(7) WF_CASE(8)
        WF_CASE(9) WF_CASE(10) WF_CASE(11) WF_CASE(12) WF_CASE(13) WF_CASE(14) WF_CASE(15) WF_CASE(16)
#undef WF_CASE
    }
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
