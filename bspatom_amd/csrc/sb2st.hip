// sb2st.hip -- stage 2 of the two-stage tridiagonalisation: band (half-width b = 64) -> tridiagonal
// by Householder bulge chasing.  Replaces the second half of LAPACK DSYTRD (reference call
// matrices.f90:248 -> DSYGV).  One workgroup (256 threads) per l-channel; the channels of a
// batch run concurrently on different CUs.
//
// Band storage (lower, LD = 2b rows so that the bulge fits): AB[d + j*LD] = A(j+d, j), d < 2b.
//
// Sweep s annihilates column s below the sub-diagonal (reflector of length L <= b acting on rows
// r0 = s+1 .. s+L, two-sided on the diagonal block), then chases the bulge down the band: each
// chase step right-applies the current reflector to the L2 x L block B below the diagonal block
// (fill-in), annihilates B's first column with a new reflector, left-applies it to the rest of
// B, and applies it two-sided to the next diagonal block D2.  B and D2 (64x64 doubles each) are
// staged in LDS; vectors and partial sums live in LDS as well.
#include "common.h"

namespace bsp {

constexpr int SB = 64;             // band half-width handled by this kernel
constexpr int TLD = SB + 1;        // LDS tile row stride (bank-conflict padding)

struct HouseOut { double beta, tau; };

// LAPACK dlarfg on x[0..L) held in LDS (vector overwritten by v, v[0] = 1).  All 256 threads call;
// wave 0 does the work.  sh[0..1] scratch.  Entries whose squares underflow are dropped (tau = 0).
__device__ static HouseOut house_lds(double *x, int L, double *sh, int tid)
{
    if (tid < 64) {
        double xi = (tid < L) ? x[tid] : 0.0;
        double sq = (tid >= 1) ? xi * xi : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
        const double alpha = __shfl(xi, 0);
        double beta, tau, scale;
        if (!(alpha * alpha + sq > 1e-280) || sq == 0.0) { beta = alpha; tau = 0.0; scale = 0.0; }
        else {
            const double nrm = sqrt(alpha * alpha + sq);
            beta = (alpha >= 0.0) ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (tid < L) x[tid] = (tid == 0) ? 1.0 : xi * scale;
        else x[tid] = 0.0;
        if (tid == 0) { sh[0] = beta; sh[1] = tau; }
    }
    __syncthreads();
    HouseOut o; o.beta = sh[0]; o.tau = sh[1];
    __syncthreads();
    return o;
}

// Two-sided update D <- H D H, H = I - tau v v^T, on the L x L symmetric tile Dt (full storage):
// p = tau D v ; alpha = -1/2 tau p^T v ; p += alpha v ; D -= v p^T + p v^T.
__device__ static void two_sided(double (*Dt)[TLD], int L, const double *v, double tau, double *p,
                                 double (*red)[SB], double *sh, int tid)
{
    const int i = tid & 63, part = tid >> 6;
    double s = 0.0;
    if (i < L)
        for (int j = part * 16; j < part * 16 + 16; ++j) s += Dt[j][i] * v[j];   // D symmetric: column i
    red[part][i] = s;
    __syncthreads();
    if (tid < 64) {
        double pi = tau * (red[0][i] + red[1][i] + red[2][i] + red[3][i]);
        double dot = (i < L) ? pi * v[i] : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off);
        const double alpha = -0.5 * tau * dot;
        p[i] = (i < L) ? pi + alpha * v[i] : 0.0;
    }
    __syncthreads();
    if (i < L) {
        const double vi = v[i], pi = p[i];
        for (int j = part * 16; j < part * 16 + 16; ++j)
            if (j < L) Dt[j][i] -= v[j] * pi + p[j] * vi;
    }
    __syncthreads();
    (void)sh;
}

__global__ __launch_bounds__(256) void sb2st_kernel(int n, int npad, double *ABall, double *dall, double *eall)
{
    __shared__ double Bt[SB][TLD];      // Bt[j][i] = B(i, j)  (column j contiguous in i)
    __shared__ double Dt[SB][TLD];      // Dt[j][i] = D(i, j)
    __shared__ double v[SB], v2[SB], w[SB];
    __shared__ double red[4][SB];
    __shared__ double sh[4];
    constexpr int LD = 2 * SB;
    const int tid = threadIdx.x, i = tid & 63, part = tid >> 6;
    const size_t ch = blockIdx.x;
    double *AB = ABall + ch * (size_t)npad * LD;

    for (int s = 0; s < n - 2; ++s) {
        int L = (n - 1 - s < SB) ? (n - 1 - s) : SB;
        if (L < 2) break;
        int r0 = s + 1;
        // ---- start of sweep: reflector from column s, rows r0 .. r0+L-1 (d = 1..L) ----
        if (tid < 64) v[tid] = (tid < L) ? AB[(size_t)s * LD + 1 + tid] : 0.0;
        __syncthreads();
        HouseOut h = house_lds(v, L, sh, tid);
        double tau = h.tau;
        if (tid < L) AB[(size_t)s * LD + 1 + tid] = (tid == 0) ? h.beta : 0.0;
        // diagonal block D = A[r0:r0+L, r0:r0+L]: D(i,j) (i>=j) = AB[(i-j) + (r0+j)*LD]
        for (int j = part; j < SB; j += 4) {
            double val = 0.0;
            if (i < L && j < L && i >= j) val = AB[(size_t)(r0 + j) * LD + (i - j)];
            Dt[j][i] = val;
        }
        __syncthreads();
        for (int j = part; j < SB; j += 4)
            if (i < j && j < L) Dt[j][i] = Dt[i][j];            // mirror: D(i,j) = D(j,i) for i<j
        __syncthreads();
        two_sided(Dt, L, v, tau, w, red, sh, tid);
        for (int j = part; j < L; j += 4)
            if (i < L && i >= j) AB[(size_t)(r0 + j) * LD + (i - j)] = Dt[j][i];
        __syncthreads();
        // ---- chase ----
        double *vc = v, *vn = v2;
        while (r0 + L < n) {
            const int L2 = (n - (r0 + L) < SB) ? (n - (r0 + L)) : SB;
            // B(i,j) = A(r0+L+i, r0+j) = AB[(L+i-j) + (r0+j)*LD], i < L2, j < L
            for (int j = part; j < SB; j += 4) {
                double val = 0.0;
                if (i < L2 && j < L) val = AB[(size_t)(r0 + j) * LD + (L + i - j)];
                Bt[j][i] = val;
            }
            __syncthreads();
            // w = B vc ; B -= tau w vc^T
            {
                double sacc = 0.0;
                for (int j = part * 16; j < part * 16 + 16; ++j) sacc += Bt[j][i] * vc[j];
                red[part][i] = sacc;
            }
            __syncthreads();
            if (tid < 64) w[i] = tau * (red[0][i] + red[1][i] + red[2][i] + red[3][i]);
            __syncthreads();
            {
                const double wi = w[i];
                for (int j = part * 16; j < part * 16 + 16; ++j) Bt[j][i] -= wi * vc[j];
            }
            __syncthreads();
            // new reflector from B(:,0)
            if (tid < 64) vn[tid] = (tid < L2) ? Bt[0][tid] : 0.0;
            __syncthreads();
            HouseOut h2 = house_lds(vn, L2, sh, tid);
            const double tau2 = h2.tau;
            if (tid < 64) Bt[0][tid] = (tid == 0) ? h2.beta : 0.0;
            __syncthreads();
            // z = vn^T B(:,1:) ; B(:,1:) -= tau2 vn z^T      (thread: column j = i, rows split in 4 parts)
            {
                const int j = i;
                double sacc = 0.0;
                if (j >= 1)
                    for (int ii = part * 16; ii < part * 16 + 16; ++ii) sacc += vn[ii] * Bt[j][ii];
                red[part][j] = sacc;
            }
            __syncthreads();
            if (tid < 64) w[i] = tau2 * (red[0][i] + red[1][i] + red[2][i] + red[3][i]);
            __syncthreads();
            {
                const int j = i;
                if (j >= 1) {
                    const double zj = w[j];
                    for (int ii = part * 16; ii < part * 16 + 16; ++ii) Bt[j][ii] -= vn[ii] * zj;
                }
            }
            __syncthreads();
            for (int j = part; j < L; j += 4)
                if (i < L2) AB[(size_t)(r0 + j) * LD + (L + i - j)] = Bt[j][i];
            // next diagonal block D2 = A[r0+L : r0+L+L2, same]
            const int rn = r0 + L;
            for (int j = part; j < SB; j += 4) {
                double val = 0.0;
                if (i < L2 && j < L2 && i >= j) val = AB[(size_t)(rn + j) * LD + (i - j)];
                Dt[j][i] = val;
            }
            __syncthreads();
            for (int j = part; j < SB; j += 4)
                if (i < j && j < L2) Dt[j][i] = Dt[i][j];
            __syncthreads();
            two_sided(Dt, L2, vn, tau2, w, red, sh, tid);
            for (int j = part; j < L2; j += 4)
                if (i < L2 && i >= j) AB[(size_t)(rn + j) * LD + (i - j)] = Dt[j][i];
            __syncthreads();
            r0 = rn; L = L2; tau = tau2;
            double *tswap = vc; vc = vn; vn = tswap;
        }
    }
    __syncthreads();
    double *d = dall + ch * (size_t)npad, *e = eall + ch * (size_t)npad;
    for (int j = tid; j < n; j += 256) {
        d[j] = AB[(size_t)j * LD];
        e[j] = (j < n - 1) ? AB[(size_t)j * LD + 1] : 0.0;
    }
}


// ------------------------------------------------------------------------------------------------
// v2: same algorithm, restructured for latency.  Per chase step (5 workgroup barriers):
//   A  all waves : issue the global loads of the NEXT step's B', D2' into registers (prefetch);
//                  partial sums of w0 = B vc
//   B  wave 0    : w = tau w0 ; x' = B(:,0) - w vc_0 ; reflector (vn, tau2, beta2) ; s = vn^T w
//   C  all waves : partial sums of z0 = vn^T B (original B) and of p0 = D2 vn
//   D  wave 0    : z = tau2 (z0 - s vc)      wave 1: p = tau2 p0 + alpha vn
//   E  all waves : B <- B - w vc^T - vn z^T (column 0 := beta2 e1), D2 <- D2 - vn p^T - p vn^T, both
//                  stored to HBM from registers; then the prefetched B', D2' replace them in LDS.
// (B - w vc^T) is never formed: (I - tau2 vn vn^T)(B - w vc^T) = B - w vc^T - tau2 vn (vn^T B - (vn^T w) vc^T).
// In-sweep barriers wait for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): the prefetch loads
// and the result stores stay in flight across them (a __syncthreads() would drain vmcnt(0) at every
// barrier).  Cross-thread HBM dependencies exist only between sweeps; one __syncthreads() per sweep
// covers them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// lane-parallel dlarfg on one wavefront: xi = element i of x (0 beyond L).  Returns v_i.
__device__ __forceinline__ double wave_house(double xi, int lane, int L, double *beta, double *tau)
{
    const double sq = wave_sum((lane >= 1 && lane < L) ? xi * xi : 0.0);
    const double alpha = __shfl(xi, 0);
    double scale;
    if (!(alpha * alpha + sq > 1e-280) || sq == 0.0) { *beta = alpha; *tau = 0.0; scale = 0.0; }
    else {
        const double nrm = sqrt(alpha * alpha + sq);
        *beta = (alpha >= 0.0) ? -nrm : nrm;
        *tau = (*beta - alpha) / *beta;
        scale = 1.0 / (alpha - *beta);
    }
    return (lane == 0) ? 1.0 : ((lane < L) ? xi * scale : 0.0);
}

// dbg (timing experiments only, results are wrong when set): bit0 = no prefetch loads, bit1 = no chase stores
__global__ __launch_bounds__(256) void sb2st_kernel_v2(int n, int npad, double *ABall, double *dall, double *eall, int dbg)
{
    __shared__ double Bt[SB][TLD];      // Bt[j][i] = B(i, j)
    __shared__ double Dt[SB][TLD];      // Dt[j][i] = D2(i, j) for i >= j (lower triangle only)
    __shared__ double va[SB], vb[SB], w[SB], z[SB], pv[SB];
    __shared__ double red[4][SB], red2[4][SB];
    __shared__ double sc[8];
    constexpr int LD = 2 * SB;
    const int tid = threadIdx.x, i = tid & 63, part = tid >> 6;
    const size_t ch = blockIdx.x;
    double *AB = ABall + ch * (size_t)npad * LD;

    for (int s = 0; s < n - 2; ++s) {
        int L = (n - 1 - s < SB) ? (n - 1 - s) : SB;
        if (L < 2) break;
        int r0 = s + 1;
        double *vc = va, *vn = vb;
        __syncthreads();          // HBM stores of the previous sweep are visible to every wave
        // ---- sweep start: reflector from column s; two-sided update of D = A[r0:r0+L, r0:r0+L] ----
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int j = part + 4 * q;
            const bool ok = (i < L && j < L && i >= j);
            const double val = AB[ok ? ((size_t)(r0 + j) * LD + (i - j)) : 0];     // unconditional load, clamped address
            Dt[j][i] = ok ? val : 0.0;
        }
        double tau;
        if (tid < 64) {
            const double xraw = AB[(tid < L) ? ((size_t)s * LD + 1 + tid) : 0];
            const double xi = (tid < L) ? xraw : 0.0;
            double beta;
            const double vi = wave_house(xi, tid, L, &beta, &tau);
            vc[tid] = vi;
            if (tid < L) AB[(size_t)s * LD + 1 + tid] = (tid == 0) ? beta : 0.0;
            if (tid == 0) sc[0] = tau;
        }
        lds_barrier();
        tau = sc[0];
        {   // p0 = D vc (partials), D symmetric from its lower triangle
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int x = part * 16 + q;
                sacc += ((x <= i) ? Dt[x][i] : Dt[i][x]) * vc[x];
            }
            red[part][i] = sacc;
        }
        lds_barrier();
        if (tid < 64) {
            double pi = tau * (red[0][i] + red[1][i] + red[2][i] + red[3][i]);
            const double dot = wave_sum(pi * vc[i]);
            pv[i] = pi + (-0.5 * tau * dot) * vc[i];
        }
        lds_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int j = part + 4 * q;
            if (i < L && j < L && i >= j)
                AB[(size_t)(r0 + j) * LD + (i - j)] = Dt[j][i] - (vc[i] * pv[j] + pv[i] * vc[j]);
        }
        bool have = (r0 + L < n);
        int L2 = have ? ((n - (r0 + L) < SB) ? (n - (r0 + L)) : SB) : 0;
        lds_barrier();             // all reads of Dt done before it is overwritten
        if (have) {                // first chase step's tiles (not touched by the update above)
            const int rn = r0 + L;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = part + 4 * q;
                const bool okb = (i < L2 && j < L), okd = (i < L2 && j < L2 && i >= j);
                const double bv = AB[okb ? ((size_t)(r0 + j) * LD + (L + i - j)) : 0];
                const double dv = AB[okd ? ((size_t)(rn + j) * LD + (i - j)) : 0];
                Bt[j][i] = okb ? bv : 0.0; Dt[j][i] = okd ? dv : 0.0;
            }
        }
        lds_barrier();
        // ---- chase ----
        while (have) {
            const int rn = r0 + L;
            const bool have_next = (rn + L2 < n);
            const int L3 = have_next ? ((n - (rn + L2) < SB) ? (n - (rn + L2)) : SB) : 0;
            // A: prefetch next tiles into registers; partial w0 = B vc
            // (L3 = 0 on the last step: every address clamps to AB[0], every value is discarded)
            double pb[16], pd[16];
            {
                const int rnn = rn + L2;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int j = part + 4 * q;
                    const bool okb = (i < L3 && j < L2), okd = (i < L3 && j < L3 && i >= j);
                    if (dbg & 1) { pb[q] = 1e-3 * (i + j); pd[q] = 1e-3 * (i - j) + (i == j ? 1.0 : 0.0); continue; }
                    pb[q] = AB[okb ? ((size_t)(rn + j) * LD + (L2 + i - j)) : 0];
                    pd[q] = AB[okd ? ((size_t)(rnn + j) * LD + (i - j)) : 0];
                }
            }
            {
                double sacc = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) { const int j = part * 16 + q; sacc += Bt[j][i] * vc[j]; }
                red[part][i] = sacc;
            }
            lds_barrier();
            // B: wave 0 builds the new reflector
            if (tid < 64) {
                const double wi = tau * (red[0][i] + red[1][i] + red[2][i] + red[3][i]);
                const double xi = (i < L2) ? (Bt[0][i] - wi * vc[0]) : 0.0;
                double beta2, tau2;
                const double vi = wave_house(xi, i, L2, &beta2, &tau2);
                const double sdot = wave_sum(vi * wi);
                w[i] = wi; vn[i] = vi;
                if (i == 0) { sc[1] = beta2; sc[2] = tau2; sc[3] = sdot; }
            }
            lds_barrier();
            const double beta2 = sc[1], tau2 = sc[2], sdot = sc[3];
            // C: partial z0 = vn^T B (thread: column j = i, rows of this part) ; partial p0 = D2 vn
            {
                double zacc = 0.0, pacc = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int x = part * 16 + q;
                    zacc += vn[x] * Bt[i][x];                                   // column i of B, rows x
                    pacc += ((x <= i) ? Dt[x][i] : Dt[i][x]) * vn[x];           // row i of symmetric D2
                }
                red[part][i] = zacc; red2[part][i] = pacc;
            }
            lds_barrier();
            // D: wave 0 -> z, wave 1 -> p
            if (tid < 64) {
                z[i] = tau2 * ((red[0][i] + red[1][i] + red[2][i] + red[3][i]) - sdot * vc[i]);
            } else if (tid < 128) {
                double pi = tau2 * (red2[0][i] + red2[1][i] + red2[2][i] + red2[3][i]);
                const double dot = wave_sum(pi * vn[i]);
                pv[i] = pi + (-0.5 * tau2 * dot) * vn[i];
            }
            lds_barrier();
            // E: updates, stores, and hand-over to the prefetched tiles
            {
                const double wi = w[i], vni = vn[i], pi = pv[i];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int j = part + 4 * q;
                    double bnew = Bt[j][i] - (wi * vc[j] + vni * z[j]);
                    if (j == 0) bnew = (i == 0) ? beta2 : 0.0;
                    const bool st_ok = !(dbg & 2);
                    if (st_ok && i < L2 && j < L) AB[(size_t)(r0 + j) * LD + (L + i - j)] = bnew;
                    if (st_ok && i < L2 && j < L2 && i >= j)
                        AB[(size_t)(rn + j) * LD + (i - j)] = Dt[j][i] - (vni * pv[j] + pi * vn[j]);
                    {
                        const bool okb = (i < L3 && j < L2), okd = (i < L3 && j < L3 && i >= j);
                        Bt[j][i] = okb ? pb[q] : 0.0; Dt[j][i] = okd ? pd[q] : 0.0;
                    }
                }
            }
            lds_barrier();
            r0 = rn; L = L2; L2 = L3; tau = tau2; have = have_next;
            double *tswap = vc; vc = vn; vn = tswap;
        }
    }
    __syncthreads();
    double *d = dall + ch * (size_t)npad, *e = eall + ch * (size_t)npad;
    for (int j = tid; j < n; j += 256) {
        d[j] = AB[(size_t)j * LD];
        e[j] = (j < n - 1) ? AB[(size_t)j * LD + 1] : 0.0;
    }
}

int launch_sb2st(int n, int npad, int b, int batch, double *d_AB, double *d_d, double *d_e, hipStream_t st)
{
    if (b != SB) return BSP_ERR_ARG;
    static int ver = -1;
    if (ver < 0) { const char *e = getenv("BSP_SB2ST_VERSION"); ver = e ? atoi(e) : 2; }
    if (ver == 1) hipLaunchKernelGGL(sb2st_kernel, dim3(batch), dim3(256), 0, st, n, npad, d_AB, d_d, d_e);
    else {
        static int dbg = -1;
        if (dbg < 0) { const char *e = getenv("BSP_SB2ST_DBG"); dbg = e ? atoi(e) : 0; }
        hipLaunchKernelGGL(sb2st_kernel_v2, dim3(batch), dim3(256), 0, st, n, npad, d_AB, d_d, d_e, dbg);
    }
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
