// sy2sb.hip -- stage 1 of the two-stage tridiagonalisation: dense symmetric -> band (half-width 64).
//
// Replaces the first half of LAPACK DSYTRD inside DSYEV/DSYGV (reference call matrices.f90:248).
// Batched over l-channels: every launch covers all channels (grid.z / grid.x = channel).
//
// Per panel p (columns c0 = p*NB .. c0+NB-1, rows r0 = c0+NB .. npad-1, m = npad - r0):
//   1. panel_qr_kernel : Householder QR of P = A[r0:, c0:c0+NB]  ->  R (left in A), explicit V
//                        (unit lower trapezoid, written twice into buf = [V | . | V]) and tau
//   2. G = V^T V (MFMA GEMM), form_T_kernel: T from G and tau (dlarft forward/columnwise)
//   3. W = V T                                   (MFMA GEMM)
//   4. Y = A22 W              -> buf middle slot  (MFMA GEMM, the SYMM-shaped half of the flops)
//   5. K = W^T Y ; Z = Y - 1/2 V K (in place)    (MFMA GEMMs)
//   6. A22 -= [V Z] [Z V]^T                      (MFMA GEMM, the SYR2K-shaped half of the flops)
// with A22 = A[r0:, r0:] kept in full symmetric storage (column-major, ld = npad).
//
// panel_qr_kernel: ONE workgroup (512 threads) per channel.  The m x 64 panel is processed in
// sub-panels of SW = 4 columns held entirely in registers (thread t owns rows t, t+512, ...;
// RPT rows per thread), so a column step costs two workgroup reductions and no memory traffic;
// after a sub-panel is factored its block reflector (I - V_s T_s^T V_s^T) is applied to the
// remaining panel columns CW = 4 at a time, streaming them through registers.
#include "common.h"
#include <vector>

namespace bsp {

constexpr int NB = 64;     // band half-width after stage 1
constexpr int PQ_THREADS = 512;
constexpr int SW = 4, CW = 4;

template <int NV>
__device__ __forceinline__ void block_allreduce(double (&v)[NV], double *red, int lane, int wave)
{
    constexpr int NW = PQ_THREADS / 64;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w * NV + i];
        v[i] = s;
    }
    __syncthreads();
}

template <int RPT>
__global__ __launch_bounds__(PQ_THREADS) void panel_qr_kernel(int npad, int r0, int c0, double *Aall,
                                                             double *bufall, double *tauall)
{
    __shared__ double red[(PQ_THREADS / 64) * SW * CW];
    __shared__ double sh_alpha;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t ch = blockIdx.x;
    const long ld = npad;
    const int m = npad - r0;
    double *P = Aall + ch * (size_t)npad * npad + (size_t)c0 * ld + r0;   // P(i,j) = P[i + j*ld]
    double *buf = bufall + ch * (size_t)npad * 3 * NB;                    // buf(i,c) = buf[i + c*npad]
    double *tau_g = tauall + ch * NB;

    for (int js = 0; js < NB; js += SW) {
        double a[RPT][SW];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int i = tid + PQ_THREADS * q;
#pragma unroll
            for (int jj = 0; jj < SW; ++jj) a[q][jj] = (i < m) ? P[i + (size_t)(js + jj) * ld] : 0.0;
        }
        double taus[SW];
        // ---- factor the sub-panel ----
#pragma unroll
        for (int jj = 0; jj < SW; ++jj) {
            const int j = js + jj;                       // pivot row (panel-relative)
            double part[1] = {0.0};
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
                if (i > j && i < m) part[0] += a[q][jj] * a[q][jj];
                if (i == j) sh_alpha = a[q][jj];
            }
            block_allreduce<1>(part, red, lane, wave);   // barriers inside publish sh_alpha too
            const double alpha = sh_alpha, sigma = part[0];
            double beta, tau, scale;
            if (j >= m - 1 || !(alpha * alpha + sigma > 1e-280) || sigma == 0.0) {
                // nothing (numerically) below the pivot: H = I; tiny entries are dropped
                beta = alpha; tau = 0.0; scale = 0.0;
            } else {
                const double nrm = sqrt(alpha * alpha + sigma);
                beta = (alpha >= 0.0) ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            taus[jj] = tau;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
                if (i > j) a[q][jj] *= scale;            // v_i
                if (i == j) a[q][jj] = beta;             // R(j,j)
            }
            // apply H_j to the remaining columns of the sub-panel
            if (jj < SW - 1) {
                double w[SW - 1];
#pragma unroll
                for (int c = 0; c < SW - 1; ++c) w[c] = 0.0;
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int i = tid + PQ_THREADS * q;
                    const double vv = (i > j) ? a[q][jj] : ((i == j) ? 1.0 : 0.0);
#pragma unroll
                    for (int c = jj + 1; c < SW; ++c) w[c - 1] += vv * a[q][c];
                }
                block_allreduce<SW - 1>(w, red, lane, wave);
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int i = tid + PQ_THREADS * q;
                    const double vv = (i > j) ? a[q][jj] : ((i == j) ? 1.0 : 0.0);
#pragma unroll
                    for (int c = jj + 1; c < SW; ++c) a[q][c] -= tau * vv * w[c - 1];
                }
            }
        }
        // ---- T_s of the sub-panel from G_s = V_s^T V_s ----
        double g[SW * SW];
#pragma unroll
        for (int x = 0; x < SW * SW; ++x) g[x] = 0.0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int i = tid + PQ_THREADS * q;
            double vv[SW];
#pragma unroll
            for (int c = 0; c < SW; ++c) vv[c] = (i > js + c) ? a[q][c] : ((i == js + c) ? 1.0 : 0.0);
#pragma unroll
            for (int x = 0; x < SW; ++x)
#pragma unroll
                for (int y = x + 1; y < SW; ++y) g[x * SW + y] += vv[x] * vv[y];
        }
        block_allreduce<SW * SW>(g, red, lane, wave);
        double Ts[SW][SW];
#pragma unroll
        for (int x = 0; x < SW; ++x)
#pragma unroll
            for (int y = 0; y < SW; ++y) Ts[x][y] = 0.0;
#pragma unroll
        for (int y = 0; y < SW; ++y) {
            Ts[y][y] = taus[y];
#pragma unroll
            for (int x = 0; x < y; ++x) {
                double s = 0.0;
#pragma unroll
                for (int z = x; z < y; ++z) s += Ts[x][z] * g[z * SW + y];
                Ts[x][y] = -taus[y] * s;
            }
        }
        // ---- write R / zeros back to the panel, V (explicit) to buf, tau ----
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int i = tid + PQ_THREADS * q;
            if (i < m) {
#pragma unroll
                for (int jj = 0; jj < SW; ++jj) {
                    const int j = js + jj;
                    const double vv = (i > j) ? a[q][jj] : ((i == j) ? 1.0 : 0.0);
                    P[i + (size_t)j * ld] = (i <= j) ? a[q][jj] : 0.0;
                    buf[i + (size_t)j * npad] = vv;
                    buf[i + (size_t)(2 * NB + j) * npad] = vv;
                }
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int jj = 0; jj < SW; ++jj) tau_g[js + jj] = taus[jj];
        }
        // ---- apply (I - V_s T_s^T V_s^T) to the remaining panel columns, CW at a time ----
        for (int cs = js + SW; cs < NB; cs += CW) {
            // pass 1: wp = V_s^T X (X streamed, not kept: registers hold V_s only)
            double wp[SW * CW];
#pragma unroll
            for (int z = 0; z < SW * CW; ++z) wp[z] = 0.0;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
                if (i < m) {
                    double x[CW];
#pragma unroll
                    for (int cc = 0; cc < CW; ++cc) x[cc] = P[i + (size_t)(cs + cc) * ld];
#pragma unroll
                    for (int c = 0; c < SW; ++c) {
                        const double vv = (i > js + c) ? a[q][c] : ((i == js + c) ? 1.0 : 0.0);
#pragma unroll
                        for (int cc = 0; cc < CW; ++cc) wp[c * CW + cc] += vv * x[cc];
                    }
                }
            }
            block_allreduce<SW * CW>(wp, red, lane, wave);
            double w2[SW * CW];
#pragma unroll
            for (int c = 0; c < SW; ++c)
#pragma unroll
                for (int cc = 0; cc < CW; ++cc) {
                    double s = 0.0;
#pragma unroll
                    for (int z = 0; z <= c; ++z) s += Ts[z][c] * wp[z * CW + cc];   // (T_s^T wp)
                    w2[c * CW + cc] = s;
                }
            // pass 2: X -= V_s w2 (X re-read from L2; each thread touches only its own rows)
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
                if (i < m) {
                    double x[CW];
#pragma unroll
                    for (int cc = 0; cc < CW; ++cc) x[cc] = P[i + (size_t)(cs + cc) * ld];
#pragma unroll
                    for (int c = 0; c < SW; ++c) {
                        const double vv = (i > js + c) ? a[q][c] : ((i == js + c) ? 1.0 : 0.0);
#pragma unroll
                        for (int cc = 0; cc < CW; ++cc) x[cc] -= vv * w2[c * CW + cc];
                    }
#pragma unroll
                    for (int cc = 0; cc < CW; ++cc) P[i + (size_t)(cs + cc) * ld] = x[cc];
                }
            }
        }
        __syncthreads();   // panel columns written by this sub-panel are read by the next one
    }
}

// ------------------------------------------------------------------------------------------------
// panel_qr2_kernel: the same factorisation with the memory latency taken off the chain.  Measured on the
// kernel above: it moves ~13 KB per row of the panel (each trailing column is read twice and written once
// per 4-column sub-panel) through ONE compute unit, with every pass waiting for its loads -- while the other
// channel group's GEMMs saturate HBM.  Here
//   * the trailing columns go through LDS in groups of CW2 = 2, double-buffered with LDS-DMA loads
//     (global_load_lds_dwordx4: no VGPRs, the next group is in flight while this one is processed; both
//     passes over a group read LDS, not memory);
//   * the next sub-panel's four columns never leave the registers (they are the first two groups);
//   * the wave-level half of every reduction is a DPP row scan instead of a ds_bpermute butterfly.
// One workgroup of 512 threads per channel, rows t, t + 512, ...; RPT <= 8 (m <= 4096: two buffers of
// 2 x 4096 doubles = 128 KB of LDS); larger panels use the kernel above.
constexpr int CW2 = 2;

template <int CTRL>
__device__ __forceinline__ double pq_dpp(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ double pq_lane(double x, int l)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return r.d;
}
__device__ __forceinline__ double pq_wave_sum(double x)
{
    x += pq_dpp<0x111>(x); x += pq_dpp<0x112>(x); x += pq_dpp<0x114>(x); x += pq_dpp<0x118>(x);   // row_shr 1, 2, 4, 8
    return (pq_lane(x, 15) + pq_lane(x, 31)) + (pq_lane(x, 47) + pq_lane(x, 63));
}
__device__ __forceinline__ void pq_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NV>
__device__ __forceinline__ void block_allreduce2(double (&v)[NV], double *red, int lane, int wave)
{
    constexpr int NW = PQ_THREADS / 64;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = pq_wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    pq_barrier();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w * NV + i];
        v[i] = s;
        // two values at a time: hoisting all NW * NV LDS reads to the front costs 2 NW NV registers and sets the
        // kernel's register count
        if ((i & 1) == 1) asm volatile("" : "+v"(v[i]), "+v"(v[i - 1])::"memory");
    }
    pq_barrier();
}

template <int RPT>
__global__ __launch_bounds__(PQ_THREADS) void panel_qr2_kernel(int npad, int r0, int c0, double *Aall,
                                                              double *bufall, double *tauall)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];          // [2][CW2][MP] + 128 (sink of idle LDS-DMA)
    constexpr int MP = PQ_THREADS * RPT;                                  // row capacity
    __shared__ double red[(PQ_THREADS / 64) * SW * SW];
    __shared__ double sh_alpha;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t ch = blockIdx.x;
    const long ld = npad;
    const int m = npad - r0;
    double *P = Aall + ch * (size_t)npad * npad + (size_t)c0 * ld + r0;   // P(i,j) = P[i + j*ld]
    double *buf = bufall + ch * (size_t)npad * 3 * NB;                    // buf(i,c) = buf[i + c*npad]
    double *tau_g = tauall + ch * NB;
    double *sink = xs + 2 * CW2 * MP;

    // LDS-DMA of columns cs, cs+1 (rows 0 .. MP-1 in chunks of 128 = 1 KB per wave instruction) into buffer b.
    // Every wave issues exactly RPT instructions per group (chunk k = wave + 8 it of CW2 * 4 RPT), so the
    // wait below can be a literal; chunks that start beyond m are fetched from row 0 into the sink.
    auto prefetch = [&](int cs, int b) {
#pragma unroll
        for (int it = 0; it < RPT; ++it) {
            const int k = wave + (PQ_THREADS / 64) * it;                  // 0 .. CW2 * 4 * RPT - 1
            const int c = k / (4 * RPT), rb = (k % (4 * RPT)) * 128;
            const bool live = rb < m;
            const int row = rb + 2 * lane;                                // m is even: a lane's pair is in or out
            const double *src = P + (size_t)(cs + c) * ld + ((live && row < m) ? row : 0);
            double *dst = live ? (xs + ((size_t)b * CW2 + c) * MP + rb) : sink;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };

    // RPT <= 4: the next sub-panel (the first two groups of the trailing update) stays in registers; at RPT = 8 the
    // second register set does not fit (spills put scratch loads, i.e. vmcnt waits, into every pass): those columns
    // go through memory like the others and each thread re-reads the rows it wrote itself.
    constexpr bool KEEP = (RPT <= 4);
    double a[RPT][SW], an[KEEP ? RPT : 1][SW];
#pragma unroll 1
    for (int js = 0; js < NB; js += SW) {
        const int ngroups = (NB - js - SW) / CW2;
        if (js == 0 || !KEEP) {
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
#pragma unroll
                for (int jj = 0; jj < SW; ++jj) a[q][jj] = (i < m) ? P[i + (size_t)(js + jj) * ld] : 0.0;
            }
        }
        if (ngroups > 0) prefetch(js + SW, 0);            // in flight during the factorisation of the sub-panel
        double taus[SW];
        // ---- factor the sub-panel (registers only) ----
#pragma unroll
        for (int jj = 0; jj < SW; ++jj) {
            const int j = js + jj;                       // pivot row (panel-relative)
            double part[1] = {0.0};
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
                if (i > j && i < m) part[0] += a[q][jj] * a[q][jj];
                if (i == j) sh_alpha = a[q][jj];
            }
            block_allreduce2<1>(part, red, lane, wave);  // barriers inside publish sh_alpha too
            const double alpha = sh_alpha, sigma = part[0];
            double beta, tau, scale;
            if (j >= m - 1 || !(alpha * alpha + sigma > 1e-280) || sigma == 0.0) {
                beta = alpha; tau = 0.0; scale = 0.0;    // nothing (numerically) below the pivot: H = I
            } else {
                const double nrm = sqrt(alpha * alpha + sigma);
                beta = (alpha >= 0.0) ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            taus[jj] = tau;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
                if (i > j) a[q][jj] *= scale;            // v_i
                if (i == j) a[q][jj] = beta;             // R(j,j)
            }
            if (jj < SW - 1) {                           // apply H_j to the remaining columns of the sub-panel
                double w[SW - 1];
#pragma unroll
                for (int c = 0; c < SW - 1; ++c) w[c] = 0.0;
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int i = tid + PQ_THREADS * q;
                    const double vv = (i > j) ? a[q][jj] : ((i == j) ? 1.0 : 0.0);
#pragma unroll
                    for (int c = jj + 1; c < SW; ++c) w[c - 1] += vv * a[q][c];
                }
                block_allreduce2<SW - 1>(w, red, lane, wave);
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int i = tid + PQ_THREADS * q;
                    const double vv = (i > j) ? a[q][jj] : ((i == j) ? 1.0 : 0.0);
#pragma unroll
                    for (int c = jj + 1; c < SW; ++c) a[q][c] -= tau * vv * w[c - 1];
                }
            }
        }
        // ---- T_s of the sub-panel from G_s = V_s^T V_s (strict upper triangle: 6 values) ----
        double g[6];                                     // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
#pragma unroll
        for (int x = 0; x < 6; ++x) g[x] = 0.0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int i = tid + PQ_THREADS * q;
            double vv[SW];
#pragma unroll
            for (int c = 0; c < SW; ++c) vv[c] = (i > js + c) ? a[q][c] : ((i == js + c) ? 1.0 : 0.0);
            g[0] += vv[0] * vv[1]; g[1] += vv[0] * vv[2]; g[2] += vv[0] * vv[3];
            g[3] += vv[1] * vv[2]; g[4] += vv[1] * vv[3]; g[5] += vv[2] * vv[3];
        }
        block_allreduce2<6>(g, red, lane, wave);
        // T (upper triangular): T(y,y) = tau_y, T(0:y, y) = -tau_y T(0:y,0:y) G(0:y, y)
        double Ts[SW][SW];
#pragma unroll
        for (int x = 0; x < SW; ++x)
#pragma unroll
            for (int y = 0; y < SW; ++y) Ts[x][y] = 0.0;
        Ts[0][0] = taus[0]; Ts[1][1] = taus[1]; Ts[2][2] = taus[2]; Ts[3][3] = taus[3];
        Ts[0][1] = -taus[1] * (Ts[0][0] * g[0]);
        Ts[0][2] = -taus[2] * (Ts[0][0] * g[1] + Ts[0][1] * g[3]);
        Ts[1][2] = -taus[2] * (Ts[1][1] * g[3]);
        Ts[0][3] = -taus[3] * (Ts[0][0] * g[2] + Ts[0][1] * g[4] + Ts[0][2] * g[5]);
        Ts[1][3] = -taus[3] * (Ts[1][1] * g[4] + Ts[1][2] * g[5]);
        Ts[2][3] = -taus[3] * (Ts[2][2] * g[5]);
        // ---- write R / zeros back to the panel, V (explicit) to buf, tau; a becomes explicit V ----
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int i = tid + PQ_THREADS * q;
#pragma unroll
            for (int jj = 0; jj < SW; ++jj) {
                const int j = js + jj;
                const double vv = (i > j) ? a[q][jj] : ((i == j) ? 1.0 : 0.0);
                if (i < m) {
                    P[i + (size_t)j * ld] = (i <= j) ? a[q][jj] : 0.0;
                    buf[i + (size_t)j * npad] = vv;
                    buf[i + (size_t)(2 * NB + j) * npad] = vv;
                }
                a[q][jj] = (i < m) ? vv : 0.0;
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int jj = 0; jj < SW; ++jj) tau_g[js + jj] = taus[jj];
        }
        // ---- apply (I - V_s T_s^T V_s^T) to the remaining panel columns, CW2 at a time, from LDS ----
#pragma unroll 1
        for (int gi = 0; gi < ngroups; ++gi) {
            const int cs = js + SW + CW2 * gi, b = gi & 1;
            if (gi + 1 < ngroups) {
                prefetch(cs + CW2, b ^ 1);
                // everything this wave issued before those RPT loads -- group gi's LDS-DMA, older stores -- is done
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPT) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pq_barrier();                                  // ... for every wave: buffer b is complete
            const double *X = xs + (size_t)b * CW2 * MP;
            double wp[SW * CW2];
#pragma unroll
            for (int z = 0; z < SW * CW2; ++z) wp[z] = 0.0;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
#pragma unroll
                for (int cc = 0; cc < CW2; ++cc) {
                    const double x = (i < m) ? X[cc * MP + i] : 0.0;
#pragma unroll
                    for (int c = 0; c < SW; ++c) wp[c * CW2 + cc] += a[q][c] * x;
                }
            }
            block_allreduce2<SW * CW2>(wp, red, lane, wave);
            // w2 = T_s^T wp, in place (row c uses rows <= c: go down from the last)
#pragma unroll
            for (int c = SW - 1; c >= 0; --c)
#pragma unroll
                for (int cc = 0; cc < CW2; ++cc) {
                    double s = 0.0;
#pragma unroll
                    for (int z = 0; z <= c; ++z) s += Ts[z][c] * wp[z * CW2 + cc];
                    wp[c * CW2 + cc] = s;
                }
            // X -= V_s w2: the first two groups are the next sub-panel and stay in registers, the rest go to memory
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = tid + PQ_THREADS * q;
#pragma unroll
                for (int cc = 0; cc < CW2; ++cc) {
                    double x = (i < m) ? X[cc * MP + i] : 0.0;
#pragma unroll
                    for (int c = 0; c < SW; ++c) x -= a[q][c] * wp[c * CW2 + cc];
                    if (KEEP && gi == 0) an[q][cc] = x;
                    else if (KEEP && gi == 1) an[q][CW2 + cc] = x;
                    else if (i < m) P[i + (size_t)(cs + cc) * ld] = x;
                }
            }
            pq_barrier();                                  // buffer b may be overwritten by the prefetch after next
        }
        if (KEEP) {
#pragma unroll
            for (int q = 0; q < RPT; ++q)
#pragma unroll
                for (int jj = 0; jj < SW; ++jj) a[q][jj] = an[q][jj];
        }
    }
}

// T (NB x NB, upper triangular, column-major) from G = V^T V and tau: T(j,j) = tau_j,
// T(0:j, j) = -tau_j T(0:j,0:j) G(0:j, j)   (LAPACK dlarft, forward / columnwise).
__global__ __launch_bounds__(64) void form_T_kernel(const double *__restrict__ Gall,
                                                   const double *__restrict__ tauall, double *Tall)
{
    // Thread i owns row i of T.  Column j needs s_i = sum_{p=i}^{j-1} T(i,p) G(p,j); T is upper triangular and its
    // columns >= j are still zero, so the sum may run over ALL p: a fixed trip count, four independent accumulators
    // and 16-byte LDS reads (row i of T and row j of G^T are contiguous) instead of a dependent chain of single reads
    // with data-dependent bounds (the first version: 200 us per panel, on the critical chain of the stage).  Unrolling is
    // limited on purpose: fully unrolled the kernel needed 268 registers, i.e. a completely empty SIMD, and waited
    // for one behind the other pipelines' GEMM workgroups (82 now: it fits the place one GEMM workgroup leaves).
    __shared__ __attribute__((aligned(16))) double T[NB][NB + 2];
    __shared__ __attribute__((aligned(16))) double Gt[NB][NB + 2];       // Gt[j][p] = G(p, j)
    const int i = threadIdx.x;
    const size_t ch = blockIdx.x;
    const double *Gg = Gall + ch * NB * NB;
    const double *tau = tauall + ch * NB;
    double *Tg = Tall + ch * NB * NB;
#pragma unroll 8
    for (int j = 0; j < NB; ++j) { Gt[j][i] = Gg[i * NB + j]; T[i][j] = 0.0; }
    const double tau_l = tau[i];                           // tau_j is broadcast from lane j: no memory access in the column loop
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < NB; ++j) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
        for (int p = 0; p < NB; p += 4) {
            const double2 t01 = *reinterpret_cast<const double2 *>(&T[i][p]);
            const double2 t23 = *reinterpret_cast<const double2 *>(&T[i][p + 2]);
            const double2 g01 = *reinterpret_cast<const double2 *>(&Gt[j][p]);
            const double2 g23 = *reinterpret_cast<const double2 *>(&Gt[j][p + 2]);
            s0 += t01.x * g01.x; s1 += t01.y * g01.y; s2 += t23.x * g23.x; s3 += t23.y * g23.y;
        }
        const double tj = __shfl(tau_l, j);
        __syncthreads();
        if (i < j) T[i][j] = -tj * ((s0 + s1) + (s2 + s3));
        if (i == j) T[i][j] = tj;
        __syncthreads();
    }
#pragma unroll 8
    for (int j = 0; j < NB; ++j) Tg[i + j * NB] = T[i][j];
}

__global__ void extract_band_kernel(int npad, const double *__restrict__ Aall, double *__restrict__ ABall)
{
    // AB[d + j*2NB] = A(j+d, j), d = 0..NB ; zero for d > NB or j+d >= npad
    const size_t ch = blockIdx.y;
    const double *A = Aall + ch * (size_t)npad * npad;
    double *AB = ABall + ch * ab_stride(npad);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npad * 2 * NB) return;
    const int d = idx % (2 * NB), j = idx / (2 * NB);
    double v = 0.0;
    if (d <= NB && j + d < npad) v = A[(size_t)j * npad + j + d];
    AB[idx] = v;
}

size_t sy2sb_work_bytes(int npad, int nb, int batch)
{
    (void)nb;
    size_t per = 2 * ((size_t)npad * 3 * NB + NB) + (size_t)npad * NB + (3 + SY2SB_SPLITK) * NB * NB;
    per += (size_t)tsqr_scr_doubles(npad) + ((size_t)tsqr_cntr_ints(npad) + 1) / 2;
    return per * batch * sizeof(double);
}

void sy2sb_carve(void *base, int npad, int nb, int batch, Sy2sbWork *w)
{
    (void)nb;
    double *p = static_cast<double *>(base);
    w->buf = p; p += (size_t)batch * npad * 3 * NB;
    w->buf2 = p; p += (size_t)batch * npad * 3 * NB;
    w->W = p; p += (size_t)batch * npad * NB;
    w->G = p; p += (size_t)batch * NB * NB;
    w->T = p; p += (size_t)batch * NB * NB;
    w->Kmat = p; p += (size_t)batch * NB * NB;
    w->tau = p; p += (size_t)batch * NB;
    w->tau2 = p; p += (size_t)batch * NB;
    w->part = p; p += (size_t)batch * SY2SB_SPLITK * NB * NB;
    w->tsqr_scr = p; p += (size_t)batch * tsqr_scr_doubles(npad);
    w->tsqr_cntr = reinterpret_cast<int *>(p);
}

template <int RPT>
static void launch_pq(int npad, int r0, int c0, int batch, double *A, double *buf, double *tau, hipStream_t st)
{
    const int use2 = opts().panel_qr >= 2;
    KScope kt(KS_PANEL_QR, st);
    if (use2 && RPT <= 8) {
        const size_t lds = (size_t)(2 * CW2 * PQ_THREADS * RPT + 128) * sizeof(double);
        static bool attr[17] = {};
        if (!attr[RPT]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(panel_qr2_kernel<RPT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr[RPT] = true;
        }
        hipLaunchKernelGGL((panel_qr2_kernel<RPT>), dim3(batch), dim3(PQ_THREADS), lds, st, npad, r0, c0, A, buf, tau);
        return;
    }
    hipLaunchKernelGGL((panel_qr_kernel<RPT>), dim3(batch), dim3(PQ_THREADS), 0, st, npad, r0, c0, A, buf, tau);
}

// Panel factorisation + T + W = V T for the panel whose columns start at c0 (enqueued on `s`).
static int panel_and_W(int npad, int c0, int batch, double *d_A, double *buf, double *tau, const Sy2sbWork &w,
                       hipStream_t s)
{
    const int r0 = c0 + NB, m = npad - r0;
    const long bsBuf = (long)npad * 3 * NB, bsW = (long)npad * NB, bsS = NB * NB;
    (void)tau;
    if (opts().panel_qr >= 3 && (opts().tsqr_max_m <= 0 || m <= opts().tsqr_max_m || m > PQ_THREADS * 16))
        return tsqr_panel(npad, r0, c0, batch, d_A, buf, w.W, w.tsqr_scr, w.tsqr_cntr, s);
    if (m > PQ_THREADS * 16) return BSP_ERR_UNSUPPORTED;              // one workgroup holds 16 rows per thread: n <= 8256
    const int rpt = (m + PQ_THREADS - 1) / PQ_THREADS;
    if (rpt <= 1) launch_pq<1>(npad, r0, c0, batch, d_A, buf, tau, s);
    else if (rpt <= 2) launch_pq<2>(npad, r0, c0, batch, d_A, buf, tau, s);
    else if (rpt <= 4) launch_pq<4>(npad, r0, c0, batch, d_A, buf, tau, s);
    else if (rpt <= 8) launch_pq<8>(npad, r0, c0, batch, d_A, buf, tau, s);
    else launch_pq<16>(npad, r0, c0, batch, d_A, buf, tau, s);
    BSP_HIP(hipGetLastError());
    int rc;
    KScope kt(KS_CHAIN, s);
    GemmDesc g{};
    g.batch = batch;
    // G = V^T V
    g.M = NB; g.N = NB; g.K = m;
    g.A = buf; g.sAm = npad; g.sAk = 1; g.bA = bsBuf;
    g.B = buf; g.sBk = 1; g.sBn = npad; g.bB = bsBuf;
    g.C = w.G; g.sCm = NB; g.sCn = 1; g.bC = bsS; g.alpha = 1.0; g.beta = 0.0;
    if ((rc = gemm_splitk_f64(g, SY2SB_SPLITK, w.part, s))) return rc;   // 64 x 64 x m: one tile per channel
    hipLaunchKernelGGL(form_T_kernel, dim3(batch), dim3(64), 0, s, w.G, tau, w.T);
    BSP_HIP(hipGetLastError());
    // W = V T   (m x 64 times 64 x 64)
    return tsmm64_f64(m, batch, buf, npad, bsBuf, w.T, bsS, w.W, npad, bsW, 1.0, 0.0, s);
}

// stage-level entry (tests): the panel factorisation alone, for the panel whose columns start at c0
int sy2sb_panel_only(int npad, int c0, int batch, double *d_A, const Sy2sbWork &w, hipStream_t st)
{
    if (npad % NB || c0 % NB || c0 + 2 * NB > npad) return BSP_ERR_ARG;
    if (opts().panel_qr >= 3) BSP_HIP(hipMemsetAsync(w.tsqr_cntr, 0, (size_t)batch * tsqr_cntr_ints(npad) * sizeof(int), st));
    return panel_and_W(npad, c0, batch, d_A, w.buf, w.tau, w, st);
}

// Look-ahead schedule: after the block column that holds the next panel has been updated (syr2k
// part 1), QR(p+1), T and W(p+1) run on a side stream while the rest of the HBM-bound update
// (part 2) proceeds on the main stream; the two [V|Z|V] buffers alternate between panels.
struct Sy2sbLane {                       // one pipeline: main + high-priority side stream and their events
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t evA = nullptr, evB = nullptr, done = nullptr;
};

// one panel step of one pipeline (p = -1: the first panel's QR, T, W)
static int sy2sb_panel(int npad, int batch, double *d_A, const Sy2sbWork &w, hipStream_t st, const Sy2sbLane &ln,
                       int lookahead, int p)
{
    hipStream_t side = ln.side;
    hipEvent_t evA = ln.evA, evB = ln.evB;
    const long ld = npad;
    const long bsA = (long)npad * npad, bsBuf = (long)npad * 3 * NB, bsW = (long)npad * NB, bsS = NB * NB;
    const int P = npad / NB - 1;
    int rc;
    if (P <= 0) return BSP_OK;
    if (p < 0) return panel_and_W(npad, 0, batch, d_A, w.buf, w.tau, w, st);
    {
        const int c0 = p * NB, r0 = c0 + NB, m = npad - r0;
        double *buf = (p & 1) ? w.buf2 : w.buf;
        double *bufn = (p & 1) ? w.buf : w.buf2;
        double *taun = (p & 1) ? w.tau : w.tau2;
        double *A22 = d_A + (size_t)r0 * ld + r0;
        GemmDesc g{};
        g.batch = batch;
        // Y = A22 W  -> buf[:, NB:2NB]   (A22 valid on 64-blocks J <= I+1 only)
        if (!opts().fused_probe &&
            (rc = symm_lower_f64(m, batch, A22, ld, bsA, w.W, npad, bsW, buf + (size_t)NB * npad, npad, bsBuf, st))) return rc;
        // K = W^T Y
        {
            KScope kt(KS_CHAIN, st);
            g.M = NB; g.N = NB; g.K = m;
            g.A = w.W; g.sAm = npad; g.sAk = 1; g.bA = bsW;
            g.B = buf + (size_t)NB * npad; g.sBk = 1; g.sBn = npad; g.bB = bsBuf;
            g.C = w.Kmat; g.sCm = 1; g.sCn = NB; g.bC = bsS; g.alpha = 1.0; g.beta = 0.0;
            if ((rc = gemm_splitk_f64(g, SY2SB_SPLITK, w.part, st))) return rc;
            // Z = Y - 1/2 V K  (in place; m x 64 times 64 x 64)
            if ((rc = tsmm64_f64(m, batch, buf, npad, bsBuf, w.Kmat, bsS, buf + (size_t)NB * npad, npad, bsBuf, -0.5, 1.0, st))) return rc;
        }
        // A22 -= [V Z] [Z V]^T
        const bool more = (p + 1 < P);
        if (lookahead && more) {
            if ((rc = syr2k_lower_f64(m, batch, A22, ld, bsA, buf, npad, bsBuf, 0, 1, 1, st))) return rc;
            BSP_HIP(hipEventRecord(evA, st));
            BSP_HIP(hipStreamWaitEvent(side, evA, 0));
            if ((rc = panel_and_W(npad, r0, batch, d_A, bufn, taun, w, side))) return rc;
            BSP_HIP(hipEventRecord(evB, side));
            // the rest of the update in `nseg2` launches: at every boundary between them the running workgroups
            // drain, which is when the panel stream's small kernels (T, W ...) find a place (BSP_SY2SB_SEGS)
            // (measured at 128 channels, n = 4096: 1 launch 307-308 ms, 2: 305, 3: 302-303, 4: 303-304, 6: 306, 8: 307;
            // small trailing matrices -- fewer than six fills of the GPU -- stay in one launch)
            const int nseg_env = opts().sy2sb_segs;
            const long nbt = (m + 127) / 128, wgs = nbt * (nbt + 1) / 2 * batch;
            const int nseg2 = (nseg_env >= 1) ? nseg_env : ((wgs >= 6 * 512) ? 3 : 1);
            for (int sg = 0; sg < nseg2; ++sg)
                if ((rc = syr2k_lower_f64(m, batch, A22, ld, bsA, buf, npad, bsBuf, sg, nseg2, 2, st))) return rc;
            BSP_HIP(hipStreamWaitEvent(st, evB, 0));
        } else {
            if ((rc = syr2k_lower_f64(m, batch, A22, ld, bsA, buf, npad, bsBuf, 0, 1, 0, st))) return rc;
            if (more && (rc = panel_and_W(npad, r0, batch, d_A, bufn, taun, w, st))) return rc;
        }
    }
    return BSP_OK;
}

static int sy2sb_pipeline(int npad, int batch, double *d_A, const Sy2sbWork &w, hipStream_t st, const Sy2sbLane &ln,
                          int lookahead)
{
    const int P = npad / NB - 1;
    int rc;
    for (int p = -1; p < P; ++p)
        if ((rc = sy2sb_panel(npad, batch, d_A, w, st, ln, lookahead, p))) return rc;
    return BSP_OK;
}

// The per-panel chain QR -> T, W -> SYMM -> K, Z -> SYR2K part 1 -> next QR is serial within a channel, and the
// panel QR (one workgroup per channel) cannot fill the chip.  The channels are therefore split into groups that run
// the whole pipeline independently on their own streams: while one group factors a panel the other keeps the
// chip busy with its GEMMs.  BSP_SY2SB_GROUPS (default 2; 1 = a single pipeline on the caller's stream).
int sy2sb_run(int npad, int nb, int batch, double *d_A, const Sy2sbWork &w, hipStream_t st)
{
    if (nb != NB || npad % NB != 0) return BSP_ERR_ARG;
    if (opts().panel_qr < 3 && npad - NB > PQ_THREADS * 16) return BSP_ERR_UNSUPPORTED;   // the one-workgroup panel kernels: n <= 8256
    if (opts().panel_qr >= 3) BSP_HIP(hipMemsetAsync(w.tsqr_cntr, 0, (size_t)batch * tsqr_cntr_ints(npad) * sizeof(int), st));
    constexpr int MAXG = 4;
    static Sy2sbLane lanes[MAXG];
    static hipEvent_t fork = nullptr;
    static bool lanes_ready = false;
    const int lookahead = opts().sy2sb_lookahead;
    int groups = opts().sy2sb_groups;
    if (groups < 1) groups = 1;
    if (groups > MAXG) groups = MAXG;
    if (!lanes_ready) {
        lanes_ready = true;
        int plo = 0, phi = 0;                      // the latency-bound panel work gets the high-priority queue
        BSP_HIP(hipDeviceGetStreamPriorityRange(&plo, &phi));
        // (Giving the panel streams compute units of their own through CU masks -- a panel-QR workgroup needs a whole
        // CU and waits between GEMM workgroups that take half a CU each -- was tried: 32/64/96 reserved CUs cost
        // 20-37 % of the stage; the GEMMs need every CU's memory pipeline.)
        for (int g = 0; g < MAXG; ++g) {
            BSP_HIP(hipStreamCreateWithFlags(&lanes[g].main, hipStreamNonBlocking));
            BSP_HIP(hipStreamCreateWithPriority(&lanes[g].side, hipStreamNonBlocking, phi));
            BSP_HIP(hipEventCreateWithFlags(&lanes[g].evA, hipEventDisableTiming));
            BSP_HIP(hipEventCreateWithFlags(&lanes[g].evB, hipEventDisableTiming));
            BSP_HIP(hipEventCreateWithFlags(&lanes[g].done, hipEventDisableTiming));
        }
        BSP_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    }
    int ng = groups;
    if (batch < 16 * ng) ng = 1;                   // small batches: one pipeline
    if (ng == 1) return sy2sb_pipeline(npad, batch, d_A, w, st, lanes[0], lookahead);
    BSP_HIP(hipEventRecord(fork, st));
    const size_t bsA = (size_t)npad * npad, bsBuf = (size_t)npad * 3 * NB, bsW = (size_t)npad * NB, bsS = (size_t)NB * NB;
    int rc = BSP_OK, c0 = 0;
    Sy2sbWork wg[MAXG];
    int cnt[MAXG], first[MAXG];
    for (int g = 0; g < ng; ++g) {
        cnt[g] = batch / ng + (g < batch % ng ? 1 : 0);
        first[g] = c0;
        wg[g] = w;
        wg[g].buf = w.buf + c0 * bsBuf; wg[g].buf2 = w.buf2 + c0 * bsBuf; wg[g].W = w.W + c0 * bsW;
        wg[g].G = w.G + c0 * bsS; wg[g].T = w.T + c0 * bsS; wg[g].Kmat = w.Kmat + c0 * bsS;
        wg[g].tau = w.tau + (size_t)c0 * NB; wg[g].tau2 = w.tau2 + (size_t)c0 * NB;
        wg[g].part = w.part + (size_t)c0 * SY2SB_SPLITK * bsS;      // [splits][cnt][nb][nb] inside this group's share
        wg[g].tsqr_scr = w.tsqr_scr + (size_t)c0 * tsqr_scr_doubles(npad);
        wg[g].tsqr_cntr = w.tsqr_cntr + (size_t)c0 * tsqr_cntr_ints(npad);
        BSP_HIP(hipStreamWaitEvent(lanes[g].main, fork, 0));
        c0 += cnt[g];
    }
    // enqueue panel by panel, the groups alternating: the host needs ~12 ms to enqueue one group's ~800 launches,
    // and a group whose launches all come after the other's starts (and ends) that much later
    const int P = npad / NB - 1;
    for (int p = -1; p < P; ++p)
        for (int g = 0; g < ng; ++g)
            if ((rc = sy2sb_panel(npad, cnt[g], d_A + first[g] * bsA, wg[g], lanes[g].main, lanes[g], lookahead, p))) return rc;
    for (int g = 0; g < ng; ++g) {
        BSP_HIP(hipEventRecord(lanes[g].done, lanes[g].main));
        BSP_HIP(hipStreamWaitEvent(st, lanes[g].done, 0));
    }
    return BSP_OK;
}

int launch_extract_band(int npad, int nb, int batch, const double *d_A, double *d_AB, hipStream_t st)
{
    if (nb != NB) return BSP_ERR_ARG;
    const int total = npad * 2 * NB;
    hipLaunchKernelGGL(extract_band_kernel, dim3((total + 255) / 256, batch), dim3(256), 0, st, npad, d_A, d_AB);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
