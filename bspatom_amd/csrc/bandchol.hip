// bandchol.hip -- banded Cholesky of S and reduction of H(l) to standard form (dense C_l).
//
// Replaces DPOTRF('U') and DSYGST(1,'U') inside DSYGV (reference call matrices.f90:248):
//   S = U^T U   (upper band, half-width b = k-1),
//   C_l = U^-T H_l U^-1 = L^-1 H_l L^-T   with L = U^T,
// batched over the l-channels.  S and H(l) are banded, so both steps are banded triangular
// recurrences; C_l is dense (L^-1 is full) and is written once, column-major, ld = npad.
//
// Band layouts (upper): SB[d*n + i] = S(i,i+d); UB[d*n + i] = U(i,i+d); HB[(l*k+d)*n + i].
// npad = n rounded up to a multiple of 64: rows/cols >= n of C are zero (L is extended by the
// identity there), which keeps the later stages free of edge cases.
#include "common.h"

namespace bsp {

// ---- banded Cholesky, one wavefront; LDS ring of the last b rows of U ----------------------
// The recurrence is serial in the column index (n steps of ~b^2 flop): what a step costs is latency.  Round 4: the band entries
// of the next CH columns are requested a whole chunk ahead into registers (the first version waited for a global load in every
// step: 0.87 us per column, 3.6 ms at n = 4096 -- on the critical path of every solve, of both routes), the pivot goes round by
// readlane instead of a shuffle through LDS, the wave synchronises with LDS-only waits, and the sum over the previous rows has a
// fixed trip count with masked addresses (3.8 -> ms at n = 4096: see the loop).
// Workgroup g factors the band at SB + g * kn (kn = k n) into UB + g * kn, rdiag + g * n, rows 0 .. jstop - 1 only: the band route
// run from both ends (crawford.hip) wants the leading half of the factor of S and of the index-reversed S, side by side.
// BMAX = the largest half-width of the instance (8: every pencil of the band route; the sum over the previous rows has BMAX terms in
// a dependent chain, masked down to b, so the narrow instance halves the longest chain of a column; same terms, same order)
// Rows jstart .. jstop - 1: a launch may continue where another stopped (the rows j - b .. j - 1 it needs come back from UB: the same
// doubles), so that the band route can start on the first blocks while the rest of the factor is still being computed.
template <int BMAX>
__global__ __launch_bounds__(64) void band_cholesky_kernel(int n, int k, int jstart, int jstop, const double *__restrict__ SB0,
                                                          double *__restrict__ UB0,
                                                          double *__restrict__ rdiag0, int *info)
{
    constexpr int CH = 16, RING = 32;
    const double *SB = SB0 + (size_t)blockIdx.x * k * n;
    double *UB = UB0 + (size_t)blockIdx.x * k * n, *rdiag = rdiag0 + (size_t)blockIdx.x * n;
    __shared__ double ring[RING][BMAX + 1];       // ring[p & 31][d] = U(p, p+d) (rows j - b .. j - 1 are live: b <= 16 < 32)
    const int b = k - 1, t = threadIdx.x;
    int bad = 0;
    const bool row = t <= b;
    const double *Sb = SB + (size_t)(row ? t : 0) * n;
    for (int e = t; e < RING * (BMAX + 1); e += 64) {                          // rows before the matrix: zeros (no bounds tests below);
        const int slot = e / (BMAX + 1), d = e % (BMAX + 1);                   // rows before jstart: what the launch before left in UB
        const int pr = jstart - RING + ((slot - jstart) & (RING - 1));         // the row of [jstart - RING, jstart) that lives in this slot
        (&ring[0][0])[e] = (pr >= 0 && d <= b && pr + d < n) ? UB[(size_t)d * n + pr] : 0.0;
    }
    double cur[CH], nxt[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) cur[c] = (row && jstart + c < n) ? Sb[jstart + c] : 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int j0 = jstart; j0 < jstop; j0 += CH) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { const int j = j0 + CH + c; nxt[c] = (row && j < n) ? Sb[j] : 0.0; }   // in flight during this chunk
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = j0 + c;
            if (j >= jstop) break;                                       // uniform
            const bool act = row && (j + t < n);
            // S(j, j+t) - sum_{q=1..b} U(j-q, j) U(j-q, j+t): a fixed trip count, addresses by masks (the first version took a
            // run-time modulo per term: ~40 instructions in front of every dependent LDS read, 0.9 us per column); terms beyond
            // the band (q + t > b) are masked, rows before the matrix read the zeros above
            double av[BMAX], bv[BMAX];
#pragma unroll
            for (int q = 1; q <= BMAX; ++q) {                            // all reads first: they do not depend on the sum
                av[q - 1] = ring[(j - q) & (RING - 1)][q];
                bv[q - 1] = ring[(j - q) & (RING - 1)][(q + t) <= BMAX ? q + t : 0];
            }
            double s = act ? cur[c] : 0.0;
#pragma unroll
            for (int q = BMAX; q >= 1; --q)                              // rows j - b .. j - 1 in this order: the first version's sum, bit for bit
                if (q <= b) s = fma(-av[q - 1], (q + t <= b) ? bv[q - 1] : 0.0, s);
            // lane 0 holds the pivot
            union { double d; int i[2]; } u, r;
            u.d = s;
            r.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
            r.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
            const double piv = r.d;
            if (!(piv > 0.0) && bad == 0) { bad = 1; if (t == 0 && *info == 0) *info = j + 1; }   // minor j+1 not PD (the first one, over all launches)
            const double dj = sqrt(piv);
            double uv = (t == 0) ? dj : s / dj;
            if (t <= BMAX) {
                if (!act) uv = 0.0;
                ring[j & (RING - 1)][t] = uv;                            // row j - 32 left the band long ago
                if (row) UB[(size_t)t * n + j] = uv;
            }
            if (t == 0) rdiag[j] = 1.0 / dj;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // one wave: its LDS operations complete in order
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) cur[c] = nxt[c];
    }
}

// ---- standard form: two banded forward substitutions along the column index ----------------
// PASS 1: Y(r,j) = (H(r,j) - sum_{p=1..B} L(j,j-p) Y(r,j-p)) / L(j,j)       (Y = H L^-T)
//         stored so that Y(j,r) sits at j*ld + r, i.e. row r of Y is contiguous at r*ld.
// PASS 2: C(r,j) = (Y(j,r) - sum_p L(j,j-p) C(r,j-p)) / L(j,j)  for j <= r   (C = (L^-1 Y)^T)
//         stored column-major C(r,j) at j*ld + r, and mirrored C(j,r) = C(r,j).
// One wavefront per 64 rows; a 64x64 result tile is staged in LDS for the transposed store.
template <int B, int PASS>
__global__ __launch_bounds__(64) void std_form_kernel(int n, int npad, int k,
                                                     const double *__restrict__ HB,
                                                     const double *__restrict__ UB,
                                                     const double *__restrict__ rdiag,
                                                     const double *__restrict__ Yin,
                                                     double *__restrict__ out, int full)
{
    __shared__ double tile[64][65];
    __shared__ double Lc[64][B + 1];   // Lc[jj][p] = L(j, j-p), p = 1..B ; Lc[jj][0] = 1/L(j,j)
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 64, r = r0 + lane;
    const size_t ch = blockIdx.y;
    const long ld = npad;
    const double *H = HB + ch * (size_t)k * n;
    const double *Y = Yin + ch * (size_t)npad * npad;
    double *O = out + ch * (size_t)npad * npad;

    double prev[B];
#pragma unroll
    for (int p = 0; p < B; ++p) prev[p] = 0.0;

    int jbeg, jend;
    if (PASS == 1) { jbeg = (r0 >= 64) ? (r0 - 64) : 0; jend = npad; }
    else { jbeg = 0; jend = r0 + 64; }

    // PASS 2 reads a 64 x 64 block of Y per step, one column (512 B) per load.  Left where they are used, the loads
    // are waited for one by one inside the dependent column recurrence (measured: 75 us per block, 1.4 TB/s); the
    // next block's 64 columns are therefore fetched into a second register set while this block is processed
    // (the workgroup is one wavefront and LDS already limits the CU to four of them, so the registers are free).
    double cur[64], nxt[64];
    if (PASS == 2) {
#pragma unroll
        for (int jj = 0; jj < 64; ++jj) cur[jj] = __builtin_nontemporal_load(&Y[(size_t)(jbeg + jj) * ld + r]);
    }
    for (int j0 = jbeg; j0 < jend; j0 += 64) {
        if (PASS == 2) {
            const bool more = (j0 + 64 < jend);                // wave-uniform; clamped address, no branch around loads
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) nxt[jj] = __builtin_nontemporal_load(&Y[(size_t)(more ? (j0 + 64 + jj) : jbeg) * ld + r]);
        }
        __syncthreads();
        for (int idx = lane; idx < 64 * (B + 1); idx += 64) {
            const int jj = idx / (B + 1), p = idx % (B + 1), j = j0 + jj;
            double v;
            if (j < n) v = (p == 0) ? rdiag[j] : ((j - p >= 0) ? UB[(size_t)p * n + (j - p)] : 0.0);
            else v = (p == 0) ? 1.0 : 0.0;
            Lc[jj][p] = v;
        }
        __syncthreads();
        double v[64];
#pragma unroll
        for (int jj = 0; jj < 64; ++jj) {
            const int j = j0 + jj;
            double acc;
            if (PASS == 1) {
                acc = 0.0;
                const int d = (r > j) ? (r - j) : (j - r);
                const int lo = (r > j) ? j : r;
                if (d <= B && r < n && j < n) acc = H[(size_t)d * n + lo];
            } else {
                acc = cur[jj];
            }
#pragma unroll
            for (int p = 1; p <= B; ++p) {
                const double yp = (p <= jj) ? v[jj - p] : prev[p - jj - 1];
                acc -= Lc[jj][p] * yp;
            }
            v[jj] = acc * Lc[jj][0];
            tile[lane][jj] = v[jj];
            if (PASS == 2) {
                if (j <= r) __builtin_nontemporal_store(v[jj], &O[(size_t)j * ld + r]);
            }
        }
#pragma unroll
        for (int p = 0; p < B; ++p) prev[p] = v[63 - p];          // prev[p] = value at j0+64-(p+1)
        if (PASS == 2) {
#pragma unroll
            for (int jj = 0; jj < 64; ++jj) cur[jj] = nxt[jj];
        }
        __syncthreads();
        // transposed store: element (r0+rr, j0+lane) of the pass result goes to (r0+rr)*ld + j0+lane
        if (PASS == 1) {
            for (int rr = 0; rr < 64; ++rr) __builtin_nontemporal_store(tile[rr][lane], &O[(size_t)(r0 + rr) * ld + j0 + lane]);
        } else if (full || j0 + 64 >= r0) {
            // the mirror image (upper triangle).  The reduction that follows keeps only the 64-blocks (I, J) with
            // J <= I + 1 valid and never reads beyond them (sy2sb.hip, gemm_f64.hip MODE 1 / 2), so the pipeline asks
            // for the first block super-diagonal only: 12 n^2 instead of 16 n^2 bytes per channel.  `full` (stage-level
            // entry point) mirrors everything.
            for (int rr = 0; rr < 64; ++rr)
                if (j0 + lane < r0 + rr) __builtin_nontemporal_store(tile[rr][lane], &O[(size_t)(r0 + rr) * ld + j0 + lane]);
        }
    }
}

int launch_band_cholesky(int n, int k, const double *d_SB, double *d_UB, double *d_rdiag, int *d_info,
                         hipStream_t st)
{
    if (k - 1 > 16 || k < 2) return BSP_ERR_ARG;
    if (k - 1 <= 8) hipLaunchKernelGGL(band_cholesky_kernel<8>, dim3(1), dim3(64), 0, st, n, k, 0, n, d_SB, d_UB, d_rdiag, d_info);
    else hipLaunchKernelGGL(band_cholesky_kernel<16>, dim3(1), dim3(64), 0, st, n, k, 0, n, d_SB, d_UB, d_rdiag, d_info);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// rows jstart .. jstop - 1, continuing a factorisation whose rows before jstart are in d_UB (jstart a multiple of 16)
int launch_band_cholesky_range(int n, int k, int jstart, int jstop, const double *d_SB, double *d_UB, double *d_rdiag, int *d_info,
                               hipStream_t st)
{
    if (k - 1 > 16 || k < 2 || jstart < 0 || jstart % 16 != 0 || jstop <= jstart || jstop > n) return BSP_ERR_ARG;
    if (k - 1 <= 8) hipLaunchKernelGGL(band_cholesky_kernel<8>, dim3(1), dim3(64), 0, st, n, k, jstart, jstop, d_SB, d_UB, d_rdiag, d_info);
    else hipLaunchKernelGGL(band_cholesky_kernel<16>, dim3(1), dim3(64), 0, st, n, k, jstart, jstop, d_SB, d_UB, d_rdiag, d_info);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// rows 0 .. jstop - 1 of the factors of the two bands at d_SB and d_SB + k n (see the kernel); info is shared: nonzero if either broke down
int launch_band_cholesky_pair(int n, int k, int jstop, const double *d_SB, double *d_UB, double *d_rdiag, int *d_info, hipStream_t st)
{
    if (k - 1 > 16 || k < 2 || jstop < 1 || jstop > n) return BSP_ERR_ARG;
    if (k - 1 <= 8) hipLaunchKernelGGL(band_cholesky_kernel<8>, dim3(2), dim3(64), 0, st, n, k, 0, jstop, d_SB, d_UB, d_rdiag, d_info);
    else hipLaunchKernelGGL(band_cholesky_kernel<16>, dim3(2), dim3(64), 0, st, n, k, 0, jstop, d_SB, d_UB, d_rdiag, d_info);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

template <int B>
static int launch_std_B(int n, int npad, int k, int nl, const double *d_HB, const double *d_UB,
                        const double *d_rdiag, double *d_Y, double *d_C, hipStream_t st, int full)
{
    dim3 grid(npad / 64, nl), block(64);
    hipLaunchKernelGGL((std_form_kernel<B, 1>), grid, block, 0, st, n, npad, k, d_HB, d_UB, d_rdiag,
                       (const double *)d_Y, d_Y, full);
    hipLaunchKernelGGL((std_form_kernel<B, 2>), grid, block, 0, st, n, npad, k, d_HB, d_UB, d_rdiag,
                       (const double *)d_Y, d_C, full);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

int launch_standard_form(int n, int npad, int k, int nl, const double *d_HB, const double *d_UB,
                         const double *d_rdiag, double *d_Y, double *d_C, hipStream_t st, int full)
{
    switch (k - 1) {
#define CASE_B(B) case B: return launch_std_B<B>(n, npad, k, nl, d_HB, d_UB, d_rdiag, d_Y, d_C, st, full);
        CASE_B(1) CASE_B(2) CASE_B(3) CASE_B(4) CASE_B(5) CASE_B(6) CASE_B(7) CASE_B(8)
        CASE_B(9) CASE_B(10) CASE_B(11) CASE_B(12) CASE_B(13) CASE_B(14) CASE_B(15)
#undef CASE_B
    default: return BSP_ERR_ARG;
    }
}

}  // namespace bsp
