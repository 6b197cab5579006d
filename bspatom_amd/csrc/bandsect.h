// bandsect.h -- device code of bandsect.hip (one eigenvalue of the banded pencil from inertia counts), shared with eigvec.hip, whose
// early_vector_kernel runs the inverse iteration in the same workgroup.  See bandsect.hip for what and why.
#pragma once
#include "common.h"

namespace bsp {
constexpr int BS_T = 256;       // shifts per round = threads
constexpr int BS_B = 8;         // half-width the window is built for (narrower bands: zeros)
constexpr int BS_ROWS = 64;     // rows staged per chunk

// One round: thread t counts the negative pivots of H - x S.  stage[2][BS_ROWS][BS_B + 1] holds (h, s) of A(r, r - d) at [r][d];
// rows r >= n are rows of the identity.
__device__ __forceinline__ int band_inertia(int n, int b, const double *__restrict__ SB, const double *__restrict__ HB, double x,
                                            double2 (*stage)[BS_ROWS][BS_B + 1])
{
    constexpr int B = BS_B;
    const int tid = threadIdx.x;
    double W[B + 1][B + 1];                                  // W[p][q] = the current A(j + p, j + q), q <= p; starts as the identity
#pragma unroll
    for (int p = 0; p <= B; ++p)
#pragma unroll
        for (int q = 0; q <= B; ++q) W[p][q] = (p == q) ? 1.0 : 0.0;
    int neg = 0;
    const int rows = n + B + 1;                              // the last pivot (column n - 1) is taken when row n + B enters
    const int nchunk = (rows + BS_ROWS - 1) / BS_ROWS;
    auto fetch = [&](int c, double2 (&v)[3]) {               // this thread's entries of chunk c
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int idx = tid + e * BS_T, rr = idx / (B + 1), d = idx - rr * (B + 1), r = c * BS_ROWS + rr;
            double2 val = make_double2(0.0, 0.0);
            if (idx < BS_ROWS * (B + 1)) {
                if (r >= n) val.x = (d == 0) ? 1.0 : 0.0;
                else if (d <= b && r - d >= 0) val = make_double2(HB[(size_t)d * n + (r - d)], SB[(size_t)d * n + (r - d)]);
            }
            v[e] = val;
        }
    };
    auto put = [&](int buf, const double2 (&v)[3]) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int idx = tid + e * BS_T;
            if (idx < BS_ROWS * (B + 1)) (&stage[buf][0][0])[idx] = v[e];
        }
    };
    double2 nx[3];
    fetch(0, nx);
    put(0, nx);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) fetch(c + 1, nx);                // in flight during this chunk
        const double2 (*rowsc)[B + 1] = stage[c & 1];
        const int rend = min(BS_ROWS, rows - c * BS_ROWS);
        for (int rr = 0; rr < rend; ++rr) {
            double d0 = W[0][0];
            neg += (d0 < 0.0) ? 1 : 0;
            if (fabs(d0) < 1e-280) d0 = -1e-280;            // an exact zero counts as negative (as dstebz does)
            double rd = __builtin_amdgcn_rcp(d0);
            rd = __builtin_fma(__builtin_fma(-d0, rd, 1.0), rd, rd);
            rd = __builtin_fma(__builtin_fma(-d0, rd, 1.0), rd, rd);
            double l[B + 1], c0[B + 1];
#pragma unroll
            for (int p = 1; p <= B; ++p) { c0[p] = W[p][0]; l[p] = c0[p] * rd; }
#pragma unroll
            for (int p = 1; p <= B; ++p)
#pragma unroll
                for (int q = 1; q <= p; ++q) W[p - 1][q - 1] = __builtin_fma(-l[p], c0[q], W[p][q]);   // updated and moved up
#pragma unroll
            for (int q = 0; q <= B; ++q) {                   // row r = c BS_ROWS + rr enters: A(r, r - B + q)
                const double2 hs = rowsc[rr][B - q];
                W[B][q] = __builtin_fma(-x, hs.y, hs.x);
            }
        }
        if (c + 1 < nchunk) put((c + 1) & 1, nx);            // that buffer was last read in chunk c - 1: everyone is past its barrier
        __syncthreads();
    }
    return neg;
}

// LDS of a multisection: the staged rows, the round's points and counts
struct BandSectLds {
    double2 stage[2][BS_ROWS][BS_B + 1];
    double sx[BS_T];
    int sc[BS_T];
    int sfirst;
};

// eigenvalue m (0-based, ascending) of the pencil; every one of the workgroup's BS_T threads calls it and gets the value
__device__ __forceinline__ double band_multisect(int n, int k, const double *__restrict__ SB, const double *__restrict__ HB, int m,
                                                 BandSectLds &L)
{
    double2 (*stage)[BS_ROWS][BS_B + 1] = L.stage;
    double *sx = L.sx;
    int *sc = L.sc;
    int &sfirst = L.sfirst;
    const int t = threadIdx.x, b = k - 1;
    double lo = -0x1p+68, hi = 0x1p+68;                      // count(lo) <= m < count(hi) (every pencil the library assembles lies inside)
    for (int round = 0; round < 14; ++round) {
        double x;
        if (round == 0) x = (t < 128) ? -ldexp(1.0, 67 - t) : ldexp(1.0, t - 188);      // -2^67 .. -2^-60, 2^-60 .. 2^67, ascending
        else x = __builtin_fma(hi - lo, (double)(t + 1) * (1.0 / (BS_T + 1)), lo);
        const int cnt = band_inertia(n, b, SB, HB, x, stage);
        sx[t] = x; sc[t] = cnt;
        if (t == 0) sfirst = BS_T;
        __syncthreads();
        if (cnt > m && x > lo && x < hi) atomicMin(&sfirst, t);          // the first usable point above eigenvalue m
        __syncthreads();
        const int f = sfirst;
        double nlo = lo, nhi = hi;
        if (f < BS_T) nhi = sx[f];
        // the last point below it that is inside the bracket and whose count allows it as a lower end
        for (int i = (f < BS_T ? f : BS_T) - 1; i >= 0; --i)
            if (sc[i] <= m && sx[i] > lo && sx[i] < nhi) { nlo = sx[i]; break; }
        __syncthreads();                                     // sx, sc, sfirst are rewritten next round
        const bool shrunk = (nhi - nlo) < (hi - lo);
        lo = nlo; hi = nhi;
        const double mid = 0.5 * (lo + hi);
        if (!shrunk || mid <= lo || mid >= hi || hi - lo <= 2.0 * 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi)) + 1e-300) break;
    }
    return 0.5 * (lo + hi);
}

}  // namespace bsp
