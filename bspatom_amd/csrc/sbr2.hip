// sbr2.hip -- the bulge chasing in TWO steps: band 64 -> band 16 -> tridiagonal.  The default for n >= 512
// (BSP_SB2ST_VERSION=0 / 9; sb2st.hip, the one-step chase, below that and as the cross-check).
//
// Replaces DSYTRD's second half inside DSYGV (reference call site matrices.f90:248) like sb2st.hip does; the reason for a
// second route is bytes: a one-column-at-a-time chase of a band of half-width b re-reads and re-writes the band once per
// sweep (or per two sweeps, sb2st.hip v7/v8): 6 n^2 b bytes per channel, 936 GB measured per 128-channel step at b = 64
// against 255 GB here.
//   step 1 (sb2sb_mfma_kernel; sb2sb_kernel = the first, all-VALU version, kept as its cross-check): BLOCK bulge chasing
//          (Bischof/Lang/Sun's SBR scheme).  A sweep takes 16 columns at once: QR of a 64 x 16 block, the block reflector
//          Q = I - V T V^T applied to three 64 x 64 tiles per chase item -- the same tiles a column sweep touches, 16 times
//          less often.  Items (sweep s, step k) with k + 3 s = t are independent (their tile sets are disjoint,
//          tools/proto_sbr.py checks it), so ONE LAUNCH PER WAVEFRONT t orders everything: no flags, no spinning,
//          bit-identical results by construction.
//   step 2 (sbr_rows_kernel<16> -- "sb16r_kernel" in round 3's notes; sb16st_kernel = the first layout, kept as its cross-check,
//          BSP_SB16_ROWS=0): one-column chase of the band of half-width 16.  Round 4: the same kernel with tiles of 8,
//          sbr_rows_kernel<8>, is the band route's chase (the band crawford.hip leaves has half-width 8).  A workgroup runs 8 consecutive sweeps, three items apart, on a sliding window of the band
//          held in LDS (512 columns x 32 rows: the band and what the sweeps leave of their bulges, which stays in the matrix
//          between passes) + one wave that only moves data; HBM sees each pass of 8 sweeps once.  P = 1 .. 8 workgroups share
//          the passes of a channel (Sb16Ctl).  sb16st_kernel: one wave per sweep, a 16 x 16 tile spread over the wave;
//          sbr_rows_kernel<B>: a chase item per 16 (8) lanes, four (eight) sweeps per wave, three waves (roles) per group of sweeps.
// Index conventions, the lag of 3 and the working band (<= 127 / <= 31 sub-diagonals) are those of tools/proto_sbr.py.
// Band storage as everywhere: AB[d + 128 j] = A(j + d, j) (sy2sb.hip::extract_band_kernel).
#include "common.h"
#include <cstdio>
#include <cstdlib>

namespace bsp {
namespace {

constexpr int LD = 128;
constexpr int B1 = 64, NB1 = 16, LAG = 3;
constexpr int XLD = 65, VLD = 17;
constexpr int SB2SB_LDS = (64 * XLD + 3 * 64 * VLD + 16 * XLD + 2 * 16 * VLD + 16) * 8;

__device__ __forceinline__ double wsum64(double v)
{
    for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// One chase item of the block scheme.  256 threads; lane = row of the tile, wave = column group.
__global__ __launch_bounds__(256) void sb2sb_kernel(int n, int npad, double *__restrict__ ABall, int t, int s_lo)
{
    extern __shared__ double lds[];
    double *X = lds, *V = X + 64 * XLD, *Y = V + 64 * VLD, *Z = Y + 64 * VLD, *W = Z + 64 * VLD, *T = W + 16 * XLD,
           *G = T + 16 * VLD, *tau = G + 16 * VLD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int s = s_lo + blockIdx.x, k = t - LAG * s;
    if (k < 0) return;
    const int j0 = NB1 * s, r0 = j0 + NB1 + B1 * k;
    if (r0 >= n) return;
    double *AB = ABall + blockIdx.y * ab_stride(npad);
    const int pc0 = k == 0 ? j0 : r0 - B1;               // first column of the left tile (the panel)
    const int ncl = k == 0 ? NB1 : B1;
    const int gi = r0 + lane;                            // this lane's row

    // ---- left tile: rows r0.., columns pc0 .. pc0+ncl
    for (int c = wv; c < ncl; c += 4) {
        const int gc = pc0 + c;
        X[c * XLD + lane] = gi < n ? AB[(size_t)gc * LD + (gi - gc)] : 0.0;
    }
    __syncthreads();
    // Householder QR of its first 16 columns; V explicit (unit diagonal, zeros above), R left in X
    for (int i = 0; i < NB1; ++i) {
        if (wv == 0) {
            const double x = X[i * XLD + lane];
            const double nrm2 = wsum64(lane > i ? x * x : 0.0);
            const double alpha = X[i * XLD + i];
            double tq = 0.0, scale = 0.0, beta = alpha;
            if (nrm2 != 0.0) {
                beta = -copysign(sqrt(alpha * alpha + nrm2), alpha);
                tq = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            V[lane * VLD + i] = lane < i ? 0.0 : (lane == i ? 1.0 : x * scale);
            if (lane == 0) tau[i] = tq;
            X[i * XLD + lane] = lane < i ? x : (lane == i ? beta : 0.0);
        }
        __syncthreads();
        const double tq = tau[i];
        for (int j = i + 1 + wv; j < NB1; j += 4) {
            const double v = V[lane * VLD + i], x = X[j * XLD + lane];
            const double dot = wsum64(v * x);
            X[j * XLD + lane] = x - tq * dot * v;
        }
        __syncthreads();
    }
    // T of Q = H_1 .. H_16 = I - V T V^T (forward, columnwise)
    {
        const int i = tid >> 4, j = tid & 15;
        double a = 0.0;
        for (int r = 0; r < 64; ++r) a += V[r * VLD + i] * V[r * VLD + j];
        G[i * VLD + j] = a;
        T[i * VLD + j] = 0.0;
    }
    __syncthreads();
    for (int i = 0; i < NB1; ++i) {
        if (tid < i) {
            double a = 0.0;
            for (int m = tid; m < i; ++m) a += T[tid * VLD + m] * G[m * VLD + i];
            T[tid * VLD + i] = -tau[i] * a;
        }
        if (tid == i) T[i * VLD + i] = tau[i];
        __syncthreads();
    }
    if (k > 0) {                                         // X(:, 16:64) <- Q^T X = X - V (T^T (V^T X))
        const int i = tid & 15, c = 16 + (tid >> 4) * 3;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int r = 0; r < 64; ++r) {
            const double v = V[r * VLD + i];
            a0 += v * X[c * XLD + r]; a1 += v * X[(c + 1) * XLD + r]; a2 += v * X[(c + 2) * XLD + r];
        }
        W[i * XLD + c] = a0; W[i * XLD + c + 1] = a1; W[i * XLD + c + 2] = a2;
        __syncthreads();
        double b0 = 0.0, b1 = 0.0, b2 = 0.0;
        for (int j = 0; j <= i; ++j) {
            const double tj = T[j * VLD + i];
            b0 += tj * W[j * XLD + c]; b1 += tj * W[j * XLD + c + 1]; b2 += tj * W[j * XLD + c + 2];
        }
        __syncthreads();
        W[i * XLD + c] = b0; W[i * XLD + c + 1] = b1; W[i * XLD + c + 2] = b2;
        __syncthreads();
        double vr[16];
        for (int q = 0; q < 16; ++q) vr[q] = V[lane * VLD + q];
        for (int cc = 16 + wv * 12; cc < 16 + wv * 12 + 12; ++cc) {
            double a = X[cc * XLD + lane];
            for (int q = 0; q < 16; ++q) a -= vr[q] * W[q * XLD + cc];
            X[cc * XLD + lane] = a;
        }
        __syncthreads();
    }
    for (int c = wv; c < ncl; c += 4) {
        const int gc = pc0 + c;
        if (gi < n) AB[(size_t)gc * LD + (gi - gc)] = X[c * XLD + lane];
    }
    __syncthreads();

    // ---- diagonal tile: D <- Q^T D Q = D - V Z^T - Z V^T,  Y = D V T,  Z = Y - 1/2 V (T^T V^T Y)
    for (int c = wv; c < 64; c += 4) {
        if (lane >= c) {
            const double v = gi < n ? AB[(size_t)(r0 + c) * LD + (lane - c)] : 0.0;
            X[c * XLD + lane] = v;
            X[lane * XLD + c] = v;
        }
    }
    __syncthreads();
    {
        double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
        for (int c = 0; c < 64; ++c) {
            const double d = X[c * XLD + lane];
            const double *vc = V + c * VLD + 4 * wv;
            y0 += d * vc[0]; y1 += d * vc[1]; y2 += d * vc[2]; y3 += d * vc[3];
        }
        double *yr = Y + lane * VLD + 4 * wv;
        yr[0] = y0; yr[1] = y1; yr[2] = y2; yr[3] = y3;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * wv + q;
        double a = 0.0;
        for (int m = 0; m <= i; ++m) a += Y[lane * VLD + m] * T[m * VLD + i];
        Z[lane * VLD + i] = a;
    }
    __syncthreads();
    {
        const int i = tid >> 4, j = tid & 15;
        double a = 0.0;
        for (int r = 0; r < 64; ++r) a += V[r * VLD + i] * Z[r * VLD + j];
        G[i * VLD + j] = a;
    }
    __syncthreads();
    {
        const int i = tid >> 4, j = tid & 15;
        double a = 0.0;
        for (int m = 0; m <= i; ++m) a += T[m * VLD + i] * G[m * VLD + j];
        W[i * XLD + j] = a;                              // M = T^T V^T Y (symmetric)
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * wv + q;
        double a = 0.0;
        for (int m = 0; m < 16; ++m) a += V[lane * VLD + m] * W[m * XLD + i];
        Z[lane * VLD + i] -= 0.5 * a;
    }
    __syncthreads();
    {
        double vr[16], zr[16];
        for (int q = 0; q < 16; ++q) { vr[q] = V[lane * VLD + q]; zr[q] = Z[lane * VLD + q]; }
        for (int c = 16 * wv; c < 16 * wv + 16; ++c) {
            if (lane >= c) {
                double a = X[c * XLD + lane];
                for (int q = 0; q < 16; ++q) a -= vr[q] * Z[c * VLD + q] + zr[q] * V[c * VLD + q];
                if (gi < n) AB[(size_t)(r0 + c) * LD + (lane - c)] = a;
            }
        }
    }
    __syncthreads();

    // ---- next bulge tile: X' <- X' Q = X' - (X' V T) V^T, rows r0+64.., columns r0..
    const int r1 = r0 + B1;
    if (r1 >= n) return;
    const int gi1 = r1 + lane;
    for (int c = wv; c < 64; c += 4) {
        const int gc = r0 + c;
        X[c * XLD + lane] = gi1 < n ? AB[(size_t)gc * LD + (gi1 - gc)] : 0.0;
    }
    __syncthreads();
    {
        double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
        for (int c = 0; c < 64; ++c) {
            const double d = X[c * XLD + lane];
            const double *vc = V + c * VLD + 4 * wv;
            y0 += d * vc[0]; y1 += d * vc[1]; y2 += d * vc[2]; y3 += d * vc[3];
        }
        double *yr = Y + lane * VLD + 4 * wv;
        yr[0] = y0; yr[1] = y1; yr[2] = y2; yr[3] = y3;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const int i = 4 * wv + q;
        double a = 0.0;
        for (int m = 0; m <= i; ++m) a += Y[lane * VLD + m] * T[m * VLD + i];
        Z[lane * VLD + i] = a;
    }
    __syncthreads();
    {
        double zr[16];
        for (int q = 0; q < 16; ++q) zr[q] = Z[lane * VLD + q];
        for (int c = 16 * wv; c < 16 * wv + 16; ++c) {
            double a = X[c * XLD + lane];
            for (int q = 0; q < 16; ++q) a -= zr[q] * V[c * VLD + q];
            if (gi1 < n) AB[(size_t)(r0 + c) * LD + (gi1 - (r0 + c))] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same item on the matrix cores.  Every product has a 16 in one dimension, the shape of v_mfma_f64_16x16x4:
//   a = A[m = lane & 15][k = lane >> 4],  b = B[k = lane >> 4][n = lane & 15],  acc[r] = C[m = (lane >> 4) + 4 r][n = lane & 15]
// (layouts as in gemm_f64.hip).  With VT = V T (from T^-1 = striu(V^T V) + diag(1/tau) by forward substitution, T itself is
// never formed):  left tile X <- X - V (VT^T X);  diagonal tile Y = D VT, M = VT^T Y, Z = Y - 1/2 V M, D <- D - V Z^T - Z V^T;
// next tile X' <- X' - (X' VT) V^T.  The QR keeps the panel in registers (wave w owns columns w, w+4, w+8, w+12), one
// LDS-only barrier per column, wave sums by DPP row scans.
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int SB2SB2_LDS = (64 * XLD + 4 * 64 * VLD + 16 * VLD + 16) * 8;

template <int CTRL>
__device__ __forceinline__ double dppm(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ double rlane(double x, int l)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return r.d;
}
__device__ __forceinline__ double wsum_dpp(double x)          // uniform result: row scans + the four row totals
{
    x += dppm<0x111>(x); x += dppm<0x112>(x); x += dppm<0x114>(x); x += dppm<0x118>(x);
    return (rlane(x, 15) + rlane(x, 31)) + (rlane(x, 47) + rlane(x, 63));
}
__device__ __forceinline__ void lds_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(256) void sb2sb_mfma_kernel(int n, int npad, double *__restrict__ ABall, int t, int s_lo)
{
    extern __shared__ double lds[];
    double *X = lds, *V = X + 64 * XLD, *VT = V + 64 * VLD, *Yb = VT + 64 * VLD, *Z = Yb + 64 * VLD, *G = Z + 64 * VLD,
           *tau = G + 16 * VLD;
    double *Gp = Z;                                       // split-K partials [4][16 x VLD] live where Z is not yet
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int s = s_lo + blockIdx.x, k = t - LAG * s;
    if (k < 0) return;
    const int j0 = NB1 * s, r0 = j0 + NB1 + B1 * k;
    if (r0 >= n) return;
    double *AB = ABall + blockIdx.y * ab_stride(npad);
    const int pc0 = k == 0 ? j0 : r0 - B1;
    const int ncl = k == 0 ? NB1 : B1;
    const int gi = r0 + lane;

    // ---- every load of the first two tiles is requested before anything waits: the panel (needed at once), the rest of the left
    // tile (needed after the QR), the diagonal tile (after the left tile); rows beyond n read row r0 and are masked at use
    const bool rin = gi < n;
    const int gr = rin ? gi : r0;
    double xc[4], lt[12], pf[16];
    for (int q = 0; q < 4; ++q) {
        const int gc = pc0 + w + 4 * q;
        xc[q] = AB[(size_t)gc * LD + (gr - gc)];
    }
    if (k > 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int gc = pc0 + 16 + w + 4 * q;
            lt[q] = AB[(size_t)gc * LD + (gr - gc)];
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = w + 4 * q;                          // column c, rows lane >= c (clamped address, masked at use)
        const bool in = lane >= c && rin;
        pf[q] = AB[in ? (size_t)(r0 + c) * LD + (lane - c) : 0];
    }
    for (int q = 0; q < 4; ++q) xc[q] = rin ? xc[q] : 0.0;
    // Column i's reflector is formed by its owner (wave i & 3) and published through LDS; every wave then applies it to its own
    // columns to the right.  With look-ahead: the owner of column i + 1 applies reflector i to that column FIRST and forms
    // reflector i + 1 at once, while the other waves are still applying reflector i -- one barrier per column and the
    // reflector's chain (norm, sqrt, two divisions) beside the updates instead of before them.  Every column sees the same
    // operations in the same order as without look-ahead: the band comes out bit-identical.
    auto make_reflector = [&](const int i) {               // by wave i & 3
        const double x = xc[i >> 2];
        const double nrm2 = wsum_dpp(lane > i ? x * x : 0.0);
        const double alpha = rlane(x, i);
        double tq = 0.0, scale = 0.0, beta = alpha;
        if (nrm2 != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + nrm2), alpha);
            tq = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        V[lane * VLD + i] = lane < i ? 0.0 : (lane == i ? 1.0 : x * scale);
        if (lane == 0) tau[i] = tq;
        xc[i >> 2] = lane < i ? x : (lane == i ? beta : 0.0);
    };
    if (w == 0) make_reflector(0);
    lds_bar();
#pragma unroll
    for (int i = 0; i < NB1 - 1; ++i) {
        const double v = V[lane * VLD + i], tq = tau[i];
        if (w == ((i + 1) & 3)) {
            const int q1 = (i + 1) >> 2;
            const double dot = wsum_dpp(v * xc[q1]);
            xc[q1] -= tq * dot * v;
            make_reflector(i + 1);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (w + 4 * q > i + 1) {
                const double dot = wsum_dpp(v * xc[q]);
                xc[q] -= tq * dot * v;
            }
        lds_bar();
    }
    for (int q = 0; q < 4; ++q) {                          // R (and zeros) back to the band
        const int gc = pc0 + w + 4 * q;
        if (gi < n) AB[(size_t)gc * LD + (gi - gc)] = xc[q];
    }
    if (k > 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) X[(16 + w + 4 * q) * XLD + lane] = rin ? lt[q] : 0.0;
    }
    // G = V^T V, split over the waves along the rows
    {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < 4; ++q) {
            const double a = V[(16 * w + 4 * q + l4) * VLD + l15];
            acc = MFMA(a, a, acc);
        }
        for (int r = 0; r < 4; ++r) Gp[w * 16 * VLD + (l4 + 4 * r) * VLD + l15] = acc[r];
    }
    lds_bar();
    {
        const int i = tid >> 4, j = tid & 15, o = i * VLD + j;
        G[o] = (Gp[o] + Gp[16 * VLD + o]) + (Gp[32 * VLD + o] + Gp[48 * VLD + o]);
    }
    lds_bar();
    if (w == 0) {                                         // VT T^-1 = V, row by row
        double vt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double a = V[lane * VLD + j];
#pragma unroll
            for (int i = 0; i < j; ++i) a -= vt[i] * G[i * VLD + j];
            vt[j] = a * tau[j];
            VT[lane * VLD + j] = vt[j];
        }
    }
    lds_bar();

    // ---- left tile
    if (k > 0) {
        if (w < 3) {                                      // W2 = VT^T X(:, 16:64), block w
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
            const int cn = 16 + 16 * w + l15;
            for (int q = 0; q < 16; ++q) acc = MFMA(VT[(4 * q + l4) * VLD + l15], X[cn * XLD + 4 * q + l4], acc);
            for (int r = 0; r < 4; ++r) Yb[(l4 + 4 * r) * XLD + cn] = acc[r];
        }
        lds_bar();
        for (int nb = 0; nb < 3; ++nb) {                  // X(:, 16:64) -= V W2, row block w
            const int cn = 16 + 16 * nb + l15;
            d4_t acc;
            for (int r = 0; r < 4; ++r) acc[r] = X[cn * XLD + 16 * w + l4 + 4 * r];
            for (int q = 0; q < 4; ++q) acc = MFMA(-V[(16 * w + l15) * VLD + 4 * q + l4], Yb[(4 * q + l4) * XLD + cn], acc);
            for (int r = 0; r < 4; ++r) X[cn * XLD + 16 * w + l4 + 4 * r] = acc[r];
        }
        lds_bar();
        for (int c = 16 + w; c < 64; c += 4) {
            const int gc = pc0 + c;
            if (gi < n) AB[(size_t)gc * LD + (gi - gc)] = X[c * XLD + lane];
        }
        lds_bar();
    }

    // ---- diagonal tile
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = w + 4 * q;
        if (lane >= c) {
            const double v = gi < n ? pf[q] : 0.0;
            X[c * XLD + lane] = v;
            X[lane * XLD + c] = v;
        }
    }
    const int r1 = r0 + B1, gi1 = r1 + lane;
    const bool more = r1 < n;
    if (more) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gc = r0 + w + 4 * q;
            pf[q] = AB[gi1 < n ? (size_t)gc * LD + (gi1 - gc) : 0];
        }
    }
    lds_bar();
    {                                                     // Y = D VT, row block w
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < 16; ++q) acc = MFMA(X[(4 * q + l4) * XLD + 16 * w + l15], VT[(4 * q + l4) * VLD + l15], acc);
        for (int r = 0; r < 4; ++r) Yb[(16 * w + l4 + 4 * r) * VLD + l15] = acc[r];
    }
    lds_bar();
    {                                                     // M = VT^T Y, split along the rows
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        for (int q = 4 * w; q < 4 * w + 4; ++q) acc = MFMA(VT[(4 * q + l4) * VLD + l15], Yb[(4 * q + l4) * VLD + l15], acc);
        for (int r = 0; r < 4; ++r) Gp[w * 16 * VLD + (l4 + 4 * r) * VLD + l15] = acc[r];
    }
    lds_bar();
    {
        const int i = tid >> 4, j = tid & 15, o = i * VLD + j;
        G[o] = (Gp[o] + Gp[16 * VLD + o]) + (Gp[32 * VLD + o] + Gp[48 * VLD + o]);
    }
    lds_bar();
    {                                                     // Z = Y - 1/2 V M, row block w
        d4_t acc;
        for (int r = 0; r < 4; ++r) acc[r] = Yb[(16 * w + l4 + 4 * r) * VLD + l15];
        for (int q = 0; q < 4; ++q) acc = MFMA(-0.5 * V[(16 * w + l15) * VLD + 4 * q + l4], G[(4 * q + l4) * VLD + l15], acc);
        for (int r = 0; r < 4; ++r) Z[(16 * w + l4 + 4 * r) * VLD + l15] = acc[r];
    }
    lds_bar();
    for (int bi = w; bi < 10; bi += 4) {                  // D -= V Z^T + Z V^T, the ten 16 x 16 blocks of the lower triangle
        const int mb = bi < 1 ? 0 : (bi < 3 ? 1 : (bi < 6 ? 2 : 3));
        const int nb = bi - (mb * (mb + 1)) / 2;
        d4_t acc;
        for (int r = 0; r < 4; ++r) acc[r] = X[(16 * nb + l15) * XLD + 16 * mb + l4 + 4 * r];
        for (int q = 0; q < 4; ++q) {
            acc = MFMA(-V[(16 * mb + l15) * VLD + 4 * q + l4], Z[(16 * nb + l15) * VLD + 4 * q + l4], acc);
            acc = MFMA(-Z[(16 * mb + l15) * VLD + 4 * q + l4], V[(16 * nb + l15) * VLD + 4 * q + l4], acc);
        }
        for (int r = 0; r < 4; ++r) X[(16 * nb + l15) * XLD + 16 * mb + l4 + 4 * r] = acc[r];
    }
    lds_bar();
    for (int c = w; c < 64; c += 4)
        if (lane >= c && gi < n) AB[(size_t)(r0 + c) * LD + (lane - c)] = X[c * XLD + lane];

    // ---- next bulge tile
    if (!more) return;
    lds_bar();
#pragma unroll
    for (int q = 0; q < 16; ++q) X[(w + 4 * q) * XLD + lane] = gi1 < n ? pf[q] : 0.0;
    lds_bar();
    {                                                     // W1 = X' VT, row block w
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < 16; ++q) acc = MFMA(X[(4 * q + l4) * XLD + 16 * w + l15], VT[(4 * q + l4) * VLD + l15], acc);
        for (int r = 0; r < 4; ++r) Yb[(16 * w + l4 + 4 * r) * VLD + l15] = acc[r];
    }
    lds_bar();
    for (int nb = 0; nb < 4; ++nb) {                      // X' -= W1 V^T
        d4_t acc;
        for (int r = 0; r < 4; ++r) acc[r] = X[(16 * nb + l15) * XLD + 16 * w + l4 + 4 * r];
        for (int q = 0; q < 4; ++q) acc = MFMA(-Yb[(16 * w + l15) * VLD + 4 * q + l4], V[(16 * nb + l15) * VLD + 4 * q + l4], acc);
        for (int r = 0; r < 4; ++r) X[(16 * nb + l15) * XLD + 16 * w + l4 + 4 * r] = acc[r];
    }
    lds_bar();
    for (int c = w; c < 64; c += 4) {
        const int gc = r0 + c;
        if (gi1 < n) AB[(size_t)gc * LD + (gi1 - gc)] = X[c * XLD + lane];
    }
}
#undef MFMA

// ---------------------------------------------------------------------------------------------------------------------
// step 2: band 16 -> tridiagonal.  One workgroup (8 waves) per channel.  Wave w of pass p runs sweep s = 8 p + w; at step t it
// works on item k = t - 3 w of its sweep.  A 16 x 16 tile lives in a wave as 4 doubles per lane: lane = (row r = lane & 15,
// column group g = lane >> 4, columns 4 g .. 4 g + 3).
constexpr int B2 = 16, NW2 = 8, WCOLS = 512, WROWS = 32;
constexpr int SB16_LDS = (WCOLS * WROWS + NW2 * 16 + 2) * 8;

template <int CTRL>
__device__ __forceinline__ double dppd(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ double rdlane(double x, int l)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return r.d;
}
// Sum over the 16 lanes of a DPP row, the total in every lane (bitwise the same in all of them: each step adds the two
// halves of a pair, and addition commutes): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror.
__device__ __forceinline__ double rsum16(double x)
{
    x += dppd<0xB1>(x);
    x += dppd<0x4E>(x);
    x += dppd<0x141>(x);
    x += dppd<0x140>(x);
    return x;
}
// four row sums at once: the same four operations per value as rsum16 (the same bits), the four dependent chains interleaved so
// that the DPP hazard slots and the add latencies of one are filled by the others (back to back the compiler pads them with s_nop)
__device__ __forceinline__ void rsum16x4(double (&x)[4])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] += dppd<0xB1>(x[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] += dppd<0x4E>(x[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] += dppd<0x141>(x[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] += dppd<0x140>(x[q]);
}
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Householder parameters from alpha and the squared norm of the rest (> 0), with IEEE sqrt and divisions.  (v_rsq_f64 /
// v_rcp_f64 + three Newton steps were 2 ms faster per 128-channel step and cost the eigenvalues next to zero their accuracy: C5
// went from 2 to 27 exceptions to 1e-10 relative; the graded pencils notice a few ulp in tau.)
__device__ __forceinline__ void house_params(double alpha, double nrm2, double &beta, double &tq, double &scale)
{
    beta = -copysign(sqrt(alpha * alpha + nrm2), alpha);
    tq = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
}

// One chase item of a wave: reflector from x = A(r0.., c0), bulge tile B <- H B (k > 0), D <- H D H, next tile B' <- B' H.
// Layouts: B and D lane = (row r, columns 4g..4g+3); B' transposed, lane = (column r, rows 4g..4g+3): every reduction stays
// inside a DPP row.
template <bool FAST>
__device__ __forceinline__ void chase_item(double *Lw, double *pS, int r0, int c0, int k, int r, int g, int cA, int cT,
                                           const int (&offD)[4])
{
#define LWI(i, c) (((((c)) & (WCOLS - 1)) << 5) + ((i) - (c)))
    const int r1 = r0 + B2;
    int ix, ixc[4], ib[4], id[4], it[4];
    if (FAST) {
        const int UB = ((c0 & (WCOLS - 1)) << 5) + (r0 - c0), UD = (r0 & (WCOLS - 1)) << 5;
        ix = UB + r;
        for (int j = 0; j < 4; ++j) {
            ixc[j] = UB + 4 * g + j;
            ib[j] = UB + cA + 31 * j;
            id[j] = UD + offD[j];
            it[j] = UD + cT + j;
        }
    } else {
        ix = LWI(r0 + r, c0);
        for (int j = 0; j < 4; ++j) {
            const int cc = 4 * g + j;
            ixc[j] = LWI(r0 + cc, c0);
            ib[j] = LWI(r0 + r, c0 + cc);
            id[j] = r >= cc ? LWI(r0 + r, r0 + cc) : LWI(r0 + cc, r0 + r);
            it[j] = LWI(r1 + cc, r0 + r);
        }
    }
#undef LWI
    const double x = Lw[ix];
    double xc[4], b[4], dv[4], bt[4];
    for (int j = 0; j < 4; ++j) {
        xc[j] = Lw[ixc[j]];
        b[j] = k > 0 ? Lw[ib[j]] : 0.0;
        dv[j] = Lw[id[j]];
        bt[j] = Lw[it[j]];
    }
    const double nrm2 = rsum16(r > 0 ? x * x : 0.0);
    const double alpha = rdlane(x, 0);
    double tq = 0.0, scale = 0.0, beta = alpha;
    if (nrm2 != 0.0) house_params(alpha, nrm2, beta, tq, scale);
    const double v = r == 0 ? 1.0 : x * scale;
    double vc[4];
    for (int j = 0; j < 4; ++j) vc[j] = (4 * g + j) == 0 ? 1.0 : xc[j] * scale;
    if (g == 0) Lw[ix] = r == 0 ? beta : 0.0;
    if (k > 0) {                                          // rest of the bulge tile: B <- H B
        double d4[4];
        for (int j = 0; j < 4; ++j) d4[j] = v * b[j];
        rsum16x4(d4);
        for (int j = 0; j < 4; ++j)
            if (4 * g + j > 0) Lw[ib[j]] = b[j] - tq * d4[j] * v;
    }
    {                                                     // next bulge tile: B' <- B' H
        double q4[4];
        for (int j = 0; j < 4; ++j) q4[j] = bt[j] * v;
        rsum16x4(q4);
        for (int j = 0; j < 4; ++j) {
            const double q = tq * q4[j];
            Lw[it[j]] = bt[j] - q * v;
        }
    }
    // diagonal tile, two-sided: p = tau D v by columns (D is symmetric), by rows through pS
    double pc[4], part = 0.0;
    for (int j = 0; j < 4; ++j) pc[j] = v * dv[j];
    rsum16x4(pc);
    for (int j = 0; j < 4; ++j) {
        pc[j] = tq * pc[j];
        part += vc[j] * pc[j];
    }
    if (r < 4) pS[4 * g + r] = r == 0 ? pc[0] : (r == 1 ? pc[1] : (r == 2 ? pc[2] : pc[3]));
    const double vtp = (rdlane(part, 0) + rdlane(part, 16)) + (rdlane(part, 32) + rdlane(part, 48));
    const double kk = 0.5 * tq * vtp;
    const double w = pS[r] - kk * v;
    for (int j = 0; j < 4; ++j) {
        const double wc = pc[j] - kk * vc[j];
        if (r >= 4 * g + j) Lw[id[j]] = dv[j] - v * wc - w * vc[j];
    }
}

// Control block of one channel when P workgroups share its passes (member w runs the passes w, w + P, ...; pass p + 1 follows
// pass p through global memory).  Same rules as the rings of sb2st.hip (documented there): the members must share an XCD
// (one L2) -- blocks b, b + 8, ... are observed to, every member reports HW_REG_XCC_ID and the ring forms only if they agree,
// otherwise (or if a member does not show up in time) member 0 runs the channel alone; data is handed over with plain
// stores + s_waitcnt vmcnt(0) + a relaxed agent-scope store of the progress word, and taken with a relaxed poll of that
// word + buffer_inv sc1 (drops the CU's L1 lines) before the plain loads.  Every spin is bounded.
struct Sb16Ctl {
    unsigned long long hs;                 // handshake: byte w = 0x10 | XCC id of member w; bit 62 COMMIT, bit 63 ABORT
    int err, pad;
    unsigned long long prog[8];            // member w: (pass << 32) | columns of that pass already stored (n when it is complete)
};
constexpr unsigned long long S16_COMMIT = 1ull << 62, S16_ABORT = 1ull << 63;

__device__ __forceinline__ void s16_wait(const unsigned long long *pollp, unsigned long long want, Sb16Ctl *C, int *status)
{
    // A wait that timed out once has set C->err: every later wait of the channel gives up after a short spin instead of its
    // full bound, so that a stalled ring member ends the launch in milliseconds with the status word set, not in minutes
    // (the channel's results are void either way: BSP_ERR_HIP is returned).
    int spin = 0;
    const int bound = __hip_atomic_load(&C->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 2000 : 20000000;
    while (__hip_atomic_load(pollp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spin > bound) {                             // never seen; ends the wait instead of the machine
            atomicExch(&C->err, 1);
            if (status) atomicExch(status, BSP_ERR_HIP);
            break;
        }
        if ((spin & 1023) == 0 && __hip_atomic_load(&C->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("buffer_inv sc1" ::: "memory");
}

__global__ __launch_bounds__(576) void sb16st_kernel(int n, int npad, int batch, double *__restrict__ ABall,
                                                     double *__restrict__ dall, double *__restrict__ eall, long long *diag,
                                                     Sb16Ctl *ctl, int P, int *status, int force_abort)
{
    extern __shared__ double lds[];
    double *Lw = lds;
    long long dacc[5] = {0, 0, 0, 0, 0}, dt0 = 0;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    double *pS = lds + WCOLS * WROWS + (wv & 7) * 16;
    // lane constants of the fast path, in doubles relative to element (r0, c0) resp. (r0, r0) of the window:
    // element (r0 + a, c + b) lies 31 b + a further
    const int cA = 124 * g + r, cT = 31 * r + 4 * g + 16;
    int offD[4];
    for (int j = 0; j < 4; ++j) offD[j] = r >= 4 * g + j ? cA + 31 * j : 31 * r + 4 * g + j;
    // ---- which channel; alone (ctl == nullptr) or member w of a ring of P workgroups
    int chn = blockIdx.x, w = 0, stride = 1;
    const unsigned long long *pollp = nullptr;
    unsigned long long *pubp = nullptr;
    Sb16Ctl *C = nullptr;
    if (ctl) {
        const int blk = blockIdx.x, grp = blk / (8 * P), rr = blk % (8 * P);
        chn = grp * 8 + (rr & 7); w = rr >> 3;
        if (chn >= batch) return;
        C = ctl + chn;
        int *modep = reinterpret_cast<int *>(lds + WCOLS * WROWS + NW2 * 16);
        if (tid == 0) {
            const unsigned long long xcc = (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu);   // HW_REG_XCC_ID
            const unsigned long long mine = (0x10ull | xcc) << (8 * w);
            unsigned long long full = 0;
            for (int q = 0; q < P; ++q) full |= 0x10ull << (8 * q);
            unsigned long long v = atomicOr(&C->hs, mine) | mine;
            for (int spin = 0; !(v & (S16_COMMIT | S16_ABORT)); ++spin) {
                if (force_abort == 1) atomicCAS(&C->hs, v, v | S16_ABORT);
                else if ((v & full) == full) atomicCAS(&C->hs, v, v | S16_COMMIT);
                else if (spin > 400000) atomicCAS(&C->hs, v, v | S16_ABORT);
                else __builtin_amdgcn_s_sleep(4);
                v = __hip_atomic_load(&C->hs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int mode;                                     // 0 = exit, 1 = ring, 2 = alone
            if (v & S16_ABORT) mode = (w == 0) ? 2 : 0;
            else {
                bool same = true;
                for (int q = 1; q < P; ++q) same = same && (((v >> (8 * q)) & 0xfu) == (v & 0xfu));
                if (force_abort == 2) same = false;
                mode = same ? 1 : ((w == 0) ? 2 : 0);
            }
            *modep = mode;
        }
        __syncthreads();
        const int mode = __builtin_amdgcn_readfirstlane(*modep);
        __syncthreads();
        if (mode == 0) return;
        if (mode == 1 && P > 1) { stride = P; pollp = &C->prog[(w + P - 1) % P]; pubp = &C->prog[w]; }
        else w = 0;
    }
    double *AB = ABall + (size_t)chn * ab_stride(npad);
#define LW(i, c) Lw[((((c)) & (WCOLS - 1)) << 5) + ((i) - (c))]

    double *dump = AB + (size_t)npad * LD + 64;              // padding behind the band (ab_stride): target of masked stores
    const int npass = (n - 2 + NW2 - 1) / NW2;
    for (int ps = w; ps < npass; ps += stride) {
        const int s0 = ps * NW2;
        int RP = s0, LP = s0 + 64;
        if (pollp && ps > 0) {                            // the first 128 columns of this pass, as the pass before left them
            if (tid == 0) {
                const int need = s0 + 128 < n ? s0 + 128 : n;
                s16_wait(pollp, ((unsigned long long)(ps - 1) << 32) + (unsigned)need, C, status);
            }
            __syncthreads();
            asm volatile("buffer_inv sc1" ::: "memory");
        }
        for (int idx = tid; idx < 64 * WROWS; idx += 576) {
            const int c = s0 + (idx >> 5), d = idx & 31;
            Lw[((c & (WCOLS - 1)) << 5) + d] = c + d < n ? AB[(size_t)c * LD + d] : 0.0;
        }
        // Wave 8 moves data and does nothing else.  Columns enter the window in blocks of 64 every fourth step, four steps
        // after their loads were issued (the registers are not touched in between, so nothing waits for HBM); columns
        // leave it with plain stores, 16 per step.  The eight chasing waves never touch global memory inside a pass.
        const int md = lane & 31, mh = lane >> 5;                   // mover lane: row md of columns 2 i + mh
        double qb[32];
        unsigned long long pw = 0;                                  // the partner's progress word as last seen
        if (wv == NW2) {
            const unsigned o0 = (unsigned)(LP + mh) * LD + md, omax = (unsigned)(n - 1) * LD;
            for (int i = 0; i < 32; ++i) {                          // unmasked value; the mask is applied at the LDS write
                unsigned o = o0 + 2 * LD * i;
                o = o < omax ? o : omax;
                qb[i] = AB[o];
            }
        }
        __syncthreads();
        const int s = s0 + wv;
        const int nsteps = (n - s0 - 1 + B2 - 1) / B2 + LAG * (NW2 - 1);
        for (int t = 0; t < nsteps; ++t) {
            int RPn = s0 + NW2 + B2 * (t - LAG * (NW2 - 1) - 1);           // columns leaving: left of the trailing wave's tiles
            if (RPn > n) RPn = n;
            if (RPn < RP) RPn = RP;
            if (wv == NW2) {
                // Order matters: everything that WAITS for memory comes first, while the only operations in flight are a step
                // old; the stores of this step go out last.  (Stores first, as in the first version, made the block's LDS writes
                // wait for stores issued a moment ago: 5500 ticks every fourth step.)
                if (diag) dt0 = (long long)__builtin_amdgcn_s_memtime();
                if ((t & 3) == 0) {
                    if (pollp && ps > 0 && LP + 64 < n) {  // the block about to be requested, as the pass before left it
                        const int need = LP + 128 < n ? LP + 128 : n;
                        const unsigned long long want = ((unsigned long long)(ps - 1) << 32) + (unsigned)need;
                        if (pw < want) s16_wait(pollp, want, C, status);      // pw was requested two steps ago
                        else asm volatile("buffer_inv sc1" ::: "memory");
                    }
                    // 32-bit offsets from uniform bases (one add per element): the data-moving wave is bound by its instructions
                    {
                        const unsigned s0_ = (unsigned)(LP + mh), lim = (unsigned)(n - md);      // column c is inside while c < n - md
                        for (int i = 0; i < 32; ++i) {
                            const unsigned c = s0_ + 2 * i;
                            Lw[(((c & (WCOLS - 1)) << 5) + md)] = c < lim ? qb[i] : 0.0;
                        }
                        const unsigned o0 = (unsigned)(LP + 64 + mh) * LD + md, omax = (unsigned)(n - 1) * LD;
                        for (int i = 0; i < 32; ++i) {
                            unsigned o = o0 + 2 * LD * i;
                            o = o < omax ? o : omax;                                          // clamped, masked at the LDS write
                            qb[i] = AB[o];
                        }
                    }
                }
                if ((t & 3) == 3 && pubp) {               // everything stored before this step has reached the L2
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0)
                        __hip_atomic_store(pubp, ((unsigned long long)ps << 32) + (unsigned)RP, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
                if ((t & 3) == 2 && pollp) pw = __hip_atomic_load(pollp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (diag) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[1] += t_ - dt0; dt0 = t_; }
                {
                    const unsigned c0_ = (unsigned)(RP + mh), dumpo = (unsigned)npad * LD + 64 + lane;
                    for (int i = 0; i < 8; ++i) {
                        const unsigned c = c0_ + 2 * i;
                        const unsigned o = c < (unsigned)RPn ? c * LD + md : dumpo;
                        AB[o] = Lw[((c & (WCOLS - 1)) << 5) + md];
                    }
                }
                if (diag) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[0] += t_ - dt0; }
            }
            if ((t & 3) == 0) LP += 64;
            RP = RPn;
            const int k = t - LAG * wv;
            const int r0 = s + 1 + B2 * k;
            if (diag) dt0 = (long long)__builtin_amdgcn_s_memtime();
            if (wv < NW2 && k >= 0 && s < n - 2 && r0 < n) {
                const int c0 = k == 0 ? s : r0 - B2;
                // a tile whose 16 columns do not wrap around the ring (9 in 10) is addressed with lane constants and immediates
                const bool fast = (c0 & (WCOLS - 1)) <= WCOLS - 16 && (r0 & (WCOLS - 1)) <= WCOLS - 16;
                if (fast) chase_item<true>(Lw, pS, r0, c0, k, r, g, cA, cT, offD);
                else chase_item<false>(Lw, pS, r0, c0, k, r, g, cA, cT, offD);
                if (diag) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[2] += t_ - dt0; dt0 = t_; }
            }
            lds_only_barrier();
            if (diag) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[3] += t_ - dt0; dacc[4] += 1; }
        }
        __syncthreads();
        int hi = LP < n ? LP : n;
        for (int idx = tid; idx < (hi - RP) * WROWS; idx += 576) {
            const int c = RP + (idx >> 5), d = idx & 31;
            AB[(size_t)c * LD + d] = Lw[((c & (WCOLS - 1)) << 5) + d];
        }
        __syncthreads();
        if (pubp && tid == 0)
            __hip_atomic_store(pubp, ((unsigned long long)ps << 32) + (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef LW
    if (diag && blockIdx.x == 0 && lane == 0)
        for (int q = 0; q < 5; ++q) diag[wv * 5 + q] = dacc[q];
    if (stride > 1 && (npass - 1) % stride != w && npass > 0) return;   // the member of the last pass has seen every pass end
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");
    double *dd = dall + (size_t)chn * npad, *ee = eall + (size_t)chn * npad;
    for (int j = tid; j < n; j += 576) {
        dd[j] = AB[(size_t)j * LD];
        ee[j] = j < n - 1 ? AB[(size_t)j * LD + 1] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// step 2, rows layout (round 3; BSP_SB16_ROWS=1, the default; sb16st_kernel above stays as the cross-check).
// The first layout spreads ONE 16 x 16 tile over a wave (4 values per lane) and pays for it in instructions: ~400 per item, 144
// of them DPP steps of 12 row sums, most of the rest addresses; eight waves of it fill the CU's issue slots.  Here a DPP row of
// 16 lanes owns an item and a wave runs FOUR sweeps (row g: sweep 4 grp + g of the pass, three items apart as before -- the
// items of a step are independent whichever wave they sit in).  A lane holds 16 values of a tile, chosen so that every product
// with the reflector is a sum INSIDE the lane:
//   bulge tile B <- H B        lane j = column j (rows 0..15):  y_j = v^T B(:, j),      B(:, j)  -= tau y_j v
//   next tile   B' <- B' H     lane i = row i (columns 0..15):  y_i = B'(i, :) v,       B'(i, :) -= tau y_i v^T
//   diagonal    D <- H D H     lane j = column j, its upper part through the symmetry:  p = tau D v, z = p - (tau/2)(v^T p) v,
//                              D(:, j) -= v z_j + z v_j   (one 16-lane sum for v^T p; z goes round through LDS)
// Three waves share the items of four sweeps.  Role 0 takes B' -- and with it the NEXT item's reflector: the column that item
// annihilates is the first column of the updated B', element j in lane j, so its norm is one row sum; beta, tau and v are formed
// right there, (beta, 0, .., 0) goes into the window in place of the column and (v, tau) into an exchange slot (two per sweep,
// by step parity), while the rest of the tile is still being updated: the square root and the two divisions, the longest
// dependent chain of an item, leave the start of the next step.  Role 1 takes D, role 2 takes B.  At the start of a step every role
// reads (v, tau) of its item from the slot as a broadcast.  (The waves are bound by the latency of their dependent chains, not by
// issue slots: one wave per SIMD and role; with D and B in one wave the step was 2860 ticks, 2475 of them that wave.)  The first item of a sweep: role 0 forms its reflector from the band
// column one step ahead (the column is final by then and nobody reads it in between, tools/proto_sbr.py).
// The kernel is a template on the tile size B: 16 (half-width <= 16: the dense route's second step, and the band route as it first
// was) and 8 (half-width <= 8: what the band route's reduction really leaves, crawford.hip).  With B = 8 an item is owned by HALF a
// DPP row, a wave runs EIGHT sweeps and a pass sixteen; a window column has 16 rows; everything else is the same program.
constexpr int NCW4 = 6, SB16R_THREADS = (NCW4 + 1) * 64;
// window element at BYTE offset `off` (the tile addresses of the chasing waves are kept in bytes: as indices every access
// paid a shift, 16 vector instructions per step of a role)
__device__ __forceinline__ double &ldsb(double *Lw, int off) { return *reinterpret_cast<double *>(reinterpret_cast<char *>(Lw) + off); }

template <int B>
struct Rw {
    static constexpr int WR = 2 * B;                   // rows of a window column: the band and what the sweeps leave of their bulges
    static constexpr int WSH = B == 16 ? 5 : 4;        // log2(WR)
    static constexpr int RPW = 64 / B;                 // items (sweeps) of a wave
    static constexpr int NSW = 2 * RPW;                // sweeps of a pass: two groups of RPW
    static constexpr int MLAG = B == 16 ? 4 : 8;       // steps between the request of a column and the wait for it (about the latency of HBM)
    static constexpr int LP0 = B * (MLAG + 3);         // columns in the window when a pass starts
    static constexpr int Z = WCOLS * WR;               // z of the diagonal-tile waves [2 groups][RPW items][B]
    static constexpr int DUMP = Z + 128;               // target of masked stores [6 waves][RPW items][B]
    static constexpr int XS = DUMP + 384;              // exchange slots [2 parities][NSW sweeps][2 B]: v (B), tau, padding
    static constexpr int XPAR = NSW * 2 * B;           // one parity's slots (256 for both B)
    static constexpr int MODE = XS + 2 * XPAR;
    static constexpr int LDS = (MODE + 2) * 8;
    static_assert(B == 16 || B == 8, "tile sizes of the rows kernel");
    static_assert((1 << WSH) == WR && XPAR == 256 && LP0 + B <= 128, "window layout");
};

// Sum over the B lanes that own an item, the total in each of them (B = 8: the first three steps of rsum16)
template <int B>
__device__ __forceinline__ double rsumB(double x)
{
    x += dppd<0xB1>(x);
    x += dppd<0x4E>(x);
    x += dppd<0x141>(x);
    if (B == 16) x += dppd<0x140>(x);
    return x;
}
// two such sums at once (the chains interleaved, same operations per value)
template <int B>
__device__ __forceinline__ void rsumBx2(double &a, double &b)
{
    a += dppd<0xB1>(a); b += dppd<0xB1>(b);
    a += dppd<0x4E>(a); b += dppd<0x4E>(b);
    a += dppd<0x141>(a); b += dppd<0x141>(b);
    if (B == 16) { a += dppd<0x140>(a); b += dppd<0x140>(b); }
}

// The reflector of the column whose element j is x0 in lane j of the item: v and tau into the slot xw, beta (lane 0) resp. zeros
// into the window at index hx (the column's own place).  Branch-free: a column that is zero below its first element gives
// tau = 0, v = e_0, beta = alpha.
template <int B>
__device__ __forceinline__ void next_reflector(double *Lw, const double x0, const int j, const int xw, const int hx)
{
    double nrm2 = j > 0 ? x0 * x0 : 0.0, alpha = j == 0 ? x0 : 0.0;
    rsumBx2<B>(nrm2, alpha);
    const bool nz = nrm2 != 0.0;
    const double beta0 = -copysign(sqrt(fma(alpha, alpha, nz ? nrm2 : 1.0)), alpha);      // never zero
    const double tq0 = (beta0 - alpha) / beta0, scale0 = 1.0 / (alpha - beta0);
    const double beta = nz ? beta0 : alpha, tq = nz ? tq0 : 0.0, scale = nz ? scale0 : 0.0;
    Lw[xw + j] = j == 0 ? 1.0 : x0 * scale;
    if (j == 0) Lw[xw + B] = tq;
    ldsb(Lw, hx) = j == 0 ? beta : 0.0;                                 // hx: a byte offset
}

// Window BYTE offset of the element (r0 + B + j, r0 + i) of an item's next tile, i = 0 .. B - 1, as base[i] + 8 (WR - 1) i: the column
// r0 + i wraps around the ring from i = 512 - (r0 mod 512) on
template <int B, bool FAST>                                            // FAST: no tile of the wave wraps
__device__ __forceinline__ void next_tile_addr(const int r0, const int j, int (&at)[B])
{
    const int cb = r0 & (WCOLS - 1), a0 = ((cb << Rw<B>::WSH) + B + j) * 8, iw = WCOLS - cb;
#pragma unroll
    for (int i = 0; i < B; ++i) at[i] = (FAST || i < iw) ? a0 : a0 - WCOLS * Rw<B>::WR * 8;
}
// ... of the element D(i, j) of its diagonal tile (bytes; offD in bytes), from the stored triangle: column r0 + min(i, j) (wraps
// when both do), (WR - 1) min + max
template <int B, bool FAST>
__device__ __forceinline__ void diag_tile_addr(const int r0, const int j, const int (&offD)[B], int (&ad)[B])
{
    const int cb = r0 & (WCOLS - 1), ud = (cb << Rw<B>::WSH) * 8, iw = WCOLS - cb, udq = j >= iw ? ud - WCOLS * Rw<B>::WR * 8 : ud;
#pragma unroll
    for (int i = 0; i < B; ++i) ad[i] = ((!FAST && i >= iw) ? udq : ud) + offD[i];
}

// role 0, the arithmetic of an item: next tile bt (lane j = row j) <- bt H, the reflector of its first column for the next item
template <int B>
__device__ __forceinline__ void chase4_next(double *Lw, double (&bt)[B], const double (&v)[B], const double tq, const int (&at)[B],
                                            const int j, const int xw)
{
    double y0 = bt[0], y1 = v[1] * bt[1], y2 = v[2] * bt[2], y3 = v[3] * bt[3];
#pragma unroll
    for (int i = 4; i < B; i += 4) {
        y0 = fma(v[i], bt[i], y0); y1 = fma(v[i + 1], bt[i + 1], y1);
        y2 = fma(v[i + 2], bt[i + 2], y2); y3 = fma(v[i + 3], bt[i + 3], y3);
    }
    const double ct = tq * ((y0 + y1) + (y2 + y3));
    next_reflector<B>(Lw, bt[0] - ct, j, xw, at[0]);                 // B'(j, 0): element j of the next item's column
#pragma unroll
    for (int i = 1; i < B; ++i) ldsb(Lw, at[i] + (Rw<B>::WR - 1) * 8 * i) = fma(-ct, v[i], bt[i]);
}

// role 1: diagonal tile d (lane j = column j) <- H d H; (v, tau) from slot xr
template <int B>
__device__ __forceinline__ void chase4_diag(double *Lw, const double (&d)[B], const int (&ad)[B], const int j, const int xr,
                                            const int zi, const int dumpi)
{
    double v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) v[i] = Lw[xr + i];
    const double tq = Lw[xr + B], vl = Lw[xr + j];
    double y0 = d[0], y1 = v[1] * d[1], y2 = v[2] * d[2], y3 = v[3] * d[3];
#pragma unroll
    for (int i = 4; i < B; i += 4) {
        y0 = fma(v[i], d[i], y0); y1 = fma(v[i + 1], d[i + 1], y1);
        y2 = fma(v[i + 2], d[i + 2], y2); y3 = fma(v[i + 3], d[i + 3], y3);
    }
    const double p = tq * ((y0 + y1) + (y2 + y3));
    const double vtp = rsumB<B>(vl * p);
    const double z = fma(-(0.5 * tq * vtp), vl, p);
    Lw[zi + j] = z;
    double zz[B];
#pragma unroll
    for (int i = 0; i < B; ++i) zz[i] = Lw[zi + i];
    // the lower triangle is what the band stores; the mirrored values go to the dump (a select on the address, no branch)
#pragma unroll
    for (int i = 0; i < B; ++i) ldsb(Lw, i >= j ? ad[i] : (dumpi + i) * 8) = fma(-zz[i], vl, fma(-v[i], z, d[i]));
}

// role 2: bulge tile of item (r0, c0): column c0 + j, rows r0 .. r0 + B - 1 (contiguous; no wrap inside a lane).  Column 0 holds
// (beta, 0, .., 0) already: lane 0 works on the dump
template <int B>
__device__ __forceinline__ void chase4_bulge(double *Lw, const int r0, const int c0, const int j, const int xr, const int dumpi)
{
    double v[B], b[B];
#pragma unroll
    for (int i = 0; i < B; ++i) v[i] = Lw[xr + i];
    const double tq = Lw[xr + B];
    int ab = (((c0 + j) & (WCOLS - 1)) << Rw<B>::WSH) + (r0 - c0 - j);
    if (j == 0) ab = dumpi;
#pragma unroll
    for (int i = 0; i < B; ++i) b[i] = Lw[ab + i];
    double w0 = b[0], w1 = v[1] * b[1], w2 = v[2] * b[2], w3 = v[3] * b[3];
#pragma unroll
    for (int i = 4; i < B; i += 4) {
        w0 = fma(v[i], b[i], w0); w1 = fma(v[i + 1], b[i + 1], w1);
        w2 = fma(v[i + 2], b[i + 2], w2); w3 = fma(v[i + 3], b[i + 3], w3);
    }
    const double cb = tq * ((w0 + w1) + (w2 + w3));
#pragma unroll
    for (int i = 0; i < B; ++i) Lw[ab + i] = fma(-cb, v[i], b[i]);
}

// Rows and columns of the band storage beyond the matrix are made zero before the rows kernel runs (wr = the rows of its window
// column), column npad (the first of the slack behind the band, ab_stride) included: the kernel's data-moving wave copies whole
// columns into LDS without looking at them, and takes column npad for every column further right.
__global__ void band_tail_zero_kernel(int n, int npad, int wr, double *__restrict__ ABall)
{
    double *AB = ABall + (size_t)blockIdx.x * ab_stride(npad);
    const int c_lo = n - (wr - 1) > 0 ? n - (wr - 1) : 0;
    for (int idx = threadIdx.x; idx < (npad + 1 - c_lo) * wr; idx += blockDim.x) {
        const int c = c_lo + idx / wr, d = idx % wr;
        if (c + d >= n) AB[(size_t)c * LD + d] = 0.0;
    }
}

// One step of the data-moving wave.  B columns enter the window per step by LDS-DMA (B = 16: four instructions of 1 KB, a column
// of the window is the first 32 rows of the band column, 256 bytes here and there; B = 8: one instruction, 128 bytes of each of 8
// columns; no registers, so no wait the compiler could place -- with register staging it made every step wait for the loads of
// the step before, HBM latency, and the data-moving wave was the longest of the step); they have landed MLAG steps later (the wait
// at the end of the step, a literal count: the counter is in order over loads and stores and a step issues 4 + 8 resp. 1 + 2 of
// them, one more with the poll) and are first touched two steps after that.  Columns leave the window with plain stores, at most
// B per step.
template <int B, int PH>
__device__ __forceinline__ void mover_step(double *Lw, double *__restrict__ AB, const int n, const int npad, const int LP,
                                           const int RP, const int RPn, const int RPold, const int ps, const int lane,
                                           const unsigned long long *pollp, unsigned long long *pubp, unsigned long long &pw,
                                           unsigned long long &ptmp, Sb16Ctl *C, int *status, const bool diag, long long (&dacc)[5])
{
    constexpr int WSH = Rw<B>::WSH;
    long long ts = diag ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define MV_STAMP(q) if (diag) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[q] += t_ - ts; ts = t_; }
    if (PH == 0 && pollp && ps > 0 && LP < n) {                    // the 4 B columns requested in these four steps, as the pass before left them
        const int need = LP + 4 * B < n ? LP + 4 * B : n;
        const unsigned long long want = ((unsigned long long)(ps - 1) << 32) + (unsigned)need;
        if (pw < want) s16_wait(pollp, want, C, status);            // pw was requested three steps ago
        else asm volatile("buffer_inv sc1" ::: "memory");
    }
    if (B == 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                               // columns LP + 4 q .. + 3: lane = (column lane >> 4, rows 2 (lane & 15), + 1)
            const int c = LP + 4 * q, cl = c + (lane >> 4);
            const double *src = AB + (size_t)(cl < npad ? cl : npad) * LD + 2 * (lane & 15);
            double *dst = Lw + ((c & (WCOLS - 1)) << WSH);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    } else {                                                        // columns LP .. LP + 7: lane = (column lane >> 3, rows 2 (lane & 7), + 1)
        const int cl = LP + (lane >> 3);
        const double *src = AB + (size_t)(cl < npad ? cl : npad) * LD + 2 * (lane & 7);
        double *dst = Lw + ((LP & (WCOLS - 1)) << WSH);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
    // The partner's progress word: requested here, looked at two steps later.  The load is hidden from the compiler (it would wait
    // for it with a count that also covers the DMA just issued); a value read too early is an OLDER progress, which only
    // sends the check above into its polling loop.
    if (PH == 1 && pollp) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(ptmp) : "v"(pollp) : "memory");
    MV_STAMP(1)                                                     // poll check and requests
    {
        // column-out lane: row md of the columns RP + mh, RP + mh + 64 / WR, ...
        const int md = lane & (Rw<B>::WR - 1), mh = lane >> WSH;
        const unsigned c0_ = (unsigned)(RP + mh), dumpo = (unsigned)npad * LD + 64 + lane;
#pragma unroll
        for (int i = 0; i < B * Rw<B>::WR / 64; ++i) {
            const unsigned c = c0_ + (64 / Rw<B>::WR) * i;
            const unsigned o = c < (unsigned)RPn ? c * LD + md : dumpo;
            AB[o] = Lw[((c & (WCOLS - 1)) << WSH) + md];
        }
    }
    MV_STAMP(0)                                                     // columns out: LDS reads, stores issued
    // All but the operations of the last MLAG steps complete: the columns requested MLAG steps ago are in the window, and the
    // stores of that step have reached the L2 -- the columns left of RPold, progress for the member that runs the next pass.  (Two
    // steps were not enough at B = 16: a step is shorter than half the latency of HBM and the wave waited for it every time.)
    if (B == 16) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    MV_STAMP(2)                                                     // the wait for the operations of MLAG steps ago
#undef MV_STAMP
    if (PH == 3 && pubp && lane == 0)
        __hip_atomic_store(pubp, ((unsigned long long)ps << 32) + (unsigned)RPold, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PH == 3 && pollp) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ptmp), hi = __builtin_amdgcn_readfirstlane((unsigned)(ptmp >> 32));
        pw = ((unsigned long long)hi << 32) | lo;
    }
}

// barrier of a chasing wave whose loads for the next step may still be in flight; its stores may not (LDS operations of a wave
// complete in order: with more than 15 (B = 16) resp. 7 (B = 8) load instructions behind the last store -- 16 / 24 resp. 8 / 12 in
// the code as compiled -- that many outstanding means the stores are done)
template <int B>
__device__ __forceinline__ void stores_done_barrier()
{
    if (B == 16) asm volatile("s_waitcnt lgkmcnt(15)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(7)\n\ts_barrier" ::: "memory");
}

// DIAG: the instrumented instance (BSP_SB2ST_DIAG; `diag` non-null); the other carries no stamp, no test for one
template <int B, bool DIAG = false>
__global__ __launch_bounds__(SB16R_THREADS) void sbr_rows_kernel(int n, int npad, int batch, double *__restrict__ ABall,
                                                                 double *__restrict__ dall, double *__restrict__ eall,
                                                                 long long *diag, Sb16Ctl *ctl, int P, int *status, int force_abort)
{
    using W = Rw<B>;
    constexpr int WSH = W::WSH, WR = W::WR, RPW = W::RPW, NSW = W::NSW, MLAG = W::MLAG, LP0 = W::LP0;
    extern __shared__ double lds[];
    double *Lw = lds;
    // the whole batch waits for the slowest channel: first on a SIMD shared with a long-running wave of another kernel (as the band
    // reduction's waves, crawford.hip)
    __builtin_amdgcn_s_setprio(3);
    long long dacc[5] = {0, 0, 0, 0, 0}, dt0 = 0;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & (B - 1), g = lane / B;
    // chasing waves: 0, 1 = roles 0, 1 of the first group of sweeps; 2, 3 = roles 0, 1 of the second; 4, 5 = role 2 of the two groups (the
    // waves with the long chains first, so that each has a SIMD where it is the only long one)
    const int grp = wv < 4 ? (wv >> 1) & 1 : wv & 1, role = wv < 4 ? wv & 1 : 2, sw = RPW * grp + g;
    const int zi = W::Z + 64 * grp + B * g, dumpi = W::DUMP + 64 * (wv % NCW4) + B * g;
    // ---- which channel; alone (ctl == nullptr) or member w of a ring of P workgroups
    int chn = blockIdx.x, w = 0, stride = 1;
    const unsigned long long *pollp = nullptr;
    unsigned long long *pubp = nullptr;
    Sb16Ctl *C = nullptr;
    if (ctl) {
        const int blk = blockIdx.x, grp = blk / (8 * P), rr = blk % (8 * P);
        chn = grp * 8 + (rr & 7); w = rr >> 3;
        if (chn >= batch) return;
        C = ctl + chn;
        int *modep = reinterpret_cast<int *>(lds + W::MODE);
        if (tid == 0) {
            const unsigned long long xcc = (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu);   // HW_REG_XCC_ID
            const unsigned long long mine = (0x10ull | xcc) << (8 * w);
            unsigned long long full = 0;
            for (int q = 0; q < P; ++q) full |= 0x10ull << (8 * q);
            unsigned long long v = atomicOr(&C->hs, mine) | mine;
            for (int spin = 0; !(v & (S16_COMMIT | S16_ABORT)); ++spin) {
                if (force_abort == 1) atomicCAS(&C->hs, v, v | S16_ABORT);
                else if ((v & full) == full) atomicCAS(&C->hs, v, v | S16_COMMIT);
                else if (spin > 400000) atomicCAS(&C->hs, v, v | S16_ABORT);
                else __builtin_amdgcn_s_sleep(4);
                v = __hip_atomic_load(&C->hs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int mode;                                     // 0 = exit, 1 = ring, 2 = alone
            if (v & S16_ABORT) mode = (w == 0) ? 2 : 0;
            else {
                bool same = true;
                for (int q = 1; q < P; ++q) same = same && (((v >> (8 * q)) & 0xfu) == (v & 0xfu));
                if (force_abort == 2) same = false;
                mode = same ? 1 : ((w == 0) ? 2 : 0);
            }
            *modep = mode;
        }
        __syncthreads();
        const int mode = __builtin_amdgcn_readfirstlane(*modep);
        __syncthreads();
        if (mode == 0) return;
        if (mode == 1 && P > 1) { stride = P; pollp = &C->prog[(w + P - 1) % P]; pubp = &C->prog[w]; }
        else w = 0;
    }
    double *AB = ABall + (size_t)chn * ab_stride(npad);
    const int npass = (n - 2 + NSW - 1) / NSW;
    for (int ps = w; ps < npass; ps += stride) {
        const int s0 = ps * NSW;
        int RP = s0, LP = s0 + LP0;                       // columns leave the window left of RP; LP is the next one to be requested
        if (pollp && ps > 0) {                            // the first LP0 + B columns of this pass, as the pass before left them
            if (tid == 0) {
                const int need = s0 + LP0 + B < n ? s0 + LP0 + B : n;
                s16_wait(pollp, ((unsigned long long)(ps - 1) << 32) + (unsigned)need, C, status);
            }
            __syncthreads();
            asm volatile("buffer_inv sc1" ::: "memory");
        }
        for (int idx = tid; idx < LP0 * WR; idx += SB16R_THREADS) {
            const int c = s0 + (idx >> WSH), d = idx & (WR - 1);
            Lw[((c & (WCOLS - 1)) << WSH) + d] = c + d < n ? AB[(size_t)c * LD + d] : 0.0;
        }
        __syncthreads();
        // the reflector of the pass's first item (sweep s0, k = 0) into its slot of step 0
        if (wv == 0 && g == 0) {
            const int hx = ((s0 & (WCOLS - 1)) << WSH) + 1 + j;
            next_reflector<B>(Lw, Lw[hx], j, W::XS, hx * 8);
        }
        const int nsteps = (n - s0 - 1 + B - 1) / B + LAG * (NSW - 1);
#define SB16R_RPN(t) \
        int RPn = s0 + NSW + B * ((t) - LAG * (NSW - 1) - 1);   /* columns leaving: left of the trailing sweep's tiles */ \
        if (RPn > n) RPn = n; \
        if (RPn < RP) RPn = RP;
        if (wv == NCW4) {                                 // the last wave moves data and does nothing else
            unsigned long long pw = 0, ptmp = 0;          // the partner's progress word as last seen / as last requested
            int RPold = RP;                               // RP MLAG steps ago
            __syncthreads();
#define SB16R_MSTEP(PH) \
            if (t + PH < nsteps) { \
                SB16R_RPN(t + PH) \
                { int r4 = s0 + NSW + B * ((t + PH - MLAG) - LAG * (NSW - 1) - 1);      /* RPn of MLAG steps ago */ \
                  RPold = r4 > n ? n : (r4 < s0 ? s0 : r4); } \
                mover_step<B, PH>(Lw, AB, n, npad, LP + B * PH, RP, RPn, RPold, ps, lane, pollp, pubp, pw, ptmp, C, status, DIAG, dacc); \
                if (DIAG) dt0 = (long long)__builtin_amdgcn_s_memtime(); \
                RP = RPn; \
                lds_only_barrier(); \
                if (DIAG) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[3] += t_ - dt0; dacc[4] += 1; } \
            }
            for (int t = 0; t < nsteps; t += 4) {
                SB16R_MSTEP(0) SB16R_MSTEP(1) SB16R_MSTEP(2) SB16R_MSTEP(3)
                LP += 4 * B;
            }
#undef SB16R_MSTEP
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LP = s0 + LP0;
        } else {
            // Roles 0 and 1 fetch the tile of their NEXT item at the end of a step, behind their stores: both tiles are final by
            // then (the lag of three leaves the diagonal tile of item k + 1 untouched from the step before item k runs, tools/proto_sbr.py;
            // the one element of the next tile that the sweep ahead reaches later -- the first element of ITS next column -- already
            // holds that column's beta, because the reflector is formed a step ahead here), so the loads of a step no longer stand
            // between the barrier and its arithmetic.  Role 0 also reads back the (v, tau) it has just written (a wave's LDS
            // operations are in order): its chain starts with the dot products.
            const int s4 = s0 + sw;
            const bool live = s4 < n - 2;
            const int xs0 = W::XS + sw * 2 * B;
            if (role == 0) {
                double bt[B], v[B], tq = 0.0;
                int at[B];
                for (int i = 0; i < B; ++i) { bt[i] = 0.0; v[i] = 0.0; at[i] = 0; }
                if (sw == 0 && live && s4 + 1 < n) {      // item (s0, 0) runs at step 0
                    next_tile_addr<B, false>(s4 + 1, j, at);
#pragma unroll
                    for (int i = 0; i < B; ++i) { bt[i] = ldsb(Lw, at[i] + (WR - 1) * 8 * i); v[i] = Lw[xs0 + i]; }
                    tq = Lw[xs0 + B];
                }
                __syncthreads();
                for (int t = 0; t < nsteps; ++t) {
                    if (DIAG) dt0 = (long long)__builtin_amdgcn_s_memtime();
                    const int k4 = t - LAG * sw, r04 = s4 + 1 + B * k4;   // item g of the wave: sweep s0 + sw, item t - 3 sw
                    const bool act = k4 >= 0 && live && r04 < n, actN = k4 >= -1 && live && r04 + B < n;
                    const int xw = xs0 + (((t + 1) & 1) << 8);
                    if (act) {
                        chase4_next<B>(Lw, bt, v, tq, at, j, xw);
                    } else if (k4 == -1 && live) {        // the sweep starts in the next step: its reflector from the band column
                        const int hx = ((s4 & (WCOLS - 1)) << WSH) + 1 + j;
                        next_reflector<B>(Lw, Lw[hx], j, xw, hx * 8);
                    }
                    // for the next step: (v, tau) back from the slot, the next item's tile (general addresses for the whole wave
                    // if the tile of one of its items wraps around the ring)
                    const bool wrapN = __builtin_amdgcn_ballot_w64(actN && ((r04 + B) & (WCOLS - 1)) > WCOLS - B) != 0;
                    if (actN) {
                        if (wrapN) next_tile_addr<B, false>(r04 + B, j, at);
                        else next_tile_addr<B, true>(r04 + B, j, at);
#pragma unroll
                        for (int i = 0; i < B; ++i) v[i] = Lw[xw + i];
                        tq = Lw[xw + B];
#pragma unroll
                        for (int i = 0; i < B; ++i) bt[i] = ldsb(Lw, at[i] + (WR - 1) * 8 * i);
                    }
                    if (DIAG) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[2] += t_ - dt0; dt0 = t_; }
                    if (__builtin_amdgcn_ballot_w64(actN) != 0) stores_done_barrier<B>();
                    else lds_only_barrier();
                    if (DIAG) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[3] += t_ - dt0; dacc[4] += 1; }
                }
            } else if (role == 1) {
                double d[B];
                int ad[B], offD[B];
                for (int i = 0; i < B; ++i) { d[i] = 0.0; ad[i] = 0; offD[i] = (i >= j ? (WR - 1) * j + i : (WR - 1) * i + j) * 8; }
                if (sw == 0 && live && s4 + 1 < n) {
                    diag_tile_addr<B, false>(s4 + 1, j, offD, ad);
#pragma unroll
                    for (int i = 0; i < B; ++i) d[i] = ldsb(Lw, ad[i]);
                }
                __syncthreads();
                for (int t = 0; t < nsteps; ++t) {
                    if (DIAG) dt0 = (long long)__builtin_amdgcn_s_memtime();
                    const int k4 = t - LAG * sw, r04 = s4 + 1 + B * k4;
                    const bool act = k4 >= 0 && live && r04 < n, actN = k4 >= -1 && live && r04 + B < n;
                    if (act) chase4_diag<B>(Lw, d, ad, j, xs0 + ((t & 1) << 8), zi, dumpi);
                    const bool wrapN = __builtin_amdgcn_ballot_w64(actN && ((r04 + B) & (WCOLS - 1)) > WCOLS - B) != 0;
                    if (actN) {
                        if (wrapN) diag_tile_addr<B, false>(r04 + B, j, offD, ad);
                        else diag_tile_addr<B, true>(r04 + B, j, offD, ad);
#pragma unroll
                        for (int i = 0; i < B; ++i) d[i] = ldsb(Lw, ad[i]);
                    }
                    if (DIAG) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[2] += t_ - dt0; dt0 = t_; }
                    if (__builtin_amdgcn_ballot_w64(actN) != 0) stores_done_barrier<B>();
                    else lds_only_barrier();
                    if (DIAG) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[3] += t_ - dt0; dacc[4] += 1; }
                }
            } else {
                __syncthreads();
                for (int t = 0; t < nsteps; ++t) {
                    if (DIAG) dt0 = (long long)__builtin_amdgcn_s_memtime();
                    const int k4 = t - LAG * sw, r04 = s4 + 1 + B * k4;
                    if (k4 > 0 && live && r04 < n) chase4_bulge<B>(Lw, r04, r04 - B, j, xs0 + ((t & 1) << 8), dumpi);
                    if (DIAG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[2] += t_ - dt0; dt0 = t_; }
                    lds_only_barrier();
                    if (DIAG) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); dacc[3] += t_ - dt0; dacc[4] += 1; }
                }
            }
        }
        if (wv != NCW4 && nsteps > 0) {                   // the chasing waves do not track RP step by step: its value after the last one
            SB16R_RPN(nsteps - 1)
            RP = RPn;
        }
        LP += 4 * B * ((nsteps + 3) / 4);                 // what was requested (far beyond n by the end of a pass)
#undef SB16R_RPN
        __syncthreads();
        int hi = LP < n ? LP : n;
        for (int idx = tid; idx < (hi - RP) * WR; idx += SB16R_THREADS) {
            const int c = RP + (idx >> WSH), d = idx & (WR - 1);
            AB[(size_t)c * LD + d] = Lw[((c & (WCOLS - 1)) << WSH) + d];
        }
        __syncthreads();
        if (pubp && tid == 0)
            __hip_atomic_store(pubp, ((unsigned long long)ps << 32) + (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (DIAG && blockIdx.x == 0 && lane == 0)
        for (int q = 0; q < 5; ++q) diag[wv * 5 + q] = dacc[q];
    if (stride > 1 && (npass - 1) % stride != w && npass > 0) return;   // the member of the last pass has seen every pass end
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");
    double *dd = dall + (size_t)chn * npad, *ee = eall + (size_t)chn * npad;
    for (int jj = tid; jj < n; jj += SB16R_THREADS) {
        dd[jj] = AB[(size_t)jj * LD];
        ee[jj] = jj < n - 1 ? AB[(size_t)jj * LD + 1] : 0.0;
    }
}

}  // namespace

int launch_sb2sb(int n, int npad, int batch, double *d_AB, hipStream_t st)
{
    static bool attr = false;
    if (!attr) {
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sb2sb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SB2SB_LDS));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sb2sb_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SB2SB2_LDS));
        attr = true;
    }
    const int mf = opts().sb2sb_mfma;
    if (opts().sb2st_diag) {
        int nb1 = 0, nb2 = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, reinterpret_cast<const void *>(sb2sb_mfma_kernel), 256, SB2SB2_LDS);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, reinterpret_cast<const void *>(sb2sb_kernel), 256, SB2SB_LDS);
        fprintf(stderr, "sb2sb: workgroups per CU by the occupancy query: mfma kernel %d (LDS %d B), first kernel %d (LDS %d B)\n", nb1,
                SB2SB2_LDS, nb2, SB2SB_LDS);
    }
    const int S = (n - NB1 + NB1 - 1) / NB1;                       // sweeps: j0 + 16 < n
    auto K = [n](int s) { const int m = n - (NB1 * s + NB1); return m > 0 ? (m + B1 - 1) / B1 : 0; };
    int tmax = -1;
    for (int s = 0; s < S; ++s)
        if (K(s) > 0 && K(s) - 1 + LAG * s > tmax) tmax = K(s) - 1 + LAG * s;
    for (int t = 0; t <= tmax; ++t) {
        int s_hi = t / LAG;
        if (s_hi > S - 1) s_hi = S - 1;
        if (t - LAG * s_hi >= K(s_hi)) continue;
        int s_lo = s_hi;
        while (s_lo > 0 && t - LAG * (s_lo - 1) < K(s_lo - 1)) --s_lo;
        KScope kt(KS_SB2SB, st);
        if (mf) hipLaunchKernelGGL(sb2sb_mfma_kernel, dim3(s_hi - s_lo + 1, batch), dim3(256), SB2SB2_LDS, st, n, npad, d_AB, t, s_lo);
        else hipLaunchKernelGGL(sb2sb_kernel, dim3(s_hi - s_lo + 1, batch), dim3(256), SB2SB_LDS, st, n, npad, d_AB, t, s_lo);
    }
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// hb = 16: band of half-width <= 16 (the dense route's second step); hb = 8: half-width <= 8 (the band route, crawford.hip), rows
// kernel with tiles of 8 only
int launch_sb16st(int n, int npad, int batch, double *d_AB, double *d_d, double *d_e, hipStream_t st, int *d_status, void *ctl, int hb)
{
    static bool attr = false;
    static int wg8 = 1;                                     // workgroups of the B = 8 kernel a CU holds (LDS: 2; registers permitting)
    if (!attr) {
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sb16st_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SB16_LDS));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sbr_rows_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Rw<16>::LDS));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sbr_rows_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Rw<8>::LDS));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sbr_rows_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Rw<16>::LDS));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sbr_rows_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    Rw<8>::LDS));
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(sbr_rows_kernel<8, false>), SB16R_THREADS,
                                                         Rw<8>::LDS) == hipSuccess && nb >= 1) wg8 = nb > 2 ? 2 : nb;
        attr = true;
    }
    if (hb != 16 && hb != 8) return BSP_ERR_ARG;
    const bool rows = hb == 8 || opts().sb16_rows != 0;
    const int nsw = hb == 8 ? Rw<8>::NSW : NW2;            // sweeps of a pass
    // workgroups per channel: as many as the chip has room for (B = 16: each needs a whole CU's LDS), at most 8
    int P;
    if (opts().sb2st_ring > 0) P = opts().sb2st_ring;
    else if (hb == 16) P = batch > 128 ? 1 : (batch > 64 ? 2 : (batch > 32 ? 4 : 8));
    else {
        int cus = 256;
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        // one workgroup per CU while that gives a channel four members, two per CU (what the LDS holds) beyond: measured at 128 /
        // 64 channels, chase in ms: four members on one CU each - / 17.7, two per CU 26.1 / 18.9 (eight members), two members 30.8 / -
        const int wgs = opts().sb8_wgs > 0 && opts().sb8_wgs < wg8 ? opts().sb8_wgs : wg8;
        for (P = 8; P > 1 && ((batch + 7) / 8) * 8 * P > cus; P /= 2) {}
        if (P < 4 && opts().sb8_wgs == 0)
            for (P = 4; P > 1 && ((batch + 7) / 8) * 8 * P > cus * wgs; P /= 2) {}
        else if (opts().sb8_wgs > 0)
            for (P = 8; P > 1 && ((batch + 7) / 8) * 8 * P > cus * wgs; P /= 2) {}
    }
    if (P > 8) P = 8;
    while (P > 1 && (n - 2) / nsw < 2 * P) P /= 2;         // fewer passes than members: nothing to share
    static Sb16Ctl *s_ctl = nullptr;
    static int cap = 0;
    Sb16Ctl *d_ctl = nullptr;
    static_assert(sizeof(Sb16Ctl) <= 112, "the caller's per-problem control block is sized for sb2st.hip's Sb8Ctl (112 bytes per channel)");
    if (P > 1 && ctl) {                                     // the problem's own control block (capi.hip: PipeBufs::sbctl)
        d_ctl = static_cast<Sb16Ctl *>(ctl);
        BSP_HIP(hipMemsetAsync(d_ctl, 0, (size_t)batch * sizeof(Sb16Ctl), st));
    }
    else if (P > 1) {
        if (cap < batch) {
            if (s_ctl) hipFree(s_ctl);
            BSP_HIP(hipMalloc(reinterpret_cast<void **>(&s_ctl), (size_t)batch * sizeof(Sb16Ctl)));
            cap = batch;
        }
        d_ctl = s_ctl;
        BSP_HIP(hipMemsetAsync(d_ctl, 0, (size_t)batch * sizeof(Sb16Ctl), st));
    }
    const int nblk = P > 1 ? ((batch + 7) / 8) * 8 * P : batch;
    const int fab = opts().sb2st_force_abort;
    auto launch = [&](long long *dbuf) {
        if (rows) hipLaunchKernelGGL(band_tail_zero_kernel, dim3(batch), dim3(256), 0, st, n, npad, 2 * hb, d_AB);
        if (hb == 8 && dbuf)
            hipLaunchKernelGGL((sbr_rows_kernel<8, true>), dim3(nblk), dim3(SB16R_THREADS), Rw<8>::LDS, st, n, npad, batch, d_AB, d_d, d_e, dbuf,
                               d_ctl, P, d_status, fab);
        else if (hb == 8)
            hipLaunchKernelGGL((sbr_rows_kernel<8, false>), dim3(nblk), dim3(SB16R_THREADS), Rw<8>::LDS, st, n, npad, batch, d_AB, d_d, d_e, dbuf,
                               d_ctl, P, d_status, fab);
        else if (rows && dbuf)
            hipLaunchKernelGGL((sbr_rows_kernel<16, true>), dim3(nblk), dim3(SB16R_THREADS), Rw<16>::LDS, st, n, npad, batch, d_AB, d_d, d_e, dbuf,
                               d_ctl, P, d_status, fab);
        else if (rows)
            hipLaunchKernelGGL((sbr_rows_kernel<16, false>), dim3(nblk), dim3(SB16R_THREADS), Rw<16>::LDS, st, n, npad, batch, d_AB, d_d, d_e, dbuf,
                               d_ctl, P, d_status, fab);
        else
            hipLaunchKernelGGL(sb16st_kernel, dim3(nblk), dim3(576), SB16_LDS, st, n, npad, batch, d_AB, d_d, d_e, dbuf, d_ctl, P,
                               d_status, fab);
    };
    if (opts().sb2st_diag) {                                // cycles per phase of the chasing waves (workgroup 0)
        long long *dbuf = nullptr, h[45];
        BSP_HIP(hipMalloc(reinterpret_cast<void **>(&dbuf), sizeof(h)));
        BSP_HIP(hipMemsetAsync(dbuf, 0, sizeof(h), st));
        launch(dbuf);
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) { hipFree(dbuf); BSP_HIP(le); }
        BSP_HIP(hipStreamSynchronize(st));
        BSP_HIP(hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost));
        hipFree(dbuf);
        fprintf(stderr, "band chase, tiles of %d: %d workgroup(s) per channel, %d per CU\n", hb, P, hb == 8 ? wg8 : 1);
        for (int w = 0; w < 9; ++w) {
            const double steps = h[w * 5 + 4] > 0 ? (double)h[w * 5 + 4] : 1.0;      // workgroup 0 may have left in mode 0: no steps
            fprintf(stderr, "sb16st wave %d: %lld steps; s_memtime ticks per step: chase item %.0f, rest + barrier %.0f (mover wave: columns out %.0f, block in %.0f)\n",
                    w, h[w * 5 + 4], (double)h[w * 5 + 2] / steps, (double)h[w * 5 + 3] / steps, (double)h[w * 5] / steps,
                    (double)h[w * 5 + 1] / steps);
        }
        return BSP_OK;
    }
    KScope kt(KS_SB16ST, st);
    launch(nullptr);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
