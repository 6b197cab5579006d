// gemm_f64.hip -- batched fp64 GEMM on v_mfma_f64_16x16x4_f64 (gfx950).
//
// Used by the dense->band stage (sy2sb.hip) for the panel-update GEMMs that replace the BLAS-3
// calls inside LAPACK DSYTRD's blocked path (DSYMM/DSYR2K-shaped work; reference call site
// matrices.f90:248 -> DSYGV -> DSYEV -> DSYTRD).
//
// Tiling: a workgroup of 4 wavefronts (256 threads) owns a BM x BN tile of C; the wavefronts
// are laid out WM x WN and each accumulates (BM/WM) x (BN/WN) as TM x TN MFMA tiles of 16x16
// (4 fp64 accumulators per lane per tile).  K is consumed in steps of BK = 16 through LDS:
// As[k][m], Bs[k][n] (k-major, row stride padded so that the four k-rows a ds_read_b64 touches
// fall in different bank halves).  MFMA f64 operand maps (guide section 3): A: lane l holds
// A[i=l&15][k=l>>4]; B: lane l holds B[k=l>>4][j=l&15]; C/D register r: row (l>>4)+4r, col l&15.
// The C store is widest when C is contiguous along j (sCn == 1); the host wrapper transposes the
// product (C^T = B^T A^T) when C is contiguous along i instead.
#include "common.h"

namespace bsp {

constexpr int BK = 16;

template <int BM, int BN, int WM, int WN, int ALAY, int BLAY>
__global__ __launch_bounds__(256) void gemm_kernel(GemmDesc g)
{
    constexpr int LDA = BM + 16, LDB = BN + 16;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    __shared__ double As[BK * LDA];
    __shared__ double Bs[BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // lower_only (kernel view): 1 keeps tiles that touch i >= j, 2 keeps tiles that touch j >= i
    if (g.lower_only == 1 && m0 + BM - 1 < n0) return;
    if (g.lower_only == 2 && n0 + BN - 1 < m0) return;
    const double *A = g.A + (long)blockIdx.z * g.bA;
    const double *B = g.B + (long)blockIdx.z * g.bB;
    double *C = g.C + (long)blockIdx.z * g.bC;

    double4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < g.K; k0 += BK) {
        // Staging loads are UNCONDITIONAL from clamped addresses, the value is selected afterwards:
        // a branch around a load makes hipcc wait vmcnt(0) per load (serialised round trips).
        // M, N are even and K is a multiple of 2 for every caller (checked in gemm_f64).
        // ---- stage A tile (BM x BK) ----
        if (ALAY == 0) {   // contiguous along m: double2 along m
#pragma unroll
            for (int it = 0; it < BK * BM / 2 / 256; ++it) {
                const int idx = tid + it * 256;
                const int kk = idx / (BM / 2), mm = (idx % (BM / 2)) * 2;
                const int gm = m0 + mm, gk = k0 + kk;
                const bool ok = (gk < g.K) && (gm + 1 < g.M);
                const double2 v = *reinterpret_cast<const double2 *>(A + (ok ? ((long)gm + (long)gk * g.sAk) : 0));
                *reinterpret_cast<double2 *>(&As[kk * LDA + mm]) = ok ? v : make_double2(0.0, 0.0);
            }
        } else {           // contiguous along k: double2 along k
#pragma unroll
            for (int it = 0; it < BM * BK / 2 / 256; ++it) {
                const int idx = tid + it * 256;
                const int mm = idx / (BK / 2), kk = (idx % (BK / 2)) * 2;
                const int gm = m0 + mm, gk = k0 + kk;
                const bool ok = (gm < g.M) && (gk + 1 < g.K);
                const double2 v = *reinterpret_cast<const double2 *>(A + (ok ? ((long)gm * g.sAm + (long)gk) : 0));
                As[kk * LDA + mm] = ok ? v.x : 0.0;
                As[(kk + 1) * LDA + mm] = ok ? v.y : 0.0;
            }
        }
        // ---- stage B tile (BK x BN) ----
        if (BLAY == 0) {   // contiguous along n
#pragma unroll
            for (int it = 0; it < BK * BN / 2 / 256; ++it) {
                const int idx = tid + it * 256;
                const int kk = idx / (BN / 2), nn = (idx % (BN / 2)) * 2;
                const int gn = n0 + nn, gk = k0 + kk;
                const bool ok = (gk < g.K) && (gn + 1 < g.N);
                const double2 v = *reinterpret_cast<const double2 *>(B + (ok ? ((long)gn + (long)gk * g.sBk) : 0));
                *reinterpret_cast<double2 *>(&Bs[kk * LDB + nn]) = ok ? v : make_double2(0.0, 0.0);
            }
        } else {           // contiguous along k
#pragma unroll
            for (int it = 0; it < BN * BK / 2 / 256; ++it) {
                const int idx = tid + it * 256;
                const int nn = idx / (BK / 2), kk = (idx % (BK / 2)) * 2;
                const int gn = n0 + nn, gk = k0 + kk;
                const bool ok = (gn < g.N) && (gk + 1 < g.K);
                const double2 v = *reinterpret_cast<const double2 *>(B + (ok ? ((long)gn * g.sBn + (long)gk) : 0));
                Bs[kk * LDB + nn] = ok ? v.x : 0.0;
                Bs[(kk + 1) * LDB + nn] = ok ? v.y : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < BK / 4; ++k4) {
            const int kr = k4 * 4 + (lane >> 4);
            double a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[kr * LDA + wm * (BM / WM) + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[kr * LDB + wn * (BN / WN) + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue ----
    const double alpha = g.alpha, beta = g.beta;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int gi = m0 + wm * (BM / WM) + i * 16 + (lane >> 4) + 4 * r;
                int gj = n0 + wn * (BN / WN) + j * 16 + (lane & 15);
                const bool ok = (gi < g.M && gj < g.N);
                double *p = C + (ok ? ((long)gi * g.sCm + (long)gj * g.sCn) : 0);
                double v = alpha * acc[i][j][r];
                if (beta != 0.0) v += beta * (*p);       // wave-uniform condition, unconditional load
                if (ok) *p = v;
            }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmDesc &g, int alay, int blay, hipStream_t st)
{
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch), block(256);
    if (alay == 0 && blay == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 0, 0>), grid, block, 0, st, g);
    else if (alay == 0 && blay == 1) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 0, 1>), grid, block, 0, st, g);
    else if (alay == 1 && blay == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, 0>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, 1>), grid, block, 0, st, g);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// Host wrapper: puts the memory-contiguous dimension of C on the MFMA's lane-fast (j) axis and
// picks a tile shape.  Requirements (checked): one of (sAm,sAk) and one of (sBk,sBn) is 1; all
// pointers 16-byte aligned and all non-unit strides / batch strides even (double2 staging loads).
int gemm_f64(const GemmDesc &gin, hipStream_t st)
{
    GemmDesc g = gin;
    if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return BSP_OK;
    if (g.sCm == 1 && g.sCn != 1) {
        // C^T = B^T A^T : kernel-A(i',k) = B(k,i'), kernel-B(k,j') = A(j',k)
        GemmDesc t = g;
        t.M = g.N; t.N = g.M;
        t.A = g.B; t.sAm = g.sBn; t.sAk = g.sBk; t.bA = g.bB;
        t.B = g.A; t.sBk = g.sAk; t.sBn = g.sAm; t.bB = g.bA;
        t.sCm = g.sCn; t.sCn = g.sCm;
        // the caller's lower triangle (i >= j) is the transposed view's upper triangle (j' >= i')
        g = t;
        if (gin.lower_only) g.lower_only = 2;
    }
    int alay, blay;
    if (g.sAm == 1) alay = 0; else if (g.sAk == 1) alay = 1; else return BSP_ERR_ARG;
    if (g.sBn == 1) blay = 0; else if (g.sBk == 1) blay = 1; else return BSP_ERR_ARG;
    auto even = [](long v) { return (v & 1) == 0; };
    if ((g.M & 1) || (g.N & 1) || (g.K & 1)) return BSP_ERR_ARG;
    if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || !even(g.bA) || !even(g.bB)) return BSP_ERR_ARG;
    if ((alay == 0 && !even(g.sAk)) || (alay == 1 && !even(g.sAm))) return BSP_ERR_ARG;
    if ((blay == 0 && !even(g.sBk)) || (blay == 1 && !even(g.sBn))) return BSP_ERR_ARG;
    if (g.M >= 128 && g.N >= 128) return launch_cfg<128, 128, 2, 2>(g, alay, blay, st);
    if (g.N >= 128) return launch_cfg<64, 128, 2, 2>(g, alay, blay, st);
    if (g.M >= 128) return launch_cfg<128, 64, 2, 2>(g, alay, blay, st);
    return launch_cfg<64, 64, 2, 2>(g, alay, blay, st);
}

}  // namespace bsp
