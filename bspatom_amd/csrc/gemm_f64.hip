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

// One k-step (depth 4) of a (16 TM) x (16 TN) wave tile: acc[i][j] += A(16 rows of block i, 4) * B(4, 16 cols of block j).
// Arow / Brow: LDS row k0 + (lane >> 4) of this wave's A / B tile.  (A core on v_mfma_f64_4x4x4, whose layouts nest in
// these -- tools/microbench/mfma4_probe.hip -- and which a register-only loop runs at 72 TFLOP/s against 36-48 for
// 16x16x4 -- tools/microbench/mfma_f64_peak.hip -- was tried: 20 instead of 8 LDS reads per step, and the kernels
// came out 2-5 % SLOWER; they are not bound by the matrix pipe.)
// LDS image of a staged tile: row k holds its columns permuted inside every aligned group of 16, column x at x ^ lds_swz(k).
// The row stride (BX + 16 doubles) puts rows k and k + 2 on the same banks; an operand that is contiguous along k in memory is
// stored with eight lanes of a 16-lane group on rows k, k + 2, .., k + 14 of ONE column -- an eight-way bank conflict per
// ds_write_b64 without the permutation (counters, profiles/r03_lds_util.json: 63 % of symm's LDS-array cycles and 70 % of
// gemm_kernel<64,64>'s were conflict cycles), none with it: the eight rows land on eight different even offsets.  A fragment
// read takes the 16 columns of a group in permuted order (the same banks); pairs of columns stay pairs (the offset is even).
__device__ __forceinline__ int lds_swz(int k) { return ((k >> 1) & 7) << 1; }

// SW = false: no operand of the kernel is staged with the transposed store (the rank-128 update): plain rows
template <int TM, int TN, bool SW = true>
__device__ __forceinline__ void mfma_step(const double *Arow, const double *Brow, int lane, int kr, double4_t (&acc)[TM][TN])
{
    double a[TM], b[TN];
    const int c = SW ? (lane & 15) ^ lds_swz(kr) : (lane & 15);
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = Arow[i * 16 + c];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = Brow[j * 16 + c];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
}


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
    const int zb = (g.ksplit > 1) ? (int)(blockIdx.z % g.batch) : (int)blockIdx.z;
    const int zs = (g.ksplit > 1) ? (int)(blockIdx.z / g.batch) : 0;
    const double *A = g.A + (long)zb * g.bA;
    const double *B = g.B + (long)zb * g.bB;
    double *C = g.C + (long)zb * g.bC + (long)zs * g.bCs;
    const int kbeg = (g.ksplit > 1) ? zs * g.kchunk : 0;
    const int Kend = (g.ksplit > 1) ? ((kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K) : g.K;

    double4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    for (int k0 = kbeg; k0 < Kend; k0 += BK) {
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
                const bool ok = (gk < Kend) && (gm + 1 < g.M);
                const double2 v = *reinterpret_cast<const double2 *>(A + (ok ? ((long)gm + (long)gk * g.sAk) : 0));
                *reinterpret_cast<double2 *>(&As[kk * LDA + (mm ^ lds_swz(kk))]) = ok ? v : make_double2(0.0, 0.0);
            }
        } else {           // contiguous along k: double2 along k
#pragma unroll
            for (int it = 0; it < BM * BK / 2 / 256; ++it) {
                const int idx = tid + it * 256;
                const int mm = idx / (BK / 2), kk = (idx % (BK / 2)) * 2;
                const int gm = m0 + mm, gk = k0 + kk;
                const bool ok = (gm < g.M) && (gk + 1 < Kend);
                const double2 v = *reinterpret_cast<const double2 *>(A + (ok ? ((long)gm * g.sAm + (long)gk) : 0));
                As[kk * LDA + (mm ^ lds_swz(kk))] = ok ? v.x : 0.0;
                As[(kk + 1) * LDA + (mm ^ lds_swz(kk))] = ok ? v.y : 0.0;         // kk is even: the same offset
            }
        }
        // ---- stage B tile (BK x BN) ----
        if (BLAY == 0) {   // contiguous along n
#pragma unroll
            for (int it = 0; it < BK * BN / 2 / 256; ++it) {
                const int idx = tid + it * 256;
                const int kk = idx / (BN / 2), nn = (idx % (BN / 2)) * 2;
                const int gn = n0 + nn, gk = k0 + kk;
                const bool ok = (gk < Kend) && (gn + 1 < g.N);
                const double2 v = *reinterpret_cast<const double2 *>(B + (ok ? ((long)gn + (long)gk * g.sBk) : 0));
                *reinterpret_cast<double2 *>(&Bs[kk * LDB + (nn ^ lds_swz(kk))]) = ok ? v : make_double2(0.0, 0.0);
            }
        } else {           // contiguous along k
#pragma unroll
            for (int it = 0; it < BN * BK / 2 / 256; ++it) {
                const int idx = tid + it * 256;
                const int nn = idx / (BK / 2), kk = (idx % (BK / 2)) * 2;
                const int gn = n0 + nn, gk = k0 + kk;
                const bool ok = (gn < g.N) && (gk + 1 < Kend);
                const double2 v = *reinterpret_cast<const double2 *>(B + (ok ? ((long)gn * g.sBn + (long)gk) : 0));
                Bs[kk * LDB + (nn ^ lds_swz(kk))] = ok ? v.x : 0.0;
                Bs[(kk + 1) * LDB + (nn ^ lds_swz(kk))] = ok ? v.y : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < BK / 4; ++k4) {
            const int kr = k4 * 4 + (lane >> 4);
            mfma_step<TM, TN>(&As[kr * LDA + wm * (BM / WM)], &Bs[kr * LDB + wn * (BN / WN)], lane, kr, acc);
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
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch * (g.ksplit > 1 ? g.ksplit : 1)), block(256);
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


// ------------------------------------------------------------------------------------------------
// Tall-skinny times small square: C[b] (m x 64) = alpha * A[b] (m x 64) * B[b] (64 x 64) + beta * C[b], all
// column-major (W = V T and Z = Y - 1/2 V K of sy2sb: twice per panel, on the stage's critical chain, and
// memory-bound: 16 flop/B).  The general kernel stages A through LDS k-step by k-step with two barriers each;
// here B (32 KB) is the only thing in LDS and every wave reads its 32 x 64 slab of A straight into MFMA operand
// registers -- lane l holds A(row l & 15, column l >> 4), i.e. one instruction fetches four 128-byte column
// segments -- with all 32 loads in flight before the first MFMA.  One workgroup = 128 rows.
__global__ __launch_bounds__(256) void tsmm64_kernel(int m, const double *__restrict__ Aall, long lda, long bsA,
                                                    const double *__restrict__ Ball, long bsB, double *Call, long ldc,
                                                    long bsC, double alpha, double beta)
{
    __shared__ double Bs[64][64 + 2];                                  // Bs[k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *A = Aall + (long)blockIdx.y * bsA;
    const double *B = Ball + (long)blockIdx.y * bsB;
    double *C = Call + (long)blockIdx.y * bsC;
    const int r0 = blockIdx.x * 128 + wave * 32;                       // this wave's 32 rows
    for (int idx = tid; idx < 64 * 64; idx += 256) { const int kk = idx & 63, nn = idx >> 6; Bs[kk][nn] = B[kk + 64 * nn]; }
    // A fragments: a[i][q] = A(r0 + 16 i + (lane & 15), 4 q + (lane >> 4)); rows beyond m read row 0 (masked at the store)
    double a[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = r0 + 16 * i + (lane & 15);
        const double *ap = A + (row < m ? row : 0) + (long)(lane >> 4) * lda;
#pragma unroll
        for (int q = 0; q < 16; ++q) a[i][q] = ap[(long)(4 * q) * lda];
    }
    double4_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        double b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[4 * q + (lane >> 4)][16 * j + (lane & 15)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i][q], acc[i][j], 0, 0, 0);
    }
    // The product is formed TRANSPOSED (C^T = B^T A^T: the B fragment as the MFMA's first operand, the A fragment as its
    // second; the two register layouts are each other's transposes), so that acc[i][j][r] = C(r0 + 16 i + (lane & 15),
    // 16 j + (lane >> 4) + 4 r): the 16 lanes of a DPP row store 16 consecutive rows of one column of the
    // column-major C -- whole 128-byte lines
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 16 * i + (lane & 15), col = 16 * j + (lane >> 4) + 4 * r;
                if (row < m) {
                    double *p = C + row + (long)col * ldc;
                    double v = alpha * acc[i][j][r];
                    if (beta != 0.0) v += beta * (*p);
                    *p = v;
                }
            }
}

int tsmm64_f64(int m, int batch, const double *A, long lda, long bsA, const double *B, long bsB, double *C, long ldc, long bsC,
               double alpha, double beta, hipStream_t st)
{
    if (m <= 0 || batch <= 0) return BSP_OK;
    hipLaunchKernelGGL(tsmm64_kernel, dim3((m + 127) / 128, batch), dim3(256), 0, st, m, A, lda, bsA, B, bsB, C, ldc, bsC, alpha,
                       beta);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// sum of the split-K slices in slice order: C(i,j) = sum_s part[s][b][i*N + j] (+ beta * C)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const double *__restrict__ part, int splits, int batch, int M, int N,
                                                           double *Call, long sCm, long sCn, long bC, double beta)
{
    const int b = blockIdx.y;
    const long mn = (long)M * N;
    for (long idx = blockIdx.x * 256 + threadIdx.x; idx < mn; idx += (long)gridDim.x * 256) {
        double s = 0.0;
        for (int q = 0; q < splits; ++q) s += part[((long)q * batch + b) * mn + idx];
        const int i = (int)(idx / N), j = (int)(idx % N);
        double *p = Call + (long)b * bC + (long)i * sCm + (long)j * sCn;
        *p = (beta != 0.0) ? s + beta * (*p) : s;
    }
}

int gemm_splitk_f64(const GemmDesc &gin, int splits, double *part, hipStream_t st)
{
    int chunk = ((gin.K + splits - 1) / splits + BK - 1) / BK * BK;
    if (chunk < 4 * BK) chunk = 4 * BK;                               // do not split below 64-deep slices
    const int ns = (gin.K + chunk - 1) / chunk;
    if (ns <= 1 || !part) return gemm_f64(gin, st);
    GemmDesc g = gin;
    g.C = part; g.sCm = gin.N; g.sCn = 1; g.bC = (long)gin.M * gin.N; g.bCs = (long)gin.batch * gin.M * gin.N;
    g.beta = 0.0; g.ksplit = ns; g.kchunk = chunk;
    int rc = gemm_f64(g, st);
    if (rc) return rc;
    const long mn = (long)gin.M * gin.N;
    dim3 grid((unsigned)((mn + 255) / 256 < 64 ? (mn + 255) / 256 : 64), gin.batch);
    hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, st, part, ns, gin.batch, gin.M, gin.N, gin.C, gin.sCm, gin.sCn,
                       gin.bC, gin.beta);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// ================================================================================================
// Pipelined kernels for the two big products of sy2sb (per panel):
//   MODE 1 "syr2k": A22 -= [V Z] [Z V]^T     (K = 128)
//   MODE 2 "symm" : Y = A22 W                 (K = m)
// Both work on A22 in column-major storage of which only the 64x64 blocks (I, J) with J <= I+1
// (lower triangle + first block super-diagonal) are kept valid: MODE 1 skips every 128x128 tile
// whose column block is more than one above its row block (45% of the flops and of the HBM traffic
// of the full update are saved), MODE 2 reads the invalid part through the transposed address.
// The invariant survives the 64-row shift of the trailing matrix from panel to panel because it is
// stated in 64-blocks while the tiles are 128 wide (derivation in DESIGN.md section 4).
//
// Pipeline per K-step (BK = 16): global loads of step t+1 go to registers before the MFMAs of
// step t; they are written to the other LDS buffer after the MFMAs; one LDS-only barrier per step.
// With beta != 0 the C tile is loaded straight into the accumulators ((beta/alpha) C) while the
// first tiles are staged, so the epilogue is a pure store.
// ================================================================================================
__device__ __forceinline__ void lds_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// timing experiments only (BSP_GEMM_DIAG): cycle stamps of one workgroup's phases, summed over its life
__device__ long long g_gemm2_diag[8];
#define G2_STAMP(k) if (DIAGG) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dacc[k] += t_ - tl; tl = t_; }

// tile staging: element (x, k) of an X-by-K operand; LAY 0: contiguous along x, stride sk along k;
// LAY 1: contiguous along k, stride sx along x.  NR = double2 registers per thread.
template <int BX, int LAY>
__device__ __forceinline__ void tile_load(double2 (&r)[BK * BX / 512], const double *__restrict__ base, long sx,
                                          long sk, int x0, int k0, int X, int K, int tid)
{
#pragma unroll
    for (int it = 0; it < BK * BX / 512; ++it) {
        const int idx = tid + it * 256;
        if (LAY == 0) {
            const int kk = idx / (BX / 2), xx = (idx % (BX / 2)) * 2;
            const int gx = x0 + xx, gk = k0 + kk;
            const bool ok = (gk < K) && (gx + 1 < X);
            const double2 v = *reinterpret_cast<const double2 *>(base + (ok ? ((long)gx + (long)gk * sk) : 0));
            r[it] = ok ? v : make_double2(0.0, 0.0);
        } else {
            const int xx = idx / (BK / 2), kk = (idx % (BK / 2)) * 2;
            const int gx = x0 + xx, gk = k0 + kk;
            const bool ok = (gx < X) && (gk + 1 < K);
            const double2 v = *reinterpret_cast<const double2 *>(base + (ok ? ((long)gx * sx + (long)gk) : 0));
            r[it] = ok ? v : make_double2(0.0, 0.0);
        }
    }
}

template <int BX, int LAY, bool SW = true>
__device__ __forceinline__ void tile_store(const double2 (&r)[BK * BX / 512], double *S, int tid)
{
    constexpr int LD = BX + 16;
    static_assert(SW || LAY == 0, "the transposed store always permutes the columns: a kernel that reads plain rows (SW = false) may only stage along x");
#pragma unroll
    for (int it = 0; it < BK * BX / 512; ++it) {
        const int idx = tid + it * 256;
        if (LAY == 0) {
            const int kk = idx / (BX / 2), xx = (idx % (BX / 2)) * 2;
            *reinterpret_cast<double2 *>(&S[kk * LD + (SW ? xx ^ lds_swz(kk) : xx)]) = r[it];
        } else {
            const int xx = idx / (BK / 2), kk = (idx % (BK / 2)) * 2;        // kk is even: rows kk and kk + 1 share the offset
            S[kk * LD + (xx ^ lds_swz(kk))] = r[it].x;
            S[(kk + 1) * LD + (xx ^ lds_swz(kk))] = r[it].y;
        }
    }
}

// The same staging with everything that does not change from K-step to K-step taken out of the loop: per thread
// the BYTE offset of each of its double2 elements relative to the operand's position at k0 (32 bits: an operand of
// sy2sb spans < 4 GB), so that a K-step's loads are `global_load_dwordx4 v, voffset, s[base]` with a uniform base
// that advances by a constant -- no 64-bit multiplies, no bounds branches in the loop (the kernels built on this
// spent ~250 instructions per K-step beside their 32-64 MFMAs).  Rows beyond X are CLAMPED instead of zero-filled:
// such a row only feeds output rows that are never stored (callers guarantee K % BK == 0, X even).  Used by MODE 2
// (symm: 132.8 -> 120.4 ms standalone).  MODE 1 keeps the plain form: with the offsets hoisted the compiler parks
// the staging registers in scratch around the MFMA block (236 ms), and staging through LDS-DMA loads instead
// (global_load_lds_dwordx4, no staging registers at all) measured 148 against 145.5 ms.
template <int BX, int LAY>
__device__ __forceinline__ void tile_offsets(unsigned (&off)[BK * BX / 512], long sx, long sk, int x0, int X, int tid)
{
#pragma unroll
    for (int it = 0; it < BK * BX / 512; ++it) {
        const int idx = tid + it * 256;
        if (LAY == 0) {
            const int kk = idx / (BX / 2), xx = (idx % (BX / 2)) * 2;
            int gx = x0 + xx; gx = (gx + 1 < X) ? gx : X - 2;
            off[it] = (unsigned)(((long)gx + (long)kk * sk) * 8);
        } else {
            const int xx = idx / (BK / 2), kk = (idx % (BK / 2)) * 2;
            int gx = x0 + xx; gx = (gx < X) ? gx : X - 1;
            off[it] = (unsigned)(((long)gx * sx + (long)kk) * 8);
        }
    }
}

#ifndef G2_NT
#define G2_NT 1
#endif
// streaming (non-temporal) 16-byte load: the A22 tiles of symm pass through once per use and should not push W and
// the panels out of the L2
__device__ __forceinline__ double2 g2_ldnt(const char *p)
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t v = G2_NT ? __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p)) : *reinterpret_cast<const d2_t *>(p);
    return make_double2(v.x, v.y);
}
// kernel view: C'(i', j') with j' memory-contiguous (sCn == 1 required).
template <int BM, int BN, int ALAY, int BLAY, int MODE, int DIAGG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm2_kernel(GemmDesc g)
{
    long long dacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tl = DIAGG ? (long long)__builtin_amdgcn_s_memtime() : 0;
    constexpr int WM = 2, WN = 2;
    constexpr int LDA = BM + 16, LDB = BN + 16;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool SW = (MODE != 1);                       // MODE 1 stages both operands along x: no transposed store, plain rows
    __shared__ double As[2][BK * LDA];
    __shared__ double Bs[2][BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (MODE == 1) {
        // 1-D grid over the VALID tiles only (column block by <= row block bx + 1), laid out XCD-aware: workgroups go
        // round-robin to the 8 XCDs by linear id, so id & 7 is the XCD; XCD x takes the channels z = x, x+8, ... one
        // after the other, row by row.  All tiles of a channel then share one L2 and its [V|Z|V] panels are fetched
        // from HBM once instead of once per XCD (PMC: 1.8x the algorithmic bytes before).
        const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
        const int nb = (g.N + BN - 1) / BN, T = g.lower_only;           // T: valid tiles per channel (from the launcher)
        const int zi = q / T;
        int t = __builtin_amdgcn_readfirstlane(q - zi * T + g.toff);    // uniform: the table below is read through the scalar cache
        bz = xcd + 8 * zi;
        if (bz >= g.batch) return;
        const int ylo = g.yoff, part1 = (g.yoff == 0 && g.nsb > 0 && g.sbpre[g.nsb - 1] == nb);   // part 1 of the look-ahead: by == 0 only
        // Within a channel the tiles go in 8 x 8 SUPER-BLOCKS (row-major inside): the 64 workgroups an XCD holds at a
        // time then share 8 row slices and 8 column slices of the [V|Z|V] panels (2 MB, L2-resident) instead of one
        // row slice and up to 32 column slices that each serve a single tile before the streaming C tiles evict them.
        int sb = 0;
        while (t >= g.sbpre[sb]) ++sb;
        if (sb) t -= g.sbpre[sb - 1];
        const int sxy = g.sbxy[sb];
        const int sy0 = 8 * (sxy >> 8);
        bx = 8 * (sxy & 255);
        for (;;) {
            const int yhi = part1 ? 0 : ((bx + 1 < nb - 1) ? bx + 1 : nb - 1);
            const int lo = (ylo > sy0) ? ylo : sy0, hi = (yhi < sy0 + 7) ? yhi : sy0 + 7;
            const int c = hi - lo + 1;
            if (c > 0 && t < c) { by = lo + t; break; }
            if (c > 0) t -= c;
            ++bx;
        }
    }
    if (MODE == 2) {
        // same XCD-aware layout for the SYMM-shaped product: all row tiles of a channel on one XCD, so that W (and
        // the A22 tiles that are read a second time through the transposed address) are shared in its L2
        const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
        const int nbx = (g.N + BN - 1) / BN;
        const int zi = q / nbx;
        bx = q - zi * nbx; by = 0;
        bz = xcd + 8 * zi;
        if (bz >= g.batch) return;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const double *A = g.A + (long)bz * g.bA;
    const double *B = g.B + (long)bz * g.bB;
    double *C = g.C + (long)bz * g.bC;
    const double alpha = g.alpha, beta = g.beta;

    const int nk = (g.K + BK - 1) / BK;
    double2 ra[BK * BM / 512], rb[BK * BN / 512];
    double2 ra1[(MODE == 2) ? BK * BM / 512 : 1], rb1[(MODE == 2) ? BK * BN / 512 : 1];   // MODE 2: second staging set
    // MODE 2: B is the symmetric A22; K-steps beyond the diagonal block of this row tile read A22^T
    const int ksw = (MODE == 2) ? (n0 + 128) : 0x7fffffff;
    constexpr int NRA = BK * BM / 512, NRB = BK * BN / 512;
    unsigned offA[NRA], offB[NRB], offT[NRB];
    tile_offsets<BM, ALAY>(offA, g.sAm, g.sAk, m0, g.M, tid);
    tile_offsets<BN, BLAY>(offB, g.sBn, g.sBk, n0, g.N, tid);
    if (MODE == 2) tile_offsets<BN, 1>(offT, g.sBk, 1, n0, g.N, tid);                             // A22(k,i): stride ld along i
    const long ksA = (ALAY == 0) ? g.sAk * 8 : 8, ksB = (BLAY == 0) ? g.sBk * 8 : 8;           // bytes per unit of k
#define G2_LOAD_A(k0_, ra)                                                                                    \
    {                                                                                                        \
        const char *kb_ = reinterpret_cast<const char *>(A) + (long)(k0_) * ksA;                             \
        _Pragma("unroll") for (int it = 0; it < NRA; ++it) ra[it] = *reinterpret_cast<const double2 *>(kb_ + offA[it]); \
    }
#define G2_LOAD_B(k0_, rb)                                                                                    \
    {                                                                                                        \
        if (MODE == 2 && (k0_) >= ksw) {                                                                     \
            const char *kb_ = reinterpret_cast<const char *>(B) + (long)(k0_) * 8;                           \
            _Pragma("unroll") for (int it = 0; it < NRB; ++it) rb[it] = g2_ldnt(kb_ + offT[it]);            \
        } else {                                                                                             \
            const char *kb_ = reinterpret_cast<const char *>(B) + (long)(k0_) * ksB;                         \
            _Pragma("unroll") for (int it = 0; it < NRB; ++it) rb[it] = g2_ldnt(kb_ + offB[it]);            \
        }                                                                                                    \
    }
#define G2_STORE_B(k0_, S_, rb)                                              \
    {                                                                        \
        if constexpr (MODE == 2) {                                           \
            if ((k0_) >= ksw) tile_store<BN, 1, true>(rb, S_, tid);          \
            else tile_store<BN, BLAY, SW>(rb, S_, tid);                      \
        } else tile_store<BN, BLAY, SW>(rb, S_, tid);                        \
    }
    if (MODE == 1) {
        tile_load<BM, ALAY>(ra, A, g.sAm, g.sAk, m0, 0, g.M, g.K, tid);
        tile_load<BN, BLAY>(rb, B, g.sBn, g.sBk, n0, 0, g.N, g.K, tid);
    } else {
        G2_LOAD_A(0, ra)
        G2_LOAD_B(0, rb)
        if (nk > 1) {                                  // MODE 2 prefetches two K-steps ahead (second register set)
            G2_LOAD_A(BK, ra1)
            G2_LOAD_B(BK, rb1)
        }
    }

    double4_t acc[TM][TN];
    // a wave's sub-tile is either completely inside C or (the last tile of an odd number of 64-blocks) completely
    // outside in one direction: the C accesses of the inside case need no per-element bounds branch and share 4 TM
    // row pointers (row (lane >> 4) + 4 r + 16 i, column offsets 128 j bytes as immediates)
    const int wr0 = m0 + wm * (BM / WM), wc0 = n0 + wn * (BN / WN);
    const bool inside = (wr0 + BM / WM <= g.M) && (wc0 + BN / WN <= g.N);
    double *Cl = C + (long)(wr0 + (lane >> 4)) * g.sCm + (long)(wc0 + (lane & 15));
    if (beta != 0.0 && inside) {
        const double sc = beta / alpha;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double *pr = Cl + (long)(16 * i + 4 * r) * g.sCm;
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] = sc * (G2_NT ? __builtin_nontemporal_load(&pr[16 * j]) : pr[16 * j]);
            }
    } else if (beta != 0.0) {
        const double sc = beta / alpha;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = m0 + wm * (BM / WM) + i * 16 + (lane >> 4) + 4 * r;
                    const int gj = n0 + wn * (BN / WN) + j * 16 + (lane & 15);
                    const bool ok = (gi < g.M && gj < g.N);
                    const double v = C[ok ? ((long)gi * g.sCm + (long)gj) : 0];
                    acc[i][j][r] = ok ? sc * v : 0.0;
                }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
    G2_STAMP(0)                                              // C, first A/B tiles: loads issued
    if (DIAGG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); G2_STAMP(1) }   // ... and arrived
    tile_store<BM, ALAY, SW>(ra, As[0], tid);
    G2_STORE_B(0, Bs[0], rb)
    lds_barrier2();
    G2_STAMP(2)

    if constexpr (MODE == 2) {
        // Two K-steps of prefetch: while tile t is multiplied, tile t+1 is in flight or in registers (set S1) and the
        // loads of tile t+2 are issued into the set tile t just left (S0); a K-step here is only 32 MFMAs per wave
        // (0.85 us), less than the memory latency under load, so one step of distance left a wait in every step.
#define G2_STEP2(t_, S0a, S0b, S1a, S1b)                                                                     \
        {                                                                                                    \
            const int cur = (t_) & 1;                                                                        \
            if ((t_) + 2 < nk) {                                                                             \
                G2_LOAD_A(((t_) + 2) * BK, S0a)                                                              \
                G2_LOAD_B(((t_) + 2) * BK, S0b)                                                              \
            }                                                                                                \
            _Pragma("unroll") for (int k4 = 0; k4 < BK / 4; ++k4) {                                          \
                const int kr = k4 * 4 + (lane >> 4);                                                         \
                mfma_step<TM, TN, SW>(&As[cur][kr * LDA + wm * (BM / WM)], &Bs[cur][kr * LDB + wn * (BN / WN)], lane, kr, acc); \
            }                                                                                                \
            if ((t_) + 1 < nk) {                                                                             \
                tile_store<BM, ALAY, SW>(S1a, As[cur ^ 1], tid);                                                 \
                G2_STORE_B(((t_) + 1) * BK, Bs[cur ^ 1], S1b)                                                \
            }                                                                                                \
            lds_barrier2();                                                                                  \
        }
        int t = 0;
        for (; t + 1 < nk; t += 2) {
            G2_STEP2(t, ra, rb, ra1, rb1)
            G2_STEP2(t + 1, ra1, rb1, ra, rb)
        }
        if (t < nk) G2_STEP2(t, ra, rb, ra1, rb1)
    } else {
    for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk) {
                if (MODE == 1) {
                    tile_load<BM, ALAY>(ra, A, g.sAm, g.sAk, m0, (t + 1) * BK, g.M, g.K, tid);
                    tile_load<BN, BLAY>(rb, B, g.sBn, g.sBk, n0, (t + 1) * BK, g.N, g.K, tid);
                } else {
                    G2_LOAD_A((t + 1) * BK, ra)
                    G2_LOAD_B((t + 1) * BK, rb)
                }
            }
    #pragma unroll
            for (int k4 = 0; k4 < BK / 4; ++k4) {
                const int kr = k4 * 4 + (lane >> 4);
                mfma_step<TM, TN, SW>(&As[cur][kr * LDA + wm * (BM / WM)], &Bs[cur][kr * LDB + wn * (BN / WN)], lane, kr, acc);
            }
            G2_STAMP(3)                                          // MFMAs of this k-tile (and issue of the next loads)
            if (t + 1 < nk) {
                if (DIAGG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); G2_STAMP(4) }   // wait for the next tile
                tile_store<BM, ALAY, SW>(ra, As[cur ^ 1], tid);
                G2_STORE_B((t + 1) * BK, Bs[cur ^ 1], rb)
            }
            lds_barrier2();
            G2_STAMP(5)
        }
}
    if (inside) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *pr = Cl + (long)(16 * i + 4 * r) * g.sCm;
#pragma unroll
                for (int j = 0; j < TN; ++j) { if (G2_NT) __builtin_nontemporal_store(alpha * acc[i][j][r], &pr[16 * j]); else pr[16 * j] = alpha * acc[i][j][r]; }
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = m0 + wm * (BM / WM) + i * 16 + (lane >> 4) + 4 * r;
                    const int gj = n0 + wn * (BN / WN) + j * 16 + (lane & 15);
                    if (gi < g.M && gj < g.N) C[(long)gi * g.sCm + (long)gj] = alpha * acc[i][j][r];
                }
    }
    G2_STAMP(6)
    if (DIAGG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G2_STAMP(7)
        if (tid == 0) for (int q = 0; q < 8; ++q) atomicAdd((unsigned long long *)&g_gemm2_diag[q], (unsigned long long)dacc[q]);
    }
}

// A22 (m x m, column-major, ld) -= P Q^T with P = buf[:, 0:128], Q = buf[:, 64:192] (ldb rows apart),
// only tiles with column block <= row block + 1.
int syr2k_lower_f64(int m, int batch, double *A22, long ld, long bsA, const double *buf, long ldb, long bsBuf, int seg, int nseg,
                    int part, hipStream_t st)
{
    GemmDesc g{};
    g.M = m; g.N = m; g.K = opts().fused_probe ? 256 : 128; g.batch = batch;   // probe: reads 128 columns past [V|Z|V] (its neighbours in the work area)
    g.A = buf + 64 * ldb; g.sAm = 1; g.sAk = ldb; g.bA = bsBuf;      // kernel-A(i'=c, k) = Q(c, k)
    g.B = buf; g.sBn = 1; g.sBk = ldb; g.bB = bsBuf;                  // kernel-B(k, j'=r) = P(r, k)
    g.C = A22; g.sCm = ld; g.sCn = 1; g.bC = bsA;                     // C'(c, r) = A22(r, c)
    g.alpha = -1.0; g.beta = 1.0;
    const int nb = (m + 127) / 128;
    // valid tiles per channel: row block bx holds column blocks ylo .. min(bx + 1, nb - 1)  (part 1: column block 0 only),
    // enumerated in 8 x 8 super-blocks (see the kernel)
    int T = 0;
    g.yoff = (part == 2) ? 1 : 0;
    const int nsx = (nb + 7) / 8;
    if (nsx * (nsx + 1) / 2 + nsx > 80) return BSP_ERR_UNSUPPORTED;   // m <= 11264
    g.nsb = 0;
    for (int sx = 0; sx < nsx; ++sx)
        for (int sy = 0; sy <= ((sx + 1 < nsx - 1) ? sx + 1 : nsx - 1); ++sy) {
            int c = 0;
            for (int bx = 8 * sx; bx < 8 * sx + 8 && bx < nb; ++bx) {
                const int yhi = (part == 1) ? 0 : ((bx + 1 < nb - 1) ? bx + 1 : nb - 1);
                const int lo = (g.yoff > 8 * sy) ? g.yoff : 8 * sy, hi = (yhi < 8 * sy + 7) ? yhi : 8 * sy + 7;
                if (hi >= lo) c += hi - lo + 1;
            }
            if (c == 0) continue;
            T += c;
            g.sbxy[g.nsb] = sx | (sy << 8); g.sbpre[g.nsb] = T;
            ++g.nsb;
        }
    if (T <= 0) return BSP_OK;
    if (nseg > 1) {                                                   // a slice of the enumeration
        const int t0 = (int)((long)T * seg / nseg), t1 = (int)((long)T * (seg + 1) / nseg);
        if (t1 <= t0) return BSP_OK;
        g.toff = t0; T = t1 - t0;
    }
    g.lower_only = T;                                                 // MODE 1 reads it as the tile count
    dim3 grid((unsigned)(8 * ((batch + 7) / 8) * T), 1, 1);
    KScope kt(KS_SYR2K, st);
    const int gd = opts().gemm_diag;
    if (gd && part == 2 && m >= 3900) {
        long long z[8] = {0};
        BSP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm2_diag), z, sizeof(z)));
        hipLaunchKernelGGL((gemm2_kernel<128, 128, 0, 0, 1, 1>), grid, dim3(256), 0, st, g);
        BSP_HIP(hipStreamSynchronize(st));
        BSP_HIP(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_gemm2_diag), sizeof(z)));
        static const char *nm[8] = {"issue C+AB0", "wait C+AB0", "LDS store+barrier", "k-tile MFMA", "wait next AB", "LDS store+barrier",
                                    "issue C stores", "drain stores"};
        long long tot = 0;
        for (int q = 0; q < 8; ++q) tot += z[q];
        fprintf(stderr, "syr2k diag (m=%d, %u workgroups, sum of wave-0 cycles %.1f M):", m, grid.x, tot / 1e6);
        for (int q = 0; q < 8; ++q) fprintf(stderr, " [%s %.1f%%]", nm[q], 100.0 * z[q] / (double)tot);
        fprintf(stderr, "\n");
    } else
    hipLaunchKernelGGL((gemm2_kernel<128, 128, 0, 0, 1>), grid, dim3(256), 0, st, g);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

// Y (m x 64, column-major, ldy) = A22 W, A22 valid on 64-blocks (I, J) with J <= I+1, W m x 64 (ldw).
int symm_lower_f64(int m, int batch, const double *A22, long ld, long bsA, const double *W, long ldw, long bsW,
                   double *Y, long ldy, long bsY, hipStream_t st)
{
    GemmDesc g{};
    g.M = 64; g.N = m; g.K = m; g.batch = batch;
    g.A = W; g.sAm = ldw; g.sAk = 1; g.bA = bsW;                      // kernel-A(i'=c, k) = W(k, c)
    g.B = A22; g.sBn = 1; g.sBk = ld; g.bB = bsA;                     // kernel-B(k, j'=i) = A22(i, k) (direct part)
    g.C = Y; g.sCm = ldy; g.sCn = 1; g.bC = bsY;                      // C'(c, i) = Y(i, c)
    g.alpha = 1.0; g.beta = 0.0; g.lower_only = 0;
    dim3 grid((unsigned)(8 * ((batch + 7) / 8) * ((m + 127) / 128)), 1, 1);
    KScope kt(KS_SYMM, st);
    hipLaunchKernelGGL((gemm2_kernel<64, 128, 1, 0, 2>), grid, dim3(256), 0, st, g);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
