// tsqr.hip -- panel factorisation of sy2sb on MANY workgroups: TSQR + Householder reconstruction (round 3; BSP_PANEL_QR=3,
// the default).  Replaces, per panel, panel_qr2_kernel + G = V^T V (split-K) + form_T_kernel + W = V T of sy2sb.hip by two
// launches; the arithmetic it stands for is the panel QR inside LAPACK DSYTRD's blocked reduction (reference call site
// matrices.f90:248 -> DSYGV -> DSYEV -> DSYTRD).
//
// Why: the one-workgroup-per-channel panel kernel streams the m x 64 panel through one CU sixteen times and runs ~1200
// barrier-separated phases; it takes ~1 ms whatever m is (kernel trace at 16 channels: 954 us average, 583 us at m = 448),
// which is 60 of the 92 ms of sy2sb at 16 channels per GPU (BASELINE configs[3] on 8 GPUs), and at 128 channels its 512-thread /
// 128 KB workgroups only start where BOTH GEMM workgroups of a CU have left.  Its row capacity (16 rows per thread) was also
// the n <= 8256 limit of the build.
//
// What (tools/proto_tsqr.py is the dense NumPy statement of the same steps, checked there against ill-conditioned and
// rank-deficient panels):
//   tsqr_tree_kernel  one workgroup (256 threads, block in REGISTERS: thread = 16 rows x 4 columns) per 256-row block of the
//                     panel: unblocked Householder QR (one LDS-only barrier per column, reductions inside DPP rows), R to
//                     scratch, explicit Q of the block in place (dorg2r order), Q to scratch.  The workgroup that delivers the
//                     LAST of the (up to) four R factors of a tree node (agent-scope release / counter / acquire: the blocks of
//                     a channel may sit on different XCDs) goes on as that node: QR of the stacked R factors, and so on to the
//                     root.  No workgroup ever waits for another one.
//                     The root then reconstructs the Householder form (Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik,
//                     IPDPS 2014): Q11 = top block of Q1 (product of 64 x 64 blocks along the leftmost path), modified LU
//                     Q11 - S' = L U with s_j = -sgn of the partially eliminated diagonal (|pivot| >= 1: no pivoting needed),
//                     T = -U S' L^-T, U^-1;  I - V T V^T with V = (Q1 - E S') U^-1 is orthogonal and maps P to [S' R; 0].
//   tsqr_apply_kernel one workgroup per 256-row block: B = product of the 64 x 64 blocks on the block's path through the tree,
//                     V = Q_block (B U^-1), W = V T on the matrix cores (two 256 x 64 x 64 products from the same A fragments),
//                     [V | . | V] and W written where sy2sb's products expect them, S' R on top of the panel, zeros below.
// Results do not depend on the batch size, on which workgroup runs a node, or on timing: bit-identical from run to run.
#include "common.h"

namespace bsp {
namespace {

constexpr int TB = 64;                 // panel width (= NB of sy2sb.hip)
constexpr int TR = 256;                // rows of a block
constexpr int XS = 65;                 // leading dimension of the 64 x 64 LDS matrices of the reconstruction (column-major)
constexpr int TSQR_MAXLEV = 8;

struct TsqrPlan {
    int nlev;                          // levels of the tree (1: a single block, which is leaf and root)
    int cnt[TSQR_MAXLEV];              // nodes per level (level 0 = the 256-row blocks)
    int base[TSQR_MAXLEV];             // index of a level's first node in the per-channel arrays
    int ntot;
};

struct TsqrArgs {
    int npad, r0, c0, m;               // panel = A[r0 .. npad-1, c0 .. c0+63], m = npad - r0 rows
    double *A; long bsA;               // dense matrices, column-major, ld = npad
    double *buf; long bsBuf;           // [V | Z | V], ld = npad, row 0 = row r0 of A (sy2sb.hip)
    double *W; long bsW;               // W = V T, ld = npad
    double *scr; long bsScr;           // scratch per channel: Q blocks | R blocks | U^-1 | T | L | S'
    int *cntr; long bsCntr;            // arrival counters per node (zero between launches)
    long offQ, offR, offUi, offT, offL, offS;
    TsqrPlan plan;
};

template <int CTRL>
__device__ __forceinline__ double tq_dpp(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, true);
    return r.d;
}
// Sum over the 16 lanes of a DPP row, the total in every lane (bitwise the same in all of them: each step adds the two halves of
// a pair): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror.
__device__ __forceinline__ double tq_rsum16(double x)
{
    x += tq_dpp<0xB1>(x);
    x += tq_dpp<0x4E>(x);
    x += tq_dpp<0x141>(x);
    x += tq_dpp<0x140>(x);
    return x;
}
__device__ __forceinline__ void tq_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the 256 x 64 block in registers: thread (rg = lane & 15, cg = tid >> 4) holds a[i][jj] = element (rg + 16 i, cg + 16 jj).
// Column j = 16 JJ + j16 lives in register column JJ of the 16 threads with cg == j16 (one DPP row); rows < 16 JJ are finished.

// Householder QR, columns 16 JJ .. 16 JJ + 15: afterwards R on and above the diagonal, the reflectors (unit diagonal implied)
// below it, tau in taus[].  Nothing (numerically) below a pivot: H = I, tiny entries are dropped (as sy2sb.hip's panel kernels do).
template <int JJ>
__device__ __forceinline__ void tq_qr_cols(double (&a)[16][4], double *vb, double *taus, int rg, int cg)
{
#pragma unroll 1
    for (int j16 = 0; j16 < 16; ++j16) {
        const int j = 16 * JJ + j16;
        double *vp = vb + (j & 1) * TR;
        if (cg == j16) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = JJ + 1; i < 16; ++i) {
                if (i & 1) s1 += a[i][JJ] * a[i][JJ]; else s0 += a[i][JJ] * a[i][JJ];
            }
            const double x0 = a[JJ][JJ];
            const double sigma = tq_rsum16((s0 + s1) + (rg > j16 ? x0 * x0 : 0.0));
            const double alpha = tq_rsum16(rg == j16 ? x0 : 0.0);
            double beta = alpha, tau = 0.0, scale = 0.0;
            if (sigma != 0.0 && (alpha * alpha + sigma > 1e-280)) {
                beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
#pragma unroll
            for (int i = JJ + 1; i < 16; ++i) {
                a[i][JJ] *= scale;
                vp[rg + 16 * i] = a[i][JJ];
            }
            const double xs = x0 * scale;
            vp[rg + 16 * JJ] = rg > j16 ? xs : (rg == j16 ? 1.0 : 0.0);
            a[JJ][JJ] = rg > j16 ? xs : (rg == j16 ? beta : x0);
            if (rg == 0) taus[j] = tau;
        }
        tq_bar();
        const double tau = taus[j];
        double v[16];
#pragma unroll
        for (int i = JJ; i < 16; ++i) v[i] = vp[rg + 16 * i];
#pragma unroll
        for (int jj = JJ; jj < 4; ++jj) {
            if (jj > JJ || cg > j16) {                 // columns right of j (a DPP row is all in or all out)
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = JJ; i < 16; ++i) {
                    if (i & 1) s1 += v[i] * a[i][jj]; else s0 += v[i] * a[i][jj];
                }
                const double w = tau * tq_rsum16(s0 + s1);
#pragma unroll
                for (int i = JJ; i < 16; ++i) a[i][jj] -= w * v[i];
            }
        }
    }
}

// Explicit Q = H_0 H_1 .. H_63 [I; 0] in place of the reflectors (LAPACK dorg2r: last reflector first; column j becomes H_j e_j
// after H_j has been applied to the columns right of it).
template <int JJ>
__device__ __forceinline__ void tq_formq_cols(double (&a)[16][4], double *vb, const double *taus, int rg, int cg)
{
#pragma unroll 1
    for (int j16 = 15; j16 >= 0; --j16) {
        const int j = 16 * JJ + j16;
        double *vp = vb + (j & 1) * TR;
        if (cg == j16) {
#pragma unroll
            for (int i = JJ + 1; i < 16; ++i) vp[rg + 16 * i] = a[i][JJ];
            vp[rg + 16 * JJ] = rg > j16 ? a[JJ][JJ] : (rg == j16 ? 1.0 : 0.0);
        }
        tq_bar();
        const double tau = taus[j];
        double v[16];
#pragma unroll
        for (int i = JJ; i < 16; ++i) v[i] = vp[rg + 16 * i];
#pragma unroll
        for (int jj = JJ; jj < 4; ++jj) {
            if (jj > JJ || cg > j16) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = JJ; i < 16; ++i) {
                    if (i & 1) s1 += v[i] * a[i][jj]; else s0 += v[i] * a[i][jj];
                }
                const double w = tau * tq_rsum16(s0 + s1);
#pragma unroll
                for (int i = JJ; i < 16; ++i) a[i][jj] -= w * v[i];
            }
        }
        if (cg == j16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i > JJ) a[i][JJ] = -tau * v[i];
                else if (i < JJ) a[i][JJ] = 0.0;
            }
            a[JJ][JJ] = rg > j16 ? -tau * v[JJ] : (rg == j16 ? 1.0 - tau : 0.0);
        }
    }
}

constexpr int TQ_LDS_TREE = (2 * TR + TB + 8 + 2 * TB * XS) * 8;       // vb | taus | flag | X | Y  = 69 184 bytes

__global__ __launch_bounds__(256) void tsqr_tree_kernel(TsqrArgs g)
{
    extern __shared__ __attribute__((aligned(16))) double tq_lds[];
    double *vb = tq_lds, *taus = vb + 2 * TR;
    int *flag = reinterpret_cast<int *>(taus + TB);
    double *X = taus + TB + 8, *Y = X + TB * XS;
    const int tid = threadIdx.x, lane = tid & 63, rg = lane & 15, cg = tid >> 4;
    const int ch = blockIdx.y;
    const TsqrPlan &pl = g.plan;
    double *scr = g.scr + (long)ch * g.bsScr;
    double *Qall = scr + g.offQ, *Rall = scr + g.offR;
    int *cntr = g.cntr + (long)ch * g.bsCntr;
    const long ld = g.npad;
    const double *P = g.A + (long)ch * g.bsA + (long)g.c0 * ld + g.r0;

    int level = 0, node = blockIdx.x;
    double a[16][4];
    {   // the 256 rows of this leaf (rows beyond m: zeros); 16 lanes read 16 consecutive rows of a column
        const int row0 = TR * node;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = row0 + rg + 16 * i;
                a[i][jj] = P[(long)(cg + 16 * jj) * ld + (r < g.m ? r : 0)];
            }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i][jj] = (row0 + rg + 16 * i < g.m) ? a[i][jj] : 0.0;
    }
    for (;;) {
        tq_qr_cols<0>(a, vb, taus, rg, cg);
        tq_qr_cols<1>(a, vb, taus, rg, cg);
        tq_qr_cols<2>(a, vb, taus, rg, cg);
        tq_qr_cols<3>(a, vb, taus, rg, cg);
        double *Rn = Rall + (long)(pl.base[level] + node) * (TB * TB);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rg + 16 * i, c = cg + 16 * jj;
                Rn[r + TB * c] = r <= c ? a[i][jj] : 0.0;
            }
        tq_formq_cols<3>(a, vb, taus, rg, cg);
        tq_formq_cols<2>(a, vb, taus, rg, cg);
        tq_formq_cols<1>(a, vb, taus, rg, cg);
        tq_formq_cols<0>(a, vb, taus, rg, cg);
        double *Qn = Qall + (long)(pl.base[level] + node) * (TR * TB);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 16; ++i) Qn[(rg + 16 * i) + TR * (cg + 16 * jj)] = a[i][jj];
        if (level == pl.nlev - 1) break;
        // ---- hand the R factor to the parent node: the workgroup whose arrival completes the node becomes it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int parent = node >> 2;
        if (tid == 0) {
            int nchild = pl.cnt[level] - 4 * parent;
            nchild = nchild > 4 ? 4 : nchild;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int *cp = cntr + pl.base[level + 1] + parent;
            const int old = __hip_atomic_fetch_add(cp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == nchild - 1;
            if (last) {
                __hip_atomic_store(cp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next panel
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        __syncthreads();
        {   // rows 64 k .. 64 k + 63 of the node = R factor of child 4 parent + k (a missing child: zeros)
            const int nchild_all = pl.cnt[level];
            const double *Rc = Rall + (long)(pl.base[level] + 4 * parent) * (TB * TB);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int k = i >> 2, r = rg + 16 * (i & 3), c = cg + 16 * jj;
                    const bool have = 4 * parent + k < nchild_all;
                    const double x = Rc[have ? (long)k * (TB * TB) + r + TB * c : 0];
                    a[i][jj] = have ? x : 0.0;
                }
        }
        ++level;
        node = parent;
    }

    // ================= root: Householder reconstruction =================================================================
    // Q11 = top 64 x 64 block of Q1 = product, along the leftmost path, of the top blocks of the nodes' explicit Q factors
    // (leaf first).  This workgroup's own factor (the root's) comes from its registers.
    const int top = pl.nlev - 1;
    double *mine = top > 0 ? Y : X;
    tq_bar();                                        // the last v buffer of the Q formation has been read by everyone
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i) mine[(rg + 16 * i) + XS * (cg + 16 * jj)] = a[i][jj];
    if (top > 0) {
        {   // X = top block of leaf 0's Q
            const double *Q0 = Qall + (long)pl.base[0] * (TR * TB);
            for (int idx = tid; idx < TB * TB; idx += 256) X[(idx & 63) + XS * (idx >> 6)] = Q0[(idx & 63) + TR * (idx >> 6)];
        }
        tq_bar();
        for (int l = 1; l <= top; ++l) {             // X <- X . (top block of the level-l node 0's Q); the root's is in Y
            const double *Bp = l < top ? Qall + (long)pl.base[l] * (TR * TB) : Y;
            const long sBn = l < top ? TR : XS;
            const int r4 = 4 * (tid & 15), c4 = 4 * (tid >> 4);
            double acc[4][4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
#pragma unroll 4
            for (int k = 0; k < 64; ++k) {
                double av[4], bv[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) av[x] = X[(r4 + x) + XS * k];
#pragma unroll
                for (int y = 0; y < 4; ++y) bv[y] = Bp[k + sBn * (c4 + y)];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[x][y] += av[x] * bv[y];
            }
            tq_bar();                                // every thread has read X
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) X[(r4 + x) + XS * (c4 + y)] = acc[x][y];
            tq_bar();
        }
    } else tq_bar();

    // ---- modified LU of X = Q11 - S' (S' chosen on the fly), in place: L strictly below the diagonal, U on and above it
    double *sg = vb;                                // the v buffers are free now
#pragma unroll 1
    for (int j = 0; j < TB; ++j) {
        const double d = X[j + XS * j];
        const double s = d >= 0.0 ? -1.0 : 1.0;
        const double piv = d - s;
        if (tid > j && tid < TB) X[tid + XS * j] = X[tid + XS * j] / piv;
        tq_bar();
        if (tid == 0) { X[j + XS * j] = piv; sg[j] = s; }
        // trailing update, thread t: rows r = j + 1 + (t & 15) + 16 a, columns c = j + 1 + (t >> 4) + 16 b
        for (int c = j + 1 + (tid >> 4); c < TB; c += 16) {
            const double ujc = X[j + XS * c];
            for (int r = j + 1 + (tid & 15); r < TB; r += 16) X[r + XS * c] -= X[r + XS * j] * ujc;
        }
        tq_bar();
    }
    // ---- Y = -U S' (upper triangle), then T = Y L^-T in place, row by row (thread i owns row i: T(i,c) -= sum_{k<c} T(i,k) L(c,k))
    for (int idx = tid; idx < TB * TB; idx += 256) {
        const int r = idx & 63, c = idx >> 6;
        Y[r + XS * c] = r <= c ? -X[r + XS * c] * sg[c] : 0.0;
    }
    tq_bar();
    double *Ui = scr + g.offUi, *Tm = scr + g.offT, *Lm = scr + g.offL, *Sm = scr + g.offS;
    if (tid >= 64 && tid < 128) {                   // wave 1: T
        const int i = tid - 64;
        for (int c = i + 1; c < TB; ++c) {
            double s = Y[i + XS * c];
            for (int k = i; k < c; ++k) s -= Y[i + XS * k] * X[c + XS * k];
            Y[i + XS * c] = s;
        }
    }
    // L (unit lower) and S' to scratch while wave 1 works: X's lower triangle is not written any more
    for (int idx = tid; idx < TB * TB; idx += 256) {
        const int r = idx & 63, c = idx >> 6;
        Lm[idx] = r > c ? X[r + XS * c] : (r == c ? 1.0 : 0.0);          // column-major, ld 64
    }
    if (tid < TB) Sm[tid] = sg[tid];
    tq_bar();
    __syncthreads();
    // ---- U^-1 in place in X's upper triangle (LAPACK dtrti2, column by column; wave 0, wave-synchronous), T to scratch meanwhile
    if (tid < 64) {
        const int r = tid;
        for (int j = 0; j < TB; ++j) {
            const double ujj = 1.0 / X[j + XS * j];
            // column j of the inverse above the diagonal: -Uinv(0:j,0:j) U(0:j,j) ujj; thread r < j computes row r
            double s = 0.0;
            if (r < j)
                for (int k = r; k < j; ++k) s += X[r + XS * k] * X[k + XS * j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // all reads of column j done (one wave: lock-step)
            __builtin_amdgcn_wave_barrier();
            if (r < j) X[r + XS * j] = -s * ujj;
            if (r == j) X[j + XS * j] = ujj;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (int idx = tid - 64; idx < TB * TB; idx += 192) {
            const int r = idx >> 6, c = idx & 63;
            Tm[idx] = r <= c ? Y[r + XS * c] : 0.0;                       // row-major, ld 64: Tm[r * 64 + c]
        }
    }
    tq_bar();
    __syncthreads();
    for (int idx = tid; idx < TB * TB; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        Ui[idx] = r <= c ? X[r + XS * c] : 0.0;                           // row-major, ld 64
    }
    // ---- top 64 rows of V (= L: exact unit lower triangle) and of W = L T
    double *buf = g.buf + (long)ch * g.bsBuf, *Wg = g.W + (long)ch * g.bsW;
    {
        const int r4 = 4 * (tid & 15), c4 = 4 * (tid >> 4);
        double acc[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
        for (int k = 0; k < r4 + 4; ++k) {                                 // L(r, k) = 0 for k > r
            double lv[4], tv[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) lv[x] = k < r4 + x ? X[(r4 + x) + XS * k] : (k == r4 + x ? 1.0 : 0.0);
#pragma unroll
            for (int y = 0; y < 4; ++y) tv[y] = k <= c4 + y ? Y[k + XS * (c4 + y)] : 0.0;
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] += lv[x] * tv[y];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int r = r4 + x, c = c4 + y;
                Wg[r + ld * c] = acc[x][y];
                const double lv = r > c ? X[r + XS * c] : (r == c ? 1.0 : 0.0);
                buf[r + ld * c] = lv;
                buf[r + ld * (2 * TB + c)] = lv;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// V = Q_block (B U^-1), W = V T for one 256-row block; S' R on top of the panel, zeros below it.
constexpr int BS = 66;                                                    // leading dimension of the row-major LDS matrices here
constexpr int TQ_LDS_APPLY = 2 * TB * BS * 8;                             // 67 584 bytes

// C = A B, 64 x 64, A and C row-major in LDS (ld BS), B from global or LDS through its strides
__device__ __forceinline__ void tq_mm64r(const double *A, const double *B, long sBk, long sBn, double *C, int tid)
{
    const int r4 = 4 * (tid >> 4), c4 = 4 * (tid & 15);
    double acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) av[x] = A[(r4 + x) * BS + k];
#pragma unroll
        for (int y = 0; y < 4; ++y) bv[y] = B[k * sBk + (c4 + y) * sBn];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] += av[x] * bv[y];
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) C[(r4 + x) * BS + (c4 + y)] = acc[x][y];
}

__global__ __launch_bounds__(256) void tsqr_apply_kernel(TsqrArgs g)
{
    extern __shared__ __attribute__((aligned(16))) double tq_lds[];
    double *M0 = tq_lds, *M1 = M0 + TB * BS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = blockIdx.y, leaf = blockIdx.x;
    const TsqrPlan &pl = g.plan;
    const double *scr = g.scr + (long)ch * g.bsScr;
    const double *Qall = scr + g.offQ;
    const double *Ui = scr + g.offUi, *Tm = scr + g.offT;
    const long ld = g.npad;
    // ---- B = product of the 64 x 64 blocks on the path leaf -> root (row block k of the parent's Q, k = position among its
    // children); B U^-1 -> BV, BV T -> BW (row-major, the layout the MFMA B fragments are read in)
    double *BV, *BW;
    if (pl.nlev == 1) {
        for (int idx = tid; idx < TB * TB; idx += 256) M0[(idx >> 6) * BS + (idx & 63)] = Ui[idx];
        tq_bar();
        __syncthreads();
        BV = M0; BW = M1;
    } else {
        int idx = leaf;
        double *cur = M0, *oth = M1;
        {
            const int k = idx & 3; idx >>= 2;
            const double *Qp = Qall + (long)(pl.base[1] + idx) * (TR * TB) + TB * k;
            for (int e = tid; e < TB * TB; e += 256) cur[(e & 63) * BS + (e >> 6)] = Qp[(e & 63) + TR * (e >> 6)];
        }
        tq_bar();
        __syncthreads();
        for (int l = 2; l < pl.nlev; ++l) {
            const int k = idx & 3; idx >>= 2;
            const double *Qp = Qall + (long)(pl.base[l] + idx) * (TR * TB) + TB * k;
            tq_mm64r(cur, Qp, 1, TR, oth, tid);
            tq_bar();
            __syncthreads();
            double *t = cur; cur = oth; oth = t;
        }
        tq_mm64r(cur, Ui, TB, 1, oth, tid);
        tq_bar();
        __syncthreads();
        BV = oth; BW = cur;
    }
    tq_mm64r(BV, Tm, TB, 1, BW, tid);
    // meanwhile-independent: the panel itself becomes [S' R; 0]
    {
        double *P = g.A + (long)ch * g.bsA + (long)g.c0 * ld + g.r0;
        const double *Rr = scr + g.offR + (long)pl.base[pl.nlev - 1] * (TB * TB);
        const double *Sm = scr + g.offS;
        const int row0 = TR * leaf;
        for (int e = tid; e < TR * TB; e += 256) {
            const int r = row0 + (e & 255), c = e >> 8;
            if (r < g.m) {
                double v = 0.0;
                if (r < TB && r <= c) v = Sm[r] * Rr[r + TB * c];
                P[r + ld * c] = v;
            }
        }
    }
    tq_bar();
    __syncthreads();
    // ---- V = Q_leaf BV, W = Q_leaf BW on the matrix cores: wave w owns rows 64 w .. 64 w + 63 of the block, in two halves of 32;
    // the products are formed transposed (MFMA(b, a)), so that a DPP row of 16 lanes stores 16 consecutive rows of one column
    // (gemm_f64.hip::tsmm64_kernel)
    const double *Ql = Qall + (long)(pl.base[0] + leaf) * (TR * TB);
    double *buf = g.buf + (long)ch * g.bsBuf, *Wg = g.W + (long)ch * g.bsW;
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int rb = 64 * wave + 32 * half;                              // block-relative first row
        double af[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) af[i][q] = Ql[(rb + 16 * i + l15) + (long)TR * (4 * q + l4)];
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            const double *Bm = which ? BW : BV;
            double4_t acc[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                double b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bm[(4 * q + l4) * BS + 16 * j + l15];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], af[i][q], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = TR * leaf + rb + 16 * i + l15, col = 16 * j + l4 + 4 * r;
                        if (row < g.m && row >= (leaf == 0 ? TB : 0)) {      // the top 64 rows were written by the root (exact L, L T)
                            if (which) Wg[row + ld * col] = acc[i][j][r];
                            else {
                                buf[row + ld * col] = acc[i][j][r];
                                buf[row + ld * (2 * TB + col)] = acc[i][j][r];
                            }
                        }
                    }
        }
    }
}

TsqrPlan make_plan(int m)
{
    TsqrPlan p{};
    int n = (m + TR - 1) / TR, b = 0, l = 0;
    for (;;) {
        p.cnt[l] = n; p.base[l] = b; b += n; ++l;
        if (n == 1 || l == TSQR_MAXLEV) break;
        n = (n + 3) / 4;
    }
    p.nlev = l; p.ntot = b;
    return p;
}

}  // namespace

// scratch of the TSQR panel factorisation per channel of order npad: doubles (Q and R factors of every tree node, U^-1, T, L, S')
// and ints (arrival counters of the nodes)
long tsqr_scr_doubles(int npad)
{
    const TsqrPlan p = make_plan(npad);
    return (long)p.ntot * (TR * TB + TB * TB) + 3L * TB * TB + TB + 64;
}
long tsqr_cntr_ints(int npad) { return make_plan(npad).ntot + 16; }

// Panel factorisation + W = V T for the panel whose rows start at r0 (columns at c0): on return (stream order) the panel holds
// [S' R; 0], buf holds V in its first and third 64-column slot, W = V T.  scr: batch * tsqr_scr_doubles(npad) doubles; cntr: batch *
// tsqr_cntr_ints(npad) ints, zero before the first use (the arrival counters return to zero by themselves).
int tsqr_panel(int npad, int r0, int c0, int batch, double *d_A, double *buf, double *W, double *scr, int *cntr, hipStream_t st)
{
    const int m = npad - r0;
    if (m <= 0 || batch <= 0) return BSP_OK;
    if (m % TB) return BSP_ERR_ARG;
    TsqrArgs g{};
    g.npad = npad; g.r0 = r0; g.c0 = c0; g.m = m;
    g.A = d_A; g.bsA = (long)npad * npad;
    g.buf = buf; g.bsBuf = (long)npad * 3 * TB;
    g.W = W; g.bsW = (long)npad * TB;
    const TsqrPlan pmax = make_plan(npad);
    g.plan = make_plan(m);
    if (g.plan.cnt[g.plan.nlev - 1] != 1) return BSP_ERR_UNSUPPORTED;        // more than 4^7 blocks
    g.bsScr = tsqr_scr_doubles(npad);
    g.scr = scr;
    g.offQ = 0;
    g.offR = (long)pmax.ntot * (TR * TB);
    g.offUi = g.offR + (long)pmax.ntot * (TB * TB);
    g.offT = g.offUi + TB * TB; g.offL = g.offT + TB * TB; g.offS = g.offL + TB * TB;
    g.cntr = cntr;
    g.bsCntr = tsqr_cntr_ints(npad);
    static bool attr = false;
    if (!attr) {
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tsqr_tree_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TQ_LDS_TREE));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tsqr_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TQ_LDS_APPLY));
        attr = true;
    }
    const dim3 grid(g.plan.cnt[0], batch);
    {
        KScope kt(KS_PANEL_QR, st);
        hipLaunchKernelGGL(tsqr_tree_kernel, grid, dim3(256), TQ_LDS_TREE, st, g);
    }
    {
        KScope kt(KS_CHAIN, st);
        hipLaunchKernelGGL(tsqr_apply_kernel, grid, dim3(256), TQ_LDS_APPLY, st, g);
    }
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
