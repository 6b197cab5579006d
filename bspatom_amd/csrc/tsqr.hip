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
    long offQ, offR, offUi, offT, offS;
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

// Update of register column block jj with the reflector (v, tau): a(:, jj) -= tau (v^T a(:, jj)) v for the rows of blocks >= I0,
// wherever `on` (a whole DPP row is on or off: its 16 lanes hold one column).  Branch-free: a row that is off subtracts zero.
#define TQ_UPD(I0, JJX, ON)                                                                 \
    {                                                                                       \
        double s0_ = 0.0, s1_ = 0.0;                                                        \
        _Pragma("unroll") for (int i = (I0); i < 16; ++i) {                                 \
            if (i & 1) s1_ += v[i] * a[i][JJX]; else s0_ += v[i] * a[i][JJX];               \
        }                                                                                   \
        const double w_ = (ON) ? tau * tq_rsum16(s0_ + s1_) : 0.0;                          \
        _Pragma("unroll") for (int i = (I0); i < 16; ++i) a[i][JJX] -= w_ * v[i];           \
    }

// Householder vector of column jn = 16 JN + jn16 (held by the 16 lanes with cg == jn16 in register column JN).  The arithmetic runs on
// ALL lanes, each on its own column (the other rows' results are discarded): no branch around the long dependent chain (norm,
// sqrt, two divisions), so that the scheduler fills it with the independent column updates that follow in the same block.
// Nothing (numerically) below the pivot: H = I, tiny entries are dropped (as sy2sb.hip's panel kernels do).
template <int JN>
__device__ __forceinline__ void tq_house(double (&a)[16][4], int jn16, double *vn, double *taus, int rg, int cg)
{
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = JN + 1; i < 16; ++i) {
        if (i & 1) s1 += a[i][JN] * a[i][JN]; else s0 += a[i][JN] * a[i][JN];
    }
    const double x0 = a[JN][JN];
    const double sigma = tq_rsum16((s0 + s1) + (rg > jn16 ? x0 * x0 : 0.0));
    const double alpha = tq_rsum16(rg == jn16 ? x0 : 0.0);
    const double n2 = alpha * alpha + sigma;
    const bool ok = sigma != 0.0 && n2 > 1e-280;
    const double nrm = sqrt(ok ? n2 : 1.0);
    const double beta = ok ? -copysign(nrm, alpha) : alpha;
    double tau = ok ? (beta - alpha) / (ok ? beta : 1.0) : 0.0;
    double scale = ok ? 1.0 / (ok ? alpha - beta : 1.0) : 0.0;
    asm volatile("" : "+v"(tau), "+v"(scale));           // computed here, not inside the branch below
    if (cg == jn16) {
#pragma unroll
        for (int i = JN + 1; i < 16; ++i) {
            a[i][JN] *= scale;
            vn[rg + 16 * i] = a[i][JN];
        }
        const double xs = x0 * scale;
        vn[rg + 16 * JN] = rg > jn16 ? xs : (rg == jn16 ? 1.0 : 0.0);
        a[JN][JN] = rg > jn16 ? xs : (rg == jn16 ? beta : x0);
        if (rg == 0) taus[16 * JN + jn16] = tau;
    }
}

// One column step j = 16 JJ + j16 of the QR: apply H_j (its v, tau are in LDS since the previous step) -- first to the register
// column block JN that holds column j + 1, then form v_{j+1} there (look-ahead: the only thing the next step waits for), then
// the other blocks.  One LDS-only barrier per column.
template <int JJ, int JN, bool NEXT>
__device__ __forceinline__ void tq_qr_step(double (&a)[16][4], double *vb, double *taus, int j16, int rg, int cg)
{
    const int j = 16 * JJ + j16;
    const double *vp = vb + (j & 1) * TR;
    double *vn = vb + ((j + 1) & 1) * TR;
    const double tau = taus[j];
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = i >= JJ ? vp[rg + 16 * i] : 0.0;
    const bool onJ = cg > j16;                            // block JJ: the columns right of j
    if constexpr (JN == JJ) TQ_UPD(JJ, JJ, onJ)
    else if constexpr (JN < 4) TQ_UPD(JJ, JN, true)
    if constexpr (NEXT) tq_house<(JN < 4 ? JN : 3)>(a, (j16 + 1) & 15, vn, taus, rg, cg);
#pragma unroll
    for (int jj = JJ; jj < 4; ++jj) {
        if (jj == JN) continue;
        if (jj == JJ) TQ_UPD(JJ, JJ, onJ)
        else TQ_UPD(JJ, jj, true)
    }
    tq_bar();
}

template <int JJ>
__device__ __forceinline__ void tq_qr_cols(double (&a)[16][4], double *vb, double *taus, int rg, int cg)
{
#pragma unroll 1
    for (int j16 = 0; j16 < 15; ++j16) tq_qr_step<JJ, JJ, true>(a, vb, taus, j16, rg, cg);
    tq_qr_step<JJ, JJ + 1, (JJ < 3)>(a, vb, taus, 15, rg, cg);
}

// Householder QR of the block: afterwards R on and above the diagonal, the reflectors (unit diagonal implied) below it, tau in taus[].
__device__ __forceinline__ void tq_qr(double (&a)[16][4], double *vb, double *taus, int rg, int cg)
{
    tq_house<0>(a, 0, vb, taus, rg, cg);
    tq_bar();
    tq_qr_cols<0>(a, vb, taus, rg, cg);
    tq_qr_cols<1>(a, vb, taus, rg, cg);
    tq_qr_cols<2>(a, vb, taus, rg, cg);
    tq_qr_cols<3>(a, vb, taus, rg, cg);
}

// Explicit Q = H_0 H_1 .. H_63 [I; 0] in place of the reflectors (LAPACK dorg2r: last reflector first; column j becomes H_j e_j
// after H_j has been applied to the columns right of it).  v_{j-1} is published while step j runs (column j - 1 is not touched by
// H_j), so a step is: read v_j, update, one barrier.
template <int JP>
__device__ __forceinline__ void tq_publish(const double (&a)[16][4], int jp16, double *vn, int rg, int cg)
{
    if (cg == jp16) {
#pragma unroll
        for (int i = JP + 1; i < 16; ++i) vn[rg + 16 * i] = a[i][JP];
        vn[rg + 16 * JP] = rg > jp16 ? a[JP][JP] : (rg == jp16 ? 1.0 : 0.0);
    }
}

template <int JJ>
__device__ __forceinline__ void tq_formq_cols(double (&a)[16][4], double *vb, const double *taus, int rg, int cg)
{
#pragma unroll 1
    for (int j16 = 15; j16 >= 0; --j16) {
        const int j = 16 * JJ + j16;
        const double *vp = vb + (j & 1) * TR;
        double *vn = vb + ((j + 1) & 1) * TR;             // (j - 1) & 1
        if (j16 > 0) tq_publish<JJ>(a, j16 - 1, vn, rg, cg);
        else if (JJ > 0) tq_publish<(JJ > 0 ? JJ - 1 : 0)>(a, 15, vn, rg, cg);
        const double tau = taus[j];
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = i >= JJ ? vp[rg + 16 * i] : 0.0;
        const bool onJ = cg > j16;
#pragma unroll
        for (int jj = JJ; jj < 4; ++jj) {
            if (jj == JJ) TQ_UPD(JJ, JJ, onJ)
            else TQ_UPD(JJ, jj, true)
        }
        if (cg == j16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i > JJ) a[i][JJ] = -tau * v[i];
                else if (i < JJ) a[i][JJ] = 0.0;
            }
            a[JJ][JJ] = rg > j16 ? -tau * v[JJ] : (rg == j16 ? 1.0 - tau : 0.0);
        }
        tq_bar();
    }
}

__device__ __forceinline__ void tq_formq(double (&a)[16][4], double *vb, const double *taus, int rg, int cg)
{
    tq_publish<3>(a, 15, vb + TR, rg, cg);                // v_63 -> buffer 63 & 1 (the QR's last read of it is behind a barrier)
    tq_bar();
    tq_formq_cols<3>(a, vb, taus, rg, cg);
    tq_formq_cols<2>(a, vb, taus, rg, cg);
    tq_formq_cols<1>(a, vb, taus, rg, cg);
    tq_formq_cols<0>(a, vb, taus, rg, cg);
}

constexpr int TQ_LDS_TREE = (2 * TR + TB + 8 + 2 * TB * XS) * 8;       // vb | taus | flag | X | Y  = 69 184 bytes

template <int MINW>
__global__ __launch_bounds__(256, MINW) void tsqr_tree_kernel(TsqrArgs g)
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
        tq_qr(a, vb, taus, rg, cg);
        double *Rn = Rall + (long)(pl.base[level] + node) * (TB * TB);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rg + 16 * i, c = cg + 16 * jj;
                Rn[r + TB * c] = r <= c ? a[i][jj] : 0.0;
            }
        tq_formq(a, vb, taus, rg, cg);
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
            double tmp[16];                               // all 16 loads of a thread in flight, then the LDS writes
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int idx = tid + 256 * u; tmp[u] = Q0[(idx & 63) + TR * (idx >> 6)]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int idx = tid + 256 * u; X[(idx & 63) + XS * (idx >> 6)] = tmp[u]; }
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

    // ---- modified LU of X = Q11 - S' (S' chosen on the fly: s_j = -sgn of the partially eliminated diagonal, |pivot| >= 1), then
    // T = -U S' L^-T and U^-1, all in REGISTERS: thread (br = tid & 15, bc = tid >> 4) owns the 4 x 4 block (rows 4 br.., columns
    // 4 bc..) of each 64 x 64 matrix; per step the pivot row / column travel through LDS (double-buffered: one barrier per step).
    // L is formed by multiplication with the reciprocal pivot, the same product wherever it is used.
    const int br = tid & 15, bc = tid >> 4;
    double x[4][4], t[4][4], ui[4][4];
#pragma unroll
    for (int ro = 0; ro < 4; ++ro)
#pragma unroll
        for (int co = 0; co < 4; ++co) x[ro][co] = X[(4 * br + ro) + XS * (4 * bc + co)];
    double *sg = taus;                               // the tau's are not needed any more
    double *Urow = vb, *Lcol = vb + 128;             // [2][64] each
#pragma unroll 1
    for (int jb = 0; jb < 16; ++jb) {
#pragma unroll
        for (int jo = 0; jo < 4; ++jo) {
            const int j = 4 * jb + jo, p = 64 * (jo & 1);
            if (br == jb) {
#pragma unroll
                for (int co = 0; co < 4; ++co) Urow[p + 4 * bc + co] = x[jo][co];
            }
            if (bc == jb) {
#pragma unroll
                for (int ro = 0; ro < 4; ++ro) Lcol[p + 4 * br + ro] = x[ro][jo];
            }
            tq_bar();
            const double d = Urow[p + j];
            const double sj = d >= 0.0 ? -1.0 : 1.0;
            const double piv = d - sj, rp = 1.0 / piv;
            double lv[4], uv[4];
#pragma unroll
            for (int ro = 0; ro < 4; ++ro) lv[ro] = (4 * br + ro > j) ? Lcol[p + 4 * br + ro] * rp : 0.0;
#pragma unroll
            for (int co = 0; co < 4; ++co) uv[co] = (4 * bc + co > j) ? Urow[p + 4 * bc + co] : 0.0;
#pragma unroll
            for (int ro = 0; ro < 4; ++ro)
#pragma unroll
                for (int co = 0; co < 4; ++co) x[ro][co] -= lv[ro] * uv[co];
            if (bc == jb) {
#pragma unroll
                for (int ro = 0; ro < 4; ++ro)
                    if (4 * br + ro > j) x[ro][jo] = lv[ro];
                if (br == jb) x[jo][jo] = piv;
            }
            if (tid == 0) sg[j] = sj;
        }
    }
    tq_bar();
    // t = Y = -U S' (upper triangle), ui = I;  then, in one loop of 64 steps: column c = it of T is final and leaves the columns
    // right of it (T L^T = Y), row k = 63 - it of U^-1 is final and leaves the rows above it (U U^-1 = I)
#pragma unroll
    for (int ro = 0; ro < 4; ++ro)
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            const int r = 4 * br + ro, c = 4 * bc + co;
            t[ro][co] = r <= c ? -x[ro][co] * sg[c] : 0.0;
            ui[ro][co] = r == c ? 1.0 : 0.0;
        }
    double *Tcol = vb, *LcT = vb + 128, *Xrow = vb + 256, *Ucol = vb + 384;      // [2][64] each
#pragma unroll 1
    for (int cb = 0; cb < 16; ++cb) {
        const int kb = 15 - cb;
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            const int kj = 3 - cj;
            const int c = 4 * cb + cj, k = 4 * kb + kj, p = 64 * (cj & 1);
            if (bc == cb) {
#pragma unroll
                for (int ro = 0; ro < 4; ++ro) {
                    Tcol[p + 4 * br + ro] = t[ro][cj];
                    LcT[p + 4 * br + ro] = (4 * br + ro > c) ? x[ro][cj] : 0.0;          // L(r, c), r > c
                }
            }
            if (br == kb) {
#pragma unroll
                for (int co = 0; co < 4; ++co) Xrow[p + 4 * bc + co] = ui[kj][co];
            }
            if (bc == kb) {
#pragma unroll
                for (int ro = 0; ro < 4; ++ro) Ucol[p + 4 * br + ro] = (4 * br + ro <= k) ? x[ro][kj] : 0.0;   // U(r, k), r <= k
            }
            tq_bar();
            {
                double tc[4], lc[4];
#pragma unroll
                for (int ro = 0; ro < 4; ++ro) tc[ro] = Tcol[p + 4 * br + ro];
#pragma unroll
                for (int co = 0; co < 4; ++co) lc[co] = (4 * bc + co > c) ? LcT[p + 4 * bc + co] : 0.0;
#pragma unroll
                for (int ro = 0; ro < 4; ++ro)
#pragma unroll
                    for (int co = 0; co < 4; ++co) t[ro][co] -= tc[ro] * lc[co];
            }
            {
                const double rp = 1.0 / Ucol[p + k];
                double xr[4], uc[4];
#pragma unroll
                for (int co = 0; co < 4; ++co) xr[co] = Xrow[p + 4 * bc + co] * rp;
#pragma unroll
                for (int ro = 0; ro < 4; ++ro) uc[ro] = (4 * br + ro < k) ? Ucol[p + 4 * br + ro] : 0.0;
#pragma unroll
                for (int ro = 0; ro < 4; ++ro)
#pragma unroll
                    for (int co = 0; co < 4; ++co) ui[ro][co] -= uc[ro] * xr[co];
                if (br == kb) {
#pragma unroll
                    for (int co = 0; co < 4; ++co) ui[kj][co] = xr[co];
                }
            }
        }
    }
    // ---- results: U^-1, T (row-major, ld 64) and S' to scratch for tsqr_apply_kernel; L (unit lower) -> X, T -> Y for W_top = L T
    double *Ui = scr + g.offUi, *Tm = scr + g.offT, *Sm = scr + g.offS;
    tq_bar();
#pragma unroll
    for (int ro = 0; ro < 4; ++ro)
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            const int r = 4 * br + ro, c = 4 * bc + co;
            Ui[r * TB + c] = r <= c ? ui[ro][co] : 0.0;
            Tm[r * TB + c] = r <= c ? t[ro][co] : 0.0;
            X[r + XS * c] = r > c ? x[ro][co] : (r == c ? 1.0 : 0.0);
            Y[r + XS * c] = r <= c ? t[ro][co] : 0.0;
        }
    if (tid < TB) Sm[tid] = sg[tid];
    tq_bar();
    // ---- top 64 rows of V (= L: exact unit lower triangle) and of W = L T
    double *buf = g.buf + (long)ch * g.bsBuf, *Wg = g.W + (long)ch * g.bsW;
    {
        const int r4 = 4 * (tid & 15), c4 = 4 * (tid >> 4);
        double acc[4][4];
#pragma unroll
        for (int xx = 0; xx < 4; ++xx)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[xx][y] = 0.0;
#pragma unroll 4
        for (int k = 0; k < TB; ++k) {                                     // L(r, k) = 0 for k > r and T(k, c) = 0 for k > c: stored zeros
            double lv[4], tv[4];
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) lv[xx] = X[(r4 + xx) + XS * k];
#pragma unroll
            for (int y = 0; y < 4; ++y) tv[y] = Y[k + XS * (c4 + y)];
#pragma unroll
            for (int xx = 0; xx < 4; ++xx)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[xx][y] += lv[xx] * tv[y];
        }
#pragma unroll
        for (int xx = 0; xx < 4; ++xx)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int r = r4 + xx, c = c4 + y;
                Wg[r + ld * c] = acc[xx][y];
                const double lv = X[r + XS * c];
                buf[r + ld * c] = lv;
                buf[r + ld * (2 * TB + c)] = lv;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// V = Q_block (B U^-1), W = V T for one 256-row block; S' R on top of the panel, zeros below it.
constexpr int BS = 66;                                                    // leading dimension of the row-major LDS matrices here
constexpr int TQ_LDS_APPLY = 2 * TB * BS * 8;                             // 67 584 bytes

// C = A B, 64 x 64, on the matrix cores: A and C row-major in LDS (ld BS), B from global memory (or LDS) through its strides, fetched
// straight into MFMA operand registers (all 64 loads of a lane in flight before the first MFMA).  Wave w computes rows 16 w .. 16 w + 15:
// a = A[m = lane & 15][k = lane >> 4], b = B[k = lane >> 4][n = lane & 15], acc[r] = C[m = (lane >> 4) + 4 r][n = lane & 15].
__device__ __forceinline__ void tq_mm64m(const double *A, const double *B, long sBk, long sBn, double *C, int lane, int wave)
{
    const int l15 = lane & 15, l4 = lane >> 4;
    double4_t acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {                    // two halves of K: 32 loads of a lane in flight at a time
        double bf[8][4];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[q][nt] = B[(long)(4 * (8 * h + q) + l4) * sBk + (long)(16 * nt + l15) * sBn];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double av = A[(16 * wave + l15) * BS + 4 * (8 * h + q) + l4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bf[q][nt], acc[nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(16 * wave + l4 + 4 * r) * BS + 16 * nt + l15] = acc[nt][r];
}

template <int MINW>
__global__ __launch_bounds__(256, MINW) void tsqr_apply_kernel(TsqrArgs g)
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
        double tmp[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) tmp[u] = Ui[tid + 256 * u];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int idx = tid + 256 * u; M0[(idx >> 6) * BS + (idx & 63)] = tmp[u]; }
        tq_bar();
        __syncthreads();
        BV = M0; BW = M1;
    } else {
        int idx = leaf;
        double *cur = M0, *oth = M1;
        {
            const int k = idx & 3; idx >>= 2;
            const double *Qp = Qall + (long)(pl.base[1] + idx) * (TR * TB) + TB * k;
            double tmp[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int e = tid + 256 * u; tmp[u] = Qp[(e & 63) + TR * (e >> 6)]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int e = tid + 256 * u; cur[(e & 63) * BS + (e >> 6)] = tmp[u]; }
        }
        tq_bar();
        __syncthreads();
        for (int l = 2; l < pl.nlev; ++l) {
            const int k = idx & 3; idx >>= 2;
            const double *Qp = Qall + (long)(pl.base[l] + idx) * (TR * TB) + TB * k;
            tq_mm64m(cur, Qp, 1, TR, oth, lane, wave);
            tq_bar();
            __syncthreads();
            double *t = cur; cur = oth; oth = t;
        }
        tq_mm64m(cur, Ui, TB, 1, oth, lane, wave);
        tq_bar();
        __syncthreads();
        BV = oth; BW = cur;
    }
    tq_mm64m(BV, Tm, TB, 1, BW, lane, wave);
    // meanwhile-independent: the panel itself becomes [S' R; 0]
    {
        double *P = g.A + (long)ch * g.bsA + (long)g.c0 * ld + g.r0;
        const double *Rr = scr + g.offR + (long)pl.base[pl.nlev - 1] * (TB * TB);
        const double *Sm = scr + g.offS;
        const int row0 = TR * leaf;
        // thread = one row of the block: 64 stores of zero (or of S' R in the top 64 rows), no loads outside the top block
        const int r = row0 + tid;
        if (r < g.m) {
            if (r < TB) {
                const double sr = Sm[r];
#pragma unroll 8
                for (int c = 0; c < TB; ++c) P[r + ld * c] = r <= c ? sr * Rr[r + TB * c] : 0.0;
            } else {
#pragma unroll 8
                for (int c = 0; c < TB; ++c) P[r + ld * c] = 0.0;
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
    g.offT = g.offUi + TB * TB; g.offS = g.offT + TB * TB;
    g.cntr = cntr;
    g.bsCntr = tsqr_cntr_ints(npad);
    static bool attr = false;
    if (!attr) {
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tsqr_tree_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, TQ_LDS_TREE));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tsqr_apply_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, TQ_LDS_APPLY));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tsqr_tree_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, TQ_LDS_TREE));
        BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tsqr_apply_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, TQ_LDS_APPLY));
        attr = true;
    }
    const dim3 grid(g.plan.cnt[0], batch);
    {
        KScope kt(KS_PANEL_QR, st);
        // tsqr_regcap = 1: the variants limited to 256 registers (they fit beside ONE rank-128-update workgroup of a CU, at the price
        // of a few spills outside the column loops)
        if (opts().tsqr_regcap) hipLaunchKernelGGL(tsqr_tree_kernel<2>, grid, dim3(256), TQ_LDS_TREE, st, g);
        else hipLaunchKernelGGL(tsqr_tree_kernel<1>, grid, dim3(256), TQ_LDS_TREE, st, g);
    }
    {
        KScope kt(KS_CHAIN, st);
        if (opts().tsqr_regcap) hipLaunchKernelGGL(tsqr_apply_kernel<2>, grid, dim3(256), TQ_LDS_APPLY, st, g);
        else hipLaunchKernelGGL(tsqr_apply_kernel<1>, grid, dim3(256), TQ_LDS_APPLY, st, g);
    }
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
