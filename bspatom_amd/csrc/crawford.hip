// crawford.hip -- band route: the banded pencil (H_l, S) to a BANDED standard-form matrix, without the dense C_l.
//
// Replaces DPOTRF + DSYGST + the dense part of DSYTRD inside DSYGV (reference call matrices.f90:248) for pencils whose
// half-width k - 1 is at most 8 (every BASELINE config except C5): the dense route forms C_l = L^-1 H_l L^-T (n^2 doubles) and
// reduces it to band form in 4/3 n^3 flop; here the band survives the reduction (Crawford 1973; blocked as in Lang 2019):
//
//   L = R_0 R_1 ... R_{N-1}, R_j = identity except block row j = [L_{j,j-1}, L_jj]  (8 x 8 blocks, N = ceil(n / 8)).
//   Step j ("elimination"): A <- R_j^-1 A R_j^-T touches block row / column j only and leaves ONE 8 x 8 block of fill, at
//   (j, j-2).  It is chased off the top: item (j, s), p = j - 2 - s, takes the RQ factorisation [A(p+2,p), A(p+2,p+1)] = [0 R] Q^T
//   and applies Q to block columns and rows (p, p+1), which moves the fill to (p+1, p-1).  Q mixes blocks below j only and
//   therefore commutes with every later R_k.  The result is block tridiagonal -- and, the sub-diagonal blocks being upper triangular
//   at the end of every sweep (crawford_corner_kernel below), of half-width 8: the input of the one-column chase on tiles of 8
//   (sbr2.hip::sbr_rows_kernel<8>); 6 n^2 b flop instead of 4/3 n^3.
//
// Both kinds of work are ONE shape: a 16 x 16 matrix Q applied as a congruence to the window W = [D_p E_p^T; E_p D_{p+1}] and
// from the left to the block in front of it, [E_{p-1}; 0] -> [E_{p-1}'; fill].  For a chase item Q is orthogonal (eight
// Householder reflectors from the RQ loop), for the elimination it is R_j^-T restricted to the window, [I -K^T; 0 Li^T] with
// Li = L_jj^-1, K = Li L_{j,j-1} -- formed once per solve, S being the same for every channel.
//
// Schedule: item (j, s) in wavefront t = 2 j + s, the elimination of step j in wavefront 2 j - 1; the items of one wavefront
// touch disjoint blocks (tools/proto_crawford.py checks it with a write log), so a wavefront is one launch over all channels:
// no flags, no spinning, bit-identical by construction.  3 N - 5 launches.
//
// One WAVE per item, no LDS memory.  16 x 16 matrices live in the accumulator layout of v_mfma_f64_16x16x4 (register q of lane
// l holds M[4 q + (l >> 4)][l & 15]): a matrix row is a DPP row of 16 lanes, so the RQ loop's dot products are row rotations
// (row_ror 8, 4, 2, 1: every lane ends with the same bits), and the layout is at once the B operand of the matrix cores and --
// for the transpose -- the A operand: P = W Q and W' = Q^T P need no transposition because W is symmetric.
//
// ORIENTATION.  The reduction runs on the index-REVERSED pencil: the fill is chased towards large r, where the entries of H are
// small.  Chased towards r = 0 it passes through the centrifugal term l (l + 1) / r^2 of every row above it and the eigenvalues
// next to zero lose a factor of 50 at l = 14 (0.015 against 0.0003 eps lambda_max, measured against 113-bit truth by the
// prototype).  The band handed to the tridiagonalisation is stored in the original order again.
#include "common.h"
#include <cstdio>
#include <mutex>

namespace bsp {
namespace {

constexpr int CB = 8;              // block size = largest half-width of the pencil this route takes
constexpr int CBB = CB * CB;

template <int CTRL>
__device__ __forceinline__ double cw_dpp(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, true);
    return r.d;
}
// sum over the 16 lanes of a DPP row, the same bits in every lane (a butterfly of commutative additions)
__device__ __forceinline__ double cw_rowsum(double x)
{
    x += cw_dpp<0x128>(x);         // row_ror:8
    x += cw_dpp<0x124>(x);
    x += cw_dpp<0x122>(x);
    x += cw_dpp<0x121>(x);
    return x;
}
__device__ __forceinline__ double cw_bperm(double x, int srclane)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_ds_bpermute(srclane << 2, u.i[0]);
    r.i[1] = __builtin_amdgcn_ds_bpermute(srclane << 2, u.i[1]);
    return r.d;
}
template <int L>
__device__ __forceinline__ double cw_lane(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], L);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], L);
    return r.d;
}

// One step of the RQ factorisation of X = [F E] (8 x 16; x0: rows 0-3, x1: rows 4-7; lane (g, c) holds rows g and 4 + g at
// column c): the reflector H = I - t u u^T on columns 0 .. 8 + I that leaves row I as (0 .. 0, beta, *), applied to the rows
// above it.  u is NOT scaled to a unit last component: u = (x_0 .. x_{LEN-1}, alpha - beta), t = -1 / (beta (alpha - beta))
// = 2 / u^T u -- one division per reflector instead of two, no multiplication of the vector (ONEDIV; an A/B switch: 4 ms faster,
// and its rounding moves the accuracy figures by what separates any two stable routes -- enough to trip the ratchet on one case,
// so LAPACK's dlarfg form with two divisions is the default).  Rows below I have zeros wherever u
// has not and are left as they are.  The reflector is accumulated into Q <- Q H.
// (Measured and not kept, profiles/r04_experiments.txt: Q built after the loop by left multiplications H_i Q with the row vector
// u_i^T Q on the matrix cores -- 4 MFMAs instead of four 16-lane sums per reflector, 24 % fewer instructions -- is SLOWER, 59.8
// against 56.2 ms for the stage: on gfx950 an fp64 MFMA costs what its 1024 multiply-adds cost on the vector pipe, and a row
// vector replicated over 16 rows wastes 15 / 16 of them.)
template <int I, bool ONEDIV>
__device__ __forceinline__ void rq_step(double &x0, double &x1, double (&q)[4], const int g, const int c)
{
    constexpr int LEN = CB + I, GI = I & 3;
    double &xr = (I < 4) ? x0 : x1;
    const double xi = cw_bperm(xr, GI * 16 + c);                       // row I in every DPP row
    const double sig = cw_rowsum(c < LEN ? xi * xi : 0.0);
    const double alpha = cw_lane<LEN>(xi);
    const double a2s = fma(alpha, alpha, sig);
    const bool ok = (a2s > 1e-280) && (sig != 0.0);                    // nothing (numerically) left of the pivot: H = I
    const double nrm = sqrt(ok ? a2s : 1.0);
    const double bt = (alpha >= 0.0) ? -nrm : nrm;
    const double amb = alpha - bt;                                     // |alpha| + nrm with alpha's sign: no cancellation
    // ONEDIV = false: LAPACK's form (dlarfg), u scaled to a unit last component, tau = (beta - alpha) / beta: two divisions
    const double sc = ONEDIV ? 1.0 : (ok ? 1.0 / amb : 0.0);
    const double t = ok ? (ONEDIV ? -1.0 / (bt * amb) : (bt - alpha) / bt) : 0.0;
    const double beta = ok ? bt : alpha;
    const double u = (c < LEN) ? (ONEDIV ? xi : xi * sc) : ((c == LEN) ? (ONEDIV ? amb : 1.0) : 0.0);
    if (I > 0) {
        const double w0 = cw_rowsum(x0 * u);
        x0 = fma(-t * w0, u, x0);
    }
    if (I > 4) {
        const double w1 = cw_rowsum(x1 * u);
        x1 = fma(-t * w1, u, x1);
    }
    if (g == GI) xr = (c < LEN) ? 0.0 : ((c == LEN) ? beta : xr);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double wq = cw_rowsum(q[r] * u);
        q[r] = fma(-t * wq, u, q[r]);
    }
}

__device__ __forceinline__ double4_t cw_mfma(double a, double b, double4_t acc)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
}

// Wavefront t: chase items j = jlo .. jlo + nch - 1 (s = t - 2 j, p = j - 2 - s) and, if jel > 0, the elimination of step jel
// (window (jel - 1, jel)).  grid = (ceil(items / 4), channels), one wave per item.
// nlh / qstride / nlead: channels blockIdx.y >= nlh take their elimination transforms from Qel + qstride and have nlead blocks (the run
// from both ends, below: the second half of the batch is the leading part of the pencil, with the factor of S itself)
template <bool ONEDIV>
__global__ __launch_bounds__(256) void crawford_item_kernel(int N, int t, int jlo, int nch, int jel, int nlh, int qstride, int nlead,
                                                           const double *__restrict__ Qel, double *Dall, double *Eall, double *Gall)
{
    if ((int)blockIdx.y >= nlh) {                                          // a leading part of nlead blocks: the items beyond do not exist
        if (jlo + nch > nlead) nch = nlead > jlo ? nlead - jlo : 0;
        if (jel >= nlead) jel = 0;
    }
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15, c8 = c & 7;
    const bool left = c < CB;
    bool elim;
    int j, p;
    if (idx < nch) { elim = false; j = jlo + idx; p = j - 2 - (t - 2 * j); }
    else if (idx == nch && jel > 0) { elim = true; j = jel; p = j - 1; }
    else return;
    const size_t chn = (size_t)blockIdx.y * N * CBB;
    double *D = Dall + chn, *E = Eall + chn, *G = Gall + chn;
    double *D0 = D + (size_t)p * CBB, *D1 = D0 + CBB, *E0 = E + (size_t)p * CBB;

    // ---- loads: the window, symmetric by construction (lower triangles of D), and the block in front of it ----
    double w[4], sd[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int R = 4 * r + g;
        const int hi = R > c8 ? R : c8, lo = R > c8 ? c8 : R;
        w[r] = *(left ? D0 + hi * CB + lo : E0 + c8 * CB + R);
        w[r + 2] = *(left ? E0 + R * CB + c8 : D1 + hi * CB + lo);
    }
    const bool side = p >= 1;
    {
        const double *Em = E + (size_t)(side ? p - 1 : 0) * CBB;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const double v = Em[(4 * r + g) * CB + c8];
            sd[r] = (side && left) ? v : 0.0;
        }
    }
    double q[4], x0 = 0.0, x1 = 0.0, xt[2] = {0.0, 0.0};
    const bool has_x = elim && (j + 1 <= N - 1);
    if (!elim) {
        const double *src = left ? G + (size_t)p * CBB + c8 : E + (size_t)(p + 1) * CBB + c8;
        x0 = src[g * CB];
        x1 = src[(4 + g) * CB];
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = (4 * r + g == c) ? 1.0 : 0.0;
        rq_step<7, ONEDIV>(x0, x1, q, g, c);
        rq_step<6, ONEDIV>(x0, x1, q, g, c);
        rq_step<5, ONEDIV>(x0, x1, q, g, c);
        rq_step<4, ONEDIV>(x0, x1, q, g, c);
        rq_step<3, ONEDIV>(x0, x1, q, g, c);
        rq_step<2, ONEDIV>(x0, x1, q, g, c);
        rq_step<1, ONEDIV>(x0, x1, q, g, c);
        rq_step<0, ONEDIV>(x0, x1, q, g, c);
    } else {
        const double *Qj = Qel + ((int)blockIdx.y >= nlh ? qstride : 0) + (size_t)j * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = Qj[(4 * r + g) * 16 + c];
        const double *Ej = E + (size_t)(has_x ? j : 0) * CBB;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const double v = Ej[c8 * CB + 4 * r + g];                 // (X^T)[8 + 4 r + g][c] = E_j[c][4 r + g]
            xt[r] = (has_x && left) ? v : 0.0;
        }
    }

    // ---- W' = Q^T (W Q) ----
    double4_t P = {0.0, 0.0, 0.0, 0.0}, Wn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) P = cw_mfma(w[r], q[r], P);
#pragma unroll
    for (int r = 0; r < 4; ++r) Wn = cw_mfma(q[r], P[r], Wn);
    // ---- [E_{p-1}'; fill] = Q^T [E_{p-1}; 0] ----
    double4_t O = {0.0, 0.0, 0.0, 0.0};
    O = cw_mfma(q[0], sd[0], O);
    O = cw_mfma(q[1], sd[1], O);

    // ---- stores ----
    if (left) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            D0[(4 * r + g) * CB + c8] = Wn[r];
            E0[(4 * r + g) * CB + c8] = Wn[r + 2];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) D1[(4 * r + g) * CB + c8] = Wn[r + 2];
    }
    if (side && left) {
        double *Em = E + (size_t)(p - 1) * CBB, *Gm = G + (size_t)(p - 1) * CBB;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (!elim) Em[(4 * r + g) * CB + c8] = O[r];              // the elimination leaves E_{p-1} as it is (Q^T = [I 0; -K Li])
            Gm[(4 * r + g) * CB + c8] = O[r + 2];
        }
    }
    if (!elim) {
        if (!left) {
            double *E1 = E + (size_t)(p + 1) * CBB;
            E1[g * CB + c8] = x0;
            E1[(4 + g) * CB + c8] = x1;
        }
    } else if (has_x) {
        // E_j <- E_j Li^T, as (X Q)^T = Q^T X^T with X = [0 E_j]: rows 8 .. 15 of the product hold (E_j Li^T)^T
        double4_t T = {0.0, 0.0, 0.0, 0.0};
        T = cw_mfma(q[2], xt[0], T);
        T = cw_mfma(q[3], xt[1], T);
        if (left) {
            double *Ej = E + (size_t)j * CBB;
#pragma unroll
            for (int r = 0; r < 2; ++r) Ej[c8 * CB + 4 * r + g] = T[r + 2];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// crawford_item4_kernel: FOUR chase items per wave.  The kernel above spends a third of its ~1400 vector instructions per item
// on the DPP moves of 16-lane row sums (every dot product of the RQ loop is a reduction ACROSS lanes).  Here the layout is
// turned round for the RQ loop, as in the band-16 chase (sbr2.hip: "a chase item per row of 16 lanes"): a DPP row of 16 lanes
// owns an item, lane r of it holds ROW r of Q (16 doubles) and, for r < 8, row r of X = [F E] -- the reflector u is the same in
// all 16 lanes (row I of X goes round through 128 bytes of LDS: written by its lane, read back by the row as a broadcast), so
// every dot product is a sum INSIDE a lane: 16 FMAs for four items at once instead of 12 instructions of rotation-and-add for
// four rows of one item.  ~40 instructions per item and reflector instead of ~170.  The Q of each item then goes through LDS
// into the accumulator layout of the matrix cores (2 KB per item) and the congruences run exactly as above, one item at a time.
// The arithmetic of an item is the first kernel's in another summation order (sums over 16 columns run 0 .. 15 inside a lane
// instead of as a butterfly): spectra agree to rounding, not bit for bit; BSP_CW_ITEMS4=0 keeps the first kernel as the cross-check.
constexpr int QLD = 18;                                                  // LDS row stride of a stored Q (doubles): 144 B, 16-byte aligned

template <int I, bool ONEDIV>
__device__ __forceinline__ void rq4_step(double (&x)[16], double (&q)[16], double *bc, const int r)
{
    constexpr int LEN = CB + I;
    // row I of X to every lane of the DPP row (bc: this item's 16 doubles of LDS; one wave, LDS operations complete in order)
    if (r == I) {
#pragma unroll
        for (int c = 0; c < 16; c += 2) *reinterpret_cast<double2 *>(bc + c) = make_double2(x[c], x[c + 1]);
    }
    // The reads below are OTHER lanes' reads of lane I's stores: a data race in the language's per-thread view, so without
    // this barrier the compiler moves the reads of the lanes r != I in front of the store and forwards lane I its own
    // registers (it did: wrong results that changed from run to run).  The hardware needs nothing: one wave's LDS
    // operations complete in order.
    asm volatile("" ::: "memory");
    double u[16];
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        if (c <= LEN) {
            const double2 v = *reinterpret_cast<const double2 *>(bc + c);
            u[c] = v.x; u[c + 1] = v.y;
        }
    }
    asm volatile("" ::: "memory");                                      // the next step's store stays behind these reads
    double sg[4] = {0.0, 0.0, 0.0, 0.0};                                // four partial sums: a chain of 16 dependent FMAs is all latency
#pragma unroll
    for (int c = 0; c < LEN; ++c) sg[c & 3] = fma(u[c], u[c], sg[c & 3]);
    const double sig = (sg[0] + sg[1]) + (sg[2] + sg[3]);
    const double alpha = u[LEN];
    const double a2s = fma(alpha, alpha, sig);
    const bool ok = (a2s > 1e-280) && (sig != 0.0);                    // nothing (numerically) left of the pivot: H = I
    const double nrm = sqrt(ok ? a2s : 1.0);
    const double bt = (alpha >= 0.0) ? -nrm : nrm;
    const double amb = alpha - bt;
    // the two divisions on operands that are safe either way and the results masked by a multiplication (x * 1.0 and x * 0.0 are
    // exact): written as selects of the quotients, the compiler puts each division under a branch of its own and the two
    // dependent sequences of eleven instructions run one after the other instead of side by side
    const double okf = ok ? 1.0 : 0.0, ambs = ok ? amb : 1.0, bts = ok ? bt : 1.0;
    const double sc = ONEDIV ? 1.0 : (1.0 / ambs) * okf;
    const double t = (ONEDIV ? -1.0 / (bts * ambs) : (bts - alpha) / bts) * okf;
    const double beta = ok ? bt : alpha;
    if (!ONEDIV) {
#pragma unroll
        for (int c = 0; c < LEN; ++c) u[c] *= sc;
    }
    u[LEN] = ONEDIV ? amb : 1.0;
    {   // rows up to and INCLUDING I (lanes r <= I); the others keep their x.  Row I goes through the same update as the rows
        // above it: it comes out as (rounding residue .. , beta, *) instead of exact zeros.  Nothing needs them exact: the F part of
        // X is never stored (it IS the annihilated fill), the part of row I left of the pivot inside E_{p+1} is below the diagonal
        // of an 8 x 8 block that is full in this band form anyway, and no later reflector touches a row below its own.  (Writing
        // the zeros was 32 conditional moves per reflector, a tenth of the kernel's instructions.)
        double ws[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c <= LEN; ++c) ws[c & 3] = fma(x[c], u[c], ws[c & 3]);
        const double w = (ws[0] + ws[1]) + (ws[2] + ws[3]);
        const double tw = (r <= I) ? -t * w : 0.0;
#pragma unroll
        for (int c = 0; c <= LEN; ++c) x[c] = fma(tw, u[c], x[c]);
    }
    (void)beta;
    {
        double ws[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c <= LEN; ++c) ws[c & 3] = fma(q[c], u[c], ws[c & 3]);
        const double w = (ws[0] + ws[1]) + (ws[2] + ws[3]);
        const double tw = -t * w;
#pragma unroll
        for (int c = 0; c <= LEN; ++c) q[c] = fma(tw, u[c], q[c]);
    }
}

// The work of one wave on its (up to) four items of wavefront t of channel `chan`: items idx0 .. idx0 + ipw - 1 of the wavefront's
// list (nch chase items j = jlo .., then the elimination jel if any).  Qsw / Bcw: the wave's LDS (4 x 16 x QLD and 4 x 16 doubles).
template <bool ONEDIV>
__device__ __forceinline__ void cw_quad(const int N, const int t, const int jlo, int nch, int jel, const int ipw, const int nlh,
                                        const int qstride, const int nlead, const int chan, const int idx0, const int lane,
                                        double *Qsw, double *Bcw, double *Wsw, const double *__restrict__ Qel, double *Dall, double *Eall,
                                        double *Gall, unsigned long long *diag)
{
    unsigned long long ts0 = 0, tsa = 0, tsb = 0, tsc = 0, tsd = 0, tse = 0;
    if (diag) ts0 = __builtin_amdgcn_s_memtime();
    if (chan >= nlh) {                                                     // a leading part of nlead blocks: the items beyond do not exist
        if (jlo + nch > nlead) nch = nlead > jlo ? nlead - jlo : 0;
        if (jel >= nlead) jel = 0;
    }
    const int items = nch + (jel > 0 ? 1 : 0);
    if (idx0 >= items) return;
    const size_t chn = (size_t)chan * N * CBB;
    double *D = Dall + chn, *E = Eall + chn, *G = Gall + chn;

    // ---- the operands of phase B, requested now by LDS-DMA (no registers: they land in Wsw while phase A computes; none of them is
    // written by phase A, which only stores E_{p+1}).  Per item two instructions of 1 KB: D_p and D_{p+1} are neighbours in memory,
    // E_p goes with the block in front of the window, E_{p-1} (E_p again where there is none). ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = idx0 + s;
        const int kd = s >= ipw ? 0 : (idx < nch ? 1 : ((idx == nch && jel > 0) ? 2 : 0));
        if (kd != 0) {                                                   // uniform
            const int jx = kd == 1 ? jlo + idx : jel, p = kd == 2 ? jel - 1 : jx - 2 - (t - 2 * jx);
            const double *srcD = D + (size_t)p * CBB + 2 * lane;
            const double *srcE = E + (size_t)((lane < 32 || p < 1) ? p : p - 1) * CBB + 2 * (lane & 31);
            // written as assembly: through the builtin the compiler waits for the DMA (vmcnt(0)) before the wave's next LDS operation
            const unsigned ldsD = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(Wsw + s * 4 * CBB);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "v"(srcD), "v"(srcE), "s"(ldsD), "s"(ldsD + 2 * CBB * 8) : "memory", "m0");
        }
    }

    // ---- phase A: the RQ loops of the wave's chase items, an item per DPP row ----
    if (idx0 < nch) {
        const int it = lane >> 4, r = lane & 15;
        const int idx = idx0 + it;
        const bool live = idx < nch && it < ipw && r < CB;               // lanes that hold a row of X
        const int j = jlo + (idx < nch ? idx : 0), p = j - 2 - (t - 2 * j);
        double x[16], q[16];
        {
            const double *gp = G + (size_t)p * CBB + (r & 7) * CB, *ep = E + (size_t)(p + 1) * CBB + (r & 7) * CB;
#pragma unroll
            for (int c = 0; c < CB; c += 2) {
                const double2 a = *reinterpret_cast<const double2 *>(gp + c), b = *reinterpret_cast<const double2 *>(ep + c);
                x[c] = live ? a.x : 0.0; x[c + 1] = live ? a.y : 0.0;
                x[CB + c] = live ? b.x : 0.0; x[CB + c + 1] = live ? b.y : 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) q[c] = (c == r) ? 1.0 : 0.0;
        double *bc = Bcw + it * 16;
        if (diag) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tsa = __builtin_amdgcn_s_memtime(); }   // X has arrived
        rq4_step<7, ONEDIV>(x, q, bc, r);
        rq4_step<6, ONEDIV>(x, q, bc, r);
        rq4_step<5, ONEDIV>(x, q, bc, r);
        rq4_step<4, ONEDIV>(x, q, bc, r);
        rq4_step<3, ONEDIV>(x, q, bc, r);
        rq4_step<2, ONEDIV>(x, q, bc, r);
        rq4_step<1, ONEDIV>(x, q, bc, r);
        rq4_step<0, ONEDIV>(x, q, bc, r);
        if (diag) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsb = __builtin_amdgcn_s_memtime(); }   // the eight reflectors
        if (live) {                                                      // E_{p+1} <- R
            double *ep = E + (size_t)(p + 1) * CBB + r * CB;
#pragma unroll
            for (int c = 0; c < CB; c += 2) *reinterpret_cast<double2 *>(ep + c) = make_double2(x[CB + c], x[CB + c + 1]);
        }
        double *qs = Qsw + it * (16 * QLD) + r * QLD;
#pragma unroll
        for (int c = 0; c < 16; c += 2) *reinterpret_cast<double2 *>(qs + c) = make_double2(q[c], q[c + 1]);
    }
    // this wave's own LDS stores (no other wave reads them) and the DMA of phase B's operands, requested a phase ago: the counter is
    // in order over loads and stores, and the only operations younger than the DMA are the four 16-byte stores of R that a wave with
    // a chase item has just issued (lanes 0 .. 7 of its first item are always live)
    if (idx0 < nch) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // Everything phase B computes from the kernel's arguments (item numbers, block addresses of four items) would otherwise be
    // hoisted in front of phase A and sit in registers across it (96 dwords spilled at three waves per SIMD): the two values it
    // all derives from are redefined here, as far as the compiler can tell.
    if (diag) tsc = __builtin_amdgcn_s_memtime();                        // R and Q stored
    int idx0b = idx0;
    size_t chnb = chn;
    asm volatile("" : "+s"(idx0b), "+s"(chnb));
    D = Dall + chnb; E = Eall + chnb; G = Gall + chnb;

    // ---- phase B: the congruences, one item at a time, in the layout of the matrix cores.  The operands of all four items are
    // requested before the first product (a slot without an item reads slot 0's addresses and is skipped): one round trip to
    // memory per wave instead of one per item. ----
    const int g = lane >> 4, c = lane & 15, c8 = c & 7;
    const bool left = c < CB;
    int kind[4], jj[4], pp[4];                                           // kind: 0 none, 1 chase, 2 elimination
    double wv[4][4], sdv[4][2], qe[4], xt[2] = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = idx0b + s;
        kind[s] = s >= ipw ? 0 : (idx < nch ? 1 : ((idx == nch && jel > 0) ? 2 : 0));
        jj[s] = kind[s] == 1 ? jlo + idx : (kind[s] == 2 ? jel : jlo + idx0b);
        pp[s] = kind[s] == 2 ? jel - 1 : jj[s] - 2 - (t - 2 * jj[s]);
        if (kind[s] == 0) { jj[s] = jj[0]; pp[s] = pp[0]; }
        const int p = pp[s];
        const double *W0 = Wsw + s * 4 * CBB, *W1 = W0 + CBB, *WE = W0 + 2 * CBB, *WM = W0 + 3 * CBB;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int R = 4 * r + g;
            const int hi = R > c8 ? R : c8, lo = R > c8 ? c8 : R;
            wv[s][r] = *(left ? W0 + hi * CB + lo : WE + c8 * CB + R);
            wv[s][r + 2] = *(left ? WE + R * CB + c8 : W1 + hi * CB + lo);
            const double v = WM[(4 * r + g) * CB + c8];
            sdv[s][r] = (p >= 1 && left) ? v : 0.0;
        }
    }
    // the elimination of this wavefront, if this wave holds it (at most one, the last item): its Q and E_j
    int se = -1;
#pragma unroll
    for (int s = 0; s < 4; ++s) if (kind[s] == 2) se = s;
    const bool has_x = se >= 0 && (jel + 1 <= N - 1);
    if (se >= 0) {
        const double *Qj = Qel + ((int)chan >= nlh ? qstride : 0) + (size_t)jel * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) qe[r] = Qj[(4 * r + g) * 16 + c];
        const double *Ej = E + (size_t)(has_x ? jel : 0) * CBB;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const double v = Ej[c8 * CB + 4 * r + g];                 // (X^T)[8 + 4 r + g][c] = E_j[c][4 r + g]
            xt[r] = (has_x && left) ? v : 0.0;
        }
    }
    if (diag) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tsd = __builtin_amdgcn_s_memtime(); }   // phase B's operands have arrived
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (kind[s] == 0) break;
        const bool elim = kind[s] == 2;
        const int p = pp[s];
        double *D0 = D + (size_t)p * CBB, *D1 = D0 + CBB, *E0 = E + (size_t)p * CBB;
        double q[4];
        if (!elim) {
            const double *qs = Qsw + s * (16 * QLD);
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = qs[(4 * r + g) * QLD + c];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = qe[r];
        }
        double4_t P = {0.0, 0.0, 0.0, 0.0}, Wn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) P = cw_mfma(wv[s][r], q[r], P);
#pragma unroll
        for (int r = 0; r < 4; ++r) Wn = cw_mfma(q[r], P[r], Wn);
        double4_t O = {0.0, 0.0, 0.0, 0.0};
        O = cw_mfma(q[0], sdv[s][0], O);
        O = cw_mfma(q[1], sdv[s][1], O);
        if (left) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                D0[(4 * r + g) * CB + c8] = Wn[r];
                E0[(4 * r + g) * CB + c8] = Wn[r + 2];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) D1[(4 * r + g) * CB + c8] = Wn[r + 2];
        }
        if (p >= 1 && left) {
            double *Em = E + (size_t)(p - 1) * CBB, *Gm = G + (size_t)(p - 1) * CBB;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (!elim) Em[(4 * r + g) * CB + c8] = O[r];
                Gm[(4 * r + g) * CB + c8] = O[r + 2];
            }
        }
        if (elim && has_x) {
            double4_t T = {0.0, 0.0, 0.0, 0.0};
            T = cw_mfma(q[2], xt[0], T);
            T = cw_mfma(q[3], xt[1], T);
            if (left) {
                double *Ej = E + (size_t)jel * CBB;
#pragma unroll
                for (int r = 0; r < 2; ++r) Ej[c8 * CB + 4 * r + g] = T[r + 2];
            }
        }
    }
    if (diag) {
        tse = __builtin_amdgcn_s_memtime();                              // the congruences issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tsf = __builtin_amdgcn_s_memtime();    // the stores have left
        if (lane == 0) {
            const bool a = tsa != 0;
            atomicAdd(diag + 0, a ? tsa - ts0 : 0ull);
            atomicAdd(diag + 1, a ? tsb - tsa : 0ull);
            atomicAdd(diag + 2, a ? tsc - tsb : tsc - ts0);
            atomicAdd(diag + 3, tsd - tsc);
            atomicAdd(diag + 4, tse - tsd);
            atomicAdd(diag + 5, tsf - tse);
            atomicAdd(diag + 6, tsf - ts0);
            atomicAdd(diag + 7, 1ull);
        }
    }
}

template <bool ONEDIV, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void crawford_item4_kernel(int N, int t, int jlo, int nch, int jel, int ipw, int nlh, int qstride,
                                                            int nlead, const double *__restrict__ Qel, double *Dall, double *Eall, double *Gall,
                                                            unsigned long long *diag)
{
    // a launch waits for its slowest wave: the waves of these launches go first on a SIMD they share with a long-running wave of
    // another kernel (the consumed eigenvector's workgroup if it was not given a CU of its own, BSP_VEC_OWN_CU=0: it has 60 ms to spare)
    __builtin_amdgcn_s_setprio(3);
    // BSP_CW_DIAG: s_memtime stamps of a wave's phases, summed over the waves of every launch (diag[0..6] ticks, diag[7] waves)
    if (diag && (((blockIdx.x & 7) | (blockIdx.y & 7)) != 0)) diag = nullptr;   // one wave in 64 reports (the sums are atomics)
    __shared__ __attribute__((aligned(16))) double Qs[NW][4][16 * QLD];  // per wave and item slot: Q, row-major, stride QLD
    __shared__ __attribute__((aligned(16))) double Bc[NW][4][16];        // per wave and item slot: the row that goes round
    __shared__ __attribute__((aligned(16))) double Ws[NW][4][4 * CBB];   // per wave and item slot: D_p, D_{p+1}, E_p, E_{p-1} (phase B's operands)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ipw = items per wave: 4; 2 or 1 only as an experiment (BSP_CW_IPW).  An item's arithmetic does not depend on the slot it
    // sits in: the choice changes no bit (tests/test_gpu_solve.py::test_band_route_properties).
    cw_quad<ONEDIV>(N, t, jlo, nch, jel, ipw, nlh, qstride, nlead, (int)blockIdx.y, (blockIdx.x * NW + wave) * ipw, lane, &Qs[wave][0][0],
                    &Bc[wave][0][0], &Ws[wave][0][0], Qel, Dall, Eall, Gall, diag);
}

// index-reversed overlap band: SBf[d][i] = S_f(i, i + d) = S(n-1-i-d, n-1-i)
// ... and, behind it, the band as it is (the pair of factorisations of the run from both ends reads the two side by side)
__global__ void crawford_flip_kernel(int n, int k, const double *__restrict__ SB, double *__restrict__ SBf)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * k) return;
    const int d = idx / n, i = idx % n;
    SBf[idx] = (i + d < n) ? SB[(size_t)d * n + (n - 1 - i - d)] : 0.0;
    SBf[(size_t)n * k + idx] = SB[idx];
}

// Per block j: Li = L_jj^-1, K = Li L_{j,j-1} (L = U^T from the Cholesky factor of the reversed overlap, identity beyond n),
// LiB[j] = Li (row-major 8 x 8) and Qel[j] = [I -K^T; 0 Li^T] (row-major 16 x 16).  One wave per block.
// blockIdx.y = which factor (0: of the reversed overlap; 1: of the overlap itself, the run from both ends only), fstride blocks apart in
// LiB / Qel.  The grid covers the blocks jfirst .. jfirst + gridDim.x - 1 (the factor may arrive in chunks: crawford_prepare).
__global__ __launch_bounds__(64) void crawford_setup_kernel(int n, int k, int fstride, int jfirst, const double *__restrict__ UB0,
                                                           double *__restrict__ LiB0, double *__restrict__ Qel0)
{
    __shared__ double Ld[CB][CB + 1], M[CB][CB + 1], Li[CB][CB + 1];
    const int j = jfirst + blockIdx.x, t = threadIdx.x, r = t >> 3, c = t & 7, b = k - 1;
    const double *UBf = UB0 + (size_t)blockIdx.y * k * n;
    double *LiB = LiB0 + (size_t)blockIdx.y * fstride * CBB, *Qel = Qel0 + (size_t)blockIdx.y * fstride * 256;
    const int i = CB * j + r;
    {
        const int ic = CB * j + c, d = r - c;
        double v = 0.0;
        if (i >= n) v = (r == c) ? 1.0 : 0.0;
        else if (d >= 0 && d <= b) v = UBf[(size_t)d * n + ic];           // L(i, ic) = U(ic, i)
        Ld[r][c] = v;
        const int im = CB * (j - 1) + c, dm = CB + r - c;
        double m = 0.0;
        if (j > 0 && i < n && dm <= b) m = UBf[(size_t)dm * n + im];
        M[r][c] = m;
        Li[r][c] = 0.0;
    }
    __syncthreads();
    if (t < CB) {                                                          // column t of Li by forward substitution
        double x[CB];
#pragma unroll
        for (int rr = 0; rr < CB; ++rr) {
            double s = (rr == t) ? 1.0 : 0.0;
#pragma unroll
            for (int kk = 0; kk < CB; ++kk)
                if (kk < rr && kk >= t) s -= Ld[rr][kk] * x[kk];
            x[rr] = (rr >= t) ? s / Ld[rr][rr] : 0.0;
        }
#pragma unroll
        for (int rr = 0; rr < CB; ++rr) Li[rr][t] = x[rr];
    }
    __syncthreads();
    double kv = 0.0;
#pragma unroll
    for (int kk = 0; kk < CB; ++kk) kv += Li[r][kk] * M[kk][c];            // K[r][c]
    LiB[(size_t)j * CBB + t] = Li[r][c];
    double *Q = Qel + (size_t)j * 256;
    Q[r * 16 + c] = (r == c) ? 1.0 : 0.0;                                  // I
    Q[c * 16 + CB + r] = -kv;                                              // -K^T: Q[c][8 + r] = -K[r][c]
    Q[(CB + r) * 16 + c] = 0.0;
    Q[(CB + c) * 16 + CB + r] = Li[r][c];                                  // Li^T
}

// D_p, E_p of the index-reversed H_l (upper band HB[d][i] = H(i, i + d)); block 0 is taken through step 0 of the
// elimination here: D_0 <- Li_0 D_0 Li_0^T, E_0 <- E_0 Li_0^T.  grid = (N, channels), one wave per block pair.
// Run from both ends (nlh < number of channels in the grid): channel blockIdx.y >= nlh is the LEADING part of the pencil of channel
// blockIdx.y - nlh in the original order, with Li_0 of the overlap's own factor (LiB + listride); N is the stride of a channel in
// blocks (the grid has the blocks of a half).
__global__ __launch_bounds__(64) void crawford_init_kernel(int n, int k, int N, int nlh, int listride, const double *__restrict__ HBall,
                                                          const double *__restrict__ LiB, double *Dall, double *Eall)
{
    __shared__ double Ds[CB][CB + 1], Es[CB][CB + 1], Ls[CB][CB + 1], Ts[CB][CB + 1];
    const int p = blockIdx.x, t = threadIdx.x, r = t >> 3, c = t & 7, b = k - 1;
    const bool fwd = (int)blockIdx.y >= nlh;
    const double *HB = HBall + (size_t)(fwd ? blockIdx.y - nlh : blockIdx.y) * k * n;
    auto Hf = [&](int i, int i2) -> double {                               // reversed indices (original ones for the leading part)
        const int hi = i > i2 ? i : i2, lo = i > i2 ? i2 : i, d = hi - lo;
        return (hi < n && d <= b) ? HB[(size_t)d * n + (fwd ? lo : n - 1 - hi)] : 0.0;
    };
    double dv = Hf(CB * p + r, CB * p + c), ev = Hf(CB * (p + 1) + r, CB * p + c);
    if (p == 0) {
        Ds[r][c] = dv; Es[r][c] = ev; Ls[r][c] = LiB[(fwd ? listride : 0) + t];
        __syncthreads();
        double s = 0.0, e2 = 0.0;
#pragma unroll
        for (int kk = 0; kk < CB; ++kk) { s += Ls[r][kk] * Ds[kk][c]; e2 += Es[r][kk] * Ls[c][kk]; }
        Ts[r][c] = s;
        __syncthreads();
        s = 0.0;
#pragma unroll
        for (int kk = 0; kk < CB; ++kk) s += Ts[r][kk] * Ls[c][kk];
        dv = s; ev = e2;
    }
    const size_t o = ((size_t)blockIdx.y * N + p) * CBB + t;
    Dall[o] = dv;
    Eall[o] = ev;
}

// ------------------------------------------------------------------------------------------------------------------------------
// THE RUN FROM BOTH ENDS (BSP_CW_SPLIT = share of the leading part in percent; an OPTION, not the default: it is faster -- 21.8
// instead of 38.2 ms at 128 channels with the cut in the middle -- and it is less accurate next to zero, because the leading part's
// fill is chased towards r = 0: ORIENTATION above; measured in profiles/r04_experiments.txt, 11).  The fill of elimination step j is
// chased over j blocks: N^2 / 2 chase items.  Cut the pencil, in the middle say: the trailing part in reversed order (channels
// 0 .. nl - 1 of the batch, exactly the process above on its first Nh blocks) and the leading part in the original order (channels
// nl .. 2 nl - 1: the same process with the factor of S itself, Nl = N - Nh <= Nh blocks; the kernels skip the items a shorter part does
// not have) are two independent pencils -- each chases its fill away from the cut, to its own end of the matrix:
// 2 (N/2)^2 / 2 = N^2 / 4 items in 3 Nh - 5 launches.  Neither touches the other's blocks: the chase
// of a part never reaches its last block, and the block that couples the parts is only ever scaled,
//     C = flip(Li_1) H(cut) Li_0^T   (rows: the leading part's last block, reversed; columns: the trailing part's last block),
// by the two eliminations next to the cut.  What is left of S is the identity plus the same expression of S's coupling block, Gamma:
// [I Gamma^T; Gamma I] = [I 0; Gamma C][..]^T with C C^T = I - Gamma Gamma^T, i.e. ONE more elimination step, in the reversed numbering
// at block Nh with L(Nh, Nh-1) = Gamma, L(Nh, Nh) = C.  Its fill is chased through the trailing part (towards large r: ORIENTATION
// above): Nh - 1 items, one per launch.  For that sweep the leading part's last block joins the trailing part's arrays as block Nh
// (a channel has Nh + 2 block slots): D[Nh] = its last diagonal block, flipped, E[Nh] = its last sub-diagonal block, flipped and
// transposed; crawford_band_kernel reads them back from there.  tools/proto_crawford_split.py states all of it densely.
//
// Qel[Nh] of factor 0 = [I -K^T; 0 Li^T] with Li = C^-1, K = Li Gamma.  One wave.
__global__ __launch_bounds__(64) void crawford_cut_setup_kernel(int n, int k, int N, int Nh, int Nl, const double *__restrict__ SBf,
                                                               const double *__restrict__ LiB, double *__restrict__ Qel, int *info)
{
    __shared__ double A[CB][CB + 1], Lf[CB][CB + 1], T[CB][CB + 1], Gm[CB][CB + 1], Li[CB][CB + 1];
    const int t = threadIdx.x, r = t >> 3, c = t & 7, b = k - 1;
    const double *L0 = LiB + (size_t)(Nh - 1) * CBB, *L1 = LiB + (size_t)N * CBB + (size_t)(Nl - 1) * CBB;
    {
        const int d = CB + r - c;                                          // S_f(8 Nh + r, 8 (Nh - 1) + c)
        A[r][c] = (d <= b) ? SBf[(size_t)d * n + CB * (Nh - 1) + c] : 0.0;
        Lf[r][c] = L1[(CB - 1 - r) * CB + (CB - 1 - c)];
        Li[r][c] = 0.0;
    }
    __syncthreads();
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < CB; ++q) v += Lf[r][q] * A[q][c];
    T[r][c] = v;
    __syncthreads();
    v = 0.0;
#pragma unroll
    for (int q = 0; q < CB; ++q) v += T[r][q] * L0[c * CB + q];
    Gm[r][c] = v;
    __syncthreads();
    v = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < CB; ++q) v -= Gm[r][q] * Gm[c][q];
    A[r][c] = v;                                                           // I - Gamma Gamma^T
    __syncthreads();
    if (t == 0) {                                                          // its Cholesky factor (lower), in place: 8 x 8, one lane
        for (int j = 0; j < CB; ++j) {
            double dj = A[j][j];
            for (int q = 0; q < j; ++q) dj -= A[j][q] * A[j][q];
            if (!(dj > 0.0)) { if (*info == 0) *info = n; dj = 1.0; }      // S is not positive definite across the cut
            dj = sqrt(dj);
            A[j][j] = dj;
            for (int i = j + 1; i < CB; ++i) {
                double e = A[i][j];
                for (int q = 0; q < j; ++q) e -= A[i][q] * A[j][q];
                A[i][j] = e / dj;
            }
        }
    }
    __syncthreads();
    if (t < CB) {                                                          // column t of C^-1 by forward substitution
        double x[CB];
#pragma unroll
        for (int rr = 0; rr < CB; ++rr) {
            double sacc = (rr == t) ? 1.0 : 0.0;
#pragma unroll
            for (int kk = 0; kk < CB; ++kk)
                if (kk < rr && kk >= t) sacc -= A[rr][kk] * x[kk];
            x[rr] = (rr >= t) ? sacc / A[rr][rr] : 0.0;
        }
#pragma unroll
        for (int rr = 0; rr < CB; ++rr) Li[rr][t] = x[rr];
    }
    __syncthreads();
    double kv = 0.0;
#pragma unroll
    for (int kk = 0; kk < CB; ++kk) kv += Li[r][kk] * Gm[kk][c];           // K[r][c]
    double *Q = Qel + (size_t)Nh * 256;
    Q[r * 16 + c] = (r == c) ? 1.0 : 0.0;
    Q[c * 16 + CB + r] = -kv;
    Q[(CB + r) * 16 + c] = 0.0;
    Q[(CB + c) * 16 + CB + r] = Li[r][c];
}

// Per channel, before the elimination at the cut: the coupling block (E[Nh-1] of the trailing part, scaled from the right by that
// part's last elimination already) takes the leading part's scaling from the left, and the leading part's last blocks move into
// slot Nh of the trailing part's arrays.  N = block slots per channel, Nfull = blocks between the two factors in LiB.
__global__ __launch_bounds__(64) void crawford_cut_prep_kernel(int N, int Nh, int Nl, int nl, int Nfull, const double *__restrict__ LiB,
                                                              double *Dall, double *Eall)
{
    __shared__ double X[CB][CB + 1], Lf[CB][CB + 1];
    const int t = threadIdx.x, r = t >> 3, c = t & 7;
    const size_t c0 = (size_t)blockIdx.x * N * CBB, c1 = (size_t)(nl + blockIdx.x) * N * CBB;
    const double *L1 = LiB + (size_t)Nfull * CBB + (size_t)(Nl - 1) * CBB;
    X[r][c] = Eall[c0 + (size_t)(Nh - 1) * CBB + t];
    Lf[r][c] = L1[(CB - 1 - r) * CB + (CB - 1 - c)];
    __syncthreads();
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < CB; ++q) v += Lf[r][q] * X[q][c];
    Eall[c0 + (size_t)(Nh - 1) * CBB + t] = v;
    Dall[c0 + (size_t)Nh * CBB + t] = Dall[c1 + (size_t)(Nl - 1) * CBB + (CB - 1 - r) * CB + (CB - 1 - c)];
    Eall[c0 + (size_t)Nh * CBB + t] = Eall[c1 + (size_t)(Nl - 2) * CBB + (CB - 1 - c) * CB + (CB - 1 - r)];
}

// block tridiagonal (reversed order) -> lower band storage of the one-column chase in the ORIGINAL order:
// AB[j * 128 + d] = A(j + d, j), d = 0 .. 31 (zero beyond the half-width handed over -- 8, or 15 with BSP_CW_BAND8=0 -- and beyond the matrix)
// What the reduction leaves is narrower than block tridiagonal: E_1 .. E_{N-2} are upper triangular -- each is the R of the last RQ
// factorisation that touched it (item (j, s) leaves E_{p+1} = R, the next item of the sweep fills E_p again and the one after it
// re-factors that; tools/proto_crawford.py and tests/test_host_cpu.py check it on the dense statement) -- so the matrix has
// half-width CB except for E_0, which the last item of every sweep leaves full.  One more RQ, E_0 = R Q^T, D_0 <- Q^T D_0 Q (block 0
// touches nothing else), and the chase that follows works on a band of half-width 8 instead of 15.  One wave per channel: lane
// i < 8 holds row i of E_0, lane 8 + i row i of D_0.
__device__ __forceinline__ double cw_rdl(double x, int l)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return r.d;
}
__global__ __launch_bounds__(64) void crawford_corner_kernel(int N, double *__restrict__ Dall, double *__restrict__ Eall)
{
    const int lane = threadIdx.x, i = lane & 7;
    const bool isE = lane < 8, isD = lane >= 8 && lane < 16;
    double *rowp = (isE ? Eall : Dall) + (size_t)blockIdx.x * N * CBB + i * CB;
    double a[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) a[c] = (isE || isD) ? rowp[c] : 0.0;
#pragma unroll
    for (int r = CB - 1; r >= 1; --r) {
        // the reflector H = I - tau v v^T on columns 0 .. r that leaves row r of E_0 as (0 .. 0, beta, *)
        double v[CB], nrm2 = 0.0;
#pragma unroll
        for (int c = 0; c < CB; ++c) v[c] = c <= r ? cw_rdl(a[c], r) : 0.0;
#pragma unroll
        for (int c = 0; c < r; ++c) nrm2 = fma(v[c], v[c], nrm2);
        if (nrm2 == 0.0) continue;                                         // uniform: the row is in shape already
        const double alpha = v[r], beta = -copysign(sqrt(fma(alpha, alpha, nrm2)), alpha);
        const double tau = (beta - alpha) / beta, scale = 1.0 / (alpha - beta);
#pragma unroll
        for (int c = 0; c < r; ++c) v[c] *= scale;
        v[r] = 1.0;
        // rows of E_0 and of D_0 alike: a <- a H
        double sdot = 0.0;
#pragma unroll
        for (int c = 0; c <= r; ++c) sdot = fma(a[c], v[c], sdot);
        sdot *= tau;
#pragma unroll
        for (int c = 0; c <= r; ++c) a[c] = fma(-sdot, v[c], a[c]);
        if (isE && i == r) {
#pragma unroll
            for (int c = 0; c < r; ++c) a[c] = 0.0;
            a[r] = beta;
        }
        // D_0 <- H (D_0 H): row i -= tau v_i (v^T (D_0 H))
        double vi = 0.0;
#pragma unroll
        for (int c = 0; c <= r; ++c) vi = (i == c) ? v[c] : vi;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            double y = 0.0;
#pragma unroll
            for (int q = 0; q <= r; ++q) y = fma(v[q], cw_rdl(a[c], CB + q), y);
            if (isD) a[c] = fma(-tau * vi, y, a[c]);
        }
    }
    if (isE || isD) {
#pragma unroll
        for (int c = 0; c < CB; ++c) rowp[c] = a[c];
    }
}

// hw = half-width handed over: 2 CB - 1 (the block tridiagonal as it stands) or CB (after crawford_corner_kernel; what lies
// beyond is rounding residue of the RQ factorisations, where LAPACK's own reductions store exact zeros)
// N = block slots per channel; Nh (blocks of the trailing part), Ntot (of the pencil), nl: the run from both ends (Nh = 0: one process
// over all N blocks)
__global__ void crawford_band_kernel(int n, int npad, int N, int hw, int Nh, int Ntot, int nl, const double *__restrict__ Dall,
                                     const double *__restrict__ Eall, double *__restrict__ ABall)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npad * 32) return;
    const int j = idx >> 5, d = idx & 31;
    const size_t ch = blockIdx.y;
    double v = 0.0;
    if (j + d < n && d <= hw) {
        const int ihi = n - 1 - j, ilo = ihi - d;                          // reversed indices, ihi >= ilo
        const int P = ihi >> 3, Pc = ilo >> 3;                             // P - Pc = 2 (d >= 9 only): outside the block tridiagonal
        const int rr = ihi & 7, cc = ilo & 7;
        if (Nh == 0) {
            const double *blk = (P == Pc ? Dall : Eall) + (ch * N + Pc) * CBB;
            v = (P - Pc <= 1) ? blk[rr * CB + cc] : 0.0;
        } else if (P - Pc <= 1) {
            // run from both ends: blocks 0 .. Nh (diagonal) / 0 .. Nh (sub-diagonal) of the reversed numbering from the trailing part's
            // arrays, the others from the leading part's (channel nl + ch), flipped: reversed block P is its block Ntot - 1 - P
            const double *t0 = (P == Pc ? Dall : Eall) + ch * N * CBB, *t1 = (P == Pc ? Dall : Eall) + (nl + ch) * N * CBB;
            if (Pc <= Nh) v = t0[(size_t)Pc * CBB + rr * CB + cc];
            else if (P == Pc) v = t1[(size_t)(Ntot - 1 - P) * CBB + (CB - 1 - cc) * CB + (CB - 1 - rr)];   // its lower triangle
            else v = t1[(size_t)(Ntot - 2 - Pc) * CBB + (CB - 1 - cc) * CB + (CB - 1 - rr)];
        }
    }
    ABall[ch * ab_stride(npad) + (size_t)j * 128 + d] = v;
}

}  // namespace

// both forms of the run fit: two bands, two factors, two sets of transforms; 2 (N + 2) block slots per channel (the run from both ends
// gives each part the slots of the longer one + 2)
size_t crawford_work_bytes(int n, int k, int nl)
{
    const size_t N = (n + CB - 1) / CB;
    return ((size_t)4 * k * n + 2 * n + 2 * N * 256 + 2 * N * CBB + (size_t)3 * nl * 2 * (N + 2) * CBB) * sizeof(double) + 64;
}

void crawford_carve(void *base, int n, int k, int nl, CrawfordWork *w)
{
    const size_t N = (n + CB - 1) / CB;
    double *p = static_cast<double *>(base);
    w->SBf = p; p += (size_t)2 * k * n;
    w->UBf = p; p += (size_t)2 * k * n;
    w->rdiagf = p; p += 2 * n;
    w->Qel = p; p += 2 * N * 256;
    w->LiB = p; p += 2 * N * CBB;
    w->D = p; p += (size_t)nl * 2 * (N + 2) * CBB;
    w->E = p; p += (size_t)nl * 2 * (N + 2) * CBB;
    w->G = p; p += (size_t)nl * 2 * (N + 2) * CBB;
    w->info = reinterpret_cast<int *>(p);
}

bool crawford_supported(int n, int k) { return k >= 2 && k - 1 <= CB && n >= 2 * CB; }

// w.info (device) receives the order of the minor of the REVERSED overlap at which its Cholesky factorisation broke down, or 0
// (run from both ends: nonzero if one of the three factorisations broke down -- the caller finds DSYGV's own info with the
// forward factorisation).
namespace {
struct CwShape { int N, Nl, Nh, Nproc, Ns; bool split; };
CwShape cw_shape(int n)
{
    CwShape c;
    c.N = (n + CB - 1) / CB;
    // from both ends: both parts whole blocks and long enough to have a chase.  BSP_CW_SPLIT = the share of the blocks, in percent,
    // that the LEADING part takes (at most half): its fill is chased towards r = 0, which is what costs accuracy (ORIENTATION).
    c.Nl = (opts().cw_split > 0 && n % CB == 0) ? (int)((long)c.N * (opts().cw_split > 50 ? 50 : opts().cw_split) / 100) : 0;
    c.split = c.Nl >= 3 && c.N - c.Nl >= 4;
    if (!c.split) c.Nl = 0;
    c.Nh = c.split ? c.N - c.Nl : 0;                                       // blocks of the trailing part (the longer one)
    c.Nproc = c.split ? c.Nh : c.N;                                        // blocks of the longest process
    c.Ns = c.split ? c.Nh + 2 : c.N;                                       // block slots of a channel of the batch
    return c;
}
// The factor in chunks (crawford_prepare with events): columns per chunk, a multiple of 128 (whole blocks, whole load chunks of the
// Cholesky kernel); one chunk for the small pencils, whose whole factor takes less than a launch or two of the reduction.
int cw_chunk_cols(int n) { return n < opts().cw_chunk_min ? n + 128 : ((n + CW_CHUNKS - 1) / CW_CHUNKS + 127) / 128 * 128; }
}  // namespace

// The part of the reduction that depends on S alone: the reversed band, its factor, the elimination transforms (the same for every
// channel).  On its own so that the caller can run it on a stream of its own beside the assembly of the H_l, as soon as S is there.
int crawford_prepare(int n, int k, const double *d_SB, const CrawfordWork &w, hipStream_t st, hipEvent_t *evc)
{
    if (!crawford_supported(n, k)) return BSP_ERR_UNSUPPORTED;
    const CwShape c = cw_shape(n);
    const int N = c.N, Nh = c.Nh, Nl = c.Nl;
    int rc;
    BSP_HIP(hipMemsetAsync(w.info, 0, sizeof(int), st));
    hipLaunchKernelGGL(crawford_flip_kernel, dim3((n * k + 255) / 256), dim3(256), 0, st, n, k, d_SB, w.SBf);
    if (c.split) {
        if ((rc = launch_band_cholesky_pair(n, k, CB * Nh, w.SBf, w.UBf, w.rdiagf, w.info, st))) return rc;
        hipLaunchKernelGGL(crawford_setup_kernel, dim3(Nh, 2), dim3(64), 0, st, n, k, N, 0, w.UBf, w.LiB, w.Qel);
        hipLaunchKernelGGL(crawford_cut_setup_kernel, dim3(1), dim3(64), 0, st, n, k, N, Nh, Nl, w.SBf, w.LiB, w.Qel, w.info);
    } else if (evc) {
        // in chunks: the same rows by the same arithmetic (a launch takes the rows before its first from the factor in memory)
        const int cc = cw_chunk_cols(n);
        for (int q = 0, j0 = 0; q < CW_CHUNKS; ++q, j0 += cc) {
            if (j0 < n) {
                const int j1 = j0 + cc < n ? j0 + cc : n, b0 = j0 / CB, b1 = j1 < n ? j1 / CB : N;
                if ((rc = launch_band_cholesky_range(n, k, j0, j1, w.SBf, w.UBf, w.rdiagf, w.info, st))) return rc;
                hipLaunchKernelGGL(crawford_setup_kernel, dim3(b1 - b0, 1), dim3(64), 0, st, n, k, N, b0, w.UBf, w.LiB, w.Qel);
            }
            BSP_HIP(hipEventRecord(evc[q], st));
        }
    } else {
        if ((rc = launch_band_cholesky(n, k, w.SBf, w.UBf, w.rdiagf, w.info, st))) return rc;
        hipLaunchKernelGGL(crawford_setup_kernel, dim3(N, 1), dim3(64), 0, st, n, k, N, 0, w.UBf, w.LiB, w.Qel);
    }
    if (evc && c.split)
        for (int q = 0; q < CW_CHUNKS; ++q) BSP_HIP(hipEventRecord(evc[q], st));
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

int crawford_run(int n, int npad, int k, int nl, const double *d_SB, const double *d_HB, const CrawfordWork &w, double *d_AB,
                 hipStream_t st, bool s_prepared, hipEvent_t *evc, hipStream_t aux0)
{
    if (!crawford_supported(n, k)) return BSP_ERR_UNSUPPORTED;
    const int N = (n + CB - 1) / CB;
    // from both ends: both parts whole blocks and long enough to have a chase.  BSP_CW_SPLIT = the share of the blocks, in percent,
    // that the LEADING part takes (at most half): its fill is chased towards r = 0, which is what costs accuracy (ORIENTATION).
    int Nl = (opts().cw_split > 0 && n % CB == 0) ? (int)((long)N * (opts().cw_split > 50 ? 50 : opts().cw_split) / 100) : 0;
    const bool split = Nl >= 3 && N - Nl >= 4;
    if (!split) Nl = 0;
    const int Nh = split ? N - Nl : 0;                                     // blocks of the trailing part (the longer one)
    const int Nproc = split ? Nh : N;                                      // blocks of the longest process
    const int Ns = split ? Nh + 2 : N;                                     // block slots of a channel of the batch
    const int ny = split ? 2 * nl : nl;                                    // channels of the batch
    const int nlead = split ? Nl : 0x7fffffff;
    int rc;
    KScope kt(KS_CRAWFORD, st);
    if (!s_prepared && (rc = crawford_prepare(n, k, d_SB, w, st))) return rc;
    if (!s_prepared) evc = nullptr;
    // the factor prepared elsewhere, in chunks: chunk 0 before block 0 is touched, the others in the loop below
    const int ccb = cw_chunk_cols(n) / CB;                                 // blocks per chunk
    int waited = 0;
    if (evc) BSP_HIP(hipStreamWaitEvent(st, evc[0], 0));
    hipLaunchKernelGGL(crawford_init_kernel, dim3(Nproc, ny), dim3(64), 0, st, n, k, Ns, nl, N * CBB, d_HB, w.LiB, w.D, w.E);
    BSP_HIP(hipGetLastError());
    const int qstride = N * 256;
    // one wavefront: chase items j = jlo .. jlo + nch - 1 and, if jel > 0, the elimination of step jel, for cy channels
    unsigned long long *dg = nullptr;
    if (opts().cw_diag && opts().cw_items4) {
        BSP_HIP(hipMalloc(reinterpret_cast<void **>(&dg), 8 * sizeof(unsigned long long)));
        BSP_HIP(hipMemsetAsync(dg, 0, 8 * sizeof(unsigned long long), st));
    }
    // c0, cy, ws: the channels c0 .. c0 + cy - 1 of the batch, on stream ws
    auto wavefront = [&](int t, int jlo, int nch, int jel, int cy, int c0 = 0, hipStream_t ws = nullptr) {
        if (!ws) ws = st;
        double *gD = w.D + (size_t)c0 * Ns * CBB, *gE = w.E + (size_t)c0 * Ns * CBB, *gG = w.G + (size_t)c0 * Ns * CBB;
        const int items = nch + (jel ? 1 : 0);
        if (opts().cw_items4) {
            const int nw = opts().cw_nw == 4 ? 4 : 1;                      // waves per workgroup (A/B; a wave never talks to another)
            // four items per wave at every launch size: fewer (BSP_CW_IPW = 1, 2) on the launches that would still fit the chip
            // was measured and is slower (profiles/r04_experiments.txt, 9) -- a wave's RQ loop costs the same for one item as for four
            const int ipw = (opts().cw_ipw == 1 || opts().cw_ipw == 2) ? opts().cw_ipw : 4;
            const int waves = (items + ipw - 1) / ipw;
            const dim3 grid((waves + nw - 1) / nw, cy), block(64 * nw);
            if (opts().cw_onediv) hipLaunchKernelGGL((crawford_item4_kernel<true, 1>), dim3(waves, cy), dim3(64), 0, ws, Ns, t, jlo, nch, jel, ipw, nl, qstride, nlead, w.Qel, gD, gE, gG, dg);
            else if (nw == 4) hipLaunchKernelGGL((crawford_item4_kernel<false, 4>), grid, block, 0, ws, Ns, t, jlo, nch, jel, ipw, nl, qstride, nlead, w.Qel, gD, gE, gG, dg);
            else hipLaunchKernelGGL((crawford_item4_kernel<false, 1>), grid, block, (size_t)opts().cw_ldspad * 1024, ws, Ns, t, jlo, nch, jel, ipw, nl, qstride, nlead, w.Qel, gD, gE, gG, dg);
        } else {
            const dim3 grid((items + 3) / 4, cy);
            if (opts().cw_onediv) hipLaunchKernelGGL(crawford_item_kernel<true>, grid, dim3(256), 0, ws, Ns, t, jlo, nch, jel, nl, qstride, nlead, w.Qel, gD, gE, gG);
            else hipLaunchKernelGGL(crawford_item_kernel<false>, grid, dim3(256), 0, ws, Ns, t, jlo, nch, jel, nl, qstride, nlead, w.Qel, gD, gE, gG);
        }
    };
    // wavefront t: eliminations 2 j - 1 = t (1 <= j <= Nproc - 1), chase items 2 j + s = t with 0 <= s <= j - 2, j <= Nproc - 1
    const int tmax = (Nproc >= 3) ? 3 * Nproc - 5 : (Nproc == 2 ? 1 : 0);
    // STREAMS (BSP_CW_STREAMS): the channels in groups, each group's launches on a stream of its own.  A launch of W waves takes
    // ceil(W / 2048) rounds of the chip (2048 = two waves per SIMD) and a wave lives as long alone as in a full round, so a launch
    // with 2050 .. 4096 waves -- two fifths of them at 128 channels -- leaves most of its second round empty; launches of another
    // group, which depend on nothing in this one, fill it.  The items and their arithmetic are the same: the same bits.
    int ngrp = 1;
    if (!split && !dg && opts().cw_streams > 1) {
        ngrp = opts().cw_streams > 4 ? 4 : opts().cw_streams;
        while (ngrp > 1 && nl / ngrp < 16) --ngrp;
    }
    // aux0: a stream of the caller's for the second group (the one that prepared S, idle by now: a process has few hardware queues, and
    // a stream more than those makes two of them wait for each other)
    static hipStream_t s_aux[3] = {nullptr, nullptr, nullptr};
    hipStream_t aux[3];
    hipEvent_t evf = nullptr, evj[3] = {nullptr, nullptr, nullptr};
    if (ngrp > 1) {
        static std::mutex mx;
        std::lock_guard<std::mutex> lk(mx);
        for (int g = 0; g < ngrp - 1; ++g)
            if (!s_aux[g] && !(g == 0 && aux0)) BSP_HIP(hipStreamCreateWithFlags(&s_aux[g], hipStreamNonBlocking));
        for (int g = 0; g < 3; ++g) aux[g] = (g == 0 && aux0) ? aux0 : s_aux[g];
        BSP_HIP(hipEventCreateWithFlags(&evf, hipEventDisableTiming));
        BSP_HIP(hipEventRecord(evf, st));
        for (int g = 0; g < ngrp - 1; ++g) BSP_HIP(hipStreamWaitEvent(aux[g], evf, 0));
    }
    for (int t = 1; t <= tmax; ++t) {
        const int jel = ((t & 1) && (t + 1) / 2 <= Nproc - 1) ? (t + 1) / 2 : 0;
        const int jlo = (t + 2 + 2) / 3;                                  // ceil((t + 2) / 3)
        const int jhi = (t / 2 < Nproc - 1) ? t / 2 : Nproc - 1;
        const int nch = (jhi >= jlo && jlo >= 2) ? jhi - jlo + 1 : 0;
        if (nch + (jel ? 1 : 0) == 0) continue;
        if (evc && jel) {                                                  // the elimination of block jel reads Qel[jel]
            const int q = jel / ccb < CW_CHUNKS - 1 ? jel / ccb : CW_CHUNKS - 1;
            for (; waited < q; ++waited)
                for (int g = 0; g < ngrp; ++g) BSP_HIP(hipStreamWaitEvent(g == 0 ? st : aux[g - 1], evc[waited + 1], 0));
        }
        for (int g = 0; g < ngrp; ++g) {
            const int c0 = (int)((long)ny * g / ngrp), c1 = (int)((long)ny * (g + 1) / ngrp);
            wavefront(t, jlo, nch, jel, c1 - c0, c0, g == 0 ? st : aux[g - 1]);
        }
    }
    if (ngrp > 1) {
        for (int g = 0; g < ngrp - 1; ++g) {
            BSP_HIP(hipEventCreateWithFlags(&evj[g], hipEventDisableTiming));
            BSP_HIP(hipEventRecord(evj[g], aux[g]));
            BSP_HIP(hipStreamWaitEvent(st, evj[g], 0));
        }
        hipEventDestroy(evf);                                              // released once the work recorded so far has passed them
        for (int g = 0; g < ngrp - 1; ++g) hipEventDestroy(evj[g]);
    }
    if (evc)
        for (; waited < CW_CHUNKS - 1; ++waited) BSP_HIP(hipStreamWaitEvent(st, evc[waited + 1], 0));   // (a pencil of very few blocks)
    if (split) {
        // the step at the cut, on the trailing parts (channels 0 .. nl - 1): elimination at block Nh, its fill chased to block 0
        hipLaunchKernelGGL(crawford_cut_prep_kernel, dim3(nl), dim3(64), 0, st, Ns, Nh, Nl, nl, N, w.LiB, w.D, w.E);
        wavefront(2 * Nh - 1, 0, 0, Nh, nl);
        for (int sft = 0; sft <= Nh - 2; ++sft) wavefront(2 * Nh + sft, Nh, 1, 0, nl);
    }
    BSP_HIP(hipGetLastError());
    if (dg) {
        unsigned long long h[8];
        BSP_HIP(hipStreamSynchronize(st));
        BSP_HIP(hipMemcpy(h, dg, sizeof(h), hipMemcpyDeviceToHost));
        hipFree(dg);
        const double wv = h[7] ? (double)h[7] : 1.0;
        fprintf(stderr, "crawford_item4_kernel, %llu waves, s_memtime ticks per wave: X arrives %.0f | eight reflectors %.0f | R, Q stored %.0f | "
                        "phase B operands arrive %.0f | congruences issued %.0f | stores leave %.0f | wave %.0f\n", h[7], h[0] / wv, h[1] / wv,
                h[2] / wv, h[3] / wv, h[4] / wv, h[5] / wv, h[6] / wv);
    }
    const bool band8 = opts().cw_band8 != 0;
    if (band8) hipLaunchKernelGGL(crawford_corner_kernel, dim3(ny), dim3(64), 0, st, Ns, w.D, w.E);
    hipLaunchKernelGGL(crawford_band_kernel, dim3((npad * 32 + 255) / 256, nl), dim3(256), 0, st, n, npad, Ns, band8 ? CB : 2 * CB - 1,
                       Nh, N, nl, w.D, w.E, d_AB);
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
