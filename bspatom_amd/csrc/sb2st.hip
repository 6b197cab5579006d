// sb2st.hip -- stage 2 of the two-stage tridiagonalisation: band (half-width b = 64) -> tridiagonal
// by Householder bulge chasing.  Replaces the second half of LAPACK DSYTRD (reference call
// matrices.f90:248 -> DSYGV).
//
// Band storage (lower, LD = 2b rows so that the bulge fits): AB[d + j*LD] = A(j+d, j), d < 2b.
//
// Sweep s annihilates column s below the sub-diagonal (reflector of length L <= b acting on rows
// r0 = s+1 .. s+L, two-sided on the diagonal block), then chases the bulge down the band: each
// chase step ("item") right-applies the current reflector to the L2 x L block B below the diagonal block
// (fill-in), annihilates B's first column with a new reflector, left-applies it to the rest of
// B, and applies it two-sided to the next diagonal block D2.
//
// Kernel generations kept here (BSP_SB2ST_VERSION; history and measurements in DESIGN.md 4.1):
//   3  one workgroup per channel, register-blocked tiles, tiles through HBM every item
//   6  two workgroups (two CUs of one XCD) per channel, sweep s+1 following sweep s through L2
//   7  one 512-thread workgroup per channel running TWO sweeps per pass with on-chip forwarding
//   8  (default) 7 paired as in 6: four sweeps in flight per channel
#include <vector>
#include "common.h"

namespace bsp {

constexpr int SB = 64;             // band half-width handled by this kernel
constexpr int TLD = SB + 1;        // LDS tile row stride (bank-conflict padding)

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

// Wave-wide sum, result uniform in every lane: DPP row scan (row_shr 1,2,4,8 -> lane 15 of each
// 16-lane row holds the row total) + four readlanes.  ~10x shorter dependent chain than a
// ds_bpermute butterfly, which matters because the reflector construction is serial.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ double read_lane(double x, int l)
{
    union { double d; int i[2]; } u, r;
    u.d = x;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return r.d;
}
__device__ __forceinline__ double wave_sum(double x)
{
    x += dpp_mov<0x111>(x);      // row_shr:1
    x += dpp_mov<0x112>(x);      // row_shr:2
    x += dpp_mov<0x114>(x);      // row_shr:4
    x += dpp_mov<0x118>(x);      // row_shr:8
    return (read_lane(x, 15) + read_lane(x, 31)) + (read_lane(x, 47) + read_lane(x, 63));
}

// lane-parallel dlarfg on one wavefront: xi = element i of x (0 beyond L).  Returns v_i.
__device__ __forceinline__ double wave_house(double xi, int lane, int L, double *beta, double *tau)
{
    const double sq = wave_sum((lane >= 1 && lane < L) ? xi * xi : 0.0);
    const double alpha = read_lane(xi, 0);
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

// ------------------------------------------------------------------------------------------------
// v3: register-blocked tiles.  Thread (bi, bj) of the 16 x 16 thread grid owns the 4 x 4 sub-blocks
// rows 4bi.., cols 4bj.. of the current B and D2 tiles AND of the prefetched next ones, all in
// registers; LDS carries only the vectors (vc, vn, w, z, p) and the 16-way partial sums of the
// matrix-vector products, which the finalising wavefront adds up.  4 LDS-only barriers per step:
//   P1 all    : issue the loads of the next step's tiles; partial w0 = B vc          -> R1[bj][i]
//   P2 wave 0 : w = tau sum R1 ; x' = B(:,0) - w vc_0 ; (vn, tau2, beta2) ; s = vn^T w
//   P3 all    : partial z0 = vn^T B -> R2[bi][j] ; partial p0 = D2 vn -> R3[bj][i] (row part),
//               R4[bi][j] (column part of the strictly-lower sub-blocks)
//   P4 wave 0 : z = tau2 (sum R2 - s vc)        wave 1 : p = tau2 (sum R3 + sum R4) + alpha vn
//   P5 all    : B, D2 updated in registers and stored; next tiles become current (no copy: the
//               step function is instantiated twice with the register sets swapped).
constexpr int RLD = SB + 4;     // row stride of the partial-sum arrays (bank spread)

struct ChaseState { int r0, L, L2, L3; double tau; };

// thread (ib, jb): rows ib + 16 ri (ri < 4), columns 4 jb + cj (cj < 4).  For a fixed (ri, cj) the 16
// lanes of a DPP row cover 16 consecutive rows of one column = one full 128-B line, so a wave-level
// load/store touches 4 whole lines (the 4x4 contiguous sub-block variant touched 16 partial ones and
// was bound by the CU's address unit).
#define SB3_ROW(ri) (ib + 16 * (ri))
#define SB3_COL(cj) (j0 + (cj))
// Element (i, j) of a tile whose top-left element sits at AB[base] lives at base + j*(LD-1) + i
// (B: base = r0*LD + L, D2: base = rn*LD): per-thread constants off[ri][cj], a wave-uniform base per
// step.  `low` bit (ri*4+cj) marks i >= j (D2 is kept as a lower triangle).  Branch-free: masked lanes
// read element 0 of the tile and discard it.
__device__ __forceinline__ void load_tiles(const double *__restrict__ AB, int r0, int L, int L2, int ib, int j0,
                                           const unsigned (&off)[4][4], unsigned low,
                                           double (&Bq)[4][4], double (&Dq)[4][4])
{
    constexpr int LD = 2 * SB;
    const double *__restrict__ Bb = AB + ((size_t)r0 * LD + L);
    const double *__restrict__ Db = AB + (size_t)(r0 + L) * LD;
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            const int i = SB3_ROW(ri), j = SB3_COL(cj);
            const bool okb = (i < L2 && j < L);
            const bool okd = ((low >> (ri * 4 + cj)) & 1u) && (i < L2);
            const double bv = Bb[okb ? off[ri][cj] : 0u];
            const double dv = Db[okd ? off[ri][cj] : 0u];
            Bq[ri][cj] = okb ? bv : 0.0;
            Dq[ri][cj] = okd ? dv : 0.0;
        }
}

struct Sb3Lds {
    double va[SB], vb[SB], w[SB], z[SB], pv[SB], x0[SB];
    double R1[16][RLD], R2[16][RLD], R3[16][RLD], R4[16][RLD];
    double sc[8];
};

// diagnostic cycle stamps (DIAG instantiation only; never part of a timed or shipped run)
#define SB3_STAMP(k)                                                             \
    if (DIAG == 1) {                                                             \
        const long long tnow_ = (long long)__builtin_amdgcn_s_memtime();         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
        acc[k] += tnow_ - tlast;                                                 \
        tlast = tnow_;                                                           \
    }

// one chase step; (Bc, Dc) current tiles, (Bn, Dn) receive the prefetch of the next step
template <int DIAG>
__device__ __forceinline__ void chase_step(double *__restrict__ AB, Sb3Lds &S, const double *vc, double *vn,
                                           ChaseState &st, int n, int tid, int ib, int j0,
                                           const unsigned (&off)[4][4], unsigned low,
                                           double (&Bc)[4][4], double (&Dc)[4][4], double (&Bn)[4][4], double (&Dn)[4][4],
                                           long long (&acc)[12], long long &tlast)
{
    constexpr int LD = 2 * SB;
    const int jb = j0 >> 2;
    const int r0 = st.r0, L = st.L, L2 = st.L2, rn = r0 + L;
    const bool have_next = (rn + L2 < n);
    const int L3 = have_next ? ((n - (rn + L2) < SB) ? (n - (rn + L2)) : SB) : 0;
    // ---- P1 ----
    SB3_STAMP(0)
    // Prefetch of the next step's tiles: ONE branch-free masked loader.  Any control flow around these
    // loads (even wave-uniform) makes hipcc drain vmcnt(0) where the register results join, which
    // serialises the whole prefetch (measured 2.2x slower).  Without a next step the bases are clamped to
    // AB and every lane is masked off.
    {
        const double *__restrict__ Bb = have_next ? (AB + ((size_t)rn * LD + L2)) : AB;
        const double *__restrict__ Db = have_next ? (AB + (size_t)(rn + L2) * LD) : AB;
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int i = SB3_ROW(ri), j = SB3_COL(cj);
                const bool okb = (i < L3) && (j < L2);
                const bool okd = ((low >> (ri * 4 + cj)) & 1u) && (i < L3);
                if (DIAG == 3) { Bn[ri][cj] = 1e-3 * (i + j); Dn[ri][cj] = (i == j) ? 2.0 : 1e-3; continue; }   // timing-only
                const double bv = Bb[okb ? off[ri][cj] : 0u];
                const double dv = Db[okd ? off[ri][cj] : 0u];
                Bn[ri][cj] = okb ? bv : 0.0;
                Dn[ri][cj] = okd ? dv : 0.0;
            }
    }
    SB3_STAMP(1)
    {
        double vcj[4];
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) vcj[cj] = vc[j0 + cj];
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            double a = 0.0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) a += Bc[ri][cj] * vcj[cj];
            S.R1[jb][SB3_ROW(ri)] = a;
        }
        if (jb == 0) {
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) S.x0[SB3_ROW(ri)] = Bc[ri][0];
        }
    }
    SB3_STAMP(2)
    lds_barrier();
    SB3_STAMP(3)
    // ---- P2: wave 0 ----
    if (tid < 64) {
        const int i = tid;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R1[q][i];
        const double wi = st.tau * a;
        const double xi = (i < L2) ? (S.x0[i] - wi * vc[0]) : 0.0;
        double beta2, tau2;
        const double vi = wave_house(xi, i, L2, &beta2, &tau2);
        const double sdot = wave_sum(vi * wi);
        S.w[i] = wi; vn[i] = vi;
        if (i == 0) { S.sc[1] = beta2; S.sc[2] = tau2; S.sc[3] = sdot; }
    }
    SB3_STAMP(4)
    lds_barrier();
    SB3_STAMP(5)
    const double beta2 = S.sc[1], tau2 = S.sc[2], sdot = S.sc[3];
    // ---- P3 ----
    double vni[4], vnj[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { vni[x] = vn[SB3_ROW(x)]; vnj[x] = vn[j0 + x]; }
    {
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            double a = 0.0;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) a += vni[ri] * Bc[ri][cj];
            S.R2[ib][j0 + cj] = a;
        }
        // D2 vn: sub-blocks with bi >= bj give the row part; strictly lower ones also the column part
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            double a = 0.0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) a += Dc[ri][cj] * vnj[cj];          // Dc is zero above the diagonal
            S.R3[jb][SB3_ROW(ri)] = a;
        }
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            double a = 0.0;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) a += (SB3_ROW(ri) != j0 + cj) ? Dc[ri][cj] * vni[ri] : 0.0;   // strictly lower
            S.R4[ib][j0 + cj] = a;
        }
    }
    SB3_STAMP(6)
    lds_barrier();
    SB3_STAMP(7)
    // ---- P4 ----
    if (tid < 64) {
        const int j = tid;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R2[q][j];
        S.z[j] = tau2 * (a - sdot * vc[j]);
    } else if (tid < 128) {
        const int i = tid - 64;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R3[q][i] + S.R4[q][i];
        const double pi = tau2 * a;
        const double dot = wave_sum(pi * vn[i]);
        S.pv[i] = pi + (-0.5 * tau2 * dot) * vn[i];
    }
    SB3_STAMP(8)
    lds_barrier();
    SB3_STAMP(9)
    // ---- P5 ----
    {
        double wi[4], zj[4], vcj[4], pi[4], pj[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            wi[x] = S.w[SB3_ROW(x)]; zj[x] = S.z[j0 + x]; vcj[x] = vc[j0 + x]; pi[x] = S.pv[SB3_ROW(x)]; pj[x] = S.pv[j0 + x];
        }
        double *__restrict__ Bb = AB + ((size_t)r0 * LD + L);
        double *__restrict__ Db = AB + (size_t)rn * LD;
        if (DIAG >= 2) {                                  // timing-only: results kept live, nothing stored
            double keep = 0.0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri)
                    keep += (Bc[ri][cj] - (wi[ri] * vcj[cj] + vni[ri] * zj[cj])) + (Dc[ri][cj] - (vni[ri] * pj[cj] + pi[ri] * vnj[cj]));
            if (keep == 1.2345e300) Bb[0] = keep;
        } else if (L == SB && L2 == SB) {
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    double bnew = Bc[ri][cj] - (wi[ri] * vcj[cj] + vni[ri] * zj[cj]);
                    if (cj == 0 && j0 == 0) bnew = (SB3_ROW(ri) == 0) ? beta2 : 0.0;
                    Bb[off[ri][cj]] = bnew;
                    if ((low >> (ri * 4 + cj)) & 1u)
                        Db[off[ri][cj]] = Dc[ri][cj] - (vni[ri] * pj[cj] + pi[ri] * vnj[cj]);
                }
        } else {
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int i = SB3_ROW(ri), j = j0 + cj;
                    double bnew = Bc[ri][cj] - (wi[ri] * vcj[cj] + vni[ri] * zj[cj]);
                    if (j == 0) bnew = (i == 0) ? beta2 : 0.0;
                    if (i < L2 && j < L) Bb[off[ri][cj]] = bnew;
                    if (((low >> (ri * 4 + cj)) & 1u) && i < L2)
                        Db[off[ri][cj]] = Dc[ri][cj] - (vni[ri] * pj[cj] + pi[ri] * vnj[cj]);
                }
        }
    }
    SB3_STAMP(10)
    st.r0 = rn; st.L = L2; st.L2 = L3; st.tau = tau2;
}

template <int DIAG>
__device__ __forceinline__ void sb2st_v3_body(int n, int npad, double *ABall, double *dall, double *eall,
                                              long long *diag, size_t ch)
{
    long long acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = (DIAG == 1) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    __shared__ Sb3Lds S;
    constexpr int LD = 2 * SB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ib = lane & 15, jb = (lane >> 4) + 4 * wave;
    const int j0 = 4 * jb;
    unsigned off[4][4], low = 0;
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            const int i = SB3_ROW(ri), j = j0 + cj;
            off[ri][cj] = (unsigned)(j * (LD - 1) + i);
            if (i >= j) low |= 1u << (ri * 4 + cj);
        }
    double *AB = ABall + ch * ab_stride(npad);
    double B0[4][4], D0[4][4], B1[4][4], D1[4][4];

    for (int s = 0; s < n - 2; ++s) {
        int L = (n - 1 - s < SB) ? (n - 1 - s) : SB;
        if (L < 2) break;
        const int r0 = s + 1;
        __syncthreads();          // HBM stores of the previous sweep are visible to every wave
        // ---- sweep start: D = A[r0:r0+L, r0:r0+L] (as a "D2" at rn = r0 with an empty B) ----
        {
            // load_tiles with (r0' = r0 - 0, L' = 0): B part masked out (j < 0 never), D2 at rn = r0
            load_tiles(AB, r0, 0, L, ib, j0, off, low, B0, D0);
        }
        if (tid < 64) {
            const double xraw = AB[(tid < L) ? ((size_t)s * LD + 1 + tid) : 0];
            const double xi = (tid < L) ? xraw : 0.0;
            double beta, tau;
            const double vi = wave_house(xi, tid, L, &beta, &tau);
            S.va[tid] = vi;
            if (tid < L) AB[(size_t)s * LD + 1 + tid] = (tid == 0) ? beta : 0.0;
            if (tid == 0) S.sc[0] = tau;
        }
        lds_barrier();
        const double tau0 = S.sc[0];
        double vni[4], vnj[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) { vni[x] = S.va[SB3_ROW(x)]; vnj[x] = S.va[j0 + x]; }
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            double a = 0.0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) a += D0[ri][cj] * vnj[cj];
            S.R3[jb][SB3_ROW(ri)] = a;
        }
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            double a = 0.0;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) a += (SB3_ROW(ri) != j0 + cj) ? D0[ri][cj] * vni[ri] : 0.0;
            S.R4[ib][j0 + cj] = a;
        }
        lds_barrier();
        if (tid < 64) {
            const int i = tid;
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) a += S.R3[q][i] + S.R4[q][i];
            const double pi = tau0 * a;
            const double dot = wave_sum(pi * S.va[i]);
            S.pv[i] = pi + (-0.5 * tau0 * dot) * S.va[i];
        }
        lds_barrier();
        {
            double pi[4], pj[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) { pi[x] = S.pv[SB3_ROW(x)]; pj[x] = S.pv[j0 + x]; }
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int i = SB3_ROW(ri);
                    if (((low >> (ri * 4 + cj)) & 1u) && i < L)
                        AB[(size_t)r0 * LD + off[ri][cj]] = D0[ri][cj] - (vni[ri] * pj[cj] + pi[ri] * vnj[cj]);
                }
        }
        // ---- chase ----
        ChaseState st;
        st.r0 = r0; st.L = L; st.tau = tau0;
        st.L2 = (r0 + L < n) ? ((n - (r0 + L) < SB) ? (n - (r0 + L)) : SB) : 0;
        if (st.L2 > 0) load_tiles(AB, r0, L, st.L2, ib, j0, off, low, B0, D0);
        double *vc = S.va, *vn = S.vb;
        if (DIAG == 1) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); acc[11] += t_ - tlast; tlast = t_; }
        while (st.L2 > 0) {
            chase_step<DIAG>(AB, S, vc, vn, st, n, tid, ib, j0, off, low, B0, D0, B1, D1, acc, tlast);
            if (st.L2 <= 0) break;
            chase_step<DIAG>(AB, S, vn, vc, st, n, tid, ib, j0, off, low, B1, D1, B0, D0, acc, tlast);
        }
    }
    if (DIAG == 1 && (tid & 63) == 0 && blockIdx.x == 0) {
        for (int q = 0; q < 12; ++q) diag[(tid >> 6) * 12 + q] = acc[q];
    }
    __syncthreads();
    double *d = dall + ch * (size_t)npad, *e = eall + ch * (size_t)npad;
    for (int j = tid; j < n; j += 256) {
        d[j] = AB[(size_t)j * LD];
        e[j] = (j < n - 1) ? AB[(size_t)j * LD + 1] : 0.0;
    }
}


template <int DIAG>
__global__ __launch_bounds__(256) void sb2st_kernel_v3(int n, int npad, double *ABall, double *dall, double *eall,
                                                      long long *diag)
{
    sb2st_v3_body<DIAG>(n, npad, ABall, dall, eall, diag, blockIdx.x);
}

// what a half does in one super-step (v6, v7)
enum { ACT_IDLE = 0, ACT_PRELOAD = 1, ACT_ITEM0 = 2, ACT_CHASE = 3 };

// ------------------------------------------------------------------------------------------------
// Helpers of the cross-workgroup hand-off (rings of v7 workgroups, "v8" below).  The first paired generation (v6: two
// one-sweep workgroups per channel, 440 ms) is in the history only; its progress-word format survives.
constexpr unsigned long long SB6_FIN = 1ull << 60;

__device__ __forceinline__ double ld_sc1(const double *p)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}
__device__ __forceinline__ bool sb6_dep_ok(unsigned long long pw, int sw, int need)
{
    if (sw == 0 || pw >= SB6_FIN) return true;
    const int osw = (int)(pw >> 20), odn = (int)(pw & 0xfffff);
    return (osw > sw - 1) || (osw == sw - 1 && odn >= need);
}

// ------------------------------------------------------------------------------------------------
// v7: TWO SWEEPS PER PASS WITH ON-CHIP FORWARDING (halves the HBM traffic of bulge chasing).
// One 512-thread workgroup per channel.  Half A (waves 0-3) runs sweep s = 0, 2, 4, ...: it loads its
// tiles from HBM (prefetched, as v3) but writes its UPDATED tiles to LDS exchange slots instead of HBM.
// Half B (waves 4-7) runs sweep s+1 two items behind: its item-k tiles are A's updated tiles of items
// k and k+1 shifted by one row and one column (checked bit-for-bit in tools/proto_forward.py), which it
// assembles from the slots; only B stores to HBM.  Per item pair 48 KB are loaded and 48 KB stored
// instead of 96 + 96.  A's only HBM stores are the finished column s and the diagonal entry (s+1,s+1).
// Both halves execute exactly five barriers per super-step in two separate straight-line loops.
constexpr int SB7_MAX_STALL = 4096;  // super-steps B may wait for A before the kernel gives up (a sweep has <= n/64 + 3)
constexpr int SB7_LAG_HBM = 6;      // A's next sweep may read what B stored this many items earlier

// per-half scratch: as Sb3Lds, but R1 (written in P1, read in P2) shares its storage with R2 (written in
// P3, read in P4) -- barriers 2 and 4/0 separate the two lifetimes -- so that both halves plus the
// exchange slots fit the CU's 160 KB of LDS
struct Sb7Lds {
    double va[SB], vb[SB], w[SB], z[SB], pv[SB], x0[SB];
    union { double R1[16][RLD]; double R2[16][RLD]; };
    double R3[16][RLD], R4[16][RLD];
    double sc[8];
};
// Exchange frames.  Sweep s+1's item-k tiles B'_k, D'_k are A's item-k tiles shifted by one row and one
// column, with the first column of A's D_k as the last column of B'_k, and the first row of A's item k+1
// tiles as their LAST row.  A stores its updated tiles UNSHIFTED, B reads them at (i+1, j+1):
//   FB[f][j][i]   = B_k(i, j)            (j < SB), written by every thread, zero outside the tile
//   FB[f][SB][i]  = D_k(i, 0)            the column that becomes B'_k's last one (and sweep s+1's x for k = 0)
//   FB[f][SB][L2] = B_{k+1}(0, 0)        written by A's item k+1 (its L equals this item's L2)
//   FD<f>(i, j)   = D_k(i, j), i >= j;   FD<f>(L2, j) = B_{k+1}(0, j) (j >= 1),  FD<f>(L2, L2) = D_{k+1}(0, 0)
// The frame index f is the PARITY OF THE SUPER-STEP in which A runs the item -- a compile-time constant in
// the unrolled pair, so every LDS address is a lane base plus an immediate.  B runs item k exactly two
// super-steps after A (same parity; checked through the tags).
struct Sb7Shared {
    Sb7Lds S[2];
    double FB[2][SB + 1][TLD];
    double FD[SB + 2][TLD];         // lower triangles incl. row SB of both frames: frame 0 at FD[j][i], frame 1 at FD[i+1][j]
    int tag[2][2];                  // per frame: sweep and item of the A item that wrote its main part
    int sweep[2], done[2], fin[2], viol, abort;
    int astate, mode;               // A's state after its last super-step (B evaluates A's cross-CU dependency too)
    unsigned long long pw;          // partner workgroup's published progress, as last polled
};

// Two workgroups (two CUs of one XCD) per channel, as in v6: this workgroup runs sweeps first, first + 4, ...
// (half A) and first + 1, first + 5, ... (half B); the partner runs the two sweeps in between.  Half A's
// tiles then come from what the PARTNER's half B stored; `prog` carries B's progress across.
//
// THE HAND-OFF BETWEEN RING MEMBERS, against the valid forms of MI355X_MICROARCH.md ("Workgroup dispatch, XCD placement &
// inter-workgroup visibility").  Data: band tiles in HBM-backed memory; producer = half B of member w-1, consumer =
// half A of member w; flag = prog[w-1] (sweep << 20 | items finished).
//   Placement is VERIFIED, not assumed: every member reports HW_REG_XCC_ID in the handshake word; unless all P ids are
//   equal the ring is dissolved and member 0 runs the channel alone (mode 2 below; also on a handshake time-out).
//   So producer and consumer always share one XCD, i.e. ONE L2, which is the coherence point of their CUs.
//   Producer  (guide: plain stores -> every storing wave's vmcnt(0) -> barrier -> [agent release] -> relaxed agent flag):
//     plain global stores of the tiles (the vector L1 is write-through: the bytes are in the XCD's L2 once the store
//     has been acknowledged) -> `s_waitcnt vmcnt(0)` in EVERY wave of half B one super-step later, at the top of its
//     next P5 where the wait is free (or in the idle path) -> barrier 0 of the following super-step -> lane 0: `__hip_atomic_store(relaxed, agent)` of the progress word
//     (global_store ... sc1).  The agent release (`buffer_wbl2 sc1`: write the L2's dirty lines back to memory) of
//     the guide's form is what makes bytes visible to ANOTHER XCD's L2; within one XCD the consumer reads the same
//     L2 the stores went to, so it is omitted -- at 1.7-6.5 us per fence it would cost as much as the super-step
//     (5.9 us) it would be issued in.  This is the one deliberate departure from the copyable form, and the reason
//     the XCD check above is mandatory rather than a speed hint.
//   Consumer  (guide: ONE relaxed poll -> ONE agent acquire -> vmcnt(0) -> barrier -> plain loads):
//     lane 0 polls with `__hip_atomic_load(relaxed, agent)` (global_load sc1: served by the L2, never by the L1); the
//     value goes through LDS (SH.pw) and is acted on only after the next workgroup barrier.  The acquire
//     (`buffer_inv sc1` + `s_waitcnt vmcnt(0)`, every wave of half A) is issued ONCE PER SWEEP, before the sweep's
//     first tile load (ACT_PRELOAD), not once per poll.  Why that suffices: a line of this CU's L1 can be stale only
//     if it was filled before the producer's store.  (i) Lines filled during this CU's previous pass over the band
//     are dropped by the per-sweep invalidate.  (ii) Within the sweep, the load of item j is issued only when the
//     published progress is >= j + margin (margin = 3), i.e. after the producer's stores of items <= j+2 were
//     acknowledged by the L2; a tile's 128-byte lines reach at most 15 rows into item j+1's tile, never further, so no
//     line is filled ahead of its producer.  Tile loads of half A are plain loads for speed (sc1 loads measured
//     +18 %: tile rows straddle lines); the final d, e read-out of member 0 uses sc1 loads (ld_sc1) because it
//     follows no invalidate.
//   What guards it: bit-identical spectra across repeated 128-channel solves and across ring sizes 2/4/8, one
//   workgroup per channel, and both fallbacks (tests/test_gpu_solve.py::test_sb2st_fallback_paths,
//   ::test_full_size_batch_is_deterministic); a stale line shows up there as a difference.
//
// control block of one channel for v7/v8: a RING of P <= SB8_MAXP workgroups (P CUs of one XCD).  Member w runs the
// sweeps 2w + 2P t (half A) and 2w + 1 + 2P t (half B); half A's tiles come from what member w-1's half B stored.
constexpr int SB8_MAXP = 8;
struct Sb8Ctl {
    unsigned long long hs;                 // handshake: byte w = 0x10 | XCC id of member w; bit 62 COMMIT, bit 63 ABORT
    int err, pad;
    unsigned long long prog[SB8_MAXP];     // published half-B progress of member w: (sweep << 20) | items ; SB6_FIN at the end
    unsigned long long nwait[2], wcycles[2];
};
constexpr unsigned long long SB8_COMMIT = 1ull << 62, SB8_ABORT = 1ull << 63;

struct Pair7 {
    int paired, stride;
    const unsigned long long *pollp;   // partner's published B progress
    unsigned long long *pubp;          // this workgroup's
    Sb8Ctl *C;
    int *status;
    int margin, hyst, lead;            // partner's published lead demanded before an item / extra once a wait began / extra at a sweep's start
};
__device__ __forceinline__ unsigned long long sb7_rfl64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// A super-step in which a half has nothing to run: the four remaining barriers and nothing else -- no
// side effects on the reflector buffers, the register tile sets or the exchange frames, which stay live
// across it (half A, waiting for the partner workgroup, "holds" for an EVEN number of super-steps so that
// the parity-indexed buffers line up again).
__device__ __forceinline__ void sb7_idle_barriers()
{
    lds_barrier(); lds_barrier(); lds_barrier(); lds_barrier();
}
static_assert(sizeof(Sb7Shared) <= 160 * 1024, "both halves' scratch and the exchange frames must fit the CU's LDS");
static_assert(TLD >= SB + 1, "row SB of the frames");

template <int F>
__device__ __forceinline__ double &fd_ref(Sb7Shared &SH, int i, int j)
{
    return F == 0 ? SH.FD[j][i] : SH.FD[i + 1][j];
}

// element offset of tile entry (ib + 16 ri, j0 + cj) from the tile's base in band storage (ld 2*SB - 1 per column)
#define SB7_OFF(ri, cj) (o0 + (unsigned)((cj) * (2 * SB - 1) + 16 * (ri)))

// diagnostic cycle stamps of v7 (DIAG instantiation only; never part of a timed or shipped run)
struct Diag7 { long long acc[12]; long long tlast; };
#define SB7_STAMP(k)                                                             \
    asm volatile("; MARK stamp=%0 par=%1 fast=%2" ::"n"(k), "n"(PAR), "n"(SB7_FASTV)); \
    if (DIAG) {                                                                  \
        const long long tnow_ = (long long)__builtin_amdgcn_s_memtime();         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
        dg.acc[k] += tnow_ - dg.tlast;                                           \
        dg.tlast = tnow_;                                                        \
    }

// ---- half A: one super-step after barrier 0.  FAST: an interior chase item (64 x 64 tiles, a full successor,
// a partner sweep) -- every range mask is true and folds away; the general instantiation handles the rest.
#define SB7_FASTV FAST
template <int PAR, int FAST, int DIAG>
__device__ __forceinline__ void body_v7A(double *__restrict__ AB, Sb7Shared &SH, int htid_, int ib_, int j0_,
                                              unsigned o0_, unsigned low_, int n,
                                              int &state, int &sw, int &done, ChaseState &st, Diag7 &dg, double &xpre,
                                              double (&Bc)[4][4], double (&Dc)[4][4], double (&Bn)[4][4], double (&Dn)[4][4], int act,
                                              const Pair7 &pc)
{
    constexpr int LD = 2 * SB;
    Sb7Lds &S = SH.S[0];
    const double *vc = PAR ? S.vb : S.va;
    double *vn = PAR ? S.va : S.vb;
    // the lane constants are re-derived per super-step: kept live across the whole loop (with everything the
    // compiler pre-computes from them: masks, row indices, LDS addresses) they do not fit the 256 registers
    int htid = htid_, ib = ib_, j0 = j0_;
    unsigned o0 = o0_, low = low_;
    asm volatile("" : "+v"(htid), "+v"(ib), "+v"(j0), "+v"(o0), "+v"(low));
    const int jb = j0 >> 2;
    const bool comp = FAST ? true : (act >= ACT_ITEM0);
    const bool item0 = FAST ? false : (act == ACT_ITEM0);
    const int L0 = (n - 1 - sw < SB) ? (n - 1 - sw) : SB;
    if (item0) { st.r0 = sw + 1; st.L = 0; st.L2 = L0; st.tau = 0.0; }
    // wave-uniform by construction; telling the compiler so keeps the tile bases in SGPRs (scalar base + lane offset
    // + immediate addressing for all 32 loads)
    const int r0 = __builtin_amdgcn_readfirstlane(comp ? st.r0 : 0);
    const int L = FAST ? SB : __builtin_amdgcn_readfirstlane(comp ? st.L : 0);
    const int L2 = FAST ? SB : __builtin_amdgcn_readfirstlane(comp ? st.L2 : 0), rn = r0 + L;
    int pr0, pL, pL2;
    bool have_pf;
    if (FAST) { pr0 = rn; pL = SB; pL2 = SB; have_pf = true; }
    else if (act == ACT_PRELOAD) { pr0 = sw + 1; pL = 0; pL2 = L0; have_pf = true; }
    else {
        have_pf = comp && (rn + L2 < n);
        pr0 = rn; pL = L2; pL2 = have_pf ? ((n - (rn + L2) < SB) ? (n - (rn + L2)) : SB) : 0;
    }
    // ---- P1: prefetch from HBM (branch-free), partial w ----
    unsigned long long pollv = 0;
    if (pc.paired) {
        // L1 coherence across the two CUs (as v6): a line this CU cached during its previous sweep has since been
        // rewritten by the partner; one completed L1 invalidate before the sweep's first load removes them all
        if (!FAST && act == ACT_PRELOAD) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        pollv = __hip_atomic_load(pc.pollp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed at the end of the step
    }
    {
        const double *__restrict__ Bb = have_pf ? (AB + ((size_t)pr0 * LD + pL)) : AB;
        const double *__restrict__ Db = have_pf ? (AB + (size_t)(pr0 + pL) * LD) : AB;
        if (FAST) {
            // one lane pointer per tile + compile-time offsets: no address arithmetic per load.  Every address is
            // valid (an upper-triangle D entry reads the tail of the previous column and is masked afterwards).
            // Only the B tile here; the D tile follows in P3.  A wave that issues all 32 loads at once sits in P1 until
            // the CU's memory pipeline has taken them (measured: 30 % of the super-step), two half bursts overlap with
            // the reductions in between.
            const double *__restrict__ Bl = Bb + o0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) Bn[ri][cj] = Bl[cj * (LD - 1) + 16 * ri];
        } else {
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int i = SB3_ROW(ri), j = SB3_COL(cj);
                    const bool okb = have_pf && (i < pL2) && (j < pL);
                    const bool okd = have_pf && ((low >> (ri * 4 + cj)) & 1u) && (i < pL2);
                    const double bv = Bb[okb ? SB7_OFF(ri, cj) : 0u];
                    const double dv = Db[okd ? SB7_OFF(ri, cj) : 0u];
                    Bn[ri][cj] = okb ? bv : 0.0;
                    Dn[ri][cj] = okd ? dv : 0.0;
                }
        }
        if (!FAST) {                                          // every wave loads (a branch around a load drains vmcnt)
            const bool okx = (act == ACT_PRELOAD) && (htid < L0);
            const double xr = AB[okx ? ((size_t)sw * LD + 1 + htid) : 0];
            xpre = (act == ACT_PRELOAD) ? (okx ? xr : 0.0) : xpre;
        }
        double vcj[4];
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) vcj[cj] = vc[j0 + cj];
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            double a = 0.0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) a += Bc[ri][cj] * vcj[cj];
            S.R1[jb][SB3_ROW(ri)] = a;
        }
        if (jb == 0) {
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) S.x0[SB3_ROW(ri)] = Bc[ri][0];
        }
    }
    SB7_STAMP(1);
    lds_barrier();
    SB7_STAMP(2);
    // ---- P2 ----
    if (htid < 64) {
        const int i = htid;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R1[q][i];
        const double wi = item0 ? 0.0 : st.tau * a;
        const double xc = item0 ? xpre : (S.x0[i] - wi * vc[0]);
        const double xi = (FAST || i < L2) ? xc : 0.0;
        double beta2, tau2;
        const double vi = wave_house(xi, i, L2, &beta2, &tau2);
        const double sdot = wave_sum(vi * wi);
        S.w[i] = wi; vn[i] = vi;
        if (i == 0) { S.sc[1] = beta2; S.sc[2] = tau2; S.sc[3] = sdot; }
    }
    SB7_STAMP(3);
    lds_barrier();
    SB7_STAMP(4);
    const double beta2 = S.sc[1], tau2 = S.sc[2], sdot = S.sc[3];
    asm volatile("" : "+v"(htid), "+v"(ib), "+v"(j0), "+v"(o0), "+v"(low));   // re-derive per phase: see the top of the body
    // ---- P3 ----
    double vni[4], vnj[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { vni[x] = vn[SB3_ROW(x)]; vnj[x] = vn[j0 + x]; }
    {
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            double a = 0.0;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) a += vni[ri] * Bc[ri][cj];
            S.R2[ib][j0 + cj] = a;
        }
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            double a = 0.0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) a += Dc[ri][cj] * vnj[cj];
            S.R3[jb][SB3_ROW(ri)] = a;
        }
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            double a = 0.0;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) a += (SB3_ROW(ri) != j0 + cj) ? Dc[ri][cj] * vni[ri] : 0.0;
            S.R4[ib][j0 + cj] = a;
        }
        if (FAST) {
            // second half of the prefetch: the D tile.  Row blocks entirely above this wave's 16 columns (ri < wave) hold
            // no lower-triangle entry: not loaded; the diagonal block is masked at the end of P5.
            const double *__restrict__ Dl = AB + (size_t)(pr0 + pL) * LD + o0;
            const int hw = __builtin_amdgcn_readfirstlane(htid >> 6);
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                if (ri >= hw) {
#pragma unroll
                    for (int cj = 0; cj < 4; ++cj) Dn[ri][cj] = Dl[cj * (LD - 1) + 16 * ri];
                } else {
#pragma unroll
                    for (int cj = 0; cj < 4; ++cj) Dn[ri][cj] = 0.0;
                }
            }
        }
    }
    SB7_STAMP(5);
    lds_barrier();
    SB7_STAMP(6);
    // ---- P4 ----
    if (htid < 64) {
        const int j = htid;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R2[q][j];
        S.z[j] = tau2 * (a - sdot * vc[j]);
    } else if (htid < 128) {
        const int i = htid - 64;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R3[q][i] + S.R4[q][i];
        const double pi = tau2 * a;
        const double dot = wave_sum(pi * vn[i]);
        S.pv[i] = pi + (-0.5 * tau2 * dot) * vn[i];
    }
    SB7_STAMP(7);
    lds_barrier();
    SB7_STAMP(8);
    asm volatile("" : "+v"(htid), "+v"(ib), "+v"(j0), "+v"(o0), "+v"(low));   // re-derive per phase: see the top of the body
    // ---- P5: updated tiles -> exchange frames; only the finished entries go to HBM ----
    if (DIAG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SB7_STAMP(10); }
    if (comp) {
        const int item = item0 ? 0 : done;
        constexpr int f = PAR, g = PAR ^ 1;
        // no partner sweep s+1 (end of the matrix): nobody picks the tiles up from LDS, store them instead
        const int Lp = (n - 2 - sw < SB) ? (n - 2 - sw) : SB;
        const bool fwd = FAST ? true : ((sw + 1 < n - 2) && (Lp >= 2));
        double *__restrict__ Bb = AB + ((size_t)r0 * LD + L);
        double *__restrict__ Db = AB + (size_t)rn * LD;
        {   // B tile (two passes keep the register pressure of each below the 256 of a 512-thread workgroup)
            double wi[4], zj[4], vcj[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) { wi[x] = S.w[SB3_ROW(x)]; zj[x] = S.z[j0 + x]; vcj[x] = vc[j0 + x]; }
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int i = SB3_ROW(ri), j = j0 + cj;
                    const bool inb = FAST ? true : ((i < L2) && (j < L));
                    double bnew = Bc[ri][cj] - (wi[ri] * vcj[cj] + vni[ri] * zj[cj]);
                    if (j == 0) bnew = (i == 0) ? beta2 : 0.0;
                    bnew = inb ? bnew : 0.0;
                    if (fwd) {
                        SH.FB[f][j][i] = bnew;
                        if (ri == 0 && ib == 0 && inb) {      // row 0 of this item = last row of item k-1's frame
                            if (j >= 1) fd_ref<g>(SH, L, j) = bnew; else SH.FB[g][SB][L] = bnew;
                        }
                        // the annihilated column is outside every tile of sweep s+1 but inside item k-1's tile of
                        // sweep s+2, and HBM still holds sweep s-1's bulge there: the zeros must land
                        if (cj == 0 && jb == 0 && i >= 1 && inb) {
                            double zero;                      // materialised here: kept in a register across the loop
                            asm volatile("v_mov_b64 %0, 0" : "=v"(zero));   // the constant gets spilled and reloaded
                            Bb[SB7_OFF(ri, 0)] = zero;        // behind an s_waitcnt vmcnt(0), i.e. behind the prefetch
                        }
                    } else if (inb) Bb[SB7_OFF(ri, cj)] = bnew;
                }
        }
        asm volatile("" ::: "memory");
        {   // D tile
            double pi[4], pj[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) { pi[x] = S.pv[SB3_ROW(x)]; pj[x] = S.pv[j0 + x]; }
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int i = SB3_ROW(ri), j = j0 + cj;
                    const bool lw = (low >> (ri * 4 + cj)) & 1u;
                    const bool isl = lw && (FAST ? true : (i < L2));
                    const double dnew = isl ? (Dc[ri][cj] - (vni[ri] * pj[cj] + pi[ri] * vnj[cj])) : 0.0;
                    if (fwd) {
                        if (lw) fd_ref<f>(SH, i, j) = dnew;
                        if (cj == 0 && jb == 0) SH.FB[f][SB][i] = dnew;   // first column: last column of B'_k / x of sweep s+1
                        if (ri == 0 && cj == 0 && ib == 0 && jb == 0) {
                            if (item0) AB[(size_t)rn * LD] = dnew;        // A(s+1, s+1) is final
                            else fd_ref<g>(SH, L, L) = dnew;
                        }
                    } else if (isl) Db[SB7_OFF(ri, cj)] = dnew;
                }
        }
        if (!FAST && item0 && htid < L2) AB[(size_t)sw * LD + 1 + htid] = (htid == 0) ? beta2 : 0.0;   // finished column s
        if (htid == 0) { SH.tag[f][0] = sw; SH.tag[f][1] = item; }
        if (FAST) {                                          // the prefetched D tile has arrived: drop its upper triangle
#pragma unroll
            for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) Dn[ri][cj] = ((low >> (ri * 4 + cj)) & 1u) ? Dn[ri][cj] : 0.0;
        }
    }
    // ---- state update + publication ----
    if (!FAST && act == ACT_PRELOAD) state = 2;
    else if (comp) {
        const bool more = FAST ? true : (rn + L2 < n);
        st.r0 = rn; st.L = L2; st.L2 = pL2; st.tau = tau2;
        done = item0 ? 1 : done + 1;
        if (more) state = 3;
        else {
            sw += pc.stride; done = 0;
            const int Lnext = (n - 1 - sw < SB) ? (n - 1 - sw) : SB;
            state = (sw < n - 2 && Lnext >= 2) ? 1 : 0;
        }
        if (htid == 0) { SH.sweep[0] = sw; SH.done[0] = done; if (state == 0) SH.fin[0] = 1; }
    }
    if (htid == 0) { SH.astate = state; if (pc.paired) SH.pw = pollv; }
}

#undef SB7_FASTV
#define SB7_FASTV (-1)
// ---- half A: one super-step -------------------------------------------------------------------
template <int PAR, int DIAG>
__device__ __forceinline__ void superstep_v7A(double *__restrict__ AB, Sb7Shared &SH, int htid, int ib, int j0,
                                              unsigned o0, unsigned low, int n,
                                              int &state, int &sw, int &done, ChaseState &st, Diag7 &dg, double &xpre,
                                              double (&Bc)[4][4], double (&Dc)[4][4], double (&Bn)[4][4], double (&Dn)[4][4],
                                              const Pair7 &pc)
{
    SB7_STAMP(9);
    lds_barrier();                                           // barrier 0
    SB7_STAMP(0);
    int act = ACT_IDLE;
    {
        const int osw = __builtin_amdgcn_readfirstlane(SH.sweep[1]), odn = __builtin_amdgcn_readfirstlane(SH.done[1]), ofin = __builtin_amdgcn_readfirstlane(SH.fin[1]);
        if (state == 1) {
            // B (this workgroup's, on its previous sweep sb = sw - stride + 1) must be (a) alone: SB7_LAG_HBM items
            // into it, so that the tiles this sweep loads are in HBM (paired: the partner's progress is checked
            // below instead), and (b) at most one item from its end: item 0 of this sweep (next super-step)
            // overwrites an exchange frame
            const int sb = sw - pc.stride + 1;
            const int kb = (n - sb - 1 + SB - 1) / SB;       // items of sweep sb
            const int lag = pc.paired ? 0 : SB7_LAG_HBM;
            const int need = (kb - 2 > lag) ? kb - 2 : lag;
            bool ok = (sw < pc.stride) || ofin || (osw > sb) || (osw == sb && odn >= need);
            if (ok && pc.paired) {
                // start a sweep with some slack behind the partner, so that timing jitter does not end in holds in
                // mid-sweep; the slack must fit the ring (4 sweeps in flight over kb items each), hence the clamp
                const int P = pc.stride / 2;
                int lead = (kb - 9 * P) / P;
                lead = lead < 0 ? 0 : (lead > pc.lead ? pc.lead : lead);
                ok = sb6_dep_ok(sb7_rfl64(SH.pw), sw, pc.margin + lead);               // not yet: idle and poll again
            }
            act = ok ? ACT_PRELOAD : ACT_IDLE;
        } else if (state == 2) act = ACT_ITEM0;
        else if (state == 3) act = ACT_CHASE;
    }
    if (__builtin_amdgcn_readfirstlane((SH.fin[0] && SH.fin[1]) || SH.abort)) { state = -1; return; }
    if (pc.paired && (state == 2 || state == 3)) {
        // this item prefetches the next one: the partner must have published SB6_MARGIN items more of sweep sw-1
        // (plus SB6_HYST once a wait has begun).  Otherwise half A HOLDS here for TWO super-steps at a time (an
        // even number keeps the parity-indexed buffers, register sets and frames aligned with half B), while half B
        // keeps running -- that is what the partner in turn waits for; blocking the whole workgroup deadlocks
        // the pair (measured).  Nothing of A's state is touched while it holds.
        const int need = ((state == 2) ? 1 : done + 1) + pc.margin;
        bool waited = false;
        for (int guard = 0; !sb6_dep_ok(sb7_rfl64(SH.pw), sw, need + (waited ? pc.hyst : 0)); ++guard) {
            waited = true;
#pragma unroll 1
            for (int r = 0; r < 2; ++r) {
                lds_barrier();                                // every wave has evaluated the condition above: only
                if (htid == 0) {                              // now may the shared words it read change
                    SH.pw = __hip_atomic_load(pc.pollp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (r == 0) { pc.C->nwait[0] += 1; if (sw * 2 < n) pc.C->nwait[1] += 1; if (state == 2) pc.C->wcycles[0] += 1; }
                    if (guard > SB7_MAX_STALL) { SH.abort = 1; SH.viol = 1; }
                }
                lds_barrier(); lds_barrier(); lds_barrier();  // the rest of this super-step ...
                lds_barrier();                                // ... and barrier 0 of the next
                if (__builtin_amdgcn_readfirstlane(SH.abort)) { state = -1; return; }
            }
        }
    }
    const bool fast = (act == ACT_CHASE) && (st.L == SB) && (st.L2 == SB) && (n - (st.r0 + 2 * SB) >= SB) && (sw + 1 < n - 2);
    if (fast) body_v7A<PAR, 1, DIAG>(AB, SH, htid, ib, j0, o0, low, n, state, sw, done, st, dg, xpre, Bc, Dc, Bn, Dn, act, pc);
    else body_v7A<PAR, 0, DIAG>(AB, SH, htid, ib, j0, o0, low, n, state, sw, done, st, dg, xpre, Bc, Dc, Bn, Dn, act, pc);
}

#undef SB7_FASTV
#define SB7_FASTV FAST
// ---- half B: one super-step after barrier 0 (FAST: interior chase item, 64 x 64 tiles) ----------
template <int PAR, int FAST, int DIAG>
__device__ __forceinline__ void body_v7B(double *__restrict__ AB, Sb7Shared &SH, int htid_, int ib_, int j0_,
                                              unsigned o0_, unsigned low_, int n,
                                              int &state, int &sw, int &done, int &stall, ChaseState &st, Diag7 &dg, int act,
                                              const Pair7 &pc, unsigned long long &pubv, unsigned long long &pendv)
{
    constexpr int LD = 2 * SB;
    Sb7Lds &S = SH.S[1];
    const double *vc = PAR ? S.vb : S.va;
    double *vn = PAR ? S.va : S.vb;
    // the lane constants are re-derived per super-step: kept live across the whole loop (with everything the
    // compiler pre-computes from them: masks, row indices, LDS addresses) they do not fit the 256 registers
    int htid = htid_, ib = ib_, j0 = j0_;
    unsigned o0 = o0_, low = low_;
    asm volatile("" : "+v"(htid), "+v"(ib), "+v"(j0), "+v"(o0), "+v"(low));
    const int jb = j0 >> 2;
    const bool comp = FAST ? true : (act >= ACT_ITEM0);
    const bool item0 = FAST ? false : (act == ACT_ITEM0);
    const int L0 = (n - 1 - sw < SB) ? (n - 1 - sw) : SB;
    if (item0) { st.r0 = sw + 1; st.L = 0; st.L2 = L0; st.tau = 0.0; }
    const int r0 = __builtin_amdgcn_readfirstlane(comp ? st.r0 : 0);
    const int L = FAST ? SB : __builtin_amdgcn_readfirstlane(comp ? st.L : 0);
    const int L2 = FAST ? SB : __builtin_amdgcn_readfirstlane(comp ? st.L2 : 0), rn = r0 + L;
    // ---- P1: this item's tiles from the frame of this super-step's parity ----
    double Bc[4][4], Dc[4][4];
    double xcol = 0.0;
    {
        constexpr int f = PAR;
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int i = SB3_ROW(ri), j = SB3_COL(cj);
                // row SB-1 of B' left of its last column lies outside even the bulged band (A never writes it)
                const bool okb = (FAST ? true : (comp && (i < L2) && (j < L))) && !(i == SB - 1 && j < L - 1);
                const bool okd = (FAST ? true : (comp && (i < L2))) && (j <= i);
                const double bv = SH.FB[f][j + 1][i + 1];
                const double cv = FAST ? bv : SH.FB[f][SB][i + 1];      // L == SB: column L-1 reads FB[f][SB][.] anyway
                const double dv = fd_ref<f>(SH, i + 1, j + 1);
                Bc[ri][cj] = okb ? ((!FAST && j == L - 1) ? cv : bv) : 0.0;
                Dc[ri][cj] = okd ? dv : 0.0;
            }
        if (!FAST && htid < 64) { const double xv = SH.FB[f][SB][htid + 1]; xcol = (item0 && htid < L2) ? xv : 0.0; }
        double vcj[4];
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) vcj[cj] = vc[j0 + cj];
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            double a = 0.0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) a += Bc[ri][cj] * vcj[cj];
            S.R1[jb][SB3_ROW(ri)] = a;
        }
        if (jb == 0) {
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) S.x0[SB3_ROW(ri)] = Bc[ri][0];
        }
    }
    SB7_STAMP(1);
    lds_barrier();
    SB7_STAMP(2);
    // ---- P2 ----
    if (htid < 64) {
        const int i = htid;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R1[q][i];
        const double wi = item0 ? 0.0 : st.tau * a;
        const double xc = item0 ? xcol : (S.x0[i] - wi * vc[0]);
        const double xi = (FAST || i < L2) ? xc : 0.0;
        double beta2, tau2;
        const double vi = wave_house(xi, i, L2, &beta2, &tau2);
        const double sdot = wave_sum(vi * wi);
        S.w[i] = wi; vn[i] = vi;
        if (i == 0) { S.sc[1] = beta2; S.sc[2] = tau2; S.sc[3] = sdot; }
    }
    SB7_STAMP(3);
    lds_barrier();
    SB7_STAMP(4);
    const double beta2 = S.sc[1], tau2 = S.sc[2], sdot = S.sc[3];
    asm volatile("" : "+v"(htid), "+v"(ib), "+v"(j0), "+v"(o0), "+v"(low));   // re-derive per phase: see the top of the body
    // ---- P3 ----
    double vni[4], vnj[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) { vni[x] = vn[SB3_ROW(x)]; vnj[x] = vn[j0 + x]; }
    {
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            double a = 0.0;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) a += vni[ri] * Bc[ri][cj];
            S.R2[ib][j0 + cj] = a;
        }
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            double a = 0.0;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) a += Dc[ri][cj] * vnj[cj];
            S.R3[jb][SB3_ROW(ri)] = a;
        }
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            double a = 0.0;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) a += (SB3_ROW(ri) != j0 + cj) ? Dc[ri][cj] * vni[ri] : 0.0;
            S.R4[ib][j0 + cj] = a;
        }
    }
    SB7_STAMP(5);
    lds_barrier();
    SB7_STAMP(6);
    // ---- P4 ----
    if (htid < 64) {
        const int j = htid;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R2[q][j];
        S.z[j] = tau2 * (a - sdot * vc[j]);
    } else if (htid < 128) {
        const int i = htid - 64;
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += S.R3[q][i] + S.R4[q][i];
        const double pi = tau2 * a;
        const double dot = wave_sum(pi * vn[i]);
        S.pv[i] = pi + (-0.5 * tau2 * dot) * vn[i];
    }
    SB7_STAMP(7);
    lds_barrier();
    SB7_STAMP(8);
    asm volatile("" : "+v"(htid), "+v"(ib), "+v"(j0), "+v"(o0), "+v"(low));   // re-derive per phase: see the top of the body
    // ---- P5: results to HBM ----
    if (pc.paired) {
        // the stores of the previous super-step are complete by now (the wait is free): what was pending becomes
        // publishable, i.e. the partner learns of an item one super-step after its stores were issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pubv = pendv;
    }
    {
        double wi[4], zj[4], vcj[4], pi[4], pj[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            wi[x] = S.w[SB3_ROW(x)]; zj[x] = S.z[j0 + x]; vcj[x] = vc[j0 + x]; pi[x] = S.pv[SB3_ROW(x)]; pj[x] = S.pv[j0 + x];
        }
        double *__restrict__ Bb = AB + ((size_t)r0 * LD + L);
        double *__restrict__ Db = AB + (size_t)rn * LD;
        double *__restrict__ Bl = Bb + o0, *__restrict__ Dl = Db + o0;   // FAST: lane pointer + compile-time offsets
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int i = SB3_ROW(ri), j = j0 + cj;
                double bnew = Bc[ri][cj] - (wi[ri] * vcj[cj] + vni[ri] * zj[cj]);
                if (j == 0) bnew = (i == 0) ? beta2 : 0.0;
                const double dnew = Dc[ri][cj] - (vni[ri] * pj[cj] + pi[ri] * vnj[cj]);
                if (FAST) {
                    Bl[cj * (LD - 1) + 16 * ri] = bnew;
                    if ((low >> (ri * 4 + cj)) & 1u) Dl[cj * (LD - 1) + 16 * ri] = dnew;
                } else {
                    if (i < L2 && j < L) Bb[SB7_OFF(ri, cj)] = bnew;
                    if (((low >> (ri * 4 + cj)) & 1u) && i < L2) Db[SB7_OFF(ri, cj)] = dnew;
                }
            }
        if (!FAST && item0 && htid < L2) AB[(size_t)sw * LD + 1 + htid] = (htid == 0) ? beta2 : 0.0;
    }
    if (comp) {
        const bool more = (rn + L2 < n);
        const int nL2 = more ? ((n - (rn + L2) < SB) ? (n - (rn + L2)) : SB) : 0;
        st.r0 = rn; st.L = L2; st.L2 = nL2; st.tau = tau2;
        done = item0 ? 1 : done + 1;
        if (more) state = 3;
        else {
            sw += pc.stride; done = 0;
            const int Lnext = (n - 1 - sw < SB) ? (n - 1 - sw) : SB;
            state = (sw < n - 2 && Lnext >= 2) ? 2 : 0;
        }
        if (htid == 0) { SH.sweep[1] = sw; SH.done[1] = done; if (state == 0) SH.fin[1] = 1; }
    }
    if (pc.paired) pendv = ((unsigned long long)sw << 20) | (unsigned)done;
    // B missed its window (cannot happen while the halves run in lock-step): give up instead of spinning;
    // written after the last barrier, so that every wave sees it after barrier 0 of the next super-step
    if (stall > SB7_MAX_STALL && htid == 0) { SH.abort = 1; SH.viol = 1; }
}

#undef SB7_FASTV
#define SB7_FASTV (-1)
// ---- half B: one super-step -------------------------------------------------------------------
template <int PAR, int DIAG>
__device__ __forceinline__ void superstep_v7B(double *__restrict__ AB, Sb7Shared &SH, int htid, int ib, int j0,
                                              unsigned o0, unsigned low, int n,
                                              int &state, int &sw, int &done, int &stall, ChaseState &st, Diag7 &dg,
                                              const Pair7 &pc, unsigned long long &pubv, unsigned long long &pendv)
{
    SB7_STAMP(9);
    lds_barrier();                                           // barrier 0
    SB7_STAMP(0);
    // publish what became publishable in the previous super-step: EVERY wave of this half has since passed its
    // s_waitcnt vmcnt(0) (and this barrier), so all stores of the items counted in pubv are complete
    if (pc.paired && htid == 0) __hip_atomic_store(pc.pubp, pubv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // item j of sweep sw needs A's items j and j+1 of sweep sw-1 (or A finished that sweep)
    int act = ACT_IDLE;
    if (state != 0) {
        const int asw = __builtin_amdgcn_readfirstlane(SH.sweep[0]), adn = __builtin_amdgcn_readfirstlane(SH.done[0]), afin = __builtin_amdgcn_readfirstlane(SH.fin[0]);
        const int j = (state == 2) ? 0 : done;
        // ... and the frame of this super-step's parity must be the one A filled with item j: true exactly two
        // super-steps after A ran it (when A ends its sweep with item j, one step after is too early)
        const bool ok = (afin || (asw > sw - 1) || (asw == sw - 1 && adn >= j + 2)) &&
                        (__builtin_amdgcn_readfirstlane(SH.tag[PAR][0]) == sw - 1 && __builtin_amdgcn_readfirstlane(SH.tag[PAR][1]) == j);
        if (ok) { act = (state == 2) ? ACT_ITEM0 : ACT_CHASE; stall = 0; }
        else ++stall;
    }
    if (__builtin_amdgcn_readfirstlane((SH.fin[0] && SH.fin[1]) || SH.abort)) { state = -1; return; }
    if (act == ACT_IDLE) {
        if (pc.paired) {                                      // keep the progress pipeline moving (see P5)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pubv = pendv;
        }
        sb7_idle_barriers();
        if (stall > SB7_MAX_STALL && htid == 0) { SH.abort = 1; SH.viol = 1; }
        return;
    }
    const bool fast = (act == ACT_CHASE) && (st.L == SB) && (st.L2 == SB);
    if (fast) body_v7B<PAR, 1, DIAG>(AB, SH, htid, ib, j0, o0, low, n, state, sw, done, stall, st, dg, act, pc, pubv, pendv);
    else body_v7B<PAR, 0, DIAG>(AB, SH, htid, ib, j0, o0, low, n, state, sw, done, stall, st, dg, act, pc, pubv, pendv);
}

template <int DIAG>
__global__ __launch_bounds__(512) void sb2st_kernel_v7(int n, int npad, int batch, double *ABall, double *dall, double *eall,
                                                       int *status, long long *diag, Sb8Ctl *ctl, int P, int margin, int hyst,
                                                       int lead, int force_abort)
{
    Diag7 dg;
    if (DIAG) { for (int q = 0; q < 12; ++q) dg.acc[q] = 0; dg.tlast = (long long)__builtin_amdgcn_s_memtime(); }
    extern __shared__ __attribute__((aligned(16))) unsigned char sb7_raw[];
    Sb7Shared &SH = *reinterpret_cast<Sb7Shared *>(sb7_raw);
    constexpr int LD = 2 * SB;
    const int tid = threadIdx.x, h = __builtin_amdgcn_readfirstlane(tid >> 8), htid = tid & 255, lane = tid & 63, hwave = htid >> 6;
    // ---- which channel; alone (ctl == nullptr) or member w of a ring of P workgroups ----
    int chn = blockIdx.x, w = 0;
    Pair7 pc; pc.paired = 0; pc.stride = 2; pc.pollp = nullptr; pc.pubp = nullptr; pc.C = nullptr; pc.status = status;
    pc.margin = margin; pc.hyst = hyst; pc.lead = lead;
    if (ctl) {
        // blocks b, b+8, b+16, ... are observed to share an XCD (round-robin dispatch): they form a ring, then VERIFY
        const int blk = blockIdx.x, grp = blk / (8 * P), rr = blk % (8 * P);
        chn = grp * 8 + (rr & 7); w = rr >> 3;
        if (chn >= batch) return;
        Sb8Ctl *C = ctl + chn;
        if (tid == 0) {
            // Handshake: 0 = exit, 1 = ring, 2 = alone.  Every member sets its byte; the one that completes the set
            // COMMITs, one that has waited too long ABORTs; both by compare-and-swap on the same word, so exactly one
            // of the two bits is ever set and every member (also one that arrives later) reads the same decision.
            // After an ABORT, or if the members turn out not to share an XCD, member 0 runs the channel alone.
            const unsigned long long xcc = (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu);   // HW_REG_XCC_ID
            const unsigned long long mine = (0x10ull | xcc) << (8 * w);
            unsigned long long full = 0;
            for (int q = 0; q < P; ++q) full |= 0x10ull << (8 * q);
            unsigned long long v = atomicOr(&C->hs, mine) | mine;
            for (int spin = 0; !(v & (SB8_COMMIT | SB8_ABORT)); ++spin) {
                if (force_abort == 1) atomicCAS(&C->hs, v, v | SB8_ABORT);           // test hook: the time-out branch
                else if ((v & full) == full) atomicCAS(&C->hs, v, v | SB8_COMMIT);
                else if (spin > 400000) atomicCAS(&C->hs, v, v | SB8_ABORT);
                else __builtin_amdgcn_s_sleep(4);
                v = __hip_atomic_load(&C->hs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int mode;
            if (v & SB8_ABORT) mode = (w == 0) ? 2 : 0;
            else {
                bool same = true;
                for (int q = 1; q < P; ++q) same = same && (((v >> (8 * q)) & 0xfu) == (v & 0xfu));
                if (force_abort == 2) same = false;                                  // test hook: the cross-XCD branch
                mode = same ? 1 : ((w == 0) ? 2 : 0);
            }
            SH.mode = mode;
        }
        __syncthreads();
        const int mode = __builtin_amdgcn_readfirstlane(SH.mode);
        if (mode == 0) return;
        if (mode == 1 && P > 1) { pc.paired = 1; pc.stride = 2 * P; pc.pollp = &C->prog[(w + P - 1) % P]; pc.pubp = &C->prog[w]; }
        else w = 0;
        pc.C = C;
    }
    const int ib = lane & 15, jb = (lane >> 4) + 4 * hwave;
    const int j0 = 4 * jb;
    const unsigned o0 = (unsigned)(j0 * (LD - 1) + ib);
    unsigned low = 0;
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
#pragma unroll
        for (int ri = 0; ri < 4; ++ri)
            if (SB3_ROW(ri) >= j0 + cj) low |= 1u << (ri * 4 + cj);
    const size_t ch = (size_t)chn;
    double *AB = ABall + ch * ab_stride(npad);
    const int first = pc.paired ? 2 * w + h : h;                 // this half's first sweep
    if (tid == 0) { SH.viol = 0; SH.abort = 0; SH.pw = 0; }
    if (htid == 0) {
        const int L0 = (n - 1 - first < SB) ? (n - 1 - first) : SB;
        const bool any = (first < n - 2) && (L0 >= 2);
        SH.sweep[h] = first; SH.done[h] = 0; SH.fin[h] = any ? 0 : 1;
        SH.tag[h][0] = -1; SH.tag[h][1] = -1;
        if (h == 0) SH.astate = any ? 1 : 0;
    }
    __syncthreads();
    int sw = first, done = 0;
    ChaseState st; st.r0 = 0; st.L = 0; st.L2 = 0; st.L3 = 0; st.tau = 0.0;
    if (h == 0) {
        int state = __builtin_amdgcn_readfirstlane(SH.fin[0]) ? 0 : 1;
        double xpre = 0.0;
        double B0[4][4], D0[4][4], B1[4][4], D1[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) { B0[a][c] = 0.0; D0[a][c] = 0.0; B1[a][c] = 0.0; D1[a][c] = 0.0; }
        for (;;) {
            superstep_v7A<0, DIAG>(AB, SH, htid, ib, j0, o0, low, n, state, sw, done, st, dg, xpre, B0, D0, B1, D1, pc);
            if (state < 0) break;
            superstep_v7A<1, DIAG>(AB, SH, htid, ib, j0, o0, low, n, state, sw, done, st, dg, xpre, B1, D1, B0, D0, pc);
            if (state < 0) break;
        }
    } else {
        int state = __builtin_amdgcn_readfirstlane(SH.fin[1]) ? 0 : 2, stall = 0;
        unsigned long long pubv = (unsigned long long)sw << 20, pendv = pubv;
        for (;;) {
            superstep_v7B<0, DIAG>(AB, SH, htid, ib, j0, o0, low, n, state, sw, done, stall, st, dg, pc, pubv, pendv);
            if (state < 0) break;
            superstep_v7B<1, DIAG>(AB, SH, htid, ib, j0, o0, low, n, state, sw, done, stall, st, dg, pc, pubv, pendv);
            if (state < 0) break;
        }
    }
    if (DIAG && lane == 0 && blockIdx.x == 0) {
        for (int q = 0; q < 12; ++q) diag[(tid >> 6) * 12 + q] = dg.acc[q];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // all stores of this workgroup are complete
    if (tid == 0 && SH.viol) {                                // exchange-frame protocol violated
        if (status) atomicExch(status, BSP_ERR_HIP);
        if (pc.C) atomicExch(&pc.C->err, 3 + 16 * (w + 2 * pc.paired));
    }
    double *d = dall + ch * (size_t)npad, *e = eall + ch * (size_t)npad;
    if (pc.paired) {
        if (tid == 0) __hip_atomic_store(pc.pubp, SB6_FIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w != 0) return;
        if (tid == 0) {                                       // member 0 writes d, e once every member is done
            const int np_ = pc.stride / 2;
            int spin = 0;
            for (int q = 1; q < np_; ++q)
                for (; spin < 8000000; ++spin) {
                    if (__hip_atomic_load(&pc.C->prog[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= SB6_FIN) break;
                    __builtin_amdgcn_s_sleep(8);
                }
            if (spin >= 8000000) { atomicExch(&pc.C->err, 2); if (status) atomicExch(status, BSP_ERR_HIP); }
        }
        __syncthreads();
        for (int j = tid; j < n; j += 512) {
            d[j] = ld_sc1(AB + (size_t)j * LD);
            e[j] = (j < n - 1) ? ld_sc1(AB + (size_t)j * LD + 1) : 0.0;
        }
        return;
    }
    for (int j = tid; j < n; j += 512) {
        d[j] = AB[(size_t)j * LD];
        e[j] = (j < n - 1) ? AB[(size_t)j * LD + 1] : 0.0;
    }
}

size_t sb2st_ctl_bytes(int batch)
{
    return (size_t)batch * sizeof(Sb8Ctl);
}

int launch_sb2st(int n, int npad, int b, int batch, double *d_AB, double *d_d, double *d_e, hipStream_t st, int *d_status,
                 void *ctl)
{
    if (b != SB) return BSP_ERR_ARG;
    int ver = opts().sb2st_version;
    if (ver == 0) ver = n >= 512 ? 9 : 8;                  // two steps wherever there is enough to chase
    // 8 (default): two-sweep workgroups in rings of P per channel; 7: one two-sweep workgroup per channel; 3: one
    // one-sweep workgroup per channel (an independent implementation of the same chase, kept as the cross-check of
    // tests/test_gpu_solve.py::test_sb2st_fallback_paths); v1, v2, v4, v6 are in the history only
    // 9: the two-step route of sbr2.hip (band 64 -> 16 -> 1)
    if (ver == 9) {
        int rc;
        if ((rc = launch_sb2sb(n, npad, batch, d_AB, st))) return rc;
        return launch_sb16st(n, npad, batch, d_AB, d_d, d_e, st, d_status, ctl);
    }
    if (ver != 3 && ver != 7 && ver != 8) return BSP_ERR_ARG;
    if (ver == 7 || ver == 8) {
        // 7: one workgroup per channel; 8: two (the partner CU runs the two sweeps in between), v6's pairing
        static bool attr7 = false;
        if (!attr7) {
            BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sb2st_kernel_v7<0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            BSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sb2st_kernel_v7<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr7 = true;
        }
        static int *d_chk = nullptr;
        const bool chk = opts().sb2st_check != 0;
        if (chk && !d_chk) BSP_HIP(hipMalloc(reinterpret_cast<void **>(&d_chk), sizeof(int)));
        if (chk) BSP_HIP(hipMemsetAsync(d_chk, 0, sizeof(int), st));
        static Sb8Ctl *s_ctl7 = nullptr;
        static int cap7 = 0;
        const int ring_env = opts().sb2st_ring;
        Sb8Ctl *d_ctl = nullptr;
        // ring size: as many workgroups per channel as the chip has CUs for (each needs a whole CU), at most 8;
        // 128 channels -> pairs, 32 channels -> rings of 8 (16 sweeps in flight per channel)
        int P = 1;
        if (ver == 8) {
            P = ring_env > 0 ? ring_env : (batch > 64 ? 2 : (batch > 32 ? 4 : 8));
            if (P > SB8_MAXP) P = SB8_MAXP;
            if (ring_env <= 0)
                while (P > 2 && n / SB < 3 * P) P /= 2;       // measured: rings beyond a sweep's length (n/64 items) still
                                                              // help down to about a third of it (n = 2048: 57 -> 46 ms)
            if (P < 2) P = 2;
            d_ctl = static_cast<Sb8Ctl *>(ctl);
            if (!d_ctl) {
                if (cap7 < batch) {
                    if (s_ctl7) hipFree(s_ctl7);
                    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&s_ctl7), (size_t)batch * sizeof(Sb8Ctl)));
                    cap7 = batch;
                }
                d_ctl = s_ctl7;
            }
            BSP_HIP(hipMemsetAsync(d_ctl, 0, (size_t)batch * sizeof(Sb8Ctl), st));
        }
        const int nblk = (ver == 8) ? ((batch + 7) / 8) * 8 * P : batch;
        const size_t lds = (sizeof(Sb7Shared) + 1023) / 1024 * 1024;
        // measured: holds 13k -> 1.6k per channel with the lead 16; 243 ms (hyst 2) .. 258 ms (hyst 16)
        const int diag7 = opts().sb2st_diag, margin = opts().sb2st_margin, hyst = opts().sb2st_hyst;
        const int lead = opts().sb2st_lead > 0 ? opts().sb2st_lead : (P == 2 ? 8 : 16);   // pairs: 216 -> 210 ms at 128 channels with 8
        const int fab = opts().sb2st_force_abort;
        if (diag7) {
            long long *dbuf = nullptr, h[96];
            BSP_HIP(hipMalloc(reinterpret_cast<void **>(&dbuf), sizeof(h)));
            hipLaunchKernelGGL(sb2st_kernel_v7<1>, dim3(nblk), dim3(512), lds, st, n, npad, batch, d_AB, d_d, d_e,
                               chk ? d_chk : d_status, dbuf, d_ctl, P, margin, hyst, lead, fab);
            BSP_HIP(hipStreamSynchronize(st));
            BSP_HIP(hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost));
            hipFree(dbuf);
            static const char *nm[12] = {"barrier0", "P1", "barrier1", "P2", "barrier2", "P3", "barrier3", "P4", "barrier4", "P5+upd", "vmcnt wait at P5", "-"};
            for (int wv = 0; wv < 8; ++wv) {
                long long tot = 0;
                for (int q = 0; q < 12; ++q) tot += h[wv * 12 + q];
                fprintf(stderr, "sb2st v7 diag wave %d (Mcycles %.1f):", wv, tot / 1e6);
                for (int q = 0; q < 12; ++q) fprintf(stderr, " [%s %.1f%%]", nm[q], 100.0 * h[wv * 12 + q] / (double)tot);
                fprintf(stderr, "\n");
            }
        } else
        hipLaunchKernelGGL(sb2st_kernel_v7<0>, dim3(nblk), dim3(512), lds, st, n, npad, batch, d_AB, d_d, d_e,
                           chk ? d_chk : d_status, (long long *)nullptr, d_ctl, P, margin, hyst, lead, fab);
        if (chk) {
            int hv = 0;
            BSP_HIP(hipStreamSynchronize(st));
            BSP_HIP(hipMemcpy(&hv, d_chk, sizeof(int), hipMemcpyDeviceToHost));
            if (hv) fprintf(stderr, "bspatom: sb2st v7 exchange-frame / pairing failure (status %d)\n", hv);
            if (d_ctl) {
                std::vector<Sb8Ctl> hc(batch);
                BSP_HIP(hipMemcpy(hc.data(), d_ctl, (size_t)batch * sizeof(Sb8Ctl), hipMemcpyDeviceToHost));
                int nerr = 0, nsolo = 0;
                double nw = 0, nw1 = 0, nw2 = 0;
                for (auto &c : hc) {
                    nerr += c.err != 0;
                    bool same = !(c.hs & SB8_ABORT);
                    for (int q = 1; q < P; ++q) same = same && (((c.hs >> (8 * q)) & 0xf) == (c.hs & 0xf));
                    nsolo += !same;
                    nw += c.nwait[0]; nw1 += c.nwait[1]; nw2 += c.wcycles[0];
                }
                fprintf(stderr, "bspatom: sb2st v8: %d channels, rings of %d, %d alone, %d errors, %.1f holds per channel (%.1f in the first half of the sweeps, %.1f before item 0)\n",
                        batch, P, nsolo, nerr, nw / batch, nw1 / batch, nw2 / batch);
                for (int c = 0, shown = 0; c < batch && shown < 8; ++c)
                    if (hc[c].err) { fprintf(stderr, "  channel %d: err %d handshake %016llx\n", c, hc[c].err, hc[c].hs); ++shown; }
                if (nerr) return BSP_ERR_HIP;
            }
            if (hv) return BSP_ERR_HIP;
        }
    }
    else if (ver == 3) {
        const int diag = opts().sb2st_diag;
        if (!diag) hipLaunchKernelGGL(sb2st_kernel_v3<0>, dim3(batch), dim3(256), 0, st, n, npad, d_AB, d_d, d_e, (long long *)nullptr);
        else if (diag == 2) hipLaunchKernelGGL(sb2st_kernel_v3<2>, dim3(batch), dim3(256), 0, st, n, npad, d_AB, d_d, d_e, (long long *)nullptr);
        else if (diag == 3) hipLaunchKernelGGL(sb2st_kernel_v3<3>, dim3(batch), dim3(256), 0, st, n, npad, d_AB, d_d, d_e, (long long *)nullptr);
        else {
            long long *dbuf = nullptr, h[48];
            BSP_HIP(hipMalloc(reinterpret_cast<void **>(&dbuf), sizeof(h)));
            hipLaunchKernelGGL(sb2st_kernel_v3<1>, dim3(batch), dim3(256), 0, st, n, npad, d_AB, d_d, d_e, dbuf);
            BSP_HIP(hipStreamSynchronize(st));
            BSP_HIP(hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost));
            hipFree(dbuf);
            static const char *nm[12] = {"P5->P1 gap", "issue loads", "P1 compute", "barrier1", "P2 (wave0 house)", "barrier2",
                                         "P3 compute", "barrier3", "P4 finalize", "barrier4", "P5 update+store", "sweep start"};
            for (int wv = 0; wv < 4; ++wv) {
                long long tot = 0;
                for (int q = 0; q < 12; ++q) tot += h[wv * 12 + q];
                fprintf(stderr, "sb2st diag wave %d (cycles, share):", wv);
                for (int q = 0; q < 12; ++q) fprintf(stderr, " [%s %lld %.1f%%]", nm[q], h[wv * 12 + q], 100.0 * h[wv * 12 + q] / (double)tot);
                fprintf(stderr, "\n");
            }
        }
    }
    BSP_HIP(hipGetLastError());
    return BSP_OK;
}

}  // namespace bsp
