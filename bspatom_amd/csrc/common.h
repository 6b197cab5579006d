// common.h -- shared declarations of libbspatom (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

// status codes of the C ABI (include/bspatom.h)
#define BSP_OK 0
#define BSP_ERR_HIP (-1)       // HIP runtime error (message on stderr)
#define BSP_ERR_ARG (-2)       // invalid argument
#define BSP_ERR_BSPLVB (-3)    // 'FATAL ERROR - BSPLVB' (reference bsplvb.f90:30-34)
#define BSP_ERR_NOGPU (-4)     // no gfx950 device visible
#define BSP_ERR_UNSUPPORTED (-5)

#define BSP_HIP(x)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "bspatom: HIP error '%s' at %s:%d\n", hipGetErrorString(e_),    \
                    __FILE__, __LINE__);                                                    \
            return BSP_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

// Batch stride (doubles) of the band storage AB[npad][128] of one l-channel.  The 1088-double
// (8.5 KiB) skew keeps the channels, which march through their bands in lock-step during sb2st, from
// hitting the same HBM channel/bank at the same instant (a power-of-two stride would).
__host__ __device__ inline size_t ab_stride(int npad) { return (size_t)npad * 128 + 1088; }

namespace bsp {

// ---- run-time switches (A/B comparisons, diagnostics, test hooks) ---------------------------------
// Read ONCE per process from the environment (BSP_* variables) at the first use; the tests flip
// them in-process through bspatom_set_option (include/bspatom.h) to drive the fallback paths.
struct Options {
    int sb2st_version = 0;       // 0: by size (9 = two steps, sbr2.hip, for n >= 512; else 8); 9 / 8 / 7 / 3 force a generation
    int sb2st_ring = 0;          // BSP_SB2ST_RING: ring size (0 = by channel count)
    int sb2st_margin = 3, sb2st_hyst = 2;
    int sb2st_lead = 0;          // BSP_SB2ST_LEAD: 0 = by ring size (pairs: 8, larger rings: 16; re-scan of round 2)
    int sb2st_check = 0;         // BSP_SB2ST_CHECK: synchronise and report ring formation / holds on stderr
    int sb2st_diag = 0;          // BSP_SB2ST_DIAG: instrumented kernel
    int sb2st_force_abort = 0;   // test hook: 1 = every ring ABORTs its handshake (member 0 runs alone); 2 = pretend the
                                 // members sit on different XCDs (same fallback through the other branch)
    int sy2sb_groups = 2, sy2sb_lookahead = 1, sy2sb_segs = 0;
    int panel_qr = 3;            // BSP_PANEL_QR: 3 = TSQR + Householder reconstruction on many workgroups (tsqr.hip); 2 = one workgroup per
                                 // channel with LDS-DMA (n <= 8256; above 4096 rows it falls back to 1), 1 = the first panel kernel
    int tsqr_max_m = 2048;       // BSP_TSQR_MAX_M: panels with more rows than this (and at most 8192) take the one-workgroup kernel (0 = no limit); a rule in m alone,
                                 // so the arithmetic does not depend on the batch size
    int tsqr_regcap = 1;         // BSP_TSQR_REGCAP: 1 = the 256-register variants of the tsqr.hip kernels (they fit beside one GEMM workgroup), 0 = uncapped
    int gemm_diag = 0;
    int bisect = 3, bisect_ept = 0;
    int bisect_secant = 1;       // BSP_BISECT_SECANT: 1 = lock-step rounds take a safeguarded secant step where a bracket holds one eigenvalue
                                 // (tridiag.hip; ~22 rounds instead of ~50); 0 = bisection only; >= 2: as 1, and the value is the
                                 // fraction of a workgroup's 1024 eigenvalues (1 / value) left to the multisection tail (1 = 8)
    int bisect_tail = 1;         // BSP_BISECT_TAIL: 0 = lock-step bisection to the end (no multisection tail), for A/B timing
    int no_eigvec_prefetch = 0;
    int vec_early = 1;           // BSP_VEC_EARLY: band route, the consumed eigenvector's eigenvalue from the pencil's inertia right after the assembly (bandsect.hip); 0: from the tridiagonal matrix at the end; 2: as 1 with the check made to fail
    int vec_own_cu = 1;          // BSP_VEC_OWN_CU: the early vector's workgroup asks for a CU's whole LDS (eigvec.hip::early_vector_kernel) -- 1: for batches of more than 32 channels (capi.hip), 2: always, 0: never
    int sb2sb_mfma = 1;          // 1: block-chasing item on the matrix cores (sbr2.hip); 0: the first, all-VALU kernel (cross-check)
    int sb16_rows = 1;           // BSP_SB16_ROWS: 1 = band 16 -> 1 with a whole chase item per DPP row, four sweeps per wave (sbr2.hip); 0 = the
                                 // first layout (one tile spread over a wave), kept as the cross-check
    int poison_c = 0;            // test hook: fill the dense C buffer with NaN bit patterns before every solve (nothing outside the
                                 // blocks the standard form writes may ever be read)
    int route = 0;               // BSP_ROUTE: 0 = by size (the band route wherever crawford_supported), 1 = dense route (standard form, sy2sb,
                                 // two-step bulge chasing: north_star's letter, and every pencil wider than 8), 2 = band route (crawford.hip:
                                 // the pencil stays banded; one-column chase on tiles of 8; UNSUPPORTED where it cannot run)
    int fused_probe = 0;         // BSP_FUSED_PROBE: TIMING EXPERIMENT ONLY (wrong results): the rank-128 update runs 16 K-steps instead of 8 and
                                 // symm is not launched -- the upper bound of what a fused update + symm sweep over A22 can gain (round-3 verdict, item 1)
    int cw_items4 = 1;           // BSP_CW_ITEMS4: 1 = crawford_item4_kernel (four chase items per wave, an item per DPP row in the RQ loop),
                                 // 0 = crawford_item_kernel (one item per wave; the cross-check)
    int cw_nw = 1;               // BSP_CW_NW: waves per workgroup of crawford_item4_kernel (1 or 4; a wave never talks to another)
    int cw_streams = 2;          // BSP_CW_STREAMS: the band reduction's channels in this many groups (1 .. 4), each on a stream of its own
    int cw_chunk_min = 1024;     // BSP_CW_CHUNK_MIN: from this many functions on, the S-only part hands its factor over in CW_CHUNKS pieces (crawford_prepare)
    int s_overlap = 1;           // BSP_S_OVERLAP: band route: the S-only part of the reduction (band Cholesky ..) on a stream of its own beside the
                                 // assembly of the H_l
    int cw_diag = 0;             // BSP_CW_DIAG: s_memtime stamps of the phases of crawford_item4_kernel's waves, averaged over a solve (stderr)
    int cw_split = 0;            // BSP_CW_SPLIT: > 0 = the band reduction runs from both ends of the pencil (n a multiple of 8); the value is the share
                                 // of the blocks, in percent, of the leading part (at most 50 = half the chase items).  NOT the default: the
                                 // leading part's fill is chased towards r = 0 and the eigenvalues next to zero pay for it (at 50 %: 21.8 instead
                                 // of 38.2 ms for the reduction, up to 13 x the error next to zero, 7345 instead of 50 of C4's 524288
                                 // eigenvalues beyond 1e-10 relative; profiles/r04_experiments.txt, 11).  0 = one process over all blocks
    int cw_band8 = 1;            // BSP_CW_BAND8: 1 = the band reduction hands over the band of half-width 8 it really leaves (one 8 x 8 block
                                 // made triangular at the end) and the chase runs on tiles of 8; 0 = half-width 15, tiles of 16 (as first built)
    int sb8_wgs = 0;             // BSP_SB8_WGS: workgroups of the tiles-of-8 chase per CU the rings are sized for (0 = what fits: 2)
    int cw_ipw = 0;              // BSP_CW_IPW: items per wave of crawford_item4_kernel (0 = 4; 1, 2: experiment 9; bit-identical results)
    int cw_ldspad = 0;           // BSP_CW_LDSPAD: KB of unused dynamic LDS per workgroup of crawford_item4_kernel (timing experiment: occupancy)
    int cw_onediv = 0;           // BSP_CW_ONEDIV: reflectors of the band route's RQ loop in the one-division form (A/B switch)
    int ktime = 0;               // 1: HIP events around every launch of the kernels in KSlot (bspatom_kernel_times; bench.py's
                                 // per-kernel roofline entries are measured with it in one extra, untimed step)
};
Options &opts();

// one process per GPU: the device of the first successfully created problem (capi.hip); -1 while none exists
int process_device();
int process_device_check(int device);          // BSP_OK, or BSP_ERR_UNSUPPORTED (message on stderr) if another device is latched
void process_device_latch(int device);

// ---- per-kernel launch timing (off unless opts().ktime) --------------------------------------------
// Two events per launch, recorded on the launch's own stream; bspatom_kernel_times() sums the elapsed times per slot after
// the device has drained.  Launches of different streams overlap, so the sums of a slot are sums of launch DURATIONS (what
// rocprofv3 --kernel-trace --stats reports), not wall time.
enum KSlot { KS_SYR2K = 0, KS_SYMM, KS_PANEL_QR, KS_CHAIN, KS_SB2SB, KS_SB16ST, KS_BISECT, KS_STDFORM, KS_CRAWFORD, KS_COUNT };
void ktime_begin(int slot, hipStream_t st);
void ktime_end(int slot, hipStream_t st);
struct KScope {
    int slot; hipStream_t st; bool on;
    KScope(int slot_, hipStream_t st_) : slot(slot_), st(st_), on(opts().ktime != 0) { if (on) ktime_begin(slot, st); }
    ~KScope() { if (on) ktime_end(slot, st); }
};

// ---- batched fp64 MFMA GEMM: C[b] = alpha * A[b] * B[b] + beta * C[b] ----------------------
// Element (i,k) of A[b] is at A + b*bA + i*sAm + k*sAk (one of sAm, sAk must be 1), likewise
// B(k,j) at B + b*bB + k*sBk + j*sBn and C(i,j) at C + b*bC + i*sCm + j*sCn.
struct GemmDesc {
    int M, N, K, batch;
    const double *A; long sAm, sAk, bA;
    const double *B; long sBk, sBn, bB;
    double *C; long sCm, sCn, bC;
    double alpha, beta;
    int lower_only;   // 1: C is square/symmetric, compute only tiles touching i >= j (col-major C)
    int yoff;         // gemm2 MODE 1: first kernel-view row block (= caller column block) of this launch
    int ksplit;       // gemm_kernel: > 1 = split-K, grid.z = batch * ksplit, slice s covers k in [s*kchunk, (s+1)*kchunk)
    int kchunk;       //              and writes alpha * partial to C + s*bCs (beta ignored); set by gemm_splitk_f64
    long bCs;
    // gemm2 MODE 1: the valid tiles are enumerated in 8 x 8 super-blocks; entry s = super-block (sbx, sby) in tile
    // units / 8, sbpre = number of valid tiles up to and including it (set by syr2k_lower_f64)
    int nsb, toff;           // toff: first tile (of the enumeration) of this launch; lower_only = number of tiles in it
    int sbpre[80];           // dwords: a uniform index then reads them with scalar loads (16-bit entries made the tile search a
    int sbxy[80];            // chain of dependent VECTOR loads, ~3 us at the head of every workgroup); sbxy = sbx | sby << 8
};
int gemm_f64(const GemmDesc &g, hipStream_t st);
// C = alpha * A B + beta * C with the K range cut into `splits` slices that run as separate workgroups (for products
// with a small C and a long K: one tile per channel cannot fill the chip); `part` holds splits * batch * M * N
// doubles; the slices are summed in a fixed order (deterministic).
int gemm_splitk_f64(const GemmDesc &g, int splits, double *part, hipStream_t st);
// C[b] (m x 64) = alpha * A[b] (m x 64) * B[b] (64 x 64, ld 64) + beta * C[b], column-major (gemm_f64.hip)
int tsmm64_f64(int m, int batch, const double *A, long lda, long bsA, const double *B, long bsB, double *C, long ldc, long bsC,
               double alpha, double beta, hipStream_t st);
// pipelined, symmetry-aware products of sy2sb (see gemm_f64.hip)
// part: 0 = all tiles, 1 = only column block 0 (look-ahead part), 2 = column blocks >= 1
// seg / nseg: the launch covers the seg-th of nseg equal slices of the tile enumeration (nseg = 1: all)
int syr2k_lower_f64(int m, int batch, double *A22, long ld, long bsA, const double *buf, long ldb, long bsBuf, int seg, int nseg,
                    int part, hipStream_t st);
int symm_lower_f64(int m, int batch, const double *A22, long ld, long bsA, const double *W, long ldw, long bsW,
                   double *Y, long ldy, long bsY, hipStream_t st);

// ---- stage kernels (launchers) -------------------------------------------------------------
// assemble.hip
int launch_point_table(int nkp, int k, int ka, int nfun, const double *d_rt, const double *d_aind,
                       const double *d_xg, const double *d_wg, const double *d_vpot, double *d_ptab,
                       int *d_left, int *d_status, hipStream_t st);
int launch_assemble_bands(int nfun, int k, int ka, int nkp, int kind_pot, const double *d_bl, int l0,
                          int nl, const double *d_ptab, const int *d_left, double *d_SB, double *d_HB,
                          hipStream_t st);
int launch_dipole_bands(int nfun, int k, int ka, int nkp, const double *d_ptab, const int *d_left, double *d_RB,
                        hipStream_t st);
// bandchol.hip
int launch_band_cholesky(int n, int k, const double *d_SB, double *d_UB, double *d_rdiag, int *d_info,
                         hipStream_t st);
int launch_band_cholesky_range(int n, int k, int jstart, int jstop, const double *d_SB, double *d_UB, double *d_rdiag, int *d_info,
                               hipStream_t st);
int launch_band_cholesky_pair(int n, int k, int jstop, const double *d_SB, double *d_UB, double *d_rdiag, int *d_info, hipStream_t st);
// full = 0: C's lower triangle and first block super-diagonal only (all the reduction reads); 1: the whole matrix
int launch_standard_form(int n, int npad, int k, int nl, const double *d_HB, const double *d_UB,
                         const double *d_rdiag, double *d_Y, double *d_C, hipStream_t st, int full = 0);
// crawford.hip: band route -- the banded pencil to a banded standard-form matrix (half-width 8) without the dense C_l
struct CrawfordWork {
    double *SBf, *UBf, *rdiagf;  // [2][k][n] overlap in reversed order and as it is, their Cholesky factors, [2][n] reciprocal pivots
    double *Qel, *LiB;           // [2][N][256] elimination transforms, [2][N][64] inverse diagonal blocks of L (N = ceil(n / 8))
    double *D, *E, *G;           // [nl][2 (N + 2)][64] diagonal / sub-diagonal blocks of the working matrix, the fill in flight (run from
                                 // both ends: [2 nl][blocks of the longer part + 2][64])
    int *info;                   // device word: order of the minor at which the Cholesky factorisation of the reversed overlap broke down
};
bool crawford_supported(int n, int k);
size_t crawford_work_bytes(int n, int k, int nl);
void crawford_carve(void *base, int n, int k, int nl, CrawfordWork *w);
// the S-only part (run: unless s_prepared).  evc (CW_CHUNKS events, or null): the factor is computed in CW_CHUNKS launches, each followed
// by the elimination transforms of its blocks and by evc[c], so that a run given the same events starts on the first blocks while the
// rest of the factor is still being computed (without them the caller orders the run behind the whole of it)
constexpr int CW_CHUNKS = 4;
int crawford_prepare(int n, int k, const double *d_SB, const CrawfordWork &w, hipStream_t st, hipEvent_t *evc = nullptr);
int crawford_run(int n, int npad, int k, int nl, const double *d_SB, const double *d_HB, const CrawfordWork &w, double *d_AB,
                 hipStream_t st, bool s_prepared = false, hipEvent_t *evc = nullptr, hipStream_t aux0 = nullptr);
// sy2sb.hip
// tsqr.hip: panel factorisation on many workgroups (TSQR + Householder reconstruction), BSP_PANEL_QR=3
long tsqr_scr_doubles(int npad);
long tsqr_cntr_ints(int npad);
int tsqr_panel(int npad, int r0, int c0, int batch, double *d_A, double *buf, double *W, double *scr, int *cntr, hipStream_t st);
struct Sy2sbWork {
    double *tsqr_scr;   // [batch][tsqr_scr_doubles(npad)]
    int *tsqr_cntr;     // [batch][tsqr_cntr_ints(npad)], zeroed at the start of every sy2sb_run
    double *buf2;   // second [V | Z | V] set (panels alternate: look-ahead QR writes one while the update reads the other)
    double *tau2;
    double *buf;    // [batch][npad][3*nb]  : [V | Z | V]
    double *W;      // [batch][npad][nb]
    double *G;      // [batch][nb][nb]
    double *T;      // [batch][nb][nb]
    double *Kmat;   // [batch][nb][nb]
    double *tau;    // [batch][nb]
    double *part;   // [SY2SB_SPLITK][batch][nb][nb] split-K partial sums of G and K
};
constexpr int SY2SB_SPLITK = 16;
size_t sy2sb_work_bytes(int npad, int nb, int batch);
void sy2sb_carve(void *base, int npad, int nb, int batch, Sy2sbWork *w);
int sy2sb_run(int npad, int nb, int batch, double *d_A, const Sy2sbWork &w, hipStream_t st);
int sy2sb_panel_only(int npad, int c0, int batch, double *d_A, const Sy2sbWork &w, hipStream_t st);
int launch_extract_band(int npad, int nb, int batch, const double *d_A, double *d_AB, hipStream_t st);
// sb2st.hip
// ctl: device scratch of sb2st_ctl_bytes(batch) bytes owned by the caller (per problem); nullptr -> a
// process-wide buffer (stage-level entry points only).
size_t sb2st_ctl_bytes(int batch);
// two-step route (sbr2.hip): band 64 -> 16 by block bulge chasing, then 16 -> tridiagonal in an LDS window
int launch_sb2sb(int n, int npad, int batch, double *d_AB, hipStream_t st);
int launch_sb16st(int n, int npad, int batch, double *d_AB, double *d_d, double *d_e, hipStream_t st, int *d_status = nullptr,
                  void *ctl = nullptr, int hb = 16);
int launch_sb2st(int n, int npad, int b, int batch, double *d_AB, double *d_d, double *d_e,
                 hipStream_t st, int *d_status = nullptr, void *ctl = nullptr);
// tridiag.hip
int launch_bisect(int n, int ldn, int batch, const double *d_d, const double *d_e, double *d_w,
                  long ldw, hipStream_t st);
int launch_bisect_one(int n, const double *d_d, const double *d_e, int m, double *d_out, hipStream_t st);
// bandsect.hip: eigenvalue m (0-based, ascending) of the banded pencil itself (upper bands, k - 1 <= 8), one workgroup
int launch_band_multisect(int n, int k, const double *d_SB, const double *d_HB, int m, double *d_out, hipStream_t st);
// eigvec.hip: that eigenvalue and its eigenvector in one launch of one workgroup (d_HB: the channel's own band; d_work: invit_work_doubles
// for one vector).  own_cu: the workgroup asks for OWN_CU_LDS bytes of LDS, so that nothing of the band route's kernels fits beside it
constexpr size_t OWN_CU_LDS = 144 * 1024;           // of 160 KB: 16 KB left
int launch_early_vector(int n, int k, const double *d_SB, const double *d_HB, int m, double *d_E, double *d_work, double *d_vec,
                        int *d_info, hipStream_t st, bool own_cu);
// eigvec.hip
// vector iv: channel chan[iv] of HB (HB + chan*k*n), eigenvalue E[iv]; work: nvec*invit_work_doubles
int launch_inverse_iteration(int n, int k, int nvec, const double *d_SB, const double *d_HB,
                             const int *d_chan, const double *d_E, double *d_work, double *d_vec,
                             int *d_info, hipStream_t st);
size_t invit_work_doubles(int n, int k);
int launch_band_apply(int n, int k, const double *d_RB, const double a[3], const double *d_x, double *d_v, hipStream_t st);
int launch_dots(int n, int m, const double *d_Z, const double *d_v, double *d_D, hipStream_t st);
int launch_wf_tabulate(int nkp, int k, int n, const double *d_rt, const double *d_c, double ra,
                       double rb, int npts, double *d_r, double *d_u, int *d_status, hipStream_t st);

// capi.hip: enqueue Cholesky -> standard form -> sy2sb -> sb2st -> bisection on `st` for nl channels
// whose upper bands are already in d_SB / d_HB.  ev (optional): 5 events recorded at the stage
// boundaries [after chol+std, after sy2sb, after sb2st, after bisect] starting from ev[0] = begin.
struct PipeBufs {
    double *UB, *rdiag, *Y, *C, *AB, *d, *e;
    void *work;
    int *info;
    int *status = nullptr;   // device word set to a BSP_ERR_* code by kernels that detect a failure
    void *sbctl = nullptr;   // sb2st pairing/progress control block (sb2st_ctl_bytes(nl))
    void *cwork = nullptr;   // band route: crawford_work_bytes(n, k, nl); Y, C, work may be null when only that route runs
    hipEvent_t *s_events = nullptr;   // band route, S-only part already enqueued elsewhere (s_prepared): CW_CHUNKS events, see crawford_prepare
    hipStream_t aux0 = nullptr;       // band route: the caller's stream for the reduction's second group of channels (crawford_run)
    hipEvent_t chase_after = nullptr; // band route: the chase waits for this event (the early vector's kernels hold LDS two workgroups of the chase need)
};
// 1 = dense route, 2 = band route, for a pencil of this size under the current switches (BSP_ROUTE)
int pipeline_route(int n, int k);
size_t pipe_bytes_per_channel(int npad);
int pipeline_enqueue(int n, int npad, int k, int nl, const double *d_SB, const double *d_HB,
                     const PipeBufs &b, double *d_Eout, hipStream_t st, hipEvent_t *ev, bool with_bisect = true, bool s_prepared = false);

}  // namespace bsp
