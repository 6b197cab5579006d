// capi.hip -- the C ABI of libbspatom (include/bspatom.h): problem handle, the batched solve
// pipeline and the stage-level entry points.  One HIP stream per problem; every stage of a solve
// is enqueued on it back-to-back (no host synchronisation between stages) and timed with events.
#include <atomic>
#include <cstring>
#include <new>
#include <vector>
#include "common.h"
#include "host_setup.h"
#include "../../include/bspatom.h"

using namespace bsp;

struct bspatom_problem {
    HostSetup hs;
    int device;
    int npad;
    hipStream_t st = nullptr;
    // device: set-up tables
    double *d_rt = nullptr, *d_aind = nullptr, *d_xg = nullptr, *d_wg = nullptr, *d_vpot = nullptr, *d_bl = nullptr;
    double *d_ptab = nullptr; int *d_left = nullptr; int *d_status = nullptr;
    bool ptab_ready = false;
    // device: per-solve buffers (sized for cap_nl channels)
    int cap_nl = 0;
    double *d_SB = nullptr, *d_HB = nullptr, *d_UB = nullptr, *d_rdiag = nullptr;
    double *d_Y = nullptr, *d_C = nullptr, *d_AB = nullptr, *d_d = nullptr, *d_e = nullptr, *d_E = nullptr;
    void *d_work = nullptr, *d_sbctl = nullptr;
    void *d_cwork = nullptr;         // band route (crawford.hip)
    int cap_dense = 0, cap_band = 0; // channels the dense buffers (Y, C, work) / the band route's work area are sized for
    int *d_info = nullptr;
    // eigenvector / wave-function scratch
    double *d_vwork = nullptr, *d_vec = nullptr, *d_wfr = nullptr, *d_wfu = nullptr, *d_Esel = nullptr;
    int *d_chan = nullptr;
    int wf_cap = 0;
    // the eigenvector the reference consumes, Hij(:, n0_ini) of channel l_ini (matrices.f90:267), computed on a side
    // stream while the batched bisection runs; bspatom_eigvec returns it when asked for exactly that state
    hipStream_t st2 = nullptr;
    hipEvent_t evx = nullptr;
    hipStream_t stS = nullptr;                // band route: the S-only part of the reduction beside the assembly of the H_l
    hipEvent_t evS = nullptr, evC[bsp::CW_CHUNKS] = {};
    bool pre_early = false, pre_early_ok = false;   // the prefetched vector's eigenvalue came from the pencil (bandsect.hip); it passed the check
    double *d_pvec = nullptr, *d_pE = nullptr;
    int *d_pinfo = nullptr;
    int pre_l = -1, pre_n0 = -1, pre_ch = 0;
    // last solve
    int last_l0 = 0, last_nl = 0;
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double ms[6] = {0, 0, 0, 0, 0, 0};
};

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

namespace bsp {
static std::atomic<int> g_process_device{-1};
int process_device() { return g_process_device.load(); }
int process_device_check(int device)
{
    const int cur = g_process_device.load();
    if (cur >= 0 && cur != device) {
        fprintf(stderr, "bspatom: this process already works on device %d; use one process per GPU (device %d refused)\n", cur, device);
        return BSP_ERR_UNSUPPORTED;
    }
    return BSP_OK;
}
void process_device_latch(int device) { int expect = -1; g_process_device.compare_exchange_strong(expect, device); }
}  // namespace bsp

// ---- run-time switches: environment once, then bspatom_set_option ------------------------------------
namespace {
struct OptName { const char *name; const char *env; int Options::*field; };
const OptName OPT_TABLE[] = {
    {"sb2st_version", "BSP_SB2ST_VERSION", &Options::sb2st_version}, {"sb2st_ring", "BSP_SB2ST_RING", &Options::sb2st_ring},
    {"sb2st_margin", "BSP_SB2ST_MARGIN", &Options::sb2st_margin}, {"sb2st_hyst", "BSP_SB2ST_HYST", &Options::sb2st_hyst},
    {"sb2st_lead", "BSP_SB2ST_LEAD", &Options::sb2st_lead}, {"sb2st_check", "BSP_SB2ST_CHECK", &Options::sb2st_check},
    {"sb2st_diag", "BSP_SB2ST_DIAG", &Options::sb2st_diag}, {"sb2st_force_abort", "BSP_SB2ST_FORCE_ABORT", &Options::sb2st_force_abort},
    {"sy2sb_groups", "BSP_SY2SB_GROUPS", &Options::sy2sb_groups}, {"sy2sb_lookahead", "BSP_SY2SB_LOOKAHEAD", &Options::sy2sb_lookahead},
    {"sy2sb_segs", "BSP_SY2SB_SEGS", &Options::sy2sb_segs}, {"panel_qr", "BSP_PANEL_QR", &Options::panel_qr},
    {"gemm_diag", "BSP_GEMM_DIAG", &Options::gemm_diag}, {"bisect", "BSP_BISECT", &Options::bisect},
    {"bisect_ept", "BSP_BISECT_EPT", &Options::bisect_ept},
    {"bisect_tail", "BSP_BISECT_TAIL", &Options::bisect_tail}, {"bisect_secant", "BSP_BISECT_SECANT", &Options::bisect_secant}, {"no_eigvec_prefetch", "BSP_NO_EIGVEC_PREFETCH", &Options::no_eigvec_prefetch}, {"vec_early", "BSP_VEC_EARLY", &Options::vec_early}, {"vec_own_cu", "BSP_VEC_OWN_CU", &Options::vec_own_cu},
    {"poison_c", "BSP_POISON_C", &Options::poison_c}, {"sb2sb_mfma", "BSP_SB2SB_MFMA", &Options::sb2sb_mfma},
    {"ktime", "BSP_KTIME", &Options::ktime}, {"tsqr_regcap", "BSP_TSQR_REGCAP", &Options::tsqr_regcap},
    {"tsqr_max_m", "BSP_TSQR_MAX_M", &Options::tsqr_max_m}, {"sb16_rows", "BSP_SB16_ROWS", &Options::sb16_rows},
    {"route", "BSP_ROUTE", &Options::route}, {"cw_onediv", "BSP_CW_ONEDIV", &Options::cw_onediv}, {"cw_items4", "BSP_CW_ITEMS4", &Options::cw_items4}, {"cw_nw", "BSP_CW_NW", &Options::cw_nw}, {"cw_ldspad", "BSP_CW_LDSPAD", &Options::cw_ldspad}, {"cw_ipw", "BSP_CW_IPW", &Options::cw_ipw}, {"cw_band8", "BSP_CW_BAND8", &Options::cw_band8}, {"cw_split", "BSP_CW_SPLIT", &Options::cw_split}, {"cw_diag", "BSP_CW_DIAG", &Options::cw_diag}, {"s_overlap", "BSP_S_OVERLAP", &Options::s_overlap}, {"cw_streams", "BSP_CW_STREAMS", &Options::cw_streams}, {"cw_chunk_min", "BSP_CW_CHUNK_MIN", &Options::cw_chunk_min}, {"sb8_wgs", "BSP_SB8_WGS", &Options::sb8_wgs},
    {"fused_probe", "BSP_FUSED_PROBE", &Options::fused_probe},
};
}  // namespace

namespace bsp {
Options &opts()
{
    static Options o = [] {
        Options v;
        for (const OptName &t : OPT_TABLE) {
            const char *e = getenv(t.env);
            if (e) v.*(t.field) = (*e == 0) ? 1 : atoi(e);     // a variable set to the empty string switches a flag on
        }
        return v;
    }();
    return o;
}

// ---- per-kernel launch timing (common.h: KSlot) -------------------------------------------------------
namespace {
struct KRec { int slot; hipEvent_t e0, e1; };
std::vector<KRec> g_krecs;                  // launches recorded since the last collection (one host thread per problem)
std::vector<hipEvent_t> g_kpool;            // events ready for reuse
hipEvent_t g_kopen[KS_COUNT];               // start event of the launch being recorded, per slot
hipEvent_t kevent()
{
    hipEvent_t e = nullptr;
    if (!g_kpool.empty()) { e = g_kpool.back(); g_kpool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}
const char *const KSLOT_NAMES[KS_COUNT] = {"gemm2_kernel<128,128,MODE 1> (rank-128 update, syr2k)", "gemm2_kernel<64,128,MODE 2> (symm Y = A22 W)",
                                          "panel_qr2_kernel / panel_qr_kernel", "sy2sb chain: G, K (split-K gemm_kernel + splitk_reduce), form_T, tsmm64 (W, Z)",
                                          "sb2sb_mfma_kernel (band 64 -> 16)", "sbr_rows_kernel<8> / <16> / sb16st_kernel (one-column chase to tridiagonal)", "bisect3_kernel",
                                          "band_cholesky_kernel + std_form_kernel",
                                          "crawford_item_kernel and its set-up kernels (band route: pencil -> band 15)"};
}  // namespace
void ktime_begin(int slot, hipStream_t st)
{
    hipEvent_t e = kevent();
    g_kopen[slot] = e;
    if (e) (void)hipEventRecord(e, st);
}
void ktime_end(int slot, hipStream_t st)
{
    hipEvent_t e0 = g_kopen[slot], e1 = kevent();
    g_kopen[slot] = nullptr;
    if (!e0 || !e1) return;
    (void)hipEventRecord(e1, st);
    g_krecs.push_back({slot, e0, e1});
}
}  // namespace bsp

extern "C" int bspatom_kernel_times(double *ms, int32_t *launches, int cap)
{
    if (!ms || !launches || cap < KS_COUNT) return BSP_ERR_ARG;
    BSP_HIP(hipDeviceSynchronize());
    for (int i = 0; i < cap; ++i) { ms[i] = 0.0; launches[i] = 0; }
    for (const auto &r : g_krecs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms[r.slot] += t; launches[r.slot] += 1; }
        g_kpool.push_back(r.e0); g_kpool.push_back(r.e1);
    }
    g_krecs.clear();
    return KS_COUNT;
}

extern "C" const char *bspatom_kernel_slot_name(int slot)
{
    return (slot >= 0 && slot < KS_COUNT) ? KSLOT_NAMES[slot] : nullptr;
}

extern "C" int bspatom_set_option(const char *name, int value)
{
    if (!name) return BSP_ERR_ARG;
    for (const OptName &t : OPT_TABLE)
        if (!strcmp(name, t.name)) { opts().*(t.field) = value; return BSP_OK; }
    return BSP_ERR_ARG;
}

extern "C" int bspatom_get_option(const char *name, int *value)
{
    if (!name || !value) return BSP_ERR_ARG;
    for (const OptName &t : OPT_TABLE)
        if (!strcmp(name, t.name)) { *value = opts().*(t.field); return BSP_OK; }
    return BSP_ERR_ARG;
}

extern "C" void bspatom_input_defaults(bspatom_input *in) { input_defaults(in); }

extern "C" int bspatom_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int bspatom_host_setup(const bspatom_input *in, bspatom_sizes *s, double *rt, double *aind, double *xg,
                                  double *wg)
{
    if (!in || !s) return BSP_ERR_ARG;
    HostSetup h;
    {
        const int drc = derive(*in, &h);
        if (drc == -5) { fprintf(stderr, "bspatom: k = %d / ka = %d exceed this build's limits k <= %d, ka <= %d\n", in->k, h.ka, BSPATOM_MAX_K, BSPATOM_MAX_KA); return BSP_ERR_UNSUPPORTED; }
        if (drc != 0) return BSP_ERR_ARG;
    }
    s->nfun = h.nfun; s->k = h.k; s->ka = h.ka; s->nkp = h.nkp; s->nointv = h.nointv;
    s->nbc1 = h.nbc1; s->nbc2 = h.nbc2; s->lmax = h.lmax; s->nintv_exp = h.nintv_exp;
    s->nintv_lin = h.nintv_lin; s->npad = round_up(h.nfun, 64);
    if (rt || aind || xg || wg) {
        build_grid(&h);
        if (rt) memcpy(rt, h.rt.data(), h.rt.size() * sizeof(double));
        if (aind) memcpy(aind, h.aind.data(), h.aind.size() * sizeof(double));
        if (xg) memcpy(xg, h.xg.data(), h.xg.size() * sizeof(double));
        if (wg) memcpy(wg, h.wg.data(), h.wg.size() * sizeof(double));
    }
    return BSP_OK;
}

static void free_solve_buffers(bspatom_problem *p)
{
    hipFree(p->d_SB); hipFree(p->d_HB); hipFree(p->d_UB); hipFree(p->d_rdiag); hipFree(p->d_Y);
    hipFree(p->d_C); hipFree(p->d_AB); hipFree(p->d_d); hipFree(p->d_e); hipFree(p->d_E); hipFree(p->d_work); hipFree(p->d_sbctl);
    hipFree(p->d_cwork);
    p->d_SB = p->d_HB = p->d_UB = p->d_rdiag = p->d_Y = p->d_C = p->d_AB = p->d_d = p->d_e = p->d_E = nullptr;
    p->d_work = nullptr; p->d_sbctl = nullptr; p->d_cwork = nullptr;
    p->cap_nl = 0; p->cap_dense = 0; p->cap_band = 0;
}

extern "C" void bspatom_problem_destroy(bspatom_problem *p)
{
    if (!p) return;
    hipSetDevice(p->device);
    free_solve_buffers(p);
    hipFree(p->d_rt); hipFree(p->d_aind); hipFree(p->d_xg); hipFree(p->d_wg); hipFree(p->d_vpot);
    hipFree(p->d_bl); hipFree(p->d_ptab); hipFree(p->d_left); hipFree(p->d_status); hipFree(p->d_info);
    hipFree(p->d_vwork); hipFree(p->d_vec); hipFree(p->d_wfr); hipFree(p->d_wfu); hipFree(p->d_Esel); hipFree(p->d_chan);
    hipFree(p->d_pvec); hipFree(p->d_pE); hipFree(p->d_pinfo);
    if (p->st2) hipStreamDestroy(p->st2);
    if (p->evx) hipEventDestroy(p->evx);
    if (p->stS) hipStreamDestroy(p->stS);
    if (p->evS) hipEventDestroy(p->evS);
    for (hipEvent_t e : p->evC) if (e) hipEventDestroy(e);
    for (auto &e : p->ev) if (e) hipEventDestroy(e);
    if (p->st) hipStreamDestroy(p->st);
    delete p;
}

template <class T>
static int upload(T **dst, const T *src, size_t count)
{
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(dst), count * sizeof(T)));
    BSP_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return BSP_OK;
}

static int problem_init(bspatom_problem *p, const bspatom_input *in, int device)
{
    {
        const int drc = derive(*in, &p->hs);
        if (drc == -5) {
            fprintf(stderr, "bspatom: B-spline order k = %d (Gauss-Legendre points ka = %d) exceeds the limits k <= %d, ka <= %d of this build\n",
                    in->k, p->hs.ka, BSPATOM_MAX_K, BSPATOM_MAX_KA);
            return BSP_ERR_UNSUPPORTED;
        }
        if (drc != 0) return BSP_ERR_ARG;
    }
    if (p->hs.nfun > BSPATOM_MAX_NFUN) {       // the reference's own limit is nfun <= 9999 (its Enl.dat record is I4, matrices.f90:391)
        fprintf(stderr, "bspatom: nfun = %d exceeds the %d functions per channel this build supports\n", p->hs.nfun, BSPATOM_MAX_NFUN);
        return BSP_ERR_UNSUPPORTED;
    }
    if (p->hs.k > BSPATOM_MAX_K || p->hs.ka > BSPATOM_MAX_KA) {
        fprintf(stderr, "bspatom: B-spline order k = %d (Gauss-Legendre points ka = %d) exceeds the limits k <= %d, ka <= %d of this build\n",
                p->hs.k, p->hs.ka, BSPATOM_MAX_K, BSPATOM_MAX_KA);
        return BSP_ERR_UNSUPPORTED;
    }
    build_grid(&p->hs);
    build_vpot(&p->hs);
    p->device = device;
    p->npad = round_up(p->hs.nfun, 64);
    BSP_HIP(hipSetDevice(device));
    BSP_HIP(hipStreamCreate(&p->st));
    for (auto &e : p->ev) BSP_HIP(hipEventCreate(&e));
    const HostSetup &h = p->hs;
    int rc;
    if ((rc = upload(&p->d_rt, h.rt.data(), h.rt.size()))) return rc;
    if ((rc = upload(&p->d_aind, h.aind.data(), h.aind.size()))) return rc;
    if ((rc = upload(&p->d_xg, h.xg.data(), h.xg.size()))) return rc;
    if ((rc = upload(&p->d_wg, h.wg.data(), h.wg.size()))) return rc;
    if ((rc = upload(&p->d_vpot, h.vpot.data(), h.vpot.size()))) return rc;
    if ((rc = upload(&p->d_bl, h.bl, (size_t)4))) return rc;
    const size_t npt = (size_t)(h.nkp - 1) * h.ka;
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_ptab), npt * (2 * h.k + 3) * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_left), npt * sizeof(int)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_status), sizeof(int)));
    BSP_HIP(hipMemset(p->d_status, 0, sizeof(int)));
    return BSP_OK;
}

extern "C" int bspatom_problem_create(const bspatom_input *in, int device, bspatom_problem **out)
{
    if (!in || !out) return BSP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        fprintf(stderr, "bspatom: no HIP device %d (libbspatom has no CPU path)\n", device);
        return BSP_ERR_NOGPU;
    }
    // One process per GPU: the kernels' opt-ins for large LDS, the side streams of sy2sb and the stage-level
    // scratch buffers are created once per process on the device of the first problem.
    // The device is latched by the first problem that was created SUCCESSFULLY and stays latched for the life of the process
    // (those per-process objects live on it); bsp_dsygv_ (dsygv.hip) honours the same latch.
    int rc0;
    if ((rc0 = bsp::process_device_check(device))) return rc0;
    bspatom_problem *p = new (std::nothrow) bspatom_problem;
    if (!p) return BSP_ERR_ARG;
    p->device = device;
    const int rc = problem_init(p, in, device);
    if (rc) { bspatom_problem_destroy(p); return rc; }      // frees whatever was created so far
    bsp::process_device_latch(device);
    *out = p;
    return BSP_OK;
}

extern "C" int bspatom_problem_sizes(const bspatom_problem *p, bspatom_sizes *s)
{
    if (!p || !s) return BSP_ERR_ARG;
    const HostSetup &h = p->hs;
    s->nfun = h.nfun; s->k = h.k; s->ka = h.ka; s->nkp = h.nkp; s->nointv = h.nointv;
    s->nbc1 = h.nbc1; s->nbc2 = h.nbc2; s->lmax = h.lmax; s->nintv_exp = h.nintv_exp;
    s->nintv_lin = h.nintv_lin; s->npad = p->npad;
    return BSP_OK;
}

extern "C" int bspatom_problem_route(const bspatom_problem *p)
{
    if (!p) return BSP_ERR_ARG;
    return pipeline_route(p->hs.nfun, p->hs.k);
}

extern "C" int bspatom_problem_grid(const bspatom_problem *p, double *rt, double *aind, double *xg, double *wg)
{
    if (!p) return BSP_ERR_ARG;
    const HostSetup &h = p->hs;
    if (rt) memcpy(rt, h.rt.data(), h.rt.size() * sizeof(double));
    if (aind) memcpy(aind, h.aind.data(), h.aind.size() * sizeof(double));
    if (xg) memcpy(xg, h.xg.data(), h.xg.size() * sizeof(double));
    if (wg) memcpy(wg, h.wg.data(), h.wg.size() * sizeof(double));
    return BSP_OK;
}

static int ensure_capacity(bspatom_problem *p, int nl)
{
    if (nl <= p->cap_nl) return BSP_OK;
    free_solve_buffers(p);
    p->last_nl = 0; p->pre_l = -1;                          // the spectra and bands of the last solve are gone
    const HostSetup &h = p->hs;
    const size_t n = h.nfun, k = h.k, np = p->npad, b = nl;
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_SB), k * n * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_HB), b * k * n * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_UB), k * n * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_rdiag), n * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_AB), b * ab_stride((int)np) * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_d), b * np * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_e), b * np * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_E), b * n * sizeof(double)));
    BSP_HIP(hipMalloc(&p->d_sbctl, sb2st_ctl_bytes(nl)));
    if (!p->d_info) BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_info), sizeof(int)));
    p->cap_nl = nl;
    return BSP_OK;
}

// The buffers of the route the next solve takes, allocated when that route is first used: the dense route's Y and C are
// 2 x nl x npad^2 doubles (34 GB at C4), the band route's work area 3 x nl x n x 8.
static int ensure_route_buffers(bspatom_problem *p, int nl, int route)
{
    const HostSetup &h = p->hs;
    const size_t np = p->npad, b = nl;
    if (route == 2 && nl > p->cap_band) {
        hipFree(p->d_cwork); p->d_cwork = nullptr; p->cap_band = 0;
        BSP_HIP(hipMalloc(&p->d_cwork, crawford_work_bytes(h.nfun, h.k, nl)));
        p->cap_band = nl;
    }
    if (route == 1 && nl > p->cap_dense) {
        hipFree(p->d_Y); hipFree(p->d_C); hipFree(p->d_work);
        p->d_Y = p->d_C = nullptr; p->d_work = nullptr; p->cap_dense = 0;
        BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_Y), b * np * np * sizeof(double)));
        BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_C), b * np * np * sizeof(double)));
        BSP_HIP(hipMalloc(&p->d_work, sy2sb_work_bytes((int)np, 64, nl)));
        p->cap_dense = nl;
    }
    return BSP_OK;
}

// enqueue point table (once) + band assembly for channels l0..l0+nl-1; `skip` channels of them are there already (their H_l and S:
// the rest goes behind them in HB and writes its copy of S, the same bits, to `sb_other`)
static int enqueue_assemble(bspatom_problem *p, int l0, int nl, int skip = 0, double *sb_other = nullptr)
{
    const HostSetup &h = p->hs;
    int rc;
    if (!p->ptab_ready) {
        if ((rc = launch_point_table(h.nkp, h.k, h.ka, h.nfun, p->d_rt, p->d_aind, p->d_xg, p->d_wg, p->d_vpot,
                                     p->d_ptab, p->d_left, p->d_status, p->st))) return rc;
        p->ptab_ready = true;
    }
    if (nl - skip <= 0) return BSP_OK;
    return launch_assemble_bands(h.nfun, h.k, h.ka, h.nkp, h.in.kind_pot, p->d_bl, l0 + skip, nl - skip, p->d_ptab, p->d_left,
                                 skip ? sb_other : p->d_SB, p->d_HB + (size_t)skip * h.k * h.nfun, p->st);
}

static int check_status(bspatom_problem *p)
{
    int st = 0;
    BSP_HIP(hipMemcpy(&st, p->d_status, sizeof(int), hipMemcpyDeviceToHost));
    return st;
}

extern "C" int bspatom_dipole_bands(bspatom_problem *p, double *RB)
{
    if (!p || !RB) return BSP_ERR_ARG;
    const HostSetup &h = p->hs;
    BSP_HIP(hipSetDevice(p->device));
    int rc;
    if (!p->ptab_ready) {
        if ((rc = launch_point_table(h.nkp, h.k, h.ka, h.nfun, p->d_rt, p->d_aind, p->d_xg, p->d_wg, p->d_vpot,
                                     p->d_ptab, p->d_left, p->d_status, p->st))) return rc;
        p->ptab_ready = true;
    }
    const size_t cnt = (size_t)3 * (2 * h.k - 1) * h.nfun;
    double *d_RB = nullptr;
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&d_RB), cnt * sizeof(double)));
    rc = launch_dipole_bands(h.nfun, h.k, h.ka, h.nkp, p->d_ptab, p->d_left, d_RB, p->st);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(RB, d_RB, cnt * sizeof(double), hipMemcpyDeviceToHost, p->st);
    if (e == hipSuccess) e = hipStreamSynchronize(p->st);
    hipFree(d_RB);
    if (rc) return rc;
    BSP_HIP(e);
    return check_status(p);
}

extern "C" int bspatom_assemble(bspatom_problem *p, int l0, int nl, double *SB, double *HB)
{
    if (!p || nl <= 0 || l0 < 0) return BSP_ERR_ARG;
    BSP_HIP(hipSetDevice(p->device));
    int rc;
    if ((rc = ensure_capacity(p, nl))) return rc;
    p->last_nl = 0; p->pre_l = -1;                          // d_HB is overwritten: eigvec / eigvecs need a new solve
    if ((rc = enqueue_assemble(p, l0, nl))) return rc;
    BSP_HIP(hipStreamSynchronize(p->st));
    if ((rc = check_status(p))) return rc;
    const size_t kn = (size_t)p->hs.k * p->hs.nfun;
    if (SB) BSP_HIP(hipMemcpy(SB, p->d_SB, kn * sizeof(double), hipMemcpyDeviceToHost));
    if (HB) BSP_HIP(hipMemcpy(HB, p->d_HB, kn * nl * sizeof(double), hipMemcpyDeviceToHost));
    return BSP_OK;
}

namespace bsp {
int pipeline_route(int n, int k)
{
    const int r = opts().route;
    if (r == 1) return 1;
    if (r == 2) return 2;
    return (crawford_supported(n, k) && n >= 32) ? 2 : 1;
}

int pipeline_enqueue(int n, int npad, int k, int nl, const double *d_SB, const double *d_HB, const PipeBufs &b,
                     double *d_Eout, hipStream_t st, hipEvent_t *ev, bool with_bisect, bool s_prepared)
{
    int rc;
    const int route = pipeline_route(n, k);
    if (route == 2) {
        // band route: the pencil stays banded (crawford.hip: half-width 8 in, half-width 8 out), the one-column chase on tiles of 8
        // takes the result as it is (BSP_CW_BAND8=0: half-width 15 out, tiles of 16)
        if (!crawford_supported(n, k)) {
            fprintf(stderr, "bspatom: BSP_ROUTE=2 (band route) takes pencils of half-width k - 1 <= 8 and n >= 16 (k = %d, n = %d)\n", k, n);
            return BSP_ERR_UNSUPPORTED;
        }
        if (!b.cwork) return BSP_ERR_ARG;
        CrawfordWork cw;
        crawford_carve(b.cwork, n, k, nl, &cw);
        if (ev) BSP_HIP(hipEventRecord(ev[1], st));
        if ((rc = crawford_run(n, npad, k, nl, d_SB, d_HB, cw, b.AB, st, s_prepared, b.s_events, b.aux0))) return rc;
        if (ev) BSP_HIP(hipEventRecord(ev[2], st));
        if (b.chase_after) BSP_HIP(hipStreamWaitEvent(st, b.chase_after, 0));
        if ((rc = launch_sb16st(n, npad, nl, b.AB, b.d, b.e, st, b.status, b.sbctl, opts().cw_band8 ? 8 : 16))) return rc;
        if (ev) BSP_HIP(hipEventRecord(ev[3], st));
        if (!with_bisect) return BSP_OK;
        if ((rc = launch_bisect(n, npad, nl, b.d, b.e, d_Eout, n, st))) return rc;
        if (ev) BSP_HIP(hipEventRecord(ev[4], st));
        return BSP_OK;
    }
    if (!b.Y || !b.C || !b.work) return BSP_ERR_ARG;
    if (opts().poison_c) BSP_HIP(hipMemsetAsync(b.C, 0xFF, (size_t)nl * npad * npad * sizeof(double), st));
    {
        KScope kt(KS_STDFORM, st);
        if ((rc = launch_band_cholesky(n, k, d_SB, b.UB, b.rdiag, b.info, st))) return rc;
        if ((rc = launch_standard_form(n, npad, k, nl, d_HB, b.UB, b.rdiag, b.Y, b.C, st))) return rc;
    }
    if (ev) BSP_HIP(hipEventRecord(ev[1], st));
    Sy2sbWork w;
    sy2sb_carve(b.work, npad, 64, nl, &w);
    if ((rc = sy2sb_run(npad, 64, nl, b.C, w, st))) return rc;
    if ((rc = launch_extract_band(npad, 64, nl, b.C, b.AB, st))) return rc;
    if (ev) BSP_HIP(hipEventRecord(ev[2], st));
    if ((rc = launch_sb2st(n, npad, 64, nl, b.AB, b.d, b.e, st, b.status, b.sbctl))) return rc;
    if (ev) BSP_HIP(hipEventRecord(ev[3], st));
    if (!with_bisect) return BSP_OK;                       // the caller enqueues other work first (see solve_impl)
    if ((rc = launch_bisect(n, npad, nl, b.d, b.e, d_Eout, n, st))) return rc;
    if (ev) BSP_HIP(hipEventRecord(ev[4], st));
    return BSP_OK;
}
}  // namespace bsp

// d_info after inverse iterations: 1 + index of a vector whose iterate had norm zero (eigvec.hip), else 0
static int invit_failed(bspatom_problem *p)
{
    int v = 0;
    BSP_HIP(hipMemcpy(&v, p->d_info, sizeof(int), hipMemcpyDeviceToHost));
    return v ? BSP_ERR_UNSUPPORTED : BSP_OK;
}

static int ensure_vec_scratch(bspatom_problem *p)
{
    const HostSetup &h = p->hs;
    if (p->d_vwork) return BSP_OK;
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_vwork), invit_work_doubles(h.nfun, h.k) * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_vec), (size_t)h.nfun * sizeof(double)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_chan), sizeof(int)));
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_Esel), sizeof(double)));
    return BSP_OK;
}

static int solve_impl(bspatom_problem *p, int l0, int nl, double *E_dev_out, double *E_host, int32_t *info)
{
    if (!p || nl <= 0 || l0 < 0) return BSP_ERR_ARG;
    BSP_HIP(hipSetDevice(p->device));
    const HostSetup &h = p->hs;
    const int n = h.nfun, np = p->npad;
    int rc;
    if ((rc = ensure_capacity(p, nl))) return rc;
    const int route = pipeline_route(n, h.k);
    if ((rc = ensure_route_buffers(p, nl, route))) return rc;
    p->last_nl = 0;                                          // valid again only when this solve has completed
    BSP_HIP(hipMemsetAsync(p->d_info, 0, sizeof(int), p->st));
    BSP_HIP(hipEventRecord(p->ev[0], p->st));
    // Band route: S is there once the first channel is assembled, and what the reduction does with S alone (the reversed band, its
    // Cholesky factor -- 1.9 ms of ONE wave --, the elimination transforms) runs on a stream of its own beside the assembly of the
    // other channels (BSP_S_OVERLAP).  The rest writes its copy of S (the same bits) into the dense route's factor buffer.
    bool s_prepared = false;
    if (route == 2 && nl > 1 && opts().s_overlap) {
        if (!p->stS) {
            BSP_HIP(hipStreamCreateWithFlags(&p->stS, hipStreamNonBlocking));
            BSP_HIP(hipEventCreateWithFlags(&p->evS, hipEventDisableTiming));
            for (hipEvent_t &e : p->evC) BSP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if ((rc = enqueue_assemble(p, l0, 1))) return rc;
        BSP_HIP(hipEventRecord(p->evS, p->st));
        BSP_HIP(hipStreamWaitEvent(p->stS, p->evS, 0));
        CrawfordWork cw;
        crawford_carve(p->d_cwork, n, h.k, nl, &cw);
        if ((rc = crawford_prepare(n, h.k, p->d_SB, cw, p->stS, p->evC))) return rc;   // the reduction waits for the chunks as it needs them
        if ((rc = enqueue_assemble(p, l0, nl, 1, p->d_UB))) return rc;
        s_prepared = true;
    } else if ((rc = enqueue_assemble(p, l0, nl))) return rc;
    BSP_HIP(hipEventRecord(p->ev[1], p->st));
    // The consumed eigenvector (l_ini, n0_ini).  Band route (BSP_VEC_EARLY): its eigenvalue from the PENCIL, by multisection on the
    // inertia of H - x S (bandsect.hip), as soon as the channel is assembled, the inverse iteration behind it, one workgroup on a CU
    // of its own (eigvec.hip::early_vector_kernel) beside the reductions -- the vector is there long before the spectra, and checked
    // against them below.  Otherwise: the
    // eigenvalue alone by multisection as soon as the tridiagonal matrices exist, then the inverse iteration on the side stream
    // beside the batched bisection.
    p->pre_l = -1; p->pre_early = false;
    const int tl = h.in.l_ini, tn0 = h.in.n0_ini;
    const bool want_vec = tl >= l0 && tl < l0 + nl && tn0 >= 1 && tn0 <= n && !opts().no_eigvec_prefetch;
    const bool early = want_vec && route == 2 && opts().vec_early;
    auto side_stream = [&]() -> int {
        if (!p->st2) {
            BSP_HIP(hipStreamCreateWithFlags(&p->st2, hipStreamNonBlocking));
            BSP_HIP(hipEventCreateWithFlags(&p->evx, hipEventDisableTiming));
            BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_pvec), (size_t)n * sizeof(double)));
            BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_pE), sizeof(double)));
            BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_pinfo), sizeof(int)));
        }
        return ensure_vec_scratch(p);
    };
    if (early) {
        if ((rc = side_stream())) return rc;
        p->pre_ch = tl - l0;                               // member: must outlive the asynchronous copy
        // on the stream that prepared S, if there is one, behind that (it has nothing else to do; with a stream more -- the main one,
        // this one, the reduction's second group's and a fourth -- two of them share a hardware queue: the reduction's second group
        // then waited behind these 20 ms, measured, 28 -> 44 ms; and the second group on THIS stream starts 1.5 ms late: 28 -> 31 ms)
        hipStream_t sv = s_prepared ? p->stS : p->st2;
        // a CU of its own only where the chase has no CU to spare for it -- more than 32 channels: fewer than eight workgroups per channel
        // or two per CU.  Below that the reduction is over before the vector is, the chase would lose the CU (32 channels: chase 13.9 ->
        // 15.9 ms), and a shared CU has room for both (with 64 channels and a shared CU: chase 16.8 -> 22.5 ms).
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device);
        const bool own_cu = opts().vec_own_cu == 2 || (opts().vec_own_cu == 1 && ((nl + 7) / 8) * 8 * 8 > cus);
        BSP_HIP(hipMemsetAsync(p->d_pinfo, 0, sizeof(int), sv));
        if (s_prepared && tl == l0) BSP_HIP(hipStreamWaitEvent(sv, p->evS, 0));     // its channel was assembled first, S with it
        else {
            BSP_HIP(hipEventRecord(p->evx, p->st));        // the bands are assembled
            BSP_HIP(hipStreamWaitEvent(sv, p->evx, 0));
        }
        if ((rc = launch_early_vector(n, h.k, p->d_SB, p->d_HB + (size_t)p->pre_ch * h.k * n, tn0 - 1, p->d_pE, p->d_vwork, p->d_pvec,
                                      p->d_pinfo, sv, own_cu))) return rc;
        BSP_HIP(hipEventRecord(p->evx, sv));
        p->pre_l = tl; p->pre_n0 = tn0; p->pre_early = true;
    }
    PipeBufs pb{p->d_UB, p->d_rdiag, p->d_Y, p->d_C, p->d_AB, p->d_d, p->d_e, p->d_work, p->d_info, p->d_status, p->d_sbctl, p->d_cwork, s_prepared ? p->evC : nullptr,
                nullptr, (early && opts().vec_early == 3) ? p->evx : nullptr};
    double *Eout = E_dev_out ? E_dev_out : p->d_E;
    if ((rc = pipeline_enqueue(n, np, h.k, nl, p->d_SB, p->d_HB, pb, Eout, p->st, &p->ev[1], false, s_prepared))) return rc;
    if (want_vec && !early) {
        if ((rc = side_stream())) return rc;
        p->pre_ch = tl - l0;
        const int ch = p->pre_ch;
        // The eigenvalue first, ALONE on the main stream (one workgroup, six multisection rounds: ~1 ms on the idle
        // GPU; beside the batched bisection it ran 7-13 ms, sharing a CU or waiting for one), then the inverse
        // iteration on the side stream beside the batched bisection.  Kernel trace before: spectra at +16 ms after the bulge
        // chasing, eigenvector at +30 ms.
        BSP_HIP(hipMemcpyAsync(p->d_chan, &p->pre_ch, sizeof(int), hipMemcpyHostToDevice, p->st));
        BSP_HIP(hipMemsetAsync(p->d_pinfo, 0, sizeof(int), p->st));
        rc = launch_bisect_one(n, p->d_d + (size_t)ch * np, p->d_e + (size_t)ch * np, tn0 - 1, p->d_pE, p->st);
        if (rc == BSP_ERR_UNSUPPORTED) rc = BSP_OK;          // the one-workgroup kernel holds the matrix in LDS (n <= ~9000): beyond
        else if (rc) return rc;                              // that bspatom_eigvec computes the vector on demand from the spectra
        else {
            BSP_HIP(hipEventRecord(p->evx, p->st));
            BSP_HIP(hipStreamWaitEvent(p->st2, p->evx, 0));
            if ((rc = launch_inverse_iteration(n, h.k, 1, p->d_SB, p->d_HB, p->d_chan, p->d_pE, p->d_vwork, p->d_pvec,
                                               p->d_pinfo, p->st2))) return rc;
            BSP_HIP(hipEventRecord(p->evx, p->st2));
            p->pre_l = tl; p->pre_n0 = tn0;
        }
    }
    if ((rc = launch_bisect(n, np, nl, p->d_d, p->d_e, Eout, n, p->st))) return rc;
    BSP_HIP(hipEventRecord(p->ev[5], p->st));
    if (p->pre_l >= 0) BSP_HIP(hipStreamWaitEvent(p->st, p->evx, 0));
    if (E_dev_out)   // keep a copy for bspatom_eigvec
        BSP_HIP(hipMemcpyAsync(p->d_E, E_dev_out, (size_t)nl * n * sizeof(double), hipMemcpyDeviceToDevice, p->st));
    BSP_HIP(hipStreamSynchronize(p->st));
    for (int i = 0; i < 5; ++i) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]);
        p->ms[i] = ms;
    }
    float tot = 0.f;
    hipEventElapsedTime(&tot, p->ev[0], p->ev[5]);
    p->ms[5] = tot;
    if ((rc = check_status(p))) return rc;
    if (p->pre_early) {
        // the early vector's eigenvalue came from inertia counts without pivoting: it must be eigenvalue tn0 of the spectra just
        // computed -- within 1e-10 of the spectrum's width, and no other eigenvalue nearer -- or the vector is forgotten
        // (bspatom_eigvec then takes the eigenvalue from the spectra, as for every other vector)
        const int m = tn0 - 1, i0 = m > 0 ? m - 1 : 0, i1 = m < n - 1 ? m + 1 : n - 1;
        double ev[3] = {0, 0, 0}, lb = 0, ends[2];
        const double *Erow = p->d_E + (size_t)p->pre_ch * n;
        BSP_HIP(hipMemcpy(ev, Erow + i0, (size_t)(i1 - i0 + 1) * sizeof(double), hipMemcpyDeviceToHost));
        BSP_HIP(hipMemcpy(&ends[0], Erow, sizeof(double), hipMemcpyDeviceToHost));
        BSP_HIP(hipMemcpy(&ends[1], Erow + n - 1, sizeof(double), hipMemcpyDeviceToHost));
        BSP_HIP(hipMemcpy(&lb, p->d_pE, sizeof(double), hipMemcpyDeviceToHost));
        const double width = fmax(fabs(ends[0]), fabs(ends[1])), em = ev[m - i0], dist = fabs(lb - em);
        bool ok = dist <= 1e-10 * width;
        for (int i = i0; i <= i1; ++i)
            if (i != m && fabs(lb - ev[i - i0]) < dist) ok = false;
        if (!ok || opts().vec_early == 2) p->pre_l = -1;     // (2: the check fails on purpose -- the fallback's test)
        p->pre_early_ok = ok;
    }
    int cinfo = 0;
    if (route == 2) {
        // The band route factors the REVERSED overlap.  If that broke down, S is not positive definite: DSYGV's info names the
        // leading minor of S itself, which the forward factorisation finds.
        CrawfordWork cw;
        crawford_carve(p->d_cwork, n, h.k, nl, &cw);
        BSP_HIP(hipMemcpy(&cinfo, cw.info, sizeof(int), hipMemcpyDeviceToHost));
        if (cinfo) {
            if ((rc = launch_band_cholesky(n, h.k, p->d_SB, p->d_UB, p->d_rdiag, p->d_info, p->st))) return rc;
            BSP_HIP(hipStreamSynchronize(p->st));
            BSP_HIP(hipMemcpy(&cinfo, p->d_info, sizeof(int), hipMemcpyDeviceToHost));
            if (!cinfo) cinfo = 1;                           // (cannot happen: both factorisations exist or neither)
        }
    } else
    BSP_HIP(hipMemcpy(&cinfo, p->d_info, sizeof(int), hipMemcpyDeviceToHost));
    if (info)
        for (int l = 0; l < nl; ++l) info[l] = cinfo ? n + cinfo : 0;       // DSYGV: n+i, B not PD
    if (E_host) BSP_HIP(hipMemcpy(E_host, p->d_E, (size_t)nl * n * sizeof(double), hipMemcpyDeviceToHost));
    p->last_l0 = l0; p->last_nl = nl;
    return BSP_OK;
}

extern "C" int bspatom_solve(bspatom_problem *p, int l0, int nl, double *E, int32_t *info)
{
    return solve_impl(p, l0, nl, nullptr, E, info);
}

extern "C" int bspatom_solve_dev(bspatom_problem *p, int l0, int nl, double *E_dev, int32_t *info)
{
    if (!E_dev) return BSP_ERR_ARG;
    return solve_impl(p, l0, nl, E_dev, nullptr, info);
}

extern "C" int bspatom_early_vector_state(const bspatom_problem *p, int32_t *state)
{
    if (!p || !state) return BSP_ERR_ARG;
    *state = !p->pre_early ? 0 : (p->pre_early_ok ? 1 : -1);
    return BSP_OK;
}

extern "C" int bspatom_last_timing(const bspatom_problem *p, double ms[6])
{
    if (!p || !ms) return BSP_ERR_ARG;
    for (int i = 0; i < 6; ++i) ms[i] = p->ms[i];
    return BSP_OK;
}

extern "C" int bspatom_eigvec(bspatom_problem *p, int l, int n0, double *c)
{
    if (!p || !c) return BSP_ERR_ARG;
    const HostSetup &h = p->hs;
    const int n = h.nfun;
    if (l < p->last_l0 || l >= p->last_l0 + p->last_nl || n0 < 1 || n0 > n) return BSP_ERR_ARG;
    BSP_HIP(hipSetDevice(p->device));
    if (l == p->pre_l && n0 == p->pre_n0) {                  // computed beside the bisection of the last solve
        int pinfo = 0;
        BSP_HIP(hipMemcpy(&pinfo, p->d_pinfo, sizeof(int), hipMemcpyDeviceToHost));
        if (pinfo) return BSP_ERR_UNSUPPORTED;              // the inverse iteration broke down (vector of norm 0)
        BSP_HIP(hipMemcpy(c, p->d_pvec, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        return BSP_OK;
    }
    int rcs;
    if ((rcs = ensure_vec_scratch(p))) return rcs;
    const int ch = l - p->last_l0;
    BSP_HIP(hipMemcpyAsync(p->d_chan, &ch, sizeof(int), hipMemcpyHostToDevice, p->st));
    BSP_HIP(hipMemcpyAsync(p->d_Esel, p->d_E + (size_t)ch * n + (n0 - 1), sizeof(double), hipMemcpyDeviceToDevice, p->st));
    BSP_HIP(hipMemsetAsync(p->d_info, 0, sizeof(int), p->st));
    int rc;
    if ((rc = launch_inverse_iteration(n, h.k, 1, p->d_SB, p->d_HB, p->d_chan, p->d_Esel, p->d_vwork, p->d_vec,
                                       p->d_info, p->st))) return rc;
    BSP_HIP(hipMemcpyAsync(c, p->d_vec, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, p->st));
    BSP_HIP(hipStreamSynchronize(p->st));
    return invit_failed(p);
}

struct DevInts {
    int *p = nullptr;
    ~DevInts() { hipFree(p); }
    int alloc(size_t n) { BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(int))); return BSP_OK; }
};

struct DevBuf {
    double *p = nullptr;
    ~DevBuf() { hipFree(p); }
    int alloc(size_t n) { BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(double))); return BSP_OK; }
    int put(const double *h, size_t n) { int rc = alloc(n); if (rc) return rc; BSP_HIP(hipMemcpy(p, h, n * sizeof(double), hipMemcpyHostToDevice)); return BSP_OK; }
    int get(double *h, size_t n) { BSP_HIP(hipMemcpy(h, p, n * sizeof(double), hipMemcpyDeviceToHost)); return BSP_OK; }
};

extern "C" int bspatom_eigvecs(bspatom_problem *p, int l, int n0, int count, double *Z)
{
    if (!p || !Z) return BSP_ERR_ARG;
    const HostSetup &h = p->hs;
    const int n = h.nfun;
    if (l < p->last_l0 || l >= p->last_l0 + p->last_nl || n0 < 1 || count < 1 || n0 + count - 1 > n) return BSP_ERR_ARG;
    BSP_HIP(hipSetDevice(p->device));
    const int ch = l - p->last_l0;
    DevBuf work, vec;
    int rc;
    // chunks bound the scratch: invit_work_doubles(n, k) per vector
    const int chunk = count < 512 ? count : 512;
    if ((rc = work.alloc((size_t)chunk * invit_work_doubles(n, h.k))) || (rc = vec.alloc((size_t)chunk * n))) return rc;
    DevInts chan;
    if ((rc = chan.alloc(chunk))) return rc;
    int *d_chan = chan.p;
    std::vector<int> hc(chunk, ch);
    hipError_t e = hipMemcpy(d_chan, hc.data(), (size_t)chunk * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(p->d_info, 0, sizeof(int), p->st);
    for (int done = 0; e == hipSuccess && rc == BSP_OK && done < count; done += chunk) {
        const int m = (count - done < chunk) ? count - done : chunk;
        rc = launch_inverse_iteration(n, h.k, m, p->d_SB, p->d_HB, d_chan, p->d_E + (size_t)ch * n + (n0 - 1 + done), work.p,
                                      vec.p, p->d_info, p->st);
        if (rc) break;
        e = hipMemcpyAsync(Z + (size_t)done * n, vec.p, (size_t)m * n * sizeof(double), hipMemcpyDeviceToHost, p->st);
        if (e == hipSuccess) e = hipStreamSynchronize(p->st);
    }
    if (rc) return rc;
    BSP_HIP(e);
    return invit_failed(p);
}

extern "C" int bspatom_dipole_elements(bspatom_problem *p, int l_ini, int n0_ini, int l_fin, int n0_fin, int count,
                                       const double a[3], double *D)
{
    if (!p || !a || !D) return BSP_ERR_ARG;
    const HostSetup &h = p->hs;
    const int n = h.nfun;
    const int lo = p->last_l0, hi = p->last_l0 + p->last_nl;
    if (l_ini < lo || l_ini >= hi || l_fin < lo || l_fin >= hi || n0_ini < 1 || n0_ini > n || n0_fin < 1 || count < 1 ||
        n0_fin + count - 1 > n) return BSP_ERR_ARG;
    BSP_HIP(hipSetDevice(p->device));
    int rc;
    if (!p->ptab_ready) {
        if ((rc = launch_point_table(h.nkp, h.k, h.ka, h.nfun, p->d_rt, p->d_aind, p->d_xg, p->d_wg, p->d_vpot,
                                     p->d_ptab, p->d_left, p->d_status, p->st))) return rc;
        p->ptab_ready = true;
    }
    const int chunk = count < 512 ? count : 512;
    DevBuf RB, work, vec, ci, v, dD;
    if ((rc = RB.alloc((size_t)3 * (2 * h.k - 1) * n)) || (rc = work.alloc((size_t)chunk * invit_work_doubles(n, h.k))) ||
        (rc = vec.alloc((size_t)chunk * n)) || (rc = ci.alloc(n)) || (rc = v.alloc(n)) || (rc = dD.alloc(chunk))) return rc;
    DevInts chan;
    if ((rc = chan.alloc(chunk))) return rc;
    int *d_chan = chan.p;
    std::vector<int> hc(chunk, l_ini - lo);
    hipError_t e = hipMemcpy(d_chan, hc.data(), sizeof(int), hipMemcpyHostToDevice);
    // v = (a0 R_r + a1 R_1/r + a2 R_d/dr) c_ini
    if (e == hipSuccess) e = hipMemsetAsync(p->d_info, 0, sizeof(int), p->st);
    if (e == hipSuccess) {
        rc = launch_dipole_bands(n, h.k, h.ka, h.nkp, p->d_ptab, p->d_left, RB.p, p->st);
        if (!rc) rc = launch_inverse_iteration(n, h.k, 1, p->d_SB, p->d_HB, d_chan, p->d_E + (size_t)(l_ini - lo) * n + (n0_ini - 1),
                                               work.p, ci.p, p->d_info, p->st);
        if (!rc) rc = launch_band_apply(n, h.k, RB.p, a, ci.p, v.p, p->st);
        if (!rc) e = hipStreamSynchronize(p->st);
    }
    // D(i) = c_fin(:, n0_fin + i) . v, the final states in chunks
    if (e == hipSuccess && !rc) {
        std::fill(hc.begin(), hc.end(), l_fin - lo);
        e = hipMemcpy(d_chan, hc.data(), (size_t)chunk * sizeof(int), hipMemcpyHostToDevice);
    }
    for (int done = 0; e == hipSuccess && rc == BSP_OK && done < count; done += chunk) {
        const int m = (count - done < chunk) ? count - done : chunk;
        rc = launch_inverse_iteration(n, h.k, m, p->d_SB, p->d_HB, d_chan, p->d_E + (size_t)(l_fin - lo) * n + (n0_fin - 1 + done),
                                      work.p, vec.p, p->d_info, p->st);
        if (!rc) rc = launch_dots(n, m, vec.p, v.p, dD.p, p->st);
        if (rc) break;
        e = hipMemcpyAsync(D + done, dD.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, p->st);
        if (e == hipSuccess) e = hipStreamSynchronize(p->st);
    }
    if (rc) return rc;
    BSP_HIP(e);
    if ((rc = invit_failed(p))) return rc;
    return check_status(p);
}

extern "C" int bspatom_write_wf(bspatom_problem *p, const double *c, int npts, double *r, double *u)
{
    if (!p || !c || !r || !u || npts < 1) return BSP_ERR_ARG;
    const HostSetup &h = p->hs;
    BSP_HIP(hipSetDevice(p->device));
    if (p->wf_cap < npts + 1) {
        hipFree(p->d_wfr); hipFree(p->d_wfu);
        BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_wfr), (size_t)(npts + 1) * sizeof(double)));
        BSP_HIP(hipMalloc(reinterpret_cast<void **>(&p->d_wfu), (size_t)(npts + 1) * sizeof(double)));
        p->wf_cap = npts + 1;
    }
    double *d_c = nullptr;
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&d_c), (size_t)h.nfun * sizeof(double)));
    hipMemcpyAsync(d_c, c, (size_t)h.nfun * sizeof(double), hipMemcpyHostToDevice, p->st);
    hipMemsetAsync(p->d_status, 0, sizeof(int), p->st);
    int rc = launch_wf_tabulate(h.nkp, h.k, h.nfun, p->d_rt, d_c, h.in.ra, h.in.rb, npts, p->d_wfr, p->d_wfu,
                                p->d_status, p->st);
    hipError_t e = hipStreamSynchronize(p->st);
    hipFree(d_c);
    if (rc) return rc;
    BSP_HIP(e);
    if ((rc = check_status(p))) { hipMemset(p->d_status, 0, sizeof(int)); return rc; }
    BSP_HIP(hipMemcpy(r, p->d_wfr, (size_t)(npts + 1) * sizeof(double), hipMemcpyDeviceToHost));
    BSP_HIP(hipMemcpy(u, p->d_wfu, (size_t)(npts + 1) * sizeof(double), hipMemcpyDeviceToHost));
    return BSP_OK;
}

// ---- stage-level entry points ----------------------------------------------------------------

static int need_gpu()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fprintf(stderr, "bspatom: no HIP device (libbspatom has no CPU path)\n");
        return BSP_ERR_NOGPU;
    }
    return BSP_OK;
}

extern "C" int bspatom_stage_gemm(int M, int N, int K, int batch, const double *A, long sAm, long sAk, long bA,
                                  long lenA, const double *B, long sBk, long sBn, long bB, long lenB, double *C,
                                  long sCm, long sCn, long bC, long lenC, double alpha, double beta)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    DevBuf dA, dB, dC;
    if ((rc = dA.put(A, lenA)) || (rc = dB.put(B, lenB)) || (rc = dC.put(C, lenC))) return rc;
    GemmDesc g{};
    g.M = M; g.N = N; g.K = K; g.batch = batch;
    g.A = dA.p; g.sAm = sAm; g.sAk = sAk; g.bA = bA;
    g.B = dB.p; g.sBk = sBk; g.sBn = sBn; g.bB = bB;
    g.C = dC.p; g.sCm = sCm; g.sCn = sCn; g.bC = bC;
    g.alpha = alpha; g.beta = beta; g.lower_only = 0;
    if ((rc = gemm_f64(g, 0))) return rc;
    BSP_HIP(hipDeviceSynchronize());
    return dC.get(C, lenC);
}

extern "C" int bspatom_stage_standard_form(int n, int k, int nl, const double *SB, const double *HB, double *UB,
                                           double *C, int32_t *info)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    const int np = round_up(n, 64);
    DevBuf dSB, dHB, dUB, dr, dY, dC;
    int *dinfo = nullptr;
    if ((rc = dSB.put(SB, (size_t)k * n)) || (rc = dHB.put(HB, (size_t)nl * k * n)) || (rc = dUB.alloc((size_t)k * n)) ||
        (rc = dr.alloc(n)) || (rc = dY.alloc((size_t)nl * np * np)) || (rc = dC.alloc((size_t)nl * np * np))) return rc;
    BSP_HIP(hipMalloc(reinterpret_cast<void **>(&dinfo), sizeof(int)));
    BSP_HIP(hipMemset(dinfo, 0, sizeof(int)));
    BSP_HIP(hipMemset(dY.p, 0, (size_t)nl * np * np * sizeof(double)));
    if ((rc = launch_band_cholesky(n, k, dSB.p, dUB.p, dr.p, dinfo, 0))) return rc;
    if ((rc = launch_standard_form(n, np, k, nl, dHB.p, dUB.p, dr.p, dY.p, dC.p, 0, 1))) return rc;
    BSP_HIP(hipDeviceSynchronize());
    int hi = 0;
    BSP_HIP(hipMemcpy(&hi, dinfo, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(dinfo);
    if (info) *info = hi;
    if (UB && (rc = dUB.get(UB, (size_t)k * n))) return rc;
    return dC.get(C, (size_t)nl * np * np);
}

extern "C" int bspatom_stage_sy2sb(int npad, int batch, const double *A, double *AB)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    if (npad % 64) return BSP_ERR_ARG;
    DevBuf dA, dAB;
    if ((rc = dA.put(A, (size_t)batch * npad * npad)) || (rc = dAB.alloc((size_t)batch * ab_stride(npad)))) return rc;
    void *work = nullptr;
    BSP_HIP(hipMalloc(&work, sy2sb_work_bytes(npad, 64, batch)));
    Sy2sbWork w;
    sy2sb_carve(work, npad, 64, batch, &w);
    rc = sy2sb_run(npad, 64, batch, dA.p, w, 0);
    if (!rc) rc = launch_extract_band(npad, 64, batch, dA.p, dAB.p, 0);
    hipError_t e = hipDeviceSynchronize();
    hipFree(work);
    if (rc) return rc;
    BSP_HIP(e);
    for (int b = 0; b < batch; ++b)      // host layout is dense [batch][npad][128]
        BSP_HIP(hipMemcpy(AB + (size_t)b * npad * 128, dAB.p + b * ab_stride(npad), (size_t)npad * 128 * sizeof(double),
                          hipMemcpyDeviceToHost));
    return BSP_OK;
}

extern "C" int bspatom_stage_panel(int npad, int c0, int batch, double *A, double *V, double *W)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    if (npad % 64 || c0 % 64 || c0 + 128 > npad || batch < 1) return BSP_ERR_ARG;
    const int m = npad - c0 - 64;
    DevBuf dA;
    if ((rc = dA.put(A, (size_t)batch * npad * npad))) return rc;
    void *work = nullptr;
    BSP_HIP(hipMalloc(&work, sy2sb_work_bytes(npad, 64, batch)));
    BSP_HIP(hipMemset(work, 0, sy2sb_work_bytes(npad, 64, batch)));
    Sy2sbWork w;
    sy2sb_carve(work, npad, 64, batch, &w);
    rc = sy2sb_panel_only(npad, c0, batch, dA.p, w, 0);
    hipError_t e = hipDeviceSynchronize();
    if (!rc && e == hipSuccess) {
        for (int b = 0; b < batch && e == hipSuccess; ++b)
            for (int c = 0; c < 64 && e == hipSuccess; ++c) {          // column c of V (first slot of [V | Z | V]) and of W, rows 0 .. m-1
                e = hipMemcpy(V + ((size_t)b * 64 + c) * m, w.buf + (size_t)b * npad * 192 + (size_t)c * npad, (size_t)m * sizeof(double), hipMemcpyDeviceToHost);
                if (e == hipSuccess)
                    e = hipMemcpy(W + ((size_t)b * 64 + c) * m, w.W + (size_t)b * npad * 64 + (size_t)c * npad, (size_t)m * sizeof(double), hipMemcpyDeviceToHost);
            }
    }
    hipFree(work);
    if (rc) return rc;
    BSP_HIP(e);
    return dA.get(A, (size_t)batch * npad * npad);
}

extern "C" int bspatom_stage_sb2st(int n, int npad, int batch, const double *AB, double *d, double *e)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    DevBuf dAB, dd, de;
    if ((rc = dAB.alloc((size_t)batch * ab_stride(npad)))) return rc;
    for (int b = 0; b < batch; ++b)
        BSP_HIP(hipMemcpy(dAB.p + b * ab_stride(npad), AB + (size_t)b * npad * 128, (size_t)npad * 128 * sizeof(double),
                          hipMemcpyHostToDevice));
    if ((rc = dd.alloc((size_t)batch * npad)) ||
        (rc = de.alloc((size_t)batch * npad))) return rc;
    if ((rc = launch_sb2st(n, npad, 64, batch, dAB.p, dd.p, de.p, 0))) return rc;
    BSP_HIP(hipDeviceSynchronize());
    if ((rc = dd.get(d, (size_t)batch * npad))) return rc;
    return de.get(e, (size_t)batch * npad);
}

extern "C" int bspatom_stage_sb2sb(int n, int npad, int batch, double *AB)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    DevBuf dAB;
    if ((rc = dAB.alloc((size_t)batch * ab_stride(npad)))) return rc;
    for (int b = 0; b < batch; ++b)
        BSP_HIP(hipMemcpy(dAB.p + b * ab_stride(npad), AB + (size_t)b * npad * 128, (size_t)npad * 128 * sizeof(double),
                          hipMemcpyHostToDevice));
    if ((rc = launch_sb2sb(n, npad, batch, dAB.p, 0))) return rc;
    BSP_HIP(hipDeviceSynchronize());
    for (int b = 0; b < batch; ++b)
        BSP_HIP(hipMemcpy(AB + (size_t)b * npad * 128, dAB.p + b * ab_stride(npad), (size_t)npad * 128 * sizeof(double),
                          hipMemcpyDeviceToHost));
    return BSP_OK;
}

extern "C" int bspatom_stage_crawford(int n, int k, int nl, const double *SB, const double *HB, double *AB, int32_t *info)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    if (n < 1 || k < 2 || nl < 1 || !SB || !HB || !AB) return BSP_ERR_ARG;
    if (!crawford_supported(n, k)) return BSP_ERR_UNSUPPORTED;
    const int npad = round_up(n, 64);
    DevBuf dSB, dHB, dAB, dW;
    if ((rc = dSB.put(SB, (size_t)k * n)) || (rc = dHB.put(HB, (size_t)nl * k * n)) || (rc = dAB.alloc((size_t)nl * ab_stride(npad))) ||
        (rc = dW.alloc(crawford_work_bytes(n, k, nl) / sizeof(double) + 1))) return rc;
    BSP_HIP(hipMemset(dAB.p, 0, (size_t)nl * ab_stride(npad) * sizeof(double)));
    CrawfordWork cw;
    crawford_carve(dW.p, n, k, nl, &cw);
    if ((rc = crawford_run(n, npad, k, nl, dSB.p, dHB.p, cw, dAB.p, 0))) return rc;
    BSP_HIP(hipDeviceSynchronize());
    int ci = 0;
    BSP_HIP(hipMemcpy(&ci, cw.info, sizeof(int), hipMemcpyDeviceToHost));
    if (info) *info = ci;
    for (int b = 0; b < nl; ++b)
        BSP_HIP(hipMemcpy(AB + (size_t)b * npad * 128, dAB.p + b * ab_stride(npad), (size_t)npad * 128 * sizeof(double),
                          hipMemcpyDeviceToHost));
    return BSP_OK;
}

extern "C" int bspatom_stage_band_eigenvalue(int n, int k, const double *SB, const double *HB, int m, double *lambda)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    if (!SB || !HB || !lambda || n < 1) return BSP_ERR_ARG;
    DevBuf dS, dH, dl;
    if ((rc = dS.put(SB, (size_t)k * n)) || (rc = dH.put(HB, (size_t)k * n)) || (rc = dl.alloc(1))) return rc;
    if ((rc = launch_band_multisect(n, k, dS.p, dH.p, m, dl.p, 0))) return rc;
    BSP_HIP(hipDeviceSynchronize());
    return dl.get(lambda, 1);
}

extern "C" int bspatom_stage_bisect(int n, int batch, const double *d, const double *e, double *w)
{
    int rc;
    if ((rc = need_gpu())) return rc;
    DevBuf dd, de, dw;
    if ((rc = dd.put(d, (size_t)batch * n)) || (rc = de.put(e, (size_t)batch * n)) || (rc = dw.alloc((size_t)batch * n))) return rc;
    if ((rc = launch_bisect(n, n, batch, dd.p, de.p, dw.p, n, 0))) return rc;
    BSP_HIP(hipDeviceSynchronize());
    return dw.get(w, (size_t)batch * n);
}
