// comm.hip -- the ONE exchange of the sharded path below Python: an RCCL all-gather of the ranks' result records.
//
// The l-loop being sharded is reference matrices.f90:242-248; the channels are independent, so the ranks (one process per GPU)
// exchange nothing while they solve and only the spectra travel at the end (north_star: "RCCL over xGMI used only
// to gather the final spectra").  The Python host does this through torch.distributed (bspatom_amd/parallel.py); this file gives
// the Fortran host -- north_star's host language -- the same collective without Python: bspatom_comm_create / _allgather /
// _destroy (include/bspatom.h).
//
// RCCL is loaded at the first bspatom_comm_create (dlopen of librccl.so.1): a one-GPU run never maps it, and a process that has
// torch's copy loaded gets that one.  The ncclUniqueId travels from rank 0 to the others through a file in a directory all ranks
// name, `ncclid.<token>`, written under another name and renamed; <token> identifies the launch (bspatom_run_token: the launcher's
// pid and start time, the same for every rank of a launch and for no other launch), so a file a crashed run left behind is never
// read.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cctype>
#include <cerrno>
#include <cstring>
#include <ctime>
#include <string>
#include "common.h"
#include "../../include/bspatom.h"

namespace {
typedef struct { char internal[128]; } NcclUniqueId;                 // rccl.h: ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128
typedef void *NcclComm;
typedef int (*GetUniqueIdFn)(NcclUniqueId *);
typedef int (*CommInitRankFn)(NcclComm *, int, NcclUniqueId, int);
typedef int (*AllGatherFn)(const void *, void *, size_t, int, NcclComm, hipStream_t);
typedef int (*CommDestroyFn)(NcclComm);
typedef const char *(*GetErrorStringFn)(int);
constexpr int NCCL_FLOAT64 = 8;                                      // ncclDataType_t: ncclFloat64 = ncclDouble = 8

struct Rccl {
    void *h = nullptr;
    GetUniqueIdFn get_id = nullptr;
    CommInitRankFn init_rank = nullptr;
    AllGatherFn all_gather = nullptr;
    CommDestroyFn destroy = nullptr;
    GetErrorStringFn errstr = nullptr;
};
Rccl &rccl()
{
    static Rccl r = [] {
        Rccl v;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            v.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (v.h) break;
        }
        if (!v.h) return v;
        v.get_id = reinterpret_cast<GetUniqueIdFn>(dlsym(v.h, "ncclGetUniqueId"));
        v.init_rank = reinterpret_cast<CommInitRankFn>(dlsym(v.h, "ncclCommInitRank"));
        v.all_gather = reinterpret_cast<AllGatherFn>(dlsym(v.h, "ncclAllGather"));
        v.destroy = reinterpret_cast<CommDestroyFn>(dlsym(v.h, "ncclCommDestroy"));
        v.errstr = reinterpret_cast<GetErrorStringFn>(dlsym(v.h, "ncclGetErrorString"));
        if (!v.get_id || !v.init_rank || !v.all_gather || !v.destroy) { dlclose(v.h); v.h = nullptr; }
        return v;
    }();
    return r;
}
int rccl_fail(const char *what, int rc)
{
    fprintf(stderr, "bspatom: RCCL %s failed: %s\n", what, rccl().errstr ? rccl().errstr(rc) : "?");
    return BSP_ERR_HIP;
}
void nap_ms(int ms)
{
    struct timespec ts = {ms / 1000, (ms % 1000) * 1000000L};
    nanosleep(&ts, nullptr);
}
}  // namespace

struct bspatom_comm {
    int rank = 0, world = 1, calls = 0;
    NcclComm comm = nullptr;
    hipStream_t st = nullptr;
    double *d_send = nullptr, *d_recv = nullptr;
    size_t cap = 0;                                                  // doubles per rank the device buffers hold
    std::string idfile;
};

// "<pid of the launcher>.<its start time in clock ticks since boot>": every rank of one launch is a child of the same
// launcher process (torch.distributed.run's agent, mpirun's daemon, a shell loop), and no other launch ever has this pair.
// TORCHELASTIC_RUN_ID is appended when the launcher sets one that is not the placeholder.
extern "C" int bspatom_run_token(char *buf, int cap)
{
    if (!buf || cap < 8) return BSP_ERR_ARG;
    const long ppid = (long)getppid();
    unsigned long long start = 0;
    char path[64], line[1024];
    snprintf(path, sizeof(path), "/proc/%ld/stat", ppid);
    if (FILE *f = fopen(path, "r")) {
        if (fgets(line, sizeof(line), f)) {
            const char *p = strrchr(line, ')');                      // the command name may hold blanks and parentheses
            int field = 2;
            for (p = p ? p + 1 : nullptr; p && *p; ++p)
                if (*p == ' ' && ++field == 22) { start = strtoull(p + 1, nullptr, 10); break; }
        }
        fclose(f);
    }
    const char *rid = getenv("TORCHELASTIC_RUN_ID");
    std::string t = std::to_string(ppid) + "." + std::to_string(start);
    if (rid && *rid && strcmp(rid, "none") != 0) {
        t += ".";
        for (const char *q = rid; *q && t.size() < 96; ++q) t += (isalnum((unsigned char)*q) ? *q : '_');
    }
    if ((int)t.size() + 1 > cap) return BSP_ERR_ARG;
    memcpy(buf, t.c_str(), t.size() + 1);
    return BSP_OK;
}

extern "C" int bspatom_comm_create(int rank, int world, const char *dir, bspatom_comm **out)
{
    if (!out || world < 1 || rank < 0 || rank >= world || !dir || !*dir) return BSP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BSP_ERR_NOGPU;
    // one process per GPU is what RCCL connects; ranks that share a device (a test box with one GPU) use the file exchange
    if (world > ndev) return BSP_ERR_UNSUPPORTED;
    if (!rccl().h) {
        fprintf(stderr, "bspatom: librccl.so.1 cannot be loaded (%s)\n", dlerror());
        return BSP_ERR_UNSUPPORTED;
    }
    int dev = bsp::process_device();
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    BSP_HIP(hipSetDevice(dev));
    char token[128];
    int rc = bspatom_run_token(token, sizeof(token));
    if (rc) return rc;
    if (mkdir(dir, 0777) != 0 && errno != EEXIST) {
        fprintf(stderr, "bspatom: cannot create %s: %s\n", dir, strerror(errno));
        return BSP_ERR_ARG;
    }
    bspatom_comm *c = new bspatom_comm;
    c->rank = rank; c->world = world;
    c->idfile = std::string(dir) + "/ncclid." + token;
    NcclUniqueId id;
    memset(&id, 0, sizeof(id));
    if (rank == 0) {
        int nr = rccl().get_id(&id);
        if (nr) { delete c; return rccl_fail("ncclGetUniqueId", nr); }
        if (world > 1) {
            const std::string tmp = c->idfile + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, sizeof(id), 1, f) != 1 || fclose(f) != 0 || rename(tmp.c_str(), c->idfile.c_str()) != 0) {
                fprintf(stderr, "bspatom: cannot write %s: %s\n", c->idfile.c_str(), strerror(errno));
                delete c;
                return BSP_ERR_ARG;
            }
        }
    } else {
        bool got = false;
        for (int tries = 0; tries < 6000 && !got; ++tries) {         // two minutes: rank 0 writes it before it solves anything
            if (FILE *f = fopen(c->idfile.c_str(), "rb")) {
                got = fread(&id, sizeof(id), 1, f) == 1;
                fclose(f);
            }
            if (!got) nap_ms(20);
        }
        if (!got) {
            fprintf(stderr, "bspatom: rank %d found no %s (is rank 0 running, with the same directory?)\n", rank, c->idfile.c_str());
            delete c;
            return BSP_ERR_HIP;
        }
    }
    int nr = rccl().init_rank(&c->comm, world, id, rank);
    if (nr) { delete c; return rccl_fail("ncclCommInitRank", nr); }
    if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) { rccl().destroy(c->comm); delete c; return BSP_ERR_HIP; }
    *out = c;
    return BSP_OK;
}

// recv[r * count .. (r + 1) * count) = rank r's send[0 .. count), on every rank; host buffers, staged through device memory.
extern "C" int bspatom_comm_allgather(bspatom_comm *c, const double *send, double *recv, long count)
{
    if (!c || !send || !recv || count <= 0) return BSP_ERR_ARG;
    if ((size_t)count > c->cap) {
        hipFree(c->d_send); hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr; c->cap = 0;
        BSP_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_send), (size_t)count * sizeof(double)));
        BSP_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_recv), (size_t)count * c->world * sizeof(double)));
        c->cap = (size_t)count;
    }
    BSP_HIP(hipMemcpyAsync(c->d_send, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->st));
    const int nr = rccl().all_gather(c->d_send, c->d_recv, (size_t)count, NCCL_FLOAT64, c->comm, c->st);
    if (nr) return rccl_fail("ncclAllGather", nr);
    c->calls += 1;
    BSP_HIP(hipMemcpyAsync(recv, c->d_recv, (size_t)count * c->world * sizeof(double), hipMemcpyDeviceToHost, c->st));
    BSP_HIP(hipStreamSynchronize(c->st));
    return BSP_OK;
}

extern "C" int bspatom_comm_collectives(const bspatom_comm *c) { return c ? c->calls : BSP_ERR_ARG; }

extern "C" void bspatom_comm_destroy(bspatom_comm *c)
{
    if (!c) return;
    if (c->st) hipStreamSynchronize(c->st);
    if (c->comm) rccl().destroy(c->comm);
    if (c->st) hipStreamDestroy(c->st);
    hipFree(c->d_send); hipFree(c->d_recv);
    if (c->rank == 0 && c->world > 1) unlink(c->idfile.c_str());      // every rank has passed ncclCommInitRank by now (the all-gather completed)
    delete c;
}
