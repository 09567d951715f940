// TEST INFRASTRUCTURE -- not part of the product, never loaded unless a test sets MBB_RCCL_LIB.
//
// A stand-in for the four RCCL entry points libmbb_hip.so binds (csrc/mbb_hip.hip load_rccl: ncclGetUniqueId,
// ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString) for ranks that are PROCESSES SHARING ONE
// DEVICE -- which real RCCL refuses (it wants a device per rank) and which is all a one-GPU box offers.  With it the
// N > 1 code of the library (mbb_comm_init, mbb_lnlike_allgather's rank-major offsets and in-place send pointer,
// mbb_allgather_f64, the sampler's in-place gather of moved rows, of the chain and of the counts) runs with
// nranks = 2, 3, ... on one GPU and is held bitwise to the unsharded evaluation
// (tests/test_zz_gpu_processes.py::test_rccl_paths_with_several_ranks_on_one_gpu).
//
// How: the unique id names a POSIX shared-memory segment; every rank allocates a staging buffer on the device and
// publishes its hipIpc handle there; an all-gather is  send -> own staging (on the caller's stream), wait, barrier,
// every peer's staging -> recv + r * bytes (on the caller's stream), wait, barrier.  It blocks the host where RCCL
// would not -- the results are the same, the timing is not, and nothing here is ever timed.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kStage = 8u << 20;

struct Seg {
    std::atomic<uint64_t> bar;
    std::atomic<int> ready[kMaxRanks];
    hipIpcMemHandle_t handle[kMaxRanks];
    std::atomic<int> broken;
};

struct Comm {
    Seg *seg = nullptr;
    char name[128];
    int n = 0, rank = 0;
    uint64_t barriers = 0;
    char *stage[kMaxRanks] = {};
};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

// every rank arrives once per barrier; nobody waits longer than 60 s for a rank that is gone
int barrier(Comm *c)
{
    const uint64_t want = ++c->barriers * (uint64_t)c->n;
    c->seg->bar.fetch_add(1, std::memory_order_acq_rel);
    const double t_end = now_s() + 60.0;
    while (c->seg->bar.load(std::memory_order_acquire) < want) {
        if (c->seg->broken.load(std::memory_order_relaxed) || now_s() > t_end) {
            c->seg->broken.store(1, std::memory_order_relaxed);
            return 6;           // ncclRemoteError
        }
        usleep(20);
    }
    return 0;
}

size_t type_size(int t)
{
    switch (t) {                // nccl.h's ncclDataType_t
    case 0: case 1: return 1;   // int8 / uint8
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: case 9: return 2;
    default: return 0;
    }
}
}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

const char *ncclGetErrorString(int r)
{
    switch (r) {
    case 0: return "no error (stand-in)";
    case 1: return "unhandled HIP error (stand-in)";
    case 4: return "invalid argument (stand-in)";
    case 6: return "a peer never arrived (stand-in)";
    default: return "error (stand-in)";
    }
}

// (a segment whose communicator was never destroyed -- a rank that died -- goes with the process that made it)
static char g_made[16][128];
static std::atomic<int> g_nmade{0};
static void unlink_made()
{
    for (int i = 0; i < g_nmade.load() && i < 16; ++i)
        if (g_made[i][0]) shm_unlink(g_made[i]);
}

int ncclGetUniqueId(ncclUniqueId *id)
{
    static std::atomic<int> serial{0};
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/mbb_rccl_standin_%d_%d_%ld", (int)getpid(), serial.fetch_add(1),
             (long)(now_s() * 1e6));
    int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, sizeof(Seg)) != 0) { close(fd); shm_unlink(id->internal); return 2; }
    close(fd);                  // (zero-filled: every counter starts at 0)
    const int k = g_nmade.fetch_add(1);
    if (k == 0) atexit(unlink_made);
    if (k < 16) memcpy(g_made[k], id->internal, 128);
    return 0;
}

int ncclCommInitRank(void **out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
    id.internal[127] = 0;
    if (strncmp(id.internal, "/mbb_rccl_standin_", 18) != 0) return 4;
    int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) return 2;
    void *m = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 2;
    Comm *c = new Comm;
    c->seg = (Seg *)m; c->n = nranks; c->rank = rank;
    memcpy(c->name, id.internal, 128);
    auto give_up = [&](int code) {              // whatever was set up so far goes; the peers' barriers time out or see `broken`
        c->seg->broken.store(1, std::memory_order_relaxed);
        for (int r = 0; r < nranks; ++r)
            if (r != rank && c->stage[r]) (void)hipIpcCloseMemHandle(c->stage[r]);
        if (c->stage[rank]) (void)hipFree(c->stage[rank]);
        munmap(c->seg, sizeof(Seg));
        delete c;
        return code;
    };
    if (hipMalloc((void **)&c->stage[rank], kStage) != hipSuccess) return give_up(1);
    if (nranks > 1) {
        if (hipIpcGetMemHandle(&c->seg->handle[rank], c->stage[rank]) != hipSuccess) return give_up(1);
        c->seg->ready[rank].store(1, std::memory_order_release);
        int rc = barrier(c);
        if (rc) return give_up(rc);
        for (int r = 0; r < nranks; ++r) {
            if (r == rank) continue;
            if (hipIpcOpenMemHandle((void **)&c->stage[r], c->seg->handle[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                c->stage[r] = nullptr;
                return give_up(1);
            }
        }
        if ((rc = barrier(c))) return give_up(rc);
    }
    *out = c;
    return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * type_size(dtype);
    if (!c || !send || !recv || type_size(dtype) == 0) return 4;
    for (size_t off = 0; off < bytes; off += kStage) {
        const size_t nb = bytes - off < kStage ? bytes - off : kStage;
        if (hipMemcpyAsync(c->stage[c->rank], (const char *)send + off, nb, hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return 1;
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;
        int rc = c->n > 1 ? barrier(c) : 0;
        if (rc) return rc;
        for (int r = 0; r < c->n; ++r)
            if (hipMemcpyAsync((char *)recv + (size_t)r * bytes + off, c->stage[r], nb, hipMemcpyDeviceToDevice, stream) !=
                hipSuccess)
                return 1;
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;
        if (c->n > 1 && (rc = barrier(c))) return rc;       // nobody refills its staging under a reader
    }
    return 0;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return 0;
    if (c->n > 1) (void)barrier(c);
    for (int r = 0; r < c->n; ++r)
        if (r != c->rank && c->stage[r]) (void)hipIpcCloseMemHandle(c->stage[r]);
    if (c->n > 1) (void)barrier(c);                         // (handles closed before the owner's memory goes)
    if (c->stage[c->rank]) (void)hipFree(c->stage[c->rank]);
    if (c->rank == 0) {
        shm_unlink(c->name);
        for (int i = 0; i < g_nmade.load() && i < 16; ++i)
            if (!strncmp(g_made[i], c->name, 128)) g_made[i][0] = 0;
    }
    munmap(c->seg, sizeof(Seg));
    delete c;
    return 0;
}

}  // extern "C"
