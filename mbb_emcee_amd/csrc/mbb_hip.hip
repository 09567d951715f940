// mbb_hip.hip -- host side and C-ABI of the MI355X likelihood hot path (gfx950).
//
// The kernels are in mbb_kernels.hip.h (one fused kernel per model variant
// evaluates, for a batch of walkers, prologue -> f_nu on every passband sample ->
// band fluxes -> chi-square / covariance form -> limits and priors -> lnL, i.e. n
// calls of the reference's likelihood.__call__, likelihood.py:790-834, in one
// launch).  This file owns contexts, device buffers, launch geometry, the sampler
// driver and the RCCL plumbing.  See include/mbb_hip.h for the boundary and
// DESIGN.md for the data layout.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <functional>
#include <string>
#include <type_traits>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mbb_hip.h"
#include "mbb_device.hip.h"
#include "mbb_kernels.hip.h"

// SMODE 6 is instantiated in mbb_flow.hip (its own compiler flags)
#define MBB_FLOW_EXT(OT, NA)                                                    \
    extern template __global__ void k_lnlike<OT, NA, 6, false>(const LikeArgs); \
    extern template __global__ void k_lnlike<OT, NA, 6, true>(const LikeArgs);
MBB_FLOW_EXT(false, false)
MBB_FLOW_EXT(false, true)
MBB_FLOW_EXT(true, false)
MBB_FLOW_EXT(true, true)
#undef MBB_FLOW_EXT
// ... and so is k_flowm, sampler form 7 (mbb_flowm.hip.h; only its LDS plan is needed here)
template <bool OPTHIN, bool NOALPHA, bool STAGE, int NP>
__global__ void k_flowm(const LikeArgs a);
#define MBB_FLOWM_EXT(OT, NA)                                                   \
    extern template __global__ void k_flowm<OT, NA, false, 1>(const LikeArgs); \
    extern template __global__ void k_flowm<OT, NA, true, 1>(const LikeArgs);
MBB_FLOWM_EXT(false, false)
MBB_FLOWM_EXT(false, true)
MBB_FLOWM_EXT(true, false)
MBB_FLOWM_EXT(true, true)
#undef MBB_FLOWM_EXT
// ... and k_flowa, form 9: the same resident run with the constructor a half-step ahead (mbb_flowa.hip.h)
template <bool OPTHIN, bool NOALPHA, bool STAGE>
__global__ void k_flowa(const LikeArgs a);
#define MBB_FLOWA_EXT(OT, NA)                                               \
    extern template __global__ void k_flowa<OT, NA, false>(const LikeArgs); \
    extern template __global__ void k_flowa<OT, NA, true>(const LikeArgs);
MBB_FLOWA_EXT(false, false)
MBB_FLOWA_EXT(false, true)
MBB_FLOWA_EXT(true, false)
MBB_FLOWA_EXT(true, true)
#undef MBB_FLOWA_EXT
static size_t flowa_lds_bytes(size_t nb, size_t npart, bool cov_in_lds, size_t W)        // = flowa_lds() of mbb_flowa.hip.h
{
    return 2 * W * (sizeof(WalkerK) + 8 * npart + 8 * 10) + 8 * W * nb + 8 * 2 * W * 8 + 16 * nb +
           (cov_in_lds ? 8 * nb * nb : 0) + 8 * (nb + 2) + 192 + 64;
}
// ... and k_serve, the likelihood of given rows as a kernel that stays resident between boundary calls (mbb_serve.hip.h)
template <bool OPTHIN, bool NOALPHA, bool STAGE, bool OVL>
__global__ void k_serve(const LikeArgs a);
#define MBB_SERVE_EXT(OT, NA)                                                      \
    extern template __global__ void k_serve<OT, NA, false, false>(const LikeArgs); \
    extern template __global__ void k_serve<OT, NA, true, false>(const LikeArgs);  \
    extern template __global__ void k_serve<OT, NA, false, true>(const LikeArgs);  \
    extern template __global__ void k_serve<OT, NA, true, true>(const LikeArgs);
MBB_SERVE_EXT(false, false)
MBB_SERVE_EXT(false, true)
MBB_SERVE_EXT(true, false)
MBB_SERVE_EXT(true, true)
#undef MBB_SERVE_EXT
constexpr unsigned long long kServeQuitHost = 0xffffull;
constexpr int kServePasses = 2;               // rows a workgroup of a resident server takes of one request, at most
static size_t serve_lds_bytes(size_t nb, size_t npart, bool cov_in_lds)                  // = serve_lds() of mbb_serve.hip.h
{
    return sizeof(WalkerK) + 8 * npart + 8 * nb + 16 + 16 * nb + (cov_in_lds ? 8 * nb * nb : 0) + 8 * (nb + 2) + 64;
}
constexpr int kFrMaxWHost = 8;                  // walkers of each half a workgroup of the resident form may own (mbb_flowa.hip.h)
constexpr int kFmPropHost = 16;
static size_t flowm_lds_bytes(size_t nb, size_t npart, bool cov_in_lds, size_t np = 1)   // = flowm_lds() of mbb_flowm.hip.h
{
    return np * 4 * sizeof(WalkerK) + 8 * (np * 4 * npart + 2 * nb + np * 4 * kFmPropHost + 2 * nb + (cov_in_lds ? nb * nb : 0)) +
           8 * (nb + 2) + 8 * (3 * 64) + 128 * np + 32;
}
#include "mbb_host_tables.h"
#include "mbb_registry.h"

static_assert(kPolyBDoubles == mbbh::kPolyBCount * mbbh::kPolyStride && mbbh::kPolyStride == mbbm::kPolyStride, "poly table size");
static_assert(kPolyCDoubles == mbbh::kPolyCCount * mbbh::kPolyStride, "poly table size");
static_assert(sizeof(mbbh::Unit) == sizeof(int4) && sizeof(mbbh::SlotRange) == sizeof(int2), "layout");


// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (code == MBB_ERR_HIP)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIPCHK(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return fail(MBB_ERR_HIP, #call, e_);     \
    } while (0)

typedef void *ncclComm_t_;
struct UniqueId { char internal[128]; };
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(ncclComm_t_ *, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t_) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;

static int load_rccl()
{
    if (g_rccl.handle) return MBB_OK;
    // A copy that is in the process already first (by its soname): a launcher that has imported torch brings torch's
    // own librccl / libamdhip64 / libhsa-runtime64 along, this library then runs on THAT HIP runtime (same soname), and
    // a second RCCL from the system path would bring a second HSA runtime with it.
    // MBB_RCCL_LIB names the library outright: a site's own RCCL build -- and what the one-GPU tests use to run the
    // N > 1 code with ranks that share a device (tests/rccl_standin/).  A path that does not load is an error, never
    // a reason to look elsewhere.
    void *h = nullptr;
    if (const char *named = getenv("MBB_RCCL_LIB")) {
        if (*named && !(h = dlopen(named, RTLD_NOW | RTLD_LOCAL)))
            return fail(MBB_ERR_RCCL, "cannot load the library MBB_RCCL_LIB names");
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MBB_ERR_RCCL, "cannot load librccl.so");
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(ncclComm_t_ *, int, UniqueId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, ncclComm_t_, hipStream_t))
        dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(ncclComm_t_))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail(MBB_ERR_RCCL, "librccl.so lacks the expected nccl* symbols");
    g_rccl.handle = h;
    return MBB_OK;
}

struct mbb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int cu_count = 256, xcds = 8;
    // model
    int opthin = 0, noalpha = 0;
    double wavenorm = 500.0;
    // bands
    int nb = 0, nseg = 0, nunit = 0, npart = 0, nchunk = 0, nq = 0;
    long t_prep_ns = 0, t_launch_ns = 0, t_wait_ns = 0;   // phases of the last mbb_lnlike_batch
    int simd_chunks[4] = {0, 0, 0, 0};   // chunks dealt to each SIMD position by the unit table
    double *d_nu = nullptr, *d_lnnu = nullptr, *d_wt = nullptr;
    double *d_poly_b = nullptr, *d_poly_c = nullptr;     // piecewise polynomials of the sample loop
    int2 *d_band_rng = nullptr;
    int32_t *d_tail_slot = nullptr;
    int4 *d_unit_tab = nullptr;
    // data
    int data_nb = 0, has_cov = 0, nsrc = 1;
    double *d_flux = nullptr, *d_ivar = nullptr, *d_invcov = nullptr;
    // limits and priors (likelihood.py:73, :83-85 defaults)
    double lowlim[5] = {1, 0.1, 1, 0.1, 1e-3};
    double uplim[6] = {INFINITY, 20.0, INFINITY, 20.0, INFINITY, INFINITY};
    uint32_t has_uplim = (1u << 1) | (1u << 3);
    double gmean[6] = {0, 0, 0, 0, 0, 0};
    double givar[6] = {1, 1, 1, 1, 1, 1};
    uint32_t has_gprior = 0;
    // staging for the host-in / host-out path
    size_t cap = 0, cap_flux = 0;
    double *d_pars = nullptr, *d_lnl = nullptr, *d_mflux = nullptr;
    int32_t *d_status = nullptr;
    double *h_pars = nullptr, *h_lnl = nullptr, *h_mflux = nullptr;   // pinned
    double *w_pars = nullptr;    // device memory the host writes through the PCIe BAR (fine-grained), or null
    double *dv_pars = nullptr, *dv_lnl = nullptr, *dv_mflux = nullptr;   // the pinned blocks as the device addresses them
    int32_t *dv_status = nullptr;
    // the served boundary (k_serve, mbb_serve.hip.h): a kernel that stays resident between a host-driven sampler's calls
    unsigned long long *w_door = nullptr;     // its doorbell: a word of fine-grained device memory the host writes through the BAR
    double *h_srv = nullptr, *dv_srv = nullptr;   // pinned: a server's result records [row]{lnl, status as a 64-bit integer}
    size_t srv_cap = 0;
    unsigned long long *h_gone = nullptr;     // pinned: the number of the last request a server saw before it left (0: still there)
    // The serve state below is a context's own -- except that a SIBLING context coming to the device tells this one's
    // server to leave (yield_server), possibly from another thread (ctypes and the _mbbfast extension release the GIL).
    // srv_mu guards it: held by the owner for the whole of mbb_lnlike_call and of use(), taken by a visitor (try_lock
    // under g_dev_mutex, so that a context cannot be destroyed between being found and being locked) around serve_stop.
    // Lock order: ctx.srv_mu, then g_dev_mutex; a thread waits for another context's srv_mu only while its own context
    // is NOT the device's server, and the server's owner never waits for a sibling: no cycle.
    std::mutex srv_mu;
    bool serving = false;
    unsigned long long srv_seq = 0;           // requests so far (the doorbell word is request number << 16 | rows)
    int srv_hot = 0;                          // boundary calls in a row with nothing else in between
    long srv_need = 0;                        // ... of which a server is started (0: opt_serve_after).  Doubled, up to 64, every
                                              // time a sibling context's visit sends this context's server away, back to
                                              // opt_serve_after once a server has answered 256 requests: two likelihoods used
                                              // in turns must not spend their time starting and stopping kernels
    long srv_run = 0;                         // requests the present server has answered
    int srv_strikes = 0;                      // servers that had to be given up in a row; three rest the feature
    long srv_rest = 0, srv_rests = 0;         // boundary calls the feature still rests for; how often it has had to
    long srv_requests = 0, srv_fallbacks = 0;
    long opt_serve = 1;                       // 0: every boundary call is a launch; 2: a server even beside other contexts (tests)
    long opt_serve_after = 3;                 // boundary calls in a row before a server is started
    long opt_serve_idle_us = 1000;            // the server leaves after this long without a request (by the clock all CUs share)
    long opt_serve_budget_us = 400;           // the host gives a served request this long before it falls back to a launch
    int opt_serve_grid = 0;                   // workgroups of a resident server (0: as many as the calls have rows, in eights)
    int srv_grid = 0;                         // ... of the one that is resident
    int srv_want = 0;                         // the most rows a call of this context has had (in eights): a server is started that wide
    long srv_resizes = 0;                     // servers that left because they were the wrong width
    int srv_busy = 0;                         // busy processes beside this one, as the latest boundary call counted them
    int opt_serve_prefetch = 32;              // record lines asked for ahead of the host's scan once the first record has turned (0: none)
    long opt_serve_lease_us = 50000;          // a server is sent away after this long in one go (0: never): processes this library
                                              // cannot see (other containers, other programs) get the CUs at least that often
    long srv_t0_ns = 0;                       // when the present server was started
    uint32_t reg_key = 0;                     // the device's name in the cross-process registry (mbb_registry.h): PCI domain:bus:device
    long srv_lease_yields = 0;                // servers sent away because their lease was up
    long srv_peer_yields = 0;                 // servers not started, or sent away, because another process is on the device
    unsigned long long buf_gen = 1;   // mbb_boundary_generation: bumped whenever the blocks below are freed and made anew
    double *call_in = nullptr;   // mbb_boundary_buffers: where the caller writes its rows (w_pars or h_pars)
    size_t call_cap = 0;         // ... and the capacity that answer was given for
    hipFunction_t mod_fn[80] = {};   // launch_api 1: the kernels' module handles, by variant (32 k_lnlike, 8 k_flowm, 8 + 8 k_flowa (slots 40..55), 16 k_serve)
    hipEvent_t *launch_ev = nullptr; // != nullptr: two events to record right before and right behind the next launch
    long opt_launch_api = 1;     // 1 hipModuleLaunchKernel with a packed argument buffer (-0.2 us per call, profiles/r04/boundary_breakdown.txt); 0 hipLaunchKernel
    double *d_gather = nullptr, *h_gather = nullptr;   // sharded boundary: every rank's lnprob, device / pinned landing place
    size_t gather_cap = 0;
    int large_bar = -1;
    int32_t *h_status = nullptr;
    // scratch for SED-level calls
    size_t sed_cap = 0, sed_out_cap = 0;
    double *d_sed_pars = nullptr, *d_sed_out = nullptr;
    int32_t *d_sed_status = nullptr;
    WalkerK *d_sed_wk = nullptr;
    // options
    long opt_wpb = 0, opt_threads = 0, opt_seg_chunks = 4, opt_debug = 0;
    long opt_prepass = -1, last_prepass = 0;   // big batches: the constructors by k_walker_pre, a lane per walker (launch_lnlike)
    double *d_pre = nullptr;                   // ... its records
    size_t pre_cap = 0;                        // (doubles)
    long opt_zero_copy = 1;   // host path: kernel reads/writes pinned host memory (26 vs 34 us per call)
    long opt_stage = -1;      // -1 auto, 0 never, 1 whenever the tables fit in LDS
    long opt_roof_wgs = 0, opt_roof_threads = 0;   // measurement: geometry of mbb_roof_probe
    long opt_vranks = 0;      // testing: run a sampler as this many shards on one GPU
    long opt_bar_params = 1;  // host path: write the parameter rows into device memory through the BAR
    long opt_pack_tails = 1;  // band leftovers share chunks, one row of 16 lanes each (0: a chunk per leftover)
    long opt_spin_budget = 20000000L;   // polls of the result slots before falling back to the stream
    long last_watch_seen = -1;          // last mbb_lnlike_batch: 1 results seen by the watch, 0 fell back, -1 no watch
    long opt_spin = 2;        // 0 block on the stream; 1 poll hipStreamQuery (measured: no gain);
                              // 2 watch the result slots in pinned memory (zero-copy batches <= 8192 rows)
    long last_stage = 0;
    long opt_lookahead = 1;   // single-GPU sampler runs prepare the next half-step's proposals ahead of the decisions they depend on
    unsigned long long flow_serial = 0;   // one-launch sampler runs started on this context
    long opt_serve_overlap = 1;    // the served kernel starts a row's quadrature beside its constructor (0: one after the other)
    long opt_flow_spin_log2 = 0;   // one-launch run: log2 of the polls before a wait gives up (0: the kernel's 22)
    hipEvent_t ev_timed[2] = {nullptr, nullptr};   // mbb_sampler_advance_timed
    long flow_fallbacks = 0;       // one-launch runs that timed out and were redone as a launch train
    // A give-up is a property of the moment (a co-tenant holding CUs), not of the context: the next run
    // takes the one-launch form again.  Only kFlowStrikes give-ups in a row rest it, for kFlowRest runs.
    long flow_strikes = 0, flow_rest = 0;
    long opt_flow_min_steps = 2;   // runs shorter than this take the launch train (a one-launch run costs
                                   // ~14 us beside its steps: 16.9 us for one step against 15.7)
    long opt_xflow = 1;       // ... also for a sharded ensemble with the one-hop exchange (SMODE 6)
    long opt_flow = 1;        // 1: ... as ONE launch per run, the half-steps handing over row by row (SMODE 5)
    long opt_flowm = 1;       // 1: ... with the quadrature of both candidates running ahead too (k_flowm, form 7)
    long opt_flowr = 1;       // 1: ensembles beyond one pair of walkers per CU run as ONE resident launch too, several walkers per
                              // workgroup, the SED constructor a half-step ahead for both outcomes of each partner's pending move
                              // (k_flowa, form 9); 0: off; 2: every eligible ensemble takes it
    long opt_flowr_walkers = 0;   // walkers per workgroup and half of that form (0: the host's choice, ceil(half / CUs))
                              // (ensembles of 258-512 walkers on 256 CUs); 1, 2 force it (testing)
    long opt_la_waves = 0;
    long opt_la_rows = 0;     // ... candidates per wave of the workgroups that do so (1, 2 or 4)
    size_t lds_granted[112] = {};   // dynamic-LDS ceiling already requested, per kernel variant
    long last_wpb = 0, last_threads = 0, last_grid = 0, last_smem = 0, last_smode = 0, last_ahead = 0;
    unsigned long long *d_stamps = nullptr;   // diagnostic build only
    // rccl
    ncclComm_t_ comm = nullptr;
    int nranks = 1, rank = 0;
    // one-hop exchange (mbb_xchg_*): one fine-grained allocation per rank, mapped by all peers:
    // [flags: 16 x u64][arrival counter][state rows: xcap x 6 doubles]
    struct Xchg {
        int n = 0, rank = 0, connected = 0;
        int users = 0;                            // samplers whose state rows live in this buffer
        size_t cap_rows = 0;
        unsigned char *base = nullptr;            // own allocation
        unsigned char *peer[16] = {};             // every rank's allocation as mapped here (own = base)
        XchgArgs *d_args = nullptr;               // the kernel's view (device memory)
        FlowX *d_flowx = nullptr;                 // the one-launch sharded run's view of every rank's copy (device memory)
        unsigned long long flow_run = 0;          // one-launch sharded runs so far (the same on every rank)
        unsigned long long seq = 0;               // launches posted so far
        long long spin_max = 4000000;
        static constexpr size_t kHeader = 256;
        double *pos6(int r) const { return reinterpret_cast<double *>(peer[r] + kHeader); }
        // behind the rows: the state of a one-launch run (FlowView, mbb_kernels.hip.h), sized for cap_rows
        double *flow(int r) const { return reinterpret_cast<double *>(peer[r] + kHeader + cap_rows * 6 * sizeof(double)); }
        unsigned long long *flags(int r) const { return reinterpret_cast<unsigned long long *>(peer[r]); }
        unsigned int *count() const { return reinterpret_cast<unsigned int *>(base + 128); }
    } x;
};

static int serve_stop(mbb_ctx *c);

// Every entry point but the boundary call comes through here: a resident server (k_serve) is told to leave first --
// it holds the CUs, and its arguments were fixed at its launch -- and the streak of boundary calls ends.
static int yield_server(mbb_ctx *c);
static int use(mbb_ctx *c)
{
    if (!c) return fail(MBB_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> own(c->srv_mu);
    c->srv_hot = 0;
    if (c->serving) return serve_stop(c);
    return yield_server(c);
}

extern "C" const char *mbb_last_error(void) { return g_err.c_str(); }

extern "C" int mbb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

struct DeviceStatics {
    bool ready = false;
    int cu_count = 0, xcds = 8;
    double *d_poly_b = nullptr, *d_poly_c = nullptr;
    std::vector<hipStream_t> idle_streams;
    std::atomic<int> live{0};        // contexts of this process on the device
    struct mbb_ctx *server = nullptr;    // the context whose k_serve is resident on the device, if any (g_dev_mutex)
    struct mbb_ctx *last_user = nullptr; // the context that came to the device last: "calls in a row" are a context's own
};
static std::mutex g_dev_mutex;
static DeviceStatics g_dev[64];

// Another context of this process has a server resident on c's device: it holds the CUs, so whatever c is about to put on
// its stream would wait for it -- it is told to leave first (and starts again after its own next few calls in a row).
// Called with c->srv_mu held and c NOT serving.
static int yield_server(mbb_ctx *c)
{
    if (c->device < 0 || c->device >= 64) return MBB_OK;
    for (long tries = 0;; ++tries) {
        mbb_ctx *o;
        {
            std::lock_guard<std::mutex> lk(g_dev_mutex);
            o = g_dev[c->device].server;
            if (g_dev[c->device].last_user != c) {
                // (another context was here in between: this one's boundary calls are not "in a row" any more)
                g_dev[c->device].last_user = c;
                c->srv_hot = 0;
            }
            if (!o || o == c) return MBB_OK;
            // found and locked inside one critical section: mbb_ctx_destroy takes the server's name off the device
            // under g_dev_mutex after its own serve_stop, so `o` is alive here and stays so while its srv_mu is held
            if (!o->srv_mu.try_lock()) o = nullptr;
        }
        if (!o) {
            // its owner is inside a call (at most the serve budget, ~0.4 ms): come back
            if ((tries & 15) == 15) std::this_thread::yield(); else __builtin_ia32_pause();
            continue;
        }
        std::lock_guard<std::mutex> theirs(o->srv_mu, std::adopt_lock);
        if (!o->serving) return MBB_OK;            // (it left by itself, or its owner sent it away, in between)
        o->srv_hot = 0;
        o->srv_need = std::min<long>(64, 2 * (o->srv_need > 0 ? o->srv_need : o->opt_serve_after));
        return serve_stop(o);
    }
}

extern "C" int mbb_ctx_create(int device, mbb_ctx **out)
{
    if (!out) return fail(MBB_ERR_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(MBB_ERR_ARG, "no such HIP device");
    HIPCHK(hipSetDevice(device));
    // What is the same for every context of a device is made once per process and shared: the device's CU
    // count and the two polynomial tables of the sample loop (read-only; never freed).  A catalogue loop makes
    // a context per source: this took a new one from 6.2 ms to the stream and a few allocations
    // (tools/probe_first_fit.py).
    if (device >= 64) return fail(MBB_ERR_ARG, "device number out of range");
    {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        DeviceStatics &d = g_dev[device];
        if (!d.ready) {
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, device));
            d.cu_count = prop.multiProcessorCount;
            int xcc = 0;                  // (MI355X: 8 XCDs of 32 CUs; the share of a device is dealt in whole rows of them)
            if (hipDeviceGetAttribute(&xcc, hipDeviceAttributeNumberOfXccs, device) != hipSuccess) (void)hipGetLastError();
            d.xcds = xcc > 0 && d.cu_count % xcc == 0 ? xcc : 8;
            std::vector<double> pb, pc;
            mbbh::build_poly_tables(pb, pc);
            HIPCHK(hipMalloc((void **)&d.d_poly_b, pb.size() * sizeof(double)));
            HIPCHK(hipMalloc((void **)&d.d_poly_c, pc.size() * sizeof(double)));
            HIPCHK(hipMemcpy(d.d_poly_b, pb.data(), pb.size() * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d.d_poly_c, pc.data(), pc.size() * sizeof(double), hipMemcpyHostToDevice));
            d.ready = true;
        }
    }
    mbb_ctx *c = new mbb_ctx();
    c->device = device;
    c->cu_count = g_dev[device].cu_count;
    c->xcds = g_dev[device].xcds;
    c->d_poly_b = g_dev[device].d_poly_b;
    c->d_poly_c = g_dev[device].d_poly_c;
    // A stream costs ~5 ms to create (a hardware queue): the streams of contexts that have gone are kept, idle,
    // for the contexts to come.
    {
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        std::vector<hipStream_t> &pool = g_dev[device].idle_streams;
        if (!pool.empty()) { c->stream = pool.back(); pool.pop_back(); }
    }
    if (!c->stream) {
        // The runtime deals the streams of one priority class to four hardware queues in turn, and a kernel that lands behind
        // a RESIDENT one on its queue (k_serve between a sampler's calls) does not start until that one leaves -- measured:
        // tools/lat_two_residents.hip, 20 ms (the resident kernel's own limit) instead of 50 us for every fourth stream.
        // Streams of another class never share a queue with it.  So the library's streams are of the class applications use
        // least, the least urgent one: whatever the application (torch, a prior of its own on the GPU) puts on its streams is
        // not held up by a resident server.  (By itself the class costs nothing: M1 and M2 the same at any priority.)
        int least = 0, most = 0;
        hipError_t es = hipDeviceGetStreamPriorityRange(&least, &most);
        if (es == hipSuccess && least != most) es = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, least);
        else es = hipErrorNotSupported;
        if (es != hipSuccess) {
            (void)hipGetLastError();
            c->stream = nullptr;
            es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        }
        if (es != hipSuccess) { delete c; return fail(MBB_ERR_HIP, "hipStreamCreateWithFlags", es); }
    }
    ++g_dev[device].live;
    {
        // the device's name for other processes: its PCI address (the HIP ordinal is renumbered by HIP_VISIBLE_DEVICES)
        int dom = 0, bus = 0, dv = 0;
        if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess) dom = 0;
        if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess) bus = device;
        if (hipDeviceGetAttribute(&dv, hipDeviceAttributePciDeviceId, device) != hipSuccess) dv = 0;
        (void)hipGetLastError();
        c->reg_key = 0x80000000u | ((uint32_t)(dom & 0x7fff) << 16) | ((uint32_t)(bus & 0xff) << 8) | (uint32_t)(dv & 0xff);
        mbbh::registry_join(c->reg_key);
    }
    *out = c;
    return MBB_OK;
}

static void free_dev(void *p) { if (p) (void)hipFree(p); }
static int xchg_free(mbb_ctx *c);
static void free_host(void *p) { if (p) (void)hipHostFree(p); }

extern "C" void mbb_ctx_destroy(mbb_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    {
        // (a sibling's visit -- yield_server on another thread -- is either over or has not found this context yet)
        std::lock_guard<std::mutex> own(c->srv_mu);
        if (c->serving) (void)serve_stop(c);
    }
    if (c->device >= 0 && c->device < 64) {
        --g_dev[c->device].live;
        std::lock_guard<std::mutex> lk(g_dev_mutex);
        if (g_dev[c->device].last_user == c) g_dev[c->device].last_user = nullptr;
        if (g_dev[c->device].server == c) g_dev[c->device].server = nullptr;
    }
    mbbh::registry_leave(c->reg_key);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->x.base) (void)xchg_free(c);
    free_dev(c->d_nu); free_dev(c->d_lnnu); free_dev(c->d_wt);
    // (d_poly_b / d_poly_c belong to the device, not to the context: mbb_ctx_create)
    free_dev(c->d_unit_tab); free_dev(c->d_band_rng); free_dev(c->d_tail_slot);
    free_dev(c->d_flux); free_dev(c->d_ivar); free_dev(c->d_invcov);
    free_dev(c->d_pars); free_dev(c->d_lnl); free_dev(c->d_mflux); free_dev(c->d_status);
    free_host(c->h_pars); free_host(c->h_lnl); free_host(c->h_mflux); free_host(c->h_status);
    free_dev(c->w_pars);
    free_dev(c->w_door); free_host(c->h_gone); free_host(c->h_srv);
    free_dev(c->d_gather); free_host(c->h_gather);
    free_dev(c->d_pre);
    free_dev(c->d_sed_pars); free_dev(c->d_sed_out); free_dev(c->d_sed_status);
    free_dev(c->d_sed_wk);
    for (int i = 0; i < 2; ++i)
        if (c->ev_timed[i]) (void)hipEventDestroy(c->ev_timed[i]);
    if (c->stream) {
        // idle (synchronised above): kept for the next context of this device, up to a handful
        std::lock_guard<std::mutex> lock(g_dev_mutex);
        if (c->device >= 0 && c->device < 64 && g_dev[c->device].idle_streams.size() < 8)
            g_dev[c->device].idle_streams.push_back(c->stream);
        else
            (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

extern "C" int mbb_set_model(mbb_ctx *c, int opthin, int noalpha, double wavenorm)
{
    if (!c) return fail(MBB_ERR_ARG, "null context");
    if (!(wavenorm > 0.0)) return fail(MBB_ERR_ARG, "wavenorm must be positive");
    c->opthin = opthin ? 1 : 0;
    c->noalpha = noalpha ? 1 : 0;
    c->wavenorm = wavenorm;
    return MBB_OK;
}

template <typename T>
static int upload(T **dptr, const std::vector<T> &h)
{
    free_dev(*dptr);
    *dptr = nullptr;
    HIPCHK(hipMalloc((void **)dptr, sizeof(T) * (h.size() ? h.size() : 1)));
    if (!h.empty())
        HIPCHK(hipMemcpy(*dptr, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return MBB_OK;
}

extern "C" int mbb_set_bands(mbb_ctx *c, const double *freq, const double *weight,
                             const int32_t *offsets, int nb)
{
    int rc = use(c);
    if (rc) return rc;
    mbbh::BandLayout L;
    const char *why = "bad band tables";
    if (mbbh::build_band_layout(freq, weight, offsets, nb, (int)c->opt_seg_chunks, c->opt_pack_tails != 0, L, &why))
        return fail(MBB_ERR_ARG, why);
    std::vector<int4> unit_tab(L.unit_tab.size());
    std::vector<int2> band_rng(L.band_rng.size());
    memcpy(unit_tab.data(), L.unit_tab.data(), unit_tab.size() * sizeof(int4));
    memcpy(band_rng.data(), L.band_rng.data(), band_rng.size() * sizeof(int2));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((rc = upload(&c->d_nu, L.nu))) return rc;
    if ((rc = upload(&c->d_lnnu, L.lnnu))) return rc;
    if ((rc = upload(&c->d_wt, L.wt))) return rc;
    if ((rc = upload(&c->d_unit_tab, unit_tab))) return rc;
    if ((rc = upload(&c->d_band_rng, band_rng))) return rc;
    if ((rc = upload(&c->d_tail_slot, L.tail_slot))) return rc;
    c->nb = nb;
    c->nchunk = L.nchunk;
    c->nseg = L.nseg;
    c->nunit = L.nunit;
    c->npart = L.npart;
    for (int g = 0; g < 4; ++g) c->simd_chunks[g] = L.simd_chunks[g];
    c->nq = L.nq;
    return MBB_OK;
}

extern "C" int mbb_set_data(mbb_ctx *c, const double *flux, const double *w, int nb, int is_cov)
{
    int rc = use(c);
    if (rc) return rc;
    if (!flux || !w || nb <= 0) return fail(MBB_ERR_ARG, "bad data");
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<double> f(flux, flux + nb);
    if ((rc = upload(&c->d_flux, f))) return rc;
    if (is_cov) {
        std::vector<double> m(w, w + (size_t)nb * nb);
        if ((rc = upload(&c->d_invcov, m))) return rc;
        std::vector<double> iv(nb, 0.0);
        if ((rc = upload(&c->d_ivar, iv))) return rc;
    } else {
        std::vector<double> iv(w, w + nb);
        if ((rc = upload(&c->d_ivar, iv))) return rc;
        free_dev(c->d_invcov);
        c->d_invcov = nullptr;
    }
    c->has_cov = is_cov ? 1 : 0;
    c->data_nb = nb;
    c->nsrc = 1;
    return MBB_OK;
}

extern "C" int mbb_set_data_multi(mbb_ctx *c, const double *flux, const double *ivar, int nb, int nsrc)
{
    int rc = use(c);
    if (rc) return rc;
    if (!flux || !ivar || nb <= 0 || nsrc <= 0) return fail(MBB_ERR_ARG, "bad data");
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<double> f(flux, flux + (size_t)nb * nsrc), iv(ivar, ivar + (size_t)nb * nsrc);
    if ((rc = upload(&c->d_flux, f))) return rc;
    if ((rc = upload(&c->d_ivar, iv))) return rc;
    free_dev(c->d_invcov);
    c->d_invcov = nullptr;
    c->has_cov = 0;
    c->data_nb = nb;
    c->nsrc = nsrc;
    return MBB_OK;
}

extern "C" int mbb_set_limits(mbb_ctx *c, const double lowlim[5], const int32_t has_uplim[6],
                              const double uplim[6])
{
    if (!c || !lowlim || !has_uplim || !uplim) return fail(MBB_ERR_ARG, "null argument");
    c->has_uplim = 0;
    for (int i = 0; i < 5; ++i) c->lowlim[i] = lowlim[i];
    for (int i = 0; i < 6; ++i) {
        c->uplim[i] = has_uplim[i] ? uplim[i] : INFINITY;      // (a wall that is not set: never a NaN in the argument block)
        if (has_uplim[i]) c->has_uplim |= (1u << i);
    }
    return MBB_OK;
}

extern "C" int mbb_set_gpriors(mbb_ctx *c, const int32_t has[6], const double mean[6],
                               const double ivar[6])
{
    if (!c || !has || !mean || !ivar) return fail(MBB_ERR_ARG, "null argument");
    c->has_gprior = 0;
    for (int i = 0; i < 6; ++i) {
        c->gmean[i] = mean[i];
        c->givar[i] = ivar[i];
        if (has[i]) c->has_gprior |= (1u << i);
    }
    return MBB_OK;
}

// How mbb_lnlike_batch waits (option "spin_wait"): 0 blocks on the stream; 1 polls
// hipStreamQuery (measured on MI355X: no faster than 0); 2, the default for zero-copy
// batches of <= 8192 rows, watches the result slots in pinned host memory and falls back
// to 0 when they have not all appeared within "spin_budget" polls.
static const uint64_t kLnlSentinel = 0x7ff8dead5eed0001ull;
static const int32_t kStatusSentinel = 0x7fffff01;

static int wait_stream(mbb_ctx *c)
{
    if (c->opt_spin == 1) {
        for (;;) {
            hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) return MBB_OK;
            if (e != hipErrorNotReady) return fail(MBB_ERR_HIP, "hipStreamQuery", e);
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}

static int ensure_capacity(mbb_ctx *c, size_t n, bool want_flux)
{
    if (n > c->cap) {
        size_t cap = c->cap ? c->cap : 256;
        while (cap < n) cap *= 2;
        HIPCHK(hipStreamSynchronize(c->stream));
        // whoever holds addresses from mbb_boundary_buffers must see this BEFORE it writes through them again
        __atomic_store_n(&c->buf_gen, c->buf_gen + 1, __ATOMIC_RELEASE);
        free_dev(c->d_pars); free_dev(c->d_lnl); free_dev(c->d_status);
        free_host(c->h_pars); free_host(c->h_lnl); free_host(c->h_status);
        c->d_pars = c->d_lnl = nullptr; c->d_status = nullptr;
        c->h_pars = c->h_lnl = nullptr; c->h_status = nullptr;
        c->cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_pars, cap * 5 * sizeof(double)));
        HIPCHK(hipMalloc((void **)&c->d_lnl, cap * sizeof(double)));
        HIPCHK(hipMalloc((void **)&c->d_status, cap * sizeof(int32_t)));
        // mapped AND coherent (fine-grained): the kernel's stores to the result slots must
        // become visible to the polling host while the kernel is still running, whatever
        // HIP_HOST_COHERENT defaults to
        const unsigned hflags = hipHostMallocMapped | hipHostMallocCoherent;
        HIPCHK(hipHostMalloc((void **)&c->h_pars, cap * 5 * sizeof(double), hflags));
        HIPCHK(hipHostMalloc((void **)&c->h_lnl, cap * sizeof(double), hflags));
        HIPCHK(hipHostMalloc((void **)&c->h_status, cap * sizeof(int32_t), hflags));
        HIPCHK(hipHostGetDevicePointer((void **)&c->dv_pars, c->h_pars, 0));
        HIPCHK(hipHostGetDevicePointer((void **)&c->dv_lnl, c->h_lnl, 0));
        HIPCHK(hipHostGetDevicePointer((void **)&c->dv_status, c->h_status, 0));
        c->call_in = nullptr; c->call_cap = 0;
        // With a large BAR the host can store into (fine-grained) device memory directly:
        // posted writes, and the kernel then reads its parameter rows from local memory
        // instead of pulling them across PCIe.
        free_dev(c->w_pars);
        c->w_pars = nullptr;
        if (c->large_bar < 0) {
            int lb = 0;
            if (hipDeviceGetAttribute(&lb, hipDeviceAttributeIsLargeBar, c->device) != hipSuccess) lb = 0;
            c->large_bar = lb;
        }
        if (c->large_bar > 0 &&
            hipExtMallocWithFlags((void **)&c->w_pars, cap * 5 * sizeof(double), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            c->w_pars = nullptr;
        }
        c->cap = cap;
    }
    if (want_flux && n * (size_t)c->nb > c->cap_flux) {
        size_t cap = c->cap * (size_t)c->nb;
        if (cap < n * (size_t)c->nb) cap = n * (size_t)c->nb;
        HIPCHK(hipStreamSynchronize(c->stream));
        free_dev(c->d_mflux); free_host(c->h_mflux);
        c->d_mflux = nullptr; c->h_mflux = nullptr; c->cap_flux = 0;
        HIPCHK(hipMalloc((void **)&c->d_mflux, cap * sizeof(double)));
        HIPCHK(hipHostMalloc((void **)&c->h_mflux, cap * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void **)&c->dv_mflux, c->h_mflux, 0));
        c->cap_flux = cap;
    }
    return MBB_OK;
}

// LDS the kernel declares statically: the 2^(j/256) table and the piecewise polynomials
// (mbb_kernels.hip.h); the rest of the CU's 160 KB is what a launch may ask for dynamically.
static size_t static_lds(const mbb_ctx *c)
{
    return sizeof(double) * kExp2N + sizeof(double) * kPolyBDoubles +
           sizeof(double) * (c->opthin ? 2 : kPolyCDoubles);
}
static size_t dynamic_lds_limit(const mbb_ctx *c) { return 160 * 1024 - static_lds(c) - 2048; }

// Dynamic LDS of a k_lnlike workgroup of `wpb` walkers, without the staged passband tables and without
// the inverse covariance.
static size_t lnlike_lds_base(const mbb_ctx *c, int wpb)
{
    return (size_t)wpb * (sizeof(WalkerK) + 8 * (size_t)c->npart + 8 * (size_t)c->nb + 16) +
           16 * (size_t)c->nb + 8 * ((size_t)c->nb + 2) + 64 * (size_t)wpb;
}

// Walkers per workgroup and workgroup size.  mid_staged: the launch is of the middle regime below and
// wants its passband tables in LDS.
static void pick_geometry(const mbb_ctx *c, int n, int &wpb, int &threads, bool *mid_staged = nullptr)
{
    if (mid_staged) *mid_staged = false;
    // Small batches (an emcee half-step) are latency bound: one walker per
    // workgroup with about one segment per wave, so the chip sees n workgroups.
    // The static tables (36-57 KB) allow two workgroups per CU, so beyond one walker
    // per CU the workgroups are 512 threads wide -- 16 waves per CU -- and share their
    // tables among up to 32 walkers (a wave then takes whole walkers, 4 each).
    const long cus = c->cu_count;
    long w = 1, t = 256;
    if (n <= cus) {
        t = (long)((c->nunit + 3) / 4) * 256;          // ~ one unit per wave
        if (t > 1024) t = 1024;
        if (c->nunit <= 4) t = 256;
    } else {
        w = (n + cus * 2 - 1) / (cus * 2);
        if (w > 32) w = 32;
        t = 512;
        // The middle regime, more than one walker per CU but at most 256 (round 3, tools/sweep_geometry.py,
        // profiles/r03/geometry_sweep.txt): ONE workgroup of 1024 threads per CU with the passband tables
        // staged in LDS once for all its walkers beats two narrower workgroups reading them through L2 by
        // 13-17 % from 500 to 8192 rows, 7 % at 32768, 3 % at 65536; at 250 000 rows it is 3 % behind.
        const long wm = std::min<long>(64, (n + cus - 1) / cus);
        const size_t tables = (size_t)c->nchunk * 64 * 3 * sizeof(double);
        if (n <= 256 * cus && c->opt_wpb <= 0 && c->opt_threads <= 0 && c->opt_stage != 0 &&
            lnlike_lds_base(c, (int)wm) + (c->has_cov ? 8 * (size_t)c->nb * c->nb : 0) + tables + 16 <= dynamic_lds_limit(c)) {
            w = wm; t = 1024;
            if (mid_staged) *mid_staged = true;
        }
    }
    if (c->opt_wpb > 0) w = c->opt_wpb > 64 ? 64 : c->opt_wpb;
    wpb = (int)w;
    if (c->opt_threads > 0) t = c->opt_threads;
    if (t < 64) t = 64;
    if (t > 1024) t = 1024;
    threads = (int)((t / 64) * 64);
    // the prologue runs on one row of 16 lanes per walker
    if (threads < 16 * wpb) threads = ((16 * wpb + 63) / 64) * 64;
}

// Look-ahead sampler runs: how the workgroups that work ahead are shaped.  One row of 16 lanes
// per candidate proposal -- 4 per walker of a half: a row keeps to one half (SMODE 5, 6) --
// `rows` of them per wave, `aw` such waves per
// workgroup.  A constructor is one dependent chain and a wave alone on its SIMD runs it fastest,
// so the candidates are spread as thinly as the CUs the movers leave free allow.
// Returns false when only the densest packing (64 candidates per workgroup) would fit: measured, the
// one-launch run is then slower than the launch train (480 walkers: 17.7 against 15.8 us per step;
// 450 walkers, 32 per workgroup: 15.0 against 15.8 -- profiles/r03/walker_sweep.txt), so the host takes
// the train from there (ensembles above ~454 walkers on 256 CUs).
static bool lookahead_plan(const mbb_ctx *c, int movers, int threads, int half, int &rows, int &aw, int &n_ahead,
                           bool *sparse = nullptr)
{
    const int pairs = 4 * half, free_cus = c->cu_count - movers;
    static const int plan[5][2] = {{1, 4}, {2, 4}, {4, 4}, {4, 8}, {4, 16}};
    rows = 1; aw = 4;
    int chosen = 4;
    for (int i = 0; i < 5; ++i) {
        rows = plan[i][0]; aw = std::min(plan[i][1], threads / 64);
        if ((pairs + rows * aw - 1) / (rows * aw) <= free_cus) { chosen = i; break; }
    }
    bool worth = chosen < 4;
    if (sparse) *sparse = chosen <= 1;        // at most 8 candidates per working-ahead workgroup
    if (c->opt_la_rows > 0) { rows = (int)(c->opt_la_rows == 4 ? 4 : (c->opt_la_rows == 2 ? 2 : 1)); worth = true; }
    if (c->opt_la_waves > 0) { aw = (int)std::min<long>(c->opt_la_waves, threads / 64); worth = true; }
    n_ahead = (pairs + rows * aw - 1) / (rows * aw);
    return worth;
}

constexpr long kFlowStrikes = 3, kFlowRest = 16;
constexpr long kServeRest = 4096;            // boundary calls the served path rests for after three lost servers in a row
static std::atomic<unsigned long long> g_flow_serial{0};   // one-launch sampler runs started in this process (their check words)

struct SamplerLaunch {
    double *pos6, *chain6;
    unsigned int *nacc;
    int *errflag;
    int s_begin, c_begin, c_count, m_count, nw, step, half, nw_src;
    double stretch_a;
    unsigned long long seed;
    unsigned long long xseq;      // > 0: one-hop exchange, number of this launch
    int persist;                  // > 0: this many half-steps in one launch (SMODE 5, 6, form 7)
    double *spec;                 // != nullptr: the one-launch run's device state
    bool xflow = false;           // one-launch run of a sharded ensemble (SMODE 6): spec is the FlowX
    bool merged = false;          // k_flowm (form 7): one workgroup per (pair of walkers, candidate)
    bool resident = false;        // k_flowa (form 9): a workgroup owns several walkers of each half, every workgroup resident
    int res_w = 1;                // ... walkers per workgroup and half
    bool res_ahead = false;       // ... with the constructor a half-step ahead (k_flowa, form 9)
    unsigned long long serial = 0;   // ... the number of its launch (in its check words and decision words)
    int parity = 0;               // ... which of the two sets of completion counters it uses
};

// One launch of a kernel of the likelihood family.  launch_api 1 (default): the module-launch entry with the argument
// block handed over as ONE packed buffer -- no per-argument marshalling in the runtime (A/B: tools/probe_boundary_breakdown.py,
// profiles/r04/boundary_breakdown.txt); with c->launch_ev set, the launch that carries the two events itself.
static int launch_packed(mbb_ctx *c, void (*kern)(const LikeArgs), int fn_slot, int grid, int threads, size_t smem, LikeArgs &a)
{
    hipEvent_t *ev = c->launch_ev;
    if (c->opt_launch_api == 1) {
        hipFunction_t &f = c->mod_fn[fn_slot];
        if (!f) HIPCHK(hipGetFuncBySymbol(&f, (const void *)kern));
        size_t sz = sizeof(a);
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        // (hipExtModuleLaunchKernel would take the two events with the launch -- the dispatch's own timestamps, no marker
        // packets: measured, it costs 8 us MORE of host time per timed region than two hipEventRecord around the launch,
        // profiles/r04/timed_region.txt)
        if (ev) HIPCHK(hipEventRecord(ev[0], c->stream));
        HIPCHK(hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, (unsigned)threads, 1, 1, (unsigned)smem, c->stream, nullptr, extra));
        if (ev) HIPCHK(hipEventRecord(ev[1], c->stream));
        return MBB_OK;
    }
    if (ev) HIPCHK(hipEventRecord(ev[0], c->stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, c->stream, a);
    HIPCHK(hipGetLastError());
    if (ev) HIPCHK(hipEventRecord(ev[1], c->stream));
    return MBB_OK;
}

static int launch_lnlike(mbb_ctx *c, const double *d_pars, int n, double *d_lnl,
                         int32_t *d_status, double *d_mflux, const SamplerLaunch *sl = nullptr)
{
    if (c->nb <= 0) return fail(MBB_ERR_STATE, "bands not set (mbb_set_bands)");
    if (c->data_nb != c->nb) return fail(MBB_ERR_STATE, "data not set or band count mismatch");
    if (n <= 0) return MBB_OK;
    LikeArgs a;
    a.xargs = nullptr;
    a.persist = 0; a.spec = nullptr;
    a.nu = c->d_nu; a.lnnu = c->d_lnnu; a.wt = c->d_wt;
    a.poly_b = c->d_poly_b; a.poly_c = c->d_poly_c;
    a.unit_tab = c->d_unit_tab; a.band_rng = c->d_band_rng; a.tail_slot = c->d_tail_slot;
    a.flux = c->d_flux; a.ivar = c->d_ivar; a.invcov = c->has_cov ? c->d_invcov : nullptr;
    a.nb = c->nb; a.nunit = c->nunit; a.npart = c->npart; a.nchunk = c->nchunk;
    a.nunorm = kUmToGHz / c->wavenorm;
    a.lnunorm = log(a.nunorm);
    for (int i = 0; i < 5; ++i) a.lowlim[i] = c->lowlim[i];
    for (int i = 0; i < 6; ++i) { a.uplim[i] = c->uplim[i]; a.gmean[i] = c->gmean[i]; a.givar[i] = c->givar[i]; }
    a.has_uplim = c->has_uplim; a.has_gprior = c->has_gprior;
    a.pars = d_pars; a.n = n; a.lnl = d_lnl; a.status = d_status; a.model_flux = d_mflux;
    int wpb, threads;
    bool mid_staged = false;
    pick_geometry(c, n, wpb, threads, &mid_staged);
    a.wpb = wpb;
    a.debug = (int)c->opt_debug;
#ifdef MBB_STAMPS
    a.stamps = c->d_stamps;
#endif
    int grid = (n + wpb - 1) / wpb;
    const size_t cov_bytes = c->has_cov ? 8 * (size_t)c->nb * c->nb : 0;
    const size_t smem_base = lnlike_lds_base(c, wpb);
    // the inverse covariance goes to LDS when it fits beside everything else
    const size_t dyn_limit = dynamic_lds_limit(c);
    a.cov_in_lds = (c->has_cov && smem_base + cov_bytes <= std::min<size_t>(64 * 1024, dyn_limit)) ? 1 : 0;
    const size_t smem = smem_base + (a.cov_in_lds ? cov_bytes : 0);
    if (smem > dyn_limit) return fail(MBB_ERR_ARG, "band tables too large for the LDS plan");
    c->last_wpb = wpb; c->last_threads = threads; c->last_grid = grid;
    a.nsrc = c->nsrc;
    // Big batches of given rows: gate, constructor and penalties by a pass of their own with a LANE per walker (k_walker_pre)
    // -- where a launch is bound by the number of instructions it issues, a row of 16 lanes per walker spends a quarter of
    // them on the constructors (cfg5's 250 000 rows: 1.31 -> 1.0x ms).  Option "prepass": -1 auto (from 64 walkers per CU),
    // 0 never, 1 always.
    c->last_prepass = 0;
    if (!sl && (c->opt_prepass > 0 || (c->opt_prepass < 0 && (long)n >= 64L * c->cu_count))) {
        const size_t need = (size_t)n * kPreWords;
        if (need > c->pre_cap) {
            size_t cap = c->pre_cap ? c->pre_cap : 16384 * (size_t)kPreWords;
            while (cap < need) cap *= 2;
            HIPCHK(hipStreamSynchronize(c->stream));
            free_dev(c->d_pre); c->d_pre = nullptr; c->pre_cap = 0;
            HIPCHK(hipMalloc((void **)&c->d_pre, cap * sizeof(double)));
            c->pre_cap = cap;
        }
        a.spec = c->d_pre;
        a.xargs = nullptr;
        static void (*const ptable[4])(const LikeArgs) = {k_walker_pre<false, false>, k_walker_pre<false, true>,
                                                          k_walker_pre<true, false>, k_walker_pre<true, true>};
        hipLaunchKernelGGL(ptable[(c->opthin ? 2 : 0) | (c->noalpha ? 1 : 0)], dim3((n + 255) / 256), dim3(256), 0, c->stream, a);
        HIPCHK(hipGetLastError());
        c->last_prepass = 1;
    }
    a.rows_per_src = 0;
    a.nw_src = 0;
    if (c->nsrc > 1) {
        if (n % c->nsrc != 0)
            return fail(MBB_ERR_ARG, "row count must be a multiple of the number of sources");
        a.rows_per_src = n / c->nsrc;
    }
    // LDS staging of the passband tables.  Measured (profiles/r01/ab_stage.txt,
    // interleaved A/B): in the latency regime (one walker per workgroup, the copy
    // hides under the prologue) it is 1.5 % faster; with many walkers per
    // workgroup it is 4-5 % slower than reading the tables through L2, because
    // 60 KB of LDS per workgroup caps residency at two workgroups per CU.
    const size_t table_bytes = (size_t)c->nchunk * 64 * 3 * sizeof(double);
    bool stage = ((wpb == 1 && n <= c->cu_count) || mid_staged) && smem + table_bytes + 16 <= dyn_limit;
    if (c->opt_stage == 0) stage = false;
    if (c->opt_stage == 1) stage = smem + table_bytes + 16 <= dyn_limit;
    const size_t smem_total = smem + (stage ? table_bytes + 16 : 0);
    c->last_smem = (long)smem_total;
    c->last_stage = stage ? 1 : 0;
    void (*kern)(const LikeArgs);
    int vi_of_kernel = 0;
    if (sl && sl->resident) {
        // sampler form 9: ceil(n / W) workgroups of 16 waves, every one resident, W walkers of each half apiece
        a.pos6 = sl->pos6; a.chain6 = sl->chain6; a.nacc = sl->nacc; a.errflag = sl->errflag;
        a.s_begin = sl->s_begin; a.c_begin = sl->c_begin; a.c_count = sl->c_count; a.nw = sl->nw;
        a.m_count = sl->m_count;
        a.step = sl->step; a.half = sl->half; a.stretch_a = sl->stretch_a; a.seed = sl->seed;
        a.nw_src = sl->nw_src;
        a.persist = sl->persist;
        a.flow_serial = c->flow_serial = sl->serial;
        a.spec = sl->spec;
        a.spec_cfg = (int)((c->opt_flow_spin_log2 & 0x3f) << 24) | (sl->parity & 1);
        a.n_ahead = 0;
        const int Wr = sl->res_w, wgs = (n + Wr - 1) / Wr, thr = 1024;
        a.wpb = Wr;
        auto lds_of = [&](bool cov) { return flowa_lds_bytes(c->nb, c->npart, cov, Wr); };
        a.cov_in_lds = (c->has_cov && lds_of(true) <= std::min<size_t>(64 * 1024, dyn_limit)) ? 1 : 0;
        const size_t sm = lds_of(a.cov_in_lds != 0);
        const bool stg = c->opt_stage != 0 && sm + table_bytes + 16 <= dyn_limit;
        const size_t sm_total = sm + (stg ? table_bytes + 16 : 0);
        if (sm_total > dyn_limit) return fail(MBB_ERR_ARG, "band tables too large for the LDS plan");
        if (wgs > c->cu_count || Wr > kFrMaxWHost)
            return fail(MBB_ERR_ARG, "the one-launch sampler run needs every workgroup resident: too many for this GPU");
        static void (*const rtable[8])(const LikeArgs) = {
            k_flowa<false, false, false>, k_flowa<false, false, true>, k_flowa<false, true, false>, k_flowa<false, true, true>,
            k_flowa<true, false, false>, k_flowa<true, false, true>, k_flowa<true, true, false>, k_flowa<true, true, true>};
        const int ri = ((c->opthin ? 2 : 0) | (c->noalpha ? 1 : 0)) * 2 + (stg ? 1 : 0);
        kern = rtable[ri];
        c->last_wpb = Wr; c->last_threads = thr; c->last_grid = wgs; c->last_smem = (long)sm_total;
        c->last_stage = stg ? 1 : 0; c->last_smode = 9; c->last_ahead = 0;
        if (static_lds(c) + sm_total > 60 * 1024) {
            size_t &g = c->lds_granted[72 + ri];
            if (sm_total > g) {
                size_t want = (sm_total + 16383) & ~(size_t)16383;
                if (want > dyn_limit) want = dyn_limit;
                HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
                g = want;
            }
        }
        return launch_packed(c, kern, 40 + ri, wgs, thr, sm_total, a);
    }
    if (sl && sl->merged) {
        // sampler form 7: 2 n workgroups of (quadrature waves + 5), every one resident; its own LDS plan
        a.pos6 = sl->pos6; a.chain6 = sl->chain6; a.nacc = sl->nacc; a.errflag = sl->errflag;
        a.s_begin = sl->s_begin; a.c_begin = sl->c_begin; a.c_count = sl->c_count; a.nw = sl->nw;
        a.m_count = sl->m_count;
        a.step = sl->step; a.half = sl->half; a.stretch_a = sl->stretch_a; a.seed = sl->seed;
        a.nw_src = sl->nw_src;
        a.persist = sl->persist;
        a.flow_serial = c->flow_serial = sl->serial;
        a.spec = sl->spec;
        a.spec_cfg = (int)((c->opt_flow_spin_log2 & 0x3f) << 24) | (sl->parity & 1);
        a.n_ahead = 0;
        const int nq = std::min(threads / 64, 11);
        const int thr = (nq + 5) * 64;
        const int np = 1;              // (round 3 also had two pairs of walkers per workgroup: superseded by form 9)
        const int wgs = 2 * n;
        a.cov_in_lds = (c->has_cov && flowm_lds_bytes(c->nb, c->npart, true, np) <= std::min<size_t>(64 * 1024, dyn_limit)) ? 1 : 0;
        const size_t sm = flowm_lds_bytes(c->nb, c->npart, a.cov_in_lds != 0, np);
        const bool stg = c->opt_stage != 0 && sm + table_bytes + 16 <= dyn_limit;
        const size_t sm_total = sm + (stg ? table_bytes + 16 : 0);
        if (sm_total > dyn_limit) return fail(MBB_ERR_ARG, "band tables too large for the LDS plan");
        if (wgs > c->cu_count)
            return fail(MBB_ERR_ARG, "the one-launch sampler run needs every workgroup resident: too many for this GPU");
        static void (*const mtable[8])(const LikeArgs) = {
            k_flowm<false, false, false, 1>, k_flowm<false, false, true, 1>, k_flowm<false, true, false, 1>,
            k_flowm<false, true, true, 1>, k_flowm<true, false, false, 1>, k_flowm<true, false, true, 1>,
            k_flowm<true, true, false, 1>, k_flowm<true, true, true, 1>};
        const int mi = ((c->opthin ? 2 : 0) | (c->noalpha ? 1 : 0)) * 2 + (stg ? 1 : 0);
        kern = mtable[mi];
        c->last_wpb = np; c->last_threads = thr; c->last_grid = wgs; c->last_smem = (long)sm_total;
        c->last_stage = stg ? 1 : 0; c->last_smode = 7; c->last_ahead = 0;
        if (static_lds(c) + sm_total > 60 * 1024) {
            size_t &g = c->lds_granted[56 + mi];
            if (sm_total > g) {
                size_t want = (sm_total + 16383) & ~(size_t)16383;
                if (want > dyn_limit) want = dyn_limit;
                HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
                g = want;
            }
        }
        return launch_packed(c, kern, 32 + mi, wgs, thr, sm_total, a);
    }
    if (sl) {
        a.pos6 = sl->pos6; a.chain6 = sl->chain6; a.nacc = sl->nacc; a.errflag = sl->errflag;
        a.s_begin = sl->s_begin; a.c_begin = sl->c_begin; a.c_count = sl->c_count; a.nw = sl->nw;
        a.m_count = sl->m_count;
        a.step = sl->step; a.half = sl->half; a.stretch_a = sl->stretch_a; a.seed = sl->seed;
        a.nw_src = sl->nw_src;
        if (sl->xseq) a.xargs = c->x.d_args;
        a.persist = sl->persist;
        if (sl->spec) {
            a.xargs = nullptr;
            a.flow_serial = c->flow_serial = ++g_flow_serial;
            // first the workgroups that prepare the next half-step -- one row of 16 lanes per
            // (walker, candidate), `rows` of them per wave, `aw` such waves per workgroup -- then
            // the movers.  A constructor is one dependent chain: a wave alone on its SIMD runs it
            // fastest, so the candidates are spread as thinly as the CUs the movers leave free allow.
            int rows, aw, n_ahead;
            lookahead_plan(c, grid, threads, sl->m_count, rows, aw, n_ahead);
            a.spec = sl->spec;
            a.spec_cfg = (rows << 8) | (aw << 16) | (int)((c->opt_flow_spin_log2 & 0x3f) << 24);
            a.n_ahead = n_ahead;
            if (a.n_ahead + grid > c->cu_count)
                return fail(MBB_ERR_ARG, "the one-launch sampler run needs every workgroup resident: too many for this GPU");
            grid = a.n_ahead + grid;
            c->last_grid = grid;
        }
    } else {
        a.pos6 = nullptr; a.chain6 = nullptr; a.nacc = nullptr; a.errflag = nullptr;
        a.s_begin = a.c_begin = a.c_count = a.m_count = a.nw = a.step = a.half = 0;
        a.stretch_a = 2.0; a.seed = 0;
    }
    {
        // forms: 0 the likelihood of given rows, 1 the half-step, 2 the half-step with the one-hop
        // exchange, 6 the one-launch look-ahead run across the ranks of a sharded ensemble (slot 3).
        // (Its single-GPU twin, SMODE 5, went in round 4: forms 7 and 9 are faster at every ensemble size.)
        const int smode = !sl ? 0 : (sl->spec ? 6 : (sl->xseq ? 2 : 1));
        if (sl && sl->spec && !sl->xflow) return fail(MBB_ERR_STATE, "one-launch run without a form");
        const int slot = smode == 6 ? 3 : smode;
        const int vi = ((c->opthin ? 2 : 0) | (c->noalpha ? 1 : 0)) * 8 + slot * 2 + (stage ? 1 : 0);
        c->last_smode = smode;
        c->last_ahead = smode == 6 ? a.n_ahead : 0;
        vi_of_kernel = vi;
#define MBB_VARIANTS(OT, NA)                                                                        \
    k_lnlike<OT, NA, 0, false>, k_lnlike<OT, NA, 0, true>, k_lnlike<OT, NA, 1, false>,              \
        k_lnlike<OT, NA, 1, true>, k_lnlike<OT, NA, 2, false>, k_lnlike<OT, NA, 2, true>,           \
        k_lnlike<OT, NA, 6, false>, k_lnlike<OT, NA, 6, true>
        static void (*const table[32])(const LikeArgs) = {
            MBB_VARIANTS(false, false), MBB_VARIANTS(false, true), MBB_VARIANTS(true, false),
            MBB_VARIANTS(true, true)};
#undef MBB_VARIANTS
        kern = table[vi];
    }
    if (static_lds(c) + smem_total > 60 * 1024) {
        // beyond 64 KB of LDS per workgroup (static + dynamic) the kernel's dynamic-LDS
        // ceiling has to be raised: once per context, variant and size, not per launch
        size_t &g = c->lds_granted[vi_of_kernel];
        if (smem_total > g) {
            size_t want = (smem_total + 16383) & ~(size_t)16383;
            if (want > dyn_limit) want = dyn_limit;
            HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)want));
            g = want;
        }
    }
    return launch_packed(c, kern, vi_of_kernel, grid, threads, smem_total, a);
}

extern "C" int mbb_lnlike_batch_device(mbb_ctx *c, const double *d_pars, int n, double *d_lnl,
                                       int32_t *d_status, double *d_mflux)
{
    int rc = use(c);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_pars || !d_lnl))) return fail(MBB_ERR_ARG, "bad batch buffers");
    return launch_lnlike(c, d_pars, n, d_lnl, d_status, d_mflux);
}

extern "C" int mbb_lnlike_repeat_device(mbb_ctx *c, const double *d_pars, int n, double *d_lnl,
                                        int32_t *d_status, int reps)
{
    int rc = use(c);
    if (rc) return rc;
    if (n <= 0 || reps <= 0 || !d_pars || !d_lnl) return fail(MBB_ERR_ARG, "bad batch buffers");
    for (int r = 0; r < reps; ++r)
        if ((rc = launch_lnlike(c, d_pars, n, d_lnl, d_status, nullptr))) return rc;
    return MBB_OK;
}

// The zero-copy host path once the parameter rows are where the kernel reads them (device memory the host
// wrote through the BAR when `push`, the pinned block otherwise): launch, watch the result slots in pinned
// memory (or wait for the stream), leave lnl / status / model flux in c->h_lnl, c->h_status, c->h_mflux.
static int lnlike_zero_copy(mbb_ctx *c, int n, bool push, bool model_flux, long t_a)
{
    int rc;
    auto now_ns = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1000000000L + ts.tv_nsec; };
    // the kernel reads the pinned parameter block and writes lnL straight
    // into pinned host memory: no copy commands on the stream at all
    double *const dp = push ? c->w_pars : c->dv_pars, *const dl = c->dv_lnl, *const df = model_flux ? c->dv_mflux : nullptr;
    int32_t *const ds = c->dv_status;
    // spin_wait = 2: do not wait for the kernel's completion signal at all.  The
    // result slots in pinned memory are pre-filled with patterns the kernel never
    // writes (a NaN with a payload; it produces the canonical NaN), and the host
    // watches them: a walker's results are final the moment they appear, before the
    // end-of-kernel bookkeeping.  status is written after lnl by the same lane.
    const bool watch = c->opt_spin == 2 && !model_flux && n <= 8192;
    c->last_watch_seen = -1;
    if (watch) {
        uint64_t *hl = reinterpret_cast<uint64_t *>(c->h_lnl);
        for (int i = 0; i < n; ++i) { hl[i] = kLnlSentinel; c->h_status[i] = kStatusSentinel; }
    }
    const long t_b = now_ns();
    if ((rc = launch_lnlike(c, dp, n, dl, ds, df))) return rc;
    const long t_c = now_ns();
    bool seen = false;
    if (watch) {
        // acquire loads: the copies out of h_lnl / h_status below must not be satisfied
        // from before the slot was seen to change
        const uint64_t *hl = reinterpret_cast<const uint64_t *>(c->h_lnl);
        const int32_t *hs = c->h_status;
        int i = 0;
        for (long spins = 0; spins < c->opt_spin_budget; ++spins) {   // default ~ tens of ms, then give up
            while (i < n && __atomic_load_n(&hl[i], __ATOMIC_ACQUIRE) != kLnlSentinel &&
                   __atomic_load_n(&hs[i], __ATOMIC_ACQUIRE) != kStatusSentinel) ++i;
            if (i == n) { seen = true; break; }
            __builtin_ia32_pause();
        }
        c->last_watch_seen = seen ? 1 : 0;
    }
    if (!seen && (rc = wait_stream(c))) return rc;
    c->t_prep_ns = t_b - t_a; c->t_launch_ns = t_c - t_b; c->t_wait_ns = now_ns() - t_c;
    return MBB_OK;
}

extern "C" int mbb_lnlike_batch(mbb_ctx *c, const double *pars, int n, double *lnl,
                                int32_t *status, double *model_flux)
{
    int rc = use(c);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!pars || !lnl))) return fail(MBB_ERR_ARG, "bad batch buffers");
    if (n == 0) return MBB_OK;
    if (c->nb <= 0) return fail(MBB_ERR_STATE, "bands not set (mbb_set_bands)");
    if ((rc = ensure_capacity(c, (size_t)n, model_flux != nullptr))) return rc;
    const size_t nbytes = (size_t)n * 5 * sizeof(double);
    auto now_ns = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1000000000L + ts.tv_nsec; };
    const long t_a = now_ns();
    const bool push = c->opt_zero_copy && c->opt_bar_params && c->w_pars;
    if (push) {
        memcpy(c->w_pars, pars, nbytes);           // CPU stores through the BAR ...
        __builtin_ia32_sfence();                   // ... drained before the doorbell is rung
    } else {
        memcpy(c->h_pars, pars, nbytes);
    }
    if (c->opt_zero_copy) {
        if ((rc = lnlike_zero_copy(c, n, push, model_flux != nullptr, t_a))) return rc;
    } else {
        HIPCHK(hipMemcpyAsync(c->d_pars, c->h_pars, nbytes, hipMemcpyHostToDevice, c->stream));
        if ((rc = launch_lnlike(c, c->d_pars, n, c->d_lnl, c->d_status,
                                model_flux ? c->d_mflux : nullptr))) return rc;
        HIPCHK(hipMemcpyAsync(c->h_lnl, c->d_lnl, (size_t)n * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_status, c->d_status, (size_t)n * sizeof(int32_t),
                              hipMemcpyDeviceToHost, c->stream));
        if (model_flux)
            HIPCHK(hipMemcpyAsync(c->h_mflux, c->d_mflux, (size_t)n * c->nb * sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
        if ((rc = wait_stream(c))) return rc;
    }
    memcpy(lnl, c->h_lnl, (size_t)n * sizeof(double));
    if (status) memcpy(status, c->h_status, (size_t)n * sizeof(int32_t));
    if (model_flux) memcpy(model_flux, c->h_mflux, (size_t)n * c->nb * sizeof(double));
    return MBB_OK;
}

// The boundary call with the copies taken out (the caller is likelihood.__call__,
// likelihood.py:790-834, once per emcee half-step).  mbb_boundary_buffers hands the binding the host
// addresses of the block it writes the parameter rows INTO -- device memory behind the BAR where there is one,
// else the pinned block the kernel reads -- and of the pinned blocks the kernel writes lnprob and row status
// to; mbb_lnlike_call then evaluates the first n rows of that block: no pointer arguments, no memcpy on either
// side, the row status looked through here.  The addresses hold until a call asks for more rows than `nmax`
// (ask again then) or the context is destroyed.
extern "C" const unsigned long long *mbb_boundary_generation(mbb_ctx *c)
{
    return c ? &c->buf_gen : nullptr;
}

extern "C" int mbb_boundary_buffers(mbb_ctx *c, int nmax, double **in, double **out, int32_t **status)
{
    int rc = use(c);
    if (rc) return rc;
    if (nmax <= 0 || !in || !out) return fail(MBB_ERR_ARG, "bad arguments");
    if ((rc = ensure_capacity(c, (size_t)nmax, false))) return rc;
    const bool push = c->opt_zero_copy && c->opt_bar_params && c->w_pars;
    c->call_in = push ? c->w_pars : c->h_pars;
    c->call_cap = c->cap;
    *in = c->call_in; *out = c->h_lnl;
    if (status) *status = c->h_status;
    return MBB_OK;
}

// ---- the served boundary ----------------------------------------------------------------------------------------
// (see mbb_serve.hip.h for what it is and why)
// Called with c->srv_mu held (by c's owner, or by a visiting sibling: yield_server).
static int serve_stop(mbb_ctx *c)
{
    if (!c->serving) return MBB_OK;
    c->serving = false;
    if (c->device >= 0 && c->device < 64) {
        std::lock_guard<std::mutex> lk(g_dev_mutex);
        if (g_dev[c->device].server == c) g_dev[c->device].server = nullptr;
    }
    // a request number the server has not seen, with the row count that means "leave"
    __atomic_store_n(c->w_door, (++c->srv_seq << 16) | kServeQuitHost, __ATOMIC_RELAXED);
    __builtin_ia32_sfence();
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}

// Start a server with the request in the launch itself.  The argument block is launch_lnlike's, filled in here for
// the fields k_serve reads.
// How many workgroups a server of this context is started with: a row each for the widest call so far, in eights -- not
// one per CU: what it does not hold is there for the launches and the servers of other processes (two pool workers of 125
// rows each have their servers side by side, 8.7 us per call in both: profiles/r05/pool_two_processes_sized.txt) --, and never
// more than this process's share of the device.
static int serve_grid(const mbb_ctx *c, int n, int share)
{
    int g = c->opt_serve_grid > 0 ? c->opt_serve_grid : std::max(c->srv_want, (n + 7) & ~7);
    return std::min(std::min(g, share), c->cu_count);
}

// (the launch itself; serve_start below claims the device around it)
static int serve_launch(mbb_ctx *c, int n, unsigned long long word, int grid)
{
    if (!c->w_door) {
        if (hipExtMallocWithFlags((void **)&c->w_door, 64, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            c->w_door = nullptr;
            return 1;
        }
        HIPCHK(hipHostMalloc((void **)&c->h_gone, 64, hipHostMallocMapped | hipHostMallocCoherent));
    }

    LikeArgs a;
    memset(&a, 0, sizeof a);
    a.nu = c->d_nu; a.lnnu = c->d_lnnu; a.wt = c->d_wt;
    a.poly_b = c->d_poly_b; a.poly_c = c->d_poly_c;
    a.unit_tab = c->d_unit_tab; a.band_rng = c->d_band_rng; a.tail_slot = c->d_tail_slot;
    a.flux = c->d_flux; a.ivar = c->d_ivar; a.invcov = c->has_cov ? c->d_invcov : nullptr;
    a.nb = c->nb; a.nunit = c->nunit; a.npart = c->npart; a.nchunk = c->nchunk;
    a.nunorm = kUmToGHz / c->wavenorm;
    a.lnunorm = log(a.nunorm);
    for (int i = 0; i < 5; ++i) a.lowlim[i] = c->lowlim[i];
    for (int i = 0; i < 6; ++i) { a.uplim[i] = c->uplim[i]; a.gmean[i] = c->gmean[i]; a.givar[i] = c->givar[i]; }
    a.has_uplim = c->has_uplim; a.has_gprior = c->has_gprior;
    a.pars = c->w_pars; a.n = n; a.lnl = c->dv_srv; a.status = nullptr; a.model_flux = nullptr;
    a.wpb = 1; a.debug = (int)c->opt_debug; a.nsrc = 1;
#ifdef MBB_STAMPS
    a.stamps = c->d_stamps;
#endif
    int wpb, threads;
    pick_geometry(c, 1, wpb, threads);
    const size_t dyn_limit = dynamic_lds_limit(c);
    const size_t table_bytes = (size_t)c->nchunk * 64 * 3 * sizeof(double);
    a.cov_in_lds = (c->has_cov && serve_lds_bytes(c->nb, c->npart, true) <= std::min<size_t>(64 * 1024, dyn_limit)) ? 1 : 0;
    const size_t sm = serve_lds_bytes(c->nb, c->npart, a.cov_in_lds != 0);
    const bool stg = c->opt_stage != 0 && sm + table_bytes + 16 <= dyn_limit;
    // The quadrature beside the constructor (mbb_serve.hip.h): every wave but the first keeps both candidates of its own
    // unit's samples in registers
    // -- with a chunk of samples at least for each of twelve waves: with fewer the extra barrier costs more than there is to
    // gain (cfg1's one chunk: 8.5-8.7 -> 8.9 us per 25-row call; option "serve_overlap" 2: regardless; 0: never)
    // -- and waves to work ahead with (a workgroup narrowed by option "block_threads" may have none besides the constructor's)
    // -- and units of at most four chunks (option "seg_chunks")
    const bool ovl = c->opt_serve_overlap != 0 && (c->nchunk >= 12 || c->opt_serve_overlap == 2) && threads >= 256 &&
                     c->opt_seg_chunks >= 1 && c->opt_seg_chunks <= 4;
    a.spec_cfg = ovl ? 1 : 0;
    const size_t sm_total = sm + (stg ? table_bytes + 16 : 0);
    if (sm_total > dyn_limit) return 1;
    HIPCHK(hipHostGetDevicePointer((void **)&a.chain6, c->h_gone, 0));
    a.pos6 = reinterpret_cast<double *>(c->w_door);
    a.seed = word;
    a.persist = (int)std::min<long>(std::max<long>(c->opt_serve_idle_us, 20), 1000000);
#define MBB_SV(OT, NA) k_serve<OT, NA, false, false>, k_serve<OT, NA, true, false>, k_serve<OT, NA, false, true>, k_serve<OT, NA, true, true>
    static void (*const stable[16])(const LikeArgs) = {MBB_SV(false, false), MBB_SV(false, true), MBB_SV(true, false), MBB_SV(true, true)};
#undef MBB_SV
    const int si = ((c->opthin ? 2 : 0) | (c->noalpha ? 1 : 0)) * 4 + (ovl ? 2 : 0) + (stg ? 1 : 0);
    void (*kern)(const LikeArgs) = stable[si];
    if (static_lds(c) + sm_total > 60 * 1024) {
        size_t &g = c->lds_granted[96 + si];
        if (sm_total > g) {
            size_t want = (sm_total + 16383) & ~(size_t)16383;
            if (want > dyn_limit) want = dyn_limit;
            HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
            g = want;
        }
    }
    *c->h_gone = 0;
    __atomic_store_n(c->w_door, word, __ATOMIC_RELAXED);
    __builtin_ia32_sfence();
    {
        int rc = launch_packed(c, kern, 56 + si, grid, threads, sm_total, a);
        if (rc) return rc;
    }
    c->srv_grid = grid;
    c->last_wpb = 1; c->last_threads = threads; c->last_grid = grid; c->last_smem = (long)sm_total;
    c->last_stage = stg ? 1 : 0; c->last_smode = 10;
    c->serving = true;
    c->srv_run = 0;
    { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); c->srv_t0_ns = ts.tv_sec * 1000000000L + ts.tv_nsec; }
    return MBB_OK;                                 // (the device has had this context's name since the claim above)
}

// Starts a resident server for this context.  The device is CLAIMED before anything is launched: two contexts on two
// threads that both found no server resident (yield_server) would otherwise both start one, the device would remember
// the second, and nobody could find the first to tell it to leave.  Whoever comes second sends its rows by
// a launch (1); its next call finds the first's server by name and tells it to go.
static int serve_start(mbb_ctx *c, int n, unsigned long long word, int grid)
{
    if (grid <= 0 || n > kServePasses * grid) return 1;
    const bool named = c->device >= 0 && c->device < 64;
    if (named) {
        std::lock_guard<std::mutex> lk(g_dev_mutex);
        mbb_ctx *&sv = g_dev[c->device].server;
        if (sv && sv != c) return 1;
        sv = c;
    }
    const int rc = serve_launch(c, n, word, grid);
    if (rc != MBB_OK && named) {
        std::lock_guard<std::mutex> lk(g_dev_mutex);
        if (g_dev[c->device].server == c) g_dev[c->device].server = nullptr;
    }
    return rc;
}

// One request: MBB_OK when the results are in the pinned slots, 1 when the rows have to go by a launch after all,
// negative on error.  The parameter rows are in c->w_pars already.
static int serve_request(mbb_ctx *c, int n, int grid)
{
    const int kPfAhead = c->opt_serve_prefetch;
    auto now_ns = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1000000000L + ts.tv_nsec; };
    const long t_a = now_ns();
    // (the records exist once a server has been started; before that serve_start makes them)
    if (!c->h_srv) {
        HIPCHK(hipHostMalloc((void **)&c->h_srv, (size_t)kServePasses * c->cu_count * 8 * kSrvStride, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void **)&c->dv_srv, c->h_srv, 0));
        c->srv_cap = (size_t)kServePasses * c->cu_count;
    }
    uint64_t *hr = reinterpret_cast<uint64_t *>(c->h_srv);
    for (int i = 0; i < n; ++i) hr[kSrvStride * i + 1] = (uint64_t)kStatusSentinel;
    const unsigned long long word = (++c->srv_seq << 16) | (unsigned long long)n;
    __builtin_ia32_sfence();                       // the caller's rows, through the BAR, before the request
    const long t_b = now_ns();
    long budget_ns = c->opt_serve_budget_us * 1000L;
    if (!c->serving) {
        int rc = serve_start(c, n, word, grid);
        if (rc) return rc;
        budget_ns += 200000L;                      // (a launch, and the tables into LDS on every CU)
    } else {
        __atomic_store_n(c->w_door, word, __ATOMIC_RELAXED);
        __builtin_ia32_sfence();
    }
    const long t_c = now_ns();
    ++c->srv_requests;
    int i = 0;
    bool seen = false;
    for (long spins = 0;; ++spins) {
        // a record is one 16-byte store: when its status word has turned, its lnl is there
        while (i < n && __atomic_load_n(&hr[kSrvStride * i + 1], __ATOMIC_ACQUIRE) != (uint64_t)kStatusSentinel) {
            // (the GPU's writes took the records' lines out of this core's caches: once the first has turned the others are
            // landing, and their misses are asked for kPfAhead lines ahead of the scan instead of one by one behind it --
            // tools/lat_doorbell.hip: the scan of 125 landed lines 0.67-1.29 -> 0.65-0.76 us.  Not before the first has
            // turned: a line that sits in the core's cache when the GPU wants to write it costs the GPU a snoop)
            if (i == 0) for (int k = 1; k < kPfAhead && k < n; ++k) __builtin_prefetch(&hr[kSrvStride * k], 0, 3);
            if (kPfAhead && i + kPfAhead < n) __builtin_prefetch(&hr[kSrvStride * (i + kPfAhead)], 0, 3);
            c->h_lnl[i] = c->h_srv[kSrvStride * i];
            c->h_status[i] = (int32_t)hr[kSrvStride * i + 1];
            ++i;
        }
        if (i == n) { seen = true; break; }
        if ((spins & 63) == 63 && (now_ns() - t_c > budget_ns || __atomic_load_n(c->h_gone, __ATOMIC_ACQUIRE) != 0)) break;
        __builtin_ia32_pause();
    }
    c->last_watch_seen = seen ? 1 : 0;
    c->t_prep_ns = t_b - t_a; c->t_launch_ns = t_c - t_b; c->t_wait_ns = now_ns() - t_c;
    if (seen) {
        c->srv_strikes = 0;
        if (++c->srv_run >= 256) c->srv_need = 0;
        return MBB_OK;
    }
    // the records did not turn: the server had left when the request came (it says so: nothing wrong, the sampler's
    // calls were further apart than its patience), or it was leaving, or it is not resident.  It is told to go, waited
    // for, and the rows go by a launch; three requests in a row lost to a server that had NOT said it was gone, and
    // the feature rests for the next kServeRest boundary calls (rounds 4-5: for the life of the context -- one bad
    // stretch on a shared box, and a sampler's every later call was a launch; the sampler forms rest the same way).
    ++c->srv_fallbacks;
    c->srv_hot = 0;
    if (__atomic_load_n(c->h_gone, __ATOMIC_ACQUIRE) == 0 && ++c->srv_strikes >= 3) {
        c->srv_strikes = 0;
        c->srv_rest = kServeRest;
        ++c->srv_rests;
    }
    int rc = serve_stop(c);
    return rc ? rc : 1;
}

// Returns MBB_OK, a negative error, or -- positive -- the status code of the first row that the reference
// would have raised for (alpha <= 0, beta < 0, no merge point: modified_blackbody.py:219-224, :294-316).
extern "C" int mbb_lnlike_call(mbb_ctx *c, int n)
{
    if (!c) return fail(MBB_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));               // (not use(): a resident server stays, the streak of calls goes on)
    std::lock_guard<std::mutex> own(c->srv_mu);    // (uncontended: ~20 ns; a sibling's visit waits for the call to end)
    int rc;
    if (n <= 0) return n == 0 ? MBB_OK : fail(MBB_ERR_ARG, "bad row count");
    if (c->nb <= 0) return fail(MBB_ERR_STATE, "bands not set (mbb_set_bands)");
    const bool push = c->opt_zero_copy && c->opt_bar_params && c->w_pars;
    if (!c->opt_zero_copy || !c->call_in || c->call_cap != c->cap || (size_t)n > c->cap ||
        c->call_in != (push ? c->w_pars : c->h_pars)) {
        if (c->serving && (rc = serve_stop(c))) return rc;
        return fail(MBB_ERR_STATE, "mbb_boundary_buffers first (or again: the buffers or the host-path options changed)");
    }
    // After a few boundary calls in a row with nothing else in between -- a sampler's loop -- the rows are handed to
    // a kernel that stays on the GPU (k_serve) instead of a launch each: while the batch is at most a row per CU and the
    // host path is the default one.  One server per device and process: whichever context comes to the device tells a
    // sibling's to leave first (use(), and the line below).
    bool can_serve = c->opt_serve && push && n <= kServePasses * c->cu_count && c->nsrc <= 1 && c->opt_spin == 2 && c->data_nb == c->nb;
    if (c->srv_rest > 0) { --c->srv_rest; can_serve = false; }      // (resting after three lost servers in a row)
    // ... and across PROCESSES (emcee's pool, mbb_fit.py:80-81 threads > 1: the likelihood pickled into workers that share the
    // GPU): a server holds a CU per workgroup, and nothing of another process fits on those -- with round 4's server on every
    // CU one worker's call waited 42 ms for the other's whole loop (profiles/r05/pool_two_processes_before.txt).  So a server is
    // as wide as the calls have rows, and no wider than this process's share of the device: the CUs divided by the processes
    // of this library that are making boundary calls on it right now (csrc/mbb_registry.h: each says so at every call; one
    // that holds a context but does not call -- a pool's parent -- is not in the way).  A call of more rows than the share goes
    // by a launch; a server that is too wide for the share, or too narrow for the call, leaves and the next starts with this
    // call.  ("serve" 2: the whole device is this process's share -- tests.)
    int share = c->cu_count, grid = 0;
    const long need_hot = c->srv_need > 0 ? c->srv_need : c->opt_serve_after;
    timespec tb = {0, 0};
    if (can_serve) {
        clock_gettime(CLOCK_MONOTONIC, &tb);                   // (one look at the clock: who is busy, and the lease)
        if (c->opt_serve != 2) {
            const uint64_t now_ms = (uint64_t)tb.tv_sec * 1000u + (uint64_t)(tb.tv_nsec / 1000000);
            // (before a server is there every call looks: a pool's workers begin together, and three servers of 128 do not fit)
            const int busy = mbbh::registry_busy(c->reg_key, now_ms, 250, !c->serving);
            // (the dispatcher deals a kernel's workgroups to the device's 8 XCDs in turn, and there they stay: what has to fit
            // is every process's ceil(workgroups / 8) into an XCD's CUs -- three servers of 85 are 33 on the first XCDs of 32,
            // one workgroup never starts and every request of its server times out)
            share = c->xcds * ((c->cu_count / c->xcds) / (busy + 1));
            c->srv_busy = busy;
        }
        c->srv_want = std::max(c->srv_want, std::min((n + 7) & ~7, c->cu_count));
        grid = serve_grid(c, n, share);
        // (a server narrower than the call has rows takes them in turns, kServePasses rows a workgroup at most: beyond that a
        // launch is the faster way)
        if (grid <= 0 || n > kServePasses * grid) {
            // (counted once per run of calls that would have been served)
            if (c->serving || ++c->srv_hot == need_hot) ++c->srv_peer_yields;
            can_serve = false;
        } else if (c->serving && (c->srv_grid > share || (n > c->srv_grid && grid > c->srv_grid))) {
            if ((rc = serve_stop(c))) return rc;
            c->srv_hot = need_hot;
            ++c->srv_resizes;
        }
    }
    // ... and not for ever in one go: processes the registry cannot see get the CUs when the lease is up (the rows of this
    // call go by a launch, a new server starts after the next few calls in a row)
    if (can_serve && c->serving && c->opt_serve_lease_us > 0) {
        if (tb.tv_sec * 1000000000L + tb.tv_nsec - c->srv_t0_ns > c->opt_serve_lease_us * 1000L) {
            can_serve = false;
            c->srv_hot = 0;
            ++c->srv_lease_yields;
        }
    }
    // (a sibling context's server: this call, served or launched, needs the CUs; and a sibling's visit in between ends this
    // context's run of calls)
    if (!c->serving && (rc = yield_server(c))) return rc;
    bool done = false;
    if (can_serve && (c->serving || ++c->srv_hot >= need_hot)) {
        rc = serve_request(c, n, grid);
        if (rc < 0) return rc;
        done = rc == MBB_OK;
    } else if (c->serving && (rc = serve_stop(c))) {
        return rc;
    }
    if (!done) {
        timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        if (push) __builtin_ia32_sfence();         // the caller's stores through the BAR, drained before the doorbell
        if ((rc = lnlike_zero_copy(c, n, push, false, ts.tv_sec * 1000000000L + ts.tv_nsec))) return rc;
    }
    const int32_t *hs = c->h_status;
    for (int i = 0; i < n; ++i)
        if (hs[i] >= 2 && hs[i] != 7) return hs[i];
    return MBB_OK;
}

// ---- device-resident ensemble sampler ----------------------------------------
constexpr size_t kChainDirectBytes = (size_t)8 << 20;   // stored chains up to this size go straight into the caller's arrays
struct mbb_sampler_state {
    int nw = 0;            // walkers per source
    int nsrc = 1;          // independent ensembles advanced together
    double *d_pos6 = nullptr;
    bool pos6_owned = true;              // false: the rows live in the context's exchange buffer
    unsigned int *d_nacc = nullptr;      // [shards][2][nsrc*per], launch-local order
    int *d_err = nullptr;
    double *d_bak = nullptr;             // one-launch run: the rows and counts it started from (R x 6 doubles, R counts)
    bool flow_used = false;              // the last enqueue took the one-launch form
    bool unchecked = false;              // an asynchronous advance was enqueued and its error flag not looked at yet
    bool lost = false;                   // such an advance gave up: the rows are no state of the chain until set again
    double *d_spec = nullptr;            // one-launch run: its device state (FlowView or FlowMView, spec_words(rows))
    int spec_form = 0;                   // the sampler form whose state d_spec holds (0: none / not to be trusted)
    int flowm_parity = 0;                // form 7: the set of completion counters the next launch uses
    double *d_chain6 = nullptr;          // [shards][nsteps][2][nsrc*per][6]
    double *d_chain_out = nullptr;       // the same chain in the caller's layout: [rows][nsteps][5] then [rows][nsteps]
    double *h_chain = nullptr;           // ... and its pinned landing place on the host (a D2H into the caller's pageable,
                                         // often untouched, arrays ran at ~1 GB/s: 18.8 us per step for 2000 stored steps)
    size_t chain_cap = 0;
    unsigned long long seed = 0, steps_done = 0;
    int rows() const { return nw * nsrc; }
};

extern "C" int mbb_sampler_create(mbb_ctx *c, int nwalkers, unsigned long long seed, void **out)
{
    int rc = use(c);
    if (rc) return rc;
    if (!out || nwalkers < 2 || (nwalkers & 1)) return fail(MBB_ERR_ARG, "nwalkers must be even");
    mbb_sampler_state *s = new mbb_sampler_state();
    s->nw = nwalkers;
    s->nsrc = c->nsrc > 0 ? c->nsrc : 1;
    s->seed = seed;
    const size_t R = (size_t)s->rows();
    if (c->x.connected) {
        // sharded with the one-hop exchange: the ensemble lives where the peers can write it
        if (R > c->x.cap_rows) { delete s; return fail(MBB_ERR_ARG, "more walkers than the exchange buffer holds"); }
        s->d_pos6 = c->x.pos6(c->x.rank);
        s->pos6_owned = false;
        ++c->x.users;
    } else {
        HIPCHK(hipMalloc((void **)&s->d_pos6, R * 6 * sizeof(double)));
    }
    HIPCHK(hipMalloc((void **)&s->d_nacc, R * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void **)&s->d_err, sizeof(int)));
    HIPCHK(hipMemset(s->d_nacc, 0, R * sizeof(unsigned int)));
    HIPCHK(hipMemset(s->d_err, 0, sizeof(int)));
    *out = s;
    return MBB_OK;
}

extern "C" int mbb_sampler_destroy(mbb_ctx *c, void *sp)
{
    int rc = use(c);
    if (rc) return rc;
    mbb_sampler_state *s = (mbb_sampler_state *)sp;
    if (!s) return MBB_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (s->pos6_owned) free_dev(s->d_pos6);
    else if (c->x.users > 0) --c->x.users;
    free_dev(s->d_nacc); free_dev(s->d_err); free_dev(s->d_chain6); free_dev(s->d_chain_out); free_dev(s->d_spec);
    if (s->h_chain) (void)hipHostFree(s->h_chain);
    free_dev(s->d_bak);
    delete s;
    return MBB_OK;
}

static int sampler_check_pending(mbb_ctx *c, mbb_sampler_state *s);

extern "C" int mbb_sampler_reset(mbb_ctx *c, void *sp)
{
    int rc = use(c);
    if (rc) return rc;
    mbb_sampler_state *s = (mbb_sampler_state *)sp;
    if (!s) return fail(MBB_ERR_ARG, "null sampler");
    (void)sampler_check_pending(c, s);       // (a give-up of an asynchronous advance is remembered, not wiped with the flag)
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemset(s->d_nacc, 0, (size_t)s->rows() * sizeof(unsigned int)));
    HIPCHK(hipMemset(s->d_err, 0, sizeof(int)));
    return MBB_OK;
}

// Set the ensemble(s): pos [nsrc][nw][5]; lnprob [nsrc][nw] or NULL (then evaluated
// on the device).  In a sharded run every rank sets the same, complete state.
extern "C" int mbb_sampler_set_state(mbb_ctx *c, void *sp, const double *pos, const double *lnprob)
{
    int rc = use(c);
    if (rc) return rc;
    mbb_sampler_state *s = (mbb_sampler_state *)sp;
    if (!s || !pos) return fail(MBB_ERR_ARG, "null argument");
    if (s->nsrc != c->nsrc) return fail(MBB_ERR_STATE, "number of sources changed since the sampler was made");
    const int R = s->rows();
    std::vector<double> lnp(R);
    if (lnprob) {
        for (int i = 0; i < R; ++i) lnp[i] = lnprob[i];
    } else {
        std::vector<int32_t> st(R);
        if ((rc = mbb_lnlike_batch(c, pos, R, lnp.data(), st.data(), nullptr))) return rc;
        for (int i = 0; i < R; ++i)
            if (st[i] >= 2) return fail(MBB_ERR_ARG, "initial position is not a valid SED or lnprob is NaN");
    }
    std::vector<double> rows((size_t)R * 6);
    for (int i = 0; i < R; ++i) {
        for (int k = 0; k < 5; ++k) rows[(size_t)i * 6 + k] = pos[(size_t)i * 5 + k];
        rows[(size_t)i * 6 + 5] = lnp[i];
        if (lnp[i] != lnp[i]) return fail(MBB_ERR_ARG, "initial lnprob is NaN");
    }
    HIPCHK(hipMemcpyAsync(s->d_pos6, rows.data(), rows.size() * sizeof(double),
                          hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (s->lost || s->unchecked) {
        // whatever an asynchronous advance left in the flag belongs to the state just replaced
        HIPCHK(hipMemset(s->d_err, 0, sizeof(int)));
        s->lost = s->unchecked = false;
        s->spec_form = 0;
    }
    return MBB_OK;
}

// What an asynchronous advance left behind, looked at once before anything is built on it: a one-launch
// run that gave up (flag 9) kept no copy to be redone from, so the rows are not a state of the chain.  Without
// this look the next run would back those rows up and, finding the stale 9 after its own launch, "redo" it from
// them and return MBB_OK (round 3's advisor finding).  Other flags stay for the run's own check.
static int sampler_check_pending(mbb_ctx *c, mbb_sampler_state *s)
{
    if (s->unchecked) {
        int err = 0;
        HIPCHK(hipMemcpyAsync(&err, s->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        s->unchecked = false;
        if (err == 9) {
            HIPCHK(hipMemset(s->d_err, 0, sizeof(int)));
            s->lost = true;
            s->spec_form = 0;
        }
    }
    if (s->lost)
        return fail(MBB_ERR_STATE, "the one-launch sampler run timed out waiting for a half-step; the ensemble it "
                                   "left behind is not a state of the chain: set the sampler's state again");
    return MBB_OK;
}

// How a run is cut into shards.  One shard = the whole half-ensemble (one GPU).
// With an RCCL communicator of G ranks the moving half is cut into G contiguous
// blocks; rank r moves block r and the blocks are exchanged by an in-place
// all-gather of the state rows, so every rank holds the whole, identical ensemble
// before the next half-step.  Option "virtual_ranks" runs the G
// shards one after another on this GPU without any collective -- the same result
// by construction, used to test the sharded launch arithmetic on one device.
struct ShardPlan { int shards, first, last, per; bool collective, xchg; };

static int shard_plan(const mbb_ctx *c, const mbb_sampler_state *s, ShardPlan &p)
{
    const int half = s->nw / 2;
    const bool xchg = c->x.connected && !s->pos6_owned && c->x.n > 1;
    p.collective = xchg || (c->comm != nullptr && c->nranks > 1);
    const int granks = xchg ? c->x.n : c->nranks, grank = xchg ? c->x.rank : c->rank;
    p.shards = p.collective ? granks : (int)(c->opt_vranks > 1 ? c->opt_vranks : 1);
    if (p.shards > 1 && s->nsrc > 1)
        return fail(MBB_ERR_ARG, "a sharded sampler run needs a single source");
    if (half % p.shards != 0)
        return fail(MBB_ERR_ARG, "nwalkers/2 must be a multiple of the number of ranks");
    p.per = half / p.shards;
    p.first = p.collective ? grank : 0;
    p.last = p.collective ? grank : p.shards - 1;
    p.xchg = xchg;
    return MBB_OK;
}

static int allgather_bytes(mbb_ctx *c, void *base, size_t bytes_per_rank)
{
    // in place: rank r's block already sits at base + r * bytes_per_rank
    int r = g_rccl.AllGather((const char *)base + (size_t)c->rank * bytes_per_rank, base,
                             bytes_per_rank, 0 /* ncclInt8 */, c->comm, c->stream);
    if (r != 0) return fail(MBB_ERR_RCCL, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclAllGather failed");
    return MBB_OK;
}

// (tev: two events for mbb_sampler_advance_timed, recorded on the stream right before the run's first launch and right
// behind its last)
static int sampler_enqueue(mbb_ctx *c, mbb_sampler_state *s, int nsteps, double stretch_a, bool store, bool backup = true,
                           hipEvent_t *tev = nullptr)
{
    s->flow_used = false;
    ShardPlan p;
    int rc = shard_plan(c, s, p);
    if (rc) return rc;
    const int nw = s->nw, half = nw / 2;
    const size_t nl = (size_t)s->nsrc * p.per;              // walkers per launch
    if (p.xchg && tev) HIPCHK(hipEventRecord(tev[0], c->stream));
    if (p.xchg) {
        // the kernel's view of the exchange for this run, in stream order before its launches
        // (pageable source: staged by the runtime before the call returns)
        XchgArgs xa;
        memset(&xa, 0, sizeof xa);
        for (int r = 0; r < c->x.n; ++r) { xa.xpos[r] = c->x.pos6(r); xa.xflag[r] = c->x.flags(r); }
        xa.xcount = c->x.count(); xa.xn = c->x.n; xa.xrank = c->x.rank;
        xa.xseq0 = c->x.seq; xa.xspin_max = c->x.spin_max;
        HIPCHK(hipMemcpyAsync(c->x.d_args, &xa, sizeof xa, hipMemcpyHostToDevice, c->stream));
    }
    SamplerLaunch sl;
    sl.pos6 = s->d_pos6; sl.errflag = s->d_err; sl.nw = s->rows(); sl.nw_src = nw;
    sl.stretch_a = stretch_a; sl.c_count = half; sl.m_count = p.per;
    sl.xseq = 0; sl.persist = 0;
    sl.spec = nullptr;
    // A sharded ensemble with the one-hop exchange, one launch per run (k_lnlike SMODE 6): every rank
    // moves its share of each half and prepares their proposals ahead, decisions / rows / words go
    // into every rank's copy of the run's state (behind the rows in the exchange buffer) at system
    // scope, and a row's half-step starts when the rows it depends on are done, on whatever GPU.
    // Around the launch: set up this rank's copy, tell the peers (a mover's first store into a
    // peer's copy waits for that peer's word), launch, tell the peers the launch has ended, wait
    // for theirs (no store of theirs is in flight any more), bring the rows up to date.
    {
        int wpb_x = 0, thr_x = 0, la_rows, la_aw, la_ahead;
        pick_geometry(c, (int)nl, wpb_x, thr_x);
        lookahead_plan(c, (int)nl, thr_x, p.per, la_rows, la_aw, la_ahead);
        if (p.xchg && c->opt_lookahead && c->opt_flow && c->opt_xflow && s->nsrc == 1 && nsteps > 0 && wpb_x == 1 &&
            la_ahead + (int)nl <= c->cu_count && (size_t)s->rows() <= c->x.cap_rows) {
            const size_t R = (size_t)s->rows();
            double *mine = c->x.flow(c->x.rank);
            const FlowView fvl = flow_view(mine, (int)R);
            sl.spec = reinterpret_cast<double *>(c->x.d_flowx);
            sl.xflow = true;
            for (int t0 = 0; t0 < nsteps; t0 += 4096) {
                const int nt = std::min(4096, nsteps - t0);
                FlowX fxh;
                memset(&fxh, 0, sizeof fxh);
                for (int r = 0; r < c->x.n; ++r) fxh.base[r] = c->x.flow(r);
                fxh.n = c->x.n; fxh.rank = c->x.rank; fxh.run = ++c->x.flow_run;
                HIPCHK(hipMemcpyAsync(c->x.d_flowx, &fxh, sizeof fxh, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(k_flow_init, dim3((unsigned)((R * 8 + 255) / 256)), dim3(256), 0, c->stream,
                                   s->d_pos6, mine, (int)R);
                hipLaunchKernelGGL(k_flow_post, dim3(1), dim3(64), 0, c->stream, c->x.d_flowx, (int)R, 0, fxh.run,
                                   (const int *)nullptr);
                HIPCHK(hipGetLastError());
                sl.s_begin = c->x.rank * p.per; sl.c_begin = half; sl.step = t0; sl.half = 0;
                sl.persist = 2 * nt;
                sl.chain6 = store ? s->d_chain6 + ((((size_t)c->x.rank * nsteps + t0) * 2) * nl) * 6 : nullptr;
                sl.nacc = s->d_nacc + (size_t)c->x.rank * 2 * nl;
                sl.seed = s->seed + 0x9E3779B97F4A7C15ull * (s->steps_done + (unsigned long long)t0 + 1ull);
                if ((rc = launch_lnlike(c, nullptr, (int)nl, nullptr, nullptr, nullptr, &sl))) return rc;
                hipLaunchKernelGGL(k_flow_post, dim3(1), dim3(64), 0, c->stream, c->x.d_flowx, (int)R, 1, fxh.run,
                                   (const int *)s->d_err);
                hipLaunchKernelGGL(k_flow_wait_end, dim3(1), dim3(64), 0, c->stream, fvl.endf, c->x.n, c->x.rank, fxh.run,
                                   c->x.spin_max, s->d_err);
                hipLaunchKernelGGL(k_flow_finish, dim3((unsigned)((R * 6 + 255) / 256)), dim3(256), 0, c->stream,
                                   s->d_pos6, mine, (int)R, 2 * nt);
                HIPCHK(hipGetLastError());
            }
            s->steps_done += (unsigned long long)nsteps;
            if (tev) HIPCHK(hipEventRecord(tev[1], c->stream));
            return MBB_OK;
        }
    }
    int wpb_1 = 0, thr_1 = 0;
    pick_geometry(c, (int)nl, wpb_1, thr_1);
    // Options "lookahead_sampler" / "flow_sampler" (default 1) -- one GPU, one ensemble, every workgroup resident: the run
    // is ONE launch per 4096 steps, rows handed over through check words instead of a launch boundary; chains are bitwise
    // those of the plain launch train, which is what several sources, very short runs and ensembles beyond 8 walkers per CU
    // and half take.  Which form (profiles/r04/walker_sweep.txt, us per MCMC step, cfg2 bands):
    //   form 7 (k_flowm): a workgroup per (pair of walkers, candidate), quadrature and constructor ahead of the decisions
    //           they depend on -- while each has a CU of its own: 6.0-6.3 up to 256 walkers;
    //   form 9 (k_flowa): a workgroup owns W = ceil(half / CUs) walkers of each half, the constructor a half-step ahead
    //           for both outcomes of the partner's pending move, a walker's quadrature starting when ITS partner has decided
    //           -- 8.4-8.9 from 258 to 512 walkers (round 3's forms there, removed this round: form 5 9.6-10.2 up to 340,
    //           form 7 with two pairs per workgroup 11.5 up to 512), 10.7-11.3 up to 1000, 15.6 at 1500, 18.6 at 2000
    //           (train: 18.9 / 19.1 / 22.4 / 26.0);
    //           (round 4's form 8, k_flowr -- the same ownership, nothing ahead -- existed for 3073-4096 walkers only and was
    //           6 % ahead there, 35.5 against 37.6 us per step at 4096: removed in round 5, form 9 takes those too).
    bool one_launch = c->opt_lookahead && c->opt_flow && p.shards == 1 && !p.collective && s->nsrc == 1 &&
                      nsteps >= (int)std::max<long>(1, c->opt_flow_min_steps);
    const bool merged = one_launch && c->opt_flowm && c->opt_flowr != 2 && 2 * (int)nl <= c->cu_count && wpb_1 == 1;
    int res_w = c->opt_flowr_walkers > 0 ? (int)std::min<long>(c->opt_flowr_walkers, kFrMaxWHost)
                                         : ((int)nl + c->cu_count - 1) / c->cu_count;
    const bool res_fits = res_w >= 1 && res_w <= kFrMaxWHost && ((int)nl + res_w - 1) / res_w <= c->cu_count;
    const bool resident = one_launch && !merged && c->opt_flowr != 0 && res_fits;
    const bool res_ahead = true;
    one_launch = merged || resident;
    if (one_launch && c->flow_rest > 0) { --c->flow_rest; one_launch = false; }   // resting after give-ups in a row
    if (one_launch) {
        const size_t R = (size_t)s->rows();
        if (!s->d_spec) {
            // zeroed: the records of a one-launch run are taken by their check words, and freshly
            // allocated memory may hold those of another sampler's run
            HIPCHK(hipMalloc((void **)&s->d_spec, spec_words(R) * sizeof(double)));
            HIPCHK(hipMemsetAsync(s->d_spec, 0, spec_words(R) * sizeof(double), c->stream));
        }
        {
            sl.spec = s->d_spec;
            // what the run starts from, kept so that a run that times out (a workgroup that is not
            // resident: another process on the GPU) can be redone as a launch train (mbb_sampler_run)
            // (mbb_sampler_advance_async does not redo anything: no copy there)
            if (backup) {
                if (!s->d_bak) HIPCHK(hipMalloc((void **)&s->d_bak, R * 6 * sizeof(double) + R * sizeof(unsigned int)));
                HIPCHK(hipMemcpyAsync(s->d_bak, s->d_pos6, R * 6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(hipMemcpyAsync(s->d_bak + R * 6, s->d_nacc, R * sizeof(unsigned int), hipMemcpyDeviceToDevice, c->stream));
            }
            s->flow_used = backup;
            if ((merged || resident) && s->spec_form != 7 && s->spec_form != 8) {
                // forms 7 and 8 find their completion counters cleared by the launch before it (either's); after
                // another form (or a run that gave up) has used the memory, once from here
                const FlowMView fvh = flowm_view(s->d_spec, (int)R);
                HIPCHK(hipMemsetAsync(fvh.done, 0, 2 * kFmRing * 16 * sizeof(unsigned long long), c->stream));
                s->flowm_parity = 0;
            }
            s->spec_form = resident ? 8 : 7;
            const bool carried = tev && nsteps <= 4096;
            if (tev && !carried) HIPCHK(hipEventRecord(tev[0], c->stream));
            for (int t0 = 0; t0 < nsteps; t0 += 4096) {
                const int nt = std::min(4096, nsteps - t0);
                // one launch, nothing before or after it: it files the rows it finds and stores the last ones back
                sl.serial = ++g_flow_serial;
                sl.parity = s->flowm_parity;
                s->flowm_parity ^= 1;
                sl.s_begin = 0; sl.c_begin = half; sl.step = t0; sl.half = 0;
                sl.persist = 2 * nt;
                sl.chain6 = store ? s->d_chain6 + ((size_t)t0 * 2 * nl) * 6 : nullptr;
                sl.nacc = s->d_nacc;
                sl.seed = s->seed + 0x9E3779B97F4A7C15ull * (s->steps_done + (unsigned long long)t0 + 1ull);
                sl.merged = merged;
                sl.resident = resident;
                sl.res_w = res_w;
                sl.res_ahead = res_ahead;
                c->launch_ev = carried ? tev : nullptr;
                rc = launch_lnlike(c, nullptr, (int)nl, nullptr, nullptr, nullptr, &sl);
                c->launch_ev = nullptr;
                if (rc) return rc;
            }
            if (tev && !carried) HIPCHK(hipEventRecord(tev[1], c->stream));
            s->steps_done += (unsigned long long)nsteps;
            return MBB_OK;
        }
    }
    sl.spec = nullptr; sl.merged = false; sl.resident = false; sl.persist = 0;
    if (tev && !p.xchg) HIPCHK(hipEventRecord(tev[0], c->stream));
    for (int t = 0; t < nsteps; ++t)
        for (int h = 0; h < 2; ++h) {
            const int hb = h ? half : 0;
            for (int r = p.first; r <= p.last; ++r) {
                sl.s_begin = hb + r * p.per; sl.c_begin = h ? 0 : half;
                sl.step = t; sl.half = h;
                sl.chain6 = store ? s->d_chain6 + ((((size_t)r * nsteps + t) * 2 + h) * nl) * 6 : nullptr;
                sl.nacc = s->d_nacc + ((size_t)r * 2 + h) * nl;
                // the RNG key advances over the whole life of the sampler, the chain index restarts
                sl.seed = s->seed + 0x9E3779B97F4A7C15ull * (s->steps_done + (unsigned long long)t + 1ull);
                sl.xseq = p.xchg ? ++c->x.seq : 0;
                if ((rc = launch_lnlike(c, nullptr, (int)nl, nullptr, nullptr, nullptr, &sl))) return rc;
            }
            if (p.collective && !p.xchg &&
                (rc = allgather_bytes(c, s->d_pos6 + (size_t)hb * 6, (size_t)p.per * 6 * sizeof(double))))
                return rc;
        }
    s->steps_done += (unsigned long long)nsteps;
    if (p.xchg) {
        // the stream is done only when every peer's last launch has landed here too; chain
        // and acceptance counts stay per rank (the caller gathers them if it wants them)
        if (c->x.seq)
            hipLaunchKernelGGL(k_xchg_wait, dim3(1), dim3(64), 0, c->stream, c->x.flags(c->x.rank), c->x.n,
                               c->x.rank, c->x.seq, c->x.spin_max, s->d_err);
        HIPCHK(hipGetLastError());
    } else if (p.collective) {   // every rank ends up with the whole chain and all counts
        if (store && (rc = allgather_bytes(c, s->d_chain6, (size_t)nsteps * 2 * nl * 6 * sizeof(double)))) return rc;
        if ((rc = allgather_bytes(c, s->d_nacc, 2 * nl * sizeof(unsigned int)))) return rc;
    }
    if (tev) HIPCHK(hipEventRecord(tev[1], c->stream));
    return MBB_OK;
}

// Advance nsteps stretch-move steps entirely on the device: 2 nsteps dependent
// launches on the context's stream, no host round trip in between.
// chain [nsrc*nw][nsteps][5] and lnprob [nsrc*nw][nsteps] (emcee's layout per
// source, results.py:154-155) may be NULL; pos_out [nsrc*nw*5], lnprob_out,
// naccepted [nsrc*nw] (running totals).
extern "C" int mbb_sampler_run(mbb_ctx *c, void *sp, int nsteps, double stretch_a, double *chain,
                               double *lnprob, double *pos_out, double *lnprob_out,
                               double *naccepted)
{
    int rc = use(c);
    if (rc) return rc;
    mbb_sampler_state *s = (mbb_sampler_state *)sp;
    if (!s || nsteps < 0 || !(stretch_a > 1.0)) return fail(MBB_ERR_ARG, "bad sampler arguments");
    if (s->nsrc != c->nsrc) return fail(MBB_ERR_STATE, "number of sources changed since the sampler was made");
    ShardPlan p;
    if ((rc = shard_plan(c, s, p))) return rc;
    const int R = s->rows(), nw = s->nw, half = nw / 2;
    const size_t nl = (size_t)s->nsrc * p.per;
    const bool store = (chain || lnprob) && nsteps > 0;
    if ((rc = sampler_check_pending(c, s))) return rc;
    if (store && (size_t)nsteps * R * 6 > s->chain_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        free_dev(s->d_chain6); s->d_chain6 = nullptr; s->chain_cap = 0;
        free_dev(s->d_chain_out); s->d_chain_out = nullptr;
        if (s->h_chain) { (void)hipHostFree(s->h_chain); s->h_chain = nullptr; }
        HIPCHK(hipMalloc((void **)&s->d_chain6, (size_t)nsteps * R * 6 * sizeof(double)));
        HIPCHK(hipMalloc((void **)&s->d_chain_out, (size_t)nsteps * R * 6 * sizeof(double)));
        // (a landing buffer the host refuses to pin -- a limit on locked memory -- is done without: the chain
        // then goes straight into the caller's arrays, slower for big chains, never wrong)
        // (only chains big enough to go through it get one: pinning costs a few hundred microseconds)
        if ((size_t)nsteps * R * 6 * sizeof(double) > kChainDirectBytes &&
            hipHostMalloc((void **)&s->h_chain, (size_t)nsteps * R * 6 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            s->h_chain = nullptr;
        }
        s->chain_cap = (size_t)nsteps * R * 6;
    }
    if ((rc = sampler_enqueue(c, s, nsteps, stretch_a, store))) return rc;
    std::vector<double> rows((size_t)R * 6);
    std::vector<unsigned int> nacc(R);
    int err = 0;
    HIPCHK(hipMemcpyAsync(rows.data(), s->d_pos6, rows.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(nacc.data(), s->d_nacc, nacc.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&err, s->d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    // The chain comes back in the caller's layout: re-ordered on the device, copied straight into the
    // caller's arrays.  (A rank of a run sharded with the one-hop exchange holds its own walkers' part
    // only: there the old way, through a staging vector, fills just those rows.)
    const bool direct = store && !p.xchg;
    bool via_pinned = false;
    std::vector<double> ch;
    if (direct) {
        const long long cells = (long long)R * nsteps;
        hipLaunchKernelGGL(k_chain_reorder, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, c->stream,
                           (const double *)s->d_chain6, s->d_chain_out, nsteps, nw, half, p.per, s->nsrc, p.shards);
        HIPCHK(hipGetLastError());
        // small chains go straight into the caller's arrays (the runtime stages them through its own pinned
        // buffers); big ones land in ours first -- a D2H into megabytes of pageable, untouched memory runs at
        // ~1 GB/s (measured: 3 MB direct 0.43 ms against 1.35 ms staged; 24 MB direct 25 ms against 6 ms staged)
        via_pinned = s->h_chain != nullptr && (size_t)cells * 6 * sizeof(double) > kChainDirectBytes;
        if (via_pinned) {
            HIPCHK(hipMemcpyAsync(s->h_chain, s->d_chain_out, (size_t)cells * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        } else {
            if (chain) HIPCHK(hipMemcpyAsync(chain, s->d_chain_out, (size_t)cells * 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            if (lnprob) HIPCHK(hipMemcpyAsync(lnprob, s->d_chain_out + (size_t)cells * 5, (size_t)cells * sizeof(double),
                                              hipMemcpyDeviceToHost, c->stream));
        }
    } else if (store) {
        ch.resize((size_t)nsteps * R * 6);
        HIPCHK(hipMemcpyAsync(ch.data(), s->d_chain6, ch.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (err) {
        HIPCHK(hipMemset(s->d_err, 0, sizeof(int)));
        if (err == 8) return fail(MBB_ERR_RCCL, "the exchange timed out: a peer did not post its launch");
        if (err == 9 && s->flow_used && s->d_bak) {
            // The one-launch run gave up waiting (not every workgroup was resident): back to the state it
            // started from and the same steps as a train of launches -- the same chain.  The next run takes
            // the one-launch form again; kFlowStrikes give-ups in a row rest it for kFlowRest runs.
            HIPCHK(hipMemcpyAsync(s->d_pos6, s->d_bak, (size_t)R * 6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(s->d_nacc, s->d_bak + (size_t)R * 6, (size_t)R * sizeof(unsigned int), hipMemcpyDeviceToDevice,
                                  c->stream));
            s->steps_done -= (unsigned long long)nsteps;
            s->spec_form = 0;
            ++c->flow_fallbacks;
            if (++c->flow_strikes >= kFlowStrikes) { c->flow_strikes = 0; c->flow_rest = kFlowRest; }
            const long keep = c->opt_flow;
            c->opt_flow = 0;                                   // (this redo only)
            rc = mbb_sampler_run(c, sp, nsteps, stretch_a, chain, lnprob, pos_out, lnprob_out, naccepted);
            c->opt_flow = keep;
            return rc;
        }
        if (err == 9) {
            // (a sharded run: nothing was kept to redo it from)
            s->spec_form = 0;
            s->lost = true;
            return fail(MBB_ERR_STATE, "the one-launch sampler run timed out waiting for a half-step; the ensemble it "
                                       "left behind is not a state of the chain: set the sampler's state again");
        }
        g_err = "lnprob returned NaN or the SED constructor rejected a proposal (row status " +
                std::to_string(err) + ")";
        return MBB_ERR_ARG;
    }
    if (s->flow_used) c->flow_strikes = 0;                  // a one-launch run that went through
    if (direct && via_pinned) {
        const size_t cells = (size_t)R * nsteps;
        if (chain) memcpy(chain, s->h_chain, cells * 5 * sizeof(double));
        if (lnprob) memcpy(lnprob, s->h_chain + cells * 5, cells * sizeof(double));
    }
    for (int i = 0; i < R; ++i) {
        if (pos_out) for (int k = 0; k < 5; ++k) pos_out[(size_t)i * 5 + k] = rows[(size_t)i * 6 + k];
        if (lnprob_out) lnprob_out[i] = rows[(size_t)i * 6 + 5];
    }
    // launch-local order -> rows: shard r, half h, launch index w = src * per + loc
    // (one-hop exchange: only this rank's shard is here; the other rows are left untouched)
    for (int r = p.xchg ? p.first : 0; r <= (p.xchg ? p.last : p.shards - 1); ++r)
        for (int h = 0; h < 2; ++h)
            for (size_t w = 0; w < nl; ++w) {
                const int src = (int)(w / p.per), loc = (int)(w - (size_t)src * p.per);
                const int row = src * nw + (h ? half : 0) + r * p.per + loc;
                if (naccepted) naccepted[row] = (double)nacc[((size_t)r * 2 + h) * nl + w];
                if (!store || direct) continue;
                for (int t = 0; t < nsteps; ++t) {
                    const double *q = &ch[((((size_t)r * nsteps + t) * 2 + h) * nl + w) * 6];
                    if (chain) for (int k = 0; k < 5; ++k) chain[((size_t)row * nsteps + t) * 5 + k] = q[k];
                    if (lnprob) lnprob[(size_t)row * nsteps + t] = q[5];
                }
            }
    return MBB_OK;
}

// Measurement helper: enqueue nsteps steps without storing a chain and without
// synchronising (HIP events around the call time the dependent launch train).
extern "C" int mbb_sampler_advance_async(mbb_ctx *c, void *sp, int nsteps, double stretch_a)
{
    int rc = use(c);
    if (rc) return rc;
    mbb_sampler_state *s = (mbb_sampler_state *)sp;
    if (!s || nsteps < 0) return fail(MBB_ERR_ARG, "bad sampler arguments");
    if (s->lost) return sampler_check_pending(c, s);
    s->unchecked = true;
    return sampler_enqueue(c, s, nsteps, stretch_a, false, false);
}

// Measurement helper: the same enqueue between the host clock and two events, in one call -- the harness
// around a short timed region is then a few hundred nanoseconds instead of four Python-to-C round trips.
// (The events ride on the launch itself when the run is one: sampler_enqueue.)
extern "C" int mbb_sampler_advance_timed(mbb_ctx *c, void *sp, int nsteps, double stretch_a, double *wall_s,
                                         float *stream_ms)
{
    int rc = use(c);
    if (rc) return rc;
    mbb_sampler_state *s = (mbb_sampler_state *)sp;
    if (!s || nsteps < 0 || !wall_s || !stream_ms) return fail(MBB_ERR_ARG, "bad sampler arguments");
    if (s->lost) return sampler_check_pending(c, s);
    s->unchecked = true;
    if (!c->ev_timed[0]) {
        HIPCHK(hipEventCreate(&c->ev_timed[0]));
        HIPCHK(hipEventCreate(&c->ev_timed[1]));
    }
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = sampler_enqueue(c, s, nsteps, stretch_a, false, false, c->ev_timed))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    const auto t1 = std::chrono::steady_clock::now();
    *wall_s = std::chrono::duration<double>(t1 - t0).count();
    HIPCHK(hipEventElapsedTime(stream_ms, c->ev_timed[0], c->ev_timed[1]));
    return MBB_OK;
}

// ---- SED-level entry points -------------------------------------------------
static int ensure_sed(mbb_ctx *c, size_t n, size_t nout)
{
    if (n > c->sed_cap) {
        size_t cap = c->sed_cap ? c->sed_cap : 64;
        while (cap < n) cap *= 2;
        HIPCHK(hipStreamSynchronize(c->stream));
        free_dev(c->d_sed_pars); free_dev(c->d_sed_status); free_dev(c->d_sed_wk);
        c->d_sed_pars = nullptr; c->d_sed_status = nullptr; c->d_sed_wk = nullptr; c->sed_cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_sed_pars, cap * 5 * sizeof(double)));
        HIPCHK(hipMalloc((void **)&c->d_sed_status, cap * sizeof(int32_t)));
        HIPCHK(hipMalloc((void **)&c->d_sed_wk, cap * sizeof(WalkerK)));
        c->sed_cap = cap;
    }
    if (nout > c->sed_out_cap) {
        size_t cap = c->sed_out_cap ? c->sed_out_cap : 1024;
        while (cap < nout) cap *= 2;
        HIPCHK(hipStreamSynchronize(c->stream));
        free_dev(c->d_sed_out); c->d_sed_out = nullptr; c->sed_out_cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_sed_out, cap * sizeof(double)));
        c->sed_out_cap = cap;
    }
    return MBB_OK;
}

template <typename F>
static void dispatch_variant(int opthin, int noalpha, F &&f)
{
    if (opthin) { if (noalpha) f(std::true_type(), std::true_type()); else f(std::true_type(), std::false_type()); }
    else { if (noalpha) f(std::false_type(), std::true_type()); else f(std::false_type(), std::false_type()); }
}

static int run_prologue(mbb_ctx *c, const double *pars, int n, int opthin, int noalpha,
                        double wavenorm, int want_peak, double *d_out6)
{
    HIPCHK(hipMemcpyAsync(c->d_sed_pars, pars, (size_t)n * 5 * sizeof(double),
                          hipMemcpyHostToDevice, c->stream));
    const int threads = 64, grid = (n + threads - 1) / threads;
    dispatch_variant(opthin, noalpha, [&](auto OT, auto NA) {
        hipLaunchKernelGGL((k_prologue<decltype(OT)::value, decltype(NA)::value>), dim3(grid),
                           dim3(threads), 0, c->stream, c->d_sed_pars, n, kUmToGHz / wavenorm,
                           log(kUmToGHz / wavenorm), want_peak, d_out6, c->d_sed_status,
                           c->d_sed_wk);
    });
    HIPCHK(hipGetLastError());
    return MBB_OK;
}

extern "C" int mbb_sed_prologue_batch(mbb_ctx *c, const double *pars, int n, int opthin,
                                      int noalpha, double wavenorm, int want_peak, double *out,
                                      int32_t *status)
{
    int rc = use(c);
    if (rc) return rc;
    if (n <= 0 || !pars || !out) return fail(MBB_ERR_ARG, "bad arguments");
    if ((rc = ensure_sed(c, (size_t)n, (size_t)n * 6))) return rc;
    if ((rc = run_prologue(c, pars, n, opthin, noalpha, wavenorm, want_peak, c->d_sed_out))) return rc;
    HIPCHK(hipMemcpyAsync(out, c->d_sed_out, (size_t)n * 6 * sizeof(double),
                          hipMemcpyDeviceToHost, c->stream));
    if (status)
        HIPCHK(hipMemcpyAsync(status, c->d_sed_status, (size_t)n * sizeof(int32_t),
                              hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}

extern "C" int mbb_sed_eval_batch(mbb_ctx *c, const double *pars, int n, int opthin, int noalpha,
                                  double wavenorm, const double *freq, int m, double *out,
                                  int32_t *status)
{
    int rc = use(c);
    if (rc) return rc;
    if (n <= 0 || m <= 0 || !pars || !freq || !out) return fail(MBB_ERR_ARG, "bad arguments");
    if (n > 65535) return fail(MBB_ERR_ARG, "at most 65535 rows per call");
    const size_t nout = (size_t)n * m;
    if ((rc = ensure_sed(c, (size_t)n, nout + (size_t)m))) return rc;
    double *d_freq = c->d_sed_out + nout;
    if ((rc = run_prologue(c, pars, n, opthin, noalpha, wavenorm, 0, nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(d_freq, freq, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int threads = 256;
    dim3 grid((m + threads - 1) / threads, n);
    dispatch_variant(opthin, noalpha, [&](auto OT, auto NA) {
        hipLaunchKernelGGL((k_sed_eval<decltype(OT)::value, decltype(NA)::value>), grid,
                           dim3(threads), 0, c->stream, c->d_sed_wk, d_freq, m, c->d_sed_out);
    });
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->d_sed_out, nout * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status)
        HIPCHK(hipMemcpyAsync(status, c->d_sed_status, (size_t)n * sizeof(int32_t),
                              hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}

extern "C" int mbb_sed_integrate_batch(mbb_ctx *c, const double *pars, int n, int opthin, int noalpha,
                                       double wavenorm, double numin, double numax, double *out,
                                       int32_t *status)
{
    int rc = use(c);
    if (rc) return rc;
    if (n <= 0 || !pars || !out || !(numin > 0.0) || !(numax > numin))
        return fail(MBB_ERR_ARG, "bad arguments");
    const int ngl = 64;
    if ((rc = ensure_sed(c, (size_t)n, (size_t)n + 2 * ngl))) return rc;
    // Gauss-Legendre nodes and weights on [-1, 1] (Newton on P_n, Abramowitz & Stegun 25.4.29)
    double gx[64], gw[64];
    for (int i = 0; i < ngl; ++i) {
        double x = cos(M_PI * (i + 0.75) / (ngl + 0.5)), pp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 0; j < ngl; ++j) { double p3 = p2; p2 = p1; p1 = ((2.0 * j + 1.0) * x * p2 - j * p3) / (j + 1.0); }
            pp = ngl * (x * p1 - p2) / (x * x - 1.0);
            const double dx = p1 / pp;
            x -= dx;
            if (fabs(dx) < 1e-16) break;
        }
        gx[i] = x;
        gw[i] = 2.0 / ((1.0 - x * x) * pp * pp);
    }
    double *d_gx = c->d_sed_out + n, *d_gw = d_gx + ngl;
    if ((rc = run_prologue(c, pars, n, opthin, noalpha, wavenorm, 0, nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(d_gx, gx, sizeof gx, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_gw, gw, sizeof gw, hipMemcpyHostToDevice, c->stream));
    dispatch_variant(opthin, noalpha, [&](auto OT, auto NA) {
        hipLaunchKernelGGL((k_sed_integrate<decltype(OT)::value, decltype(NA)::value>), dim3(n),
                           dim3(64), 0, c->stream, c->d_sed_wk, numin, numax, d_gx, d_gw, ngl,
                           8, c->d_sed_out);
    });
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->d_sed_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status)
        HIPCHK(hipMemcpyAsync(status, c->d_sed_status, (size_t)n * sizeof(int32_t),
                              hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}

// Measurement helper: the empirical roof of the sample arithmetic (k_roof).
extern "C" int mbb_roof_probe(mbb_ctx *c, const double pars[5], int reps, double *seconds,
                              double *lane_slots, double *clock_mhz)
{
    int rc = use(c);
    if (rc) return rc;
    if (!pars || reps <= 0 || !seconds || !lane_slots) return fail(MBB_ERR_ARG, "bad arguments");
    if (c->nb <= 0) return fail(MBB_ERR_STATE, "bands not set (mbb_set_bands)");
    const int threads = c->opt_roof_threads > 0 ? (int)c->opt_roof_threads : 512;
    const int grid = (int)(c->opt_roof_wgs > 0 ? c->opt_roof_wgs : 2) * c->cu_count;
    if ((rc = ensure_sed(c, 1, (size_t)grid * threads + 2))) return rc;
    if ((rc = run_prologue(c, pars, 1, c->opthin, c->noalpha, c->wavenorm, 0, nullptr))) return rc;
    unsigned long long *d_clk = reinterpret_cast<unsigned long long *>(c->d_sed_out + (size_t)grid * threads);
    LikeArgs a;
    memset(&a, 0, sizeof a);
    a.nu = c->d_nu; a.lnnu = c->d_lnnu; a.wt = c->d_wt;
    a.poly_b = c->d_poly_b; a.poly_c = c->d_poly_c;
    a.nchunk = c->nchunk;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) {            // the first pass warms clocks and caches
        HIPCHK(hipEventRecord(e0, c->stream));
        dispatch_variant(c->opthin, c->noalpha, [&](auto OT, auto NA) {
            hipLaunchKernelGGL((k_roof<decltype(OT)::value, decltype(NA)::value>), dim3(grid), dim3(threads),
                               0, c->stream, a, c->d_sed_wk, reps, c->d_sed_out, d_clk);
        });
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e1, c->stream));
        HIPCHK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
    *seconds = ms * 1e-3;
    *lane_slots = (double)grid * (threads / 64) * (double)reps * c->nchunk * 64.0;
    if (clock_mhz) {
        unsigned long long clk[2] = {0, 0};
        HIPCHK(hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
        *clock_mhz = clk[1] ? 100.0 * (double)clk[0] / (double)clk[1] : 0.0;
    }
    return MBB_OK;
}

extern "C" int mbb_fnu_eval(mbb_ctx *c, int opthin, int noalpha, const double *freq, int n,
                            double T, double beta, double x0, double alpha, double normfac,
                            double xmerge, double kappa, double *out)
{
    int rc = use(c);
    if (rc) return rc;
    if (n <= 0 || !freq || !out) return fail(MBB_ERR_ARG, "bad arguments");
    if ((rc = ensure_sed(c, 1, (size_t)2 * n))) return rc;
    double *d_freq = c->d_sed_out + n;
    HIPCHK(hipMemcpyAsync(d_freq, freq, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int threads = 256, grid = (n + threads - 1) / threads;
    dispatch_variant(opthin, noalpha, [&](auto OT, auto NA) {
        hipLaunchKernelGGL((k_fnu_explicit<decltype(OT)::value, decltype(NA)::value>), dim3(grid),
                           dim3(threads), 0, c->stream, d_freq, n, T, beta, x0, alpha, normfac,
                           xmerge, kappa, c->d_sed_out);
    });
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->d_sed_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}

// ---- plumbing ---------------------------------------------------------------
extern "C" int mbb_malloc(mbb_ctx *c, size_t bytes, void **dptr)
{
    int rc = use(c);
    if (rc) return rc;
    if (!dptr) return fail(MBB_ERR_ARG, "null dptr");
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 1));
    return MBB_OK;
}
extern "C" int mbb_free(mbb_ctx *c, void *dptr)
{
    int rc = use(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (dptr) HIPCHK(hipFree(dptr));
    return MBB_OK;
}
extern "C" int mbb_memcpy_h2d(mbb_ctx *c, void *dst, const void *src, size_t bytes)
{
    int rc = use(c);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}
extern "C" int mbb_memcpy_d2h(mbb_ctx *c, void *dst, const void *src, size_t bytes)
{
    int rc = use(c);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}
extern "C" int mbb_sync(mbb_ctx *c)
{
    int rc = use(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return MBB_OK;
}
extern "C" void *mbb_stream(mbb_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int mbb_event_create(mbb_ctx *c, void **ev)
{
    int rc = use(c);
    if (rc) return rc;
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void *)e;
    return MBB_OK;
}
extern "C" int mbb_event_record(mbb_ctx *c, void *ev)
{
    int rc = use(c);
    if (rc) return rc;
    HIPCHK(hipEventRecord((hipEvent_t)ev, c->stream));
    return MBB_OK;
}
extern "C" int mbb_event_elapsed_ms(mbb_ctx *c, void *start, void *stop, float *ms)
{
    int rc = use(c);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MBB_OK;
}
extern "C" int mbb_event_destroy(mbb_ctx *c, void *ev)
{
    int rc = use(c);
    if (rc) return rc;
    HIPCHK(hipEventDestroy((hipEvent_t)ev));
    return MBB_OK;
}

#ifdef MBB_STAMPS
extern "C" int mbb_stamps(mbb_ctx *c, unsigned long long *host, int nblocks)
{
    int rc = use(c);
    if (rc) return rc;
    if (!c->d_stamps) {
        HIPCHK(hipMalloc((void **)&c->d_stamps, 32 * sizeof(unsigned long long) * 65536));
        HIPCHK(hipMemset(c->d_stamps, 0, 32 * sizeof(unsigned long long) * 65536));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mbbd::g_pstamps), &c->d_stamps, sizeof(c->d_stamps)));
    }
    if (host) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(host, c->d_stamps, 32 * sizeof(unsigned long long) * nblocks, hipMemcpyDeviceToHost));
    }
    return MBB_OK;
}
#endif

extern "C" int mbb_set_option(mbb_ctx *c, const char *name, long value)
{
    if (!c || !name) return fail(MBB_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> own(c->srv_mu);
    if (c->serving) {                               // (a resident server runs with the options of its launch)
        HIPCHK(hipSetDevice(c->device));
        int rc = serve_stop(c);
        if (rc) return rc;
    }
    c->srv_hot = 0;
    if (!strcmp(name, "walkers_per_group")) c->opt_wpb = value;
    else if (!strcmp(name, "block_threads")) c->opt_threads = value;
    else if (!strcmp(name, "zero_copy")) c->opt_zero_copy = value;
    else if (!strcmp(name, "seg_chunks")) c->opt_seg_chunks = value;
    else if (!strcmp(name, "debug")) c->opt_debug = value;
    else if (!strcmp(name, "stage_tables")) c->opt_stage = value;
    else if (!strcmp(name, "spin_wait")) c->opt_spin = value;
    else if (!strcmp(name, "spin_budget")) c->opt_spin_budget = value < 0 ? 0 : value;
    else if (!strcmp(name, "pack_tails")) c->opt_pack_tails = value;
    else if (!strcmp(name, "bar_params")) c->opt_bar_params = value;
    else if (!strcmp(name, "launch_api")) c->opt_launch_api = value;
    else if (!strcmp(name, "serve_overlap")) c->opt_serve_overlap = value;
    else if (!strcmp(name, "serve")) { c->opt_serve = value; c->srv_strikes = 0; c->srv_rest = 0; }
    else if (!strcmp(name, "serve_after")) { c->opt_serve_after = value < 1 ? 1 : value; c->srv_need = 0; }
    else if (!strcmp(name, "serve_idle_us")) c->opt_serve_idle_us = value;
    else if (!strcmp(name, "serve_budget_us")) c->opt_serve_budget_us = value < 1 ? 1 : value;
    else if (!strcmp(name, "serve_lease_us")) c->opt_serve_lease_us = value < 0 ? 0 : value;
    else if (!strcmp(name, "serve_grid")) c->opt_serve_grid = value < 0 ? 0 : (int)std::min<long>(value, 1 << 20);
    else if (!strcmp(name, "serve_prefetch")) c->opt_serve_prefetch = value < 0 ? 0 : (value > 256 ? 256 : (int)value);
    else if (!strcmp(name, "prepass")) c->opt_prepass = value;
    else if (!strcmp(name, "virtual_ranks")) c->opt_vranks = value;
    else if (!strcmp(name, "xchg_spin_max")) c->x.spin_max = value;
    else if (!strcmp(name, "lookahead_sampler")) c->opt_lookahead = value;
    else if (!strcmp(name, "flow_sampler")) c->opt_flow = value;
    else if (!strcmp(name, "merged_flow_sampler")) c->opt_flowm = value;
    else if (!strcmp(name, "resident_sampler")) c->opt_flowr = value;
    else if (!strcmp(name, "resident_walkers")) c->opt_flowr_walkers = value;
    else if (!strcmp(name, "resident_ahead")) { /* (round 4's choice between forms 8 and 9: form 8 is gone, accepted and ignored) */ }
    else if (!strcmp(name, "flow_spin_log2")) c->opt_flow_spin_log2 = value;
    else if (!strcmp(name, "flow_min_steps")) c->opt_flow_min_steps = value;
    else if (!strcmp(name, "sharded_flow_sampler")) c->opt_xflow = value;
    else if (!strcmp(name, "lookahead_rows")) c->opt_la_rows = value;
    else if (!strcmp(name, "lookahead_waves")) c->opt_la_waves = value;
    else if (!strcmp(name, "roof_wgs_per_cu")) c->opt_roof_wgs = value;
    else if (!strcmp(name, "roof_threads")) c->opt_roof_threads = value;
    else return fail(MBB_ERR_ARG, "unknown option");
    return MBB_OK;
}

extern "C" int mbb_get_info(mbb_ctx *c, const char *name, long *value)
{
    if (!c || !name || !value) return fail(MBB_ERR_ARG, "null argument");
    if (!strcmp(name, "nb")) *value = c->nb;
    else if (!strcmp(name, "nseg")) *value = c->nseg;
    else if (!strcmp(name, "nunit")) *value = c->nunit;
    else if (!strcmp(name, "last_prep_ns")) *value = c->t_prep_ns;
    else if (!strcmp(name, "serving")) *value = c->serving ? 1 : 0;
    else if (!strcmp(name, "serve_requests")) *value = c->srv_requests;
    else if (!strcmp(name, "serve_fallbacks")) *value = c->srv_fallbacks;
    else if (!strcmp(name, "last_prepass")) *value = c->last_prepass;
    else if (!strcmp(name, "serve_enabled")) *value = c->opt_serve;
    else if (!strcmp(name, "device_peers")) *value = mbbh::registry_peers(c->reg_key, true);
    else if (!strcmp(name, "serve_peer_yields")) *value = c->srv_peer_yields;
    else if (!strcmp(name, "serve_resizes")) *value = c->srv_resizes;
    else if (!strcmp(name, "serve_rests")) *value = c->srv_rests;
    else if (!strcmp(name, "serve_resting")) *value = c->srv_rest;
    else if (!strcmp(name, "device_busy")) *value = c->srv_busy;
    else if (!strcmp(name, "serve_grid")) *value = c->serving ? c->srv_grid : 0;
    else if (!strcmp(name, "serve_lease_yields")) *value = c->srv_lease_yields;
    else if (!strcmp(name, "last_launch_ns")) *value = c->t_launch_ns;
    else if (!strcmp(name, "last_wait_ns")) *value = c->t_wait_ns;
    else if (!strcmp(name, "last_watch_seen")) *value = c->last_watch_seen;
    else if (!strcmp(name, "simd_chunks_max")) *value = *std::max_element(c->simd_chunks, c->simd_chunks + 4);
    else if (!strcmp(name, "simd_chunks_min")) *value = *std::min_element(c->simd_chunks, c->simd_chunks + 4);
    else if (!strcmp(name, "nchunk")) *value = c->nchunk;
    else if (!strcmp(name, "nq")) *value = c->nq;
    else if (!strcmp(name, "cu_count")) *value = c->cu_count;
    else if (!strcmp(name, "last_wpb")) *value = c->last_wpb;
    else if (!strcmp(name, "last_threads")) *value = c->last_threads;
    else if (!strcmp(name, "last_grid")) *value = c->last_grid;
    else if (!strcmp(name, "flow_fallbacks")) *value = c->flow_fallbacks;
    else if (!strcmp(name, "flow_resting")) *value = c->flow_rest;
    else if (!strcmp(name, "last_kernel_form")) *value = c->last_smode;     // k_lnlike's SMODE of the last launch
    else if (!strcmp(name, "last_workgroups_ahead")) *value = c->last_ahead;
    else if (!strcmp(name, "last_smem")) *value = c->last_smem;
    else if (!strcmp(name, "last_stage")) *value = c->last_stage;
    else if (!strcmp(name, "device")) *value = c->device;
    else if (!strcmp(name, "nranks")) *value = c->x.connected ? c->x.n : c->nranks;
    else if (!strcmp(name, "rank")) *value = c->x.connected ? c->x.rank : c->rank;
    else if (!strcmp(name, "xchg_launches")) *value = (long)c->x.seq;
    else return fail(MBB_ERR_ARG, "unknown info key");
    return MBB_OK;
}

// ---- one-hop exchange between the ranks of a sharded sampler run ----------------
// (the direct exchange on the xGMI mesh instead of a ring
// collective for a 6 KB message.)  Replaces emcee's pool, mbb_fit.py:80-81.
static int xchg_free(mbb_ctx *c)
{
    for (int r = 0; r < c->x.n; ++r)
        if (r != c->x.rank && c->x.peer[r]) (void)hipIpcCloseMemHandle(c->x.peer[r]);
    free_dev(c->x.base);
    free_dev(c->x.d_args);
    free_dev(c->x.d_flowx);
    c->x = mbb_ctx::Xchg();
    return MBB_OK;
}

extern "C" int mbb_xchg_open(mbb_ctx *c, int nranks, int rank, int max_rows, unsigned char handle[64])
{
    int rc = use(c);
    if (rc) return rc;
    if (nranks < 1 || nranks > 16 || rank < 0 || rank >= nranks || max_rows <= 0 || !handle)
        return fail(MBB_ERR_ARG, "bad exchange layout (at most 16 ranks)");
    if (c->x.base) return fail(MBB_ERR_STATE, "this context already has an exchange (mbb_xchg_close first)");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    const size_t bytes = mbb_ctx::Xchg::kHeader + (size_t)max_rows * 6 * sizeof(double) +
                         spec_words((size_t)max_rows) * sizeof(double);
    // fine-grained: peers' system-scope stores must be visible to this device's loads while
    // kernels run on both sides
    HIPCHK(hipExtMallocWithFlags((void **)&c->x.base, bytes, hipDeviceMallocFinegrained));
    HIPCHK(hipMemset(c->x.base, 0, bytes));
    HIPCHK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, c->x.base);
    if (e != hipSuccess) { free_dev(c->x.base); c->x.base = nullptr; return fail(MBB_ERR_HIP, "hipIpcGetMemHandle", e); }
    memcpy(handle, &h, 64);
    c->x.n = nranks; c->x.rank = rank; c->x.cap_rows = (size_t)max_rows; c->x.connected = 0; c->x.seq = 0;
    c->x.peer[rank] = c->x.base;
    return MBB_OK;
}

// handles: nranks x 64 bytes, entry r as returned by rank r's mbb_xchg_open (entry `rank` is ignored).
// Every rank must have returned from mbb_xchg_open before any calls this.
extern "C" int mbb_xchg_connect(mbb_ctx *c, const unsigned char *handles)
{
    int rc = use(c);
    if (rc) return rc;
    if (!c->x.base || !handles) return fail(MBB_ERR_STATE, "mbb_xchg_open first");
    if (c->x.connected) return fail(MBB_ERR_STATE, "exchange already connected");
    for (int r = 0; r < c->x.n; ++r) {
        if (r == c->x.rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * 64, 64);
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(MBB_ERR_HIP, "hipIpcOpenMemHandle", e);
        c->x.peer[r] = (unsigned char *)p;
    }
    HIPCHK(hipMalloc((void **)&c->x.d_args, sizeof(XchgArgs)));
    HIPCHK(hipMalloc((void **)&c->x.d_flowx, sizeof(FlowX)));
    c->x.connected = 1;
    return MBB_OK;
}

extern "C" int mbb_xchg_close(mbb_ctx *c)
{
    int rc = use(c);
    if (rc) return rc;
    if (c->x.users > 0) return fail(MBB_ERR_STATE, "a sampler still lives in the exchange buffer (mbb_sampler_destroy first)");
    HIPCHK(hipStreamSynchronize(c->stream));
    return xchg_free(c);
}

// ---- RCCL -------------------------------------------------------------------
extern "C" int mbb_comm_unique_id(char id[128])
{
    int rc = load_rccl();
    if (rc) return rc;
    UniqueId u;
    memset(&u, 0, sizeof u);
    int r = g_rccl.GetUniqueId(&u);
    if (r != 0) return fail(MBB_ERR_RCCL, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclGetUniqueId failed");
    memcpy(id, u.internal, 128);
    return MBB_OK;
}

extern "C" int mbb_comm_init(mbb_ctx *c, int nranks, int rank, const char id[128])
{
    int rc = use(c);
    if (rc) return rc;
    if (nranks < 1 || rank < 0 || rank >= nranks || !id) return fail(MBB_ERR_ARG, "bad rank layout");
    if (c->comm) return fail(MBB_ERR_STATE, "this context already has a communicator (mbb_comm_destroy first)");
    if ((rc = load_rccl())) return rc;
    UniqueId u;
    memcpy(u.internal, id, 128);
    int r = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
    if (r != 0) return fail(MBB_ERR_RCCL, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclCommInitRank failed");
    c->nranks = nranks;
    c->rank = rank;
    return MBB_OK;
}

extern "C" int mbb_comm_destroy(mbb_ctx *c)
{
    int rc = use(c);
    if (rc) return rc;
    if (c->comm) {
        HIPCHK(hipStreamSynchronize(c->stream));
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->nranks = 1;
    c->rank = 0;
    return MBB_OK;
}

extern "C" int mbb_allgather_f64(mbb_ctx *c, const double *d_send, double *d_recv, int count);

// One sharded half-step in one call: this rank's n walkers through the fused
// kernel, then the all-gather of their log-probabilities, both on the context's
// stream.  d_all holds nranks*n doubles, rank-major; d_lnl is this rank's slice
// of d_all (in-place all-gather) or a separate buffer.
extern "C" int mbb_lnlike_allgather_device(mbb_ctx *c, const double *d_pars, int n, double *d_lnl,
                                           int32_t *d_status, double *d_all)
{
    int rc = use(c);
    if (rc) return rc;
    if (n <= 0 || !d_pars || !d_lnl || !d_all) return fail(MBB_ERR_ARG, "bad batch buffers");
    if ((rc = launch_lnlike(c, d_pars, n, d_lnl, d_status, nullptr))) return rc;
    return mbb_allgather_f64(c, d_lnl, d_all, n);
}

// The sharded boundary as ONE call, host arrays in and out (what emcee's pool does per half-step,
// mbb_fit.py:80-81: every worker evaluates its share of the rows, everybody gets all the answers): this
// rank's n parameter rows go to the device without a copy command (through the BAR, or read by the kernel
// from pinned memory), the fused kernel writes its lnprob into this rank's slice of the gather buffer and
// its row status straight into pinned host memory, ONE in-place ncclAllGather of n doubles per rank, one
// copy of the nranks*n gathered values to a pinned landing buffer, one stream wait.
extern "C" int mbb_lnlike_allgather(mbb_ctx *c, const double *pars, int n, double *all, int32_t *status)
{
    int rc = use(c);
    if (rc) return rc;
    if (n <= 0 || !pars || !all) return fail(MBB_ERR_ARG, "bad batch buffers");
    if (c->nb <= 0) return fail(MBB_ERR_STATE, "bands not set (mbb_set_bands)");
    const int nranks = c->comm ? c->nranks : 1, rank = c->comm ? c->rank : 0;
    if (!c->comm && c->nranks != 1) return fail(MBB_ERR_STATE, "communicator not initialised");
    if ((rc = ensure_capacity(c, (size_t)n, false))) return rc;
    const size_t total = (size_t)nranks * n;
    if (total > c->gather_cap) {
        size_t cap = c->gather_cap ? c->gather_cap : 1024;
        while (cap < total) cap *= 2;
        HIPCHK(hipStreamSynchronize(c->stream));
        free_dev(c->d_gather); free_host(c->h_gather);
        c->d_gather = c->h_gather = nullptr; c->gather_cap = 0;
        HIPCHK(hipMalloc((void **)&c->d_gather, cap * sizeof(double)));
        HIPCHK(hipHostMalloc((void **)&c->h_gather, cap * sizeof(double), hipHostMallocDefault));
        c->gather_cap = cap;
    }
    const size_t nbytes = (size_t)n * 5 * sizeof(double);
    const bool push = c->opt_zero_copy && c->opt_bar_params && c->w_pars;
    double *dp;
    int32_t *ds;
    if (push) {
        memcpy(c->w_pars, pars, nbytes);
        __builtin_ia32_sfence();
        dp = c->w_pars;
    } else {
        memcpy(c->h_pars, pars, nbytes);
        if (c->opt_zero_copy) HIPCHK(hipHostGetDevicePointer((void **)&dp, c->h_pars, 0));
        else {
            HIPCHK(hipMemcpyAsync(c->d_pars, c->h_pars, nbytes, hipMemcpyHostToDevice, c->stream));
            dp = c->d_pars;
        }
    }
    if (c->opt_zero_copy) HIPCHK(hipHostGetDevicePointer((void **)&ds, c->h_status, 0));
    else ds = c->d_status;
    double *mine = c->d_gather + (size_t)rank * n;
    if ((rc = launch_lnlike(c, dp, n, mine, ds, nullptr))) return rc;
    if ((rc = mbb_allgather_f64(c, mine, c->d_gather, n))) return rc;
    HIPCHK(hipMemcpyAsync(c->h_gather, c->d_gather, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (!c->opt_zero_copy)
        HIPCHK(hipMemcpyAsync(c->h_status, c->d_status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if ((rc = wait_stream(c))) return rc;
    memcpy(all, c->h_gather, total * sizeof(double));
    if (status) memcpy(status, c->h_status, (size_t)n * sizeof(int32_t));
    return MBB_OK;
}

extern "C" int mbb_allgather_f64(mbb_ctx *c, const double *d_send, double *d_recv, int count)
{
    int rc = use(c);
    if (rc) return rc;
    if (!d_send || !d_recv || count < 0) return fail(MBB_ERR_ARG, "bad buffers");
    if (!c->comm) {
        if (c->nranks != 1) return fail(MBB_ERR_STATE, "communicator not initialised");
        if (d_send != d_recv)
            HIPCHK(hipMemcpyAsync(d_recv, d_send, (size_t)count * sizeof(double),
                                  hipMemcpyDeviceToDevice, c->stream));
        return MBB_OK;
    }
    int r = g_rccl.AllGather(d_send, d_recv, (size_t)count, 8 /* ncclFloat64 */, c->comm, c->stream);
    if (r != 0) return fail(MBB_ERR_RCCL, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclAllGather failed");
    return MBB_OK;
}
