// mbb_kernels.hip.h -- the HIP kernels of the likelihood hot path (gfx950).
//
// k_lnlike       the fused per-walker likelihood (and, SAMPLER, the stretch-move
//                half-step): n calls of likelihood.__call__ (likelihood.py:790-834)
// k_prologue     modified_blackbody.__init__ + max_wave for n rows
// k_sed_eval     f_nu on a frequency grid for n rows
// k_sed_integrate  freq_integrate for n rows
// k_fnu_explicit fnu.pyx's four functions with explicit scalars
//
// Included by mbb_hip.hip (the C-ABI / host side) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mbb_device.hip.h"
#include "mbb_flow_index.h"

using namespace mbbd;

constexpr int kPolyBDoubles = (mbbm::kPolyBMax * 8 + 1) * mbbm::kPolyStride;      // mbbh::kPolyBCount rows of kPolyStride doubles
constexpr int kPolyCDoubles = (mbbm::kPolyCMax * 8 + 1) * mbbm::kPolyStride;      // mbbh::kPolyCCount

// ---------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------
// One-hop exchange of the moved state rows between the ranks of a sharded run.  Every rank
// holds the whole ensemble in a fine-grained buffer its peers have mapped (hipIpc*): a
// walker that moves is stored into every rank's copy by the lane that accepted it, system
// scope; the last walker of the launch to have done so raises this launch's number in
// every peer's flag word, and the next launch's prologue waits for all peers' flags
// before it reads a row.  Lives in device memory, written once per enqueued run.
struct XchgArgs {
    double *xpos[16];               // every rank's pos6 (entry xrank is this rank's own)
    unsigned long long *xflag[16];  // every rank's flag array [xn]; a rank writes word [xrank] of each
    unsigned int *xcount;           // this rank's arrival counter (walkers of the launch done)
    int xn, xrank;
    unsigned long long xseq0;       // launches posted before this run: launch (step, half) is
                                    // number xseq0 + 2 step + half + 1, waits for the one before
    long long xspin_max;            // polls before a waiting launch gives up (errflag = 8)
};

struct LikeArgs {
    // passband tables, chunk-padded: band b owns chunks; every chunk is 64 samples
    const double *nu;         // [nchunk*64] GHz   (padding: 1.0)
    const double *lnnu;       // [nchunk*64] log(nu)  (padding: 0.0)
    const double *wt;         // [nchunk*64] sedmult*normfac (padding: 0.0)
    const double *poly_b;     // [kPolyBCount*kPolyStride] piecewise polynomials of x^3/expm1(x) (mbb_host_tables.h)
    const double *poly_c;     // [kPolyCCount*kPolyStride] ... of 1 - e^-y
    const int4 *unit_tab;     // [nunit] {result slot, first chunk, end chunk, kind} in dealing order;
                              // kind 0: a segment, reduced to one slot; 1: a chunk of 64 single-sample
                              // bands, lane l's value goes to slot + l
                              // 2: a tail chunk (slot field = its index k): row r of 16 lanes is the
                              // leftover of some band and goes to tail_slot[4k + r] (-1: unused row)
    const int2 *band_rng;     // [nb] the band's result slots [s0, s1)
    const int32_t *tail_slot; // [4 * tail chunks]
    const double *flux;       // [nb]
    const double *ivar;       // [nb]
    const double *invcov;     // [nb*nb] or nullptr
    int cov_in_lds;           // C^-1 copied to LDS (it fits) or read from global
    int nb, nunit, npart, nchunk;   // units and result slots per walker
    double nunorm;            // um_to_GHz / wavenorm, GHz
    double lnunorm;           // log(nunorm)
    double lowlim[5];
    double uplim[6];
    double gmean[6];
    double givar[6];
    uint32_t has_uplim;       // bit i
    uint32_t has_gprior;      // bit i
    // batch
    union {
        const double *pars;   // [n*5]  (SMODE 0)
        unsigned long long flow_serial;   // SMODE 5: number of this launch in the context's life, in the
                              // check words of the records (a record left by an earlier run never fits)
    };
    int n;
    int wpb;                  // walkers per block
    int debug;                // status carries root-finder iterations << 8
    double *lnl;              // [n]
    int32_t *status;          // [n] or nullptr
    double *model_flux;       // [n*nb] or nullptr
#ifdef MBB_STAMPS
    unsigned long long *stamps;   // diagnostic build only: [grid*8] s_memtime values
#endif
    // ---- stretch-move half-step (SAMPLER instantiation only) ----------------
    // State rows are (T, beta, lambda0, alpha, fnorm, lnprob).  This launch moves
    // m_count rows per source starting at s_begin, using partners drawn from
    // [c_begin, c_begin + c_count).  chain6 / nacc are launch-local: entry w belongs to
    // the w-th walker of this launch (the host keeps the map back to rows), so that a
    // rank of a sharded run owns one contiguous block it can all-gather in place.
    double *pos6;             // [nw*6]
    double *chain6;           // [n*6] or nullptr: this launch's slot of the chain
    unsigned int *nacc;       // [n] accepted moves of this launch's walkers
    int *errflag;             // set to a row status >= 2 if lnprob is NaN / invalid
    int s_begin, c_begin, c_count, m_count, nw;
    int step, half;
    // (arguments of variants that never meet share storage, so that the block stays at 480
    // bytes for every variant)
    int persist;              // SMODE 5, 6, form 7: half-steps in this launch (step = number of the first step)
    double *spec;             // SMODE 5, 6, form 7: the one-launch run's device state (FlowView / FlowX / FlowMView)
    // ---- independent sources sharing the band tables (cfg5): flux/ivar are
    // [nsrc*nb]; plain mode: source = row / rows_per_src; sampler mode: the state is
    // [nsrc][nw_src][6] and a launch covers nsrc * c_count walkers
    int nsrc, rows_per_src, nw_src;
    double stretch_a;
    unsigned long long seed;
    // ---- one-hop exchange between the ranks of a sharded run (SMODE 2 only; mbb_xchg_*):
    // kept out of the argument block -- every launch of every variant pays for the size of
    // that block (48 bytes more, crossing 512, cost the single-GPU sampler 2.7 % per step:
    // tools/lat_kernarg.hip, profiles/r02/lat_kernarg.txt)
    union {
        const XchgArgs *xargs;
        // ---- one-launch runs (SMODE 5, 6, form 7): see k_lnlike, k_flowm
        struct {
            int spec_cfg;     // bit 0: form 7, which of the two sets of completion counters this launch uses;
                              // bits 8-15: candidates per wave (1, 2 or 4 rows of 16 lanes),
                              // bits 16-23: waves of a workgroup working ahead that take candidates,
                              // bits 24-31: SMODE 5, log2 of the polls before a wait gives up (0: 22)
            int n_ahead;      // workgroups 0 .. n_ahead-1 work ahead (dispatched first: theirs is the
                              // longer path), the rest move walkers
        };
    };
};
// k_serve's result records (pinned host memory): {lnl, status} every kSrvStride doubles -- a cache line each
// (profiles/r05/served_boundary.txt; -DMBB_SRV_STRIDE=2: packed, four to a line, for the A/B)
#ifndef MBB_WC_ROW
#define MBB_WC_ROW true          // mbb_walker_consts.inc: the constructor on a row of 16 lanes (k_walker_pre: on one)
#endif
// k_walker_pre's records: {WalkerK: 13 words, pen_u, pen_g, -}
constexpr int kPreWords = 16;
static_assert(sizeof(mbbd::WalkerK) == 13 * sizeof(double), "a walker's record is 13 words");
#ifndef MBB_SRV_STRIDE
#define MBB_SRV_STRIDE 8
#endif
constexpr int kSrvStride = MBB_SRV_STRIDE;
#ifndef MBB_STAMPS
static_assert(sizeof(LikeArgs) == 480, "the argument block: every launch of every variant pays for its size");
#endif
// Device state of a one-launch run, one allocation of 8-byte words (nw = state rows).
// A record: WalkerK (13 words), proposal (5), 4 ln z, ln u, the two penalties = 22 elements.
// SMODE 5 (FlowView): everything a row publishes is indexed by the number m of the move it
// belongs to, mod kFlowSlots -- a mover more than four half-steps ahead of the slowest waits, so
// four slots are never overwritten under a reader:
//   rec   [nw][kFlowSlots][2][kFlowRec]  the proposal records of move m, one per candidate: word
//                   2c is element c of the record, word 2c + 1 is
//                   (half-step of the move + 1) XOR that element's bits -- a reader takes an element
//                   when the pair fits, whenever and in whatever order the two stores arrive, so
//                   the writer neither waits for its stores nor raises a flag after them
//   st    [kFlowSlots][nw][8]            the row after move m (slot 0: as the run found it)
//   seq   [nw]      the half-step after the last one whose row has LANDED in st (0: none)
//   done  [8][16]   this GPU's moves completed per half-step mod 8; the last one of a half-step says
//                   so in pub (the lag guard)
//   mseq  [nw][kFlowSlots]  2 x (half-step of move m + 1) + (it was accepted): written the moment
//                   the move is decided, before the row itself
//   pub   [16][8]   per rank of a sharded run (SMODE 6; one rank otherwise) and half-step mod 8: that
//                   half-step + 1 once all of the rank's moves of it are complete
//   startf, endf [16]  per rank: the number of the run whose state is set up / whose launch has ended
// In a sharded run every rank holds all of it (its own records only) in memory its peers have
// mapped, and whoever publishes a decision, a row or a word stores it into every rank's copy.
constexpr int kFlowRecN = 22, kFlowRec = 48;
struct FlowView {
    double *rec, *st;
    unsigned long long *seq, *done, *mseq, *pub, *startf, *endf;
};
__host__ __device__ constexpr size_t spec_words(size_t nw)
{
    return nw * ((size_t)kFlowSlots * 2 * kFlowRec + kFlowSlots * 8 + 1 + kFlowSlots) + 8 * 16 + 16 * 8 + 32;
}
__host__ __device__ __forceinline__ FlowView flow_view(double *spec, int nw)
{
    FlowView v;
    v.rec = spec;
    v.st = spec + (size_t)nw * kFlowSlots * 2 * kFlowRec;
    v.seq = reinterpret_cast<unsigned long long *>(v.st + (size_t)nw * kFlowSlots * 8);
    v.done = v.seq + nw;
    v.mseq = v.done + 8 * 16;
    v.pub = v.mseq + (size_t)nw * kFlowSlots;
    v.startf = v.pub + 16 * 8;
    v.endf = v.startf + 16;
    return v;
}
// A sharded one-launch run (SMODE 6): where every rank's copy is mapped here.  In device memory;
// a.spec points at it (and base[rank] is this rank's own copy).
struct FlowX {
    double *base[16];
    int n, rank;
    unsigned long long run;      // number of this run (the same on every rank)
};
// (flow_cnt, flow_seq: mbb_flow_index.h)

__device__ __forceinline__ double ld_sys(const double *p)      // system-scope load (bypasses L1)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(double *p, double v)    // system-scope (write-through) store
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ld_dev(const double *p)      // device-scope: sc1 load, L2-served
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_dev(double *p, double v)    // device-scope: sc1 store, through to L2
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// Role-local view of the kernel arguments (the one-launch sampler kernels: k_lnlike SMODE 5/6, k_flowm).
// Their workgroups or waves take roles that share one kernel and one 480-byte argument block, and the
// compiler fetches a by-value kernel argument where the function begins: every field any role touches is
// then live in scalar registers across the branch to the roles -- far more than the 102 there are -- and
// the overflow comes back as v_readlane wherever the spiller puts it.  So each role reads the block
// through a pointer to the kernel-argument segment (constant address space: scalar loads) that it
// launders at its own top: nothing a role needs can be fetched before its block is entered and nothing
// another role needs is live inside it.
// A field named with MBB_PIN at the top of a role arrives with the rest of the role's batch; one that is
// first touched inside a branch is a scalar load -- a round trip to the scalar cache -- where it is used,
// and in these kernels that is on the chain between a decision and its publication (measured: role-local
// arguments WITHOUT the pins cost form 7 +6 % per step, with them -2.3 %; profiles/r03/ab_role_args.txt).
// So every field of a role's hot path is pinned (the later reads of the same address are the same value
// to the compiler); what depends on run-time switches (priors, limits) stays a load inside its branch.
struct LikeArgs;
typedef const __attribute__((address_space(4))) LikeArgs CLikeArgs;
__device__ __forceinline__ CLikeArgs *role_args(CLikeArgs *p)
{
    asm volatile("" : "+s"(p));
    return p;
}
#define MBB_KERNARGS() ((CLikeArgs *)__builtin_amdgcn_kernarg_segment_ptr())
#define MBB_ROLE_ARGS() CLikeArgs &a = *role_args(ka)
#define MBB_PIN(x) asm volatile("" ::"s"(x))

// Philox4x32-10 (Salmon et al. 2011), counter = (row, 2 step + half), key = seed.
__device__ __forceinline__ void philox4x32(unsigned int c[4], unsigned int k0, unsigned int k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a high and a low
        // half: integer multiplies are quarter rate and this sits on the latency path
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
        const unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
        const unsigned int n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// The stretch move's proposal c - z (c - s), one rounding per coordinate wherever it is formed.
__device__ __forceinline__ double stretch_q(double cv, double sv, double zz)
{
    return __builtin_fma(-zz, cv - sv, cv);
}

// Philox draw of state row `row` at half-step (step, half): z of the stretch move, the
// partner's index in the other half, the uniform of the accept test.  The key (`seed`) carries the step's number in
// the sampler's LIFE, the counter the row and the half: what is drawn for a step does not depend on how a run was cut
// into launches -- run_mcmc(p0, 64) twice and run_mcmc(p0, 128) make the same chain (rounds 1-5 also counted the
// step's place in its launch, `step`: another grouping was another chain -- ADVICE r05).
__device__ __forceinline__ void stretch_draw(int row, int step, int half, unsigned long long seed,
                                             double stretch_a, int c_count, double &zz, int &pj, double &u3)
{
    (void)step;
    unsigned int c4[4] = {(unsigned int)row, (unsigned int)half, 0u, 0u};
    philox4x32(c4, (unsigned int)seed, (unsigned int)(seed >> 32));
    const double u1 = fma((double)(c4[0] >> 5), 67108864.0, (double)(c4[1] >> 6)) * (1.0 / 9007199254740992.0);   // (exact)
    const double u2 = (double)c4[2] * (1.0 / 4294967296.0);
    u3 = ((double)c4[3] + 0.5) * (1.0 / 4294967296.0);
    const double sq = fma(stretch_a - 1.0, u1, 1.0);
    zz = sq * sq / stretch_a;
    pj = (int)(u2 * (double)c_count);
    if (pj >= c_count) pj = c_count - 1;
}

// Block = blockDim.x/64 waves working on `wpb` consecutive walkers.
//   phase 1: prologue, one row of 16 lanes per walker     -> LDS
//   phase 2: (walker, segment) units dealt round-robin to waves; a lane strides
//            over the segment's samples, then one wave64 shuffle reduction
//   phase 3: band sums in fixed order, then one lane per walker forms lnL
// Summation order depends only on the band tables, never on the batch, so a
// walker's result is bitwise independent of which launch / GPU evaluates it.
// SMODE: 0 the likelihood of given rows; 1 the stretch-move half-step; 2 the half-step of a
// sharded run with the one-hop exchange (its own instantiation: carried as run-time
// branches and extra arguments in the single-GPU sampler kernel it cost that kernel 0.75 us
// per launch).
// 5 and 6, the one-launch look-ahead run: a whole run of half-steps in ONE
// launch, every workgroup resident; the proposals of the NEXT half-step -- draw, SED constructor,
// penalties -- are prepared while this one is being decided, for both outcomes of each partner's
// pending move, by workgroups of their own (blockIdx < n_ahead); a mover picks the record its
// partner's decision points at and starts at the quadrature; a row's half-step starts when the
// rows it depends on are done (FlowView: per-row words polled with bounded spins, write-through
// stores, no grid-wide barrier).  6: the same across the ranks of a sharded ensemble.  Bitwise the
// chain of SMODE 1; 9.9 us per step against 15.6.
// (Rounds 1-2 also had SMODE 3, a one-launch run behind a grid-wide hand-off, and SMODE 4, the
// look-ahead as extra workgroups of every launch of a train: both measured slower than what
// replaced them -- profiles/r02/persistent_sampler.txt, lookahead_notes.txt -- and removed in round 3.)
template <bool OPTHIN, bool NOALPHA, int SMODE, bool STAGE>
__global__ void __launch_bounds__(1024) k_lnlike(const LikeArgs a_val)
{
    static_assert(SMODE == 0 || SMODE == 1 || SMODE == 2 || SMODE == 5 || SMODE == 6, "no such sampler form");
    constexpr bool SAMPLER = SMODE != 0, XCHG = SMODE == 2, XF = SMODE == 6, FLOW = SMODE == 5 || XF;
    // The one-launch forms have two kinds of workgroups (movers, and those that work ahead) in one kernel:
    // they read the argument block through role-local views (see MBB_ROLE_ARGS); the others take it by value.
    CLikeArgs *const ka = MBB_KERNARGS();
    auto &a = *[&]() { if constexpr (FLOW) return role_args(ka); else return &a_val; }();
    // SMODE 6, the one-launch run of a sharded ensemble: this rank's copy of the run's state and
    // its peers'; words and rows that cross GPUs are read and written at system scope
    const FlowX *const fx = XF ? reinterpret_cast<const FlowX *>(a.spec) : nullptr;
    const int npeer = XF ? fx->n : 1, xrank = XF ? fx->rank : 0;
    auto peer_view = [&](int pr) { return flow_view(XF ? fx->base[pr] : a.spec, a.nw); };
    auto fl_ld = [&](const double *q) { if constexpr (XF) return ld_sys(q); else return ld_dev(q); };
    auto fl_st = [&](double *q, double v) { if constexpr (XF) st_sys(q, v); else st_dev(q, v); };
    auto fl_ldw = [&](const unsigned long long *q) {
        if constexpr (XF) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto fl_stw = [&](unsigned long long *q, unsigned long long v) {
        if constexpr (XF) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ __align__(16) double s_tab[kExp2N];                     // 2^(j/256) for the sample loop
    __shared__ __align__(16) double s_pb[kPolyBDoubles];    // x^3/expm1(x), piecewise degree 7
    __shared__ __align__(16) double s_pc[OPTHIN ? 2 : kPolyCDoubles];   // 1 - e^-y (thick only)
    const int W = a.wpb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwave = blockDim.x >> 6;
    const int nun = a.nunit, npart = a.npart, nb = a.nb;
    WalkerK *wk = reinterpret_cast<WalkerK *>(smem_raw);
    double *partial = reinterpret_cast<double *>(wk + W);   // [W*npart]
    double *mflux = partial + (size_t)W * npart;            // [W*nb]
    double *pen = mflux + (size_t)W * nb;                   // [W*2]
    double *s_flux = pen + 2 * (size_t)W;                   // [nb]
    double *s_ivar = s_flux + nb;                           // [nb]
    double *s_invcov = s_ivar + nb;                         // [nb*nb] when a.invcov
    int2 *s_band = reinterpret_cast<int2 *>(s_invcov + (a.cov_in_lds ? (size_t)nb * nb : 0));  // [nb]
    // SAMPLER: per walker the proposal q[5], (dim-1) log z, old lnprob, log u
    double *prop = reinterpret_cast<double *>(s_band + nb + 1);              // [W*8]
    // STAGE: the passband tables themselves (nu, log nu, weight), [nchunk*64] each
    // (offset arithmetic on smem_raw, not on a pointer cast to an integer: the latter
    // loses the LDS address space and every table read becomes a flat load)
    const size_t tab_off = ((size_t)(reinterpret_cast<unsigned char *>(prop + 8 * (size_t)W) - smem_raw) + 15) &
                           ~(size_t)15;
    double *s_nu = reinterpret_cast<double *>(smem_raw + tab_off);
    double *s_lnnu = s_nu + (STAGE ? a.nchunk * 64 : 0);
    double *s_wt = s_lnnu + (STAGE ? a.nchunk * 64 : 0);
    const int w0 = (FLOW ? (int)blockIdx.x - a.n_ahead : (int)blockIdx.x) * W;
    // SMODE 5: polls before a wait gives up and ends the run with error 9 (~1.5 us each)
    // (taken from the argument block where it is needed, not kept in registers across the chains; once
    // the error flag is up, every wait notices within a few polls and the run drains)
#define MBB_FLOW_SPIN_LIMIT (1ll << (((a.spec_cfg >> 24) & 0x3f) ? ((a.spec_cfg >> 24) & 0x3f) : 22))
    // The compiler fetches kernel arguments where they are first used, one exposed
    // scalar-cache round trip (~200 cycles) each; on the latency path that is a
    // dozen of them.  Ask for the hot ones here so that they arrive in one batch.
#define PIN(x) asm volatile("" ::"s"(x))
#define MBB_LNLIKE_PINS() do { \
    PIN(a.n); PIN(a.pars); PIN(a.nunorm); PIN(a.lnunorm); PIN(a.has_uplim); PIN(a.has_gprior); \
    PIN(a.lowlim[0]); PIN(a.lowlim[1]); PIN(a.lowlim[2]); PIN(a.lowlim[3]); PIN(a.lowlim[4]); \
    PIN(a.unit_tab); PIN(a.lnl); PIN(a.status); PIN(a.model_flux); PIN(a.invcov); PIN(a.nsrc); \
    PIN(a.debug); PIN(a.flux); PIN(a.ivar); PIN(a.rows_per_src); } while (0)
    if constexpr (!FLOW) {
    PIN(a.n); PIN(a.pars); PIN(a.nunorm); PIN(a.lnunorm); PIN(a.has_uplim); PIN(a.has_gprior);
    PIN(a.lowlim[0]); PIN(a.lowlim[1]); PIN(a.lowlim[2]); PIN(a.lowlim[3]); PIN(a.lowlim[4]);
    PIN(a.unit_tab); PIN(a.lnl); PIN(a.status); PIN(a.model_flux); PIN(a.invcov); PIN(a.nsrc);
    PIN(a.debug); PIN(a.flux); PIN(a.ivar); PIN(a.rows_per_src);
    }
#ifdef MBB_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();   // before the kernarg arrives
#define STAMP(i) do { if (tid == 0 && a.stamps && blockIdx.x < 65536) a.stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMPD(i, dep) do { if (tid == 0 && a.stamps && blockIdx.x < 65536) { asm volatile("" ::"v"(dep)); a.stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
    if (tid == 0 && a.stamps && blockIdx.x < 65536) a.stamps[blockIdx.x * 32 + 7] = t_entry;
// every wave's own stamp number `page` (1..7): slot 16 + wave of block blockIdx + 256 page
#define WSTAMP(page) do { if ((tid & 63) == 0 && a.stamps && blockIdx.x < 256) a.stamps[(blockIdx.x + 256 * (page)) * 32 + 16 + (tid >> 6)] = __builtin_amdgcn_s_memtime(); } while (0)
#define WSTAMPD(page, dep) do { asm volatile("" ::"v"(dep)); WSTAMP(page); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMPD(i, dep) do { } while (0)
#define WSTAMP(page) do { } while (0)
#define WSTAMPD(page, dep) do { } while (0)
#endif
    STAMP(0);

    // ---- the workgroups that work ahead --------------------------------------------------
    // Row (16 lanes) `pair` = (walker, candidate): the walker's proposal for its next half-step,
    // its SED constants and penalties, under the assumption that its partner -- moving in the
    // half-step before -- stays (candidate 0) or moves to the proposal it is being tested on
    // (candidate 1; formed again here from the same draw and the same rows, so it is the value
    // the mover's own record holds).
    if constexpr (FLOW) {
        if ((int)blockIdx.x < a.n_ahead) {
            MBB_ROLE_ARGS();
            PIN(a.spec_cfg); PIN(a.m_count); PIN(a.persist); PIN(a.c_count); PIN(a.s_begin); PIN(a.step); PIN(a.seed);
            PIN(a.stretch_a); PIN(a.nw); PIN(a.flow_serial); PIN(a.errflag); PIN(a.nunorm); PIN(a.lnunorm);
            PIN(a.has_uplim); PIN(a.has_gprior);
            PIN(a.lowlim[0]); PIN(a.lowlim[1]); PIN(a.lowlim[2]); PIN(a.lowlim[3]); PIN(a.lowlim[4]);
            const int rpw = (a.spec_cfg >> 8) & 0xff, aw = (a.spec_cfg >> 16) & 0xff;
            const int pair = ((int)blockIdx.x * aw + wave) * rpw + (lane >> 4);
            // a row of lanes keeps to one half of the ensemble, every other half-step (a proposal
            // takes about a half-step to prepare; what can be fetched before the decision it waits
            // for is fetched during the half-step in between)
            const int wh = (pair >> 1) & 1;
            const bool active = wave < aw && (lane >> 4) < rpw && pair < 4 * a.m_count;
            const int loc = pair >> 2, cand = pair & 1;
            if (wave >= aw) return;                           // nothing is shared, spare waves leave
            // SMODE 5: half-step j of the run is prepared as soon as the rows it starts from are
            // there -- the state as of the start of half-step j - 1 -- while j - 1 is still moving
            const FlowView fv = peer_view(xrank);
            const int nj = a.persist;
            for (int j = wh; j < nj; j += 2) {
            // the draws: the walker's own for the half-step being prepared, and the one its
            // partner is moving on meanwhile
            const int hj = j & 1;
            const int sb = hj ? a.c_count : 0;      // the half that moves then / the other
            const int ob = hj ? 0 : a.c_count;
            const int tn = a.step + (j >> 1);
            const unsigned long long seed_n = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(j >> 1);
            const int tp = a.step + ((j - 1) >> 1), hp = hj ^ 1;
            const unsigned long long seed_p = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)((j - 1) >> 1);
            const bool c1 = cand && j > 0;
            const int rown = sb + a.s_begin + loc;      // (s_begin = this rank's offset in a half)
            STAMP(11);
            double zz = 1.0, u3 = 0.5, zp = 1.0, up;
            int pj = 0, pjp = 0;
            if (active) {
                stretch_draw(rown, tn, hj, seed_n, a.stretch_a, a.c_count, zz, pj, u3);
                if (c1) stretch_draw(ob + pj, tp, hp, seed_p, a.stretch_a, a.c_count, zp, pjp, up);
            }
            double snv[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, cpos[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, cpv[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            int m_next = 0;                                   // the number of the move being prepared
            {
                // The rows this proposal starts from -- the walker's own and, for candidate 1, its
                // partner's partner -- made their last move in half-step j - 2.  Such a row is put
                // together here from what was known before that move was decided: the row as it
                // was, and the proposal it was tested on -- formed again from that row, its partner's
                // row of the time and the same draw, so the same bits as the record its mover used --
                // so that when the decision arrives nothing is left to fetch: the constructor starts
                // one hand-off after the decision.  (Only rows and decision words are read: what a
                // sharded run would have to replicate.)
                const int m_s = flow_cnt(hj, j - 1), m_o = flow_cnt(hj ^ 1, j - 1);   // moves made, as of j - 1
                m_next = m_s + 1;
                const int g = j - 2;                                                  // half-step of move m_s
                const int m_q = flow_cnt(hj ^ 1, g);                                  // the other half's moves before g
                const int l16 = lane & 15, base = lane & 48;
                const int pprow = sb + pjp, prow = ob + pj;
                int qr = 0, qp = 0;                           // partners of the two rows in half-step g,
                double zr = 1.0, zq = 1.0;                    // and their stretch factors
                if (active && m_s > 0) {
                    const int tg = a.step + (g >> 1);
                    const unsigned long long seed_g = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(g >> 1);
                    double u0;
                    stretch_draw(rown, tg, hj, seed_g, a.stretch_a, a.c_count, zr, qr, u0);
                    if (c1) stretch_draw(pprow, tg, hj, seed_g, a.stretch_a, a.c_count, zq, qp, u0);
                }
                auto spin = [&](const unsigned long long *word, unsigned long long need, bool watch, int shift) {
                    unsigned long long v = 0;
                    long long spins = 0;
                    for (;;) {
                        bool ok = true;
                        if (watch) { v = fl_ldw(word); ok = (v >> shift) >= need; }
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                        ++spins;
                        if (spins > MBB_FLOW_SPIN_LIMIT ||
                            ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            atomicMax(a.errflag, 9);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    return v;
                };
                // A row's last proposal comes from its record where that is on this GPU (always the
                // walker's own; its partner's partner's unless the run is sharded): the candidate its
                // mover took is the one the earlier decision of ITS partner says.  A row of another
                // rank's has no record here: its proposal is formed again from the row it was
                // proposed from, which has to have landed.
                constexpr bool rec_r = true, rec_p = !XF;
                // (1) what was settled a half-step or more ago, one lane of the row each:
                //  0, 1: for the two rows, that earlier decision (record) or the landing of the row the
                //        move m_s was proposed from
                //  2   : the partner's row before its pending move;  3, 4: the two rows before move m_s
                const unsigned long long *w1 = fv.seq;
                unsigned long long n1 = 0;
                int sh1 = 0;
                bool watch1 = false;
                if (active) {
                    const unsigned long long nq = (unsigned long long)flow_seq(hj ^ 1, m_q);
                    const unsigned long long nold = (unsigned long long)flow_seq(hj, m_s - 1);
                    switch (l16) {
                    case 0:
                        w1 = rec_r ? fv.mseq + (size_t)(ob + qr) * kFlowSlots + (m_q % kFlowSlots) : fv.seq + (ob + qr);
                        n1 = nq; sh1 = rec_r ? 1 : 0; watch1 = m_s > 0 && m_q > 0; break;
                    case 1:
                        w1 = rec_p ? fv.mseq + (size_t)(ob + qp) * kFlowSlots + (m_q % kFlowSlots) : fv.seq + (ob + qp);
                        n1 = nq; sh1 = rec_p ? 1 : 0; watch1 = c1 && m_s > 0 && m_q > 0; break;
                    case 2: w1 = fv.seq + prow; n1 = (unsigned long long)flow_seq(hj ^ 1, m_o); watch1 = m_o > 0; break;
                    case 3: w1 = fv.seq + rown; n1 = nold; watch1 = m_s > 1; break;
                    case 4: w1 = fv.seq + pprow; n1 = nold; watch1 = c1 && m_s > 1; break;
                    default: break;
                    }
                }
                const unsigned long long v1 = spin(w1, n1, watch1, sh1);
                const int cr = (rec_r && m_s > 0 && m_q > 0) ? (int)(__shfl(v1, base + 0) & 1ull) : 0;
                const int cp = (rec_p && m_s > 0 && m_q > 0) ? (int)(__shfl(v1, base + 1) & 1ull) : 0;
                // the rows -> this row's corner of LDS (the polynomial table's place), an element or
                // two per lane: [0,5) the walker's row as it was, [5,10) the row its last move was
                // proposed from, [10,20) the same two for the partner's partner, [20,25) the partner
                double *scr = s_pb + (size_t)(wave * 4 + (lane >> 4)) * 32;
                for (long long tries = 0;; ++tries) {
                    bool bad = false;
                    if (active) {
                        const int so = (m_s > 0 ? m_s - 1 : 0) % kFlowSlots, sq = m_q % kFlowSlots;
                        const double *o_r = fv.st + ((size_t)so * a.nw + rown) * 8, *o_p = fv.st + ((size_t)so * a.nw + pprow) * 8;
                        // the proposal of move m_s: elements 13..17 of the record (each with its check
                        // word), or the row it was proposed from
                        const double *n_r = rec_r ? fv.rec + (((size_t)rown * kFlowSlots + (m_s % kFlowSlots)) * 2 + cr) * kFlowRec + 26
                                                  : fv.st + ((size_t)sq * a.nw + ob + qr) * 8;
                        const double *n_p = rec_p ? fv.rec + (((size_t)pprow * kFlowSlots + (m_s % kFlowSlots)) * 2 + cp) * kFlowRec + 26
                                                  : fv.st + ((size_t)sq * a.nw + ob + qp) * 8;
                        const double *sp = fv.st + ((size_t)(m_o % kFlowSlots) * a.nw + prow) * 8;
                        const unsigned long long tag_g = (a.flow_serial << 32) | (unsigned long long)(g + 1);
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int e = l16 + 16 * t;
                            if (e < 25) {
                                const int grp = e / 5, i = e - 5 * grp;
                                const bool from_rec = (grp == 1 && rec_r) || (grp == 3 && rec_p);
                                const double *src = grp == 0 ? o_r : (grp == 1 ? n_r : (grp == 2 ? o_p : (grp == 3 ? n_p : sp)));
                                const bool want = grp == 0 || grp == 4 || (grp == 1 && m_s > 0) || (grp == 2 && c1) || (grp == 3 && c1 && m_s > 0);
                                double v = 0.0;
                                if (want && from_rec) {
                                    v = ld_dev(src + 2 * i);
                                    const unsigned long long chk = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src) + 2 * i + 1,
                                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    bad = bad || (chk ^ (unsigned long long)__double_as_longlong(v)) != tag_g;
                                } else if (want) {
                                    v = fl_ld(src + i);
                                }
                                scr[e] = v;
                            }
                        }
                    }
                    // (a record element that is not there yet: again -- its mover used it a half-step
                    // ago, so this does not happen; bounded all the same)
                    if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
                    if (tries > MBB_FLOW_SPIN_LIMIT) { atomicMax(a.errflag, 9); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                // a proposal that did not come from a record is formed now, in place of the row it
                // was proposed from: after the decisions only a selection is left
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (active && m_s > 0 && l16 < 10) {
                    const int i = l16 < 5 ? l16 : l16 - 5, o = l16 < 5 ? 0 : 10;
                    if ((l16 < 5 && !rec_r) || (l16 >= 5 && c1 && !rec_p))
                        scr[o + 5 + i] = stretch_q(scr[o + 5 + i], scr[o + i], l16 < 5 ? zr : zq);
                }
                // (2) the decisions of half-step j - 2 (lanes 0, 1): the hand-off this chain waits for
                const unsigned long long *w2 = fv.mseq + (size_t)(l16 == 0 ? rown : pprow) * kFlowSlots + (m_s % kFlowSlots);
                const unsigned long long v2 = spin(w2, (unsigned long long)flow_seq(hj, m_s),
                                                   active && m_s > 0 && (l16 == 0 || (l16 == 1 && c1)), 1);
                STAMP(12);
                const bool ar = m_s > 0 && (__shfl(v2, base + 0) & 1ull), ap = m_s > 0 && (__shfl(v2, base + 1) & 1ull);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 5; ++i) { snv[i] = scr[(ar ? 5 : 0) + i]; cpv[i] = scr[(ap ? 15 : 10) + i]; cpos[i] = scr[20 + i]; }
            }
            if (active) {
                if (c1) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) cpos[i] = stretch_q(cpv[i], cpos[i], zp);
                }
                double p[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) p[i] = stretch_q(cpos[i], snv[i], zz);
                STAMPD(8, p[0] + p[1] + p[2] + p[3] + p[4]);
                double lo[4];
                vlog<true>(lo, p[0], p[2], zz, u3);
                WalkerK k;
                k.hokt9 = k.lhokt9 = k.beta = k.bp3 = k.cq = k.alpha = k.lx0 = k.xmerge = k.cbb = k.cpl = k.kap = k.peak = 0.0;
                k.status = ROW_SKIP;
                k.pad = 0;
                double pen_u = 0.0, pen_g = 0.0;
                const double lT = lo[0], lL = lo[1];
#include "mbb_walker_consts.inc"
                if ((tid & 15) == 0) {
                    double *rec = fv.rec + (((size_t)rown * kFlowSlots + (m_next % kFlowSlots)) * 2 + cand) * kFlowRec;
                    {
                        // element by element, each with its check word; nothing to wait for
                        const unsigned long long tag = (a.flow_serial << 32) | (unsigned long long)(j + 1);
                        auto put = [&](int c, double v) {
                            st_dev(rec + 2 * c, v);
                            __hip_atomic_store(reinterpret_cast<unsigned long long *>(rec) + 2 * c + 1,
                                               tag ^ (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        };
                        put(0, k.hokt9); put(1, k.lhokt9); put(2, k.beta); put(3, k.bp3); put(4, k.cq); put(5, k.alpha);
                        put(6, k.lx0); put(7, k.xmerge); put(8, k.cbb); put(9, k.cpl); put(10, k.kap); put(11, k.peak);
                        put(12, __longlong_as_double((long long)(((unsigned long long)(unsigned int)k.pad << 32) | (unsigned int)k.status)));
#pragma unroll
                        for (int i = 0; i < 5; ++i) put(13 + i, p[i]);
                        put(18, 4.0 * lo[2]); put(19, lo[3]); put(20, pen_u); put(21, pen_g);
                    }
                }
                STAMPD(10, pen_u + pen_g);
            }
            }   // half-steps of a one-launch run
            STAMP(6);
            return;
        }
    }

    if constexpr (FLOW) {
        // a mover starts from loads, not from arithmetic: everything its first instructions need (the
        // workgroups that work ahead have left above: these loads are the movers' alone)
        MBB_LNLIKE_PINS();
        PIN(a.spec); PIN(a.n_ahead); PIN(a.spec_cfg); PIN(a.s_begin); PIN(a.c_begin); PIN(a.c_count);
        PIN(a.step); PIN(a.half); PIN(a.seed); PIN(a.stretch_a); PIN(a.nw); PIN(a.poly_b); PIN(a.poly_c);
        PIN(a.nu); PIN(a.lnnu); PIN(a.wt); PIN(a.nchunk); PIN(a.band_rng); PIN(a.nb); PIN(a.wpb);
        PIN(a.nunit); PIN(a.npart); PIN(a.pos6); PIN(a.nacc); PIN(a.chain6); PIN(a.errflag); PIN(a.tail_slot);
    }

    // A walker's prologue runs on one row of 16 lanes (mbb_device.hip.h, "rows"), so
    // the first ceil(16 W / 64) waves are prologue waves.  The other waves meanwhile
    // stage what the later phases need in LDS and pull their first segment's samples
    // and the index table into this CU's L1, so that nothing after the barrier waits
    // on L2; without spare waves every wave stages first.
    const int pwaves = min(nwave, (16 * W + 63) >> 6);
    if ((!FLOW || (int)blockIdx.x >= a.n_ahead) && (wave >= pwaves || pwaves == nwave)) {
        const int t0 = (pwaves == nwave) ? tid : tid - 64 * pwaves;
        const int nt = (pwaves == nwave) ? (int)blockDim.x : (int)blockDim.x - 64 * pwaves;
        const double2 *gb = reinterpret_cast<const double2 *>(a.poly_b);
        const double2 *gc = reinterpret_cast<const double2 *>(a.poly_c);
        const double2 *g0 = reinterpret_cast<const double2 *>(a.nu);
        const double2 *g1 = reinterpret_cast<const double2 *>(a.lnnu);
        const double2 *g2 = reinterpret_cast<const double2 *>(a.wt);
        double2 *lb = reinterpret_cast<double2 *>(s_pb);
        double2 *lc = reinterpret_cast<double2 *>(s_pc);
        double2 *l0 = reinterpret_cast<double2 *>(s_nu);
        double2 *l1 = reinterpret_cast<double2 *>(s_lnnu);
        double2 *l2 = reinterpret_cast<double2 *>(s_wt);
        constexpr int nB = kPolyBDoubles / 2, nC = OPTHIN ? 0 : kPolyCDoubles / 2;
        const int n2 = STAGE ? a.nchunk * 32 : 0;              // double2 elements per passband array
        if constexpr (!FLOW) {
            // table by table, a sweep at a time: the copy trickles along beside the constructor
            // (asked for all at once it fills the CU's load queue and the constructor wave's own
            // few loads wait behind it: +15 % on the 125-walker launch)
            for (int i = t0; i < kExp2N; i += nt) s_tab[i] = kExp2Tab[i];
            for (int i = t0; i < nB; i += nt) lb[i] = gb[i];
            for (int i = t0; i < nC; i += nt) lc[i] = gc[i];
            for (int b = t0; b < nb; b += nt) { s_flux[b] = a.flux[b]; s_ivar[b] = a.ivar[b]; }
            for (int b = t0; b < nb; b += nt) s_band[b] = a.band_rng[b];
            for (int i = t0; i < n2; i += nt) { l0[i] = g0[i]; l1[i] = g1[i]; l2[i] = g2[i]; }
        } else {
            // A mover of a look-ahead run has no constructor to hide the copy behind (other
            // workgroups run it), so every table is asked for before the first one is stored: one
            // exposed round trip for the lot (~3500 cycles at the start of a launch, when nothing
            // is in L2 yet) instead of one per table and sweep.  Up to three sweeps per table go
            // through registers; what is left (few staging threads, long tables) follows in
            // plain loops.  The loads are pinned in front of the stores: left alone the compiler
            // sinks each load into the branch of its store and waits for it there.
#define MBB_PIN2(v) asm volatile("" : "+v"(v.x), "+v"(v.y))
            // (After that first round trip the copy is bound by the rate at which the CU's address
            // unit takes loads -- 64 bytes a cycle, 16 cycles per full wave -- so lanes past the
            // end of a table ask for nothing.)
            const int i0 = t0, i1 = t0 + nt, i2 = t0 + 2 * nt;
            const double2 z2 = make_double2(0.0, 0.0);
            double2 ve = z2, vb0 = z2, vb1 = z2, vb2 = z2, vc0 = z2, vc1 = z2;
            double2 p00 = z2, p01 = z2, p10 = z2, p11 = z2, p20 = z2, p21 = z2, fiv = z2;
            int2 brv = make_int2(0, 0);
            if (STAGE) {
                if (i0 < n2) { p00 = g0[i0]; p10 = g1[i0]; p20 = g2[i0]; }
                if (i1 < n2) { p01 = g0[i1]; p11 = g1[i1]; p21 = g2[i1]; }
            }
            if (i0 < kExp2N / 2) ve = reinterpret_cast<const double2 *>(kExp2Tab)[i0];
            if (i0 < nB) vb0 = gb[i0];
            if (i1 < nB) vb1 = gb[i1];
            if (i2 < nB) vb2 = gb[i2];
            if (!OPTHIN) {
                if (i0 < nC) vc0 = gc[i0];
                if (i1 < nC) vc1 = gc[i1];
            }
            if (t0 < nb) { fiv = make_double2(a.flux[t0], a.ivar[t0]); brv = a.band_rng[t0]; }
            if (STAGE) { MBB_PIN2(p00); MBB_PIN2(p10); MBB_PIN2(p20); MBB_PIN2(p01); MBB_PIN2(p11); MBB_PIN2(p21); }
            MBB_PIN2(ve); MBB_PIN2(vb0); MBB_PIN2(vb1); MBB_PIN2(vb2);
            if (!OPTHIN) { MBB_PIN2(vc0); MBB_PIN2(vc1); }
            MBB_PIN2(fiv);
#undef MBB_PIN2
            if (STAGE) {
                if (i0 < n2) { l0[i0] = p00; l1[i0] = p10; l2[i0] = p20; }
                if (i1 < n2) { l0[i1] = p01; l1[i1] = p11; l2[i1] = p21; }
            }
            if (i0 < kExp2N / 2) reinterpret_cast<double2 *>(s_tab)[i0] = ve;
            if (i0 < nB) lb[i0] = vb0;
            if (i1 < nB) lb[i1] = vb1;
            if (i2 < nB) lb[i2] = vb2;
            if (!OPTHIN) {
                if (i0 < nC) lc[i0] = vc0;
                if (i1 < nC) lc[i1] = vc1;
            }
            if (t0 < nb) { s_flux[t0] = fiv.x; s_ivar[t0] = fiv.y; s_band[t0] = brv; }
            for (int i = t0 + nt; i < kExp2N / 2; i += nt) reinterpret_cast<double2 *>(s_tab)[i] = reinterpret_cast<const double2 *>(kExp2Tab)[i];
            for (int i = t0 + 3 * nt; i < nB; i += nt) lb[i] = gb[i];
            for (int i = t0 + 2 * nt; i < nC; i += nt) lc[i] = gc[i];
            for (int i = t0 + 2 * nt; i < n2; i += nt) { l0[i] = g0[i]; l1[i] = g1[i]; l2[i] = g2[i]; }
            for (int b = t0 + nt; b < nb; b += nt) { s_flux[b] = a.flux[b]; s_ivar[b] = a.ivar[b]; s_band[b] = a.band_rng[b]; }
        }
        if (a.cov_in_lds)
            for (int i = t0; i < nb * nb; i += nt) s_invcov[i] = a.invcov[i];
        if (!STAGE && wave >= pwaves) {
            const int u = wave;
            if (u < W * nun) {
                const int4 us = a.unit_tab[u % nun];
                double t = 0.0;
                for (int c = us.y; c < us.z; ++c) {
                    const int i = c * 64 + lane;
                    t += a.nu[i] + a.lnnu[i] + a.wt[i];
                }
                asm volatile("" ::"v"(t));
            }
        }
    }

    // this wave's first quadrature unit: asked for now, it is here when phase 2 begins
    const int nunit = W * nun;
    int4 us_first = make_int4(0, 0, 0, 0);
    if (wave < nunit) us_first = a.unit_tab[wave % nun];
    int tail_first = -1;                  // ... and (a one-launch run) should it be a tail chunk, the slot of this lane's row
    if (FLOW && wave < nunit && us_first.w == 2) tail_first = a.tail_slot[4 * us_first.x + (lane >> 4)];

    // FLOW: a.persist half-steps in this launch; otherwise one pass with the launch's values
    const int niter = FLOW ? a.persist : 1;
    // SMODE 5, wave 0: the workgroup's two walkers (one of each half) as they are, element l in
    // lane l < 8 -- nobody else writes them; and the row whose word is still to be published
    double own_half[2] = {0.0, 0.0};
    int pend_row = -1, pend_it = 0;
    // (a sharded run) this wave's place in the count of the half-step before that one, looked at a
    // half-step after it was asked for: whoever came last tells every rank that this rank is through
    unsigned long long cnt_seen = 0;
    int cnt_it = -1;
    for (int it = 0; it < niter; ++it) {
    const int L_step = FLOW ? a.step + (it >> 1) : a.step, L_half = FLOW ? (it & 1) : a.half;
    const int L_s_begin = FLOW ? (L_half ? a.c_count : 0) + a.s_begin : a.s_begin;
    const int L_c_begin = FLOW ? (L_half ? 0 : a.c_count) : a.c_begin;
    const unsigned long long L_seed = FLOW ? a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(it >> 1) : a.seed;
    double *const L_chain6 = (FLOW && a.chain6) ? a.chain6 + (size_t)it * a.n * 6 : a.chain6;
    unsigned int *const L_nacc = FLOW ? a.nacc + (size_t)L_half * a.n : a.nacc;

    // A mover of a one-launch run: its proposal record was written by a workgroup working ahead; wave 0
    // picks the candidate its partner's decision says and lays it out in LDS as phase 1 would have
    double own_reg = 0.0;          // lane l < 8 of wave 0: element l of the walker's state row
    if constexpr (FLOW) {
        if (wave == 0) {
            const int row = L_s_begin + w0;                   // one ensemble, one walker per workgroup
            const FlowView fv = peer_view(xrank);
            const double *rec = fv.rec + ((size_t)row * kFlowSlots + ((flow_cnt(L_half, it) + 1) % kFlowSlots)) * 2 * kFlowRec;
            double r0 = 0.0, r1 = 0.0, flag = 0.0;
            {
                STAMP(11);
                // SMODE 5: this half-step starts when both candidates of the walker's record are
                // there and its partner's move of the half-step before is decided (and nobody is
                // more than four half-steps behind: the state slots and the records are reused)
                double zz, u3;
                int pj;
                stretch_draw(row, L_step, L_half, L_seed, a.stretch_a, a.c_count, zz, pj, u3);
                const int prow = L_c_begin + pj;
                const int m_par = flow_cnt(L_half ^ 1, it);
                // One loop, everything asked for in the same round: lane c < 22 the record's element c
                // of both candidates (taken when its check word fits), lane 22 the partner's decision,
                // lane 23 the lag guard.  When all of it is there already, that is one round trip.
                const unsigned long long tag = (a.flow_serial << 32) | (unsigned long long)(it + 1);
                const unsigned long long need_p = (unsigned long long)flow_seq(L_half ^ 1, m_par);
                // lane 23: this GPU's count of half-step it - kFlowLag (the lag guard); a sharded run
                // besides: lanes 24 ..: the other ranks' words for that half-step (whichever of a
                // rank's movers completes a half-step last tells them), lanes 40 .. (first half-step):
                // every peer has set up its copy of this run
                const int gl = lane - 24, sl = lane - 40;
                const bool guard = XF && gl >= 0 && gl < npeer && gl != xrank && it >= kFlowLag;
                const bool started = XF && it == 0 && sl >= 0 && sl < npeer;
                const unsigned long long need_g =
                    started ? fx->run : (guard ? (unsigned long long)(it - kFlowLag + 1)
                                               : (unsigned long long)a.n * (unsigned long long)(((it - kFlowLag) >> 3) + 1));
                const unsigned long long *word =
                    lane == 22 ? fv.mseq + (size_t)prow * kFlowSlots + (m_par % kFlowSlots)
                               : (started ? fv.startf + sl
                                          : (guard ? fv.pub + gl * 8 + ((it - kFlowLag) & 7) : fv.done + ((it - kFlowLag) & 7) * 16));
                const bool watch = (lane == 22 && need_p > 0) || (lane == 23 && it >= kFlowLag) || guard || started;
                const int c = lane < kFlowRecN ? lane : 0;
                unsigned long long pv = 0;
                bool rec_ok = false, word_ok = !watch;
                long long spins = 0;
                for (;;) {
                    if (lane < kFlowRecN && !rec_ok) {
                        r0 = ld_dev(rec + 2 * c);
                        r1 = ld_dev(rec + kFlowRec + 2 * c);
                        const unsigned long long t0 = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(rec) + 2 * c + 1,
                                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long t1 = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(rec) + kFlowRec + 2 * c + 1,
                                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        rec_ok = (t0 ^ (unsigned long long)__double_as_longlong(r0)) == tag &&
                                 (t1 ^ (unsigned long long)__double_as_longlong(r1)) == tag;
                    }
                    if (watch && !word_ok) {
                        pv = fl_ldw(word);
                        word_ok = (lane == 22 ? pv >> 1 : pv) >= (lane == 22 ? need_p : need_g);
                    }
                    if (pend_row >= 0) {
                        // The row this wave stored at the end of the half-step before has landed by
                        // now -- memory operations of a wave complete in the order they were issued,
                        // and the loads above came after those stores; the explicit wait costs nothing
                        // then.  Its word, for the workgroups that work ahead from that row.
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (XF) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");     // across GPUs: a full release
                        if (XF && lane < npeer) fl_stw(peer_view(lane).seq + pend_row, (unsigned long long)(pend_it + 1));
                        if (lane == 0) {
                            if (!XF) fl_stw(fv.seq + pend_row, (unsigned long long)(pend_it + 1));
                            if constexpr (XF) {
                                if (cnt_it >= 0 && cnt_seen + 1 == (unsigned long long)a.n * (unsigned long long)((cnt_it >> 3) + 1))
                                    for (int pr = 0; pr < npeer; ++pr)
                                        if (pr != xrank)
                                            fl_stw(peer_view(pr).pub + xrank * 8 + (cnt_it & 7), (unsigned long long)(cnt_it + 1));
                                cnt_seen = __hip_atomic_fetch_add(fv.done + (pend_it & 7) * 16, 1ull, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT);
                                cnt_it = pend_it;
                            } else {
                                __hip_atomic_fetch_add(fv.done + (pend_it & 7) * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                        pend_row = -1;
                    }
                    if (__builtin_amdgcn_ballot_w64((lane < kFlowRecN && !rec_ok) || !word_ok) == 0) break;
                    ++spins;
                    if (spins > MBB_FLOW_SPIN_LIMIT ||
                        ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        atomicMax(a.errflag, 9);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (it < 2) {
                    if (lane < 8) own_reg = fl_ld(fv.st + (size_t)row * 8 + lane);
                } else {
                    own_reg = L_half ? own_half[1] : own_half[0];
                }
                flag = (need_p > 0 && (__shfl(pv, 22) & 1ull)) ? 1.0 : 0.0;
                STAMP(12);
            }
            const double v = flag != 0.0 ? r1 : r0;
            double *wkd = reinterpret_cast<double *>(wk);
            if (lane < 13) wkd[lane] = v;
            else if (lane < 18) prop[lane - 13] = v;          // the proposal
            else if (lane == 18) prop[5] = v;                 // (dim - 1) ln z
            else if (lane == 19) prop[7] = v;                 // ln u
            else if (lane == 20) pen[0] = v;
            else if (lane == 21) pen[1] = v;
            if (lane == 5) prop[6] = own_reg;                 // the walker's current lnprob
            STAMPD(10, v);
        }
    }
    // ---- phase 1: gate + prologue + parameter-only penalties, one row per walker
    // (the host guarantees blockDim.x >= 16 W)
    if constexpr (!FLOW)
    if (const int j = tid >> 4; j < W) {
        const bool lead = (tid & 15) == 0;                    // the lane that writes to LDS
        const int w = w0 + j;
        WalkerK k;
        k.status = ROW_SKIP;
        k.pad = 0;
        double pen_u = 0.0, pen_g = 0.0;
        bool pre_done = false;
        if (w < a.n) {
            double p[5], lT, lL = 0.0;
            if (SAMPLER) {
                // stretch move (Goodman & Weare 2010; what emcee does per half-step,
                // mbb_fit.py:533/:542): z ~ g(z) on [1/a, a], partner from the other
                // half, proposal q = c - z (c - s)
                const int src = w / a.m_count, loc = w - src * a.m_count;
                const int row = src * a.nw_src + L_s_begin + loc;
                double zz, u3;
                int pj;
                stretch_draw(row, L_step, L_half, L_seed, a.stretch_a, a.c_count, zz, pj, u3);
                const double *srow = a.pos6 + (size_t)row * 6;
                const double *crow = a.pos6 + (size_t)(src * a.nw_src + L_c_begin + pj) * 6;
                constexpr bool xchg = XCHG;
                if (xchg) {
                    const XchgArgs &x = *a.xargs;
                    const unsigned long long xseq = x.xseq0 + 2ull * (unsigned)L_step + (unsigned)L_half + 1ull;
                    if (xseq > 1) {
                        // the partner rows were moved by the previous launch, on any rank: wait
                        // until every peer has posted that launch (lane l of the wave watches
                        // peer l's word), bounded so that a lost peer cannot hang the GPU
                        // (a launch that already failed -- errflag set -- is not waited for again)
                        const unsigned long long *mine = x.xflag[x.xrank];
                        const int l = tid & 63;
                        long long spins = __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 8
                                              ? x.xspin_max : 0;
                        for (;;) {
                            unsigned long long v = xseq;
                            if (l < x.xn && l != x.xrank)
                                v = __hip_atomic_load(mine + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            if (__builtin_amdgcn_ballot_w64(v + 1 < xseq) == 0) break;
                            if (++spins > x.xspin_max) { atomicMax(a.errflag, 8); break; }
                            __builtin_amdgcn_s_sleep(8);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                    }
                }
                double srow5 = 0.0;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const double cv = xchg ? ld_sys(crow + i) : crow[i];
                    const double sv = xchg ? ld_sys(srow + i) : srow[i];
                    p[i] = stretch_q(cv, sv, zz);
                }
                srow5 = xchg ? ld_sys(srow + 5) : srow[5];
                double lo[4];
                vlog<true>(lo, p[0], p[2], zz, u3);
                lT = lo[0]; lL = lo[1];
                if (lead) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) prop[j * 8 + i] = p[i];
                    prop[j * 8 + 5] = 4.0 * lo[2];            // (dim - 1) ln z, dim = 5
                    prop[j * 8 + 6] = srow5;
                    prop[j * 8 + 7] = lo[3];                  // ln u
                }
            } else if (a.spec) {
                // big batches: gate, constructor and penalties were worked out by k_walker_pre, a LANE per walker (below), and
                // are fetched: lane l of the row takes word l of the walker's record {WalkerK: 13 words, pen_u, pen_g, -}
                const double v = a.spec[(size_t)w * kPreWords + (tid & 15)];
                const int l = tid & 15;
                double *wkd = reinterpret_cast<double *>(wk + j);
                if (l < 13) wkd[l] = v;
                else if (l < 15) pen[2 * j + (l - 13)] = v;
                pre_done = true;
            } else {
#pragma unroll
                for (int i = 0; i < 5; ++i) p[i] = a.pars[(size_t)w * 5 + i];
                STAMPD(8, p[0] + p[1] + p[2] + p[3] + p[4]);
                if (OPTHIN) {
                    double lo[1];
                    vlog<true>(lo, p[0]);
                    lT = lo[0];
                } else {
                    double lo[2];
                    vlog<true>(lo, p[0], p[2]);
                    lT = lo[0]; lL = lo[1];
                }
            }
            if (!pre_done) {
#include "mbb_walker_consts.inc"
            }
        }
        STAMPD(10, pen_u + pen_g);
        if (lead && !pre_done) {
            if (k.status == ROW_OK) wk[j] = k;
            else { wk[j].status = k.status; wk[j].pad = k.pad; }
            pen[2 * j] = pen_u;
            pen[2 * j + 1] = pen_g;
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);
    WSTAMP(1);

    // ---- phase 2: passband quadrature (response.py:572-576) -----------------
    auto T_nu = [&](int i) { if constexpr (STAGE) return s_nu[i]; else return a.nu[i]; };
    auto T_ln = [&](int i) { if constexpr (STAGE) return s_lnnu[i]; else return a.lnnu[i]; };
    auto T_wt = [&](int i) { if constexpr (STAGE) return s_wt[i]; else return a.wt[i]; };
    const SampleTabs tabs = {s_tab, s_pb, s_pc};
    auto do_unit = [&](const WalkerK &k, const int j, const int4 us, const bool first_unit) {
        const int s = us.x, c0 = us.y, c1 = us.z;
        double acc = 0.0;
        int c = c0;
        for (; c + 2 <= c1; c += 2) {          // two chunks per step
            double n0, l0, q0, n1, l1, q1;
            if constexpr (STAGE) {
                const int i0 = c * 64 + lane, i1 = i0 + 64;
                n0 = s_nu[i0]; l0 = s_lnnu[i0]; q0 = s_wt[i0];
                n1 = s_nu[i1]; l1 = s_lnnu[i1]; q1 = s_wt[i1];
            } else {
                // (the chunk is the wave's, uniform: a scalar base per table and the lane as a 32-bit offset -- indexed by
                // chunk * 64 + lane the compiler formed three 64-bit addresses per chunk pair on the vector unit)
                const int cu = __builtin_amdgcn_readfirstlane(c);
                typedef const __attribute__((address_space(1))) double *gptr;       // (global memory, said so: the pin below hides where the pointers came from)
                gptr pn = (gptr)(a.nu + (size_t)cu * 64), pl = (gptr)(a.lnnu + (size_t)cu * 64), pw = (gptr)(a.wt + (size_t)cu * 64);
                // (pinned in scalar registers: left alone the compiler re-associates base + lane into a vector of its own
                // per table and adds the chunk on the vector unit)
                asm volatile("" : "+s"(pn), "+s"(pl), "+s"(pw));
                n0 = pn[lane]; l0 = pl[lane]; q0 = pw[lane];
                n1 = pn[lane + 64]; l1 = pl[lane + 64]; q1 = pw[lane + 64];
            }
            const double f0 = fnu_sample<OPTHIN, NOALPHA, true, false>(k, n0, l0, &tabs);
            const double f1 = fnu_sample<OPTHIN, NOALPHA, true, false>(k, n1, l1, &tabs);
            acc = fma(f0, q0, acc);
            acc = fma(f1, q1, acc);
        }
        if (c < c1) {
            const int i = c * 64 + lane;
            const double f = fnu_sample<OPTHIN, NOALPHA, true, false>(k, T_nu(i), T_ln(i), &tabs);
            acc = fma(f, T_wt(i), acc);
        }
        WSTAMPD(3, acc);
        if (us.w == 0) {
            acc = wave_sum_l63(acc);           // (the total is lane 63's)
            if (lane == 63) partial[j * npart + s] = acc;
        } else if (us.w == 2) {                               // four band leftovers, one per row
            acc = row_sum(acc);
            if ((lane & 15) == 0) {
                const int sl = (FLOW && first_unit) ? tail_first : a.tail_slot[4 * s + (lane >> 4)];
                if (sl >= 0) partial[j * npart + sl] = acc;
            }
        } else {
            partial[j * npart + s + lane] = acc;              // 64 single-sample bands
        }
    };
    if (W >= nwave) {
        // many walkers per workgroup (big batches): a wave takes whole walkers, so the
        // walker's constants are fetched once and there is no (walker, unit) arithmetic;
        // every walker has the same units, so the waves stay balanced
        for (int j = wave; j < W; j += nwave) {
            if (wk[j].status != ROW_OK) continue;                 // wave-uniform
            const WalkerK k = wk[j];
            for (int uu = 0; uu < nun; ++uu) do_unit(k, j, a.unit_tab[uu], false);
        }
    } else {
        // few walkers (an emcee half-step: one per workgroup): the units of a walker are
        // dealt to the waves; the table deals the segments so that the four SIMDs (wave mod
        // 4) of the CU get equal numbers of chunks; which wave sums a segment does not change it
        for (int u = wave; u < nunit; u += nwave) {
            const int j = u / nun;
            const int4 us = (u == wave) ? us_first : a.unit_tab[u - j * nun];
            if (wk[j].status != ROW_OK) continue;                 // wave-uniform
            const WalkerK k = wk[j];
            WSTAMPD(2, k.cbb);
            do_unit(k, j, us, u == wave);
            WSTAMP(4);
        }
    }
    STAMP(3);
#ifdef MBB_STAMPS       // every wave's arrival at the second barrier (slots 17..31: waves 1..15)
    if ((tid & 63) == 0 && tid > 0 && a.stamps && blockIdx.x < 65536) a.stamps[blockIdx.x * 32 + 16 + (tid >> 6)] = __builtin_amdgcn_s_memtime();
#endif

    // ---- phase 3: one wave per walker, one lane per band (likelihood.py:821-834)
    // Whatever does not depend on the segment sums is fetched before the barrier, for
    // the wave's first walker and the lane's first band: a wave that is done with its
    // quadrature early would only wait there.  The code after the barrier is peeled
    // the same way (FIRST), so that the usual case -- one walker per wave, at most 64
    // bands -- runs without the bookkeeping of the general loops.
    const bool multi = a.nsrc > 1;
    // data of walker w's source: the LDS copy for one source, global for many
    auto band_data = [&](int w, int b, double &fb, double &ib) {
        if (multi) {
            const int src = SAMPLER ? (w / a.m_count) : (w / a.rows_per_src);
            fb = a.flux[(size_t)src * nb + b];
            ib = a.ivar[(size_t)src * nb + b];
        } else {
            fb = s_flux[b];
            ib = s_ivar[b];
        }
    };
    int st_first = ROW_SKIP, sb_first0 = 0, sb_first1 = 0, pad_first = 0;
    double fb_first = 0.0, ib_first = 0.0, pen_u_first = 0.0, pen_g_first = 0.0, cbb_first = 0.0;
    double *lnl_first = nullptr;
    int32_t *status_first = nullptr;
    double q_first[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // SAMPLER: the proposal record
    int row_first = 0;                                               // SAMPLER: the walker's state row
    if (SAMPLER && wave < W) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q_first[i] = prop[wave * 8 + i];
        const int wf = w0 + wave, srcf = wf / a.m_count;
        row_first = srcf * a.nw_src + L_s_begin + (wf - srcf * a.m_count);
    }
    if (wave < W) {
        st_first = wk[wave].status;
        pad_first = wk[wave].pad;
        cbb_first = wk[wave].cq;
        pen_u_first = pen[2 * wave];
        pen_g_first = pen[2 * wave + 1];
        if (a.lnl) lnl_first = a.lnl + (w0 + wave);
        if (a.status) status_first = a.status + (w0 + wave);
        if (lane < nb) {
            const int2 rng = s_band[lane];
            sb_first0 = rng.x;
            sb_first1 = rng.y;
            band_data(w0 + wave, lane, fb_first, ib_first);
        }
    }
    __syncthreads();
    STAMP(4);

    auto epilogue = [&](const int j, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const int st = FIRST ? st_first : wk[j].status;
        if (st == ROW_SKIP) return;                            // wave-uniform
        const int w = w0 + j;
        const double pen_u = FIRST ? pen_u_first : pen[2 * j];
        const double pen_g = FIRST ? pen_g_first : pen[2 * j + 1];
        double acc = 0.0;
        if (st == ROW_OK) {
            double *mf = mflux + (size_t)j * nb;
            const double *pj = partial + j * npart;
            const double cbb = FIRST ? cbb_first : wk[j].cq;   // normfac (h/kT)^2: the samples were summed without them
            auto band = [&](const int b, auto firstb_c) {      // band flux, fixed order
                constexpr bool FB = decltype(firstb_c)::value;
                double sum = 0.0;
                int sg0 = sb_first0, sg1 = sb_first1;
                if (!FB) { const int2 rng = s_band[b]; sg0 = rng.x; sg1 = rng.y; }
                for (int sg = sg0; sg < sg1; sg += 4) {        // four reads per wait, same order
                    const int l = sg1 - 1;
                    const double q0 = pj[sg], q1 = pj[min(sg + 1, l)], q2 = pj[min(sg + 2, l)],
                                 q3 = pj[min(sg + 3, l)];
                    sum += q0;
                    if (sg + 1 < sg1) sum += q1;
                    if (sg + 2 < sg1) sum += q2;
                    if (sg + 3 < sg1) sum += q3;
                }
                sum *= cbb;
                if (a.model_flux) a.model_flux[(size_t)w * nb + b] = sum;
                double fb = fb_first, ib = ib_first;
                if (!FB) band_data(w, b, fb, ib);
                const double d = fb - sum;                     // :821
                if (a.invcov) mf[b] = d;
                else acc = fma(d * d, ib, acc);                // :825
            };
            if (lane < nb) band(lane, first_c);
            if (nb > 64)
                for (int b = lane + 64; b < nb; b += 64) band(b, std::false_type{});
            if (a.invcov) {                                    // :823
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < nb; i += 64) {
                    double t = 0.0;
                    const double *row = (a.cov_in_lds ? s_invcov : a.invcov) + (size_t)i * nb;
                    for (int jj = 0; jj < nb; ++jj) t = fma(row[jj], mf[jj], t);
                    acc = fma(mf[i], t, acc);
                }
            }
            STAMPD(5, acc);
            acc = (nb <= 16) ? wave_sum_row0(acc) : wave_sum(acc);
            STAMPD(16, acc);
        } else if (a.model_flux) {
            for (int b = lane; b < nb; b += 64) a.model_flux[(size_t)w * nb + b] = __builtin_nan("");
        }
        double old5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        if constexpr (FLOW) {
#pragma unroll
            for (int i = 0; i < 5; ++i) old5[i] = __shfl(own_reg, i);
        }
        int flow_accept = 0;
        double flow_r = 0.0;
        if (lane == 0) {
            double r;
            if (st == ROW_BELOW_LOWLIM) r = -__builtin_inf();
            else if (st != ROW_OK) r = __builtin_nan("");
            else {
                r = fma(-0.5, acc, pen_u);                     // :828
                if (a.has_gprior) r += pen_g;                  // :830-831
            }
            double *lnl_out = FIRST ? lnl_first : (a.lnl ? a.lnl + w : nullptr);
            int32_t *status_out = FIRST ? status_first : (a.status ? a.status + w : nullptr);
            if (SAMPLER) {
                // accept with probability min(1, z^(dim-1) P(q)/P(s))
                int row = row_first;
                if (!FIRST) { const int src = w / a.m_count; row = src * a.nw_src + L_s_begin + (w - src * a.m_count); }
                double *srow = a.pos6 + (size_t)row * 6;
                double q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = FIRST ? q_first[i] : prop[j * 8 + i];
                if (st >= 2 || r != r) atomicMax(a.errflag, st >= 2 ? st : (int)ROW_NONFINITE);
                const bool accept = (q[5] + r - q[6]) > q[7];
                if (XCHG) {
                    // sharded run: the new row goes into every rank's copy of the ensemble
                    // (a rejected move changes nothing anywhere), then this walker is counted;
                    // the last one of the launch tells every peer that the launch is complete
                    const XchgArgs &x = *a.xargs;
                    const unsigned long long xseq = x.xseq0 + 2ull * (unsigned)L_step + (unsigned)L_half + 1ull;
                    if (accept) {
                        for (int pr = 0; pr < x.xn; ++pr) {
                            double *dst = x.xpos[pr] + (size_t)row * 6;
#pragma unroll
                            for (int i = 0; i < 5; ++i) st_sys(dst + i, q[i]);
                            st_sys(dst + 5, r);
                        }
                        atomicAdd(&L_nacc[w], 1u);
                    }
                    if (L_chain6) {
                        double *crow = L_chain6 + (size_t)w * 6;
#pragma unroll
                        for (int i = 0; i < 5; ++i) crow[i] = accept ? q[i] : ld_sys(srow + i);
                        crow[5] = accept ? r : q[6];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the rows are out
                    const unsigned int done = __hip_atomic_fetch_add(x.xcount, 1u, __ATOMIC_ACQ_REL,
                                                                     __HIP_MEMORY_SCOPE_AGENT);
                    if (done + 1 == (unsigned int)a.n) {
                        __hip_atomic_store(x.xcount, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (int pr = 0; pr < x.xn; ++pr)
                            if (pr != x.xrank)
                                __hip_atomic_store(x.xflag[pr] + x.xrank, xseq, __ATOMIC_RELEASE,
                                                   __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                } else if (FLOW) {
                    // the decision first -- all the partner's next mover waits for --, then the row
                    // as it is after this half-step, write-through, to its next slot (every rank's
                    // copy in a sharded run); the row's own word follows at the start of the next
                    // half-step, when the stores have landed.  pos6 is brought up to date once, when
                    // the launch has ended (k_flow_finish); counts and chain are for the host: plain stores.
                    const FlowView fv = peer_view(xrank);
                    const int m_new = flow_cnt(L_half, it) + 1;
                    if constexpr (!XF)
                        fl_stw(fv.mseq + (size_t)row * kFlowSlots + (m_new % kFlowSlots),
                               2ull * (unsigned long long)(it + 1) + (accept ? 1ull : 0ull));
                    if constexpr (!XF) {
                        double *dst = fv.st + ((size_t)(m_new % kFlowSlots) * a.nw + row) * 8;
#pragma unroll
                        for (int i = 0; i < 5; ++i) fl_st(dst + i, accept ? q[i] : old5[i]);
                        fl_st(dst + 5, accept ? r : q[6]);
                    }                                          // (a sharded run: below, a lane per rank)
                    if (accept) atomicAdd(&L_nacc[w], 1u);
                    if (L_chain6) {
                        double *crow = L_chain6 + (size_t)w * 6;
#pragma unroll
                        for (int i = 0; i < 5; ++i) crow[i] = accept ? q[i] : old5[i];
                        crow[5] = accept ? r : q[6];
                    }
                    flow_accept = accept ? 1 : 0;
                    flow_r = r;
                } else {
                if (accept) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) srow[i] = q[i];
                    srow[5] = r;
                    atomicAdd(&L_nacc[w], 1u);                 // no-return atomic: nothing waits on it
                }
                if (L_chain6) {
                    double *crow = L_chain6 + (size_t)w * 6;
#pragma unroll
                    for (int i = 0; i < 5; ++i) crow[i] = accept ? q[i] : srow[i];
                    crow[5] = accept ? r : q[6];
                }
                }
            }
            if (lnl_out) *lnl_out = r;
            if (status_out) *status_out = a.debug ? (st | ((FIRST ? pad_first : wk[j].pad) << 8)) : st;
        }
        if constexpr (FLOW) {
            // the walker as it is now stays in this wave's registers for its next half-step
            const int acc_u = __builtin_amdgcn_readfirstlane(flow_accept);
            if constexpr (XF) {
                // the decision, into every rank's copy at once (lane pr stores to rank pr)
                if (lane < npeer)
                    fl_stw(peer_view(lane).mseq + (size_t)row_first * kFlowSlots + ((flow_cnt(L_half, it) + 1) % kFlowSlots),
                           2ull * (unsigned long long)(it + 1) + (acc_u ? 1ull : 0ull));
            }
            const double r_u = __shfl(flow_r, 0);
            double nv = own_reg;
            if (acc_u) {
                double t = r_u;                                // lane 5: lnprob
#pragma unroll
                for (int i = 0; i < 5; ++i) t = (lane == i) ? q_first[i] : t;
                nv = lane < 6 ? t : nv;
            }
            if (L_half) own_half[1] = nv; else own_half[0] = nv;
            if constexpr (XF) {
                // the row as it is now into every rank's copy: lane pr stores to rank pr, an
                // instruction per element instead of one per element and rank
                const int m_new = flow_cnt(L_half, it) + 1;
                double *dst = peer_view(lane < npeer ? lane : 0).st + ((size_t)(m_new % kFlowSlots) * a.nw + row_first) * 8;
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const double ve = __shfl(nv, e);
                    if (lane < npeer) fl_st(dst + e, ve);
                }
            }
            pend_row = row_first;
            pend_it = it;
        }
    };
    if (wave < W) epilogue(wave, std::true_type{});
    if (W > nwave)
        for (int j = wave + nwave; j < W; j += nwave) epilogue(j, std::false_type{});
    }   // half-steps of a one-launch run
    STAMP(6);
#undef MBB_FLOW_SPIN_LIMIT
#undef PIN
}

// SMODE 5: slot 0 of the state from the sampler's rows (accept flags clear), all words zero.
static __global__ void k_flow_init(const double *pos6, double *spec, int nw)
{
    const FlowView fv = flow_view(spec, nw);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw * 8) {
        const int row = i >> 3, e = i & 7;
        fv.st[i] = e < 6 ? pos6[(size_t)row * 6 + e] : 0.0;
    }
    const int nwords = nw * (1 + kFlowSlots) + 8 * 16 + 16 * 8;   // seq, done, mseq, pub are contiguous
    for (int k = i; k < nwords; k += gridDim.x * blockDim.x) fv.seq[k] = 0ull;
}

// SMODE 5, 6, when a launch of `nhalf` half-steps has ended: the sampler's rows from the slots
// the last moves went to (every row is in this GPU's copy, whoever moved it).
static __global__ void k_flow_finish(double *pos6, double *spec, int nw, int nhalf)
{
    const FlowView fv = flow_view(spec, nw);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw * 6) return;
    const int row = i / 6, e = i - 6 * row;
    const int slot = flow_cnt(row < nw / 2 ? 0 : 1, nhalf) % kFlowSlots;
    pos6[i] = __hip_atomic_load(fv.st + ((size_t)slot * nw + row) * 8 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The chain of a run, from the order the launches wrote it in -- [shard][step][half][launch index][6] --
// to the layout the caller gets (emcee's, results.py:153-154): chain [row][step][5] and lnprob
// [row][step], behind one another in `out`.  One thread per (row, step).  Doing this on the device and
// copying straight into the caller's arrays replaced a D2H into a staging vector plus three nested host
// loops: 3.8 -> see profiles/r03 us per step of overhead for a stored run of 250 walkers.
static __global__ void k_chain_reorder(const double *chain6, double *out, int nsteps, int nw, int half, int per,
                                       int nsrc, int shards)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long R = (long long)nsrc * nw;
    if (i >= R * nsteps) return;
    const int row = (int)(i / nsteps), t = (int)(i - (long long)row * nsteps);
    const int src = row / nw, rin = row - src * nw;
    const int h = rin >= half ? 1 : 0, rh = rin - h * half;
    const int r = rh / per, loc = rh - r * per;
    const size_t nl = (size_t)nsrc * per;
    const size_t w = (size_t)src * per + loc;
    const double *q = chain6 + ((((size_t)r * nsteps + t) * 2 + h) * nl + w) * 6;
    double *c = out + (size_t)i * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) c[k] = q[k];
    out[(size_t)R * nsteps * 5 + i] = q[5];
    (void)shards;
}

// ---- sampler form 7 (k_flowm, mbb_flowm.hip.h): the state of a run, its set-up and its end
struct FlowMView {
    double *prop, *row;
    unsigned long long *mseq, *done;
};
constexpr int kFmWords = 16;   // words per proposal / per row
// (kFmSlots, kFmLag, kFmRing: mbb_flow_index.h)
__host__ __device__ constexpr size_t flowm_words(size_t nw)
{
    // (the completion counters begin on a 256-byte boundary: 250 atomic adds per half-step on a word that shared its
    // 128-byte line with the last rows' decision words, which their partners poll, cost every second launch of a
    // sampler 2 % -- profiles/r04/done_counters_alignment.txt)
    return ((nw * ((size_t)kFmSlots * 2 * kFmWords + kFmSlots * kFmWords + kFmMseq) + 31) & ~(size_t)31) + 2 * kFmRing * 16;
}
static_assert(flowm_words(1000) <= spec_words(1000) && flowm_words(2) <= spec_words(2), "form 7 lives in the allocation of forms 5/6");
__host__ __device__ __forceinline__ FlowMView flowm_view(double *spec, int nw)
{
    FlowMView v;
    v.prop = spec;
    v.row = spec + (size_t)nw * kFmSlots * 2 * kFmWords;
    v.mseq = reinterpret_cast<unsigned long long *>(v.row + (size_t)nw * kFmSlots * kFmWords);
    v.done = reinterpret_cast<unsigned long long *>(spec) + (flowm_words((size_t)nw) - 2 * kFmRing * 16);
    return v;
}
// SMODE 6: tells every rank that this rank's copy is set up for run `run` (which = 0) or that
// its launch of that run has ended (which = 1; with the top bit set if the launch gave up: what it
// stored into the peers' copies after that is not to be trusted, so they must fail too).  One wave.
static __global__ void k_flow_post(const FlowX *fx, int nw, int which, unsigned long long run, const int *errflag)
{
    const int pr = threadIdx.x;
    if (pr >= fx->n) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    const FlowView v = flow_view(fx->base[pr], nw);
    const unsigned long long bad = (which && errflag && *errflag != 0) ? (1ull << 63) : 0ull;
    __hip_atomic_store((which ? v.endf : v.startf) + fx->rank, run | bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// SMODE 6: holds the stream until every peer's launch of run `run` has ended (none of its stores
// into this rank's copy is in flight any more); a peer that gave up, or does not answer, fails
// this rank's run too.  One wave.
static __global__ void k_flow_wait_end(const unsigned long long *endf, int xn, int xrank, unsigned long long run,
                                       long long spin_max, int *errflag)
{
    const int l = threadIdx.x;
    long long spins = 0;
    for (;;) {
        unsigned long long v = run;
        if (l < xn && l != xrank) v = __hip_atomic_load(endf + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__builtin_amdgcn_ballot_w64((v & ~(1ull << 63)) < run) == 0) {
            if (__builtin_amdgcn_ballot_w64((v >> 63) != 0) != 0) atomicMax(errflag, 9);
            break;
        }
        if (++spins > spin_max) { atomicMax(errflag, 8); break; }
        __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// Sharded run, one-hop exchange: holds the stream until every peer has posted launch
// number `seq` (its moved rows are then all in this rank's copy of the ensemble).  One wave.
static __global__ void k_xchg_wait(const unsigned long long *mine, int xn, int xrank, unsigned long long seq,
                            long long spin_max, int *errflag)
{
    const int l = threadIdx.x;
    long long spins = 0;
    for (;;) {
        unsigned long long v = seq;
        if (l < xn && l != xrank) v = __hip_atomic_load(mine + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__builtin_amdgcn_ballot_w64(v < seq) == 0) break;
        if (++spins > spin_max) { atomicMax(errflag, 8); break; }
        __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// Measurement only (bench.py, roofline): the sample arithmetic of phase 2 and nothing
// else -- every wave of a chip-filling grid walks all passband chunks `reps` times with
// one walker's constants; no prologue, no reductions, no epilogue.  Its samples/s is the
// empirical roof the fused kernel's quadrature is priced against.
template <bool OPTHIN, bool NOALPHA>
__global__ void __launch_bounds__(1024) k_roof(const LikeArgs a, const WalkerK *wk, int reps, double *out,
                                              unsigned long long *clk)
{
    // shader cycles (s_memtime) against the 100 MHz reference (s_memrealtime) over the whole
    // loop of workgroup 0: the clock the chip holds under this load (MI355X_MICROARCH.md,
    // "DVFS give-back" item 6); stored where nothing else reads it
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __shared__ __align__(16) double s_tab[kExp2N];
    __shared__ __align__(16) double s_pb[kPolyBDoubles];
    __shared__ __align__(16) double s_pc[OPTHIN ? 2 : kPolyCDoubles];
    const int tid = threadIdx.x, lane = tid & 63, nt = blockDim.x;
    for (int i = tid; i < kExp2N; i += nt) s_tab[i] = kExp2Tab[i];
    for (int i = tid; i < kPolyBDoubles; i += nt) s_pb[i] = a.poly_b[i];
    if (!OPTHIN)
        for (int i = tid; i < kPolyCDoubles; i += nt) s_pc[i] = a.poly_c[i];
    __syncthreads();
    const SampleTabs tabs = {s_tab, s_pb, s_pc};
    const WalkerK k = wk[0];
    double acc = 0.0;
    // Two samples per step, the next step's table values fetched before this step's
    // arithmetic (`prefetch`): what the fused kernel's loop does.
    const int npair = a.nchunk >> 1;
    for (int r = 0; r < reps; ++r) {
        double n0 = a.nu[lane], l0 = a.lnnu[lane], q0 = a.wt[lane];
        double n1 = a.nu[64 + lane], l1 = a.lnnu[64 + lane], q1 = a.wt[64 + lane];
        for (int pr = 0; pr < npair; ++pr) {
            const int nx = (pr + 1 < npair) ? (pr + 1) * 128 + lane : lane;
            const double pn0 = a.nu[nx], pl0 = a.lnnu[nx], pq0 = a.wt[nx];
            const double pn1 = a.nu[nx + 64], pl1 = a.lnnu[nx + 64], pq1 = a.wt[nx + 64];
            const double hk8 = 8.0 * k.hokt9;
            const double Xs[2] = {hk8 * n0, hk8 * n1};
            const bool wien0 = !NOALPHA && Xs[0] > 8.0 * k.xmerge, wien1 = !NOALPHA && Xs[1] > 8.0 * k.xmerge;
            const bool far = !(Xs[0] <= kXFar8) || !(Xs[1] <= kXFar8);
            double fs[2];
            if (__builtin_amdgcn_ballot_w64(wien0 || wien1 || far) == 0) {
                fs[0] = fnu_bb_tab<OPTHIN>(k, Xs[0], l0, &tabs);
                fs[1] = fnu_bb_tab<OPTHIN>(k, Xs[1], l1, &tabs);
            } else if (!NOALPHA && __builtin_amdgcn_ballot_w64(!(wien0 && wien1)) == 0) {
                fs[0] = fnu_wien_tab(k, l0, &tabs);
                fs[1] = fnu_wien_tab(k, l1, &tabs);
            } else {
                fs[0] = fnu_sample<OPTHIN, NOALPHA, true, false>(k, n0, l0, &tabs);
                fs[1] = fnu_sample<OPTHIN, NOALPHA, true, false>(k, n1, l1, &tabs);
            }
            acc = fma(fs[0], q0, acc);
            acc = fma(fs[1], q1, acc);
            n0 = pn0; l0 = pl0; q0 = pq0; n1 = pn1; l1 = pl1; q1 = pq1;
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = acc;
    if (clk && blockIdx.x == 0 && tid == 0) {
        asm volatile("" ::"v"(acc));
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// Gate, SED constructor and parameter-only penalties of n rows, a LANE per row (k_lnlike does them on a row of 16 lanes per
// walker: right for an emcee half-step, where a constructor is latency, but ~600 wave instructions per walker -- a quarter
// of everything a 250 000-row launch executes, profiles/r04/pmc_valu_cfg5_v5.json -- where only their number counts; a
// lane per walker makes it ~40).  The same text, the same values (MBB_WC_ROW false: every exp where the row form deals
// them to its lanes): records {WalkerK, pen_u, pen_g} in a.spec, kPreWords words each, which k_lnlike's phase 1 fetches.
template <bool OPTHIN, bool NOALPHA>
__global__ void __launch_bounds__(256) k_walker_pre(const LikeArgs a)
{
    [[maybe_unused]] const int tid = threadIdx.x;           // (the diagnostic build's stamps inside the text)
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.n) return;
    double p[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) p[i] = a.pars[(size_t)w * 5 + i];
    const double lT = m_log(p[0]), lL = OPTHIN ? 0.0 : m_log(p[2]);      // vlog<true> of k_lnlike's phase 1: the same m_log
    WalkerK k;
    k.hokt9 = k.lhokt9 = k.beta = k.bp3 = k.cq = k.alpha = k.lx0 = k.xmerge = k.cbb = k.cpl = k.kap = k.peak = 0.0;
    k.status = ROW_SKIP;
    k.pad = 0;
    double pen_u = 0.0, pen_g = 0.0;
#undef MBB_WC_ROW
#define MBB_WC_ROW false
#include "mbb_walker_consts.inc"
#undef MBB_WC_ROW
#define MBB_WC_ROW true
    double *rec = a.spec + (size_t)w * kPreWords;
    const double *kd = reinterpret_cast<const double *>(&k);
#pragma unroll
    for (int i = 0; i < 13; ++i) rec[i] = kd[i];
    rec[13] = pen_u;
    rec[14] = pen_g;
}

// modified_blackbody.__init__ + max_wave for n rows, one lane per row.
template <bool OPTHIN, bool NOALPHA>
__global__ void k_prologue(const double *pars, int n, double nunorm, double lnunorm,
                           int want_peak, double *out, int32_t *status, WalkerK *wk_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double T = pars[i * 5 + 0], beta = pars[i * 5 + 1], lambda0 = pars[i * 5 + 2],
                 alpha = pars[i * 5 + 3], fnorm = pars[i * 5 + 4];
    const double nan = __builtin_nan("");
    SedScalars s;
    s.normfac = s.xmerge = s.kappa = s.hcokt = s.hokt9 = s.lhokt9 = nan;
    s.lx0 = 0.0;
    const double p5[5] = {T, beta, lambda0, alpha, fnorm};
    int st = ROW_NONFINITE;
    if (finite5(p5))
        st = sed_prologue<OPTHIN, NOALPHA, false>(T, beta, alpha, fnorm, m_log(T),
                                                  OPTHIN ? 0.0 : m_log(lambda0), nunorm, lnunorm, s);
    double peak = nan;
    WalkerK k;
    k.status = st;
    if (st == ROW_OK) {
        make_walker_k<OPTHIN, NOALPHA>(beta, alpha, s, k);
        if (want_peak) {
            int pst;
            peak = sed_peak_wave<OPTHIN, false>(T, beta, k.lx0, s.hcokt, pst);
            if (pst != ROW_OK) st = pst;
        }
    }
    k.peak = peak;
    if (out) {
        out[i * 6 + 0] = s.normfac;
        out[i * 6 + 1] = s.xmerge;
        out[i * 6 + 2] = s.kappa;
        out[i * 6 + 3] = OPTHIN ? nan : s.hcokt / lambda0;         // x0, :232
        out[i * 6 + 4] = NOALPHA ? nan : s.hcokt / s.xmerge;        // wavemerge :382-388
        out[i * 6 + 5] = peak;
    }
    if (status) status[i] = st;
    if (wk_out) wk_out[i] = k;
}

// f_nu of row blockIdx.y on a common frequency grid (modified_blackbody.py:441-554)
template <bool OPTHIN, bool NOALPHA>
__global__ void k_sed_eval(const WalkerK *wk, const double *freq, int m, double *out)
{
    const WalkerK k = wk[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double r = __builtin_nan("");
    if (k.status == ROW_OK) {
        const double nu = freq[i];
        r = fnu_sample<OPTHIN, NOALPHA>(k, nu, m_log(nu));
    }
    out[(size_t)blockIdx.y * m + i] = r;
}

// freq_integrate (modified_blackbody.py:639-674) for row blockIdx.x: the integral of
// f_nu over [numin, numax] (GHz), in mJy GHz.  The reference calls scipy's adaptive
// quad; here the integral is taken in t = log(nu), split at the merge frequency
// and at nu0, each piece cut into npanel panels with an ngl-point Gauss-Legendre
// rule: one node per lane, one wave per row.
template <bool OPTHIN, bool NOALPHA>
__global__ void k_sed_integrate(const WalkerK *wk, double numin, double numax, const double *glx,
                                const double *glw, int ngl, int npanel, double *out)
{
    const WalkerK k = wk[blockIdx.x];
    const int lane = threadIdx.x;
    double total = __builtin_nan("");
    if (k.status == ROW_OK) {
        const double t0 = m_log(numin), t1 = m_log(numax);
        // break points in t = log nu: the merge frequency (f_nu is only C1 there) and,
        // for the optically thick model, nu0 (for large beta the optical-depth factor
        // is nearly a step there)
        double tm = t1, tz = t0;
        if (!NOALPHA) tm = fmin(fmax(m_log(k.xmerge) - k.lhokt9, t0), t1);
        if (!OPTHIN) tz = fmin(fmax(k.lx0 - k.lhokt9, t0), tm);
        const double edge[4] = {t0, tz, tm, t1};
        double acc = 0.0;
        for (int piece = 0; piece < 3; ++piece) {
            const double a = edge[piece], b = edge[piece + 1];
            if (!(b > a)) continue;
            // keep a node that rounds across the merge point on its own side
            WalkerK kk = k;
            if (!NOALPHA) kk.xmerge = (piece == 2) ? 0.0 : __builtin_inf();
            const double pw = (b - a) / npanel, half = 0.5 * pw;
            for (int pn = 0; pn < npanel; ++pn) {
                const double mid = fma(pn + 0.5, pw, a);
                for (int i = lane; i < ngl; i += 64) {
                    const double t = fma(half, glx[i], mid);
                    const double nu = m_exp(t);
                    acc = fma(fnu_sample<OPTHIN, NOALPHA>(kk, nu, t) * nu, half * glw[i], acc);
                }
            }
        }
        total = wave_sum(acc);
    }
    if (lane == 0) out[blockIdx.x] = total;
}

// fnu.pyx:9-108 with explicit scalars
template <bool OPTHIN, bool NOALPHA>
__global__ void k_fnu_explicit(const double *freq, int n, double T, double beta, double x0,
                               double alpha, double normfac, double xmerge, double kappa,
                               double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SedScalars s;
    s.normfac = normfac; s.xmerge = xmerge; s.kappa = kappa; s.hcokt = 0.0;
    s.hokt9 = m_div(1e9 * kH / kK, T);
    s.lhokt9 = kLog1e9HoK - m_log(T);
    s.lx0 = OPTHIN ? 0.0 : m_log(x0);
    WalkerK k;
    make_walker_k<OPTHIN, NOALPHA>(beta, alpha, s, k);
    const double nu = freq[i];
    out[i] = fnu_sample<OPTHIN, NOALPHA>(k, nu, m_log(nu));
}

