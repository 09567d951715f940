// mbb_flowm.hip.h -- k_flowm, the one-launch sampler run with every stage of a half-step working
// ahead (sampler form 7; single GPU, single ensemble).  Included by mbb_flow.hip only.
//
// k_lnlike's one-launch form (SMODE 5) prepares a walker's proposal ahead -- the draw, the SED
// constructor and the penalties, for both outcomes of its partner's pending move -- but its mover
// waits for that partner's decision BEFORE the passband quadrature: hand-over + quadrature + band
// sums + accept test are one chain per half-step.  Here the quadrature runs ahead as well.  One
// workgroup per (pair of walkers, candidate): workgroup 2 w + c takes, in half-step j, walker w
// of the half that moves then, under the assumption c that its partner's pending move (half-step
// j - 1) is rejected (c = 0) or accepted (c = 1).  Inside the workgroup three kinds of waves form a
// pipeline, handing over through words in LDS (no workgroup barrier after the set-up):
//   C  three waves, each taking every third half-step (a proposal takes ~1.5 half-steps of a wave's
//      time).  A wave's four rows of 16 lanes run the constructor for the four outcomes of the two moves
//      of half-step j - 2 the proposal depends on (the walker's own, and its partner's partner's for
//      c = 1) -- the values either way are known once half-step j - 3 is decided -- and when those
//      two decisions arrive only a selection is left: that row's constants go to LDS for Q, its
//      proposal to the run's state (what other workgroups form their rows from).
//   Q  the quadrature waves (the units of k_lnlike's phase 2, dealt the same way).
//   E  two waves, one per half of the ensemble: band sums and lnL for this candidate (phase 3 of
//      k_lnlike), formed as soon as Q is through.  Then the partner's decision of half-step j - 1 says
//      which of the two sibling workgroups holds the proposal the chain actually makes: that one does
//      the accept test and publishes decision, row, chain entry.
// A decision therefore depends on the one a half-step earlier only through a selection, on the one
// two half-steps earlier through quadrature and band sums, and on the one three half-steps earlier
// through the constructor: the chain per half-step is the largest of (hand-over + accept test),
// (hand-over + quadrature + band sums) / 2 and (hand-over + constructor + quadrature + band sums) / 3
// instead of their sum.  The price is twice the quadrature and up to four times the constructor
// work, on CUs that were waiting.  Measured (tools/probe_flowm.py, probe_chain_flowm.py; the bench
// workload): 6.2 us per MCMC step against 10.2 for SMODE 5; what a half-step waits for is, in this
// order, the quadrature (a third), the constructor (a quarter), and the hand-over of the decisions of
// three half-steps back.
// Same draws, same arithmetic per candidate, same order of every sum: the chain is bitwise that of
// the launch train (SMODE 1).
//
// State of a run (FlowMView, in the allocation forms 5/6 use for theirs), everything filed under the
// number m of the move it belongs to, mod kFmSlots (mbb_flow_index.h; the lag guard keeps a
// slot from being overwritten under a reader):
//   prop [nw][kFmSlots][2][16]  the proposal of move m, per candidate: word 2i is coordinate i,
//                                 word 2i + 1 its check word
//   row  [kFmSlots][nw][16]     the row after move m (T, beta, lambda0, alpha, fnorm, lnprob), same pairs;
//                                 slot 0 also holds what the launch found (m = 0)
//   mseq [nw][kFmSlots]         the decision: (serial of the launch << 32) | 2 x (half-step of move m + 1) + (it was accepted)
//   done [2][kFmRing][16]       workgroups through with half-step j, running total per j mod kFmRing; a launch
//                                 uses one of the two sets and clears the other for the sampler's next launch
// A check word is (serial of the launch << 32 | half-step of the move + 1) XOR the bits of the value:
// a reader takes an element when the pair fits, whenever and in whatever order the two stores arrive,
// so nobody waits for stores to land or raises a flag after them, and nothing left in memory by an
// earlier run ever fits.  That also makes a run ONE launch with nothing before or after it: a
// workgroup files its own two walkers' rows as it finds them (slot 0) when it starts -- whoever needs
// them polls the check words like any others -- and the workgroup that decides a row's last move of the
// launch stores it back into the sampler's rows.
#pragma once
#include "mbb_kernels.hip.h"

// polls a wait of a one-launch run may make before it gives up: 2^v for option "flow_spin_log2" = v in 1..62, 2^22 for 0,
// and none for 63 -- the first word that is not there ends the run (the give-up tests: whether a budget of two polls is
// ever exceeded depends on the timing of the day)
__device__ __forceinline__ long long flow_spin_limit(int spec_cfg)
{
    const int v = (spec_cfg >> 24) & 0x3f;
    return v == 63 ? 0ll : 1ll << (v ? v : 22);
}

// LDS control words (ints) of a k_flowm workgroup
// (kFmNC C waves, kFmNB hand-over records in LDS: mbb_flow_index.h)
constexpr int kFmReady = 0;    // [kFmNB] half-step + 1 of the record last handed to Q through buffer b
constexpr int kFmQDone = 4;    // [kFmNB] Q waves that have finished a unit pass over buffer b, running total
constexpr int kFmEDone = 8;    // [kFmNB] half-step + 1 of the last record of buffer b E is through with
constexpr int kFmPen = 16;     // [kFmNB] half-step + 1 of the record of buffer b whose two penalties are there (they follow the
                               // record: the quadrature does not wait for them, the accept test does)
constexpr int kFmStaged = 12;  //        Q and E waves that have copied their share of the tables to LDS
constexpr int kFmProp = 16;    // doubles per hand-over record besides WalkerK: proposal 0..4, (dim-1) ln z,
                               // ln u, the two penalties, the walker's row as it is (9..13)
// dynamic LDS of a k_flowm launch besides the staged passband tables (bytes)
// (np = pairs of walkers a workgroup serves: the hand-over records and their control words are per pair)
__host__ __device__ constexpr size_t flowm_lds(size_t nb, size_t npart, bool cov_in_lds, size_t np = 1)
{
    return np * kFmNB * sizeof(WalkerK) + 8 * (np * kFmNB * npart + 2 * nb + np * kFmNB * kFmProp + 2 * nb + (cov_in_lds ? nb * nb : 0)) +
           8 * (nb + 2) + 8 * (kFmNC * 64) + 128 * np + 32;
}

// The lane number as the compiler cannot see through it: what a C wave derives from it (which item a lane
// fetches, its offsets and masks) is then derived again for every proposal instead of being kept in
// registers across the loop over half-steps -- 103 VGPRs instead of 122 and 66 spilled scalars instead of 87
// at the same speed with one pair per workgroup (6.184 against 6.181 us per step), and no scratch with two
// (kept across the loop over the pairs those values cost four registers the constructor then spilled).
__device__ __forceinline__ int fm_loop_lane(int l)
{
    asm volatile("" : "+v"(l));
    return l;
}
// An element and its check word: ONE 16-byte store, ONE 16-byte load (device scope: sc1, through to / served by L2).
// The pair is 16-byte aligned (records are 16 doubles, elements at even words).  Whether the 16 bytes land together
// does not matter to the protocol -- a reader takes the element only when value and check word fit, in whatever order
// they arrive -- but it halves the requests: rounds 3-5 stored and loaded the two words separately, and of the 256 KB
// per half-step the counters saw (FETCH_SIZE / WRITE_SIZE count requests, 32 or 64 bytes each, not payload) most
// were these pairs (VERDICT r05 item 6 ii; profiles/r06/form7.txt).
__device__ __forceinline__ void fm_put(double *pair, double v, unsigned long long tag)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long vb = (unsigned long long)__double_as_longlong(v), chk = tag ^ vb;
    const u32x4 d = {(unsigned int)vb, (unsigned int)(vb >> 32), (unsigned int)chk, (unsigned int)(chk >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(pair), "v"(d) : "memory");
}
__device__ __forceinline__ bool fm_get(const double *pair, unsigned long long tag, double &v)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 d;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(pair) : "memory");
    const unsigned long long vb = ((unsigned long long)d.y << 32) | d.x, chk = ((unsigned long long)d.w << 32) | d.z;
    v = __longlong_as_double((long long)vb);
    return (chk ^ vb) == tag;
}

// What every role needs of the launch, declared inside the role after its own argument pointer (what a
// role does not use is dead code there): band counts, the layout of the dynamic LDS, the run's state.
#define MBB_FM_COMMON() \
    const int nun = a.nunit, npart = a.npart, nb = a.nb; \
    const int wbase = ((int)blockIdx.x >> 1) * NP, cand = (int)blockIdx.x & 1; \
    WalkerK *wk0 = reinterpret_cast<WalkerK *>(smem_raw); \
    double *partial0 = reinterpret_cast<double *>(wk0 + NP * kFmNB); \
    double *mflux_all = partial0 + NP * kFmNB * (size_t)npart; \
    double *prop0 = mflux_all + 2 * nb; \
    double *s_flux = prop0 + NP * kFmNB * kFmProp; \
    double *s_ivar = s_flux + nb; \
    double *s_invcov = s_ivar + nb; \
    int2 *s_band = reinterpret_cast<int2 *>(s_invcov + (a.cov_in_lds ? (size_t)nb * nb : 0)); \
    double *cscr = reinterpret_cast<double *>(s_band + nb + 1); \
    int *ctl0 = reinterpret_cast<int *>(cscr + kFmNC * 64); \
    const size_t tab_off = ((size_t)(reinterpret_cast<unsigned char *>(ctl0 + 32 * NP) - smem_raw) + 15) & ~(size_t)15; \
    double *s_nu = reinterpret_cast<double *>(smem_raw + tab_off); \
    double *s_lnnu = s_nu + (STAGE ? a.nchunk * 64 : 0); \
    double *s_wt = s_lnnu + (STAGE ? a.nchunk * 64 : 0); \
    const long long spin_limit = flow_spin_limit(a.spec_cfg); \
    const FlowMView fv = flowm_view(a.spec, a.nw); \
    const unsigned long long serial32 = a.flow_serial << 32; \
    unsigned long long *const done_set = fv.done + (size_t)(a.spec_cfg & 1) * kFmRing * 16; \
    const int niter = a.persist;
// One workgroup serves NP pairs of walkers.  Only NP = 1 is instantiated since round 4 (up to two walkers per CU);
// round 3's NP = 2 for ensembles of up to four per CU -- the roles taking the pairs one after another in every
// half-step, each pair with hand-over records and control words of its own: 11.5 us per step -- was superseded by
// k_flowa (8.2-8.6).  The loops over `vp` below are the shape that form had; with NP = 1 they are one pass.
#define MBB_FM_PAIRS_LOOP _Pragma("clang loop unroll(disable)")      /* (one body for both pairs, not two copies) */
#define MBB_FM_PAIR(vp) \
    const int w = wbase + (vp); \
    if constexpr (NP > 1) { if (w >= a.n) continue; }      /* (an odd number of pairs: the last workgroup's second is not there) */ \
    WalkerK *const wk = wk0 + (vp) * kFmNB; \
    double *const partial = partial0 + (size_t)(vp) * kFmNB * npart; \
    double *const prop = prop0 + (vp) * kFmNB * kFmProp; \
    int *const ctl = ctl0 + 32 * (vp); \
    (void)wk; (void)partial; (void)prop; (void)ctl
// ... and the waits.  Hand-over words in LDS: a wave's LDS operations execute in the order it issued
// them, so data then word (writer) and word then data (reader) need no wait in between, only the
// compiler's order; every wait is bounded and gives up once the run's error flag is up.
#define MBB_FM_ORDER() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define MBB_FM_WAITS() \
    auto dec_ok = [&](unsigned long long v, unsigned long long need) {   /* a decision word of this launch whose half-step + 1 is at least `need` */ \
        return (v >> 32) == (serial32 >> 32) && ((v & 0xffffffffull) >> 1) >= need; \
    }; \
    auto lds_wait = [&](int *word, int need) { \
        long long spins = 0; \
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) { \
            ++spins; \
            if (spins > spin_limit * 16 || \
                ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { \
                atomicMax(a.errflag, 9); \
                break; \
            } \
            __builtin_amdgcn_s_sleep(1); \
        } \
        MBB_FM_ORDER(); \
    }; \
    auto lds_post = [&](int *word, int v) { \
        MBB_FM_ORDER(); \
        __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    }; \
    (void)dec_ok; (void)lds_wait; (void)lds_post
// (the wave's number as a scalar: the roles are then uniform branches, not exec-masked regions)
#ifdef MBB_WAVE_VECTOR
#define MBB_WAVE_ID(t) ((t) >> 6)
#else
#define MBB_WAVE_ID(t) __builtin_amdgcn_readfirstlane((t) >> 6)
#endif

template <bool OPTHIN, bool NOALPHA, bool STAGE, int NP>
__global__ void __launch_bounds__(1024) k_flowm(const LikeArgs a)
{
    static_assert(NP == 1, "one pair of walkers per workgroup (the two-pair shape is not built or tested any more)");
    CLikeArgs *const ka = MBB_KERNARGS();
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ __align__(16) double s_tab[kExp2N];
    __shared__ __align__(16) double s_pb[kPolyBDoubles];
    __shared__ __align__(16) double s_pc[OPTHIN ? 2 : kPolyCDoubles];
    const int tid = threadIdx.x, lane = tid & 63, wave = MBB_WAVE_ID(tid);
    const int nwave = blockDim.x >> 6, nq = nwave - kFmNC - 2;
    // Which wave does what.  The constructor is one long dependent chain and runs fastest on a SIMD it
    // does not share with the quadrature's bursts: waves 3, 7, 11 and 15 (one SIMD: a workgroup's waves
    // go round the four in turn) are the three C waves and E0, wave 14 is E1, the eleven others are Q
    // waves, numbered in order.  (Fewer than 16 waves: the last five.)  Three C waves, a half-step in
    // three each, and two E waves, one per half of the ensemble: a proposal takes ~1.5 half-steps of a
    // wave's time, an E pass ~0.4, and one that had to queue behind the wave's previous one was what a
    // half-step waited for most often (tools/probe_chain_flowm.py).
    const bool spread = nwave == 16;
    const int role = spread ? ((wave & 3) == 3 ? 1 + (wave >> 2) : (wave == 14 ? 2 + kFmNC : 0))
                            : (wave < nq ? 0 : 1 + wave - nq);                    // 0 Q, 1..3 C, 4 E0, 5 E1
    const int qi = spread ? wave - (wave >> 2) : wave;                           // Q wave number (14 is not one)
    // ---- set-up, once per launch ---------------------------------------------------------------
    // control words clear; this pair's two rows as the sampler holds them -> slot 0 of the run's state,
    // with this launch's check words (no kernel before this one; candidate 0's workgroup does it);
    // workgroup 0 clears the completion counters of the sampler's NEXT launch.  The tables the
    // quadrature and the band sums need go to LDS from the Q and E waves, behind a counter of their
    // own: the C waves start on the launch's first proposals meanwhile (a launch's fixed cost is what
    // a short run is made of: tools/probe_flowm_short.py).
    {
        MBB_ROLE_ARGS();
        MBB_FM_COMMON();
        if (tid < 32 * NP) ctl0[tid] = 0;
        if (cand == 0 && tid < 12 * NP && wbase + tid / 12 < a.n) {
            const int t12 = tid % 12;
            const int r = (t12 < 6 ? 0 : a.c_count) + wbase + tid / 12, e = t12 < 6 ? t12 : t12 - 6;
            fm_put(fv.row + (size_t)r * kFmWords + 2 * e, a.pos6[(size_t)r * 6 + e], serial32);
        }
        if (blockIdx.x == 0 && tid < kFmRing * 16)
            __hip_atomic_store(fv.done + (size_t)((a.spec_cfg & 1) ^ 1) * kFmRing * 16 + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)done_set; (void)niter; (void)spin_limit;
    }
    __syncthreads();
    // diagnostic build: cycles a wave spends in each part of its loop, summed over the launch
    // -> stamps[(workgroup * 16 + wave) * 8 + part] (tools/probe_stamps_flowm.py)
#ifdef MBB_STAMPS
    // (when this workgroup got past its set-up, on the clock all CUs share: tools/probe_flowm_start.py)
    if (tid == 0 && a.stamps) a.stamps[(1u << 20) + 4096 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#define FM_T(kk) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[kk] += t_now - t_last; t_last = t_now; } while (0)
#define FM_TD(kk, dep) do { asm volatile("" ::"v"(dep)); FM_T(kk); } while (0)
#define FM_TOUT() do { if (lane == 0 && a.stamps) for (int kk = 0; kk < 8; ++kk) a.stamps[((size_t)blockIdx.x * 16 + wave) * 8 + kk] = t_acc[kk]; } while (0)
    if (lane == 0 && a.stamps) {                                  // where the wave runs: HW_ID (SIMD in bits 4-5, CU 8-11, SE 13-15)
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.stamps[(1u << 20) + blockIdx.x * 16 + wave] = ((unsigned long long)xcc << 32) | hwid;
    }
// ... and when things happened in the launch's last 64 half-steps, on the clock all CUs share (100 MHz):
// log[(workgroup * 64 + half-step mod 64) * 8 + event], tools/probe_chain_flowm.py
#define FM_EV(jj, ev) do { if (lane == 0 && a.stamps && (jj) >= niter - 64) \
        a.stamps[(1u << 20) + 8192 + (((size_t)blockIdx.x * 64 + ((jj) & 63)) * 16 + (ev))] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FM_EVV(jj, ev, val) do { if (lane == 0 && a.stamps && (jj) >= niter - 64) \
        a.stamps[(1u << 20) + 8192 + (((size_t)blockIdx.x * 64 + ((jj) & 63)) * 16 + (ev))] = (unsigned long long)(val); } while (0)
#else
#define FM_EV(jj, ev) do { } while (0)
#define FM_EVV(jj, ev, val) do { } while (0)
#define FM_T(kk) do { } while (0)
#define FM_TD(kk, dep) do { } while (0)
#define FM_TOUT() do { } while (0)
#endif

    if (role == 0 || role > kFmNC) {
        MBB_ROLE_ARGS();
        MBB_FM_COMMON();
        MBB_FM_WAITS();
        // tables -> LDS: the Q and E waves' threads, numbered through
        const int ns = nq + 2, si = role == 0 ? qi : nq + (role - 1 - kFmNC);
        const int t0 = si * 64 + lane, nt = ns * 64;
        const double2 *gb = reinterpret_cast<const double2 *>(a.poly_b);
        const double2 *gc = reinterpret_cast<const double2 *>(a.poly_c);
        double2 *lb = reinterpret_cast<double2 *>(s_pb);
        double2 *lc = reinterpret_cast<double2 *>(s_pc);
        for (int i = t0; i < kExp2N; i += nt) s_tab[i] = kExp2Tab[i];
        for (int i = t0; i < kPolyBDoubles / 2; i += nt) lb[i] = gb[i];
        if (!OPTHIN)
            for (int i = t0; i < kPolyCDoubles / 2; i += nt) lc[i] = gc[i];
        for (int bb = t0; bb < nb; bb += nt) { s_flux[bb] = a.flux[bb]; s_ivar[bb] = a.ivar[bb]; s_band[bb] = a.band_rng[bb]; }
        if (a.cov_in_lds)
            for (int i = t0; i < nb * nb; i += nt) s_invcov[i] = a.invcov[i];
        if (STAGE) {
            const int n2 = a.nchunk * 32;
            const double2 *g0 = reinterpret_cast<const double2 *>(a.nu), *g1 = reinterpret_cast<const double2 *>(a.lnnu),
                          *g2 = reinterpret_cast<const double2 *>(a.wt);
            double2 *l0 = reinterpret_cast<double2 *>(s_nu), *l1 = reinterpret_cast<double2 *>(s_lnnu),
                    *l2 = reinterpret_cast<double2 *>(s_wt);
            for (int i = t0; i < n2; i += nt) { l0[i] = g0[i]; l1[i] = g1[i]; l2[i] = g2[i]; }
        }
        // every share is in LDS before any of these waves reads a table
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        MBB_FM_ORDER();
        if (lane == 0) __hip_atomic_fetch_add(ctl0 + kFmStaged, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_wait(ctl0 + kFmStaged, ns);
    }
    // (the chains of C and E are what a half-step waits for: ahead of the Q wave they share a SIMD with)
    if (role != 0) __builtin_amdgcn_s_setprio(3);
    // =========================== Q: the passband quadrature ====================================
    if (role == 0) {
        MBB_ROLE_ARGS();
        MBB_FM_COMMON();
        MBB_FM_WAITS();
        MBB_PIN(a.unit_tab); MBB_PIN(a.tail_slot); MBB_PIN(a.errflag);
        if (!STAGE) { MBB_PIN(a.nu); MBB_PIN(a.lnnu); MBB_PIN(a.wt); }
        auto T_nu = [&](int i) { if constexpr (STAGE) return s_nu[i]; else return a.nu[i]; };
        auto T_ln = [&](int i) { if constexpr (STAGE) return s_lnnu[i]; else return a.lnnu[i]; };
        auto T_wt = [&](int i) { if constexpr (STAGE) return s_wt[i]; else return a.wt[i]; };
        const SampleTabs tabs = {s_tab, s_pb, s_pc};
        int4 us_first = make_int4(0, 0, 0, 0);
        if (qi < nun) us_first = a.unit_tab[qi];
        int tail_first = -1;
        if (qi < nun && us_first.w == 2) tail_first = a.tail_slot[4 * us_first.x + (lane >> 4)];
        for (int it = 0; it < niter; ++it)
        MBB_FM_PAIRS_LOOP
        for (int vp = 0; vp < NP; ++vp) {
            MBB_FM_PAIR(vp);
            const int b = it & (kFmNB - 1);
            lds_wait(ctl + kFmReady + b, it + 1);
            FM_T(0);
            if (qi == 1) FM_EV(it, 14);
            const WalkerK *wkb = wk + b;
            double *part = partial + (size_t)b * npart;
            if (wkb->status == ROW_OK) {                          // wave-uniform
                const WalkerK k = *wkb;
                for (int u = qi; u < nun; u += nq) {
                    const int4 us = (u == qi) ? us_first : a.unit_tab[u];
                    const int s = us.x, c0 = us.y, c1 = us.z;
                    double acc = 0.0;
                    int c = c0;
                    for (; c + 2 <= c1; c += 2) {                 // two chunks per step (k_lnlike, do_unit)
                        const int i0 = c * 64 + lane, i1 = i0 + 64;
                        const double n0 = T_nu(i0), l0 = T_ln(i0), q0 = T_wt(i0);
                        const double n1 = T_nu(i1), l1 = T_ln(i1), q1 = T_wt(i1);
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
                    if (us.w == 0) {
                        acc = wave_sum_l63(acc);           // (the total is lane 63's)
                        if (lane == 63) part[s] = acc;
                    } else if (us.w == 2) {
                        acc = row_sum(acc);
                        if ((lane & 15) == 0) {
                            const int sl = (u == qi) ? tail_first : a.tail_slot[4 * s + (lane >> 4)];
                            if (sl >= 0) part[sl] = acc;
                        }
                    } else {
                        part[s + lane] = acc;
                    }
                }
            }
            MBB_FM_ORDER();
#ifdef MBB_STAMPS
            if (lane == 0) {
                const int cnt = __hip_atomic_fetch_add(ctl + kFmQDone + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cnt + 1 == nq * ((it / kFmNB) + 1)) FM_EV(it, 15);      // the last Q wave through with this record
            }
#else
            if (lane == 0) __hip_atomic_fetch_add(ctl + kFmQDone + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
            FM_T(1);
        }
        FM_TOUT();
        return;
    }

    // =========================== E: band sums, lnL, and the move if it is this candidate's =====
    if (role > kFmNC) {
        MBB_ROLE_ARGS();
        MBB_FM_COMMON();
        MBB_FM_WAITS();
        MBB_PIN(a.n); MBB_PIN(a.nw); MBB_PIN(a.c_count); MBB_PIN(a.step); MBB_PIN(a.seed); MBB_PIN(a.stretch_a);
        MBB_PIN(a.has_gprior); MBB_PIN(a.invcov); MBB_PIN(a.cov_in_lds); MBB_PIN(a.pos6); MBB_PIN(a.chain6);
        MBB_PIN(a.nacc); MBB_PIN(a.errflag);
        double *mflux = mflux_all + (size_t)(role - 1 - kFmNC) * nb;     // (the two E waves run side by side)
        for (int it = role - 1 - kFmNC; it < niter; it += 2)
        MBB_FM_PAIRS_LOOP
        for (int vp = 0; vp < NP; ++vp) {
            MBB_FM_PAIR(vp);
            const int b = it & (kFmNB - 1);
            const int L_step = a.step + (it >> 1), L_half = it & 1;
            const unsigned long long L_seed = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(it >> 1);
            const int row = (L_half ? a.c_count : 0) + w, c_begin = L_half ? 0 : a.c_count;
            // (the walker's partner in this half-step: the draw C made for the proposal; E's first look
            // must not wait for C, so it is made again here -- E has two half-steps per pass)
            double zz, u3;
            int pj;
            stretch_draw(row, L_step, L_half, L_seed, a.stretch_a, a.c_count, zz, pj, u3);
            const int prow = c_begin + pj;
            const int m_par = flow_cnt(L_half ^ 1, it), m_s = flow_cnt(L_half, it);
            const unsigned long long need_p = (unsigned long long)flow_seq(L_half ^ 1, m_par);
            // one loop, one round trip when everything is there: lane 21 the walker's lnprob as it is
            // (element 5 of its row after move m_s), lane 22 the partner's decision of half-step it - 1,
            // lane 23 the lag guard
            const double *lnp_p = fv.row + ((size_t)(m_s % kFmSlots) * a.nw + row) * kFmWords + 10;
            const unsigned long long tag_s = serial32 | (unsigned long long)flow_seq(L_half, m_s);
            const unsigned long long need_g = 2ull * (unsigned long long)a.n * (unsigned long long)(((it - kFmLag) / kFmRing) + 1);
            const unsigned long long *word = lane == 22 ? fv.mseq + (size_t)prow * kFmMseq + (m_par % kFmSlots)
                                                        : done_set + ((it - kFmLag) & (kFmRing - 1)) * 16;
            const bool watch = (lane == 22 && need_p > 0) || (lane == 23 && it >= kFmLag);
            // The band sums do not wait for the partner: as soon as Q is through they are formed, between
            // asking for the words and looking at the answers; whichever comes last -- the partner's
            // decision or the sums -- is followed by the accept test alone.
            const WalkerK *wkb = wk + b;
            const double *pr = prop + b * kFmProp;
            int st = ROW_SKIP;
            double cbb = 0.0, pen_u = 0.0, pen_g = 0.0, lnz4 = 0.0, lnu = 0.0, acc = 0.0;
            double q[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, old5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            auto sums = [&]() {
                lds_wait(ctl + kFmPen + b, it + 1);                    // (C posts the penalties behind the record)
                st = wkb->status;
                cbb = wkb->cq; pen_u = pr[7]; pen_g = pr[8];
#pragma unroll
                for (int i = 0; i < 5; ++i) { q[i] = pr[i]; old5[i] = pr[9 + i]; }
                lnz4 = pr[5]; lnu = pr[6];
                if (st == ROW_OK) {
                    const double *pj2 = partial + (size_t)b * npart;
                    auto band = [&](const int bb) {                    // band flux, fixed order (k_lnlike, phase 3)
                        // (the product and the difference are separate roundings, as there: the library is
                        // built with -ffp-contract=off and every fused multiply-add is an explicit fma())
                        double sum = 0.0;
                        const int2 rng = s_band[bb];
                        for (int sg = rng.x; sg < rng.y; sg += 4) {
                            const int l = rng.y - 1;
                            const double q0 = pj2[sg], q1 = pj2[min(sg + 1, l)], q2 = pj2[min(sg + 2, l)], q3 = pj2[min(sg + 3, l)];
                            sum += q0;
                            if (sg + 1 < rng.y) sum += q1;
                            if (sg + 2 < rng.y) sum += q2;
                            if (sg + 3 < rng.y) sum += q3;
                        }
                        sum *= cbb;
                        const double d = s_flux[bb] - sum;             // likelihood.py:821
                        if (a.invcov) mflux[bb] = d;
                        else acc = fma(d * d, s_ivar[bb], acc);        // :825
                    };
                    for (int bb = lane; bb < nb; bb += 64) band(bb);
                    if (a.invcov) {                                    // :823
                        MBB_FM_ORDER();
                        for (int i = lane; i < nb; i += 64) {
                            double t = 0.0;
                            const double *crow = (a.cov_in_lds ? s_invcov : a.invcov) + (size_t)i * nb;
                            for (int jj = 0; jj < nb; ++jj) t = fma(crow[jj], mflux[jj], t);
                            acc = fma(mflux[i], t, acc);
                        }
                    }
                    acc = (nb <= 16) ? wave_sum_row0(acc) : wave_sum(acc);
                }
                MBB_FM_ORDER();
                if (lane == 0) lds_post(ctl + kFmEDone + b, it + 1);   // buffer b may be written again
            };
            const int q_need = nq * ((it / kFmNB) + 1);
            FM_EV(it, 9);
#ifdef MBB_STAMPS
            unsigned long long t_ok = 0;
#endif
            unsigned long long pv = 0;
            double lnp = 0.0;
            bool ok = !(watch || lane == 21), have_sums = false;
            long long spins = 0;
            for (;;) {
                // (asked for ...)
                double lv = 0.0;
                unsigned long long lchk = 0, wv = 0;
                const bool ask_l = lane == 21 && !ok, ask_w = watch && !ok;
                if (ask_l) {
                    lv = ld_dev(lnp_p);
                    lchk = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(lnp_p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (ask_w) wv = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (... the sums meanwhile, if Q is through ...)
                if (!have_sums && __hip_atomic_load(ctl + kFmQDone + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= q_need) {
                    MBB_FM_ORDER();
                    FM_EV(it, 4);
                    sums();
                    have_sums = true;
                }
                // (... and looked at)
                if (ask_l) { lnp = lv; ok = (lchk ^ (unsigned long long)__double_as_longlong(lv)) == tag_s; }
                if (ask_w) { pv = wv; ok = lane == 22 ? dec_ok(pv, need_p) : pv >= need_g; }
#ifdef MBB_STAMPS
                if ((ask_l || ask_w) && ok) t_ok = __builtin_amdgcn_s_memrealtime();
#endif
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                ++spins;
                if (spins > spin_limit ||
                    ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    atomicMax(a.errflag, 9);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const int flag = (need_p > 0 && (__shfl(pv, 22) & 1ull)) ? 1 : 0;
            const bool mine = flag == cand;                       // this workgroup's candidate is the chain's proposal
            const double lnp_cur = __shfl(lnp, 21);
            FM_TD(0, lnp_cur);
            FM_EV(it, 5);
            FM_EVV(it, 7, prow);
#ifdef MBB_STAMPS
            { const unsigned long long t21 = __shfl(t_ok, 21), t22 = __shfl(t_ok, 22), t23 = __shfl(t_ok, 23);
              FM_EVV(it, 10, t21); FM_EVV(it, 11, t22); FM_EVV(it, 12, t23); }
#endif
            if (!have_sums) {
                lds_wait(ctl + kFmQDone + b, q_need);
                FM_T(1);
                FM_EV(it, 4);
                sums();
            }
            FM_TD(2, acc);
            {
                // (every lane: the band total is in all of them)
                double r;
                if (st == ROW_BELOW_LOWLIM) r = -__builtin_inf();
                else if (st != ROW_OK) r = __builtin_nan("");
                else {
                    r = fma(-0.5, acc, pen_u);                     // :828
                    if (a.has_gprior) r += pen_g;                  // :830-831
                }
                if (mine) {
                    // (after a wait has given up -- error 9 -- the stages run on with whatever is in their
                    // buffers: such a status must not outrank the 9 the host falls back on)
                    if (lane == 0 && (st >= 2 || r != r)) atomicMax(a.errflag, (st >= 2 && st <= (int)ROW_NONFINITE) ? st : (int)ROW_NONFINITE);
                    const bool accept = (lnz4 + r - lnp_cur) > lnu;
                    const int m_new = m_s + 1;
                    // the decision first, then the row as it is after this half-step, an element and its
                    // check word per lane; counts and chain are for the host: plain stores
                    FM_EV(it, 6);
                    if (lane == 0)
                        __hip_atomic_store(fv.mseq + (size_t)row * kFmMseq + (m_new % kFmSlots),
                                           serial32 | (2ull * (unsigned long long)(it + 1) + (accept ? 1ull : 0ull)), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    const double lnp_new = accept ? r : lnp_cur;
                    double ve = lnp_new;
#pragma unroll
                    for (int i = 0; i < 5; ++i) ve = ((lane & 7) == i) ? (accept ? q[i] : old5[i]) : ve;
                    if (lane < 6) {
                        fm_put(fv.row + ((size_t)(m_new % kFmSlots) * a.nw + row) * kFmWords + 2 * lane, ve,
                               serial32 | (unsigned long long)(it + 1));
                        // the row's last move of the launch: back into the sampler's rows (no kernel after this one)
                        if (it + 2 >= niter) a.pos6[(size_t)row * 6 + lane] = ve;
                    }
                    else if (lane >= 8 && lane < 14 && a.chain6)
                        a.chain6[((size_t)it * a.n + w) * 6 + (lane - 8)] = ve;
                    if (lane == 0 && accept) atomicAdd(a.nacc + (size_t)L_half * a.n + w, 1u);
                }
                if (lane == 0) __hip_atomic_fetch_add(done_set + (it & (kFmRing - 1)) * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            FM_T(3);
        }
        FM_TOUT();
        return;
    }

    // =========================== C: the proposals, worked out ahead of their decisions =========
    {
        MBB_ROLE_ARGS();
        MBB_FM_COMMON();
        MBB_FM_WAITS();
        MBB_PIN(a.c_count); MBB_PIN(a.step); MBB_PIN(a.seed); MBB_PIN(a.stretch_a); MBB_PIN(a.nw); MBB_PIN(a.errflag);
        MBB_PIN(a.lowlim[0]); MBB_PIN(a.lowlim[1]); MBB_PIN(a.lowlim[2]); MBB_PIN(a.lowlim[3]); MBB_PIN(a.lowlim[4]);
        MBB_PIN(a.nunorm); MBB_PIN(a.lnunorm); MBB_PIN(a.has_uplim); MBB_PIN(a.has_gprior);
        const int cb = role - 1;                                  // this wave takes the half-steps j = cb mod kFmNC
        const int lane_w = lane;
        double *scr = cscr + (size_t)cb * 64;
        auto spin = [&](const unsigned long long *word, unsigned long long need, bool watch) {    // for decision words
            unsigned long long v = 0;
            long long spins = 0;
            for (;;) {
                bool ok = true;
                if (watch) { v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = dec_ok(v, need); }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                ++spins;
                if (spins > spin_limit ||
                    ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    atomicMax(a.errflag, 9);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            return v;
        };
        for (int j = cb; j < niter; j += kFmNC)
        MBB_FM_PAIRS_LOOP
        for (int vp = 0; vp < NP; ++vp) {
            MBB_FM_PAIR(vp);
            const int lane = fm_loop_lane(lane_w);
            const int vrow = lane >> 4, l16 = lane & 15, base = lane & 48;
            const int hj = j & 1;
            const int sb = hj ? a.c_count : 0, ob = hj ? 0 : a.c_count;    // the half that moves in j / the other
            const int tn = a.step + (j >> 1);
            const unsigned long long seed_n = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(j >> 1);
            const int tp = a.step + ((j - 1) >> 1), hp = hj ^ 1;
            const unsigned long long seed_p = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)((j - 1) >> 1);
            const bool c1 = cand && j > 0;
            const int rown = sb + w;
            double zz = 1.0, u3 = 0.5, zp = 1.0, up;
            int pj = 0, pjp = 0;
            stretch_draw(rown, tn, hj, seed_n, a.stretch_a, a.c_count, zz, pj, u3);
            if (c1) stretch_draw(ob + pj, tp, hp, seed_p, a.stretch_a, a.c_count, zp, pjp, up);
            // The rows this proposal starts from -- the walker's own and, for candidate 1, its
            // partner's partner -- made their last move (number m_s) in half-step j - 2.  Either is put
            // together from what was known before that move was decided: the row as it was, and the
            // proposal it was tested on, whose candidate is the one the decision of ITS partner in
            // half-step j - 3 says (k_lnlike, SMODE 5, the workgroups that work ahead).
            const int m_s = flow_cnt(hj, j - 1), m_o = flow_cnt(hj ^ 1, j - 1);
            const int m_next = m_s + 1;
            const int g = j - 2;
            const int m_q = flow_cnt(hj ^ 1, g);
            const int pprow = sb + pjp, prow = ob + pj;
            int qr = 0, qp = 0;
            if (m_s > 0) {
                const int tg = a.step + (g >> 1);
                const unsigned long long seed_g = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(g >> 1);
                double u0, z0;
                stretch_draw(rown, tg, hj, seed_g, a.stretch_a, a.c_count, z0, qr, u0);
                if (c1) stretch_draw(pprow, tg, hj, seed_g, a.stretch_a, a.c_count, z0, qp, u0);
            }
            int snv_off = 0, cpv_off = 15;
            // (1) everything that is settled once half-step j - 3 is decided, one item per lane, all asked
            // for in the same round (one round trip when it is all there) -> LDS:
            //   lane 0, 1    the two decisions of j - 3 that say which candidate the rows' last proposals were
            //   lane 2..6    the walker's row before its last move              -> scr[0..5)
            //   lane 7..11   its partner's partner's                            -> scr[15..20)
            //   lane 12..16  its partner's row as it is (before the pending move) -> scr[30..35)
            //   lane 17..26  the proposals of the walker's last move, both candidates -> scr[5..15)
            //   lane 27..36  the same for the partner's partner                 -> scr[20..30)
            {
                const bool has = m_s > 0;
                const int so = (has ? m_s - 1 : 0) % kFmSlots;
                const unsigned long long tag_old = serial32 | (unsigned long long)flow_seq(hj, has ? m_s - 1 : 0);
                const unsigned long long tag_g = serial32 | (unsigned long long)(g + 1);
                const unsigned long long tag_o = serial32 | (unsigned long long)flow_seq(hj ^ 1, m_o);
                const double *src = fv.row;
                unsigned long long tag = 0, need = 0;
                const unsigned long long *wsrc = fv.mseq;
                int slot = -1, kind = 0;                              // kind 1: a decision word, 2: an element
                if (lane < 2) {
                    const int r = lane == 0 ? qr : qp;
                    wsrc = fv.mseq + (size_t)(ob + r) * kFmMseq + (m_q % kFmSlots);
                    need = (unsigned long long)flow_seq(hj ^ 1, m_q);
                    kind = (has && m_q > 0 && (lane == 0 || c1)) ? 1 : 0;
                } else if (lane < 7) {
                    src = fv.row + ((size_t)so * a.nw + rown) * kFmWords + 2 * (lane - 2); tag = tag_old; slot = lane - 2; kind = 2;
                } else if (lane < 12) {
                    src = fv.row + ((size_t)so * a.nw + pprow) * kFmWords + 2 * (lane - 7); tag = tag_old; slot = 15 + lane - 7; kind = c1 ? 2 : 0;
                } else if (lane < 17) {
                    src = fv.row + ((size_t)(m_o % kFmSlots) * a.nw + prow) * kFmWords + 2 * (lane - 12); tag = tag_o; slot = 30 + lane - 12; kind = 2;
                } else if (lane < 27) {
                    const int c = (lane - 17) / 5, i = (lane - 17) - 5 * c;
                    src = fv.prop + (((size_t)rown * kFmSlots + (m_s % kFmSlots)) * 2 + c) * kFmWords + 2 * i; tag = tag_g; slot = 5 + lane - 17;
                    kind = has ? 2 : 0;
                } else if (lane < 37) {
                    const int c = (lane - 27) / 5, i = (lane - 27) - 5 * c;
                    src = fv.prop + (((size_t)pprow * kFmSlots + (m_s % kFmSlots)) * 2 + c) * kFmWords + 2 * i; tag = tag_g; slot = 20 + lane - 27;
                    kind = (has && c1) ? 2 : 0;
                }
                FM_EV(j, 13);
                bool ok = kind == 0;
                double v = 0.0;
                unsigned long long dv = 0;
                long long spins = 0;
                for (;;) {
                    if (kind == 2 && !ok) ok = fm_get(src, tag, v);
                    if (kind == 1 && !ok) { dv = __hip_atomic_load(wsrc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = dec_ok(dv, need); }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                    ++spins;
                    if (spins > spin_limit ||
                        ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        atomicMax(a.errflag, 9);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (slot >= 0) scr[slot] = v;
                FM_TD(0, v);
                FM_EV(j, 0);
                // (the rows whose decisions this proposal waits for: partner, partner's partner, the two partners of j - 2)
                FM_EVV(j, 8, (unsigned long long)prow | ((unsigned long long)pprow << 10) | ((unsigned long long)(ob + qr) << 20) |
                                 ((unsigned long long)(ob + qp) << 30) | ((unsigned long long)(c1 ? 1 : 0) << 40) | ((unsigned long long)(m_s > 0 ? 1 : 0) << 41));
                MBB_FM_ORDER();
                // (a decision word that was not waited for reads as candidate 0)
                const int cr = (int)(__shfl(dv, 0) & 1ull), cp = (int)(__shfl(dv, 1) & 1ull);
                // this row of lanes' assumption about the two moves of half-step j - 2
                const bool ar_v = has && (vrow & 1), ap_v = has && c1 && (vrow & 2);
                snv_off = ar_v ? 5 + 5 * cr : 0;
                cpv_off = ap_v ? 20 + 5 * cp : 15;
            }
            double snv[5], cpos[5], cpv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) { snv[i] = scr[snv_off + i]; cpv[i] = scr[cpv_off + i]; cpos[i] = scr[30 + i]; }
            FM_T(1);
            if (c1) {
#pragma unroll
                for (int i = 0; i < 5; ++i) cpos[i] = stretch_q(cpv[i], cpos[i], zp);
            }
            double p[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) p[i] = stretch_q(cpos[i], snv[i], zz);
            double lo[4];
            vlog<true>(lo, p[0], p[2], zz, u3);
            WalkerK k;
            k.hokt9 = k.lhokt9 = k.beta = k.bp3 = k.cq = k.alpha = k.lx0 = k.xmerge = k.cbb = k.cpl = k.kap = k.peak = 0.0;
            k.status = ROW_SKIP;
            k.pad = 0;
            double pen_u = 0.0, pen_g = 0.0;
            const double lT = lo[0], lL = lo[1];
            // (2) the two decisions of half-step j - 2 say which row of lanes was right.  They are asked
            // for once towards the end of the constructor, so that the answer is there when it is through
            // (a decision that arrives later is polled for afterwards)
            const unsigned long long *w2 = fv.mseq + (size_t)(l16 == 0 ? rown : pprow) * kFmMseq + (m_s % kFmSlots);
            const bool watch2 = m_s > 0 && (l16 == 0 || (l16 == 1 && c1));
            const unsigned long long need2 = (unsigned long long)flow_seq(hj, m_s);
            unsigned long long v2 = 0;
#define MBB_WC_AFTER_PROLOGUE if (watch2) v2 = __hip_atomic_load(w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#define MBB_WC_PENALTIES_LATER
#include "mbb_walker_consts.inc"
#undef MBB_WC_PENALTIES_LATER
#undef MBB_WC_AFTER_PROLOGUE
#ifdef MBB_STAMPS
            {
                asm volatile("" ::"v"(pen_u + pen_g + k.cbb));
                const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_last;
                if (dt > t_acc[6]) t_acc[6] = dt;
                if (t_acc[7] == 0 || dt < t_acc[7]) t_acc[7] = dt;
            }
#endif
            FM_TD(2, pen_u + pen_g + k.cbb);
            FM_EV(j, 1);
            if (__builtin_amdgcn_ballot_w64(watch2 && !dec_ok(v2, need2)) != 0) v2 = spin(w2, need2, watch2);
            const bool ar = m_s > 0 && (__shfl(v2, base + 0) & 1ull), ap = m_s > 0 && c1 && (__shfl(v2, base + 1) & 1ull);
            const int vsel = (ar ? 1 : 0) | (ap ? 2 : 0);
            FM_TD(3, v2);
            FM_EV(j, 2);
            // the record buffer must be free: E is through with half-step j - 2
            const int bj = j & (kFmNB - 1);
            if (j >= kFmNB) lds_wait(ctl + kFmEDone + bj, j - kFmNB + 1);
            FM_T(4);
            if (vrow == vsel && l16 == 0) {
                wk[bj] = k;
                double *pr = prop + bj * kFmProp;
#pragma unroll
                for (int i = 0; i < 5; ++i) { pr[i] = p[i]; pr[9 + i] = snv[i]; }
                pr[5] = 4.0 * lo[2];                              // (dim - 1) ln z, dim = 5
                pr[6] = lo[3];                                    // ln u
                lds_post(ctl + kFmReady + bj, j + 1);             // (the quadrature may start; the penalties follow)
#ifdef MBB_STAMPS
                if (a.stamps && j >= niter - 64)
                    a.stamps[(1u << 20) + 8192 + (((size_t)blockIdx.x * 64 + (j & 63)) * 16 + 3)] = __builtin_amdgcn_s_memrealtime();
#endif
                // the proposal, for the workgroups that form rows from it: element by element, each
                // with its check word
                double *rec = fv.prop + (((size_t)rown * kFmSlots + (m_next % kFmSlots)) * 2 + cand) * kFmWords;
                const unsigned long long tag = serial32 | (unsigned long long)(j + 1);
#pragma unroll
                for (int i = 0; i < 5; ++i) fm_put(rec + 2 * i, p[i], tag);
            }
            // the five parameters' walls and priors, behind the hand-over: their first reader is the accept test
#include "mbb_walker_penalties.inc"
            if (vrow == vsel && l16 == 0) {
                double *pr = prop + bj * kFmProp;
                pr[7] = pen_u;
                pr[8] = pen_g;
                lds_post(ctl + kFmPen + bj, j + 1);
            }
            FM_T(5);
        }
        FM_TOUT();
    }
#undef FM_EV
#undef FM_EVV
#undef FM_T
#undef FM_TD
#undef FM_TOUT
}
