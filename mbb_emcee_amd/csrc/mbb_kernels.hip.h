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

using namespace mbbd;

constexpr int kPolyBDoubles = (64 * 8 + 1) * 8;      // mbbh::kPolyBCount intervals x 8 coefficients
constexpr int kPolyCDoubles = (40 * 8 + 1) * 8;      // mbbh::kPolyCCount

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
    const double *poly_b;     // [kPolyBCount*8] piecewise polynomials of x/expm1(x) (mbb_host_tables.h)
    const double *poly_c;     // [kPolyCCount*8] ... of (1 - e^-y)/y
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
    const double *pars;       // [n*5]
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
    int persist;              // SMODE 3: half-steps in this launch (step = number of the first step)
    unsigned int *gbar;       // SMODE 3: eight arrival counters, 128 bytes apart, zero at launch
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
    const XchgArgs *xargs;
};

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
// per launch); 3 a whole run of half-steps in ONE launch (single GPU, single source, one
// walker per workgroup, every workgroup resident): the tables are staged once and the
// dependence between half-steps is carried by a fence-free hand-off inside the kernel
// instead of a kernel boundary -- the moved row goes out with write-through (sc1) stores
// from one lane, that lane then adds to its XCD's arrival counter, and the next half-step's
// prologue wave polls the eight counters and reads its rows with sc1 loads
// (MI355X_MICROARCH.md, "Valid forms"; tools/lat_grid_barrier.hip: 2.5 us per hand-off).
// Measured (tools/probe_persist.py): 21.2 us per step against 15.5 with one launch per
// half-step -- the body of a half-step is 7.6 us either way and a dependent kernel
// boundary costs this launch train far less than the hand-off -- so the host uses it only
// on request (option "persistent_sampler" 1); chains are bitwise the same.
template <bool OPTHIN, bool NOALPHA, int SMODE, bool STAGE>
__global__ void __launch_bounds__(1024) k_lnlike(const LikeArgs a)
{
    constexpr bool SAMPLER = SMODE != 0, XCHG = SMODE == 2, PERSIST = SMODE == 3;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ Exp2Entry s_tab[kExp2N];                     // 2^(j/256) for the sample loop
    __shared__ __align__(16) double s_pb[kPolyBDoubles];    // x/expm1(x), piecewise degree 7
    __shared__ __align__(16) double s_pc[OPTHIN ? 2 : kPolyCDoubles];   // (1 - e^-y)/y (thick only)
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
    const int w0 = blockIdx.x * W;
    // The compiler fetches kernel arguments where they are first used, one exposed
    // scalar-cache round trip (~200 cycles) each; on the latency path that is a
    // dozen of them.  Ask for the hot ones here so that they arrive in one batch.
#define PIN(x) asm volatile("" ::"s"(x))
    PIN(a.n); PIN(a.pars); PIN(a.nunorm); PIN(a.lnunorm); PIN(a.has_uplim); PIN(a.has_gprior);
    PIN(a.lowlim[0]); PIN(a.lowlim[1]); PIN(a.lowlim[2]); PIN(a.lowlim[3]); PIN(a.lowlim[4]);
    PIN(a.unit_tab); PIN(a.lnl); PIN(a.status); PIN(a.model_flux); PIN(a.invcov); PIN(a.nsrc);
    PIN(a.debug); PIN(a.flux); PIN(a.ivar); PIN(a.rows_per_src);
#undef PIN
#ifdef MBB_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();   // before the kernarg arrives
#define STAMP(i) do { if (tid == 0 && a.stamps && blockIdx.x < 65536) a.stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMPD(i, dep) do { if (tid == 0 && a.stamps && blockIdx.x < 65536) { asm volatile("" ::"v"(dep)); a.stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
    if (tid == 0 && a.stamps && blockIdx.x < 65536) a.stamps[blockIdx.x * 32 + 7] = t_entry;
#else
#define STAMP(i) do { } while (0)
#define STAMPD(i, dep) do { } while (0)
#endif
    STAMP(0);

    // A walker's prologue runs on one row of 16 lanes (mbb_device.hip.h, "rows"), so
    // the first ceil(16 W / 64) waves are prologue waves.  The other waves meanwhile
    // stage what the later phases need in LDS and pull their first segment's samples
    // and the index table into this CU's L1, so that nothing after the barrier waits
    // on L2; without spare waves every wave stages first.
    const int pwaves = min(nwave, (16 * W + 63) >> 6);
    if (wave >= pwaves || pwaves == nwave) {
        const int t0 = (pwaves == nwave) ? tid : tid - 64 * pwaves;
        const int nt = (pwaves == nwave) ? (int)blockDim.x : (int)blockDim.x - 64 * pwaves;
        for (int i = t0; i < kExp2N; i += nt) s_tab[i] = kExp2Tab[i];
        {
            const double2 *gb = reinterpret_cast<const double2 *>(a.poly_b);
            double2 *lb = reinterpret_cast<double2 *>(s_pb);
            for (int i = t0; i < kPolyBDoubles / 2; i += nt) lb[i] = gb[i];
            if (!OPTHIN) {
                const double2 *gc = reinterpret_cast<const double2 *>(a.poly_c);
                double2 *lc = reinterpret_cast<double2 *>(s_pc);
                for (int i = t0; i < kPolyCDoubles / 2; i += nt) lc[i] = gc[i];
            }
        }
        for (int b = t0; b < nb; b += nt) { s_flux[b] = a.flux[b]; s_ivar[b] = a.ivar[b]; }
        for (int b = t0; b < nb; b += nt) s_band[b] = a.band_rng[b];
        if (a.cov_in_lds)
            for (int i = t0; i < nb * nb; i += nt) s_invcov[i] = a.invcov[i];
        if (STAGE) {
            // passband tables -> LDS, 16 B per lane
            const int n2 = a.nchunk * 32;                      // double2 elements per array
            const double2 *g0 = reinterpret_cast<const double2 *>(a.nu);
            const double2 *g1 = reinterpret_cast<const double2 *>(a.lnnu);
            const double2 *g2 = reinterpret_cast<const double2 *>(a.wt);
            double2 *l0 = reinterpret_cast<double2 *>(s_nu);
            double2 *l1 = reinterpret_cast<double2 *>(s_lnnu);
            double2 *l2 = reinterpret_cast<double2 *>(s_wt);
            for (int i = t0; i < n2; i += nt) { l0[i] = g0[i]; l1[i] = g1[i]; l2[i] = g2[i]; }
        } else if (wave >= pwaves) {
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

    // PERSIST: a.persist half-steps in this launch; otherwise one pass with the launch's values
    unsigned xcc = 0;
    if (PERSIST) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7u; }
    const int niter = PERSIST ? a.persist : 1;
    for (int it = 0; it < niter; ++it) {
    const int L_step = PERSIST ? a.step + (it >> 1) : a.step, L_half = PERSIST ? (it & 1) : a.half;
    const int L_s_begin = PERSIST ? (L_half ? a.c_count : 0) : a.s_begin;
    const int L_c_begin = PERSIST ? (L_half ? 0 : a.c_count) : a.c_begin;
    const unsigned long long L_seed = PERSIST ? a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(it >> 1) : a.seed;
    double *const L_chain6 = (PERSIST && a.chain6) ? a.chain6 + (size_t)it * a.n * 6 : a.chain6;
    unsigned int *const L_nacc = PERSIST ? a.nacc + (size_t)L_half * a.n : a.nacc;

    // ---- phase 1: gate + prologue + parameter-only penalties, one row per walker
    // (the host guarantees blockDim.x >= 16 W)
    if (const int j = tid >> 4; j < W) {
        const bool lead = (tid & 15) == 0;                    // the lane that writes to LDS
        const int w = w0 + j;
        WalkerK k;
        k.status = ROW_SKIP;
        k.pad = 0;
        double pen_u = 0.0, pen_g = 0.0;
        if (w < a.n) {
            double p[5], lT, lL = 0.0;
            if (SAMPLER) {
                // stretch move (Goodman & Weare 2010; what emcee does per half-step,
                // mbb_fit.py:533/:542): z ~ g(z) on [1/a, a], partner from the other
                // half, proposal q = c - z (c - s)
                const int src = w / a.m_count, loc = w - src * a.m_count;
                const int row = src * a.nw_src + L_s_begin + loc;
                unsigned int c4[4] = {(unsigned int)row, (unsigned int)(2 * L_step + L_half), 0u, 0u};
                philox4x32(c4, (unsigned int)L_seed, (unsigned int)(L_seed >> 32));
                const double u1 = ((double)(c4[0] >> 5) * 67108864.0 + (double)(c4[1] >> 6)) *
                                  (1.0 / 9007199254740992.0);
                const double u2 = (double)c4[2] * (1.0 / 4294967296.0);
                const double u3 = ((double)c4[3] + 0.5) * (1.0 / 4294967296.0);
                const double sq = (a.stretch_a - 1.0) * u1 + 1.0;
                const double zz = sq * sq / a.stretch_a;
                int pj = (int)(u2 * (double)a.c_count);
                if (pj >= a.c_count) pj = a.c_count - 1;
                const double *srow = a.pos6 + (size_t)row * 6;
                const double *crow = a.pos6 + (size_t)(src * a.nw_src + L_c_begin + pj) * 6;
                if (PERSIST && it > 0) {
                    // every walker of the previous half-step must have stored its row: the eight
                    // arrival counters (one per XCD) add up to n per completed half-step.  Only
                    // this wave reads state rows, so no workgroup barrier is needed after the poll.
                    const unsigned target = (unsigned)it * (unsigned)a.n;
                    const int l = tid & 63;
                    long long spins = 0;
                    for (;;) {
                        unsigned v = 0;
                        if (l < 8) v = __hip_atomic_load(a.gbar + l * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
                        if (__builtin_amdgcn_ballot_w64(l == 0 && v < target) == 0) break;
                        if (++spins > (1ll << 22)) { atomicMax(a.errflag, 9); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
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
                    const double cv = xchg ? ld_sys(crow + i) : (PERSIST ? ld_dev(crow + i) : crow[i]);
                    const double sv = xchg ? ld_sys(srow + i) : (PERSIST ? ld_dev(srow + i) : srow[i]);
                    p[i] = cv - zz * (cv - sv);
                }
                srow5 = xchg ? ld_sys(srow + 5) : (PERSIST ? ld_dev(srow + 5) : srow[5]);
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
            bool ok = true;                                   // likelihood.py:643-670
#pragma unroll
            for (int i = 0; i < 5; ++i) ok = ok && !(p[i] < a.lowlim[i]);
            if (!ok) {
                k.status = ROW_BELOW_LOWLIM;
            } else if (!finite5(p)) {
                // NaN passes the reference's `<` gate and its SED arithmetic then
                // yields NaN; +-inf does the same.  Short-cut both to NaN.
                k.status = ROW_NONFINITE;
            } else {
                SedScalars s;
                k.status = sed_prologue<OPTHIN, NOALPHA, true>(p[0], p[1], p[3], p[4], lT, lL,
                                                               a.nunorm, a.lnunorm, s, &k.pad);
                STAMPD(9, s.normfac);
                if (k.status == ROW_OK) {
                    make_walker_k<OPTHIN, NOALPHA>(p[1], p[3], s, k);
                    if (a.has_uplim | a.has_gprior) {          // one test when there are none
                    // _uplim_prior, likelihood.py:672-717
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        if (((a.has_uplim >> i) & 1u) && p[i] > a.uplim[i]) {
                            double lw = 0.02 * (a.uplim[i] - a.lowlim[i]);
                            double d = p[i] - a.uplim[i];
                            pen_u -= 0.5 * d * d / (lw * lw);
                        }
                    double peak = 0.0;
                    if (((a.has_uplim | a.has_gprior) >> 5) & 1u) {
                        int pst;
                        peak = sed_peak_wave<OPTHIN, true>(p[0], p[1], k.lx0, s.hcokt, pst);
                        if (pst != ROW_OK) k.status = pst;
                        k.peak = peak;
                    }
                    if (((a.has_uplim >> 5) & 1u) && peak > a.uplim[5]) {  // :710-715
                        double lw = 0.02 * a.uplim[5], d = peak - a.uplim[5];
                        pen_u -= 0.5 * d * d / (lw * lw);
                    }
                    // _gprior, likelihood.py:719-752
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        if ((a.has_gprior >> i) & 1u) {
                            double d = p[i] - a.gmean[i];
                            pen_g -= 0.5 * a.givar[i] * d * d;
                        }
                    if ((a.has_gprior >> 5) & 1u) {
                        double d = peak - a.gmean[5];
                        pen_g -= 0.5 * a.givar[5] * d * d;
                    }
                    }
                }
            }
        }
        STAMPD(10, pen_u + pen_g);
        if (lead) {
            if (k.status == ROW_OK) wk[j] = k;
            else { wk[j].status = k.status; wk[j].pad = k.pad; }
            pen[2 * j] = pen_u;
            pen[2 * j + 1] = pen_g;
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    // ---- phase 2: passband quadrature (response.py:572-576) -----------------
    auto T_nu = [&](int i) { if constexpr (STAGE) return s_nu[i]; else return a.nu[i]; };
    auto T_ln = [&](int i) { if constexpr (STAGE) return s_lnnu[i]; else return a.lnnu[i]; };
    auto T_wt = [&](int i) { if constexpr (STAGE) return s_wt[i]; else return a.wt[i]; };
    const SampleTabs tabs = {s_tab, s_pb, s_pc};
    auto do_unit = [&](const WalkerK &k, const int j, const int4 us) {
        const int s = us.x, c0 = us.y, c1 = us.z;
        double acc = 0.0;
        int c = c0;
        for (; c + 2 <= c1; c += 2) {          // two chunks per step
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
            acc = wave_sum(acc);
            if (lane == 0) partial[j * npart + s] = acc;
        } else if (us.w == 2) {                               // four band leftovers, one per row
            acc = row_sum(acc);
            if ((lane & 15) == 0) {
                const int sl = a.tail_slot[4 * s + (lane >> 4)];
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
            for (int uu = 0; uu < nun; ++uu) do_unit(k, j, a.unit_tab[uu]);
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
            do_unit(k, j, us);
        }
    }
    STAMP(3);

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
        cbb_first = wk[wave].cbb;
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
            const double cbb = FIRST ? cbb_first : wk[j].cbb;  // normfac: the samples were summed without it
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
        if (lane == 0) {
            double r;
            if (st == ROW_BELOW_LOWLIM) r = -__builtin_inf();
            else if (st != ROW_OK) r = __builtin_nan("");
            else {
                r = -0.5 * acc;
                r += pen_u;                                    // :828
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
                } else if (PERSIST) {
                    // the row goes out write-through, the stores are waited for, then this
                    // walker is counted on its XCD's counter: the next half-step starts from that
                    if (accept) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) st_dev(srow + i, q[i]);
                        st_dev(srow + 5, r);
                        atomicAdd(&L_nacc[w], 1u);
                    }
                    if (L_chain6) {
                        double *crow = L_chain6 + (size_t)w * 6;
#pragma unroll
                        for (int i = 0; i < 5; ++i) crow[i] = accept ? q[i] : ld_dev(srow + i);
                        crow[5] = accept ? r : q[6];
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_fetch_add(a.gbar + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    };
    if (wave < W) epilogue(wave, std::true_type{});
    if (W > nwave)
        for (int j = wave + nwave; j < W; j += nwave) epilogue(j, std::false_type{});
    }   // half-steps of a persistent run
    STAMP(6);
}

// Sharded run, one-hop exchange: holds the stream until every peer has posted launch
// number `seq` (its moved rows are then all in this rank's copy of the ensemble).  One wave.
__global__ void k_xchg_wait(const unsigned long long *mine, int xn, int xrank, unsigned long long seq,
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
    __shared__ Exp2Entry s_tab[kExp2N];
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
            const double xs[2] = {k.hokt9 * n0, k.hokt9 * n1};
            const double lxs[2] = {k.lhokt9 + l0, k.lhokt9 + l1};
            const bool wien0 = !NOALPHA && xs[0] > k.xmerge, wien1 = !NOALPHA && xs[1] > k.xmerge;
            const bool far = !(xs[0] <= 64.0) || !(xs[1] <= 64.0);
            double fs[2];
            if (__builtin_amdgcn_ballot_w64(wien0 || wien1 || far) == 0) {
                fnu_bb_tab_n<OPTHIN, 2>(k, xs, lxs, &tabs, fs);
            } else if (!NOALPHA && __builtin_amdgcn_ballot_w64(!(wien0 && wien1)) == 0) {
                fnu_wien_tab_n<2>(k, lxs, &tabs, fs);
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
                const double mid = a + (pn + 0.5) * pw;
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

