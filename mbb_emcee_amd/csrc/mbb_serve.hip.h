// mbb_serve.hip.h -- k_serve: the likelihood of given rows (k_lnlike SMODE 0, one walker per workgroup) as a kernel
// that STAYS on the GPU between the calls of a host-driven sampler and is rung through the PCIe BAR.
// Included by mbb_flow.hip only.
//
// What a boundary call (likelihood.__call__ on host arrays: SURVEY.md 8d metric M1, the lnprob callable emcee calls
// once per half-step, mbb_fit.py:80-81) costs beside its kernel is the launch: ~2.5 us in the runtime's launch call,
// ~3 us from the doorbell to the first wave, the tables into LDS again -- 14 us per 125-row call of which the kernel
// is under 7 (profiles/r04/boundary_breakdown.txt).  A sampler's calls come one after another with nothing but host
// work in between, so after a few of them in a row the host starts THIS kernel instead: a workgroup per row of the widest call (each on a CU
// of its own; mbb_hip.hip serve_grid), tables
// staged once, and then per request
//   host:   rows -> the parameter block in device memory (through the BAR), sfence, the request word -> the doorbell
//           (same allocation, same path: PCIe keeps posted writes in order), watches the result slots in pinned memory
//   kernel: thread 0 of every workgroup polls the doorbell (fine-grained memory: not cached); a new request number
//           -> workgroup b < n evaluates row b exactly as k_lnlike does (same constructor text, same units in the same
//           order, same order of the band sums: bitwise the launch's results) and writes lnl / status into the pinned
//           result slots, which the host sees turn.
// A row's evaluation is not the launch's three phases one after the other (constructor, quadrature, band sums): what the
// quadrature needs of the constructor for a sample on the blackbody side of the merge point -- h/kT, its log, beta, log x0 --
// and for one on the power-law side -- alpha, h/kT again: x^-alpha, without kappa -- is a division and two logs away from the
// parameters, and what takes the constructor its time, the merge point itself (a root), only says WHICH side a sample is on.
// So once those few scalars are in LDS every wave but the first works out BOTH candidates of every sample of ITS unit (unit
// u is wave u + 1's) and keeps them in registers while the first wave's row of 16 lanes finishes the constructor; behind it
// a unit is, per chunk, a comparison, a selection, the product with kappa and the fma with the weight -- sample by sample in
// the launch's order, each value the very one fnu_sample would have formed, so that every partial sum is bit for bit the
// launch's -- and its reduction (spec_cfg bit 0; units beyond the waves' number are evaluated behind the constructor, as in
// k_lnlike; without the bit -- few samples, a narrow workgroup -- the phases run one after the other).
// The kernel leaves when told to (the doorbell says QUIT: any other use of the context, its destruction) or when
// workgroup 0 has seen no request for `idle` microseconds (it then writes QUIT itself so that every workgroup
// follows); every workgroup besides has a safety limit of its own (four times that).  Whatever goes wrong -- a request
// written while the kernel was leaving, a workgroup that was not resident -- shows on the host as result slots that do
// not turn within its budget: it then says QUIT, waits for the stream, and evaluates the rows by a launch.
#pragma once
#include "mbb_kernels.hip.h"

constexpr unsigned long long kServeQuit = 0xffffull;           // the request's row count that means "leave"

// dynamic LDS of a k_serve launch besides the staged passband tables (bytes)
__host__ __device__ constexpr size_t serve_lds(size_t nb, size_t npart, bool cov_in_lds)
{
    return sizeof(WalkerK) + 8 * npart + 8 * nb + 16 + 16 * nb + (cov_in_lds ? 8 * nb * nb : 0) + 8 * (nb + 2) + 64;
}

// Arguments (LikeArgs fields of variants that never meet share storage): pars = the parameter block, lnl = the pinned
// result records [row]{lnl, status as a 64-bit integer}, pos6 = the doorbell (one 8-byte word: request number << 16 | rows), seed = the request the
// launch itself carries (served at once), persist = idle limit in microseconds, chain6 = the host's "gone" word (pinned).
template <bool OPTHIN, bool NOALPHA, bool STAGE, bool OVL>
__global__ void __launch_bounds__(1024) k_serve(const LikeArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ __align__(16) double s_tab[kExp2N];
    __shared__ __align__(16) double s_pb[kPolyBDoubles];
    __shared__ __align__(16) double s_pc[OPTHIN ? 2 : kPolyCDoubles];
    __shared__ unsigned long long s_req[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwave = blockDim.x >> 6;
    const int nun = a.nunit, npart = a.npart, nb = a.nb;
    WalkerK *wk = reinterpret_cast<WalkerK *>(smem_raw);
    double *partial = reinterpret_cast<double *>(wk + 1);                 // [npart]
    double *mflux = partial + npart;                                      // [nb]
    double *pen = mflux + nb;                                             // [2]
    double *s_flux = pen + 2;
    double *s_ivar = s_flux + nb;
    double *s_invcov = s_ivar + nb;
    int2 *s_band = reinterpret_cast<int2 *>(s_invcov + (a.cov_in_lds ? (size_t)nb * nb : 0));
    const size_t tab_off = ((size_t)(reinterpret_cast<unsigned char *>(s_band + nb + 1) - smem_raw) + 15) & ~(size_t)15;
    double *s_nu = reinterpret_cast<double *>(smem_raw + tab_off);
    double *s_lnnu = s_nu + (STAGE ? a.nchunk * 64 : 0);
    double *s_wt = s_lnnu + (STAGE ? a.nchunk * 64 : 0);
    const unsigned long long *door = reinterpret_cast<const unsigned long long *>(a.pos6);
    // OVL: the quadrature starts beside the constructor (see above) -- a template parameter, so that each instantiation holds
    // one of the two ways only: with both in one kernel the constructor's chain was compiled with spills to scratch
    constexpr bool overlap = OVL;
    __shared__ WalkerK kfin;
    __shared__ double s_row[5];          // `overlap`: the row's five values for the wave that works out the penalties

    // ---- once: the tables and the data to LDS.  By every wave but the first, which goes straight for its row: the first
    // request is in the launch and the constructor needs none of this -- the first
    // barrier every wave passes (behind the scalars with OVL, behind the constructor without) is also the one behind the
    // staging.  (A workgroup narrowed to one wave stages as it did.)
    if (wave > 0 || nwave == 1) {
        const int nt = (int)blockDim.x - (nwave > 1 ? 64 : 0), tid = (int)threadIdx.x - (nwave > 1 ? 64 : 0);
        const double2 *gb = reinterpret_cast<const double2 *>(a.poly_b);
        const double2 *gc = reinterpret_cast<const double2 *>(a.poly_c);
        double2 *lb = reinterpret_cast<double2 *>(s_pb);
        double2 *lc = reinterpret_cast<double2 *>(s_pc);
        for (int i = tid; i < kExp2N; i += nt) s_tab[i] = kExp2Tab[i];
        for (int i = tid; i < kPolyBDoubles / 2; i += nt) lb[i] = gb[i];
        if (!OPTHIN)
            for (int i = tid; i < kPolyCDoubles / 2; i += nt) lc[i] = gc[i];
        for (int b = tid; b < nb; b += nt) { s_flux[b] = a.flux[b]; s_ivar[b] = a.ivar[b]; s_band[b] = a.band_rng[b]; }
        if (a.cov_in_lds)
            for (int i = tid; i < nb * nb; i += nt) s_invcov[i] = a.invcov[i];
        if (STAGE) {
            const int n2 = a.nchunk * 32;
            const double2 *g0 = reinterpret_cast<const double2 *>(a.nu), *g1 = reinterpret_cast<const double2 *>(a.lnnu),
                          *g2 = reinterpret_cast<const double2 *>(a.wt);
            double2 *l0 = reinterpret_cast<double2 *>(s_nu), *l1 = reinterpret_cast<double2 *>(s_lnnu),
                    *l2 = reinterpret_cast<double2 *>(s_wt);
            for (int i = tid; i < n2; i += nt) { l0[i] = g0[i]; l1[i] = g1[i]; l2[i] = g2[i]; }
        }
    }
    auto T_nu = [&](int i) { if constexpr (STAGE) return s_nu[i]; else return a.nu[i]; };
    auto T_ln = [&](int i) { if constexpr (STAGE) return s_lnnu[i]; else return a.lnnu[i]; };
    auto T_wt = [&](int i) { if constexpr (STAGE) return s_wt[i]; else return a.wt[i]; };
    const SampleTabs tabs = {s_tab, s_pb, s_pc};
    // (with `overlap` unit u is wave u + 1's: wave 0 has the constructor)
    int4 us_first = make_int4(0, 0, 0, 0);
    {
        const int u = overlap ? wave - 1 : wave;
        if (u >= 0 && u < nun) us_first = a.unit_tab[u];
    }
    if (nwave == 1) __syncthreads();

    // diagnostic build: when things happened in a workgroup's latest request, on the clock all CUs share (100 MHz):
    // stamps[workgroup * 16 + event], tools/probe_serve_stamps.py
#ifdef MBB_STAMPS
// (MBB_STAMPS_MIN: only "request seen" and "result stored" -- every stamp is a scalar memory read, a wait and a store ON
// the path it times, ~0.1-0.3 us each: the seven together make a request a microsecond longer than it is)
#ifdef MBB_STAMPS_MIN
#define SV_EV(ev) do { if ((ev) == 0 || (ev) == 6) { if (tid == 0 && a.stamps) a.stamps[(size_t)blockIdx.x * 16 + (ev)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define SV_EV(ev) do { if (tid == 0 && a.stamps) a.stamps[(size_t)blockIdx.x * 16 + (ev)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
// ... and per wave (workgroups 0..63): stamps[4096 + workgroup * 64 + wave * 4 + k]
#ifdef MBB_STAMPS_MIN
#define SV_WV(k) do { } while (0)
#else
#define SV_WV(k) do { if (lane == 0 && a.stamps && blockIdx.x < 64) a.stamps[4096 + (size_t)blockIdx.x * 64 + wave * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#else
#define SV_EV(ev) do { } while (0)
#define SV_WV(k) do { } while (0)
#endif
    unsigned long long cur = a.seed;                 // the request the launch carries
    const long long idle = a.persist > 0 ? (long long)a.persist : 1000;
    for (int turn = 0;; ++turn) {
        const int n = (int)(cur & 0xffffull);
        // (a server narrower than the request has rows -- its process's share of the device, mbb_hip.hip -- takes its rows in
        // turns: workgroup b rows b, b + workgroups, ...)
        for (int w = (int)blockIdx.x; w < n; w += (int)gridDim.x) {
            // ---- phase 1 (k_lnlike's, SMODE 0): gate, constructor, parameter-only penalties on one row of 16 lanes --
            // in two parts when the quadrature may start on the first
            SV_EV(0);
            bool mine = false;
            double p[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, lT = 0.0, lL = 0.0;
            int st0 = ROW_OK;
            if (tid < 16) {
                mine = true;
                // the row's five values in ONE request (lane i takes element i; the block is fine-grained memory the
                // host has just written through the BAR: not cached, every request goes to memory -- five requests
                // per workgroup, one per element with all lanes on the same address, cost 125 rows 8 us)
                const double pe = tid < 5 ? __hip_atomic_load(a.pars + (size_t)w * 5 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
#pragma unroll
                for (int i = 0; i < 5; ++i) p[i] = __shfl(pe, i);
#ifdef MBB_STAMPS
                asm volatile("" ::"v"(p[0]));
#endif
                SV_EV(1);
                if (OPTHIN) {
                    double lo[1];
                    vlog<true>(lo, p[0]);
                    lT = lo[0];
                } else {
                    double lo[2];
                    vlog<true>(lo, p[0], p[2]);
                    lT = lo[0]; lL = lo[1];
                }
                if constexpr (overlap) {
                    // the gate (likelihood.py:643-670; NaN and +-inf: mbb_walker_consts.inc), once, here
                    bool okl = true;
#pragma unroll
                    for (int i = 0; i < 5; ++i) okl = okl && !(p[i] < a.lowlim[i]);
                    st0 = !okl ? (int)ROW_BELOW_LOWLIM : (!finite5(p) ? (int)ROW_NONFINITE : (int)ROW_OK);
                }
                if (overlap && tid == 0) {
                    // what a sample needs on either side of the merge point, exactly as sed_prologue / make_walker_k form it
                    // (the constructor below forms them again, into a record of its own)
                    const bool ok = st0 == ROW_OK && !(!NOALPHA && p[3] <= 0.0) && !(p[1] < 0.0);
#pragma unroll
                    for (int i = 0; i < 5; ++i) s_row[i] = p[i];
                    WalkerK k0;
                    k0.hokt9 = m_div(1e9 * kH / kK, p[0]);
                    k0.lhokt9 = kLog1e9HoK - lT;
                    k0.beta = p[1]; k0.bp3 = p[1] + 3.0; k0.cq = 0.0;
                    k0.alpha = NOALPHA ? 0.0 : p[3];
                    k0.lx0 = OPTHIN ? 0.0 : k0.lhokt9 + kLogUmToGHz - lL;
                    k0.xmerge = __builtin_inf();
                    k0.cbb = k0.cpl = k0.kap = k0.peak = 0.0;
                    k0.status = ok ? ROW_OK : ROW_SKIP;
                    k0.pad = 0;
                    wk[0] = k0;
                }
            }
            if (overlap) __syncthreads();
            SV_EV(2);
            const bool ahead = overlap && wk[0].status == ROW_OK;           // workgroup-uniform
            // One unit's sum from samples evaluated now, behind the constructor (k_lnlike's do_unit): two chunks per step.
            auto classic_unit = [&](const int4 us, const WalkerK &k) {
                const int s = us.x, c0 = us.y, c1 = us.z;
                double acc = 0.0;
                int c = c0;
                for (; c + 2 <= c1; c += 2) {
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
                return acc;
            };
            auto store_unit = [&](const int4 us, double acc) {
                const int s = us.x;
                if (us.w == 0) {
                    acc = wave_sum_l63(acc);           // (the total is lane 63's)
                    if (lane == 63) partial[s] = acc;
                } else if (us.w == 2) {
                    acc = row_sum(acc);
                    if ((lane & 15) == 0) {
                        const int sl = a.tail_slot[4 * s + (lane >> 4)];
                        if (sl >= 0) partial[sl] = acc;
                    }
                } else {
                    partial[s + lane] = acc;
                }
            };
            // Two paths that never meet inside a wave, so that neither's registers are live in the other (the candidates kept
            // ahead are 16 registers; the constructor's chain wants most of the file): with `overlap` every wave but the first
            // takes the first path whole -- barriers included: every wave passes two, whichever way it goes.
            if (overlap && wave > 0) {
                // The penalties that depend on the five parameters alone (soft upper walls, Gaussian priors: likelihood.py:672-752)
                // are no part of the constructor's chain: the last wave -- it has the fewest units, if any -- works them out
                // meanwhile, term by term as mbb_walker_consts.inc does (on the constructor's row of lanes they cost it 600 to
                // 1700 cycles of comparisons against the argument block and of branches: tools/lat_ctor.hip); the peak
                // wavelength's two terms, which follow them, are added in phase 3.
                if (ahead && wave == nwave - 1 && lane == 0) {
                    double pen_u = 0.0, pen_g = 0.0;
                    if (a.has_uplim | a.has_gprior) {
                        double q[5];
#pragma unroll
                        for (int i = 0; i < 5; ++i) q[i] = s_row[i];
#pragma unroll
                        for (int i = 0; i < 5; ++i)
                            if (((a.has_uplim >> i) & 1u) && q[i] > a.uplim[i]) {
                                double lw = 0.02 * (a.uplim[i] - a.lowlim[i]);
                                double d = q[i] - a.uplim[i];
                                pen_u -= 0.5 * d * d / (lw * lw);
                            }
#pragma unroll
                        for (int i = 0; i < 5; ++i)
                            if ((a.has_gprior >> i) & 1u) {
                                double d = q[i] - a.gmean[i];
                                pen_g = fma(-0.5 * a.givar[i] * d, d, pen_g);
                            }
                    }
                    pen[0] = pen_u;
                    pen[1] = pen_g;
                }
                double rg_f[4], rg_w[4];             // (h/kT nu and the weight are read again behind the barrier: registers)
                SV_WV(0);
                if (ahead) {
                    const WalkerK k = wk[0];
                    const int c0 = us_first.y, c1 = us_first.z;          // (no unit: c0 == c1 == 0)
                    // The merge point lies between x = 2 + alpha and x = 3 + alpha + beta (thick: thick_merge_root's bracket;
                    // thin: a - 1 < xmerge < a with a = 3 + alpha + beta).  A sample at or below the first is never on the
                    // power-law side, one beyond the second always: each gets the one candidate it can need.  (A chunk is
                    // 64 neighbouring frequencies of one band: the lanes mostly agree.)
                    const double xlo = 2.0 + k.alpha, xhi = 3.0 + k.alpha + k.beta;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // (one sample at a time, in this order: two at once, as classic_unit takes them, want more registers
                        // than there are beside the candidates kept)
                        MBB_FENCE();
                        if (c0 + j < c1) {
                            const int i = (c0 + j) * 64 + lane;
                            const double nu = T_nu(i), ln = T_ln(i);
                            const double x = k.hokt9 * nu;
                            rg_f[j] = 0.0;
                            if (NOALPHA || !(x > xhi)) rg_f[j] = fnu_sample<OPTHIN, NOALPHA, true, false>(k, nu, ln, &tabs);
                            if constexpr (!NOALPHA) {
                                rg_w[j] = 0.0;
                                if (x > xlo) rg_w[j] = wien_pow_tab(k, ln, &tabs);
                            }
                        }
                    }
                }
                SV_WV(1);
                __syncthreads();
                SV_WV(2);
                // ---- phase 2: a unit is a comparison, a selection and an fma per chunk, then its reduction
                // (kfin.status is ROW_OK only for a row the scalars were made for: `ahead` holds)
                if (kfin.status == ROW_OK) {
                    const int c0 = us_first.y, c1 = us_first.z;
                    if (c0 < c1) {
                        // (everything asked of LDS at once -- frequencies and weights of all four chunks, clamped to the unit's
                        // last one where it has fewer, and the record's three values: one wait, not five in a row)
                        double nx[4], qw[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = min(c0 + j, c1 - 1) * 64 + lane;
                            nx[j] = T_nu(i);
                            qw[j] = T_wt(i);
                        }
                        const double xm = kfin.xmerge, kap = kfin.kap, hk = kfin.hokt9;
                        double acc = 0.0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (c0 + j < c1) {
                                double f = rg_f[j];
                                if constexpr (!NOALPHA) {
                                    const double pl = kap * rg_w[j];                 // fnu_wien_tab's value
                                    f = (hk * nx[j] > xm) ? pl : f;                  // (fnu_sample's comparison)
                                }
                                acc = fma(f, qw[j], acc);
                            }
                        store_unit(us_first, acc);
                    }
                    // (more units than waves to keep them: the rest as a launch does them)
                    if (wave - 1 + (nwave - 1) < nun) {
                        const WalkerK k = kfin;
                        for (int u = wave - 1 + (nwave - 1); u < nun; u += nwave - 1) {
                            const int4 us = a.unit_tab[u];
                            store_unit(us, classic_unit(us, k));
                        }
                    }
                }
                SV_WV(3);
                __syncthreads();
            } else {
                if (mine) {
                    if (ahead) __builtin_amdgcn_s_setprio(3);
                    WalkerK k;
                    k.status = ROW_SKIP;
                    k.pad = 0;
                    double pen_u = 0.0, pen_g = 0.0;
                    if constexpr (overlap) {
                        // the constructor and nothing else: the gate was taken before the barrier, the parameter-only
                        // penalties are another wave's (above), the peak wavelength's terms are added in phase 3
                        k.status = st0;
                        k.peak = 0.0;
                        if (st0 == ROW_OK) {
                            SedScalars s;
                            k.status = sed_prologue<OPTHIN, NOALPHA, true>(p[0], p[1], p[3], p[4], lT, lL, a.nunorm, a.lnunorm, s, &k.pad);
                            if (k.status == ROW_OK) {
                                make_walker_k<OPTHIN, NOALPHA>(p[1], p[3], s, k);
                                if (((a.has_uplim | a.has_gprior) >> 5) & 1u) {
                                    int pst;
                                    const double peak = sed_peak_wave<OPTHIN, true>(p[0], p[1], k.lx0, s.hcokt, pst);
                                    if (pst != ROW_OK) k.status = pst;
                                    k.peak = peak;
                                }
                            }
                        }
                        if (tid == 0) kfin = k;
                    } else {
#if defined(MBB_STAMPS) && defined(MBB_STAMPS_FINE)
                        if (tid == 0 && a.stamps) a.stamps[blockIdx.x * 32 + 8] = __builtin_amdgcn_s_memtime();      // (one row: no clash with SV_EV's)
#endif
#include "mbb_walker_consts.inc"
#if defined(MBB_STAMPS) && defined(MBB_STAMPS_FINE)
                        if (tid == 0 && a.stamps) { asm volatile("" ::"v"(pen_u + pen_g + k.cbb)); a.stamps[blockIdx.x * 32 + 10] = __builtin_amdgcn_s_memtime(); }
#endif
                        if (tid == 0) {
                            kfin = k;
                            pen[0] = pen_u;
                            pen[1] = pen_g;
                        }
                    }
                    SV_EV(3);
                    if (ahead) __builtin_amdgcn_s_setprio(0);
                }
                __syncthreads();
                // ---- phase 2 (k_lnlike's): the walker's units dealt to the waves
                SV_EV(4);
                if (!overlap && kfin.status == ROW_OK) {
                    const WalkerK k = kfin;
                    for (int u = wave; u < nun; u += nwave) {
                        const int4 us = (u == wave) ? us_first : a.unit_tab[u];
                        store_unit(us, classic_unit(us, k));
                    }
                }
                __syncthreads();
            }
            SV_EV(5);
            // ---- phase 3 (k_lnlike's): band sums in fixed order, lnL -> the pinned result slots
            if (wave == 0) {
                const int st = kfin.status;
                double acc = 0.0;
                if (st == ROW_OK) {
                    const double cbb = kfin.cq;
                    auto band = [&](const int b) {
                        double sum = 0.0;
                        const int2 rng = s_band[b];
                        for (int sg = rng.x; sg < rng.y; sg += 4) {
                            const int le = rng.y - 1;
                            const double q0 = partial[sg], q1 = partial[min(sg + 1, le)], q2 = partial[min(sg + 2, le)], q3 = partial[min(sg + 3, le)];
                            sum += q0;
                            if (sg + 1 < rng.y) sum += q1;
                            if (sg + 2 < rng.y) sum += q2;
                            if (sg + 3 < rng.y) sum += q3;
                        }
                        sum *= cbb;
                        const double d = s_flux[b] - sum;              // likelihood.py:821
                        if (a.invcov) mflux[b] = d;
                        else acc = fma(d * d, s_ivar[b], acc);         // :825
                    };
                    for (int b = lane; b < nb; b += 64) band(b);
                    if (a.invcov) {                                    // :823
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        for (int i = lane; i < nb; i += 64) {
                            double t = 0.0;
                            const double *crow = (a.cov_in_lds ? s_invcov : a.invcov) + (size_t)i * nb;
                            for (int jj = 0; jj < nb; ++jj) t = fma(crow[jj], mflux[jj], t);
                            acc = fma(mflux[i], t, acc);
                        }
                    }
                    acc = (nb <= 16) ? wave_sum_row0(acc) : wave_sum(acc);
                }
                if (lane == 0) {
                    double r;
                    if (st == ROW_BELOW_LOWLIM) r = -__builtin_inf();
                    else if (st != ROW_OK) r = __builtin_nan("");
                    else {
                        double pen_u = pen[0], pen_g = pen[1];
                        if (overlap && (((a.has_uplim | a.has_gprior) >> 5) & 1u)) {
                            // the peak wavelength's wall and prior, behind the five parameters' as in mbb_walker_consts.inc
                            const double peak = kfin.peak;
                            if (((a.has_uplim >> 5) & 1u) && peak > a.uplim[5]) {  // likelihood.py:710-715
                                double lw = 0.02 * a.uplim[5], d = peak - a.uplim[5];
                                pen_u -= 0.5 * d * d / (lw * lw);
                            }
                            if ((a.has_gprior >> 5) & 1u) {                        // :748-750
                                double d = peak - a.gmean[5];
                                pen_g = fma(-0.5 * a.givar[5] * d, d, pen_g);
                            }
                        }
                        r = fma(-0.5, acc, pen_u);                     // :828
                        if (a.has_gprior) r += pen_g;                  // :830-831
                    }
                    // lnl and status as ONE 16-byte store into the row's result record (pinned host memory): every
                    // store is a PCIe write of its own and they leave one after the other -- two per row cost 125 rows
                    // ~2 us more than one.  The host takes a row when its status word has turned.  A record has a cache
                    // line of its own (kSrvStride): four to a line, each write was a read-modify-write of a line the host
                    // was polling -- 125 rows 9.2 -> 8.3 us (tools/ab_m1.py); the whole line in one 64-byte write: 8.5.
                    // (system scope -- straight through to the host -- spelled out: there is no 16-byte atomic store to
                    // ask the compiler for, and a non-temporal one may stay in L2 until the kernel ends)
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    const long long rb = __double_as_longlong(r), sb = (long long)(a.debug ? (st | (kfin.pad << 8)) : st);
                    v4i rec;
                    rec.x = (int)rb; rec.y = (int)(rb >> 32); rec.z = (int)sb; rec.w = (int)(sb >> 32);
                    double *dst = a.lnl + kSrvStride * (size_t)w;
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(rec) : "memory");
                    SV_EV(6);
                }
            }
            // (another row of this request: the first wave is through with this row's sums and penalties before any wave writes
            // the next row's)
            if (w + (int)gridDim.x < n) __syncthreads();
        }
        // ---- the next request: thread 0 watches the doorbell, the workgroup follows it
        if (tid == 0) {
            unsigned long long v = cur;
            // (idle time by the clock all CUs share, 100 MHz -- not by counting polls: a poll takes 0.3-1 us depending on who
            // else is polling, and a server that leaves after 0.35 ms instead of the millisecond asked for does not outlast a
            // sibling's start)
            const unsigned long long limit = 100ull * (unsigned long long)((blockIdx.x == 0) ? idle : 4 * idle + 64);
            const unsigned long long t_idle = __builtin_amdgcn_s_memrealtime();
            unsigned polls = 0;
            for (;;) {
                v = __hip_atomic_load(door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v != cur) break;
                if ((++polls & 7u) == 0u && __builtin_amdgcn_s_memrealtime() - t_idle > limit) {
                    v = (cur & ~0xffffull) | kServeQuit;
                    // workgroup 0 has seen nothing for `idle` microseconds: it says so where everybody looks
                    if (blockIdx.x == 0)
                        __hip_atomic_store(const_cast<unsigned long long *>(door), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            s_req[turn & 1] = v;
#if defined(MBB_STAMPS) && !defined(MBB_STAMPS_MIN)
            if (a.stamps) a.stamps[(size_t)blockIdx.x * 16 + 7] = __builtin_amdgcn_s_memrealtime();     // (the word was seen)
#endif
        }
        __syncthreads();
        cur = s_req[turn & 1];
        if ((cur & 0xffffull) == kServeQuit) break;
    }
    // gone: the host may look here instead of asking the runtime
    if (blockIdx.x == 0 && tid == 0 && a.chain6)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.chain6), cur >> 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
