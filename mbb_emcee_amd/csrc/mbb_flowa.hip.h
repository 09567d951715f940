// mbb_flowa.hip.h -- k_flowa, sampler form 9: the resident run for ensembles of any size up to 8 walkers per CU and
// half, with the SED constructor running one half-step ahead (single GPU, single ensemble).  Included by mbb_flow.hip.
//
// Round 4's k_flowr (form 8, removed in round 5) made the launch train resident: workgroup g owns walkers [g W, g W + W) of both halves and does a
// half-step as k_lnlike's launch does it -- constructor, then quadrature, then accept test, one after the other: 6.3 us
// per half-step with one walker per workgroup, 11.5 with four, of which the constructor is 3.6 whatever W is (the
// walkers' rows of 16 lanes run it in lock-step).  Here the constructor of half-step j + 1 runs WHILE half-step j is in
// the quadrature, on waves of its own.  What it needs and does not have yet is the partner's row after ITS pending
// move (half-step j): but that row is one of two known things -- the partner's row as it is (the move is rejected) or
// the proposal the partner is being tested on (accepted), which the partner's owner publishes the moment it is chosen.
// So every walker's proposal is constructed for both candidates (two rows of 16 lanes per walker, the same lock-step),
// and when the partner's decision arrives only a selection is left: that candidate's constants go to LDS for the
// quadrature.  Same draws, same proposal arithmetic (one rounding per coordinate wherever it is formed), same
// constructor text, same units in the same order, same order of the band sums: the chain is bitwise the launch
// train's.  The idea is form 5's (k_lnlike SMODE 5, one walker per workgroup, the constructor on workgroups of
// their own); the state and its check words are form 7's (FlowMView).  Twice the constructor work, the quadrature --
// what bounds a half-step from two walkers per CU on -- once.
//
// Waves of a workgroup (16):
//   C  ceil(2 W / 4) waves (numbers 3, 7, 11, 15: one SIMD, away from the quadrature's bursts): row r of 16 lanes of C
//      wave i is (walker (4 i + r) / 2, candidate (4 i + r) % 2).  Per half-step j: the draw; one polling loop for
//      the partner's row before its pending move (candidate 0) or its published proposal (candidate 1), an element
//      per lane with its check word; proposal, constructor, penalties; then the partner's decision of half-step
//      j - 1 selects a row: its constants and proposal record go to LDS, its proposal to the run's state for the
//      workgroups whose walkers will draw this one as partner in j + 1.
//   Q  the others, numbered through: the (walker, unit) pairs of k_lnlike's phase 2, a walker's as soon as its record is
//      there; then Q wave q < W forms walker q's band sums and lnL, does the accept test and publishes: the decision word
//      first, then the row (each element with its check word), chain entry, count; the owned rows stay in LDS.
// Hand-over inside the workgroup is by words in LDS (a wave's LDS operations execute in issue order): `ready` (one per
// walker: its record of half-step j is in LDS -- written the moment its partner's decision has selected the candidate),
// `qdone` (Q waves through with the half-step's units), `edone` (walkers decided).  Across workgroups: check words,
// decision words, and the lag guard (form 7's completion counters) --
// a C wave enters half-step j only when every C wave of the grid is through with j - kFmLag, which keeps the four
// slots of rows, proposals and decision words from being rewritten under a reader (tests/_flowa_model.py restates
// the protocol on the host).  Every wait is bounded and watches the run's error flag.
#pragma once
#include "mbb_flowm.hip.h"

constexpr int kFaMaxW = 8;      // walkers per workgroup and half
constexpr int kFaNB = 2;        // hand-over record buffers in LDS: half-step j uses buffer j mod kFaNB
constexpr int kFaRec = 10;      // doubles per proposal record besides WalkerK: proposal 0..4, (dim-1) ln z, ln u, the two penalties

// dynamic LDS of a k_flowa launch besides the staged passband tables (bytes)
__host__ __device__ constexpr size_t flowa_lds(size_t nb, size_t npart, bool cov_in_lds, size_t W)
{
    return kFaNB * W * (sizeof(WalkerK) + 8 * npart + 8 * kFaRec) + 8 * W * nb + 8 * 2 * W * 8 + 16 * nb +
           (cov_in_lds ? 8 * nb * nb : 0) + 8 * (nb + 2) + 192 + 64;
}

template <bool OPTHIN, bool NOALPHA, bool STAGE>
__global__ void __launch_bounds__(1024) k_flowa(const LikeArgs a_val)
{
    CLikeArgs *const ka = MBB_KERNARGS();
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ __align__(16) double s_tab[kExp2N];
    __shared__ __align__(16) double s_pb[kPolyBDoubles];
    __shared__ __align__(16) double s_pc[OPTHIN ? 2 : kPolyCDoubles];
    const int tid = threadIdx.x, lane = tid & 63, wave = MBB_WAVE_ID(tid);
    // What every role needs of the launch (declared inside the role, after its own argument pointer)
#define MBB_FA_COMMON() \
    const int W = a.wpb; \
    const int nun = a.nunit, npart = a.npart, nb = a.nb; \
    const int nC = (2 * W + 3) >> 2, nQ = 16 - nC; \
    WalkerK *wk0 = reinterpret_cast<WalkerK *>(smem_raw);                         /* [kFaNB][W] */ \
    double *partial0 = reinterpret_cast<double *>(wk0 + kFaNB * W);              /* [kFaNB][W * npart] */ \
    double *rec0 = partial0 + kFaNB * (size_t)W * npart;                         /* [kFaNB][W * kFaRec] */ \
    double *mflux = rec0 + kFaNB * (size_t)W * kFaRec;                           /* [W * nb] */ \
    double *own = mflux + (size_t)W * nb;                                        /* [2][W][8] */ \
    double *s_flux = own + 2 * (size_t)W * 8; \
    double *s_ivar = s_flux + nb; \
    double *s_invcov = s_ivar + nb; \
    int2 *s_band = reinterpret_cast<int2 *>(s_invcov + (a.cov_in_lds ? (size_t)nb * nb : 0)); \
    int *ctl = reinterpret_cast<int *>(s_band + nb + 1);                         /* edone[2] at 4; ready[kFaNB][8] at 16; qdone[kFaNB] at 32, 40 */ \
    const size_t tab_off = ((size_t)(reinterpret_cast<unsigned char *>(ctl + 48) - smem_raw) + 15) & ~(size_t)15; \
    double *s_nu = reinterpret_cast<double *>(smem_raw + tab_off); \
    double *s_lnnu = s_nu + (STAGE ? a.nchunk * 64 : 0); \
    double *s_wt = s_lnnu + (STAGE ? a.nchunk * 64 : 0); \
    const int wbase = (int)blockIdx.x * W; \
    const int Wl = min(W, a.n - wbase); \
    const FlowMView fv = flowm_view(a.spec, a.nw); \
    const unsigned long long serial32 = a.flow_serial << 32; \
    unsigned long long *const done_set = fv.done + (size_t)(a.spec_cfg & 1) * kFmRing * 16; \
    const long long spin_limit = flow_spin_limit(a.spec_cfg); \
    const int niter = a.persist; \
    int *const c_ready = ctl + 16, *const c_qdone = ctl + 32, *const c_edone = ctl + 4; \
    (void)nun; (void)nQ; (void)mflux; (void)own; (void)s_flux; (void)s_ivar; (void)s_invcov; (void)s_nu; (void)s_lnnu; (void)s_wt; \
    (void)Wl; (void)done_set; (void)spin_limit; (void)niter; (void)c_ready; (void)c_qdone; (void)c_edone; (void)partial0; (void)rec0

    // ---- set-up, once per launch (every thread): tables and data to LDS; the owned rows as the sampler holds them
    // into LDS and, with this launch's check words, into slot 0 of the run's state; workgroup 0 clears the completion
    // counters of the sampler's NEXT launch
    {
        MBB_ROLE_ARGS();
        MBB_FA_COMMON();
        const int nt = (int)blockDim.x;
        if (tid < 48) ctl[tid] = 0;
        if (tid < 2 * W * 6) {
            const int hh = tid / (W * 6), l = (tid - hh * W * 6) / 6, e = tid % 6;
            if (l < Wl) {
                const int r = (hh ? a.c_count : 0) + wbase + l;
                const double v = a.pos6[(size_t)r * 6 + e];
                own[(size_t)(hh * W + l) * 8 + e] = v;
                fm_put(fv.row + (size_t)r * kFmWords + 2 * e, v, serial32);
            }
        }
        if (blockIdx.x == 0 && tid < kFmRing * 16)
            __hip_atomic_store(fv.done + (size_t)((a.spec_cfg & 1) ^ 1) * kFmRing * 16 + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        (void)nC;
    }
    __syncthreads();

    // diagnostic build: when things happened in the launch's last 64 half-steps, on the clock all CUs share (100 MHz):
    // stamps[(workgroup * 64 + half-step mod 64) * 16 + event], tools/probe_chain_flowa.py
#ifdef MBB_STAMPS
#define FA_EV(jj, ev) do { if (lane == 0 && a.stamps && (jj) >= niter - 64) \
        a.stamps[(((size_t)blockIdx.x * 64 + ((jj) & 63)) * 16 + (ev))] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FA_EV(jj, ev) do { } while (0)
#endif
    const int W_u = ka->wpb;
    const int nC_u = (2 * W_u + 3) >> 2;
    const bool is_c = (wave & 3) == 3 && (wave >> 2) < nC_u;
    // running totals the counters of buffer (j & 1) have reached when half-step j is through
#define MBB_FA_TURN(j) (((j) >> 1) + 1)

    if (!is_c) {
        // =========================== Q: quadrature, then band sums + accept test =====================
        MBB_ROLE_ARGS();
        MBB_FA_COMMON();
        MBB_FM_WAITS();
        MBB_PIN(a.unit_tab); MBB_PIN(a.tail_slot); MBB_PIN(a.errflag); MBB_PIN(a.n); MBB_PIN(a.nw); MBB_PIN(a.c_count);
        MBB_PIN(a.has_gprior); MBB_PIN(a.invcov); MBB_PIN(a.cov_in_lds); MBB_PIN(a.pos6); MBB_PIN(a.chain6); MBB_PIN(a.nacc);
        if (!STAGE) { MBB_PIN(a.nu); MBB_PIN(a.lnnu); MBB_PIN(a.wt); }
        const int qi = wave - min(wave >> 2, nC);          // Q wave number: the C waves (3, 7, ..) below it skipped
        auto T_nu = [&](int i) { if constexpr (STAGE) return s_nu[i]; else return a.nu[i]; };
        auto T_ln = [&](int i) { if constexpr (STAGE) return s_lnnu[i]; else return a.lnnu[i]; };
        auto T_wt = [&](int i) { if constexpr (STAGE) return s_wt[i]; else return a.wt[i]; };
        const SampleTabs tabs = {s_tab, s_pb, s_pc};
        // The walkers are taken one by one, each as soon as its record is there (its partner's decision has come in), in
        // whatever order that is; of walker l's units -- numbers l nun + i of the workgroup's list -- this wave has every
        // nQ-th, as when the list was taken in one sweep behind one word: then the workgroup sat idle 2.4 us of a half-step's
        // 10.5 at four walkers, waiting for the last of its four partners' decisions (tools/probe_chain_flowa.py; 21.0 -> 18.6 us
        // per step at 2000 walkers, 17.3 -> 15.6 at 1500, 12.0 -> 11.3 at 1000).  The accept tests stay behind the whole sweep,
        // on the first Wl waves: done walker by walker by whichever wave brought a walker's last unit in, the wave that was last
        // once fell further behind with every test and took them all (23.2 at 2000 walkers); done by the waves with the least
        // to do, as soon as a walker's units were in, 19.5 (profiles/r04/walker_sweep.txt).
        for (int it = 0; it < niter; ++it) {
            const int bj = it & (kFaNB - 1), h = it & 1;
            const WalkerK *wk = wk0 + bj * W;
            double *partial = partial0 + (size_t)bj * W * npart;
            const double *rec = rec0 + (size_t)bj * W * kFaRec;
            const int turn = MBB_FA_TURN(it);
            if (qi == 0) FA_EV(it, 0);
            // ---- a walker's band sums, lnL and move (k_lnlike, phase 3, SAMPLER): what other workgroups wait for
            auto accept_test = [&](const int j) {
                const int st = wk[j].status;
                const int row = (h ? a.c_count : 0) + wbase + j;
                const double *rj = rec + (size_t)j * kFaRec;
                const double pen_u = rj[7], pen_g = rj[8];
                double acc = 0.0;
                if (st == ROW_OK) {
                    double *mf = mflux + (size_t)j * nb;
                    const double *pj2 = partial + j * npart;
                    const double cbb = wk[j].cq;
                    auto band = [&](const int b) {
                        double sum = 0.0;
                        const int2 rng = s_band[b];
                        for (int sg = rng.x; sg < rng.y; sg += 4) {
                            const int le = rng.y - 1;
                            const double q0 = pj2[sg], q1 = pj2[min(sg + 1, le)], q2 = pj2[min(sg + 2, le)], q3 = pj2[min(sg + 3, le)];
                            sum += q0;
                            if (sg + 1 < rng.y) sum += q1;
                            if (sg + 2 < rng.y) sum += q2;
                            if (sg + 3 < rng.y) sum += q3;
                        }
                        sum *= cbb;
                        const double d = s_flux[b] - sum;          // likelihood.py:821
                        if (a.invcov) mf[b] = d;
                        else acc = fma(d * d, s_ivar[b], acc);     // :825
                    };
                    for (int b = lane; b < nb; b += 64) band(b);
                    if (a.invcov) {                                // :823
                        MBB_FM_ORDER();
                        for (int i = lane; i < nb; i += 64) {
                            double t = 0.0;
                            const double *crow = (a.cov_in_lds ? s_invcov : a.invcov) + (size_t)i * nb;
                            for (int jj = 0; jj < nb; ++jj) t = fma(crow[jj], mf[jj], t);
                            acc = fma(mf[i], t, acc);
                        }
                    }
                    acc = (nb <= 16) ? wave_sum_row0(acc) : wave_sum(acc);
                }
                double r;
                if (st == ROW_BELOW_LOWLIM) r = -__builtin_inf();
                else if (st != ROW_OK) r = __builtin_nan("");
                else {
                    r = fma(-0.5, acc, pen_u);                     // :828
                    if (a.has_gprior) r += pen_g;                  // :830-831
                }
                double *orow = own + (size_t)(h * W + j) * 8;
                const double lnp_cur = orow[5];
                if (lane == 0 && (st >= 2 || r != r)) atomicMax(a.errflag, (st >= 2 && st <= (int)ROW_NONFINITE) ? st : (int)ROW_NONFINITE);
                const bool accept = (rj[5] + r - lnp_cur) > rj[6];  // min(1, z^(dim-1) P(q)/P(s)) against u
                const int m_new = flow_cnt(h, it) + 1;
                // the decision first -- all that the constructors of the next half-step wait for
                if (lane == 0)
                    __hip_atomic_store(fv.mseq + (size_t)row * kFmMseq + (m_new % kFmSlots),
                                       serial32 | (2ull * (unsigned long long)(it + 1) + (accept ? 1ull : 0ull)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                // element (lane & 7) of the row as it is after this half-step
                double ve = accept ? r : lnp_cur;
#pragma unroll
                for (int i = 0; i < 5; ++i) ve = ((lane & 7) == i) ? (accept ? rj[i] : orow[i]) : ve;
                MBB_FM_ORDER();                                    // (every lane has read the old row)
                if (lane < 6) {
                    fm_put(fv.row + ((size_t)(m_new % kFmSlots) * a.nw + row) * kFmWords + 2 * lane, ve,
                           serial32 | (unsigned long long)(it + 1));
                    orow[lane] = ve;
                    // the row's last move of the launch: back into the sampler's rows (no kernel after this one)
                    if (it + 2 >= niter) a.pos6[(size_t)row * 6 + lane] = ve;
                } else if (lane >= 8 && lane < 14 && a.chain6) {
                    a.chain6[((size_t)it * a.n + (wbase + j)) * 6 + (lane - 8)] = ve;
                }
                if (lane == 0 && accept) atomicAdd(a.nacc + (size_t)h * a.n + (wbase + j), 1u);
                MBB_FM_ORDER();
                if (lane == 0) __hip_atomic_fetch_add(c_edone + bj, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            unsigned todo = (1u << Wl) - 1u;
            int nth = 0;
            while (todo) {
                // the next walker whose record is there (a wave that waits here must not take issue slots from those of its
                // SIMD that are still on a walker: lowest priority, and a longer nap between looks)
                int j = -1;
                {
                    long long spins = 0;
                    __builtin_amdgcn_s_setprio(0);
                    for (;;) {
                        unsigned t = todo;
                        while (t) {
                            const int b = __builtin_ctz(t);
                            t &= t - 1;
                            if (__hip_atomic_load(c_ready + bj * 8 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= turn) { j = b; break; }
                        }
                        if (j >= 0) break;
                        ++spins;
                        if (spins > spin_limit * 16 ||
                            ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            atomicMax(a.errflag, 9);
                            j = __builtin_ctz(todo);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    __builtin_amdgcn_s_setprio(1);
                    MBB_FM_ORDER();
                }
                j = __builtin_amdgcn_readfirstlane(j);
                todo &= ~(1u << j);
                if (qi == 0 && nth == 0) FA_EV(it, 1);
                for (int ui = ((qi - j * nun) % nQ + nQ) % nQ; ui < nun; ui += nQ) {
                    const int4 us = a.unit_tab[ui];
                    if (wk[j].status != ROW_OK) continue;             // wave-uniform
                    const WalkerK k = wk[j];
                    const int s = us.x, c0 = us.y, c1 = us.z;
                    double acc = 0.0;
                    int c = c0;
                    for (; c + 2 <= c1; c += 2) {                     // two chunks per step (k_lnlike, do_unit)
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
                        if (lane == 63) partial[j * npart + s] = acc;
                    } else if (us.w == 2) {
                        acc = row_sum(acc);
                        if ((lane & 15) == 0) {
                            const int sl = a.tail_slot[4 * s + (lane >> 4)];
                            if (sl >= 0) partial[j * npart + sl] = acc;
                        }
                    } else {
                        partial[j * npart + s + lane] = acc;
                    }
                }
                if (qi == 0 && nth == Wl - 1) FA_EV(it, 2);
                ++nth;
            }
            MBB_FM_ORDER();
            if (lane == 0) __hip_atomic_fetch_add(c_qdone + bj * 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (qi < Wl) {
                __builtin_amdgcn_s_setprio(0);
                lds_wait(c_qdone + bj * 8, nQ * turn);
                __builtin_amdgcn_s_setprio(2);
                if (qi == 0) FA_EV(it, 3);
                accept_test(qi);
                if (qi == 0) FA_EV(it, 4);
                __builtin_amdgcn_s_setprio(1);
            }
        }
        return;
    }

    // =========================== C: the proposals, constructed for both candidates ==================
    {
        MBB_ROLE_ARGS();
        MBB_FA_COMMON();
        MBB_FM_WAITS();
        MBB_PIN(a.c_count); MBB_PIN(a.step); MBB_PIN(a.seed); MBB_PIN(a.stretch_a); MBB_PIN(a.nw); MBB_PIN(a.errflag); MBB_PIN(a.n);
        MBB_PIN(a.lowlim[0]); MBB_PIN(a.lowlim[1]); MBB_PIN(a.lowlim[2]); MBB_PIN(a.lowlim[3]); MBB_PIN(a.lowlim[4]);
        MBB_PIN(a.nunorm); MBB_PIN(a.lnunorm); MBB_PIN(a.has_uplim); MBB_PIN(a.has_gprior);
        __builtin_amdgcn_s_setprio(3);
        const int ci = wave >> 2;
        const int lane_w = lane;
        for (int j = 0; j < niter; ++j) {
            const int lane = fm_loop_lane(lane_w);
            const int rr = 4 * ci + (lane >> 4), l = rr >> 1, cand = rr & 1, l16 = lane & 15, base = lane & 48;
            const int h = j & 1, bj = j & (kFaNB - 1);
            const int sb = h ? a.c_count : 0, ob = h ? 0 : a.c_count;   // the half that moves in j / the other
            const int row = sb + wbase + l;
            const bool c1 = cand && j > 0;
            const bool active = l < Wl && (cand == 0 || j > 0);
            double zz = 1.0, u3 = 0.5;
            int pj = 0;
            if (active) stretch_draw(row, a.step + (j >> 1), h, a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(j >> 1),
                                     a.stretch_a, a.c_count, zz, pj, u3);
            const int prow = ob + pj;
            // the partner's pending move (half-step j - 1) is its move number m1; before it the row is version m0
            const int m1 = flow_cnt(h ^ 1, j), m0 = flow_cnt(h ^ 1, j - 1);
            // the walker's own row must be final (its move of half-step j - 2) and the record buffer free
            if (j >= kFaNB) lds_wait(c_edone + bj, Wl * MBB_FA_TURN(j - kFaNB));
            // (1) one polling loop: lane e < 5 of the row of lanes takes element e of the partner's row as it is
            // (candidate 0) or of the proposal it is being tested on (candidate 1); lane 5 of the workgroup's first
            // C wave: the lag guard
            const double *src = c1 ? fv.prop + (((size_t)prow * kFmSlots + (m1 % kFmSlots)) * 2) * kFmWords + 2 * (l16 < 5 ? l16 : 0)
                                   : fv.row + ((size_t)(m0 % kFmSlots) * a.nw + prow) * kFmWords + 2 * (l16 < 5 ? l16 : 0);
            const unsigned long long tag = serial32 | (unsigned long long)(c1 ? j : flow_seq(h ^ 1, m0));
            const bool want_e = active && l16 < 5, want_g = ci == 0 && lane == 5 && j >= kFmLag;
            const unsigned long long need_g = (unsigned long long)gridDim.x * (unsigned long long)nC * (unsigned long long)(((j - kFmLag) / kFmRing) + 1);
            const unsigned long long *gword = done_set + ((j - kFmLag) & (kFmRing - 1)) * 16;
            if (ci == 0) FA_EV(j, 8);
            double pv = 0.0;
            {
                bool ok = !(want_e || want_g);
                long long spins = 0;
                for (;;) {
                    if (want_e && !ok) ok = fm_get(src, tag, pv);
                    if (want_g && !ok) ok = __hip_atomic_load(gword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need_g;
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                    ++spins;
                    if (spins > spin_limit ||
                        ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        atomicMax(a.errflag, 9);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (ci == 0) FA_EV(j, 9);
            WalkerK k;
            k.hokt9 = k.lhokt9 = k.beta = k.bp3 = k.cq = k.alpha = k.lx0 = k.xmerge = k.cbb = k.cpl = k.kap = k.peak = 0.0;
            k.status = ROW_SKIP;
            k.pad = 0;
            double pen_u = 0.0, pen_g = 0.0;
            double p[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, lo[4] = {0.0, 0.0, 0.0, 0.0};
            // (2) the partner's decision of half-step j - 1 says which candidate was right: asked for once towards
            // the end of the constructor, polled for afterwards if it is not there yet
            const unsigned long long *w2 = fv.mseq + (size_t)prow * kFmMseq + (m1 % kFmSlots);
            const bool watch2 = active && j > 0 && l16 == 0 && cand == 0;
            unsigned long long v2 = 0;
            if (active) {
                const double *srow = own + (size_t)(h * W + l) * 8;
#pragma unroll
                for (int i = 0; i < 5; ++i) p[i] = stretch_q(__shfl(pv, base + i), srow[i], zz);
                vlog<true>(lo, p[0], p[2], zz, u3);
                const double lT = lo[0], lL = lo[1];
#define MBB_WC_AFTER_PROLOGUE if (watch2) v2 = __hip_atomic_load(w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#include "mbb_walker_consts.inc"
#undef MBB_WC_AFTER_PROLOGUE
            }
            if (ci == 0) FA_EV(j, 10);
            // (3) walker by walker, as the decisions come in: the row of the right candidate puts its constants and its
            // proposal record into LDS and the proposal into the run's state, for the workgroups whose walkers draw this one
            // as partner in half-step j + 1; then the walker's `ready` word -- the quadrature starts on it at once
            {
                const int pbase = lane & 32;                      // first lane of the walker's two rows: the one that watches
                bool handed = !(l < Wl);
                long long spins = 0;
                for (;;) {
                    const unsigned long long v2p = __shfl(v2, pbase);
                    const bool ok = j == 0 || dec_ok(v2p, (unsigned long long)j);
                    const bool now = !handed && ok;
                    if (now) {
                        const int sel = (j > 0 && (v2p & 1ull)) ? 1 : 0;
                        if (cand == sel && l16 == 0) {
                            WalkerK *wk = wk0 + bj * W;
                            double *rj = rec0 + ((size_t)bj * W + l) * kFaRec;
                            if (k.status == ROW_OK) wk[l] = k;
                            else { wk[l].status = k.status; wk[l].pad = k.pad; }
#pragma unroll
                            for (int i = 0; i < 5; ++i) rj[i] = p[i];
                            rj[5] = 4.0 * lo[2];                  // (dim - 1) ln z, dim = 5
                            rj[6] = lo[3];                        // ln u
                            rj[7] = pen_u;
                            rj[8] = pen_g;
                            double *pr = fv.prop + (((size_t)row * kFmSlots + ((flow_cnt(h, j) + 1) % kFmSlots)) * 2) * kFmWords;
                            const unsigned long long ptag = serial32 | (unsigned long long)(j + 1);
#pragma unroll
                            for (int i = 0; i < 5; ++i) fm_put(pr + 2 * i, p[i], ptag);
                        }
                        handed = true;
                    }
                    MBB_FM_ORDER();                               // (the record's stores are issued before the word's)
                    if (now && lane == pbase) __hip_atomic_store(c_ready + bj * 8 + l, MBB_FA_TURN(j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (__builtin_amdgcn_ballot_w64(!handed) == 0) break;
                    ++spins;
                    if (spins > spin_limit ||
                        ((spins & 255) == 8 && __hip_atomic_load(a.errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                        atomicMax(a.errflag, 9);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if (watch2 && !handed) v2 = __hip_atomic_load(w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (ci == 0) FA_EV(j, 11);
            MBB_FM_ORDER();
            if (ci == 0) FA_EV(j, 12);
            // this wave has read what it needs of half-step j - 1's rows, proposals and decisions: the lag guard
            if (lane == 0) __hip_atomic_fetch_add(done_set + (j & (kFmRing - 1)) * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#undef MBB_FA_COMMON
#undef MBB_FA_TURN
#undef FA_EV
}
