// mbb_flowr.hip.h -- k_flowr, sampler form 8: the launch train made resident (single GPU, single ensemble).
// Included by mbb_flow.hip only.
//
// Ensembles with more than one pair of walkers per CU cannot have a workgroup per (pair, candidate)
// (k_flowm, form 7) and fell to a train of one launch per half-step: 19 us per step at 1000 walkers, 26 at 2000
// (cfg3's ensemble on one GPU), of which the kernel boundary -- launch ramp, the tables into LDS again, the
// tail of the slowest workgroup -- is about half.  Here a run is ONE launch per 4096 steps: workgroup g owns
// walkers [g W, g W + W) of BOTH halves of the ensemble (W = walkers per CU per half, up to 8: 4096 walkers
// on 256 CUs), keeps their rows in LDS, its tables staged once, and does for them, half-step after half-step,
// exactly what k_lnlike's half-step launch (SMODE 1) does for its walkers -- the same draw, the same proposal
// arithmetic, the same constructor text (mbb_walker_consts.inc), the same units in the same order, the same
// order of the band sums and of the accept test -- so the chain is bitwise the launch train's.  Nothing is
// computed ahead and nothing twice: with several walkers per CU the quadrature is throughput, not latency,
// and speculating on it (form 7) doubles what bounds the half-step.
//
// What replaces the launch boundary: a walker's proposal needs its partner's row -- any row of the other
// half, moved a half-step earlier by whatever workgroup owns it.  Rows are published as form 7 publishes
// them (FlowMView.row, filed under the number m of the move mod kFmSlots, every element with a check word
// carrying the launch's serial and the half-step): a reader takes an element when the pair fits, in
// whatever order the two stores land, so the writer neither waits for its stores nor raises a flag.  A
// half-step of a workgroup therefore starts when the rows IT depends on are there -- no grid-wide barrier --
// and the lag guard (completion counters per half-step mod kFmRing, form 7's) keeps a slot from being
// rewritten under a reader: a workgroup enters half-step j only when every workgroup has read what it
// needed for j - kFmLag.  Every wait is bounded and watches the run's error flag; a run that gives up
// (a workgroup not resident) is redone by the host as a launch train, as for the other one-launch forms.
// Inside a workgroup the phases of a half-step are separated by two workgroup barriers, as in k_lnlike; the
// hand-over records in LDS are double-buffered by half-step parity so that the waves that finish the accept
// test late do not hold up the next proposals.
#pragma once
#include "mbb_kernels.hip.h"

constexpr int kFrMaxW = 8;     // walkers per workgroup and half (two prologue waves of four rows of 16 lanes)

// dynamic LDS of a k_flowr launch besides the staged passband tables (bytes)
__host__ __device__ constexpr size_t flowr_lds(size_t nb, size_t npart, bool cov_in_lds, size_t W)
{
    return 2 * W * (sizeof(WalkerK) + 8 * npart + 8 * 8 + 8 * 2) + 8 * W * nb + 8 * 2 * W * 8 + 16 * nb +
           (cov_in_lds ? 8 * nb * nb : 0) + 8 * (nb + 2) + 64;
}

template <bool OPTHIN, bool NOALPHA, bool STAGE>
__global__ void __launch_bounds__(1024) k_flowr(const LikeArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ Exp2Entry s_tab[kExp2N];
    __shared__ __align__(16) double s_pb[kPolyBDoubles];
    __shared__ __align__(16) double s_pc[OPTHIN ? 2 : kPolyCDoubles];
    const int W = a.wpb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwave = blockDim.x >> 6;
    const int nun = a.nunit, npart = a.npart, nb = a.nb;
    // LDS: per buffer (half-step parity) the walkers' constants, segment sums, proposal records, penalties;
    // then the difference vectors of the covariance form, the owned rows [2 halves][W][8], the data, the tables
    WalkerK *wk0 = reinterpret_cast<WalkerK *>(smem_raw);                  // [2][W]
    double *partial0 = reinterpret_cast<double *>(wk0 + 2 * W);           // [2][W * npart]
    double *prop0 = partial0 + 2 * (size_t)W * npart;                     // [2][W * 8]
    double *pen0 = prop0 + 2 * (size_t)W * 8;                             // [2][W * 2]
    double *mflux = pen0 + 2 * (size_t)W * 2;                             // [W * nb]
    double *own = mflux + (size_t)W * nb;                                 // [2][W][8]
    double *s_flux = own + 2 * (size_t)W * 8;                             // [nb]
    double *s_ivar = s_flux + nb;                                         // [nb]
    double *s_invcov = s_ivar + nb;                                       // [nb * nb] when it fits
    int2 *s_band = reinterpret_cast<int2 *>(s_invcov + (a.cov_in_lds ? (size_t)nb * nb : 0));   // [nb]
    const size_t tab_off = ((size_t)(reinterpret_cast<unsigned char *>(s_band + nb + 1) - smem_raw) + 15) & ~(size_t)15;
    double *s_nu = reinterpret_cast<double *>(smem_raw + tab_off);
    double *s_lnnu = s_nu + (STAGE ? a.nchunk * 64 : 0);
    double *s_wt = s_lnnu + (STAGE ? a.nchunk * 64 : 0);
    const int wbase = (int)blockIdx.x * W;
    const int Wl = min(W, a.n - wbase);                                   // (the last workgroup may own fewer)
    const FlowMView fv = flowm_view(a.spec, a.nw);
    const unsigned long long serial32 = a.flow_serial << 32;
    unsigned long long *const done_set = fv.done + (size_t)(a.spec_cfg & 1) * kFmRing * 16;
    const long long spin_limit = flow_spin_limit(a.spec_cfg);
    const int niter = a.persist;

    // ---- set-up, once per launch: tables and data to LDS; the owned rows as the sampler holds them into LDS
    // and, with this launch's check words, into slot 0 of the run's state (no kernel before this one);
    // workgroup 0 clears the completion counters of the sampler's NEXT launch
    {
        const int nt = (int)blockDim.x;
        const double2 *gb = reinterpret_cast<const double2 *>(a.poly_b);
        const double2 *gc = reinterpret_cast<const double2 *>(a.poly_c);
        double2 *lb = reinterpret_cast<double2 *>(s_pb);
        double2 *lc = reinterpret_cast<double2 *>(s_pc);
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
    __syncthreads();

    auto T_nu = [&](int i) { if constexpr (STAGE) return s_nu[i]; else return a.nu[i]; };
    auto T_ln = [&](int i) { if constexpr (STAGE) return s_lnnu[i]; else return a.lnnu[i]; };
    auto T_wt = [&](int i) { if constexpr (STAGE) return s_wt[i]; else return a.wt[i]; };
    const SampleTabs tabs = {s_tab, s_pb, s_pc};
    // this wave's first quadrature unit (the same in every half-step)
    const int nunit = W * nun;
    int4 us_first = make_int4(0, 0, 0, 0);
    if (wave < nunit) us_first = a.unit_tab[wave % nun];

    for (int it = 0; it < niter; ++it) {
        const int h = it & 1, buf = it & 1;
        const int L_step = a.step + (it >> 1);
        const unsigned long long L_seed = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(it >> 1);
        const int sb = h ? a.c_count : 0, ob = h ? 0 : a.c_count;       // the half that moves / the other
        WalkerK *const wk = wk0 + buf * W;
        double *const partial = partial0 + (size_t)buf * W * npart;
        double *const prop = prop0 + (size_t)buf * W * 8;
        double *const pen = pen0 + (size_t)buf * W * 2;

        // ---- phase 1 (k_lnlike's, SAMPLER): draw, partner's row, proposal, gate, constructor, penalties --
        // one row of 16 lanes per owned walker
        if (const int l = tid >> 4; l < W) {
            const bool lead = (tid & 15) == 0;
            const int l16 = tid & 15, base = lane & 48;
            const bool have = l < Wl;
            const int row = sb + wbase + l;
            WalkerK k;
            k.status = ROW_SKIP;
            k.pad = 0;
            double pen_u = 0.0, pen_g = 0.0;
            double zz = 1.0, u3 = 0.5;
            int pj = 0;
            if (have) stretch_draw(row, L_step, h, L_seed, a.stretch_a, a.c_count, zz, pj, u3);
            // the partner's row as it is after the half-step before (element e in lane e of the walker's row of
            // lanes, taken when its check word fits); lane 5 of the workgroup's first row: the lag guard
            const int m_par = flow_cnt(h ^ 1, it);
            const double *src = fv.row + ((size_t)(m_par % kFmSlots) * a.nw + (ob + pj)) * kFmWords + 2 * (l16 < 5 ? l16 : 0);
            const unsigned long long tag = serial32 | (unsigned long long)flow_seq(h ^ 1, m_par);
            const bool want_e = have && l16 < 5, want_g = tid == 5 && it >= kFmLag;
            const unsigned long long need_g = (unsigned long long)gridDim.x * (unsigned long long)(((it - kFmLag) / kFmRing) + 1);
            const unsigned long long *gword = done_set + ((it - kFmLag) & (kFmRing - 1)) * 16;
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
            if (have) {
                const double *srow = own + (size_t)(h * W + l) * 8;
                double p[5], lT, lL = 0.0;
#pragma unroll
                for (int i = 0; i < 5; ++i) p[i] = stretch_q(__shfl(pv, base + i), srow[i], zz);
                const double srow5 = srow[5];
                double lo[4];
                vlog<true>(lo, p[0], p[2], zz, u3);
                lT = lo[0]; lL = lo[1];
                if (lead) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) prop[l * 8 + i] = p[i];
                    prop[l * 8 + 5] = 4.0 * lo[2];            // (dim - 1) ln z, dim = 5
                    prop[l * 8 + 6] = srow5;                  // the walker's lnprob as it is
                    prop[l * 8 + 7] = lo[3];                  // ln u
                }
#include "mbb_walker_consts.inc"
            }
            if (lead) {
                if (k.status == ROW_OK) wk[l] = k;
                else { wk[l].status = k.status; wk[l].pad = k.pad; }
                pen[2 * l] = pen_u;
                pen[2 * l + 1] = pen_g;
            }
        }
        __syncthreads();
        // this workgroup has read what it needs of the rows of half-step it - 1: counted for the lag guard
        if (tid == 0) __hip_atomic_fetch_add(done_set + (it & (kFmRing - 1)) * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // ---- phase 2 (k_lnlike's): the (walker, unit) pairs dealt to the waves ---------------------
        for (int u = wave; u < nunit; u += nwave) {
            const int j = u / nun;
            const int4 us = (u == wave) ? us_first : a.unit_tab[u - j * nun];
            if (wk[j].status != ROW_OK) continue;                 // wave-uniform
            const WalkerK k = wk[j];
            const int s = us.x, c0 = us.y, c1 = us.z;
            double acc = 0.0;
            int c = c0;
            for (; c + 2 <= c1; c += 2) {                         // two chunks per step (k_lnlike, do_unit)
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
        __syncthreads();

        // ---- phase 3 (k_lnlike's, SAMPLER): band sums in fixed order, lnL, the accept test, the move ----
        // walker j on wave nwave - 1 - j: the first waves, which run the next proposals, are through at once
        if (const int j = nwave - 1 - wave; j < Wl) {
            const int st = wk[j].status;
            const int row = sb + wbase + j;
            const double pen_u = pen[2 * j], pen_g = pen[2 * j + 1];
            double acc = 0.0;
            if (st == ROW_OK) {
                double *mf = mflux + (size_t)j * nb;
                const double *pj2 = partial + j * npart;
                const double cbb = wk[j].cbb;
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
                    const double d = s_flux[b] - sum;              // likelihood.py:821
                    if (a.invcov) mf[b] = d;
                    else acc = fma(d * d, s_ivar[b], acc);         // :825
                };
                for (int b = lane; b < nb; b += 64) band(b);
                if (a.invcov) {                                    // :823
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
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
                r = fma(-0.5, acc, pen_u);                         // :828
                if (a.has_gprior) r += pen_g;                      // :830-831
            }
            double q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = prop[j * 8 + i];
            if (lane == 0 && (st >= 2 || r != r)) atomicMax(a.errflag, (st >= 2 && st <= (int)ROW_NONFINITE) ? st : (int)ROW_NONFINITE);
            const bool accept = (q[5] + r - q[6]) > q[7];          // min(1, z^(dim-1) P(q)/P(s)) against u
            double *orow = own + (size_t)(h * W + j) * 8;
            // element (lane & 7) of the row as it is after this half-step
            double ve = accept ? r : q[6];
#pragma unroll
            for (int i = 0; i < 5; ++i) ve = ((lane & 7) == i) ? (accept ? q[i] : orow[i]) : ve;
            const int m_new = flow_cnt(h, it) + 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();                       // (every lane has read the old row)
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
        }
    }
}
