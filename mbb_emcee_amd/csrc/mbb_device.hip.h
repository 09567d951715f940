// mbb_device.hip.h -- device-side fp64 model of the modified blackbody path (gfx950).
//
// Everything here runs on the GPU only.  Reference citations are relative to
// the reference's mbb_emcee/ directory.
//
// Formulation.  The reference evaluates, per quadrature sample (fnu.pyx:9-108),
//     x = (1e9 h / k T) nu
//     thin :  normfac x^(beta+3) / expm1(x)
//     thick: -normfac expm1(-(x/x0)^beta) x^3 / expm1(x)
//     Wien side (x > xmerge):  normfac kappa x^(-alpha)
// with libm pow().  Here log(nu) is tabulated once per passband sample on the
// host and log T, log lambda0 are taken once per walker, so every power becomes
// one exp():  x^p = exp(p (log(1e9 h/k) - log T + log nu)).  That leaves
// 1 (Wien side), 2 (thin) or 3 (thick) exp-class operations and one division
// per sample, and no log/pow anywhere in the sample loop.  The per-walker
// prologue (modified_blackbody.py:168-337) is written in the same log space,
// so it needs two logs and a handful of exps; its root finds iterate on
// u = log x for the same reason.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mbb_math.hip.h"

namespace mbbd {

using mbbm::m_div;
using mbbm::m_exp;
using mbbm::m_expm1;
using mbbm::m_log;
using mbbm::m_exp_t;
using mbbm::kExp2Tab;
using mbbm::kExp2N;
using mbbm::polyrow_eval;

// modified_blackbody.py:15-18
constexpr double kH = 6.6260693e-34;      // J s
constexpr double kK = 1.3806505e-23;      // J / K
constexpr double kC_um = 299792458e6;     // um / s
constexpr double kUmToGHz = 299792458e-3;
constexpr double kLog1e9HoK = -3.036713188283967;     // log(1e9 h / k)
constexpr double kLogUmToGHz = 12.610845707563017;    // log(299792.458)

// diagnostic build only: s_memtime stamps from inside the prologue (tools/probe_stamps.py)
// (MBB_STAMPS_FINE besides: every stamp is a scalar memory read, a wait and a global store by one lane -- a few hundred
// cycles each ON the chain they time; the coarse events of a diagnostic build are only honest without them)
#ifdef MBB_STAMPS
__device__ unsigned long long *g_pstamps;
#endif
#if defined(MBB_STAMPS) && defined(MBB_STAMPS_FINE)
#define PSTAMP(i, dep) do { if (threadIdx.x == 0 && blockIdx.x < 65536 && g_pstamps) { \
    asm volatile("" ::"v"(dep)); g_pstamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define PSTAMP(i, dep) do { } while (0)
#endif

enum RowStatus : int { ROW_OK = 0, ROW_BELOW_LOWLIM = 1, ROW_BAD_ALPHA = 2,
                       ROW_BAD_BETA = 3, ROW_NOCONV = 6, ROW_NONFINITE = 7, ROW_SKIP = -1 };

// Per-walker constants of the sample loop (what modified_blackbody.__init__ leaves
// in _normfac/_xmerge/_kappa/_x0, recast for the exp-only formulation).
struct WalkerK {
    double hokt9;    // 1e9 h / (k T), per GHz                 fnu.pyx:16
    double lhokt9;   // log(hokt9)
    double beta;
    double bp3;      // beta + 3                               fnu.pyx:19
    double cq;       // normfac (h/kT)^2: the band scale of the fused kernels' quadrature, which sums f_nu / x^2 against
                     // weights w nu^2 (mbb_host_tables.h) -- rounds 2-5 kept beta + 2 here
    double alpha;
    double lx0;      // log(x0), thick only
    double xmerge;   // +inf when there is no Wien-side power law
    double cbb;      // normfac
    double cpl;      // normfac * kappa
    double kap;      // kappa (the fused kernel sums unscaled samples and applies normfac per band)
    double peak;     // lambda_peak in um (only when requested)
    int status;
    int pad;
};

// SED scalars in the reference's own terms (for parity of the constructor).
struct SedScalars { double normfac, xmerge, kappa, hcokt, hokt9, lhokt9, lx0; };

__device__ __forceinline__ bool finite5(const double *p)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 5; ++i) ok = ok && (fabs(p[i]) <= 1.7976931348623157e308);
    return ok;
}

// ---- rows ---------------------------------------------------------------------
// In the fused kernel a walker's prologue is latency: nothing else in its workgroup
// can start before it, and a lone lane issues one dependent fp64 operation every
// ~9 cycles.  So the prologue runs on a *row* of 16 lanes (one DPP row) per walker.
// All lanes of a row hold the same scalars and execute the same instructions;
// where the algebra has several independent exp / expm1 / log evaluations they are
// dealt to the lanes of the row (lane i takes argument i) and one call does them
// all, the results coming back through DPP row broadcasts.  ROW = false is the same
// arithmetic, value for value, on a single lane (one-off kernels).
template <int N>
__device__ __forceinline__ double row_bcast(double v)      // lane N of each row -> its row
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + N, 0xf, 0xf, false);   // row_newbcast:N
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int K, int I = 0>
__device__ __forceinline__ void row_scatter(double res, double (&out)[K])
{
    if constexpr (I < K) {
        out[I] = row_bcast<I>(res);
        row_scatter<K, I + 1>(res, out);
    }
}

// lane i of the row takes the i-th argument (scalars, not an array: an indexable
// array would be spilled to scratch and fetched back by lane number)
template <int I = 0, typename... Rest>
__device__ __forceinline__ double row_pick(double first, Rest... rest)
{
    if constexpr (sizeof...(rest) == 0) return first;
    else return ((int)(threadIdx.x & 15) == I) ? first : row_pick<I + 1>(rest...);
}

// out[i] = expm1(arg_i) where bit i of M1 is set, exp(arg_i) elsewhere
template <bool ROW, unsigned M1, int K, typename... A>
__device__ __forceinline__ void vexp(double (&out)[K], A... arg)
{
    static_assert(sizeof...(arg) == K, "one output per argument");
    if constexpr (ROW) {
        int k;
        const double q = mbbm::expm1_reduced(mbbm::reduce_ln2(row_pick(arg...), k));
        const double e = ldexp(1.0 + q, k);                     // m_exp
        const double t = ldexp(1.0, k);
        const double m = fma(t, q, t - 1.0);                    // m_expm1
        row_scatter<K>(((M1 >> (threadIdx.x & 15)) & 1u) ? m : e, out);
    } else {
        const double a[K] = {arg...};
#pragma unroll
        for (int i = 0; i < K; ++i) out[i] = ((M1 >> i) & 1u) ? m_expm1(a[i]) : m_exp(a[i]);
    }
}

template <bool ROW, int K, typename... A>
__device__ __forceinline__ void vlog(double (&out)[K], A... arg)
{
    static_assert(sizeof...(arg) == K, "one output per argument");
    if constexpr (ROW) {
        row_scatter<K>(m_log(row_pick(arg...)), out);
    } else {
        const double a[K] = {arg...};
#pragma unroll
        for (int i = 0; i < K; ++i) out[i] = m_log(a[i]);
    }
}

// h(y) = y / expm1(y) and its derivative, E = expm1(y); y = (x/x0)^beta >= 0.
// Large y: the reference catches OverflowError and uses 0
// (modified_blackbody.py:144-150).
__device__ __forceinline__ void h_and_dh(double y, double E, double &h, double &dh)
{
    const double rE = m_div(1.0, E);
    const double hh = y * rE, dd = (fma(-y, rE, 1.0) - y) * rE;
    const bool tiny = y < 1e-4, big = !(y < 700.0);
    h = big ? 0.0 : (tiny ? fma(y * y, 1.0 / 12.0, fma(-0.5, y, 1.0)) : hh);
    dh = big ? 0.0 : (tiny ? fma(y, 1.0 / 6.0, -0.5) : dd);
}

// Root of alpha_merge_eqn (modified_blackbody.py:122-151)
//   g(x) = x - (1 - e^-x) (3 + alpha + beta h(y)),  y = (x/x0)^beta
// for the optically thick model, solved for u = log x.  The reference brackets by
// halving from 0.1 and doubling from 15 and then calls brentq (:286-322).
// g(2+alpha) < 0 < g(3+alpha+beta) holds for every alpha, beta >= 0 because
// 0 <= h <= 1, and g has a single sign change, so that interval brackets the same
// root.  Working in u makes the two exps of an evaluation independent
// (x = e^u, y = e^(beta (u - log x0))) and leaves log(xmerge) = u for the caller.
//
// Stage 1, fp32 with the hardware exp/rcp: the 16 lanes of the row evaluate g at 16
// equispaced points of the bracket, the count of negative values names the
// sub-interval holding the root, three rounds shrink the bracket 3375-fold and a
// secant step through its ends leaves ~1e-7 in u.  No derivative, no data-dependent
// branch.
// (raw v_exp_f32 / v_rcp_f32: 1 ulp, no denormal or division fix-up code)
__device__ __forceinline__ float merge_eqn_f32(float u, float alpha3, float beta, float betal2,
                                               float lx0)
{
    const float kL2e = 1.44269504088896341f;
    const float x = __builtin_amdgcn_exp2f(u * kL2e);
    const float y = __builtin_amdgcn_exp2f((u - lx0) * betal2);
    const float rE = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y * kL2e) - 1.0f);
    float h = (y < 0.02f) ? fmaf(y, fmaf(y, 1.0f / 12.0f, -0.5f), 1.0f) : y * rE;
    h = (y < 80.0f) ? h : 0.0f;
    const float om = 1.0f - __builtin_amdgcn_exp2f(x * -kL2e);
    return fmaf(-om, fmaf(beta, h, alpha3), x);              // x - (1 - e^-x)(3 + alpha + beta h)
}

template <bool ROW>
__device__ inline float thick_merge_root_f32(float alpha, float beta, float lx0, float ulo, float uhi)
{
    const float alpha3 = 3.0f + alpha, betal2 = beta * 1.44269504088896341f;
    float glo = -1.0f, ghi = 1.0f;
#pragma unroll
    for (int round = 0; round < 3; ++round) {
        const float du = (uhi - ulo) * (1.0f / 15.0f);
        int j;
        if constexpr (ROW) {
            const int lane = threadIdx.x & 63;
            const float g = merge_eqn_f32(fmaf((float)(lane & 15), du, ulo), alpha3, beta, betal2, lx0);
            const unsigned long long neg = __builtin_amdgcn_ballot_w64(g < 0.0f);
            j = __popc((unsigned)(neg >> (lane & 48)) & 0xffffu);
            j = min(max(j, 1), 15);
            if (round == 2) {
                glo = __shfl(g, (lane & 48) + j - 1);
                ghi = __shfl(g, (lane & 48) + j);
            }
        } else {
            // one lane: the same 16 values one after the other, counted, none kept (an indexed array of them
            // cost k_prologue 100 B of scratch per lane); the two at the ends of the chosen sub-interval are
            // only wanted after the last round and are evaluated again there -- the same bits
            j = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                j += (merge_eqn_f32(fmaf((float)i, du, ulo), alpha3, beta, betal2, lx0) < 0.0f) ? 1 : 0;
            j = min(max(j, 1), 15);
            if (round == 2) {
                glo = merge_eqn_f32(fmaf((float)(j - 1), du, ulo), alpha3, beta, betal2, lx0);
                ghi = merge_eqn_f32(fmaf((float)j, du, ulo), alpha3, beta, betal2, lx0);
            }
        }
        const float nlo = fmaf((float)(j - 1), du, ulo), nhi = fmaf((float)j, du, ulo);
        ulo = nlo; uhi = nhi;
    }
    const float u = ulo - glo * (uhi - ulo) * __builtin_amdgcn_rcpf(ghi - glo);
    return (u >= ulo && u <= uhi) ? u : 0.5f * (ulo + uhi);
}

// Stage 2, fp64 Newton in u with the analytic derivative
//   dg/du = x (1 - e^-x A) - (1 - e^-x) beta^2 h'(y) y,   A = 3 + alpha + beta h
// inside the analytic bracket, accepted when a step is below 1e-6 (quadratic
// convergence: < 1e-12 left; brentq in the reference stops at 2e-12).  Normally one
// evaluation.  An evaluation is two rounds of independent exps -- (x, y), then
// (e^-x, expm1(y)) -- so each round is one row call.  PB: the prologue's own
// normalisation exps ride in the idle lanes of the first evaluation:
//   round 1 also gives  pb[0] = exp(pb[0]),  pb[1] = expm1(pb[1]),  pb[2] = exp(pb[2])
//   round 2 also gives  pb[3] = expm1(-pb[0])
template <bool ROW, bool PB>
__device__ inline double thick_merge_root(double alpha, double beta, double lx0, int &status,
                                          double &xroot, double &yroot, double (&pb)[4],
                                          int *iters = nullptr, double *kappa = nullptr)
{
    const double xlo = 2.0 + alpha, xhi = 3.0 + alpha + beta;
    const float kLn2 = 0.693147180559945309f;
    const float fulo = __builtin_amdgcn_logf((float)xlo) * kLn2 - 1e-5f;
    const float fuhi = __builtin_amdgcn_logf((float)xhi) * kLn2 + 1e-5f;
    double ulo = (double)fulo, uhi = (double)fuhi;
    double u = (double)thick_merge_root_f32<ROW>((float)alpha, (float)beta, (float)lx0, fulo, fuhi);
    if (!(u >= ulo && u <= uhi)) u = 0.5 * (ulo + uhi);
    PSTAMP(11, u);
    status = ROW_NOCONV;
    xroot = 0.0; yroot = 0.0;
    if (PB && kappa) *kappa = __builtin_nan("");
    for (int it = 0; it < 80; ++it) {
        double x, y, em, E;
        double k0 = 0.0;                  // PB: kappa at this evaluation point
        if (PB && it == 0) {
            // kappa = x^(3+alpha) (1 - e^-y) / expm1(x) (modified_blackbody.py:326-328) needs
            // three more exps of the same x and y: they ride in this evaluation's two rounds
            double o1[6];
            vexp<ROW, 0x08u>(o1, u, beta * (u - lx0), pb[0], pb[1], pb[2], (3.0 + alpha) * u);
            x = o1[0]; y = o1[1]; pb[0] = o1[2]; pb[1] = o1[3]; pb[2] = o1[4];
            PSTAMP(12, x + y + pb[0] + pb[1] + pb[2]);
            double o2[5];
            vexp<ROW, 0x1Eu>(o2, -x, y, -pb[0], -y, x);
            em = o2[0]; E = o2[1]; pb[3] = o2[2];
            PSTAMP(13, em + E + pb[3]);
            k0 = m_div(o1[5] * -o2[3], o2[4]);
        } else {
            double o1[2];
            vexp<ROW, 0x00u>(o1, u, beta * (u - lx0));
            x = o1[0]; y = o1[1];
            double o2[2];
            vexp<ROW, 0x02u>(o2, -x, y);
            em = o2[0]; E = o2[1];
        }
        double h, dh;
        h_and_dh(y, E, h, dh);
        const double om = 1.0 - em;
        // (every fused multiply-add is written out: the library is built with -ffp-contract=off, so
        // that this arithmetic rounds the same wherever it is inlined -- the sampler forms are held
        // to one another bit for bit)
        const double A = fma(beta, h, 3.0 + alpha);
        const double g = fma(-om, A, x);
        const double dg = fma(x, fma(-em, A, 1.0), -(om * beta * beta * dh * y));
        if (iters) *iters = it + 1;
        if (g == 0.0) { xroot = x; yroot = y; status = ROW_OK; if (PB && kappa && it == 0) *kappa = k0; break; }
        if (g < 0.0) ulo = u; else uhi = u;
        const double step = m_div(-g, dg);          // dg > 0 around the root
        if (fabs(step) <= 1e-6) {
            // x e^step and y e^(beta step) to third order: exact to 1e-24
            u += step;
            xroot = x * fma(step, fma(step, fma(step, 1.0 / 6.0, 0.5), 1.0), 1.0);
            const double bs = beta * step;
            yroot = y * fma(bs, fma(bs, fma(bs, 1.0 / 6.0, 0.5), 1.0), 1.0);
            status = ROW_OK;
            // kappa is stationary at the root: d ln kappa / du = s(u) = A - x/(1 - e^-x) =
            // -g/(1 - e^-x), zero there (it IS the merge condition), so over the step
            // ln kappa changes by s step + s' step^2/2 = s step/2 (s = -s' step to first
            // order): kappa(root) = kappa(u) (1 + s step / 2), error O(step^3) < 1e-18
            if (PB && kappa && it == 0) *kappa = k0 * fma(-0.5 * step, m_div(g, om), 1.0);
            break;
        }
        double un = u + step;
        if (!(un > ulo && un < uhi)) {
            un = 0.5 * (ulo + uhi);
            if (!(un > ulo && un < uhi)) { xroot = x; yroot = y; status = ROW_OK; break; }
        }
        u = un;
    }
    return u;
}

// Root of x = a (1 - e^-x), a > 1: the thin merge point a + W0(-a e^-a)
// (modified_blackbody.py:246-254) and, with a = 3 + beta, the thin SED peak.
// F(x) = x - a(1 - e^-x) is convex and F(a) > 0, so Newton from x = a descends
// monotonically onto the root.
// Three fp32 steps with the hardware exp/rcp first (a e^-x < 1 from x = a on, so the
// slope stays positive), then fp64 steps until one is below 1e-8 x: two as a rule.
__device__ inline double thin_fixed_point(double a)
{
    const float af = (float)a;
    float xf = af;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const float e = __builtin_amdgcn_exp2f(xf * -1.44269504088896341f);
        xf -= (xf - af * (1.0f - e)) * __builtin_amdgcn_rcpf(1.0f - af * e);
    }
    double x = (xf > 0.0f && xf <= af) ? (double)xf : a;
    for (int it = 0; it < 60; ++it) {
        const double e = m_exp(-x);
        const double F = fma(-a, 1.0 - e, x), dF = fma(-a, e, 1.0);
        const double step = m_div(-F, dF);
        x += step;
        if (fabs(step) <= 1e-8 * fabs(x)) break;       // quadratic: next error < 1 ulp
    }
    return x;
}

// modified_blackbody.__init__ (modified_blackbody.py:168-337).
// lT = log T and lL = log lambda0 come from the caller's vlog (the sampler adds its
// own logs to that call); nunorm = um_to_GHz / wavenorm and lnunorm = log(nunorm)
// are per-fit constants.  Quotients with a positive denominator use m_div.
template <bool OPTHIN, bool NOALPHA, bool ROW>
__device__ inline int sed_prologue(double T, double beta, double alpha, double fnorm, double lT,
                                   double lL, double nunorm, double lnunorm, SedScalars &s,
                                   int *iters = nullptr)
{
    // (s is left untouched on the two error returns, and xmerge / kappa without alpha)
    if (!NOALPHA && alpha <= 0.0) return ROW_BAD_ALPHA;             // :219-221
    if (beta < 0.0) return ROW_BAD_BETA;                            // :222-224
    const double hokt9 = m_div(1e9 * kH / kK, T);                   // fnu.pyx:16, per GHz
    s.hokt9 = hokt9;
    s.hcokt = hokt9 * kUmToGHz;                                     // :228, h c / k T in um
    const double xnorm = hokt9 * nunorm;                            // :233
    const double lhokt9 = kLog1e9HoK - lT;
    s.lhokt9 = lhokt9;
    const double lxnorm = lhokt9 + lnunorm;                         // log(xnorm)
    s.lx0 = 0.0;
    int status = ROW_OK;
    if (OPTHIN) {
        // fnorm expm1(xnorm) / xnorm^(3+beta)                       :240-241, :268-269
        if (NOALPHA) {
            double o1[2];
            vexp<ROW, 0x01u>(o1, xnorm, -(3.0 + beta) * lxnorm);
            s.normfac = fnorm * o1[0] * o1[1];
        } else {
            const double a = 3.0 + alpha + beta;                    // :253-254
            s.xmerge = thin_fixed_point(a);
            double lxm[1];
            vlog<ROW>(lxm, s.xmerge);
            double o1[5];
            vexp<ROW, 0x09u>(o1, xnorm, -(3.0 + beta) * lxnorm, a * lxm[0], s.xmerge, alpha * lxnorm);
            s.kappa = m_div(o1[2], o1[3]);                          // :259-261
            if (xnorm > s.xmerge)                                   // :264-266
                s.normfac = m_div(fnorm * o1[4], s.kappa);
            else
                s.normfac = fnorm * o1[0] * o1[1];
        }
    } else {
        const double lx0 = lhokt9 + kLogUmToGHz - lL;               // log(x0), :232
        s.lx0 = lx0;
        // (xnorm/x0)^beta; -fnorm expm1(xnorm) / (expm1(-(..)^beta) xnorm^3)   :274-276, :335-337
        if (NOALPHA) {
            double o1[2];
            vexp<ROW, 0x02u>(o1, beta * (lxnorm - lx0), xnorm);
            double o2[1];
            vexp<ROW, 0x01u>(o2, -o1[0]);
            s.normfac = m_div(fnorm * o1[1], -o2[0] * (xnorm * xnorm * xnorm));
        } else {
            double xm, ym;
            double pb[4] = {beta * (lxnorm - lx0), xnorm, alpha * lxnorm, 0.0};
            double kfast;
            const double um = thick_merge_root<ROW, true>(alpha, beta, lx0, status, xm, ym, pb, iters, &kfast); // :286-322
            PSTAMP(14, um + xm + ym);
            const double bbnorm = m_div(fnorm * pb[1], -pb[3] * (xnorm * xnorm * xnorm));
            s.xmerge = xm;
            // -xm^(3+alpha) expm1(-(xm/x0)^beta) / expm1(xm)          :326-328
            if (kfast == kfast) {
                s.kappa = kfast;         // the usual case: one evaluation, kappa came with it
            } else {
                double o3[3];
                vexp<ROW, 0x06u>(o3, (3.0 + alpha) * um, -ym, xm);
                s.kappa = m_div(o3[0] * -o3[1], o3[2]);
            }
            PSTAMP(15, s.kappa);
            if (xnorm > xm)                                         // :331-333
                s.normfac = m_div(fnorm * pb[2], s.kappa);
            else
                s.normfac = bbnorm;
        }
    }
    return status;
}

// max_wave (modified_blackbody.py:581-637).  Setting the derivative of
// S_nu (x^(3+beta)/expm1(x), or (1-e^-y) x^3/expm1(x)) to zero gives
//   thin : x = (3+beta)(1 - e^-x)
//   thick: x = (1 - e^-x)(3 + beta h(y))       -- alpha_merge_eqn with alpha = 0
// which is what the reference's _snudev root (:556-579) solves numerically.
template <bool OPTHIN, bool ROW>
__device__ inline double sed_peak_wave(double T, double beta, double lx0, double hcokt,
                                       int &status)
{
    status = ROW_OK;
    if (OPTHIN) {
        if (beta == 0.0) {                                          // :600-604
            double numax_bb = 2.82144 * kK * T / kH;
            return kC_um / numax_bb;
        }
        return hcokt / thin_fixed_point(3.0 + beta);
    }
    double xp, yp, pb[4];
    thick_merge_root<ROW, false>(0.0, beta, lx0, status, xp, yp, pb);
    return hcokt / xp;
}

template <bool OPTHIN, bool NOALPHA>
__device__ inline void make_walker_k(double beta, double alpha, const SedScalars &s, WalkerK &w)
{
    w.hokt9 = s.hokt9;
    w.lhokt9 = s.lhokt9;
    // (exponents beyond 1e80 are held there: the sample loop multiplies them by a logarithm, and the product must stay
    // below 1e90 for its exp (mbb_math.hip.h, reduce_ln2_256) -- which gives 0, 1 or inf for such powers whatever the
    // exponent's exact size)
    w.beta = fmin(beta, 1.0e80);
    w.bp3 = beta + 3.0;
    w.cq = s.normfac * (s.hokt9 * s.hokt9);
    w.alpha = NOALPHA ? 0.0 : fmin(alpha, 1.0e80);
    w.lx0 = OPTHIN ? 0.0 : s.lx0;
    w.xmerge = NOALPHA ? __builtin_inf() : s.xmerge;
    w.cbb = s.normfac;
    w.cpl = NOALPHA ? 0.0 : s.normfac * s.kappa;
    w.kap = NOALPHA ? 0.0 : s.kappa;
}

// The tables of the sample loop, all in LDS: 2^(j/256) for exp (mbb_math.hip.h) and the
// piecewise polynomials of b(x) = x/expm1(x) and C(y) = 1 - e^-y (mbb_host_tables.h).
struct SampleTabs { const double *e; const double *b; const double *c; };

constexpr double kLnCMax = 3.6109179126442243;       // log(kPolyCMax = 37): e^-37 < 2^-53, 1 - e^-y is 1 beyond
static_assert(mbbm::kPolyCMax == 37, "kLnCMax is log(37)");
constexpr double kXFar8 = 8.0 * mbbm::kPolyBMax;   // in units of X = 8 x: where the table of b ends

// One quadrature sample: f_nu at frequency nu (GHz), lnnu = log(nu).
// fnu.pyx:9-108, the four kernels.
// TAB = true (the passband loop of the fused kernels; `tabs` points into LDS; always with
// SCALE = false): the value is f_nu / (normfac x^2) -- the Planck factor x^3/expm1(x) is
// x^2 b(x), and the x^2 = (h/kT)^2 nu^2 is not formed per sample: nu^2 is in the weight
// table, (h/kT)^2 goes with normfac into the one factor per band (WalkerK::cq).  b(x) and
// the optical-depth factor C(y) = 1 - e^-y are read off piecewise degree-7 polynomials --
// 10 operations each (row, fraction, address, 7 fma) instead of an expm1 (19) plus, for b,
// a division (10) -- so a sample costs one exp (the power (x/x0)^beta, x^beta or
// x^-(alpha+2)), never a division, and the thick model's sample is C(y) b(x): two
// look-ups and ONE multiplication (rounds 2-5: (y c(y)) ((x x) b(x)), four).  Both tables
// are indexed by eight times their argument: X = (8 h/kT) nu costs what x did, and 8 y
// comes out of the exp for free (m_exp_t's KADD).  For x > 48, beyond the table, 1 - e^-x
// is 1 to the last bit and b(x) = x e^-x.
// TAB = false (one-off evaluations): polynomial exp/expm1 and a true division, the
// same formulas as the reference term by term; the parity tests hold the two against
// each other on the passband grids.
// SCALE = false: without the factor normfac -- the fused kernel applies it once per band
// instead of once per sample (the Wien side is then kappa x^-alpha).
#define MBB_FENCE() __builtin_amdgcn_sched_barrier(0)
// Blackbody side, 0 < x <= 48 (X = 8x): f_nu / (normfac x^2).  The exponent of the power comes straight from
// log(nu) by one fma with two constants of the walker -- log x itself is never formed.
constexpr double kLn8 = 2.0794415416798357;
template <bool OPTHIN>
__device__ __forceinline__ double fnu_bb_tab(const WalkerK &w, double X, double lnnu, const SampleTabs *tabs)
{
    const double b = polyrow_eval(tabs->b, X);
    if (OPTHIN) {
        return m_exp_t(fma(w.beta, lnnu, w.beta * w.lhokt9), tabs->e) * b;      // x^beta b(x)          :24-25, :51
    } else {
        // 8 y = 8 (x/x0)^beta = exp(beta (log x - log x0) + log 8), y held at kPolyCMax             :74, :105
        const double Y = m_exp_t(fmin(fma(w.beta, lnnu, fma(w.beta, w.lhokt9 - w.lx0, kLn8)), kLnCMax + kLn8), tabs->e);
        return polyrow_eval(tabs->c, Y) * b;                                    // (1 - e^-y) b(x)      :75-76, :106
    }
}

// Wien side (x > xmerge): x^-(alpha+2), and with kappa: f_nu / (normfac x^2) = kappa x^-alpha / x^2    :48-49, :102-103
__device__ __forceinline__ double wien_pow_tab(const WalkerK &w, double lnnu, const SampleTabs *tabs)
{
    const double ma2 = -(w.alpha + 2.0);
    return m_exp_t(fma(ma2, lnnu, ma2 * w.lhokt9), tabs->e);
}
__device__ __forceinline__ double fnu_wien_tab(const WalkerK &w, double lnnu, const SampleTabs *tabs)
{
    return w.kap * wien_pow_tab(w, lnnu, tabs);
}

template <bool OPTHIN, bool NOALPHA, bool TAB = false, bool SCALE = true>
__device__ __forceinline__ double fnu_sample(const WalkerK &w, double nu, double lnnu,
                                             const SampleTabs *tabs = nullptr)
{
    auto scaled = [&](double v) { if constexpr (SCALE) return w.cbb * v; else return v; };
    if constexpr (TAB) {
        static_assert(!SCALE, "the table path yields f_nu / (normfac x^2): the band scale is WalkerK::cq");
        // (8 h/kT) nu = 8 x, bit for bit (a power of two); the comparisons in those units
        const double X = (8.0 * w.hokt9) * nu;                          // > 0
        if (!NOALPHA) {
            if (X > 8.0 * w.xmerge) return fnu_wien_tab(w, lnnu, tabs);
        }
        if (X <= kXFar8) return fnu_bb_tab<OPTHIN>(w, X, lnnu, tabs);
        const double x = 0.125 * X;
        const double b = x * m_exp_t(-x, tabs->e);                      // beyond the table: b(x) = x e^-x
        if (OPTHIN) {
            return m_exp_t(fma(w.beta, lnnu, w.beta * w.lhokt9), tabs->e) * b;
        } else {
            const double Y = m_exp_t(fmin(fma(w.beta, lnnu, fma(w.beta, w.lhokt9 - w.lx0, kLn8)), kLnCMax + kLn8), tabs->e);
            return polyrow_eval(tabs->c, Y) * b;
        }
    } else {
        const double x = w.hokt9 * nu;                                  // > 0
        const double lx = w.lhokt9 + lnnu;
        if (!NOALPHA) {
            if (x > w.xmerge) return (SCALE ? w.cpl : w.kap) * m_exp(-w.alpha * lx);
        }
        if (OPTHIN) {
            return scaled(m_div(m_exp(w.bp3 * lx), m_expm1(x)));
        } else {
            const double y = m_exp(w.beta * (lx - w.lx0));
            return scaled(m_div(-m_expm1(-y) * (x * x * x), m_expm1(x)));
        }
    }
}

// wave64 sum through the DPP crossbar (no LDS traffic): butterflies inside each row of 16 lanes, then row_bcast15 /
// row_bcast31 carry the row totals upward; the grand total lands in lane 63.  The tree is the balanced one over the
// lanes in their order -- ((l0 + l1) + (l2 + l3)) + ... per row, then (R3 + R2) + (R1 + R0) -- and every form of every
// kernel sums a unit by it: that is what makes a walker's band fluxes the same bit for bit wherever they are formed.
// A stage is two DPP moves and an add.  (Rounds 2-5 wrote the moves as update_dpp with the value itself as the
// `old` operand -- a register the instruction overwrites, so the compiler copied it first: five instructions a stage,
// 34 for a wave's sum where 20 do; a quarter of everything a 250 000-row launch executed was these reductions.)  The
// carries go to every row: rows 0 and 2 then hold sums nobody reads, lane 63's is the tree's.
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int plo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    const int phi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(phi, plo);
}

__device__ __forceinline__ double row_sum(double v)    // every lane: the total of its row
{
    v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);      // row_half_mirror
    v = dpp_add<0x140>(v);      // row_mirror
    return v;
}

__device__ __forceinline__ double wave_sum_l63(double v)     // the total in lane 63 (the other lanes: partial sums)
{
    v = row_sum(v);
    v = dpp_add<0x142>(v);      // row_bcast15: rows 1 and 3 take R0 and R2 in
    v = dpp_add<0x143>(v);      // row_bcast31: row 3 takes R1 + R0 in
    return v;
}

__device__ __forceinline__ double wave_sum(double v)         // uniform: the total in every lane
{
    v = wave_sum_l63(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// The same total when only lanes 0..15 hold non-zero terms (the other rows would add
// exact zeros): stop after the row stage.
__device__ __forceinline__ double wave_sum_row0(double v)
{
    v = row_sum(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 0);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 0);
    return __hiloint2double(hi, lo);
}

}  // namespace mbbd
