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
using mbbm::m_expm1_t;
using mbbm::Exp2Entry;
using mbbm::kExp2Tab;

// modified_blackbody.py:15-18
constexpr double kH = 6.6260693e-34;      // J s
constexpr double kK = 1.3806505e-23;      // J / K
constexpr double kC_um = 299792458e6;     // um / s
constexpr double kUmToGHz = 299792458e-3;
constexpr double kLog1e9HoK = -3.036713188283967;     // log(1e9 h / k)
constexpr double kLogUmToGHz = 12.610845707563017;    // log(299792.458)

enum RowStatus : int { ROW_OK = 0, ROW_BELOW_LOWLIM = 1, ROW_BAD_ALPHA = 2,
                       ROW_BAD_BETA = 3, ROW_NOCONV = 6, ROW_NONFINITE = 7, ROW_SKIP = -1 };

// Per-walker constants of the sample loop (what modified_blackbody.__init__ leaves
// in _normfac/_xmerge/_kappa/_x0, recast for the exp-only formulation).
struct WalkerK {
    double hokt9;    // 1e9 h / (k T), per GHz                 fnu.pyx:16
    double lhokt9;   // log(hokt9)
    double beta;
    double bp3;      // beta + 3                               fnu.pyx:19
    double alpha;
    double lx0;      // log(x0), thick only
    double xmerge;   // +inf when there is no Wien-side power law
    double cbb;      // normfac
    double cpl;      // normfac * kappa
    double peak;     // lambda_peak in um (only when requested)
    int status;
    int pad;
};

// SED scalars in the reference's own terms (for parity of the constructor).
struct SedScalars { double normfac, xmerge, kappa, x0, hcokt, lhokt9, lx0; };

__device__ __forceinline__ bool finite5(const double *p)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 5; ++i) ok = ok && (fabs(p[i]) <= 1.7976931348623157e308);
    return ok;
}

// h(y) = y / expm1(y) and its derivative; y = (x/x0)^beta >= 0.
// Large y: the reference catches OverflowError and uses 0
// (modified_blackbody.py:144-150).
__device__ __forceinline__ void h_and_dh(double y, double &h, double &dh)
{
    if (!(y < 700.0)) { h = 0.0; dh = 0.0; }
    else if (y < 1e-4) { h = 1.0 - 0.5 * y + y * y * (1.0 / 12.0); dh = -0.5 + y * (1.0 / 6.0); }
    else {
        const double rE = m_div(1.0, m_expm1(y));
        h = y * rE;
        dh = (1.0 - h - y) * rE;
    }
}

// Root of alpha_merge_eqn (modified_blackbody.py:122-151)
//   g(x) = x - (1 - e^-x) (3 + alpha + beta h(y)),  y = (x/x0)^beta
// for the optically thick model, solved for u = log x.  The reference brackets by
// halving from 0.1 and doubling from 15 and then calls brentq (:286-322).
// g(2+alpha) < 0 < g(3+alpha+beta) holds for every alpha, beta >= 0 because
// 0 <= h <= 1, and g has a single sign change, so that interval brackets the same
// root.  Newton in u with the analytic derivative
//   dg/du = x (1 - e^-x A) - (1 - e^-x) beta^2 h'(y) y
// converges quadratically; a bracket keeps it safe.  Working in u makes the two
// exps of an evaluation independent (x = e^u, y = e^(beta (u - log x0))) and
// leaves log(xmerge) = u for the caller.
// fp32 pre-solve of the same equation: a few Newton steps with the hardware
// exp/log (microseconds matter here: the prologue is one lane per walker and
// nothing else in the workgroup can start before it).  Good to ~1e-6 in u.
__device__ inline float thick_merge_root_f32(float alpha, float beta, float lx0,
                                             float ulo, float uhi, float u)
{
    // four plain Newton steps, clamped to the bracket: from the midpoint that is
    // float precision for every (alpha, beta, x0); no convergence test, so all
    // lanes take the same path and rounding noise cannot trigger a slow fallback
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const float x = __expf(u);
        const float y = __expf(beta * (u - lx0));
        float h, dh;
        if (!(y < 80.0f)) { h = 0.0f; dh = 0.0f; }
        else if (y < 0.02f) { h = 1.0f - 0.5f * y + y * y * (1.0f / 12.0f); dh = -0.5f + y * (1.0f / 6.0f); }
        else { const float rE = __frcp_rn(__expf(y) - 1.0f); h = y * rE; dh = (1.0f - h - y) * rE; }
        const float em = __expf(-x), om = 1.0f - em;
        const float A = 3.0f + alpha + beta * h;
        const float g = x - om * A;
        const float dg = x * (1.0f - em * A) - om * beta * beta * dh * y;
        u = fminf(fmaxf(u - g * __frcp_rn(dg), ulo), uhi);
    }
    return u;
}

__device__ inline double thick_merge_root(double alpha, double beta, double lx0, int &status,
                                          double &xroot, double &yroot, int *iters = nullptr)
{
    const double xlo = 2.0 + alpha, xhi = 3.0 + alpha + beta;
    const float fulo = __logf((float)xlo) - 1e-5f, fuhi = __logf((float)xhi) + 1e-5f;
    double ulo = (double)fulo, uhi = (double)fuhi;
    double u = (double)thick_merge_root_f32((float)alpha, (float)beta, (float)lx0, fulo, fuhi,
                                            __logf((float)(0.5 * (xlo + xhi))));
    status = ROW_NOCONV;
    xroot = 0.0; yroot = 0.0;
    for (int it = 0; it < 80; ++it) {
        const double x = m_exp(u);
        const double y = m_exp(beta * (u - lx0));
        double h, dh;
        h_and_dh(y, h, dh);
        const double em = m_exp(-x), om = 1.0 - em;
        const double A = 3.0 + alpha + beta * h;
        const double g = x - om * A;
        const double dg = x * (1.0 - em * A) - om * beta * beta * dh * y;
        if (iters) *iters = it + 1;
        if (g == 0.0) { xroot = x; yroot = y; status = ROW_OK; break; }
        if (g < 0.0) ulo = u; else uhi = u;
        const double step = -g / dg;
        if (fabs(step) <= 1e-6) {
            // Newton is quadratic with |g''/2g'| < 1: the error left after a step
            // of 1e-6 is below 1e-12 (the fp32 pre-solve normally leaves ~1e-7, so
            // ~1e-14; brentq in the reference stops at 2e-12).  One fp64 evaluation.
            // x e^step and y e^(beta step) to third order: exact to 1e-24
            u += step;
            xroot = x * (1.0 + step * (1.0 + step * (0.5 + step * (1.0 / 6.0))));
            const double bs = beta * step;
            yroot = y * (1.0 + bs * (1.0 + bs * (0.5 + bs * (1.0 / 6.0))));
            status = ROW_OK;
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
__device__ inline double thin_fixed_point(double a)
{
    double x = a;
    for (int it = 0; it < 60; ++it) {
        const double e = m_exp(-x);
        const double F = x - a * (1.0 - e), dF = 1.0 - a * e;
        const double step = -F / dF;
        x += step;
        if (fabs(step) <= 1e-8 * fabs(x)) break;       // quadratic: next error < 1 ulp
    }
    return x;
}

// modified_blackbody.__init__ (modified_blackbody.py:168-337).
// lnunorm = log(um_to_GHz / wavenorm), a per-fit constant.
template <bool OPTHIN, bool NOALPHA>
__device__ inline int sed_prologue(double T, double beta, double lambda0, double alpha,
                                   double fnorm, double wavenorm, double lnunorm,
                                   SedScalars &s, int *iters = nullptr)
{
    const double nan = __builtin_nan("");
    s.normfac = nan; s.xmerge = nan; s.kappa = nan; s.x0 = nan; s.hcokt = nan;
    s.lhokt9 = nan; s.lx0 = 0.0;
    if (!NOALPHA && alpha <= 0.0) return ROW_BAD_ALPHA;             // :219-221
    if (beta < 0.0) return ROW_BAD_BETA;                            // :222-224
    const double hcokt = kH * kC_um / (kK * T);                     // :228
    s.hcokt = hcokt;
    const double xnorm = hcokt / wavenorm;                          // :233
    const double lhokt9 = kLog1e9HoK - m_log(T);
    s.lhokt9 = lhokt9;
    const double lxnorm = lhokt9 + lnunorm;                         // log(xnorm)
    int status = ROW_OK;
    if (OPTHIN) {
        // fnorm expm1(xnorm) / xnorm^(3+beta)                       :240-241, :268-269
        const double bbnorm = fnorm * m_expm1(xnorm) * m_exp(-(3.0 + beta) * lxnorm);
        if (NOALPHA) {
            s.normfac = bbnorm;
        } else {
            const double a = 3.0 + alpha + beta;                    // :253-254
            s.xmerge = thin_fixed_point(a);
            s.kappa = m_exp(a * m_log(s.xmerge)) / m_expm1(s.xmerge);   // :259-261
            if (xnorm > s.xmerge)                                   // :264-266
                s.normfac = fnorm * m_exp(alpha * lxnorm) / s.kappa;
            else
                s.normfac = bbnorm;
        }
    } else {
        const double x0 = hcokt / lambda0;                          // :232
        s.x0 = x0;
        const double lx0 = lhokt9 + kLogUmToGHz - m_log(lambda0);   // log(x0)
        s.lx0 = lx0;
        // (xnorm/x0)^beta; -fnorm expm1(xnorm) / (expm1(-(..)^beta) xnorm^3)   :274-276, :335-337
        const double ynorm = m_exp(beta * (lxnorm - lx0));
        const double bbnorm = fnorm * m_expm1(xnorm) /
            (-m_expm1(-ynorm) * (xnorm * xnorm * xnorm));
        if (NOALPHA) {
            s.normfac = bbnorm;
        } else {
            double xm, ym;
            const double um = thick_merge_root(alpha, beta, lx0, status, xm, ym, iters); // :286-322
            s.xmerge = xm;
            // -xm^(3+alpha) expm1(-(xm/x0)^beta) / expm1(xm)          :326-328
            s.kappa = m_exp((3.0 + alpha) * um) * -m_expm1(-ym) / m_expm1(xm);
            if (xnorm > xm)                                         // :331-333
                s.normfac = fnorm * m_exp(alpha * lxnorm) / s.kappa;
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
template <bool OPTHIN>
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
    double xp, yp;
    thick_merge_root(0.0, beta, lx0, status, xp, yp);
    return hcokt / xp;
}

template <bool OPTHIN, bool NOALPHA>
__device__ inline void make_walker_k(double T, double beta, double alpha,
                                     const SedScalars &s, WalkerK &w)
{
    w.hokt9 = 1e9 * kH / (kK * T);                                  // fnu.pyx:16
    w.lhokt9 = s.lhokt9;
    w.beta = beta;
    w.bp3 = beta + 3.0;
    w.alpha = NOALPHA ? 0.0 : alpha;
    w.lx0 = OPTHIN ? 0.0 : s.lx0;
    w.xmerge = NOALPHA ? __builtin_inf() : s.xmerge;
    w.cbb = s.normfac;
    w.cpl = NOALPHA ? 0.0 : s.normfac * s.kappa;
}

// One quadrature sample: f_nu at frequency nu (GHz), lnnu = log(nu).
// fnu.pyx:9-108, the four kernels.  tab != nullptr selects the table-driven
// exp/expm1 (the table sits in LDS; the passband loop), nullptr the polynomial
// ones (one-off evaluations).
template <bool OPTHIN, bool NOALPHA, bool TAB = false>
__device__ __forceinline__ double fnu_sample(const WalkerK &w, double nu, double lnnu,
                                             const Exp2Entry *tab = nullptr)
{
    auto ex = [&](double v) { if constexpr (TAB) return m_exp_t(v, tab); else return m_exp(v); };
    auto em1 = [&](double v) { if constexpr (TAB) return m_expm1_t(v, tab); else return m_expm1(v); };
    const double x = w.hokt9 * nu;
    const double lx = w.lhokt9 + lnnu;
    if (!NOALPHA) {
        if (x > w.xmerge) return w.cpl * ex(-w.alpha * lx);         // :48-49, :102-103
    }
    if (OPTHIN) {
        return w.cbb * m_div(ex(w.bp3 * lx), em1(x));               // :24-25, :51
    } else {
        const double y = ex(w.beta * (lx - w.lx0));                 // :74, :105
        return w.cbb * m_div(-em1(-y) * (x * x * x), em1(x));       // :75-76, :106
    }
}

// wave64 sum through the DPP crossbar (no LDS traffic): butterflies inside each
// row of 16 lanes, then row_bcast15 / row_bcast31 carry the row totals upward;
// the grand total lands in lane 63 and is broadcast through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(phi, plo);
}

__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);     // row_mirror: every lane holds its row's total
    v = dpp_add<0x142, 0xa>(v);     // row_bcast15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);    // uniform: the total in every lane
}

}  // namespace mbbd
