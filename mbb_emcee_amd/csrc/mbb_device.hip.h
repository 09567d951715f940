// mbb_device.hip.h -- device-side fp64 math of the modified blackbody path (gfx950).
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
// host, so every power becomes one exp():  x^p = exp(p (log(1e9 h/kT) + log nu)).
// That leaves 1 (Wien side), 2 (thin) or 3 (thick) exp-class operations and one
// division per sample and no log/pow in the inner loop.  The per-walker prologue
// (modified_blackbody.py:168-337) keeps pow(): it runs once per walker.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mbbd {

// modified_blackbody.py:15-18
constexpr double kH = 6.6260693e-34;      // J s
constexpr double kK = 1.3806505e-23;      // J / K
constexpr double kC_um = 299792458e6;     // um / s
constexpr double kUmToGHz = 299792458e-3;

enum RowStatus : int { ROW_OK = 0, ROW_BELOW_LOWLIM = 1, ROW_BAD_ALPHA = 2,
                       ROW_BAD_BETA = 3, ROW_NOCONV = 6, ROW_SKIP = -1 };

__device__ __forceinline__ double d_exp(double x) { return exp(x); }
__device__ __forceinline__ double d_expm1(double x) { return expm1(x); }
__device__ __forceinline__ double d_log(double x) { return log(x); }
__device__ __forceinline__ double d_pow(double x, double y) { return pow(x, y); }

// Per-walker constants of the inner loop (what modified_blackbody.__init__ leaves
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
struct SedScalars { double normfac, xmerge, kappa, x0, hcokt; };

// h(y) = y / expm1(y) and its derivative; y = (x/x0)^beta >= 0.
// Large y: the reference catches OverflowError and uses 0
// (modified_blackbody.py:144-150).
__device__ __forceinline__ void h_and_dh(double y, double &h, double &dh)
{
    if (!(y < 700.0)) { h = 0.0; dh = 0.0; }
    else if (y < 1e-4) { h = 1.0 - 0.5 * y + y * y * (1.0 / 12.0); dh = -0.5 + y * (1.0 / 6.0); }
    else { double E = d_expm1(y); h = y / E; dh = (1.0 - h - y) / E; }
}

// alpha_merge_eqn (modified_blackbody.py:122-151) and its x-derivative.
//   g(x) = x - (1 - e^-x) (3 + alpha + beta h(y)),  y = (x/x0)^beta
__device__ __forceinline__ double merge_g(double x, double alpha, double beta,
                                          double lx0, double &dg)
{
    double y = d_exp(beta * (d_log(x) - lx0));
    double h, dh;
    h_and_dh(y, h, dh);
    double em = d_exp(-x), om = 1.0 - em;
    double A = 3.0 + alpha + beta * h;
    dg = 1.0 - em * A - om * beta * dh * (beta * y / x);
    return x - om * A;
}

// Root of alpha_merge_eqn for the optically thick model.  The reference brackets
// by halving from 0.1 and doubling from 15 and then calls brentq
// (modified_blackbody.py:286-322).  g(2+alpha) < 0 < g(3+alpha+beta) holds for
// every alpha, beta >= 0 because 0 <= h <= 1, and g has a single sign change, so
// that interval brackets the same root; a safeguarded Newton iteration on it
// converges to a few ulp (the reference stops at xtol = 2e-12).
__device__ inline double thick_merge_root(double alpha, double beta, double lx0, int &status,
                                          int *iters = nullptr)
{
    double lo = 2.0 + alpha, hi = 3.0 + alpha + beta;
    double x = 0.5 * (lo + hi);
    status = ROW_NOCONV;
    for (int it = 0; it < 80; ++it) {
        double dg, g = merge_g(x, alpha, beta, lx0, dg);
        if (iters) *iters = it + 1;
        if (g == 0.0) { status = ROW_OK; break; }
        if (g < 0.0) lo = x; else hi = x;
        const double step = -g / dg;
        // Newton converges quadratically here (|g''/2g'| < 1): once a step is
        // below 1e-8 x the error left after taking it is below one ulp.
        if (fabs(step) <= 1e-8 * fabs(x)) { x += step; status = ROW_OK; break; }
        double xn = x + step;
        if (!(xn > lo && xn < hi)) {
            xn = 0.5 * (lo + hi);
            if (!(xn > lo && xn < hi)) { x = xn; status = ROW_OK; break; }  // bracket is 1 ulp
        }
        x = xn;
    }
    return x;
}

// Root of x = a (1 - e^-x), a > 1: the thin merge point a + W0(-a e^-a)
// (modified_blackbody.py:246-254) and, with a = 3 + beta, the thin SED peak.
// F(x) = x - a(1 - e^-x) is convex and F(a) > 0, so Newton from x = a descends
// monotonically onto the root.
__device__ inline double thin_fixed_point(double a)
{
    double x = a;
    for (int it = 0; it < 60; ++it) {
        double e = d_exp(-x);
        double F = x - a * (1.0 - e), dF = 1.0 - a * e;
        double step = -F / dF;
        x += step;
        if (fabs(step) <= 1e-8 * fabs(x)) break;       // quadratic: next error < 1 ulp
    }
    return x;
}

// modified_blackbody.__init__ (modified_blackbody.py:168-337).
template <bool OPTHIN, bool NOALPHA>
__device__ inline int sed_prologue(double T, double beta, double lambda0, double alpha,
                                   double fnorm, double wavenorm, SedScalars &s,
                                   int *iters = nullptr)
{
    const double nan = __builtin_nan("");
    s.normfac = nan; s.xmerge = nan; s.kappa = nan; s.x0 = nan; s.hcokt = nan;
    if (!NOALPHA && alpha <= 0.0) return ROW_BAD_ALPHA;             // :219-221
    if (beta < 0.0) return ROW_BAD_BETA;                            // :222-224
    const double hcokt = kH * kC_um / (kK * T);                     // :228
    s.hcokt = hcokt;
    const double xnorm = hcokt / wavenorm;                          // :233
    int status = ROW_OK;
    if (OPTHIN) {
        if (NOALPHA) {                                              // :240-241
            s.normfac = fnorm * d_expm1(xnorm) / d_pow(xnorm, 3.0 + beta);
        } else {
            const double a = 3.0 + alpha + beta;                    // :253-254
            s.xmerge = thin_fixed_point(a);
            s.kappa = d_pow(s.xmerge, a) / d_expm1(s.xmerge);       // :259-261
            if (xnorm > s.xmerge)                                   // :264-269
                s.normfac = fnorm * d_pow(xnorm, alpha) / s.kappa;
            else
                s.normfac = fnorm * d_expm1(xnorm) / d_pow(xnorm, 3.0 + beta);
        }
    } else {
        const double x0 = hcokt / lambda0;                          // :232
        s.x0 = x0;
        if (NOALPHA) {                                              // :274-276
            s.normfac = -fnorm * d_expm1(xnorm) /
                (d_expm1(-d_pow(xnorm / x0, beta)) * (xnorm * xnorm * xnorm));
        } else {
            s.xmerge = thick_merge_root(alpha, beta, d_log(x0), status, iters); // :286-322
            s.kappa = -d_pow(s.xmerge, 3.0 + alpha) *                // :326-328
                d_expm1(-d_pow(s.xmerge / x0, beta)) / d_expm1(s.xmerge);
            if (xnorm > s.xmerge) {                                 // :331-337
                s.normfac = fnorm * d_pow(xnorm, alpha) / s.kappa;
            } else {
                double expmfac = d_expm1(-d_pow(xnorm / x0, beta));
                s.normfac = -fnorm * d_expm1(xnorm) / (xnorm * xnorm * xnorm * expmfac);
            }
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
    return hcokt / thick_merge_root(0.0, beta, lx0, status);
}

template <bool OPTHIN, bool NOALPHA>
__device__ inline void make_walker_k(double T, double beta, double alpha,
                                     const SedScalars &s, WalkerK &w)
{
    w.hokt9 = 1e9 * kH / (kK * T);                                  // fnu.pyx:16
    w.lhokt9 = d_log(w.hokt9);
    w.beta = beta;
    w.bp3 = beta + 3.0;
    w.alpha = NOALPHA ? 0.0 : alpha;
    w.lx0 = OPTHIN ? 0.0 : d_log(s.x0);
    w.xmerge = NOALPHA ? __builtin_inf() : s.xmerge;
    w.cbb = s.normfac;
    w.cpl = NOALPHA ? 0.0 : s.normfac * s.kappa;
}

// One quadrature sample: f_nu at frequency nu (GHz), lnnu = log(nu).
// fnu.pyx:9-108, the four kernels.
template <bool OPTHIN, bool NOALPHA>
__device__ __forceinline__ double fnu_sample(const WalkerK &w, double nu, double lnnu)
{
    const double x = w.hokt9 * nu;
    const double lx = w.lhokt9 + lnnu;
    if (!NOALPHA) {
        if (x > w.xmerge) return w.cpl * d_exp(-w.alpha * lx);      // :48-49, :102-103
    }
    if (OPTHIN) {
        return w.cbb * d_exp(w.bp3 * lx) / d_expm1(x);              // :24-25, :51
    } else {
        const double y = d_exp(w.beta * (lx - w.lx0));              // :74, :105
        return w.cbb * (-d_expm1(-y)) * (x * x * x) / d_expm1(x);   // :75-76, :106
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;       // valid in lane 0
}

}  // namespace mbbd
