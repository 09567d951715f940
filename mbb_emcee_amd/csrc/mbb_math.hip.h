// mbb_math.hip.h -- lean fp64 exp / expm1 / log / divide for the gfx950 kernels.
//
// The device library's exp/expm1/pow carry special-case handling and long
// Horner chains that the likelihood path does not need: its arguments are finite,
// its quotients have positive finite-or-inf denominators, and it is either
// latency bound (one row of lanes per walker in the prologue) or VALU-issue bound (the
// passband loop).  These versions use one shared range reduction, an Estrin
// polynomial (dependency depth 5 instead of 13) and no branches.
//
// Accuracy (tools/test_math_host.cpp, 4e6 points each, vs long double libm):
//   m_exp, m_expm1  <= 2 ulp over [-745, 709.7];  m_log <= 2 ulp;  m_div <= 1.5 ulp.
//
// Compiles for the host too (MBB_MATH_HOST) so the accuracy test runs on CPU.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef MBB_MATH_HOST
#define MBB_HD inline
static inline double m_rcp_seed(double b) { return (double)(1.0f / (float)b); }
#else
#include <hip/hip_runtime.h>
#define MBB_HD __device__ __forceinline__
static __device__ __forceinline__ double m_rcp_seed(double b) { return __builtin_amdgcn_rcp(b); }
#endif

namespace mbbm {

// e^r - 1 on |r| <= ln2/2 as r + r^2 * s(r), s = 1/2! + r/3! + ... + r^11/13!
MBB_HD double expm1_reduced(double r)
{
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = fma(r, 1.0 / 6.0, 0.5);                          // c2 + c3 r
    const double a1 = fma(r, 1.0 / 120.0, 1.0 / 24.0);                 // c4 + c5 r
    const double a2 = fma(r, 1.0 / 5040.0, 1.0 / 720.0);               // c6 + c7 r
    const double a3 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0);           // c8 + c9 r
    const double a4 = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0);       // c10 + c11 r
    const double a5 = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);   // c12 + c13 r
    const double b0 = fma(a1, r2, a0);
    const double b1 = fma(a3, r2, a2);
    const double b2 = fma(a5, r2, a4);
    const double s = fma(b2, r8, fma(b1, r4, b0));
    return fma(r2, s, r);
}

// x = k ln2 + r, |r| <= ln2/2.  x is clamped to [-800, 800]: beyond that the
// results are 0 / inf anyway and the clamp keeps +-inf from turning into NaN.
MBB_HD double reduce_ln2(double x, int &k)
{
    x = fmin(fmax(x, -800.0), 800.0);
    const double kd = rint(x * 1.4426950408889634074);
    double r = fma(kd, -6.93147180559945286227e-01, x);        // ln2 hi (nearest double)
    r = fma(kd, -2.31904681384629955842e-17, r);               // ln2 lo
    k = (int)kd;
    return r;
}

MBB_HD double m_exp(double x)
{
    int k;
    const double q = expm1_reduced(reduce_ln2(x, k));
    return ldexp(1.0 + q, k);
}

// expm1(x) = 2^k (e^r - 1) + (2^k - 1)
MBB_HD double m_expm1(double x)
{
    int k;
    const double q = expm1_reduced(reduce_ln2(x, k));
    const double t = ldexp(1.0, k);                // inf for k >= 1024, as wanted
    return fma(t, q, t - 1.0);
}

// ---- table-driven variants for the sample loop --------------------------------
// x = (256 k + j) ln2/256 + r, |r| <= ln2/512:  e^x = 2^k 2^(j/256) e^r with 2^(j/256)
// from a 256-entry table of doubles (2 KB, held in LDS by the kernels; the `hi` words of
// tools/gen_exp2_table.py's table: 2^(j/256) rounded to nearest) and a degree-4
// polynomial for e^r - 1 (truncation r^5/120 <= 3.8e-17): 4 polynomial operations
// instead of 15, one fma for the table value.  (Rounds 2-5 carried a hi/lo pair per
// entry and one more operation per exp: 0.6 ulp instead of 1.1 here, which nothing
// downstream sees -- the stated tolerance is 1e-12 -- at 8 more bytes of LDS per sample.)
constexpr int kExp2N = 256;
#ifdef MBB_MATH_HOST
static const double kExp2Tab[kExp2N] = {
#else
__device__ __align__(16) const double kExp2Tab[kExp2N] = {
#endif
#include "mbb_exp2_tab.inc"
};

// n = round(x 256/ln2) as an integer, r = x - n ln2/256.  The integer comes out of a
// SATURATING conversion (v_cvt_i32_f64), so an argument far outside exp's range needs no
// clamp: n pins at INT_MIN / INT_MAX, ldexp by n >> 8 = -+8388608 gives 0 / inf, and r --
// formed from the unsaturated double, off by ~1e-16 |x| there -- stays small enough for its fourth power to be
// finite (and 1 + r + .. + r^4/24 is positive for every r) as long as |x| < 1e90, which the callers' exponents (held
// at 1e80) times a logarithm are.  (Rounds 2-5 rounded by adding
// 1.5 2^52 and read the integer off the mantissa: two operations instead of three, but
// wrong beyond |x| = 5.8e6, hence a v_max and a v_min in front of every exp.)
MBB_HD double reduce_ln2_256(double x, int &n)
{
    const double nd = rint(x * 3.69329930467574627e+02);      // 256 / ln2
    n = (int)nd;
#ifdef MBB_MATH_HOST
    if (nd >= 2147483647.0) n = 2147483647;                    // (what the device's conversion does by itself)
    if (nd <= -2147483648.0) n = -2147483647 - 1;
#endif
    double r = fma(nd, -2.70760617406228627e-03, x);           // ln2/256 hi
    r = fma(nd, -9.05877661658710765e-20, r);                  // ln2/256 lo
    return r;
}

MBB_HD double expm1_small(double r)        // |r| <= ln2/512: r + r^2/2 + r^3/6 + r^4/24
{
    const double r2 = r * r;
    const double a0 = fma(r, 1.0 / 6.0, 0.5);
    return fma(r2, fma(r2, 1.0 / 24.0, a0), r);
}

// e^x for any finite x (0 below the range, inf above); tab[j] = 2^(j/256)
MBB_HD double m_exp_t(double x, const double *tab)
{
    int n;
    const double q = expm1_small(reduce_ln2_256(x, n));
    const double t = tab[n & (kExp2N - 1)];
    return ldexp(fma(t, q, t), n >> 8);
}

// Piecewise degree-7 polynomials (mbb_host_tables.h) of a function of x >= 0 on intervals
// [i/8, (i+1)/8): row i holds eight coefficients, lowest order first, in t = 8x - i, and
// starts kPolyStride doubles after row i - 1.  The caller hands over X = 8x: the row is its
// integer part (v_cvt_i32_f64) and t its fraction (v_fract_f64) -- two operations where
// rounds 2-5 spent three adds on an interval centred on i/8.  Then Horner: 7 fma.
// kPolyStride = 10: rows of 64 bytes sit on four bank positions of the LDS (a 16-byte read
// covers 4 of 64 banks, row i starts at bank 16 i mod 64) and lanes whose samples fall into
// rows i and i + 4 collide -- 22 % of round 5's LDS cycles; 80-byte rows start at bank
// 20 i mod 64, sixteen different 4-bank positions before they repeat.
constexpr int kPolyStride = 10;
// how far the two tables go (mbb_host_tables.h): b(x) on [0, kPolyBMax], beyond which b(x) = x e^-x to the last bit;
// C(y) on [0, kPolyCMax], beyond which 1 - e^-y is 1 (e^-37 < 2^-53)
constexpr int kPolyBMax = 48, kPolyCMax = 37;
MBB_HD double polyrow_eval(const double *tab, double X)
{
#ifdef MBB_MATH_HOST
    const int i = (int)X;
    const double t = X - floor(X);
#else
    const int i = (int)X;
    const double t = __builtin_amdgcn_fract(X);
#endif
#ifdef MBB_MATH_HOST
    const double *c = tab + kPolyStride * i;
#else
    // (v_mul_u32_u24: the row times its stride at full rate -- a plain 32-bit multiply is a quarter-rate instruction)
    const double *c = reinterpret_cast<const double *>(reinterpret_cast<const char *>(tab) + __mul24(i, kPolyStride * 8));
#endif
    double p = fma(c[7], t, c[6]);
    p = fma(p, t, c[5]);
    p = fma(p, t, c[4]);
    p = fma(p, t, c[3]);
    p = fma(p, t, c[2]);
    p = fma(p, t, c[1]);
    return fma(p, t, c[0]);
}

// a / b for finite a and finite b != 0, or b = +inf (the quotient is then 0).
// v_rcp_f64 is good to 4.6e-8 (measured); one Newton step squares that and the
// residual correction of the quotient multiplies the two errors: <= 1 ulp
// (1e6 random pairs against long double, same as with two steps).
MBB_HD double m_div(double a, double b)
{
    b = fmin(b, 8.0e307);
    double r = m_rcp_seed(b);
    r = r * fma(-b, r, 2.0);
#ifdef MBB_MATH_HOST
    r = r * fma(-b, r, 2.0);       // the float seed of the host build is coarser
    r = r * fma(-b, r, 2.0);
#endif
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// log(x), x > 0 finite normal.  x = 2^e m, m in [sqrt(1/2), sqrt(2));
// log m = 2 atanh(s), s = (m-1)/(m+1), series in s^2 to s^21.
MBB_HD double m_log(double x)
{
    int e;
    double m = frexp(x, &e);                // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double f = m - 1.0;
    const double s = m_div(f, m + 1.0);
    const double z = s * s, z2 = z * z, z4 = z2 * z2;
    const double a0 = fma(z, 2.0 / 5.0, 2.0 / 3.0);
    const double a1 = fma(z, 2.0 / 9.0, 2.0 / 7.0);
    const double a2 = fma(z, 2.0 / 13.0, 2.0 / 11.0);
    const double a3 = fma(z, 2.0 / 17.0, 2.0 / 15.0);
    const double a4 = fma(z, 2.0 / 21.0, 2.0 / 19.0);
    const double p = fma(fma(a4, z4, fma(a3, z2, a2)), z4, fma(a1, z2, a0));
    // log m = 2s + s z p ; write 2s = f - s f  (exact identity: s = f/(2+f))
    const double lm = fma(-s, fma(-z, p, f), f);
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
}

}  // namespace mbbm
