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
// x = (128 k + j) ln2/128 + r, |r| <= ln2/256:  e^x = 2^k 2^(j/128) e^r with
// 2^(j/128) from a 128-entry hi/lo table (held in LDS by the kernels) and a
// degree-5 polynomial for e^r - 1: 5 polynomial operations instead of 15.
struct Exp2Entry { double hi, lo; };
#ifdef MBB_MATH_HOST
static const Exp2Entry kExp2Tab[128] = {
#else
__device__ const Exp2Entry kExp2Tab[128] = {
#endif

    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.0163da9fb3335p+0, 0x1.b61299ab8cdb7p-54},
    {0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56},
    {0x1.04315e86e7f85p+0, -0x1.0a31c1977c96ep-54},
    {0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55},
    {0x1.0706b29ddf6dep+0, -0x1.c91dfe2b13c27p-55},
    {0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57},
    {0x1.09e3ecac6f383p+0, 0x1.1487818316136p-54},
    {0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54},
    {0x1.0cc922b7247f7p+0, 0x1.01edc16e24f71p-54},
    {0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59},
    {0x1.0fb66affed31bp+0, -0x1.b9bedc44ebd7bp-57},
    {0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54},
    {0x1.12abdc06c31ccp+0, -0x1.1b514b36ca5c7p-58},
    {0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54},
    {0x1.15a98c8a58e51p+0, 0x1.2406ab9eeab0ap-55},
    {0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55},
    {0x1.18af9388c8deap+0, -0x1.11023d1970f6cp-54},
    {0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55},
    {0x1.1bbe084045cd4p+0, -0x1.95386352ef607p-54},
    {0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54},
    {0x1.1ed5022fcd91dp+0, -0x1.1df98027bb78cp-54},
    {0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55},
    {0x1.21f49917ddc96p+0, 0x1.2a97e9494a5eep-55},
    {0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54},
    {0x1.251ce4fb2a63fp+0, 0x1.ac155bef4f4a4p-55},
    {0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55},
    {0x1.284dfe1f56381p+0, -0x1.a4c3a8c3f0d7ep-54},
    {0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55},
    {0x1.2b87fd0dad990p+0, -0x1.10adcd6381aa4p-59},
    {0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54},
    {0x1.2ecafa93e2f56p+0, 0x1.1ca0f45d52383p-56},
    {0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55},
    {0x1.32170fc4cd831p+0, 0x1.a9ce78e18047cp-55},
    {0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54},
    {0x1.356c55f929ff1p+0, -0x1.b5cee5c4e4628p-55},
    {0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54},
    {0x1.38cae6d05d866p+0, -0x1.e958d3c9904bdp-54},
    {0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56},
    {0x1.3c32dc313a8e5p+0, -0x1.efff8375d29c3p-54},
    {0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55},
    {0x1.3fa4504ac801cp+0, -0x1.7d023f956f9f3p-54},
    {0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58},
    {0x1.431f5d950a897p+0, -0x1.1c7dde35f7999p-55},
    {0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59},
    {0x1.46a41ed1d0057p+0, 0x1.c944bd1648a76p-54},
    {0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56},
    {0x1.4a32af0d7d3dep+0, 0x1.9cb62f3d1be56p-54},
    {0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56},
    {0x1.4dcb299fddd0dp+0, 0x1.8ecdbbc6a7833p-54},
    {0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54},
    {0x1.516daa2cf6642p+0, -0x1.f768569bd93efp-55},
    {0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55},
    {0x1.551a4ca5d920fp+0, -0x1.d689cefede59bp-55},
    {0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54},
    {0x1.58d12d497c7fdp+0, 0x1.295e15b9a1de8p-55},
    {0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54},
    {0x1.5c9268a5946b7p+0, 0x1.c4b1b816986a2p-60},
    {0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54},
    {0x1.605e1b976dc09p+0, -0x1.3e2429b56de47p-54},
    {0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54},
    {0x1.6434634ccc320p+0, -0x1.c483c759d8933p-55},
    {0x1.6623882552225p+0, -0x1.bb60987591c34p-54},
    {0x1.68155d44ca973p+0, 0x1.038ae44f73e65p-57},
    {0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54},
    {0x1.6c012750bdabfp+0, -0x1.2895667ff0b0dp-56},
    {0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57},
    {0x1.6ff7df9519484p+0, -0x1.83c0f25860ef6p-55},
    {0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55},
    {0x1.73f9a48a58174p+0, -0x1.0a8d96c65d53cp-54},
    {0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54},
    {0x1.780694fde5d3fp+0, 0x1.866b80a02162dp-54},
    {0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55},
    {0x1.7c1ed0130c132p+0, 0x1.f124cd1164dd6p-54},
    {0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56},
    {0x1.80427543e1a12p+0, -0x1.27c86626d972bp-54},
    {0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54},
    {0x1.8471a4623c7adp+0, -0x1.8d684a341cdfbp-55},
    {0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54},
    {0x1.88ac7d98a6699p+0, 0x1.994c2f37cb53ap-54},
    {0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54},
    {0x1.8cf3216b5448cp+0, -0x1.0d55e32e9e3aap-56},
    {0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55},
    {0x1.9145b0b91ffc6p+0, -0x1.dd6792e582524p-54},
    {0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57},
    {0x1.95a44cbc8520fp+0, -0x1.64b7c96a5f039p-56},
    {0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54},
    {0x1.9a0f170ca07bap+0, -0x1.173bd91cee632p-54},
    {0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56},
    {0x1.9e86319e32323p+0, 0x1.824ca78e64c6ep-56},
    {0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54},
    {0x1.a309bec4a2d33p+0, 0x1.6305c7ddc36abp-54},
    {0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54},
    {0x1.a799e1330b358p+0, 0x1.bcb7ecac563c7p-54},
    {0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54},
    {0x1.ac36bbfd3f37ap+0, -0x1.f9234cae76cd0p-55},
    {0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54},
    {0x1.b0e07298db666p+0, -0x1.bdef54c80e425p-54},
    {0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57},
    {0x1.b59728de5593ap+0, -0x1.c71dfbbba6de3p-54},
    {0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56},
    {0x1.ba5b030a1064ap+0, -0x1.efcd30e54292ep-54},
    {0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55},
    {0x1.bf2c25bd71e09p+0, -0x1.efdca3f6b9c73p-54},
    {0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55},
    {0x1.c40ab5fffd07ap+0, 0x1.b4537e083c60ap-54},
    {0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54},
    {0x1.c8f6d9406e7b5p+0, 0x1.1acbc48805c44p-56},
    {0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56},
    {0x1.cdf0b555dc3fap+0, -0x1.dd83b53829d72p-55},
    {0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54},
    {0x1.d2f87080d89f2p+0, -0x1.d487b719d8578p-54},
    {0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55},
    {0x1.d80e316c98398p+0, -0x1.11ec18beddfe8p-54},
    {0x1.da9e603db3285p+0, 0x1.c2300696db532p-54},
    {0x1.dd321f301b460p+0, 0x1.2da5778f018c3p-54},
    {0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54},
    {0x1.e264614f5a129p+0, -0x1.7b627817a1496p-54},
    {0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55},
    {0x1.e7a51fbc74c83p+0, 0x1.2d522ca0c8de2p-54},
    {0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54},
    {0x1.ecf482d8e67f1p+0, -0x1.c93f3b411ad8cp-54},
    {0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54},
    {0x1.f252b376bba97p+0, 0x1.3a1a5bf0d8e43p-54},
    {0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54},
    {0x1.f7bfdad9cbe14p+0, -0x1.dbb12d006350ap-54},
    {0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55},
    {0x1.fd3c22b8f71f1p+0, 0x1.2eb74966579e7p-57},
};

// LO / HI: clamp x to [-800, 800] on that side; a caller that knows its argument's
// range leaves the clamp out
template <bool LO = true, bool HI = true>
MBB_HD double reduce_ln2_128(double x, int &n)
{
    if (LO) x = fmax(x, -800.0);
    if (HI) x = fmin(x, 800.0);
    // round(x 128/ln2) by adding 1.5 2^52: the integer lands in the low mantissa
    // bits (read as an int) and the subtraction gives it back as a double -- one
    // add instead of a round and a convert
    const double shifted = fma(x, 184.66496523378731, 6755399441055744.0);
#ifdef MBB_MATH_HOST
    union { double d; int64_t i; } u; u.d = shifted; n = (int)(u.i & 0xffffffff);
#else
    n = __double2loint(shifted);
#endif
    const double nd = shifted - 6755399441055744.0;
    double r = fma(nd, -5.41521234812457272e-03, x);           // ln2/128 hi
    r = fma(nd, -1.81175532330184346e-19, r);                  // ln2/128 lo
    return r;
}

MBB_HD double expm1_small(double r)        // |r| <= ln2/256: r + r^2/2 + ... + r^5/120
{
    const double r2 = r * r;
    const double a0 = fma(r, 1.0 / 6.0, 0.5);
    const double a1 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    return fma(r2, fma(r2, a1, a0), r);
}

template <bool LO = true, bool HI = true>
MBB_HD double m_exp_t(double x, const Exp2Entry *tab)
{
    int n;
    const double q = expm1_small(reduce_ln2_128<LO, HI>(x, n));
    const Exp2Entry t = tab[n & 127];
    return ldexp(t.hi + fma(t.hi, q, t.lo), n >> 7);
}

template <bool LO = true, bool HI = true>
MBB_HD double m_expm1_t(double x, const Exp2Entry *tab)
{
    int n;
    const double q = expm1_small(reduce_ln2_128<LO, HI>(x, n));
    const Exp2Entry t = tab[n & 127];
    const double p = ldexp(1.0, n >> 7);           // inf for k >= 1024, as wanted
    const double S = t.hi * p;
    return (S - 1.0) + fma(S, q, t.lo * p);
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
    const double lm = f - s * (f - z * p);
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
}

}  // namespace mbbm
