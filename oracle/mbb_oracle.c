/*
 * mbb_oracle.c -- CPU restatement of the mbb_emcee per-walker likelihood path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (mbb_emcee_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against tests/golden/{sed,lnlike,passbands}.npz, which were produced by
 * the reference itself (tests/golden/make_golden.py imports the reference's
 * modules unmodified) and against the reference's own known-answer tests
 * (mbb_emcee/tests/test_modified_blackbody.py:6-69).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/mbb_emcee/).  Third-party algorithms the reference calls
 * and that are not in its tree are restated from their published form:
 *   scipy 1.15.3  scipy.optimize.brentq   (Brent 1973; xtol 2e-12, rtol 4 eps,
 *                                          maxiter 100 -- scipy defaults)
 *   scipy 1.15.3  scipy.special.lambertw  (Halley iteration, Corless et al. 1996)
 *   numpy 2.2.6   ndarray.sum()           (pairwise summation, blocks of 128,
 *                                          8 accumulators)
 *
 * Plain C99, double precision, libm pow/expm1/exp exactly as the reference's
 * Cython kernel uses them.  OpenMP (over walkers) only in the *_batch entry.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* modified_blackbody.py:15-18 */
static const double C_UM = 299792458e6;     /* um / s */
static const double H_PLANCK = 6.6260693e-34;
static const double K_BOLTZ = 1.3806505e-23;
static const double UM_TO_GHZ = 299792458e-3;

enum { MBBO_OK = 0, MBBO_BELOW_LOWLIM = 1, MBBO_BAD_ALPHA = 2, MBBO_BAD_BETA = 3,
       MBBO_BRACKET_LOW = 4, MBBO_BRACKET_HIGH = 5, MBBO_NOCONV = 6,
       MBBO_PEAK_BRACKET = 7 };

typedef struct {
    int opthin, noalpha;
    double T, beta, lambda0, alpha, fnorm, wavenorm;
    double hcokt, x0, xnorm, normfac, xmerge, kappa;
} mbbo_sed;

typedef double (*fn1)(double, const void *);

/* ---- scipy.optimize.brentq, classic Brent-Dekker ----------------------- */
static double brentq(fn1 f, const void *arg, double xa, double xb, int *status)
{
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    double xpre = xa, xcur = xb, xblk = 0.0;
    double fpre = f(xpre, arg), fcur = f(xcur, arg), fblk = 0.0;
    double spre = 0.0, scur = 0.0;
    *status = MBBO_OK;
    if (fpre == 0.0) return xpre;
    if (fcur == 0.0) return xcur;
    if (signbit(fpre) == signbit(fcur)) { *status = MBBO_NOCONV; return NAN; }
    for (int it = 0; it < 100; ++it) {
        if (fpre != 0.0 && fcur != 0.0 && signbit(fpre) != signbit(fcur)) {
            xblk = xpre; fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        double delta = (xtol + rtol * fabs(xcur)) / 2.0;
        double sbis = (xblk - xcur) / 2.0;
        if (fcur == 0.0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);        /* secant */
            } else {                                   /* inverse quadratic */
                double dpre = (fpre - fcur) / (xpre - xcur);
                double dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) /
                       (dblk * dpre * (fblk - fpre));
            }
            double lim = fmin(fabs(spre), 3.0 * fabs(sbis) - delta);
            if (2.0 * fabs(stry) < lim) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0.0 ? delta : -delta);
        fcur = f(xcur, arg);
    }
    *status = MBBO_NOCONV;
    return xcur;
}

/* ---- scipy.special.lambertw(z).real, principal branch, -1/e < z < 0 ---- */
static double lambertw0(double z)
{
    /* start from the branch-point series, then Halley on w e^w = z */
    double p = sqrt(2.0 * (M_E * z + 1.0));
    double w = -1.0 + p - p * p / 3.0 + 11.0 / 72.0 * p * p * p;
    for (int it = 0; it < 100; ++it) {
        double ew = exp(w), wew = w * ew, wewz = wew - z;
        double wn = w - wewz / (wew + ew - (w + 2.0) * wewz / (2.0 * w + 2.0));
        if (fabs(wn - w) <= 1e-8 * fabs(wn)) return wn;
        w = wn;
    }
    return w;
}

/* ---- modified_blackbody.py:122-151 alpha_merge_eqn ---------------------- */
typedef struct { double alpha, beta, x0; } merge_arg;
static double alpha_merge_eqn(double x, const void *p)
{
    const merge_arg *a = (const merge_arg *)p;
    double xox0beta = pow(x / a->x0, a->beta);
    double bterm;
    /* Python raises OverflowError from ** (result inf) or from math.expm1
       (arg > ~709.78); both are caught and give bterm = 0 (:144-150) */
    if (isinf(xox0beta) || xox0beta > 709.782712893384) bterm = 0.0;
    else bterm = xox0beta / expm1(xox0beta);
    return x - (1.0 - exp(-x)) * (3.0 + a->alpha + a->beta * bterm);
}

/* ---- modified_blackbody.py:168-337 __init__ ----------------------------- */
int mbbo_sed_init(mbbo_sed *s, double T, double beta, double lambda0,
                  double alpha, double fnorm, double wavenorm,
                  int noalpha, int opthin)
{
    memset(s, 0, sizeof(*s));
    s->T = T; s->beta = beta; s->lambda0 = lambda0; s->alpha = alpha;
    s->fnorm = fnorm; s->wavenorm = wavenorm;
    s->noalpha = noalpha; s->opthin = opthin;
    s->xmerge = NAN; s->kappa = NAN; s->x0 = NAN;
    if (!noalpha && alpha <= 0.0) return MBBO_BAD_ALPHA;         /* :219-221 */
    if (beta < 0.0) return MBBO_BAD_BETA;                        /* :222-224 */
    s->hcokt = H_PLANCK * C_UM / (K_BOLTZ * T);                  /* :228 */
    if (!opthin) s->x0 = s->hcokt / lambda0;                     /* :232 */
    s->xnorm = s->hcokt / wavenorm;                              /* :233 */
    const double xnorm = s->xnorm;
    if (opthin) {
        if (noalpha) {                                           /* :240-241 */
            s->normfac = fnorm * expm1(xnorm) / pow(xnorm, 3.0 + beta);
        } else {
            double a = 3.0 + alpha + beta;                       /* :253-254 */
            s->xmerge = a + lambertw0(-a * exp(-a));
            s->kappa = pow(s->xmerge, 3.0 + alpha + beta) / expm1(s->xmerge);
            if (xnorm > s->xmerge)                               /* :264-269 */
                s->normfac = fnorm * pow(xnorm, alpha) / s->kappa;
            else
                s->normfac = fnorm * expm1(xnorm) / pow(xnorm, 3.0 + beta);
        }
    } else {
        if (noalpha) {                                           /* :274-276 */
            s->normfac = -fnorm * expm1(xnorm) /
                (expm1(-pow(xnorm / s->x0, beta)) * pow(xnorm, 3.0));
        } else {
            merge_arg ma = { alpha, beta, s->x0 };
            double a = 0.1, aval = alpha_merge_eqn(a, &ma);      /* :286-300 */
            int it = 0;
            while (aval >= 0.0) {
                a /= 2.0; aval = alpha_merge_eqn(a, &ma);
                if (it > 100) return MBBO_BRACKET_LOW;
                ++it;
            }
            double b = 15.0, bval = alpha_merge_eqn(b, &ma);     /* :302-317 */
            it = 0;
            while (bval <= 0.0) {
                b *= 2.0; bval = alpha_merge_eqn(b, &ma);
                if (it > 100) return MBBO_BRACKET_HIGH;
                ++it;
            }
            int st;
            s->xmerge = brentq(alpha_merge_eqn, &ma, a, b, &st); /* :320-322 */
            if (st) return st;
            s->kappa = -pow(s->xmerge, 3.0 + alpha) *            /* :326-328 */
                expm1(-pow(s->xmerge / s->x0, beta)) / expm1(s->xmerge);
            if (xnorm > s->xmerge) {                             /* :331-337 */
                s->normfac = fnorm * pow(xnorm, alpha) / s->kappa;
            } else {
                double expmfac = expm1(-pow(xnorm / s->x0, beta));
                s->normfac = -fnorm * expm1(xnorm) /
                    (pow(xnorm, 3.0) * expmfac);
            }
        }
    }
    return MBBO_OK;
}

/* ---- fnu.pyx:9-108, the four kernels, freq in GHz ----------------------- */
void mbbo_fnu(const mbbo_sed *s, const double *freq, int n, double *out)
{
    const double hokt9 = 1e9 * H_PLANCK / (K_BOLTZ * s->T);      /* fnu.pyx:16 */
    if (s->opthin && s->noalpha) {                               /* :9-27 */
        const double bp3 = s->beta + 3.0;
        for (int i = 0; i < n; ++i) {
            double cx = hokt9 * freq[i];
            out[i] = s->normfac * pow(cx, bp3) / expm1(cx);
        }
    } else if (s->opthin) {                                      /* :31-53 */
        const double bp3 = s->beta + 3.0;
        for (int i = 0; i < n; ++i) {
            double cx = hokt9 * freq[i], r;
            if (cx > s->xmerge) r = s->kappa * pow(cx, -s->alpha);
            else r = pow(cx, bp3) / expm1(cx);
            out[i] = s->normfac * r;
        }
    } else if (s->noalpha) {                                     /* :57-78 */
        for (int i = 0; i < n; ++i) {
            double cx = hokt9 * freq[i];
            double x0b = pow(cx / s->x0, s->beta);
            out[i] = -s->normfac * expm1(-x0b) * pow(cx, 3.0) / expm1(cx);
        }
    } else {                                                     /* :82-108 */
        for (int i = 0; i < n; ++i) {
            double cx = hokt9 * freq[i], r;
            if (cx > s->xmerge) {
                r = s->kappa * pow(cx, -s->alpha);
            } else {
                double x0b = pow(cx / s->x0, s->beta);
                r = -expm1(-x0b) * pow(cx, 3.0) / expm1(cx);
            }
            out[i] = s->normfac * r;
        }
    }
}

/* modified_blackbody.py:535-552 __call__: wavelengths in um -> f_nu */
void mbbo_sed_call(const mbbo_sed *s, const double *wave, int n, double *out,
                   double *scratch)
{
    for (int i = 0; i < n; ++i) scratch[i] = UM_TO_GHZ / wave[i]; /* :551 */
    mbbo_fnu(s, scratch, n, out);
}

/* ---- numpy pairwise summation (ndarray.sum on a contiguous f8 vector) --- */
static double pairwise_sum(const double *a, ptrdiff_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (ptrdiff_t i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        ptrdiff_t i;
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        ptrdiff_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* ---- modified_blackbody.py:556-637 _snudev + max_wave ------------------- */
static double snudev(double x, const void *p)
{
    const mbbo_sed *s = (const mbbo_sed *)p;
    double efac = expm1(x);
    if (s->opthin) {                                             /* :564-567 */
        return pow(x, 2.0 + s->beta) * (3.0 + s->beta) / efac -
               exp(x) * pow(x, 3.0 + s->beta) / pow(efac, 2.0);
    }
    double xx0 = x / s->x0;                                      /* :569-579 */
    double xx0b = pow(xx0, s->beta);
    if (isinf(xx0b))            /* OverflowError branch (:577-579) */
        return 3.0 * pow(x, 2.0) / efac - exp(x) * pow(x, 3.0) / pow(efac, 2.0);
    double ebfac = -expm1(-xx0b);
    return 3.0 * pow(x, 2.0) * ebfac / efac -
           exp(x) * pow(x, 3.0) * ebfac / pow(efac, 2.0) +
           s->beta * pow(x, 3.0) * exp(-xx0b) * xx0b / (x * efac);
}

int mbbo_max_wave(const mbbo_sed *s, double *out)
{
    const double xmax_bb = 2.82144;                              /* :600 */
    if (s->opthin && s->beta == 0.0) {                           /* :602-604 */
        double numax_bb = xmax_bb * K_BOLTZ * s->T / H_PLANCK;
        *out = C_UM / numax_bb;
        return MBBO_OK;
    }
    double a = xmax_bb / 2.0, aval = snudev(a, s);               /* :608-618 */
    int it = 0;
    while (aval <= 0.0) {
        if (it > 20) return MBBO_PEAK_BRACKET;
        a /= 2.0; aval = snudev(a, s); ++it;
    }
    double b = xmax_bb * 2.0, bval = snudev(b, s);               /* :621-630 */
    it = 0;
    while (bval >= 0.0) {
        if (it > 20) return MBBO_PEAK_BRACKET;
        b *= 2.0; bval = snudev(b, s); ++it;
    }
    int st;
    double xmax = brentq(snudev, s, a, b, &st);                  /* :633 */
    if (st) return st;
    double numax = xmax * K_BOLTZ * s->T / H_PLANCK;             /* :636-637 */
    *out = C_UM / numax;
    return MBBO_OK;
}

/* ---- likelihood configuration (likelihood.py:24-117, :158-232, :330-357) */
typedef struct {
    int opthin, noalpha;
    double wavenorm;
    int nb;                     /* number of data points / bands            */
    int response_integrate;     /* 1: bands are passbands, 0: plain wavelengths */
    const double *wave;         /* response: concatenated band wavelengths (um);
                                   else: nb data wavelengths                */
    const double *sedmult;      /* concatenated trapezoid*response weights  */
    const double *normfac;      /* [nb] pipeline normalisation (signed)     */
    const int32_t *offsets;     /* [nb+1] into wave / sedmult               */
    const double *flux;         /* [nb] */
    const double *ivar;         /* [nb] 1/unc^2                             */
    const double *invcov;       /* [nb*nb] row-major or NULL                */
    double lowlim[5];
    int32_t has_uplim[6];
    double uplim[6];
    int32_t has_gprior[6];
    double gprior_mean[6];
    double gprior_ivar[6];
} mbbo_like;

/* response.py:544-576 for one band (non-delta) */
static double band_flux(const mbbo_sed *s, const double *wave,
                        const double *sedmult, double normfac, int n,
                        double *buf /* 3n scratch */)
{
    double *f = buf, *prod = buf + n, *fr = buf + 2 * n;
    mbbo_sed_call(s, wave, n, f, fr);
    for (int i = 0; i < n; ++i) prod[i] = f[i] * sedmult[i];
    return pairwise_sum(prod, n) * normfac;                      /* :575-576 */
}

static int maxn(const mbbo_like *L)
{
    int m = L->nb;
    if (L->response_integrate)
        for (int b = 0; b < L->nb; ++b) {
            int n = L->offsets[b + 1] - L->offsets[b];
            if (n > m) m = n;
        }
    return m;
}

/* likelihood.py:790-834 __call__.  model_flux may be NULL. */
int mbbo_lnlike(const mbbo_like *L, const double *pars, double *lnl,
                double *model_flux, double *scratch /* >= 3*maxn + 2*nb */)
{
    const int nb = L->nb;
    for (int i = 0; i < 5; ++i)                                  /* :643-670 */
        if (pars[i] < L->lowlim[i]) { *lnl = -INFINITY; return MBBO_BELOW_LOWLIM; }
    mbbo_sed s;
    int st = mbbo_sed_init(&s, pars[0], pars[1], pars[2], pars[3], pars[4],
                           L->wavenorm, L->noalpha, L->opthin);  /* :754-768 */
    if (st) { *lnl = NAN; return st; }
    const int mn = maxn(L);
    double *mflux = scratch + 3 * (size_t)mn, *diff = mflux + nb;
    if (L->response_integrate) {                                 /* :813-815 */
        for (int b = 0; b < nb; ++b) {
            int o = L->offsets[b], n = L->offsets[b + 1] - o;
            mflux[b] = band_flux(&s, L->wave + o, L->sedmult + o,
                                 L->normfac[b], n, scratch);
        }
    } else {                                                     /* :817 */
        mbbo_sed_call(&s, L->wave, nb, mflux, scratch);
    }
    if (model_flux) memcpy(model_flux, mflux, sizeof(double) * nb);
    for (int b = 0; b < nb; ++b) diff[b] = L->flux[b] - mflux[b]; /* :821 */
    double r;
    if (L->invcov) {                                             /* :823 */
        double acc = 0.0;
        for (int i = 0; i < nb; ++i) {
            double t = 0.0;
            for (int j = 0; j < nb; ++j) t += L->invcov[i * nb + j] * diff[j];
            acc += diff[i] * t;
        }
        r = -0.5 * acc;
    } else {                                                     /* :825 */
        double *t = scratch;
        for (int b = 0; b < nb; ++b) t[b] = diff[b] * diff[b] * L->ivar[b];
        r = -0.5 * pairwise_sum(t, nb);
    }
    /* _uplim_prior, likelihood.py:672-717 */
    double pen = 0.0;
    for (int i = 0; i < 5; ++i)
        if (L->has_uplim[i] && pars[i] > L->uplim[i]) {
            double w = 0.02 * (L->uplim[i] - L->lowlim[i]);
            double d = pars[i] - L->uplim[i];
            pen -= 0.5 * d * d / (w * w);
        }
    double peak = NAN;
    if (L->has_uplim[5] || L->has_gprior[5]) {
        st = mbbo_max_wave(&s, &peak);
        if (st) { *lnl = NAN; return st; }
    }
    if (L->has_uplim[5] && peak > L->uplim[5]) {                 /* :710-715 */
        double w = 0.02 * L->uplim[5], d = peak - L->uplim[5];
        pen -= 0.5 * d * d / (w * w);
    }
    r += pen;
    /* _gprior, likelihood.py:719-752 */
    int any = 0;
    for (int i = 0; i < 6; ++i) any |= L->has_gprior[i];
    if (any) {
        double g = 0.0;
        for (int i = 0; i < 5; ++i)
            if (L->has_gprior[i]) {
                double d = pars[i] - L->gprior_mean[i];
                g -= 0.5 * L->gprior_ivar[i] * d * d;
            }
        if (L->has_gprior[5]) {
            double d = peak - L->gprior_mean[5];
            g -= 0.5 * L->gprior_ivar[5] * d * d;
        }
        r += g;
    }
    *lnl = r;
    return MBBO_OK;
}

/* Batch over walkers -- what emcee's map() does one row at a time
 * (mbb_fit.py:80-81).  nthreads <= 1: serial.  Returns the worst status. */
int mbbo_lnlike_batch(const mbbo_like *L, const double *pars, int n,
                      double *lnl, int32_t *status, double *model_flux,
                      int nthreads)
{
    const int mn = maxn(L);
    const size_t ns = 3 * (size_t)mn + 2 * (size_t)L->nb + 8;
    int worst = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(max : worst)
#endif
    {
        double *scratch = (double *)malloc(sizeof(double) * ns);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int i = 0; i < n; ++i) {
            int st = mbbo_lnlike(L, pars + 5 * (size_t)i, lnl + i,
                                 model_flux ? model_flux + (size_t)i * L->nb : NULL,
                                 scratch);
            if (status) status[i] = st;
            if (st > worst && st != MBBO_BELOW_LOWLIM) worst = st;
        }
        free(scratch);
    }
    return worst;
}

int mbbo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

size_t mbbo_sizeof_sed(void) { return sizeof(mbbo_sed); }
size_t mbbo_sizeof_like(void) { return sizeof(mbbo_like); }
