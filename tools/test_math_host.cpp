// Accuracy of mbb_math.hip.h compiled for the host against long double libm.
#define MBB_MATH_HOST
#include "../mbb_emcee_amd/csrc/mbb_math.hip.h"
#include <cstdio>
#include <cstdlib>
#include <random>
static double ulp_err(double got, long double ref) {
    if (ref == 0) return got == 0 ? 0 : 1e9;
    double r = (double)ref; int e; frexp(r, &e);
    long double u = ldexpl(1.0L, e - 53);
    return (double)(fabsl((long double)got - ref) / u);
}
int main() {
    std::mt19937_64 g(1);
    const int N = 4000000;
    double me = 0, mm = 0, ml = 0, md = 0, mm_small = 0, te = 0;
    std::uniform_real_distribution<double> U(0, 1);
    for (int i = 0; i < N; ++i) {
        double x = -700 + (709.7 + 700) * U(g);
        if (i % 3 == 0) x = -40 + 80 * U(g);
        if (i % 3 == 1) x = (U(g) - 0.5) * 2.0;
        double e = ulp_err(mbbm::m_exp(x), expl((long double)x));
        if (e > me) me = e;
        double m = ulp_err(mbbm::m_expm1(x), expm1l((long double)x));
        if (m > mm) mm = m;
        double e2 = ulp_err(mbbm::m_exp_t(x, mbbm::kExp2Tab), expl((long double)x));
        if (e2 > te) te = e2;
        // the piecewise polynomials' evaluation is tested with the tables themselves (tests/test_host_cpu.py::test_poly_tables_accuracy)
        double xs = ldexp(U(g) - 0.5, -(int)(U(g) * 60));
        double s = ulp_err(mbbm::m_expm1(xs), expm1l((long double)xs));
        if (s > mm_small) mm_small = s;
        double lx = exp(-30 + 60 * U(g));
        if (i % 2) lx = 0.5 + U(g);
        double l = ulp_err(mbbm::m_log(lx), logl((long double)lx));
        if (l > ml) ml = l;
        double a = exp(-20 + 40 * U(g)), b = exp(-30 + 110 * U(g));
        double d = ulp_err(mbbm::m_div(a, b), (long double)a / (long double)b);
        if (d > md) md = d;
    }
    printf("max ulp: exp %.3f expm1 %.3f expm1(small) %.3f log %.3f div %.3f\n", me, mm, mm_small, ml, md);
    printf("table : exp %.3f\n", te);
    // no clamp in front of the table exp: far outside the range it must still give 0 / inf, never NaN or a wrong sign
    printf("table edge: exp(800)=%g exp(-800)=%g exp(1e9)=%g exp(-1e9)=%g exp(1e20)=%g exp(-1e20)=%g exp(1e89)=%g exp(-1e89)=%g\n",
           mbbm::m_exp_t(800, mbbm::kExp2Tab), mbbm::m_exp_t(-800, mbbm::kExp2Tab), mbbm::m_exp_t(1e9, mbbm::kExp2Tab),
           mbbm::m_exp_t(-1e9, mbbm::kExp2Tab), mbbm::m_exp_t(1e20, mbbm::kExp2Tab), mbbm::m_exp_t(-1e20, mbbm::kExp2Tab),
           mbbm::m_exp_t(1e89, mbbm::kExp2Tab), mbbm::m_exp_t(-1e89, mbbm::kExp2Tab));
    printf("edge: exp(800)=%g exp(-800)=%g exp(inf)=%g exp(-inf)=%g expm1(710)=%g expm1(-800)=%g expm1(0)=%g div(1,inf)=%g exp(nan)=%g\n",
           mbbm::m_exp(800), mbbm::m_exp(-800), mbbm::m_exp(INFINITY), mbbm::m_exp(-INFINITY), mbbm::m_expm1(710.0),
           mbbm::m_expm1(-800), mbbm::m_expm1(0.0), mbbm::m_div(1.0, INFINITY), mbbm::m_exp(NAN));
    return 0;
}
