// Per-path VALU instruction counts of one quadrature sample:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/cnt.s tools/count_sample_ops.hip
//   python tools/count_sample_ops.py /tmp/cnt.s
// Each kernel evaluates exactly one path between two markers; x and log x come in registers.
#include "../mbb_emcee_amd/csrc/mbb_device.hip.h"
using namespace mbbd;
#define PATH(NAME, EXPR)                                                                          \
    __global__ void NAME(const WalkerK *wk, const double *in, double *out, const Exp2Entry *e,    \
                         const double *b, const double *c)                                        \
    {                                                                                             \
        __shared__ Exp2Entry s_e[kExp2N];                                                         \
        __shared__ double s_b[4104], s_c[2568];                                                   \
        for (int i = threadIdx.x; i < kExp2N; i += 64) s_e[i] = e[i];                             \
        for (int i = threadIdx.x; i < 4104; i += 64) s_b[i] = b[i];                               \
        for (int i = threadIdx.x; i < 2568; i += 64) s_c[i] = c[i];                               \
        __syncthreads();                                                                          \
        const SampleTabs tabs = {s_e, s_b, s_c};                                                  \
        const WalkerK w = wk[0];                                                                  \
        double x = in[threadIdx.x], lx = in[64 + threadIdx.x];                                    \
        asm volatile("; MARK_BEGIN " #NAME : "+v"(x), "+v"(lx));                                  \
        double f = EXPR;                                                                          \
        asm volatile("; MARK_END " #NAME : "+v"(f));                                              \
        out[threadIdx.x] = f;                                                                     \
    }
PATH(bb_thick, fnu_bb_tab<false>(w, x, lx, &tabs))
PATH(bb_thin, fnu_bb_tab<true>(w, x, lx, &tabs))
PATH(wien, fnu_wien_tab(w, lx, &tabs))
PATH(bb_thick_far, ((x * m_exp_t<true, false>(-x, tabs.e)) * (x * x)) *
                       (m_exp_t<true, false>(fmin(w.beta * (lx - w.lx0), kLn40), tabs.e)))
PATH(exp_table, m_exp_t(x, tabs.e))
PATH(poly8, poly8_eval(tabs.b, x))
PATH(ref_thick_notab, (m_div(-m_expm1(-m_exp(w.beta * (lx - w.lx0))) * (x * x * x), m_expm1(x))))
