// Single-lane latency of the prologue's building blocks (s_memtime), MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../mbb_emcee_amd/csrc/mbb_device.hip.h"
using namespace mbbd;
#define T0 unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define LAP(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long t1 = __builtin_amdgcn_s_memtime(); out[i] = (double)(t1 - t0); t0 = __builtin_amdgcn_s_memtime(); }
__global__ void k(const double *pars, double *out, double *sink)
{
    if (threadIdx.x != 0) return;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double T = pars[0], beta = pars[1], lambda0 = pars[2], alpha = pars[3], fnorm = pars[4];
    double acc = T + beta + lambda0 + alpha + fnorm;
    asm volatile("" ::"v"(acc));
    LAP(0)                                    // global load of the row
    double lT = m_log(T);            asm volatile("" ::"v"(lT));   LAP(1)
    double e1 = m_exp(-lT);          asm volatile("" ::"v"(e1));   LAP(2)
    double e2 = m_expm1(lT);         asm volatile("" ::"v"(e2));   LAP(3)
    double d1 = T / beta;            asm volatile("" ::"v"(d1));   LAP(4)
    double d2 = m_div(T, beta);      asm volatile("" ::"v"(d2));   LAP(5)
    const double hcokt = kH * kC_um / (kK * T);
    const double lhokt9 = kLog1e9HoK - lT;
    const double lx0 = lhokt9 + kLogUmToGHz - m_log(lambda0);
    asm volatile("" ::"v"(lx0));                                   LAP(6)
    const float fulo = __logf((float)(2.0 + alpha)) - 1e-5f, fuhi = __logf((float)(3.0 + alpha + beta)) + 1e-5f;
    float up = thick_merge_root_f32((float)alpha, (float)beta, (float)lx0, fulo, fuhi, __logf((float)(2.5 + alpha + 0.5 * beta)));
    asm volatile("" ::"v"(up));                                    LAP(7)
    int st; double xm, ym; int it = 0;
    double um = thick_merge_root(alpha, beta, lx0, st, xm, ym, &it);
    asm volatile("" ::"v"(um), "v"(xm), "v"(ym));                  LAP(8)
    double kappa = m_exp((3.0 + alpha) * um) * -m_expm1(-ym) / m_expm1(xm);
    asm volatile("" ::"v"(kappa));                                 LAP(9)
    SedScalars s;
    int st2 = sed_prologue<false, false>(T, beta, lambda0, alpha, fnorm, 500.0, 6.396, s);
    asm volatile("" ::"v"(s.normfac), "v"(st2));                   LAP(10)
    double pk = sed_peak_wave<false>(T, beta, lx0, hcokt, st);
    asm volatile("" ::"v"(pk));                                    LAP(11)
    sink[0] = acc + lT + e1 + e2 + d1 + d2 + lx0 + up + um + kappa + s.normfac + pk + it;
}
int main() {
    double h[5] = {13.76405235, 1.49341579, 630.66745827, 3.61276087, 42.99353427}, *d, *o, *s;
    hipMalloc(&d, 40); hipMalloc(&o, 16 * 8); hipMalloc(&s, 8);
    hipMemcpy(d, h, 40, hipMemcpyHostToDevice);
    double r[16];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, s);
        hipDeviceSynchronize();
    }
    hipMemcpy(r, o, 16 * 8, hipMemcpyDeviceToHost);
    const char *nm[] = {"row load", "m_log", "m_exp", "m_expm1", "IEEE div", "m_div", "lx0 (div+log)", "fp32 presolve",
                        "presolve + fp64 finish", "kappa", "whole sed_prologue", "peak (max_wave)"};
    for (int i = 0; i < 12; ++i) printf("%-22s %6.0f cycles\n", nm[i], r[i]);
    return 0;
}
