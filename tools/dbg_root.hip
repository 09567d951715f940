#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../mbb_emcee_amd/csrc/mbb_device.hip.h"
using namespace mbbd;
__global__ void k(double T, double beta, double lambda0, double alpha)
{
    double hcokt = kH * kC_um / (kK * T), x0 = hcokt / lambda0, lx0 = d_log(x0);
    double lo = 2.0 + alpha, hi = 3.0 + alpha + beta, x = 0.5 * (lo + hi);
    for (int it = 0; it < 12; ++it) {
        double dg, g = merge_g(x, alpha, beta, lx0, dg);
        double y = d_exp(beta * (d_log(x) - lx0));
        printf("it %d x=%.17g g=%.6e dg=%.6e y=%.6e lo=%.17g hi=%.17g\n", it, x, g, dg, y, lo, hi);
        if (g == 0.0) break;
        if (g < 0.0) lo = x; else hi = x;
        double xn = x - g / dg;
        if (!(xn > lo && xn < hi)) { xn = 0.5 * (lo + hi); printf("   bisect\n"); }
        x = xn;
    }
}
int main() {
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, 13.76405235, 1.49341579, 630.66745827, 3.61276087);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, 12.4, 1.9, 610.0, 3.1);
    hipDeviceSynchronize();
    return 0;
}
