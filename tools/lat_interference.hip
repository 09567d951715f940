// How much does a single latency-bound wave (a dependent fp64 chain, like the prologue)
// slow down when the other waves of its workgroup are busy?  Wave 0 times 2000 dependent
// FMAs; the other 15 waves do nothing, or run FMA / LDS / global-load loops, on the other
// SIMDs only (wave & 3 != 0) or on all of them.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE, bool ALL, bool STRAIGHT, int KIND = 0>
__global__ void __launch_bounds__(1024) k(double *out, const double *g, int n)
{
    __shared__ double lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = i * 1e-3;
    __syncthreads();
    if (wave == 0) {
        double a = 1.0 + lane * 1e-9;
        const double b = 1.0000001, c = 1e-9;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (KIND == 1) {                // VALU + a scalar move per step (constants are materialised like this)
            for (int i = 0; i < 125; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    unsigned sc; asm volatile("s_mov_b32 %0, 0x3ff00000" : "=s"(sc));
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                    asm volatile("" ::"s"(sc));
                }
            }
        } else if (KIND == 2) {         // DPP row broadcast pair per step
            for (int i = 0; i < 125; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    int lo = __double2loint(a), hi = __double2hiint(a);
                    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150, 0xf, 0xf, false);
                    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150, 0xf, 0xf, false);
                    a = __hiloint2double(hi, lo);
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                }
            }
        } else if (KIND == 3) {         // fp32 transcendental chain
            float f = 1.0f + lane * 1e-6f;
            for (int i = 0; i < 125; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) { asm volatile("v_exp_f32 %0, %0" : "+v"(f)); asm volatile("v_rcp_f32 %0, %0" : "+v"(f)); }
            }
            a += f;
        } else if (KIND == 4) {         // select / convert / ldexp mix as in the exp reduction
            for (int i = 0; i < 125; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    asm volatile("v_rndne_f64 %0, %0" : "+v"(a));
                    asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a));
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                }
            }
        } else if (STRAIGHT) {                 // 2000 instructions of straight-line code: every one is fetched
#pragma unroll
            for (int j = 0; j < 2000; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
        } else {
            for (int i = 0; i < 125; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            }
        }
        asm volatile("" ::"v"(a));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[blockIdx.x * 2] = (double)(t1 - t0) / 2000.0; out[blockIdx.x * 2 + 1] = a; }
    } else if (ALL || (wave & 3) != 0) {
        double a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, s = 0.0;
        const double b = 1.0000001, c = 1e-9;
        unsigned idx = (lane * 37 + wave * 101) & 4095;
        for (int i = 0; i < n; ++i) {
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s += lds[idx]; idx = (idx * 5 + 1) & 4095; }
            } else if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { s += g[(idx + 4096 * j + (i & 7) * 16384) & 0xfffff]; idx = (idx * 5 + 1) & 4095; }
            } else if (MODE == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { asm volatile("v_rcp_f64 %0, %0" : "+v"(a0)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); }
            }
        }
        if (a0 + a1 + a2 + a3 + s == 123.456) out[0] = s;
    }
}
template <int MODE, bool ALL, bool STRAIGHT = false, int KIND = 0> static void run(const char *name, double *o, const double *g, int n)
{
    hipLaunchKernelGGL((k<MODE, ALL, STRAIGHT, KIND>), dim3(125), dim3(1024), 0, 0, o, g, n); hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, ALL, STRAIGHT, KIND>), dim3(125), dim3(1024), 0, 0, o, g, n); hipDeviceSynchronize();
    double h[250]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 125; ++i) m += h[2 * i];
    printf("%-58s %.2f ticks per dependent FMA in wave 0\n", name, m / 125);
}
int main()
{
    double *o, *g; hipMalloc(&o, 8 * 250); hipMalloc(&g, 8 << 20); hipMemset(g, 0, 8 << 20);
    run<0, false>("other waves idle", o, g, 0);
    run<1, false>("fp64 FMA loops on the other three SIMDs", o, g, 400);
    run<1, true>("fp64 FMA loops on all SIMDs (3 waves share wave 0's)", o, g, 400);
    run<2, false>("LDS reads on the other three SIMDs", o, g, 400);
    run<3, false>("global loads on the other three SIMDs", o, g, 100);
    run<4, false>("v_rcp_f64 + FMA on the other three SIMDs", o, g, 400);
    run<0, false, true>("straight-line chain, other waves idle", o, g, 0);
    run<1, false, true>("straight-line chain, FMA loops on the other three SIMDs", o, g, 400);
    run<1, true, true>("straight-line chain, FMA loops on all SIMDs", o, g, 400);
    run<2, false, true>("straight-line chain, LDS reads on the other three SIMDs", o, g, 400);
    run<0, false, false, 1>("FMA + s_mov per step, idle", o, g, 0);
    run<1, false, false, 1>("FMA + s_mov per step, FMA loops on the other SIMDs", o, g, 600);
    run<1, true, false, 1>("FMA + s_mov per step, FMA loops on all SIMDs", o, g, 600);
    run<0, false, false, 2>("DPP pair + FMA per step, idle", o, g, 0);
    run<1, false, false, 2>("DPP pair + FMA per step, FMA loops on the other SIMDs", o, g, 900);
    run<2, false, false, 2>("DPP pair + FMA per step, LDS reads on the other SIMDs", o, g, 900);
    run<0, false, false, 3>("v_exp_f32 + v_rcp_f32 chain, idle", o, g, 0);
    run<4, false, false, 3>("v_exp_f32 + v_rcp_f32 chain, v_rcp_f64+FMA on the other SIMDs", o, g, 900);
    run<1, false, false, 3>("v_exp_f32 + v_rcp_f32 chain, FMA loops on the other SIMDs", o, g, 900);
    run<0, false, false, 4>("rndne/ldexp/FMA chain, idle", o, g, 0);
    run<1, false, false, 4>("rndne/ldexp/FMA chain, FMA loops on the other SIMDs", o, g, 900);
    return 0;
}
