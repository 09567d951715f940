// Calibrates s_memtime against HIP events and measures the dependent-issue latency of
// the VALU operations the prologue is built from (one wave on an otherwise idle CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long ticks, unsigned long long *out)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t;
    do { t = __builtin_amdgcn_s_memtime(); } while (t - t0 < ticks);
    out[0] = t - t0;
}
template <int KIND>
__global__ void chain(double *out, double seed, int n)
{
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
    float fa = (float)a, fb = 1.0000001f, fc = 1e-9f, f2 = fa + 1, f3 = fa + 2, f4 = fa + 3;
    double b2 = a + 1, b3 = a + 2, b4 = a + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));
            if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(fa));
            if (KIND == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));
            if (KIND == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
            if (KIND == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(fa));
            if (KIND == 7) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a));
            if (KIND == 8) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a) : "v"(i));
            if (KIND == 9) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                             asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b2) : "v"(b), "v"(c)); }
            if (KIND == 10) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                              asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b2) : "v"(b), "v"(c));
                              asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b3) : "v"(b), "v"(c));
                              asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b4) : "v"(b), "v"(c)); }
            if (KIND == 11) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));
                              asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(fb), "v"(fc));
                              asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(fb), "v"(fc));
                              asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f4) : "v"(fb), "v"(fc)); }
            if (KIND == 12) { asm volatile("v_exp_f32 %0, %0" : "+v"(fa)); asm volatile("v_exp_f32 %0, %0" : "+v"(f2));
                              asm volatile("v_exp_f32 %0, %0" : "+v"(f3)); asm volatile("v_exp_f32 %0, %0" : "+v"(f4)); }
        }
    }
    asm volatile("" ::"v"(a), "v"(fa), "v"(b2), "v"(b3), "v"(b4), "v"(f2), "v"(f3), "v"(f4));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = (double)(t1 - t0) / (16.0 * n); out[1] = a + fa; }
}
int main()
{
    unsigned long long *d; double *o; hipMalloc(&d, 8); hipMalloc(&o, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 100000000ull, d);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
    printf("s_memtime: %llu ticks in %.3f ms -> %.1f MHz\n", t, ms, t / ms / 1e3);
    const char *nm[] = {"v_fma_f64", "v_fma_f32", "v_exp_f32", "v_rcp_f64", "v_mul_f64", "v_add_f64", "v_rcp_f32", "v_ldexp_f64", "v_cvt_f64_i32", "2x fma_f64 (per pair)", "4x fma_f64 (per quad)", "4x fma_f32 (per quad)", "4x exp_f32 (per quad)"};
    double h[2];
#define RUN(K) hipLaunchKernelGGL(chain<K>, dim3(1), dim3(64), 0, 0, o, 1.0, 1000); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(chain<K>, dim3(1), dim3(64), 0, 0, o, 1.0, 1000); hipDeviceSynchronize(); \
    hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); printf("%-22s dependent: %.2f ticks/op\n", nm[K], h[0]);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
    return 0;
}
