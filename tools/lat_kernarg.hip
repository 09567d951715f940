// Does the size of a kernel's argument block cost launch time?  A chain of dependent
// launches (same stream) of a kernel that reads its first and last argument words, for
// argument blocks of 64 ... 1024 bytes; 125 workgroups of 1024 threads like the headline launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lat_kernarg tools/lat_kernarg.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Args { double *out; double v[N]; };
template <int N> __global__ void __launch_bounds__(1024) k(const Args<N> a)
{
    if (threadIdx.x == 0) a.out[blockIdx.x] = a.v[0] + a.v[N - 1];
}
template <int N> static void run(double *d)
{
    Args<N> a; a.out = d; for (int i = 0; i < N; ++i) a.v[i] = i;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<N>, dim3(125), dim3(1024), 0, 0, a);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k<N>, dim3(125), dim3(1024), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("argument block %4zu bytes: %.3f us per launch\n", sizeof(a), ms / 2.0);
}
int main()
{
    double *d; hipMalloc(&d, 8 * 256);
    for (int r = 0; r < 2; ++r) {
        run<7>(d); run<15>(d); run<23>(d); run<31>(d); run<39>(d); run<47>(d); run<55>(d); run<57>(d); run<59>(d);
        run<61>(d); run<62>(d); run<63>(d); run<64>(d); run<65>(d); run<71>(d); run<95>(d); run<127>(d);
    }
    return 0;
}
