// Cost of letting two workgroups share one walker: each writes a few partial sums,
// agent-scope release, an atomic ticket; the second to arrive reads the other's sums.
// Compared with the same kernel without the exchange.  Pairs on the same XCD
// (blocks b, b+8) and on different XCDs (2j, 2j+1).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>   // 0: no exchange, 1: pair = (2j, 2j+1), 2: pair = (b, b+8); 3, 4: the same two
                      // pairings with the fence-free hand-off of MI355X_MICROARCH.md ("Valid forms":
                      // sc1 stores, every storing wave's vmcnt(0), one returning agent-scope add, the
                      // later arriver reads with sc1 loads)
__global__ void k(double *scratch, unsigned *ticket, double *out, unsigned long long *ticks, int work)
{
    const int b = blockIdx.x;
    int j, half;
    if (MODE == 2 || MODE == 4) { j = (b / 16) * 8 + (b % 8); half = (b / 8) & 1; }
    else { j = b >> 1; half = b & 1; }
    // some arithmetic standing in for the quadrature
    double acc = threadIdx.x * 1e-3 + b;
    for (int i = 0; i < work; ++i) acc = fma(acc, 1.0000001, 1e-9);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double total = acc;
    if (MODE >= 3) {
        if (threadIdx.x < 16)
            __hip_atomic_store(&scratch[(size_t)b * 16 + threadIdx.x], acc + threadIdx.x, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ unsigned old_s;
        if (threadIdx.x == 0)
            old_s = __hip_atomic_fetch_add(&ticket[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (old_s == 1) {
            const int other = (MODE == 4) ? (half ? b - 8 : b + 8) : (b ^ 1);
            double v = 0.0;
            if (threadIdx.x < 16)
                v = __hip_atomic_load(&scratch[(size_t)other * 16 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            total += v;
            if (threadIdx.x == 0) __hip_atomic_store(&ticket[j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 16) out[(size_t)j * 16 + threadIdx.x] = total;
        }
    } else if (MODE != 0) {
        if (threadIdx.x < 16) scratch[(size_t)b * 16 + threadIdx.x] = acc + threadIdx.x;
        __syncthreads();
        __shared__ unsigned old_s;
        if (threadIdx.x == 0) {
            __threadfence();
            old_s = __hip_atomic_fetch_add(&ticket[j], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
        }
        __syncthreads();
        if (old_s == 1) {                       // second to arrive: finish the walker
            const int other = (MODE == 2) ? (half ? b - 8 : b + 8) : (b ^ 1);
            double v = 0.0;
            if (threadIdx.x < 16)
                v = __hip_atomic_load(&scratch[(size_t)other * 16 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            total += v;
            if (threadIdx.x == 0) ticket[j] = 0;
            if (threadIdx.x < 16) out[(size_t)j * 16 + threadIdx.x] = total;
        }
    } else if (threadIdx.x < 16) out[(size_t)b * 16 + threadIdx.x] = total;
    if (threadIdx.x == 0) ticks[b] = __builtin_amdgcn_s_memtime() - t0;
}
template <int MODE> static void run(const char *name, int blocks, int work, double *s, unsigned *t, double *o, unsigned long long *tk)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, s, t, o, tk, work);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, s, t, o, tk, work);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, tk, 8 * blocks, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; double mean = 0; for (int i = 0; i < blocks; ++i) { if (h[i] > mx) mx = h[i]; mean += h[i]; }
    printf("%-34s work %5d: %.2f us per launch; exchange section %.0f ticks mean, %llu max\n", name, work, ms, mean / blocks, mx);
}
int main()
{
    double *s, *o; unsigned *t; unsigned long long *tk;
    hipMalloc(&s, 8 * 16 * 256); hipMalloc(&o, 8 * 16 * 256); hipMalloc(&t, 4 * 256); hipMalloc(&tk, 8 * 256);
    hipMemset(t, 0, 4 * 256);
    for (int work : {0, 2000}) {
        run<0>("no exchange, 250 blocks", 250, work, s, t, o, tk);
        run<1>("pairs (2j, 2j+1), different XCDs", 250, work, s, t, o, tk);
        run<2>("pairs (b, b+8), same XCD", 240, work, s, t, o, tk);
        run<3>("fence-free, pairs (2j, 2j+1)", 250, work, s, t, o, tk);
        run<4>("fence-free, pairs (b, b+8)", 240, work, s, t, o, tk);
    }
    return 0;
}
