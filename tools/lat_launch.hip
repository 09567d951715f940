// What a kernel boundary costs on this GPU, and what a grid-wide barrier inside a
// persistent kernel costs instead (125 workgroups, one per CU, bounded spins).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty(int *p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void gridbar(unsigned *cnt, unsigned *gen, int iters, unsigned long long *ticks, int *fail)
{
    const unsigned nb = gridDim.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned g = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                  // release this block's writes
            const unsigned arrived = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == nb - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gen, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                int spins = 0;
                while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000) { *fail = 1; break; }      // never hang the GPU
                }
            }
            __threadfence();
        }
        ++g;
        __syncthreads();
        if (*fail) break;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = __builtin_amdgcn_s_memtime() - t0;
}
int main()
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int blocks : {1, 125}) for (int thr : {64, 768}) {
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty, dim3(blocks), dim3(thr), 0, 0, nullptr);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty, dim3(blocks), dim3(thr), 0, 0, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel %3d x %3d threads, back to back: %.2f us per launch\n", blocks, thr, ms * 1e3 / 2000);
    }
    unsigned *cnt, *gen; unsigned long long *ticks; int *fail;
    hipMalloc(&cnt, 4); hipMalloc(&gen, 4); hipMalloc(&ticks, 8); hipMalloc(&fail, 4);
    for (int blocks : {125, 250}) {
        hipMemset(cnt, 0, 4); hipMemset(gen, 0, 4); hipMemset(fail, 0, 4);
        const int iters = 2000;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(gridbar, dim3(blocks), dim3(256), 0, 0, cnt, gen, iters, ticks, fail);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        int hf; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
        printf("grid barrier, %d workgroups: %.2f us per barrier%s\n", blocks, ms * 1e3 / iters, hf ? "  (SPIN LIMIT HIT)" : "");
    }
    return 0;
}
