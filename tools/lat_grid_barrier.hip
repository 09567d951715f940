// Cost of a grid-wide hand-off inside one persistent kernel, fence-free form
// (MI355X_MICROARCH.md "Valid forms", first table row): one lane per workgroup stores
// its 48-byte record with sc1 (write-through) stores, waits for them, adds to a counter
// (one per XCD); every workgroup polls the counters with sc1 loads until all have
// arrived, then reads another workgroup's record with sc1 loads and checks it.
// 125 workgroups of 1024 threads, one per CU, like the sampler's half-step.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lat_grid_barrier tools/lat_grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(1024) k(double *rows, unsigned int *counters, int iters, int nwg,
                                          unsigned long long *ticks, int *bad, int work)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    __shared__ double s_val[8];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double acc = tid * 1e-3;
    for (int it = 1; it <= iters; ++it) {
        for (int i = 0; i < work; ++i) acc = fma(acc, 1.0000001, 1e-9);      // stands in for the half-step
        if (tid == 0) {
            double *mine = rows + ((size_t)(it & 1) * nwg + b) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                __hip_atomic_store(mine + i, (double)(it * 1000 + b) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&counters[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 64) {                       // wave 0 polls: lane l < 8 watches XCD l's counter
            const unsigned target = (unsigned)it * (unsigned)nwg;
            for (;;) {
                unsigned v = 0;
                if (tid < 8) v = __hip_atomic_load(&counters[tid * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
                if (__builtin_amdgcn_readfirstlane(v) >= target) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        // read the record of another workgroup (as the next half-step reads its partner)
        if (tid < 6) {
            const int other = (b * 7 + it) % nwg;
            const double v = __hip_atomic_load(rows + ((size_t)(it & 1) * nwg + other) * 8 + tid, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
            if (v != (double)(it * 1000 + other) + tid) atomicAdd(bad, 1);
            s_val[tid] = v;
        }
        __syncthreads();
        acc += s_val[tid % 6] * 1e-30;
    }
    if (tid == 0) { ticks[b] = __builtin_amdgcn_s_memtime() - t0; rows[(size_t)2 * nwg * 8 + b] = acc; }
}
int main()
{
    const int nwg = 125, iters = 2000;
    double *rows; unsigned int *cnt; unsigned long long *tk; int *bad;
    hipMalloc(&rows, 8 * (2 * nwg * 8 + nwg)); hipMalloc(&cnt, 4 * 8 * 32); hipMalloc(&tk, 8 * nwg); hipMalloc(&bad, 4);
    for (int work : {0, 500, 2000}) {
        hipMemset(cnt, 0, 4 * 8 * 32); hipMemset(bad, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(1024), 0, 0, rows, cnt, iters, nwg, tk, bad, work);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("work %4d: %.3f us per iteration (%d iterations, %d workgroups), stale reads: %d\n", work,
               ms * 1e3 / iters, iters, nwg, hb);
    }
    return 0;
}
