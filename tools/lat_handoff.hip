// One-way latency of a flag hand-off between two waves on different CUs: wave A stores k to its
// word (write-through, agent scope), wave B polls A's word until it reads k and stores k to its
// own word, A polls that ...  Time of N round trips / 2N.  Pairs on different XCDs (workgroups
// 2j, 2j+1) or on the same XCD (b, b+8); polling one load at a time (with or without s_sleep) or
// with several loads in flight; alone on the chip or with `noise` other pairs doing the same.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lat_handoff tools/lat_handoff.hip && tools/lat_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long ldw(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int POLL>   // 0: load, check, s_sleep 1;  1: load, check;  2: four loads in flight, ~a quarter round trip apart
__device__ __forceinline__ bool wait_for(const unsigned long long *w, unsigned long long k)
{
    if (POLL == 2) {
        unsigned long long v0 = ldw(w);
        __builtin_amdgcn_s_sleep(2);
        unsigned long long v1 = ldw(w);
        __builtin_amdgcn_s_sleep(2);
        unsigned long long v2 = ldw(w);
        __builtin_amdgcn_s_sleep(2);
        unsigned long long v3 = ldw(w);
        for (int i = 0; i < (1 << 20); ++i) {
            if (v0 >= k) return true;
            v0 = ldw(w);
            if (v1 >= k) return true;
            v1 = ldw(w);
            if (v2 >= k) return true;
            v2 = ldw(w);
            if (v3 >= k) return true;
            v3 = ldw(w);
        }
        return false;
    }
    for (int i = 0; i < (1 << 22); ++i) {
        if (ldw(w) >= k) return true;
        if (POLL == 0) __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <int POLL>
__global__ void k_ping(unsigned long long *words, int n, int same_xcd, unsigned long long *ticks, int *err)
{
    const int b = blockIdx.x;
    int pair, side;
    if (same_xcd) { pair = (b / 16) * 8 + (b % 8); side = (b / 8) & 1; }
    else { pair = b >> 1; side = b & 1; }
    unsigned long long *mine = words + (size_t)(2 * pair + side) * 16, *other = words + (size_t)(2 * pair + (side ^ 1)) * 16;
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= n; ++i) {
        if (side == 0) {
            __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!wait_for<POLL>(other, (unsigned long long)i)) { atomicMax(err, 1); break; }
        } else {
            if (!wait_for<POLL>(other, (unsigned long long)i)) { atomicMax(err, 1); break; }
            __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (side == 0) ticks[pair] = __builtin_amdgcn_s_memrealtime() - t0;      // 100 MHz
}

int main()
{
    unsigned long long *words, *ticks;
    int *err;
    const int maxpairs = 128;
    hipMalloc(&words, maxpairs * 2 * 16 * 8);
    hipMalloc(&ticks, maxpairs * 8);
    hipMalloc(&err, 4);
    const int n = 2000;
    for (int same = 0; same < 2; ++same)
        for (int pairs : {1, 8, 64, 120})
            for (int poll = 0; poll < 3; ++poll) {
                if (same && pairs > 64) continue;
                hipMemset(words, 0, maxpairs * 2 * 16 * 8);
                hipMemset(err, 0, 4);
                const int grid = same ? ((pairs + 7) / 8) * 16 : 2 * pairs;
                if (poll == 0) hipLaunchKernelGGL(k_ping<0>, dim3(grid), dim3(64), 0, 0, words, n, same, ticks, err);
                if (poll == 1) hipLaunchKernelGGL(k_ping<1>, dim3(grid), dim3(64), 0, 0, words, n, same, ticks, err);
                if (poll == 2) hipLaunchKernelGGL(k_ping<2>, dim3(grid), dim3(64), 0, 0, words, n, same, ticks, err);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(maxpairs);
                int e;
                hipMemcpy(h.data(), ticks, pairs * 8, hipMemcpyDeviceToHost);
                hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.begin() + pairs);
                printf("%s XCD, %3d pairs at once, %s: one-way hand-off %.0f ns median, %.0f ns slowest pair%s\n",
                       same ? "same     " : "different", pairs,
                       poll == 0 ? "load-check-sleep  " : (poll == 1 ? "load-check        " : "four loads in flight"),
                       h[pairs / 2] * 10.0 / (2.0 * n), h[pairs - 1] * 10.0 / (2.0 * n), e ? "  (TIMED OUT)" : "");
            }
    return 0;
}
