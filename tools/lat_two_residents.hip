// Two kernels of ONE process that stay on the GPU at the same time, each on a stream of its own: does the second start while the
// first is resident?  (Two servers of two likelihoods side by side: profiles/r05/served_boundary.txt 16.)  By how many other
// streams the process makes BETWEEN the two (the runtime deals the streams of a priority class to four hardware queues in turn)
// and by the streams' priority classes.
// Every kernel is 128 workgroups x 1024 threads with 100 KB of LDS (a CU each), says "here" in a pinned word and spins until told
// to leave or for 20 ms at most.
//     hipcc --offload-arch=gfx950 -O3 -o tools/lat_two_residents tools/lat_two_residents.hip && tools/lat_two_residents
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <immintrin.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k_stay(volatile unsigned long long *here, const volatile unsigned long long *leave, unsigned long long id)
{
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) lds[0] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((unsigned long long *)here, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load((const unsigned long long *)leave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != id &&
               __builtin_amdgcn_s_memrealtime() - t0 < 2000000ull)          // (20 ms of the 100 MHz clock)
            __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    CHK(hipSetDevice(0));
    unsigned long long *h = nullptr, *d = nullptr;
    CHK(hipHostMalloc((void **)&h, 256, hipHostMallocMapped | hipHostMallocCoherent));
    for (int i = 0; i < 32; ++i) h[i] = 0;
    CHK(hipHostGetDevicePointer((void **)&d, h, 0));
    CHK(hipFuncSetAttribute((const void *)k_stay, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    int plo = 0, phi = 0;
    CHK(hipDeviceGetStreamPriorityRange(&plo, &phi));
    printf("stream priorities: least urgent %d, most urgent %d\n", plo, phi);
    printf("second resident kernel (128 workgroups, a CU each) launched while the first is resident: us until it says 'here' (20000+: not before the first left)\n");
    volatile unsigned long long *hv = h;
    unsigned long long id = 0;
    for (int between : {2, 3, 7}) {
        const int extra = 2;
        for (int mode = 0; mode < 6; ++mode) {            // 0: both default; 1 / 2: second high / low; 3: second made WithPriority(0); 4: FIRST high, second default; 5: both high
            std::vector<hipStream_t> others(extra);
            for (auto &s : others) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            // (the other streams have been used: a stream gets its hardware queue with its first work)
            for (auto &s : others) { hipLaunchKernelGGL(k_stay, dim3(1), dim3(64), 1024, s, d + 8, d + 9, 0ull); }
            for (auto &s : others) CHK(hipStreamSynchronize(s));
            hipStream_t s1, s2;
            if (mode >= 4) CHK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, phi));
            else CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
            // (... and `between` more made, and used, between the two that matter: if streams are dealt to the hardware queues in
            // turn, the second of the two lands on the first's queue when `between` + 1 is a multiple of their number)
            std::vector<hipStream_t> mids(between);
            for (auto &s : mids) { if (mode == 5) CHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, phi)); else CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
            for (auto &s : mids) { hipLaunchKernelGGL(k_stay, dim3(1), dim3(64), 1024, s, d + 8, d + 9, 0ull); }
            for (auto &s : mids) CHK(hipStreamSynchronize(s));
            if (mode == 0 || mode == 4) CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            else CHK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, mode == 1 || mode == 5 ? phi : (mode == 2 ? plo : 0)));
            double worst = 0.0;
            for (int rep = 0; rep < 5; ++rep) {
                const unsigned long long a = ++id, b = ++id;
                hv[0] = 0; hv[1] = 0; hv[2] = 0; hv[3] = 0;
                hipLaunchKernelGGL(k_stay, dim3(128), dim3(1024), 100 * 1024, s1, d + 0, d + 2, a);
                while (hv[0] != a) _mm_pause();
                const double t0 = now_us();
                hipLaunchKernelGGL(k_stay, dim3(128), dim3(1024), 100 * 1024, s2, d + 1, d + 3, b);
                double t = -1.0;
                while (now_us() - t0 < 30000.0) {
                    if (hv[1] == b) { t = now_us() - t0; break; }
                    _mm_pause();
                }
                hv[2] = a; hv[3] = b;
                _mm_sfence();
                CHK(hipStreamSynchronize(s1)); CHK(hipStreamSynchronize(s2));
                if (t < 0.0) t = 30000.0;
                if (t > worst) worst = t;
            }
            printf("  %d streams made between the two, second stream %-19s : %9.1f us at worst of 5\n", between,
                   mode == 0 ? "default priority" : mode == 1 ? "high priority" : mode == 2 ? "low priority" : mode == 3 ? "WithPriority(0)" : mode == 4 ? "default (1st high)" : "high (all high)", worst);
            CHK(hipStreamDestroy(s1)); CHK(hipStreamDestroy(s2));
            for (auto &s : mids) CHK(hipStreamDestroy(s));
            for (auto &s : others) CHK(hipStreamDestroy(s));
        }
    }
    return 0;
}
