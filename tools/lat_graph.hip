// A synchronous call that is ONE kernel launch (the boundary's launch-per-call path: 2.9 us in the runtime's launch call, ~3 us
// from there to the first wave): does a captured graph of that one kernel get it to the GPU sooner than the launch call?
// The kernel writes its argument into a pinned word the host watches (as k_lnlike's watched launch does).
//     hipcc --offload-arch=gfx950 -O3 -o tools/lat_graph tools/lat_graph.hip && tools/lat_graph
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <immintrin.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { unsigned long long *flag; unsigned long long seq; double pad[56]; };   // (480 bytes, as the library's block)

__global__ void k_flag(const Args a)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void report(const char *what, std::vector<double> &call, std::vector<double> &total)
{
    std::sort(call.begin(), call.end()); std::sort(total.begin(), total.end());
    printf("  %-64s call p50 %6.2f us   call -> flag seen p50 %6.2f  p90 %6.2f\n", what, call[call.size() / 2], total[total.size() / 2], total[total.size() * 9 / 10]);
}

int main()
{
    CHK(hipSetDevice(0));
    unsigned long long *h_flag = nullptr, *d_flag = nullptr;
    CHK(hipHostMalloc((void **)&h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHK(hipHostGetDevicePointer((void **)&d_flag, h_flag, 0));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Args a{};
    a.flag = d_flag;
    const int N = 3000;
    volatile unsigned long long *hf = h_flag;
    printf("one kernel of 125 workgroups x 1024 threads that writes a pinned word; %d calls each, 2 us apart\n", N);
    for (int grid : {1, 125}) {
        // (a) the launch call
        {
            std::vector<double> call, total;
            for (int i = 1; i <= N; ++i) {
                a.seq = (unsigned long long)i;
                const double t0 = now_us();
                hipLaunchKernelGGL(k_flag, dim3(grid), dim3(1024), 0, st, a);
                const double t1 = now_us();
                while (*hf != a.seq) _mm_pause();
                const double t2 = now_us();
                if (i > 100) { call.push_back(t1 - t0); total.push_back(t2 - t0); }
                while (now_us() < t2 + 2.0) _mm_pause();
            }
            CHK(hipStreamSynchronize(st));
            char w[96]; snprintf(w, sizeof w, "hipLaunchKernelGGL, grid %d", grid);
            report(w, call, total);
        }
        // (b) a graph of that one kernel node, its argument replaced before every launch
        {
            hipGraph_t g; hipGraphExec_t ge; hipGraphNode_t node;
            CHK(hipGraphCreate(&g, 0));
            void *kargs[1] = {&a};
            hipKernelNodeParams kp{};
            kp.func = (void *)k_flag; kp.gridDim = dim3(grid); kp.blockDim = dim3(1024); kp.sharedMemBytes = 0; kp.kernelParams = kargs; kp.extra = nullptr;
            *hf = 0; a.seq = 0;
            CHK(hipGraphAddKernelNode(&node, g, nullptr, 0, &kp));
            CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            std::vector<double> call, total, setp;
            unsigned long long base = 1000000ull * (grid + 1);
            for (int i = 1; i <= N; ++i) {
                a.seq = base + i;
                const double t0 = now_us();
                CHK(hipGraphExecKernelNodeSetParams(ge, node, &kp));
                const double tm = now_us();
                CHK(hipGraphLaunch(ge, st));
                const double t1 = now_us();
                while (*hf != a.seq) _mm_pause();
                const double t2 = now_us();
                if (i > 100) { call.push_back(t1 - t0); total.push_back(t2 - t0); setp.push_back(tm - t0); }
                while (now_us() < t2 + 2.0) _mm_pause();
            }
            CHK(hipStreamSynchronize(st));
            std::sort(setp.begin(), setp.end());
            char w[96]; snprintf(w, sizeof w, "graph of one node (set params %.2f us + launch), grid %d", setp[setp.size() / 2], grid);
            report(w, call, total);
            // (c) the same graph launched as it is (the kernel would read its argument from memory the host writes)
            std::vector<double> call2, total2;
            unsigned long long last = a.seq;
            for (int i = 1; i <= N; ++i) {
                *hf = 0;
                const double t0 = now_us();
                CHK(hipGraphLaunch(ge, st));
                const double t1 = now_us();
                while (*hf != last) _mm_pause();
                const double t2 = now_us();
                if (i > 100) { call2.push_back(t1 - t0); total2.push_back(t2 - t0); }
                while (now_us() < t2 + 2.0) _mm_pause();
            }
            CHK(hipStreamSynchronize(st));
            snprintf(w, sizeof w, "the graph launched as it is, grid %d", grid);
            report(w, call2, total2);
            CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
        }
    }
    return 0;
}
