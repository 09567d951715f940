// Ping-pong between the host and ONE resident wave, by where the two words live (VERDICT r04 item 7's A/B):
//   doorbell   D = device memory the host writes through the PCIe BAR   | H = pinned host memory the wave polls over PCIe
//   record     H = pinned host memory the wave writes over PCIe         | D = device memory the host polls through the BAR
// The round trip (host writes the doorbell -> host sees the record) with nothing computed in between is the floor of everything
// the served boundary spends outside its kernel.  Also: the same with 125 workgroups answering (the host scans 125 records).
//     hipcc --offload-arch=gfx950 -O3 -o tools/lat_doorbell tools/lat_doorbell.hip && tools/lat_doorbell
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <immintrin.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every workgroup's thread 0 polls the doorbell and answers with {seq, its own id} in its own 64-byte line; a doorbell of
// ~0 ends the kernel, and so does `limit` polls without a change (a wave that could spin for ever is not launched here)
// the shape of the answer: 1 = one 16-byte store, 2 = four lanes x 16 bytes (the whole line), 3 = eight lanes x 8 bytes (the whole line)
template <int SHAPE>
__global__ void k_echo_s(const volatile uint64_t *door, uint64_t *rec, long limit, int answering) {
    if (threadIdx.x >= 8) return;
    uint64_t seen = 0;
    long idle = 0;
    for (;;) {
        const uint64_t d = __hip_atomic_load((const uint64_t *)door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (d == ~0ull) break;
        if (d != seen) {
            seen = d;
            idle = 0;
            if ((int)blockIdx.x < answering) {
                uint64_t *r = rec + blockIdx.x * 8;
                typedef int v4i __attribute__((ext_vector_type(4)));
                v4i rec4;
                rec4.x = (int)d; rec4.y = (int)(d >> 32); rec4.z = (int)d; rec4.w = (int)(d >> 32);
                if (SHAPE == 1) {
                    if (threadIdx.x == 0) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(r), "v"(rec4) : "memory");
                } else if (SHAPE == 2) {
                    if (threadIdx.x < 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(r + 2 * threadIdx.x), "v"(rec4) : "memory");
                } else {
                    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(r + threadIdx.x), "v"(d) : "memory");
                }
            }
        } else if (++idle > limit) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

// thread 0 keeps FOUR polls of the doorbell in flight (a poll of fine-grained device memory takes ~1 us to come back: with one in
// flight a new request number is seen a whole latency and half a period after it lands, with four a latency and an eighth)
__global__ void k_echo_k4(const uint64_t *door, uint64_t *rec, long limit, int answering, int sleepy) {
    if (threadIdx.x != 0) return;
    uint64_t seen = 0, r0, r1, r2, r3;
    long idle = 0;
#define LD(r) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(r) : "v"(door) : "memory"); if (sleepy) __builtin_amdgcn_s_sleep(4);
#define WT(r) asm volatile("s_waitcnt vmcnt(3)" : "+v"(r) : : "memory");
    LD(r0) LD(r1) LD(r2) LD(r3)
    bool go = true;
    while (go) {
        uint64_t d;
#define STEP(r)                                                                                                              \
        WT(r) d = r;                                                                                                        \
        if (go && d == ~0ull) go = false;                                                                                   \
        if (go && d != seen) {                                                                                              \
            seen = d; idle = 0;                                                                                             \
            if ((int)blockIdx.x < answering) __hip_atomic_store(rec + blockIdx.x * 8, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
        } else if (++idle > limit) go = false;                                                                              \
        LD(r)
        STEP(r0) STEP(r1) STEP(r2) STEP(r3)
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
#undef LD
#undef WT
#undef STEP
}

// every workgroup polls ITS OWN 64-byte request line (lanes 0..3, 16 bytes each: one read of the line; the request number is the
// line's last word, the row would be the words before it) and answers with one 16-byte record in its own line
template <int SLEEP>
__global__ void k_echo_own(const uint64_t *req, uint64_t *rec, long limit, int answering) {
    if (threadIdx.x >= 64) return;
    typedef int v4i __attribute__((ext_vector_type(4)));
    const uint64_t *mine = req + blockIdx.x * 8 + 2 * (threadIdx.x & 3);
    uint64_t seen = 0;
    long idle = 0;
    for (;;) {
        v4i got;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(mine) : "memory");
        // (the request number sits in lane 3's upper half)
        const uint64_t mine_d = ((uint64_t)(unsigned)got.w << 32) | (unsigned)got.z;
        const uint64_t d = __builtin_amdgcn_readlane((int)(mine_d), 3) | ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(mine_d >> 32), 3) << 32);
        if (d == ~0ull) break;
        if (d != seen) {
            seen = d;
            idle = 0;
            // is the line ONE snapshot?  every row word must be of this request when the request number is
            const uint64_t lo = ((uint64_t)(unsigned)got.y << 32) | (unsigned)got.x;
            const int l = threadIdx.x & 3;
            const bool bad = threadIdx.x < 3 && (lo != d + 2 * l || (l < 2 && mine_d != d + 2 * l + 1));
            const bool torn = __builtin_amdgcn_ballot_w64(bad) != 0;
            if ((int)blockIdx.x < answering && threadIdx.x == 0) {
                v4i rec4;
                rec4.x = (int)d; rec4.y = (int)(d >> 32); rec4.z = torn ? 0 : got.x; rec4.w = torn ? 0 : got.y;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(rec + blockIdx.x * 8), "v"(rec4) : "memory");
            }
        } else if (++idle > limit) break;
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// ONE workgroup polls its request line in pinned host memory with FOUR reads in flight (lanes 0..7, 8 bytes each: one read of the
// line; the request number is the line's last word)
__global__ void k_echo_own4(const uint64_t *req, uint64_t *rec, long limit, int gap) {
    if (threadIdx.x >= 8) return;
    const uint64_t *mine = req + blockIdx.x * 8 + threadIdx.x;
    uint64_t seen = 0, r0, r1, r2, r3;
    long idle = 0;
#define LD(r) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(r) : "v"(mine) : "memory"); if (gap) __builtin_amdgcn_s_sleep(1);
#define WT(r) asm volatile("s_waitcnt vmcnt(3)" : "+v"(r) : : "memory");
    LD(r0) LD(r1) LD(r2) LD(r3)
    bool go = true;
    while (go) {
        uint64_t d;
#define STEP(r)                                                                                                              \
        WT(r)                                                                                                               \
        d = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)(r >> 32), 7) << 32) | (unsigned)__builtin_amdgcn_readlane((int)r, 7); \
        if (go && d == ~0ull) go = false;                                                                                   \
        if (go && d != seen) {                                                                                              \
            seen = d; idle = 0;                                                                                             \
            const bool bad = threadIdx.x < 5 && r != d + threadIdx.x;                                                       \
            const bool torn = __builtin_amdgcn_ballot_w64(bad) != 0;                                                        \
            if (threadIdx.x == 0) { __hip_atomic_store(rec + blockIdx.x * 8 + 1, torn ? 0 : d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  \
                                    __hip_atomic_store(rec + blockIdx.x * 8, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } \
        } else if (++idle > limit) go = false;                                                                              \
        LD(r)
        STEP(r0) STEP(r1) STEP(r2) STEP(r3)
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
#undef LD
#undef WT
#undef STEP
}

template <int SLEEP>
__global__ void k_echo_v(const volatile uint64_t *door, volatile uint64_t *rec, long limit, int stride, int answering) {
    if (threadIdx.x != 0) return;
    uint64_t seen = 0;
    long idle = 0;
    for (;;) {
        const uint64_t d = __hip_atomic_load((const uint64_t *)door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (d == ~0ull) break;
        if (d != seen) {
            seen = d;
            idle = 0;
            if ((int)blockIdx.x < answering)
                __hip_atomic_store((uint64_t *)rec + blockIdx.x * stride, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (++idle > limit) break;
        __builtin_amdgcn_s_sleep(SLEEP);
    }
}

__global__ void k_echo(const volatile uint64_t *door, volatile uint64_t *rec, long limit) {
    if (threadIdx.x != 0) return;
    uint64_t seen = 0;
    long idle = 0;
    for (;;) {
        const uint64_t d = __hip_atomic_load((const uint64_t *)door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (d == ~0ull) break;
        if (d != seen) {
            seen = d;
            idle = 0;
            __hip_atomic_store((uint64_t *)rec + blockIdx.x * 8, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (++idle > limit) break;
        __builtin_amdgcn_s_sleep(2);
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    CHK(hipSetDevice(0));
    uint64_t *d_door = nullptr, *d_rec = nullptr, *h_door = nullptr, *h_rec = nullptr, *hd_door = nullptr, *hd_rec = nullptr;
    const int maxwg = 256;
    CHK(hipExtMallocWithFlags((void **)&d_door, 64, hipDeviceMallocFinegrained));
    CHK(hipExtMallocWithFlags((void **)&d_rec, maxwg * 64, hipDeviceMallocFinegrained));
    CHK(hipHostMalloc((void **)&h_door, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHK(hipHostMalloc((void **)&h_rec, maxwg * 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHK(hipHostGetDevicePointer((void **)&hd_door, h_door, 0));
    CHK(hipHostGetDevicePointer((void **)&hd_rec, h_rec, 0));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("ping-pong host <-> one resident wave per workgroup; medians / p90 of 4000 round trips, us\n");
    for (int wgs : {1, 16}) {
        for (int dm = 0; dm < 2; ++dm) {
            for (int rm = 0; rm < 2; ++rm) {
                volatile uint64_t *door_host = dm == 0 ? d_door : h_door;       // (the host's view)
                uint64_t *door_dev = dm == 0 ? d_door : hd_door;
                volatile uint64_t *rec_host = rm == 0 ? h_rec : d_rec;
                uint64_t *rec_dev = rm == 0 ? hd_rec : d_rec;
                *door_host = 0;
                for (int i = 0; i < maxwg * 8; ++i) rec_host[i] = 0;
                _mm_sfence();
                CHK(hipStreamSynchronize(st));
                hipLaunchKernelGGL(k_echo, dim3(wgs), dim3(64), 0, st, door_dev, rec_dev, 4000000L);
                std::vector<double> t;
                bool lost = false;
                for (uint64_t s = 1; s <= 4200 && !lost; ++s) {
                    const double t0 = now_us();
                    *door_host = s;
                    _mm_sfence();
                    for (int w = 0; w < wgs; ++w) {
                        long spins = 0;
                        while (rec_host[w * 8] != s) {
                            _mm_pause();
                            if ((++spins & 1023) == 0 && now_us() - t0 > 2.0e6) { lost = true; break; }
                        }
                        if (lost) break;
                    }
                    const double t1 = now_us();
                    if (s > 200) t.push_back(t1 - t0);
                    // (a call of the boundary is not back to back with the next: Python runs in between)
                    const double until = t1 + 1.0;
                    while (now_us() < until) _mm_pause();
                }
                *door_host = ~0ull;
                _mm_sfence();
                CHK(hipStreamSynchronize(st));
                if (lost || t.empty()) { printf("  %3d workgroups  doorbell %s  record %s : no answer\n", wgs, dm ? "H" : "D", rm ? "D" : "H"); continue; }
                std::sort(t.begin(), t.end());
                printf("  %3d workgroups  doorbell in %-26s record in %-36s  p50 %6.2f  p90 %6.2f  min %6.2f\n", wgs,
                       dm ? "pinned host memory" : "device memory (BAR write)", rm ? "device memory (host reads the BAR)" : "pinned host memory (GPU writes)",
                       t[t.size() / 2], t[t.size() * 9 / 10], t[0]);
            }
        }
    }
    printf("\n125 workgroups (doorbell through the BAR, records in pinned host memory), by what the 125 do:\n");
    struct V { const char *what; int wgs, stride, answering, sleep; };
    const V vs[] = {{"1 workgroup", 1, 8, 1, 8},
                    {"125 poll, only the first answers", 125, 8, 1, 8},
                    {"125 poll, only the LAST answers", 125, 8, -1, 8},
                    {"125 answer, a 64-byte line each", 125, 8, 125, 8},
                    {"125 answer, 32 bytes apart", 125, 4, 125, 8},
                    {"125 answer, 16 bytes apart", 125, 2, 125, 8},
                    {"125 answer, 8 bytes apart", 125, 1, 125, 8},
                    {"125 answer, a line each, s_sleep 0", 125, 8, 125, 0},
                    {"125 answer, a line each, s_sleep 32", 125, 8, 125, 32},
                    {"125 answer, one 16-byte store", 125, 8, 125, 101},
                    {"125 answer, the line by 4 lanes x 16 B", 125, 8, 125, 102},
                    {"125 answer, the line by 8 lanes x 8 B", 125, 8, 125, 103},
                    {"125 answer, a line each; scan 8 us late", 125, 8, 125, 8},
                    {"125 answer; scan 8 us late, prefetch 16 ahead", 125, 8, 125, 8},
                    {"125 answer; scan 8 us late, prefetch 32 ahead", 125, 8, 125, 8},
                    {"125 answer, a line each, prefetch 16 ahead", 125, 8, 125, 8},
                    {"125 answer, a line each, prefetch 32 ahead", 125, 8, 125, 8},
                    {"1 workgroup, four polls in flight", 1, 8, 1, 204},
                    {"1 workgroup, four polls in flight, s_sleep 4", 1, 8, 1, 205},
                    {"125 answer, four polls in flight", 125, 8, 125, 204},
                    {"125 answer, four polls in flight, s_sleep 4", 125, 8, 125, 205},
                    {"256 poll, 125 answer, four in flight, sl 4", 256, 8, 125, 205},
                    {"256 poll, 125 answer, a line each", 256, 8, 125, 8},
                    {"256 answer, a line each", 256, 8, 256, 8}};
    for (const V &v : vs) {
        for (int i = 0; i < maxwg * 8; ++i) h_rec[i] = 0;
        *(volatile uint64_t *)d_door = 0;
        _mm_sfence();
        CHK(hipStreamSynchronize(st));
        const int answering = v.answering < 0 ? v.wgs : v.answering;
        const int first = v.answering < 0 ? v.wgs - 1 : 0;
        // (for "only the last answers" every workgroup but the last is refused by a rec pointer trick: answering = wgs and the
        //  host looks at the last record alone)
        const bool late = std::string(v.what).find("late") != std::string::npos;
        const int ahead = std::string(v.what).find("prefetch 16") != std::string::npos ? 16 : std::string(v.what).find("prefetch 32") != std::string::npos ? 32 : 0;
        if (v.sleep == 204 || v.sleep == 205) hipLaunchKernelGGL(k_echo_k4, dim3(v.wgs), dim3(64), 0, st, d_door, hd_rec, 4000000L, answering, v.sleep - 204);
        else if (v.sleep == 101) hipLaunchKernelGGL(k_echo_s<1>, dim3(v.wgs), dim3(64), 0, st, d_door, hd_rec, 4000000L, answering);
        else if (v.sleep == 102) hipLaunchKernelGGL(k_echo_s<2>, dim3(v.wgs), dim3(64), 0, st, d_door, hd_rec, 4000000L, answering);
        else if (v.sleep == 103) hipLaunchKernelGGL(k_echo_s<3>, dim3(v.wgs), dim3(64), 0, st, d_door, hd_rec, 4000000L, answering);
        else if (v.sleep == 0) hipLaunchKernelGGL(k_echo_v<0>, dim3(v.wgs), dim3(64), 0, st, d_door, hd_rec, 4000000L, v.stride, answering);
        else if (v.sleep == 32) hipLaunchKernelGGL(k_echo_v<32>, dim3(v.wgs), dim3(64), 0, st, d_door, hd_rec, 4000000L, v.stride, answering);
        else hipLaunchKernelGGL(k_echo_v<8>, dim3(v.wgs), dim3(64), 0, st, d_door, hd_rec, 4000000L, v.stride, answering);
        std::vector<double> t;
        bool lost = false;
        volatile uint64_t *rh = h_rec;
        for (uint64_t q = 1; q <= 4200 && !lost; ++q) {
            double t0 = now_us();
            *(volatile uint64_t *)d_door = q;
            _mm_sfence();
            if (late) {                                     // (every record has landed: what is timed is the host's scan alone)
                while (now_us() < t0 + 8.0) _mm_pause();
                t0 = now_us();
            }
            for (int w = first; w < answering; ++w) {
                long spins = 0;
                if (ahead && w == first) while (rh[w * v.stride] != q) _mm_pause();      // (prefetches only once the first record is in)
                if (ahead && w == first) for (int k = 1; k < ahead && w + k < answering; ++k) _mm_prefetch((const char *)(h_rec + (w + k) * v.stride), _MM_HINT_T0);
                if (ahead && w + ahead < answering) _mm_prefetch((const char *)(h_rec + (w + ahead) * v.stride), _MM_HINT_T0);
                while (rh[w * v.stride] != q) {
                    _mm_pause();
                    if ((++spins & 1023) == 0 && now_us() - t0 > 2.0e6) { lost = true; break; }
                }
                if (lost) break;
            }
            const double t1 = now_us();
            if (q > 200) t.push_back(t1 - t0);
            const double until = t1 + 1.0;
            while (now_us() < until) _mm_pause();
        }
        *(volatile uint64_t *)d_door = ~0ull;
        _mm_sfence();
        CHK(hipStreamSynchronize(st));
        if (lost || t.empty()) { printf("  %-40s no answer\n", v.what); continue; }
        std::sort(t.begin(), t.end());
        printf("  %-40s p50 %6.2f  p90 %6.2f  min %6.2f\n", v.what, t[t.size() / 2], t[t.size() * 9 / 10], t[0]);
    }
    printf("\nevery workgroup polls its own request line in pinned host memory (row + request number in one line), records as above:\n");
    uint64_t *h_req = nullptr, *hd_req = nullptr;
    CHK(hipHostMalloc((void **)&h_req, maxwg * 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHK(hipHostGetDevicePointer((void **)&hd_req, h_req, 0));
    for (int sleep : {0, 8, 100, 101}) for (int wgs : {1, 16, 125, 256}) {
        if (sleep >= 100 && wgs != 1) continue;
        for (int i = 0; i < maxwg * 8; ++i) { h_rec[i] = 0; h_req[i] = 0; }
        _mm_sfence();
        CHK(hipStreamSynchronize(st));
        const int answering = wgs > 125 ? 125 : wgs;
        if (sleep >= 100) hipLaunchKernelGGL(k_echo_own4, dim3(wgs), dim3(64), 0, st, hd_req, hd_rec, 8000000L, sleep - 100);
        else if (sleep) hipLaunchKernelGGL(k_echo_own<8>, dim3(wgs), dim3(64), 0, st, hd_req, hd_rec, 2000000L, answering);
        else hipLaunchKernelGGL(k_echo_own<0>, dim3(wgs), dim3(64), 0, st, hd_req, hd_rec, 2000000L, answering);
        std::vector<double> t;
        bool lost = false;
        volatile uint64_t *rh = h_rec;
        volatile uint64_t *qh = h_req;
        for (uint64_t q = 1; q <= 4200 && !lost; ++q) {
            const double t0 = now_us();
            for (int w = 0; w < answering; ++w) {               // (a row's five words and the request number behind them)
                for (int k = 0; k < 5; ++k) qh[w * 8 + k] = q + k;
                qh[w * 8 + 7] = q;
            }
            for (int w = 0; w < answering; ++w) {
                long spins = 0;
                while (rh[w * 8] != q) {
                    _mm_pause();
                    if ((++spins & 1023) == 0 && now_us() - t0 > 2.0e6) { lost = true; break; }
                }
                if (lost) break;
                if (rh[w * 8 + 1] != q) { printf("torn line: request %llu workgroup %d\n", (unsigned long long)q, w); lost = true; break; }
            }
            const double t1 = now_us();
            if (q > 200) t.push_back(t1 - t0);
            const double until = t1 + 1.0;
            while (now_us() < until) _mm_pause();
        }
        for (int w = 0; w < maxwg; ++w) qh[w * 8 + 7] = ~0ull;
        _mm_sfence();
        CHK(hipStreamSynchronize(st));
        if (lost || t.empty()) { printf("  %3d workgroups, s_sleep %d: no answer\n", wgs, sleep); continue; }
        std::sort(t.begin(), t.end());
        printf("  %3d workgroups (%3d answer), s_sleep %d   p50 %6.2f  p90 %6.2f  min %6.2f\n", wgs, answering, sleep, t[t.size() / 2], t[t.size() * 9 / 10], t[0]);
    }
    return 0;
}
