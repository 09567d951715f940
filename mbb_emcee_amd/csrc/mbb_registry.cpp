// mbb_registry.cpp -- see mbb_registry.h.
#include "mbb_registry.h"

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <mutex>

namespace mbbh {
namespace {

constexpr uint32_t kMagic = 0x4d424233u;      // "MBB3"
constexpr int kSlots = 256, kKeys = 16;

struct Slot {
    std::atomic<int32_t> pid;                 // 0: free
    std::atomic<uint32_t> key[kKeys];         // 0: none
    std::atomic<uint64_t> beat[kKeys];        // when the process last made a boundary call on that device (CLOCK_MONOTONIC, ms; 0: never)
    std::atomic<uint64_t> born;               // the process's start time (clock ticks since boot, /proc/<pid>/stat field 22;
                                              // 0: unknown): a pid handed out again to somebody else is not mistaken for the owner
};
struct Table {
    std::atomic<uint32_t> magic;
    std::atomic<uint64_t> gen;                // bumped at every join / leave / reclaim
    Slot slot[kSlots];
};
static_assert(std::atomic<int32_t>::is_always_lock_free && std::atomic<uint64_t>::is_always_lock_free, "plain words in shared memory");

struct Local {
    std::mutex mu;
    Table *tab = nullptr;
    bool tried = false;
    int32_t pid = 0;                          // the process this state belongs to (a fork()ed child starts afresh)
    int my = -1;                              // my slot
    uint32_t keys[kKeys] = {};
    int joins[kKeys] = {};
    uint64_t gen_seen = ~0ull;                // generation the cached counts are of
    int peers[kKeys] = {};
    uint64_t scan_ms[kKeys] = {};             // when the busy peers of a key were last counted, and of which generation
    uint64_t scan_gen[kKeys] = {};
    int busy[kKeys] = {};
    char name[64] = {};
} g;

// /proc/<pid>/stat field 22 (0: not to be had -- no /proc, another pid namespace)
uint64_t born_of(int32_t pid)
{
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%d/stat", (int)pid);
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return 0;
    const ssize_t n = read(fd, buf, sizeof buf - 1);
    close(fd);
    if (n <= 0) return 0;
    buf[n] = 0;
    const char *q = strrchr(buf, ')');          // (the command may hold spaces and brackets: fields are counted behind it)
    if (!q) return 0;
    int field = 2;
    for (++q; *q; ++q)
        if (*q == ' ' && ++field == 22) return strtoull(q + 1, nullptr, 10);
    return 0;
}

// Is the process that wrote the slot still there?  (Its pid answers, and -- where both are known -- was born when it said.)
bool alive(int32_t pid, uint64_t born)
{
    if (pid <= 0 || (kill(pid, 0) != 0 && errno == ESRCH)) return false;
    if (born == 0) return true;
    const uint64_t now = born_of(pid);
    return now == 0 || now == born;
}

// When the library goes (the process ends, or dlclose): the slot is given back.  A destructor of the library's own, not
// atexit(): a handler registered there would outlive an unloaded library.  (No lock: whoever still calls in is gone too; a
// slot left behind by a process that was killed is reclaimed by whoever counts next.)
__attribute__((destructor)) void registry_unload()
{
    if (g.tab && g.my >= 0 && g.pid == (int32_t)getpid()) {
        for (int i = 0; i < kKeys; ++i) { g.tab->slot[g.my].key[i].store(0, std::memory_order_relaxed); g.tab->slot[g.my].beat[i].store(0, std::memory_order_relaxed); }
        g.tab->slot[g.my].born.store(0, std::memory_order_relaxed);
        g.tab->slot[g.my].pid.store(0, std::memory_order_release);
        g.tab->gen.fetch_add(1, std::memory_order_release);
        g.my = -1;
    }
}

// A child of fork() is a process of its own.  getpid() is a system call with today's C libraries -- not something to make
// at every boundary call --, so the parent keeps a word on a page the kernel hands to a child EMPTY (MADV_WIPEONFORK): a
// look that finds it empty is a child's first.  (Where that cannot be had the pid is asked for every time.)
volatile int *g_mine = nullptr;
unsigned g_looks = 0;

// g.mu held.  Maps the table (once per process) and claims a slot.
bool attach()
{
    if (g_mine && *g_mine == 1 && g.tab && g.my >= 0 && (++g_looks & 1023u)) return true;      // (every call but one in 1024)
    const int32_t me = (int32_t)getpid();
    if (g.pid != me) {                        // first use, or a child of fork(): nothing of the parent's is ours
        g.pid = me; g.my = -1; g.gen_seen = ~0ull;
        memset(g.keys, 0, sizeof g.keys); memset(g.joins, 0, sizeof g.joins); memset(g.peers, 0, sizeof g.peers);
        memset(g.scan_ms, 0, sizeof g.scan_ms); memset(g.scan_gen, 0xff, sizeof g.scan_gen); memset(g.busy, 0, sizeof g.busy);
    }
    if (!g.tab) {
        if (g.tried) return false;
        g.tried = true;
        const char *forced = getenv("MBB_REGISTRY_NAME");       // (tests: a table of their own)
        if (forced && forced[0] == '/') snprintf(g.name, sizeof g.name, "%s", forced);
        else {
            // One table per user AND per PID namespace: whether a slot's owner is still there is asked of kill() and /proc,
            // which answer for the asker's namespace only -- two containers that share /dev/shm would take each other's
            // live processes for dead ones and claim their slots.  Processes of another namespace are simply
            // not in this table: they are "processes the library cannot see" and get what those get (the CUs a server
            // does not hold at once, all of them when its lease is up).
            struct stat ns;
            const unsigned long long ino = stat("/proc/self/ns/pid", &ns) == 0 ? (unsigned long long)ns.st_ino : 0ull;
            snprintf(g.name, sizeof g.name, "/mbb_hip_registry3_%u_%llx", (unsigned)getuid(), ino);
        }
        int fd = shm_open(g.name, O_RDWR | O_CREAT, 0600);
        if (fd < 0) return false;
        if (ftruncate(fd, (off_t)sizeof(Table)) != 0) { close(fd); return false; }
        void *p = mmap(nullptr, sizeof(Table), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return false;
        g.tab = static_cast<Table *>(p);      // (a fresh object is all zeroes: every slot free, generation 0)
        uint32_t none = 0;
        g.tab->magic.compare_exchange_strong(none, kMagic);
        if (g.tab->magic.load() != kMagic) { munmap(p, sizeof(Table)); g.tab = nullptr; return false; }
    }
    if (!g_mine) {
#ifdef MADV_WIPEONFORK
        void *pg = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (pg != MAP_FAILED) {
            if (madvise(pg, 4096, MADV_WIPEONFORK) == 0) g_mine = static_cast<volatile int *>(pg);
            else munmap(pg, 4096);
        }
#endif
    }
    if (g_mine) *g_mine = 1;
    if (g.my >= 0 && g.tab->slot[g.my].pid.load(std::memory_order_acquire) == me) return true;
    g.my = -1;
    for (int pass = 0; pass < 2 && g.my < 0; ++pass)
        for (int i = 0; i < kSlots && g.my < 0; ++i) {
            Slot &sl = g.tab->slot[i];
            int32_t cur = sl.pid.load(std::memory_order_acquire);
            // first pass: free slots; second: also slots of processes that are gone -- freed first (the birth time with
            // them), so that a slot never shows a live pid beside somebody else's birth time
            if (cur != 0) {
                if (pass == 0 || alive(cur, sl.born.load(std::memory_order_acquire))) continue;
                if (!sl.pid.compare_exchange_strong(cur, 0, std::memory_order_acq_rel)) continue;
                sl.born.store(0, std::memory_order_release);
            }
            int32_t none = 0;
            if (sl.pid.compare_exchange_strong(none, me, std::memory_order_acq_rel)) {
                for (int k = 0; k < kKeys; ++k) { sl.key[k].store(0, std::memory_order_relaxed); sl.beat[k].store(0, std::memory_order_relaxed); }
                sl.born.store(born_of(me), std::memory_order_release);
                g.my = i;
            }
        }
    return g.my >= 0;
}

int key_index(uint32_t key, bool make)
{
    for (int i = 0; i < kKeys; ++i)
        if (g.keys[i] == key) return i;
    if (make)
        for (int i = 0; i < kKeys; ++i)
            if (g.keys[i] == 0) { g.keys[i] = key; g.joins[i] = 0; return i; }
    return -1;
}

}  // namespace

const char *registry_name()
{
    std::lock_guard<std::mutex> lk(g.mu);
    attach();
    return g.name;
}

int registry_join(uint32_t key)
{
    if (key == 0) return 0;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!attach()) return 0;
    const int i = key_index(key, true);
    if (i < 0) return 0;                      // (more than kKeys devices: the rest go unseen)
    if (++g.joins[i] == 1) {
        g.tab->slot[g.my].beat[i].store(0, std::memory_order_relaxed);
        g.tab->slot[g.my].key[i].store(key, std::memory_order_release);
        g.tab->gen.fetch_add(1, std::memory_order_acq_rel);
    }
    return g.joins[i];
}

int registry_leave(uint32_t key)
{
    if (key == 0) return 0;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.tab || g.pid != (int32_t)getpid() || g.my < 0) return 0;
    const int i = key_index(key, false);
    if (i < 0 || g.joins[i] <= 0) return 0;
    if (--g.joins[i] == 0) {
        g.tab->slot[g.my].key[i].store(0, std::memory_order_release);
        g.tab->gen.fetch_add(1, std::memory_order_acq_rel);
        g.keys[i] = 0;
    }
    return g.joins[i];
}

int registry_peers(uint32_t key, bool recount)
{
    if (key == 0) return 0;
    // (an uncontended lock and one load of shared memory per boundary call: ~30 ns of a ~10 us call)
    std::lock_guard<std::mutex> lk(g.mu);
    if (!attach()) return 0;
    const int i = key_index(key, false);
    if (i < 0) return 0;
    const uint64_t gen = g.tab->gen.load(std::memory_order_acquire);
    if (!recount && gen == g.gen_seen) return g.peers[i];
    memset(g.peers, 0, sizeof g.peers);
    const int32_t me = (int32_t)getpid();
    bool reclaimed = false;
    for (int s = 0; s < kSlots; ++s) {
        int32_t pid = g.tab->slot[s].pid.load(std::memory_order_acquire);
        if (pid == 0 || pid == me) continue;
        if (!alive(pid, g.tab->slot[s].born.load(std::memory_order_acquire))) {
            // gone without a word: its slot is free again
            if (g.tab->slot[s].pid.compare_exchange_strong(pid, 0, std::memory_order_acq_rel)) {
                g.tab->slot[s].born.store(0, std::memory_order_release);     // (at worst a new owner's: then it counts by its pid alone)
                reclaimed = true;
            }
            continue;
        }
        for (int k = 0; k < kKeys; ++k) {
            const uint32_t theirs = g.tab->slot[s].key[k].load(std::memory_order_acquire);
            if (theirs == 0) continue;
            for (int m = 0; m < kKeys; ++m)
                if (g.keys[m] == theirs) ++g.peers[m];
        }
    }
    if (reclaimed) g.tab->gen.fetch_add(1, std::memory_order_acq_rel);
    g.gen_seen = gen;
    return g.peers[i];
}

int registry_busy(uint32_t key, uint64_t now_ms, uint64_t window_ms, bool recount)
{
    if (key == 0) return 0;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!attach()) return 0;
    const int i = key_index(key, false);
    if (i < 0) return 0;
    g.tab->slot[g.my].beat[i].store(now_ms ? now_ms : 1, std::memory_order_relaxed);       // this process's own call
    const uint64_t gen = g.tab->gen.load(std::memory_order_acquire);
    if (!recount && now_ms == g.scan_ms[i] && gen == g.scan_gen[i]) return g.busy[i];       // (counted this millisecond)
    // Memory reads only: a process that is gone stops calling, and that is all that is asked here (its slot is reclaimed
    // by registry_peers' recount or by the next process that needs one)
    const int32_t me = (int32_t)getpid();
    int n = 0;
    for (int s = 0; s < kSlots; ++s) {
        const int32_t pid = g.tab->slot[s].pid.load(std::memory_order_acquire);
        if (pid == 0 || pid == me) continue;
        for (int k = 0; k < kKeys; ++k) {
            if (g.tab->slot[s].key[k].load(std::memory_order_acquire) != key) continue;
            const uint64_t b = g.tab->slot[s].beat[k].load(std::memory_order_relaxed);
            if (b != 0 && b + window_ms >= now_ms) ++n;
        }
    }
    g.scan_ms[i] = now_ms; g.scan_gen[i] = gen; g.busy[i] = n;
    return n;
}

}  // namespace mbbh

// ---- C hooks for the CPU tests (tests/test_host_cpu.py) ----------------------------------------------------------
extern "C" int mbbh_registry_join(uint32_t key) { return mbbh::registry_join(key); }
extern "C" int mbbh_registry_leave(uint32_t key) { return mbbh::registry_leave(key); }
extern "C" int mbbh_registry_peers(uint32_t key, int recount) { return mbbh::registry_peers(key, recount != 0); }
extern "C" int mbbh_registry_busy(uint32_t key, unsigned long long now_ms, unsigned long long window_ms) { return mbbh::registry_busy(key, now_ms, window_ms, false); }
extern "C" int mbbh_registry_busy_now(uint32_t key, unsigned long long now_ms, unsigned long long window_ms) { return mbbh::registry_busy(key, now_ms, window_ms, true); }
// (tests / tools: what the per-call look costs, ns -- `iters` calls a microsecond of the clock apart)
extern "C" double mbbh_registry_busy_cost(uint32_t key, int iters)
{
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    int sink = 0;
    for (int i = 0; i < iters; ++i) {
        timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
        sink += mbbh::registry_busy(key, (uint64_t)t.tv_sec * 1000u + (uint64_t)(t.tv_nsec / 1000000), 250, false);
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    return ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec) + (sink < 0 ? 1 : 0)) / (iters > 0 ? iters : 1);
}
extern "C" const char *mbbh_registry_name(void) { return mbbh::registry_name(); }
