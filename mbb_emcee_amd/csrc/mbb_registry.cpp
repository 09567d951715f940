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
#include <unistd.h>

#include <atomic>
#include <mutex>

namespace mbbh {
namespace {

constexpr uint32_t kMagic = 0x4d424231u;      // "MBB1"
constexpr int kSlots = 256, kKeys = 16;

struct Slot {
    std::atomic<int32_t> pid;                 // 0: free
    std::atomic<uint32_t> key[kKeys];         // 0: none
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
    char name[64] = {};
} g;

bool alive(int32_t pid)
{
    return pid > 0 && (kill(pid, 0) == 0 || errno != ESRCH);
}

void at_exit()
{
    // (no lock: the process is going; a slot left behind would be reclaimed by its dead pid anyway)
    if (g.tab && g.my >= 0 && g.pid == (int32_t)getpid()) {
        for (int i = 0; i < kKeys; ++i) g.tab->slot[g.my].key[i].store(0, std::memory_order_relaxed);
        g.tab->slot[g.my].pid.store(0, std::memory_order_release);
        g.tab->gen.fetch_add(1, std::memory_order_release);
    }
}

// g.mu held.  Maps the table (once per process) and claims a slot.
bool attach()
{
    const int32_t me = (int32_t)getpid();
    if (g.pid != me) {                        // first use, or a child of fork(): nothing of the parent's is ours
        g.pid = me; g.my = -1; g.gen_seen = ~0ull;
        memset(g.keys, 0, sizeof g.keys); memset(g.joins, 0, sizeof g.joins); memset(g.peers, 0, sizeof g.peers);
    }
    if (!g.tab) {
        if (g.tried) return false;
        g.tried = true;
        const char *forced = getenv("MBB_REGISTRY_NAME");       // (tests: a table of their own)
        if (forced && forced[0] == '/') snprintf(g.name, sizeof g.name, "%s", forced);
        else snprintf(g.name, sizeof g.name, "/mbb_hip_registry_%u", (unsigned)getuid());
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
        atexit(at_exit);
    }
    if (g.my >= 0 && g.tab->slot[g.my].pid.load(std::memory_order_acquire) == me) return true;
    g.my = -1;
    for (int pass = 0; pass < 2 && g.my < 0; ++pass)
        for (int i = 0; i < kSlots && g.my < 0; ++i) {
            int32_t cur = g.tab->slot[i].pid.load(std::memory_order_acquire);
            // first pass: free slots; second: slots of processes that are gone
            if (pass == 0 ? cur != 0 : alive(cur)) continue;
            if (g.tab->slot[i].pid.compare_exchange_strong(cur, me, std::memory_order_acq_rel)) {
                for (int k = 0; k < kKeys; ++k) g.tab->slot[i].key[k].store(0, std::memory_order_relaxed);
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
        if (!alive(pid)) {
            // gone without a word: its slot is free again
            if (g.tab->slot[s].pid.compare_exchange_strong(pid, 0, std::memory_order_acq_rel)) reclaimed = true;
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

}  // namespace mbbh

// ---- C hooks for the CPU tests (tests/test_host_cpu.py) ----------------------------------------------------------
extern "C" int mbbh_registry_join(uint32_t key) { return mbbh::registry_join(key); }
extern "C" int mbbh_registry_leave(uint32_t key) { return mbbh::registry_leave(key); }
extern "C" int mbbh_registry_peers(uint32_t key, int recount) { return mbbh::registry_peers(key, recount != 0); }
extern "C" const char *mbbh_registry_name(void) { return mbbh::registry_name(); }
