// mbb_registry.h -- which processes of this library are on which GPU (host-only, no HIP types).
//
// Why: the served boundary (k_serve, mbb_serve.hip.h) keeps a workgroup on a CU of its own, with most of the CU's LDS,
// between a sampler's calls.  Inside one process that is arbitrated by yield_server (mbb_hip.hip).  Across
// processes -- emcee's pool, reference mbb_fit.py:80-81 with threads > 1: the likelihood pickled into workers that
// share the GPU -- nothing of another process fits on the CUs a server holds: with a server on every CU, measured, one
// worker's call waited 42 ms for the other's whole loop (profiles/r05/pool_two_processes_before.txt).  So a server is as
// wide as its process's calls have rows and no wider than the process's share of the device: the CUs divided by the
// processes that are making boundary calls on it (registry_busy).
//
// How: a small table in POSIX shared memory (/dev/shm/mbb_hip_registry3_<uid>_<pid namespace>), one slot per process: its pid and
// the keys (PCI domain:bus:device -- not the HIP ordinal, which HIP_VISIBLE_DEVICES renumbers) of the devices it
// holds contexts on; a generation word, bumped at every change, makes the per-call check one load of shared memory.
// A slot whose process is gone (kill(pid, 0) == ESRCH, or the pid belongs to a process born at another time than the slot
// says: /proc/<pid>/stat) is reclaimed by whoever counts next; a count that says
// "somebody else" is made again every so often, so a peer that died without a word is not believed for long.
// Processes that do not share /dev/shm (other containers) or do not use this library are not seen: for those the
// server's lease (option "serve_lease_us") bounds how long it holds the GPU in one go.
// Compiled into libmbb_hip.so by hipcc and, on its own, by gcc for the CPU tests (`make -C oracle
// libmbb_hosttables.so`, tests/test_host_cpu.py::test_device_registry_*).
#pragma once
#include <stdint.h>

namespace mbbh {

// Joins / leaves are counted per key inside the process: the key is published on the first join and withdrawn on
// the last leave.  All three return 0 (or the count) when the registry cannot be had (no /dev/shm): nobody is seen.
int registry_join(uint32_t key);
int registry_leave(uint32_t key);
// Other live processes registered on `key`.  `recount`: do not trust the cached answer.
int registry_peers(uint32_t key, bool recount);
// Other processes registered on `key` that made a boundary call on it within the last `window_ms` (they say so here, with
// `now_ms` of one clock all processes read: CLOCK_MONOTONIC in ms); also notes this process's own call.  Counted at most once
// per millisecond unless `recount` (a process about to start a server looks every time: workers of a pool begin together),
// from memory alone.  A process that holds a context but is not calling (a pool's parent) is not in the way.
int registry_busy(uint32_t key, uint64_t now_ms, uint64_t window_ms, bool recount);
// (tests) the name of the shared-memory object this process uses
const char *registry_name();

}  // namespace mbbh
