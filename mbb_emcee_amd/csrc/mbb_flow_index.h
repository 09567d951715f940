// mbb_flow_index.h -- index arithmetic of the one-launch look-ahead sampler run (k_lnlike SMODE 5),
// shared by the kernel and by the host-side model of its hand-over protocol
// (tests/test_host_cpu.py::test_flow_protocol_model, through the hooks in mbb_host_tables.cpp).
//
// Half-steps are numbered j = 0, 1, 2, ... over a launch; half h = j & 1 of the ensemble moves in
// half-step j (h = 0: rows [0, n/2), h = 1: the rest).  Everything a row publishes is filed under
// the number m = 1, 2, ... of the move it belongs to, in slot m mod kFlowSlots; slot 0 also holds
// what the launch found (m = 0).
#pragma once
#if defined(__HIPCC__)
#define MBB_FLOW_HD __host__ __device__
#else
#define MBB_FLOW_HD
#endif

constexpr int kFlowSlots = 4;      // slots per row
constexpr int kFlowLag = 4;        // a mover of half-step j waits until every move of j - kFlowLag is complete

// moves half h has completed before half-step j
MBB_FLOW_HD constexpr int flow_cnt(int h, int j) { return (j - h + 1) > 0 ? (j - h + 1) >> 1 : 0; }
// what a row's word says once its m-th move is published: the half-step of that move plus one
MBB_FLOW_HD constexpr int flow_seq(int h, int m) { return m > 0 ? h + 2 * m - 1 : 0; }

// ---- sampler form 7 (k_flowm, mbb_flowm.hip.h): the same numbering of half-steps and moves
constexpr int kFmSlots = 4;    // slots per row (moves filed mod this)
constexpr int kFmMseq = kFmSlots;    // decision words per row as laid out, the stride of mseq (a 128-byte line per row, 16,
                                     // is no faster: profiles/r04/done_counters_alignment.txt)
constexpr int kFmLag = 4;      // a workgroup at half-step j waits until every workgroup is through with j - kFmLag
constexpr int kFmRing = 8;     // completion counters, by half-step mod this (a power of two >= 2 kFmLag)
// (8 slots and a lag of 8 were tried: 6.49 against 6.29 us per step -- the lag guard is not what a half-step waits for)
constexpr int kFmNC = 3;       // C waves of a workgroup: wave c takes the half-steps j = c mod kFmNC
constexpr int kFmNB = 4;       // hand-over records in a workgroup's LDS: half-step j uses buffer j mod kFmNB
