"""A host-side model of the hand-over protocol of the one-launch look-ahead sampler run (k_lnlike
SMODE 5), used by tests/test_host_cpu.py.  It restates WHO waits for WHAT and
who reads and writes which slot -- with the kernel's own index arithmetic (mbb_flow_index.h, through
the C hook) -- and runs the actors in random or adversarial order, every publication delayed at
random.  Each slot remembers the number of the move whose data it holds; a read that finds another
number than the one it is entitled to is a violation: a slot reused under a reader.

Actors (n2 walkers per half, rows 0..n2-1 are half 0):
  mover(r, j)        r moves in half-step j = h, h+2, ...            (one per row, in order)
  worker(r, c, j)    prepares candidate c of r's move in half-step j  (one per row and candidate, in order)
A mover's completion is published later than its decision (the row's stores have to land), which is
what the workers of the row's old state and the lag guard look at.
"""
import ctypes as C
import random


class Index(object):
    def __init__(self, lib):
        lib.mbbh_flow_index.argtypes = [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
        self.lib = lib
        self.slots, self.lag = self._q(0, 0, 0)[2:]

    def _q(self, h, j, m):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.lib.mbbh_flow_index(h, j, m, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return a.value, b.value, c.value, d.value

    def cnt(self, h, j):
        return self._q(h, j, 0)[0]

    def seq(self, h, m):
        return self._q(h, 0, m)[1]


class Violation(Exception):
    pass


def run(ix, n2, nsteps, rng, guard=True, partner=None, stall=None, max_events=2000000):
    """Returns (largest lead of any mover over the slowest, events, all done).  partner(r, j) ->
    index in the other half; stall = (row, half-step, events): that row's candidate-0 worker does
    not run for that half-step until so many events have passed (an adversary: everything that
    depends on it must wait, and what it reads when it wakes up must still be there)."""
    S, LAG = ix.slots, ix.lag
    G = 2 * nsteps
    half_of = lambda r: 0 if r < n2 else 1
    rows = list(range(2 * n2))
    if partner is None:
        table = {}
        partner = lambda r, j: table.setdefault((r, j), rng.randrange(n2))
    prow = lambda r, j: (n2 if half_of(r) == 0 else 0) + partner(r, j)
    # published state: what each slot holds (move numbers), and the words
    st = {(r, s): (0 if s == 0 else None) for r in rows for s in range(S)}
    rec = {(r, s, c): None for r in rows for s in range(S) for c in (0, 1)}
    mseq = {(r, s): None for r in rows for s in range(S)}         # move number decided
    seq = {r: 0 for r in rows}
    done = {}                                                     # half-step -> completions published
    next_move = {r: half_of(r) for r in rows}                     # the half-step the row's mover does next
    next_work = {(r, c): half_of(r) for r in rows for c in (0, 1)}
    pending = []                                                  # delayed publications: (kind, ...)
    lead = 0

    def expect(store, key, want, what):
        if store[key] != want:
            raise Violation("%s %s holds move %s, reader wants %s" % (what, key, store[key], want))

    def mover_ready(r):
        j = next_move[r]
        if j >= G:
            return False
        h = half_of(r)
        m = ix.cnt(h, j) + 1
        if rec[(r, m % S, 0)] != m or rec[(r, m % S, 1)] != m:    # check words: only this move's record fits
            return False
        p, mp = prow(r, j), ix.cnt(1 - h, j)
        if mp > 0 and mseq[(p, mp % S)] != mp:
            if mseq[(p, mp % S)] is not None and mseq[(p, mp % S)] > mp:
                raise Violation("decision word of row %d overwritten under mover of row %d" % (p, r))
            return False
        if guard and j >= LAG and done.get(j - LAG, 0) < n2:
            return False
        return True

    def mover_step(r):
        j = next_move[r]
        h = half_of(r)
        m = ix.cnt(h, j) + 1
        expect(rec, (r, m % S, 0), m, "record"); expect(rec, (r, m % S, 1), m, "record")
        mseq[(r, m % S)] = m                                      # the decision, at once
        pending.append(("row", r, m, j))                          # the row and its word, later
        next_move[r] = j + 2

    def worker_ready(r, c):
        j = next_work[(r, c)]
        if j >= G or (stall and stall[:2] == (r, j) and c == 0 and events[0] < stall[2]):
            return False
        h = half_of(r)
        m_s, m_o, g = ix.cnt(h, j - 1), ix.cnt(1 - h, j - 1), j - 2
        p = prow(r, j)
        pp = (0 if h == 0 else n2) + partner(p, j - 1) if (c == 1 and j > 0) else None
        both = [r] + ([pp] if pp is not None else [])
        if m_o > 0 and seq[p] < ix.seq(1 - h, m_o):
            return False
        for x in both:
            if m_s > 1 and seq[x] < ix.seq(h, m_s - 1):
                return False
            if m_s > 0:
                m_q = ix.cnt(1 - h, g)
                if m_q > 0 and seq[prow(x, g)] < ix.seq(1 - h, m_q):   # the row x's last move was proposed from
                    return False
                if mseq[(x, m_s % S)] != m_s:                     # the decision of half-step j - 2
                    return False if mseq[(x, m_s % S)] is None or mseq[(x, m_s % S)] < m_s else _raise(
                        "decision word of row %d overwritten under its worker" % x)
        return True

    def _raise(msg):
        raise Violation(msg)

    def worker_step(r, c):
        j = next_work[(r, c)]
        h = half_of(r)
        m_s, m_o = ix.cnt(h, j - 1), ix.cnt(1 - h, j - 1)
        p = prow(r, j)
        pp = (0 if h == 0 else n2) + partner(p, j - 1) if (c == 1 and j > 0) else None
        expect(st, (p, m_o % S), m_o, "row of the partner")
        for x in [r] + ([pp] if pp is not None else []):
            expect(st, (x, (m_s - 1 if m_s > 0 else 0) % S), m_s - 1 if m_s > 0 else 0, "row as it was")
            if m_s > 0:
                m_q = ix.cnt(1 - h, j - 2)
                expect(st, (prow(x, j - 2), m_q % S), m_q, "row the last move was proposed from")
        rec[(r, (m_s + 1) % S, c)] = m_s + 1                      # element by element, no flag
        next_work[(r, c)] = j + 2

    events = [0]
    while events[0] < max_events:
        ready = ([("m", r) for r in rows if mover_ready(r)] +
                 [("w", r, c) for r in rows for c in (0, 1) if worker_ready(r, c)])
        # a wave's publications stay in the order it made them (a mover workgroup owns a row of
        # each half; a worker row owns one row and candidate)
        owner = lambda e: ("row", e[1] % n2)
        seen, pubs = set(), []
        for i, e in enumerate(pending):
            if owner(e) not in seen:
                seen.add(owner(e))
                pubs.append(("p", i))
        choices = ready + pubs
        if not choices:
            if stall and events[0] < stall[2]:
                events[0] = stall[2]                              # everything waits for the sleeper: wake it
                continue
            break
        ev = rng.choice(choices)
        events[0] += 1
        if ev[0] == "m":
            mover_step(ev[1])
        elif ev[0] == "w":
            worker_step(ev[1], ev[2])
        else:
            kind, a, b, j = pending.pop(ev[1])
            st[(a, b % S)] = b
            seq[a] = max(seq[a], j + 1)
            done[j] = done.get(j, 0) + 1
        fastest = max(next_move.values()) - 2
        slowest = min(next_move[r] - 2 for r in rows)
        lead = max(lead, fastest - slowest)
    finished = all(next_move[r] >= G for r in rows)
    return lead, events[0], finished
