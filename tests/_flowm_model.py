"""A host-side model of the hand-over protocol of sampler form 7 (k_flowm, mbb_flowm.hip.h; DESIGN.md
section 9, "Every stage ahead"), used by tests/test_host_cpu.py.  Like _flow_model.py for SMODE 5 it
restates WHO waits for WHAT and who reads and writes which slot, with the kernel's own index arithmetic
(mbb_flow_index.h through the C hooks), and runs the actors in random or adversarial order with every
store to memory landing at a random later time.  Every slot remembers the move whose data it holds --
what the check words say in the kernel -- and a reader that finds a LATER move than the one it is entitled
to would spin for ever there: a violation, a slot reused under a reader.

Actors, per workgroup (pair w, candidate c):
  C(j)   prepares the proposal of row (half j & 1, w) for half-step j; three waves, wave j mod kFmNC,
         each in order; needs what is settled once j - 3 is decided, then the two decisions of j - 2,
         then a free hand-over record (E through with j - kFmNB)
  E(j)   two waves, one per half, each in order: needs C(j) (the quadrature is inside the workgroup), the
         partner's decision of j - 1, the row's lnprob after its last move, the lag guard; publishes
         the move if its candidate is the one the partner's decision selects
"""
import ctypes as C
import random
from _flow_model import Index, Violation


def consts(lib):
    v = [C.c_int() for _ in range(5)]
    lib.mbbh_flowm_consts.argtypes = [C.POINTER(C.c_int)] * 5
    lib.mbbh_flowm_consts(*[C.byref(x) for x in v])
    return [x.value for x in v]          # slots, lag, ring, C waves, LDS records


def run(ix, lib, n2, nsteps, rng, guard=True, partner=None, stall=None, max_events=4000000):
    """Returns (largest lead of any E over the slowest, events, all done).  stall = (workgroup, half-step,
    events): that workgroup's C does not take that half-step until so many events have passed."""
    S, LAG, RING, NC, NB = consts(lib)
    G = 2 * nsteps
    half_of = lambda r: 0 if r < n2 else 1
    if partner is None:
        table = {}
        partner = lambda r, j: table.setdefault((r, j), rng.randrange(n2))
    prow = lambda r, j: (n2 if half_of(r) == 0 else 0) + partner(r, j)
    rows = list(range(2 * n2))
    wgs = [(w, c) for w in range(n2) for c in (0, 1)]
    row_of = lambda w, j: (n2 if (j & 1) else 0) + w
    # memory: what each slot holds
    prop = {(r, s, c): None for r in rows for s in range(S) for c in (0, 1)}
    row = {(r, s): (0 if s == 0 else None) for r in rows for s in range(S)}
    mseq = {(r, s): None for r in rows for s in range(S)}          # (move, accepted)
    done = {}
    pending = []                                                   # stores on their way: (store, key, value)
    next_c = {(g, k): k for g in wgs for k in range(NC)}           # next half-step of C wave k
    c_done = {g: set() for g in wgs}                               # half-steps whose record is handed over
    next_e = {(g, h): h for g in wgs for h in (0, 1)}
    e_done = {g: set() for g in wgs}
    events = [0]
    lead = 0

    def have(store, key, want, what):
        """True if the slot holds exactly what the reader wants; a later move is a violation"""
        got = store[key]
        gm = got[0] if isinstance(got, tuple) else got
        if gm is not None and gm > want:
            raise Violation("%s %s holds move %s, reader wants %s" % (what, key, gm, want))
        return gm == want

    def c_needs(g, j, check):
        w, c = g
        h = j & 1
        r = row_of(w, j)
        c1 = c == 1 and j > 0
        m_s, m_o, m_q = ix.cnt(h, j - 1), ix.cnt(1 - h, j - 1), ix.cnt(1 - h, j - 2)
        p = prow(r, j)
        pp = ((0 if h == 0 else n2) + partner(p, j - 1)) if c1 else None
        ok = True
        so = (m_s - 1 if m_s > 0 else 0)
        ok &= check(row, (p, m_o % S), m_o, "partner's row")
        for x in [r] + ([pp] if pp is not None else []):
            ok &= check(row, (x, so % S), so, "row before its last move")
            if m_s > 0:
                for cc in (0, 1):
                    ok &= check(prop, (x, m_s % S, cc), m_s, "proposal of the last move")
                if m_q > 0:
                    ok &= check(mseq, (prow(x, j - 2), m_q % S), m_q, "decision of j - 3")
                ok &= check(mseq, (x, m_s % S), m_s, "decision of j - 2")
        return ok

    def c_ready(g, k):
        j = next_c[(g, k)]
        if j >= G or (stall and stall[0] == g and stall[1] == j and events[0] < stall[2]):
            return False
        if j >= NB and (j - NB) not in e_done[g]:                  # the hand-over record is still in use
            return False
        return c_needs(g, j, have)

    def c_step(g, k):
        j = next_c[(g, k)]
        w, c = g
        m_next = ix.cnt(j & 1, j - 1) + 1
        pending.append((prop, (row_of(w, j), m_next % S, c), m_next))
        c_done[g].add(j)
        next_c[(g, k)] = j + NC

    def e_ready(g, h):
        j = next_e[(g, h)]
        if j >= G or j not in c_done[g]:
            return False
        w, c = g
        r = row_of(w, j)
        m_s, m_par = ix.cnt(h, j), ix.cnt(1 - h, j)
        if m_par > 0 and not have(mseq, (prow(r, j), m_par % S), m_par, "partner's decision"):
            return False
        if not have(row, (r, m_s % S), m_s, "the row's lnprob"):
            return False
        if guard and j >= LAG and done.get(j - LAG, 0) < 2 * n2:
            return False
        return True

    def e_step(g, h):
        j = next_e[(g, h)]
        w, c = g
        r = row_of(w, j)
        m_par = ix.cnt(1 - h, j)
        flag = mseq[(prow(r, j), m_par % S)][1] if m_par > 0 else 0
        if flag == c:                                              # this candidate is the chain's proposal
            m_new = ix.cnt(h, j) + 1
            pending.append((mseq, (r, m_new % S), (m_new, rng.randrange(2))))
            pending.append((row, (r, m_new % S), m_new))
        pending.append((done, j, 1))
        e_done[g].add(j)
        next_e[(g, h)] = j + 2

    while events[0] < max_events:
        ready = ([("c", g, k) for g in wgs for k in range(NC) if c_ready(g, k)] +
                 [("e", g, h) for g in wgs for h in (0, 1) if e_ready(g, h)])
        choices = ready + [("p", i) for i in range(len(pending))]
        if not choices:
            if stall and events[0] < stall[2]:
                events[0] = stall[2]
                continue
            break
        # (stores land sooner rather than later most of the time: keeps the model's runs short)
        ev = rng.choice(choices) if (not pending or rng.random() < 0.5) else ("p", rng.randrange(len(pending)))
        events[0] += 1
        if ev[0] == "c":
            c_step(ev[1], ev[2])
        elif ev[0] == "e":
            e_step(ev[1], ev[2])
        else:
            store, key, val = pending.pop(ev[1])
            if store is done:
                done[key] = done.get(key, 0) + 1
            else:
                old = store[key]
                om = old[0] if isinstance(old, tuple) else old
                nm = val[0] if isinstance(val, tuple) else val
                if om is None or nm > om:                          # (a slot is written by one wave, in order)
                    store[key] = val
        prog = [min(next_e[(g, 0)], next_e[(g, 1)] - 1) for g in wgs]
        lead = max(lead, max(prog) - min(prog))
    finished = all(next_e[(g, h)] >= G for g in wgs for h in (0, 1))
    return lead, events[0], finished
