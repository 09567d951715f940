"""A host-side model of the hand-over protocol of the resident sampler forms (k_flowa, form 9: mbb_flowa.hip.h;
round 4's form 8 used the same protocol), used by tests/test_host_cpu.py.  Like _flowm_model.py for form 7 it
restates WHO waits for WHAT and who reads and writes which slot, with the kernel's own index arithmetic
(mbb_flow_index.h through the C hooks), and runs the actors in random or adversarial order with every store to
memory landing at a random later time.  Every slot remembers the move whose data it holds -- what the check words
say in the kernel -- and a reader that finds a LATER move than the one it is entitled to would spin for ever
there: a violation, a slot reused under a reader.

Workgroup g owns walkers [g W, g W + W) of both halves.  Actors per workgroup:
  C(j)  form 9: the constructor waves' pass for half-step j, in order.  Needs, for every owned walker of the half
        that moves in j: its partner's row BEFORE the partner's pending move (candidate 0) and, from the second
        half-step on, the proposal of that pending move (candidate 1) and then its decision; the lag guard (every
        workgroup's C through with j - kFmLag); its own E through with j - 2 (the owned rows are final, the record
        buffer is free).  Publishes the proposal of each owned walker's move and counts itself for the guard.
        (Since round 4 the kernel publishes a walker's proposal as soon as THAT walker's partner has decided rather than
        when all of the workgroup's have: the same store into the same slot within the same half-step, only earlier --
        what protects the slot's previous content is the guard at the half-step's entry, which the model keeps.)
        form 8: phase 1 of the half-step: needs every partner's row AFTER its last move (half-step j - 1), the lag
        guard, its own E(j - 1).
  E(j)  quadrature, accept test and publication of half-step j for the owned walkers: needs C(j); publishes the
        decision words and the rows after the move.
"""
import random
from _flow_model import Violation
from _flowm_model import consts


def run(ix, lib, n2, W, nsteps, rng, ahead=True, guard=True, partner=None, stall=None, max_events=2000000):
    """Returns (largest lead of any workgroup's E over the slowest, events, all done).  stall = (workgroup,
    half-step, events): that workgroup's C does not take that half-step until so many events have passed."""
    S, LAG, RING, _, _ = consts(lib)
    G = 2 * nsteps
    ngrp = (n2 + W - 1) // W
    if partner is None:
        table = {}
        partner = lambda r, j: table.setdefault((r, j), rng.randrange(n2))
    half_of = lambda r: 0 if r < n2 else 1
    prow = lambda r, j: (n2 if half_of(r) == 0 else 0) + partner(r, j)
    owned = lambda g, j: [(n2 if (j & 1) else 0) + w for w in range(g * W, min(n2, g * W + W))]
    rows = list(range(2 * n2))
    prop = {(r, s): None for r in rows for s in range(S)}
    row = {(r, s): (0 if s == 0 else None) for r in rows for s in range(S)}
    mseq = {(r, s): None for r in rows for s in range(S)}
    done = {}
    pending = []
    next_c = {g: 0 for g in range(ngrp)}
    next_e = {g: 0 for g in range(ngrp)}
    events = [0]
    lead = 0

    def have(store, key, want, what):
        got = store[key]
        gm = got[0] if isinstance(got, tuple) else got
        if gm is not None and gm > want:
            raise Violation("%s %s holds move %s, reader wants %s" % (what, key, gm, want))
        return gm == want

    def c_ready(g):
        j = next_c[g]
        if j >= G or (stall and stall[0] == g and stall[1] == j and events[0] < stall[2]):
            return False
        back = 2 if ahead else 1
        if j >= back and next_e[g] <= j - back:                    # its own E of j - 2 (form 8: j - 1) is not through
            return False
        if guard and j >= LAG and done.get(j - LAG, 0) < ngrp:
            return False
        h = j & 1
        ok = True
        for r in owned(g, j):
            p = prow(r, j)
            if ahead:
                m1, m0 = ix.cnt(1 - h, j), ix.cnt(1 - h, j - 1)
                ok &= have(row, (p, m0 % S), m0, "partner's row before its pending move")
                if j > 0:
                    ok &= have(prop, (p, m1 % S), m1, "partner's pending proposal")
                    ok &= have(mseq, (p, m1 % S), m1, "partner's decision")
            else:
                m = ix.cnt(1 - h, j)
                ok &= have(row, (p, m % S), m, "partner's row")
        return ok

    def c_step(g):
        j = next_c[g]
        if ahead:
            for r in owned(g, j):
                m_new = ix.cnt(j & 1, j) + 1
                pending.append((prop, (r, m_new % S), m_new))
        pending.append((done, j, 1))
        next_c[g] = j + 1

    def e_ready(g):
        j = next_e[g]
        return j < G and next_c[g] > j

    def e_step(g):
        j = next_e[g]
        for r in owned(g, j):
            m_new = ix.cnt(j & 1, j) + 1
            pending.append((mseq, (r, m_new % S), (m_new, rng.randrange(2))))
            pending.append((row, (r, m_new % S), m_new))
        next_e[g] = j + 1

    while events[0] < max_events:
        ready = [("c", g) for g in range(ngrp) if c_ready(g)] + [("e", g) for g in range(ngrp) if e_ready(g)]
        choices = ready + [("p", i) for i in range(len(pending))]
        if not choices:
            if stall and events[0] < stall[2]:
                events[0] = stall[2]
                continue
            break
        ev = rng.choice(choices) if (not pending or rng.random() < 0.5) else ("p", rng.randrange(len(pending)))
        events[0] += 1
        if ev[0] == "c":
            c_step(ev[1])
        elif ev[0] == "e":
            e_step(ev[1])
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
        prog = list(next_e.values())
        lead = max(lead, max(prog) - min(prog))
    return lead, events[0], all(v >= G for v in next_e.values())
