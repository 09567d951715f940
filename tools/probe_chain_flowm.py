"""Diagnostic build only (-DMBB_STAMPS): the critical path of sampler form 7, from event times of the launch's
last 64 half-steps (s_memrealtime, 10 ns ticks; all CUs share that clock).  From a late decision the path is walked
back, at every join along the input that arrived last, and the time is booked to the kind of segment."""
import os, sys, ctypes as C, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.environ.get("MBB_STAMPS_LIB", os.path.join(ROOT, "tools", "libmbb_hip_stamps.so")); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU
like, flux = make_likelihood(0)
ctx = like._sync_device()
lib = ctx.lib
lib.mbb_stamps(ctx.h, None, 0)
for o in sys.argv[1:]:
    k, v = o.split("="); ctx.set_option(k, int(v))
smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
smp.run_mcmc(walkers(1)[:NW_PER_GPU], 100, storechain=False)
NS = 500
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0); smp.advance_async(NS); ctx.record(e1); ctx.sync()
print("%.3f us per step, form %d" % (ctx.elapsed_ms(e0, e1) * 1e3 / NS, ctx.info("last_kernel_form")))
nb = int(ctx.info("last_grid")); half = nb // 2
big = np.zeros((32768 + 256 + nb * 32 + 8, 32), dtype=np.uint64)
lib.mbb_stamps(ctx.h, big.ctypes.data_as(C.c_void_p), big.shape[0])
ev = big.reshape(-1)[(1 << 20) + 8192:(1 << 20) + 8192 + nb * 64 * 16].reshape(nb, 64, 16).astype(np.int64)
niter = 2 * NS
T = lambda g, j, e: ev[g, j & 63, e] * 10.0               # ns


def decider(pair, j):
    """workgroup that published the decision of (pair, half-step j) and when"""
    a, b = T(2 * pair, j, 6), T(2 * pair + 1, j, 6)
    return (2 * pair, a) if a >= b else (2 * pair + 1, b)


book = collections.Counter()
hops = collections.Counter()
paths = 0
jlast = niter - 3
for g0 in range(0, nb, 2):
    g, t = decider(g0 >> 1, jlast)
    j = jlast
    while j > niter - 50:
        # decision of (g, j) at T6: the later of E's words (partner's decision of j - 1) and the sums (Q)
        t6, t5, t4 = T(g, j, 6), T(g, j, 5), T(g, j, 4)
        dep = ev[g, j & 63, 8]
        prow, pprow, rq, rp, c1, has = dep & 1023, (dep >> 10) & 1023, (dep >> 20) & 1023, (dep >> 30) & 1023, (dep >> 40) & 1, (dep >> 41) & 1
        w = g >> 1
        t9 = T(g, j, 9)                                    # E turned to this half-step (it is through with j - 1)
        if t5 > t4 + 400:                                  # (the sums take ~300 ns: words later than that -> they were last)
            book["accept test + publish after E's words"] += t6 - t5
            g2, td = decider(prow % half, j - 1)
            if t9 > td + 300:                              # the decision was out before E looked: E itself was late
                book["E's first look (it was busy with j - 1 until then)"] += t5 - t9
                hops["E busy with the workgroup's other walker (j - 1)"] += 1
                # where E was before: the end of its pass over j - 1, whatever that waited for
                t = t9
                j = j - 1
                t6 = t9                                    # (its publish / count came last)
                t5, t4 = T(g, j, 5), T(g, j, 4)
                book["E: end of pass j - 1 after the later of its words and sums"] += t9 - max(t5, t4)
                if t5 > t4 + 400:
                    g2, td = decider((ev[g, j & 63, 8] & 1023) % half, j - 1)
                    book["hand-over: decision of j - 1 -> E's words satisfied"] += t5 - td
                    hops["via the partner's decision of j - 1"] += 1
                    g, j = g2, j - 1
                    continue
                # else fall through to the Q / C path of (g, j)
                t6 = max(t5, t4)
                dep = ev[g, j & 63, 8]
                prow, pprow, rq, rp, c1, has = dep & 1023, (dep >> 10) & 1023, (dep >> 20) & 1023, (dep >> 30) & 1023, (dep >> 40) & 1, (dep >> 41) & 1
            else:
                book["hand-over: decision of j - 1 -> E's words satisfied"] += t5 - td
                hops["via the partner's decision of j - 1"] += 1
                g, j = g2, j - 1
                continue
        book["band sums + accept test + publish"] += t6 - t4
        t3, t2, t1, t0 = T(g, j, 3), T(g, j, 2), T(g, j, 1), T(g, j, 0)
        t14 = T(g, j, 14)
        t15 = T(g, j, 15)
        book["quadrature: Q wave 1 takes the record -> the last Q wave is through"] += t15 - max(t14, t3)
        book["quadrature: last Q wave through -> E starts the sums"] += t4 - t15
        book["quadrature: record waits for Q wave 1 (busy with the other half's)"] += max(t14, t3) - t3
        book["selection + hand-over to Q"] += t3 - t2
        if t2 - t1 > 700:                                  # W2 had to wait beyond one look: a decision of j - 2 was last
            cands = [decider(w, j - 2)] + ([decider(pprow % half, j - 2)] if c1 else [])
            g2, td = max(cands, key=lambda x: x[1])
            book["hand-over: decision of j - 2 -> W2 satisfied"] += t2 - td
            hops["via a decision of j - 2"] += 1
            g, j = g2, j - 2
            continue
        book["W2 look after the constructor"] += t2 - t1
        book["constructor"] += t1 - t0
        if not has:
            break
        cands = [decider(prow % half, j - 3), decider(rq % half, j - 3)] + ([decider(rp % half, j - 3)] if c1 else [])
        g2, td = max(cands, key=lambda x: x[1])
        t13 = T(g, j, 13)
        if t13 > td + 300:
            book["C turns to this proposal late (busy with its previous one): first look"] += t0 - t13
            book["C: decision of j - 3 out -> C turns to the proposal"] += t13 - td
        else:
            book["hand-over: decision of j - 3 -> W1 satisfied"] += t0 - td
        hops["via a decision of j - 3"] += 1
        g, j = g2, j - 3
    paths += 1
    book["_half_steps"] += jlast - j
tot = sum(v for k, v in book.items() if not k.startswith("_"))
print("critical paths walked back from %d decisions of half-step %d: %.0f ns per half-step" % (paths, jlast, tot / book["_half_steps"]))
for k, v in sorted(book.items(), key=lambda x: -x[1]):
    if not k.startswith("_"):
        print("  %5.1f %%  %6.0f ns per half-step  %s" % (100 * v / tot, v / book["_half_steps"], k))
print("joins taken:", dict(hops))

# which of E's three words came last (half-steps where E was not late itself)
last = collections.Counter(); lat = collections.defaultdict(list)
for j in range(niter - 50, niter - 2):
    for g in range(nb):
        t9, t21, t22, t23 = T(g, j, 9), T(g, j, 10), T(g, j, 11), T(g, j, 12)
        k = max((t21, "lnprob of the walker's row"), (t22, "partner's decision"), (t23, "lag guard"))[1]
        last[k] += 1
        prow = ev[g, j & 63, 7]
        td = decider(int(prow) % half, j - 1)[1]
        if t22 > 0 and t9 < td:
            lat["partner's decision out -> seen by a waiting E"].append(t22 - td)
        if t23 > 0:
            tl = max(max(T(x, j - 4, 5), T(x, j - 4, 4)) for x in range(nb))
            lat["last workgroup through with j - 4 (its words and sums) -> lag guard seen"].append(t23 - tl)
print("E's last word:", dict(last))
for k, v in lat.items():
    print("  %s: median %.0f p10 %.0f p90 %.0f ns (%d)" % (k, np.median(v), np.percentile(v, 10), np.percentile(v, 90), len(v)))
