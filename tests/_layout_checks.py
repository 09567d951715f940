"""Property checks of the band-layout builder (mbb_emcee_amd/csrc/mbb_host_tables.cpp,
build_band_layout) through its C hooks.  Shared by tests/test_host_cpu.py (plain gcc
build) and tests/_asan_worker.py (AddressSanitizer + UBSan build)."""
import ctypes as C

import numpy as np

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def load(path):
    lib = C.CDLL(path)
    lib.mbbh_band_layout.argtypes = [_dp, _dp, _ip, C.c_int, C.c_int, C.c_int, _ip, _dp, _dp, _dp,
                                     _ip, _ip, _ip, C.c_int, C.c_int]
    lib.mbbh_band_layout.restype = C.c_int
    lib.mbbh_poly_counts.argtypes = [C.POINTER(C.c_int)] * 3
    lib.mbbh_poly_tables.argtypes = [_dp, _dp]
    return lib


def layout(lib, sizes, seg_chunks=4, pack_tails=1, rng=None):
    """Build the layout of bands with the given sample counts.  Frequencies are distinct
    integers >= 2 and weights small non-zero integers, so that every checksum below is exact."""
    rng = rng or np.random.RandomState(0)
    sizes = np.asarray(sizes, dtype=np.int64)
    nq = int(sizes.sum())
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    freq = (2.0 + rng.permutation(nq)).astype(np.float64)
    weight = rng.randint(1, 8, nq).astype(np.float64) * rng.choice([-1.0, 1.0], nq)
    nb = len(sizes)
    cap_chunks = int(np.sum((sizes + 63) // 64)) + nb + 2
    cap_units = cap_chunks + nb + 2
    counts = np.zeros(9, dtype=np.int32)
    nu = np.full(cap_chunks * 64, np.nan); lnnu = np.full(cap_chunks * 64, np.nan); wt = np.full(cap_chunks * 64, np.nan)
    unit_tab = np.full(4 * cap_units, -7, dtype=np.int32)
    band_rng = np.full(2 * nb, -7, dtype=np.int32)
    tail_slot = np.full(4 * cap_chunks, -7, dtype=np.int32)
    rc = lib.mbbh_band_layout(freq.ctypes.data_as(_dp), weight.ctypes.data_as(_dp), offsets.ctypes.data_as(_ip), nb,
                              int(seg_chunks), int(pack_tails), counts.ctypes.data_as(_ip), nu.ctypes.data_as(_dp),
                              lnnu.ctypes.data_as(_dp), wt.ctypes.data_as(_dp), unit_tab.ctypes.data_as(_ip),
                              band_rng.ctypes.data_as(_ip), tail_slot.ctypes.data_as(_ip), cap_chunks, cap_units)
    assert rc == 0, rc
    nchunk, nunit, npart, nseg, nq_out = (int(x) for x in counts[:5])
    return dict(sizes=sizes, offsets=offsets, freq=freq, weight=weight, nchunk=nchunk, nunit=nunit, npart=npart,
                nseg=nseg, nq=nq_out, simd=counts[5:9].copy(), nu=nu[:nchunk * 64], lnnu=lnnu[:nchunk * 64],
                wt=wt[:nchunk * 64], unit_tab=unit_tab[:4 * nunit].reshape(-1, 4), band_rng=band_rng.reshape(-1, 2),
                tail_slot=tail_slot, seg_chunks=seg_chunks, untouched=(nu[nchunk * 64:], unit_tab[4 * nunit:]))


def check(L):
    """Every sample once; chunks tiled by the units; slots tiled by the bands; each band's
    slots hold exactly the band's samples (what k_lnlike's phases 2 and 3 rely on)."""
    sizes, nchunk, nunit, npart = L["sizes"], L["nchunk"], L["nunit"], L["npart"]
    nb = len(sizes)
    assert L["nq"] == sizes.sum()
    # nothing written beyond what was reported
    assert np.all(np.isnan(L["untouched"][0])) and np.all(L["untouched"][1] == -7)
    nu, wt, lnnu = L["nu"], L["wt"], L["lnnu"]
    assert not np.isnan(nu).any() and not np.isnan(wt).any() and not np.isnan(lnnu).any()
    pad = (nu == 1.0) & (wt == 0.0) & (lnnu == 0.0)
    real = ~pad
    # every sample exactly once, with its own weight TIMES nu^2 (the quadrature sums f/x^2: mbb_host_tables.h), and
    # log(nu) beside it
    assert real.sum() == L["nq"]
    order = np.argsort(nu[real])
    assert np.array_equal(nu[real][order], np.sort(L["freq"]))
    assert np.array_equal(wt[real][order], (L["weight"] * L["freq"] ** 2)[np.argsort(L["freq"])])
    assert np.array_equal(lnnu[real], np.log(nu[real]))
    # the units tile the chunks
    ut = L["unit_tab"]
    cover = np.zeros(nchunk, dtype=np.int64)
    for slot, c0, c1, kind in ut:
        assert 0 <= c0 < c1 <= nchunk and kind in (0, 1, 2)
        assert c1 - c0 <= max(1, L["seg_chunks"])
        cover[c0:c1] += 1
    assert np.all(cover == 1)
    assert L["simd"].sum() == nchunk and L["nseg"] == int(np.sum(ut[:, 3] == 0))
    # positions g, g+4, ... of the table belong to SIMD g: its chunk count is what was reported
    for g in range(4):
        assert int(np.sum(ut[g::4, 2] - ut[g::4, 1])) == int(L["simd"][g])
    # run the table the way the kernel does, with the exact checksum w * nu per sample
    part = np.full(npart, np.nan)
    contrib = (wt * nu).reshape(nchunk, 64)
    for slot, c0, c1, kind in ut:
        if kind == 0:
            assert np.isnan(part[slot]); part[slot] = contrib[c0:c1].sum()
        elif kind == 2:
            for r in range(4):
                sl = L["tail_slot"][4 * slot + r]
                if sl >= 0:
                    assert np.isnan(part[sl]); part[sl] = contrib[c0, 16 * r:16 * r + 16].sum()
                else:
                    assert np.all(pad.reshape(nchunk, 64)[c0, 16 * r:16 * r + 16])
        else:
            assert np.all(np.isnan(part[slot:slot + 64])); part[slot:slot + 64] = contrib[c0]
    # the bands' slot ranges tile [0, S), S <= npart < S + 64, and sum to the band's samples
    br = L["band_rng"]
    seen = np.zeros(npart, dtype=np.int64)
    for b in range(nb):
        s0, s1 = br[b]
        assert 0 <= s0 < s1 <= npart
        seen[s0:s1] += 1
        lo, hi = L["offsets"][b], L["offsets"][b + 1]
        assert part[s0:s1].sum() == np.sum(L["weight"][lo:hi] * L["freq"][lo:hi] ** 3), b
    S = int(seen.sum())
    assert np.all(seen[:S] == 1) and np.all(seen[S:] == 0) and npart - S < 64
    return True


EDGE_SIZES = [2, 15, 16, 17, 32, 33, 47, 48, 49, 63, 64, 65, 80, 81, 127, 128, 129, 193]


def run_all(lib, nrandom=1000, seed=5):
    """The 18 band sizes around the row and chunk boundaries (each alone, and all together),
    then nrandom random layouts: 1-40 bands of 1-700 samples, single-sample bands mixed in,
    both tail layouts, segment lengths 1-6."""
    rng = np.random.RandomState(seed)
    n = 0
    for pack in (1, 0):
        for sz in EDGE_SIZES:
            check(layout(lib, [sz], 4, pack, rng)); n += 1
        check(layout(lib, EDGE_SIZES, 4, pack, rng)); n += 1
        check(layout(lib, [1] * 5, 4, pack, rng)); n += 1           # delta-function photometry
        check(layout(lib, [1] * 130, 4, pack, rng)); n += 1         # three packed chunks
    for _ in range(nrandom):
        nb = int(rng.randint(1, 41))
        kind = rng.rand(nb)
        sizes = np.where(kind < 0.2, 1, np.where(kind < 0.6, rng.randint(2, 130, nb), rng.randint(2, 700, nb)))
        check(layout(lib, sizes, int(rng.randint(1, 7)), int(rng.randint(2)), rng)); n += 1
    return n
