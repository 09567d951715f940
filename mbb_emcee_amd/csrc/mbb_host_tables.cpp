// mbb_host_tables.cpp -- see mbb_host_tables.h.  Host-only, no HIP.
#include "mbb_host_tables.h"

#include <math.h>

#include <algorithm>
#include <functional>

namespace mbbh {

int build_band_layout(const double *freq, const double *weight, const int32_t *offsets, int nb,
                      int seg_chunks, bool pack_tails, BandLayout &L, const char **err)
{
    static const char *msgs[] = {"", "bad band tables", "offsets[0] must be 0", "empty band",
                                 "non-positive frequency"};
    auto bad = [&](int code) { if (err) *err = msgs[-code]; return code; };
    if (!freq || !weight || !offsets || nb <= 0) return bad(-1);
    if (offsets[0] != 0) return bad(-2);
    const int segc = seg_chunks > 0 ? seg_chunks : 4;
    for (int b = 0; b < nb; ++b) {
        if (offsets[b + 1] - offsets[b] <= 0) return bad(-3);
        for (int i = offsets[b]; i < offsets[b + 1]; ++i)
            if (!(freq[i] > 0.0) || !isfinite(freq[i])) return bad(-4);
    }
    L = BandLayout();
    std::vector<double> &nu = L.nu, &lnnu = L.lnnu, &wt = L.wt;
    // the weight goes in times nu^2: the sample loop sums f_nu / x^2 (x = h nu / kT: the factor x^2 of the Planck term is
    // nu^2 times a constant of the walker, applied once per band with normfac) -- two multiplications per sample saved
    auto push = [&](int i) { nu.push_back(freq[i]); lnnu.push_back(log(freq[i])); wt.push_back(weight[i] * (freq[i] * freq[i])); };
    auto pad = [&]() { nu.push_back(1.0); lnnu.push_back(0.0); wt.push_back(0.0); };
    // A band with a passband is cut into chunks of 64 samples; segments of <= segc chunks
    // are one unit of work and one result slot each.  What is left over at the end of a
    // band (fewer than 49 samples) does not get a chunk of its own: the leftovers of all
    // bands share *tail chunks*, each leftover in whole rows of 16 lanes, one result slot
    // per row (a row total is the first stage of the wave reduction anyway).  Single-sample
    // bands (delta-function photometry, the reference's default: response.py:572-574,
    // likelihood.py:817) are packed 64 to a chunk, lane = band, one slot per lane, no
    // reduction.  Slots are numbered band by band, so a band's flux is the sum of its
    // slots [s0, s1) in that order.  Table order: full chunks, tail chunks, delta chunks.
    struct Tail { int band, first, count; };
    std::vector<Unit> units;                       // {slot | tail chunk, c0, c1, kind}
    std::vector<Tail> tails;
    std::vector<SlotRange> band_rng(nb);
    std::vector<int> tail_slot_of_row;             // slot of each tail row, in packing order
    int chunk = 0, slot = 0, nseg = 0;
    for (int b = 0; b < nb; ++b) {
        const int n = offsets[b + 1] - offsets[b];
        if (n == 1) continue;
        int full = n / 64, rem = n % 64;
        int trows = (rem + 15) / 16;
        if (!pack_tails || trows == 4) { full += rem ? 1 : 0; trows = 0; rem = 0; }
        band_rng[b].s0 = slot;
        for (int cc = 0; cc < full; cc += segc) {
            units.push_back(Unit{slot++, chunk + cc, chunk + (cc + segc < full ? cc + segc : full), 0});
            ++nseg;
        }
        const int in_full = n - rem;               // samples that sit in full chunks
        for (int i = 0; i < full * 64; ++i) {
            if (i < in_full) push(offsets[b] + i);
            else pad();
        }
        chunk += full;
        for (int r = 0; r < trows; ++r) {
            tails.push_back({b, offsets[b] + in_full + 16 * r, (rem - 16 * r < 16) ? rem - 16 * r : 16});
            tail_slot_of_row.push_back(slot++);
        }
        band_rng[b].s1 = slot;
    }
    const int ntc = ((int)tails.size() + 3) / 4;
    std::vector<int32_t> tail_slot(4 * (size_t)(ntc > 0 ? ntc : 1), -1);
    for (int k = 0; k < ntc; ++k) {
        for (int r = 0; r < 4; ++r) {
            const size_t t = 4 * (size_t)k + r;
            const int cnt = t < tails.size() ? tails[t].count : 0;
            if (t < tails.size()) tail_slot[t] = tail_slot_of_row[t];
            for (int i = 0; i < 16; ++i) {
                if (i < cnt) push(tails[t].first + i);
                else pad();
            }
        }
        units.push_back(Unit{k, chunk + k, chunk + k + 1, 2});
    }
    chunk += ntc;
    int nd = 0;
    for (int b = 0; b < nb; ++b)
        if (offsets[b + 1] - offsets[b] == 1) {
            band_rng[b] = SlotRange{slot + nd, slot + nd + 1};
            push(offsets[b]);
            ++nd;
        }
    const int ndc = (nd + 63) / 64;
    for (int i = nd; i < ndc * 64; ++i) pad();
    for (int k = 0; k < ndc; ++k) units.push_back(Unit{slot + 64 * k, chunk + k, chunk + k + 1, 1});
    // Dealing order.  Waves w, w+4, w+8, ... of a workgroup share a SIMD and unit u goes
    // to wave u mod nwave, so position i of the table lands on SIMD i mod 4.  Longest
    // units first, each to the least loaded SIMD that still has a position free.
    const int nunit = (int)units.size();
    auto len = [&](int u) { return units[u].c1 - units[u].c0; };
    std::vector<int> by_size(nunit);
    for (int i = 0; i < nunit; ++i) by_size[i] = i;
    std::stable_sort(by_size.begin(), by_size.end(), [&](int x, int y) { return len(x) > len(y); });
    std::vector<int> mine[4];
    int load[4] = {0, 0, 0, 0}, cap[4];
    for (int g = 0; g < 4; ++g) cap[g] = (nunit - g + 3) / 4;            // positions g, g+4, ...
    for (int k = 0; k < nunit; ++k) {
        int best = -1;
        for (int g = 0; g < 4; ++g) {
            if ((int)mine[g].size() >= cap[g]) continue;
            if (best < 0 || load[g] < load[best]) best = g;
        }
        mine[best].push_back(by_size[k]);
        load[best] += len(by_size[k]);
    }
    // The greedy deal can be a chunk off when the SIMDs have different numbers of
    // positions.  With few units (the latency regime, where it matters) search for the
    // deal with the smallest maximum: depth first over the units by decreasing length,
    // bounded, pruned at the best maximum found so far.
    if (nunit <= 24) {
        int total = 0;
        for (int u = 0; u < nunit; ++u) total += len(u);
        const int floor_max = (total + 3) / 4;
        int best_max = *std::max_element(load, load + 4);
        std::vector<int> where(nunit, 0), best_where;
        int cur[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
        long nodes = 0;
        std::function<void(int)> dfs = [&](int k) {
            if (best_max == floor_max || ++nodes > 200000) return;
            if (k == nunit) {
                const int m = *std::max_element(cur, cur + 4);
                if (m < best_max) { best_max = m; best_where = where; }
                return;
            }
            const int l = len(by_size[k]);
            for (int g = 0; g < 4; ++g) {
                if (cnt[g] >= cap[g] || cur[g] + l >= best_max) continue;
                cur[g] += l; ++cnt[g]; where[k] = g;
                dfs(k + 1);
                cur[g] -= l; --cnt[g];
            }
        };
        dfs(0);
        if (!best_where.empty()) {
            for (int g = 0; g < 4; ++g) { mine[g].clear(); load[g] = 0; }
            for (int k = 0; k < nunit; ++k) {
                mine[best_where[k]].push_back(by_size[k]);
                load[best_where[k]] += len(by_size[k]);
            }
        }
    }
    L.unit_tab.assign(nunit, Unit{0, 0, 0, 0});
    for (int g = 0; g < 4; ++g) {
        std::stable_sort(mine[g].begin(), mine[g].end(), [&](int x, int y) { return len(x) > len(y); });
        for (size_t i = 0; i < mine[g].size(); ++i) L.unit_tab[g + 4 * i] = units[mine[g][i]];
    }
    L.band_rng = band_rng;
    L.tail_slot = tail_slot;
    L.nb = nb;
    L.nchunk = chunk + ndc;
    L.nseg = nseg;
    L.nunit = nunit;
    L.npart = slot + 64 * ndc;
    for (int g = 0; g < 4; ++g) L.simd_chunks[g] = load[g];
    L.nq = offsets[nb];
    return 0;
}

// ---- piecewise polynomials -----------------------------------------------------
namespace {

typedef long double ld;

ld b_fun(ld x)        // x / expm1(x), 1 at 0
{
    if (fabsl(x) < 1e-6L) return 1.0L - x / 2 + x * x / 12;       // next term x^4/720
    return x / expm1l(x);
}

ld c_over_y(ld y)     // C(y) / y = (1 - exp(-y)) / y, 1 at 0
{
    if (fabsl(y) < 1e-6L) return 1.0L - y / 2 + y * y / 6;        // next term y^3/24
    return -expm1l(-y) / y;
}

ld C_fun(ld y) { return -expm1l(-y); }                            // 1 - exp(-y)

// Coefficients, lowest order first, of the polynomial of degree n - 1 in t (0 <= t <= 1) that
// interpolates g(t) at the n Chebyshev nodes of [0, 1].  Solved in s = 2 t - 1, where the
// Vandermonde matrix of the nodes is benign (Gauss-Jordan with partial pivoting in long
// double), then expanded in t by the binomial theorem.
template <typename G>
void cheb_fit(G g, int n, ld *out)
{
    constexpr int NMAX = kPolyDeg + 1;
    ld s[NMAX], V[NMAX][2 * NMAX], a[NMAX];
    const ld pi = acosl(-1.0L);
    for (int k = 0; k < n; ++k) s[k] = cosl((2 * k + 1) * pi / (2 * n));
    for (int r = 0; r < n; ++r) {
        ld p = 1.0L;
        for (int c = 0; c < n; ++c) { V[r][c] = p; p *= s[r]; }
        for (int c = 0; c < n; ++c) V[r][n + c] = (r == c) ? 1.0L : 0.0L;
    }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r)
            if (fabsl(V[r][col]) > fabsl(V[piv][col])) piv = r;
        if (piv != col)
            for (int c = 0; c < 2 * n; ++c) std::swap(V[piv][c], V[col][c]);
        const ld d = V[col][col];
        for (int c = 0; c < 2 * n; ++c) V[col][c] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            const ld m = V[r][col];
            if (m != 0.0L)
                for (int c = 0; c < 2 * n; ++c) V[r][c] -= m * V[col][c];
        }
    }
    ld val[NMAX];
    for (int k = 0; k < n; ++k) val[k] = g((1.0L + s[k]) / 2);
    for (int k = 0; k < n; ++k) {                     // coefficient of s^k
        a[k] = 0.0L;
        for (int j = 0; j < n; ++j) a[k] += V[k][n + j] * val[j];
    }
    // sum_k a_k (2 t - 1)^k = sum_j t^j 2^j sum_{k >= j} a_k C(k, j) (-1)^(k - j)
    for (int j = 0; j < n; ++j) {
        ld cj = 0.0L;
        for (int k = j; k < n; ++k) {
            ld binom = 1.0L;
            for (int m = 0; m < j; ++m) binom = binom * (ld)(k - m) / (ld)(m + 1);
            cj += a[k] * binom * (((k - j) & 1) ? -1.0L : 1.0L);
        }
        out[j] = ldexpl(cj, j);
    }
}

// `zero_order`: the order of the function's zero at the origin (row 0 is t^zero_order times the interpolant of
// `over`, the function divided by that power of its argument)
void fit_table(ld (*f)(ld), ld (*over)(ld), int zero_order, int count, std::vector<double> &out)
{
    constexpr int N = kPolyDeg + 1;
    const ld h = 0.125L;
    out.assign((size_t)count * kPolyStride, 0.0);
    for (int i = 0; i < count; ++i) {
        ld c[N] = {0};
        if (i == 0) {
            // f(h t) = (h t)^z over(h t): coefficients of t^z .. t^7 from the interpolant of h^z over(h t)
            ld q[N];
            const ld hz = powl(h, (ld)zero_order);
            cheb_fit([&](ld t) { return hz * over(h * t); }, N - zero_order, q);
            for (int k = zero_order; k < N; ++k) c[k] = q[k - zero_order];
        } else {
            const ld x0 = (ld)i * h;
            cheb_fit([&](ld t) { return f(x0 + h * t); }, N, c);
        }
        for (int k = 0; k < N; ++k) out[(size_t)i * kPolyStride + k] = (double)c[k];
    }
}

}  // namespace

void build_poly_tables(std::vector<double> &b, std::vector<double> &c)
{
    fit_table(b_fun, b_fun, 0, kPolyBCount, b);
    fit_table(C_fun, c_over_y, 1, kPolyCCount, c);
}

}  // namespace mbbh

// ---- C hooks for the CPU tests (the sanitizer build binds these through ctypes) --------
extern "C" int mbbh_poly_counts(int *nb_intervals, int *nc_intervals, int *ncoef)
{
    *nb_intervals = mbbh::kPolyBCount;
    *nc_intervals = mbbh::kPolyCCount;
    *ncoef = mbbh::kPolyStride;          // doubles per row: eight coefficients and the padding
    return 0;
}

extern "C" int mbbh_poly_tables(double *b, double *c)
{
    std::vector<double> vb, vc;
    mbbh::build_poly_tables(vb, vc);
    std::copy(vb.begin(), vb.end(), b);
    std::copy(vc.begin(), vc.end(), c);
    return 0;
}

// counts[9] = nchunk, nunit, npart, nseg, nq, simd_chunks[4]; the arrays must hold
// cap_chunks * 64 samples, cap_units units (4 ints each), nb slot ranges (2 ints each)
// and 4 * cap_chunks tail slots.  Returns the builder's code, or -9 if a capacity is short.
extern "C" int mbbh_band_layout(const double *freq, const double *weight, const int32_t *offsets, int nb,
                                int seg_chunks, int pack_tails, int32_t *counts, double *nu, double *lnnu,
                                double *wt, int32_t *unit_tab, int32_t *band_rng, int32_t *tail_slot,
                                int cap_chunks, int cap_units)
{
    mbbh::BandLayout L;
    const char *err = nullptr;
    const int rc = mbbh::build_band_layout(freq, weight, offsets, nb, seg_chunks, pack_tails != 0, L, &err);
    if (rc) return rc;
    if (L.nchunk > cap_chunks || L.nunit > cap_units || (int)L.tail_slot.size() > 4 * cap_chunks) return -9;
    counts[0] = L.nchunk; counts[1] = L.nunit; counts[2] = L.npart; counts[3] = L.nseg; counts[4] = L.nq;
    for (int g = 0; g < 4; ++g) counts[5 + g] = L.simd_chunks[g];
    std::copy(L.nu.begin(), L.nu.end(), nu);
    std::copy(L.lnnu.begin(), L.lnnu.end(), lnnu);
    std::copy(L.wt.begin(), L.wt.end(), wt);
    for (int u = 0; u < L.nunit; ++u) {
        unit_tab[4 * u] = L.unit_tab[u].slot; unit_tab[4 * u + 1] = L.unit_tab[u].c0;
        unit_tab[4 * u + 2] = L.unit_tab[u].c1; unit_tab[4 * u + 3] = L.unit_tab[u].kind;
    }
    for (int b = 0; b < nb; ++b) { band_rng[2 * b] = L.band_rng[b].s0; band_rng[2 * b + 1] = L.band_rng[b].s1; }
    std::copy(L.tail_slot.begin(), L.tail_slot.end(), tail_slot);
    return 0;
}

// index arithmetic of the one-launch sampler run (mbb_flow_index.h), for the protocol model
#include "mbb_flow_index.h"
extern "C" int mbbh_flow_index(int h, int j, int m, int *cnt, int *seq, int *slots, int *lag)
{
    *cnt = flow_cnt(h, j);
    *seq = flow_seq(h, m);
    *slots = kFlowSlots;
    *lag = kFlowLag;
    return 0;
}

// ... and the constants of sampler form 7 (tests/_flowm_model.py)
extern "C" int mbbh_flowm_consts(int *slots, int *lag, int *ring, int *nc, int *nb)
{
    *slots = kFmSlots; *lag = kFmLag; *ring = kFmRing; *nc = kFmNC; *nb = kFmNB;
    return 0;
}
