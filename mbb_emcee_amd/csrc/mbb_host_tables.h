// mbb_host_tables.h -- host-only table builders of the likelihood hot path.
//
// Plain C++ (no HIP types): compiled into libmbb_hip.so by hipcc and, on its own,
// by gcc with -fsanitize=address,undefined for the CPU test target
// (`make -C oracle asan`, tests/test_host_cpu.py).
//
//   build_band_layout  the chunk / unit / result-slot layout of the passband samples (frequency, its log, and the
//                      weight times nu^2: the kernels sum f_nu / x^2) that k_lnlike reads (what response.__call__, response.py:572-576,
//                      does band by band becomes one table walked by all waves)
//   build_poly_tables  piecewise polynomials of the two smooth factors of the sample
//                      arithmetic, b(x) = x / expm1(x) and C(y) = 1 - exp(-y)
#pragma once
#include <stdint.h>
#include <vector>

namespace mbbh {

struct Unit { int32_t slot, c0, c1, kind; };     // same layout as HIP's int4
struct SlotRange { int32_t s0, s1; };            // same layout as HIP's int2

struct BandLayout {
    std::vector<double> nu, lnnu, wt;            // [nchunk*64]
    std::vector<Unit> unit_tab;                  // [nunit] in dealing order
    std::vector<SlotRange> band_rng;             // [nb]
    std::vector<int32_t> tail_slot;              // [4 * max(1, tail chunks)]
    int nb = 0, nseg = 0, nunit = 0, npart = 0, nchunk = 0, nq = 0;
    int simd_chunks[4] = {0, 0, 0, 0};
};

// Returns 0, or a negative code with *err set to a static string:
//   -1 bad arguments, -2 offsets[0] != 0, -3 empty band, -4 non-positive or non-finite frequency
int build_band_layout(const double *freq, const double *weight, const int32_t *offsets, int nb,
                      int seg_chunks, bool pack_tails, BandLayout &out, const char **err);

// Piecewise degree-7 polynomials on the intervals [i/8, (i+1)/8):
//   b(x) = x / expm1(x)     rows i = 0 .. kPolyBCount-1   (x in [0, 48]: the Planck factor x^3/expm1(x) of fnu.pyx:25, :76
//                                                          is x^2 b(x), and the x^2 is in the weights: build_band_layout)
//   C(y) = 1 - exp(-y)      rows i = 0 .. kPolyCCount-1   (y in [0, 37]: the optical-depth factor of fnu.pyx:75, :106)
// A row is kPolyStride doubles: eight coefficients, lowest order first, in t = 8x - i (0 <= t < 1), then padding
// (zeros) -- rows 80 bytes apart sit on sixteen different bank positions of the LDS, rows of 64 bytes on four
// (mbb_math.hip.h, polyrow_eval).  Each polynomial interpolates the function at the eight Chebyshev nodes of its
// interval (computed in long double).  Row 0 of C is built differently, because C ~ y at the origin and a plain
// interpolant's error does not vanish with it: it is t times the degree-6 interpolant of C/y, which keeps the
// RELATIVE accuracy down to y = 0 (3e-15 there; the other rows and all of b: 2-3 ulp evaluated by Horner's rule in
// double; tests/test_host_cpu.py::test_poly_tables_accuracy).
// (Rounds 2-5 tabulated c(y) = (1 - e^-y)/y, on intervals centred on i/8, and paid three more multiplications per
// sample -- x^2, its product with b, y times c -- and three adds where the row and t now cost a convert and a fract.)
constexpr int kPolyDeg = 7;
constexpr int kPolyStride = 10;
constexpr int kPolyBCount = 48 * 8 + 1;        // beyond x = 48, b(x) = x e^-x to the last bit (the kernels' far branch)
constexpr int kPolyCCount = 37 * 8 + 1;        // beyond y = 37, 1 - e^-y = 1 (e^-37 < 2^-53): the kernels clamp y there
void build_poly_tables(std::vector<double> &b, std::vector<double> &c);

}  // namespace mbbh
