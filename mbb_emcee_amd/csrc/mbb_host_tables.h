// mbb_host_tables.h -- host-only table builders of the likelihood hot path.
//
// Plain C++ (no HIP types): compiled into libmbb_hip.so by hipcc and, on its own,
// by gcc with -fsanitize=address,undefined for the CPU test target
// (`make -C oracle asan`, tests/test_host_cpu.py).
//
//   build_band_layout  the chunk / unit / result-slot layout of the passband samples
//                      that k_lnlike reads (what response.__call__, response.py:572-576,
//                      does band by band becomes one table walked by all waves)
//   build_poly_tables  piecewise polynomials of the two smooth factors of the sample
//                      arithmetic, b(x) = x / expm1(x) and c(y) = (1 - exp(-y)) / y
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

// Piecewise degree-7 polynomials on intervals of width 1/8 centred on i/8:
//   b(x) = x / expm1(x)        i = 0 .. kPolyBCount-1   (x in [0, 64])
//   c(y) = (1 - exp(-y)) / y   i = 0 .. kPolyCCount-1   (y in [0, 40])
// eight coefficients per interval, lowest order first, in t = x - i/8 (|t| <= 1/16).
// Each polynomial interpolates the function at the eight Chebyshev nodes of its
// interval (computed in long double); evaluated by Horner's rule in double they agree
// with the function to 2 ulp (tests/test_host_cpu.py::test_poly_tables_accuracy).
constexpr int kPolyDeg = 7;
constexpr int kPolyBCount = 64 * 8 + 1;
constexpr int kPolyCCount = 40 * 8 + 1;
void build_poly_tables(std::vector<double> &b, std::vector<double> &c);

}  // namespace mbbh
