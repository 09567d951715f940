// The SED constructor (sed_prologue, mbb_device.hip.h) as one dependent chain on ONE row of 16 lanes of one wave on an
// otherwise idle CU -- how it runs in k_serve / k_lnlike with a walker per workgroup -- timed in s_memtime ticks per call,
// whole and in pieces: each piece in a loop of its own whose next input depends on the last result.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/lat_ctor tools/lat_ctor.hip && tools/lat_ctor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../mbb_emcee_amd/csrc/mbb_kernels.hip.h"
using namespace mbbd;

constexpr double kNunorm = 299792.458 / 500.0;

template <int WHAT>
__global__ void piece(double *out, int n, double T, double beta, double lam0, double alpha, double fnorm)
{
    __shared__ Exp2Entry s_tab[kExp2N];
    for (int i = threadIdx.x; i < kExp2N; i += blockDim.x) s_tab[i] = kExp2Tab[i];
    __syncthreads();
    const double lnunorm = log(kNunorm);
    double acc = 0.0, carry = 0.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        const double Ti = T + carry, li = lam0 + carry;
        double r = 0.0;
        if (WHAT == 0) {                                   // the two logs
            double lo[2];
            vlog<true>(lo, Ti, li);
            r = lo[0] + lo[1];
        } else if (WHAT == 1) {                            // fp32 bracket + secant only
            const float lx0 = (float)(-3.0 - 1e-3 * Ti);
            r = (double)thick_merge_root_f32<true>((float)alpha, (float)beta, lx0, 1.6f, 2.06f);
        } else if (WHAT == 2) {                            // the root: fp32 stage + fp64 Newton (with the ride-along exps)
            int st; double xr, yr, pb[4] = {0.1, 2.0 + 1e-3 * Ti, 3.0, 0.0}, kf;
            r = thick_merge_root<true, true>(alpha, beta, -3.0 - 1e-3 * Ti, st, xr, yr, pb, nullptr, &kf);
            r += xr + yr + kf + pb[0] + pb[1] + pb[2] + pb[3];
        } else if (WHAT == 3) {                            // the whole constructor, logs included
            double lo[2];
            vlog<true>(lo, Ti, li);
            SedScalars s;
            int it;
            const int st = sed_prologue<false, false, true>(Ti, beta, alpha, fnorm, lo[0], lo[1], kNunorm, lnunorm, s, &it);
            WalkerK k;
            make_walker_k<false, false>(beta, alpha, s, k);
            r = k.cbb + k.cpl + k.xmerge + k.lx0 + k.hokt9 + st;
        } else if (WHAT == 4) {                            // one vexp round of 6 (what a Newton evaluation does twice)
            double o[6];
            vexp<true, 0x08u>(o, 1.8 + 1e-3 * Ti, 0.3 + 1e-3 * Ti, 0.1, 2.0, 3.0, 9.0);
            r = o[0] + o[1] + o[2] + o[3] + o[4] + o[5];
        } else if (WHAT == 5) {                            // one division
            r = m_div(1.0 + Ti, 3.0 + li);
        }
        acc += r;
        carry = (r != r) ? 1e-9 : r * 1e-300;              // (depends on the result, changes nothing)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = (double)(t1 - t0) / n; out[1] = acc; }
}

// The constructor as the kernels run it: gate, sed_prologue, make_walker_k, parameter-only penalties (mbb_walker_consts.inc) with
// the limits and priors in the kernel argument block, the row's five values arriving from memory, the record going to LDS.
struct LimBlock { double lowlim[5], uplim[6], gmean[6], givar[6], nunorm, lnunorm; unsigned has_uplim, has_gprior; };
struct LimView { const double *lowlim, *uplim, *gmean, *givar; double nunorm, lnunorm; unsigned has_uplim, has_gprior; };
template <bool OPTHIN, bool NOALPHA, int VAR>
__global__ void __launch_bounds__(1024) ctor_text(const LikeArgs a, double *out, int n)
{
    __shared__ WalkerK kfin;
    __shared__ double pen[2];
    __shared__ LimBlock s_lim;
    const int tid = threadIdx.x;
    if (tid == 0) {
        for (int i = 0; i < 5; ++i) s_lim.lowlim[i] = a.lowlim[i];
        for (int i = 0; i < 6; ++i) { s_lim.uplim[i] = a.uplim[i]; s_lim.gmean[i] = a.gmean[i]; s_lim.givar[i] = a.givar[i]; }
        s_lim.nunorm = a.nunorm; s_lim.lnunorm = a.lnunorm; s_lim.has_uplim = a.has_uplim; s_lim.has_gprior = a.has_gprior;
    }
    __syncthreads();
    double acc = 0.0, carry = 0.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (tid < 16) {
        for (int it = 0; it < n; ++it) {
            double p[5];
            if (VAR == 0 || VAR == 3) {
                const double pe = tid < 5 ? __hip_atomic_load(a.pars + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + carry : 0.0;
#pragma unroll
                for (int i = 0; i < 5; ++i) p[i] = __shfl(pe, i);
            } else {
                p[0] = a.gmean[0] + carry; p[1] = a.gmean[1] + carry; p[2] = a.gmean[2] + carry; p[3] = a.gmean[3] + carry; p[4] = a.gmean[4] + carry;
            }
            double lo[2];
            vlog<true>(lo, p[0], p[2]);
            const double lT = lo[0], lL = lo[1];
            WalkerK k;
            k.status = ROW_SKIP; k.pad = 0;
            double pen_u = 0.0, pen_g = 0.0;
#define STAMPD(i, dep) do { } while (0)
            if (VAR == 4 || VAR == 5) {
                // the bare constructor in this kernel's context (the argument block by value): no gate, no penalties
                SedScalars s;
                k.status = sed_prologue<OPTHIN, NOALPHA, true>(p[0], p[1], p[3], p[4], lT, lL, a.nunorm, a.lnunorm, s, VAR == 5 ? &k.pad : nullptr);
                make_walker_k<OPTHIN, NOALPHA>(p[1], p[3], s, k);
            } else if (VAR == 6) {
                // ... with the gate in front
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 5; ++i) ok = ok && !(p[i] < a.lowlim[i]);
                if (!ok) k.status = ROW_BELOW_LOWLIM;
                else if (!finite5(p)) k.status = ROW_NONFINITE;
                else {
                    SedScalars s;
                    k.status = sed_prologue<OPTHIN, NOALPHA, true>(p[0], p[1], p[3], p[4], lT, lL, a.nunorm, a.lnunorm, s, nullptr);
                    if (k.status == ROW_OK) make_walker_k<OPTHIN, NOALPHA>(p[1], p[3], s, k);
                }
            } else if (VAR >= 2) {
                // the limits and priors through LDS instead of the argument block's scalar registers
                const LimView a = {s_lim.lowlim, s_lim.uplim, s_lim.gmean, s_lim.givar, s_lim.nunorm, s_lim.lnunorm,
                                   (unsigned)__builtin_amdgcn_readfirstlane((int)s_lim.has_uplim), (unsigned)__builtin_amdgcn_readfirstlane((int)s_lim.has_gprior)};
#include "../mbb_emcee_amd/csrc/mbb_walker_consts.inc"
            } else {
#include "../mbb_emcee_amd/csrc/mbb_walker_consts.inc"
            }
            if (tid == 0) { kfin = k; pen[0] = pen_u; pen[1] = pen_g; }
            acc += k.cbb + k.xmerge;
            carry = (k.cbb != k.cbb) ? 1e-9 : k.cbb * 1e-300;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { out[0] = (double)(t1 - t0) / n; out[1] = acc + kfin.cbb + pen[0]; }
}

int main()
{
    double *o; hipMalloc(&o, 16);
    double h[2];
    const char *nm[] = {"two logs (vlog)", "fp32 bracket + secant", "root: fp32 + fp64 Newton", "whole constructor", "one vexp round of six", "one m_div"};
#define RUN(K) for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(piece<K>, dim3(1), dim3(64), 0, 0, o, 2000, 12.3, 1.8, 600.0, 3.0, 40.0); hipDeviceSynchronize(); } \
    hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); printf("%-28s %8.1f ticks per call   (check %.6g)\n", nm[K], h[0], h[1] / 2000);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    {
        LikeArgs a;
        memset(&a, 0, sizeof a);
        double hp[5] = {12.3, 1.8, 600.0, 3.0, 40.0}, *dp;
        hipMalloc(&dp, 40); hipMemcpy(dp, hp, 40, hipMemcpyHostToDevice);
        a.pars = dp;
        a.nunorm = kNunorm; a.lnunorm = log(kNunorm);
        const double low[5] = {1, 0.1, 1, 0.1, 1e-3}, up[6] = {INFINITY, 20.0, 3300.0, 20.0, INFINITY, INFINITY};
        for (int i = 0; i < 5; ++i) a.lowlim[i] = low[i];
        for (int i = 0; i < 6; ++i) { a.uplim[i] = up[i]; a.givar[i] = 1.0; }
        a.has_uplim = (1u << 1) | (1u << 2) | (1u << 3);
        for (int i = 0; i < 5; ++i) a.gmean[i] = hp[i];
#define RUNT(VAR, what) for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((ctor_text<false, false, VAR>), dim3(1), dim3(64), 0, 0, a, o, 2000); hipDeviceSynchronize(); } \
        hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); printf("%-60s %8.1f ticks per call   (check %.6g)\n", what, h[0], h[1] / 2000);
        RUNT(0, "constructor text, row from memory, walls in the argument block")
        RUNT(1, "... row from the argument block")
        RUNT(2, "... row from the argument block, walls and priors through LDS")
        RUNT(3, "... row from memory, walls and priors through LDS")
        a.has_uplim = 0;
        RUNT(1, "no upper walls: row from the argument block")
        RUNT(0, "no upper walls, row from memory")
        RUNT(2, "no upper walls, through LDS")
        RUNT(4, "bare sed_prologue + make_walker_k in this kernel")
        RUNT(5, "... with the iteration count stored")
        RUNT(6, "... with the gate in front, no penalties")
    }
    return 0;
}
