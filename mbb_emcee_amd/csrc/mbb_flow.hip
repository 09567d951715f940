// mbb_flow.hip -- the one-launch sampler runs (k_flowm, k_flowa; k_lnlike SMODE 6, sharded) in a translation unit
// of its own, because it wants other code generation than the rest of the library: the kernel
// is a loop over half-steps around two long dependent chains, and with the default pipeline the
// compiler hoists every loop-invariant value out of that loop, runs out of registers and
// reloads the spilled values inside the chains (172 bytes of scratch per lane).  Built with
//   -mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm
// it needs no scratch at all (mbb_emcee_amd/build.py).  The other kernels keep the default
// flags: measured, they gain nothing from these.
#include "mbb_kernels.hip.h"

#define MBB_FLOW_INST(OT, NA)                                            \
    template __global__ void k_lnlike<OT, NA, 6, false>(const LikeArgs); \
    template __global__ void k_lnlike<OT, NA, 6, true>(const LikeArgs);
MBB_FLOW_INST(false, false)
MBB_FLOW_INST(false, true)
MBB_FLOW_INST(true, false)
MBB_FLOW_INST(true, true)
#undef MBB_FLOW_INST

#include "mbb_flowm.hip.h"
#define MBB_FLOWM_INST(OT, NA)                                            \
    template __global__ void k_flowm<OT, NA, false, 1>(const LikeArgs); \
    template __global__ void k_flowm<OT, NA, true, 1>(const LikeArgs);
MBB_FLOWM_INST(false, false)
MBB_FLOWM_INST(false, true)
MBB_FLOWM_INST(true, false)
MBB_FLOWM_INST(true, true)
#undef MBB_FLOWM_INST

#include "mbb_flowa.hip.h"
#define MBB_FLOWA_INST(OT, NA)                                     \
    template __global__ void k_flowa<OT, NA, false>(const LikeArgs); \
    template __global__ void k_flowa<OT, NA, true>(const LikeArgs);
MBB_FLOWA_INST(false, false)
MBB_FLOWA_INST(false, true)
MBB_FLOWA_INST(true, false)
MBB_FLOWA_INST(true, true)
#undef MBB_FLOWA_INST

#include "mbb_serve.hip.h"
#define MBB_SERVE_INST(OT, NA)                                            \
    template __global__ void k_serve<OT, NA, false, false>(const LikeArgs); \
    template __global__ void k_serve<OT, NA, true, false>(const LikeArgs);  \
    template __global__ void k_serve<OT, NA, false, true>(const LikeArgs);  \
    template __global__ void k_serve<OT, NA, true, true>(const LikeArgs);
MBB_SERVE_INST(false, false)
MBB_SERVE_INST(false, true)
MBB_SERVE_INST(true, false)
MBB_SERVE_INST(true, true)
#undef MBB_SERVE_INST
