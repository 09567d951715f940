/*
 * mbb_hip.h -- C-ABI of the MI355X (gfx950) likelihood hot path.
 *
 * This is the drop-in boundary for the per-walker likelihood of mbb_emcee:
 * plain pointers and sizes, no C++ or framework types.  Every entry point
 * names the reference interface it replaces (paths relative to the
 * reference's mbb_emcee/ directory).  The reference-side binding (ctypes)
 * is shown in INTEGRATION.md; mbb_emcee_amd/_native.py is that binding.
 *
 * Conventions
 *   - all floating point data is IEEE float64, arrays are C-contiguous;
 *   - parameter rows are (T, beta, lambda0, alpha, fnorm)   likelihood.py:20-22;
 *   - the caller owns every buffer passed in; nothing is retained after return
 *     except what the set_* calls copy to the device;
 *   - functions return 0 on success, a negative mbb_error on failure, and never
 *     throw; mbb_last_error() gives the text of the last failure in this thread;
 *   - one context = one device + one HIP stream; a context is not thread-safe,
 *     any number of contexts may coexist;
 *   - per-row status codes (int32): see mbb_row_status.
 */
#ifndef MBB_HIP_H
#define MBB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mbb_ctx mbb_ctx;

enum mbb_error {
    MBB_OK = 0,
    MBB_ERR_HIP = -1,        /* a HIP runtime call failed (no GPU, OOM, ...) */
    MBB_ERR_ARG = -2,        /* invalid argument                             */
    MBB_ERR_STATE = -3,      /* bands / data not set yet                     */
    MBB_ERR_RCCL = -4        /* RCCL missing or a collective failed          */
};

/* Per-row outcome.  Rows with status >= 2 correspond to Python exceptions in
 * the reference's SED constructor (modified_blackbody.py:219-224, :294-316). */
enum mbb_row_status {
    MBB_ROW_OK = 0,
    MBB_ROW_BELOW_LOWLIM = 1, /* lnL = -inf          likelihood.py:806-807   */
    MBB_ROW_BAD_ALPHA = 2,    /* alpha <= 0          modified_blackbody.py:219 */
    MBB_ROW_BAD_BETA = 3,     /* beta < 0            modified_blackbody.py:222 */
    MBB_ROW_NOCONV = 6,       /* merge / peak root not found                 */
    MBB_ROW_NONFINITE = 7     /* NaN/inf parameter: lnL = NaN, as the reference */
};

const char *mbb_last_error(void);
int mbb_device_count(void);

/* ---- context ----------------------------------------------------------- */
/* Replaces: construction of `likelihood` + `modified_blackbody` state
 * (likelihood.py:24-117).  opthin / noalpha / wavenorm as in likelihood.py:62-64. */
int mbb_ctx_create(int device, mbb_ctx **out);
void mbb_ctx_destroy(mbb_ctx *ctx);
int mbb_set_model(mbb_ctx *ctx, int opthin, int noalpha, double wavenorm);

/* Replaces: the per-band state built by response.setup (response.py:252-332)
 * and consumed by response.__call__ (response.py:572-576).
 *   freq   [offsets[nb]]  sample frequencies in GHz (response._freq)
 *   weight [offsets[nb]]  response._sedmult * response._normfac; a delta band
 *                         (response.py:336-374) or a plain photometric
 *                         wavelength (likelihood.py:817) is one sample, weight 1
 *   offsets[nb+1]         band b owns samples offsets[b] .. offsets[b+1]-1   */
int mbb_set_bands(mbb_ctx *ctx, const double *freq, const double *weight,
                  const int32_t *offsets, int nb);

/* Replaces: likelihood.set_phot / set_cov state (likelihood.py:158-232, :330-357).
 * is_cov == 0: w is ivar[nb] = 1/unc^2; is_cov != 0: w is inverse covariance [nb*nb]. */
int mbb_set_data(mbb_ctx *ctx, const double *flux, const double *w, int nb, int is_cov);

/* Batched multi-source mode (BASELINE.json configs[4]): nsrc independent SEDs
 * observed through the same bands, flux / ivar [nsrc*nb] (diagonal errors).
 * Afterwards a batch of n rows is read as nsrc groups of n/nsrc consecutive rows,
 * group g being compared with source g's data.  The reference fits one source
 * per process (likelihood.py:158-232 holds one flux vector); this is the same
 * likelihood applied to many. */
int mbb_set_data_multi(mbb_ctx *ctx, const double *flux, const double *ivar, int nb, int nsrc);

/* Replaces: _lowlim / _has_uplim / _uplim (likelihood.py:73, :83-85, :227-229);
 * index 5 is the lambda_peak ghost parameter (likelihood.py:710-715). */
int mbb_set_limits(mbb_ctx *ctx, const double lowlim[5], const int32_t has_uplim[6],
                   const double uplim[6]);
/* Replaces: Gaussian priors (likelihood.py:87-92, :513-541, :719-752). */
int mbb_set_gpriors(mbb_ctx *ctx, const int32_t has[6], const double mean[6],
                    const double ivar[6]);

/* ---- the hot path ------------------------------------------------------- */
/* Replaces: n calls of likelihood.__call__(pars) (likelihood.py:790-834), i.e.
 * what emcee's map(lnprobfn, rows) does per half-step (mbb_fit.py:80-81).
 * pars [n*5] host, lnl [n] host, status [n] host (may be NULL),
 * model_flux [n*nb] host (may be NULL).  Synchronous. */
int mbb_lnlike_batch(mbb_ctx *ctx, const double *pars, int n, double *lnl,
                     int32_t *status, double *model_flux);

/* The same boundary call with the copies taken out (what likelihood.__call__ uses for an (n, 5) array,
 * likelihood.py:790-834, once per emcee half-step).  mbb_boundary_buffers returns host addresses that hold
 * until more than nmax rows are asked for or the context goes: *in -- the caller writes its parameter rows
 * [n x 5] straight INTO it (device memory behind the PCIe BAR where there is a large BAR, else the pinned
 * block the kernel reads); *out / *status (may be NULL) -- pinned blocks the kernel writes lnprob [n] and row
 * status [n] to.  mbb_lnlike_call(ctx, n) evaluates the first n rows of *in: same kernel, same results as
 * mbb_lnlike_batch, bit for bit.  It returns MBB_OK, a negative error (MBB_ERR_STATE: ask for the buffers
 * again -- capacity or the zero_copy / bar_params options changed), or, positive, the row status of the first
 * row the reference would have raised ValueError for (2 alpha <= 0, 3 beta < 0, 6 no merge point:
 * modified_blackbody.py:219-224, :294-316); rows below a lower limit are -inf with status 1, as ever. */
int mbb_boundary_buffers(mbb_ctx *ctx, int nmax, double **in, double **out, int32_t **status);
int mbb_lnlike_call(mbb_ctx *ctx, int n);
/* The address of a word that lives as long as the context and changes whenever the blocks mbb_boundary_buffers
 * handed out are freed (ANY entry point of the context asked to hold more rows than they do: mbb_lnlike_batch,
 * mbb_lnlike_allgather, ...).  A binding that caches the addresses reads the word right after mbb_boundary_buffers
 * and compares it BEFORE every write through them; a different value means: ask for the buffers again. */
const unsigned long long *mbb_boundary_generation(mbb_ctx *ctx);

/* Same computation on device-resident buffers, enqueued on the context's
 * stream, asynchronous.  d_model_flux / d_status may be NULL. */
int mbb_lnlike_batch_device(mbb_ctx *ctx, const double *d_pars, int n,
                            double *d_lnl, int32_t *d_status, double *d_model_flux);

/* Measurement helper: the same launch enqueued `reps` times back to back from
 * C (no host work in between), so that HIP events around the call time the
 * kernel itself.  Asynchronous. */
int mbb_lnlike_repeat_device(mbb_ctx *ctx, const double *d_pars, int n, double *d_lnl,
                             int32_t *d_status, int reps);

/* Measurement helper (bench.py roofline): the sample arithmetic of
 * fnu.pyx:9-108 alone -- every wave of a chip-filling grid walks all passband
 * samples `reps` times with the constants of one parameter row; no prologue, no
 * reductions.  lane_slots = samples evaluated including chunk padding; clock_mhz (may be
 * NULL) = the shader clock the chip held meanwhile, s_memtime against the 100 MHz
 * s_memrealtime over workgroup 0's loop.  Synchronous. */
int mbb_roof_probe(mbb_ctx *ctx, const double pars[5], int reps, double *seconds,
                   double *lane_slots, double *clock_mhz);

/* ---- device-resident ensemble sampler ------------------------------------- */
/* Replaces: emcee.EnsembleSampler(nwalkers, 5, like).run_mcmc(p0, nsteps) as
 * driven by mbb_fitter.run (mbb_fit.py:80-81, :533, :542): the affine-invariant
 * stretch move (Goodman & Weare 2010) with the proposal, the fused likelihood
 * and the accept/reject step of a half-ensemble in ONE kernel launch, 2 nsteps
 * dependent launches per call and no host round trip in between.  emcee is not
 * part of the reference tree, so parity is statistical.
 * chain [nw][nsteps][5] and lnprob [nw][nsteps] use emcee's layout
 * (results.py:154-155).  After mbb_set_data_multi the sampler advances nsrc
 * independent ensembles of nwalkers each in the same launches; every array then
 * has a leading nsrc dimension.  "Fixed" parameters work as in the reference: a column
 * of p0 with zero scatter is preserved exactly by the stretch move
 * (mbb_fit.py:442-443). */
int mbb_sampler_create(mbb_ctx *ctx, int nwalkers, unsigned long long seed, void **sampler);
int mbb_sampler_destroy(mbb_ctx *ctx, void *sampler);
int mbb_sampler_reset(mbb_ctx *ctx, void *sampler);          /* zero the acceptance counts */
int mbb_sampler_set_state(mbb_ctx *ctx, void *sampler, const double *pos, const double *lnprob);
int mbb_sampler_run(mbb_ctx *ctx, void *sampler, int nsteps, double stretch_a, double *chain,
                    double *lnprob, double *pos_out, double *lnprob_out, double *naccepted);
int mbb_sampler_advance_async(mbb_ctx *ctx, void *sampler, int nsteps, double stretch_a);
/* Measurement helper (bench.py's timed region on one GPU): nsteps steps as mbb_sampler_advance_async
 * enqueues them, bracketed inside ONE call by the host clock and by two events on the context's stream,
 * recorded right before the run's first launch and right behind its last: clock; enqueue; stream wait; clock.
 * The stream must be idle on entry (mbb_sync before).  wall_s = host seconds from before the enqueue to after
 * the wait; stream_ms = between the events.  No reference counterpart. */
int mbb_sampler_advance_timed(mbb_ctx *ctx, void *sampler, int nsteps, double stretch_a, double *wall_s,
                              float *stream_ms);

/* ---- SED-level entry points (parity + the modified_blackbody class) ----- */
/* Replaces: modified_blackbody.__init__ (modified_blackbody.py:168-337) and
 * max_wave (:581-637) for n parameter rows.
 * out [n*6] = normfac, xmerge, kappa, x0, wavemerge, max_wave (NaN where the
 * reference has None); want_peak == 0 skips max_wave. */
int mbb_sed_prologue_batch(mbb_ctx *ctx, const double *pars, int n, int opthin,
                           int noalpha, double wavenorm, int want_peak,
                           double *out, int32_t *status);

/* Replaces: modified_blackbody.__call__ / f_nu (modified_blackbody.py:441-554)
 * for n parameter rows on a common grid of m frequencies (GHz): out [n*m]. */
int mbb_sed_eval_batch(mbb_ctx *ctx, const double *pars, int n, int opthin,
                       int noalpha, double wavenorm, const double *freq, int m,
                       double *out, int32_t *status);

/* Replaces: modified_blackbody.freq_integrate (modified_blackbody.py:639-674, scipy
 * quad of f_nu) for n rows: out[n] = integral of f_nu d nu over [numin, numax] GHz,
 * in mJy GHz (the reference multiplies by 1e-17 to get erg/s/cm^2).  Used by the
 * chain post-processing (results.py:627-674, L_IR). */
int mbb_sed_integrate_batch(mbb_ctx *ctx, const double *pars, int n, int opthin, int noalpha,
                            double wavenorm, double numin, double numax, double *out,
                            int32_t *status);

/* Replaces: fnu.fnueval_{thin,thick}_{noalpha,walpha} (fnu.pyx:9-108) with the
 * same explicit scalars; unused ones are ignored. */
int mbb_fnu_eval(mbb_ctx *ctx, int opthin, int noalpha, const double *freq, int n,
                 double T, double beta, double x0, double alpha, double normfac,
                 double xmerge, double kappa, double *out);

/* ---- device plumbing for callers that keep data resident ---------------- */
int mbb_malloc(mbb_ctx *ctx, size_t bytes, void **dptr);
int mbb_free(mbb_ctx *ctx, void *dptr);
int mbb_memcpy_h2d(mbb_ctx *ctx, void *dst, const void *src, size_t bytes);
int mbb_memcpy_d2h(mbb_ctx *ctx, void *dst, const void *src, size_t bytes);
int mbb_sync(mbb_ctx *ctx);
void *mbb_stream(mbb_ctx *ctx);             /* hipStream_t of the context     */
/* HIP events on the context's stream, for timing launches where they run */
int mbb_event_create(mbb_ctx *ctx, void **ev);
int mbb_event_record(mbb_ctx *ctx, void *ev);
int mbb_event_elapsed_ms(mbb_ctx *ctx, void *start, void *stop, float *ms);
int mbb_event_destroy(mbb_ctx *ctx, void *ev);

/* Tunables: "walkers_per_group" (0 = auto), "block_threads" (0 = auto),
 * "zero_copy" (host path reads/writes pinned host memory from the kernel),
 * "spin_wait" (how mbb_lnlike_batch waits: 0 blocks on the stream, 1 polls it,
 * 2 -- the default -- watches the result slots in pinned memory, which are final
 * before the kernel's completion signal is; "spin_budget" = polls before it falls back
 * to blocking on the stream, 0 forces the fallback), "bar_params" (host path writes the
 * parameter rows into device memory through the PCIe BAR), "serve" (default 1: after "serve_after" (3) boundary calls in a row
 * with nothing else in between -- a sampler's loop -- mbb_lnlike_call hands its rows to a kernel that STAYS on the GPU
 * between the calls and is rung through the BAR (k_serve: no launch per call; same results bit for bit), while a batch
 * is at most two rows per CU (a workgroup of the kernel takes a row; a call of more rows than it has workgroups, two).  One such kernel per device and process: any other entry point on the context, and any entry
 * point of ANOTHER context of the process that comes to the device (which also ends this context's run of calls, and
 * doubles, up to 64, the calls in a row it needs before its next server -- back to "serve_after" once a server has
 * answered 256 requests: likelihoods used in turns do not spend their time starting and stopping kernels), tells it to
 * leave first; it leaves by itself "serve_idle_us" (1000) after the last request; a request
 * whose results do not appear within "serve_budget_us" (400) is evaluated by a launch instead and three such in a row
 * rest the feature for the next 4096 boundary calls (mbb_get_info "serve_rests": how often, "serve_resting": calls left).  A server holds a CU per workgroup, and nothing of another PROCESS fits on those (emcee's pool,
 * mbb_fit.py:80-81 with threads > 1): so it is as wide as the calls have rows (in eights; "serve_grid" > 0: that many
 * workgroups; mbb_get_info "serve_grid": the resident one's) and no wider than this process's share of the device -- the CUs
 * divided (in whole rows of the 8 XCDs) by the processes of this library that are making boundary calls on it right now (a table in POSIX shared memory
 * keyed by the device's PCI address, each process noting its calls: csrc/mbb_registry.h; mbb_get_info "device_peers": registered, "device_busy": calling).  Two
 * workers of 125 rows have their servers side by side; a server narrower than a call has rows takes two rows a workgroup
 * (three workers: 80 CUs each); a call of more than twice the share's rows goes by a launch
 * ("serve_peer_yields"); a server too wide for the share or too narrow for the call leaves and the next starts with the same
 * call ("serve_resizes").  For processes that table cannot show, a server is sent away after "serve_lease_us" (50000;
 * 0: never) in one go -- the rows of that call go by a launch, the next server starts after the next few calls in a row
 * ("serve_lease_yields").
 * "serve" 0: a launch per call; 2: the whole device is this process's share whoever else is there (tests).  "serve_prefetch" (32; 0 to
 * 256): once a request's first record has turned the host asks for that many record lines ahead of its scan.  "prepass" (-1 auto: from
 * 64 rows per CU; 0 never; 1 always): a likelihood launch of given rows is preceded by k_walker_pre, which works out gate, SED
 * constructor and penalties with a LANE per row (k_lnlike: a row of 16 lanes per walker -- right where a constructor is
 * latency, a quarter of all instructions where a launch is bound by their number); same values bit for bit
 * (mbb_get_info "last_prepass").  "serve_overlap" (default 1): that
 * kernel starts a row's passband quadrature beside its SED constructor -- the blackbody-side value of every sample,
 * which needs none of the constructor's merge point, into a buffer in LDS -- and sums the units from the buffer when
 * the constructor is through, when the bands have at least 12 chunks of 64 samples (2: with fewer too); 0: one after
 * the other, as a launch does it; the same results either way.
 * mbb_get_info "serving", "serve_requests", "serve_fallbacks"), "launch_api" (how the likelihood launch of given rows
 * is handed to the runtime: 1, the default, hipModuleLaunchKernel with the argument block as one packed buffer;
 * 0 hipLaunchKernel -- 0.2 us more per call: profiles/r04/boundary_breakdown.txt), "seg_chunks", "pack_tails" (band
 * leftovers share chunks; takes effect at the next mbb_set_bands), "stage_tables",
 * "virtual_ranks", "debug", "roof_threads" / "roof_wgs_per_cu" (measurement only: the geometry of
 * mbb_roof_probe), "xchg_spin_max" (polls before a launch waiting for a peer
 * gives up); the forms of the single-GPU sampler, same chains bit for bit (which one a run takes by default, and
 * what each costs: sampler_enqueue in mbb_hip.hip, profiles/r04/walker_sweep.txt):
 * "lookahead_sampler" (default 1; 0: the plain train of one launch per half-step),
 * "flow_sampler" (default 1: one launch per 4096 steps, every workgroup resident, rows handed over through check
 * words instead of a launch boundary; 0: the plain train as well),
 * "flow_min_steps" (runs shorter than this take the plain train: a one-launch run costs ~14 us
 * beside its steps; default from profiles/r03/flowm_short_runs.txt),
 * "merged_flow_sampler" (default 1: ensembles of up to two walkers per CU take "form 7", k_flowm -- one workgroup per
 * pair of walkers and candidate; the passband quadrature of both proposals a walker can end up making, and the SED
 * constructor for every outcome still open, run ahead of the decisions they depend on; 0: those ensembles take the
 * resident forms below too),
 * "resident_sampler" (default 1: ensembles beyond that run as ONE launch per 4096 steps as well, a workgroup owning
 * W = ceil(half / CUs) walkers of each half, up to 8; 0: off, the plain train; 2: every eligible ensemble takes it),
 * that run -- "form 9", k_flowa -- constructs every walker's proposal a half-step ahead, for both outcomes of its partner's
 * pending move, beside the quadrature of the half-step before ("resident_ahead", round 4's choice between it and a form
 * with nothing ahead, is accepted and ignored: that form is gone),
 * "resident_walkers" (walkers per workgroup and half of the resident forms; 0 = the host's choice),
 * "lookahead_rows" / "lookahead_waves" (the sharded one-launch run: 0 = the host's choice of candidates per wave and
 * waves per workgroup among those that work ahead), "sharded_flow_sampler" (default 1: a sharded run with the
 * one-hop exchange is one launch per 4096 steps on every rank too; 0: one launch per half-step),
 * "flow_spin_log2" (0 = 22: log2 of the polls before a wait -- 63: no polls, the first miss, for tests --
 * inside the one-launch run gives up; mbb_sampler_run then redoes the run as a launch train and
 * counts it in mbb_get_info "flow_fallbacks"; the next run takes the one-launch form again, and only
 * three give-ups in a row rest it for the next 16 runs -- mbb_get_info "flow_resting" says how many of
 * those are left.  mbb_sampler_advance_async keeps nothing to redo a run from: there a give-up surfaces
 * at the next mbb_sampler_run as MBB_ERR_STATE and the sampler wants mbb_sampler_set_state again).
 * mbb_get_info "last_kernel_form" says which form ran. */
int mbb_set_option(mbb_ctx *ctx, const char *name, long value);
int mbb_get_info(mbb_ctx *ctx, const char *name, long *value);

/* ---- multi-GPU: one process per GPU, lnprob all-gather over RCCL -------- */
/* Replaces: emcee's multiprocessing pool selected by threads= (mbb_fit.py:80-81).
 * id is an opaque 128-byte ncclUniqueId made on rank 0 and handed to the other
 * ranks by the launcher (any side channel).  The collective library is librccl as the process
 * finds it, or -- environment variable MBB_RCCL_LIB -- the one library named there and no other
 * (a name that does not load is MBB_ERR_RCCL; the one-GPU tests name a stand-in that lets ranks
 * share a device: tests/rccl_standin/). */
int mbb_comm_unique_id(char id[128]);
int mbb_comm_init(mbb_ctx *ctx, int nranks, int rank, const char id[128]);
int mbb_comm_destroy(mbb_ctx *ctx);
/* every rank contributes count doubles; d_recv holds nranks*count, rank-major */
int mbb_allgather_f64(mbb_ctx *ctx, const double *d_send, double *d_recv, int count);
/* one sharded half-step: fused kernel on this rank's n rows, then the all-gather
 * of their lnprob into d_all [nranks*n]; asynchronous on the context's stream */
int mbb_lnlike_allgather_device(mbb_ctx *ctx, const double *d_pars, int n, double *d_lnl,
                                int32_t *d_status, double *d_all);
/* the same half-step as one synchronous call on host arrays -- literally what emcee's pool does per
 * half-step (mbb_fit.py:80-81): pars [n x 5] are this rank's rows, all [nranks*n] receives every rank's
 * lnprob (rank-major), status [n] (may be NULL) this rank's row status.  No copy command for the
 * parameters or the status (BAR / pinned memory as in mbb_lnlike_batch), one in-place ncclAllGather,
 * one copy of the gathered vector into a pinned landing buffer, one stream wait.  Without a communicator
 * (one rank) it is mbb_lnlike_batch through the same buffers. */
int mbb_lnlike_allgather(mbb_ctx *ctx, const double *pars, int n, double *all, int32_t *status);

/* ---- multi-GPU, device-resident sampler: one-hop exchange of the moved walkers ----- */
/* Replaces: emcee's multiprocessing pool (mbb_fit.py:80-81) for the device-resident
 * sampler, without a collective library.  Every rank keeps the whole ensemble in a
 * fine-grained buffer that all peers map (hipIpc*): the lane that accepts a move stores
 * the new state row into every rank's copy, and the launch's last walker raises the
 * launch number in every peer's flag word, which the next launch's prologue waits for.
 * Set-up: mbb_xchg_open on every rank -> exchange the 64-byte handles by any side
 * channel -> mbb_xchg_connect on every rank -> mbb_sampler_create (the sampler then lives
 * in the exchange buffer; nwalkers <= max_rows).  The launcher must also synchronise
 * the ranks between mbb_sampler_set_state and the first mbb_sampler_run / advance, and
 * before a later set_state.  In this mode mbb_sampler_run fills chain / lnprob /
 * naccepted for this rank's walkers only (pos_out and lnprob_out are complete).
 * mbb_xchg_close is refused while a sampler created under the exchange is alive. */
int mbb_xchg_open(mbb_ctx *ctx, int nranks, int rank, int max_rows, unsigned char handle[64]);
int mbb_xchg_connect(mbb_ctx *ctx, const unsigned char *handles /* nranks x 64 */);
int mbb_xchg_close(mbb_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MBB_HIP_H */
