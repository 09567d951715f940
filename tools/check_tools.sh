#!/bin/bash
# Runs every Python tool once with its defaults (or small arguments) under a time limit and records the return code:
#     bash tools/check_tools.sh       (GPU box; needs tools/libmbb_hip_stamps.so = the -DMBB_STAMPS -DMBB_STAMPS_FINE build and
#                                      tools/libmbb_hip_cur.so = any second build of the library for the A/B tools)
# -> gpurun_out/tools_check.txt.  A tool that has rotted against the library shows here, not when it is needed.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/tools_check.txt
: > $out
run() { local t0=$(date +%s); timeout -k 5 ${LIMIT:-100} "$@" > gpurun_out/tools_check_last.log 2>&1; local rc=$?; echo "rc=$rc $(( $(date +%s) - t0 ))s  $*" >> $out; if [ $rc -ne 0 ]; then tail -4 gpurun_out/tools_check_last.log | sed 's/^/      /' >> $out; fi; }
S=tools/libmbb_hip_stamps.so
run python3 tools/probe_stamps.py
run python3 tools/probe_serve_stamps.py 125 1
run python3 tools/probe_stamps_flowm.py
run python3 tools/probe_chain_flowm.py
run python3 tools/probe_chain_flowa.py
run python3 tools/probe_flowm_start.py 20
run python3 tools/probe_flowm.py
run python3 tools/probe_flowm_short.py
run python3 tools/probe_timed_region.py
run python3 tools/probe_served_boundary.py
run python3 tools/probe_serve_overlap.py
run python3 tools/probe_host_phases.py
run python3 tools/probe_boundary_breakdown.py
run python3 tools/probe_first_fit.py
run python3 tools/probe_clock.py
run python3 tools/probe_traffic_calib.py
run python3 tools/probe_prepass.py
run python3 tools/probe_pool.py 2 300
run python3 tools/probe_root_sweep.py
run python3 tools/sweep_geometry.py
run python3 tools/sweep_walkers.py 512
run python3 tools/bench_cfg5.py --quick
run python3 tools/bench_configs.py cfg1
run python3 tools/ab_stage.py
run python3 tools/ab_boundary.py
run python3 tools/ab_sampler.py
run python3 tools/ab_two_libs.py tools/libmbb_hip_cur.so mbb_emcee_amd/libmbb_hip.so
run python3 tools/ab_m1.py tools/libmbb_hip_cur.so mbb_emcee_amd/libmbb_hip.so 1
run python3 tools/ab_sampler_step.py tools/libmbb_hip_cur.so mbb_emcee_amd/libmbb_hip.so 1
run python3 tools/ab_option.py serve_prefetch 0 32 --rounds 1
run python3 tools/probe_other_streams.py
run python3 tools/soak_flowm_sizes.py 200
run python3 tools/soak_resident_sizes.py 100 300
run python3 tools/soak_random_configs.py 1 2
run python3 tools/soak_served_options.py
run python3 tools/soak_served_random.py 1 20
cat $out
