#!/bin/bash
# Round 3 additions to the profile set (tools/run_profiles.sh collects the bench workload's):
#   bash tools/run_profiles_r03.sh <tag>
# VALU counter passes of cfg1 and cfg4 (tools/bench_configs.py --profile: 400 plain launches of a
# half-ensemble + 350 sampler steps each) -> gpurun_out/<tag>/pmc_valu_cfg{1,4}.json, and the kernel
# trace of the same commands.  Counter passes carry --kernel-trace only.
set -o pipefail
tag=${1:-prof_r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
export MBB_BENCH_FULL_LINE=1     # (the summarizers read legs that the driver's short line leaves to the side file)
cd $R
VALU="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
HS=$((2 * (50 + 300)))
for cfg in cfg1 cfg4; do
  timeout -k 10 300 rocprofv3 --pmc $VALU --kernel-trace --output-format csv -d $O/pmc_valu_$cfg -- python3 tools/bench_configs.py $cfg --profile > $O/valu_$cfg.log 2>&1 || exit 1
  python3 tools/summarize_valu.py $O/pmc_valu_$cfg $O/pmc_valu_$cfg.json "rocprofv3 --pmc $VALU --kernel-trace -- python3 tools/bench_configs.py $cfg --profile" "400 plain launches of a half-ensemble, then 350 steps of the device sampler in its default form" $HS > /dev/null || exit 2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python3 tools/bench_configs.py $cfg --profile > $O/stats_$cfg.log 2>&1 || exit 3
  cp $O/stats_$cfg/*/*_kernel_stats.csv $O/kernel_stats_$cfg.csv
  echo "$cfg done"
done
ls $O
