#!/bin/bash
# Round 4 additions to the profile set (tools/run_profiles.sh collects the bench workload's):
#   bash tools/run_profiles_r04.sh <tag>
# the cfg1 / cfg4 counter and kernel-trace passes (tools/run_profiles_r03.sh), and the kernel trace of the resident
# sampler forms on ensembles of 512 to 4096 walkers (tools/sweep_walkers.py: k_flowa / k_flowr against the launch train).
set -o pipefail
tag=${1:-prof_r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
cd $R
bash tools/run_profiles_r03.sh $tag || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_big -- python3 tools/sweep_walkers.py 512 1000 2000 4096 > $O/sweep_big.txt 2> $O/stats_big.log || exit 2
cp $O/stats_big/*/*_kernel_stats.csv $O/kernel_stats_large_ensembles.csv
echo "large ensembles done"
ls $O
