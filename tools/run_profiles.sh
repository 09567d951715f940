#!/bin/bash
# Collects the round's profile set on the GPU box: bash tools/run_profiles.sh <tag>
# (bench line, rocprofv3 kernel stats of the same command, HBM traffic counters in two
# passes, VALU counters of the bench workload's launches and of the 250 000-walker launch).
# Output under gpurun_out/<tag>/; summaries are copied to profiles/ by hand.
# Counter passes carry --kernel-trace only (no other trace domain next to --pmc).
set -o pipefail
tag=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
export MBB_BENCH_FULL_LINE=1     # (the summarizers read legs that the driver's short line leaves to the side file)
cd $R
VALU="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SHORT="--steps 300 --warmup 50 --no-extras"
# (how many half-steps the sampler kernel covers in such a run: bench.py says on its line, the summaries read the logs)
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
cp $R/gpurun_out/bench_full.json $O/bench_full.json
# the driver's own command, as the driver sees it: the short line on stdout, everything else in the side file
( unset MBB_BENCH_FULL_LINE; timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err ) || exit 1
cp $R/gpurun_out/bench_full.json $O/bench_driver_command_full.json
echo "bench done"
# kernel trace + stats of the timed region alone (default K and W): the dominant kernel's three launches
# (rehearsal, warm-up, timed), tools/summarize_stats.py holds the timed one against the line's HIP events
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-extras > $O/stats_line.json 2> $O/stats.log || exit 2
python3 tools/summarize_stats.py $O/stats $O/stats_line.json $O/kernel_time.json > /dev/null || exit 2
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $SHORT > $O/fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $SHORT > $O/write.log 2>&1 || exit 4
echo "traffic done"
timeout -k 10 300 rocprofv3 --pmc $VALU --kernel-trace --output-format csv -d $O/pmc_valu_cfg2 -- python3 bench.py $SHORT > $O/valu_cfg2.log 2>&1 || exit 5
timeout -k 10 400 rocprofv3 --pmc $VALU --kernel-trace --output-format csv -d $O/pmc_valu -- python3 tools/bench_cfg5.py --quick > $O/valu.log 2>&1 || exit 6
# the same chain as a train of plain launches: the algorithmic work of a half-step (nothing computed twice)
MBB_BENCH_PLAIN_TRAIN=1 timeout -k 10 300 rocprofv3 --pmc $VALU --kernel-trace --output-format csv -d $O/pmc_valu_plain -- python3 bench.py $SHORT > $O/valu_plain.log 2>&1 || exit 7
echo "valu done"
python3 tools/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/fetch.log,$O/write.log > /dev/null
python3 tools/summarize_valu.py $O/pmc_valu_cfg2 $O/pmc_valu_cfg2.json "rocprofv3 --pmc $VALU --kernel-trace -- python3 bench.py $SHORT" "" $O/valu_cfg2.log > /dev/null
python3 tools/summarize_valu.py $O/pmc_valu_plain $O/pmc_valu_plain.json "MBB_BENCH_PLAIN_TRAIN=1 rocprofv3 --pmc $VALU --kernel-trace -- python3 bench.py $SHORT" "the sampler as one plain launch per half-step" > /dev/null
python3 tools/summarize_valu.py $O/pmc_valu $O/pmc_valu_cfg5.json "rocprofv3 --pmc $VALU --kernel-trace -- python3 tools/bench_cfg5.py --quick" > /dev/null
cp $O/stats/*/*_kernel_stats.csv $O/kernel_stats.csv
ls $O
