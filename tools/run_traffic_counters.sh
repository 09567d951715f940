#!/bin/bash
# HBM-side traffic of the bench workload's sampler kernel per half-step (FETCH_SIZE and WRITE_SIZE in passes of their own):
#     bash tools/run_traffic_counters.sh <tag>   -> gpurun_out/<tag>/pmc_traffic.json
set -o pipefail
tag=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
export MBB_BENCH_FULL_LINE=1
cd $R
SHORT="--steps 300 --warmup 50 --no-extras"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $SHORT > $O/fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $SHORT > $O/write.log 2>&1 || exit 4
python3 tools/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/fetch.log,$O/write.log > /dev/null
python3 - <<PY
import json
d=json.load(open("$O/pmc_traffic.json"))
for k,v in d["kernels"].items():
    if "traffic_bytes_per_half_step" in v:
        hs=v["half_steps_in_all_launches"]
        print(k, "fetch KB/half-step %.1f  write KB/half-step %.1f  total bytes %.0f" % (v["FETCH_SIZE"]["sum_KB"]/hs["fetch_pass"], v["WRITE_SIZE"]["sum_KB"]/hs["write_pass"], v["traffic_bytes_per_half_step"]))
PY
