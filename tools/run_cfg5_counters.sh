set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6_cfg5a; mkdir -p $O
python tools/ab_two_libs.py tools/libmbb_hip_head.so mbb_emcee_amd/libmbb_hip.so 7 > $O/ab.txt 2>&1 || exit 1
tail -4 $O/ab.txt | head -3
VALU="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
timeout -k 10 300 rocprofv3 --pmc $VALU --kernel-trace --output-format csv -d $O/pmc_valu -- python3 tools/bench_cfg5.py --quick > $O/valu.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_lds -- python3 tools/bench_cfg5.py --quick > $O/lds.log 2>&1 || exit 3
python3 tools/summarize_valu.py $O/pmc_valu $O/pmc_valu_cfg5.json "rocprofv3 --pmc VALU --kernel-trace -- python3 tools/bench_cfg5.py --quick" > /dev/null
python3 - <<'PY'
import csv,glob,json,numpy as np
O="gpurun_out/r6_cfg5a"
p=glob.glob(O+"/pmc_lds/*/*_counter_collection.csv")[0]
per={}
for r in csv.DictReader(open(p)):
    if "k_lnlike<" not in r["Kernel_Name"]: continue
    g=int(r.get("Grid_Size",0) or 0)
    per.setdefault((r["Kernel_Name"],g),{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
out={}
for (k,g),c in per.items():
    out["%s grid %d"%(k[:60],g)]={cn:float(np.median(v)) for cn,v in c.items()}|{"n":len(next(iter(c.values())))}
json.dump(out,open(O+"/pmc_lds_cfg5.json","w"),indent=1)
for k,v in out.items(): print(k,v)
d=json.load(open(O+"/pmc_valu_cfg5.json"))
for k,v in d["kernels"].items(): print(k[:50], {a:b for a,b in v.items() if not isinstance(b,dict)}, v.get("counters_per_launch"))
PY
