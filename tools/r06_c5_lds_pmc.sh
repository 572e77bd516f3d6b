#!/bin/bash
# LDS / matrix-pipe counters of config 5's kernels ("LDS-read-bound" needs a counter behind it): one --pmc pass each
# (kernel-trace only), bench.py --workload c5 with two steps.   -> gpurun_out/r06c5/*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06c5; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="--workload c5 --no-cpu-baseline --no-other-workloads --no-power --steps 2 --warmup 1 --min-seconds 0.01"
for set in lds util; do
  (cd $R && timeout 300 rocprofv3 -i tools/pmc_$set.txt --kernel-trace --output-format csv -d $O/pmc_$set -- python bench.py $B > $O/pmc_$set.log 2>&1)
done
python - <<PY
import csv, collections, glob
O = "$O"
def load(d):
    cnt = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return cnt, dur
cl, dl = load(O + "/pmc_lds")
print("kernel                                                            n   us    | per launch: SQ_BUSY_CYCLES  MFMA_BUSY  LDS_IDX_ACTIVE  LDS_BANK_CONFLICT  INSTS_LDS | LDS active / busy   conflict / active   MFMA busy / (busy x 4 SIMD ... see note)")
for k, v in sorted(cl.items(), key=lambda kv: -sum(dl.get(kv[0], [0]))):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    us = sum(dl[k]) / len(dl[k]) / 1e3 if dl.get(k) else 0
    if us < 30: continue
    busy = max(a.get("SQ_BUSY_CYCLES", 1), 1)
    print(f"{k[:64]:64s} {len(dl[k]):3d} {us:8.1f} | {a.get('SQ_BUSY_CYCLES',0):14.3e} {a.get('SQ_VALU_MFMA_BUSY_CYCLES',0):10.3e} {a.get('SQ_LDS_IDX_ACTIVE',0):14.3e} {a.get('SQ_LDS_BANK_CONFLICT',0):14.3e} {a.get('SQ_INSTS_LDS',0):10.3e} | "
          f"{a.get('SQ_LDS_IDX_ACTIVE',0)/busy:8.3f} {a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_LDS_IDX_ACTIVE',1),1):10.3f} {a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/busy:10.3f}")
PY
python $R/tools/pmc_util_summary.py $O/pmc_util 256 | cut -c1-170
rm -rf $O/pmc_*/*/*.db 2>/dev/null
