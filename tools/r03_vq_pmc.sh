#!/bin/bash
# Round 3: the standalone quantizer kernel under rocprofv3 -- kernel-trace stats (true kernel duration) and two SQ counter
# passes (stall attribution, instruction counts), on the model-like rows of tools/vq_phase.py at VQ_ROWS rows.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r3_vq_pmc}
rm -rf $O; mkdir -p $O
for n in ${VQ_ROWS_LIST:-262144 2097152}; do
  VQ_ROWS=$n timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -- python $R/tools/vq_phase.py > $O/stats_$n.log 2>&1
  VQ_ROWS=$n timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
     --output-format csv -d $O/a_$n -- python $R/tools/vq_phase.py > $O/a_$n.log 2>&1
  VQ_ROWS=$n timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
     --output-format csv -d $O/b_$n -- python $R/tools/vq_phase.py > $O/b_$n.log 2>&1
done
python - <<PY > $O/summary.txt
import csv, glob, collections
for n in "${VQ_ROWS_LIST:-262144 2097152}".split():
    print("== rows", n)
    for f in glob.glob("$O/stats_%s/**/*kernel_stats.csv" % n, recursive=True):
        for r in csv.DictReader(open(f)):
            if "vq_" in r["Name"]: print("   stats:", r["Name"][:70], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min_ns", r["MinNs"])
    for d in ("a", "b"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (d, n), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in sorted(agg.items()):
            if "vq_track" not in k and "vq_sweep" not in k: continue
            print("  ", k)
            print("      ", {m: round(sum(v) / len(v)) for m, v in sorted(c.items())})
PY
cat $O/summary.txt
