#!/bin/bash
# SQ stall-attribution counters of the headline step (one pass, kernel-trace only): where the wave cycles of each conv kernel go.
# WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES (quad-cycles);
# VALU_MFMA_BUSY_CYCLES in cycles (32 per 32x32x16 MFMA).  MI355X_MICROARCH.md, "rocprofv3 PMC slots".
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_sq
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
   --output-format csv -d $O/a -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --min-seconds 0 > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
   --output-format csv -d $O/b -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --min-seconds 0 > $O/b.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        if "vqvae::" not in k or "pack" in k or "wscale" in k or "prepare" in k: continue
        print(k)
        print("   ", {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
