#!/bin/bash
# round 5: SQ counters of the stand-alone quantizer under the torch-free harness (seconds per pass):
#   tools/r05_vq_pmc.sh TAG libname [mults=4] [forms=0]      -> gpurun_out/r05_vq_pmc_TAG.txt
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; lib=$R/vqvae_amd/build/variants/libvqvae_$2.so; mults=${3:-4}; forms=${4:-0}
O=$R/gpurun_out/r05_vq_pmc_$tag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VQ_AB_MULTS=$mults VQ_AB_FORMS=$forms
run() { timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$pass -- $R/tools/ubench/vq_ab $R/tools/data/vq_c3.bin 10 $lib > $O/$pass.log 2>&1; }
pass=a; run SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES
pass=b; run SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass=c; run SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES
pass=d; run FETCH_SIZE
pass=e; run WRITE_SIZE
pass=s; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- $R/tools/ubench/vq_ab $R/tools/data/vq_c3.bin 30 $lib > $O/s.log 2>&1
python3 - <<PY > $R/gpurun_out/r05_vq_pmc_$tag.txt
import csv, glob, collections
O="$O"
print("lib $2 mults $mults forms $forms")
for f in glob.glob(O + "/s/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vq_" in r["Name"]: print("   stats:", r["Name"][:70], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min_ns", r["MinNs"])
for d in "abcde":
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        if "vq_track" not in k: continue
        print("  ", k)
        print("      ", {m: round(sum(v) / len(v)) for m, v in sorted(c.items())})
PY
cat $R/gpurun_out/r05_vq_pmc_$tag.txt
