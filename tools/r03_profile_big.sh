#!/bin/bash
# BASELINE configs 5 and 4 on one GPU: bench line + rocprofv3 kernel stats of the same command (tools/rocprof_summary.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03big; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for wl in c5 c4; do
(cd $R && timeout 300 python bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1) > $O/bench_$wl.json
(cd $R && timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_$wl -- python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 --min-seconds 0.05 > $O/prof_$wl.log 2>&1)
DB=$(find $O/prof_$wl -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 14 > $O/kernel_stats_$wl.txt
rm -rf $O/prof_$wl
cut -c1-260 $O/bench_$wl.json; head -14 $O/kernel_stats_$wl.txt | cut -c1-150
done
