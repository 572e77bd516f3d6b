#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04t; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof -- python tools/train_bench.py 4096 hip 13 > $O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 40 > $O/train_kernel_stats.txt
rm -rf $O/prof
cat $O/prof.log | tail -2; cut -c1-170 $O/train_kernel_stats.txt
