#!/bin/bash
# rocprofv3 kernel stats of the training step (main.py:74-78 without the optimizer, B = 4096): tools/train_bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06t; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof -- python tools/train_bench.py 4096 hip 16 > $O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 70 > $O/train_kernel_stats.txt
rm -rf $O/prof
cat $O/prof.log | tail -2; cut -c1-170 $O/train_kernel_stats.txt
