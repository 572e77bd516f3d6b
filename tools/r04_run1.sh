#!/bin/bash
# Round-4 first GPU pass: the whole -m gpu suite on the per-output-channel weight scales (+ the new heterogeneous-scale parity
# tests), the bench line, and the rocprofv3 kernel stats of the config-3 step.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -60 $O/gpu_tests.log | cut -c1-400
(timeout 400 python bench.py --no-other-workloads 2>/dev/null | tail -1) > $O/bench_c3.json
cut -c1-1200 $O/bench_c3.json
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --no-cpu-baseline --no-other-workloads --steps 20 --min-seconds 0.2 > $O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 24 > $O/kernel_stats.txt
rm -rf $O/prof
head -14 $O/kernel_stats.txt | cut -c1-160
