#!/bin/bash
# Round-4 measurement pass on the GPU box (every step under its own `timeout`): bench line, rocprofv3 kernel stats of the
# same command, FETCH_SIZE / WRITE_SIZE (separate --pmc passes, kernel-trace only) for the bench step and for the VQ
# kernel on a stream beyond the Infinity Cache, and the traffic JSON bench.py reads.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04; rm -rf $O; mkdir -p $O          # gpurun MERGES into existing directories: stale counter files would be averaged in
cd /tmp; export TMPDIR=/tmp
for c in fetch write; do
  (cd $R && timeout 180 rocprofv3 -i tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/pmc_$c -- python bench.py --no-cpu-baseline --no-other-workloads --steps 2 --warmup 1 --min-seconds 0.01 > $O/pmc_$c.log 2>&1)
  (cd $R && timeout 180 rocprofv3 -i tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/vq_pmc_$c -- python tools/vq_traffic.py > $O/vq_pmc_$c.log 2>&1)
done
(cd $R && python tools/pmc_traffic.py c3 4096 $O/pmc_fetch $O/pmc_write $O/vq_pmc_fetch $O/vq_pmc_write 4194304 $O/hbm_traffic_c3.json) > $O/traffic.txt 2>&1
# the bench line is printed AFTER the traffic file of the same build exists (bench.py reads profiles/hbm_traffic_c3.json)
cp $O/hbm_traffic_c3.json $R/profiles/hbm_traffic_c3.json
(cd $R && timeout 400 python bench.py 2>/dev/null | tail -1) > $O/bench_c3.json
(cd $R && timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --no-cpu-baseline --no-other-workloads --steps 20 --min-seconds 0.2 > $O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 24 > $O/kernel_stats.txt
rm -rf $O/prof/*/*.db.bak $O/pmc_*/*/*.db $O/vq_pmc_*/*/*.db 2>/dev/null
cut -c1-600 $O/bench_c3.json; head -18 $O/kernel_stats.txt | cut -c1-150; cat $O/traffic.txt
