#!/bin/bash
# Round-6 measurement pass on the GPU box (every step under its own `timeout`): parity suite, smoke, FETCH_SIZE / WRITE_SIZE passes
# (separate --pmc runs, kernel-trace only) for configs 3 / 4 / 5 and for the quantizer on a stream beyond the Infinity Cache, the traffic
# JSONs bench.py reads (keyed by template instance and row count), the bench lines, rocprofv3 kernel stats of the same commands.
#   tools/r06_final.sh [skip_tests]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O
cd $R
if [ -z "$1" ]; then
  (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) >> $O/pytest_gpu.txt; tail -2 $O/pytest_gpu.txt
fi
cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --no-other-workloads --no-power --steps 2 --warmup 1 --min-seconds 0.01"
for c in fetch write; do
  (cd $R && timeout 200 rocprofv3 -i tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/pmc_c3_$c -- python bench.py $B > $O/pmc_c3_$c.log 2>&1)
  (cd $R && timeout 200 rocprofv3 -i tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/vq_pmc_$c -- python tools/vq_traffic.py > $O/vq_pmc_$c.log 2>&1)
  for w in c4 c5; do
    (cd $R && timeout 240 rocprofv3 -i tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/pmc_${w}_$c -- python bench.py --workload $w $B > $O/pmc_${w}_$c.log 2>&1)
  done
done
(cd $R && python tools/pmc_traffic.py c3 4096 $O/pmc_c3_fetch $O/pmc_c3_write $O/vq_pmc_fetch $O/vq_pmc_write 4194304 $O/hbm_traffic_c3.json) > $O/traffic.txt 2>&1
(cd $R && python tools/pmc_traffic.py c4 512 $O/pmc_c4_fetch $O/pmc_c4_write $O/pmc_c4_fetch $O/pmc_c4_write 1605632 $O/hbm_traffic_c4.json) >> $O/traffic.txt 2>&1
(cd $R && python tools/pmc_traffic.py c5 1024 $O/pmc_c5_fetch $O/pmc_c5_write $O/pmc_c5_fetch $O/pmc_c5_write 4194304 $O/hbm_traffic_c5.json) >> $O/traffic.txt 2>&1
for w in c3 c4 c5; do [ -s $O/hbm_traffic_$w.json ] && cp $O/hbm_traffic_$w.json $R/profiles/hbm_traffic_$w.json; done
# the bench lines are printed AFTER the traffic files of the same build exist
(cd $R && timeout 600 python bench.py 2>$O/bench_c3.err | tail -1) > $O/bench_c3.json
for w in c4 c5; do (cd $R && timeout 300 python bench.py --workload $w --no-cpu-baseline --no-other-workloads --no-power 2>/dev/null | tail -1) > $O/bench_$w.json; done
for w in c3 c4 c5; do
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -- python bench.py --workload $w --no-cpu-baseline --no-other-workloads --no-power --steps 10 --min-seconds 0.2 > $O/prof_$w.log 2>&1)
  DB=$(find $O/prof_$w -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 24 > $O/${w}_kernel_stats.txt
done
rm -rf $O/prof_*/*/*.db* $O/pmc_*/*/*.db $O/vq_pmc_*/*/*.db 2>/dev/null
cut -c1-700 $O/bench_c3.json; echo; head -14 $O/c3_kernel_stats.txt | cut -c1-150; cat $O/traffic.txt | cut -c1-600
