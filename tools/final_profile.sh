#!/bin/bash
# Round-end measurement pass on the GPU box: parity suite, bench line, rocprofv3 kernel stats of the same command,
# and FETCH_SIZE / WRITE_SIZE (every step under its own `timeout`: a hung counter pass must not eat the box) (separate --pmc passes, kernel-trace only) for every kernel of the step.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > $O/pytest.txt
(cd $R && timeout 300 python bench.py 2>/dev/null | tail -1) > $O/bench.json
(cd $R && timeout 180 rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --no-cpu-baseline --steps 20 > $O/prof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 24 > $O/kernel_stats.txt
for c in fetch write; do
  (cd $R && timeout 120 rocprofv3 -i tools/pmc_$c.txt --kernel-trace --output-format csv -d $O/pmc_$c -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/pmc_$c.log 2>&1)
  python $R/tools/pmc_summary.py $O/pmc_$c > $O/pmc_$c.txt
done
rm -rf $O/prof/*/*.db.bak 2>/dev/null
cat $O/pytest.txt; cut -c1-400 $O/bench.json; head -16 $O/kernel_stats.txt | cut -c1-150; grep -A1 "n=" $O/pmc_fetch.txt | head -40
