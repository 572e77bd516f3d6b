#!/bin/bash
# Round 4, second session: measurement pass on the GPU box -- parity suite, smoke, then tools/r04_profile.sh (PMC traffic of the step and
# of the quantizer, the bench line, rocprofv3 kernel stats of the same command).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > gpurun_out/r04b_pytest_full.txt
cat gpurun_out/r04b_pytest_full.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/r04b_smoke.txt
cat gpurun_out/r04b_smoke.txt
bash tools/r04_profile.sh
cp gpurun_out/r04b_pytest_full.txt gpurun_out/r04b_smoke.txt gpurun_out/r04/
# BASELINE config 4 (its K = 1024 quantizer on the stream-tracker kernel's four-wave form): kernel stats of bench.py --workload c4
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04/prof_c4 -- python bench.py --workload c4 --no-cpu-baseline --no-other-workloads --steps 5 --min-seconds 0.2 > $R/gpurun_out/r04/prof_c4.log 2>&1)
DB=$(find $R/gpurun_out/r04/prof_c4 -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB 20 > $R/gpurun_out/r04/c4_kernel_stats.txt
rm -rf $R/gpurun_out/r04/prof_c4/*/*.db* 2>/dev/null
head -14 $R/gpurun_out/r04/c4_kernel_stats.txt | cut -c1-150
