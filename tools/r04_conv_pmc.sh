#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_conv_pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 180 rocprofv3 -i tools/pmc_lds.txt --kernel-trace --output-format csv -d $O/pmc -- python bench.py --no-cpu-baseline --no-other-workloads --steps 2 --warmup 1 --min-seconds 0.01 > $O/pmc.log 2>&1)
python $R/tools/pmc_summary.py $O/pmc > $O/summary.txt
grep -A17 "h2_kernel" $O/summary.txt | grep -v "^--" | head -90
