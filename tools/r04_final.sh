#!/bin/bash
# Round-4 measurement pass on the GPU box: parity suite, smoke, the profile pass of tools/r04_profile.sh (PMC traffic, bench line,
# rocprofv3 kernel stats of the same command), the quantizer's SQ counters, the training step's kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4) > gpurun_out/r04_pytest_full.txt
cat gpurun_out/r04_pytest_full.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/r04_smoke.txt
cat gpurun_out/r04_smoke.txt
bash tools/r04_profile.sh
cp gpurun_out/r04_pytest_full.txt gpurun_out/r04_smoke.txt gpurun_out/r04/
VQ_ROWS_LIST="262144 2097152" bash tools/r03_vq_pmc.sh r04_vq_pmc > /dev/null 2>&1; cat gpurun_out/r04_vq_pmc/summary.txt | cut -c1-400
bash tools/r04_train_profile.sh > /dev/null 2>&1; head -30 gpurun_out/r04t/train_kernel_stats.txt | cut -c1-150
