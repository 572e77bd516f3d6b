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
