#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > gpurun_out/r04/pytest_full.txt
cat gpurun_out/r04/pytest_full.txt
bash tools/r04_profile.sh
