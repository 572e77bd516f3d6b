#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/train_bench.py 4096 hip 20 2>&1 | grep backend
bash tools/r04_run9.sh 2>&1 | grep -E "conv_tile8|igemm_bf3|conv_in_rows|backend" | cut -c1-150
