#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_conv_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -30 $O/gpu_tests.log | cut -c1-300
