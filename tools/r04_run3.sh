#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_vq_gpu.py tests/test_parity_hetero_gpu.py tests/test_fuzz_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
grep -v "worst error" $O/gpu_tests.log | tail -15 | cut -c1-300
timeout 300 python tools/vq_ab4.py > $O/vq_ab4.txt 2>&1
cat $O/vq_ab4.txt | cut -c1-600
timeout 300 python tools/vq_repro_hetero.py 2>&1 | grep -v amdgpu | tail -12
