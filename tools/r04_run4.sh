#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_vq_gpu.py -m gpu -q --maxfail=5 -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
timeout 300 python tools/vq_ab4.py 2>&1 | grep -v amdgpu | cut -c1-700
done
