#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_pixelcnn.py tests/test_capi.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python tools/pixelcnn_bench.py 2>&1 | grep -v amdgpu | tail -6
