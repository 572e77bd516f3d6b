#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_vq_gpu.py -m gpu -q -x 2>&1 | tail -5
