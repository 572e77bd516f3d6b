#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_training_gpu.py -m gpu -q -x -s -k "full_model" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-900
