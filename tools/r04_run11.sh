#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_training_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/train_bench.py 4096 hip 20 2>&1 | grep backend
timeout 300 python - <<'PY'
import torch, json, bench
print(json.dumps(bench.training_step(torch, torch.device("cuda:0"))))
PY
