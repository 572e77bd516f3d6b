#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
timeout 400 python bench.py --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['calibration']['mfma_fp16_random_tflops'], d['roofline']['avg_kernel_us'])"
