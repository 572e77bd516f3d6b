#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r04
timeout 300 python tools/vq_ab4.py 2>&1 | grep -v amdgpu | cut -c1-420
timeout 400 python bench.py --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['roofline'])[:600])"
timeout 600 python -m pytest tests/test_vq_gpu.py tests/test_capi.py -m gpu -q -x 2>&1 | tail -3
