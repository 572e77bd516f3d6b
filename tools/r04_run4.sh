#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_vq_gpu.py -m gpu -q --maxfail=5 -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
echo "--- new (pipelined sweep)"; timeout 300 python tools/vq_ab4.py 2>&1 | grep -v amdgpu | cut -c1-400
echo "--- old sweep"; VQVAE_HIP_LIB_OVERRIDE=$R/vqvae_amd/build/variants/libvqvae_sweepold.so timeout 300 python tools/vq_ab4.py 2>&1 | grep -v amdgpu | cut -c1-400
done
