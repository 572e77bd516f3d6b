#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
echo "--- aux 0"; timeout 300 python tools/vq_ab4.py 2>&1 | grep -v amdgpu | cut -c1-420
for v in 2 16 18; do echo "--- aux $v"; VQVAE_HIP_LIB_OVERRIDE=$R/vqvae_amd/build/variants/libvqvae_aux$v.so timeout 300 python tools/vq_ab4.py 2>&1 | grep -v amdgpu | cut -c1-420; done
done
