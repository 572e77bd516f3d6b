#!/bin/bash
# round 5: torch-free A/B of quantizer builds + the GPU test suite in one gpurun call
#   tools/r05_vq_call.sh TAG ITERS "lib names" [pytest args...]      (VQ_AB_MULTS / VQ_AB_FORMS pass through)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=$1; iters=$2; names=$3; shift 3
libs=""; for n in $names; do libs="$libs vqvae_amd/build/variants/libvqvae_$n.so"; done
TRACE_OUT=gpurun_out/vqtrace_$tag timeout 300 tools/ubench/vq_ab ${VQ_AB_DATA:-tools/data/vq_c3.bin} $iters $libs > gpurun_out/r05_vq_$tag.txt 2>&1
tail -n 60 gpurun_out/r05_vq_$tag.txt
if [ $# -gt 0 ]; then timeout 900 python -m pytest "$@" > gpurun_out/r05_pytest_$tag.txt 2>&1; tail -n 15 gpurun_out/r05_pytest_$tag.txt; fi
