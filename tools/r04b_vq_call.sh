#!/bin/bash
# torch-free A/B of quantizer builds: tools/r04b_vq_call.sh TAG ITERS lib-names...   (VQ_AB_MULTS / VQ_AB_FORMS / VQ_AB_DATA pass through)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=$1; iters=$2; shift 2
libs=""; for n in "$@"; do libs="$libs vqvae_amd/build/variants/libvqvae_$n.so"; done
TRACE_OUT=gpurun_out/vqtrace_$tag timeout 300 tools/ubench/vq_ab ${VQ_AB_DATA:-tools/data/vq_c3.bin} $iters $libs > gpurun_out/r04b_vq_$tag.txt 2>&1
grep -c . gpurun_out/r04b_vq_$tag.txt
