#!/bin/bash
# Variant of libvqvae_hip.so that differs from the current tree only in vq_track.hip's compile-time switches (-DVQ_TRACE, -DVQ_TRACE2,
# -DVQ_KO_*):   tools/build_vq_variant.sh NAME [-DFOO ...]   ->  vqvae_amd/build/variants/libvqvae_NAME.so
exec "$(dirname "$0")/build_src_variant.sh" vq_track.hip "$@"
