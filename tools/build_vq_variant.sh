#!/bin/bash
# Variant of libvqvae_hip.so that differs from the current tree only in vq_track.hip's compile-time switches:
#   tools/build_vq_variant.sh NAME [-DFOO ...]   ->  vqvae_amd/build/variants/libvqvae_NAME.so
# (every other object is taken from vqvae_amd/build/, i.e. run `python -m vqvae_amd.build` first)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=vqvae_amd/build/variants; mkdir -p $out/$name
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function"
hipcc $FLAGS "$@" -c vqvae_amd/csrc/vq_track.hip -o $out/$name/vq_track.hip.o
objs=$(ls vqvae_amd/build/*.hip.o | grep -v vq_track.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libvqvae_$name.so $objs $out/$name/vq_track.hip.o
echo $out/libvqvae_$name.so
