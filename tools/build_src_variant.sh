#!/bin/bash
# Variant of libvqvae_hip.so that differs from the current tree in ONE source's compile-time switches:
#   tools/build_src_variant.sh SRC.hip NAME [-DFOO ...]   ->  vqvae_amd/build/variants/libvqvae_NAME.so
# (every other object is taken from vqvae_amd/build/, i.e. run `python -m vqvae_amd.build` first)
set -e
cd "$(dirname "$0")/.."
src=$1; name=$2; shift 2
out=vqvae_amd/build/variants; mkdir -p $out/$name
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function"
[ "$src" = vq_track.hip ] && [ -z "$NO_EXTRA" ] && FLAGS="$FLAGS -fno-slp-vectorize"     # (vqvae_amd/build.py EXTRA_FLAGS; NO_EXTRA=1 builds without)
hipcc $FLAGS "$@" -c vqvae_amd/csrc/$src -o $out/$name/$src.o
objs=$(ls vqvae_amd/build/*.hip.o | grep -v "/$src.o")
# (_lib.load() wants vqvae_source_fingerprint in every library it opens: the generated object of the last full build)
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libvqvae_$name.so $objs $out/$name/$src.o vqvae_amd/build/source_fingerprint.o
echo $out/libvqvae_$name.so
