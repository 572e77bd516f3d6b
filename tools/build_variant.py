#!/usr/bin/env python3
"""Build a variant of libvqvae_hip.so with extra -D flags into vqvae_amd/build/variants/ (A/B and knock-out runs).

    python tools/build_variant.py NAME -DFOO=1 -DBAR         ->  vqvae_amd/build/variants/libvqvae_NAME.so
Tools pick it up through VQVAE_BENCH_LIB=<path>."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import build as B

name, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(B.HERE, "build", "variants")
os.makedirs(os.path.join(out, name), exist_ok=True)
procs, objs = [], []
for src in B.sources():
    obj = os.path.join(out, name, os.path.basename(src) + ".o")
    objs.append(obj)
    procs.append(subprocess.Popen([B.hipcc(), *B.flags_for(src), *defs, "-c", src, "-o", obj]))
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(out, f"libvqvae_{name}.so")
B.link([*objs, B.fingerprint_object(os.path.join(out, name))], lib)      # (_lib.load() wants vqvae_source_fingerprint in every library)
print(lib)
