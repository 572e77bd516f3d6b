"""Builds libvqvae_hip.so in-tree with hipcc for gfx950 (no torch extension machinery,
no hipify: the sources are HIP, the ABI is plain C -- include/vqvae_hip.h)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvqvae_hip.so")

FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


# Per-source additions.  vq_track.hip: hipcc's SLP vectoriser packs the epilogue's fp32 adds / multiplies into v_pk_* pairs plus the
# v_mov shuffles that feed them -- more issue slots, not fewer, on gfx950 (profiles/r05_vq_notes.txt).
EXTRA_FLAGS = {"vq_track.hip": ["-fno-slp-vectorize"]}


def flags_for(src: str):
    return [*FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), [])]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; libvqvae_hip.so cannot be built")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


FP_MARKER = "VQVAE_SRC_FP="        # in front of the fingerprint inside the library file: readable without loading the library


def library_fingerprint(path: str = LIB):
    """the fingerprint a built library carries (None: built before round 5's guard, or not ours)"""
    with open(path, "rb") as f:
        blob = f.read()
    i = blob.find(FP_MARKER.encode())
    return blob[i + len(FP_MARKER):i + len(FP_MARKER) + 16].decode(errors="replace") if i >= 0 else None


def source_fingerprint() -> str:
    """sha256 (16 hex digits) over the names and bytes of vqvae_amd/csrc/*: what the library is built from.  The build links it in
    (vqvae_source_fingerprint()), `_lib.load()` compares it with the sources it finds, bench.py stamps profiles with it."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def stale() -> bool:
    if not os.path.exists(LIB) or library_fingerprint() != source_fingerprint():
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(HERE, "..", "include", "vqvae_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc(), *flags_for(src), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    # the fingerprint of the sources, as a function of the library (a generated one-line translation unit: no object above depends on it)
    fp_src = os.path.join(objdir, "source_fingerprint.cpp")
    with open(fp_src, "w") as f:
        f.write('extern "C" __attribute__((visibility("default"))) const char *vqvae_source_fingerprint(void) '
                f'{{ static const char s[] = "{FP_MARKER}{source_fingerprint()}"; return &s[{len(FP_MARKER)}]; }}\n')
    fp_obj = os.path.join(objdir, "source_fingerprint.o")
    subprocess.check_call([hipcc(), "-O1", "-fPIC", "-x", "c++", "-c", fp_src, "-o", fp_obj])
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, fp_obj]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
