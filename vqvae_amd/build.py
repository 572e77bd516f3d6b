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


def have_sources() -> bool:
    """is this a source tree (csrc/ with the kernels) or an installed one that only carries the built library?"""
    return bool(sources())


def source_fingerprint() -> str:
    """sha256 (16 hex digits) over the names and bytes of vqvae_amd/csrc/*.hip and *.h -- what the library is built from, and
    nothing else that may lie there (editor backups, .orig files of a patch).  The build links it in
    (vqvae_source_fingerprint()), `_lib.load()` compares it with the sources it finds, bench.py stamps profiles with it."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def fingerprint_object(objdir: str) -> str:
    """compile the one-line translation unit that makes the sources' fingerprint a function of the library
    (vqvae_source_fingerprint(); every library `_lib.load()` opens must export it -- the A/B variants of tools/ too) -> object path"""
    os.makedirs(objdir, exist_ok=True)
    fp_src = os.path.join(objdir, "source_fingerprint.cpp")
    with open(fp_src, "w") as f:
        f.write('extern "C" __attribute__((visibility("default"))) const char *vqvae_source_fingerprint(void) '
                f'{{ static const char s[] = "{FP_MARKER}{source_fingerprint()}"; return &s[{len(FP_MARKER)}]; }}\n')
    fp_obj = os.path.join(objdir, "source_fingerprint.o")
    subprocess.check_call([hipcc(), "-O1", "-fPIC", "-x", "c++", "-c", fp_src, "-o", fp_obj])
    return fp_obj


def link(objs, out: str) -> str:
    """link to a temporary name and rename into place: a process that opens `out` meanwhile (another rank of the same node) sees
    the old library or the new one, never a half-written file"""
    tmp = f"{out}.tmp.{os.getpid()}"
    try:
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs])
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return out


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
    # the fingerprint of the sources, as a function of the library (a generated translation unit: no object above depends on it)
    link([*objs, fingerprint_object(objdir)], LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
