"""Host-side check of the stream tracker's logic (vqvae_amd/csrc/vq_track.h, the per-lane part of round 3's quantizer
kernel): tests/host/trk_harness.cpp includes the SAME header the HIP kernel compiles, emulates the two accumulator lanes
of every row over synthetic screen matrices and compares every verdict with a brute-force scan -- closed rows name the
one code at or above the threshold, open rows' exact tasks cover every code at or above it, the rescan is never taken
spuriously.  No GPU needed; the GPU tests then check the kernel's outputs bit for bit against the reference."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    exe = str(tmp_path_factory.mktemp("trk") / "trk_harness")
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-o", exe,
                           os.path.join(ROOT, "tests", "host", "trk_harness.cpp")])
    return exe


@pytest.mark.parametrize("rows,K,seed,mode", [
    (20000, 512, 0, 0),     # gaussian scores: ~4 % open
    (20000, 512, 1, 1),     # up to six planted near ties per row
    (20000, 512, 2, 2),     # planted bit-identical maxima
    (20000, 512, 3, 3),     # negative, tiny scores (keys of negative floats)
    (5000, 500, 7, 1),      # K % 32 != 0: padding codes in the last tile
    (5000, 33, 8, 1),       # one real code in the second tile
    (5000, 31, 3, 2),
    (2000, 1, 9, 0),        # single code
    (5000, 600, 9, 2),
    (5000, 1000, 4, 1),     # 32 tiles: the 6-bit cell field is full
])
def test_tracker_verdicts_against_brute_force(harness, rows, K, seed, mode):
    out = subprocess.run([harness, str(rows), str(K), str(seed), str(mode)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["violations"] == 0
    assert r["closed"] + r["open"] + r["hard"] == rows
    if mode in (0, 3) and K >= 500:
        assert r["closed"] > 0.9 * rows and r["hard"] < 0.01 * rows      # the common case stays on the cheap path
