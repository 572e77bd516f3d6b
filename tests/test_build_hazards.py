"""Assembly-level check of every HIP source (no GPU): (1) a 16-byte buffer store whose soffset is an SGPR must not be followed
within two instructions by a vector write of its data registers.  hipcc (ROCm 7.2) does not insert wait states for that
form, and MI355X does corrupt the stored data (round 3: z_q of vq_track_kernel_d64, last dword of lanes 12..15 of a row).
tools/hazard_scan.py has the pattern; the sources avoid it by construction (no soffset register on wide stores that are
followed by arithmetic), this test keeps it that way after every edit / compiler change.  (2) round 4: the fdot2 miscompile
(four v_dot2c_f32_f16 reading ONE register where four different components were meant; csrc/common.h, sqsum8_f16) must not
reappear in any source's assembly."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_unguarded_store_data_overwrite(tmp_path):
    import pytest
    sys.path.insert(0, ROOT)
    from tools import hazard_scan
    from vqvae_amd import build as hip_build
    try:
        hip_build.hipcc()
    except RuntimeError:
        pytest.skip("hipcc not found: the assembly scan needs the ROCm compiler (the runtime bit-exact z_q tests are the primary guard)")

    flags = [f for f in hip_build.FLAGS if f not in ("-fPIC", "-fvisibility=hidden")]

    def asm(src):
        out = str(tmp_path / (os.path.basename(src) + ".s"))
        subprocess.check_call([hip_build.hipcc(), *flags, "-S", "--cuda-device-only", "-o", out, src],
                              stderr=subprocess.DEVNULL)
        return out

    with ThreadPoolExecutor(8) as ex:
        files = list(ex.map(asm, hip_build.sources()))
    assert len(files) >= 10
    sites = sum(hazard_scan.scan(f) for f in files)
    assert sites == 0, f"{sites} unguarded store-data overwrite(s) / miscompiled dot2c chain(s): see the output above"
