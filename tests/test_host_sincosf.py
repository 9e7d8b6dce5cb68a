"""The BFGS back end builds its float transformations with the host libm's float sine and cosine
(lamp_amd/csrc/host_sincosf.h restates glibc's algorithm, so that a device run can be compared with
a host run of the same solver evaluation by evaluation).  Compiled for the host: the restatement
must return the bits of the host's own sinf / cosf."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_restated_sinf_cosf_return_the_host_libm_bits(tmp_path):
    exe = str(tmp_path / "host_sincosf")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                    "-I", os.path.join(ROOT, "lamp_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "host_sincosf.cc"), "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
