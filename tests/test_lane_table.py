"""The bookkeeping that decides which calls may work on a context at the same time
(lamp_amd/csrc/lane_table.h: submits take a lane each, registration takes all of them) is plain
host code: compiled with ThreadSanitizer and driven from twelve threads, no two grants may share a
lane, a grant of all lanes may overlap nothing, and no data race may be reported."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_lane_grants_never_overlap(tmp_path):
    exe = str(tmp_path / "lane_table")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread",
                    "-I", os.path.join(ROOT, "lamp_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "lane_table.cc"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr, r.stderr
    assert r.stdout.startswith("errors 0 "), r.stdout
