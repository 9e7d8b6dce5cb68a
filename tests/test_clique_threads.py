"""The TEASER++ clique heuristic evaluates windows of restarts on several threads; its result is
defined by the serial order.  Compiled for the host with ThreadSanitizer (teaser.h has no device
code): any thread count must return the serial clique, without a data race."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_clique_is_thread_count_independent(tmp_path):
    exe = str(tmp_path / "clique_threads")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread",
                    "-I", os.path.join(ROOT, "lamp_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "clique_threads.cc"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr, r.stderr
    assert r.stdout.count("same 1 1") == 6, r.stdout
