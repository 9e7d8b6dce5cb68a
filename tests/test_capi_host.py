"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
and refuses to run without a gfx950 device (no fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lamp_gicp.h")).read()
    return sorted(set(re.findall(r"\b(lgicp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lamp_amd import capi
    L = capi.lib()
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(capi.EXPORTS) == syms


def test_struct_layouts_match_header_sizes():
    # sizes the C compiler sees (checked against a tiny C program's sizeof output)
    import subprocess, tempfile, ctypes
    from lamp_amd import capi
    src = '#include "lamp_gicp.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu\\n",' \
          'sizeof(lgicp_config),sizeof(lgicp_candidate),sizeof(lgicp_result),sizeof(lgicp_stats));}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"),
                               "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [ctypes.sizeof(capi.Config), ctypes.sizeof(capi.Candidate),
                     ctypes.sizeof(capi.Result), ctypes.sizeof(capi.Stats)]


def test_default_config_mirrors_reference_yaml():
    from lamp_amd import capi
    base = capi.default_config(True)
    robot = capi.default_config(False)
    # loop_closure/config/laser_parameters.yaml: base :205-263, robot :58-109
    assert (base.corr_dist, base.iterations, base.max_tolerable_fitness) == (0.5, 200, 0.9)
    assert (robot.corr_dist, robot.iterations, robot.max_tolerable_fitness) == (0.1, 5, 0.4)
    assert base.tf_epsilon == 1e-10 and base.optimizer_iterations == 50
    assert base.init_method == capi.INIT_FEATURES and base.covariance_method == capi.COV_POINT2PLANE
    assert base.num_prev_scans == 2 and base.num_next_scans == 2
    assert base.accumulate_source == 0 and robot.accumulate_source == 1


def test_no_device_fails_loudly():
    import torch  # only to ask whether a GPU exists; the library itself never imports torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from lamp_amd import capi
    with pytest.raises(capi.LgicpError):
        capi.Context()


def test_host_mirror_cpu_logic():
    """C++ mirror of LoopComputation / IcpLoopComputation: parameter loading, reclosing filter,
    edge creation, queue and status publishing (no GPU needed)."""
    import subprocess
    exe = os.path.join(ROOT, "lamp_amd", "host", "test_loop_computation")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "lamp_amd", "host")])
    out = subprocess.check_output([exe, "--host-only"], timeout=60)
    assert b"host_only" in out
    # the C++ cut of a candidate list over devices (loop_closure/PairSharding.h) equals the Python
    # one the multi-process harness uses (lamp_amd/sharding.py), on the same pseudo-random list
    import json
    from lamp_amd.sharding import partition_pairs
    got = json.loads(out.decode().strip().splitlines()[-1])["partition"]
    state = [12345]

    def nxt():
        state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        return state[0] >> 33

    sizes = {k: 8000 + nxt() % 24000 for k in range(40)}
    pairs = []
    for _ in range(200):
        a = nxt() % 40
        b = nxt() % 40
        pairs.append((a, b))
    assert got == partition_pairs(pairs, sizes, 3)
    assert sorted(i for part in got for i in part) == list(range(200))
