"""The one JSON line bench.py prints: every field the harness reads is there, with the meaning the
harness expects, on a workload small enough to run in seconds (32 pairs of 4,096-point keyframes)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--pairs", "32",
           "--points", "4096", "--keyframes", "16", "--cpu-sample", "4", *extra]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=280).stdout
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, lines          # exactly one line on stdout
    return json.loads(lines[0])


def test_default_line_has_what_the_harness_reads():
    d = _bench("--no-sub-records")
    assert d["metric"] == "loop-closure GICP pairs verified/sec" and d["unit"] == "pairs/s"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value = pairs of all steps / time of the timed region
    assert abs(d["value"] - d["config"]["pairs_per_step"] / (1e-3 * d["ms_per_step"])) <= 1e-6 * d["value"]
    assert d["value_single_submit"] > 0 and d["value_cold"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "fp32_valu") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["achieved"] > 0 and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["frac"] <= 1.0 and r["avg_launch_ms"] > 0 and "traffic" in r
    rr = d["roofline_reduce"]
    assert rr["bound"] == "hbm" and 0 < rr["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["unit"] == "pairs/s" and c["cores"] >= 1 and c["kind"] == "port" and c["sample"]
    # per-stage times come from the exclusive one-stream phase (the timed region runs without them)
    st = d["config"]["stage_ms_per_step"]
    assert st["nn"] > 0 and st["reduce"] > 0 and st["solve"] > 0
    assert d["config"]["pairs_ok_fraction"] > 0.8


def test_sub_records_describe_the_other_workloads():
    d = _bench("--no-cpu", "--no-cold")
    for key in ("accumulated", "solver_bfgs", "c3_list"):
        rec = d[key]
        assert rec["value"] > 0 and rec["unit"] == "pairs/s" and rec["submits_in_flight"] == 1
        assert rec["pairs_ok_fraction_rank0"] > 0.8
    assert "nn_one_stream" in d["accumulated"]
    assert d["solver_bfgs"]["value"] < d["value_single_submit"]      # the reference's optimiser is the slow one
