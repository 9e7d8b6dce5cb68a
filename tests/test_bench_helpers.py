"""Host-side helpers of the measurement harness (no GPU): PMC summaries are folded the way the bench
line reports them."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmc_kernel_folds_template_instantiations():
    b = _bench()
    summary = {"kernels": {
        "lgicp::k<false, 2>": {"FETCH_SIZE": {"launches": 2, "sum": 200.0, "mean_per_launch": 100.0},
                               "WRITE_SIZE": {"launches": 2, "sum": 20.0, "mean_per_launch": 10.0}},
        "lgicp::k<false, 8>": {"FETCH_SIZE": {"launches": 6, "sum": 1800.0, "mean_per_launch": 300.0},
                               "WRITE_SIZE": {"launches": 6, "sum": 60.0, "mean_per_launch": 10.0}},
        "lgicp::plain": {"FETCH_SIZE": {"launches": 1, "sum": 1.0, "mean_per_launch": 1.0}},
    }}
    k = b.pmc_kernel(summary, "lgicp::k")
    assert k["FETCH_SIZE"]["launches"] == 8 and k["FETCH_SIZE"]["mean_per_launch"] == 250.0
    # FETCH_SIZE counts 128-byte requests at 64 bytes (KiB units): doubled; WRITE_SIZE as is
    assert k["hbm_bytes_per_launch"] == 1024.0 * (2 * 250.0 + 10.0)
    assert b.pmc_kernel(summary, "lgicp::plain")["FETCH_SIZE"]["sum"] == 1.0
    assert b.pmc_kernel(summary, "lgicp::absent") == {} and b.pmc_kernel(None, "x") == {}


def test_pmc_summary_tool_classifies_lookup_passes(tmp_path):
    rows = []
    def add(dispatch, kernel, counter, value):
        rows.append({"Dispatch_Id": dispatch, "Kernel_Name": kernel, "Counter_Name": counter,
                     "Counter_Value": value})
    nn = "lgicp::nn_pruned_kernel(lgicp::CloudDev const*)"
    # a full sweep: 2000 VALU instructions per wave; a certified pass: 100
    for d, per_wave in ((1, 2000.0), (2, 100.0)):
        add(d, nn, "SQ_WAVES", 1000.0)
        add(d, nn, "SQ_INSTS_VALU", per_wave * 1000.0)
        add(d, nn, "GRBM_GUI_ACTIVE", 8.0 * 100000.0)
    p = tmp_path / "pass_counter_collection.csv"
    with open(p, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    out = tmp_path / "summary.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(out), "w", "c", str(p)])
    s = json.load(open(out))
    classes = s["nn_pruned_kernel_by_pass_class"]
    sweep = [v for k, v in classes.items() if k.startswith("sweep")][0]
    cert = [v for k, v in classes.items() if k.startswith("mostly certified")][0]
    # 4 cycles per wave instruction on one of 1024 SIMDs, GRBM_GUI_ACTIVE summed over 8 XCDs
    assert abs(sweep["valu_issue_utilisation"] - 4 * 2000.0 * 1000.0 / (1024 * 100000.0)) < 1e-12
    assert abs(cert["valu_issue_utilisation"] - 4 * 100.0 * 1000.0 / (1024 * 100000.0)) < 1e-12
    assert s["kernels"]["lgicp::nn_pruned_kernel"]["SQ_WAVES"]["launches"] == 2


def test_steps_in_flight_run_once_each_and_report_in_order():
    """bench.run_steps_in_flight: every step runs exactly once, on the lane its index belongs to;
    results are handed over in step order whatever order the lanes finish in; `depth` steps really
    are in flight together; an exception in a lane or in the consumer ends the run and is raised."""
    import threading
    import time
    b = _bench()
    ran, seen = [], []
    lock = threading.Lock()
    in_flight, peak = [0], [0]

    def submit_step(k, i):
        with lock:
            in_flight[0] += 1
            peak[0] = max(peak[0], in_flight[0])
        time.sleep(0.002 * ((i * 7) % 5))      # lanes finish out of order
        with lock:
            in_flight[0] -= 1
            ran.append((k, i))
        return i * i

    b.run_steps_in_flight(23, 4, 3, submit_step, lambda i, v: seen.append((i, v)))
    assert seen == [(i, i * i) for i in range(4, 27)]
    assert sorted(i for _, i in ran) == list(range(4, 27))
    assert all(k == (i - 4) % 3 for k, i in ran)
    assert peak[0] == 3

    def failing(k, i):
        if i == 9:
            raise RuntimeError("lane failed")
        return i

    try:
        b.run_steps_in_flight(20, 0, 4, failing, lambda i, v: None)
        raise AssertionError("no exception")
    except RuntimeError as e:
        assert "lane failed" in str(e)

    def consumer(i, v):
        if i == 5:
            raise ValueError("consumer failed")

    try:
        b.run_steps_in_flight(12, 0, 2, lambda k, i: i, consumer)
        raise AssertionError("no exception")
    except ValueError as e:
        assert "consumer failed" in str(e)
