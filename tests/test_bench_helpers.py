"""Host-side arithmetic of bench.py (no GPU): counters -> HBM figures, PMC traffic scaling."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_hbm_kernels_bytes_and_rates():
    c = {"evals": 1000, "children_scanned": 900000, "levels": 2500, "edges_created": 300000,
         "select_ms": 2.0, "finish_ms": 1.0}
    sel, fin = bench.hbm_kernels(c, 361, 362, steps=10)
    assert sel["kernel"] == "k_select" and sel["bound"] == "hbm" and sel["unit"] == "GB/s"
    sel_bytes = 1000 * (8 * 384 + 17 * 361 + 362) + 12 * 900000 + 8 * 2500
    assert abs(sel["bytes_per_sim"] - sel_bytes / 1000) < 1e-9
    assert abs(sel["achieved"] - sel_bytes / 2.0e-3 / 1e9) < 1e-6 and abs(sel["frac"] - sel["achieved"] / 8000.0) < 1e-12
    fin_bytes = 1000 * 5 * 362 + 18 * 300000 + 8 * 2500
    assert abs(fin["achieved"] - fin_bytes / 1.0e-3 / 1e9) < 1e-6 and abs(fin["ms_per_round"] - 0.1) < 1e-12


def test_measured_traffic_scales_with_boards():
    t, src = bench.measured_traffic(3993, 19, 256)
    t2, _ = bench.measured_traffic(7986, 19, 256)
    assert t > 1e9 and abs(t2 - 2 * t) <= 2 and "FETCH_SIZE" in src
    assert bench.measured_traffic(1024, 9, 256) == (None, None)      # other board size: no counters taken


def test_slot_count_for_whole_rounds_of_conv_tiles():
    from alphazero_amd.engine import slots_for_whole_tile_rounds
    assert slots_for_whole_tile_rounds(4096, 19, 19) == 3993          # 22 rounds of 256 tiles x 256 points
    assert slots_for_whole_tile_rounds(4096, 8, 8) == 4096            # exactly 4 rounds
    assert slots_for_whole_tile_rounds(4096, 15, 15) == 4074          # 14 rounds as 7 groups of 582 boards
    assert slots_for_whole_tile_rounds(100, 19, 19) == 100            # below one round: unchanged
    for n in (300, 1000, 5000):
        m = slots_for_whole_tile_rounds(n, 9, 9)
        rounds = (n * 81) // 65536
        assert m <= n and (m == n if rounds == 0 else m * 81 <= rounds * 65536 < (m + 3) * 81)   # (whole boards per group)


def test_gpu_telemetry_parser_reads_rocm_smi_json():
    text = ('WARNING: AMD GPU device(s) is/are in a low-power state.\n{"card0": {"temperature_hotspot (C)": "58", "average_umc_activity (%)": "7", '
            '"current_socket_power (W)": "1355", "current_gfxclks (MHz)": "[\'2103\', \'2051\', \'N/A\']", "ppt_residency_acc (Count)": "644259", '
            '"socket_thm_residency_acc (Count)": "0", "average_socket_power (W)": "N/A", "jpeg_activity (%)": "[\'N/A\']"}}')
    rec = bench.PowerSampler.parse(text)
    assert rec["socket_power_w"] == 1355.0 and rec["hotspot_c"] == 58.0 and rec["memory_controller_activity_pct"] == 7.0
    assert rec["power_limit_residency"] == 644259.0 and rec["thermal_residency"] == 0.0
    assert "gfxclk_mhz" not in rec                    # a list with an unreadable entry is dropped, not half-read


# ---- `python bench.py --gpus N` as its own launcher (no GPU needed: the ranks here are a stand-in script) -----------------

_STUB = """
import json, os, sys, time
r = int(os.environ["RANK"])
if os.environ.get("STUB_FAIL_RANK") == str(r):
    sys.exit(3)
if "STUB_FAIL_RANK" in os.environ:
    time.sleep(60)          # a healthy rank that would wait for the lost one in a collective
print(json.dumps({k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[1:]}))
"""


def _run_launcher(tmp_path, n, extra_env=None):
    import subprocess
    import sys
    stub = tmp_path / "rank_stub.py"
    stub.write_text(_STUB)
    code = ("import importlib.util, sys; spec = importlib.util.spec_from_file_location('bench', %r); b = importlib.util.module_from_spec(spec); "
            "spec.loader.exec_module(b); a = b.parse_args(['--gpus', '%d', '--steps', '3']); sys.exit(b.launch_ranks(a, ['--gpus', '%d', '--steps', '3'], script=%r))"
            % (os.path.join(ROOT, "bench.py"), n, n, str(stub)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)


def test_plain_command_with_gpus_n_starts_n_ranks(tmp_path):
    import json
    r = _run_launcher(tmp_path, 3)
    assert r.returncode == 0, r.stderr
    out = [json.loads(x) for x in r.stdout.strip().splitlines()]
    err = [json.loads(x) for x in r.stderr.strip().splitlines() if x.startswith("{")]
    assert len(out) == 1 and out[0]["RANK"] == "0"                       # only rank 0 owns stdout (the one JSON line)
    assert sorted(e["RANK"] for e in err) == ["1", "2"]
    for e in out + err:
        assert e["WORLD_SIZE"] == "3" and e["LOCAL_RANK"] == e["RANK"] and e["MASTER_ADDR"] == "127.0.0.1"
        assert e["MASTER_PORT"] == out[0]["MASTER_PORT"] and e["argv"] == ["--gpus", "3", "--steps", "3"]


def test_launcher_reports_a_failed_rank_and_ends_the_others(tmp_path):
    import time
    t0 = time.time()
    r = _run_launcher(tmp_path, 2, {"STUB_FAIL_RANK": "1"})
    assert r.returncode == 3 and "rank 1 exited with status 3" in r.stderr
    assert time.time() - t0 < 30                                         # rank 0's 60 s wait was cut short


def test_the_launcher_is_taken_only_without_a_rank_in_the_environment():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if _args.gpus > 1 and "RANK" not in os.environ:' in src      # under torchrun the ranks exist already
    # the launcher itself must not initialise the GPU: nothing of torch or the engine is imported before it runs
    head = src[:src.index("def main(args):")]
    assert "import torch" not in head.split("def launch_ranks")[1].split("def exchange_exercise")[0]


def test_telemetry_is_off_under_a_profiler(monkeypatch):
    for k in list(os.environ):
        if k.startswith(("ROCP", "HSA_TOOLS")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    assert not bench.PowerSampler.under_profiler()
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.PowerSampler.under_profiler()


def test_post_region_guard_prints_the_line_once_with_whatever_finished():
    import json
    lines = []
    out = {"value": 1.5, "exchange": None}
    g = bench.PostRegionGuard(out, 0, 30.0, emit=lines.append, leave=lambda code: None)
    g.record("exchange", lambda: {"ok": True})
    g.record("tower_parity", lambda: 1 / 0)          # a failing exercise is reported in its place
    g.emit()
    g.emit()
    g.done()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 1.5 and d["exchange"] == {"ok": True} and "ZeroDivisionError" in d["tower_parity"]["error"]
    assert "post_region_note" not in d


def test_post_region_guard_abandons_a_hanging_exercise():
    """An exercise that never returns (a collective that hangs) may not cost the measurement: after the deadline the
    line goes out as it stands and the process is told to leave; what the hung exercise returns later is ignored."""
    import json
    import threading
    lines, left = [], threading.Event()
    out = {"value": 2.5, "exchange": None}
    g = bench.PostRegionGuard(out, 0, 0.2, emit=lines.append, leave=lambda code: left.set())
    release = threading.Event()
    g.record("exchange", lambda: (release.wait(5.0), {"late": True})[1])   # blocks past the deadline
    assert left.wait(5.0)
    release.set()
    g.emit()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 2.5 and d["exchange"] is None and "did not finish" in d["post_region_note"]


def test_post_region_guard_is_silent_on_other_ranks():
    lines, left = [], []
    g = bench.PostRegionGuard(None, 3, 0.05, emit=lines.append, leave=left.append)
    import time
    time.sleep(0.3)
    assert lines == [] and left == [0]


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    try:    # a zombie that nobody has reaped yet is not running any more
        with open("/proc/%d/stat" % pid) as f:
            return f.read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_ranks_do_not_outlive_a_launcher_that_is_stopped_or_killed(tmp_path):
    """SIGTERM: the launcher ends its ranks.  SIGKILL (no handler can run): the ranks were started with a parent-death
    signal and go with it - a rank left behind would keep its GPU from the next run."""
    import signal
    import subprocess
    import sys
    import time
    stub = tmp_path / "sleeper.py"
    stub.write_text("import os, sys, time\nopen(os.path.join(%r, 'pid_' + os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(120)\n" % str(tmp_path))
    code = ("import importlib.util, sys; spec = importlib.util.spec_from_file_location('bench', %r); b = importlib.util.module_from_spec(spec); "
            "spec.loader.exec_module(b); a = b.parse_args(['--gpus', '2']); sys.exit(b.launch_ranks(a, ['--gpus', '2'], script=%r))"
            % (os.path.join(ROOT, "bench.py"), str(stub)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    for sig in (signal.SIGTERM, signal.SIGKILL):
        for f in tmp_path.glob("pid_*"):
            f.unlink()
        p = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t0 = time.time()
        while len(list(tmp_path.glob("pid_*"))) < 2 and time.time() - t0 < 60:
            time.sleep(0.1)
        pids = [int(f.read_text()) for f in tmp_path.glob("pid_*")]
        assert len(pids) == 2 and all(_alive(x) for x in pids)
        p.send_signal(sig)
        p.wait(timeout=30)
        t0 = time.time()
        while any(_alive(x) for x in pids) and time.time() - t0 < 20:
            time.sleep(0.1)
        assert not any(_alive(x) for x in pids), sig
