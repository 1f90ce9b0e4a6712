"""bench.py's watchdog for launcher-less multi-GPU runs (`python bench.py --gpus N`: the driver's N = 1 command line form),
driven with stub children on the CPU: one budget for all attempts, the first attempt gets 60 % of it, an attempt the
watchdog had to end (deadline, signal, exit code) is carried into the final line, and the fallback's line is relayed.
Children are fresh processes (tests/launcher_stub.py); nothing here touches a GPU."""
import io
import json
import os
import sys
import time
import types
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

STUB = os.path.join(ROOT, "tests", "launcher_stub.py")


def launch(monkeypatch, plan, comm="auto", budget=6.0):
    monkeypatch.setenv("STUB_PLAN", json.dumps(plan))
    args = types.SimpleNamespace(comm=comm, launch_timeout=budget, gpus=2)
    seen = []

    def child(mode, port, rest):
        seen.append((mode, port, list(rest)))
        return [sys.executable, STUB] + rest + [f"--comm={mode}"]
    # (sys.stdout.write is what self_launch relays the line with: capture through a pipe-like buffer)
    buf = io.StringIO()
    t0 = time.time()
    with redirect_stdout(buf):
        rc = bench.self_launch(args, ["--gpus", "2", "--comm", comm, "--steps", "5"], child_cmd=child, min_budget=1.0)
    return rc, buf.getvalue(), seen, time.time() - t0


def test_hung_first_attempt_is_killed_at_60_percent_and_named_in_the_line(monkeypatch):
    rc, out, seen, took = launch(monkeypatch, {"auto": "sleep", "torch": "line"}, budget=6.0)
    assert rc == 0
    line = json.loads(out.strip().splitlines()[-1])
    assert [m for m, _, _ in seen] == ["auto", "torch"]
    assert all("--comm" not in r and not any(a.startswith("--comm=") for a in r) for _, _, r in seen)      # the mode is the attempt's
    given_up = line["comm"]["transports_given_up"]
    assert given_up == [{"mode": "auto", "why": "killed at deadline (4 s)"}], given_up        # 60 % of 6 s
    assert line["comm"]["mode"] == "torch"
    assert 3.0 <= took <= 5.9, took                    # the fallback fitted under the caller's limit


def test_child_dying_on_a_signal_is_reported_as_that_signal(monkeypatch):
    rc, out, seen, _ = launch(monkeypatch, {"auto": "abort", "torch": "line"})
    assert rc == 0
    line = json.loads(out.strip().splitlines()[-1])
    assert line["comm"]["transports_given_up"] == [{"mode": "auto", "why": "exit signal 6"}]


def test_both_lost_attempts_end_in_a_failure_not_a_line(monkeypatch):
    rc, out, seen, took = launch(monkeypatch, {"auto": "exit3", "torch": "sleep"}, budget=4.0)
    assert rc == 1 and out.strip() == ""
    assert [m for m, _, _ in seen] == ["auto", "torch"]
    assert took <= 5.5                                   # the second attempt got what was left of the ONE budget, not a new one


def test_explicit_mode_is_one_attempt(monkeypatch):
    rc, out, seen, _ = launch(monkeypatch, {"p2p": "line"}, comm="p2p")
    assert rc == 0 and [m for m, _, _ in seen] == ["p2p"]
    assert json.loads(out.strip().splitlines()[-1])["comm"]["transports_given_up"] == []
