"""Bounded, seeded slices of the fuzzers (scripts/fuzz_*.py) as part of the GPU suite.

Every real memory bug of this code base so far was found by one of these fuzzers, not by a fixed parity case (round 2: an
out-of-bounds read of the stored prices that depended on where the allocator had put the array — scripts/fuzz_sharded.py).
Each slice runs in a CHILD process (a fresh process, nothing re-execs this one) in guard mode (DOPF_GUARD=1: every device
array of the library ends on the last bytes of its own mapping, so an access past an array's end is a fault that names
it), with fixed seeds, sized for <= ~30 s each. The scripts compare the HIP path with the oracle's exact mode (and with
other launch chains of the library) on random cases; a slice passes when the script reports `bad 0`."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# script, cases, seed, extra environment, what it pins
SLICES = [
    ("fuzz_parity", 260, 411, {}, "one-step parity on random small cases (networks and copper plates, ragged shapes)"),
    ("fuzz_free", 110, 412, {}, "free runs side by side: warm starts, certificates, hand-over to the scan body"),
    ("fuzz_lean", 60, 413, {}, "the lean copper-plate storage body vs oracle and vs the general body"),
    ("fuzz_sharded", 110, 414, {"GPU_MAX_HW_QUEUES": "8"}, "2-3 shards on one device vs a single context"),
    ("fuzz_quiet", 45, 415, {}, "the quiet network chain (no k_slack launch) bit for bit against the chain it replaces"),
    ("fuzz_net_wide", 10, 416, {}, "wide networks: 12-300 nodes, up to 400 lines, 24-192 steps"),
    ("fuzz_oracle", 40, 417, {"ORACLE_MODE": "0"}, "HIP vs the LITERAL mode of the oracle (term-by-term QP, interior point)"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("script,n,seed,env,what", SLICES, ids=[s[0] for s in SLICES])
def test_fuzz_slice_in_guard_mode(script, n, seed, env, what):
    path = os.path.join(ROOT, "scripts", script + ".py")
    if script == "fuzz_oracle":
        path = os.path.join(ROOT, "scripts", "fuzz_parity.py")      # the literal mode is a switch of the parity fuzzer
    r = subprocess.run([sys.executable, path, str(n), str(seed)], cwd=ROOT, capture_output=True, text=True, timeout=420,
                       env=dict(os.environ, DOPF_GUARD="1", **env))
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-1500:]
    assert r.returncode == 0, f"{script}: {what}\n{tail}"
    done = [l for l in r.stdout.splitlines() if l.startswith("done:")]
    assert done, tail
    assert "MISMATCH" not in r.stdout and "SOLVER FAILURES" not in r.stdout, tail
    m = re.search(r"bad (\d+)", done[-1])
    assert m and int(m.group(1)) == 0, done[-1]
