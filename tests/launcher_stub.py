"""A stand-in for one launcher-started multi-GPU attempt of bench.py (tests/test_bench_launcher.py): behaves as the plan in
STUB_PLAN says for the --comm mode it is started with — "sleep" (hangs past any budget), "abort" (dies on SIGABRT),
"exit3" (plain failure), "line" (prints one JSON line that carries what the watchdog handed down about earlier attempts)."""
import json
import os
import signal
import sys
import time

mode = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--comm=")][0]
what = json.loads(os.environ["STUB_PLAN"])[mode]
if what == "sleep":
    print("not a result line")
    sys.stdout.flush()
    time.sleep(600)
elif what == "abort":
    os.kill(os.getpid(), signal.SIGABRT)
elif what == "exit3":
    sys.exit(3)
else:
    print("chatter before the line")
    print(json.dumps({"metric": "stub", "comm": {"mode": mode, "transports_given_up": json.loads(os.environ.get("DOPF_BENCH_LOST_ATTEMPTS", "[]"))}}))
