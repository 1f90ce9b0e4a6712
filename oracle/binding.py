"""ctypes binding of the CPU oracle (oracle/libdopf_oracle.so) — TEST INFRASTRUCTURE ONLY.

The oracle exports the C ABI of include/dopf.h with the prefix ``oracle_`` (plus a solve-mode argument on
create and a few oracle-only entry points), so the product package's generic ``CApi``/``Engine`` classes can
drive it. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing in
decentralopf.jl_amd/ does (tests/test_capi_symbols.py checks).
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import dopf_pkg  # noqa: E402

dopf_pkg.load()
from decentralopf_jl_amd import _capi  # noqa: E402

ORACLE_LIB = os.path.join(ROOT, "oracle", "libdopf_oracle.so")
MODE_LITERAL, MODE_EXACT = 0, 1


class OracleApi(_capi.CApi):
    def __init__(self, path: str = ORACLE_LIB):
        super().__init__(path, "oracle_", create_extra=(C.c_int32,))
        ctxp, dp, ip = C.c_void_p, _capi.c_double_p, _capi.c_int32_p
        self._sig("set_threads", None, [ctxp, C.c_int32])
        self._sig("get_agent_slacks", C.c_int, [ctxp, C.c_int32, dp, dp])
        self._sig("calculate_ptdf", C.c_int, [C.c_int32, C.c_int32, ip, ip, dp, C.c_int32, dp])
        self._sig("qp_solve", C.c_int, [C.c_int32, C.c_int32] + [dp] * 8 + [ip])


def set_threads(engine: _capi.Engine, n: int) -> None:
    """OpenMP threads of an oracle engine (1 = serial)."""
    engine.api.set_threads(engine._ctx, int(n))
