"""ctypes binding of the C ABI declared in include/dopf.h.

The product library is csrc/libdopf_hip.so (hand-written HIP for gfx950). There is NO CPU
fallback: if the library is missing or HIP cannot start, loading raises.

``CApi`` is generic over the symbol prefix so that another library exporting the same
signatures can be driven by the same Engine class (the test suite's checker does that from
oracle/binding.py); nothing in this package knows about such a library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "csrc", "libdopf_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class DopfProblem(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("L", C.c_int32), ("T", C.c_int32), ("G", C.c_int32), ("S", C.c_int32),
        ("demand", c_double_p), ("ptdf", c_double_p), ("f_max", c_double_p),
        ("gen_mc", c_double_p), ("gen_pmax", c_double_p), ("gen_node", c_int32_p),
        ("sto_mc", c_double_p), ("sto_pmax", c_double_p), ("sto_emax", c_double_p),
        ("sto_node", c_int32_p),
    ]


class DopfParams(C.Structure):
    _fields_ = [
        ("gamma", C.c_double), ("w_flow", C.c_double), ("w_prox", C.c_double),
        ("eps", C.c_double), ("mask_thr", C.c_double),
        ("max_iters", C.c_int32), ("n_agents_global", C.c_int32),
        ("device", C.c_int32), ("flags", C.c_int32),
        ("stream", C.c_void_p),
    ]


class DopfTiming(C.Structure):
    _fields_ = [("tables_ms", C.c_double), ("gen_ms", C.c_double), ("sto_ms", C.c_double),
                ("slack_ms", C.c_double), ("reduce_ms", C.c_double), ("dual_ms", C.c_double),
                ("iter_ms", C.c_double), ("empty_ms", C.c_double), ("iters", C.c_int32),
                ("agents_fused", C.c_int32), ("tail_fused", C.c_int32),
                ("slack_in_dual", C.c_int32), ("quiet", C.c_int32), ("sto_lean", C.c_int32), ("persist", C.c_int32)]


class DopfCentralResult(C.Structure):
    _fields_ = [("objective", C.c_double), ("dual_objective", C.c_double), ("primal_infeasibility", C.c_double),
                ("gap", C.c_double), ("iterations", C.c_int32), ("converged", C.c_int32)]


F_NO_GRAPH = 1
F_OVERLAP_AGENTS = 2
F_NO_WARM_START = 4
F_NO_ROW_SKIP = 8
F_NO_TAIL_FUSE = 4096
F_TIME_CALLS = 8192
F_NO_FUSE = 16
F_COMM_HOST = 64
F_DEBUG_ROOT_CAP = 128
F_KEEP_DELTAS = 256
F_COMM_GRAPH = 512
F_COMM_P2P = 1024
F_DEBUG_LEAVE = 2048
F_STO_GENERAL = 16384
F_NO_QUIET = 32768
F_XCHG_OWNER = 65536
F_XCHG_ALLGATHER = 131072
F_NO_TAIL_XCHG = 262144
F_NET_SMALL_ITEMS = 524288
F_PERSIST = 1048576
COMM_ID_BYTES = 128
XCHG_HANDLE_BYTES = 64


class DopfError(RuntimeError):
    pass


def _f64(a, n=None) -> np.ndarray:
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if n is not None and arr.size != n:
        raise ValueError(f"expected {n} doubles, got {arr.size}")
    return arr


def _i32(a, n=None) -> np.ndarray:
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1))
    if n is not None and arr.size != n:
        raise ValueError(f"expected {n} int32, got {arr.size}")
    return arr


def _dp(arr: Optional[np.ndarray]):
    return None if arr is None else arr.ctypes.data_as(c_double_p)


class CApi:
    """One loaded shared library exporting <prefix>create/iterate/... (include/dopf.h)."""

    def __init__(self, path: str, prefix: str = "dopf_", create_extra=()):
        if not os.path.exists(path):
            raise DopfError(
                f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the HIP path.")
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        p = prefix
        L = self.lib
        ctxp = C.c_void_p
        self.has_mode = bool(create_extra)          # <prefix>create takes extra trailing arguments
        create = getattr(L, p + "create")
        create.restype = C.c_int
        create.argtypes = [C.POINTER(ctxp), C.POINTER(DopfProblem), C.POINTER(DopfParams)] + list(create_extra)
        self._create = create
        self._sig("destroy", None, [ctxp])
        self._sig("last_error", C.c_char_p, [ctxp])
        self._sig("iterate", C.c_int, [ctxp, C.c_int32, c_int32_p, c_int32_p])
        self._sig("local_update", C.c_int, [ctxp])
        self._sig("apply_consensus", C.c_int, [ctxp])
        self._sig("consensus_size", C.c_int64, [ctxp])
        self._sig("consensus_ptr", C.c_void_p, [ctxp])
        self._sig("sync", C.c_int, [ctxp, c_int32_p, c_int32_p])
        self._sig("get_duals", C.c_int, [ctxp, c_double_p, c_double_p, c_double_p])
        self._sig("get_duals_used", C.c_int, [ctxp, c_double_p, c_double_p, c_double_p])
        self._sig("get_primal", C.c_int, [ctxp, c_double_p, c_double_p, c_double_p, c_double_p])
        self._sig("get_consensus", C.c_int, [ctxp, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p])
        self._sig("get_residuals", C.c_int, [ctxp, c_double_p, c_double_p, c_double_p, c_int32_p])
        self._sig("get_nodal_price", C.c_int, [ctxp, C.c_int32, c_double_p])
        self._sig("set_state", C.c_int, [ctxp] + [c_double_p] * 8 + [C.c_int32])
        if prefix == "dopf_":
            self._sig("bind_consensus", C.c_int, [ctxp, C.c_void_p])
            self._sig("solver_failures", C.c_int64, [ctxp])
            self._sig("iterate_timed", C.c_int, [ctxp, C.c_int32, C.POINTER(DopfTiming)])
            self._sig("debug_stats", C.c_int, [ctxp, C.POINTER(C.c_uint64)])
            self._sig("version", C.c_char_p, [])
            self._sig("default_params", None, [C.POINTER(DopfParams)])
            self._sig("get_agent_slacks", C.c_int, [ctxp, C.c_int32, c_double_p, c_double_p])
            self._sig("get_agent_penalty", C.c_int, [ctxp, C.c_int32, c_double_p, c_double_p])
            self._sig("get_residual_vectors", C.c_int, [ctxp, c_double_p, c_double_p, c_double_p])
            self._sig("get_penalty_sums", C.c_int, [ctxp, c_double_p])
            self._sig("central_solve", C.c_int, [C.POINTER(DopfProblem), C.POINTER(DopfParams), C.c_double, C.c_int32,
                                                 C.POINTER(DopfCentralResult)] + [c_double_p] * 9)
            self._sig("get_node_results", C.c_int, [ctxp, c_double_p, c_double_p, c_double_p])
            self._sig("last_call_ms", C.c_double, [ctxp])
            # consensus sum across GPUs inside the library (RCCL, loaded on first use)
            self._sig("comm_unique_id", C.c_int, [C.c_void_p])
            self._sig("comm_init", C.c_int, [ctxp, C.c_int32, C.c_int32, C.c_void_p])
            self._sig("xchg_export", C.c_int, [ctxp, C.c_int32, C.c_void_p])
            self._sig("xchg_init", C.c_int, [ctxp, C.c_int32, C.c_int32, C.c_void_p])
            self._sig("comm_info", C.c_int, [ctxp, c_int32_p, c_int32_p, c_int32_p])
            self._sig("multi_create", C.c_int, [C.POINTER(ctxp), C.POINTER(DopfProblem), C.POINTER(DopfParams), C.c_int32, c_int32_p])
            self._sig("multi_destroy", None, [ctxp])
            self._sig("multi_last_error", C.c_char_p, [ctxp])
            self._sig("multi_iterate", C.c_int, [ctxp, C.c_int32, c_int32_p, c_int32_p])
            self._sig("multi_get_primal", C.c_int, [ctxp, c_double_p, c_double_p, c_double_p, c_double_p])
            self._sig("multi_size", C.c_int32, [ctxp])
            self._sig("multi_ctx", ctxp, [ctxp, C.c_int32])

    def _sig(self, name, restype, argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        f.argtypes = argtypes
        setattr(self, name, f)


_hip_api: Optional[CApi] = None
_runtime_pinned = False


def _pin_hip_runtime():
    """One HIP runtime per process. PyTorch's wheels bundle their own libamdhip64.so / libhsa-runtime64.so (same SONAMEs as
    the system ROCm's, requested under another file name), so a process that loads libdopf_hip.so first (system runtime) and
    imports torch later ends up with TWO runtimes: the second one finds no GPU ("No HIP GPUs are available"), and a system
    RCCL bound to the other copy's runtime is an ABI mismatch. When PyTorch is installed but not imported yet, its copy of the
    runtime is loaded here first: libdopf_hip.so's DT_NEEDED libamdhip64.so.7 then binds to it by SONAME, and a later
    `import torch` finds its runtime already in place. Without PyTorch (a C or Julia host) nothing happens: system ROCm."""
    global _runtime_pinned
    if _runtime_pinned:
        return
    _runtime_pinned = True
    import sys
    if "torch" in sys.modules:
        return                          # its runtime is loaded already; ours will bind to it
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return                  # a HIP runtime is mapped already (whoever loaded it): never add a second one
    except OSError:
        pass
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return                  # (an unusable bundle: leave everything to the system runtime)


def hip_api() -> CApi:
    """The product library. Raises (never falls back) if it is not built or cannot load."""
    global _hip_api
    if _hip_api is None:
        _pin_hip_runtime()
        _hip_api = CApi(HIP_LIB_PATH, "dopf_")
    return _hip_api


def default_params(**kw) -> DopfParams:
    """The reference's literals: gamma 0.3, flow weight 10, prox 1,
    eps 1e-3, mask threshold 1e-2."""
    q = DopfParams(gamma=0.3, w_flow=10.0, w_prox=1.0, eps=1e-3, mask_thr=1e-2, max_iters=0,
                   n_agents_global=0, device=-1, flags=0, stream=None)
    for k, v in kw.items():
        if not hasattr(q, k):
            raise TypeError(f"unknown parameter {k}")
        setattr(q, k, v)
    return q


class Engine:
    """A context of the C ABI with numpy in/out. Mirrors include/dopf.h one to one."""

    def __init__(self, api: CApi, *, N, L, T, demand, ptdf, f_max, gen_mc, gen_pmax, gen_node,
                 sto_mc, sto_pmax, sto_emax, sto_node, params: Optional[DopfParams] = None,
                 mode: Optional[int] = None):
        self.api = api
        self.N, self.L, self.T = int(N), int(L), int(T)
        gen_mc = _f64(gen_mc)
        sto_mc = _f64(sto_mc)
        self.G, self.S = gen_mc.size, sto_mc.size
        keep = dict(
            demand=_f64(demand, self.N * self.T), ptdf=_f64(ptdf, self.L * self.N),
            f_max=_f64(f_max, self.L), gen_mc=gen_mc, gen_pmax=_f64(gen_pmax, self.G),
            gen_node=_i32(gen_node, self.G), sto_mc=sto_mc, sto_pmax=_f64(sto_pmax, self.S),
            sto_emax=_f64(sto_emax, self.S), sto_node=_i32(sto_node, self.S))
        prob = DopfProblem(N=self.N, L=self.L, T=self.T, G=self.G, S=self.S)
        for k, v in keep.items():
            ptr = v.ctypes.data_as(c_int32_p if v.dtype == np.int32 else c_double_p)
            setattr(prob, k, ptr)
        self.params = params if params is not None else default_params()
        self._ctx = C.c_void_p()
        args = [C.byref(self._ctx), C.byref(prob), C.byref(self.params)]
        if api.has_mode:
            args.append(C.c_int32(0 if mode is None else mode))
        rc = api._create(*args)
        if rc != 0:
            msg = api.last_error(None)
            raise DopfError(f"{api.prefix}create failed ({rc}): {msg.decode() if msg else ''}")

    # -- lifecycle -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.api.destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            msg = self.api.last_error(self._ctx)
            raise DopfError(f"{self.api.prefix}* failed ({rc}): {msg.decode() if msg else ''}")

    # -- iteration -----------------------------------------------------------------------------
    def iterate(self, n_iters: int):
        done, conv = C.c_int32(0), C.c_int32(0)
        self._chk(self.api.iterate(self._ctx, int(n_iters), C.byref(done), C.byref(conv)))
        return done.value, bool(conv.value)

    def local_update(self):
        self._chk(self.api.local_update(self._ctx))

    def apply_consensus(self):
        self._chk(self.api.apply_consensus(self._ctx))

    def consensus_size(self) -> int:
        return int(self.api.consensus_size(self._ctx))

    def consensus_ptr(self) -> int:
        return int(self.api.consensus_ptr(self._ctx) or 0)

    def bind_consensus(self, device_ptr: int):
        self._chk(self.api.bind_consensus(self._ctx, C.c_void_p(device_ptr)))

    def sync(self):
        it, conv = C.c_int32(0), C.c_int32(0)
        self._chk(self.api.sync(self._ctx, C.byref(it), C.byref(conv)))
        return it.value, bool(conv.value)

    def get_node_results(self):
        """ResultNode.{generation, discharge, charge} of the last result, each (N, T) (src/structures/results.jl:19-35)."""
        outs = [np.zeros(self.N * self.T) for _ in range(3)]
        self._chk(self.api.get_node_results(self._ctx, *[_dp(o) for o in outs]))
        return tuple(o.reshape(self.T, self.N).T.copy() for o in outs)

    def last_call_ms(self) -> float:
        """F_TIME_CALLS: device-side milliseconds of the last iterate() call's launches (-1 if not measured)."""
        return float(self.api.last_call_ms(self._ctx))

    def solver_failures(self) -> int:
        return int(self.api.solver_failures(self._ctx))

    def iterate_timed(self, n_iters: int) -> dict:
        """HIP-event timing of every kernel of the chain (eager launches), averages in ms."""
        t = DopfTiming()
        self._chk(self.api.iterate_timed(self._ctx, int(n_iters), C.byref(t)))
        return {k: getattr(t, k) for k, _ in DopfTiming._fields_}

    def warm_start_stats(self):
        """(storages the warm-start kernel solved, storages it left to the scan kernel) in the LAST iteration."""
        out = (C.c_uint64 * 15)()
        self._chk(self.api.debug_stats(self._ctx, out))
        return int(out[3]), int(out[4])

    # -- consensus sum across ranks inside the library (one process per GPU) ---------------------
    def comm_unique_id(self) -> bytes:
        """128 opaque bytes (ncclUniqueId): rank 0 creates them, every rank passes them to comm_init."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = self.api.comm_unique_id(buf)
        if rc != 0:
            msg = self.api.last_error(None)
            raise DopfError(f"dopf_comm_unique_id failed ({rc}): {msg.decode() if msg else ''}")
        return buf.raw

    def comm_init(self, world: int, rank: int, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        self._chk(self.api.comm_init(self._ctx, int(world), int(rank), buf))

    # -- the same sum by the peer exchange (direct stores into the peers' memory, no collective library) ----------
    def xchg_export(self, world: int) -> bytes:
        """Allocates this rank's receive area; 64 opaque bytes (hipIpcMemHandle_t) to gather over all ranks."""
        buf = C.create_string_buffer(XCHG_HANDLE_BYTES)
        self._chk(self.api.xchg_export(self._ctx, int(world), buf))
        return buf.raw

    def xchg_init(self, world: int, rank: int, handles):
        """handles: the world handles in rank order (list of bytes or one bytes object)."""
        blob = b"".join(handles) if not isinstance(handles, (bytes, bytearray)) else bytes(handles)
        if len(blob) != world * XCHG_HANDLE_BYTES:
            raise ValueError("need world x 64 bytes of handles")
        buf = C.create_string_buffer(blob, len(blob))
        self._chk(self.api.xchg_init(self._ctx, int(world), int(rank), buf))

    def comm_info(self):
        """(world, rank, collective captured in the hipGraph?)"""
        w, r, g = C.c_int32(1), C.c_int32(0), C.c_int32(0)
        self._chk(self.api.comm_info(self._ctx, C.byref(w), C.byref(r), C.byref(g)))
        return w.value, r.value, bool(g.value)

    # -- getters -------------------------------------------------------------------------------
    def _duals(self, fn):
        lam = np.zeros(self.T)
        mu = np.zeros(self.L * self.T)
        rho = np.zeros(self.L * self.T)
        self._chk(fn(self._ctx, _dp(lam), _dp(mu), _dp(rho)))
        # Julia shapes: lambda (T), mu/rho (L, T) column-major
        return lam, mu.reshape(self.T, self.L).T.copy(), rho.reshape(self.T, self.L).T.copy()

    def get_duals(self):
        return self._duals(self.api.get_duals)

    def get_duals_used(self):
        return self._duals(self.api.get_duals_used)

    def get_primal(self):
        """P (G,T), D, C, E (S,T)."""
        P = np.zeros(self.G * self.T)
        D = np.zeros(self.S * self.T)
        Cc = np.zeros(self.S * self.T)
        E = np.zeros(self.S * self.T)
        self._chk(self.api.get_primal(self._ctx, _dp(P), _dp(D), _dp(Cc), _dp(E)))
        return (P.reshape(self.G, self.T), D.reshape(self.S, self.T),
                Cc.reshape(self.S, self.T), E.reshape(self.S, self.T))

    def get_consensus(self):
        """injection (N,T), avg_U, avg_K, line_utilization (L,T), total_costs."""
        inj = np.zeros(self.N * self.T)
        aU = np.zeros(self.L * self.T)
        aK = np.zeros(self.L * self.T)
        fl = np.zeros(self.L * self.T)
        cost = C.c_double(0)
        self._chk(self.api.get_consensus(self._ctx, _dp(inj), _dp(aU), _dp(aK), _dp(fl), C.byref(cost)))
        r = lambda v, rows: v.reshape(self.T, rows).T.copy()
        return r(inj, self.N), r(aU, self.L), r(aK, self.L), r(fl, self.L), cost.value

    def get_residuals(self):
        a, b, c, it = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int32(0)
        self._chk(self.api.get_residuals(self._ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(it)))
        return a.value, b.value, c.value, it.value

    def get_residual_vectors(self):
        """Convergence.{lambda_res, mue_res, rho_res}[end]: |dual change| per entry, (T), (L,T), (L,T)."""
        lam = np.zeros(self.T)
        mu = np.zeros(self.L * self.T)
        rho = np.zeros(self.L * self.T)
        self._chk(self.api.get_residual_vectors(self._ctx, _dp(lam), _dp(mu), _dp(rho)))
        return lam, mu.reshape(self.T, self.L).T.copy(), rho.reshape(self.T, self.L).T.copy()

    def get_agent_slacks(self, agent: int):
        """ResultGenerator/ResultStorage.U, .K of the last solve, (L,T) each; agent: generators first."""
        U = np.zeros(self.L * self.T)
        K = np.zeros(self.L * self.T)
        self._chk(self.api.get_agent_slacks(self._ctx, int(agent), _dp(U), _dp(K)))
        return U.reshape(self.T, self.L).T.copy(), K.reshape(self.T, self.L).T.copy()

    def get_agent_penalty(self, agent: int, delta=None):
        """PenaltyTerm(energy_balance, upper_flow, lower_flow) of the agent's last solve, (T) each. `delta`: the
        agent's injection change of that iteration; needed on a copper plate (the device keeps it only with lines)."""
        pen = np.zeros(3 * self.T)
        d = None if delta is None else _f64(delta, self.T)
        self._chk(self.api.get_agent_penalty(self._ctx, int(agent), _dp(d), _dp(pen)))
        return pen[:self.T].copy(), pen[self.T:2 * self.T].copy(), pen[2 * self.T:].copy()

    def get_penalty_sums(self):
        """Result.penalty_term (results.jl:66-70): the three penalty vectors summed over all agents, (T) each. Networks
        with F_KEEP_DELTAS (the device must hold the injection changes of the last x-update)."""
        pen = np.zeros(3 * self.T)
        self._chk(self.api.get_penalty_sums(self._ctx, _dp(pen)))
        return pen[:self.T].copy(), pen[self.T:2 * self.T].copy(), pen[2 * self.T:].copy()

    def get_nodal_price(self, which: int = 0):
        out = np.zeros(self.N * self.T)
        self._chk(self.api.get_nodal_price(self._ctx, int(which), _dp(out)))
        return out.reshape(self.T, self.N).T.copy()

    def set_state(self, *, P=None, D=None, C_=None, avg_U=None, avg_K=None, lam=None, mu=None,
                  rho=None, iteration: int = 1):
        """Matrices in Julia shape: P (G,T) etc. agent-major rows; avg_U/mu/rho (L,T)."""
        def am(a, rows):  # (rows, T) -> [t + T*row]
            return None if a is None else _f64(np.asarray(a, dtype=np.float64).reshape(rows, self.T))

        def cm(a, rows):  # (rows, T) -> column-major [r + rows*t]
            return None if a is None else _f64(np.asarray(a, dtype=np.float64).reshape(rows, self.T).T)
        bufs = [am(P, self.G), am(D, self.S), am(C_, self.S), cm(avg_U, self.L), cm(avg_K, self.L),
                None if lam is None else _f64(lam, self.T), cm(mu, self.L), cm(rho, self.L)]
        self._chk(self.api.set_state(self._ctx, *[_dp(b) for b in bufs], int(iteration)))


class _ShardView(Engine):
    """A shard's context inside a MultiEngine (owned by the dopf_multi object: never destroyed from here)."""

    def __init__(self, api: CApi, ctx, N, L, T, G, S):     # noqa: super().__init__ deliberately not called
        self.api, self._ctx = api, ctx
        self.N, self.L, self.T, self.G, self.S = N, L, T, G, S

    def close(self):
        self._ctx = C.c_void_p()


class MultiEngine:
    """One process, n GPUs: dopf_multi_* (the library shards the agents, owns the RCCL communicator and one host
    thread per device). Replicated state (duals, consensus, prices, residuals) is read from shard 0."""

    def __init__(self, api: CApi, n_gpus: int, *, N, L, T, demand, ptdf, f_max, gen_mc, gen_pmax, gen_node,
                 sto_mc, sto_pmax, sto_emax, sto_node, params: Optional[DopfParams] = None, devices=None):
        self.api = api
        self.N, self.L, self.T = int(N), int(L), int(T)
        gen_mc = _f64(gen_mc)
        sto_mc = _f64(sto_mc)
        self.G, self.S = gen_mc.size, sto_mc.size
        keep = dict(
            demand=_f64(demand, self.N * self.T), ptdf=_f64(ptdf, self.L * self.N),
            f_max=_f64(f_max, self.L), gen_mc=gen_mc, gen_pmax=_f64(gen_pmax, self.G),
            gen_node=_i32(gen_node, self.G), sto_mc=sto_mc, sto_pmax=_f64(sto_pmax, self.S),
            sto_emax=_f64(sto_emax, self.S), sto_node=_i32(sto_node, self.S))
        prob = DopfProblem(N=self.N, L=self.L, T=self.T, G=self.G, S=self.S)
        for k, v in keep.items():
            setattr(prob, k, v.ctypes.data_as(c_int32_p if v.dtype == np.int32 else c_double_p))
        self.params = params if params is not None else default_params()
        dev = None if devices is None else _i32(devices, n_gpus)
        self._m = C.c_void_p()
        rc = api.multi_create(C.byref(self._m), C.byref(prob), C.byref(self.params), int(n_gpus),
                              None if dev is None else dev.ctypes.data_as(c_int32_p))
        if rc != 0:
            msg = api.multi_last_error(None)
            raise DopfError(f"dopf_multi_create failed ({rc}): {msg.decode() if msg else ''}")
        self.n = int(api.multi_size(self._m))

    def _chk(self, rc):
        if rc != 0:
            msg = self.api.multi_last_error(self._m)
            raise DopfError(f"dopf_multi_* failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_m", None) is not None and self._m.value:
            self.api.multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def iterate(self, n_iters: int):
        done, conv = C.c_int32(0), C.c_int32(0)
        self._chk(self.api.multi_iterate(self._m, int(n_iters), C.byref(done), C.byref(conv)))
        return done.value, bool(conv.value)

    def get_primal(self):
        P = np.zeros(self.G * self.T)
        D = np.zeros(self.S * self.T)
        Cc = np.zeros(self.S * self.T)
        E = np.zeros(self.S * self.T)
        self._chk(self.api.multi_get_primal(self._m, _dp(P), _dp(D), _dp(Cc), _dp(E)))
        return (P.reshape(self.G, self.T), D.reshape(self.S, self.T), Cc.reshape(self.S, self.T), E.reshape(self.S, self.T))

    def shard(self, i: int = 0) -> Engine:
        """Engine view of shard i's context (getters for the replicated state; do not iterate it directly)."""
        ctx = C.c_void_p(self.api.multi_ctx(self._m, int(i)))
        if not ctx.value:
            raise IndexError(i)
        def cut(total):              # the split dopf_multi_create makes (contiguous, remainder to the first shards)
            base, rem = divmod(total, self.n)
            return base + (1 if i < rem else 0)
        return _ShardView(self.api, ctx, self.N, self.L, self.T, cut(self.G), cut(self.S))


def central_solve(api: CApi, *, N, L, T, demand, ptdf, f_max, gen_mc, gen_pmax, gen_node, sto_mc, sto_pmax, sto_emax,
                  sto_node, tol: float = 1e-8, max_iters: int = 200000, params: Optional[DopfParams] = None) -> dict:
    """dopf_central_solve: the central reference (src/opf_central_reference.jl) as one LP solved on the GPU by a first-order
    primal-dual method. Arguments as Engine (PackedProblem.engine_kwargs()). Returns objective, gap, iterations and the
    reference script's outputs in Julia shapes: P (G,T), D/C/E (S,T), system_price (T), nodal_price (N,T),
    line_utilization (L,T)."""
    N, L, T = int(N), int(L), int(T)
    gen_mc = _f64(gen_mc)
    sto_mc = _f64(sto_mc)
    G, S = gen_mc.size, sto_mc.size
    keep = dict(demand=_f64(demand, N * T), ptdf=_f64(ptdf, L * N), f_max=_f64(f_max, L), gen_mc=gen_mc,
                gen_pmax=_f64(gen_pmax, G), gen_node=_i32(gen_node, G), sto_mc=sto_mc, sto_pmax=_f64(sto_pmax, S),
                sto_emax=_f64(sto_emax, S), sto_node=_i32(sto_node, S))
    prob = DopfProblem(N=N, L=L, T=T, G=G, S=S)
    for k, v in keep.items():
        setattr(prob, k, v.ctypes.data_as(c_int32_p if v.dtype == np.int32 else c_double_p))
    q = params if params is not None else default_params()
    res = DopfCentralResult()
    P, D, Cc, E = np.zeros(G * T), np.zeros(S * T), np.zeros(S * T), np.zeros(S * T)
    lam, nodal, flow = np.zeros(T), np.zeros(N * T), np.zeros(L * T)
    fu, fl = np.zeros(L * T), np.zeros(L * T)
    rc = api.central_solve(C.byref(prob), C.byref(q), float(tol), int(max_iters), C.byref(res), _dp(P), _dp(D), _dp(Cc), _dp(E),
                           _dp(lam), _dp(nodal), _dp(flow), _dp(fu), _dp(fl))
    if rc != 0:
        msg = api.last_error(None)
        raise DopfError(f"dopf_central_solve failed ({rc}): {msg.decode() if msg else ''}")
    return dict(objective=res.objective, dual_objective=res.dual_objective, primal_infeasibility=res.primal_infeasibility,
                gap=res.gap, iterations=res.iterations, converged=bool(res.converged),
                P=P.reshape(G, T), D=D.reshape(S, T), C=Cc.reshape(S, T), E=E.reshape(S, T), system_price=lam,
                nodal_price=nodal.reshape(T, N).T.copy(), line_utilization=flow.reshape(T, L).T.copy(),
                flow_upper_dual=fu.reshape(T, L).T.copy(), flow_lower_dual=fl.reshape(T, L).T.copy())
