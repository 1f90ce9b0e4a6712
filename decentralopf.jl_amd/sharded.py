"""Agents sharded over ranks (one process per GPU), consensus by ONE all-reduce per iteration.

The reference has no parallelism at all; what shards is the agent loop of
optimize_all_subproblems! (src/optimization/subproblems.jl:1-17): every agent reads only iteration
k-1, and the only cross-agent data flow is the sum over agents of nodal injections and slacks
(Result(...), src/structures/results.jl:72-106). Each rank therefore owns a contiguous slice of the
node-sorted agent lists plus a replica of the O((N+L)T) consensus state, and per iteration does

    local_update()      x-updates of its agents + local sums  -> consensus buffer (N*T + 2*L*T + 1)
    all_reduce(SUM)     RCCL over xGMI (backend "nccl"); gloo in the CPU tests
    apply_consensus()   averages, flows, lambda/mu/rho update, residuals, stop test — replicated,
                        so duals stay identical on all ranks without a second collective

`ShardedADMM` is generic over the C-ABI library it drives: the product passes the HIP library and a
CUDA tensor bound as the consensus buffer; the CPU tests pass a host library and a numpy view.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional

import numpy as np

from . import _capi
from .network import PackedProblem


def host_consensus_view(engine: _capi.Engine) -> np.ndarray:
    """numpy view of a HOST consensus buffer (host libraries only)."""
    n = engine.consensus_size()
    ptr = engine.consensus_ptr()
    return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))


class ShardedADMM:
    def __init__(self, problem: PackedProblem, rank: int, world: int, *, api: Optional[_capi.CApi] = None,
                 all_reduce: Optional[Callable[[], None]] = None, mode: Optional[int] = None,
                 device: Optional[int] = None, n_agents_global_override: Optional[int] = None, **params):
        self.rank, self.world = rank, world
        self.problem = problem
        self.shard = problem.shard(rank, world)
        self.n_agents_global = n_agents_global_override or (problem.G + problem.S)
        self._tensor = None
        kw = dict(params)
        kw["n_agents_global"] = self.n_agents_global
        if api is None:                       # product path: HIP + RCCL via torch.distributed
            import torch
            import torch.distributed as dist
            if device is None:
                device = torch.cuda.current_device()
            dev = torch.device("cuda", device)
            # one dedicated (non-null) stream carries the kernels AND, as torch's current stream, orders
            # the RCCL all-reduce between dopf_local_update and dopf_apply_consensus
            self.stream = torch.cuda.Stream(device=dev)
            kw["device"] = device
            kw["stream"] = self.stream.cuda_stream
            self.engine = _capi.Engine(_capi.hip_api(), params=_capi.default_params(**kw),
                                       **self.shard.engine_kwargs())
            self._tensor = torch.zeros(self.engine.consensus_size(), dtype=torch.float64, device=dev)
            torch.cuda.synchronize(dev)
            self.engine.bind_consensus(self._tensor.data_ptr())
            if all_reduce is None:
                if world > 1:
                    t = self._tensor

                    def all_reduce():                 # runs on the engine's stream: step() makes it current
                        dist.all_reduce(t, op=dist.ReduceOp.SUM)
                else:
                    all_reduce = lambda: None
        else:
            self.engine = _capi.Engine(api, params=_capi.default_params(**kw), mode=mode,
                                       **self.shard.engine_kwargs())
            if all_reduce is None:
                if world > 1:
                    raise ValueError("a host library needs an all_reduce callable")
                all_reduce = lambda: None
        self._all_reduce = all_reduce

    def step(self, n: int = 1) -> None:
        """n iterations, enqueued without host synchronisation (converged state is frozen on device)."""
        e = self.engine
        if self._tensor is not None:          # product path: the collective must queue on the kernels' stream
            import torch
            with torch.cuda.stream(self.stream):      # entered once per call, not once per iteration
                for _ in range(n):
                    e.local_update()
                    self._all_reduce()
                    e.apply_consensus()
            return
        for _ in range(n):
            e.local_update()
            self._all_reduce()
            e.apply_consensus()

    def sync(self):
        return self.engine.sync()

    def run(self, max_iters: int, check_every: int = 16):
        """Iterate until the stop test holds (checked every `check_every` iterations) or max_iters."""
        done = 0
        while done < max_iters:
            n = min(check_every, max_iters - done)
            self.step(n)
            done += n
            it, conv = self.sync()
            if conv:
                return it, True
        return self.sync()

    def gather_primal(self):
        """This rank's slice of P, D, C, E with the global index ranges it covers."""
        P, D, C, E = self.engine.get_primal()
        return dict(gen_range=self.shard.meta["gen_range"], sto_range=self.shard.meta["sto_range"],
                    P=P, D=D, C=C, E=E)
