"""decentralopf.jl_amd — MI355X-native ADMM consensus-OPF inner loop (hot path of
rockstaedt/DecentralOPF.jl) behind the C ABI of include/dopf.h.

The directory name contains a dot, so it is loaded through ``dopf_pkg.load()`` (repo root) under
the module name ``decentralopf_jl_amd``.
"""
from . import _capi, admm, central, network, sharded, synth
from .central import CentralResult, central_reference, central_reference_on_device, solve_central_packed
from .admm import ADMM, calculate_iteration, export_results, get_nodal_price, run
from .sharded import ShardedADMM
from ._capi import DopfError, Engine, default_params, hip_api
from .network import (Generator, Line, Node, PackedProblem, Storage, calculate_ptdf, pack,
                      three_node_case)

__all__ = ["DopfError", "Engine", "default_params", "hip_api", "Generator", "Line", "Node",
           "PackedProblem", "Storage", "calculate_ptdf", "pack", "three_node_case", "_capi",
           "network", "synth", "admm", "sharded", "ADMM", "calculate_iteration",
           "export_results", "get_nodal_price", "run", "ShardedADMM", "central", "CentralResult",
           "central_reference", "central_reference_on_device", "solve_central_packed"]
