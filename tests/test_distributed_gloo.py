"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups driving ShardedADMM.
Compute stand-in = the oracle library (tests only); what is under test is the sharding, the
consensus-buffer all-reduce and the replicated dual update of decentralopf.jl_amd/sharded.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ORACLE_LIB, ROOT, build_oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, iters, out_dir):
    sys.path.insert(0, ROOT)
    import dopf_pkg
    dopf_pkg.load()
    from decentralopf_jl_amd import _capi, synth
    from decentralopf_jl_amd.sharded import ShardedADMM, host_consensus_view
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pp = synth.synthetic_case(**case)
    from oracle.binding import OracleApi
    api = OracleApi(ORACLE_LIB)
    holder = {}

    def all_reduce():
        dist.all_reduce(holder["t"], op=dist.ReduceOp.SUM)

    sh = ShardedADMM(pp, rank, world, api=api, all_reduce=all_reduce, mode=1, eps=0.0, gamma=0.05)
    holder["t"] = torch.from_numpy(host_consensus_view(sh.engine))      # shares memory with the C buffer
    sh.step(iters)
    it, conv = sh.sync()
    lam, mu, rho = sh.engine.get_duals()
    prim = sh.gather_primal()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lam=lam, mu=mu, rho=rho, P=prim["P"], D=prim["D"],
             C=prim["C"], gr=np.asarray(prim["gen_range"]), sr=np.asarray(prim["sto_range"]), it=it,
             inj=sh.engine.get_consensus()[0])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single(oracle_api, tmp_path, world):
    build_oracle()
    case = dict(n_gen=26, n_sto=7, T=6, N=3, L=3, seed=8, fmax_factor=0.8, fmax_min=5)
    iters = 10
    mp.spawn(_worker, args=(world, _free_port(), case, iters, str(tmp_path)), nprocs=world, join=True)
    from decentralopf_jl_amd import synth
    from helpers import make_engine, state_of
    pp = synth.synthetic_case(**case)
    ref = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=0.05)
    ref.iterate(iters)
    want = state_of(ref)
    P = np.zeros_like(want["P"])
    D = np.zeros_like(want["D"])
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert int(z["it"]) == iters + 1
        for k in ("lam", "mu", "rho", "inj"):
            assert np.abs(z[k] - want[k]).max() < 1e-9, (r, k)          # replicated state identical on all ranks
        P[z["gr"][0]:z["gr"][1]] = z["P"]
        D[z["sr"][0]:z["sr"][1]] = z["D"]
    assert np.abs(P - want["P"]).max() < 1e-9 and np.abs(D - want["D"]).max() < 1e-9
