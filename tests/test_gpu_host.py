"""The host mirror and the sharded driver on the real HIP backend (world size 1)."""
import numpy as np
import pytest

import decentralopf_jl_amd as pkg
from decentralopf_jl_amd import synth

pytestmark = pytest.mark.gpu


def test_admm_driver_on_hip(three_node, thesis, tmp_path):
    nodes, lines, gens, stos, _ = three_node
    admm = pkg.ADMM(0.3, nodes, gens, stos, lines)              # default backend = libdopf_hip
    pkg.run(admm)
    assert admm.convergence.all and admm.iteration == 476 and len(admm.results) == 476
    assert np.abs(pkg.get_nodal_price(admm, admm.iteration) - np.asarray(thesis["admm"]["nodal_price"])).max() < 6e-5
    assert abs(admm.results[-1].total_costs - 14034.5056) < 1e-3
    pkg.export_results(admm, "TNS", parent_dir=str(tmp_path) + "/")
    assert (tmp_path / "TNS_duals.csv").read_text().splitlines()[1] == "1,lambda,1,,0.0"
    fast = pkg.run(pkg.ADMM(0.3, nodes, gens, stos, lines, record=False))
    assert fast.iteration == 476 and abs(fast.results[0].total_costs - 14034.5056) < 1e-3


def test_sharded_driver_world1_on_hip(hip_api):
    import torch
    from helpers import make_engine, state_of
    pp = synth.synthetic_case(500, 50, 24, seed=3)
    g = 1.0 / 550
    sh = pkg.ShardedADMM(pp, 0, 1, eps=0.0, gamma=g)
    sh.step(25)
    assert sh.sync() == (26, False)
    ref = make_engine(hip_api, pp, eps=0.0, gamma=g)
    ref.iterate(25)
    a, b = state_of(sh.engine), state_of(ref)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_sharded_driver_with_rccl_allreduce_world1(hip_api):
    """A one-rank nccl (= RCCL) process group: the all-reduce really runs between dopf_local_update and
    dopf_apply_consensus on the engine's stream; results must equal the plain engine bit for bit."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from helpers import make_engine, state_of
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        pp = synth.synthetic_case(400, 40, 24, N=3, L=3, seed=6, fmax_factor=0.8, fmax_min=5)
        g = 0.01
        sh = pkg.ShardedADMM(pp, 0, 1, eps=0.0, gamma=g)
        tens, st = sh._tensor, sh.stream

        def all_reduce():
            with torch.cuda.stream(st):
                dist.all_reduce(tens, op=dist.ReduceOp.SUM)
        sh._all_reduce = all_reduce
        sh.step(20)
        assert sh.sync() == (21, False)
        ref = make_engine(hip_api, pp, eps=0.0, gamma=g)
        ref.iterate(20)
        a, b = state_of(sh.engine), state_of(ref)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    finally:
        dist.destroy_process_group()
