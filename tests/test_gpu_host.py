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
