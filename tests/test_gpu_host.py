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
    # the sharded driver runs the three-launch chain (partial rows -> k_reduce -> dual kernel): bit for bit the plain
    # engine's with the same chain; the one-launch form of the plain engine adds the same numbers as integers
    from decentralopf_jl_amd import _capi
    ref = make_engine(hip_api, pp, eps=0.0, gamma=g, flags=_capi.F_NO_TAIL_FUSE)
    ref.iterate(25)
    a, b = state_of(sh.engine), state_of(ref)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    one = make_engine(hip_api, pp, eps=0.0, gamma=g)
    one.iterate(25)
    c = state_of(one)
    for k in a:
        if a[k].size:
            assert np.abs(a[k] - c[k]).max() <= 1e-9 * (1.0 + np.abs(c[k]).max()), k


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


def test_export_results_on_hip_matches_reference_dump(three_node, golden, tmp_path):
    """Every row of the three CSVs written from a HIP run against the reference's own dump (results/TNS_*.csv, kept
    iterations), in the reference's row order (src/helpers/output.jl:14-85: lambda, rho, mue; charge before discharge)."""
    import csv
    nodes, lines, gens, stos, _ = three_node
    admm = pkg.ADMM(0.3, nodes, gens, stos, lines, eps=0.0, max_iters=12, record_slacks=True)
    pkg.run(admm)
    pkg.export_results(admm, "TNS", parent_dir=str(tmp_path) + "/")
    gold = golden["TNS"]["iterations"]
    n_it = 12
    rows = list(csv.reader(open(tmp_path / "TNS_duals.csv")))
    assert rows[0] == ["iteration", "dual", "timestep", "line", "value"] and len(rows) == 1 + n_it * 2 * (1 + 3 + 3)
    order = [(name, i, t, l) for name, nl in (("lambda", 0), ("rho", 3), ("mue", 3)) for i in range(1, n_it + 1)
             for t in (1, 2) for l in (range(1, nl + 1) if nl else [""])]
    checked = 0
    for r, (name, i, t, l) in zip(rows[1:], order):
        assert r[:4] == [str(i), name, str(t), str(l)]
        if str(i) in gold:
            g = gold[str(i)]
            want = g["lam"][t - 1] if name == "lambda" else g["mu" if name == "mue" else "rho"][l - 1][t - 1]
            assert abs(float(r[4]) - want) < 1e-5
            checked += 1
    assert checked >= 10 * 14
    rows = list(csv.reader(open(tmp_path / "TNS_generators.csv")))
    assert rows[0] == ["iteration", "generator", "timestep", "generation"] and len(rows) == 1 + 4 * n_it * 2
    k = 1
    for gi, g in enumerate(gens):
        for i in range(1, n_it + 1):
            for t in (1, 2):
                assert rows[k][:3] == [str(i), g.name, str(t)]
                if str(i) in gold:
                    assert abs(float(rows[k][3]) - gold[str(i)]["P"][gi][t - 1]) < 1e-5
                k += 1
    rows = list(csv.reader(open(tmp_path / "TNS_storages.csv")))
    assert rows[0] == ["iteration", "storage", "timestep", "charge", "discharge"] and len(rows) == 1 + n_it * 2
    k = 1
    for i in range(1, n_it + 1):
        for t in (1, 2):
            assert rows[k][:3] == [str(i), "battery", str(t)]
            if str(i) in gold:
                assert abs(float(rows[k][3]) - gold[str(i)]["C"][t - 1]) < 1e-5 and abs(float(rows[k][4]) - gold[str(i)]["D"][t - 1]) < 1e-5
            k += 1
    # the per-unit diagnostics of ResultGenerator / ResultStorage (results.jl:1-17) came along
    r = admm.results[-1].of(gens[0])
    assert r.U.shape == (3, 2) and r.K.shape == (3, 2) and r.penalty_term.energy_balance.shape == (2,)
    assert len(admm.convergence.lambda_res) == 11 and admm.convergence.mue_res[-1].shape == (3, 2)


def test_bench_line_keeps_the_driver_contract(tmp_path):
    """`python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line with the keys the driver and the judge read
    (metric/value/unit/n_gpus/steps/warmup/ms_per_step/..., roofline, config) — here on the small workload, side runs off."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3",
                        "--workload", "config1", "--no-also", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "agent_subproblem_updates_per_sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 3
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["workload"] == "config1" and "model" not in d["config"]
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert abs(d["value"] - d["config"]["agents_per_gpu"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_node_results_are_the_sums_over_the_units_of_each_node(hip_api, three_node):
    """dopf_get_node_results = ResultNode.{generation, discharge, charge} (src/structures/results.jl:19-35, filled by `update`
    in the agent loop of Result(...)): on the shipped case and on a seeded network with empty nodes, against numpy sums of
    the primal arrays; injection = generation + discharge - charge - demand; the host's Result carries them per node."""
    from helpers import make_engine
    for pp, kw in ((three_node[4], dict()),
                   (synth.synthetic_case(25, 6, 8, N=30, L=50, seed=17, fmax_factor=0.6, fmax_min=5), dict(gamma=0.05)),
                   (synth.synthetic_case(3000, 300, 24, seed=31), dict(gamma=1.0 / 3300))):
        e = make_engine(hip_api, pp, eps=0.0, **kw)
        e.iterate(30)
        P, D, C, _E = e.get_primal()
        g, d, c = e.get_node_results()
        want = [np.zeros((pp.N, pp.T)) for _ in range(3)]
        np.add.at(want[0], np.asarray(pp.gen_node, dtype=np.int64), P)
        np.add.at(want[1], np.asarray(pp.sto_node, dtype=np.int64), D)
        np.add.at(want[2], np.asarray(pp.sto_node, dtype=np.int64), C)
        for a, b in zip((g, d, c), want):
            assert a.shape == (pp.N, pp.T) and np.abs(a - b).max() <= 1e-9 * (1.0 + np.abs(b).max())
        inj = e.get_consensus()[0]
        assert np.abs(g + d - c - np.asarray(pp.demand).reshape(pp.N, pp.T) - inj).max() <= 1e-8 * (1.0 + np.abs(inj).max())
    nodes, lines, gens, stos = three_node[:4]
    admm = pkg.ADMM(0.3, nodes, gens, stos, lines)
    pkg.calculate_iteration(admm)
    r = admm.results[-1]
    assert np.allclose(sum(r.of_node(n).generation for n in nodes), r.generation)
    assert np.allclose(r.of_node(stos[0].node).discharge, r.of(stos[0]).discharge)
