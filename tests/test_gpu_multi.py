"""The consensus sum across GPUs inside the library: dopf_comm_* (one process per GPU) and dopf_multi_* (one process,
n GPUs). On the one-GPU box RCCL runs at world size 1 and the sharding logic of dopf_multi_* runs over the host-sum
debugging transport (several shards on one device); the two-rank RCCL test needs two GPUs and skips otherwise."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before the library loads RCCL: a process must not end up with two copies, PyTorch ships its own)

from decentralopf_jl_amd import _capi, synth
from helpers import make_engine, max_diff, state_of

pytestmark = pytest.mark.gpu

CASES = {
    "network": dict(n_gen=300, n_sto=40, T=24, N=3, L=3, seed=4, fmax_factor=0.8, fmax_min=5),
    "copper plate": dict(n_gen=3000, n_sto=400, T=24, seed=4),
    "copper plate T96": dict(n_gen=700, n_sto=90, T=96, seed=5),
}


def _case(name):
    kw = dict(CASES[name])
    return synth.synthetic_case(kw.pop("n_gen"), kw.pop("n_sto"), kw.pop("T"), **kw)


@pytest.mark.parametrize("flags", [0, _capi.F_NO_GRAPH], ids=["graph", "eager"])
def test_comm_world1_rccl_in_the_graph(hip_api, flags):
    """A one-rank RCCL communicator owned by the library: the all-reduce sits between the local sums and the dual step
    of every iteration, inside the captured graph; results equal the plain engine's bit for bit."""
    pp = _case("network")
    ref = make_engine(hip_api, pp, eps=0.0, gamma=0.01)
    ref.iterate(40)
    e = make_engine(hip_api, pp, eps=0.0, gamma=0.01, flags=flags)
    e.comm_init(1, 0, e.comm_unique_id())
    assert e.iterate(40) == (40, False)
    world, rank, in_graph = e.comm_info()
    assert (world, rank) == (1, 0) and in_graph == (flags == 0)
    a, b = state_of(e), state_of(ref)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    with pytest.raises(_capi.DopfError):
        e.comm_init(1, 0, e.comm_unique_id())          # one communicator per context


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("n", [1, 2, 3])
def test_multi_shards_equal_one_context(hip_api, name, n):
    """dopf_multi_*: the library cuts the agent lists into n shards and sums their consensus buffers every iteration
    (host transport here: all shards on the one GPU); trajectory = the single context's to rounding."""
    pp = _case(name)
    # (the literal flow weight makes the network case a 2-cycle that amplifies rounding: few iterations, as in
    # test_hip_sharded_contexts_equal_one)
    g = 0.01 if name == "network" else 1.0 / (pp.G + pp.S)
    ref = make_engine(hip_api, pp, eps=0.0, gamma=g)
    m = _capi.MultiEngine(hip_api, n, params=_capi.default_params(eps=0.0, gamma=g, flags=_capi.F_COMM_HOST if n > 1 else 0),
                          **pp.engine_kwargs())
    assert m.n == n
    for k in ((1, 4, 7) if name == "network" else (1, 7, 30)):
        ref.iterate(k)
        assert m.iterate(k) == (k, False)
        want = state_of(ref)
        P, D, C, E = m.get_primal()
        for a, b in zip((P, D, C, E), (want["P"], want["D"], want["C"], want["E"])):
            assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
        for i in range(n):
            got = state_of(m.shard(i))
            for key in ("lam", "mu", "rho", "inj", "avg_U", "avg_K", "flow", "cost"):
                if want[key].size:
                    assert np.abs(got[key] - want[key]).max() <= 1e-9 * max(1.0, np.abs(want[key]).max()), (key, i)
    m.close()


def test_multi_stops_like_check_convergence(hip_api, three_node):
    m = _capi.MultiEngine(hip_api, 2, params=_capi.default_params(flags=_capi.F_COMM_HOST), **three_node[4].engine_kwargs())
    done, conv = m.iterate(2000)
    assert conv and done == 476 and m.shard(0).get_residuals()[3] == 476 and m.shard(1).get_residuals()[3] == 476
    assert m.iterate(5) == (0, True)
    assert abs(m.shard(0).get_consensus()[4] - 14034.5056) < 1e-3


def test_multi_rejects_what_it_cannot_do(hip_api, three_node):
    import torch
    kw = three_node[4].engine_kwargs()
    ndev = torch.cuda.device_count()
    with pytest.raises(_capi.DopfError):          # RCCL transport: one distinct device per shard
        _capi.MultiEngine(hip_api, ndev + 1, **kw)
    with pytest.raises(_capi.DopfError):
        _capi.MultiEngine(hip_api, 2, devices=[0, 0], **kw)
    with pytest.raises(_capi.DopfError):
        _capi.MultiEngine(hip_api, 0, **kw)


def test_solver_failure_surfaces_as_an_error(hip_api):
    """A storage whose root search hits its iteration cap must not pass silently (DOPF_E_SOLVER): forced here by the
    debug flag that lowers the scan kernel's cap to 2."""
    pp = synth.synthetic_case(60, 30, 24, seed=8)
    e = make_engine(hip_api, pp, eps=0.0, gamma=0.02, flags=_capi.F_NO_WARM_START | _capi.F_DEBUG_ROOT_CAP)
    with pytest.raises(_capi.DopfError, match="tolerance"):
        for _ in range(30):
            e.iterate(1)
    assert e.solver_failures() > 0
    e.iterate(0)                                   # reported once per occurrence: no new failure, no error
    ok = make_engine(hip_api, pp, eps=0.0, gamma=0.02, flags=_capi.F_NO_WARM_START)
    ok.iterate(30)
    assert ok.solver_failures() == 0


def test_agent_slacks_penalties_and_residual_vectors(hip_api, oracle_api):
    """ResultGenerator/ResultStorage.{U, K, penalty_term} and Convergence.*_res, recomputed on request."""
    pp = synth.synthetic_case(40, 8, 6, N=3, L=3, seed=21, fmax_factor=0.6, fmax_min=5)
    h = make_engine(hip_api, pp, eps=0.0, gamma=0.05, flags=_capi.F_KEEP_DELTAS)
    o = make_engine(oracle_api, pp, mode=1, eps=0.0, gamma=0.05)
    dp = _capi.c_double_p
    with pytest.raises(_capi.DopfError):          # without the flag the device does not keep every agent's change
        make_engine(hip_api, pp, eps=0.0, gamma=0.05).get_agent_slacks(0)
    for it in range(6):
        before = state_of(h)
        lam0, mu0, rho0 = h.get_duals()
        h.iterate(1)
        o.iterate(1)
        after = state_of(h)
        for a in (0, 7, pp.G - 1, pp.G, pp.G + pp.S - 1):
            U, K = h.get_agent_slacks(a)
            Uo, Ko = np.zeros(pp.L * pp.T), np.zeros(pp.L * pp.T)
            assert oracle_api.get_agent_slacks(o._ctx, a, Uo.ctypes.data_as(dp), Ko.ctypes.data_as(dp)) == 0
            assert np.abs(U - Uo.reshape(pp.T, pp.L).T).max() < 1e-8 and np.abs(K - Ko.reshape(pp.T, pp.L).T).max() < 1e-8
            # penalty terms from their definition (penalty_terms.jl:3-37) with the consensus state the solve read
            d = (after["P"][a] - before["P"][a]) if a < pp.G else \
                ((after["D"] - after["C"])[a - pp.G] - (before["D"] - before["C"])[a - pp.G])
            h_col = pp.ptdf[:, (pp.gen_node[a] if a < pp.G else pp.sto_node[a - pp.G])]
            fl = before["flow"] + np.outer(h_col, d)
            eb, up, lo = h.get_agent_penalty(a)
            assert np.abs(eb - (before["inj"].sum(axis=0) + d) ** 2).max() < 1e-6 * max(1.0, eb.max())
            assert np.abs(up - ((fl + U - pp.f_max[:, None]) ** 2).sum(axis=0)).max() < 1e-6 * max(1.0, up.max())
            assert np.abs(lo - ((K - fl - pp.f_max[:, None]) ** 2).sum(axis=0)).max() < 1e-6 * max(1.0, lo.max())
            eb2, _, _ = h.get_agent_penalty(a, delta=d)         # same with the change passed in
            assert np.abs(eb2 - eb).max() < 1e-9 * max(1.0, eb.max())
        # Result.penalty_term (results.jl:66-70): the sum over ALL agents in one device pass = the per-agent terms added up
        eb_s, up_s, lo_s = h.get_penalty_sums()
        acc = np.zeros((3, pp.T))
        for a in range(pp.G + pp.S):
            acc += np.asarray(h.get_agent_penalty(a))
        for got, want in zip((eb_s, up_s, lo_s), acc):
            assert np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max())
        lam1, mu1, rho1 = h.get_duals()
        rl, rm, rr = h.get_residual_vectors()
        assert np.array_equal(rl, np.abs(lam1 - lam0)) and np.array_equal(rm, np.abs(mu1 - mu0)) and np.array_equal(rr, np.abs(rho1 - rho0))
        res = h.get_residuals()
        assert res[0] == rl.max() and res[1] == rm.max() and res[2] == rr.max() or it == 0
    # copper plate: no slacks, the penalty needs the change from the caller
    cp = synth.synthetic_case(10, 3, 4, seed=2)
    e = make_engine(hip_api, cp, eps=0.0, gamma=0.05)
    p0 = e.get_primal()[0]
    s0 = e.get_consensus()[0].sum(axis=0)
    e.iterate(1)
    d = e.get_primal()[0][2] - p0[2]
    assert np.allclose(e.get_agent_penalty(2, delta=d)[0], (s0 + d) ** 2)
    with pytest.raises(_capi.DopfError):
        e.get_agent_penalty(2)
    with pytest.raises(_capi.DopfError):          # no lines: the device does not keep the agents' changes
        e.get_penalty_sums()


def test_two_ranks_two_gpus_rccl(hip_api, tmp_path):
    """One process per GPU, the library's own communicator: needs two devices (skips on the one-GPU box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank_worker.py")
    idf = str(tmp_path / "uid.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", idf, str(tmp_path / f"out{r}.npz"), "25"], env=env)
             for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    pp = synth.synthetic_case(600, 80, 24, N=3, L=3, seed=11, fmax_factor=0.8, fmax_min=5)
    ref = make_engine(hip_api, pp, eps=0.0, gamma=0.01)
    ref.iterate(25)
    want = state_of(ref)
    outs = [np.load(str(tmp_path / f"out{r}.npz")) for r in range(2)]
    for key in ("lam", "mu", "rho", "inj", "avg_U", "avg_K", "flow", "cost"):
        for o in outs:
            assert np.abs(o[key] - want[key]).max() <= 1e-9 * max(1.0, np.abs(want[key]).max()), key
        assert np.array_equal(outs[0][key], outs[1][key]), key          # replicated state: identical on both ranks
    assert np.abs(np.concatenate([o["P"] for o in outs]) - want["P"]).max() < 1e-9
    assert outs[0]["comm"][0] == 2


@pytest.mark.parametrize("transport", ["rccl", "p2p"])
def test_multi_network_on_two_devices(hip_api, transport):
    """dopf_multi_* with one shard per DEVICE on a network whose dual/price kernel needs the raised dynamic-LDS limit
    (118 nodes / 186 lines: ~90 KB): the limit is a per-device function attribute — raised once per process it was missing on
    every device but the first (advisor, round 3). Needs two devices (skips on the one-GPU box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    pp = synth.baseline_config(3, scale=0.02)
    A = pp.G + pp.S
    kw = dict(eps=0.0, gamma=1.0 / A, w_flow=0.3 / A)
    ref = make_engine(hip_api, pp, **kw)
    ref.iterate(12)
    want = state_of(ref)
    m = _capi.MultiEngine(hip_api, 2, params=_capi.default_params(flags=_capi.F_COMM_P2P if transport == "p2p" else 0, **kw),
                          devices=[0, 1], **pp.engine_kwargs())
    assert m.iterate(12) == (12, False)
    for i in range(2):
        got = state_of(m.shard(i))
        for key in ("lam", "mu", "rho", "inj", "avg_U", "avg_K", "flow"):
            if want[key].size:
                assert np.abs(got[key] - want[key]).max() <= 1e-8 * max(1.0, np.abs(want[key]).max()), (key, i)
    m.close()


def test_peer_exchange_shards_on_one_device(hip_api):
    """DOPF_F_COMM_P2P through dopf_multi_*: the consensus sum by the exchange kernels (every shard stores its vector into
    every shard's receive area, flags, sum in rank order), inside the iteration graphs, each shard on its own stream and
    host thread — here with all shards on the ONE GPU. tests/p2p_worker.py runs, in a process of its own: 2 and 3 shards
    against a single context on a network and two copper plates (trajectory to rounding, replicated state bitwise equal
    on all shards), a consensus vector of several 2048-double chunks, the stop at iteration 476, a peer that never sends.
    Its own process because shards that share one device need a hardware queue each — a waiting exchange kernel must not
    sit in the queue in front of the peer it waits for — so the child runs with GPU_MAX_HW_QUEUES=8 and nothing else
    alive; one shard per device, the real thing, cannot run into that."""
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "p2p_worker.py")
    r = subprocess.run([sys.executable, worker], env=dict(os.environ, GPU_MAX_HW_QUEUES="8", DOPF_XCHG_TIMEOUT_MS="4000"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "p2p worker: ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    for line in ("equal network 2", "equal network 3", "equal copper plate 3", "equal copper plate T96 2", "three-launch exchange ok", "chunks ok", "reduce-scatter ok", "comm-quiet ok", "stop ok", "missing peer ok"):
        assert line in r.stdout, (line, r.stdout[-1500:])


@pytest.mark.parametrize("world,form", [(2, "all-gather"), (3, "all-gather"), (3, "reduce-scatter")])
def test_ranks_in_separate_processes_peer_exchange(hip_api, tmp_path, world, form):
    """One process per rank, receive areas shared through hipIpc handles — the form a launcher-started multi-GPU run
    uses. With fewer GPUs than ranks the ranks share device 0 (the exchange does not care where a peer's memory lives).
    reduce-scatter: the owner form of the exchange (k_xchg_rs), forced on this one-chunk vector."""
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank_worker.py")
    idf = str(tmp_path / "handle")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DOPF_XCHG_TIMEOUT_MS="20000")
    xflags = str(_capi.F_XCHG_OWNER if form == "reduce-scatter" else 0)
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), idf, str(tmp_path / f"out{r}.npz"), "25", "xchg", xflags], env=env)
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    pp = synth.synthetic_case(600, 80, 24, N=3, L=3, seed=11, fmax_factor=0.8, fmax_min=5)
    ref = make_engine(hip_api, pp, eps=0.0, gamma=0.01)
    ref.iterate(25)
    want = state_of(ref)
    outs = [np.load(str(tmp_path / f"out{r}.npz")) for r in range(world)]
    for key in ("lam", "mu", "rho", "inj", "avg_U", "avg_K", "flow", "cost"):
        for o in outs:
            assert np.abs(o[key] - want[key]).max() <= 1e-9 * max(1.0, np.abs(want[key]).max()), key
            assert np.array_equal(outs[0][key], o[key]), key
    assert np.abs(np.concatenate([o["P"] for o in outs]) - want["P"]).max() < 1e-9
    assert outs[0]["comm"][0] == world


