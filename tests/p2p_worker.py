"""Started by tests/test_gpu_multi.py (GPU_MAX_HW_QUEUES=8, nothing else alive in the process): the peer exchange of
dopf_multi_* with several shards on ONE device. See the test's docstring."""
import ctypes as C
import gc
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dopf_pkg  # noqa: E402

dopf_pkg.load()
from decentralopf_jl_amd import _capi, network, synth  # noqa: E402
from helpers import make_engine, state_of  # noqa: E402

hip = _capi.hip_api()
CASES = {
    "network": dict(n_gen=300, n_sto=40, T=24, N=3, L=3, seed=4, fmax_factor=0.8, fmax_min=5),
    "copper plate": dict(n_gen=3000, n_sto=400, T=24, seed=4),
    "copper plate T96": dict(n_gen=700, n_sto=90, T=96, seed=5),
}
KEYS = ("lam", "mu", "rho", "inj", "avg_U", "avg_K", "flow", "cost")


def quiet_state(engine):
    q = (C.c_int64 * 3)()
    assert hip.lib.dopf_debug_quiet(engine._ctx, q) == 0
    return int(q[0]), int(q[1]), int(q[2])          # the chain without k_slack / k_reduce: allowed, in use for the next call, times parked


def compare(pp, n, steps, kw, tol, xflags=0, quiet_seen=None):
    ref = make_engine(hip, pp, **kw)
    wants = []
    for k in steps:
        ref.iterate(k)
        wants.append(state_of(ref))
    ref.close()
    gc.collect()
    m = _capi.MultiEngine(hip, n, params=_capi.default_params(flags=_capi.F_COMM_P2P | xflags, **kw), devices=[0] * n, **pp.engine_kwargs())
    assert m.shard(0).comm_info()[0] == n
    for k, want in zip(steps, wants):
        assert m.iterate(k) == (k, False)
        for a, b in zip(m.get_primal(), (want["P"], want["D"], want["C"], want["E"])):
            assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())
        first = state_of(m.shard(0))
        for i in range(n):
            got = state_of(m.shard(i))
            for key in KEYS:
                if want[key].size:
                    assert np.abs(got[key] - want[key]).max() <= tol * max(1.0, np.abs(want[key]).max()), (key, i, k)
                    assert np.array_equal(got[key], first[key]), (key, i)        # rank-order sums: bitwise the same
        if quiet_seen is not None:
            qs = [quiet_state(m.shard(i)) for i in range(n)]
            assert all(q == qs[0] for q in qs), qs                               # every shard takes the same decision
            quiet_seen.append(qs[0])
    m.close()
    gc.collect()


for name, case in CASES.items():
    kw = dict(case)
    pp = synth.synthetic_case(kw.pop("n_gen"), kw.pop("n_sto"), kw.pop("T"), **kw)
    # (the literal flow weight makes the network case a 2-cycle that amplifies rounding: few iterations)
    g = 0.01 if name == "network" else 1.0 / (pp.G + pp.S)
    for n in (2, 3):
        compare(pp, n, (1, 4, 7) if name == "network" else (1, 7, 30), dict(eps=0.0, gamma=g), 1e-9)
        print("equal", name, n, flush=True)

# copper plates above ran the exchange inside the tail block of the one-launch iteration; the older place —
# inside the one-block dual kernel of the three-launch chain — stays reachable and tested
kwc = dict(CASES["copper plate T96"])
ppc = synth.synthetic_case(kwc.pop("n_gen"), kwc.pop("n_sto"), kwc.pop("T"), **kwc)
compare(ppc, 2, (1, 7, 30), dict(eps=0.0, gamma=1.0 / (ppc.G + ppc.S)), 1e-9, xflags=_capi.F_NO_TAIL_XCHG)
print("three-launch exchange ok", flush=True)

# a consensus vector of several chunks (30 nodes, 50 lines, 24 steps: 3 121 doubles), three shards
pp = synth.synthetic_case(240, 30, 24, N=30, L=50, seed=41, fmax_factor=0.7, fmax_min=5)
A = pp.G + pp.S
assert pp.N * pp.T + 2 * pp.L * pp.T + 1 > 2048
compare(pp, 3, (1, 5, 21), dict(eps=0.0, gamma=1.0 / A, w_flow=0.3 / A), 1e-8)
print("chunks ok", flush=True)

# the reduce-scatter + all-gather form of the exchange (every chunk has an owner that adds the ranks' copies and hands the
# sum to everybody): forced on the cases above, and chosen by the library itself on a vector of more than one chunk per
# shard (40 nodes, 60 lines, 48 steps: 7 681 doubles = 4 chunks, three shards)
compare(pp, 3, (1, 5, 21), dict(eps=0.0, gamma=1.0 / A, w_flow=0.3 / A), 1e-8, xflags=_capi.F_XCHG_OWNER)
kwn = dict(CASES["network"])
ppn = synth.synthetic_case(kwn.pop("n_gen"), kwn.pop("n_sto"), kwn.pop("T"), **kwn)
for n in (2, 3):
    compare(ppn, n, (1, 4, 7), dict(eps=0.0, gamma=0.01), 1e-9, xflags=_capi.F_XCHG_OWNER)
pp4 = synth.synthetic_case(320, 40, 48, N=40, L=60, seed=43, fmax_factor=0.7, fmax_min=5)
A4 = pp4.G + pp4.S
assert (pp4.N * pp4.T + 2 * pp4.L * pp4.T + 1 + 2047) // 2048 > 3
compare(pp4, 3, (1, 5, 15), dict(eps=0.0, gamma=1.0 / A4, w_flow=0.3 / A4), 1e-8)
print("reduce-scatter ok", flush=True)

# Networks whose dual step is the one-launch kernel (here 118 nodes / 186 lines, 2 % of configs[3]'s agents): while no line is
# flagged a shard's chain on the exchange is k_net_agents, k_slack, the exchange of the NODE SUMS, the dual/price kernel — which
# forms the slack sums behind the exchange from the summed injections' changes (no k_reduce launch; DevView::slackGlobal). A
# dual step that flags a line parks the chain on every shard alike and the host goes back to the chain with k_reduce.
ppq = synth.baseline_config(3, scale=0.02)
Aq = ppq.G + ppq.S
for n, xf in ((2, 0), (3, 0), (3, _capi.F_XCHG_ALLGATHER)):
    seen = []
    # (on this small share lines get flagged again now and then: no flag after iterations 22-28, 45-54 and 166-196 — the calls end
    # inside those windows, so the next call starts on the chain without k_reduce and is parked by the first new flag)
    compare(ppq, n, (1, 4, 19, 26, 130, 40), dict(eps=0.0, gamma=1.0 / Aq, w_flow=0.3 / Aq), 1e-9, xflags=xf, quiet_seen=seen)
    assert all(s[0] == 1 for s in seen) and [s[1] for s in seen] == [0, 0, 1, 1, 1, 0] and seen[-1][2] >= 2, seen     # allowed, ran, was parked
    print("network chain without k_reduce on the exchange", n, "shards: in use after calls", [s[1] for s in seen], "parked", seen[-1][2], flush=True)
# (DOPF_F_NO_QUIET keeps a context on the chain with k_reduce)
seen = []
compare(ppq, 2, (1, 4, 60), dict(eps=0.0, gamma=1.0 / Aq, w_flow=0.3 / Aq), 1e-9, xflags=_capi.F_NO_QUIET, quiet_seen=seen)
assert all(s[0] == 0 and s[1] == 0 for s in seen), seen
print("comm-quiet ok", flush=True)

# the reference's shipped case: stops like check_convergence! on both shards
nodes, lines, gens, stos = network.three_node_case()
pp3 = network.pack(nodes, gens, stos, lines)
m = _capi.MultiEngine(hip, 2, params=_capi.default_params(flags=_capi.F_COMM_P2P), devices=[0, 0], **pp3.engine_kwargs())
done, conv = m.iterate(2000)
assert conv and done == 476 and m.shard(1).get_residuals()[3] == 476
assert m.iterate(5) == (0, True)
assert abs(m.shard(0).get_consensus()[4] - 14034.5056) < 1e-3
m.close()
gc.collect()
print("stop ok", flush=True)

# a rank whose peer never sends must not hang: the wait is bounded (wall clock), the kernel ends, the call returns an error
os.environ["DOPF_XCHG_TIMEOUT_MS"] = "300"
kw = dict(CASES["copper plate"])
pp = synth.synthetic_case(kw.pop("n_gen"), kw.pop("n_sto"), kw.pop("T"), **kw)
m = _capi.MultiEngine(hip, 2, params=_capi.default_params(eps=0.0, gamma=1e-3, flags=_capi.F_COMM_P2P), devices=[0, 0], **pp.engine_kwargs())
lonely = m.shard(1)
for n_it in (3, 40):                     # sticky, and later exchanges do not wait again
    try:
        lonely.iterate(n_it)
        raise SystemExit("a lonely shard iterated without its peer")
    except _capi.DopfError as e:
        assert "did not arrive" in str(e), e
m.close()
print("missing peer ok", flush=True)
print("p2p worker: ok")
