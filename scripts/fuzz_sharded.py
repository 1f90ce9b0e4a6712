"""Random cases split over 2-4 HIP contexts on the one GPU (consensus buffers summed by hand, as the all-reduce would),
against a single context: the multi-rank arithmetic without RCCL.  usage: python scripts/fuzz_sharded.py [n] [seed]"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
from helpers import make_engine, state_of
hip = _capi.CApi(os.environ["DOPF_LIB"], "dopf_") if os.environ.get("DOPF_LIB") else _capi.hip_api()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
start = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # (replay: draw the first cases without running them)
verbose = len(sys.argv) > 4
worst_all, bad = 0.0, 0
t0 = time.time()
for k in range(n_cases):
    net = rng.random() < 0.5
    N = int(rng.integers(2, 9)) if net else int(rng.choice([1, 1, 4]))
    L = int(rng.integers(N - 1, 2 * N + 1)) if net else 0
    T = int(rng.choice([3, 8, 12, 24, 48, 96]))
    G, S = int(rng.integers(4, 300)), int(rng.integers(0, 60))
    case = dict(n_gen=G, n_sto=S, T=T, N=N, L=L, seed=int(rng.integers(1, 10**6)))
    if net:
        case.update(fmax_factor=float(rng.choice([0.8, 1.0, 1.5])), fmax_min=float(rng.choice([5, 20])))
    try:
        pp = synth.synthetic_case(**case)
    except ValueError:
        continue
    A = G + S
    K = int(rng.integers(2, 5))
    params = dict(gamma=float(rng.choice([1.0, 0.3])) / A, w_flow=(0.1 / A if net else 10.0), eps=0.0)
    iters = int(rng.integers(5, 40))
    if k < start:
        continue
    if verbose:
        print(k, case, 'shards', K, params, 'iters', iters, flush=True)
    ref = make_engine(hip, pp, **params)
    engs = [make_engine(hip, pp.shard(r, K), n_agents_global=A, **params) for r in range(K)]
    bufs = [torch.zeros(e.consensus_size(), dtype=torch.float64, device="cuda") for e in engs]
    for e, b in zip(engs, bufs):
        e.bind_consensus(b.data_ptr())
    for it_ in range(iters):
        if verbose: print('  it', it_, flush=True)
        for e in engs: e.local_update()
        for e in engs: e.sync()
        total = sum(bufs[1:], bufs[0].clone())
        for b in bufs: b.copy_(total)
        torch.cuda.synchronize()
        for e in engs: e.apply_consensus()
        for e in engs: e.sync()
    if verbose: print('  loop done', flush=True)
    ref.iterate(iters)
    if verbose: print('  ref done', flush=True)
    want = state_of(ref)
    if verbose: print('  ref state', flush=True)
    got = []
    for e in engs:
        got.append(state_of(e))
        if verbose: print('  shard state', len(got), flush=True)
    w = 0.0
    for key in ("lam", "mu", "rho", "inj", "flow"):
        if want[key].size:
            for g in got:
                w = max(w, float(np.abs(g[key] - want[key]).max()) / max(1.0, float(np.abs(want[key]).max())))
    if G:
        w = max(w, float(np.abs(np.concatenate([g["P"] for g in got]) - want["P"]).max()) / max(1.0, float(np.abs(want["P"]).max())))
    if S:
        w = max(w, float(np.abs(np.concatenate([g["E"] for g in got]) - want["E"]).max()) / max(1.0, float(np.abs(want["E"]).max())))
    worst_all = max(worst_all, w)
    if w > 1e-7:
        bad += 1
        print("MISMATCH", case, "shards", K, params, w, flush=True)
    for e in engs + [ref]: e.close()
    if k % 20 == 19:
        print(f"{k+1} cases, worst relative difference {worst_all:.2e}, bad {bad}, {time.time()-t0:.0f}s", flush=True)
print(f"done: {n_cases} cases, worst {worst_all:.2e}, bad {bad}")
