"""The quiet chain (networks: no k_slack launch while no line is flagged; the dual/price kernel forms the node sums, parks the chain when
its dual step flags a line) against the chain that always launches k_slack (DOPF_F_NO_QUIET keeps a context on it): random wide
networks, a few hundred iterations in calls of random length, every array of the C ABI bit for bit after every call.
usage: python scripts/fuzz_quiet.py [n_cases] [seed]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, dopf_pkg
pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth
from helpers import make_engine, state_of
hip = _capi.hip_api()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, used, parked_total = 0, 0, 0
t0 = time.time()
for k in range(n_cases):
    N = int(rng.choice([12, 30, 64, 118, 128, 129, 200, 256]))
    L = int(min(rng.integers(N - 1, 2 * N), 256))
    T = int(rng.choice([24, 48, 73, 96, 168]))
    while max(N, L) * T <= 4096:
        T *= 2
    T = min(T, 192)
    calm = rng.random() < 0.6          # many agents per node, loose limits: the flags go away after the first iterations
    G, S = (int(rng.integers(1500, 4000)), int(rng.integers(100, 400))) if calm else (int(rng.integers(40, 600)), int(rng.integers(4, 60)))
    case = dict(n_gen=G, n_sto=S, T=T, N=N, L=L, seed=int(rng.integers(1, 10**6)),
                fmax_factor=float(rng.choice([1.5, 2.0]) if calm else rng.choice([0.8, 1.0, 1.5])), fmax_min=float(rng.choice([5, 20, 50])))
    try:
        pp = synth.synthetic_case(**case)
    except ValueError:
        continue
    A = G + S
    params = dict(gamma=float(rng.choice([1.0 / A, 0.5 / A, 2.0 / A])), w_flow=float(rng.choice([0.3 / A, 1.0 / A, 0.1 / A, 3.0 / A])),
                  eps=float(rng.choice([0.0, 0.0, 1e-3])), max_iters=int(rng.choice([0, 0, 150])))
    a = make_engine(hip, pp, **params)
    b = make_engine(hip, pp, flags=_capi.F_NO_QUIET, **params)          # the chain the quiet one replaces: k_slack in every iteration
    q = (C.c_int64 * 3)()
    ok, was_quiet = True, 0
    done = 0
    while done < 400 and ok:
        n = int(rng.choice([1, 1, 2, 3, 5, 16, 17, 40]))
        ra, rb = a.iterate(n), b.iterate(n)
        done += n
        hip.lib.dopf_debug_quiet(a._ctx, q)
        was_quiet = max(was_quiet, q[1])
        if ra != rb or a.get_residuals() != b.get_residuals():
            print("MISMATCH (status)", case, params, done, ra, rb, a.get_residuals(), b.get_residuals(), flush=True); ok = False; break
        sa, sb = state_of(a), state_of(b)
        for key in sa:
            if sa[key].size and not np.array_equal(sa[key], sb[key]):           # (the cost too: both chains add it in the same order)
                print("MISMATCH", case, params, done, key, float(np.abs(sa[key] - sb[key]).max()), flush=True); ok = False; break
        if ra[1]:
            break
    hip.lib.dopf_debug_quiet(a._ctx, q)
    if q[0] != 1: print('quiet chain not allowed for', case, params, flush=True)
    bad += 0 if ok else 1
    used += was_quiet
    parked_total += q[2]
    a.close(); b.close()
    if k % 5 == 4:
        print(f"{k+1} cases, bad {bad}, quiet chain used in {used}, parked {parked_total} times, {time.time()-t0:.0f}s", flush=True)
print(f"done: {n_cases} cases, bad {bad}, quiet chain used in {used}, parked {parked_total} times")
