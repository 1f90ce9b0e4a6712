"""Started by tests/test_gpu_parity.py with DOPF_GUARD=1: every device array of the library then ends on the last byte of
its own mapping, so that a read or write past an array's end is a GPU memory fault (process dies) instead of a silent
access to a neighbour. Ragged shapes on purpose: storage counts that do not fill a lane group, nodes without units,
one line, shards."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dopf_pkg  # noqa: E402

dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth  # noqa: E402
from helpers import make_engine  # noqa: E402

assert os.environ.get("DOPF_GUARD")
hip = _capi.hip_api()
cases = [
    dict(n_gen=287, n_sto=57, T=24, N=2, L=1, seed=653544, fmax_factor=0.8, fmax_min=20.0),      # found by scripts/fuzz_sharded.py
    dict(n_gen=7, n_sto=41, T=12, N=6, L=5, seed=489871, fmax_factor=0.8, fmax_min=5.0),
    dict(n_gen=145, n_sto=53, T=3, N=5, L=6, seed=529681, fmax_factor=0.8, fmax_min=20.0),
    dict(n_gen=138, n_sto=48, T=96, N=1, L=0, seed=95602),
    dict(n_gen=1000, n_sto=3, T=24, N=1, L=0, seed=3),
    dict(n_gen=5, n_sto=70, T=48, N=4, L=0, seed=4),
    dict(n_gen=333, n_sto=1, T=168, N=3, L=3, seed=5, fmax_factor=1.0, fmax_min=5.0),
]
for case in cases:
    pp = synth.synthetic_case(**case)
    A = pp.G + pp.S
    kw = dict(gamma=1.0 / A, eps=0.0)
    if pp.L:
        kw["w_flow"] = 0.3 / A
    for K in (1, 3):
        engs = [make_engine(hip, pp.shard(r, K), n_agents_global=A, **kw) for r in range(K)]
        bufs = [torch.zeros(e.consensus_size(), dtype=torch.float64, device="cuda") for e in engs]
        for e, b in zip(engs, bufs):
            e.bind_consensus(b.data_ptr())
        for it in range(6):
            for e in engs:
                e.local_update()
            for e in engs:
                e.sync()
            total = sum(bufs[1:], bufs[0].clone())       # what the all-reduce would do
            for b in bufs:
                b.copy_(total)
            torch.cuda.synchronize()
            for e in engs:
                e.apply_consensus()
            for e in engs:
                e.sync()
        for e in engs:
            e.get_primal()
            e.close()
    e = make_engine(hip, pp, **kw)
    e.iterate(20)
    e.close()
print("guard worker: ok")
