"""One rank of a multi-process HIP run with the in-library communicator (dopf_comm_init): started by
tests/test_gpu_multi.py, one process per GPU. argv: rank world id_file out_file n_iters"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dopf_pkg  # noqa: E402

dopf_pkg.load()
from decentralopf_jl_amd import _capi, synth  # noqa: E402
from helpers import make_engine, state_of  # noqa: E402

rank, world, id_file, out_file, n_iters = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
pp = synth.synthetic_case(600, 80, 24, N=3, L=3, seed=11, fmax_factor=0.8, fmax_min=5)
A = pp.G + pp.S
e = make_engine(_capi.hip_api(), pp.shard(rank, world), eps=0.0, gamma=0.01, n_agents_global=A, device=rank)
if rank == 0:
    uid = e.comm_unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.05)
    uid = open(id_file, "rb").read()
e.comm_init(world, rank, uid)
e.iterate(n_iters)
st = state_of(e)
np.savez(out_file, comm=np.asarray(e.comm_info(), dtype=np.int64), **st)
