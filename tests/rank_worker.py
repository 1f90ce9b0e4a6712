"""One rank of a multi-process HIP run with the in-library communicator (dopf_comm_init): started by
tests/test_gpu_multi.py, one process per GPU (or per rank on one GPU for the peer exchange). argv: rank world id_file out_file
n_iters [rccl|xchg] [extra DOPF_F_* flags]"""
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
transport = sys.argv[6] if len(sys.argv) > 6 else "rccl"
xflags = int(sys.argv[7]) if len(sys.argv) > 7 else 0
import torch  # noqa: E402,F401  (before the library loads RCCL)

pp = synth.synthetic_case(600, 80, 24, N=3, L=3, seed=11, fmax_factor=0.8, fmax_min=5)
A = pp.G + pp.S
e = make_engine(_capi.hip_api(), pp.shard(rank, world), eps=0.0, gamma=0.01, n_agents_global=A, flags=xflags,
                device=rank % torch.cuda.device_count())


def publish(path, blob):
    with open(path + ".tmp", "wb") as f:
        f.write(blob)
    os.replace(path + ".tmp", path)


def fetch(path):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise SystemExit("nothing at " + path)
        time.sleep(0.05)
    return open(path, "rb").read()


if transport == "xchg":
    # peer exchange: every rank exports the handle of its receive area, all ranks read all handles
    publish(f"{id_file}.{rank}", e.xchg_export(world))
    e.xchg_init(world, rank, [fetch(f"{id_file}.{r}") for r in range(world)])
else:
    if rank == 0:
        publish(id_file, e.comm_unique_id())
    e.comm_init(world, rank, fetch(id_file))
done = 0
while done < n_iters:                       # several calls: graphs of 16, 4 and 1 iterations all get used
    k = min(n_iters - done, 21)
    e.iterate(k)
    done += k
st = state_of(e)
np.savez(out_file, comm=np.asarray(e.comm_info(), dtype=np.int64), **st)
