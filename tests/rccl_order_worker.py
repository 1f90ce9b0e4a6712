"""Child process of tests/test_capi_symbols.py: the library has to load RCCL BEFORE PyTorch is imported, then PyTorch arrives.
The process must end with ONE librccl and ONE HIP runtime mapped (the copies PyTorch bundles), whatever the calls returned
(there may be no GPU here), and exit cleanly."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C

import dopf_pkg

dopf_pkg.load()
from decentralopf_jl_amd import _capi

assert "torch" not in sys.modules
api = _capi.hip_api()                       # pins the HIP runtime PyTorch bundles (when PyTorch is installed), loads libdopf_hip.so
buf = (C.c_char * _capi.COMM_ID_BYTES)()
rc = api.comm_unique_id(buf)                # loads RCCL: the copy next to the mapped runtime
print("comm_unique_id rc", rc, flush=True)
import torch  # noqa: E402,F401

torch.zeros(3).sum().item()
rc2 = api.comm_unique_id(buf)               # and again with PyTorch in the process: still one copy
print("comm_unique_id rc", rc2, flush=True)


def mapped(prefix):
    out = set()
    with open("/proc/self/maps") as f:
        for line in f:
            p = line.split()[-1]
            if os.path.basename(p).startswith(prefix):
                out.add(os.path.realpath(p))
    return sorted(out)


r, h = mapped("librccl.so"), mapped("libamdhip64.so")
print("rccl", r, "hip", h, flush=True)
assert len(r) <= 1 and len(h) == 1, (r, h)
assert rc2 != -4                             # DOPF_E_UNSUPPORTED would mean "two copies of RCCL mapped"
print("rccl order worker: ok", flush=True)
