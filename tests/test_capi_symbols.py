"""The C-ABI library loads and exports every symbol include/dopf.h declares. No compute calls:
this runs on the CPU-only build container (hipcc cross-compiles, libamdhip64 loads without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from decentralopf_jl_amd import _capi


def _declared():
    text = open(os.path.join(ROOT, "include", "dopf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dopf_[a-z_]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for must in ("dopf_create", "dopf_destroy", "dopf_iterate", "dopf_local_update", "dopf_apply_consensus",
                 "dopf_get_duals", "dopf_get_primal", "dopf_get_consensus", "dopf_get_residuals", "dopf_set_state",
                 "dopf_last_error", "dopf_get_nodal_price", "dopf_bind_consensus"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_capi.HIP_LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_capi.HIP_LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.dopf_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.dopf_version()


def test_struct_layouts_match_the_header():
    # dopf_problem: 5 int32 (+4 pad) + 10 pointers; dopf_params: 5 doubles + 4 int32 + pointer
    assert ctypes.sizeof(_capi.DopfProblem) == 24 + 10 * 8
    assert ctypes.sizeof(_capi.DopfParams) == 5 * 8 + 4 * 4 + 8
    q = _capi.DopfParams()
    ctypes.CDLL(_capi.HIP_LIB_PATH).dopf_default_params(ctypes.byref(q))
    assert (q.gamma, q.w_flow, q.w_prox, q.eps, q.mask_thr, q.device) == (0.3, 10.0, 1.0, 1e-3, 1e-2, -1)
    # dopf_timing and dopf_central_result: the ctypes mirrors name the header's fields, in the header's order
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dopf.h")).read(), flags=re.S)
    for cname, mirror in (("dopf_timing", _capi.DopfTiming), ("dopf_central_result", _capi.DopfCentralResult)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if decl:
                ctype, names = decl.split(" ", 1)
                fields += [(n.strip(), {"double": ctypes.c_double, "int32_t": ctypes.c_int32}[ctype]) for n in names.split(",")]
        assert [(n, t) for n, t in mirror._fields_] == fields, cname


def test_no_cpu_fallback_in_product_package():
    """The product package never loads, links or names the oracle library."""
    pkg_dir = os.path.join(ROOT, "decentralopf.jl_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".jl")):
                text = open(os.path.join(dirpath, f)).read()
                assert "libdopf_oracle" not in text and "dopf_oracle" not in text, f
                assert "oracle_create" not in text and "ORACLE_LIB" not in text, f


def _build_c_example(tmp_path, name="three_node"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "decentralopf.jl_amd", "csrc")
    exe = str(tmp_path / name)
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", name + ".c"),
                    "-o", exe, "-L" + lib_dir, "-ldopf_hip", "-Wl,-rpath," + lib_dir], check=True)
    return exe


def test_c_example_builds_against_the_abi_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/three_node.c: plain C against include/dopf.h + libdopf_hip.so. Here (no GPU) it must refuse to run."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    r = subprocess.run([_build_c_example(tmp_path, "three_node_multi")], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_example_reproduces_the_reference_run(tmp_path):
    """The shipped three-node case from C: 476 iterations, cost 14034.51 (thesis Table 16), prices of Table 17."""
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "converged after 476 iterations, total cost 14034.51" in r.stdout
    assert "t=1 nodal prices -36.597 -15.216 -30.000" in r.stdout           # thesis Table 17 (tests/golden/thesis_tables.json)
    assert "t=2 nodal prices -81.976 -4.013 -30.000" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 2, 3])
def test_c_example_sharded_run_equals_the_reference_run(tmp_path, shards):
    """examples/three_node_multi.c: dopf_multi_* with the peer exchange from plain C (all shards on device 0 here)."""
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path, "three_node_multi"), str(shards)], capture_output=True, text=True,
                       env=dict(os.environ, DOPF_XCHG_TIMEOUT_MS="5000"))
    assert r.returncode == 0, r.stderr
    assert "converged after 476 iterations, total cost 14034.51 (%d shards)" % shards in r.stdout
    assert "t=1 nodal prices -36.597 -15.216 -30.000" in r.stdout
    assert "t=2 nodal prices -81.976 -4.013 -30.000" in r.stdout


def test_library_and_pytorch_share_one_rccl_and_one_hip_runtime_whatever_the_order():
    """Round 2's abort inside dopf_create (gpurun_out/r2_bis*.log): the library had dlopen'ed the SYSTEM's RCCL because
    PyTorch's copy was not mapped yet; PyTorch's arrived later. Now the library loads the RCCL that sits next to the HIP
    runtime the process runs on, the Python loader pins PyTorch's runtime before libdopf_hip.so when PyTorch is installed,
    and a second copy is refused (DOPF_E_UNSUPPORTED) instead of used. Child process: load the library, make it load
    RCCL, THEN import torch — one librccl, one libamdhip64 mapped, clean exit."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_order_worker.py")
    r = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl order worker: ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])


def test_a_second_rccl_in_the_process_is_refused_not_used():
    """Force the hazard: the system's librccl is loaded by hand next to the one the library picked. The next call that
    needs RCCL returns DOPF_E_UNSUPPORTED and names both copies."""
    import subprocess
    import sys
    sys_rccl = "/opt/rocm/lib/librccl.so.1"
    if not os.path.exists(sys_rccl):
        pytest.skip("no system RCCL to collide with")
    code = (
        "import os, sys, ctypes as C\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import dopf_pkg; dopf_pkg.load()\n"
        "from decentralopf_jl_amd import _capi\n"
        "api = _capi.hip_api()\n"
        "buf = (C.c_char * _capi.COMM_ID_BYTES)()\n"
        "assert api.comm_unique_id(buf) == 0\n"
        "maps = open('/proc/self/maps').read()\n"
        f"if {sys_rccl!r} in maps or os.path.realpath({sys_rccl!r}) in maps: print('system copy in use: nothing to collide'); sys.exit(0)\n"
        f"C.CDLL({sys_rccl!r}, mode=C.RTLD_LOCAL)\n"
        "rc = api.comm_unique_id(buf)\n"
        "msg = api.last_error(None).decode()\n"
        "print(rc, msg)\n"
        "assert rc == -4 and 'two copies of RCCL' in msg, (rc, msg)\n"
        "print('refused')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ("refused" in r.stdout or "nothing to collide" in r.stdout), (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
