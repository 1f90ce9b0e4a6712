import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import dopf_pkg  # noqa: E402

pkg = dopf_pkg.load()
from decentralopf_jl_amd import _capi  # noqa: E402

ORACLE_LIB = os.path.join(ROOT, "oracle", "libdopf_oracle.so")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def build_oracle():
    src = os.path.join(ROOT, "oracle", "dopf_oracle.c")
    if not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-B"], check=True, stdout=subprocess.DEVNULL)
    return ORACLE_LIB


@pytest.fixture(scope="session")
def oracle_api():
    """The CPU oracle — the checker. Never part of the product path."""
    from oracle.binding import OracleApi
    return OracleApi(build_oracle())


@pytest.fixture(scope="session")
def hip_api():
    """The product library on a real GPU; fails loudly if it is missing (no fallback)."""
    return _capi.hip_api()


@pytest.fixture(scope="session")
def three_node():
    nodes, lines, gens, stos = pkg.three_node_case()
    return nodes, lines, gens, stos, pkg.pack(nodes, gens, stos, lines)


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return {n: load_golden(n) for n in ("TNS", "big_gamma", "wrong_weight")}


@pytest.fixture(scope="session")
def thesis():
    return load_golden("thesis_tables")
