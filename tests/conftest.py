import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """the C-ABI library and the oracle must exist; build them if this checkout is fresh (CPU-only is fine)"""
    import subprocess
    lib = os.path.join(ROOT, "metakssd_amd", "lib", "libmetakssd_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "metakssd_amd", "csrc")])
    ora = os.path.join(ROOT, "oracle", "libkssd_oracle.so")
    if not os.path.exists(ora) or not os.path.exists(os.path.join(ROOT, "oracle", "kssd_oracle_cli")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


# name -> (k, subk, drlevel, seed); the same seeds as oracle/check_vs_ref.py and tests/golden/make_golden.py
SHUF_SPECS = {
    "L3K11": (11, 6, 3, 11),   # BASELINE config: 22-mers, 1/4096 accepted, table 33 554 393
    "L3K10": (10, 6, 3, 10),   # config 5: table 2 097 143
    "L2K11": (11, 5, 2, 211),  # config 5: 16 components, table 536 870 909
    "L3K9": (9, 6, 3, 9),      # table 131 071
    "L0K6": (6, 3, 0, 6),      # every k-mer accepted, table 131 071: collisions from tiny inputs
    "L1K7": (7, 4, 1, 7),      # 1/16 accepted, table 131 071
}

_shuf_cache = {}


@pytest.fixture(scope="session")
def shufs():
    from metakssd_amd import capi

    def get(name):
        if name not in _shuf_cache:
            if name == "L0K6z":  # L0K6 with inner substring 0 mapped to 0: poly-A/T reads give key 0
                import numpy as np
                base = get("L0K6")
                s = capi.Shuf.generate(6, 3, 0, 6)
                t = s.table
                j = int(np.nonzero(base.table == 0)[0][0])
                t[j], t[0] = t[0], 0
                _shuf_cache[name] = s
            else:
                k, subk, drl, seed = SHUF_SPECS[name]
                _shuf_cache[name] = capi.Shuf.generate(k, subk, drl, seed)
        return _shuf_cache[name]

    return get


@pytest.fixture(scope="session")
def oracle_for():
    from oracle_binding import Oracle

    def make(shuf):
        return Oracle(shuf.c.id, shuf.c.k, shuf.c.subk, shuf.c.drlevel, shuf.table)

    return make
