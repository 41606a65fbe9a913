"""a slice of tools/fuzz_parity.py in the GPU suite: random geometries, flavours, strides, pushes and dirty reads, engine vs
oracle bit for bit (the tool itself has been run over thousands of cases per seed; see its docstring)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,sparse,front", [(11, "0", None), (12, "1", None), (13, "0", "6"), (14, "0", "11"), (15, "1", "5"), (16, "1", "0")])
def test_randomized_engine_vs_oracle(seed, sparse, front):
    """front: MK_OPT_FRONT_BITS -- a front table of 64 slots overflows into the big table in almost every case and closes
    after the first launch, one of 2048 slots holds most small sketches whole"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--cases", "250", "--seed", str(seed),
                        "--sparse", sparse] + (["--front-bits", front] if front else []), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "250 cases, 0 mismatches" in out, out[-1500:]


def test_randomized_setop_vs_numpy_and_oracle():
    """mk_setop_* (set -u/-q/-i/-s/-g, composite join) on random id lists around the chunk boundaries, with duplicates, zeros and
    the top of the 32-bit range: numpy's definition of the sets, the oracle's per-taxon table (also over-full ones)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_setop.py"), "--cases", "500", "--seed", "41"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "500 cases, 0 mismatches" in out, out[-1500:]
